/* -*- c++ -*- -----------------------------------------------------------------------------------
   MI355X-native angular-EAM pair style: LAMMPS-facing adapter.

   Same class name, style name and virtual surface as the CPU plugin
   (lammps/lammps-plugins USER-AEAM/pair_aeam.h:14-41), including the four per-pair comm callbacks.
   All arithmetic happens in libmdpair_hip.so (include/mdpair_hip.h).
-------------------------------------------------------------------------------------------------- */
#ifdef PAIR_CLASS
// clang-format off
PairStyle(aeam,PairAEAM);
// clang-format on
#else

#ifndef MDP_PAIR_AEAM_H
#define MDP_PAIR_AEAM_H

#include "pair.h"

#include "mdpair_hip.h"

namespace LAMMPS_NS {

class PairAEAM : public Pair {
 public:
  PairAEAM(class LAMMPS *);
  ~PairAEAM() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  void init_style() override;
  double init_one(int, int) override;

  int pack_forward_comm(int, int *, double *, int, int *) override;
  void unpack_forward_comm(int, int, double *) override;
  int pack_reverse_comm(int, int, double *) override;
  void unpack_reverse_comm(int, int *, double *) override;
  double memory_usage() override;
  void *extract(const char *, int &) override;

 protected:
  int nve_linked;             // set by fix nve/mdp: x, v and f of the owned atoms stay on the device between reneighborings
  mdp_ctx *bricks;            // set by fix nve/mdp on several ranks: its context holds this rank's brick, whole steps run there
  int bricks_ev;              // ... and whether it opened the current step with energy / virial
  int style_id;               // MDP_STYLE_AEAM (what the fix sets its own context up with)
  int nmax;                   // allocated size of the per-atom host arrays
  double cutforcesq, cutmax;
  double *rho, *fp;           // host mirrors: rho (owned), fp = Fptmp*F' (owned, then ghosts via forward_comm)

  mdp_ctx *dev;
  mdp_aeam_file *potfile;     // parsed AlSi.aeam-style file (owns the table storage)
  mdp_aeam_tables tables;
  bool tables_built;
  int nelements;
  static constexpr int MAXEL = 64; // (a bound on what a file may declare; everything else is sized from the file)
  char elements[MAXEL][16];
  double element_mass[MAXEL];
  double *cut_el;             // [nelements*nelements], points into potfile
  int nall_uploaded;
  bool device_lists;          // lists built on the device from the positions; the host's list is checked, not read

  void allocate();
  void open_device();
  void fail_one(int code, const char *what);
  void compute_bricks();
};

}    // namespace LAMMPS_NS

#endif
#endif
