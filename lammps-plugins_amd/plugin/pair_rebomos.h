/* -*- c++ -*- -----------------------------------------------------------------------------------
   MI355X-native REBO Mo-S pair style: LAMMPS-facing adapter.

   Same class name, style name and virtual surface as the CPU plugin
   (lammps/lammps-plugins USER-REBOMOS/pair_rebomos.h:14-39) so that `pair_style rebomos` /
   `pair_coeff * * MoS.REBO.set5b Mo S` scripts run unchanged.  All arithmetic happens in
   libmdpair_hip.so (include/mdpair_hip.h); this class only moves the host's data across the C-ABI.
-------------------------------------------------------------------------------------------------- */
#ifdef PAIR_CLASS
// clang-format off
PairStyle(rebomos,PairREBOMoS);
// clang-format on
#else

#ifndef MDP_PAIR_REBOMOS_H
#define MDP_PAIR_REBOMOS_H

#include "pair.h"

#include "mdpair_hip.h"

namespace LAMMPS_NS {

class PairREBOMoS : public Pair {
 public:
  PairREBOMoS(class LAMMPS *);
  ~PairREBOMoS() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  void init_style() override;
  double init_one(int, int) override;
  double memory_usage() override;
  void *extract(const char *, int &) override;

 protected:
  mdp_ctx *dev;                 // device context (one GPU per rank)
  bool host_list = false;       // MDP_REBOMOS_HOST_LIST=1: lists from the rows LAMMPS built (mdp_rebomos_host_list)
  int nve_linked;               // set by fix nve/mdp: x, v and f of the owned atoms stay on the device between reneighborings
  mdp_ctx *bricks;              // set by fix nve/mdp on several ranks: its context holds this rank's brick, whole steps run there
  int bricks_ev;                // ... and whether it opened the current step with energy / virial
  int style_id;                 // MDP_STYLE_REBOMOS (what the fix sets its own context up with)
  mdp_rebomos_params params;    // the 61 file scalars after mixing
  bool params_read;
  double cut3rebo;              // 3 * rcmax_MM, the list cutoff the style asks the host for
  int nall_uploaded;            // atoms on the device match the host's (nlocal+nghost) of the last upload
  double device_bytes;

  void allocate();
  void open_device();
  void fail_one(int code, const char *what);
  void compute_bricks();
};

}    // namespace LAMMPS_NS

#endif
#endif
