/* -*- c++ -*- -----------------------------------------------------------------------------------
   `fix nve/mdp`: velocity-Verlet NVE on the device, for runs whose pair style is one of this plugin's.

   The reference's styles run under the host's `fix nve` (in.rebomos-bulk:27): atom->f comes down and atom->x goes
   up across the link every step.  This fix does the two half-kicks and the drift on the device through the pair
   style's context (Pair::extract("mdp_ctx")), so that between two reneighborings nothing per atom crosses the link.
   A plugin registering a fix style is what the reference repository itself does (USER-BFIELD/bfieldplugin.cpp:15-29,
   creator.v2; virtuals USER-BFIELD/fix_bfield.h:33-38).

   fix ID all nve/mdp [hostcheck yes|no] [bricks yes|no]     (defaults no, no: see fix_nve_mdp.cpp)

   On several MPI ranks the fix runs the library's own domain decomposition (csrc/domain.hip: one brick per rank on
   Comm's processor grid, halo / migration / `check yes` decision on the device, RCCL between the GPUs) on a context of
   its own, from the atoms each rank owns at setup; the host's Comm and Neighbor idle for the length of the run and get
   the atoms back -- wherever they migrated to -- on output steps and at the end (fix_nve_mdp.cpp, "bricks").
-------------------------------------------------------------------------------------------------- */
#ifdef FIX_CLASS
// clang-format off
FixStyle(nve/mdp,FixNVEMDP);
// clang-format on
#else

#ifndef MDP_FIX_NVE_MDP_H
#define MDP_FIX_NVE_MDP_H

#include "fix.h"

#include "mdpair_hip.h"

namespace LAMMPS_NS {

class FixNVEMDP : public Fix {
 public:
  FixNVEMDP(class LAMMPS *, int, char **);
  ~FixNVEMDP() override;
  int setmask() override;
  void init() override;
  void setup(int) override;
  void initial_integrate(int) override;
  void final_integrate() override;
  void post_run() override;
  void reset_dt() override;

 protected:
  mdp_ctx **ctxp;      // the pair style's device context (created in its init_style)
  int *pair_linked;    // the pair style's "positions and forces stay on the device" switch
  long downloads;      // steps on which the host's x / v were brought up to date (statistics)
  int hostcheck;       // `hostcheck yes`: Neighbor::decide() keeps looking at atom->x, which is downloaded for it
  int took_delay;      // init() raised neighbor->delay (`check yes`: the device's check decides) ...
  int saved_delay;     // ... from this value, which the destructor restores
  static constexpr int kDelayTaken = 1 << 30;

  class Pair *linked_to; // the pair style ctxp / pair_linked / bricks_slot point into (see the destructor)
  // several ranks ("bricks")
  int bricks;          // comm->nprocs > 1, or `bricks yes`: the steps run on bctx
  int bricks_kw;       // `bricks yes`
  long one_rank_builds = 0;
  mdp_ctx *bctx;       // the fix's own context: this rank's brick
  mdp_ctx **bricks_slot;   // the pair style's pointer to it (set while a run is under way: its compute() ends the steps there)
  int *bricks_ev;      // the pair style's copy of "this step was opened with energy / virial"
  int style_id, comm_up, pending_final, step_ev;

  mdp_ctx *ctx() const { return ctxp ? *ctxp : nullptr; }
  void to_host(bool forces);
  void init_bricks();
  void bricks_to_host();
  void fail(mdp_ctx *c);
  int taken_delay() const;
};

}    // namespace LAMMPS_NS

#endif
#endif
