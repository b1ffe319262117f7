/* -*- c++ -*- -----------------------------------------------------------------------------------
   `fix nve/mdp`: velocity-Verlet NVE on the device, for runs whose pair style is one of this plugin's.

   The reference's styles run under the host's `fix nve` (in.rebomos-bulk:27): atom->f comes down and atom->x goes
   up across the link every step.  This fix does the two half-kicks and the drift on the device through the pair
   style's context (Pair::extract("mdp_ctx")), so that between two reneighborings nothing per atom crosses the link.
   A plugin registering a fix style is what the reference repository itself does (USER-BFIELD/bfieldplugin.cpp:15-29,
   creator.v2; virtuals USER-BFIELD/fix_bfield.h:33-38).

   fix ID all nve/mdp [hostcheck yes|no]      (default no: see fix_nve_mdp.cpp)
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

  mdp_ctx *ctx() const { return ctxp ? *ctxp : nullptr; }
  void to_host(bool forces);
};

}    // namespace LAMMPS_NS

#endif
#endif
