// Loader of the MI355X-native `rebomos` pair style.  Exports the one C symbol LAMMPS' `plugin load`
// looks up (same contract as lammps/lammps-plugins USER-REBOMOS/rebomosplugin.cpp:14-28).
#include "lammpsplugin.h"
#include "version.h"

#include "fix_nve_mdp.h"
#include "pair_rebomos.h"

namespace {
void *make_pair_rebomos(void *lmp)
{
  return new LAMMPS_NS::PairREBOMoS(static_cast<LAMMPS_NS::LAMMPS *>(lmp));
}
void *make_fix_nve_mdp(void *lmp, int narg, char **arg)
{
  return new LAMMPS_NS::FixNVEMDP(static_cast<LAMMPS_NS::LAMMPS *>(lmp), narg, arg);
}
}    // namespace

extern "C" void lammpsplugin_init(void *lmp, void *handle, void *regfunc)
{
  // the struct may live on the stack: the host copies it during registration; the strings are literals
  lammpsplugin_t desc;
  desc.version = LAMMPS_VERSION;
  desc.style = "pair";
  desc.name = "rebomos";
  desc.info = "REBO Mo-S pair style, MI355X (gfx950) HIP kernels v1.0";
  desc.author = "lammps-plugins_amd";
  desc.creator.v1 = &make_pair_rebomos;
  desc.handle = handle;
  reinterpret_cast<lammpsplugin_regfunc>(regfunc)(&desc, lmp);

  // ... and the time-integration fix that keeps x, v and f on the device for this style (fix_nve_mdp.h); a second style
  // from one plugin file, registered like the first (fix styles take the three-argument factory, creator.v2)
  desc.style = "fix";
  desc.name = "nve/mdp";
  desc.info = "NVE integration on the device for the MI355X pair styles of this plugin v1.0";
  desc.creator.v2 = &make_fix_nve_mdp;
  reinterpret_cast<lammpsplugin_regfunc>(regfunc)(&desc, lmp);
}
