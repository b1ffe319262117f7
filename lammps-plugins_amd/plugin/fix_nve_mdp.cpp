/* ------------------------------------------------------------------------------------------------
   fix nve/mdp -- see fix_nve_mdp.h.  What runs where:

     initial_integrate   v += dtf f/m, x += dt v, periodic images refreshed            device (mdp_hnve_initial)
     Pair::compute       no position upload, forces stay on the device               device
     final_integrate     v += dtf f/m                                                device (mdp_hnve_final)

   and the host's atom->x / atom->v are brought up to date only on steps where the host reads them:
     * a reneighboring: Neighbor::decide() is asked for one through force_reneighbor / next_reneighbor when the
       device's own displacement check (half the skin, read one step late, with a margin) fires.  That check IS
       `neigh_modify check yes` -- the same quantity Neighbor::check_distance() computes from atom->x, seen earlier (the
       margin) -- so under `check yes` (the default; in.rebomos-bulk runs with every 1 delay 0 check yes,
       log.rebomos-bulk.1:41, sample.in:18 with every 1 delay 1 check yes) the fix takes the host's own check out of the
       steps: init() raises neighbor->delay beyond any run and the destructor puts it back.  The inputs of the
       reference then run unchanged at the device's speed; the host's x would be stale for its check anyway.
       `fix ID all nve/mdp hostcheck yes` keeps the host's settings as they are instead and downloads x and v on every
       step on which decide() looks at atom->x (the behaviour of the first version: correct for every setting, one
       download per look).  Under `check no` decide() rebuilds by the calendar, and x / v come down for exactly those steps;
     * thermo / dump steps (output->next) and the last step of a run.
   One MPI rank only: the pair style then keeps the periodic images itself (mdp_set_box_host), so ghosts follow their
   owners without the host's forward_comm.
-------------------------------------------------------------------------------------------------- */
#include "fix_nve_mdp.h"

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "neighbor.h"
#include "output.h"
#include "pair.h"
#include "update.h"

#include <cstdio>
#include <cstdlib>
#include <string>

using namespace LAMMPS_NS;
using namespace FixConst;

FixNVEMDP::FixNVEMDP(LAMMPS *lmp, int narg, char **arg)
    : Fix(lmp, narg, arg), ctxp(nullptr), pair_linked(nullptr), downloads(0), hostcheck(0), took_delay(0), saved_delay(0)
{
  if (narg != 3 && narg != 5) error->all(FLERR, "Illegal fix nve/mdp command");
  if (narg == 5) {
    const std::string key = arg[3], val = arg[4];
    if (key != "hostcheck" || (val != "yes" && val != "no")) error->all(FLERR, "Illegal fix nve/mdp command");
    hostcheck = val == "yes";
  }
  time_integrate = 1;
  force_reneighbor = 1;
  next_reneighbor = -1;
}

FixNVEMDP::~FixNVEMDP()
{
  if (took_delay) neighbor->delay = saved_delay;
  if (pair_linked) *pair_linked = 0;
  if (ctx()) (void) mdp_hnve_off(ctx());
}

int FixNVEMDP::setmask() { return INITIAL_INTEGRATE | FINAL_INTEGRATE; }

void FixNVEMDP::init()
{
  if (!force->pair) error->all(FLERR, "Fix nve/mdp requires a pair style");
  int dim = 0;
  ctxp = static_cast<mdp_ctx **>(force->pair->extract("mdp_ctx", dim));
  pair_linked = static_cast<int *>(force->pair->extract("mdp_nve_linked", dim));
  if (!ctxp || !pair_linked || !ctx())
    error->all(FLERR, "Fix nve/mdp requires a pair style of this plugin (rebomos or aeam)");
  if (comm->nprocs != 1)
    error->all(FLERR, "Fix nve/mdp needs a single MPI rank: the pair style then keeps the periodic images itself");
  if (mdp_hnve_setup(ctx(), update->dt, force->ftm2v, atom->mass, atom->ntypes) != MDP_OK)
    error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
  *pair_linked = 1; // from the next compute on (the setup compute uploads atoms AND velocities)
  next_reneighbor = -1;
  // `check yes`: the device's displacement check decides (see the head of this file); the host's own look at atom->x --
  // a pass over every owned atom per step, of positions that are not current on the host -- leaves the steps
  if (!hostcheck && neighbor->dist_check) {
    if (!took_delay) saved_delay = neighbor->delay;
    took_delay = 1;
    neighbor->delay = kDelayTaken;
  }
}

void FixNVEMDP::reset_dt()
{
  if (ctx() && mdp_hnve_setup(ctx(), update->dt, force->ftm2v, atom->mass, atom->ntypes) != MDP_OK)
    error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
}

void FixNVEMDP::to_host(bool forces)
{
  const int n = atom->nlocal;
  if (mdp_hnve_download(ctx(), n ? atom->x[0] : nullptr, n ? atom->v[0] : nullptr, (forces && n) ? atom->f[0] : nullptr) != MDP_OK)
    error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
  downloads++;
}

void FixNVEMDP::initial_integrate(int /*vflag*/)
{
  int moved = 0, dangerous = 0;
  if (mdp_hnve_initial(ctx(), &moved, &dangerous) != MDP_OK)
    error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
  const bigint now = update->ntimestep;
  if (getenv("MDP_DEBUG")) fprintf(stderr, "[fix nve/mdp] step %ld moved %d dangerous %d next_reneighbor %ld ago %d\n", (long) now, moved, dangerous, (long) next_reneighbor, neighbor->ago);
  if (moved && next_reneighbor < now) next_reneighbor = now + 1; // (a request for THIS step stands: decide() has not seen it yet)
  // will Neighbor::decide() of this step reneighbor, or read atom->x to find out?  Then the host needs x (and, for the
  // exchange of atoms between the periodic faces, v) now.  (Not under `check yes` without `hostcheck yes`: delay was raised.)
  const int ago = neighbor->ago + 1;
  const bool host_looks = ago >= neighbor->delay && ago % neighbor->every == 0;
  if (next_reneighbor == now || host_looks) to_host(false);
}

// MDP_FIX_STATS=1: one line per run on how often the host's x / v were brought up to date (bench.py reads it)
void FixNVEMDP::post_run()
{
  if (comm->me == 0 && getenv("MDP_FIX_STATS"))
    printf("fix nve/mdp: %ld downloads of x and v in this run (reneighborings the device asked for, output steps, the last step)%s\n",
           downloads, took_delay ? "; check yes decided on the device" : "");
  downloads = 0;
}

void FixNVEMDP::final_integrate()
{
  if (mdp_hnve_final(ctx()) != MDP_OK) error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
  const bigint now = update->ntimestep;
  if (now == output->next || now == update->laststep) to_host(false); // thermo, dumps, the state a run ends with
}
