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
   On one MPI rank the pair style keeps the periodic images itself (mdp_set_box_host), so ghosts follow their owners
   without the host's forward_comm.

   `fix ID all nve/mdp bricks yes` on ONE rank runs the mode described next with a single brick (no communicator): the
   periodic images, the lists and every reneighboring are then the device's too (mdp_md_integrate_check, mdp_dd_reneighbor,
   mdp_md_compute) and the host's Neighbor idles for the length of the run -- 2.88 against 3.08 ms per step at 3.98 M atoms
   from rest, and for hot runs the difference between a reneighboring of 2 ms on the device and one on the host (sample.in's
   alloy at 1.0 M atoms: 1.12 ms per step).  Not the default on one rank: per-atom energy / virial and a non-periodic box
   need the mode above.

   Several ranks ("bricks").  There the host's Comm owns ghosts and migration, on host arrays -- which is what this fix
   takes out of the steps.  So it runs the library's own decomposition instead, as minihost/ddhost.cpp does without a
   LAMMPS around it:
     init()              a context of its own (same device as the pair style's), the style's parameters / tables on it
                         (Pair::extract); neighbor->delay beyond any run: the host neither checks nor reneighbors
     setup()             this rank's owned atoms (x, v, type, tag as the host holds them after ITS setup) -> the brick
                         (mdp_md_setup, mdp_dd_setup on comm->procgrid, rank = comm->me: LAMMPS' own bricks); RCCL's id from
                         rank 0 with MPI_Bcast(world); first halo and lists (mdp_dd_comm_reneighbor); forces of step 0
     initial_integrate   mdp_dd_comm_step_begin: half-kick + drift, the `check yes` decision from the word that rode in the
                         previous halo, migration + lists or the start of this step's halo, what needs no remote ghost
     Pair::compute       mdp_dd_comm_step_end (compute_bricks in the adapters): the rest of the step; eng_vdwl / virial of
                         this rank's atoms on steps that ask (the host sums over ranks as it always does)
     final_integrate     output steps and the last step: atoms come back -- as many as the brick owns NOW (atom->nlocal
                         and, through atom->avec->grow, the arrays follow), x, v, type, tag
     post_run            the pair style is released: the next run starts from the host's arrays like the first
   What the host still does every step is idle work on stale arrays (Comm::forward_comm / reverse_comm of its ghosts, a
   memset of f); arrays never shrink below nlocal + the stale ghosts, so that work stays inside them.  Mask, image flags
   and per-atom properties beyond the atomic style's are not carried -- group all, atom_style atomic.  Run in the mini-host
   on 2, 4 and 8 ranks (tests/test_gpu_minilmp_ranks.py: log.rebomos-bulk.4's rows, the one-rank thermo of hot runs with
   migration); against a real LAMMPS this mode is unverified.
-------------------------------------------------------------------------------------------------- */
#include "fix_nve_mdp.h"

#include "atom.h"
#include "atom_vec.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "neighbor.h"
#include "output.h"
#include "pair.h"
#include "update.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

using namespace LAMMPS_NS;
using namespace FixConst;

FixNVEMDP::FixNVEMDP(LAMMPS *lmp, int narg, char **arg)
    : Fix(lmp, narg, arg), ctxp(nullptr), pair_linked(nullptr), downloads(0), hostcheck(0), took_delay(0), saved_delay(0),
      linked_to(nullptr), bricks(0), bricks_kw(0), bctx(nullptr), bricks_slot(nullptr), bricks_ev(nullptr), style_id(0), comm_up(0), pending_final(0), step_ev(0)
{
  if (narg < 3 || (narg - 3) % 2) error->all(FLERR, "Illegal fix nve/mdp command");
  for (int k = 3; k + 1 < narg; k += 2) {
    const std::string key = arg[k], val = arg[k + 1];
    if ((key != "hostcheck" && key != "bricks") || (val != "yes" && val != "no")) error->all(FLERR, "Illegal fix nve/mdp command");
    if (key == "hostcheck") hostcheck = val == "yes";
    else bricks_kw = val == "yes";
  }
  time_integrate = 1;
  force_reneighbor = 1;
  next_reneighbor = -1;
}

FixNVEMDP::~FixNVEMDP()
{
  // LAMMPS::destroy() deletes Neighbor and Force (and with it the pair style) BEFORE Modify and its fixes, and a new
  // `pair_style` command replaces the pair object under a fix that lives on: what was extracted from the pair style is
  // touched only while force->pair is still the object it came from
  if (took_delay && neighbor) neighbor->delay = saved_delay;
  const bool pair_alive = force && force->pair && force->pair == linked_to;
  if (pair_alive && pair_linked) *pair_linked = 0;
  if (pair_alive && bricks_slot) *bricks_slot = nullptr;
  if (bctx) {
    if (comm_up) (void) mdp_dd_comm_destroy(bctx);
    mdp_destroy(bctx);
  } else if (pair_alive && ctx())
    (void) mdp_hnve_off(ctx());
}

// Neighbor::init() -- which runs behind Modify::init() -- insists that a non-zero delay be a multiple of `every`
int FixNVEMDP::taken_delay() const
{
  const int every = neighbor->every > 0 ? neighbor->every : 1;
  return kDelayTaken - kDelayTaken % every;
}

void FixNVEMDP::fail(mdp_ctx *c) { error->one(FLERR, std::string("Fix nve/mdp: ") + (c ? mdp_last_error(c) : "no device context")); }

int FixNVEMDP::setmask() { return INITIAL_INTEGRATE | FINAL_INTEGRATE; }

void FixNVEMDP::init()
{
  if (!force->pair) error->all(FLERR, "Fix nve/mdp requires a pair style");
  int dim = 0;
  linked_to = force->pair;
  ctxp = static_cast<mdp_ctx **>(force->pair->extract("mdp_ctx", dim));
  pair_linked = static_cast<int *>(force->pair->extract("mdp_nve_linked", dim));
  if (!ctxp || !pair_linked || !ctx())
    error->all(FLERR, "Fix nve/mdp requires a pair style of this plugin (rebomos or aeam)");
  if (comm->nprocs != 1 || bricks_kw) { // several ranks, or `bricks yes` on one: the library's decomposition runs the steps
    init_bricks();
    return;
  }
  if (mdp_hnve_setup(ctx(), update->dt, force->ftm2v, atom->mass, atom->ntypes) != MDP_OK)
    error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
  *pair_linked = 1; // from the next compute on (the setup compute uploads atoms AND velocities)
  next_reneighbor = -1;
  // `check yes`: the device's displacement check decides (see the head of this file); the host's own look at atom->x --
  // a pass over every owned atom per step, of positions that are not current on the host -- leaves the steps
  if (!hostcheck && neighbor->dist_check) {
    if (!took_delay) saved_delay = neighbor->delay;
    took_delay = 1;
    neighbor->delay = taken_delay();
  }
}

// ---- several ranks: the library's decomposition on a context of the fix's own (see the head of this file) ----------

void FixNVEMDP::init_bricks()
{
  int dim = 0;
  bricks_slot = static_cast<mdp_ctx **>(force->pair->extract("mdp_bricks_ctx", dim));
  bricks_ev = static_cast<int *>(force->pair->extract("mdp_bricks_ev", dim));
  const int *sid = static_cast<int *>(force->pair->extract("mdp_style", dim));
  if (!bricks_slot || !bricks_ev || !sid)
    error->all(FLERR, "Fix nve/mdp requires a pair style of this plugin (rebomos or aeam)");
  if (hostcheck) error->all(FLERR, "Fix nve/mdp: hostcheck yes needs the host's arrays current: one MPI rank without `bricks yes`");
  if (!domain->xperiodic || !domain->yperiodic || !domain->zperiodic)
    error->all(FLERR, "Fix nve/mdp on several MPI ranks (or with bricks yes) needs a periodic box");
  // the library numbers its bricks as MPI_Cart_create numbers LAMMPS' default grid (`processors * * * map cart`: the last
  // dimension fastest); another mapping (map xyz, numa, a custom file) would hand every rank another rank's brick
  if ((comm->myloc[0] * comm->procgrid[1] + comm->myloc[1]) * comm->procgrid[2] + comm->myloc[2] != comm->me)
    error->all(FLERR, "Fix nve/mdp on several MPI ranks needs the default mapping of ranks to the processor grid (processors ... map cart)");
  style_id = *sid;
  if (!bctx) {
    const int ndev = mdp_device_count();
    int id = ndev > 0 ? comm->me % ndev : 0; // (the pair style's rule, pair_rebomos.cpp open_device)
    if (const char *env = getenv("MDP_DEVICE")) id = atoi(env);
    if (mdp_create(&bctx, id) != MDP_OK) error->one(FLERR, "Fix nve/mdp: cannot create a device context");
  }
  if (style_id == 1) {
    const mdp_rebomos_params *P = static_cast<mdp_rebomos_params *>(force->pair->extract("mdp_rebomos_params", dim));
    if (!P) error->all(FLERR, "Fix nve/mdp: the pair style has no parameters yet (pair_coeff)");
    if (mdp_rebomos_set_params(bctx, P) != MDP_OK) fail(bctx);
  } else {
    const mdp_aeam_tables *T = static_cast<mdp_aeam_tables *>(force->pair->extract("mdp_aeam_tables", dim));
    if (!T) error->all(FLERR, "Fix nve/mdp: the pair style has no tables yet (pair_coeff)");
    if (mdp_aeam_set_tables(bctx, T) != MDP_OK) fail(bctx);
  }
  bricks = 1;
  *bricks_slot = nullptr; // (the setup compute of this run is the host's: its arrays are the current ones)
  // neither a check nor a reneighboring of the host's during the run: both read arrays that are not current
  if (!took_delay) saved_delay = neighbor->delay;
  took_delay = 1;
  neighbor->delay = taken_delay();
  next_reneighbor = -1;
}

// Modify::setup, behind the host's own setup (exchange, borders, lists, forces of step 0 in host mode)
void FixNVEMDP::setup(int /*vflag*/)
{
  if (!bricks) return;
  int dim = 0;
  const int n = atom->nlocal;
  const double skin = neighbor->skin, cutghost = force->pair->cutforce + skin;
  mdp_md_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.style = style_id;
  cfg.nlocal = n;
  cfg.ntypes = atom->ntypes;
  cfg.skin = skin;
  cfg.dt = update->dt;
  cfg.ftm2v = force->ftm2v;
  cfg.mvv2e = force->mvv2e;
  // provisional bounds (the library sets the brick's at every reneighboring): the box's Cartesian hull + the ghost shell
  const double *h = domain->h; // xprd, yprd, zprd, yz, xz, xy
  cfg.bbox_lo[0] = domain->boxlo[0] + fmin(0.0, h[5]) + fmin(0.0, h[4]) - cutghost - 2.0;
  cfg.bbox_hi[0] = domain->boxlo[0] + h[0] + fmax(0.0, h[5]) + fmax(0.0, h[4]) + cutghost + 2.0;
  cfg.bbox_lo[1] = domain->boxlo[1] + fmin(0.0, h[3]) - cutghost - 2.0;
  cfg.bbox_hi[1] = domain->boxlo[1] + h[1] + fmax(0.0, h[3]) + cutghost + 2.0;
  cfg.bbox_lo[2] = domain->boxlo[2] - cutghost - 2.0;
  cfg.bbox_hi[2] = domain->boxlo[2] + h[2] + cutghost + 2.0;
  const int *map = style_id == 1 ? static_cast<int *>(force->pair->extract("mdp_map", dim)) : nullptr;
  const int idummy = 0;
  const double ddummy[3] = {0, 0, 0};
  const double xdummy[3] = {0, 0, 0};
  if (mdp_md_setup(bctx, &cfg, n ? atom->x[0] : xdummy, n ? atom->v[0] : xdummy, atom->type, atom->tag, atom->mass, map, &idummy,
                   ddummy, &idummy, &idummy) != MDP_OK)
    fail(bctx);
  mdp_dd_config dd;
  memset(&dd, 0, sizeof dd);
  for (int d = 0; d < 3; d++) {
    dd.boxlo[d] = domain->boxlo[d];
    dd.procgrid[d] = comm->procgrid[d];
  }
  for (int k = 0; k < 6; k++) dd.h[k] = h[k];
  dd.rank = comm->me;
  dd.cutghost = cutghost;
  if (mdp_dd_setup(bctx, &dd) != MDP_OK) fail(bctx);
  if (comm->nprocs == 1) { // one brick: its periodic images are the library's, no communicator
    if (mdp_dd_reneighbor(bctx) != MDP_OK) fail(bctx);
    if (mdp_md_compute(bctx, 0, 0) != MDP_OK) fail(bctx);
    pending_final = 0;
    *bricks_slot = bctx;
    return;
  }
  if (!comm_up) { // RCCL's unique id: rank 0 makes it, MPI hands it round
    unsigned char uid[128];
    memset(uid, 0, sizeof uid);
    int ok = 1;
    if (comm->me == 0) ok = mdp_dd_comm_unique_id(uid) == MDP_OK;
    MPI_Bcast(uid, (int) sizeof uid, MPI_BYTE, 0, world);
    if (!ok) error->one(FLERR, "Fix nve/mdp: no RCCL library could be loaded");
    if (mdp_dd_comm_init(bctx, uid) != MDP_OK) fail(bctx);
    comm_up = 1;
  }
  if (mdp_dd_comm_reneighbor(bctx) != MDP_OK) fail(bctx);
  // forces of step 0 for the first half-kick (the host printed its own setup thermo from its host-mode compute)
  int rc;
  if (style_id == 2) {
    rc = mdp_md_aeam_density(bctx, 0);
    if (rc == MDP_OK) rc = mdp_dd_comm_forward_scalar(bctx);
    if (rc == MDP_OK) rc = mdp_md_aeam_force(bctx, 0, 0);
    if (rc == MDP_OK) rc = mdp_dd_comm_reverse(bctx);
  } else
    rc = mdp_md_compute(bctx, 0, 0);
  if (rc != MDP_OK) fail(bctx);
  pending_final = 0;
  *bricks_slot = bctx; // from now on Pair::compute ends the steps this fix opens
}

// the atoms the brick owns NOW, in the brick's order, into the host's arrays
void FixNVEMDP::bricks_to_host()
{
  long long di[8];
  if (mdp_dd_info(bctx, di, nullptr, nullptr) != MDP_OK) fail(bctx);
  const int n = (int) di[0];
  if (n + atom->nghost > atom->nmax) atom->avec->grow(n + atom->nghost); // (the host's idle passes over its stale ghosts stay inside)
  if (n) {
    if (mdp_md_download(bctx, atom->x[0], atom->v[0], nullptr, nullptr) != MDP_OK) fail(bctx);
    if (mdp_md_download_int(bctx, "tag", atom->tag) != MDP_OK) fail(bctx);
    if (mdp_md_download_int(bctx, "type", atom->type) != MDP_OK) fail(bctx);
  }
  atom->nlocal = n;
  downloads++;
}

void FixNVEMDP::reset_dt()
{
  if (bricks) {
    if (bricks_slot && *bricks_slot) error->all(FLERR, "Fix nve/mdp on several MPI ranks: the timestep cannot change during a run");
    return; // (setup() hands update->dt to the brick)
  }
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

void FixNVEMDP::initial_integrate(int vflag)
{
  if (bricks) {
    const bigint now = update->ntimestep;
    step_ev = (vflag || now == output->next || now == update->laststep) ? 1 : 0;
    *bricks_ev = step_ev | (comm->nprocs == 1 ? 2 : 0);
    if (comm->nprocs == 1) { // one brick: the deferred on-device `check yes` flag, the lists rebuilt on the device when it fired
      int moved = 0, dangerous = 0;
      if (mdp_md_integrate_check(bctx, pending_final, &moved, &dangerous) != MDP_OK) fail(bctx);
      if (moved) {
        if (mdp_dd_reneighbor(bctx) != MDP_OK) fail(bctx);
        one_rank_builds++;
      }
    } else {
      int ren = 0;
      if (mdp_dd_comm_step_begin(bctx, pending_final, -1, step_ev, step_ev, &ren) != MDP_OK) fail(bctx);
    }
    pending_final = step_ev ? 0 : 1; // (Pair::compute ends the step with the half-kick deferred on steps without output)
    return;
  }
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
  if (bricks) {
    long long info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (comm->nprocs > 1) (void) mdp_dd_comm_step_info(bctx, info);
    else info[3] = one_rank_builds;
    one_rank_builds = 0;
    if (comm->me == 0 && getenv("MDP_FIX_STATS"))
      printf("fix nve/mdp: %d bricks, %lld reneighborings on the device, %ld returns of the atoms to the host in this run\n", comm->nprocs,
             info[3], downloads);
    downloads = 0;
    *bricks_slot = nullptr; // the next run's setup is the host's again, from the arrays the last step brought back
    return;
  }
  if (comm->me == 0 && getenv("MDP_FIX_STATS"))
    printf("fix nve/mdp: %ld downloads of x and v in this run (reneighborings the device asked for, output steps, the last step)%s\n",
           downloads, took_delay ? "; check yes decided on the device" : "");
  downloads = 0;
}

void FixNVEMDP::final_integrate()
{
  if (bricks) { // (the half-kick is the library's: in mdp_dd_comm_step_end, or fused into the next step's begin)
    const bigint now = update->ntimestep;
    if (now == output->next || now == update->laststep) bricks_to_host();
    return;
  }
  if (mdp_hnve_final(ctx()) != MDP_OK) error->one(FLERR, std::string("Fix nve/mdp: ") + mdp_last_error(ctx()));
  const bigint now = update->ntimestep;
  if (now == output->next || now == update->laststep) to_host(false); // thermo, dumps, the state a run ends with
}
