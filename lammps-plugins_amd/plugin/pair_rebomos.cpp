/* ------------------------------------------------------------------------------------------------
   MI355X-native REBO Mo-S pair style: LAMMPS-facing adapter (host C++).

   Mirrors the host-visible behaviour of lammps/lammps-plugins USER-REBOMOS/pair_rebomos.cpp:
   constructor flags (:57-72), settings (:144-147), coeff (:153-203), init_style (:209-238),
   init_one (:244-274), compute (:102-111) and the error messages of each.  The force/energy
   arithmetic itself (REBO_neigh, FREBO, bondorder, FLJ) runs in hand-written HIP kernels behind the
   C-ABI of include/mdpair_hip.h.

   Differences a host can observe, by design:
     * owner-computes: compute() adds complete forces to OWNED atoms and nothing to ghosts, so the
       host's reverse_comm of f carries zeros for this style;
     * the global virial is tallied explicitly on the device (no_virial_fdotr = 1), because
       x.f over ghosts is only valid for the scatter formulation;
     * per-atom virial (compute stress/atom) is complete on owned atoms, nothing on ghosts.
-------------------------------------------------------------------------------------------------- */
#include "pair_rebomos.h"

#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "output.h"
#include "update.h"
#include "utils.h"

#include <cstring>
#include <string>

using namespace LAMMPS_NS;

PairREBOMoS::PairREBOMoS(LAMMPS *lmp) : Pair(lmp)
{
  // pair_rebomos.cpp:59-64
  single_enable = 0;
  restartinfo = 0;
  one_coeff = 1;
  ghostneigh = 1;
  manybody_flag = 1;
  centroidstressflag = CENTROID_NOTAVAIL;
  // the device tallies the pair virial itself (see header comment)
  no_virial_fdotr = 1;

  dev = nullptr;
  nve_linked = 0;
  bricks = nullptr;
  bricks_ev = 0;
  style_id = 1;
  params_read = false;
  cut3rebo = 0.0;
  nall_uploaded = -1;
  device_bytes = 0.0;
  memset(&params, 0, sizeof params);
}

PairREBOMoS::~PairREBOMoS()
{
  if (dev) mdp_destroy(dev);
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
    memory->destroy(cutghost);
    delete[] map;
    map = nullptr;
  }
}

void PairREBOMoS::fail_one(int code, const char *what)
{
  std::string msg = std::string("Pair style rebomos (MI355X): ") + what + " failed";
  if (code == MDP_EOVERFLOW) msg = "Neighbor list overflow, boost neigh_modify one";    // pair_rebomos.cpp:350
  if (dev) msg += std::string(": ") + mdp_last_error(dev);
  error->one(FLERR, msg);
}

void PairREBOMoS::open_device()
{
  if (dev) return;
  const int ndev = mdp_device_count();
  if (ndev <= 0) error->all(FLERR, "Pair style rebomos (MI355X) needs a HIP device; there is no CPU fallback");
  int id = comm->me % ndev;
  if (const char *env = getenv("MDP_DEVICE")) id = atoi(env);
  const int rc = mdp_create(&dev, id);
  if (rc != MDP_OK) error->one(FLERR, "Pair style rebomos (MI355X): cannot create a device context");
  if (params_read && mdp_rebomos_set_params(dev, &params) != MDP_OK) fail_one(MDP_EINVAL, "parameter upload");
}

void PairREBOMoS::allocate()
{
  allocated = 1;
  const int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) setflag[i][j] = 0;
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  memory->create(cutghost, n + 1, n + 1, "pair:cutghost");
  delete[] map;
  map = new int[n + 1];
}

void PairREBOMoS::settings(int narg, char ** /*arg*/)
{
  if (narg != 0) error->all(FLERR, "Illegal pair_style command");
}

void PairREBOMoS::coeff(int narg, char **arg)
{
  if (!allocated) allocate();
  const int n = atom->ntypes;

  if (narg != 3 + n) error->all(FLERR, "Incorrect args for pair coefficients");
  if (strcmp(arg[0], "*") != 0 || strcmp(arg[1], "*") != 0)
    error->all(FLERR, "Incorrect args for pair coefficients");

  // atom type -> element: Mo (or legacy M) = 0, S = 1, NULL = -1   (pair_rebomos.cpp:168-179)
  map[0] = -1;
  for (int i = 3; i < narg; i++) {
    int el;
    if (strcmp(arg[i], "NULL") == 0)
      el = -1;
    else if (strcmp(arg[i], "Mo") == 0 || strcmp(arg[i], "M") == 0)
      el = 0;
    else if (strcmp(arg[i], "S") == 0)
      el = 1;
    else
      error->all(FLERR, "Incorrect args for pair coefficients");
    map[i - 2] = el;
  }

  // potential file: 61 scalars + mixing rules, shared front end in libmdpair_hip.so
  char why[512] = "";
  const std::string path = utils::get_potential_file_path(arg[2]);
  if (mdp_rebomos_read_file(path.empty() ? arg[2] : path.c_str(), &params, why, (int) sizeof why) != MDP_OK)
    error->one(FLERR, why[0] ? why : "reading rebomos potential file failed");
  params_read = true;
  if (dev && mdp_rebomos_set_params(dev, &params) != MDP_OK) fail_one(MDP_EINVAL, "parameter upload");

  int count = 0;
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) {
      setflag[i][j] = 0;
      if (map[i] >= 0 && map[j] >= 0) {
        setflag[i][j] = 1;
        count++;
      }
    }
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

void PairREBOMoS::init_style()
{
  if (atom->tag_enable == 0) error->all(FLERR, "Pair style REBOMoS requires atom IDs");
  if (force->newton_pair == 0) error->all(FLERR, "Pair style REBOMoS requires newton pair on");
  // atom types mapped to NULL (pair hybrid) are invisible to the device lists: map[] = -1 goes down as is

  // full neighbor list including neighbors of ghosts (pair_rebomos.cpp:218)
  neighbor->add_request(this, NeighConst::REQ_FULL | NeighConst::REQ_GHOST);

  open_device();
  nall_uploaded = -1;
  // MDP_REBOMOS_HOST_LIST=1: the lists are subsets of the rows LAMMPS built (exclusions and special bonds act as in the
  // reference) instead of being built from the positions -- see mdp_rebomos_host_list
  const char *ehl = getenv("MDP_REBOMOS_HOST_LIST");
  host_list = ehl && atoi(ehl) != 0;
  if (mdp_rebomos_host_list(dev, host_list ? 1 : 0) != MDP_OK) fail_one(MDP_EINVAL, "list mode");
}

double PairREBOMoS::init_one(int i, int j)
{
  if (setflag[i][j] == 0) error->all(FLERR, "All pair coeffs are not set");
  const int ii = map[i], jj = map[j];
  // list cutoff = 3 REBO distances of the largest element; ghost-list cutoff = REBO cutoff
  // (pair_rebomos.cpp:257-261)
  cut3rebo = 3.0 * params.rcmax[0][0];
  cutghost[i][j] = cutghost[j][i] = params.rcmax[ii][jj];
  return cut3rebo;
}

// fix nve/mdp on several ranks: the step was opened by its initial_integrate (mdp_dd_comm_step_begin on the fix's own
// context: integrate, reneighbor or start the halo, centres that need no remote ghost); this is the rest of the step.
// The host's atom arrays are not read and not written; energy and virial of this rank's atoms on the steps that ask.
void PairREBOMoS::compute_bricks()
{
  if (eflag_atom || vflag_atom)
    error->all(FLERR, "Pair style rebomos (MI355X): per-atom energy / virial is not available while fix nve/mdp keeps the atoms on its bricks");
  const int want = (eflag_global || vflag_global) ? 1 : 0;
  if (want && !(bricks_ev & 1))
    error->all(FLERR, "Pair style rebomos (MI355X): energy / virial asked for on a step fix nve/mdp opened without them");
  const int ev = (bricks_ev & 1) ? 1 : 0;
  int rc;
  if (bricks_ev & 2) { // one rank (`bricks yes`): no exchange to wait for -- compute, then the half-kick now or with the next step's
    rc = mdp_md_compute(bricks, ev, ev);
    if (rc == MDP_OK) rc = ev ? mdp_md_final_integrate(bricks) : mdp_md_defer_final(bricks);
  } else
    rc = mdp_dd_comm_step_end(bricks, ev, ev, ev ? 0 : 1);
  if (rc != MDP_OK) error->one(FLERR, std::string("Pair style rebomos (MI355X): ") + mdp_last_error(bricks));
  if (want) {
    double t[9];
    if (mdp_md_thermo(bricks, t) != MDP_OK) error->one(FLERR, std::string("Pair style rebomos (MI355X): ") + mdp_last_error(bricks));
    if (eflag_global) eng_vdwl = t[1];
    if (vflag_global)
      for (int k = 0; k < 6; k++) virial[k] = t[2 + k];
  }
}

void PairREBOMoS::compute(int eflag, int vflag)
{
  ev_init(eflag, vflag);
  if (bricks) {
    compute_bricks();
    return;
  }

  const int nlocal = atom->nlocal, nall = atom->nlocal + atom->nghost;
  const bool linked = nve_linked && comm->nprocs == 1;
  if (linked && host_list)
    error->all(FLERR, "Pair style rebomos (MI355X): fix nve/mdp keeps the atoms on the device and cannot be combined with MDP_REBOMOS_HOST_LIST=1");
  int rc;
  // the box of this step: on one periodic rank the library gives the images their positions itself, as
  // Comm::forward_comm does (owner + whole box vectors), and takes the owned atoms' positions only
  // (lists from the host's rows: the rows index the host's own ghosts, which then come up with the positions)
  rc = mdp_set_box_host(dev, comm->nprocs == 1 && !host_list ? domain->h : nullptr);
  if (rc != MDP_OK) fail_one(rc, "box");
  if (neighbor->ago == 0 || nall != nall_uploaded) {
    // the host rebuilt its list this step: atoms may have migrated / been re-sorted
    rc = mdp_set_atoms_host(dev, nlocal, atom->nghost, nall ? atom->x[0] : nullptr, atom->type, atom->tag,
                            atom->ntypes, map);
    if (rc != MDP_OK) fail_one(rc, "atom upload");
    // the device builds its own trimmed lists from the positions; the host's list (requested in
    // init_style for API parity and for the ghost shell it implies) only contributes its skin
    if (list->inum != nlocal) error->one(FLERR, "Pair style rebomos (MI355X): neighbor list does not match nlocal");
    if (host_list) {
      rc = mdp_set_neighbors_host(dev, list->inum, list->gnum, list->ilist, list->numneigh, list->firstneigh, neighbor->skin);
      if (rc != MDP_OK) fail_one(rc, "neighbor list upload");
    } else {
      rc = mdp_set_skin(dev, neighbor->skin);
      if (rc != MDP_OK) fail_one(rc, "skin upload");
      // ... which is only the reference's result when the host's list is the plain geometric one: the reference
      // walks the host's entries (pair_rebomos.cpp:328-330, 490-495), so exclusions or special bonds must stop the run
      rc = mdp_rebomos_check_host_list(dev, list->inum, list->ilist, list->numneigh, list->firstneigh,
                                       cut3rebo + neighbor->skin);
      if (rc != MDP_OK) fail_one(rc, "neighbor list check");
    }
    nall_uploaded = nall;
    // fix nve/mdp integrates on the device: the velocities go with the atoms (the host's are current on this step)
    if (linked) {
      rc = mdp_hnve_upload_v(dev, nlocal ? atom->v[0] : nullptr);
      if (rc != MDP_OK) fail_one(rc, "velocity upload");
    }
  } else if (!linked) {
    rc = mdp_set_positions_host(dev, nall ? atom->x[0] : nullptr);
    if (rc != MDP_OK) fail_one(rc, "position upload");
  } // (linked: the device moved the atoms itself, mdp_hnve_initial)

  const int ef = (eflag_global ? MDP_EFLAG_GLOBAL : 0) | (eflag_atom ? MDP_EFLAG_ATOM : 0);
  const int vf = (vflag_global ? MDP_VFLAG_GLOBAL : 0) | (vflag_atom ? MDP_VFLAG_ATOM : 0);
  // the forces' only reader is on the device too -- unless the host tallies or writes something this step
  const bool f_stays = linked && !ef && !vf && update->ntimestep != output->next;
  rc = mdp_rebomos_compute_host(dev, ef, vf, (nlocal && !f_stays) ? atom->f[0] : nullptr, &eng_vdwl, virial, eatom,
                                (vflag_atom && vatom) ? vatom[0] : nullptr);
  if (rc != MDP_OK) fail_one(rc, "compute");
}

void *PairREBOMoS::extract(const char *str, int &dim)
{
  // what fix nve/mdp needs of this style: its device context and the switch that keeps x, v and f there
  dim = 0;
  if (strcmp(str, "mdp_ctx") == 0) return (void *) &dev;
  if (strcmp(str, "mdp_nve_linked") == 0) return (void *) &nve_linked;
  // ... and on several ranks, where the fix runs the bricks on a context of its own: the style's parameters for it
  if (strcmp(str, "mdp_bricks_ctx") == 0) return (void *) &bricks;
  if (strcmp(str, "mdp_bricks_ev") == 0) return (void *) &bricks_ev;
  if (strcmp(str, "mdp_style") == 0) return (void *) &style_id;
  if (strcmp(str, "mdp_rebomos_params") == 0) return params_read ? (void *) &params : nullptr;
  if (strcmp(str, "mdp_map") == 0) return (void *) map;
  return nullptr;
}

double PairREBOMoS::memory_usage()
{
  // the reference reports the REBO lists it holds (pair_rebomos.cpp:1113-1124); here they live on the device:
  // candidate / tile lists, slot forces, staging of x and f -- plus the pinned host staging of one x and one f array
  device_bytes = dev ? mdp_device_bytes(dev) : 0.0;
  double bytes = device_bytes;
  bytes += (double) (atom->nlocal + atom->nghost) * (3 * sizeof(double) + sizeof(int));
  return bytes;
}
