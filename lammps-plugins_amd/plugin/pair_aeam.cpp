/* ------------------------------------------------------------------------------------------------
   MI355X-native angular-EAM pair style: LAMMPS-facing adapter (host C++).

   Mirrors the host-visible behaviour of lammps/lammps-plugins USER-AEAM/pair_aeam.cpp:
   constructor flags (:36-59), settings (:513-517), coeff (:523-595), init_style (:601-609),
   init_one (:615-621), the comm callbacks (:946-990) and compute (:110-479) with their error
   messages.  Density, embedding and force passes run in HIP kernels behind include/mdpair_hip.h.

   compute() = density half on the device -> this style's own forward_comm of fp (as the reference
   does at :307, but here the exchange is REQUIRED: ghosts' fp feed the gather formulation) -> force
   half.  Forces on ghosts are produced only by angular (three-body) centres and are folded by the
   host's reverse_comm of f, as for the reference.
-------------------------------------------------------------------------------------------------- */
#include "pair_aeam.h"

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

PairAEAM::PairAEAM(LAMMPS *lmp) : Pair(lmp)
{
  // pair_aeam.cpp:38-58
  restartinfo = 0;
  manybody_flag = 1;
  unit_convert_flag = utils::get_supported_conversions(utils::ENERGY);
  comm_forward = 1;
  comm_reverse = 1;
  one_coeff = 1;
  no_virial_fdotr = 1;    // explicit pair virial from the device (gather formulation)

  nmax = 0;
  rho = fp = nullptr;
  cutforcesq = cutmax = 0.0;
  dev = nullptr;
  nve_linked = 0;
  bricks = nullptr;
  bricks_ev = 0;
  style_id = 2;
  potfile = nullptr;
  tables_built = false;
  nelements = 0;
  cut_el = nullptr;
  nall_uploaded = -1;
  device_lists = false;
  memset(&tables, 0, sizeof tables);
}

PairAEAM::~PairAEAM()
{
  memory->destroy(rho);
  memory->destroy(fp);
  if (dev) mdp_destroy(dev);
  if (potfile) mdp_aeam_file_free(potfile);
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
    delete[] map;
    map = nullptr;
  }
}

void PairAEAM::fail_one(int code, const char *what)
{
  std::string msg = std::string("Pair style aeam (MI355X): ") + what + " failed";
  if (dev) msg += std::string(": ") + mdp_last_error(dev);
  (void) code;
  error->one(FLERR, msg);
}

void PairAEAM::open_device()
{
  if (dev) return;
  const int ndev = mdp_device_count();
  if (ndev <= 0) error->all(FLERR, "Pair style aeam (MI355X) needs a HIP device; there is no CPU fallback");
  int id = comm->me % ndev;
  if (const char *env = getenv("MDP_DEVICE")) id = atoi(env);
  if (mdp_create(&dev, id) != MDP_OK) error->one(FLERR, "Pair style aeam (MI355X): cannot create a device context");
}

void PairAEAM::allocate()
{
  allocated = 1;
  const int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) setflag[i][j] = 0;
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  delete[] map;
  map = new int[n + 1];
  for (int i = 0; i <= n; i++) map[i] = -1;
}

void PairAEAM::settings(int narg, char ** /*arg*/)
{
  if (narg != 0) error->all(FLERR, "Illegal pair_style command");
}

void PairAEAM::coeff(int narg, char **arg)
{
  if (!allocated) allocate();
  const int n = atom->ntypes;

  if (narg != 3 + n) error->all(FLERR, "Incorrect args for pair coefficients");
  if (strcmp(arg[0], "*") != 0 || strcmp(arg[1], "*") != 0)
    error->all(FLERR, "Incorrect args for pair coefficients");

  // potential file (pair_aeam.cpp:536-551)
  if (potfile) {
    mdp_aeam_file_free(potfile);
    potfile = nullptr;
  }
  tables_built = false;
  char why[512] = "";
  const std::string path = utils::get_potential_file_path(arg[2]);
  if (mdp_aeam_file_read(path.empty() ? arg[2] : path.c_str(), &potfile, why, (int) sizeof why) != MDP_OK)
    error->one(FLERR, why[0] ? why : "Cannot open AEAM potential file");
  char names[MAXEL * 16 + 16] = "";
  int nnon = 0, nang = 0;
  mdp_aeam_file_info(potfile, &nelements, &nnon, &nang, element_mass, MAXEL, names, (int) sizeof names);
  {
    int k = 0;
    char *save = nullptr;
    for (char *tok = strtok_r(names, " ", &save); tok && k < MAXEL; tok = strtok_r(nullptr, " ", &save), k++) {
      strncpy(elements[k], tok, 15);
      elements[k][15] = 0;
    }
  }

  // map atom types to elements (pair_aeam.cpp:555-566) ...
  for (int i = 3; i < narg; i++) {
    if (strcmp(arg[i], "NULL") == 0) {
      map[i - 2] = -1;
      continue;
    }
    int j;
    for (j = 0; j < nelements; j++)
      if (strcmp(arg[i], elements[j]) == 0) break;
    if (j < nelements)
      map[i - 2] = j;
    else
      error->all(FLERR, "No matching element in AEAM potential file");
  }
  // ... and insist on the file's element order (pair_aeam.cpp:568-572)
  for (int i = 3; i < narg; i++)
    if (i - 3 >= nelements || strcmp(arg[i], elements[i - 3]) != 0)
      error->all(FLERR, "no matching atom order of input file and potential file");

  int count = 0;
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) {
      setflag[i][j] = 0;
      if (map[i] >= 0 && map[j] >= 0) {
        setflag[i][j] = 1;
        if (i == j) atom->set_mass(FLERR, i, element_mass[map[i]]);
        count++;
      }
    }
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

void PairAEAM::init_style()
{
  if (force->newton_pair == 0) error->all(FLERR, "Pair style aeam requires newton pair on");

  // file2array + array2spline (pair_aeam.cpp:605-606), then the tables go to the device
  if (!potfile) error->all(FLERR, "All pair coeffs are not set");
  if (mdp_aeam_file_build(potfile, atom->ntypes, map, &tables) != MDP_OK)
    error->all(FLERR, "Pair style aeam (MI355X): building the spline tables failed");
  tables_built = true;
  cut_el = const_cast<double *>(tables.cut);
  open_device();
  if (mdp_aeam_set_tables(dev, &tables) != MDP_OK) fail_one(MDP_EINVAL, "table upload");
  // lists built on the device from the positions (the host's list is only checked); MDP_AEAM_HOST_LIST=1 streams
  // the host's list instead
  const char *ehl = getenv("MDP_AEAM_HOST_LIST");
  device_lists = !(ehl && atoi(ehl) != 0);
  if (mdp_aeam_device_lists(dev, device_lists ? 1 : 0) != MDP_OK) fail_one(MDP_EINVAL, "list mode");

  neighbor->add_request(this, NeighConst::REQ_FULL);
  nall_uploaded = -1;
}

double PairAEAM::init_one(int i, int j)
{
  if (setflag[i][j] == 0) error->all(FLERR, "All pair coeffs are not set");
  // per-type-pair cutoff straight from the file (pair_aeam.cpp:618-620)
  cutmax = tables_built ? tables.cut[(i - 1) * tables.nelements + (j - 1)] : 0.0;
  cutforcesq = cutmax * cutmax;
  return cutmax;
}

// fix nve/mdp on several ranks: the step was opened by its initial_integrate (mdp_dd_comm_step_begin on the fix's own
// context: integrate, reneighbor or start the halo, the density of the tiles that reach no remote ghost); this is the
// rest of the step -- fp out and the ghosts' three-body forces back between the bricks on the device, not through
// pack_forward_comm / Comm::reverse_comm.  The host's atom arrays are not read and not written.
void PairAEAM::compute_bricks()
{
  if (eflag_atom || vflag_atom)
    error->all(FLERR, "Pair style aeam (MI355X): per-atom energy / virial is not available while fix nve/mdp keeps the atoms on its bricks");
  const int want = (eflag_global || vflag_global) ? 1 : 0;
  if (want && !(bricks_ev & 1))
    error->all(FLERR, "Pair style aeam (MI355X): energy / virial asked for on a step fix nve/mdp opened without them");
  const int ev = (bricks_ev & 1) ? 1 : 0;
  int rc;
  if (bricks_ev & 2) { // one rank (`bricks yes`): no exchange to wait for -- compute, then the half-kick now or with the next step's
    rc = mdp_md_compute(bricks, ev, ev);
    if (rc == MDP_OK) rc = ev ? mdp_md_final_integrate(bricks) : mdp_md_defer_final(bricks);
  } else
    rc = mdp_dd_comm_step_end(bricks, ev, ev, ev ? 0 : 1);
  if (rc != MDP_OK) error->one(FLERR, std::string("Pair style aeam (MI355X): ") + mdp_last_error(bricks));
  if (want) {
    double t[9];
    if (mdp_md_thermo(bricks, t) != MDP_OK) error->one(FLERR, std::string("Pair style aeam (MI355X): ") + mdp_last_error(bricks));
    if (eflag_global) eng_vdwl = t[1];
    if (vflag_global)
      for (int k = 0; k < 6; k++) virial[k] = t[2 + k];
  }
}

void PairAEAM::compute(int eflag, int vflag)
{
  ev_init(eflag, vflag);
  if (bricks) {
    compute_bricks();
    return;
  }

  if (atom->nmax > nmax) {
    memory->destroy(rho);
    memory->destroy(fp);
    nmax = atom->nmax;
    memory->create(rho, nmax, "pair:rho");
    memory->create(fp, nmax, "pair:fp");
  }

  const int nlocal = atom->nlocal, nall = atom->nlocal + atom->nghost;
  const bool linked = nve_linked && comm->nprocs == 1;
  int rc;
  // the box of this step: on one periodic rank the library keeps the images itself (positions, fp, their share of the
  // three-body forces), as Comm::forward_comm / reverse_comm would
  rc = mdp_set_box_host(dev, comm->nprocs == 1 ? domain->h : nullptr);
  if (rc != MDP_OK) fail_one(rc, "box");
  if (neighbor->ago == 0 || nall != nall_uploaded) {
    rc = mdp_set_atoms_host(dev, nlocal, atom->nghost, nall ? atom->x[0] : nullptr, atom->type, atom->tag,
                            atom->ntypes, nullptr);
    if (rc != MDP_OK) fail_one(rc, "atom upload");
    if (device_lists) {
      // the device derives its lists from the positions (as the rebomos style does); the host's list is requested
      // for the ghost shell it implies and must be the plain geometric one -- checked, not read
      rc = mdp_set_skin(dev, neighbor->skin);
      if (rc != MDP_OK) fail_one(rc, "skin upload");
      rc = mdp_aeam_check_host_list(dev, list->inum, list->ilist, list->numneigh, list->firstneigh, neighbor->skin);
      if (rc != MDP_OK) fail_one(rc, "neighbor list check");
    } else {
      rc = mdp_set_neighbors_host(dev, list->inum, 0, list->ilist, list->numneigh, list->firstneigh, neighbor->skin);
      if (rc != MDP_OK) fail_one(rc, "neighbor list upload");
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

  // passes 1+2 on the device; fp (owned) comes back for the halo -- unless every ghost is an image the library keeps
  // itself: then rho and fp (private to this class, as in the reference) never leave the device
  const bool local_halo = mdp_host_ghosts_derived(dev) == 1;
  rc = mdp_aeam_density_host(dev, ef, local_halo ? nullptr : fp, local_halo ? nullptr : rho, &eng_vdwl, eatom);
  if (rc != MDP_OK) fail_one(rc, "density pass");

  // communicate the derivative of the embedding function (pair_aeam.cpp:307)
  if (!local_halo) comm->forward_comm(this);

  // (linked to fix nve/mdp: the forces' only reader is on the device too -- unless the host tallies or writes this step)
  const bool f_stays = linked && local_halo && !ef && !vf && update->ntimestep != output->next;
  rc = mdp_aeam_force_host(dev, ef, vf, local_halo ? nullptr : fp, (nall && !f_stays) ? atom->f[0] : nullptr, &eng_vdwl,
                           virial, eatom, (vflag_atom && vatom) ? vatom[0] : nullptr);
  if (rc != MDP_OK) fail_one(rc, "force pass");
}

void *PairAEAM::extract(const char *str, int &dim)
{
  // what fix nve/mdp needs of this style: its device context and the switch that keeps x, v and f there
  dim = 0;
  if (strcmp(str, "mdp_ctx") == 0) return (void *) &dev;
  if (strcmp(str, "mdp_nve_linked") == 0) return (void *) &nve_linked;
  // ... and on several ranks, where the fix runs the bricks on a context of its own: the style's tables for it
  if (strcmp(str, "mdp_bricks_ctx") == 0) return (void *) &bricks;
  if (strcmp(str, "mdp_bricks_ev") == 0) return (void *) &bricks_ev;
  if (strcmp(str, "mdp_style") == 0) return (void *) &style_id;
  if (strcmp(str, "mdp_aeam_tables") == 0) return tables_built ? (void *) &tables : nullptr;
  return nullptr;
}

/* ---- per-pair comm callbacks, same packing as pair_aeam.cpp:946-990 ---------------------------- */

int PairAEAM::pack_forward_comm(int n, int *list, double *buf, int /*pbc_flag*/, int * /*pbc*/)
{
  int m = 0;
  for (int i = 0; i < n; i++) buf[m++] = fp[list[i]];
  return m;
}

void PairAEAM::unpack_forward_comm(int n, int first, double *buf)
{
  int m = 0;
  const int last = first + n;
  for (int i = first; i < last; i++) fp[i] = buf[m++];
}

int PairAEAM::pack_reverse_comm(int n, int first, double *buf)
{
  // the device computes rho for owned atoms from the full list; ghosts carry nothing
  int m = 0;
  const int last = first + n;
  for (int i = first; i < last; i++) buf[m++] = 0.0;
  return m;
}

void PairAEAM::unpack_reverse_comm(int n, int *list, double *buf)
{
  int m = 0;
  for (int i = 0; i < n; i++) rho[list[i]] += buf[m++];
}

double PairAEAM::memory_usage()
{
  double bytes = (double) maxeatom * sizeof(double);
  bytes += (double) maxvatom * 6 * sizeof(double);
  bytes += 2.0 * nmax * sizeof(double);
  if (dev) bytes += mdp_device_bytes(dev); // repacked list, spline tables and work arrays held on the GPU
  return bytes;
}
