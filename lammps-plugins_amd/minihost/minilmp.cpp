// minilmp -- a small MD host that implements the slice of LAMMPS the two pair-style
// plugins need: input-script subset, lattice/create_atoms, periodic ghost
// images (triclinic), binned full(+ghost) neighbor lists in LAMMPS' paged int** form, the Verlet
// loop of fix nve (and a simple Nose-Hoover fix nvt), thermo output in LAMMPS' log format, and
// `plugin load X.so` through dlopen -> lammpsplugin_init -> factory -> Pair*.
//
// It exists because neither this container nor the GPU box has a LAMMPS binary: it lets
// `in.rebomos-bulk` run against rebomosplugin.so exactly as `lmp -in in.rebomos-bulk` would
// (host mode of the C-ABI: x up / f down across PCIe every step).  It is NOT LAMMPS: a plugin
// built against these headers loads here only.
//
// `minilmp -np N`: N ranks as N threads of this process, each with its own host objects (as N MPI processes of LAMMPS
// have), a brick of the box in lamda space on LAMMPS' processor grid, ghosts from the neighbouring bricks, migration
// at reneighborings, Comm::forward_comm / reverse_comm of a pair style BETWEEN ranks through its pack / unpack
// callbacks, thermo sums over ranks.  `in.rebomos-bulk` on four ranks then runs as log.rebomos-bulk.4 shows it
// (2 by 2 by 1 grid, Nlocal 72, Nghost 2768 / 2775, the same thermo rows).  Every rank script-parses the whole input
// and creates ALL atoms (identical IDs on every rank and for every N); the bricks keep their own at the first `run`.
#include "lammps_host_api.h"

#include <dlfcn.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

using namespace LAMMPS_NS;

// =================================================================================================
// host API implementation
// =================================================================================================
struct HostAbort : std::runtime_error {
  using std::runtime_error::runtime_error;
};

static thread_local int t_rank = 0; // (`-np N`: the rank this thread is)

void Error::all(const std::string &file, int line, const std::string &msg)
{
  std::ostringstream o;
  if (t_rank == 0) o << "ERROR: " << msg << " (" << file << ":" << line << ")"; // (every rank stops; rank 0 says why)
  throw HostAbort(o.str());
}
void Error::one(const std::string &file, int line, const std::string &msg)
{
  std::ostringstream o;
  o << "ERROR on proc " << t_rank << ": " << msg << " (" << file << ":" << line << ")";
  throw HostAbort(o.str());
}
void Error::warning(const std::string &file, int line, const std::string &msg)
{
  fprintf(stderr, "WARNING: %s (%s:%d)\n", msg.c_str(), file.c_str(), line);
}

int utils::get_supported_conversions(const int property) { return property == ENERGY ? (METAL2REAL | REAL2METAL) : NOCONVERT; }

std::string utils::get_potential_file_path(const std::string &name)
{
  struct stat st;
  if (stat(name.c_str(), &st) == 0) return name;
  if (const char *dirs = getenv("LAMMPS_POTENTIALS")) {
    std::stringstream ss(dirs);
    std::string d;
    while (std::getline(ss, d, ':')) {
      const std::string p = d + "/" + name;
      if (stat(p.c_str(), &st) == 0) return p;
    }
  }
  return "";
}

void Atom::set_mass(const char *, int, int itype, double value)
{
  if (itype >= 1 && itype <= ntypes) mass[itype] = value;
}

Pair::~Pair() {}

void Pair::ev_setup(int eflag, int vflag)
{
  evflag = 1;
  eflag_either = eflag;
  eflag_global = eflag & 1;
  eflag_atom = eflag & 2;
  vflag_either = vflag;
  vflag_global = vflag & 3;
  vflag_atom = vflag & 4;
  const int nall = atom->nlocal + atom->nghost;
  if (eflag_atom && nall > maxeatom) {
    maxeatom = atom->nmax;
    memory->destroy(eatom);
    memory->create(eatom, maxeatom, "pair:eatom");
  }
  if (vflag_atom && nall > maxvatom) {
    maxvatom = atom->nmax;
    memory->destroy(vatom);
    memory->create(vatom, maxvatom, 6, "pair:vatom");
  }
  if (eflag_global) eng_vdwl = eng_coul = 0.0;
  if (vflag_global)
    for (int i = 0; i < 6; i++) virial[i] = 0.0;
  if (eflag_atom)
    for (int i = 0; i < nall; i++) eatom[i] = 0.0;
  if (vflag_atom)
    for (int i = 0; i < nall; i++)
      for (int k = 0; k < 6; k++) vatom[i][k] = 0.0;
  if (vflag_global == 2 && no_virial_fdotr == 0) {
    vflag_fdotr = 1;
    vflag_global = 0;
    if (vflag_atom == 0) vflag_either = 0;
    if (vflag_either == 0 && eflag_either == 0) evflag = 0;
  } else
    vflag_fdotr = 0;
}

void Pair::virial_fdotr_compute()
{
  double **x = atom->x, **f = atom->f;
  const int nall = atom->nlocal + atom->nghost;
  for (int i = 0; i < nall; i++) {
    virial[0] += x[i][0] * f[i][0];
    virial[1] += x[i][1] * f[i][1];
    virial[2] += x[i][2] * f[i][2];
    virial[3] += x[i][0] * f[i][1];
    virial[4] += x[i][0] * f[i][2];
    virial[5] += x[i][1] * f[i][2];
  }
}

void Pair::init()
{
  if (!allocated) error->all(FLERR, "All pair coeffs are not set");
  init_style();
  cutforce = 0.0;
  const int n = atom->ntypes;
  for (int i = 1; i <= n; i++)
    for (int j = i; j <= n; j++) {
      const double cut = init_one(i, j);
      cutsq[i][j] = cutsq[j][i] = cut * cut;
      cutforce = MAX(cutforce, cut);
    }
}

// =================================================================================================
// the host proper
// =================================================================================================
namespace {

const double BOLTZ = 8.617343e-5, MVV2E = 1.0364269e-4, FTM2V = 1.0 / 1.0364269e-4, NKTV2P = 1.6021765e6;

struct Vec3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};

struct Region {
  bool prism = false;
  double lo[3], hi[3], tilt[3]; // in lattice units as given
};

struct Host;

// ranks other than 0 keep quiet on stdout
thread_local bool t_mute = false;
#define printf(...) (t_mute ? 0 : std::printf(__VA_ARGS__))

// the ranks of `-np N` (threads) meet here: a barrier that a failed rank breaks, scratch for sums over ranks
struct World {
  int n = 1;
  std::vector<Host *> host;
  std::vector<double> red;
  std::vector<char> bbuf; // MPI_Bcast
  static constexpr int kRed = 16;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  long generation = 0;
  bool dead = false;
  explicit World(int n_) : n(n_), host(n_, nullptr), red((size_t) n_ * kRed, 0.0) {}
  void barrier()
  {
    if (n == 1) return;
    std::unique_lock<std::mutex> lk(m);
    if (dead) throw HostAbort("");
    const long g = generation;
    if (++waiting == n) {
      waiting = 0;
      generation++;
      cv.notify_all();
      return;
    }
    cv.wait(lk, [&] { return generation != g || dead; });
    if (generation == g) throw HostAbort(""); // (another rank failed and said why)
  }
  void kill()
  {
    std::lock_guard<std::mutex> lk(m);
    dead = true;
    cv.notify_all();
  }
};

struct GhostRec {
  int src, idx; // owner rank and its index there
  Vec3 shift;
};
struct MoveRec {
  double x[3], v[3];
  int type, tag;
};

struct HostAtomVec : AtomVec {
  Host *h = nullptr;
  void grow(int n) override;
};

thread_local World *t_world = nullptr; // (the world of this rank thread, for MPI_Bcast)

struct PeriodicComm : Comm {
  Host *h = nullptr;
  void forward_comm(Pair *pair) override;
  void reverse_comm(Pair *pair) override;
};

struct Host {
  LAMMPS lmp;
  Memory memory;
  Error error;
  Atom atom;
  Force force;
  Neighbor neighbor;
  NeighList list;
  PeriodicComm comm;
  HostAtomVec avec;
  Domain domain;
  Update update;
  Output output;
  Pair *pair = nullptr;
  Fix *fix = nullptr; // a time-integration fix style from a plugin (fix nve/mdp); built-in nve / nvt: fix_style

  // registry of plugin styles
  std::map<std::string, lammpsplugin_factory1 *> pair_styles;
  std::map<std::string, lammpsplugin_factory2 *> fix_styles;
  std::vector<void *> handles;

  // box: lo, prd, tilt (xy,xz,yz)
  double boxlo[3] = {0, 0, 0}, prd[3] = {1, 1, 1}, tilt[3] = {0, 0, 0};
  bool box_exists = false;
  // lattice
  std::string lat_style = "none";
  double lat_scale = 1.0, a1[3] = {1, 0, 0}, a2[3] = {0, 1, 0}, a3[3] = {0, 0, 1}, lat_origin[3] = {0, 0, 0};
  std::vector<Vec3> basis;
  double latsp[3] = {1, 1, 1};
  std::map<std::string, Region> regions;

  // atoms (owned + ghost), contiguous storage behind the double** views
  std::vector<double> xs, vs, fs;
  std::vector<double *> xrow, vrow, frow;
  std::vector<int> types, tags;
  std::vector<double> masses;
  std::vector<int> ghost_owner;
  std::vector<Vec3> ghost_shift;
  std::vector<double> xhold;

  // `-np N`: this rank's brick, its ghosts from other ranks, what it sends them
  World *world = nullptr;
  int me = 0, np = 1;
  bool decomposed = false;              // the bricks kept their own atoms (first `run`)
  double slo[3] = {0, 0, 0}, shi[3] = {1, 1, 1};
  std::vector<GhostRec> ghosts;         // ordered by owner rank
  std::vector<int> from_first, from_count;       // my ghosts owned by rank q: [from_first[q], + from_count[q])
  std::vector<std::vector<int>> sendlist;        // my atoms that are ghosts on rank r, in r's order
  std::vector<std::vector<MoveRec>> outbox;      // atoms leaving for rank r
  std::vector<std::vector<double>> cbuf;         // pair-style comm buffers, one per peer
  long natoms_all = 0;

  // neighbor storage
  std::vector<int> nb_store, ilist_v, numneigh_v;
  std::vector<std::pair<int, int>> excl_types; // neigh_modify exclude type M N
  std::vector<int *> firstneigh_v;
  double skin = 2.0;
  int nbuilds = 0;

  // run settings
  double dt = 0.001;
  int thermo_every = 0;
  std::vector<std::string> thermo_cols = {"step", "temp", "epair", "emol", "etotal", "press"};
  std::string fix_style = "";
  double nvt_t0 = 0, nvt_t1 = 0, nvt_damp = 0.1, nvt_eta_dot = 0.0;
  long step = 0;
  bool quiet = false;

  bool multi() const { return np > 1 && decomposed; }

  // sums over ranks, in rank order on every rank (so all ranks hold the same bits)
  void sum(double *v, int k)
  {
    if (!multi()) return;
    for (int i = 0; i < k; i++) world->red[(size_t) me * World::kRed + i] = v[i];
    world->barrier();
    for (int i = 0; i < k; i++) {
      double t = 0.0;
      for (int q = 0; q < np; q++) t += world->red[(size_t) q * World::kRed + i];
      v[i] = t;
    }
    world->barrier();
  }
  bool any(bool flag)
  {
    double v = flag ? 1.0 : 0.0;
    sum(&v, 1);
    return v != 0.0;
  }
  std::vector<double> gather(double v)
  {
    std::vector<double> out(np, v);
    if (!multi()) return out;
    world->red[(size_t) me * World::kRed] = v;
    world->barrier();
    for (int q = 0; q < np; q++) out[q] = world->red[(size_t) q * World::kRed];
    world->barrier();
    return out;
  }

  // ---------------------------------------------------------------- bricks (`-np N`)
  // LAMMPS' processor grid (ProcMap::onelevel_grid / best_factors): the factorisation of N with the least surface
  // between bricks, first found on ties; ranks on it as MPI_Cart_create places them (x slowest)
  void choose_grid()
  {
    const double a[3] = {prd[0], 0, 0}, b[3] = {tilt[0], prd[1], 0}, c[3] = {tilt[1], tilt[2], prd[2]};
    auto cross_len = [](const double *u, const double *v) {
      const double x = u[1] * v[2] - u[2] * v[1], y = u[2] * v[0] - u[0] * v[2], z = u[0] * v[1] - u[1] * v[0];
      return sqrt(x * x + y * y + z * z);
    };
    const double ab = cross_len(a, b), ac = cross_len(a, c), bc = cross_len(b, c);
    double best = 1e300;
    int g[3] = {np, 1, 1};
    for (int i = 1; i <= np; i++) {
      if (np % i) continue;
      for (int j = 1; j <= np / i; j++) {
        if ((np / i) % j) continue;
        const int k = np / i / j;
        const double surf = ab / i / j + ac / i / k + bc / j / k;
        if (surf < best) {
          best = surf;
          g[0] = i; g[1] = j; g[2] = k;
        }
      }
    }
    for (int d = 0; d < 3; d++) comm.procgrid[d] = g[d];
    comm.myloc[0] = me / (g[1] * g[2]);
    comm.myloc[1] = (me / g[2]) % g[1];
    comm.myloc[2] = me % g[2];
    for (int d = 0; d < 3; d++) {
      slo[d] = (double) comm.myloc[d] / g[d];
      shi[d] = (double) (comm.myloc[d] + 1) / g[d];
      domain.sublo_lamda[d] = slo[d];
      domain.subhi_lamda[d] = shi[d];
    }
  }
  int rank_of(const double *l) const
  {
    int loc[3];
    for (int d = 0; d < 3; d++) loc[d] = std::min(comm.procgrid[d] - 1, std::max(0, (int) (l[d] * comm.procgrid[d])));
    return (loc[0] * comm.procgrid[1] + loc[1]) * comm.procgrid[2] + loc[2];
  }

  // owned atoms := the given ones (no ghosts)
  void set_owned(const std::vector<double> &x, const std::vector<double> &v, const std::vector<int> &ty, const std::vector<int> &tg)
  {
    const int n = (int) ty.size();
    atom.nlocal = n;
    atom.nghost = 0;
    set_views(n);
    std::copy(x.begin(), x.end(), xs.begin());
    std::copy(ty.begin(), ty.end(), types.begin());
    std::copy(tg.begin(), tg.end(), tags.begin());
    set_vviews();
    std::copy(v.begin(), v.end(), vs.begin());
  }

  // first `run` on N ranks: every rank holds all atoms (wrapped); each keeps those of its brick
  void decompose()
  {
    if (np == 1 || decomposed) return;
    const int n = atom.nlocal;
    natoms_all = n;
    std::vector<double> x, v;
    std::vector<int> ty, tg;
    for (int i = 0; i < n; i++) {
      double l[3];
      x2lamda(xrow[i], l);
      if (rank_of(l) != me) continue;
      x.insert(x.end(), xs.begin() + 3 * (size_t) i, xs.begin() + 3 * (size_t) i + 3);
      v.insert(v.end(), vs.begin() + 3 * (size_t) i, vs.begin() + 3 * (size_t) i + 3);
      ty.push_back(types[i]);
      tg.push_back(tags[i]);
    }
    set_owned(x, v, ty, tg);
    world->host[me] = this;
    outbox.assign(np, {});
    cbuf.assign(np, {});
    decomposed = true;
  }

  // Comm::exchange: atoms that left the brick move to the rank that owns them now (positions are wrapped)
  void exchange()
  {
    if (!multi()) return;
    const int n = atom.nlocal;
    std::vector<double> x, v;
    std::vector<int> ty, tg;
    for (auto &o : outbox) o.clear();
    for (int i = 0; i < n; i++) {
      double l[3];
      x2lamda(xrow[i], l);
      const int dest = rank_of(l);
      if (dest == me) {
        x.insert(x.end(), xs.begin() + 3 * (size_t) i, xs.begin() + 3 * (size_t) i + 3);
        v.insert(v.end(), vs.begin() + 3 * (size_t) i, vs.begin() + 3 * (size_t) i + 3);
        ty.push_back(types[i]);
        tg.push_back(tags[i]);
      } else {
        MoveRec r;
        for (int d = 0; d < 3; d++) {
          r.x[d] = xs[3 * (size_t) i + d];
          r.v[d] = vs[3 * (size_t) i + d];
        }
        r.type = types[i];
        r.tag = tags[i];
        outbox[dest].push_back(r);
      }
    }
    world->barrier();
    for (int q = 0; q < np; q++) {
      if (q == me) continue;
      for (const MoveRec &r : world->host[q]->outbox[me]) {
        x.insert(x.end(), r.x, r.x + 3);
        v.insert(v.end(), r.v, r.v + 3);
        ty.push_back(r.type);
        tg.push_back(r.tag);
      }
    }
    world->barrier();
    set_owned(x, v, ty, tg);
  }

  Host()
  {
    lmp.memory = &memory;
    lmp.error = &error;
    lmp.atom = &atom;
    lmp.comm = &comm;
    lmp.domain = &domain;
    lmp.force = &force;
    lmp.neighbor = &neighbor;
    lmp.update = &update;
    lmp.output = &output;
    comm.h = this;
    avec.h = this;
    atom.avec = &avec;
  }

  // ---------------------------------------------------------------- geometry
  void h_matrix(double h[3][3]) const
  {
    h[0][0] = prd[0]; h[0][1] = tilt[0]; h[0][2] = tilt[1];
    h[1][0] = 0;      h[1][1] = prd[1];  h[1][2] = tilt[2];
    h[2][0] = 0;      h[2][1] = 0;       h[2][2] = prd[2];
  }
  void x2lamda(const double *x, double *l) const
  {
    // invert upper-triangular h: x - lo = h * lamda
    l[2] = (x[2] - boxlo[2]) / prd[2];
    l[1] = ((x[1] - boxlo[1]) - tilt[2] * l[2]) / prd[1];
    l[0] = ((x[0] - boxlo[0]) - tilt[0] * l[1] - tilt[1] * l[2]) / prd[0];
  }
  void lamda2x(const double *l, double *x) const
  {
    x[0] = boxlo[0] + prd[0] * l[0] + tilt[0] * l[1] + tilt[1] * l[2];
    x[1] = boxlo[1] + prd[1] * l[1] + tilt[2] * l[2];
    x[2] = boxlo[2] + prd[2] * l[2];
  }
  double volume() const { return prd[0] * prd[1] * prd[2]; }

  void lattice2box(double &x, double &y, double &z) const
  {
    const double x1 = (a1[0] * x + a2[0] * y + a3[0] * z) * lat_scale;
    const double y1 = (a1[1] * x + a2[1] * y + a3[1] * z) * lat_scale;
    const double z1 = (a1[2] * x + a2[2] * y + a3[2] * z) * lat_scale;
    x = x1 + latsp[0] * lat_origin[0];
    y = y1 + latsp[1] * lat_origin[1];
    z = z1 + latsp[2] * lat_origin[2];
  }

  void setup_lattice()
  {
    // lattice spacings = extent of the unit cell's bounding box
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++)
          for (int d = 0; d < 3; d++) {
            const double v = (i * a1[d] + j * a2[d] + k * a3[d]) * lat_scale;
            lo[d] = std::min(lo[d], v);
            hi[d] = std::max(hi[d], v);
          }
    for (int d = 0; d < 3; d++) latsp[d] = hi[d] - lo[d];
  }

  // ---------------------------------------------------------------- atom storage
  void set_views(int nall)
  {
    xs.resize((size_t) 3 * nall);
    fs.resize((size_t) 3 * nall);
    types.resize(nall);
    tags.resize(nall);
    xrow.resize(nall + 1);
    frow.resize(nall + 1);
    for (int i = 0; i < nall; i++) {
      xrow[i] = xs.data() + 3 * (size_t) i;
      frow[i] = fs.data() + 3 * (size_t) i;
    }
    if (nall == 0) { // keep x[0] / f[0] valid
      xs.resize(3);
      fs.resize(3);
      xrow[0] = xs.data();
      frow[0] = fs.data();
    }
    atom.x = xrow.data();
    atom.f = frow.data();
    atom.type = types.data();
    atom.tag = tags.data();
    atom.nmax = nall;
    atom.mass = masses.data();
  }
  void set_vviews()
  {
    const int n = atom.nlocal;
    vs.resize((size_t) 3 * std::max(n, 1));
    vrow.resize(n + 1);
    for (int i = 0; i < n; i++) vrow[i] = vs.data() + 3 * (size_t) i;
    atom.v = vrow.data();
  }

  // AtomVec::grow as a style calls it: room for n atoms, contents kept (fix nve/mdp brings back more atoms than it took)
  void grow_arrays(int n)
  {
    if (n <= (int) types.size()) return;
    set_views(n); // (std::vector::resize keeps what is there)
    vs.resize((size_t) 3 * n, 0.0);
    vrow.resize(n + 1);
    for (int i = 0; i < n; i++) vrow[i] = vs.data() + 3 * (size_t) i;
    atom.v = vrow.data();
  }

  // ---------------------------------------------------------------- ghosts
  double comm_cutoff() const
  {
    double c = pair ? pair->cutforce : 0.0;
    return c + skin;
  }

  void wrap_owned()
  {
    for (int i = 0; i < atom.nlocal; i++) {
      double l[3];
      x2lamda(xrow[i], l);
      for (int d = 0; d < 3; d++) l[d] -= floor(l[d]);
      lamda2x(l, xrow[i]);
    }
  }

  void build_ghosts()
  {
    if (multi()) {
      build_ghosts_multi();
      return;
    }
    const int n = atom.nlocal;
    const double cut = comm_cutoff();
    // lamda-space half widths: cut * |row_d(h^-1)|
    double c[3];
    {
      const double hinv00 = 1.0 / prd[0], hinv11 = 1.0 / prd[1], hinv22 = 1.0 / prd[2];
      const double hinv01 = -tilt[0] / (prd[0] * prd[1]);
      const double hinv02 = (tilt[0] * tilt[2] - prd[1] * tilt[1]) / (prd[0] * prd[1] * prd[2]);
      const double hinv12 = -tilt[2] / (prd[1] * prd[2]);
      c[0] = cut * sqrt(hinv00 * hinv00 + hinv01 * hinv01 + hinv02 * hinv02);
      c[1] = cut * sqrt(hinv11 * hinv11 + hinv12 * hinv12);
      c[2] = cut * hinv22;
    }
    std::vector<double> lam((size_t) 3 * n);
    for (int i = 0; i < n; i++) x2lamda(xs.data() + 3 * (size_t) i, lam.data() + 3 * (size_t) i);
    ghost_owner.clear();
    ghost_shift.clear();
    int r[3];
    for (int d = 0; d < 3; d++) r[d] = (int) ceil(c[d]);
    double h[3][3];
    h_matrix(h);
    for (int sx = -r[0]; sx <= r[0]; sx++)
      for (int sy = -r[1]; sy <= r[1]; sy++)
        for (int sz = -r[2]; sz <= r[2]; sz++) {
          if (!sx && !sy && !sz) continue;
          const int s[3] = {sx, sy, sz};
          for (int i = 0; i < n; i++) {
            bool in = true;
            for (int d = 0; d < 3 && in; d++) {
              const double l = lam[3 * (size_t) i + d] + s[d];
              in = l >= -c[d] && l < 1.0 + c[d];
            }
            if (!in) continue;
            ghost_owner.push_back(i);
            Vec3 sh;
            for (int d = 0; d < 3; d++) sh[d] = h[d][0] * sx + h[d][1] * sy + h[d][2] * sz;
            ghost_shift.push_back(sh);
          }
        }
    const int ng = (int) ghost_owner.size();
    // re-seat views, keeping owned data
    std::vector<double> xo(xs.begin(), xs.begin() + 3 * (size_t) n);
    std::vector<int> to(types.begin(), types.begin() + n), go(tags.begin(), tags.begin() + n);
    set_views(n + ng);
    std::copy(xo.begin(), xo.end(), xs.begin());
    std::copy(to.begin(), to.end(), types.begin());
    std::copy(go.begin(), go.end(), tags.begin());
    atom.nghost = ng;
    for (int g = 0; g < ng; g++) {
      types[n + g] = types[ghost_owner[g]];
      tags[n + g] = tags[ghost_owner[g]];
    }
    refresh_ghosts();
  }

  void lamda_halo(double c[3]) const // lamda-space half widths of the ghost shell: cut * |row_d(h^-1)|
  {
    const double cut = comm_cutoff();
    const double hinv00 = 1.0 / prd[0], hinv11 = 1.0 / prd[1], hinv22 = 1.0 / prd[2];
    const double hinv01 = -tilt[0] / (prd[0] * prd[1]);
    const double hinv02 = (tilt[0] * tilt[2] - prd[1] * tilt[1]) / (prd[0] * prd[1] * prd[2]);
    const double hinv12 = -tilt[2] / (prd[1] * prd[2]);
    c[0] = cut * sqrt(hinv00 * hinv00 + hinv01 * hinv01 + hinv02 * hinv02);
    c[1] = cut * sqrt(hinv11 * hinv11 + hinv12 * hinv12);
    c[2] = cut * hinv22;
  }

  // Comm::borders on N ranks: every atom of any rank (and any periodic image of it) inside this brick widened by the
  // ghost cutoff in lamda space -- what LAMMPS' six sequential swaps collect --, grouped by the rank that owns it
  void build_ghosts_multi()
  {
    const int n = atom.nlocal;
    double c[3], h[3][3];
    lamda_halo(c);
    h_matrix(h);
    int r[3];
    for (int d = 0; d < 3; d++) r[d] = (int) ceil(c[d]);
    world->barrier(); // (every rank's owned atoms are final)
    std::vector<GhostRec> recs;
    std::vector<double> gx;
    std::vector<int> gty, gtg;
    from_first.assign(np, 0);
    from_count.assign(np, 0);
    for (int q = 0; q < np; q++) {
      const Host *o = world->host[q];
      const int nq = o->atom.nlocal;
      from_first[q] = (int) recs.size();
      std::vector<double> lam((size_t) 3 * nq);
      for (int i = 0; i < nq; i++) o->x2lamda(o->xs.data() + 3 * (size_t) i, lam.data() + 3 * (size_t) i);
      for (int sx = -r[0]; sx <= r[0]; sx++)
        for (int sy = -r[1]; sy <= r[1]; sy++)
          for (int sz = -r[2]; sz <= r[2]; sz++) {
            if (q == me && !sx && !sy && !sz) continue;
            const int s[3] = {sx, sy, sz};
            Vec3 sh;
            for (int d = 0; d < 3; d++) sh[d] = h[d][0] * sx + h[d][1] * sy + h[d][2] * sz;
            for (int i = 0; i < nq; i++) {
              bool in = true;
              for (int d = 0; d < 3 && in; d++) {
                const double l = lam[3 * (size_t) i + d] + s[d];
                in = l >= slo[d] - c[d] && l < shi[d] + c[d];
              }
              if (!in) continue;
              recs.push_back({q, i, sh});
              for (int d = 0; d < 3; d++) gx.push_back(o->xs[3 * (size_t) i + d] + sh[d]);
              gty.push_back(o->types[i]);
              gtg.push_back(o->tags[i]);
            }
          }
      from_count[q] = (int) recs.size() - from_first[q];
    }
    world->barrier(); // (nobody reads my arrays any more: they may move)
    const int ng = (int) recs.size();
    std::vector<double> xo(xs.begin(), xs.begin() + 3 * (size_t) n);
    std::vector<int> to(types.begin(), types.begin() + n), go(tags.begin(), tags.begin() + n);
    set_views(n + ng);
    std::copy(xo.begin(), xo.end(), xs.begin());
    std::copy(to.begin(), to.end(), types.begin());
    std::copy(go.begin(), go.end(), tags.begin());
    std::copy(gx.begin(), gx.end(), xs.begin() + 3 * (size_t) n);
    std::copy(gty.begin(), gty.end(), types.begin() + n);
    std::copy(gtg.begin(), gtg.end(), tags.begin() + n);
    atom.nghost = ng;
    ghosts.swap(recs);
    world->barrier(); // (every rank's ghost records are published)
    sendlist.assign(np, {});
    for (int q = 0; q < np; q++) {
      const Host *o = world->host[q];
      for (int g = o->from_first[me]; g < o->from_first[me] + o->from_count[me]; g++) sendlist[q].push_back(o->ghosts[g].idx);
    }
    world->barrier();
  }

  void refresh_ghosts() // forward comm of x
  {
    if (multi()) {
      const int n = atom.nlocal;
      world->barrier(); // (every rank has moved its atoms)
      for (size_t g = 0; g < ghosts.size(); g++) {
        const double *xo = world->host[ghosts[g].src]->xs.data() + 3 * (size_t) ghosts[g].idx;
        for (int d = 0; d < 3; d++) xs[3 * (n + g) + d] = xo[d] + ghosts[g].shift[d];
      }
      world->barrier();
      return;
    }
    const int n = atom.nlocal, ng = atom.nghost;
    for (int g = 0; g < ng; g++)
      for (int d = 0; d < 3; d++) xs[3 * (size_t) (n + g) + d] = xs[3 * (size_t) ghost_owner[g] + d] + ghost_shift[g][d];
  }

  void fold_ghost_forces() // reverse comm of f
  {
    if (multi()) {
      world->barrier(); // (every rank's forces of this step are in its array)
      for (int q = 0; q < np; q++) {
        const Host *o = world->host[q];
        const double *fo = o->fs.data() + 3 * (size_t) (o->atom.nlocal + o->from_first[me]);
        for (size_t k = 0; k < sendlist[q].size(); k++)
          for (int d = 0; d < 3; d++) fs[3 * (size_t) sendlist[q][k] + d] += fo[3 * k + d];
      }
      world->barrier();
      return;
    }
    const int n = atom.nlocal, ng = atom.nghost;
    for (int g = 0; g < ng; g++)
      for (int d = 0; d < 3; d++) fs[3 * (size_t) ghost_owner[g] + d] += fs[3 * (size_t) (n + g) + d];
  }

  // ---------------------------------------------------------------- neighbor lists (full/bin/ghost)
  void build_neighbor_lists()
  {
    const int n = atom.nlocal, nall = n + atom.nghost, nt = atom.ntypes;
    const bool want_ghost = (neighbor.request_flags & NeighConst::REQ_GHOST) != 0;
    double cutmax = 0.0;
    std::vector<double> cown((size_t) (nt + 1) * (nt + 1), 0.0), cgh((size_t) (nt + 1) * (nt + 1), 0.0);
    for (int i = 1; i <= nt; i++)
      for (int j = 1; j <= nt; j++) {
        const double c = sqrt(pair->cutsq[i][j]) + skin;
        cown[(size_t) i * (nt + 1) + j] = c * c;
        cutmax = std::max(cutmax, c);
        if (want_ghost && pair->cutghost) {
          const double g = pair->cutghost[i][j] + skin;
          cgh[(size_t) i * (nt + 1) + j] = g * g;
        }
      }
    // bins over the Cartesian bounding box of all atoms
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nall; i++)
      for (int d = 0; d < 3; d++) {
        lo[d] = std::min(lo[d], xs[3 * (size_t) i + d]);
        hi[d] = std::max(hi[d], xs[3 * (size_t) i + d]);
      }
    int nb[3];
    double inv[3];
    const double binsize = 0.5 * cutmax;
    for (int d = 0; d < 3; d++) {
      const double len = std::max(hi[d] - lo[d], 1e-9) * (1.0 + 1e-12);
      nb[d] = std::max(1, (int) floor(len / binsize));
      inv[d] = nb[d] / len;
    }
    const size_t nbins = (size_t) nb[0] * nb[1] * nb[2];
    std::vector<int> head(nbins + 1, 0), order(nall), bin_of(nall);
    for (int i = 0; i < nall; i++) {
      int c[3];
      for (int d = 0; d < 3; d++) c[d] = std::min(nb[d] - 1, std::max(0, (int) ((xs[3 * (size_t) i + d] - lo[d]) * inv[d])));
      bin_of[i] = c[0] + nb[0] * (c[1] + nb[1] * c[2]);
      head[bin_of[i] + 1]++;
    }
    for (size_t b = 0; b < nbins; b++) head[b + 1] += head[b];
    {
      std::vector<int> fill(head.begin(), head.end() - 1);
      for (int i = 0; i < nall; i++) order[fill[bin_of[i]]++] = i;
    }
    std::vector<char> excl((size_t) (nt + 1) * (nt + 1), 0);
    for (const auto &e : excl_types)
      if (e.first >= 1 && e.first <= nt && e.second >= 1 && e.second <= nt)
        excl[(size_t) e.first * (nt + 1) + e.second] = excl[(size_t) e.second * (nt + 1) + e.first] = 1;
    nb_store.clear();
    numneigh_v.assign(nall, 0);
    std::vector<size_t> start(nall, 0);
    ilist_v.clear();
    int gnum = 0;
    for (int i = 0; i < nall; i++) {
      const bool owned = i < n;
      if (!owned && !want_ghost) break;
      const std::vector<double> &ctab = owned ? cown : cgh;
      ilist_v.push_back(i);
      if (!owned) gnum++;
      start[i] = nb_store.size();
      int c[3];
      for (int d = 0; d < 3; d++) c[d] = std::min(nb[d] - 1, std::max(0, (int) ((xs[3 * (size_t) i + d] - lo[d]) * inv[d])));
      const double *xi = xs.data() + 3 * (size_t) i;
      const int ti = types[i];
      for (int z = std::max(c[2] - 2, 0); z <= std::min(c[2] + 2, nb[2] - 1); z++)
        for (int y = std::max(c[1] - 2, 0); y <= std::min(c[1] + 2, nb[1] - 1); y++) {
          const size_t b0 = std::max(c[0] - 2, 0) + (size_t) nb[0] * (y + (size_t) nb[1] * z);
          const size_t b1 = std::min(c[0] + 2, nb[0] - 1) + (size_t) nb[0] * (y + (size_t) nb[1] * z);
          for (int p = head[b0]; p < head[b1 + 1]; p++) {
            const int j = order[p];
            if (j == i || excl[(size_t) ti * (nt + 1) + types[j]]) continue;
            const double *xj = xs.data() + 3 * (size_t) j;
            const double dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
            if (dx * dx + dy * dy + dz * dz <= ctab[(size_t) ti * (nt + 1) + types[j]]) nb_store.push_back(j);
          }
        }
      numneigh_v[i] = (int) (nb_store.size() - start[i]);
      if (numneigh_v[i] > neighbor.oneatom) error.one(FLERR, "Neighbor list overflow, boost neigh_modify one");
    }
    firstneigh_v.assign(nall, nullptr);
    for (int i : ilist_v) firstneigh_v[i] = nb_store.data() + start[i];
    list.inum = n;
    list.gnum = gnum;
    list.ilist = ilist_v.data();
    list.numneigh = numneigh_v.data();
    list.firstneigh = firstneigh_v.data();
    pair->list = &list;
    neighbor.ago = 0;
    nbuilds++;
    xhold.assign(xs.begin(), xs.begin() + 3 * (size_t) n);
  }

  bool check_distance() const
  {
    const double trig = 0.25 * skin * skin;
    for (int i = 0; i < atom.nlocal; i++) {
      double d2 = 0;
      for (int d = 0; d < 3; d++) {
        const double dd = xs[3 * (size_t) i + d] - xhold[3 * (size_t) i + d];
        d2 += dd * dd;
      }
      if (d2 > trig) return true;
    }
    return false;
  }

  // ---------------------------------------------------------------- thermo
  double kinetic() // (a sum over ranks: called by every rank)
  {
    double ke = 0;
    for (int i = 0; i < atom.nlocal; i++) {
      const double *v = vs.data() + 3 * (size_t) i;
      ke += masses[types[i]] * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
    sum(&ke, 1);
    return 0.5 * MVV2E * ke;
  }
  double dof() const { return 3.0 * (multi() ? (double) natoms_all : (double) atom.nlocal) - 3.0; }
  double temperature() { return dof() > 0 ? 2.0 * kinetic() / (dof() * BOLTZ) : 0.0; }

  void print_thermo_header()
  {
    std::string s = "  ";
    for (auto &c : thermo_cols) {
      std::string n = c == "step" ? "Step" : c == "temp" ? "Temp" : c == "press" ? "Press" : c == "pe" ? "PotEng"
          : c == "ke"                                                                              ? "KinEng"
          : c == "etotal"                                                                          ? "TotEng"
          : c == "vol"                                                                             ? "Volume"
          : c == "cellgamma"                                                                       ? "CellGamma"
          : c == "epair"                                                                           ? "E_pair"
          : c == "emol"                                                                            ? "E_mol"
                                                                                                   : c;
      char buf[32];
      snprintf(buf, sizeof buf, c == "step" ? "%6s    " : " %-14s", n.c_str());
      s += buf;
    }
    printf("%s\n", s.c_str());
  }

  void print_thermo()
  {
    const double ke = kinetic(), t = dof() > 0 ? 2.0 * ke / (dof() * BOLTZ) : 0.0;
    double tot[4] = {pair ? pair->eng_vdwl : 0.0, pair->virial[0], pair->virial[1], pair->virial[2]};
    sum(tot, 4);
    const double pe = tot[0], *vir = tot + 1;
    const double press = (dof() * BOLTZ * t + vir[0] + vir[1] + vir[2]) / (3.0 * volume()) * NKTV2P;
    std::string s;
    for (auto &c : thermo_cols) {
      char buf[64];
      if (c == "step") {
        snprintf(buf, sizeof buf, "%10ld", step);
      } else {
        double v = 0;
        if (c == "temp") v = t;
        else if (c == "press") v = press;
        else if (c == "pe" || c == "epair") v = pe;
        else if (c == "ke") v = ke;
        else if (c == "etotal") v = pe + ke;
        else if (c == "vol") v = volume();
        else if (c == "cellgamma") {
          // angle between a and b edge vectors
          const double bx = tilt[0], by = prd[1];
          v = acos(bx / sqrt(bx * bx + by * by)) * 180.0 / M_PI;
        }
        snprintf(buf, sizeof buf, "   %-14.8g", v);
        buf[18] = 0; // 3 spaces + 14 chars, like LAMMPS' "{:<14.8g}" columns
      }
      s += buf;
    }
    printf("%s\n", s.c_str());
    fflush(stdout);
  }

  // ---------------------------------------------------------------- Verlet
  void force_clear() { std::fill(fs.begin(), fs.end(), 0.0); }

  void sync_domain() // Domain::set_global_box
  {
    domain.xprd = prd[0]; domain.yprd = prd[1]; domain.zprd = prd[2];
    domain.xy = tilt[0]; domain.xz = tilt[1]; domain.yz = tilt[2];
    domain.triclinic = tilt[0] != 0.0 || tilt[1] != 0.0 || tilt[2] != 0.0;
    for (int d = 0; d < 3; d++) {
      domain.boxlo[d] = boxlo[d];
      domain.boxhi[d] = boxlo[d] + prd[d];
      domain.h[d] = prd[d];
    }
    domain.h[3] = tilt[2]; domain.h[4] = tilt[1]; domain.h[5] = tilt[0];
  }

  void compute_forces(int eflag, int vflag)
  {
    sync_domain();
    force_clear();
    pair->compute(eflag, vflag);
    fold_ghost_forces();
  }

  void run(long nsteps)
  {
    if (!pair) error.all(FLERR, "run: no pair style defined");
    if (fix_style.empty()) error.warning(FLERR, "No fixes with time integration, atoms won't move");
    pair->init();
    force.pair = pair;
    update.dt = dt;
    update.ntimestep = step;
    update.laststep = step + nsteps;
    skin = neighbor.skin;
    wrap_owned();
    if (np > 1 && !decomposed) decompose();
    else exchange();
    build_ghosts();
    build_neighbor_lists();
    if (!multi()) set_vviews();
    if (fix) fix->init(); // (LAMMPS::init: force->init() before modify->init())
    // Neighbor::init(), behind Modify::init() in LAMMPS::init(): its check of the settings a fix may have touched
    if (neighbor.delay > 0 && neighbor.delay % neighbor.every != 0)
      error.all(FLERR, "Neighbor delay must be 0 or multiple of every setting");
    printf("Neighbor list info ...\n  update: every = %d steps, delay = %d steps, check = %s\n", neighbor.every, neighbor.delay,
           neighbor.dist_check ? "yes" : "no");
    printf("  max neighbors/atom: %d, page size: %d\n  master list distance cutoff = %g\n  ghost atom cutoff = %g\n",
           neighbor.oneatom, neighbor.pgsize, pair->cutforce + skin, comm_cutoff());
    printf("  pair %s, perpetual\n      attributes: full, newton on%s\n", "style", (neighbor.request_flags & NeighConst::REQ_GHOST) ? ", ghost" : "");
    const int every = thermo_every;
    output.next = output.next_thermo = step; // (setup: thermo of the initial state)
    compute_forces(1, 2);
    if (fix) fix->setup(2);
    print_thermo_header();
    print_thermo();
    const double dtf = 0.5 * dt * FTM2V;
    const long first = step;
    const auto t0 = std::chrono::steady_clock::now();
    int nbuild0 = nbuilds;
    for (long k = 1; k <= nsteps; k++) {
      step = first + k;
      update.ntimestep = step;
      { // Output::next: the next thermo step, or the last step of the run
        long nt = every > 0 ? (step + every - 1) / every * every : first + nsteps;
        output.next = output.next_thermo = std::min<long>(nt, first + nsteps);
      }
      if (fix) { // Verlet::run with a time-integration fix style from a plugin
        fix->initial_integrate((every > 0 && step % every == 0) || k == nsteps ? 2 : 0); // (ev_set: this step's vflag)
        // Neighbor::decide(): fixes that ask for a reneighboring on this step, then every / delay / check
        bool nflag = fix->force_reneighbor && fix->next_reneighbor == step;
        if (!nflag) {
          neighbor.ago++;
          if (neighbor.ago >= neighbor.delay && neighbor.ago % neighbor.every == 0) nflag = neighbor.dist_check ? any(check_distance()) : true;
        }
        if (nflag) {
          wrap_owned();
          exchange();
          build_ghosts();
          build_neighbor_lists();
        }
        const bool out = every > 0 && (step % every == 0);
        const bool last = k == nsteps;
        if (out || last)
          compute_forces(1, 2);
        else { // (the style adds nothing to the host's f on such a step -- its reader is on the device: no force_clear
               //  of 24 bytes per atom, no fold of ghost forces.  Verlet::force_clear of a real LAMMPS is a memset per step.)
          sync_domain();
          pair->compute(0, 0);
        }
        fix->final_integrate();
        if (out || last) print_thermo();
        continue;
      }
      const int n = atom.nlocal;
      double tscale = 1.0;
      if (fix_style == "nvt") { // single Nose-Hoover thermostat, half step
        const double ttarget = nvt_t0 + (nvt_t1 - nvt_t0) * (double) (k - 1) / (double) nsteps;
        const double tcur = temperature();
        nvt_eta_dot += 0.5 * dt * (tcur / ttarget - 1.0) / (nvt_damp * nvt_damp);
        tscale = exp(-0.5 * dt * nvt_eta_dot);
      }
      for (int i = 0; i < n; i++) {
        const double s = dtf / masses[types[i]];
        for (int d = 0; d < 3; d++) {
          double &v = vs[3 * (size_t) i + d];
          v = v * tscale + s * fs[3 * (size_t) i + d];
          xs[3 * (size_t) i + d] += dt * v;
        }
      }
      if (any(check_distance())) {
        wrap_owned();
        exchange();
        build_ghosts();
        build_neighbor_lists();
      } else {
        refresh_ghosts();
        neighbor.ago++;
      }
      const bool out = every > 0 && (step % every == 0);
      const bool last = k == nsteps;
      compute_forces((out || last) ? 1 : 0, (out || last) ? 2 : 0);
      for (int i = 0; i < atom.nlocal; i++) { // (nlocal: atoms may have changed ranks at the reneighboring)
        const double s = dtf / masses[types[i]];
        for (int d = 0; d < 3; d++) vs[3 * (size_t) i + d] += s * fs[3 * (size_t) i + d];
      }
      if (fix_style == "nvt") {
        const double ttarget = nvt_t0 + (nvt_t1 - nvt_t0) * (double) k / (double) nsteps;
        const double tcur = temperature();
        nvt_eta_dot += 0.5 * dt * (tcur / ttarget - 1.0) / (nvt_damp * nvt_damp);
        const double sc = exp(-0.5 * dt * nvt_eta_dot);
        for (auto &v : vs) v *= sc;
      }
      if (out || last) print_thermo();
    }
    const double loop = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (fix) fix->post_run(); // (Modify::post_run, behind Verlet::run in Run::command)
    const long nat = multi() ? natoms_all : (long) atom.nlocal;
    printf("Loop time of %g on %d procs for %ld steps with %ld atoms\n\n", loop, np, nsteps, nat);
    if (nsteps > 0 && loop > 0) {
      const double sps = nsteps / loop;
      printf("Performance: %.3f ns/day, %.3f hours/ns, %.3f timesteps/s, %.3f katom-step/s\n", sps * dt * 86.4,
             1.0 / (sps * dt * 3.6), sps, sps * nat / 1000.0);
    }
    long nn = 0;
    for (int i = 0; i < std::min<int>(atom.nlocal, (int) numneigh_v.size()); i++) nn += numneigh_v[i];
    if (multi()) { // per-rank counts as LAMMPS' Finish prints them (ave / max / min), and rank by rank
      const std::vector<double> nl = gather(atom.nlocal), ng = gather(atom.nghost), nf = gather((double) nn);
      auto line = [&](const char *name, const std::vector<double> &v) {
        double a = 0, hi = v[0], lo = v[0];
        for (double x : v) {
          a += x;
          hi = std::max(hi, x);
          lo = std::min(lo, x);
        }
        printf("%-10s %10g ave %11g max %11g min\n", name, a / v.size(), hi, lo);
        return a;
      };
      printf("\n");
      line("Nlocal:", nl);
      line("Nghost:", ng);
      const double tot = line("FullNghs:", nf);
      for (int q = 0; q < np; q++) printf("rank %d: Nlocal %d  Nghost %d  FullNghs %ld\n", q, (int) nl[q], (int) ng[q], (long) nf[q]);
      printf("\nTotal # of neighbors = %ld\nAve neighs/atom = %g\nNeighbor list builds = %d\n\n", (long) tot, nat ? tot / nat : 0.0,
             nbuilds - nbuild0);
      return;
    }
    printf("\nNlocal:    %d\nNghost:    %d\nFullNghs:  %ld\nAve neighs/atom = %g\nNeighbor list builds = %d\n\n", atom.nlocal,
           atom.nghost, nn, atom.nlocal ? (double) nn / atom.nlocal : 0.0, nbuilds - nbuild0 - 0);
  }
};

void HostAtomVec::grow(int n) { h->grow_arrays(n); }

void PeriodicComm::forward_comm(Pair *pair)
{
  // owner -> ghost through the style's own pack/unpack callbacks (one double per atom)
  if (h->multi()) { // between ranks: one buffer per peer, packed by the owner, unpacked into the peer's block of ghosts
    const int nf = std::max(1, pair->comm_forward), np = h->np, me = h->me;
    for (int r = 0; r < np; r++) {
      h->cbuf[r].resize(h->sendlist[r].size() * (size_t) nf);
      if (!h->sendlist[r].empty()) pair->pack_forward_comm((int) h->sendlist[r].size(), h->sendlist[r].data(), h->cbuf[r].data(), 0, nullptr);
    }
    h->world->barrier();
    for (int q = 0; q < np; q++)
      if (h->from_count[q])
        pair->unpack_forward_comm(h->from_count[q], h->atom.nlocal + h->from_first[q], h->world->host[q]->cbuf[me].data());
    h->world->barrier();
    return;
  }
  const int ng = h->atom.nghost, n = h->atom.nlocal;
  if (!ng) return;
  std::vector<double> buf((size_t) ng * std::max(1, pair->comm_forward));
  std::vector<int> lst(h->ghost_owner.begin(), h->ghost_owner.end());
  pair->pack_forward_comm(ng, lst.data(), buf.data(), 0, nullptr);
  pair->unpack_forward_comm(ng, n, buf.data());
}

void PeriodicComm::reverse_comm(Pair *pair)
{
  if (h->multi()) {
    const int nr = std::max(1, pair->comm_reverse), np = h->np, me = h->me;
    for (int q = 0; q < np; q++) {
      h->cbuf[q].resize((size_t) h->from_count[q] * nr);
      if (h->from_count[q]) pair->pack_reverse_comm(h->from_count[q], h->atom.nlocal + h->from_first[q], h->cbuf[q].data());
    }
    h->world->barrier();
    for (int r = 0; r < np; r++)
      if (!h->sendlist[r].empty())
        pair->unpack_reverse_comm((int) h->sendlist[r].size(), h->sendlist[r].data(), h->world->host[r]->cbuf[me].data());
    h->world->barrier();
    return;
  }
  const int ng = h->atom.nghost, n = h->atom.nlocal;
  if (!ng) return;
  std::vector<double> buf((size_t) ng * std::max(1, pair->comm_reverse));
  std::vector<int> lst(h->ghost_owner.begin(), h->ghost_owner.end());
  pair->pack_reverse_comm(ng, n, buf.data());
  pair->unpack_reverse_comm(ng, lst.data(), buf.data());
}

// =================================================================================================
// input script
// =================================================================================================
double eval_expr(const std::string &s, size_t &p);
double eval_atom(const std::string &s, size_t &p)
{
  while (p < s.size() && isspace((unsigned char) s[p])) p++;
  if (p < s.size() && s[p] == '(') {
    p++;
    const double v = eval_expr(s, p);
    while (p < s.size() && s[p] != ')') p++;
    p++;
    return v;
  }
  if (p < s.size() && (s[p] == '-' || s[p] == '+')) {
    const char c = s[p++];
    const double v = eval_atom(s, p);
    return c == '-' ? -v : v;
  }
  size_t used = 0;
  const double v = std::stod(s.substr(p), &used);
  p += used;
  return v;
}
double eval_term(const std::string &s, size_t &p)
{
  double v = eval_atom(s, p);
  for (;;) {
    while (p < s.size() && isspace((unsigned char) s[p])) p++;
    if (p < s.size() && (s[p] == '*' || s[p] == '/')) {
      const char c = s[p++];
      const double r = eval_atom(s, p);
      v = c == '*' ? v * r : v / r;
    } else
      return v;
  }
}
double eval_expr(const std::string &s, size_t &p)
{
  double v = eval_term(s, p);
  for (;;) {
    while (p < s.size() && isspace((unsigned char) s[p])) p++;
    if (p < s.size() && (s[p] == '+' || s[p] == '-')) {
      const char c = s[p++];
      const double r = eval_term(s, p);
      v = c == '+' ? v + r : v - r;
    } else
      return v;
  }
}

std::string substitute(const std::string &line)
{
  // $(expr) immediate evaluation, printed like LAMMPS' "%.15g"-ish default (20 significant digits)
  std::string out;
  for (size_t i = 0; i < line.size();) {
    if (line[i] == '$' && i + 1 < line.size() && line[i + 1] == '(') {
      size_t depth = 0, j = i + 1;
      for (; j < line.size(); j++) {
        if (line[j] == '(') depth++;
        if (line[j] == ')' && --depth == 0) break;
      }
      size_t p = 0;
      const std::string inner = line.substr(i + 2, j - i - 2);
      char buf[64];
      snprintf(buf, sizeof buf, "%.20g", eval_expr(inner, p));
      out += buf;
      i = j + 1;
    } else
      out += line[i++];
  }
  return out;
}

uint64_t splitmix(uint64_t &s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
double uniform01(uint64_t &s) { return (splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
double gaussian(uint64_t &s)
{
  double u1 = uniform01(s), u2 = uniform01(s);
  if (u1 < 1e-300) u1 = 1e-300;
  return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

thread_local Host *g_host = nullptr; // (the host of this rank thread, for the plugin registration callback)

struct Script {
  Host &H;
  explicit Script(Host &h) : H(h) {}

  static std::vector<std::string> split(const std::string &s)
  {
    std::vector<std::string> w;
    std::istringstream is(s);
    std::string t;
    while (is >> t) w.push_back(t);
    return w;
  }

  void region_bounds(const Region &r, double lo[3], double hi[3], double tl[3]) const
  {
    for (int d = 0; d < 3; d++) {
      lo[d] = r.lo[d] * H.latsp[d];
      hi[d] = r.hi[d] * H.latsp[d];
    }
    tl[0] = r.tilt[0] * H.latsp[0];
    tl[1] = r.tilt[1] * H.latsp[0];
    tl[2] = r.tilt[2] * H.latsp[1];
  }

  void create_atoms(const std::vector<std::string> &w)
  {
    if (!H.box_exists) H.error.all(FLERR, "Create_atoms command before simulation box is defined");
    if (H.decomposed) H.error.all(FLERR, "minilmp -np N: create_atoms after the first run is not supported");
    const int deftype = std::stoi(w[1]);
    std::vector<int> btype(H.basis.size(), deftype);
    for (size_t k = 3; k + 2 < w.size() + 0; k++)
      if (w[k] == "basis") {
        const int b = std::stoi(w[k + 1]), t = std::stoi(w[k + 2]);
        if (b < 1 || b > (int) H.basis.size()) H.error.all(FLERR, "Invalid basis setting in create_atoms command");
        btype[b - 1] = t;
      }
    // lattice index range that covers the box: corners of the box in lattice coordinates
    double h[3][3];
    H.h_matrix(h);
    double A[3][3] = {{H.a1[0], H.a2[0], H.a3[0]}, {H.a1[1], H.a2[1], H.a3[1]}, {H.a1[2], H.a2[2], H.a3[2]}};
    for (auto &row : A)
      for (double &v : row) v *= H.lat_scale;
    // inverse of A
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
        A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    double Ai[3][3];
    Ai[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
    Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
    Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    Ai[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
    Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
    Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    Ai[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
    Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
    Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    int ilo[3] = {1 << 30, 1 << 30, 1 << 30}, ihi[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
    for (int cx = 0; cx < 2; cx++)
      for (int cy = 0; cy < 2; cy++)
        for (int cz = 0; cz < 2; cz++) {
          const double l[3] = {(double) cx, (double) cy, (double) cz};
          double x[3];
          H.lamda2x(l, x);
          for (int d = 0; d < 3; d++) x[d] -= H.latsp[d] * H.lat_origin[d];
          for (int d = 0; d < 3; d++) {
            const double u = Ai[d][0] * x[0] + Ai[d][1] * x[1] + Ai[d][2] * x[2];
            ilo[d] = std::min(ilo[d], (int) floor(u) - 1);
            ihi[d] = std::max(ihi[d], (int) ceil(u) + 1);
          }
        }
    std::vector<double> xn;
    std::vector<int> tn;
    for (int k = ilo[2]; k <= ihi[2]; k++)
      for (int j = ilo[1]; j <= ihi[1]; j++)
        for (int i = ilo[0]; i <= ihi[0]; i++)
          for (size_t b = 0; b < H.basis.size(); b++) {
            double x = i + H.basis[b][0], y = j + H.basis[b][1], z = k + H.basis[b][2];
            H.lattice2box(x, y, z);
            const double p[3] = {x, y, z};
            double l[3];
            H.x2lamda(p, l);
            // LAMMPS: periodic lower bound -EPS, upper bound 1-2EPS in lamda (EPS = 1e-6)
            bool in = true;
            for (int d = 0; d < 3; d++) in = in && l[d] >= -1.0e-6 && l[d] < 1.0 - 2.0e-6;
            if (!in) continue;
            xn.insert(xn.end(), p, p + 3);
            tn.push_back(btype[b]);
          }
    const int n0 = H.atom.nlocal, nadd = (int) tn.size();
    std::vector<double> xo(H.xs.begin(), H.xs.begin() + 3 * (size_t) n0);
    std::vector<int> to(H.types.begin(), H.types.begin() + n0), go(H.tags.begin(), H.tags.begin() + n0);
    H.atom.nlocal = n0 + nadd;
    H.atom.nghost = 0;
    H.set_views(n0 + nadd);
    std::copy(xo.begin(), xo.end(), H.xs.begin());
    std::copy(to.begin(), to.end(), H.types.begin());
    std::copy(go.begin(), go.end(), H.tags.begin());
    for (int a = 0; a < nadd; a++) {
      for (int d = 0; d < 3; d++) H.xs[3 * (size_t) (n0 + a) + d] = xn[3 * (size_t) a + d];
      H.types[n0 + a] = tn[a];
      H.tags[n0 + a] = n0 + a + 1;
    }
    std::vector<double> vo(H.vs.begin(), H.vs.end());
    H.set_vviews();
    std::fill(H.vs.begin(), H.vs.end(), 0.0);
    std::copy(vo.begin(), vo.begin() + std::min(vo.size(), (size_t) 3 * n0), H.vs.begin());
    H.atom.natoms = H.atom.nlocal;
    printf("Created %d atoms\n", nadd);
  }

  void replicate(int nx, int ny, int nz)
  {
    if (H.decomposed) H.error.all(FLERR, "minilmp -np N: replicate after the first run is not supported");
    const int n0 = H.atom.nlocal;
    double h[3][3];
    H.h_matrix(h);
    std::vector<double> xo(H.xs.begin(), H.xs.begin() + 3 * (size_t) n0), vo(H.vs.begin(), H.vs.begin() + 3 * (size_t) n0);
    std::vector<int> to(H.types.begin(), H.types.begin() + n0);
    const int nn = n0 * nx * ny * nz;
    H.atom.nlocal = nn;
    H.atom.nghost = 0;
    H.set_views(nn);
    H.set_vviews();
    int a = 0;
    for (int k = 0; k < nz; k++)
      for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++)
          for (int q = 0; q < n0; q++, a++) {
            for (int d = 0; d < 3; d++) {
              H.xs[3 * (size_t) a + d] = xo[3 * (size_t) q + d] + i * h[d][0] + j * h[d][1] + k * h[d][2];
              H.vs[3 * (size_t) a + d] = vo[3 * (size_t) q + d];
            }
            H.types[a] = to[q];
            H.tags[a] = a + 1;
          }
    H.prd[0] *= nx;
    H.prd[1] *= ny;
    H.prd[2] *= nz;
    H.tilt[0] *= ny;
    H.tilt[1] *= nz;
    H.tilt[2] *= nz;
    H.atom.natoms = nn;
    printf("Replicated to %d atoms\n", nn);
    if (H.np > 1) {
      H.choose_grid();
      printf("  %d by %d by %d MPI processor grid\n", H.comm.procgrid[0], H.comm.procgrid[1], H.comm.procgrid[2]);
    }
  }

  void command(const std::string &raw)
  {
    std::string line = raw;
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    line = substitute(line);
    auto w = split(line);
    if (w.empty()) return;
    if (!H.quiet) printf("%s\n", raw.c_str());
    const std::string &c = w[0];
    auto need = [&](size_t n) {
      if (w.size() < n) H.error.all(FLERR, "Illegal " + c + " command");
    };
    if (c == "units") {
      need(2);
      if (w[1] != "metal") H.error.all(FLERR, "minilmp supports units metal only");
    } else if (c == "atom_style" || c == "dimension" || c == "boundary" || c == "atom_modify" || c == "echo" ||
               c == "log" || c == "dump" || c == "dump_modify" || c == "restart" || c == "comm_modify") {
      // accepted, fixed behaviour: atomic, 3d, p p p
    } else if (c == "newton") {
      need(2);
      H.force.newton_pair = (w[1] == "on");
    } else if (c == "lattice") {
      need(3);
      H.lat_style = w[1];
      H.lat_scale = std::stod(w[2]);
      H.basis.clear();
      for (int d = 0; d < 3; d++) H.lat_origin[d] = 0.0;
      const double e1[3] = {1, 0, 0}, e2[3] = {0, 1, 0}, e3[3] = {0, 0, 1};
      std::copy(e1, e1 + 3, H.a1);
      std::copy(e2, e2 + 3, H.a2);
      std::copy(e3, e3 + 3, H.a3);
      if (w[1] == "fcc") {
        H.basis = {{{0, 0, 0}}, {{0.5, 0.5, 0}}, {{0.5, 0, 0.5}}, {{0, 0.5, 0.5}}};
      } else if (w[1] == "sc") {
        H.basis = {{{0, 0, 0}}};
      } else if (w[1] == "bcc") {
        H.basis = {{{0, 0, 0}}, {{0.5, 0.5, 0.5}}};
      } else if (w[1] != "custom")
        H.error.all(FLERR, "Illegal lattice command");
      for (size_t k = 3; k < w.size();) {
        auto vec = [&](double *o) {
          if (k + 3 >= w.size()) H.error.all(FLERR, "Illegal lattice command");
          for (int d = 0; d < 3; d++) o[d] = std::stod(w[k + 1 + d]);
          k += 4;
        };
        if (w[k] == "a1") vec(H.a1);
        else if (w[k] == "a2") vec(H.a2);
        else if (w[k] == "a3") vec(H.a3);
        else if (w[k] == "origin") vec(H.lat_origin);
        else if (w[k] == "basis") {
          Vec3 b;
          vec(b.v);
          H.basis.push_back(b);
        } else
          H.error.all(FLERR, "Illegal lattice command");
      }
      H.setup_lattice();
      printf("Lattice spacing in x,y,z = %.8g %.8g %.8g\n", H.latsp[0], H.latsp[1], H.latsp[2]);
    } else if (c == "region") {
      need(9);
      Region r;
      r.prism = w[2] == "prism";
      if (!r.prism && w[2] != "block") H.error.all(FLERR, "minilmp supports region block|prism only");
      for (int d = 0; d < 3; d++) {
        r.lo[d] = std::stod(w[3 + 2 * d]);
        r.hi[d] = std::stod(w[4 + 2 * d]);
        r.tilt[d] = 0;
      }
      if (r.prism) {
        need(12);
        for (int d = 0; d < 3; d++) r.tilt[d] = std::stod(w[9 + d]);
      }
      H.regions[w[1]] = r;
    } else if (c == "create_box") {
      need(3);
      const int nt = std::stoi(w[1]);
      if (!H.regions.count(w[2])) H.error.all(FLERR, "Create_box region ID does not exist");
      const Region &r = H.regions[w[2]];
      double lo[3], hi[3], tl[3];
      region_bounds(r, lo, hi, tl);
      for (int d = 0; d < 3; d++) {
        H.boxlo[d] = lo[d];
        H.prd[d] = hi[d] - lo[d];
        H.tilt[d] = tl[d];
      }
      H.atom.ntypes = nt;
      H.masses.assign(nt + 1, 0.0);
      H.atom.mass = H.masses.data();
      H.box_exists = true;
      H.atom.nlocal = H.atom.nghost = 0;
      H.set_views(0);
      H.set_vviews();
      if (r.prism)
        printf("Created triclinic box = (%.8g %.8g %.8g) to (%.8g %.8g %.8g) with tilt (%.8g %.8g %.8g)\n", lo[0], lo[1],
               lo[2], hi[0], hi[1], hi[2], tl[0], tl[1], tl[2]);
      else
        printf("Created orthogonal box = (%.8g %.8g %.8g) to (%.8g %.8g %.8g)\n", lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
      if (H.np > 1) { // (a later `replicate` scales every edge alike here or not at all in the tests; LAMMPS re-maps too)
        H.choose_grid();
        printf("  %d by %d by %d MPI processor grid\n", H.comm.procgrid[0], H.comm.procgrid[1], H.comm.procgrid[2]);
      }
    } else if (c == "create_atoms") {
      need(3);
      create_atoms(w);
    } else if (c == "replicate") {
      need(4);
      replicate(std::stoi(w[1]), std::stoi(w[2]), std::stoi(w[3]));
    } else if (c == "mass") {
      need(3);
      const int t = std::stoi(w[1]);
      if (t < 1 || t > H.atom.ntypes) H.error.all(FLERR, "Invalid type for mass set");
      H.masses[t] = std::stod(w[2]);
    } else if (c == "plugin") {
      need(3);
      if (w[1] != "load") H.error.all(FLERR, "minilmp supports `plugin load` only");
      std::string sofile = w[2];
      {
        struct stat st;
        if (sofile.find('/') == std::string::npos && stat(sofile.c_str(), &st) == 0) sofile = "./" + sofile;
      }
      void *dso = dlopen(sofile.c_str(), RTLD_NOW | RTLD_GLOBAL);
      if (!dso) H.error.all(FLERR, std::string("Open of file ") + w[2] + " failed: " + dlerror());
      void *sym = dlsym(dso, "lammpsplugin_init");
      if (!sym) H.error.all(FLERR, "Plugin symbol lookup failure in file " + w[2] + ": lammpsplugin_init");
      H.handles.push_back(dso);
      const size_t before = H.pair_styles.size() + H.fix_styles.size();
      // registration callback: the host behind the LAMMPS* it is handed
      struct Trampoline {
        static void regfunc(lammpsplugin_t *p, void *lmp)
        {
          Host *host = g_host;
          if (!host || lmp != (void *) &host->lmp || !p || !p->style || !p->name) return;
          const bool is_pair = strcmp(p->style, "pair") == 0, is_fix = strcmp(p->style, "fix") == 0;
          if (!is_pair && !is_fix) {
            fprintf(stderr, "WARNING: plugin style %s/%s ignored (minilmp hosts pair and fix styles only)\n", p->style, p->name);
            return;
          }
          if (strcmp(p->version, LAMMPS_VERSION) != 0)
            fprintf(stderr, "WARNING: plugin %s was compiled for LAMMPS version %s, host is %s\n", p->name, p->version,
                    LAMMPS_VERSION);
          if (is_fix && host->fix_styles.count(p->name)) { // (LAMMPS' own message for a style that is loaded already)
            fprintf(stderr, "WARNING: Ignoring load of fix style %s: must unload existing %s plugin first\n", p->name, p->name);
            return;
          }
          if (is_pair) host->pair_styles[p->name] = p->creator.v1;
          else host->fix_styles[p->name] = p->creator.v2;
          printf("Loading plugin: %s by %s\n", p->info, p->author);
        }
      };
      g_host = &H;
      reinterpret_cast<lammpsplugin_initfunc>(sym)(&H.lmp, dso, (void *) &Trampoline::regfunc);
      printf("Loaded %zu plugins from %s\n", H.pair_styles.size() + H.fix_styles.size() - before, w[2].c_str());
    } else if (c == "pair_style") {
      need(2);
      if (!H.pair_styles.count(w[1])) H.error.all(FLERR, "Unrecognized pair style '" + w[1] + "' (load its plugin first)");
      delete H.pair;
      H.pair = static_cast<Pair *>(H.pair_styles[w[1]](&H.lmp));
      std::vector<char *> args;
      for (size_t k = 2; k < w.size(); k++) args.push_back(const_cast<char *>(w[k].c_str()));
      H.pair->settings((int) args.size(), args.data());
    } else if (c == "pair_coeff") {
      if (!H.pair) H.error.all(FLERR, "Pair_coeff command before pair_style is defined");
      std::vector<char *> args;
      for (size_t k = 1; k < w.size(); k++) args.push_back(const_cast<char *>(w[k].c_str()));
      H.pair->coeff((int) args.size(), args.data());
    } else if (c == "neighbor") {
      need(2);
      H.neighbor.skin = std::stod(w[1]);
    } else if (c == "neigh_modify") {
      for (size_t k = 1; k + 1 < w.size(); k += 2) {
        if (w[k] == "exclude") { // neigh_modify exclude type M N: no list entries between atoms of types M and N
          if (k + 3 >= w.size() || w[k + 1] != "type") H.error.all(FLERR, "minilmp supports `neigh_modify exclude type M N` only");
          const int a = std::stoi(w[k + 2]), b = std::stoi(w[k + 3]);
          H.excl_types.push_back({a, b});
          k += 2;
          continue;
        }
        if (w[k] == "one") H.neighbor.oneatom = std::stoi(w[k + 1]);
        if (w[k] == "page") H.neighbor.pgsize = std::stoi(w[k + 1]);
        if (w[k] == "every") H.neighbor.every = std::max(1, std::stoi(w[k + 1]));
        if (w[k] == "delay") H.neighbor.delay = std::max(0, std::stoi(w[k + 1]));
        if (w[k] == "check") H.neighbor.dist_check = w[k + 1] == "yes" ? 1 : 0;
      }
    } else if (c == "set") {
      // set region ID type/fraction T f seed  (own RNG -- not LAMMPS' RanMars)
      need(7);
      if (w[1] != "region" || w[3] != "type/fraction") H.error.all(FLERR, "minilmp supports `set region ID type/fraction` only");
      const int t = std::stoi(w[4]);
      const double frac = std::stod(w[5]);
      uint64_t seed = std::stoull(w[6]);
      int count = 0;
      for (int i = 0; i < H.atom.nlocal; i++) {
        uint64_t s = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t) H.tags[i]);
        if (uniform01(s) < frac) {
          H.types[i] = t;
          count++;
        }
      }
      {
        double cnt = count;
        H.sum(&cnt, 1);
        count = (int) cnt;
      }
      printf("Setting atom values ...\n  %d settings made for type/fraction\n", count);
    } else if (c == "velocity") {
      need(5);
      if (w[1] != "all" || w[2] != "create") H.error.all(FLERR, "minilmp supports `velocity all create T seed` only");
      const double T = std::stod(w[3]);
      uint64_t seed = std::stoull(w[4]);
      const int n = H.atom.nlocal;
      double p[3] = {0, 0, 0}, mt = 0;
      for (int i = 0; i < n; i++) {
        uint64_t s = seed ^ (0xD1B54A32D192ED03ull * (uint64_t) H.tags[i]);
        const double m = H.masses[H.types[i]];
        for (int d = 0; d < 3; d++) {
          H.vs[3 * (size_t) i + d] = gaussian(s) / sqrt(m);
          p[d] += m * H.vs[3 * (size_t) i + d];
        }
        mt += m;
      }
      {
        double t4[4] = {p[0], p[1], p[2], mt};
        H.sum(t4, 4);
        p[0] = t4[0]; p[1] = t4[1]; p[2] = t4[2]; mt = t4[3];
      }
      for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) H.vs[3 * (size_t) i + d] -= p[d] / mt;
      const double t = H.temperature();
      if (t > 0)
        for (int i = 0; i < 3 * n; i++) H.vs[i] *= sqrt(T / t);
    } else if (c == "fix") {
      need(4);
      if (w[3] == "nve")
        H.fix_style = "nve";
      else if (w[3] == "nvt") {
        need(8);
        H.fix_style = "nvt";
        H.nvt_t0 = std::stod(w[5]);
        H.nvt_t1 = std::stod(w[6]);
        H.nvt_damp = std::stod(w[7]);
        H.nvt_eta_dot = 0.0;
      } else if (H.fix_styles.count(w[3])) { // a fix style from a plugin: fix ID group style args
        delete H.fix;
        std::vector<char *> args;
        for (size_t k = 1; k < w.size(); k++) args.push_back(const_cast<char *>(w[k].c_str()));
        H.fix = static_cast<Fix *>(H.fix_styles[w[3]](&H.lmp, (int) args.size(), args.data()));
        if (!(H.fix->setmask() & (FixConst::INITIAL_INTEGRATE | FixConst::FINAL_INTEGRATE)) || !H.fix->time_integrate)
          H.error.all(FLERR, "minilmp hosts time-integration fix styles only");
        H.fix_style = "plugin";
      } else
        H.error.all(FLERR, "minilmp supports fix nve|nvt and time-integration fix styles of loaded plugins only");
    } else if (c == "timestep") {
      need(2);
      H.dt = std::stod(w[1]);
    } else if (c == "thermo") {
      need(2);
      H.thermo_every = std::stoi(w[1]);
    } else if (c == "thermo_style") {
      need(2);
      if (w[1] == "custom") H.thermo_cols.assign(w.begin() + 2, w.end());
      else if (w[1] == "one") H.thermo_cols = {"step", "temp", "epair", "emol", "etotal", "press"};
    } else if (c == "thermo_modify") {
    } else if (c == "run") {
      need(2);
      H.run(std::stol(w[1]));
    } else
      H.error.all(FLERR, "Unknown command: " + raw);
  }

  void file(std::istream &in)
  {
    std::string line, acc;
    while (std::getline(in, line)) {
      // strip a trailing comment before looking for the continuation character
      size_t end = line.find_last_not_of(" \t\r");
      if (end != std::string::npos && line[end] == '&') {
        acc += line.substr(0, end) + " ";
        continue;
      }
      acc += line;
      command(acc);
      acc.clear();
    }
    if (!acc.empty()) command(acc);
  }
};

} // namespace

// the one MPI call a plugin style makes (lammps_host_api.h): root's bytes to every rank thread
extern "C" int MPI_Bcast(void *buffer, int count, MPI_Datatype, int root, MPI_Comm)
{
  World *w = t_world;
  if (!w || w->n == 1 || count <= 0) return 0;
  if (t_rank == root) w->bbuf.assign((const char *) buffer, (const char *) buffer + count);
  w->barrier();
  if (t_rank != root) memcpy(buffer, w->bbuf.data(), (size_t) count);
  w->barrier();
  return 0;
}

int main(int argc, char **argv)
{
  std::string infile;
  bool quiet = false;
  int np = 1;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if ((a == "-in" || a == "-i") && i + 1 < argc)
      infile = argv[++i];
    else if (a == "-np" && i + 1 < argc)
      np = std::max(1, atoi(argv[++i]));
    else if (a == "-quiet")
      quiet = true;
    else if (a == "-h" || a == "-help") {
      printf("usage: minilmp [-np N] -in script   (subset of LAMMPS input; -np N: N ranks as threads; see INTEGRATION.md)\n");
      return 0;
    }
  }
  printf("minilmp (mini-host for the MI355X pair-style plugins; API subset of LAMMPS %s)\n", LAMMPS_VERSION);
  std::string text;
  if (infile.empty()) {
    std::stringstream ss;
    ss << std::cin.rdbuf();
    text = ss.str();
  } else {
    std::ifstream f(infile);
    if (!f) {
      fprintf(stderr, "ERROR: Cannot open input script %s\n", infile.c_str());
      return 1;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    text = ss.str();
  }
  World world(np);
  std::vector<int> rc(np, 0);
  // one rank: the whole host in this thread.  N ranks: N threads, each with its own host objects, as N MPI processes have
  auto rank_main = [&](int me) {
    t_mute = me != 0;
    t_rank = me;
    t_world = &world;
    try {
      Host H;
      H.quiet = quiet;
      H.world = &world;
      H.me = H.comm.me = me;
      H.np = H.comm.nprocs = np;
      world.host[me] = &H;
      Script S(H);
      std::istringstream in(text);
      S.file(in);
      world.barrier(); // (no rank tears its atoms down while another still reads them)
      // LAMMPS::destroy(): Force (and its pair style) goes before Modify (and its fixes)
      delete H.pair;
      H.pair = nullptr;
      H.force.pair = nullptr;
      delete H.fix;
      H.fix = nullptr;
    } catch (const HostAbort &e) {
      if (e.what()[0]) fprintf(stderr, "%s\n", e.what());
      rc[me] = 1;
      world.kill();
    } catch (const std::exception &e) {
      fprintf(stderr, "ERROR: %s\n", e.what());
      rc[me] = 1;
      world.kill();
    }
  };
  if (np == 1)
    rank_main(0);
  else {
    std::vector<std::thread> th;
    for (int r = 0; r < np; r++) th.emplace_back(rank_main, r);
    for (auto &t : th) t.join();
  }
  for (int r = 0; r < np; r++)
    if (rc[r]) return 1;
  return 0;
}
