// ddhost.cpp -- a C++ host for device-resident multi-GPU runs of both styles (rebomos, aeam): one brick of the box per GPU, one host
// thread per GPU, everything through the C-ABI of include/mdpair_hip.h (no Python, no torch, no MPI).
//
// What the reference gets from LAMMPS on several MPI ranks (USER-REBOMOS/log.rebomos-bulk.4: processor grid :22, thermo
// rows :54-56, "Comm" :67, per-rank Nlocal / Nghost :72-75) -- Verlet::run, fix nve, Comm::exchange / borders /
// forward_comm, thermo -- with the style, the integrator AND the domain decomposition on the devices:
//     setup      mdp_create, mdp_rebomos_read_file / _set_params, mdp_md_setup, mdp_dd_setup, mdp_dd_comm_init
//     per step   mdp_dd_comm_step_begin  (integrate, `neigh_modify check yes` decision from the word that rode in the previous
//                                         step's halo, reneighbor or start the position exchange, interior work)
//                mdp_dd_comm_step_end    (exchange arrives, rest of PairREBOMoS::compute, final half-kick or its deferral)
//     output     mdp_md_thermo + mdp_dd_comm_allreduce on thermo steps
// The ranks are threads of this one process; rank r drives GPU r mod (number of GPUs).  On a box with fewer GPUs than
// ranks RCCL refuses to connect two ranks on one device: MDP_RCCL_LIBRARY=<tests/native/libfake_rccl.so> (the
// repository's test double) lets them share it -- a rehearsal, and the program says so.
//
// usage: ddhost [-style rebomos|aeam] [-ranks N] [-replicate a b c] [-steps K] [-thermo T] [-temp K] [-seed S]
//               [-drift vx vy vz] [-pot file] [-frac2 f] [-dump prefix]
//   -style aeam    USER-AEAM/sample.in's system: fcc Al, a = 4.045, `replicate` = conventional cells per dimension, a fraction
//                  -frac2 (0.0075, sample.in:19) of the atoms switched to type 2 (Si) by this program's own hash of the atom
//                  id (LAMMPS' `set type/fraction` generator is out of scope), skin 1.0 (sample.in:17), AlSi.aeam
//   -dump prefix   every rank writes prefix.<rank>: int nlocal, then nlocal records {int tag; double x[3], v[3]} (tests)
#include "mdpair_hip.h"

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// metal units
constexpr double BOLTZ = 8.617343e-5, MVV2E = 1.0364269e-4, FTM2V = 1.0 / 1.0364269e-4, NKTV2P = 1.6021765e6;

struct Box {
  double lo[3], prd[3], xy, xz, yz;
  double volume() const { return prd[0] * prd[1] * prd[2]; }
  void x2lamda(const double *x, double *l) const
  { // LAMMPS Domain::x2lamda for a restricted triclinic box
    const double d[3] = {x[0] - lo[0], x[1] - lo[1], x[2] - lo[2]};
    l[2] = d[2] / prd[2];
    l[1] = (d[1] - yz * l[2]) / prd[1];
    l[0] = (d[0] - xy * l[1] - xz * l[2]) / prd[0];
  }
};

struct System {
  Box box;
  std::vector<double> x, v; // [n][3]
  std::vector<int> type, tag;
  double mass[3] = {0.0, 95.95, 32.065}; // per type, 1-based (aeam: from the potential file)
  int n() const { return (int) type.size(); }
};

// the 288-atom triclinic 2H-MoS2 cell of in.rebomos-bulk:3-25 (lattice custom ..., origin 0.1, region prism 0 4 0 8 0 1
// tilt -2 0 0), replicated (LAMMPS `replicate`: copies shifted by whole box vectors, tags in creation order)
System build_system(const int rep[3])
{
  const double a1[3] = {3.1903157234, 0.0, 0.0}, a2[3] = {-1.5964590311, 2.7651481541, 0.0}, a3[3] = {0.0, 0.0, 13.9827680588};
  const double basis[6][3] = {{0.0, 0.0, 0.75},           {0.0, 0.0, 0.25},           {2.0 / 3.0, 1.0 / 3.0, 0.862008989},
                              {1.0 / 3.0, 2.0 / 3.0, 0.137990996}, {1.0 / 3.0, 2.0 / 3.0, 0.362008989}, {2.0 / 3.0, 1.0 / 3.0, 0.637991011}};
  const int btype[6] = {1, 1, 2, 2, 2, 2};
  double cmin[3] = {1e300, 1e300, 1e300}, cmax[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int k = 0; k < 2; k++)
        for (int d = 0; d < 3; d++) {
          const double c = i * a1[d] + j * a2[d] + k * a3[d];
          cmin[d] = std::fmin(cmin[d], c);
          cmax[d] = std::fmax(cmax[d], c);
        }
  double lat[3], origin[3];
  for (int d = 0; d < 3; d++) {
    lat[d] = cmax[d] - cmin[d]; // lattice spacings = extent of the unit cell's bounding box (log.rebomos-bulk.1:17)
    origin[d] = 0.1 * lat[d];
  }
  Box cell{{0, 0, 0}, {4 * lat[0], 8 * lat[1], lat[2]}, -2.0 * lat[0], 0.0, 0.0};
  std::vector<double> cx;
  std::vector<int> ct;
  for (int k = -2; k < 3; k++)
    for (int j = -3; j < 12; j++)
      for (int i = -10; i < 14; i++)
        for (int b = 0; b < 6; b++) {
          double p[3], l[3];
          for (int d = 0; d < 3; d++)
            p[d] = (i + basis[b][0]) * a1[d] + (j + basis[b][1]) * a2[d] + (k + basis[b][2]) * a3[d] + origin[d];
          cell.x2lamda(p, l);
          bool in = true;
          for (int d = 0; d < 3; d++) in = in && l[d] >= -1.0e-6 && l[d] < 1.0 - 2.0e-6;
          if (!in) continue;
          cx.insert(cx.end(), p, p + 3);
          ct.push_back(btype[b]);
        }
  System s;
  s.box = Box{{0, 0, 0}, {cell.prd[0] * rep[0], cell.prd[1] * rep[1], cell.prd[2] * rep[2]}, cell.xy * rep[1], 0.0, 0.0};
  const double hx[3] = {cell.prd[0], 0, 0}, hy[3] = {cell.xy, cell.prd[1], 0}, hz[3] = {cell.xz, cell.yz, cell.prd[2]};
  for (int k = 0; k < rep[2]; k++)
    for (int j = 0; j < rep[1]; j++)
      for (int i = 0; i < rep[0]; i++)
        for (size_t a = 0; a < ct.size(); a++) {
          for (int d = 0; d < 3; d++) s.x.push_back(cx[3 * a + d] + i * hx[d] + j * hy[d] + k * hz[d]);
          s.type.push_back(ct[a]);
        }
  s.tag.resize(s.type.size());
  for (size_t a = 0; a < s.tag.size(); a++) s.tag[a] = (int) a + 1;
  s.v.assign(s.x.size(), 0.0);
  return s;
}

// USER-AEAM/sample.in:8-11,19: fcc lattice of rep conventional cells, a fraction of the atoms of type 2
System build_fcc(const int rep[3], double a, double frac2, uint64_t seed)
{
  System s;
  s.box = Box{{0, 0, 0}, {rep[0] * a, rep[1] * a, rep[2] * a}, 0.0, 0.0, 0.0};
  const double basis[4][3] = {{0, 0, 0}, {0.5, 0.5, 0}, {0.5, 0, 0.5}, {0, 0.5, 0.5}};
  for (int k = 0; k < rep[2]; k++)
    for (int j = 0; j < rep[1]; j++)
      for (int i = 0; i < rep[0]; i++)
        for (int b = 0; b < 4; b++) {
          s.x.push_back((i + basis[b][0]) * a);
          s.x.push_back((j + basis[b][1]) * a);
          s.x.push_back((k + basis[b][2]) * a);
          uint64_t h = (seed ^ (0x9E3779B97F4A7C15ull * (uint64_t) (s.type.size() + 1)));
          h ^= h >> 33;
          h *= 0xFF51AFD7ED558CCDull;
          h ^= h >> 33;
          h *= 0xC4CEB9FE1A85EC53ull;
          h ^= h >> 33;
          s.type.push_back((h >> 11) * (1.0 / 9007199254740992.0) < frac2 ? 2 : 1);
        }
  s.tag.resize(s.type.size());
  for (size_t q = 0; q < s.tag.size(); q++) s.tag[q] = (int) q + 1;
  s.v.assign(s.x.size(), 0.0);
  return s;
}

// `velocity all create T seed`-like (own generator: LAMMPS' RanPark is out of scope): Gaussian, zero net momentum, exactly T
void create_velocities(System &s, double temp, uint64_t seed, const double drift[3])
{
  const int n = s.n();
  if (temp > 0.0) {
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
    auto u01 = [&]() {
      st ^= st >> 12;
      st ^= st << 25;
      st ^= st >> 27;
      return ((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
    };
    for (int i = 0; i < n; i++)
      for (int d = 0; d < 3; d++) {
        const double u1 = u01() + 1e-300, u2 = u01();
        s.v[3 * i + d] = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2) / std::sqrt(s.mass[s.type[i]]);
      }
    double p[3] = {0, 0, 0}, mt = 0;
    for (int i = 0; i < n; i++) {
      for (int d = 0; d < 3; d++) p[d] += s.mass[s.type[i]] * s.v[3 * i + d];
      mt += s.mass[s.type[i]];
    }
    double ke = 0;
    for (int i = 0; i < n; i++)
      for (int d = 0; d < 3; d++) {
        s.v[3 * i + d] -= p[d] / mt;
        ke += s.mass[s.type[i]] * s.v[3 * i + d] * s.v[3 * i + d];
      }
    const double t = MVV2E * ke / ((3.0 * n - 3.0) * BOLTZ);
    const double f = std::sqrt(temp / t);
    for (auto &c : s.v) c *= f;
  }
  for (int i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) s.v[3 * i + d] += drift[d];
}

// 1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1 (log.rebomos-bulk.4:22), 8 -> 2x2x2; otherwise the most cubic factorisation
void proc_grid(int n, int g[3])
{
  g[0] = g[1] = g[2] = 1;
  int left = n;
  for (int d = 0; left > 1; d = (d + 1) % 3) {
    int f = 2;
    while (left % f) f++;
    g[d] *= f;
    left /= f;
  }
  // (largest factors first along x)
  for (int a = 0; a < 3; a++)
    for (int b = a + 1; b < 3; b++)
      if (g[b] > g[a]) std::swap(g[a], g[b]);
}

struct Barrier { // (a rank that fails breaks it, so that the others do not wait for it)
  std::mutex m;
  std::condition_variable cv;
  int n, count = 0, gen = 0;
  bool broken = false;
  explicit Barrier(int n_) : n(n_) {}
  void wait()
  {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return;
    const int g = gen;
    if (++count == n) {
      count = 0;
      gen++;
      cv.notify_all();
    } else
      cv.wait(lk, [&] { return gen != g || broken; });
  }
  void fail()
  {
    std::lock_guard<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }
};

struct Args {
  int ranks = 1, rep[3] = {2, 2, 1}, steps = 20, thermo = 10;
  double temp = 0.0, drift[3] = {0, 0, 0};
  uint64_t seed = 4928459;
  int style = 1; // 1 rebomos, 2 aeam
  double frac2 = 0.0075;
  std::string pot, dump;
};

struct Shared {
  Args a;
  System s;
  int grid[3];
  Barrier bar;
  unsigned char uid[128];
  int failed = 0;
  double loop_s = 0.0;
  std::vector<long long> nlocal, nghost;
  long long info[8] = {0};
  explicit Shared(int n) : bar(n), nlocal(n, 0), nghost(n, 0) {}
};

#define CK(call)                                                                                       \
  do {                                                                                                 \
    const int rc_ = (call);                                                                            \
    if (rc_ != MDP_OK) {                                                                               \
      fprintf(stderr, "ERROR on rank %d: %s -> %d: %s\n", rank, #call, rc_, ctx ? mdp_last_error(ctx) : ""); \
      S.failed = 1;                                                                                    \
      S.bar.fail();                                                                                    \
      return;                                                                                          \
    }                                                                                                  \
  } while (0)

void thermo_row(Shared &S, mdp_ctx *ctx, int rank, bool multi, long step)
{
  double t[9];
  CK(mdp_md_thermo(ctx, t));
  double v[8] = {t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]};
  if (multi) CK(mdp_dd_comm_allreduce(ctx, v, 8, 0));
  if (rank != 0) return;
  const double n = S.s.n(), dof = 3.0 * n - 3.0;
  const double temp = 2.0 * v[0] / (dof * BOLTZ);
  const double press = (dof * BOLTZ * temp + v[2] + v[3] + v[4]) / (3.0 * S.s.box.volume()) * NKTV2P;
  printf("%10ld %14.8g %14.8g %14.8g %14.8g\n", step, temp, press, v[1], v[0]);
  fflush(stdout);
}

void rank_main(Shared &S, int rank)
{
  const Args &A = S.a;
  const int N = A.ranks;
  const bool multi = N > 1;
  mdp_ctx *ctx = nullptr;
  const int ndev = mdp_device_count();
  if (ndev < 1) {
    fprintf(stderr, "ERROR: ddhost needs a HIP device; there is no CPU fallback\n");
    S.failed = 1;
    return;
  }
  CK(mdp_create(&ctx, rank % ndev));
  double skin, cutghost;
  mdp_aeam_file *af = nullptr; // (the tables point into it: it lives as long as the context)
  if (A.style == 1) {
    mdp_rebomos_params P;
    char err[512] = {0};
    if (mdp_rebomos_read_file(A.pot.c_str(), &P, err, 512) != MDP_OK) {
      fprintf(stderr, "ERROR: %s\n", err);
      S.failed = 1;
      S.bar.fail();
      return;
    }
    CK(mdp_rebomos_set_params(ctx, &P));
    skin = 2.0;
    cutghost = 3.0 * P.rcmax[0][0] + skin; // cut3rebo + skin (pair_rebomos.cpp:257, log.rebomos-bulk.1:43)
  } else {
    char err[512] = {0};
    if (mdp_aeam_file_read(A.pot.c_str(), &af, err, 512) != MDP_OK) {
      fprintf(stderr, "ERROR: %s\n", err);
      S.failed = 1;
      S.bar.fail();
      return;
    }
    mdp_aeam_tables T;
    const int amap[3] = {0, 0, 1}; // pair_coeff * * AlSi.aeam Al Si: type 1 -> element 0, type 2 -> element 1
    CK(mdp_aeam_file_build(af, 2, amap, &T));
    CK(mdp_aeam_set_tables(ctx, &T));
    skin = 1.0; // sample.in:17
    double cmax = 0.0;
    for (int q = 0; q < T.nelements * T.nelements; q++) cmax = std::fmax(cmax, T.cut[q]);
    cutghost = cmax + skin;
  }
  // this rank's brick of the box (lamda space, as LAMMPS' Comm brick does for triclinic boxes)
  const System &G = S.s;
  std::vector<double> x, v;
  std::vector<int> type, tag;
  for (int i = 0; i < G.n(); i++) {
    double l[3];
    G.box.x2lamda(&G.x[3 * i], l);
    int c[3];
    for (int d = 0; d < 3; d++) {
      l[d] -= std::floor(l[d]);
      c[d] = (int) std::floor(l[d] * S.grid[d]);
      c[d] = c[d] < 0 ? 0 : (c[d] >= S.grid[d] ? S.grid[d] - 1 : c[d]);
    }
    if ((c[0] * S.grid[1] + c[1]) * S.grid[2] + c[2] != rank) continue;
    x.insert(x.end(), &G.x[3 * i], &G.x[3 * i] + 3);
    v.insert(v.end(), &G.v[3 * i], &G.v[3 * i] + 3);
    type.push_back(G.type[i]);
    tag.push_back(G.tag[i]);
  }
  mdp_md_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.style = A.style;
  cfg.nlocal = (int) type.size();
  cfg.ntypes = 2;
  cfg.skin = skin;
  cfg.dt = 0.001;
  cfg.ftm2v = FTM2V;
  cfg.mvv2e = MVV2E;
  // provisional bounds (the library sets the brick's at every reneighboring): the box's Cartesian hull + the ghost shell
  const Box &B = G.box;
  const double hullx0 = B.lo[0] + std::fmin(0.0, B.xy) + std::fmin(0.0, B.xz), hullx1 = B.lo[0] + B.prd[0] + std::fmax(0.0, B.xy) + std::fmax(0.0, B.xz);
  cfg.bbox_lo[0] = hullx0 - cutghost - 2.0;
  cfg.bbox_hi[0] = hullx1 + cutghost + 2.0;
  cfg.bbox_lo[1] = B.lo[1] + std::fmin(0.0, B.yz) - cutghost - 2.0;
  cfg.bbox_hi[1] = B.lo[1] + B.prd[1] + std::fmax(0.0, B.yz) + cutghost + 2.0;
  cfg.bbox_lo[2] = B.lo[2] - cutghost - 2.0;
  cfg.bbox_hi[2] = B.lo[2] + B.prd[2] + cutghost + 2.0;
  const int map[3] = {0, 0, 1}; // type 1 -> Mo, type 2 -> S (pair_coeff * * MoS.REBO.set5b M S)
  const int idummy = 0;
  const double ddummy[3] = {0, 0, 0};
  CK(mdp_md_setup(ctx, &cfg, x.data(), v.data(), type.data(), tag.data(), G.mass, A.style == 1 ? map : nullptr, &idummy, ddummy, &idummy,
                  &idummy));
  mdp_dd_config dd;
  memset(&dd, 0, sizeof dd);
  for (int d = 0; d < 3; d++) {
    dd.boxlo[d] = B.lo[d];
    dd.procgrid[d] = S.grid[d];
  }
  dd.h[0] = B.prd[0];
  dd.h[1] = B.prd[1];
  dd.h[2] = B.prd[2];
  dd.h[3] = B.yz;
  dd.h[4] = B.xz;
  dd.h[5] = B.xy;
  dd.rank = rank;
  dd.cutghost = cutghost;
  CK(mdp_dd_setup(ctx, &dd));
  if (multi) {
    if (rank == 0) {
      char name[256];
      const int dbl = mdp_dd_comm_library(name, 256);
      if (dbl < 0) {
        fprintf(stderr, "ERROR: no RCCL library could be loaded\n");
        S.failed = 1;
      } else {
        printf("RCCL library: %s%s\n", name, dbl == 1 ? "  (TEST DOUBLE: the ranks share GPUs, host-staged -- a rehearsal)" : "");
        if (mdp_dd_comm_unique_id(S.uid) != MDP_OK) S.failed = 1;
      }
    }
    S.bar.wait();
    if (S.failed) return;
    CK(mdp_dd_comm_init(ctx, S.uid));
    CK(mdp_dd_comm_reneighbor(ctx));
  } else
    CK(mdp_dd_reneighbor(ctx));
  if (multi && A.style == 2) { // PairAEAM::compute with its own exchanges, blocking order (pair_aeam.cpp:257,307 + reverse_comm of f)
    CK(mdp_md_aeam_density(ctx, 1));
    CK(mdp_dd_comm_forward_scalar(ctx));
    CK(mdp_md_aeam_force(ctx, 1, 1));
    CK(mdp_dd_comm_reverse(ctx));
  } else
    CK(mdp_md_compute(ctx, 1, 1));
  long long di[8];
  CK(mdp_dd_info(ctx, di, nullptr, nullptr));
  S.nlocal[rank] = di[0];
  S.nghost[rank] = di[1] + di[3];
  S.bar.wait();
  if (S.failed) return;
  if (rank == 0) {
    printf("  %d by %d by %d processor grid (bricks in lamda space, one GPU thread each)\n", S.grid[0], S.grid[1], S.grid[2]);
    printf("%10s %14s %14s %14s %14s\n", "Step", "Temp", "Press", "PotEng", "KinEng");
  }
  thermo_row(S, ctx, rank, multi, 0);
  if (S.failed) return;
  CK(mdp_sync(ctx));
  S.bar.wait();
  if (S.failed) return;
  const auto t0 = std::chrono::steady_clock::now();
  bool pending = false;
  for (long step = 1; step <= A.steps; step++) {
    const int ev = (A.thermo > 0 && step % A.thermo == 0) || step == A.steps ? 1 : 0;
    if (multi) {
      int ren = 0;
      CK(mdp_dd_comm_step_begin(ctx, pending ? 1 : 0, -1, ev, ev, &ren));
      CK(mdp_dd_comm_step_end(ctx, ev, ev, ev ? 0 : 1));
    } else { // one GPU: the deferred on-device `check yes` flag (the answer of the previous step's check)
      int moved = 0, dangerous = 0;
      CK(mdp_md_integrate_check(ctx, pending ? 1 : 0, &moved, &dangerous));
      if (moved) CK(mdp_dd_reneighbor(ctx));
      CK(mdp_md_compute(ctx, ev, ev));
      if (ev) CK(mdp_md_final_integrate(ctx));
      else CK(mdp_md_defer_final(ctx));
    }
    pending = !ev;
    if (ev) {
      thermo_row(S, ctx, rank, multi, step);
      if (S.failed) return;
    }
  }
  CK(mdp_sync(ctx));
  S.bar.wait();
  if (rank == 0) S.loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  CK(mdp_dd_info(ctx, di, nullptr, nullptr));
  S.nlocal[rank] = di[0];
  S.nghost[rank] = di[1] + di[3];
  if (rank == 0) {
    for (int k = 0; k < 8; k++) S.info[k] = 0;
    if (multi) CK(mdp_dd_comm_step_info(ctx, S.info));
    S.info[3] = di[4];
  }
  if (!A.dump.empty()) {
    const int n = (int) di[0];
    std::vector<double> xo(3 * (size_t) n + 3), vo(3 * (size_t) n + 3);
    std::vector<int> to(n + 1);
    CK(mdp_md_download(ctx, xo.data(), vo.data(), nullptr, nullptr));
    CK(mdp_md_download_int(ctx, "tag", to.data()));
    FILE *f = fopen((A.dump + "." + std::to_string(rank)).c_str(), "wb");
    if (f) {
      fwrite(&n, sizeof n, 1, f);
      for (int i = 0; i < n; i++) {
        fwrite(&to[i], sizeof(int), 1, f);
        fwrite(&xo[3 * (size_t) i], sizeof(double), 3, f);
        fwrite(&vo[3 * (size_t) i], sizeof(double), 3, f);
      }
      fclose(f);
    }
  }
  if (multi) CK(mdp_dd_comm_destroy(ctx));
  mdp_destroy(ctx);
  if (af) mdp_aeam_file_free(af);
}

} // namespace

int main(int argc, char **argv)
{
  Args a;
  for (int i = 1; i < argc; i++) {
    const std::string k = argv[i];
    auto need = [&](int n) {
      if (i + n >= argc) {
        fprintf(stderr, "ddhost: %s needs %d value(s)\n", k.c_str(), n);
        exit(2);
      }
    };
    if (k == "-ranks") need(1), a.ranks = atoi(argv[++i]);
    else if (k == "-replicate") need(3), a.rep[0] = atoi(argv[i + 1]), a.rep[1] = atoi(argv[i + 2]), a.rep[2] = atoi(argv[i + 3]), i += 3;
    else if (k == "-steps") need(1), a.steps = atoi(argv[++i]);
    else if (k == "-thermo") need(1), a.thermo = atoi(argv[++i]);
    else if (k == "-temp") need(1), a.temp = atof(argv[++i]);
    else if (k == "-seed") need(1), a.seed = strtoull(argv[++i], nullptr, 10);
    else if (k == "-drift") need(3), a.drift[0] = atof(argv[i + 1]), a.drift[1] = atof(argv[i + 2]), a.drift[2] = atof(argv[i + 3]), i += 3;
    else if (k == "-pot") need(1), a.pot = argv[++i];
    else if (k == "-frac2") need(1), a.frac2 = atof(argv[++i]);
    else if (k == "-style") {
      need(1);
      const std::string v = argv[++i];
      if (v != "rebomos" && v != "aeam") {
        fprintf(stderr, "ddhost: -style rebomos|aeam\n");
        return 2;
      }
      a.style = v == "aeam" ? 2 : 1;
    }
    else if (k == "-dump") need(1), a.dump = argv[++i];
    else {
      fprintf(stderr, "ddhost: unknown option %s\n", k.c_str());
      return 2;
    }
  }
  if (a.ranks < 1 || a.ranks > 64 || a.rep[0] < 1 || a.rep[1] < 1 || a.rep[2] < 1 || a.steps < 0) {
    fprintf(stderr, "ddhost: bad arguments\n");
    return 2;
  }
  if (a.pot.empty()) a.pot = a.style == 1 ? "../tests/golden/potentials/MoS.REBO.set5b" : "../tests/golden/potentials/AlSi.aeam";
  Shared S(a.ranks);
  S.a = a;
  if (a.style == 1)
    S.s = build_system(a.rep);
  else {
    S.s = build_fcc(a.rep, 4.045, a.frac2, 7683797);
    // masses of the file's elements (one read on the main thread: the rank threads read the file again for their tables)
    mdp_aeam_file *f = nullptr;
    char err[512] = {0};
    int ne = 0, nn = 0, na = 0;
    double m[64] = {0};
    if (mdp_aeam_file_read(a.pot.c_str(), &f, err, 512) != MDP_OK) {
      fprintf(stderr, "ERROR: %s\n", err);
      return 1;
    }
    mdp_aeam_file_info(f, &ne, &nn, &na, m, 64, nullptr, 0);
    mdp_aeam_file_free(f);
    if (ne < 2) {
      fprintf(stderr, "ERROR: %s defines %d element(s); -style aeam maps two atom types\n", a.pot.c_str(), ne);
      return 1;
    }
    S.s.mass[1] = m[0];
    S.s.mass[2] = m[1];
  }
  create_velocities(S.s, a.temp, a.seed, a.drift);
  proc_grid(a.ranks, S.grid);
  if (a.style == 1)
    printf("ddhost: REBO-MoS bulk, in.rebomos-bulk cell replicated %d x %d x %d = %d atoms, %d rank(s), %d steps\n", a.rep[0], a.rep[1],
           a.rep[2], S.s.n(), a.ranks, a.steps);
  else
    printf("ddhost: AEAM AlSi, fcc a = 4.045, %d x %d x %d cells = %d atoms (%.2f %% type 2), %d rank(s), %d steps\n", a.rep[0], a.rep[1],
           a.rep[2], S.s.n(), 100.0 * a.frac2, a.ranks, a.steps);
  fflush(stdout);
  std::vector<std::thread> th;
  for (int r = 0; r < a.ranks; r++) th.emplace_back(rank_main, std::ref(S), r);
  for (auto &t : th) t.join();
  if (S.failed) return 1;
  printf("Loop time of %g on %d procs for %d steps with %d atoms\n\n", S.loop_s, a.ranks, a.steps, S.s.n());
  if (S.loop_s > 0 && a.steps > 0)
    printf("Performance: %.3f ns/day, %.3f timesteps/s, %.3f Matom-step/s\n\n", a.steps / S.loop_s * 0.001 * 86.4, a.steps / S.loop_s,
           (double) S.s.n() * a.steps / S.loop_s / 1e6);
  for (int r = 0; r < a.ranks; r++) printf("rank %d: Nlocal %lld  Nghost %lld\n", r, S.nlocal[r], S.nghost[r]);
  static const char *pol[6] = {"split", "lead", "blocking", "first", "inline", "undecided"};
  printf("Neighbor list builds = %lld\nDangerous builds = %lld\n", S.info[3], S.info[4]);
  if (a.ranks > 1) printf("Overlap policy = %s\n", pol[S.info[5] >= 0 && S.info[5] < 5 ? S.info[5] : 5]);
  return 0;
}
