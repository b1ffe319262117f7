// rebomos.hip -- REBO Mo-S hot path for gfx950 (wave64), FP64 throughout.
//
// Replaces PairREBOMoS::REBO_neigh / FREBO / bondorder / FLJ
// (USER-REBOMOS/pair_rebomos.cpp:281-352, 358-447, 571-847, 453-558; pair_rebomos.h:68-211).
//
// The reference walks half of the pairs (tag parity) and scatters forces onto i, j, k, l, including
// ghosts.  A scatter of ~430 FP64 atomics per atom would be bound by the chip's atomic rate, so the
// device formulation is owner-computes and atomic-free:
//
//   E = sum_c E_c,   E_c = sum_{m in N(c)} 1/2 [ V_R(r_cm) + p_cm V_A(r_cm) ]
//   p_cm = [1 + sum_{q in N(c), q != m} w_cq G_c(cos(m,c,q)) + P_c(N_c)]^(-1/2)
//
// which is the reference's energy regrouped by the *centre* atom (b_ij = (p_ij + p_ji)/2).  E_c
// depends only on x_c and the <= ~12 REBO neighbours of c, so
//   1. rebo_centre_kernel<G>: G lanes per centre (owned atoms AND the ghost atoms that neighbour
//      them).  Neighbour geometry is staged in LDS, the O(n^2) angular sums run out of LDS, and the
//      force of E_c on each neighbour slot m is written as one aligned record fnbr[c][m] = {F, e/2};
//      the centre's own share -sum_m F goes to fown[c].
//   2. rebo_lj_tile_kernel: Lennard-Jones over the full list in both directions (no parity rule needed).
//      One workgroup = one tile of 16 two-atom clusters; the union of their neighbourhoods is gathered
//      ONCE into LDS and the cluster rows are 16-bit indices into it (rebo_lj_gather_kernel is the
//      older per-cluster-list form, kept as the fallback when a union does not fit LDS).  The same kernel
//      gathers the REBO forces  F_a = fown[a] + sum_{c in N(a)} fnbr[c][slot of a]  through the
//      reverse-slot table, reduces across lanes with wave shuffles and stores f[a] once.
// Nothing is written to ghost atoms; the explicit virial replaces virial_fdotr_compute.
#include "mdp_common.h"

#include <type_traits>

namespace {

#ifndef MDP_LJ_WAVES
#define MDP_LJ_WAVES 4
#endif
constexpr double kPi = 3.14159265358979323846;
constexpr double kTol = 1.0e-9; // pair_rebomos.cpp:52

__device__ __forceinline__ void wave_lds_fence()
{
  // LDS operations of one wave execute in order; this only stops the compiler from moving
  // LDS reads above the writes of other lanes of the same wave.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the G consecutive lanes of a group, any G <= 16 (the xor butterfly needs a power of two):
// bounded tree towards lane 0 of the group, then broadcast
template <int G> __device__ __forceinline__ double group_sum_any(double v, const int s, const int lane);

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Work items that are neighbours
// in space (consecutive along the Hilbert curve) share most of what they gather, so every XCD is given one
// contiguous stretch of the items instead of every eighth one.
__device__ __forceinline__ int xcd_contiguous(const int b, const int n)
{
  const int q = n >> 3, r = n & 7, xcd = b & 7, idx = b >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

template <int W> __device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int G> __device__ __forceinline__ double group_sum_any(double v, const int s, const int lane)
{
  if constexpr ((G & (G - 1)) == 0) {
    return group_sum<G>(v);
  } else {
    static_assert(G <= 16, "tree below starts at offset 8");
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const double t = __shfl_down(v, o, 64);
      if (s < o && s + o < G) v += t;
    }
    return __shfl(v, lane - s, 64);
  }
}

__device__ __forceinline__ int wave_max_int(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// 1/x to ~1 ulp: hardware seed + two Newton steps (5 instructions instead of the ~12 of an IEEE divide;
// the result feeds products whose tolerance is 1e-9 relative)
__device__ __forceinline__ double fast_rcp(double x)
{
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// 1/sqrt(x) to ~1 ulp: hardware seed + two Newton steps (replaces an IEEE sqrt and an IEEE divide)
__device__ __forceinline__ double rsqrt_nr(double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}

// switching function, pair_rebomos.h:195-211
__device__ __forceinline__ double sp_switch(double r, double rmin, double rinv, double &dw)
{
  const double t = (r - rmin) * rinv;
  if (t <= 0.0) {
    dw = 0.0;
    return 1.0;
  }
  if (t >= 1.0) {
    dw = 0.0;
    return 0.0;
  }
  double s, c;
  sincospi(t, &s, &c);
  dw = -0.5 * kPi * s * rinv;
  return 0.5 * (1.0 + c);
}

// sin(pi u) and cos(pi u) for u in [0, 1/2] (the only range the G(cos) blend needs): fold to [0, 1/4] and
// evaluate the Taylor series in v^2 there (truncation < 5e-17); ~25 instructions instead of the general
// sincospi's argument reduction and quadrant logic.
__device__ __forceinline__ void sincospi_half(const double u, double &sn, double &cs)
{
  const bool fold = u > 0.25;
  const double v = fold ? 0.5 - u : u;
  const double w = v * v;
  constexpr double S[8] = {3.14159265358979312e+00, -5.16771278004996937e+00, 2.55016403987734508e+00, -5.99264529320791883e-01, 8.21458866111281910e-02, -7.37043094571434784e-03, 4.66302805767612337e-04, -2.19153534478302037e-05};
  constexpr double C[9] = {1.00000000000000000e+00, -4.93480220054467900e+00, 4.05871212641676760e+00, -1.33526276885458928e+00, 2.35330630358893123e-01, -2.58068913900140508e-02, 1.92957430940392206e-03, -1.04638104924845650e-04, 4.30306958703294391e-06};
  double ps = S[7], pc = C[8];
#pragma unroll
  for (int k = 6; k >= 0; k--) ps = fma(ps, w, S[k]);
#pragma unroll
  for (int k = 7; k >= 0; k--) pc = fma(pc, w, C[k]);
  ps *= v;
  sn = fold ? pc : ps;
  cs = fold ? ps : pc;
}

__device__ __forceinline__ double poly6(const double *c, double x, double &d)
{
  double g = c[6] * x, dg = 6.0 * c[6] * x;
  g += c[5];
  dg += 5.0 * c[5];
  g *= x;
  dg *= x;
  g += c[4];
  dg += 4.0 * c[4];
  g *= x;
  dg *= x;
  g += c[3];
  dg += 3.0 * c[3];
  g *= x;
  dg *= x;
  g += c[2];
  dg += 2.0 * c[2];
  g *= x;
  dg *= x;
  g += c[1];
  dg += c[1];
  g *= x;
  g += c[0];
  d = dg;
  return g;
}

__device__ __forceinline__ double poly6v(const double *c, double x)
{
  return (((((c[6] * x + c[5]) * x + c[4]) * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0];
}

// G(cos) only (first pass over the neighbour pairs)
__device__ __forceinline__ double gspline_val(const double *cb, const double *cg, double c)
{
  const double gcos = poly6v(cb, c);
  if (c < 0.5) return gcos;
  const double gamma = poly6v(cg, c);
  const double psi = 0.5 * (1.0 - cospi(2.0 * (c - 0.5)));
  return gcos + psi * (gamma - gcos);
}

// G(cos) and dG/dcos, pair_rebomos.h:68-167.  cb/cg: the centre element's b0..b6 / bg0..bg6.
__device__ __forceinline__ double gspline(const double *cb, const double *cg, double c, double &dgdc)
{
  if (c < 0.5) { // caller clamps to [-1,1] (pair_rebomos.cpp:617-618)
    return poly6(cb, c, dgdc);
  }
  double dgcos, dgamma;
  const double gcos = poly6(cb, c, dgcos);
  const double gamma = poly6(cg, c, dgamma);
  // psi = (1 - cos 2 pi u)/2 = sin^2(pi u),  psi' = pi sin 2 pi u = 2 pi sin(pi u) cos(pi u),  u = c - 1/2
  double s, co;
  sincospi_half(c - 0.5, s, co);
  const double psi = s * s;
  const double dpsi = 2.0 * kPi * s * co;
  dgdc = dgcos + dpsi * (gamma - gcos) + psi * (dgamma - dgcos);
  return gcos + psi * (gamma - gcos);
}

// ------------------------------------------------------------------------------------------------
// centre kernel
// ------------------------------------------------------------------------------------------------
// per-slot LDS record of a centre's neighbour: dx dy dz r w dw C p 1/r V_A   (d = x_c - x_m)
constexpr int kRec = 10;
// the fast kernels need neither p nor V_A in LDS: dx dy dz r w dw C 1/r
constexpr int kRecF = 9, kInvF = 7; // stride 9 doubles: 16 lanes reading 16 records touch 16 distinct bank pairs

struct CentreOut {
  double e_acc, v0, v1, v2, v3, v4, v5;
};

// shared epilogue of a slot: radial terms (pair_rebomos.cpp:411-441, 683-725), store, tallies
__device__ __forceinline__ void finish_slot(const RebomosDev &P, const int tc, const int je_m, const int off,
                                            const double dp, const bool owned, const int eflag, const double mx,
                                            const double my, const double mz, const double mr, const double mw,
                                            const double mdw, const double mp, const double mrinv, const double mVA,
                                            double &fx, double &fy, double &fz, const double acc1, const double Csum,
                                            double *__restrict__ fnbr, double &eh, CentreOut &o)
{
  const int tm = ((unsigned) je_m) >> 30;
  const int pt = tc * 2 + tm;
  const double ux = mx * mrinv, uy = my * mrinv, uz = mz * mrinv;
  double radial = (acc1 + Csum * dp) * mdw; // dw_cm [ sum_q C_q G + P'(N) sum_j C_j ]
  double ehalf = 0.0;
  if (mw > kTol) {
    const double ex = exp(-P.alpha[pt] * mr);
    const double pre = mw * P.A[pt] * ex;
    const double VR = pre * (1.0 + P.Q[pt] * mrinv);
    double dVR = pre * (-P.alpha[pt] - P.Q[pt] * mrinv * mrinv - P.Q[pt] * P.alpha[pt] * mrinv);
    const double swl = fast_rcp(mw) * mdw; // (dw/dr)/w
    dVR += VR * swl;
    double dVA = -P.beta[pt] * mVA;
    dVA += mVA * swl;
    radial += 0.5 * (dVR + mp * dVA);
    ehalf = 0.5 * (VR + mp * mVA);
  }
  fx += radial * ux;
  fy += radial * uy;
  fz += radial * uz;
  // slot forces are filed under the neighbour's (static) candidate slot: the gather finds the
  // reverse slot through a table built once per list build
  // one aligned 32-byte record per slot: force on the neighbour and its share of the pair energy
  const int tslot = je_m & 0x3FFFFFFF;
  eh = 0.5 * ehalf;
  reinterpret_cast<double4 *>(fnbr)[off + tslot] = make_double4(fx, fy, fz, eh);
  if (owned) {
    o.e_acc += ehalf;
    // virial of the cluster: sum_m (x_m - x_c) (x) F_m = -sum_m d_m (x) F_m
    o.v0 -= mx * fx;
    o.v1 -= my * fy;
    o.v2 -= mz * fz;
    o.v3 -= mx * fy;
    o.v4 -= mx * fz;
    o.v5 -= my * fz;
  }
}

__device__ __forceinline__ void centre_tally(const CentreOut &o, double *__restrict__ acc, const int eflag,
                                             const int vflag)
{
  // partial sums go to one of MDP_ACC_SLOTS slots (by block) so that a million waves do not queue
  // on seven addresses; acc_reduce_kernel folds the slots afterwards
  const int lane = threadIdx.x & 63;
  double *slot = acc + MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)));
  if (eflag & MDP_EFLAG_GLOBAL) {
    const double e = group_sum<64>(o.e_acc);
    if (lane == 0) atomicAdd(&slot[0], e);
  }
  if (vflag & MDP_VFLAG_GLOBAL) {
    const double v0 = group_sum<64>(o.v0), v1 = group_sum<64>(o.v1), v2 = group_sum<64>(o.v2);
    const double v3 = group_sum<64>(o.v3), v4 = group_sum<64>(o.v4), v5 = group_sum<64>(o.v5);
    if (lane == 0) {
      atomicAdd(&slot[1], v0);
      atomicAdd(&slot[2], v1);
      atomicAdd(&slot[3], v2);
      atomicAdd(&slot[4], v3);
      atomicAdd(&slot[5], v4);
      atomicAdd(&slot[6], v5);
    }
  }
}

// phase A for one candidate: distance test against rcmax (pair_rebomos.cpp:337), ballot compaction into
// the centre's LDS slots in candidate order.  Returns nothing; updates n / active / nsum.
// A candidate a hair outside rcmax (rsq within 1e-14 relative) gets an explicit ZERO slot record.  The gather of an
// owned atom next to a periodic image of a centre reads the slot the centre ITSELF keeps for the atom's image
// (rev_kernel: image centres are not computed at all), and decides with its own distance test, whose operands are
// rounded differently: fl(x_a - fl(x_o + s)) against fl(x_o - fl(x_a - s)).  When the two tests disagree the pair
// sits at rcmax to the last bits, where the switching function -- and with it every term of the record -- is zero.
template <int G, int CAP, int REC>
__device__ __forceinline__ void centre_take(const RebomosDev &P, const int tc, const double4 xc, const bool valid,
                                            const int t, const double4 xj, const int s, const int glane0,
                                            const int base, double *rec, int *je, int &n,
                                            unsigned long long &active, double &nsum, double4 *__restrict__ slot4,
                                            double *__restrict__ vslot)
{
  const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
  const double dx = xc.x - xj.x, dy = xc.y - xj.y, dz = xc.z - xj.z;
  const double rsq = dx * dx + dy * dy + dz * dz;
  const int tj = (int) xj.w;
  const double rc2 = P.rcmaxsq[tc * 2 + tj];
  const bool pred = valid && rsq < rc2;
  if (valid && !pred && rsq < rc2 * (1.0 + 1.0e-14)) { // (practically never taken)
    slot4[t] = make_double4(0.0, 0.0, 0.0, 0.0);
    if (vslot)
      for (int k = 0; k < 6; k++) vslot[6 * (size_t) t + k] = 0.0;
  }
  const unsigned long long bal = __ballot(pred);
  const unsigned long long gb = (bal >> glane0) & gmask;
  const int pos = n + __popcll(gb & ((1ull << s) - 1ull));
  if (pred && pos < CAP) {
    const int pt = tc * 2 + tj;
    const double rinv = rsqrt_nr(rsq); // r and 1/r from one reciprocal square root
    const double r = rsq * rinv;
    double dw;
    const double w = sp_switch(r, P.rcmin[pt], P.rcinv[pt], dw);
    double *q = rec + pos * REC;
    q[0] = dx;
    q[1] = dy;
    q[2] = dz;
    q[3] = r;
    q[4] = w;
    q[5] = dw;
    q[REC == kRec ? 8 : kInvF] = rinv; // the fast kernels' short record keeps 1/r in slot 7
    je[pos] = t | (tj << 30);
    nsum += w; // nM + nS (pair_rebomos.cpp:339-342); only their sum is ever used (:628, h:175)
  }
  n += __popcll(gb);
  if (base < 64) active |= gb << base;
}

template <int G> struct CentreCfg {
  static constexpr int CAP = G;          // fast kernel: every neighbour has its own lane
  static constexpr int GPW = 64 / G;     // centres per wave (G = 12: five, lanes 60..63 idle)
  static constexpr int WPB = G == 12 ? 3 : 4; // waves per block (12-lane groups: 31 KB of LDS per block, 5 blocks/CU)
  static constexpr int STRIDE = CAP * kRecF + 2; // small pad: spreads the groups over the LDS banks
  // {G(cos), G'(cos)} per unordered neighbour pair, circulant layout: the pair (m, m+d mod n) lives at
  // [d-1][m], i.e. every entry is private to the lane that computes it (no index arithmetic, no conflicts)
  static constexpr int MSTRIDE = 2 * (G / 2) * G + 2;
  static constexpr int UA = G >= 32 ? 1 : (G >= 12 ? 2 : 4); // candidate chunks in flight in phase A
};

// slot of the unordered pair (a < b) in the triangular matrix of a G-slot group

// ---- fast centre kernel: G lanes per centre, at most G neighbours (slot m = lane) --------------------
// A centre whose coordination has outgrown its lane group since the last list build is handed to
// rebo_centre_general_kernel through the overflow list.
// LIST: the centres come from a list filled on the device earlier in the same step (the lane-per-centre kernel's
// centres that found a fourth neighbour: centres[-1] holds their number), the candidates from the rows themselves.
// The host sizes the grid from the count it saw a step ago (h_count, a pinned word block 0 refreshes); entries beyond
// what this grid covers are passed on to the general kernel's list, so no count is ever trusted.
template <int G, bool LIST = false>
__global__ __launch_bounds__(64 * CentreCfg<G>::WPB) void rebo_centre_kernel(
    const RebomosDev P, const int *__restrict__ centres, const int ncent_arg, const int nlocal,
    const double4 *__restrict__ xq, const int *__restrict__ cand_off, const int *__restrict__ cand,
    const int *__restrict__ pk, unsigned long long *__restrict__ amask, double *__restrict__ fnbr,
    double *__restrict__ fown, double *__restrict__ acc, int *__restrict__ ovf, const int eflag, const int vflag,
    const int tc /* element of every centre of this launch: a scalar, and with it all per-element constants */)
{
  using C = CentreCfg<G>;
  __shared__ double s_rec[C::WPB * C::GPW * C::STRIDE];
  __shared__ double s_mat[C::WPB * C::GPW * C::MSTRIDE];
  __shared__ int s_je[C::WPB * C::GPW * C::CAP]; // element (bit 30) | candidate slot of the neighbour

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int s = lane % G;          // lane within the centre's group
  const int glane0 = lane - s;     // first lane of the group
  const int gw = lane / G;         // group within the wave (G = 12: the last four lanes form no group)
  const bool lane_ok = gw < C::GPW;
  const int grp_in_block = (tid >> 6) * C::GPW + (lane_ok ? gw : C::GPW - 1);
  const long long gid = (long long) (LIST ? (int) blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x)) * (C::WPB * C::GPW) + grp_in_block;
  const int ncent = LIST ? centres[-1] : ncent_arg;
  const bool have = lane_ok && gid < ncent;
  if (LIST) {
    int *h_count = reinterpret_cast<int *>(const_cast<int *>(pk)); // (LIST: `pk` carries the pinned host word instead)
    if (blockIdx.x == 0 && tid == 0) *h_count = ncent;
    const long long covered = (long long) gridDim.x * (C::WPB * C::GPW);
    for (long long e = covered + (long long) blockIdx.x * blockDim.x + tid; e < ncent; e += (long long) gridDim.x * blockDim.x)
      ovf[1 + atomicAdd(&ovf[0], 1)] = centres[e]; // beyond this grid: the general kernel's list
  }

  double *rec = s_rec + (size_t) grp_in_block * C::STRIDE;
  double *mat = s_mat + (size_t) grp_in_block * C::MSTRIDE;
  int *je = s_je + grp_in_block * C::CAP;

  // The first UA*G candidates of every centre of this class were packed contiguously in class order when the
  // lists were built (pack_cand_kernel), so they are requested together with the centre's id: the head of the
  // kernel is two dependent loads deep (id | candidates -> coordinates) instead of four.
  constexpr int W = C::UA * G;
  int jp[C::UA];
  if (!LIST) {
#pragma unroll
    for (int u = 0; u < C::UA; u++) jp[u] = have ? pk[(size_t) gid * W + u * G + s] : -1;
  }
  int c = 0, off = 0, nc = 0;
  double4 xc = make_double4(0, 0, 0, 0);
  if (have) {
    c = centres[gid];
    off = cand_off[c];
    nc = cand_off[c + 1] - off;
    xc = xq[c];
  }
  if (LIST) {
#pragma unroll
    for (int u = 0; u < C::UA; u++) jp[u] = (have && u * G + s < nc) ? cand[off + u * G + s] : -1;
  }

  // ---- phase A: filter the candidates to the current REBO set (pair_rebomos.cpp:328-344);
  // UA chunks of G candidates are loaded before the first is used
  int n = 0;
  double nsum = 0.0;
  unsigned long long active = 0ull; // bit t: candidate t is inside rcmax right now (group-uniform)
  {
    double4 xj[C::UA];
#pragma unroll
    for (int u = 0; u < C::UA; u++) xj[u] = xq[jp[u] >= 0 ? jp[u] : c];
#pragma unroll
    for (int u = 0; u < C::UA; u++)
      centre_take<G, C::CAP, kRecF>(P, tc, xc, jp[u] >= 0, u * G + s, xj[u], s, glane0, u * G, rec, je, n, active, nsum,
                                    reinterpret_cast<double4 *>(fnbr) + off, nullptr);
  }
  const int ncw = wave_max_int(nc);
  for (int base = W; base < ncw; base += W) { // rows longer than the packed part (rare)
    int jj[C::UA];
    double4 xj[C::UA];
#pragma unroll
    for (int u = 0; u < C::UA; u++) {
      const int t = base + u * G + s;
      jj[u] = t < nc ? cand[off + t] : c;
    }
#pragma unroll
    for (int u = 0; u < C::UA; u++) xj[u] = xq[jj[u]];
#pragma unroll
    for (int u = 0; u < C::UA; u++) {
      const int t = base + u * G + s;
      centre_take<G, C::CAP, kRecF>(P, tc, xc, t < nc, t, xj[u], s, glane0, base + u * G, rec, je, n, active, nsum,
                                    reinterpret_cast<double4 *>(fnbr) + off, nullptr);
    }
  }
  if (have && s == 0) amask[c] = active;
  const bool outgrown = n > C::CAP; // the general kernel takes this centre
  if (outgrown) {
    if (s == 0) ovf[1 + atomicAdd(&ovf[0], 1)] = c;
    n = 0;
  }
  const double Ntot = group_sum_any<G>(nsum, s, lane);
  wave_lds_fence();

  // centre-element constants
  double cb[7], cg[7];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    cb[k] = P.b[tc][k];
    cg[k] = P.bg[tc][k];
  }
  // P(N) and dP/dN, pair_rebomos.h:173-179
  const double ea = exp(-P.a[tc][2] * Ntot);
  const double dp = -P.a[tc][0] + P.a[tc][1] * P.a[tc][2] * ea;
  const double PS = -P.a[tc][0] * (Ntot - 1.0) - P.a[tc][1] * ea + P.a[tc][3];

  const int nw = wave_max_int(n);
  CentreOut o = {0, 0, 0, 0, 0, 0, 0};
  const bool owned = have && c < nlocal;

  const int m = s;
  const bool act = m < n;
  double mx = 0, my = 0, mz = 0, mr = 1, mw = 0, mdw = 0, mri = 1;
  if (act) {
    const double *q = rec + m * kRecF;
    mx = q[0];
    my = q[1];
    mz = q[2];
    mr = q[3];
    mw = q[4];
    mdw = q[5];
    mri = q[kInvF];
  }
  const double ux = mx * mri, uy = my * mri, uz = mz * mri;
  // -- every unordered pair (m,q) ONCE: circulant enumeration q = (m+d) mod n, d = 1..n/2 (for even n
  //    the distance-n/2 pairs are taken by the lower half of the lanes).  The lane of m evaluates G, G' and
  //    keeps them in its private LDS column; what the pair contributes to the partner q is handed over with
  //    a lane shuffle (lane q reads from lane q-d).  S_m = sum_q w_q G(m,q) (pair_rebomos.cpp:607-630) is
  //    complete when the loop ends -- no second pass over the pairs.
  const int lane_base = (threadIdx.x & 63) - s;
  double S = 0.0;
  for (int d = 1; d <= nw / 2; d++) {
    const bool mine = act && (2 * d < n || (2 * d == n && m < d));
    double give = 0.0; // w_m G(m,q): the partner's share
    if (mine) {
      int qi = m + d;
      qi = qi >= n ? qi - n : qi;
      const double *q = rec + qi * kRecF;
      double cs = (ux * q[0] + uy * q[1] + uz * q[2]) * q[kInvF];
      cs = fmin(cs, 1.0);
      cs = fmax(cs, -1.0);
      double dg, g;
      // (wave-uniform test: the blend of the two polynomials is skipped when no pair of this trip has cos >= 1/2)
      if (__any(cs >= 0.5)) g = gspline(cb, cg, cs, dg);
      else g = poly6(cb, cs, dg);
      double *e = mat + 2 * ((d - 1) * G + m);
      e[0] = g;
      e[1] = dg;
      S += q[4] * g;
      give = mw * g;
    }
    int src = m - d;
    src = src < 0 ? src + n : src;
    const double got = __shfl(give, lane_base + (act ? src : m), 64);
    S += act ? got : 0.0;
  }
  // -- phase B: p_cm, C_m (pair_rebomos.cpp:607-630)
  double mC = 0.0, mp = 0.0, mVA = 0.0;
  if (act) {
    const int pt = tc * 2 + (((unsigned) je[m]) >> 30);
    mp = rsqrt_nr(1.0 + S + PS);
    mVA = -mw * P.B[pt] * exp(-P.beta[pt] * mr);
    mC = (mw > kTol) ? 0.5 * mVA * (-0.5 * mp * mp * mp) : 0.0;
    rec[m * kRecF + 6] = mC;
  }
  const double Csum = group_sum_any<G>(mC, s, lane);
  wave_lds_fence();
  // -- phase C: forces on the slots (pair_rebomos.cpp:634-725), again once per unordered pair: the lane of
  //    m adds its own part and ships the partner's part (force on q and C_m G for q's radial term)
  double fx = 0, fy = 0, fz = 0, acc1 = 0;
  for (int d = 1; d <= nw / 2; d++) {
    const bool mine = act && (2 * d < n || (2 * d == n && m < d));
    double sx = 0, sy = 0, sz = 0, sa = 0;
    if (mine) {
      int qi = m + d;
      qi = qi >= n ? qi - n : qi;
      const double *q = rec + qi * kRecF;
      const double qrinv = q[kInvF];
      const double qx = q[0] * qrinv, qy = q[1] * qrinv, qz = q[2] * qrinv;
      double cs = ux * qx + uy * qy + uz * qz;
      cs = fmin(cs, 1.0);
      cs = fmax(cs, -1.0);
      const double *e = mat + 2 * ((d - 1) * G + m);
      const double g = e[0], dg = e[1];
      // (C_m w_q + C_q w_m) G'(cos) d cos / d x ; d cos/d x_m = -(u_q - cos u_m)/r_m ; force = -gradient
      const double common = (mC * q[4] + q[6] * mw) * dg;
      const double cm = common * mri, cq = common * qrinv;
      fx += cm * (qx - cs * ux);
      fy += cm * (qy - cs * uy);
      fz += cm * (qz - cs * uz);
      acc1 += q[6] * g;
      sx = cq * (ux - cs * qx);
      sy = cq * (uy - cs * qy);
      sz = cq * (uz - cs * qz);
      sa = mC * g;
    }
    int src = m - d;
    src = src < 0 ? src + n : src;
    const int from = lane_base + (act ? src : m);
    const double rx = __shfl(sx, from, 64), ry = __shfl(sy, from, 64), rz = __shfl(sz, from, 64);
    const double ra = __shfl(sa, from, 64);
    if (act) {
      fx += rx;
      fy += ry;
      fz += rz;
      acc1 += ra;
    }
  }
  double eh = 0.0;
  if (act)
    finish_slot(P, tc, je[m], off, dp, owned, eflag, mx, my, mz, mr, mw, mdw, mp, mri, mVA, fx, fy, fz, acc1, Csum,
                fnbr, eh, o);
  // the centre's own share: minus the sum of its slot forces, plus the centre halves of the pair energies.
  // Written once here so that the gather only has to follow the reverse slots.
  const double ox = group_sum_any<G>(act ? fx : 0.0, s, lane), oy = group_sum_any<G>(act ? fy : 0.0, s, lane), oz = group_sum_any<G>(act ? fz : 0.0, s, lane);
  const double oe = (eflag & MDP_EFLAG_ATOM) ? group_sum_any<G>(eh, s, lane) : 0.0;
  if (owned && s == 0 && !outgrown) reinterpret_cast<double4 *>(fown)[c] = make_double4(-ox, -oy, -oz, oe);
  centre_tally(o, acc, eflag, vflag);
}

// ---- centres with at most three neighbours: ONE LANE per centre ------------------------------------------
// An S atom of MoS2 has three Mo-S bonds and nothing else inside rcmax, and S centres are two thirds of all
// centres.  In 4-lane groups a wave served 16 of them: ~600 instructions for 48 bonds, most lanes idle in most of
// them, two dependent load levels in front -- 43 % VALU issue, 61 % of the wave cycles waiting (round-3 counters).
// Here a lane walks the packed candidates of its centre by itself (16 gathers in flight per lane), keeps the three
// neighbours in registers (static indexing: the k-th neighbour's atom and slot are picked up with selects, its
// geometry is then fetched again -- an L1 hit), and evaluates the three pairs straight-line: no LDS, no shuffles,
// 64 centres per wave.  A centre that finds a fourth neighbour goes to the overflow list like any centre that
// outgrows its lane group; the classification keeps centres with a candidate near rcmax out of this class.
constexpr int kC3Block = 256; // (64-thread blocks measured the same: the kernel is bound by memory traffic)
__global__ __launch_bounds__(kC3Block) void rebo_centre3_kernel(
    const RebomosDev P, const int *__restrict__ centres, const int ncent, const int nlocal,
    const double4 *__restrict__ xq, const int *__restrict__ cand_off, const int *__restrict__ cand,
    const int *__restrict__ pk, unsigned long long *__restrict__ amask, double *__restrict__ fnbr,
    double *__restrict__ fown, double *__restrict__ acc, int *__restrict__ ovf3 /* count at [-1] */,
    int *__restrict__ ovf_general /* or null */, const int eflag, const int vflag, const int tc)
{
  constexpr int W = CentreCfg<4>::UA * 4; // packed candidates per centre of this class (pack_cand_kernel)
  static_assert(W == 16, "four int4 per centre");
  // (blocks are dealt round-robin to the 8 XCDs, each with its own L2: every XCD takes one contiguous stretch of the
  //  class list -- atoms in curve order -- so that the neighbours its centres gather are shared within ITS L2)
  const long long gid = (long long) xcd_contiguous(blockIdx.x, gridDim.x) * kC3Block + threadIdx.x;
  const bool have = gid < ncent;
  int jp[W];
  {
    const int4 *__restrict__ p4 = reinterpret_cast<const int4 *>(pk) + (have ? gid : 0) * (W / 4);
#pragma unroll
    for (int q = 0; q < W / 4; q++) {
      const int4 v = p4[q];
      jp[4 * q] = have ? v.x : -1;
      jp[4 * q + 1] = have ? v.y : -1;
      jp[4 * q + 2] = have ? v.z : -1;
      jp[4 * q + 3] = have ? v.w : -1;
    }
  }
  int c = 0, off = 0, nc = 0;
  double4 xc = make_double4(0, 0, 0, 0);
  if (have) {
    c = centres[gid];
    off = cand_off[c];
    nc = cand_off[c + 1] - off;
    xc = xq[c];
  }
  double4 *__restrict__ slot4 = reinterpret_cast<double4 *>(fnbr) + off;
  const double rc2_0 = P.rcmaxsq[tc * 2], rc2_1 = P.rcmaxsq[tc * 2 + 1];

  // ---- phase A: which candidates are inside rcmax now (pair_rebomos.cpp:328-344); the first three are kept
  int n = 0, nj0 = 0, nj1 = 0, nj2 = 0, ts0 = 0, ts1 = 0, ts2 = 0;
  unsigned long long active = 0ull;
  auto take = [&](const int t, const int j, const bool valid, const double4 xj) {
    const double dx = xc.x - xj.x, dy = xc.y - xj.y, dz = xc.z - xj.z;
    const double rsq = dx * dx + dy * dy + dz * dz;
    const double rc2 = ((int) xj.w) ? rc2_1 : rc2_0;
    const bool pred = valid && rsq < rc2;
    if (valid && !pred && rsq < rc2 * (1.0 + 1.0e-14)) slot4[t] = make_double4(0.0, 0.0, 0.0, 0.0); // (see centre_take)
    const bool s0 = pred && n == 0, s1 = pred && n == 1, s2 = pred && n == 2;
    nj0 = s0 ? j : nj0;
    nj1 = s1 ? j : nj1;
    nj2 = s2 ? j : nj2;
    ts0 = s0 ? t : ts0;
    ts1 = s1 ? t : ts1;
    ts2 = s2 ? t : ts2;
    n += pred ? 1 : 0;
    if (pred && t < 64) active |= 1ull << t;
  };
#pragma unroll
  for (int h = 0; h < 2; h++) { // two batches of eight gathers
    double4 xj[W / 2];
#pragma unroll
    for (int u = 0; u < W / 2; u++) xj[u] = xq[jp[h * (W / 2) + u] >= 0 ? jp[h * (W / 2) + u] : c];
#pragma unroll
    for (int u = 0; u < W / 2; u++) take(h * (W / 2) + u, jp[h * (W / 2) + u], jp[h * (W / 2) + u] >= 0, xj[u]);
  }
  for (int t = W; __any(t < nc); t++) { // rows longer than the packed part (rare)
    const bool valid = t < nc;
    const int j = valid ? cand[off + t] : c;
    take(t, j, valid, xq[j]);
  }
  if (have) amask[c] = active;
  const bool outgrown = n > 3; // a fourth neighbour: the 8-lane-group kernel takes this centre right after this launch
  if (outgrown) {
    const int at = atomicAdd(&ovf3[-1], 1); // (counted in either mode: the host watches this number)
    if (ovf_general) ovf_general[1 + atomicAdd(&ovf_general[0], 1)] = c; // no 8-lane launch behind this one: general kernel
    else ovf3[at] = c;
    n = 0;
  }

  // ---- the (at most) three neighbours, in registers
  int nj[3] = {nj0, nj1, nj2}, ts[3] = {ts0, ts1, ts2};
  double dx[3], dy[3], dz[3], r[3], w[3], dw[3], ri[3], ux[3], uy[3], uz[3];
  int tj[3];
  bool act[3];
  double Ntot = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    act[k] = k < n;
    const double4 xj = xq[act[k] ? nj[k] : c];
    tj[k] = act[k] ? (int) xj.w : 0;
    const double ex = xc.x - xj.x, ey = xc.y - xj.y, ez = xc.z - xj.z;
    const double rsq = act[k] ? ex * ex + ey * ey + ez * ez : 1.0;
    const int pt = tc * 2 + tj[k];
    ri[k] = rsqrt_nr(rsq); // r and 1/r from one reciprocal square root
    r[k] = rsq * ri[k];
    double dwk = 0.0;
    const double wk = sp_switch(r[k], P.rcmin[pt], P.rcinv[pt], dwk);
    w[k] = act[k] ? wk : 0.0;
    dw[k] = act[k] ? dwk : 0.0;
    dx[k] = act[k] ? ex : 0.0;
    dy[k] = act[k] ? ey : 0.0;
    dz[k] = act[k] ? ez : 0.0;
    ux[k] = dx[k] * ri[k];
    uy[k] = dy[k] * ri[k];
    uz[k] = dz[k] * ri[k];
    Ntot += w[k]; // nM + nS (pair_rebomos.cpp:339-342)
  }
  double cb[7], cg[7];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    cb[k] = P.b[tc][k];
    cg[k] = P.bg[tc][k];
  }
  // P(N) and dP/dN, pair_rebomos.h:173-179
  const double ea = exp(-P.a[tc][2] * Ntot);
  const double dp = -P.a[tc][0] + P.a[tc][1] * P.a[tc][2] * ea;
  const double PS = -P.a[tc][0] * (Ntot - 1.0) - P.a[tc][1] * ea + P.a[tc][3];

  // ---- the three unordered pairs (0,1) (0,2) (1,2): cos, G, G' once each (pair_rebomos.cpp:607-630)
  double cs[3], g[3], dg[3], S[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const int m = e == 2 ? 1 : 0, q = e == 0 ? 1 : 2;
    double cv = ux[m] * ux[q] + uy[m] * uy[q] + uz[m] * uz[q];
    cv = fmin(cv, 1.0);
    cv = fmax(cv, -1.0);
    cs[e] = cv;
    double d1;
    // (wave-uniform test: the blend of the two polynomials only when some centre of the wave has cos >= 1/2)
    if (__any(cv >= 0.5)) g[e] = gspline(cb, cg, cv, d1); // (wave-uniform: the blend only when some centre needs it)
    else g[e] = poly6(cb, cv, d1);
    dg[e] = d1;
    const bool both = act[m] && act[q];
    S[m] += both ? w[q] * g[e] : 0.0;
    S[q] += both ? w[m] * g[e] : 0.0;
  }
  double p[3], VA[3], C[3], Csum = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int pt = tc * 2 + tj[k];
    p[k] = rsqrt_nr(1.0 + S[k] + PS);
    VA[k] = -w[k] * P.B[pt] * exp(-P.beta[pt] * r[k]);
    C[k] = (act[k] && w[k] > kTol) ? 0.5 * VA[k] * (-0.5 * p[k] * p[k] * p[k]) : 0.0;
    Csum += C[k];
  }
  // ---- forces on the slots (pair_rebomos.cpp:634-725)
  double fx[3] = {0, 0, 0}, fy[3] = {0, 0, 0}, fz[3] = {0, 0, 0}, a1[3] = {0, 0, 0};
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const int m = e == 2 ? 1 : 0, q = e == 0 ? 1 : 2;
    if (!(act[m] && act[q])) continue;
    // (C_m w_q + C_q w_m) G'(cos) d cos / d x ; d cos/d x_m = -(u_q - cos u_m)/r_m ; force = -gradient
    const double common = (C[m] * w[q] + C[q] * w[m]) * dg[e];
    const double cm = common * ri[m], cq = common * ri[q];
    fx[m] += cm * (ux[q] - cs[e] * ux[m]);
    fy[m] += cm * (uy[q] - cs[e] * uy[m]);
    fz[m] += cm * (uz[q] - cs[e] * uz[m]);
    fx[q] += cq * (ux[m] - cs[e] * ux[q]);
    fy[q] += cq * (uy[m] - cs[e] * uy[q]);
    fz[q] += cq * (uz[m] - cs[e] * uz[q]);
    a1[m] += C[q] * g[e];
    a1[q] += C[m] * g[e];
  }
  CentreOut o = {0, 0, 0, 0, 0, 0, 0};
  const bool owned = have && c < nlocal;
  double ox = 0.0, oy = 0.0, oz = 0.0, oe = 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (!act[k]) continue;
    double eh = 0.0;
    finish_slot(P, tc, ts[k] | (tj[k] << 30), off, dp, owned, eflag, dx[k], dy[k], dz[k], r[k], w[k], dw[k], p[k], ri[k],
                VA[k], fx[k], fy[k], fz[k], a1[k], Csum, fnbr, eh, o);
    ox += fx[k];
    oy += fy[k];
    oz += fz[k];
    oe += eh;
  }
  // the centre's own share: minus the sum of its slot forces, plus the centre halves of the pair energies
  if (owned && !outgrown)
    reinterpret_cast<double4 *>(fown)[c] = make_double4(-ox, -oy, -oz, (eflag & MDP_EFLAG_ATOM) ? oe : 0.0);
  centre_tally(o, acc, eflag, vflag);
}

// ---- general centre kernel: 32 lanes per centre, up to 64 neighbours, no pair matrix ------------------
// Works through the overflow list of the fast kernels (normally empty).
// VATOM: also the per-atom virial in the reference's split (v_tally3 thirds, v_tally2 / ev_tally halves,
// pair_rebomos.cpp:444,707-711,725): the centre's share goes to vatom[c], every neighbour's share to
// vslot[c][slot] for the gather.  All centres take this kernel on the (rare) steps that ask for it.
template <bool VATOM>
__global__ __launch_bounds__(256) void rebo_centre_general_kernel(
    const RebomosDev P, const int *__restrict__ list, const int list_count, const int nlocal,
    const double4 *__restrict__ xq, const int *__restrict__ cand_off, const int *__restrict__ cand,
    unsigned long long *__restrict__ amask, double *__restrict__ fnbr, double *__restrict__ fown,
    double *__restrict__ vslot, double *__restrict__ vatom, double *__restrict__ acc, int *__restrict__ flags,
    const int eflag, const int vflag, int *__restrict__ h_cnt4 = nullptr, const int ovf_stride = 0,
    int *__restrict__ h_gen = nullptr)
{
  constexpr int G = 32, CAP = 64, STRIDE = CAP * kRec + 2;
  // (the last centre kernel of a step: publishes how many centres outgrew the lane-per-centre kernel, per list, and how
  //  many this kernel itself was handed -- pinned words, statistics and grid sizes of later computes)
  if (h_cnt4 && list_count < 0 && blockIdx.x == 0 && threadIdx.x < 4) h_cnt4[threadIdx.x] = list[(size_t) (threadIdx.x + 1) * ovf_stride];
  if (h_gen && list_count < 0 && blockIdx.x == 0 && threadIdx.x == 4) *h_gen = list[0];
  __shared__ double s_rec[8 * STRIDE];
  __shared__ int s_je[8 * CAP];
  // list_count < 0: overflow list of the fast kernels, {count, ids...}; else an explicit list of centres
  const int ncent = list_count < 0 ? list[0] : list_count;
  const int *ovf = list_count < 0 ? list : list - 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int s = lane % G;
  const int glane0 = lane - s;
  const int grp_in_block = tid / G;
  double *rec = s_rec + (size_t) grp_in_block * STRIDE;
  int *je = s_je + grp_in_block * CAP;
  CentreOut o = {0, 0, 0, 0, 0, 0, 0};
  // wave-uniform trip count: both groups of a wave iterate together
  for (long long g0 = (long long) blockIdx.x * 8; g0 < ncent; g0 += (long long) gridDim.x * 8) {
    const long long gid = g0 + grp_in_block;
    const bool have = gid < ncent;
    int c = 0, off = 0, nc = 0, tc = 0;
    double4 xc = make_double4(0, 0, 0, 0);
    if (have) {
      c = ovf[1 + gid];
      off = cand_off[c];
      nc = cand_off[c + 1] - off;
      xc = xq[c];
      tc = (int) xc.w;
    }
    int n = 0;
    double nsum = 0.0;
    unsigned long long active = 0ull;
    const int ncw = wave_max_int(nc);
    for (int base = 0; base < ncw; base += G) {
      const int t = base + s;
      const int j = t < nc ? cand[off + t] : c;
      const double4 xj = xq[j];
      centre_take<G, CAP, kRec>(P, tc, xc, t < nc, t, xj, s, glane0, base, rec, je, n, active, nsum,
                                reinterpret_cast<double4 *>(fnbr) + off, VATOM ? vslot + 6 * (size_t) off : nullptr);
    }
    if (n > CAP) {
      if (s == 0) atomicOr(&flags[0], 1);
      n = CAP;
    }
    if (have && s == 0) amask[c] = active;
    const double Ntot = group_sum<G>(nsum);
    wave_lds_fence();
    double cb[7], cg[7];
#pragma unroll
    for (int k = 0; k < 7; k++) {
      cb[k] = P.b[tc][k];
      cg[k] = P.bg[tc][k];
    }
    const double ea = exp(-P.a[tc][2] * Ntot);
    const double dp = -P.a[tc][0] + P.a[tc][1] * P.a[tc][2] * ea;
    const double PS = -P.a[tc][0] * (Ntot - 1.0) - P.a[tc][1] * ea + P.a[tc][3];
    const int nw = wave_max_int(n);
    const bool owned = have && c < nlocal;
    double csum_part = 0.0;
    for (int mb = 0; mb < nw; mb += G) {
      const int m = mb + s;
      const bool act = m < n;
      double mx = 0, my = 0, mz = 0, mr = 1, mw = 0, mri = 1;
      if (act) {
        const double *q = rec + m * kRec;
        mx = q[0];
        my = q[1];
        mz = q[2];
        mr = q[3];
        mw = q[4];
        mri = q[8];
      }
      const double ux = mx * mri, uy = my * mri, uz = mz * mri;
      double S = 0.0;
      for (int qi = 0; qi < nw; qi++) {
        if (act && qi < n && qi != m) {
          const double *q = rec + qi * kRec;
          double cs = (ux * q[0] + uy * q[1] + uz * q[2]) * q[8];
          cs = fmin(cs, 1.0);
          cs = fmax(cs, -1.0);
          S += q[4] * gspline_val(cb, cg, cs);
        }
      }
      if (act) {
        const int pt = tc * 2 + (((unsigned) je[m]) >> 30);
        const double p = 1.0 / sqrt(1.0 + S + PS);
        const double VA = -mw * P.B[pt] * exp(-P.beta[pt] * mr);
        const double Cm = (mw > kTol) ? 0.5 * VA * (-0.5 * p * p * p) : 0.0;
        double *q = rec + m * kRec;
        q[6] = Cm;
        q[7] = p;
        q[9] = VA;
        csum_part += Cm;
      }
    }
    const double Csum = group_sum<G>(csum_part);
    wave_lds_fence();
    double ox = 0, oy = 0, oz = 0, oe = 0; // this lane's part of the centre's own share
    for (int mb = 0; mb < nw; mb += G) {
      const int m = mb + s;
      const bool act = m < n;
      double mx = 0, my = 0, mz = 0, mr = 1, mw = 0, mdw = 0, mC = 0, mp = 0, mri = 1, mVA = 0;
      if (act) {
        const double *q = rec + m * kRec;
        mx = q[0];
        my = q[1];
        mz = q[2];
        mr = q[3];
        mw = q[4];
        mdw = q[5];
        mC = q[6];
        mp = q[7];
        mri = q[8];
        mVA = q[9];
      }
      const double ux = mx * mri, uy = my * mri, uz = mz * mri;
      double fx = 0, fy = 0, fz = 0, acc1 = 0;
      double vs[6] = {0, 0, 0, 0, 0, 0}, vc[6] = {0, 0, 0, 0, 0, 0}; // slot share / centre share (VATOM)
      for (int qi = 0; qi < nw; qi++) {
        if (act && qi < n && qi != m) {
          const double *q = rec + qi * kRec;
          const double qrinv = q[8];
          double cs = (ux * q[0] + uy * q[1] + uz * q[2]) * qrinv;
          cs = fmin(cs, 1.0);
          cs = fmax(cs, -1.0);
          double dg;
          const double g = gspline(cb, cg, cs, dg);
          const double coef = (mC * q[4] + q[6] * mw) * dg * mri;
          const double qx = q[0] * qrinv, qy = q[1] * qrinv, qz = q[2] * qrinv;
          fx += coef * (qx - cs * ux);
          fy += coef * (qy - cs * uy);
          fz += coef * (qz - cs * uz);
          acc1 += q[6] * g;
          if (VATOM) {
            // triplet (bond m, third body q) and its mirror (bond q, third body m): v_tally3(i,j,k,fj,fk,rji,rki)
            const double third = 1.0 / 3.0;
            const double am[3] = {ux - cs * qx, uy - cs * qy, uz - cs * qz}; // (u_m - cos u_q)
            const double aq[3] = {qx - cs * ux, qy - cs * uy, qz - cs * uz}; // (u_q - cos u_m)
            const double um[3] = {ux, uy, uz}, uq[3] = {qx, qy, qz};
            const double dm[3] = {mx, my, mz}, dq[3] = {q[0], q[1], q[2]};
            const double T1 = mC * q[4] * dg, R1 = mC * q[5] * (g + dp);   // bond m: on m / on q
            const double T2 = q[6] * mw * dg, R2 = q[6] * mdw * (g + dp);  // bond q: on q / on m
            double fj[3], fk[3], gj[3], gk[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
              fj[d] = T1 * aq[d] * mri;                    // on m from bond m
              fk[d] = T1 * am[d] * qrinv + R1 * uq[d];     // on q from bond m
              gj[d] = T2 * am[d] * qrinv;                  // on q from bond q
              gk[d] = T2 * aq[d] * mri + R2 * um[d];       // on m from bond q
            }
            const int ia[6] = {0, 1, 2, 0, 0, 1}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
            for (int k6 = 0; k6 < 6; k6++) {
              const double v1 = -dm[ia[k6]] * fj[ib[k6]] - dq[ia[k6]] * fk[ib[k6]]; // bond m, third q
              const double v2 = -dq[ia[k6]] * gj[ib[k6]] - dm[ia[k6]] * gk[ib[k6]]; // bond q, third m
              vs[k6] += third * (v1 + v2);
              vc[k6] += third * v1;
            }
          }
        }
      }
      if (act) {
        double eh = 0.0;
        finish_slot(P, tc, je[m], off, dp, owned, eflag, mx, my, mz, mr, mw, mdw, mp, mri, mVA, fx, fy, fz, acc1,
                    Csum, fnbr, eh, o);
        ox += fx;
        oy += fy;
        oz += fz;
        oe += eh;
        if (VATOM) {
          // v_tally2(i,j,tmp2,rij) with tmp2 = -C_m P' dw_m / r_m and ev_tally's pair part with this
          // centre's half of fpair: both give half to either end
          const int pt = tc * 2 + (((unsigned) je[m]) >> 30);
          double fh = -mC * dp * mdw * mri;
          if (mw > kTol) {
            const double ex = exp(-P.alpha[pt] * mr);
            const double pre = mw * P.A[pt] * ex;
            const double VR = pre * (1.0 + P.Q[pt] * mri);
            double dVR = pre * (-P.alpha[pt] - P.Q[pt] * mri * mri - P.Q[pt] * P.alpha[pt] * mri);
            dVR += VR / mw * mdw;
            double dVA = -P.beta[pt] * mVA;
            dVA += mVA / mw * mdw;
            fh += -0.5 * (dVR + mp * dVA) * mri;
          }
          const double dm[3] = {mx, my, mz};
          const int ia[6] = {0, 1, 2, 0, 0, 1}, ib[6] = {0, 1, 2, 1, 2, 2};
          const int tslot = je[m] & 0x3FFFFFFF;
#pragma unroll
          for (int k6 = 0; k6 < 6; k6++) {
            const double vh = 0.5 * dm[ia[k6]] * dm[ib[k6]] * fh;
            vs[k6] += vh;
            vc[k6] += vh;
            vslot[6 * (size_t) (off + tslot) + k6] = vs[k6];
          }
        }
      }
      if (VATOM) { // the centre's own share: sum over the slots of this group
#pragma unroll
        for (int k6 = 0; k6 < 6; k6++) {
          const double t = group_sum<G>(act ? vc[k6] : 0.0);
          if (owned && s == 0) vatom[6 * (size_t) c + k6] += t;
        }
      }
    }
    ox = group_sum<G>(ox);
    oy = group_sum<G>(oy);
    oz = group_sum<G>(oz);
    oe = group_sum<G>(oe);
    if (owned && s == 0) reinterpret_cast<double4 *>(fown)[c] = make_double4(-ox, -oy, -oz, oe);
    wave_lds_fence();
  }
  centre_tally(o, acc, eflag, vflag);
}

// ------------------------------------------------------------------------------------------------
// Lennard-Jones over the cluster pair list + gather of the REBO slot forces
// ------------------------------------------------------------------------------------------------
// The per-atom gather x[j] is what bounds this loop (one 32-byte gather feeds ~25 flops and the L1
// tag rate saturates), so MDP_CLUSTER consecutive (Morton-ordered, hence adjacent) atoms share ONE
// union neighbour list: every gathered x[j] is tested against all four cluster atoms from registers.

// loop-invariant Lennard-Jones parameters of one (cluster atom, neighbour element) pair type
struct LJPar {
  double lo, hi, sw; // rsq windows (pair_rebomos.cpp:518-532 decided on rij, reproduced in rsq space)
  double c1, c2, c3, c4; // lj1..lj4
};

__device__ __forceinline__ LJPar lj_load(const RebomosDev &P, const int pt)
{
  LJPar q;
  q.lo = P.lj_rsq_lo[pt];
  q.hi = P.lj_rsq_hi[pt];
  q.sw = P.lj_rsq_sw[pt];
  q.c1 = P.lj1[pt];
  q.c2 = P.lj2[pt];
  q.c3 = P.lj3[pt];
  q.c4 = P.lj4[pt];
  return q;
}

// the same parameters picked from the two uniform (scalar-register) candidates by the atom's element: no
// vector memory instruction, so nothing but the index prefetch ever counts on vmcnt inside the tile loop
__device__ __forceinline__ LJPar lj_select(const RebomosDev &P, const int ta, const int seg)
{
  LJPar q;
  q.lo = ta ? P.lj_rsq_lo[2 + seg] : P.lj_rsq_lo[seg];
  q.hi = ta ? P.lj_rsq_hi[2 + seg] : P.lj_rsq_hi[seg];
  q.sw = ta ? P.lj_rsq_sw[2 + seg] : P.lj_rsq_sw[seg];
  q.c1 = ta ? P.lj1[2 + seg] : P.lj1[seg];
  q.c2 = ta ? P.lj2[2 + seg] : P.lj2[seg];
  q.c3 = ta ? P.lj3[2 + seg] : P.lj3[seg];
  q.c4 = ta ? P.lj4[2 + seg] : P.lj4[seg];
  return q;
}

template <bool EV, int SEG = -1>
__device__ __forceinline__ void lj_pair(const RebomosDev &P, const LJPar &q, const int pt, const double4 &xa,
                                        const double4 &xj, double &fx, double &fy, double &fz, double &e,
                                        const int vflag, double &v0, double &v1, double &v2, double &v3, double &v4,
                                        double &v5)
{
  const double dx = xa.x - xj.x, dy = xa.y - xj.y, dz = xa.z - xj.z;
  const double rsq = dx * dx + dy * dy + dz * dz;
  // FLJ windows (pair_rebomos.cpp:518-543).  The 12-6 branch is evaluated without divergence and
  // selected by the window predicate, so the independent evaluations of a lane interleave; the cubic
  // inner spline (rcLJmin <= r < 0.95 sigma, empty in the equilibrium crystal) sits behind a wave-uniform test.
  // (both sides of every select are plain values: `c ? expression : 0` would compile to a branch)
  const bool inwin = rsq >= q.lo && rsq <= q.hi;
  const double rs = inwin ? rsq : 1.0;
  const double r2inv = fast_rcp(rs);
  const double r6inv = r2inv * r2inv * r2inv;
  const double f12 = r6inv * (q.c1 * r6inv - q.c2) * r2inv;
  double fpair = inwin ? f12 : 0.0;
  double V = 0.0;
  if (EV) {
    const double v12 = r6inv * (q.c3 * r6inv - q.c4);
    V = inwin ? v12 : 0.0;
  }
  const bool cubic = inwin && rsq < q.sw;
  if (__any(cubic)) {
    if (cubic) {
      const double rij = sqrt(rsq);
      double rmin, k2, k3;
      if (SEG < 0) {
        rmin = P.rcLJmin[pt];
        k2 = P.ljc2[pt];
        k3 = P.ljc3[pt];
      } else { // neighbour element known at compile time, pt = the atom's element: scalar candidates (lj_select)
        rmin = pt ? P.rcLJmin[2 + SEG] : P.rcLJmin[SEG];
        k2 = pt ? P.ljc2[2 + SEG] : P.ljc2[SEG];
        k3 = pt ? P.ljc3[2 + SEG] : P.ljc3[SEG];
      }
      const double drp = rij - rmin;
      if (EV) V = drp * drp * (drp * k3 + k2);
      fpair = -drp * (3.0 * drp * k3 + 2.0 * k2) / rij;
    }
  }
  fx += dx * fpair;
  fy += dy * fpair;
  fz += dz * fpair;
  if (EV) {
    e += 0.5 * V; // both directions of every pair are visited: half the energy each
    if (vflag) {
      const double h = 0.5 * fpair;
      v0 += dx * dx * h;
      v1 += dy * dy * h;
      v2 += dz * dz * h;
      v3 += dx * dy * h;
      v4 += dx * dz * h;
      v5 += dy * dz * h;
    }
  }
}

// group reductions, stores and global tallies shared by the Lennard-Jones kernels
template <int CL, int L>
__device__ __forceinline__ void lj_store(const bool have, const int kc, const int s, const int lane, const int nlocal,
                                         const double e_lj, double (&fx)[CL], double (&fy)[CL], double (&fz)[CL],
                                         double (&ee)[CL], double v0, double v1, double v2, double v3, double v4,
                                         double v5, double *__restrict__ f, double *__restrict__ eatom,
                                         double *__restrict__ acc, const int eflag, const int vflag,
                                         const int accumulate)
{
#pragma unroll
  for (int c = 0; c < CL; c++) {
    fx[c] = group_sum<L>(fx[c]);
    fy[c] = group_sum<L>(fy[c]);
    fz[c] = group_sum<L>(fz[c]);
    if (eflag & MDP_EFLAG_ATOM) ee[c] = group_sum<L>(ee[c]);
  }
  if (have && s < CL) {
#pragma unroll
    for (int c = 0; c < CL; c++)
      if (c == s && kc * CL + c < nlocal) {
        const int ia = kc * CL + c;
        double *fo = f + 3 * (size_t) ia;
        if (accumulate) {
          fo[0] += fx[c];
          fo[1] += fy[c];
          fo[2] += fz[c];
        } else {
          fo[0] = fx[c];
          fo[1] = fy[c];
          fo[2] = fz[c];
        }
        if (eflag & MDP_EFLAG_ATOM) {
          if (accumulate)
            eatom[ia] += ee[c];
          else
            eatom[ia] = ee[c];
        }
      }
  }
  double *slot = acc + MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)));
  if (eflag & MDP_EFLAG_GLOBAL) {
    const double et = group_sum<64>(e_lj);
    if (lane == 0) atomicAdd(&slot[0], et);
  }
  if (vflag & MDP_VFLAG_GLOBAL) {
    v0 = group_sum<64>(v0);
    v1 = group_sum<64>(v1);
    v2 = group_sum<64>(v2);
    v3 = group_sum<64>(v3);
    v4 = group_sum<64>(v4);
    v5 = group_sum<64>(v5);
    if (lane == 0) {
      atomicAdd(&slot[1], v0);
      atomicAdd(&slot[2], v1);
      atomicAdd(&slot[3], v2);
      atomicAdd(&slot[4], v3);
      atomicAdd(&slot[5], v4);
      atomicAdd(&slot[6], v5);
    }
  }
}

// 12-6 branch only, straight-line: pairs inside the cubic inner spline (rcLJmin <= r < 0.95 sigma; none in
// an equilibrium crystal) are evaluated as 12-6 here and flagged; rebo_lj_cubic_kernel then replaces them.
template <bool EV>
__device__ __forceinline__ unsigned long long lj_pair_fast(const LJPar &q, const double4 &xa, const double4 &xj, double &fx,
                                                           double &fy, double &fz, double &e, const int vflag, double &v0,
                                                           double &v1, double &v2, double &v3, double &v4, double &v5,
                                                           unsigned long long &cub)
{
  const double dx = xa.x - xj.x, dy = xa.y - xj.y, dz = xa.z - xj.z;
  const double rsq = dx * dx + dy * dy + dz * dz;
  const bool inwin = rsq >= q.lo && rsq <= q.hi;
  // wave-uniform flag kept in scalar registers: the compare masks are ANDed/ORed by the scalar unit
  const unsigned long long mcub = __builtin_amdgcn_fcmp(rsq, q.lo, 3 /*oge*/) & __builtin_amdgcn_fcmp(rsq, q.sw, 4 /*olt*/);
  cub |= mcub;
  // 1/rsq: hardware seed (24 bits, measured) + ONE Newton step = 2e-15 relative.  Evaluated for every
  // entry: rsq = 0 (the atom itself) gives inf/NaN, rsq = 1e60 (dummy entry) gives 0 -- both are discarded
  // by the select below, which never propagates its unselected operand.
  double r2inv = __builtin_amdgcn_rcp(rsq);
  r2inv = fma(r2inv, fma(-rsq, r2inv, 1.0), r2inv);
  const double r6inv = r2inv * r2inv * r2inv;
  const double f12 = r6inv * (q.c1 * r6inv - q.c2) * r2inv;
  const double fpair = inwin ? f12 : 0.0;
  fx += dx * fpair;
  fy += dy * fpair;
  fz += dz * fpair;
  if (EV) {
    const double v12 = r6inv * (q.c3 * r6inv - q.c4);
    e += inwin ? 0.5 * v12 : 0.0;
    if (vflag) {
      const double h = 0.5 * fpair;
      v0 += dx * dx * h;
      v1 += dy * dy * h;
      v2 += dz * dz * h;
      v3 += dx * dy * h;
      v4 += dx * dz * h;
      v5 += dy * dz * h;
    }
  }
  return mcub; // the lanes of THIS call whose pair sits on the cubic inner spline (a wave-level value)
}

// tail of both Lennard-Jones kernels: REBO slot-force gather, group reductions, stores, global tallies
template <int CL, int L, bool GATHER>
__device__ __forceinline__ void lj_finish(const bool have, const int kc, const int s, const int lane, const int nlocal,
                                          const bool (&real)[CL], double (&fx)[CL], double (&fy)[CL],
                                          double (&fz)[CL], double (&ee)[CL], double v0, double v1, double v2,
                                          double v3, double v4, double v5, const int *__restrict__ cand_off,
                                          const unsigned long long *__restrict__ amask,
                                          const int *__restrict__ rev, const double *__restrict__ fnbr,
                                          const double *__restrict__ fown, double *__restrict__ f,
                                          double *__restrict__ eatom, double *__restrict__ acc, const int eflag,
                                          const int vflag, const int accumulate)
{
  double e_lj = 0.0;
#pragma unroll
  for (int c = 0; c < CL; c++) e_lj += real[c] ? ee[c] : 0.0;

  // ---- gather the REBO cluster forces: own centre (-sum of slot forces) + neighbour centres.
  // Slot (a,t) is active iff bit t of amask[a]; the REBO relation is symmetric, so the reverse slot
  // rev[a][t] (static between list builds) is active too and holds what centre j pushes onto a.
  // L/CL lanes work on each atom of the cluster.
  if (GATHER) {
    const int mine = s % CL;
    const int ia = kc * CL + mine;
    if (have && ia < nlocal) {
      const int off = cand_off[ia];
      const int nc = cand_off[ia + 1] - off;
      const unsigned long long act = amask[ia];
      double gx = 0, gy = 0, gz = 0, ge = 0;
      if (s / CL == 0) { // the centre's own share (written by the centre kernel)
        const double4 own = reinterpret_cast<const double4 *>(fown)[ia];
        gx = own.x;
        gy = own.y;
        gz = own.z;
        ge = own.w;
      }
      for (int t = s / CL; t < nc; t += L / CL) {
        if (!((act >> t) & 1ull)) continue;
        const int ra = rev[off + t];
        if (ra >= 0) {
          const double4 oj = reinterpret_cast<const double4 *>(fnbr)[ra];
          gx += oj.x;
          gy += oj.y;
          gz += oj.z;
          ge += oj.w;
        }
      }
#pragma unroll
      for (int c = 0; c < CL; c++)
        if (c == mine) {
          fx[c] += gx;
          fy[c] += gy;
          fz[c] += gz;
          ee[c] += ge; // per-atom energy only (the centre kernel tallies the global REBO energy)
        }
    }
  }

  lj_store<CL, L>(have, kc, s, lane, nlocal, e_lj, fx, fy, fz, ee, v0, v1, v2, v3, v4, v5, f, eatom, acc, eflag, vflag,
                  accumulate);
}

template <int CL, int L, bool EV, bool GATHER, int U = 2>
__global__ __launch_bounds__(256, MDP_LJ_WAVES) void rebo_lj_gather_kernel(
    const RebomosDev P, const int nlocal, const int *__restrict__ order, const int first, const int nclus,
    const double4 *__restrict__ xq,
    const long long *__restrict__ lj_off, const int *__restrict__ lj_split, const int *__restrict__ lj,
    const int *__restrict__ cand_off, const unsigned long long *__restrict__ amask, const int *__restrict__ rev,
    const double *__restrict__ fnbr, const double *__restrict__ fown, double *__restrict__ f,
    double *__restrict__ eatom, double *__restrict__ acc, const int eflag, const int vflag, const int accumulate)
{
  // U = list entries per lane in flight (x CL pair evaluations each)
  static_assert(L % CL == 0, "lanes per cluster must be a multiple of the cluster size");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int s = lane % L;
  const long long k64 = (long long) blockIdx.x * (256 / L) + tid / L;
  const bool have = k64 < nclus;
  // clusters are taken in the order of the interior/boundary partition (or as they come when order == null)
  const int kc = have ? (order ? order[first + (int) k64] : first + (int) k64) : 0;

  double4 xa[CL];
  int ta[CL];
  bool real[CL]; // the last cluster is padded with copies of its last atom
  double fx[CL], fy[CL], fz[CL], ee[CL];
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
#pragma unroll
  for (int c = 0; c < CL; c++) {
    const int ia = kc * CL + c;
    real[c] = have && ia < nlocal;
    xa[c] = xq[ia < nlocal ? ia : nlocal - 1];
    ta[c] = (int) xa[c].w;
    if (ta[c] < 0) { // type mapped to NULL: takes part in nothing
      real[c] = false;
      ta[c] = 0;
    }
    if (!real[c]) xa[c].x = -1.0e30; // padding / NULL atom: outside every window
    fx[c] = fy[c] = fz[c] = ee[c] = 0.0;
  }

  if (have) {
    const long long b = lj_off[kc];
    const int cnt = (int) (lj_off[kc + 1] - b);
    const int split = lj_split[kc]; // entries [0,split) are Mo, [split,cnt) are S
    const int *row = lj + b;
    const int self = kc * CL < nlocal ? kc * CL : nlocal - 1;
#pragma unroll 1
    for (int seg = 0; seg < 2; seg++) { // neighbour element is uniform inside a segment
      const int kb = seg ? split : 0, ke = seg ? cnt : split;
      LJPar q[CL];
#pragma unroll
      for (int c = 0; c < CL; c++) q[c] = lj_load(P, ta[c] * 2 + seg);
      // software pipeline: indices run two iterations ahead, coordinate gathers one iteration ahead of
      // the arithmetic, so a lane always has 2U 16-byte gathers in flight while it computes
      int j0[U], j1[U];
      double4 x0[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = kb + u * L + s;
        j0[u] = k < ke ? row[k] : -1;
      }
      // padding entries (j < 0) are moved far away: they fall outside every window without a branch
#pragma unroll
      for (int u = 0; u < U; u++) {
        x0[u] = xq[j0[u] >= 0 ? j0[u] : self];
        x0[u].x = j0[u] >= 0 ? x0[u].x : 1.0e30;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int k = kb + U * L + u * L + s;
        j1[u] = k < ke ? row[k] : -1;
      }
      for (int k0 = kb; k0 < ke; k0 += U * L) {
        double4 x1[U];
        int j2[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          x1[u] = xq[j1[u] >= 0 ? j1[u] : self];
          x1[u].x = j1[u] >= 0 ? x1[u].x : 1.0e30;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int k = k0 + 2 * U * L + u * L + s;
          j2[u] = k < ke ? row[k] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
          for (int c = 0; c < CL; c++)
            lj_pair<EV>(P, q[c], ta[c] * 2 + seg, xa[c], x0[u], fx[c], fy[c], fz[c], ee[c], vflag, v0, v1, v2, v3, v4,
                        v5);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          j0[u] = j1[u];
          x0[u] = x1[u];
          j1[u] = j2[u];
        }
      }
    }
  }
  lj_finish<CL, L, GATHER>(have, kc, s, lane, nlocal, real, fx, fy, fz, ee, v0, v1, v2, v3, v4, v5, cand_off, amask, rev,
                           fnbr, fown, f, eatom, acc, eflag, vflag, accumulate);
}

// ------------------------------------------------------------------------------------------------
// Lennard-Jones over TILE lists.  One workgroup = one tile = MDP_TILE consecutive clusters (32 atoms, a
// compact Morton blob).  The union of their neighbourhoods (~1300 atoms) is gathered from global memory
// ONCE into LDS (structure of arrays, 24 bytes per atom); the cluster rows then carry 16-bit indices into
// that union.  The old kernel's bound was the L1 tag rate of its per-lane gathers (each gathered atom
// was used by 2 pair evaluations); here a gathered atom serves ~13, and the inner loop reads LDS.
// ------------------------------------------------------------------------------------------------
#define MDP_TILE 16 // clusters per tile = 256 threads / 16 lanes per cluster

// Corrections for the pairs of a tile's rows that sit on the cubic inner spline (rcLJmin <= r < 0.95 sigma,
// pair_rebomos.cpp:532-543): none in an equilibrium crystal, a few per atom in a strained, hot or disordered structure.
// rebo_lj_tile_kernel evaluates every pair as 12-6 and puts the tiles with a flagged pair on a list; this kernel follows
// it and walks that list (grid-stride: the count is read on the device).  Per tile: the union is staged as there, the
// rows are scanned once more -- LDS reads, one distance and three compares per pair -- and the flagged (entry, atom)
// pairs of a 16-lane group are compacted into a small LDS queue, which the group then works off with all its lanes:
// the correction (cubic minus the 12-6 value already added) costs a square root and a division, and a trip of 128
// pairs typically holds one or two flagged ones.  (The first version re-walked the rows inside the tile kernel with
// per-lane parameters and evaluated the correction where it stood: 5.7 ms of a 7.1 ms kernel in a 3 300 K melt; as a
// function called from the tile kernel the pass cost the crystal 6 % through spills around the call.)
// fixtab[pt][12] = lo hi sw lj1 lj2 lj3 lj4 rcLJmin c2 c3 - -.
template <int CL, bool EV>
__global__ __launch_bounds__(256) void rebo_lj_cubic_kernel(
    const double *__restrict__ fixtab, const int *__restrict__ fix_list, const int nlocal, const int nclus,
    const double4 *__restrict__ xq, const int cap, const int *__restrict__ tu, const int *__restrict__ tile_nu,
    const long long *__restrict__ lj_off, const int *__restrict__ lj_len, const int *__restrict__ lj_split,
    const unsigned short *__restrict__ lj16, double *__restrict__ f, double *__restrict__ eatom, double *__restrict__ acc,
    const int eflag, const int vflag, int *__restrict__ h_count /* pinned: the host sizes later grids from it */)
{
  constexpr int L = 16, kCQ = 48;
  extern __shared__ double s_pos[]; // [capL][3]
  __shared__ unsigned short s_cq[MDP_TILE][kCQ];
  const int tid = threadIdx.x, lane = tid & 63, s = lane % L, glane0 = lane - s;
  unsigned short *cq = s_cq[tid / L];
  const unsigned long long below = (1ull << s) - 1ull;
  const int nfix = fix_list[0];
  if (blockIdx.x == 0 && tid == 0) *h_count = nfix;
  for (int e = blockIdx.x; e < nfix; e += gridDim.x) {
    const int t = fix_list[1 + e];
    const int kc = t * MDP_TILE + tid / L;
    const bool have = kc < nclus;
    const int nU = tile_nu[2 * t];
    const int *__restrict__ mem = tu + (size_t) t * cap;
    for (int u = tid; u < nU; u += 256) {
      const double4 v = xq[mem[u]];
      s_pos[3 * u] = v.x;
      s_pos[3 * u + 1] = v.y;
      s_pos[3 * u + 2] = v.z;
    }
    if (tid == 0) {
      s_pos[3 * nU] = 1.0e30;
      s_pos[3 * nU + 1] = 0.0;
      s_pos[3 * nU + 2] = 0.0;
    }
    const long long b = lj_off[kc];
    const int cnt = __builtin_amdgcn_readfirstlane(lj_len ? lj_len[kc] : (int) (lj_off[kc + 1] - b));
    const int split = __builtin_amdgcn_readfirstlane(lj_split[kc]);
    const unsigned short *__restrict__ row = lj16 + b;
    double4 xa[CL];
    int ta[CL];
    bool real[CL];
    double fx[CL], fy[CL], fz[CL], ee[CL];
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
#pragma unroll
    for (int c = 0; c < CL; c++) {
      real[c] = have && kc * CL + c < nlocal;
      xa[c] = xq[real[c] ? kc * CL + c : 0];
      ta[c] = (int) xa[c].w;
      if (ta[c] < 0) {
        real[c] = false;
        ta[c] = 0;
      }
      fx[c] = fy[c] = fz[c] = ee[c] = 0.0;
    }
    __syncthreads();
    int nq = 0; // items in this group's queue (the same in its 16 lanes)
    auto flush = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int i = s; i < nq; i += L) {
        const int it = (int) cq[i];
        const int li = it & 0xFFF, seg = (it >> 12) & 1, c1 = (it >> 13) & 1;
        const double ax = c1 ? xa[CL - 1].x : xa[0].x, ay = c1 ? xa[CL - 1].y : xa[0].y, az = c1 ? xa[CL - 1].z : xa[0].z;
        const double *__restrict__ q = fixtab + 12 * ((c1 ? ta[CL - 1] : ta[0]) * 2 + seg);
        const double dx = ax - s_pos[3 * li], dy = ay - s_pos[3 * li + 1], dz = az - s_pos[3 * li + 2];
        const double rsq = dx * dx + dy * dy + dz * dz;
        // (1/rsq exactly as lj_pair_fast computed it, so that the 12-6 value subtracted here is the one added there)
        double r2inv = __builtin_amdgcn_rcp(rsq);
        r2inv = fma(r2inv, fma(-rsq, r2inv, 1.0), r2inv);
        const double r6inv = r2inv * r2inv * r2inv;
        const double f12 = r6inv * (q[3] * r6inv - q[4]) * r2inv;
        const double rij = sqrt(rsq);
        const double drp = rij - q[7];
        const double fc = -drp * (3.0 * drp * q[9] + 2.0 * q[8]) / rij;
        const double df = fc - f12;
#pragma unroll
        for (int c = 0; c < CL; c++) {
          const double dfc = (CL == 1 || c == c1) ? df : 0.0;
          fx[c] += dx * dfc;
          fy[c] += dy * dfc;
          fz[c] += dz * dfc;
        }
        if (EV) {
          const double v12 = r6inv * (q[5] * r6inv - q[6]);
          const double V = drp * drp * (drp * q[9] + q[8]);
#pragma unroll
          for (int c = 0; c < CL; c++) ee[c] += (CL == 1 || c == c1) ? 0.5 * (V - v12) : 0.0;
          if (vflag) {
            const double h = 0.5 * df;
            v0 += dx * dx * h;
            v1 += dy * dy * h;
            v2 += dz * dz * h;
            v3 += dx * dy * h;
            v4 += dx * dz * h;
            v5 += dy * dz * h;
          }
        }
      }
      nq = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int seg = 0; seg < 2; seg++) {
      const int kb = seg ? split : 0, ke = seg ? cnt : split;
      double lo[CL], hi[CL], sw[CL];
#pragma unroll
      for (int c = 0; c < CL; c++) {
        const double *__restrict__ q = fixtab + 12 * (ta[c] * 2 + seg);
        lo[c] = q[0];
        hi[c] = q[1];
        sw[c] = q[2];
      }
      int li_next = kb + s < ke ? (int) row[kb + s] : nU;
      for (int k = kb + s; k < ke; k += L) { // (whole 16-lane steps, equally many for the rows of a wave)
        const int li = li_next;
        li_next = k + L < ke ? (int) row[k + L] : nU;
        const double xjx = s_pos[3 * li], xjy = s_pos[3 * li + 1], xjz = s_pos[3 * li + 2];
#pragma unroll
        for (int c = 0; c < CL; c++) {
          const double dx = xa[c].x - xjx, dy = xa[c].y - xjy, dz = xa[c].z - xjz;
          const double rsq = dx * dx + dy * dy + dz * dz;
          const bool hit = real[c] && rsq >= lo[c] && rsq <= hi[c] && rsq < sw[c]; // (the dummy entry is far outside)
          const unsigned long long m = (__ballot(hit) >> glane0) & 0xFFFFull;
          if (hit) cq[nq + __popcll(m & below)] = (unsigned short) (li | (seg << 12) | ((CL == 1 ? 0 : c) << 13));
          nq += __popcll(m);
        }
        if (__any(nq > kCQ - 2 * L)) flush(); // (room for one more trip of both atoms in every group)
      }
    }
    flush();
    double e_fix = 0.0;
#pragma unroll
    for (int c = 0; c < CL; c++) e_fix += ee[c];
    lj_store<CL, L>(have, kc, s, lane, nlocal, e_fix, fx, fy, fz, ee, v0, v1, v2, v3, v4, v5, f, eatom, acc, eflag, vflag,
                    /*accumulate=*/1);
    __syncthreads(); // (the next tile's union overwrites s_pos)
  }
}

// The follow-up of the QUEUE variant of rebo_lj_tile_kernel: per listed tile every 16-lane group works off the two
// queues it wrote -- items (index into the union | segment << 12) of the (entry, atom) pairs on the cubic inner spline --
// with all its lanes: position of the neighbour through the union's member list, (cubic - 12-6) exactly as
// rebo_lj_cubic_kernel's flush computes it, group reduction, read-modify-write of f.  No union is staged and no row is
// walked.  The counts are zeroed again here (they are written only when non-zero); a tile one of whose queues overflowed
// goes to `walk_list` and is taken by rebo_lj_cubic_kernel behind this launch.
template <int CL, bool EV>
__global__ __launch_bounds__(256) void rebo_lj_cubicq_kernel(
    const double *__restrict__ fixtab, const int *__restrict__ fix_list, int *__restrict__ walk_list, const int nlocal,
    const int nclus, const double4 *__restrict__ xq, const int cap, const int *__restrict__ tu,
    const unsigned short *__restrict__ cq, int *__restrict__ cq_cnt, const int cq_cap, double *__restrict__ f,
    double *__restrict__ eatom, double *__restrict__ acc, const int eflag, const int vflag,
    int *__restrict__ h_count /* pinned: the host sizes later grids and picks the variant from it */)
{
  constexpr int L = 16;
  const int tid = threadIdx.x, lane = tid & 63, s = lane % L;
  const int nfix = fix_list[0];
  if (blockIdx.x == 0 && tid == 0) *h_count = nfix;
  for (int e = blockIdx.x; e < nfix; e += gridDim.x) {
    const int t = fix_list[1 + e];
    const int kc = t * MDP_TILE + tid / L;
    const bool have = kc < nclus;
    const int *__restrict__ mem = tu + (size_t) t * cap;
    int *__restrict__ cnt = cq_cnt + ((size_t) t * MDP_TILE + tid / L) * CL;
    int nq[CL];
    bool over = false;
#pragma unroll
    for (int c = 0; c < CL; c++) {
      nq[c] = cnt[c];
      over = over || nq[c] > cq_cap;
    }
    over = __syncthreads_or(over) != 0;
    if (s == 0) {
#pragma unroll
      for (int c = 0; c < CL; c++)
        if (nq[c]) cnt[c] = 0;
    }
    if (over) { // (block-uniform)
      if (tid == 0) walk_list[1 + atomicAdd(&walk_list[0], 1)] = t;
      continue;
    }
    double4 xa[CL];
    int ta[CL];
    double fx[CL], fy[CL], fz[CL], ee[CL];
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
#pragma unroll
    for (int c = 0; c < CL; c++) {
      xa[c] = xq[have && kc * CL + c < nlocal ? kc * CL + c : 0];
      ta[c] = (int) xa[c].w;
      ta[c] = ta[c] < 0 ? 0 : ta[c];
      fx[c] = fy[c] = fz[c] = ee[c] = 0.0;
    }
    const unsigned short *__restrict__ q0 = cq + ((size_t) t * MDP_TILE + tid / L) * CL * cq_cap;
#pragma unroll
    for (int c = 0; c < CL; c++) {
      for (int i = s; i < nq[c]; i += L) {
        const int it = (int) q0[c * cq_cap + i];
        const int li = it & 0xFFF, seg = (it >> 12) & 1;
        const double4 xj = xq[mem[li]];
        const double *__restrict__ q = fixtab + 12 * (ta[c] * 2 + seg);
        const double dx = xa[c].x - xj.x, dy = xa[c].y - xj.y, dz = xa[c].z - xj.z;
        const double rsq = dx * dx + dy * dy + dz * dz;
        // (1/rsq exactly as lj_pair_fast computed it, so that the 12-6 value subtracted here is the one added there)
        double r2inv = __builtin_amdgcn_rcp(rsq);
        r2inv = fma(r2inv, fma(-rsq, r2inv, 1.0), r2inv);
        const double r6inv = r2inv * r2inv * r2inv;
        const double f12 = r6inv * (q[3] * r6inv - q[4]) * r2inv;
        const double rij = sqrt(rsq);
        const double drp = rij - q[7];
        const double fc = -drp * (3.0 * drp * q[9] + 2.0 * q[8]) / rij;
        const double df = fc - f12;
        fx[c] += dx * df;
        fy[c] += dy * df;
        fz[c] += dz * df;
        if (EV) {
          const double v12 = r6inv * (q[5] * r6inv - q[6]);
          const double V = drp * drp * (drp * q[9] + q[8]);
          ee[c] += 0.5 * (V - v12);
          if (vflag) {
            const double h = 0.5 * df;
            v0 += dx * dx * h;
            v1 += dy * dy * h;
            v2 += dz * dz * h;
            v3 += dx * dy * h;
            v4 += dx * dz * h;
            v5 += dy * dz * h;
          }
        }
      }
    }
    double e_fix = 0.0;
#pragma unroll
    for (int c = 0; c < CL; c++) e_fix += ee[c];
    lj_store<CL, L>(have, kc, s, lane, nlocal, e_fix, fx, fy, fz, ee, v0, v1, v2, v3, v4, v5, f, eatom, acc, eflag, vflag,
                    /*accumulate=*/1);
  }
}

// QUEUE (a hot system: most tiles were listed in the compute before): the flagged (entry, atom) pairs are not left for
// a second walk of the rows -- every 16-lane group appends them, as 16-bit items (index into the union | segment << 12),
// to its own two queues in device memory (one per atom of the cluster: cq[((tile * 16 + group) * 2 + atom) * cq_cap ...],
// counts in cq_cnt, written only when non-zero and zeroed again by their reader), and rebo_lj_cubicq_kernel adds the
// corrections from the queues alone.  The appends sit behind a wave-uniform branch on the compare mask.  The variant
// without QUEUE is the kernel as it was: a crystal never runs this one (launch_lj decides from the last list's length).
template <bool EV, bool GATHER, int WAVES, int CL, bool QUEUE = false>
__global__ __launch_bounds__(256, WAVES) void rebo_lj_tile_kernel(
    const RebomosDev P, const int nlocal, const int *__restrict__ order, const int first, const int nclus,
    const double4 *__restrict__ xq, const int cap, const int capL, const int skip_above,
    const int *__restrict__ tu, const int *__restrict__ tile_nu, const long long *__restrict__ lj_off,
    const int *__restrict__ lj_len /* row lengths of the pruned rows, or null: the rows as built */,
    const int *__restrict__ lj_split, const unsigned short *__restrict__ lj16, const int *__restrict__ cand_off,
    const unsigned long long *__restrict__ amask, const int *__restrict__ rev, const int *__restrict__ rev16,
    const double *__restrict__ fnbr, const double *__restrict__ fown, double *__restrict__ f,
    double *__restrict__ eatom, double *__restrict__ acc,
    const int eflag, const int vflag, const int accumulate, int *__restrict__ fix_list, int *__restrict__ fix_stamp,
    const int stamp, unsigned short *__restrict__ cq = nullptr, int *__restrict__ cq_cnt = nullptr, const int cq_cap = 0)
{
  constexpr int L = 16; // lanes per row; CL atoms per row (1: every atom walks its own neighbourhood, nothing of a partner's)
  constexpr int U = 1; // row entries per lane and iteration (segments are padded to L entries, so U * L must be L)
  constexpr int SK = 3; // union members per thread whose index loads are issued unconditionally (cap >= 2048)
  extern __shared__ double s_pos[]; // [capL][3]: x y z of union member u at s_pos + 3u (one address, three offsets)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int s = lane % L;
  const int bx = xcd_contiguous(blockIdx.x, gridDim.x);
  const int t = order ? order[first + bx] : first + bx;
  const int kc = t * MDP_TILE + tid / L; // row (every tile has MDP_TILE rows; absent clusters: all-dummy rows)

  // A workgroup lives for ~10 us, so a chain of dependent global round trips (~1 us each) at its head or
  // tail is what would bound the kernel.  Everything that depends on the tile number alone is therefore
  // requested at once: union size and member indices, row bounds, the cluster's own atoms ...
  // Natural tile order (order == null, the single-GPU case) saves the round trip through the order list;
  // the rare tiles whose union exceeds this launch's LDS allocation are then predicated to empty here (no
  // branch, so the loads below stay independent of this one) and taken by the "large" launch.
  const int nU_all = tile_nu[2 * t];
  const bool live = nU_all <= skip_above;
  const int nU = live ? nU_all : 0;
  const bool have = kc < nclus && live;
  const int *__restrict__ mem = tu + (size_t) t * cap;
  int sidx[SK];
#pragma unroll
  for (int k = 0; k < SK; k++) sidx[k] = mem[tid + 256 * k]; // rows are cap >= 2048 long: always in bounds
  const long long b = lj_off[kc];
  // segment lengths are equal for the four clusters of a wave by construction (tile_scan_kernel): scalars
  const int cnt = __builtin_amdgcn_readfirstlane(live ? (lj_len ? lj_len[kc] : (int) (lj_off[kc + 1] - b)) : 0);
  const int split = __builtin_amdgcn_readfirstlane(live ? lj_split[kc] : 0);
  const unsigned short *__restrict__ row = lj16 + b;
  double4 xa[CL];
#pragma unroll
  for (int c = 0; c < CL; c++) xa[c] = xq[have && kc * CL + c < nlocal ? kc * CL + c : nlocal - 1];
  // ... (second round) the coordinates of the union, the head of both row segments ...
  {
    double4 sv[SK]; // all gathers in flight together (index clamped instead of a branch around each load)
#pragma unroll
    for (int k = 0; k < SK; k++) sv[k] = xq[tid + 256 * k < nU ? sidx[k] : 0];
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int u = tid + 256 * k;
      if (u < nU) {
        s_pos[3 * u] = sv[k].x;
        s_pos[3 * u + 1] = sv[k].y;
        s_pos[3 * u + 2] = sv[k].z;
      }
    }
  }
  for (int u = tid + 256 * SK; u < nU; u += 256) {
    const double4 v = xq[mem[u]];
    s_pos[3 * u] = v.x;
    s_pos[3 * u + 1] = v.y;
    s_pos[3 * u + 2] = v.z;
  }
  if (tid == 0) { // slot nU: what the padding entries of the rows point at -- outside every window
    s_pos[3 * nU] = 1.0e30;
    s_pos[3 * nU + 1] = 0.0;
    s_pos[3 * nU + 2] = 0.0;
  }
  // (both row segments are padded to whole 16-lane steps with the dummy index, so no load below needs a
  // bounds check; reads run at most a few steps past a segment into the following rows / the slack)
  int jh[2][3 * U]; // first 3U entries per lane of the Mo segment and of the S segment
#pragma unroll
  for (int seg = 0; seg < 2; seg++) {
#pragma unroll
    for (int u = 0; u < 3 * U; u++) jh[seg][u] = (int) row[(seg ? split : 0) + u * L + s];
  }
  // ... and the REBO slot forces of this atom: own centre (-sum of its slot forces) plus what the neighbour
  // centres push onto it through the reverse slots.  Done here, not after the loops, so that its three
  // dependent loads (row bounds -> reverse slot -> slot force) overlap the staging chain; L/CL lanes per atom.
  const int mine = s % CL;
  double gx = 0, gy = 0, gz = 0, ge = 0;
  if (GATHER) {
    const int ia_g = kc * CL + mine;
    const bool g_on = have && ia_g < nlocal;
    const int ia_c = g_on ? ia_g : 0;
    // two reverse slots per lane (8 lanes per atom cover 16), every load unconditional with a clamped
    // address so that all of a level is in flight together
    const int t0 = s / CL, t1 = CL > 1 ? t0 + L / CL : t0; // (one atom per row: 16 lanes cover the 16 slots, one each)
    const unsigned long long g_act = g_on ? amask[ia_c] : 0ull;
    const int r0 = rev16[(size_t) ia_c * 16 + t0], r1 = rev16[(size_t) ia_c * 16 + t1];
    const double4 own = reinterpret_cast<const double4 *>(fown)[ia_c]; // the centre's own share
    const bool ok0 = ((g_act >> t0) & 1ull) && r0 >= 0, ok1 = CL > 1 && ((g_act >> t1) & 1ull) && r1 >= 0;
    const double4 a0 = reinterpret_cast<const double4 *>(fnbr)[ok0 ? r0 : 0];
    const double4 a1 = reinterpret_cast<const double4 *>(fnbr)[ok1 ? r1 : 0];
    if (g_on && t0 == 0) {
      gx = own.x;
      gy = own.y;
      gz = own.z;
      ge = own.w;
    }
    gx += (ok0 ? a0.x : 0.0) + (ok1 ? a1.x : 0.0);
    gy += (ok0 ? a0.y : 0.0) + (ok1 ? a1.y : 0.0);
    gz += (ok0 ? a0.z : 0.0) + (ok1 ? a1.z : 0.0);
    ge += (ok0 ? a0.w : 0.0) + (ok1 ? a1.w : 0.0);
    if (g_act >> 16) { // rare: more than 16 candidates, the rest through the row itself
      const int g_off = cand_off[ia_c];
      const int g_nc = cand_off[ia_c + 1] - g_off;
      for (int tt = 16 + s / CL; tt < g_nc; tt += L / CL) {
        if (!((g_act >> tt) & 1ull)) continue;
        const int ra = rev[g_off + tt];
        if (ra >= 0) {
          const double4 oj = reinterpret_cast<const double4 *>(fnbr)[ra];
          gx += oj.x;
          gy += oj.y;
          gz += oj.z;
          ge += oj.w;
        }
      }
    }
  }

  int ta[CL];
  bool real[CL];
  double fx[CL], fy[CL], fz[CL], ee[CL];
  double v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
#pragma unroll
  for (int c = 0; c < CL; c++) {
    real[c] = have && kc * CL + c < nlocal;
    ta[c] = (int) xa[c].w;
    if (ta[c] < 0) {
      real[c] = false;
      ta[c] = 0;
    }
    if (!real[c]) xa[c].x = -1.0e30;
    fx[c] = c == mine ? gx : 0.0;
    fy[c] = c == mine ? gy : 0.0;
    fz[c] = c == mine ? gz : 0.0;
    ee[c] = 0.0;
  }
  __syncthreads();

  // One row segment (neighbour element SEG, a compile-time constant so that every parameter is a select
  // between two scalar registers).  Two iterations per trip with two index sets (je: even, jo: odd
  // iterations) and two coordinate sets: an index set is re-requested from global memory the moment its
  // LDS reads are issued and is needed again two iterations later.  No register is copied and the loads
  // are unconditional (address clamped, validity re-derived at use), so the only wait inside the loop is
  // for the load issued a full trip earlier.
  unsigned long long cub = 0;
  // QUEUE: this group's two queues (entries so far: the same number in its 16 lanes)
  int qn[CL];
#pragma unroll
  for (int c = 0; c < CL; c++) qn[c] = 0;
  const int glane0 = lane - s;
  unsigned short *__restrict__ cqg = QUEUE ? cq + ((size_t) t * MDP_TILE + tid / L) * CL * cq_cap : nullptr;
  auto push = [&](const int c, const unsigned long long m, const int li, const int seg) {
    const unsigned mg = (unsigned) (m >> glane0) & 0xFFFFu; // the group's 16 bits of the wave's compare mask
    const int at = qn[c] + __popc(mg & ((1u << s) - 1u));
    if (((mg >> s) & 1u) && at < cq_cap) cqg[c * cq_cap + at] = (unsigned short) (li | (seg << 12));
    qn[c] += __popc(mg);
  };
  auto segment = [&](auto segc) {
    constexpr int SEG = decltype(segc)::value;
    const int kb = SEG ? split : 0, ke = SEG ? cnt : split;
    LJPar q[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) q[c] = lj_select(P, ta[c], SEG);
    int je[U], jo[U];
    int ie[U], io[U]; // QUEUE: the entries xe / xo were read for (their index registers are re-requested at once)
    double4 xe[U], xo[U];
    const unsigned short *__restrict__ rp = row + kb + s; // this lane's column of the segment
#pragma unroll
    for (int u = 0; u < U; u++) {
      const double *p = s_pos + 3 * jh[SEG][u]; // iteration 0
      xe[u] = make_double4(p[0], p[1], p[2], 0.0);
      ie[u] = jh[SEG][u];
      io[u] = 0;
      jo[u] = jh[SEG][U + u];     // iteration 1
      je[u] = jh[SEG][2 * U + u]; // iteration 2
    }
#pragma unroll 1
    for (int k0 = kb; k0 < ke; k0 += 2 * U * L) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double *p = s_pos + 3 * jo[u];
        xo[u] = make_double4(p[0], p[1], p[2], 0.0);
        if (QUEUE) io[u] = jo[u];
      }
#pragma unroll
      for (int u = 0; u < U; u++) jo[u] = (int) rp[3 * U * L + u * L];
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int c = 0; c < CL; c++) {
          const unsigned long long m =
              lj_pair_fast<EV>(q[c], xa[c], xe[u], fx[c], fy[c], fz[c], ee[c], vflag, v0, v1, v2, v3, v4, v5, cub);
          if (QUEUE && m) push(c, m, ie[u], SEG);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const double *p = s_pos + 3 * je[u];
        xe[u] = make_double4(p[0], p[1], p[2], 0.0);
        if (QUEUE) ie[u] = je[u];
      }
#pragma unroll
      for (int u = 0; u < U; u++) je[u] = (int) rp[4 * U * L + u * L];
      rp += 2 * U * L;
      if (k0 + U * L < ke) {
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
          for (int c = 0; c < CL; c++) {
            const unsigned long long m =
                lj_pair_fast<EV>(q[c], xa[c], xo[u], fx[c], fy[c], fz[c], ee[c], vflag, v0, v1, v2, v3, v4, v5, cub);
            if (QUEUE && m) push(c, m, io[u], SEG);
          }
        }
      }
    }
  };
  segment(std::integral_constant<int, 0>{});
  segment(std::integral_constant<int, 1>{});

  // Some pair of this wave sat on the cubic inner spline: the tile goes onto the list of rebo_lj_cubic_kernel, which
  // follows this launch and adds the corrections (the first flagged wave of a tile appends it; `stamp` changes with
  // every compute, so the per-tile words need no reset).  Nothing of that path costs this kernel a register.
  if (cub && lane == 0 && atomicExch(&fix_stamp[t], stamp) != stamp) fix_list[1 + atomicAdd(&fix_list[0], 1)] = t;
  if (QUEUE && s == 0) { // (a count beyond the capacity tells the reader that this tile needs the walk of the rows)
#pragma unroll
    for (int c = 0; c < CL; c++)
      if (qn[c]) cq_cnt[((size_t) t * MDP_TILE + tid / L) * CL + c] = qn[c];
  }

  double e_lj = 0.0;
#pragma unroll
  for (int c = 0; c < CL; c++) e_lj += real[c] ? ee[c] : 0.0;
  if (GATHER) { // per-atom REBO energy rides along after the LJ total has been taken
#pragma unroll
    for (int c = 0; c < CL; c++)
      if (c == mine) ee[c] += ge;
  }
  lj_store<CL, L>(have, kc, s, lane, nlocal, e_lj, fx, fy, fz, ee, v0, v1, v2, v3, v4, v5, f, eatom, acc, eflag, vflag,
                  accumulate);
}

// REBO slot-force gather alone: f[a] += -sum_t fnbr[a][t] + sum_t fnbr[rev[a][t]] (and the per-atom REBO
// energy).  Used when the Lennard-Jones kernel runs in two parts around the halo exchange.
template <int L>
__global__ __launch_bounds__(256) void rebo_gather_kernel(const int nlocal, const int *__restrict__ cand_off,
                                                          const unsigned long long *__restrict__ amask,
                                                          const int *__restrict__ rev,
                                                          const double *__restrict__ fnbr,
                                                          const double *__restrict__ fown, double *__restrict__ f,
                                                          double *__restrict__ eatom, const int eflag,
                                                          const double *__restrict__ vslot,
                                                          double *__restrict__ vatom)
{
  const int s = threadIdx.x % L;
  const long long a64 = (long long) blockIdx.x * (256 / L) + threadIdx.x / L;
  const bool have = a64 < nlocal;
  const int ia = have ? (int) a64 : 0;
  double gx = 0, gy = 0, gz = 0, ge = 0;
  double gv[6] = {0, 0, 0, 0, 0, 0};
  if (have) {
    const int off = cand_off[ia];
    const int nc = cand_off[ia + 1] - off;
    const unsigned long long act = amask[ia];
    if (s == 0) { // the centre's own share (written by the centre kernel)
      const double4 own = reinterpret_cast<const double4 *>(fown)[ia];
      gx = own.x;
      gy = own.y;
      gz = own.z;
      ge = own.w;
    }
    for (int t = s; t < nc; t += L) {
      if (!((act >> t) & 1ull)) continue;
      const int ra = rev[off + t];
      if (ra >= 0) {
        const double4 oj = reinterpret_cast<const double4 *>(fnbr)[ra];
        gx += oj.x;
        gy += oj.y;
        gz += oj.z;
        ge += oj.w;
        if (vslot) // what the neighbour centre's cluster energy contributes to this atom's virial
#pragma unroll
          for (int k6 = 0; k6 < 6; k6++) gv[k6] += vslot[6 * (size_t) ra + k6];
      }
    }
  }
  gx = group_sum<L>(gx);
  gy = group_sum<L>(gy);
  gz = group_sum<L>(gz);
  if (eflag & MDP_EFLAG_ATOM) ge = group_sum<L>(ge);
  if (vslot)
#pragma unroll
    for (int k6 = 0; k6 < 6; k6++) gv[k6] = group_sum<L>(gv[k6]);
  if (have && s == 0) {
    double *fo = f + 3 * (size_t) ia;
    fo[0] += gx;
    fo[1] += gy;
    fo[2] += gz;
    if (eflag & MDP_EFLAG_ATOM) eatom[ia] += ge;
    if (vslot)
#pragma unroll
      for (int k6 = 0; k6 < 6; k6++) vatom[6 * (size_t) ia + k6] += gv[k6];
  }
}

// Lennard-Jones part of the per-atom virial (ev_tally, pair_rebomos.cpp:554): half of every pair's virial
// to either end; with both directions visited each atom simply keeps half of what it sees.  Separate small
// kernel (8 lanes per atom over the atom's cluster list) so that the hot kernel carries no extra registers.
template <int CL>
__global__ __launch_bounds__(256) void rebo_lj_vatom_kernel(const RebomosDev P, const int nlocal,
                                                            const double4 *__restrict__ xq,
                                                            const long long *__restrict__ lj_off,
                                                            const int *__restrict__ lj,
                                                            const unsigned short *__restrict__ lj16,
                                                            const int *__restrict__ tu, const int cap,
                                                            const int *__restrict__ tile_nu,
                                                            double *__restrict__ vatom, const int tile_rows)
{
  constexpr int L = 8;
  const int s = threadIdx.x % L;
  const long long a64 = (long long) blockIdx.x * (256 / L) + threadIdx.x / L;
  const bool have = a64 < nlocal;
  const int a = have ? (int) a64 : 0;
  const double4 xa = xq[a];
  const int ta = (int) xa.w;
  double v[6] = {0, 0, 0, 0, 0, 0};
  if (have && ta >= 0) {
    const int kc = a / CL;
    const int *mem = lj16 ? tu + (size_t) (kc / tile_rows) * cap : nullptr; // tile lists: rows index the union
    const int nU = lj16 ? tile_nu[2 * (kc / tile_rows)] : -1;
    for (long long k = lj_off[kc] + s; k < lj_off[kc + 1]; k += L) {
      if (lj16 && lj16[k] == nU) continue; // row padding
      const double4 xj = xq[lj16 ? mem[lj16[k]] : lj[k]];
      const int pt = ta * 2 + (int) xj.w;
      const LJPar q = lj_load(P, pt);
      double fx = 0, fy = 0, fz = 0, e = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
      lj_pair<true>(P, q, pt, xa, xj, fx, fy, fz, e, 1, d0, d1, d2, d3, d4, d5); // d* = dx dx fpair / 2 ...
      v[0] += d0;
      v[1] += d1;
      v[2] += d2;
      v[3] += d3;
      v[4] += d4;
      v[5] += d5;
    }
  }
#pragma unroll
  for (int k6 = 0; k6 < 6; k6++) {
    v[k6] = group_sum<L>(v[k6]);
    if (have && s == 0) vatom[6 * (size_t) a + k6] += v[k6];
  }
}

// launch classes of the Lennard-Jones tiles: key = (tile union too large for the small LDS allocation).  One-hot
// flags, scanned separately, give a stable order [small | large] (keys 2, 3 -- is_bnd -- are not in use).
__global__ void unit_class_kernel(const int n, const int *__restrict__ is_bnd, const int *__restrict__ tile_nu,
                                  const int small_limit, int *__restrict__ flag /* [4][n+1] */)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int key = (is_bnd && is_bnd[k] ? 2 : 0) + (tile_nu && tile_nu[2 * k] > small_limit ? 1 : 0);
#pragma unroll
  for (int q = 0; q < 4; q++) flag[(size_t) q * (n + 1) + k] = q == key;
}

__global__ void unit_order_kernel(const int n, const int *__restrict__ flag, const int *__restrict__ pos /* [4][n+2] */,
                                  const int b1, const int b2, const int b3, int *__restrict__ order)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int base[4] = {0, b1, b2, b3};
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (flag[(size_t) q * (n + 1) + k]) order[base[q] + pos[(size_t) q * (n + 2) + k]] = k;
}

// cluster pair list straight from the bin grid: every atom j within rcLJmax+skin of ANY atom of the
// cluster, Mo neighbours first, then S (so the LJ parameters are loop invariants of each segment).
// 16 lanes per cluster sweep the stencil cells (each x-run of cells is contiguous in the sorted order),
// ballot-compacting the hits; consecutive clusters are spatial neighbours and share their stencil in cache.
template <int CL, bool FILL>
__global__ __launch_bounds__(256) void cluster_build_kernel(const MdpGrid g, const RebomosDev P, const int nclus,
                                                            const int nlocal, const double4 *__restrict__ xq,
                                                            const int *__restrict__ perm,
                                                            const int *__restrict__ cell_start, int *__restrict__ cnt,
                                                            int *__restrict__ split, const long long *__restrict__ off,
                                                            int *__restrict__ out)
{
  constexpr int L = 16;
  const int lane = threadIdx.x & 63;
  const int s = lane % L;
  const int glane0 = lane - s;
  const long long k64 = (long long) blockIdx.x * (256 / L) + threadIdx.x / L;
  const bool have = k64 < nclus;
  const int k = have ? (int) k64 : 0;
  double4 xa[CL];
  int ta[CL];
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
#pragma unroll
  for (int c = 0; c < CL; c++) {
    const int ia = k * CL + c < nlocal ? k * CL + c : nlocal - 1;
    xa[c] = xq[ia];
    ta[c] = (int) xa[c].w;
    int cc[3];
    cc[0] = (int) ((xa[c].x - g.lo[0]) * g.inv[0]);
    cc[1] = (int) ((xa[c].y - g.lo[1]) * g.inv[1]);
    cc[2] = (int) ((xa[c].z - g.lo[2]) * g.inv[2]);
#pragma unroll
    for (int d = 0; d < 3; d++) {
      cc[d] = cc[d] < 0 ? 0 : (cc[d] >= g.n[d] ? g.n[d] - 1 : cc[d]);
      lo[d] = cc[d] < lo[d] ? cc[d] : lo[d];
      hi[d] = cc[d] > hi[d] ? cc[d] : hi[d];
    }
  }
  const int R = g.range;
  // wave-uniform stencil bounds (the 4 clusters of a wave may differ by a cell): loop the union
  int zlo = max(lo[2] - R, 0), zhi = min(hi[2] + R, g.n[2] - 1);
  int ylo = max(lo[1] - R, 0), yhi = min(hi[1] + R, g.n[1] - 1);
  const int xlo = max(lo[0] - R, 0), xhi = min(hi[0] + R, g.n[0] - 1);
  if (!have) {
    zlo = ylo = 1 << 30;
    zhi = yhi = -1;
  }
  int wzlo = zlo, wzhi = zhi, wylo = ylo, wyhi = yhi;
#pragma unroll
  for (int o = 32; o >= L; o >>= 1) {
    wzlo = min(wzlo, __shfl_xor(wzlo, o, 64));
    wzhi = max(wzhi, __shfl_xor(wzhi, o, 64));
    wylo = min(wylo, __shfl_xor(wylo, o, 64));
    wyhi = max(wyhi, __shfl_xor(wyhi, o, 64));
  }
  int n0 = 0, n1 = 0;
  int *row0 = (FILL && have) ? out + off[k] : nullptr;
  int *row1 = (FILL && have) ? row0 + split[k] : nullptr;
  const unsigned long long below = (1ull << s) - 1ull;
  for (int z = wzlo; z <= wzhi; z++)
    for (int y = wylo; y <= wyhi; y++) {
      const bool rowin = have && z >= zlo && z <= zhi && y >= ylo && y <= yhi;
      int pb = 0, pe = 0;
      if (rowin) {
        pb = cell_start[xlo + g.n[0] * (y + g.n[1] * z)];
        pe = cell_start[xhi + g.n[0] * (y + g.n[1] * z) + 1];
      }
      const int lenw = wave_max_int(pe - pb);
      for (int base = 0; base < lenw; base += L) {
        const int p = pb + base + s;
        bool k0 = false, k1 = false;
        int j = 0;
        if (p < pe) {
          j = perm[p];
          const double4 xj = xq[j];
          const int tj = (int) xj.w;
          bool keep = false;
#pragma unroll
          for (int c = 0; c < CL; c++) {
            const double dx = xa[c].x - xj.x, dy = xa[c].y - xj.y, dz = xa[c].z - xj.z;
            keep = keep || (ta[c] >= 0 && dx * dx + dy * dy + dz * dz <= P.ljlist_cutsq[ta[c] * 2 + tj]);
          }
          keep = keep && tj >= 0; // NULL-mapped neighbours are invisible to this style
          k0 = keep && tj == 0;
          k1 = keep && tj != 0;
        }
        const unsigned long long b0 = (__ballot(k0) >> glane0) & 0xFFFFull;
        const unsigned long long b1 = (__ballot(k1) >> glane0) & 0xFFFFull;
        if (FILL) {
          if (k0) row0[n0 + __popcll(b0 & below)] = j;
          if (k1) row1[n1 + __popcll(b1 & below)] = j;
        }
        n0 += __popcll(b0);
        n1 += __popcll(b1);
      }
    }
  if (!FILL && have && s == 0) {
    cnt[k] = n0 + n1;
    split[k] = n0;
  }
}

// ---- tile lists -------------------------------------------------------------------------------
// Pass 1, one workgroup per tile: sweep the stencil of the tile's bounding cells; every candidate is
// tested against all atoms of the tile (broadcast LDS reads) giving a 16-bit mask of the clusters that
// list it.  Each wave takes every fourth (y,z) row and compacts its members with ballots into its own
// LDS segment (Mo from the front, S from the back); the segments are then concatenated Mo-first, so the
// result is deterministic.  Outputs: the union (tu), the masks, and the row lengths of the 16 clusters.
template <int CL>
__global__ __launch_bounds__(256) void tile_scan_kernel(const MdpGrid g, const RebomosDev P, const int nclus,
                                                        const int nlocal, const double4 *__restrict__ xq,
                                                        const int *__restrict__ perm,
                                                        const int *__restrict__ cell_start, const int cap,
                                                        int *__restrict__ tu, unsigned short *__restrict__ tmask,
                                                        int *__restrict__ tile_nu, int *__restrict__ cnt,
                                                        int *__restrict__ split, int *__restrict__ tile_flag,
                                                        const double4 *__restrict__ xq_cell)
{
  constexpr int NA = MDP_TILE * CL;
  extern __shared__ int s_dyn[];
  const int wcap = cap / 2; // per-wave segment (a wave sees about a quarter of the union)
  int *s_idx = s_dyn;                                                    // [4][wcap]
  unsigned short *s_m = (unsigned short *) (s_dyn + 4 * wcap);           // [4][wcap]
  unsigned short *s_fm = s_m + 4 * wcap;                                 // [cap] masks in final order
  __shared__ double4 s_xa[NA];
  __shared__ double s_cut[NA][2];
  __shared__ int s_lo[3], s_hi[3], s_n0[4], s_n1[4], s_over, s_rowsum;
  __shared__ double s_bb[6], s_cmax; // bounding box of the tile's atoms (lo x y z | hi x y z), their largest list cutoff
  const int t = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 3) {
    s_lo[tid] = 1 << 30;
    s_hi[tid] = -1;
  }
  if (tid == 0) s_over = s_rowsum = 0;
  __syncthreads();
  if (tid < NA) {
    const int ia = t * NA + tid;
    bool valid = ia < nlocal;
    const double4 xa = xq[valid ? ia : (nlocal > 0 ? nlocal - 1 : 0)];
    int ta = (int) xa.w;
    if (ta < 0) valid = false; // NULL-mapped atom: lists nothing
    ta = ta > 1 ? 1 : ta;      // two classes: type 0 | every other type (a style with more types sorts them out itself)
    s_xa[tid] = xa;
    s_cut[tid][0] = valid ? P.ljlist_cutsq[ta * 2 + 0] : -1.0;
    s_cut[tid][1] = valid ? P.ljlist_cutsq[ta * 2 + 1] : -1.0;
    {
      // (the NA <= 64 atom threads are the first lanes of wave 0: min / max by shuffles; absent atoms do not count)
      static_assert(NA <= 64, "the tile's atoms are held by one wave");
      double lo3[3] = {valid ? xa.x : 1.0e300, valid ? xa.y : 1.0e300, valid ? xa.z : 1.0e300};
      double hi3[3] = {valid ? xa.x : -1.0e300, valid ? xa.y : -1.0e300, valid ? xa.z : -1.0e300};
      double cm = valid ? fmax(s_cut[tid][0], s_cut[tid][1]) : -1.0;
#pragma unroll
      for (int o = 1; o < NA; o <<= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
          lo3[d] = fmin(lo3[d], __shfl_xor(lo3[d], o, 64));
          hi3[d] = fmax(hi3[d], __shfl_xor(hi3[d], o, 64));
        }
        cm = fmax(cm, __shfl_xor(cm, o, 64));
      }
      if (tid == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
          s_bb[d] = lo3[d];
          s_bb[3 + d] = hi3[d];
        }
        s_cmax = cm;
      }
    }
    if (valid) {
      int cc[3];
      cc[0] = (int) ((xa.x - g.lo[0]) * g.inv[0]);
      cc[1] = (int) ((xa.y - g.lo[1]) * g.inv[1]);
      cc[2] = (int) ((xa.z - g.lo[2]) * g.inv[2]);
#pragma unroll
      for (int d = 0; d < 3; d++) {
        cc[d] = cc[d] < 0 ? 0 : (cc[d] >= g.n[d] ? g.n[d] - 1 : cc[d]);
        atomicMin(&s_lo[d], cc[d]);
        atomicMax(&s_hi[d], cc[d]);
      }
    }
  }
  __syncthreads();
  const int R = g.range;
  const bool any = s_hi[0] >= 0;
  const int xlo = max(s_lo[0] - R, 0), xhi = min(s_hi[0] + R, g.n[0] - 1);
  const int ylo = max(s_lo[1] - R, 0), yhi = min(s_hi[1] + R, g.n[1] - 1);
  const int zlo = max(s_lo[2] - R, 0), zhi = min(s_hi[2] + R, g.n[2] - 1);
  const int ny = yhi - ylo + 1, nz = zhi - zlo + 1;
  const int nrows = any ? ny * nz : 0;
  int *seg_i = s_idx + wave * wcap;
  unsigned short *seg_m = s_m + wave * wcap;
  const unsigned long long below = (1ull << lane) - 1ull;
  int n0 = 0, n1 = 0;
  bool over = false;
  // The candidates of the (y,z) rows of cells are taken as ONE concatenated list, 64 per wave and trip: a row by
  // itself holds 15-40 atoms, so a wave walking row after row had a quarter to a half of its lanes busy in front of
  // two dependent round trips per row.  (The order in which members are found does not matter: tile_sort_kernel
  // sorts every segment by atom index.)
  constexpr int kRows = 256;
  __shared__ int s_pb[kRows], s_off[kRows + 1], s_wsum[4];
  for (int rbase = 0; rbase < nrows; rbase += kRows) {
    const int nr = nrows - rbase < kRows ? nrows - rbase : kRows;
    int pb = 0, len = 0;
    if (tid < nr) {
      const int r = rbase + tid;
      const int y = ylo + r % ny, z = zlo + r / ny;
      // The stencil is a box of cells, what can hold a neighbour is the tile's bounding box grown by the largest list
      // cutoff -- about half of that box: a row of cells farther than the cutoff from the bounding box (in the y-z plane)
      // is left out, and of the others only the cells along x that the remaining distance reaches.  Conservative by
      // construction (cell and box faces, a 1e-9 A margin), so the union is what the full sweep finds.
      const double cy0 = g.lo[1] + y / g.inv[1], cy1 = g.lo[1] + (y + 1) / g.inv[1];
      const double cz0 = g.lo[2] + z / g.inv[2], cz1 = g.lo[2] + (z + 1) / g.inv[2];
      // (the first and the last cell of a dimension also hold what lies beyond the grid: that face is at infinity)
      const double dy = y < g.n[1] - 1 && s_bb[1] > cy1 ? s_bb[1] - cy1 : (y > 0 && cy0 > s_bb[4] ? cy0 - s_bb[4] : 0.0);
      const double dz = z < g.n[2] - 1 && s_bb[2] > cz1 ? s_bb[2] - cz1 : (z > 0 && cz0 > s_bb[5] ? cz0 - s_bb[5] : 0.0);
      const double left = s_cmax - dy * dy - dz * dz; // squared reach along x
      if (left >= -1.0e-9) {
        const double rx = sqrt(left > 0.0 ? left : 0.0) + 1.0e-9;
        int x0 = (int) floor((s_bb[0] - rx - g.lo[0]) * g.inv[0]), x1 = (int) floor((s_bb[3] + rx - g.lo[0]) * g.inv[0]);
        x0 = x0 < xlo ? xlo : x0;
        x1 = x1 > xhi ? xhi : x1;
        if (x0 <= x1) {
          pb = cell_start[x0 + g.n[0] * (y + g.n[1] * z)];
          len = cell_start[x1 + g.n[0] * (y + g.n[1] * z) + 1] - pb;
        }
      }
    }
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; w++) wbase += s_wsum[w];
    const int total = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    s_pb[tid] = pb;
    s_off[tid] = wbase + incl - len;
    if (tid == 0) s_off[kRows] = total;
    __syncthreads();
    // the candidate of the NEXT trip is located and its two loads (position in cell order, atom index) are in flight
    // while the current one is tested against the tile's atoms: a trip was two dependent round trips to memory in
    // front of 32 distance tests, at five waves per SIMD
    const auto locate = [&](const int gi) {
      int lo = 0, hi = kRows; // s_off[lo] <= gi < s_off[hi]  (rows past nr are empty: offset = total)
#pragma unroll
      for (int it = 0; it < 8; it++) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= gi) lo = mid;
        else hi = mid;
      }
      return s_pb[lo] + gi - s_off[lo];
    };
    double4 xn = make_double4(0.0, 0.0, 0.0, -1.0);
    int jn = 0;
    if (wave * 64 + lane < total) {
      const int p = locate(wave * 64 + lane);
      xn = xq_cell[p];
      jn = perm[p];
    }
    for (int g0 = wave * 64; g0 < total && !over; g0 += 256) {
      const int gi = g0 + lane;
      unsigned m = 0;
      int j = 0, tj = 0;
      const double4 xj = xn;
      const int jcur = jn;
      if (gi + 256 < total) {
        const int p = locate(gi + 256);
        xn = xq_cell[p];
        jn = perm[p];
      }
      if (gi < total) {
        j = jcur;
        tj = (int) xj.w;
        tj = tj > 1 ? 1 : tj;
        if (tj >= 0) {
#pragma unroll 8
          for (int a = 0; a < NA; a++) {
            const double4 xa = s_xa[a];
            const double dx = xa.x - xj.x, dy = xa.y - xj.y, dz = xa.z - xj.z;
            if (dx * dx + dy * dy + dz * dz <= s_cut[a][tj]) m |= 1u << (a / CL);
          }
        }
      }
      const bool k0 = m != 0 && tj == 0, k1 = m != 0 && tj != 0;
      const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
      const int c0 = __popcll(b0), c1 = __popcll(b1);
      if (n0 + n1 + c0 + c1 > wcap) {
        over = true;
        break;
      }
      if (k0) {
        const int pos = n0 + __popcll(b0 & below);
        seg_i[pos] = j;
        seg_m[pos] = (unsigned short) m;
      }
      if (k1) {
        const int pos = wcap - 1 - (n1 + __popcll(b1 & below));
        seg_i[pos] = j;
        seg_m[pos] = (unsigned short) m;
      }
      n0 += c0;
      n1 += c1;
    }
    __syncthreads(); // (s_pb / s_off are rewritten by the next block of rows)
  }
  if (lane == 0) {
    s_n0[wave] = n0;
    s_n1[wave] = n1;
    if (over) s_over = 1;
  }
  __syncthreads();
  const int N0 = s_n0[0] + s_n0[1] + s_n0[2] + s_n0[3];
  const int N1 = s_n1[0] + s_n1[1] + s_n1[2] + s_n1[3];
  int nU = N0 + N1;
  if (s_over || nU > cap - 1) { // (slot nU is the kernel's dummy entry, so nU <= cap-1)
    if (tid == 0) {
      atomicOr(&tile_flag[0], 1);
      tile_nu[2 * t] = tile_nu[2 * t + 1] = 0;
    }
    if (tid < MDP_TILE) cnt[t * MDP_TILE + tid] = split[t * MDP_TILE + tid] = 0;
    return;
  }
  int base0 = 0, base1 = N0;
  for (int w = 0; w < wave; w++) {
    base0 += s_n0[w];
    base1 += s_n1[w];
  }
  int *mem = tu + (size_t) t * cap;
  unsigned short *mm = tmask + (size_t) t * cap;
  for (int i = lane; i < n0; i += 64) {
    const int u = base0 + i;
    const unsigned short m = seg_m[i];
    mem[u] = seg_i[i];
    mm[u] = m;
    s_fm[u] = m;
  }
  for (int i = lane; i < n1; i += 64) {
    const int u = base1 + i, src = wcap - 1 - i;
    const unsigned short m = seg_m[src];
    mem[u] = seg_i[src];
    mm[u] = m;
    s_fm[u] = m;
  }
  __syncthreads();
  const int gq = tid / 16, sq = tid % 16;
  int c0 = 0, c1 = 0;
  for (int u = sq; u < nU; u += 16) {
    const int bit = (s_fm[u] >> gq) & 1;
    c0 += u < N0 ? bit : 0;
    c1 += u < N0 ? 0 : bit;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    c0 += __shfl_xor(c0, o, 64);
    c1 += __shfl_xor(c1, o, 64);
  }
  // Row layout: every tile has MDP_TILE rows (absent clusters get all-dummy rows); both segments are padded
  // with the dummy index to whole 16-lane steps AND to the longest of the four clusters that share a wave
  // in the compute kernel, so that its loops are wave-uniform (idle lanes would cost the same time).
  int p0 = (c0 + 15) & ~15, p1 = (c1 + 15) & ~15;
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    p0 = max(p0, __shfl_xor(p0, o, 64));
    p1 = max(p1, __shfl_xor(p1, o, 64));
  }
  const int kc = t * MDP_TILE + gq;
  if (sq == 0) {
    cnt[kc] = p0 + p1;
    split[kc] = p0;
    atomicAdd(&s_rowsum, p0 + p1);
  }
  __syncthreads();
  if (tid == 0) {
    tile_nu[2 * t] = nU;
    tile_nu[2 * t + 1] = N0;
    // running maxima over all tiles, two words for the whole grid.  A look first -- the word only grows, so a stale smaller
    // value costs at most a redundant atomic -- leaves a handful of atomics per launch instead of two per tile on ONE
    // address (the look goes to the L2, where the atomics land: a CU's L1 would keep the first value it saw).  Measured:
    // no difference at 124 416 tiles (the atomics are fire-and-forget at the end of a workgroup); the per-WAVE atomics
    // with a return value in classify_kernel were the ones that cost a millisecond.
    if (nU > __hip_atomic_load(&tile_flag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&tile_flag[1], nU);
    if (s_rowsum > __hip_atomic_load(&tile_flag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(&tile_flag[2], s_rowsum); // row entries of the whole tile (kernels that stage a tile's rows in LDS)
  }
}

// Between the passes: the members of a union are sorted by atom index (each element segment by itself).  The
// kernels that stage a union gather xq[member] with consecutive lanes taking consecutive members; atoms are stored
// along a Hilbert curve, so after the sort neighbouring lanes mostly hit the same 128-byte line (four atoms) instead
// of one line each -- the staging gathers were a quarter of the L1 lookups of the AEAM tile kernels.
__global__ __launch_bounds__(256) void tile_sort_kernel(const int cap, const int *__restrict__ tile_nu,
                                                        int *__restrict__ tu, unsigned short *__restrict__ tmask)
{
  extern __shared__ unsigned long long s_key[];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int nU = tile_nu[2 * t], N0 = tile_nu[2 * t + 1];
  int *mem = tu + (size_t) t * cap;
  unsigned short *mm = tmask + (size_t) t * cap;
  for (int seg = 0; seg < 2; seg++) {
    const int b = seg ? N0 : 0, n = (seg ? nU : N0) - b;
    if (n <= 1) continue; // (block-uniform)
    int np = 2;
    while (np < n) np <<= 1;
    for (int i = tid; i < np; i += 256)
      s_key[i] = i < n ? (((unsigned long long) (unsigned) mem[b + i]) << 16) | mm[b + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        // every thread takes whole compare-exchange pairs (i, i + j): q-th pair, i = q with a zero inserted at bit log2 j
        for (int q = tid; q < (np >> 1); q += 256) {
          const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), p = i | j;
          const unsigned long long x = s_key[i], y = s_key[p];
          if ((x > y) == ((i & k) == 0)) {
            s_key[i] = y;
            s_key[p] = x;
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < n; i += 256) {
      const unsigned long long v = s_key[i];
      mem[b + i] = (int) (v >> 16);
      mm[b + i] = (unsigned short) (v & 0xFFFFull);
    }
    __syncthreads();
  }
}

// tile_sort_kernel and tile_fill_kernel (below) in one launch: the sorted masks stay in LDS and the rows are filled from
// there -- the masks of a tile are needed by nothing else, so their way back to global memory and a launch with its
// dependent loads (tile -> union size -> masks) are saved.
__global__ __launch_bounds__(256) void tile_sort_fill_kernel(const int cap, const int np_max, const int *__restrict__ tile_nu,
                                                             int *__restrict__ tu, const unsigned short *__restrict__ tmask,
                                                             const int nclus, const long long *__restrict__ off,
                                                             const int *__restrict__ split, unsigned short *__restrict__ lj16)
{
  extern __shared__ unsigned long long s_key[];                                   // [np_max] keys of the segment being sorted
  unsigned short *s_fm = reinterpret_cast<unsigned short *>(s_key + np_max);      // [cap] masks in final order
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int nU = tile_nu[2 * t], N0 = tile_nu[2 * t + 1];
  int *mem = tu + (size_t) t * cap;
  const unsigned short *mm = tmask + (size_t) t * cap;
  // (everything the fill needs from global memory is requested before the sort)
  const int gq = tid / 16, sq = tid % 16, glane0 = lane - sq;
  const int kc = t * MDP_TILE + gq;
  const bool have = kc < nclus;
  const long long b0 = off[kc];
  const int len[2] = {split[kc], (int) (off[kc + 1] - b0) - split[kc]}; // padded segment lengths (tile_scan_kernel)
  for (int seg = 0; seg < 2; seg++) {
    const int b = seg ? N0 : 0, n = (seg ? nU : N0) - b;
    if (n <= 1) { // (block-uniform)
      if (n == 1 && tid == 0) s_fm[b] = mm[b];
      continue;
    }
    int np = 2;
    while (np < n) np <<= 1;
    for (int i = tid; i < np; i += 256)
      s_key[i] = i < n ? (((unsigned long long) (unsigned) mem[b + i]) << 16) | mm[b + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int q = tid; q < (np >> 1); q += 256) {
          const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), p = i | j;
          const unsigned long long x = s_key[i], y = s_key[p];
          if ((x > y) == ((i & k) == 0)) {
            s_key[i] = y;
            s_key[p] = x;
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < n; i += 256) {
      const unsigned long long v = s_key[i];
      mem[b + i] = (int) (v >> 16);
      s_fm[b + i] = (unsigned short) (v & 0xFFFFull);
    }
    __syncthreads();
  }
  __syncthreads();
  unsigned short *row = lj16 + b0;
  const unsigned long long below = (1ull << sq) - 1ull;
  for (int seg = 0; seg < 2; seg++) {
    const int ub = seg ? N0 : 0, ue = seg ? nU : N0;
    int n = 0;
    for (int base = ub; base < ue; base += 16) {
      const int u = base + sq;
      const bool k = have && u < ue && ((s_fm[u < ue ? u : ub] >> gq) & 1);
      const unsigned long long bal = (__ballot(k) >> glane0) & 0xFFFFull;
      if (k) row[n + __popcll(bal & below)] = (unsigned short) u;
      n += __popcll(bal);
    }
    for (int q = n + sq; q < len[seg]; q += 16) row[q] = (unsigned short) nU; // padding: the dummy slot
    row += len[seg];
  }
}

// sorts the unions; with `fill` also writes the rows (then tile_fill_kernel must not follow).  *filled tells the caller.
static int tile_sort_launch(mdp_ctx *c, int ntile, int nclus = 0, bool fill = false, bool *filled = nullptr)
{
  if (filled) *filled = false;
  if (const char *e = getenv("MDP_TILE_SORT"))
    if (atoi(e) == 0) return MDP_OK;
  int np = 2;
  while (np < c->tile_maxu) np <<= 1;
  if (fill) {
    const size_t lds = (size_t) np * sizeof(unsigned long long) + (size_t) c->tile_cap * sizeof(unsigned short);
    if (lds > 48 * 1024)
      MDP_HIP(c, hipFuncSetAttribute((const void *) tile_sort_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    tile_sort_fill_kernel<<<ntile, 256, lds, c->stream>>>(c->tile_cap, np, c->tile_nu.p, c->tu.p, c->tmask.p, nclus,
                                                          c->lj_off.p, c->lj_split.p, c->lj16.p);
    MDP_HIP(c, hipGetLastError());
    if (filled) *filled = true;
    return MDP_OK;
  }
  const size_t lds = (size_t) np * sizeof(unsigned long long);
  if (lds > 48 * 1024)
    MDP_HIP(c, hipFuncSetAttribute((const void *) tile_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  tile_sort_kernel<<<ntile, 256, lds, c->stream>>>(c->tile_cap, c->tile_nu.p, c->tu.p, c->tmask.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// Pass 2: each cluster (16 lanes) walks its tile's masks in union order and keeps the entries with its bit;
// the Mo segment and the S segment are each padded to a whole 16-lane step with the dummy index nU
__global__ __launch_bounds__(256) void tile_fill_kernel(const int nclus, const int cap,
                                                        const int *__restrict__ tile_nu,
                                                        const unsigned short *__restrict__ tmask,
                                                        const long long *__restrict__ off,
                                                        const int *__restrict__ split,
                                                        unsigned short *__restrict__ lj16)
{
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int gq = tid / 16, sq = tid % 16, glane0 = lane - sq;
  const int kc = t * MDP_TILE + gq;
  const bool have = kc < nclus;
  const int nU = tile_nu[2 * t], N0 = tile_nu[2 * t + 1];
  const unsigned short *mm = tmask + (size_t) t * cap;
  const long long b = off[kc];
  const int len[2] = {split[kc], (int) (off[kc + 1] - b) - split[kc]}; // padded segment lengths (tile_scan_kernel)
  unsigned short *row = lj16 + b;
  const unsigned long long below = (1ull << sq) - 1ull;
  for (int seg = 0; seg < 2; seg++) {
    const int ub = seg ? N0 : 0, ue = seg ? nU : N0;
    int n = 0;
    for (int base = ub; base < ue; base += 16) {
      const int u = base + sq;
      const bool k = have && u < ue && ((mm[u] >> gq) & 1);
      const unsigned long long bal = (__ballot(k) >> glane0) & 0xFFFFull;
      if (k) row[n + __popcll(bal & below)] = (unsigned short) u;
      n += __popcll(bal);
    }
    for (int q = n + sq; q < len[seg]; q += 16) row[q] = (unsigned short) nU; // padding: the dummy slot
    row += len[seg];
  }
}

// Dynamic pruning of the rows (between list builds): every row entry is tested against the cluster's atoms at
// the CURRENT positions and kept when it lies within (upper window bound + buffer) of one of them; the kept entries
// are compacted into a second set of rows at the same offsets (segments padded as tile_scan_kernel pads them).
// The list skin -- a third of the entries of a list built with 1 A of it -- then costs a pass of this kernel every
// few tens of steps instead of an evaluation in every step; the rows as built stay, the buffer has its own
// displacement trigger (moved_kernel), and an entry is never missed: it enters a window only after the two atoms
// together moved the buffer, i.e. one of them half of it.
struct PruneLimits {
  double rsq[4]; // (window upper bound + buffer)^2 per pair type ti * 2 + tj
};
template <int CL>
__global__ __launch_bounds__(256) void tile_prune_kernel(const PruneLimits lim, const int nlocal, const int nclus,
                                                         const double4 *__restrict__ xq, const int cap, const int capL,
                                                         const int *__restrict__ tu, const int *__restrict__ tile_nu,
                                                         const long long *__restrict__ lj_off,
                                                         const int *__restrict__ lj_split,
                                                         const unsigned short *__restrict__ lj16,
                                                         unsigned short *__restrict__ lj16_in,
                                                         int *__restrict__ len_in, int *__restrict__ split_in)
{
  // The tile's rows (contiguous in memory) are staged in LDS with 16-byte loads, compacted there in place -- a
  // row's kept entries never outrun its read position, and a row belongs to one 16-lane group of one wave -- and
  // written back whole with 16-byte stores: no 2-byte global traffic, no global load inside the loop.  What lies
  // behind a row's new length stays what it was (valid indices: the kernels read a few entries past a segment).
  constexpr int L = 16, SK = 3;
  extern __shared__ double s_pos[]; // [capL][3], then the rows
  unsigned short *__restrict__ s_rows = reinterpret_cast<unsigned short *>(s_pos + 3 * (size_t) capL);
  const int tid = threadIdx.x, lane = tid & 63, s = lane % L, glane0 = lane - s;
  const int t = blockIdx.x;
  const int kc = t * MDP_TILE + tid / L;
  const int nU = tile_nu[2 * t];
  const int *__restrict__ mem = tu + (size_t) t * cap;
  const long long rb = lj_off[(size_t) t * MDP_TILE];
  const int rtot = (int) (lj_off[(size_t) t * MDP_TILE + MDP_TILE] - rb); // entries of the whole tile (multiple of 16)
  {
    int sidx[SK];
#pragma unroll
    for (int k = 0; k < SK; k++) sidx[k] = mem[tid + 256 * k]; // (rows are cap >= 2048 long: always in bounds)
    double4 sv[SK];
#pragma unroll
    for (int k = 0; k < SK; k++) sv[k] = xq[tid + 256 * k < nU ? sidx[k] : 0];
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(lj16 + rb);
    uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(s_rows);
    for (int e = tid; e * 8 < rtot; e += 256) dst[e] = src[e];
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int u = tid + 256 * k;
      if (u < nU) {
        s_pos[3 * u] = sv[k].x;
        s_pos[3 * u + 1] = sv[k].y;
        s_pos[3 * u + 2] = sv[k].z;
      }
    }
    for (int u = tid + 256 * SK; u < nU; u += 256) {
      const double4 v = xq[mem[u]];
      s_pos[3 * u] = v.x;
      s_pos[3 * u + 1] = v.y;
      s_pos[3 * u + 2] = v.z;
    }
  }
  const long long b = lj_off[kc];
  const int cnt = __builtin_amdgcn_readfirstlane((int) (lj_off[kc + 1] - b));
  const int split = __builtin_amdgcn_readfirstlane(lj_split[kc]);
  unsigned short *__restrict__ row = s_rows + (int) (b - rb);
  double4 xa[CL];
  bool real[CL];
  int ta[CL];
#pragma unroll
  for (int c = 0; c < CL; c++) {
    const int ia = kc * CL + c;
    real[c] = kc < nclus && ia < nlocal;
    xa[c] = xq[real[c] ? ia : nlocal - 1];
    ta[c] = (int) xa[c].w;
    if (ta[c] < 0) { // type mapped to NULL: takes part in nothing
      real[c] = false;
      ta[c] = 0;
    }
    ta[c] = ta[c] > 1 ? 1 : ta[c]; // (the two classes of tile_scan_kernel)
  }
  __syncthreads();
  const unsigned long long below = (1ull << s) - 1ull;
  int pseg[2];
  int base = 0;
#pragma unroll
  for (int seg = 0; seg < 2; seg++) {
    const int kb = seg ? split : 0, ke = seg ? cnt : split;
    double lim_c[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) lim_c[c] = lim.rsq[ta[c] * 2 + seg];
    int n = 0;
    for (int k = kb; k < ke; k += L) { // (segments are whole 16-lane steps, equally long for the rows of a wave)
      const int li = (int) row[k + s];
      bool keep = false;
      if (li < nU) {
        const double xj = s_pos[3 * li], yj = s_pos[3 * li + 1], zj = s_pos[3 * li + 2];
#pragma unroll
        for (int c = 0; c < CL; c++) {
          const double dx = xa[c].x - xj, dy = xa[c].y - yj, dz = xa[c].z - zj;
          keep = keep || (real[c] && dx * dx + dy * dy + dz * dz <= lim_c[c]);
        }
      }
      const unsigned long long gb = (__ballot(keep) >> glane0) & 0xFFFFull;
      // (the read of this trip is complete for the whole wave before any lane writes: same instruction stream)
      if (keep) row[base + n + __popcll(gb & below)] = (unsigned short) li;
      n += __popcll(gb);
    }
    int p = (n + 15) & ~15;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) p = max(p, __shfl_xor(p, o, 64));
    for (int q = n + s; q < p; q += L) row[base + q] = (unsigned short) nU; // padding: the dummy slot
    pseg[seg] = p;
    base += p;
  }
  if (s == 0) {
    split_in[kc] = pseg[0];
    len_in[kc] = pseg[0] + pseg[1];
  }
  __syncthreads();
  {
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(s_rows);
    uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(lj16_in + rb);
    for (int e = tid; e * 8 < rtot; e += 256) dst[e] = src[e];
  }
}

// does the tile's union reach a remote ghost?  Such tiles wait for the halo.
__global__ __launch_bounds__(256) void tile_boundary_kernel(const int ntile, const int cap, const int remote_start,
                                                            const int *__restrict__ tile_nu,
                                                            const int *__restrict__ tu, int *__restrict__ is_int,
                                                            int *__restrict__ is_bnd)
{
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntile) return;
  const int *mem = tu + (size_t) t * cap;
  const int nU = tile_nu[2 * t];
  int hit = 0;
  for (int u = lane; u < nU; u += 64) hit |= mem[u] >= remote_start;
  hit = __any(hit);
  if (lane == 0) {
    is_int[t] = !hit;
    is_bnd[t] = hit;
  }
}

// ------------------------------------------------------------------------------------------------
// repack at every neighbor (re)build: master CSR list -> REBO candidates + trimmed LJ list
// ------------------------------------------------------------------------------------------------
constexpr int RP_L = 16; // lanes per atom in the list-building kernels

// REBO candidate lists (r <= rcmax + inner skin) straight from the bin grid.
// MODE 0: count for owned atoms and mark the ghost atoms that neighbour them (they are centres too)
// MODE 1: count for the marked ghosts            MODE 2: fill (all atoms with a non-empty row)
// One sweep instead of two: MODES 0 and 1 also WRITE what they count, into rows of a fixed stride kCandStride (`off`
// null, `cand` = the staging rows; a row holds what the 64-bit active mask can address, longer rows stop the build
// anyway), and cand_compact_kernel moves them to their CSR offsets once the counts are scanned -- same candidates in
// the same order as the second sweep (MODE 2, kept for MDP_CAND_TWO_PASS=1) writes.
constexpr int kCandStride = 64;
template <int MODE>
__global__ __launch_bounds__(256) void cand_build_kernel(const MdpGrid g, const int R, const RebomosDev P,
                                                         const int nall, const int nlocal,
                                                         const double4 *__restrict__ xq, const int *__restrict__ perm,
                                                         const int *__restrict__ cell_start, int *__restrict__ cnt,
                                                         const int *__restrict__ off, int *__restrict__ cand,
                                                         int *__restrict__ is_centre,
                                                         const int mark_from /* ghosts below it are periodic images of owned atoms: no centres (rev_kernel) */,
                                                         const double4 *__restrict__ xq_cell /* xq[perm[p]]: coalesced, and no load depends on another */)
{
  const int lane = threadIdx.x & 63;
  const int s = lane % RP_L;
  const int glane0 = lane - s;
  const long long i64 = (long long) blockIdx.x * (256 / RP_L) + threadIdx.x / RP_L + (MODE == 1 ? nlocal : 0);
  const int ihi = MODE == 0 ? nlocal : nall;
  bool have = i64 < ihi;
  const int i = have ? (int) i64 : 0;
  if (MODE == 1 && have && !is_centre[i]) have = false; // ghosts far from every owned atom need no row
  if (MODE == 2 && have && off[i + 1] == off[i]) have = false;
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  if (ti < 0) have = false; // type mapped to NULL (pair_rebomos.cpp:169-171): not part of this style
  int cx = (int) ((xi.x - g.lo[0]) * g.inv[0]), cy = (int) ((xi.y - g.lo[1]) * g.inv[1]),
      cz = (int) ((xi.z - g.lo[2]) * g.inv[2]);
  cx = cx < 0 ? 0 : (cx >= g.n[0] ? g.n[0] - 1 : cx);
  cy = cy < 0 ? 0 : (cy >= g.n[1] ? g.n[1] - 1 : cy);
  cz = cz < 0 ? 0 : (cz >= g.n[2] ? g.n[2] - 1 : cz);
  int zlo = max(cz - R, 0), zhi = min(cz + R, g.n[2] - 1), ylo = max(cy - R, 0), yhi = min(cy + R, g.n[1] - 1);
  const int xlo = max(cx - R, 0), xhi = min(cx + R, g.n[0] - 1);
  if (!have) {
    zlo = ylo = 1 << 30;
    zhi = yhi = -1;
  }
  int wzlo = zlo, wzhi = zhi, wylo = ylo, wyhi = yhi;
#pragma unroll
  for (int o = 32; o >= RP_L; o >>= 1) {
    wzlo = min(wzlo, __shfl_xor(wzlo, o, 64));
    wzhi = max(wzhi, __shfl_xor(wzhi, o, 64));
    wylo = min(wylo, __shfl_xor(wylo, o, 64));
    wyhi = max(wyhi, __shfl_xor(wyhi, o, 64));
  }
  int n = 0;
  int *row = (MODE == 2 && have) ? cand + off[i] : nullptr;
  int *stage = (MODE != 2 && have && cand) ? cand + (size_t) i * kCandStride : nullptr;
  const unsigned long long below = (1ull << s) - 1ull;
  // the stencil is a box of cells around the atom's cell, what can hold a candidate is a sphere: a row of cells out of
  // the atom's reach in the y-z plane is skipped, of the others only the cells along x that the remaining distance
  // reaches are read (conservative: the first / last cell of a dimension also holds what lies beyond the grid, so that
  // face is at infinity) -- a little under half of the box.  All of it in single precision, in units of cells, rounded
  // outward by 1e-3 of a cell (float resolves 6e-5 cells at the 1024 cells a dimension has at most): a few instructions per row.
  const float reach = have ? (float) sqrt(fmax(P.cand_cutsq[ti * 2], P.cand_cutsq[ti * 2 + 1])) * 1.00001f : 0.0f;
  const float ivx = (float) g.inv[0], ivy = (float) g.inv[1], ivz = (float) g.inv[2];
  const float fx = (float) ((xi.x - g.lo[0]) * g.inv[0]), fy = (float) ((xi.y - g.lo[1]) * g.inv[1]),
              fz = (float) ((xi.z - g.lo[2]) * g.inv[2]); // the atom in cell coordinates
  const float iry = 1.0f / (reach * ivy + 1.0e-3f), irz = 1.0f / (reach * ivz + 1.0e-3f); // 1 / reach in cells
  for (int z = wzlo; z <= wzhi; z++)
    for (int y = wylo; y <= wyhi; y++) {
      bool rowin = have && z >= zlo && z <= zhi && y >= ylo && y <= yhi;
      int pb = 0, pe = 0;
      if (rowin) {
        // distance of the atom from the row's cells in y and z, in cells (0 inside the row; the first / last cell of a
        // dimension reaches to infinity on its outer side), as fractions of the reach
        const float dy = y < g.n[1] - 1 && fy > (float) (y + 1) ? fy - (float) (y + 1) : (y > 0 && fy < (float) y ? (float) y - fy : 0.0f);
        const float dz = z < g.n[2] - 1 && fz > (float) (z + 1) ? fz - (float) (z + 1) : (z > 0 && fz < (float) z ? (float) z - fz : 0.0f);
        const float qy = dy * iry, qz = dz * irz;
        const float left = 1.0f - qy * qy - qz * qz;
        rowin = left >= -1.0e-5f;
        if (rowin) {
          const float rxc = (reach * ivx) * sqrtf(left > 0.0f ? left : 0.0f) + 2.0e-3f; // reach along x in cells
          int x0 = (int) floorf(fx - rxc), x1 = (int) floorf(fx + rxc);
          x0 = x0 < xlo ? xlo : x0;
          x1 = x1 > xhi ? xhi : x1;
          if (x0 <= x1) {
            pb = cell_start[x0 + g.n[0] * (y + g.n[1] * z)];
            pe = cell_start[x1 + g.n[0] * (y + g.n[1] * z) + 1];
          }
        }
      }
      const int lenw = wave_max_int(pe - pb);
      for (int base = 0; base < lenw; base += RP_L) {
        const int p = pb + base + s;
        bool keep = false;
        int j = 0;
        if (p < pe) {
          j = perm[p];
          const double4 xj = xq_cell[p];
          const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
          const int tj = (int) xj.w;
          keep = j != i && tj >= 0 && (dx * dx + dy * dy + dz * dz) <= P.cand_cutsq[ti * 2 + tj];
        }
        const unsigned long long bk = (__ballot(keep) >> glane0) & 0xFFFFull;
        if (keep) {
          if (MODE == 2) row[n + __popcll(bk & below)] = j;
          if (MODE != 2 && stage && n + __popcll(bk & below) < kCandStride) stage[n + __popcll(bk & below)] = j;
          if (MODE == 0 && j >= mark_from) is_centre[j] = 1;
        }
        n += __popcll(bk);
      }
    }
  if (MODE != 2 && have && s == 0) cnt[i] = n;
}

// staging rows -> CSR (RP_L lanes per atom; rows longer than the stride are truncated here and refused by rev_kernel)
__global__ __launch_bounds__(256) void cand_compact_kernel(const int nall, const int *__restrict__ off,
                                                           const int *__restrict__ stage, int *__restrict__ cand)
{
  const int s = threadIdx.x % RP_L;
  const long long i64 = (long long) blockIdx.x * (256 / RP_L) + threadIdx.x / RP_L;
  if (i64 >= nall) return;
  const int i = (int) i64, b = off[i];
  const int n = off[i + 1] - b;
  const int *__restrict__ src = stage + (size_t) i * kCandStride;
  // (a row longer than the stride: the build is refused by rev_kernel, but the kernels queued behind this one walk the
  //  row first -- its tail repeats a valid index instead of holding whatever the buffer held)
  for (int k = s; k < n; k += RP_L) cand[b + k] = src[k < kCandStride ? k : kCandStride - 1];
}

// ---- lists from the HOST's neighbor list (MDP_REBOMOS_HOST_LIST=1, mdp_rebomos_host_list) -------------------------
// The reference walks the rows LAMMPS built (REBO_neigh, pair_rebomos.cpp:328-330; FLJ, :490-495, both masked with
// NEIGHMASK), so whatever the host left out of them -- `neigh_modify exclude`, special bonds with weight 0 -- is no
// pair of the style.  These two kernels take their candidates from those rows (CSR copy on the device,
// mdp_set_neighbors_host) instead of the bin grid and produce exactly what cand_build_kernel / tile_scan_kernel
// produce; everything behind them is shared.  Rows of ghost atoms are used where the host has them (REQ_GHOST).
template <int MODE> // 0: owned atoms (+ marks the ghosts that neighbour them), 1: the marked ghosts
__global__ __launch_bounds__(256) void cand_csr_kernel(const RebomosDev P, const int nall, const int nlocal,
                                                       const double4 *__restrict__ xq,
                                                       const long long *__restrict__ nb_off,
                                                       const int *__restrict__ nb, int *__restrict__ cnt,
                                                       int *__restrict__ stage_all, int *__restrict__ is_centre,
                                                       const int mark_from)
{
  const int lane = threadIdx.x & 63;
  const int s = lane % RP_L;
  const int glane0 = lane - s;
  const long long i64 = (long long) blockIdx.x * (256 / RP_L) + threadIdx.x / RP_L + (MODE == 1 ? nlocal : 0);
  bool have = i64 < (MODE == 0 ? nlocal : nall);
  const int i = have ? (int) i64 : 0;
  if (MODE == 1 && have && !is_centre[i]) have = false;
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  if (ti < 0) have = false;
  const long long b = have ? nb_off[i] : 0;
  const int len = have ? (int) (nb_off[i + 1] - b) : 0;
  const int lenw = wave_max_int(len);
  int *stage = have ? stage_all + (size_t) i * kCandStride : nullptr;
  const unsigned long long below = (1ull << s) - 1ull;
  int n = 0;
  for (int base = 0; base < lenw; base += RP_L) {
    const int k = base + s;
    bool keep = false;
    int j = 0;
    if (k < len) {
      j = nb[b + k];
      const double4 xj = xq[j];
      const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
      const int tj = (int) xj.w;
      keep = j != i && tj >= 0 && (dx * dx + dy * dy + dz * dz) <= P.cand_cutsq[ti * 2 + tj];
    }
    const unsigned long long bk = (__ballot(keep) >> glane0) & 0xFFFFull;
    if (keep) {
      const int pos = n + __popcll(bk & below);
      if (pos < kCandStride) stage[pos] = j;
      if (MODE == 0 && j >= mark_from) is_centre[j] = 1;
    }
    n += __popcll(bk);
  }
  if (have && s == 0) cnt[i] = n;
}

// tile_scan_kernel with the union gathered from the host rows of the tile's atoms: the same atom is listed by many of
// them, so the members go through a hash table in LDS (key = atom index, value = the 16-bit cluster mask | element
// class << 16) and are compacted element 0 first.  tile_sort_kernel orders every segment by atom index afterwards, so
// the order of insertion leaves no trace.
template <int CL>
__global__ __launch_bounds__(256) void tile_scan_csr_kernel(const RebomosDev P, const int nclus, const int nlocal,
                                                            const double4 *__restrict__ xq,
                                                            const long long *__restrict__ nb_off,
                                                            const int *__restrict__ nb, const int cap,
                                                            int *__restrict__ tu, unsigned short *__restrict__ tmask,
                                                            int *__restrict__ tile_nu, int *__restrict__ cnt,
                                                            int *__restrict__ split, int *__restrict__ tile_flag)
{
  constexpr int NA = MDP_TILE * CL;
  extern __shared__ int s_dyn[];
  const int HS = 2 * cap; // (power of two: cap is)
  int *s_key = s_dyn;                                       // [HS]
  int *s_val = s_dyn + HS;                                  // [HS]
  unsigned short *s_fm = (unsigned short *) (s_dyn + 2 * HS); // [cap] masks in final order
  __shared__ double4 s_xa[NA];
  __shared__ double s_cut[NA][2];
  __shared__ int s_count, s_over, s_rowsum, s_w0[4], s_w1[4];
  const int t = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int h = tid; h < HS; h += 256) {
    s_key[h] = -1;
    s_val[h] = 0;
  }
  if (tid == 0) s_count = s_over = s_rowsum = 0;
  if (tid < NA) {
    const int ia = t * NA + tid;
    bool valid = ia < nlocal;
    const double4 xa = xq[valid ? ia : (nlocal > 0 ? nlocal - 1 : 0)];
    int ta = (int) xa.w;
    if (ta < 0) valid = false;
    ta = ta > 1 ? 1 : ta;
    s_xa[tid] = xa;
    s_cut[tid][0] = valid ? P.ljlist_cutsq[ta * 2 + 0] : -1.0;
    s_cut[tid][1] = valid ? P.ljlist_cutsq[ta * 2 + 1] : -1.0;
  }
  __syncthreads();
  // every wave walks the rows of NA/4 atoms, 64 entries a trip
  for (int a = wave; a < NA; a += 4) {
    const int ia = t * NA + a;
    if (ia >= nlocal || s_cut[a][0] < 0.0) continue; // (wave-uniform)
    const double4 xa = s_xa[a];
    const long long b = nb_off[ia];
    const int len = (int) (nb_off[ia + 1] - b);
    for (int k = lane; k < len; k += 64) {
      const int j = nb[b + k];
      const double4 xj = xq[j];
      int tj = (int) xj.w;
      tj = tj > 1 ? 1 : tj;
      if (tj < 0 || j == ia) continue; // (the host lists no atom in its own row; guarded anyway)
      const double dx = xa.x - xj.x, dy = xa.y - xj.y, dz = xa.z - xj.z;
      if (!(dx * dx + dy * dy + dz * dz <= s_cut[a][tj])) continue;
      const int val = (1 << (a / CL)) | (tj << 16);
      unsigned h = ((unsigned) j * 2654435761u) & (unsigned) (HS - 1);
      for (int probe = 0; probe < HS; probe++) {
        const int old = atomicCAS(&s_key[h], -1, j);
        if (old == -1) {
          if (atomicAdd(&s_count, 1) >= cap - 1) s_over = 1; // (slot nU is the kernels' dummy entry: nU <= cap - 1)
        }
        if (old == -1 || old == j) {
          atomicOr(&s_val[h], val);
          break;
        }
        h = (h + 1) & (unsigned) (HS - 1);
      }
    }
  }
  __syncthreads();
  if (s_over) {
    if (tid == 0) {
      atomicOr(&tile_flag[0], 1);
      tile_nu[2 * t] = tile_nu[2 * t + 1] = 0;
    }
    if (tid < MDP_TILE) cnt[t * MDP_TILE + tid] = split[t * MDP_TILE + tid] = 0;
    return;
  }
  // compaction: thread tid owns the slots [tid * HS/256, (tid+1) * HS/256)
  const int per = HS / 256;
  int m0 = 0, m1 = 0;
  for (int q = 0; q < per; q++) {
    const int h = tid * per + q;
    if (s_key[h] >= 0) {
      if ((s_val[h] >> 16) & 1) m1++;
      else m0++;
    }
  }
  int i0 = m0, i1 = m1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u0 = __shfl_up(i0, o, 64), u1 = __shfl_up(i1, o, 64);
    if (lane >= o) {
      i0 += u0;
      i1 += u1;
    }
  }
  if (lane == 63) {
    s_w0[wave] = i0;
    s_w1[wave] = i1;
  }
  __syncthreads();
  const int N0 = s_w0[0] + s_w0[1] + s_w0[2] + s_w0[3];
  const int N1 = s_w1[0] + s_w1[1] + s_w1[2] + s_w1[3];
  const int nU = N0 + N1;
  int p0 = i0 - m0, p1 = N0 + i1 - m1;
  for (int w = 0; w < wave; w++) {
    p0 += s_w0[w];
    p1 += s_w1[w];
  }
  int *mem = tu + (size_t) t * cap;
  unsigned short *mm = tmask + (size_t) t * cap;
  for (int q = 0; q < per; q++) {
    const int h = tid * per + q;
    const int j = s_key[h];
    if (j < 0) continue;
    const int v = s_val[h];
    const int u = ((v >> 16) & 1) ? p1++ : p0++;
    mem[u] = j;
    mm[u] = (unsigned short) (v & 0xFFFF);
    s_fm[u] = (unsigned short) (v & 0xFFFF);
  }
  __syncthreads();
  // row lengths: as tile_scan_kernel
  const int gq = tid / 16, sq = tid % 16;
  int c0 = 0, c1 = 0;
  for (int u = sq; u < nU; u += 16) {
    const int bit = (s_fm[u] >> gq) & 1;
    c0 += u < N0 ? bit : 0;
    c1 += u < N0 ? 0 : bit;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    c0 += __shfl_xor(c0, o, 64);
    c1 += __shfl_xor(c1, o, 64);
  }
  int q0 = (c0 + 15) & ~15, q1 = (c1 + 15) & ~15;
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    q0 = max(q0, __shfl_xor(q0, o, 64));
    q1 = max(q1, __shfl_xor(q1, o, 64));
  }
  const int kc = t * MDP_TILE + gq;
  if (sq == 0) {
    cnt[kc] = q0 + q1;
    split[kc] = q0;
    atomicAdd(&s_rowsum, q0 + q1);
  }
  __syncthreads();
  if (tid == 0) {
    tile_nu[2 * t] = nU;
    tile_nu[2 * t + 1] = N0;
    if (nU > __hip_atomic_load(&tile_flag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&tile_flag[1], nU);
    if (s_rowsum > __hip_atomic_load(&tile_flag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&tile_flag[2], s_rowsum);
  }
}

// positions at list-build time (all atoms, ghosts included) and the displacement trigger
__global__ void hold_all_kernel(const int nall, const double4 *__restrict__ xq, mdp_hold_t *__restrict__ xhold)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nall) return;
  const double4 x = xq[i];
  xhold[3 * (size_t) i] = (mdp_hold_t) x.x;
  xhold[3 * (size_t) i + 1] = (mdp_hold_t) x.y;
  xhold[3 * (size_t) i + 2] = (mdp_hold_t) x.z;
}

// flag[0]: someone moved beyond the trigger; flag[1]: beyond the hard limit (half the inner skin)
// flag[2], flag[3]: the same against the positions of the last pruning of the rows (xprune, may be null)
__global__ __launch_bounds__(256) void moved_kernel(const int nall, const double trigsq, const double hardsq,
                                                    const double4 *__restrict__ xq,
                                                    const mdp_hold_t *__restrict__ xhold, int *__restrict__ flag,
                                                    const mdp_hold_t *__restrict__ xprune, const double ptrigsq,
                                                    const double phardsq)
{
  bool far = false, toofar = false, pfar = false, ptoofar = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nall; i += gridDim.x * 256) {
    const double4 x = xq[i];
    const double dx = x.x - xhold[3 * (size_t) i], dy = x.y - xhold[3 * (size_t) i + 1],
                 dz = x.z - xhold[3 * (size_t) i + 2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    far = far || d2 > trigsq;
    toofar = toofar || d2 > hardsq;
    if (xprune) {
      const double px = x.x - xprune[3 * (size_t) i], py = x.y - xprune[3 * (size_t) i + 1],
                   pz = x.z - xprune[3 * (size_t) i + 2];
      const double p2 = px * px + py * py + pz * pz;
      pfar = pfar || p2 > ptrigsq;
      ptoofar = ptoofar || p2 > phardsq;
    }
  }
  if (xprune) {
    if (__any(pfar) && (threadIdx.x & 63) == 0) flag[2] = 1;
    if (__any(ptoofar) && (threadIdx.x & 63) == 0) flag[3] = 1;
  }
  // `flag` is pinned HOST memory (zeroed by the host before the launch): plain idempotent stores, visible when the
  // kernel has completed -- no memset and no copy engine in the per-step path
  if (__any(far) && (threadIdx.x & 63) == 0) flag[0] = 1;
  if (__any(toofar) && (threadIdx.x & 63) == 0) flag[1] = 1;
}

// rev[slot of j in cand(a)] = absolute slot of a in cand(j), for owned a (static between list builds)
// rev16: the first 16 reverse slots of every owned atom at a fixed stride (-1 beyond the row), so that the
// gather reaches a slot record in two dependent loads instead of three (no row offset to fetch first)
// A neighbour j that is a periodic image of an owned atom o (nlocal <= j < self_end; resident runs know owner and
// shift) is no centre of its own: the force its cluster would put on a is the force o's cluster puts on the image
// a' = a - shift_j (translation invariance), which o's centre stores under a' anyway.  The reverse slot of (a, j) is
// therefore o's slot of a', found by tag and position.  A pair for which it is not found sits at the outer edge of the
// list skin on one side only (distances rounded differently) and cannot become active before the next list build.
__global__ __launch_bounds__(256) void rev_kernel(const int nlocal, const int *__restrict__ cand_off,
                                                  const int *__restrict__ cand, int *__restrict__ rev,
                                                  int *__restrict__ rev16, int *__restrict__ flags,
                                                  const int self_end, const double4 *__restrict__ xq,
                                                  const int *__restrict__ tag, const int *__restrict__ ghost_owner,
                                                  const RebomosDev P)
{
  const int s = threadIdx.x % RP_L;
  const long long a64 = (long long) blockIdx.x * (256 / RP_L) + threadIdx.x / RP_L;
  if (a64 >= nlocal) return;
  const int a = (int) a64;
  const int off = cand_off[a], nc = cand_off[a + 1] - off;
  if (nc > 64 && s == 0) atomicOr(&flags[1], 1); // the active mask holds 64 candidates
  for (int t = s; t < nc; t += RP_L) {
    const int j = cand[off + t];
    int r = -1;
    if (j >= nlocal && j < self_end) { // a periodic image: its owner's slot of MY image
      const int o = ghost_owner[j - nlocal];
      // the image's shift as it is NOW (x_j - x_o): the box may have changed since the images were derived (fix npt /
      // deform refresh them at owner + count * h of the step), a shift stored then would miss a' by count * dh
      const double4 xa = xq[a], xj = xq[j], xo = xq[o];
      const double px = xa.x - (xj.x - xo.x), py = xa.y - (xj.y - xo.y), pz = xa.z - (xj.z - xo.z);
      const int ta = tag[a];
      const int oo = cand_off[o], no = cand_off[o + 1] - oo;
      for (int u = 0; u < no; u++) {
        const int k = cand[oo + u];
        if (tag[k] != ta) continue;
        const double4 xk = xq[k];
        const double ex = xk.x - px, ey = xk.y - py, ez = xk.z - pz;
        if (ex * ex + ey * ey + ez * ez < 1.0e-6) {
          r = oo + u;
          break;
        }
      }
      if (r < 0) { // only a pair at the outer edge of the list skin may lack its mirror (see above): anything that can
                   // come inside rcmax before the next list build (half the inner skin of motion on either side) must not
        const double dx = xa.x - xj.x, dy = xa.y - xj.y, dz = xa.z - xj.z;
        const int pt = 2 * (int) xa.w + (int) xj.w;
        const double rskin = sqrt(P.cand_cutsq[pt]) - sqrt(P.rcmaxsq[pt]);
        const double rlim = sqrt(P.rcmaxsq[pt]) + 0.5 * rskin;
        if (dx * dx + dy * dy + dz * dz < rlim * rlim) atomicOr(&flags[2], 1);
      }
    } else {
      const int oj = cand_off[j], nj = cand_off[j + 1] - oj;
      for (int u = 0; u < nj; u++)
        if (cand[oj + u] == a) {
          r = oj + u;
          break;
        }
    }
    rev[off + t] = r;
    if (t < 16) rev16[(size_t) a * 16 + t] = r;
  }
  static_assert(RP_L == 16, "one lane per rev16 entry");
  if (s >= nc) rev16[(size_t) a * 16 + s] = -1;
}

// current REBO coordination -> lane-group class, appended to the class lists (one atomic per wave
// and class: 4.8 M single-lane atomics on four counters took longer than the list build itself)
__global__ __launch_bounds__(256) void classify_kernel(const RebomosDev P, const int nall, const int nlocal,
                                                       const double4 *__restrict__ xq,
                                                       const int *__restrict__ cand_off, const int *__restrict__ cand,
                                                       const int *__restrict__ is_centre, int *__restrict__ class_list,
                                                       int *__restrict__ class_count, const int remote_start)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int k = -1;
  if (i < nall && (i < nlocal || is_centre[i]) && cand_off[i + 1] > cand_off[i]) {
    const double4 xi = xq[i];
    const int ti = (int) xi.w;
    int n = 0;
    bool bnd = i >= remote_start; // the centre or one of its candidates is a remote ghost: needs this step's halo
    for (int q = cand_off[i]; q < cand_off[i + 1]; q++) {
      const int j = cand[q];
      bnd = bnd || j >= remote_start;
      const double4 xj = xq[j];
      const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
      n += (dx * dx + dy * dy + dz * dz) < P.rcmaxsq[ti * 2 + (int) xj.w];
    }
    // smallest lane group that holds the current coordination (one lane per neighbour, some to spare); class 0 is
    // the lane-per-centre kernel with room for exactly three neighbours (a centre that finds a fourth before the next
    // list build -- S-S pairs of MoS2 sit 0.13 A outside rcmax -- is passed to the 8-lane-group kernel in the same step)
    k = (n <= 3) ? 0 : (n <= 7) ? 1 : (n <= 12) ? 2 : (n <= 14) ? 3 : 4; // 1 / 8 / 12 / 16 / 32 lanes
    k = 2 * k + (ti != 0); // classes are per (lane-group size, element): the element is then uniform per launch
    if (bnd) k += MDP_NCLASS_HALF;
  }
  // one atomic per class and BLOCK (a wave's share found by a prefix over the block's waves in LDS): the class counters
  // are twenty addresses, and an atomic per wave on the two or three busy ones was 75 000 atomics per address at
  // 4.6 M atoms, served one after the other
  __shared__ int s_cnt[4][MDP_NCLASS], s_base[MDP_NCLASS];
  const int wave = threadIdx.x >> 6;
  unsigned long long mine = 0ull;
#pragma unroll
  for (int kk = 0; kk < MDP_NCLASS; kk++) {
    const unsigned long long m = __ballot(k == kk);
    if (lane == 0) s_cnt[wave][kk] = __popcll(m);
    if (k == kk) mine = m;
  }
  __syncthreads();
  if (threadIdx.x < MDP_NCLASS) {
    const int kk = threadIdx.x;
    const int total = s_cnt[0][kk] + s_cnt[1][kk] + s_cnt[2][kk] + s_cnt[3][kk];
    s_base[kk] = total ? atomicAdd(&class_count[kk], total) : 0;
  }
  __syncthreads();
  if (k >= 0) {
    int base = s_base[k];
    for (int w = 0; w < wave; w++) base += s_cnt[w][k];
    class_list[(size_t) k * nall + base + __popcll(mine & ((1ull << lane) - 1ull))] = i;
  }
}

// the first W candidates of every centre of one class, contiguous in class order (-1 beyond the row)
__global__ __launch_bounds__(256) void pack_cand_kernel(const int n, const int W, const int *__restrict__ list,
                                                        const int *__restrict__ cand_off,
                                                        const int *__restrict__ cand, int *__restrict__ pk)
{
  const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long) n * W) return;
  const int gid = (int) (idx / W), t = (int) (idx % W);
  const int c = list[gid];
  const int off = cand_off[c], nc = cand_off[c + 1] - off;
  pk[idx] = t < nc ? cand[off + t] : -1;
}

__global__ void zero_small_kernel(double *acc, int n)
{
  if (threadIdx.x < n) acc[threadIdx.x] = 0.0;
}

} // namespace

// smallest double q with sqrt(q) >= c / largest q with sqrt(q) <= c: lets the kernel branch on rsq
// exactly as the reference branches on rij = sqrt(rsq)
static double rsq_first_ge(double c)
{
  double q = c * c;
  while (sqrt(q) >= c) q = nextafter(q, 0.0);
  while (sqrt(q) < c) q = nextafter(q, 1.0e300);
  return q;
}
static double rsq_last_le(double c)
{
  double q = c * c;
  while (sqrt(q) <= c) q = nextafter(q, 1.0e300);
  while (sqrt(q) > c) q = nextafter(q, 0.0);
  return q;
}

static double powint_h(double x, int n)
{
  double yy = 1.0, ww = x;
  for (int nn = n; nn != 0; nn >>= 1, ww *= ww)
    if (nn & 1) yy *= ww;
  return yy;
}

void mdp_rebomos_fill_dev(mdp_ctx *c, double skin)
{
  const mdp_rebomos_params &p = c->rebomos_host;
  RebomosDev &d = c->rebomos;
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) {
      const int k = a * 2 + b;
      d.rcmin[k] = p.rcmin[a][b];
      d.rcmax[k] = p.rcmax[a][b];
      d.rcmaxsq[k] = p.rcmaxsq[a][b];
      d.rcinv[k] = 1.0 / (p.rcmax[a][b] - p.rcmin[a][b]);
      d.Q[k] = p.Q[a][b];
      d.alpha[k] = p.alpha[a][b];
      d.A[k] = p.A[a][b];
      d.B[k] = p.BIJc[a][b];
      d.beta[k] = p.Beta[a][b];
      d.lj_rsq_lo[k] = rsq_first_ge(p.rcLJmin[a][b]);
      d.lj_rsq_hi[k] = rsq_last_le(p.rcLJmax[a][b]);
      d.lj_rsq_sw[k] = rsq_first_ge(0.95 * p.sigma[a][b]);
      d.lj1[k] = p.lj1[a][b];
      d.lj2[k] = p.lj2[a][b];
      d.lj3[k] = p.lj3[a][b];
      d.lj4[k] = p.lj4[a][b];
      d.rcLJmin[k] = p.rcLJmin[a][b];
      { // pair_rebomos.cpp:533-538
        const double sg = p.sigma[a][b], ep = p.epsilon[a][b];
        const double dr = 0.95 * sg - p.rcLJmin[a][b];
        const double r6 = powint_h((sg / (0.95 * sg)), 6);
        const double vdw = 4 * ep * r6 * (r6 - 1.0);
        const double dvdw = (-4 * ep / (0.95 * sg)) * r6 * (12.0 * r6 - 6.0);
        const double c2 = ((3.0 / dr) * vdw - dvdw) / dr;
        const double c3 = (vdw / (dr * dr) - c2) / dr;
        d.ljc2[k] = c2;
        d.ljc3[k] = c3;
      }
      const double cc = p.rcmax[a][b] + skin, cl = p.rcLJmax[a][b] + skin;
      d.cand_cutsq[k] = cc * cc;
      d.ljlist_cutsq[k] = cl * cl;
    }
  for (int t = 0; t < 2; t++) {
    for (int k = 0; k < 7; k++) {
      d.b[t][k] = p.b[k][t];
      d.bg[t][k] = p.bg[k][t];
    }
    for (int k = 0; k < 4; k++) d.a[t][k] = p.a[k][t];
  }
  // the cubic inner spline's parameters per pair type in device memory (rebo_lj_cubic_kernel reads them)
  double tab[48];
  for (int k = 0; k < 4; k++) {
    const double row[12] = {d.lj_rsq_lo[k], d.lj_rsq_hi[k], d.lj_rsq_sw[k], d.lj1[k], d.lj2[k], d.lj3[k], d.lj4[k],
                            d.rcLJmin[k], d.ljc2[k], d.ljc3[k], 0.0, 0.0};
    for (int q = 0; q < 12; q++) tab[12 * k + q] = row[q];
  }
  if (c->lj_fixtab.reserve(48) == hipSuccess) (void) hipMemcpy(c->lj_fixtab.p, tab, sizeof tab, hipMemcpyHostToDevice);
}

// ------------------------------------------------------------------------------------------------
// The style's own neighbor structures, built on the device from the positions at hand:
//   * REBO candidate lists (r <= rcmax + s_in) for owned atoms and the ghosts next to them,
//     reverse-slot table, lane-group classes
//   * Lennard-Jones cluster pair lists (r <= rcLJmax + s_in)
// s_in ("inner skin") <= the host's skin: the lists stay valid until some atom has moved s_in/2, which
// the device checks itself every compute (moved_kernel); the host's list only defines the ghost shell.
// MDP_DIAG=1: what the list builder saw when a candidate row outgrew the 64-bit active mask (small systems only:
// everything is copied to the host).  Answers "which input was wrong" for a spurious overflow.
static void repack_diag(mdp_ctx *c)
{
  const int nall = c->nall, nlocal = c->nlocal;
  if (nall <= 0 || nall > (1 << 22)) return;
  std::vector<double> x(4 * (size_t) nall);
  std::vector<int> off(nall + 2), cs;
  (void) hipMemcpy(x.data(), c->xq.p, sizeof(double) * 4 * nall, hipMemcpyDeviceToHost);
  (void) hipMemcpy(off.data(), c->cand_off.p, sizeof(int) * (nall + 1), hipMemcpyDeviceToHost);
  const long long ncell = (long long) c->grid.n[0] * c->grid.n[1] * c->grid.n[2];
  cs.resize((size_t) ncell + 2);
  (void) hipMemcpy(cs.data(), c->cell_start.p, sizeof(int) * (ncell + 1), hipMemcpyDeviceToHost);
  int worst = 0, wi = -1, nbad = 0, nzero = 0, nout = 0, nonmono = 0;
  for (int i = 0; i < nall; i++) {
    const int n = off[i + 1] - off[i];
    if (n > worst) worst = n, wi = i;
    const double *p = &x[4 * (size_t) i];
    if (!(p[0] == p[0]) || !(p[1] == p[1]) || !(p[2] == p[2])) nbad++;
    if (p[0] == 0.0 && p[1] == 0.0 && p[2] == 0.0) nzero++;
    for (int d = 0; d < 3; d++)
      if (p[d] < c->bbox_lo[d] || p[d] > c->bbox_hi[d]) {
        nout++;
        break;
      }
  }
  for (long long k = 0; k < ncell; k++)
    if (cs[k + 1] < cs[k]) nonmono++;
  fprintf(stderr,
          "[mdp diag] ctx %p rank %d: nlocal %d nall %d (self %d, remote from %d) grid %dx%dx%d cells; longest row %d at atom %d "
          "(%s) x=(%.3f %.3f %.3f) elem %.0f; NaN positions %d, all-zero positions %d, outside the bin box %d; "
          "cell_start[ncell]=%d non-monotonic cells %d; cand_total %d\n",
          (void *) c, c->dd.on ? c->dd.G.rank : -1, nlocal, nall, c->remote_start - nlocal, c->remote_start, c->grid.n[0],
          c->grid.n[1], c->grid.n[2], worst, wi, wi < nlocal ? "owned" : (wi < c->remote_start ? "self image" : "remote ghost"),
          wi >= 0 ? x[4 * (size_t) wi] : 0.0, wi >= 0 ? x[4 * (size_t) wi + 1] : 0.0, wi >= 0 ? x[4 * (size_t) wi + 2] : 0.0,
          wi >= 0 ? x[4 * (size_t) wi + 3] : 0.0, nbad, nzero, nout, cs[(size_t) ncell], nonmono, off[nall]);
  if (wi >= 0) { // how many atoms really sit within 5 A of the worst one, and how many of them are coincident
    int near = 0, same = 0;
    for (int j = 0; j < nall; j++) {
      if (j == wi) continue;
      const double dx = x[4 * (size_t) j] - x[4 * (size_t) wi], dy = x[4 * (size_t) j + 1] - x[4 * (size_t) wi + 1],
                   dz = x[4 * (size_t) j + 2] - x[4 * (size_t) wi + 2];
      const double r2 = dx * dx + dy * dy + dz * dz;
      near += r2 < 25.0;
      same += r2 < 1e-6;
    }
    fprintf(stderr, "[mdp diag]   atoms within 5 A of it: %d, coincident with it: %d\n", near, same);
  }
}

int mdp_rebomos_repack(mdp_ctx *c)
{
  if (!c->have_rebomos) return mdp_fail(c, MDP_ESTATE, "rebomos parameters not set");
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  c->host_check_armed = false;
  double s_in = c->skin > 0.0 ? c->skin : 2.0;
  {
    // Inner skin (A): starts at 1.0 and ADAPTS -- when the displacement trigger fires again within 200 computes
    // (thermal vibration reaching half the skin: every ~60 steps at 300 K), the next lists get 0.2 A more, up to
    // the host's skin.  Rows grow ~5 % per step of 0.2 A; a rebuild costs ~10 steps' worth of compute.
    // MDP_INNER_SKIN fixes the value instead.
    double want = c->skin_inner_auto;
    if (const char *e = getenv("MDP_INNER_SKIN")) {
      want = atof(e);
    } else if (c->stale_rebuild && c->computes_since_build < 200 && want + 0.2 < s_in + 1e-9 &&
               want + 0.2 < c->skin_inner_cap - 1e-9) { // (never back to a skin whose candidate rows overflowed)
      want += 0.2;
      c->skin_inner_auto = want;
    }
    c->stale_rebuild = false;
    c->computes_since_build = 0;
    if (want > 0.0 && want < s_in) s_in = want;
  }
  c->skin_inner = s_in;
  mdp_rebomos_fill_dev(c, s_in);
  const int nall = c->nall, nlocal = c->nlocal;
  hipStream_t st = c->stream;
  MDP_HIP(c, c->cand_cnt.reserve(nall + 1));
  MDP_HIP(c, c->cand_off.reserve(nall + 2));
  // candidates and rows from the host's neighbor list instead of the bin grid (mdp_rebomos_host_list): see
  // cand_csr_kernel.  One atom per row then: the two atoms of a cluster share a row, and a pair the host excluded for
  // one of them must not come in through the other.
  const bool from_host = c->rebo_host_list;
  if (from_host && (!c->neigh_set || !c->nb_off.p || c->md))
    return mdp_fail(c, MDP_ESTATE, "rebomos: lists from the host's neighbor list were asked for (mdp_rebomos_host_list) but no list was handed over (mdp_set_neighbors_host)");
  const int cl = from_host ? 1 : MDP_CLUSTER;
  // Lennard-Jones list layout: tiles of 16 two-atom rows; MDP_LJ_TILE=0: per-cluster lists of global indices (the
  // fallback when a union outgrows LDS).  (Tiles of 32 one-atom rows evaluate 25 % fewer pairs and measured slower,
  // 1.35 against 1.28 ms at 3.98 M atoms -- every LDS read and row index then serves one atom: DESIGN.md section 4.)
  bool want16 = cl == 2 || from_host;
  if (const char *e = getenv("MDP_LJ_TILE")) want16 = want16 && (atoi(e) != 0 || from_host);
  c->cluster = cl;
  const int nclus = (nlocal + cl - 1) / cl;
  c->nclus = nclus;
  const int nrow_max = ((nclus + MDP_TILE - 1) / MDP_TILE) * MDP_TILE + MDP_TILE; // whole tiles of rows
  MDP_HIP(c, c->lj_split.reserve(nrow_max + 1));
  MDP_HIP(c, c->lj_cnt.reserve(nrow_max + 1));
  MDP_HIP(c, c->lj_off.reserve(nrow_max + 2));
  MDP_HIP(c, c->is_center.reserve(nall + 1));
  MDP_HIP(c, c->amask.reserve(nall + 1));
  // overflow lists: [0] the general kernel's, [1..4] the lane-per-centre kernel's per (part, element); each {count, ids...}
  MDP_HIP(c, c->ovf.reserve((size_t) MDP_NOVF_LISTS * (nall + 2))); // (mdp_common.h: five lists of centres, two of tiles)
  c->ovf_stride = nall + 2;
  c->acc_prezeroed = false; // (the counters sit at new places: the next mdp_acc_begin zeroes them itself)
  // multi-GPU runs hide the halo exchange behind the REBO centres that reach no remote ghost.  (Round 1 split the
  // Lennard-Jones tiles instead, which needs the slot gather as a kernel of its own: 0.495 against 0.462 ms per
  // step of an 8-GPU sub-domain)
  const bool centre_split = c->md && c->remote_start < nall;
  c->centre_split = centre_split;
  MDP_HIP(c, c->class_list.reserve((size_t) (centre_split ? MDP_NCLASS : MDP_NCLASS_HALF) * nall + 8));
  MDP_HIP(c, c->class_count.reserve(MDP_NCLASS));
  MDP_HIP(c, c->xhold_all.reserve((size_t) 3 * nall + 3));
  MDP_HIP(c, hipMemsetAsync(c->is_center.p, 0, sizeof(int) * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->cand_cnt.p, 0, sizeof(int) * (nall + 1), st));
  MDP_HIP(c, hipMemsetAsync(c->amask.p, 0, sizeof(unsigned long long) * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->flags.p, 0, sizeof(int) * 4, st));
  MDP_HIP(c, hipMemsetAsync(c->class_count.p, 0, sizeof(int) * MDP_NCLASS, st));
  // one bin grid serves both lists (cell width >= (rcLJmax + s_in)/2)
  double ljcut = 0.0, candcut = 0.0;
  for (int k = 0; k < 4; k++) {
    ljcut = ljcut > c->rebomos.ljlist_cutsq[k] ? ljcut : c->rebomos.ljlist_cutsq[k];
    candcut = candcut > c->rebomos.cand_cutsq[k] ? candcut : c->rebomos.cand_cutsq[k];
  }
  ljcut = sqrt(ljcut);
  candcut = sqrt(candcut);
  MDP_TRY(mdp_bin_atoms(c, ljcut, c->bbox_lo, c->bbox_hi));
  const int Rc = candcut <= 0.5 * ljcut ? 1 : 2; // cells are >= ljcut/2 wide
  // Resident runs know which ghosts are periodic images of owned atoms (owner, shift): those are no centres of their
  // own, the gather of their owned neighbours reads the owner centre's slots instead (rev_kernel).
  // Host mode knows the same once the library keeps the images itself (mdp_set_box_host): every ghost is one.
  const bool host_images = !c->md && c->host_ghosts_derived && c->ghost_owner.p && c->host_tag_dev.p;
  const int self_end = host_images ? nall
                       : (c->md && c->ghost_owner.p && c->tag.p && c->remote_start > nlocal)
                           ? (c->remote_start < nall ? c->remote_start : nall) : nlocal;
  const int *tag_dev = host_images ? c->host_tag_dev.p : c->tag.p; // tags in the device's atom order
  const int per_block = 256 / RP_L;
  const int nghost = nall - nlocal;
  static const bool two_pass_env = getenv("MDP_CAND_TWO_PASS") && atoi(getenv("MDP_CAND_TWO_PASS")) != 0; // (A/B, tests)
  const bool two_pass = two_pass_env && !from_host;
  int *stage = nullptr;
  if (!two_pass) {
    MDP_HIP(c, c->cand_stage.reserve((size_t) nall * kCandStride + kCandStride));
    stage = c->cand_stage.p;
  }
  if (from_host) {
    if (nlocal)
      cand_csr_kernel<0><<<(nlocal + per_block - 1) / per_block, 256, 0, st>>>(
          c->rebomos, nall, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->cand_cnt.p, stage, c->is_center.p, self_end);
    if (nghost)
      cand_csr_kernel<1><<<(nghost + per_block - 1) / per_block, 256, 0, st>>>(
          c->rebomos, nall, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->cand_cnt.p, stage, c->is_center.p, self_end);
  } else {
    if (nlocal)
      cand_build_kernel<0><<<(nlocal + per_block - 1) / per_block, 256, 0, st>>>(
          c->grid, Rc, c->rebomos, nall, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p, c->cand_cnt.p, nullptr,
          stage, c->is_center.p, self_end, c->xq_cell.p);
    if (nghost)
      cand_build_kernel<1><<<(nghost + per_block - 1) / per_block, 256, 0, st>>>(
          c->grid, Rc, c->rebomos, nall, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p, c->cand_cnt.p, nullptr,
          stage, c->is_center.p, self_end, c->xq_cell.p);
  }
  MDP_HIP(c, hipGetLastError());
  MDP_TRY(mdp_scan_exclusive_int(c, c->cand_cnt.p, c->cand_off.p, nall));
  // Lennard-Jones lists: tile lists for the default cluster size, unless switched off or a union outgrows LDS
  bool tiled = want16 && nclus > 0;
  const int ntile = (nclus + MDP_TILE - 1) / MDP_TILE;
  if (tiled) {
    int cap = c->tile_cap > 0 ? c->tile_cap : 2048;
    MDP_HIP(c, c->tile_flag.reserve(4));
    MDP_HIP(c, c->tile_nu.reserve((size_t) 2 * ntile + 2));
    MDP_HIP(c, c->lj_fix_stamp.reserve((size_t) ntile + 1));
    MDP_HIP(c, hipMemsetAsync(c->lj_fix_stamp.p, 0, sizeof(int) * ((size_t) ntile + 1), st)); // (no compute has stamp 0)
    for (;;) {
      MDP_HIP(c, c->tu.reserve((size_t) ntile * cap));
      MDP_HIP(c, c->tmask.reserve((size_t) ntile * cap));
      MDP_HIP(c, hipMemsetAsync(c->tile_flag.p, 0, sizeof(int) * 3, st));
      const size_t lds = (size_t) 14 * cap; // 4 wave segments of cap/2 (int + ushort) + cap ushort
#define MDP_TS(CLV)                                                                                                   \
  do {                                                                                                                \
    if (lds > 48 * 1024)                                                                                              \
      MDP_HIP(c, hipFuncSetAttribute((const void *) tile_scan_kernel<CLV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int) lds));                                                                     \
    tile_scan_kernel<CLV><<<ntile, 256, lds, st>>>(c->grid, c->rebomos, nclus, nlocal, c->xq.p, c->cell_perm.p,         \
                                                   c->cell_start.p, cap, c->tu.p, c->tmask.p, c->tile_nu.p,           \
                                                   c->lj_cnt.p, c->lj_split.p, c->tile_flag.p, c->xq_cell.p);         \
  } while (0)
      if (from_host) {
        const size_t ldsh = (size_t) 18 * cap; // hash table of 2 cap (key, value) slots + cap masks
#define MDP_TSH(CLV)                                                                                                  \
  do {                                                                                                                \
    if (ldsh > 48 * 1024)                                                                                             \
      MDP_HIP(c, hipFuncSetAttribute((const void *) tile_scan_csr_kernel<CLV>,                                        \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsh));                        \
    tile_scan_csr_kernel<CLV><<<ntile, 256, ldsh, st>>>(c->rebomos, nclus, nlocal, c->xq.p, c->nb_off.p, c->nb.p, cap,   \
                                                        c->tu.p, c->tmask.p, c->tile_nu.p, c->lj_cnt.p, c->lj_split.p, \
                                                        c->tile_flag.p);                                              \
  } while (0)
        if (cl == 1) MDP_TSH(1);
        else MDP_TSH(2);
#undef MDP_TSH
      } else if (cl == 1) MDP_TS(1);
      else MDP_TS(2);
#undef MDP_TS
      MDP_HIP(c, hipGetLastError());
      int tf[3] = {0, 0, 0};
      MDP_TRY(mdp_read_one(c, c->tile_flag.p, sizeof(int) * 3, tf));
      if (!tf[0]) {
        c->tile_cap = cap;
        c->tile_maxu = tf[1];
        c->tile_rowmax = tf[2];
        break;
      }
      cap *= 2; // a union outgrew the segment: retry larger, give up beyond what LDS can stage
      if (cap > 4096) {
        tiled = false;
        break;
      }
    }
  }
  c->lj_tiled = tiled;
  c->ntile = tiled ? ntile : 0;
  if (tiled) c->tile_rows_cl = cl;
  if (tiled && getenv("MDP_DEBUG")) {
    std::vector<int> h(2 * (size_t) ntile);
    MDP_HIP(c, hipMemcpy(h.data(), c->tile_nu.p, sizeof(int) * 2 * ntile, hipMemcpyDeviceToHost));
    double sum = 0;
    for (int k = 0; k < ntile; k++) sum += h[2 * k];
    fprintf(stderr, "[mdp] tile lists: %d tiles, cap %d, union mean %.1f max %d\n", ntile, c->tile_cap, sum / ntile,
            c->tile_maxu);

    int hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < ntile; k++) hist[h[2 * k] / 200 < 8 ? h[2 * k] / 200 : 8]++;
    for (int k = 0; k < 9; k++) fprintf(stderr, "[mdp]   union %4d.. : %d tiles\n", 200 * k, hist[k]);
  }
  if (nclus && !tiled && from_host)
    return mdp_fail(c, MDP_ENOTIMPL, "rebomos: lists from the host's neighbor list need the tile lists (two-atom clusters, unions within LDS)");
  if (nclus && !tiled) {
    const int gb = (nclus + 15) / 16;
#define MDP_CB(CLV, FILLV, OFFP, OUTP)                                                                              \
  cluster_build_kernel<CLV, FILLV><<<gb, 256, 0, st>>>(c->grid, c->rebomos, nclus, nlocal, c->xq.p, c->cell_perm.p, \
                                                       c->cell_start.p, c->lj_cnt.p, c->lj_split.p, OFFP, OUTP)
    if (cl == 1) MDP_CB(1, false, nullptr, nullptr);
    else if (cl == 2) MDP_CB(2, false, nullptr, nullptr);
    else MDP_CB(4, false, nullptr, nullptr);
  }
  MDP_HIP(c, hipGetLastError());
  const int nrow = tiled ? ntile * MDP_TILE : nclus;
  MDP_TRY(mdp_scan_exclusive_i64(c, c->lj_cnt.p, c->lj_off.p, nrow));
  int cand_total = 0;
  long long lj_total = 0;
  {
    const MdpRead rd[2] = {{c->cand_off.p + nall, sizeof(int), &cand_total}, {c->lj_off.p + nrow, sizeof(long long), &lj_total}};
    MDP_TRY(mdp_read_small(c, rd, 2));
  }
  c->cand_total = cand_total;
  c->lj_total = lj_total;
  if (tiled && getenv("MDP_DEBUG"))
    fprintf(stderr, "[mdp] row entries incl. padding: %lld (%.1f per cluster)\n", lj_total, (double) lj_total / nclus);
  MDP_HIP(c, c->cand.reserve((size_t) cand_total + 1));
  if (tiled)
    MDP_HIP(c, c->lj16.reserve((size_t) lj_total + 256)); // slack: rows are read a few steps past their end
  else
    MDP_HIP(c, c->lj.reserve((size_t) lj_total + 1));
  MDP_HIP(c, c->rev.reserve((size_t) cand_total + 1));
  MDP_HIP(c, c->rev16.reserve((size_t) 16 * nlocal + 16));
  MDP_HIP(c, c->fnbr.reserve((size_t) 4 * cand_total + 4));
  MDP_HIP(c, c->fown.reserve((size_t) 4 * nall + 4));
  MDP_HIP(c, hipMemsetAsync(c->fown.p, 0, sizeof(double) * 4 * nall, st)); // atoms that are no centre (NULL type) keep 0
  if (nall && stage)
    cand_compact_kernel<<<(nall + per_block - 1) / per_block, 256, 0, st>>>(nall, c->cand_off.p, stage, c->cand.p);
  else if (nall)
    cand_build_kernel<2><<<(nall + per_block - 1) / per_block, 256, 0, st>>>(
        c->grid, Rc, c->rebomos, nall, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p, nullptr, c->cand_off.p,
        c->cand.p, c->is_center.p, self_end, c->xq_cell.p);
  bool rows_filled = false;
  if (tiled) MDP_TRY(tile_sort_launch(c, ntile, nclus, /*fill=*/true, &rows_filled));
  if (tiled && !rows_filled) // (MDP_TILE_SORT=0: the unions stay as found)
    tile_fill_kernel<<<ntile, 256, 0, st>>>(nclus, c->tile_cap, c->tile_nu.p, c->tmask.p, c->lj_off.p, c->lj_split.p,
                                            c->lj16.p);
  c->prune_valid = false; // (new rows: the pruned copy is made at the next compute that wants it)
  c->prune_stale = false;
  c->prune_epoch++;
  if (nclus && !tiled) {
    const int gb = (nclus + 15) / 16;
    if (cl == 1) MDP_CB(1, true, c->lj_off.p, c->lj.p);
    else if (cl == 2) MDP_CB(2, true, c->lj_off.p, c->lj.p);
    else MDP_CB(4, true, c->lj_off.p, c->lj.p);
  }
#undef MDP_CB
  MDP_HIP(c, hipGetLastError());
  // launch classes of the units (tiles, or clusters without tile lists): small/large union for the LDS allocation
  // of the tile kernel
  const int nunit = tiled ? ntile : nclus;
  c->lj_ordered = false;
  for (int q = 0; q <= 4; q++) c->lj_class_base[q] = q ? nunit : 0; // everything in class 0
  // the tile kernel is latency-bound (measured: t = 0.65 ms + 3.96 ms / resident workgroups per CU), so the
  // common case is sized for FIVE workgroups per CU; the rare larger unions get their own launch
  int kSmallUnion = 1210; // (1210 + 1) * 24 B + 1.5 KB of static LDS = 30.6 KB: five workgroups and their allocation granules fit 160 KB
  if (const char *e = getenv("MDP_TILE_SMALL")) kSmallUnion = atoi(e) > 0 ? atoi(e) : kSmallUnion; // (tests)
  c->tile_small = tiled ? (c->tile_maxu < kSmallUnion ? c->tile_maxu : kSmallUnion) : 0;
  if (nunit > 0 && tiled && c->tile_maxu > kSmallUnion) {
    MDP_HIP(c, c->cl_flag.reserve((size_t) 6 * (nunit + 1)));
    MDP_HIP(c, c->cl_pos.reserve((size_t) 4 * (nunit + 2)));
    MDP_HIP(c, c->cl_order.reserve(nunit + 1));
    int *flag4 = c->cl_flag.p;
    unit_class_kernel<<<(nunit + 255) / 256, 256, 0, st>>>(nunit, nullptr,
                                                           tiled ? c->tile_nu.p : nullptr, kSmallUnion, flag4);
    MDP_HIP(c, hipGetLastError());
    int total[4] = {0, 0, 0, 0};
    MdpRead rd[4];
    for (int q = 0; q < 4; q++) {
      int *pos = c->cl_pos.p + (size_t) q * (nunit + 2);
      MDP_TRY(mdp_scan_exclusive_int(c, flag4 + (size_t) q * (nunit + 1), pos, nunit));
      rd[q] = {pos + nunit, sizeof(int), &total[q]};
    }
    MDP_TRY(mdp_read_small(c, rd, 4));
    for (int q = 0; q < 4; q++) c->lj_class_base[q + 1] = c->lj_class_base[q] + total[q];
    unit_order_kernel<<<(nunit + 255) / 256, 256, 0, st>>>(nunit, flag4, c->cl_pos.p, c->lj_class_base[1],
                                                           c->lj_class_base[2], c->lj_class_base[3], c->cl_order.p);
    MDP_HIP(c, hipGetLastError());
    c->lj_ordered = true;
  }
  if (nall)
    classify_kernel<<<(nall + 255) / 256, 256, 0, st>>>(c->rebomos, nall, nlocal, c->xq.p, c->cand_off.p, c->cand.p,
                                                        c->is_center.p, c->class_list.p, c->class_count.p,
                                                        centre_split ? c->remote_start : 0x7fffffff);
  MDP_HIP(c, hipGetLastError());
  if (nlocal)
    rev_kernel<<<(nlocal + per_block - 1) / per_block, 256, 0, st>>>(nlocal, c->cand_off.p, c->cand.p, c->rev.p,
                                                                     c->rev16.p, c->flags.p, self_end, c->xq.p, tag_dev,
                                                                     c->ghost_owner.p, c->rebomos);
  if (nall) hold_all_kernel<<<(nall + 255) / 256, 256, 0, st>>>(nall, c->xq.p, c->xhold_all.p);
  MDP_HIP(c, hipGetLastError());
  int hflags[4] = {0, 0, 0, 0};
  {
    const MdpRead rd[2] = {{c->class_count.p, sizeof(int) * MDP_NCLASS, c->h_class_count}, {c->flags.p, sizeof(int) * 4, hflags}};
    MDP_TRY(mdp_read_small(c, rd, 2));
  }
  if (hflags[2])
    return mdp_fail(c, MDP_ESTATE, "rebomos: a periodic image within rcmax + half the inner skin of an owned atom has no mirror "
                                   "slot in its owner's row (images inconsistent with the owned atoms' positions)");
  if (hflags[1]) {
    if (getenv("MDP_DIAG")) repack_diag(c);
    // A candidate row holds the atoms within rcmax + inner skin and the active mask has 64 bits.  Rows that
    // overflow because of the SKIN (a dense system: the reference's REBO list itself, pair_rebomos.cpp:337-350,
    // would still be short) are rebuilt with half the skin, down to 0.2 A: then only 65 atoms inside rcmax
    // itself (3.8 A: six times the density of MoS2) stop the run, with the reference's words.
    if (c->skin_inner > 0.2 + 1e-9 && !getenv("MDP_INNER_SKIN")) {
      fprintf(stderr, "[mdp] rebomos: a candidate row (atoms within rcmax + inner skin) outgrew 64 entries at an inner skin "
                      "of %.2f A; the style's lists are rebuilt with %.2f A and keep that cap (mdp_md_list_state)\n",
              c->skin_inner, 0.5 * c->skin_inner > 0.2 ? 0.5 * c->skin_inner : 0.2);
      c->skin_inner_cap = c->skin_inner;
      c->skin_inner_auto = 0.5 * c->skin_inner > 0.2 ? 0.5 * c->skin_inner : 0.2;
      c->stale_rebuild = false;
      return mdp_rebomos_repack(c);
    }
    return mdp_fail(c, MDP_EOVERFLOW, "rebomos: an atom has more than 64 neighbours inside rcmax+skin (Neighbor list overflow)");
  }
  { // packed candidate heads of the lane-group classes (widths = UA*G of rebo_centre_kernel<G>), per element
    const int width[5] = {CentreCfg<4>::UA * 4, CentreCfg<8>::UA * 8, CentreCfg<12>::UA * 12, CentreCfg<16>::UA * 16,
                          CentreCfg<32>::UA * 32};
    // (the boundary half of a class directly behind its interior half: a launch over both -- the blocking orders of a
    //  multi-GPU step, which have no use for two launches per class -- then finds one contiguous block)
    size_t total = 0;
    for (int g = 0; g < MDP_NCLASS_HALF; g++)
      for (int half = 0; half < 2; half++) {
        const int k = g + half * MDP_NCLASS_HALF;
        c->pk_base[k] = total;
        total += (size_t) c->h_class_count[k] * width[g / 2];
      }
    MDP_HIP(c, c->pk_cand.reserve(total + 1));
    if (centre_split) { // ... and the two halves' centre lists once more as one list per class
      size_t mtot = 0;
      for (int g = 0; g < MDP_NCLASS_HALF; g++) {
        c->merged_base[g] = mtot;
        mtot += (size_t) c->h_class_count[g] + c->h_class_count[g + MDP_NCLASS_HALF];
      }
      MDP_HIP(c, c->class_merged.reserve(mtot + 1));
      for (int g = 0; g < MDP_NCLASS_HALF; g++) {
        const int n0 = c->h_class_count[g], n1 = c->h_class_count[g + MDP_NCLASS_HALF];
        if (n0)
          MDP_HIP(c, hipMemcpyAsync(c->class_merged.p + c->merged_base[g], c->class_list.p + (size_t) g * nall, sizeof(int) * n0,
                                    hipMemcpyDeviceToDevice, st));
        if (n1)
          MDP_HIP(c, hipMemcpyAsync(c->class_merged.p + c->merged_base[g] + n0,
                                    c->class_list.p + (size_t) (g + MDP_NCLASS_HALF) * nall, sizeof(int) * n1,
                                    hipMemcpyDeviceToDevice, st));
      }
    }
    for (int k = 0; k < MDP_NCLASS; k++) {
      const int w = width[(k % MDP_NCLASS_HALF) / 2];
      const long long n = (long long) c->h_class_count[k] * w;
      if (n > 0)
        pack_cand_kernel<<<(unsigned) ((n + 255) / 256), 256, 0, st>>>(c->h_class_count[k], w,
                                                                       c->class_list.p + (size_t) k * nall,
                                                                       c->cand_off.p, c->cand.p,
                                                                       c->pk_cand.p + c->pk_base[k]);
    }
    MDP_HIP(c, hipGetLastError());
  }
  c->rebo_packed = true;
  c->style_builds++;
  return MDP_OK;
}

// The same tile lists for another two-type style (AEAM, resident mode): `cutsq[ti*2+tj]` = squared list radius of
// the pair; needs the bin grid of the current positions (mdp_bin_atoms).  Fills tu / tile_nu / lj_off / lj_split /
// lj16 and c->nclus, ntile, tile_cap, tile_maxu; *ok = false when a union does not fit (caller keeps its CSR path).
int mdp_tile_lists_build(mdp_ctx *c, const double cutsq[4], int cl, bool *ok)
{
  *ok = false;
  const int nlocal = c->nlocal;
  hipStream_t st = c->stream;
  if (cl != 1) cl = 2;
  const int nclus = (nlocal + cl - 1) / cl;
  if (nclus <= 0) return MDP_OK;
  const int ntile = (nclus + MDP_TILE - 1) / MDP_TILE;
  const int nrow = ntile * MDP_TILE;
  RebomosDev P = {};
  for (int k = 0; k < 4; k++) P.ljlist_cutsq[k] = cutsq[k];
  MDP_HIP(c, c->lj_split.reserve(nrow + 1));
  MDP_HIP(c, c->lj_cnt.reserve(nrow + 1));
  MDP_HIP(c, c->lj_off.reserve(nrow + 2));
  MDP_HIP(c, c->tile_flag.reserve(4));
  MDP_HIP(c, c->tile_nu.reserve((size_t) 2 * ntile + 2));
  int cap = c->tile_cap > 0 ? c->tile_cap : 2048;
  for (;;) {
    MDP_HIP(c, c->tu.reserve((size_t) ntile * cap));
    MDP_HIP(c, c->tmask.reserve((size_t) ntile * cap));
    MDP_HIP(c, hipMemsetAsync(c->tile_flag.p, 0, sizeof(int) * 3, st));
    const size_t lds = (size_t) 14 * cap;
    if (cl == 1) {
      if (lds > 48 * 1024)
        MDP_HIP(c, hipFuncSetAttribute((const void *) tile_scan_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) lds));
      tile_scan_kernel<1><<<ntile, 256, lds, st>>>(c->grid, P, nclus, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p,
                                                   cap, c->tu.p, c->tmask.p, c->tile_nu.p, c->lj_cnt.p, c->lj_split.p,
                                                   c->tile_flag.p, c->xq_cell.p);
    } else {
      if (lds > 48 * 1024)
        MDP_HIP(c, hipFuncSetAttribute((const void *) tile_scan_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) lds));
      tile_scan_kernel<2><<<ntile, 256, lds, st>>>(c->grid, P, nclus, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p,
                                                   cap, c->tu.p, c->tmask.p, c->tile_nu.p, c->lj_cnt.p, c->lj_split.p,
                                                   c->tile_flag.p, c->xq_cell.p);
    }
    MDP_HIP(c, hipGetLastError());
    int tf[3] = {0, 0, 0};
    MDP_TRY(mdp_read_one(c, c->tile_flag.p, sizeof(int) * 3, tf));
    if (!tf[0]) {
      c->tile_cap = cap;
      c->tile_maxu = tf[1];
      c->tile_rowmax = tf[2];
      break;
    }
    cap *= 2;
    if (cap > 4096) return MDP_OK; // *ok stays false
  }
  MDP_TRY(mdp_scan_exclusive_i64(c, c->lj_cnt.p, c->lj_off.p, nrow));
  long long total = 0;
  MDP_TRY(mdp_read_one(c, c->lj_off.p + nrow, sizeof(long long), &total));
  MDP_HIP(c, c->lj16.reserve((size_t) total + 256));
  bool rows_filled = false;
  MDP_TRY(tile_sort_launch(c, ntile, nclus, /*fill=*/true, &rows_filled));
  if (!rows_filled)
    tile_fill_kernel<<<ntile, 256, 0, st>>>(nclus, c->tile_cap, c->tile_nu.p, c->tmask.p, c->lj_off.p, c->lj_split.p,
                                            c->lj16.p);
  MDP_HIP(c, hipGetLastError());
  c->nclus = nclus;
  c->ntile = ntile;
  c->lj_total = total;
  c->tile_rows_cl = cl;
  c->prune_valid = false; // (new rows)
  c->prune_stale = false;
  c->prune_epoch++;
  if (getenv("MDP_DEBUG"))
    fprintf(stderr, "[mdp] tile lists (generic): %d tiles, cap %d, largest union %d, %.1f row entries per cluster\n", ntile,
            c->tile_cap, c->tile_maxu, (double) total / nclus);
  *ok = true;
  return MDP_OK;
}

// `neigh_modify check yes`, done by the style for its own lists: has any atom (ghosts included) moved
// more than half the inner skin since they were built?
//   host mode     : checked before every compute (the host synchronises each step anyway)
//   resident mode : the check rides in the integrate kernel (owned atoms) and the halo unpack (remote ghosts) of
//                   step n -- csrc/md.hip, MdpStyleCheck -- and is read at step n+1 (pinned flags + event), so the
//                   CPU never waits for the GPU inside the MD loop and no kernel of its own reads the positions
//                   again; the trigger is lowered by kStaleMargin to cover the one step of extra motion, and a true
//                   violation is counted as a "dangerous build".  (kPruneMargin: the same for the pruned rows.)

static int rebomos_check_launch(mdp_ctx *c, const double trig)
{
  hipStream_t st = c->stream;
  const int nall = c->nall;
  const int grid = (nall + 255) / 256 < 2048 ? (nall + 255) / 256 : 2048;
  const double hard = 0.5 * c->skin_inner;
  int *h = (int *) (c->h_pinned + 24); // no check is in flight here: the caller has waited for the previous one
  h[0] = h[1] = h[2] = h[3] = 0;
  // (pruned rows: their own, smaller trigger against the positions of the last pruning)
  const bool pr = c->prune_valid;
  double ptrig = 0.5 * c->prune_buf - kPruneMargin * mdp_margin_scale(c);
  if (ptrig < 0.25 * c->prune_buf) ptrig = 0.25 * c->prune_buf;
  const double phard = 0.5 * c->prune_buf;
  moved_kernel<<<grid, 256, 0, st>>>(nall, trig * trig, hard * hard, c->xq.p, c->xhold_all.p, h,
                                     pr ? c->xhold_prune.p : nullptr, ptrig * ptrig, phard * phard);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// host mode: the displacement check of the compute that follows, launched where the stream is waited for anyway
// (mdp_set_positions_host) instead of with a wait of its own
int mdp_rebomos_host_precheck(mdp_ctx *c)
{
  c->host_check_armed = false;
  if (c->md || !c->have_rebomos || c->have_aeam || !c->rebo_packed || !c->nall) return MDP_OK;
  MDP_TRY(rebomos_check_launch(c, 0.5 * c->skin_inner));
  c->host_check_armed = true;
  return MDP_OK;
}

static int rebomos_lists_stale(mdp_ctx *c, bool &stale)
{
  stale = false;
  if (!c->nall) return MDP_OK;
  int *h = (int *) (c->h_pinned + 24);
  if (!c->md && c->host_check_armed) { // host mode: launched behind the upload, which has waited for it already
    c->host_check_armed = false;
    stale = h[0] != 0;
    if (c->prune_valid && h[2]) c->prune_stale = true;
    return MDP_OK;
  }
  const bool deferred_host = !c->md && c->hn_on && c->hn_deferred_check && !c->check_now; // (mdp_hnve_initial armed it)
  if ((!c->md && !deferred_host) || c->check_now) { // immediate (host mode; resident mode right after the host rewrote the positions)
    mdp_sflag_drop(c); // (words of a check armed before the positions were rewritten)
    c->check_now = false;
    MDP_TRY(rebomos_check_launch(c, 0.5 * c->skin_inner));
    MDP_HIP(c, hipStreamSynchronize(c->stream));
    stale = h[0] != 0;
    if (c->prune_valid && h[2]) c->prune_stale = true;
    return MDP_OK;
  }
  // deferred by one compute: the words the integrate kernel / halo unpack of the previous step wrote
  bool toofar = false;
  MDP_TRY(mdp_sflag_collect(c, &stale, &toofar));
  if (toofar) c->dangerous_builds++;
  return MDP_OK;
}

template <int G>
static void launch_centre(mdp_ctx *c, int kg, int eflag, int vflag, int part)
{
  for (int elem = 0; elem < 2; elem++) {
    // part 0: interior centres, 1: boundary centres, 2: both halves of the class in one launch (class_merged; their
    // packed candidates are contiguous: mdp_rebomos_repack)
    const int k = 2 * kg + elem + (part == 1 ? MDP_NCLASS_HALF : 0);
    const int n = c->h_class_count[k] + (part == 2 ? c->h_class_count[k + MDP_NCLASS_HALF] : 0);
    if (n <= 0) continue;
    const int *list = part == 2 ? c->class_merged.p + c->merged_base[k] : c->class_list.p + (size_t) k * c->nall;
    constexpr int per_block = CentreCfg<G>::WPB * CentreCfg<G>::GPW;
    const int grid = (n + per_block - 1) / per_block;
    rebo_centre_kernel<G><<<grid, 64 * CentreCfg<G>::WPB, 0, c->stream>>>(c->rebomos, list, n, c->nlocal,
                                                       c->xq.p, c->cand_off.p, c->cand.p, c->pk_cand.p + c->pk_base[k],
                                                       c->amask.p, c->fnbr.p, c->fown.p, c->acc.p, c->ovf.p, eflag, vflag,
                                                       elem);
  }
}

// centres with at most three neighbours (class 0 of either element): one lane per centre
static void launch_centre3(mdp_ctx *c, int eflag, int vflag, int part)
{
  for (int elem = 0; elem < 2; elem++) {
    const int k = elem + (part == 1 ? MDP_NCLASS_HALF : 0); // (part 2: both halves in one launch, as launch_centre)
    const int n = c->h_class_count[k] + (part == 2 ? c->h_class_count[k + MDP_NCLASS_HALF] : 0);
    if (n <= 0) continue;
    const int *clist = part == 2 ? c->class_merged.p + c->merged_base[k] : c->class_list.p + (size_t) k * c->nall;
    // Its centres with a fourth neighbour (S-S pairs of MoS2 dip below rcmax at 300 K) are counted per (part, elem);
    // the count of a step is published to a pinned word by the last centre kernel of the step and read here TWO computes
    // later, without a wait of its own: two sets of words alternate (ovf_par), and the set read now was written by the
    // compute before the last, which the deferred displacement check of this compute has waited for (mdp_sflag_collect)
    // -- so the choice of path below is the same in every run of the same trajectory.  While it is zero -- a cold crystal -- such a centre goes straight to the general
    // kernel's list; once centres do overflow they are collected on a list of their own ...
    const int q = (part == 1 ? 2 : 0) + elem; // (a launch over both halves collects on the interior half's list)
    int *list = c->ovf.p + (size_t) (q + 1) * c->ovf_stride + 1; // (count at list[-1]; zeroed with the accumulators)
    int *h_cnt = (int *) (c->h_pinned + 40) + 4 * c->ovf_par + q;
    if (*h_cnt > 0) c->ovf3_hot[q] = 64;      // (hysteresis: stay in list mode for 64 computes after the last overflow)
    else if (c->ovf3_hot[q] > 0) c->ovf3_hot[q]--;
    const bool list_mode = c->ovf3_hot[q] > 0;
    rebo_centre3_kernel<<<(n + kC3Block - 1) / kC3Block, kC3Block, 0, c->stream>>>(c->rebomos, clist, n,
                                                                c->nlocal, c->xq.p, c->cand_off.p, c->cand.p,
                                                                c->pk_cand.p + c->pk_base[k], c->amask.p, c->fnbr.p,
                                                                c->fown.p, c->acc.p, list, list_mode ? nullptr : c->ovf.p,
                                                                eflag, vflag, elem);
    if (!list_mode) continue;
    // ... and go through the 8-lane-group kernel at once; grid from the count seen two computes ago,
    // the kernel itself hands what it does not cover to the general kernel
    constexpr int per_block = CentreCfg<8>::WPB * CentreCfg<8>::GPW;
    // (the count is two computes old: 4 x + two blocks -- at the onset of overflows it grows by 1.8 x per step --; what a grid does not cover still reaches the general kernel,
    //  but WHICH entries those are depends on the order the list was filled in -- with a generous grid the paths, and
    //  with them the last bits of the forces, are the same in every run)
    long long est = 4 * (long long) *h_cnt + 2 * per_block;
    if (est > n) est = n;
    const int grid = (int) ((est + per_block - 1) / per_block);
    rebo_centre_kernel<8, true><<<grid, 64 * CentreCfg<8>::WPB, 0, c->stream>>>(
        c->rebomos, list, 0, c->nlocal, c->xq.p, c->cand_off.p, c->cand.p, h_cnt, c->amask.p, c->fnbr.p, c->fown.p,
        c->acc.p, c->ovf.p, eflag, vflag, elem);
  }
}

// (Re-)prune the tile rows from the current positions (tile_prune_kernel) and remember those positions.
// lim_rsq[ti * 2 + tj]: (largest pair distance the kernels act on + buffer)^2.
int mdp_tile_prune(mdp_ctx *c, const double lim_rsq[4])
{
  hipStream_t st = c->stream;
  const int nrow = c->ntile * MDP_TILE;
  MDP_HIP(c, c->lj_len_in.reserve(nrow + 1));
  MDP_HIP(c, c->lj_split_in.reserve(nrow + 1));
  MDP_HIP(c, c->xhold_prune.reserve((size_t) 3 * c->nall + 3));
  if (c->prune_copied_epoch != c->prune_epoch) { // rows were rebuilt: the slack behind the last row as well
    MDP_HIP(c, c->lj16_in.reserve((size_t) c->lj_total + 256)); // (reads past a segment's end must find valid indices)
    MDP_HIP(c, hipMemcpyAsync(c->lj16_in.p + c->lj_total, c->lj16.p + c->lj_total, sizeof(unsigned short) * 256,
                              hipMemcpyDeviceToDevice, st));
    c->prune_copied_epoch = c->prune_epoch;
  }
  PruneLimits lim;
  for (int k = 0; k < 4; k++) lim.rsq[k] = lim_rsq[k];
  const int capL = (c->tile_maxu + 1 + 1) & ~1; // (the rows behind it start 16-byte aligned)
  const size_t lds = (size_t) capL * 3 * sizeof(double) + (size_t) 2 * c->tile_rowmax + 32;
  if (c->tile_rowmax <= 0 || lds > 160 * 1024) { // (rows of a tile do not fit LDS next to its union: no pruning)
    c->prune_valid = false;
    return MDP_OK;
  }
#define MDP_TP(CLV)                                                                                                    \
  do {                                                                                                                 \
    if (lds > 48 * 1024)                                                                                               \
      MDP_HIP(c, hipFuncSetAttribute((const void *) tile_prune_kernel<CLV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int) lds));                                                                      \
    tile_prune_kernel<CLV><<<c->ntile, 256, lds, st>>>(lim, c->nlocal, c->nclus, c->xq.p, c->tile_cap, capL, c->tu.p,    \
                                                       c->tile_nu.p, c->lj_off.p, c->lj_split.p, c->lj16.p,            \
                                                       c->lj16_in.p, c->lj_len_in.p, c->lj_split_in.p);               \
  } while (0)
  if (c->tile_rows_cl == 1)
    MDP_TP(1);
  else
    MDP_TP(2);
#undef MDP_TP
  if (c->nall) hold_all_kernel<<<(c->nall + 255) / 256, 256, 0, st>>>(c->nall, c->xq.p, c->xhold_prune.p);
  MDP_HIP(c, hipGetLastError());
  c->prune_valid = true;
  c->prune_stale = false;
  c->prune_epoch++; // (a displacement check launched before this says nothing about the new reference)
  c->prune_copied_epoch = c->prune_epoch;
  c->prunes++;
  c->computes_since_prune = 0;
  return MDP_OK;
}

// the buffer adapts like the inner skin: a pruning costs about a third of a pass over the rows, so a trigger that
// fires within a dozen computes (thermal vibration reaching half the buffer) widens it, one that stays quiet for
// long narrows it.  fired: the trigger of the current pruning has fired (MDP_PRUNE_BUFFER fixes the buffer).
void mdp_prune_adapt(mdp_ctx *c, const double buf_max, const bool fired)
{
  const char *eb = getenv("MDP_PRUNE_BUFFER");
  const double fixed_buf = eb ? atof(eb) : 0.0;
  if (fixed_buf > 0.0) {
    c->prune_buf = fixed_buf;
    return;
  }
  if (!fired) return;
  if (c->computes_since_prune < 12 && c->prune_buf + 0.1 <= buf_max + 1e-9) c->prune_buf += 0.1;
  else if (c->computes_since_prune > 60 && c->prune_buf - 0.05 >= 0.2 - 1e-9) c->prune_buf -= 0.05;
}

// A style without list upkeep of its own (aeam) keeps the pruned rows current with this: reads the flags the
// integrate kernel and the halo unpack of the previous step left (MdpStyleCheck) and prunes (again) when needed.
// Call before the first kernel that walks the rows; positions (ghosts included) must be current -- unless
// may_prune is false: then a pruning that is due is only reported (*due; the caller comes back once the halo has
// arrived).
int mdp_prune_upkeep(mdp_ctx *c, const double cut[4], const double skin, const bool may_prune, bool *due)
{
  if (due) *due = false;
  const char *ep = getenv("MDP_PRUNE");
  const int prune_on = ep ? atoi(ep) : 1;
  if (!prune_on || !c->md || c->ntile <= 0 || c->tile_rows_cl != 2) {
    c->prune_valid = false;
    return MDP_OK;
  }
  MDP_TRY(mdp_sflag_collect(c, nullptr, nullptr)); // the previous step's check (integrate kernel, halo unpack)
  mdp_prune_adapt(c, skin - 0.2, c->prune_valid && c->prune_stale);
  if (!(c->prune_buf < skin)) {
    c->prune_valid = false;
    return MDP_OK;
  }
  if (!c->prune_valid || c->prune_stale) {
    if (!may_prune) {
      if (due) *due = true;
      return MDP_OK;
    }
    double lim[4];
    for (int k = 0; k < 4; k++) lim[k] = (cut[k] + c->prune_buf) * (cut[k] + c->prune_buf);
    MDP_TRY(mdp_tile_prune(c, lim));
  }
  c->computes_since_prune++;
  return MDP_OK;
}

// force_clear (optional) + compute on the device; results stay on the device (f, eatom, acc)
// one launch class of the Lennard-Jones units (see unit_class_kernel): 0 small unions, 1 large unions (2, 3: unused)
// Which variant of the tile kernel this compute takes (once per compute: the first class launch decides, the follow-up
// kernels read lj_queue_now).  Hot means: the compute before listed more than an eighth of the tiles for the cubic
// follow-up (its count, published to a pinned word by that follow-up, read without a wait) -- then walking every listed
// tile's rows a second time costs more than the tile kernel itself (a 3 300 K melt: 1.97 against 1.55 ms) and the pairs
// are queued where they are found instead.  A crystal lists nothing and keeps the kernel without the queue code.
// MDP_LJ_QUEUE = 0 / 1 forces the variant (tests, A/B).
static bool lj_queue_mode(mdp_ctx *c)
{
  const char *fe = getenv("MDP_LJ_QUEUE"); // (read per compute: the tests switch it within one process)
  const int forced = fe ? (atoi(fe) != 0 ? 1 : 0) : -1;
  bool on = false;
  if (c->lj_tiled && c->cluster == 2 && c->ntile > 0 && forced != 0) {
    const int listed = *(const int *) (c->h_pinned + 45);
    on = forced == 1 || (long long) listed * 8 > c->ntile;
  }
  const char *ce0 = getenv("MDP_LJ_QUEUE_CAP");
  const bool recap = ce0 && c->lj_cq_cap && atoi(ce0) >= 1 && atoi(ce0) <= 4096 && atoi(ce0) != c->lj_cq_cap;
  if (on && (c->lj_cq_tiles != c->ntile || !c->lj_cq.p || recap)) { // (first hot compute after a list build with another tile count)
    int kCap = 64; // items per queue: ten times what a 3 300 K melt of MoS2 puts into one (6.8 on average)
    if (const char *ce = getenv("MDP_LJ_QUEUE_CAP")) kCap = atoi(ce) >= 1 && atoi(ce) <= 4096 ? atoi(ce) : kCap; // (tests: overflow)
    if (c->lj_cq.reserve((size_t) c->ntile * MDP_TILE * 2 * kCap + 8) != hipSuccess ||
        c->lj_cq_cnt.reserve((size_t) c->ntile * MDP_TILE * 2 + 8) != hipSuccess ||
        hipMemsetAsync(c->lj_cq_cnt.p, 0, sizeof(int) * ((size_t) c->ntile * MDP_TILE * 2 + 8), c->stream) != hipSuccess)
      on = false; // (no memory for the queues: the walk serves)
    else {
      c->lj_cq_cap = kCap;
      c->lj_cq_tiles = c->ntile;
    }
  }
  c->lj_queue_now = on;
  return on;
}

static int launch_lj(mdp_ctx *c, int klass, bool gather, int eflag, int vflag, bool accumulate)
{
  int first = c->lj_class_base[klass], count = c->lj_class_base[klass + 1] - first;
  if (count <= 0) return MDP_OK;
  hipStream_t st = c->stream;
  const int *order = c->lj_ordered ? c->cl_order.p : nullptr;
  int skip_above = 1 << 30;
  if (c->lj_tiled && order && klass == 0) { // natural order, large tiles predicated away
    order = nullptr;
    first = 0;
    count = c->ntile;
    skip_above = c->tile_small;
  }
  const bool ev = eflag || vflag; // force-only steps take the variant without energy/virial arithmetic
  if (c->lj_tiled) {
    const bool pruned = c->prune_valid; // rows pruned to the pairs that can be inside a window right now
    const bool queue = c->lj_queue_now; // (decided once per compute: mdp_rebomos_run_end)
    const bool small = !(klass & 1);
    const int capL = ((small ? c->tile_small : c->tile_maxu) + 1 + 7) & ~7;
    const size_t lds = (size_t) capL * 3 * sizeof(double);
#define MDP_LJT_(EVV, GV, WV, CLV, QV)                                                                              \
  do {                                                                                                              \
    if (lds > 48 * 1024)                                                                                            \
      MDP_HIP(c, hipFuncSetAttribute((const void *) rebo_lj_tile_kernel<EVV, GV, WV, CLV, QV>,                      \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                       \
    rebo_lj_tile_kernel<EVV, GV, WV, CLV, QV><<<count, 256, lds, st>>>(                                             \
        c->rebomos, c->nlocal, order, first, c->nclus, c->xq.p, c->tile_cap, capL, skip_above, c->tu.p,            \
        c->tile_nu.p, c->lj_off.p, pruned ? c->lj_len_in.p : nullptr, pruned ? c->lj_split_in.p : c->lj_split.p,    \
        pruned ? c->lj16_in.p : c->lj16.p, c->cand_off.p, c->amask.p, c->rev.p, c->rev16.p,                         \
        c->fnbr.p,                                                                                                  \
        c->fown.p, c->f.p, c->eatom.p, c->acc.p, eflag, vflag, accumulate ? 1 : 0,                                 \
        c->ovf.p + (size_t) 5 * c->ovf_stride, c->lj_fix_stamp.p, c->lj_stamp, c->lj_cq.p, c->lj_cq_cnt.p,          \
        c->lj_cq_cap);                                                                                              \
  } while (0)
#define MDP_LJT(EVV, GV, WV)                                                                                        \
  do {                                                                                                              \
    if (c->cluster == 1) MDP_LJT_(EVV, GV, WV, 1, false);                                                           \
    else if (queue) MDP_LJT_(EVV, GV, WV, 2, true);                                                                 \
    else MDP_LJT_(EVV, GV, WV, 2, false);                                                                           \
  } while (0)
    // the force-only variants fit 5 waves per SIMD (<= 102 VGPRs); with small unions LDS allows 5 workgroups too
    if (ev && gather) MDP_LJT(true, true, 4);
    else if (ev) MDP_LJT(true, false, 4);
    else if (gather) MDP_LJT(false, true, 5);
    else MDP_LJT(false, false, 5);
#undef MDP_LJT
#undef MDP_LJT_
    return MDP_OK;
  }
  constexpr int L = 16;
  const int grid = (count + 256 / L - 1) / (256 / L);
#define MDP_LJ(CLV, EVV, GV)                                                                                        \
  rebo_lj_gather_kernel<CLV, L, EVV, GV><<<grid, 256, 0, st>>>(                                                      \
      c->rebomos, c->nlocal, order, first, count, c->xq.p, c->lj_off.p, c->lj_split.p, c->lj.p, c->cand_off.p,       \
      c->amask.p, c->rev.p, c->fnbr.p, c->fown.p, c->f.p, c->eatom.p, c->acc.p, eflag, vflag, accumulate ? 1 : 0)
#define MDP_LJ2(CLV)                                                                                                \
  do {                                                                                                              \
    if (ev && gather) MDP_LJ(CLV, true, true);                                                                      \
    else if (ev) MDP_LJ(CLV, true, false);                                                                          \
    else if (gather) MDP_LJ(CLV, false, true);                                                                      \
    else MDP_LJ(CLV, false, false);                                                                                 \
  } while (0)
  if (c->cluster == 1) MDP_LJ2(1);
  else if (c->cluster == 4) MDP_LJ2(4);
  else MDP_LJ2(2);
#undef MDP_LJ2
#undef MDP_LJ
  return MDP_OK;
}

// behind the Lennard-Jones tile launches of a compute: the corrections of the pairs on the cubic inner spline, for the
// tiles those launches listed (normally none: the kernel reads the count on the device and leaves)
static int launch_lj_cubic(mdp_ctx *c, int eflag, int vflag)
{
  if (!c->lj_tiled || c->ntile <= 0) return MDP_OK;
  const bool pruned = c->prune_valid;
  const bool queue = c->lj_queue_now;
  if (queue) { // the queues the tile launches of this compute filled; tiles whose queues overflowed go on to the walk below
    int *h_count = (int *) (c->h_pinned + 45);
    long long want = (long long) *h_count + *h_count / 4 + 256;
    const int grid = (int) (want < c->ntile ? want : c->ntile);
    const bool ev = eflag || vflag;
    const int *fix_list = c->ovf.p + (size_t) 5 * c->ovf_stride;
    int *walk_list = c->ovf.p + (size_t) 6 * c->ovf_stride;
    if (ev)
      rebo_lj_cubicq_kernel<2, true><<<grid, 256, 0, c->stream>>>(c->lj_fixtab.p, fix_list, walk_list, c->nlocal, c->nclus,
                                                                  c->xq.p, c->tile_cap, c->tu.p, c->lj_cq.p, c->lj_cq_cnt.p,
                                                                  c->lj_cq_cap, c->f.p, c->eatom.p, c->acc.p, eflag, vflag, h_count);
    else
      rebo_lj_cubicq_kernel<2, false><<<grid, 256, 0, c->stream>>>(c->lj_fixtab.p, fix_list, walk_list, c->nlocal, c->nclus,
                                                                   c->xq.p, c->tile_cap, c->tu.p, c->lj_cq.p, c->lj_cq_cnt.p,
                                                                   c->lj_cq_cap, c->f.p, c->eatom.p, c->acc.p, eflag, vflag, h_count);
    MDP_HIP(c, hipGetLastError());
  }
  const int capL = (c->tile_maxu + 1 + 7) & ~7;
  const size_t lds = (size_t) capL * 3 * sizeof(double);
  // a workgroup per listed tile, from the count an earlier compute published (any grid covers the list: the kernel
  // strides over it; a crystal lists nothing and gets 256 workgroups that leave at once)
  // (queue mode: this walk only serves the tiles the queue kernel passed on -- normally none; its count has its own word)
  int *h_count = (int *) (c->h_pinned + (queue ? 47 : 45));
  long long want = (long long) *h_count + *h_count / 4 + 256;
  const int grid = (int) (want < c->ntile ? want : c->ntile);
  const int *fix_list = c->ovf.p + (size_t) (queue ? 6 : 5) * c->ovf_stride;
#define MDP_LJC(CLV, EVV)                                                                                            \
  do {                                                                                                                \
    if (lds > 48 * 1024)                                                                                              \
      MDP_HIP(c, hipFuncSetAttribute((const void *) rebo_lj_cubic_kernel<CLV, EVV>,                                   \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                         \
    rebo_lj_cubic_kernel<CLV, EVV><<<grid, 256, lds, c->stream>>>(                                                    \
        c->lj_fixtab.p, fix_list, c->nlocal, c->nclus, c->xq.p, c->tile_cap, c->tu.p, c->tile_nu.p, c->lj_off.p,      \
        pruned ? c->lj_len_in.p : nullptr, pruned ? c->lj_split_in.p : c->lj_split.p,                                 \
        pruned ? c->lj16_in.p : c->lj16.p, c->f.p, c->eatom.p, c->acc.p, eflag, vflag, h_count);                      \
  } while (0)
  const bool ev = eflag || vflag;
  if (c->cluster == 1) {
    if (ev) MDP_LJC(1, true);
    else MDP_LJC(1, false);
  } else {
    if (ev) MDP_LJC(2, true);
    else MDP_LJC(2, false);
  }
#undef MDP_LJC
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// parts: bit 0 = interior centres, bit 1 = boundary centres (+ the overflow pass, which must follow both)
// which (interior part only): bit 0 = the lane-per-centre kernel, bit 1 = the lane-group kernels; `first` = this is the
// first centre launch of the compute
static int launch_centres(mdp_ctx *c, int eflag, int vflag, int parts, int which = 3, bool first = true)
{
  if ((parts & 1) && first) c->ovf_par ^= 1; // (a new compute: the set of pinned overflow counts it reads first and writes last)
  hipStream_t st = c->stream; // (the overflow counter ovf[0] was zeroed by mdp_acc_begin of this compute)
  if (c->centre_split && parts == 3 && which == 3) {
    // nothing of this compute ran behind an exchange (a blocking order of the multi-GPU step): one launch per class over
    // both of its halves instead of two -- the boundary halves are 6 % of the centres and a third of the launches
    launch_centre3(c, eflag, vflag, 2);
    launch_centre<8>(c, 1, eflag, vflag, 2);
    launch_centre<12>(c, 2, eflag, vflag, 2);
    launch_centre<16>(c, 3, eflag, vflag, 2);
    launch_centre<32>(c, 4, eflag, vflag, 2);
  } else
    for (int part = 0; part < 2; part++) {
      if (!((parts >> part) & 1)) continue;
      const int w = part == 0 ? which : 3;
      if (w & 1) launch_centre3(c, eflag, vflag, part);
      if (w & 2) {
        launch_centre<8>(c, 1, eflag, vflag, part);
        launch_centre<12>(c, 2, eflag, vflag, part);
        launch_centre<16>(c, 3, eflag, vflag, part);
        launch_centre<32>(c, 4, eflag, vflag, part);
      }
    }
  MDP_HIP(c, hipGetLastError());
  if (!(parts & 2)) return MDP_OK;
  // centres that outgrew their lane group since the last build (normally none: the kernel reads the
  // count from the device and exits)
  int total = 0;
  for (int k = 0; k < MDP_NCLASS; k++) total += c->h_class_count[k];
  const int grid = total > 0 ? (total / 8 + 1 < 64 ? total / 8 + 1 : 64) : 0; // grid-stride; normally nothing to do
  mdp_time_mark(c, 1); // (marks 1 -> 2: the general kernel, i.e. the centres that outgrew their lane group)
  if (grid)
    rebo_centre_general_kernel<false><<<grid, 256, 0, st>>>(c->rebomos, c->ovf.p, -1, c->nlocal, c->xq.p,
                                                            c->cand_off.p, c->cand.p, c->amask.p, c->fnbr.p,
                                                            c->fown.p, nullptr, nullptr, c->acc.p, c->flags.p,
                                                            eflag, vflag, (int *) (c->h_pinned + 40) + 4 * c->ovf_par,
                                                            c->ovf_stride, (int *) (c->h_pinned + 44));
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// per-atom virial steps: every centre goes through the general kernel's VATOM variant
static int launch_centres_vatom(mdp_ctx *c, int eflag, int vflag)
{
  hipStream_t st = c->stream;
  for (int k = 0; k < MDP_NCLASS; k++) {
    const int n = c->h_class_count[k];
    if (n <= 0) continue;
    const int grid = n / 8 + 1 < 2048 ? n / 8 + 1 : 2048;
    rebo_centre_general_kernel<true><<<grid, 256, 0, st>>>(c->rebomos, c->class_list.p + (size_t) k * c->nall, n,
                                                           c->nlocal, c->xq.p, c->cand_off.p, c->cand.p, c->amask.p,
                                                           c->fnbr.p, c->fown.p, c->vslot.p, c->vatom.p, c->acc.p,
                                                           c->flags.p, eflag, vflag);
  }
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// First half of compute(): everything that does not need this step's REMOTE ghost positions -- the list
// upkeep and the REBO centres whose candidate sets reach no remote ghost.  Runs while the halo exchange is in flight.
int mdp_rebomos_run_begin(mdp_ctx *c, int eflag, int vflag)
{
  c->computes_since_build++;
  if (c->rebo_packed) {
    bool stale = false;
    MDP_TRY(rebomos_lists_stale(c, stale));
    if (stale) {
      c->rebo_packed = false;
      c->stale_rebuild = true; // the style's own trigger, not the host's reneighboring
    }
  }
  if (!c->rebo_packed) MDP_TRY(mdp_rebomos_repack(c));
  MDP_TRY(mdp_acc_begin(c, eflag || vflag));
  c->lj_stamp = c->lj_stamp >= 0x7ffffff0 ? 1 : c->lj_stamp + 1; // (the tiles' "listed in this compute" words: never 0)
  mdp_time_mark(c, 0);
  if (vflag & MDP_VFLAG_ATOM) {
    MDP_HIP(c, c->vatom.reserve((size_t) 6 * c->nall + 6));
    MDP_HIP(c, c->vslot.reserve((size_t) 6 * c->cand_total + 6));
    MDP_HIP(c, hipMemsetAsync(c->vatom.p, 0, sizeof(double) * 6 * c->nall, c->stream));
  }
  // (overlap_mode: what of the interior work runs here, behind the start of the halo exchange -- MdpDomain::ov_policy)
  c->centres_early = 0;
  if (c->centre_split && !(vflag & MDP_VFLAG_ATOM) && c->overlap_mode != 2) {
    c->centres_early = c->overlap_mode == 3 ? 1 : 3;
    MDP_TRY(launch_centres(c, eflag, vflag, /*interior*/ 1, c->centres_early));
  }
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// Second half: the remaining REBO centres, then the fused Lennard-Jones + slot-gather kernel over all tiles
int mdp_rebomos_run_end(mdp_ctx *c, int eflag, int vflag)
{
  hipStream_t st = c->stream;
  const bool va = (vflag & MDP_VFLAG_ATOM) != 0;
  if (va)
    MDP_TRY(launch_centres_vatom(c, eflag, vflag));
  else if (!c->centre_split)
    MDP_TRY(launch_centres(c, eflag, vflag, 3));
  else if (c->centres_early == 3)
    MDP_TRY(launch_centres(c, eflag, vflag, /*boundary*/ 2));
  else // the interior kernels that did not run behind the exchange, then the boundary centres
    MDP_TRY(launch_centres(c, eflag, vflag, 3, 3 & ~c->centres_early, /*first=*/c->centres_early == 0));
  if (va) mdp_time_mark(c, 1);
  mdp_time_mark(c, 2);
  (void) lj_queue_mode(c); // which variant the tile launches of this compute take (and their follow-up reads)
  if (va) {
    mdp_time_mark(c, 3);
    c->prune_valid = false; // (these paths walk the rows as built)
    for (int k = 0; k < 4; k++) MDP_TRY(launch_lj(c, k, false, eflag, vflag, false));
    MDP_TRY(launch_lj_cubic(c, eflag, vflag));
    if (c->nlocal) {
      rebo_gather_kernel<8><<<(c->nlocal + 31) / 32, 256, 0, st>>>(c->nlocal, c->cand_off.p, c->amask.p, c->rev.p,
                                                                   c->fnbr.p, c->fown.p, c->f.p, c->eatom.p, eflag,
                                                                   va ? c->vslot.p : nullptr, c->vatom.p);
      if (va) {
        const int grid = (c->nlocal + 31) / 32;
        const unsigned short *lj16 = c->lj_tiled ? c->lj16.p : nullptr;
        if (c->cluster == 1) rebo_lj_vatom_kernel<1><<<grid, 256, 0, st>>>(c->rebomos, c->nlocal, c->xq.p, c->lj_off.p, c->lj.p, lj16, c->tu.p, c->tile_cap, c->tile_nu.p, c->vatom.p, MDP_TILE);
        else if (c->cluster == 4) rebo_lj_vatom_kernel<4><<<grid, 256, 0, st>>>(c->rebomos, c->nlocal, c->xq.p, c->lj_off.p, c->lj.p, lj16, c->tu.p, c->tile_cap, c->tile_nu.p, c->vatom.p, MDP_TILE);
        else rebo_lj_vatom_kernel<2><<<grid, 256, 0, st>>>(c->rebomos, c->nlocal, c->xq.p, c->lj_off.p, c->lj.p, lj16, c->tu.p, c->tile_cap, c->tile_nu.p, c->vatom.p, MDP_TILE);
      }
    }
  } else {
    // the kernels walk pruned rows (see tile_prune_kernel) -- resident runs with the deferred displacement trigger,
    // host mode with the blocking check every compute does anyway (rebomos_lists_stale); a per-atom-virial step
    // reads the rows as built
    const char *ep = getenv("MDP_PRUNE");
    const int prune_on = ep ? atoi(ep) : 1;
    if (prune_on && c->lj_tiled && c->ntile > 0) {
      mdp_prune_adapt(c, c->skin_inner - 0.2, c->prune_valid && c->prune_stale);
      if (c->prune_buf < c->skin_inner) { // (a buffer as wide as the skin prunes nothing)
        if (!c->prune_valid || c->prune_stale) {
          double lim[4];
          for (int k = 0; k < 4; k++) {
            const double r = sqrt(c->rebomos.lj_rsq_hi[k]) + c->prune_buf;
            lim[k] = r * r;
          }
          MDP_TRY(mdp_tile_prune(c, lim));
        }
        c->computes_since_prune++;
      } else
        c->prune_valid = false;
    } else
      c->prune_valid = false;
    mdp_time_mark(c, 3); // (a row pruning, when one was due, lies between marks 2 and 3)
    for (int k = 0; k < 4; k++) MDP_TRY(launch_lj(c, k, /*gather=*/true, eflag, vflag, false));
    MDP_TRY(launch_lj_cubic(c, eflag, vflag));
  }
  MDP_HIP(c, hipGetLastError());
  mdp_time_mark(c, 4);
  return mdp_acc_end(c, eflag || vflag);
}

// force_clear + compute on the device; results stay on the device (f, eatom, acc)
int mdp_rebomos_run(mdp_ctx *c, int eflag, int vflag, bool zero_f)
{
  (void) zero_f; // owned forces are always overwritten; the host-mode caller adds them on the host
  MDP_TRY(mdp_rebomos_run_begin(c, eflag, vflag));
  return mdp_rebomos_run_end(c, eflag, vflag);
}
