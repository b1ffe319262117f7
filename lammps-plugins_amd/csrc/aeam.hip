// aeam.hip -- angular-EAM hot path for gfx950 (wave64), FP64 throughout.
//
// Replaces PairAEAM::compute (USER-AEAM/pair_aeam.cpp:110-479): pass 1 density (:164-253),
// pass 2 embedding (:264-303), pass 3 pair + embedding + angular forces (:309-476).
//
// The reference scatters every (i,j) visit onto f[i] AND f[j] (ghosts included).  On the GPU that
// would be ~260 FP64 atomics per atom, so the two-body part is regrouped as a gather: an owned atom
// a collects, from its own full list, both visits that touch it,
//     F_a -= d (fpair_{a->j} + fpair_{j->a}),   d = x_j - x_a
//     fpair_{i->j} = -[i metal] q_i f'_{ti,tj}(r)/r - 1/2 phi'_{ti,tj}(r)/r      (pair_aeam.cpp:371-376)
// where q_i = Fptmp_i * F'_i.  q_j of ghost neighbours is the one quantity that has to be
// communicated -- exactly the style's own forward_comm of fp (pair_aeam.cpp:307).
// Angular (covalent) centres are rare (0.75 % in sample.in): each gets a whole wave, stages its
// in-range neighbours in LDS, loops over the (j<k) triplets and scatters fj / fk with FP64 atomics
// (these are the only forces that land on ghost atoms).
//
// aeam_density_kernel<L>   L lanes per owned metal atom: rho_i = sum_j f(r)            (A1)
// aeam_density_ang_kernel  one wave per owned angular atom: triplet density             (A1)
// aeam_embed_kernel        F, F', q_i, energy                                           (A2)
// aeam_force_kernel<L>     L lanes per owned atom, both visits of every pair            (A3)
// aeam_force_ang_kernel    one wave per owned angular atom, triplet forces              (A4)
#include "mdp_common.h"

#include <type_traits>

namespace {

constexpr int AE_L = 8;      // lanes per atom in the list-streaming kernels
constexpr int ANG_CAP = 160; // LDS slots per angular centre
constexpr double kCutDec = 1.5; // pair_aeam.cpp:188

template <int W> __device__ __forceinline__ double lane_sum(double v)
{
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// spline row + fractional coordinate (pair_aeam.cpp:195-201)
__device__ __forceinline__ const double *spline_row(const double *__restrict__ tab, int table, int nmax1, double r,
                                                    double rdr, int nr, double &p)
{
  p = r * rdr + 1.0;
  int m = (int) p;
  m = m < nr - 1 ? m : nr - 1;
  p -= m;
  p = p < 1.0 ? p : 1.0;
  return tab + ((size_t) table * nmax1 + m) * 7;
}

__device__ __forceinline__ double sp_val(const double *c, double p) { return ((c[3] * p + c[4]) * p + c[5]) * p + c[6]; }
__device__ __forceinline__ double sp_der(const double *c, double p) { return (c[0] * p + c[1]) * p + c[2]; }

// The same coefficients re-laid for the streaming kernels: one 32-byte aligned record per row and use,
//   val4[table][row] = {c3,c4,c5,c6}  (value),   der4[table][row] = {c0,c1,c2,0}  (derivative)
// so a lookup is a single aligned 32-byte gather instead of 56 bytes straddling cache lines.
__device__ __forceinline__ double v4_val(const double4 c, double p) { return ((c.x * p + c.y) * p + c.z) * p + c.w; }
__device__ __forceinline__ double d4_der(const double4 c, double p) { return (c.x * p + c.y) * p + c.z; }

__device__ __forceinline__ int spline_index(double r, double rdr, int nr, double &p)
{
  p = r * rdr + 1.0;
  int m = (int) p;
  m = m < nr - 1 ? m : nr - 1;
  p -= m;
  p = p < 1.0 ? p : 1.0;
  return m;
}

// per-pair-type parameters of one centre type against every neighbour type, kept in registers
// (selected with compares on the neighbour's type: no per-lane loads from the parameter block)
template <int NT, typename T> __device__ __forceinline__ T pick(const T (&a)[NT], const int t)
{
  T v = a[0];
#pragma unroll
  for (int k = 1; k < NT; k++) v = (t == k) ? a[k] : v;
  return v;
}

template <int NT> struct PairPar {
  double cut[NT], rdr[NT];
  int nr[NT], trho[NT], tz2r[NT];
  __device__ __forceinline__ double cut_(const int t) const { return pick<NT>(cut, t); }
  __device__ __forceinline__ double rdr_(const int t) const { return pick<NT>(rdr, t); }
  __device__ __forceinline__ int nr_(const int t) const { return pick<NT>(nr, t); }
  __device__ __forceinline__ int trho_(const int t) const { return pick<NT>(trho, t); }
  __device__ __forceinline__ int tz2r_(const int t) const { return pick<NT>(tz2r, t); }
};
// NT = 0: any number of atom types (more than MDP_AEAM_MAXT: the reference sizes everything from the file,
// pair_aeam.cpp:752-872) -- the parameters of a pair are read from the block in device memory, per neighbour
template <> struct PairPar<0> {
  const double *gc, *gr;
  const int *gn, *gt, *gz;
  int base, stride;
  __device__ __forceinline__ double cut_(const int t) const { return gc[base + t * stride]; }
  __device__ __forceinline__ double rdr_(const int t) const { return gr[base + t * stride]; }
  __device__ __forceinline__ int nr_(const int t) const { return gn[base + t * stride]; }
  __device__ __forceinline__ int trho_(const int t) const { return gt[base + t * stride]; }
  __device__ __forceinline__ int tz2r_(const int t) const { return gz[base + t * stride]; }
};

template <int NT> __device__ __forceinline__ PairPar<NT> load_pairpar(const AeamDev &A, const int ti, const bool transposed)
{
  PairPar<NT> q;
  if constexpr (NT == 0) {
    q.gc = A.g_cut;
    q.gr = A.g_rdr;
    q.gn = A.g_nr;
    q.gt = A.g_t2rhor;
    q.gz = A.g_t2z2r;
    q.base = transposed ? ti : ti * A.ntypes;
    q.stride = transposed ? A.ntypes : 1;
  } else {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int pt = transposed ? t * A.ntypes + ti : ti * A.ntypes + t;
      q.cut[t] = A.cut[pt];
      q.rdr[t] = A.rdr[pt];
      q.nr[t] = A.nr[pt];
      q.trho[t] = A.t2rhor[pt];
      q.tz2r[t] = A.t2z2r[pt];
    }
  }
  return q;
}

// ---- pass 1, metal centres (pair_aeam.cpp:174-205) ---------------------------------------------------
template <int L, int NT>
__global__ __launch_bounds__(256) void aeam_density_kernel(const AeamDev A, const int nlocal,
                                                           const double4 *__restrict__ xq,
                                                           const long long *__restrict__ nb_off,
                                                           const int *__restrict__ nb, double *__restrict__ rho)
{
  constexpr int U = 2;
  const int s = threadIdx.x % L;
  const long long i64 = (long long) blockIdx.x * (256 / L) + threadIdx.x / L;
  const bool have = i64 < nlocal;
  const int i = have ? (int) i64 : 0;
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  double acc = 0.0;
  const bool metal = ti < A.nnonangular;
  if (have && metal) {
    const PairPar<NT> q = load_pairpar<NT>(A, ti, false);
    const int nm1 = A.nrmax + 1;
    const long long b = nb_off[i], e = nb_off[i + 1];
    for (long long k0 = b; k0 < e; k0 += U * L) {
      int jj[U];
      double4 xj[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long k = k0 + u * L + s;
        jj[u] = k < e ? nb[k] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; u++) xj[u] = xq[jj[u] >= 0 ? jj[u] : i];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (jj[u] < 0) continue;
        const double dx = xj[u].x - xi.x, dy = xj[u].y - xi.y, dz = xj[u].z - xi.z;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const int tj = (int) xj[u].w;
        if (r <= q.cut_(tj)) { // CutDec applies only when BOTH are angular (pair_aeam.cpp:187-190)
          double p;
          const int m = spline_index(r, q.rdr_(tj), q.nr_(tj), p);
          acc += v4_val(A.rhor_v4[(size_t) q.trho_(tj) * nm1 + m], p);
        }
      }
    }
  }
  acc = lane_sum<L>(acc);
  if (have && metal && s == 0) rho[i] = acc;
}

// ------------------------------------------------------------------------------------------------------
// Tile-list variants (resident mode and device-built lists).  Same scheme as the REBO-MoS
// Lennard-Jones kernel (csrc/rebomos.hip): one workgroup = one tile of 16 two-atom clusters; the UNION of their
// neighbourhoods is gathered once into LDS, the cluster rows are 16-bit indices into it, segmented by the
// neighbour's type (so every table selector is a choice between two scalars by the cluster atom's type) and
// padded to wave-uniform lengths with a dummy entry that lies outside every cutoff.  What remains in global
// memory per pair are the spline rows themselves.
// ------------------------------------------------------------------------------------------------------
// r and 1/r from one reciprocal square root: hardware seed + two Newton steps (~1 ulp) instead of an IEEE sqrt
// and an IEEE divide (~45 instructions); only the tile kernels use it (they are arithmetic-bound, the CSR
// kernels are not)
__device__ __forceinline__ double rsqrt_nr(double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}

// one Newton step: ~2e-15 relative (the seed is good to 2^-26).  Enough for the persistent kernels' row coordinate
// (a spline is continuous across rows) and 1/r, five orders below the parity tolerance
__device__ __forceinline__ double rsqrt_n1(double x)
{
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}

constexpr int kTile = 16; // clusters per tile (= MDP_TILE of rebomos.hip)

__device__ __forceinline__ int xcd_contiguous(const int b, const int n)
{
  const int q = n >> 3, r = n & 7, xcd = b & 7, idx = b >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// parameters of the visit (i = cluster atom of type ta, j of type TJ), or transposed (i = j's type, j = ta)
struct TilePar {
  double cut, rdr;
  int nr, trho, tz2r, pair; // pair = ti * 2 + tj of the visit
};
template <int TJ, bool TRANSPOSED> __device__ __forceinline__ TilePar tile_par(const AeamDev &A, const int ta)
{
  // ntypes == 2: pair index ti * 2 + tj
  constexpr int p0 = TRANSPOSED ? TJ * 2 + 0 : 0 * 2 + TJ, p1 = TRANSPOSED ? TJ * 2 + 1 : 1 * 2 + TJ;
  TilePar q;
  q.cut = ta ? A.cut[p1] : A.cut[p0];
  q.rdr = ta ? A.rdr[p1] : A.rdr[p0];
  q.nr = ta ? A.nr[p1] : A.nr[p0];
  q.trho = ta ? A.t2rhor[p1] : A.t2rhor[p0];
  q.tz2r = ta ? A.t2z2r[p1] : A.t2z2r[p0];
  q.pair = ta ? p1 : p0;
  return q;
}

// More than two atom types (MULTI): the tile lists keep their two segments -- neighbours of type 0 | neighbours of
// any other type -- and the kernels read the type of an entry of the second segment from LDS next to its
// coordinates; the per-pair-type parameters of all ntypes^2 pairs sit in LDS as well (the two-type kernels select
// them from scalars by the cluster atom's type).  Layout behind the union's records:
//   double cut[P], rdr[P]; int nr[P], trho[P], tz2r[P]  (P = MAXT^2)   then   int type[capL]
constexpr int kParN = MDP_AEAM_MAXT * MDP_AEAM_MAXT;
constexpr size_t kParBytes = (size_t) kParN * (2 * sizeof(double) + 3 * sizeof(int));
struct ParLds {
  const double *cut, *rdr;
  const int *nr, *trho, *tz2r;
};
__device__ __forceinline__ ParLds par_fill(const AeamDev &A, double *base, const int tid)
{
  double *cut = base, *rdr = base + kParN;
  int *nr = reinterpret_cast<int *>(base + 2 * kParN), *trho = nr + kParN, *tz2r = trho + kParN;
  const int np = A.ntypes * A.ntypes;
  if (tid < np) {
    cut[tid] = A.cut[tid];
    rdr[tid] = A.rdr[tid];
    nr[tid] = A.nr[tid];
    trho[tid] = A.t2rhor[tid];
    tz2r[tid] = A.t2z2r[tid];
  }
  return ParLds{cut, rdr, nr, trho, tz2r};
}
__device__ __forceinline__ TilePar par_at(const ParLds &T, const int pt)
{
  TilePar q;
  q.cut = T.cut[pt];
  q.rdr = T.rdr[pt];
  q.nr = T.nr[pt];
  q.trho = T.trho[pt];
  q.tz2r = T.tz2r[pt];
  q.pair = pt;
  return q;
}

// pass 1, metal centres (pair_aeam.cpp:174-205)
template <int CL, bool MULTI>
__global__ __launch_bounds__(256) void aeam_tile_density_kernel(
    const AeamDev A, const int nlocal, const int nclus, const int t_begin, const double4 *__restrict__ xq, const int cap,
    const int capL, const int *__restrict__ tu, const int *__restrict__ tile_nu, const long long *__restrict__ lj_off,
    const int *__restrict__ lj_len /* lengths of pruned rows, or null */, const int *__restrict__ lj_split,
    const unsigned short *__restrict__ lj16, double *__restrict__ rho)
{
  constexpr int L = 16, SK = 3;
  extern __shared__ double s_pos[]; // [capL][3]  (MULTI: then the parameter block and type[capL])
  const int tid = threadIdx.x, lane = tid & 63, s = lane % L;
  const int t = t_begin + xcd_contiguous(blockIdx.x, gridDim.x); // the launch walks tiles [t_begin, t_begin + grid)
  const int kc = t * kTile + tid / L;
  const bool have = kc < nclus;
  const int nU = tile_nu[2 * t];
  const int *__restrict__ mem = tu + (size_t) t * cap;
  int sidx[SK];
#pragma unroll
  for (int k = 0; k < SK; k++) sidx[k] = mem[tid + 256 * k];
  const long long b = lj_off[kc];
  const int cnt = __builtin_amdgcn_readfirstlane(lj_len ? lj_len[kc] : (int) (lj_off[kc + 1] - b));
  const int split = __builtin_amdgcn_readfirstlane(lj_split[kc]);
  const unsigned short *__restrict__ row = lj16 + b;
  double4 xa[CL];
#pragma unroll
  for (int c = 0; c < CL; c++) xa[c] = xq[have && kc * CL + c < nlocal ? kc * CL + c : nlocal - 1];
  ParLds T = {};
  int *s_ty = nullptr;
  if (MULTI) {
    T = par_fill(A, s_pos + 3 * (size_t) capL, tid);
    s_ty = reinterpret_cast<int *>(reinterpret_cast<char *>(s_pos + 3 * (size_t) capL) + kParBytes);
  }
  {
    double4 sv[SK];
#pragma unroll
    for (int k = 0; k < SK; k++) sv[k] = xq[tid + 256 * k < nU ? sidx[k] : 0];
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int u = tid + 256 * k;
      if (u < nU) {
        s_pos[3 * u] = sv[k].x;
        s_pos[3 * u + 1] = sv[k].y;
        s_pos[3 * u + 2] = sv[k].z;
        if (MULTI) s_ty[u] = (int) sv[k].w;
      }
    }
  }
  for (int u = tid + 256 * SK; u < nU; u += 256) {
    const double4 v = xq[mem[u]];
    s_pos[3 * u] = v.x;
    s_pos[3 * u + 1] = v.y;
    s_pos[3 * u + 2] = v.z;
    if (MULTI) s_ty[u] = (int) v.w;
  }
  if (tid == 0) {
    s_pos[3 * nU] = 1.0e30;
    s_pos[3 * nU + 1] = 0.0;
    s_pos[3 * nU + 2] = 0.0;
    if (MULTI) s_ty[nU] = 0;
  }
  int ta[CL];
  bool metal[CL];
  double acc[CL];
#pragma unroll
  for (int c = 0; c < CL; c++) {
    ta[c] = (int) xa[c].w;
    metal[c] = have && kc * CL + c < nlocal && ta[c] < A.nnonangular;
    acc[c] = 0.0;
  }
  __syncthreads();
  const int nm1 = A.nrmax + 1;
  auto segment = [&](auto tjc) {
    constexpr int TJ = decltype(tjc)::value;
    const int kb = TJ ? split : 0, ke = TJ ? cnt : split;
    TilePar q[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) q[c] = MULTI ? par_at(T, ta[c] * A.ntypes) : tile_par<TJ, false>(A, ta[c]);
    int li_next = kb + s < ke ? (int) row[kb + s] : nU; // the next entry's index is requested one trip ahead
    for (int k = kb + s; k < ke; k += L) {
      const int li = li_next;
      li_next = k + L < ke ? (int) row[k + L] : nU;
      const double *p3 = s_pos + 3 * li;
      const double xj = p3[0], yj = p3[1], zj = p3[2];
      if (MULTI && TJ == 1) { // the entry's own type
        const int tj = s_ty[li];
#pragma unroll
        for (int c = 0; c < CL; c++) q[c] = par_at(T, ta[c] * A.ntypes + tj);
      }
#pragma unroll
      for (int c = 0; c < CL; c++) {
        const double dx = xj - xa[c].x, dy = yj - xa[c].y, dz = zj - xa[c].z;
        const double rsq = dx * dx + dy * dy + dz * dz;
        const double r = rsq * rsqrt_nr(rsq > 0.0 ? rsq : 1.0);
        // rsq > 0: the union holds the cluster's own atoms too.  CutDec applies only when BOTH are angular
        // (pair_aeam.cpp:187-190), never here.
        const bool in = metal[c] && rsq > 0.0 && r <= q[c].cut;
        if (in) { // (a third of the entries are skin / padding: no spline row fetched for them)
          double pf;
          const int m = spline_index(r, q[c].rdr, q[c].nr, pf);
          acc[c] += v4_val(A.rhor_v4[(size_t) q[c].trho * nm1 + m], pf);
        }
      }
    }
  };
  segment(std::integral_constant<int, 0>{});
  segment(std::integral_constant<int, 1>{});
#pragma unroll
  for (int c = 0; c < CL; c++) acc[c] = lane_sum<L>(acc[c]);
  if (s < CL) {
#pragma unroll
    for (int c = 0; c < CL; c++)
      if (c == s && metal[c]) rho[kc * CL + c] = acc[c];
  }
}

// pass 3, pair part (pair_aeam.cpp:309-393), force only: both visits that touch the cluster atom, as in
// aeam_force_kernel.  LDS record of a union member: x y z q with q = Fptmp*F' of metal neighbours, 0 otherwise.
// EV: also the pair energy (global and per atom) and the global virial of this rank's own visits, tallied exactly
// as aeam_force_kernel<.., true> does (ev_tally of the visit i = a: pair_aeam.cpp:386-393).
template <int CL, bool EV, bool MULTI>
__global__ __launch_bounds__(256) void aeam_tile_force_kernel(
    const AeamDev A, const int nlocal, const int nclus, const int t_begin, const double4 *__restrict__ xq,
    const double *__restrict__ fp, const int cap, const int capL, const int *__restrict__ tu, const int *__restrict__ tile_nu,
    const long long *__restrict__ lj_off, const int *__restrict__ lj_len /* lengths of pruned rows, or null */,
    const int *__restrict__ lj_split, const unsigned short *__restrict__ lj16,
    double *__restrict__ f, double *__restrict__ eatom, double *__restrict__ acc, const int eflag, const int vflag)
{
  constexpr int L = 16, SK = 3;
  extern __shared__ double s_rec[]; // [capL][4]
  double4 *__restrict__ s4 = reinterpret_cast<double4 *>(s_rec);
  const int tid = threadIdx.x, lane = tid & 63, s = lane % L;
  const int t = t_begin + xcd_contiguous(blockIdx.x, gridDim.x); // the launch walks tiles [t_begin, t_begin + grid)
  const int kc = t * kTile + tid / L;
  const bool have = kc < nclus;
  const int nU = tile_nu[2 * t], N0 = tile_nu[2 * t + 1]; // members [0,N0) are of type 0
  const int *__restrict__ mem = tu + (size_t) t * cap;
  int sidx[SK];
#pragma unroll
  for (int k = 0; k < SK; k++) sidx[k] = mem[tid + 256 * k];
  const long long b = lj_off[kc];
  const int cnt = __builtin_amdgcn_readfirstlane(lj_len ? lj_len[kc] : (int) (lj_off[kc + 1] - b));
  const int split = __builtin_amdgcn_readfirstlane(lj_split[kc]);
  const unsigned short *__restrict__ row = lj16 + b;
  double4 xa[CL];
  double qa[CL];
#pragma unroll
  for (int c = 0; c < CL; c++) {
    const int ia = have && kc * CL + c < nlocal ? kc * CL + c : nlocal - 1;
    xa[c] = xq[ia];
    qa[c] = fp[ia];
  }
  ParLds T = {};
  int *s_ty = nullptr;
  if (MULTI) {
    T = par_fill(A, s_rec + 4 * (size_t) capL, tid);
    s_ty = reinterpret_cast<int *>(reinterpret_cast<char *>(s_rec + 4 * (size_t) capL) + kParBytes);
  }
  {
    double4 sv[SK];
    double sq[SK];
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int j = tid + 256 * k < nU ? sidx[k] : 0;
      sv[k] = xq[j];
      sq[k] = fp[j];
    }
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int u = tid + 256 * k;
      if (u < nU) {
        const int tu_ = MULTI ? (int) sv[k].w : (u < N0 ? 0 : 1);
        s4[u] = make_double4(sv[k].x, sv[k].y, sv[k].z, tu_ < A.nnonangular ? sq[k] : 0.0);
        if (MULTI) s_ty[u] = tu_;
      }
    }
  }
  for (int u = tid + 256 * SK; u < nU; u += 256) {
    const int j = mem[u];
    const double4 v = xq[j];
    const int tu_ = MULTI ? (int) v.w : (u < N0 ? 0 : 1);
    s4[u] = make_double4(v.x, v.y, v.z, tu_ < A.nnonangular ? fp[j] : 0.0);
    if (MULTI) s_ty[u] = tu_;
  }
  if (tid == 0) {
    s4[nU] = make_double4(1.0e30, 0.0, 0.0, 0.0);
    if (MULTI) s_ty[nU] = 0;
  }
  int ta[CL];
  bool real[CL];
  double fx[CL], fy[CL], fz[CL], ea[CL];
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
#pragma unroll
  for (int c = 0; c < CL; c++) {
    ta[c] = (int) xa[c].w;
    real[c] = have && kc * CL + c < nlocal;
    if (!(ta[c] < A.nnonangular)) qa[c] = 0.0; // (1 - deli): angular centres embed through the three-body kernel
    fx[c] = fy[c] = fz[c] = ea[c] = 0.0;
  }
  __syncthreads();
  const int nm1 = A.nrmax + 1;
  auto segment = [&](auto tjc) {
    constexpr int TJ = decltype(tjc)::value;
    const int kb = TJ ? split : 0, ke = TJ ? cnt : split;
    TilePar qA[CL], qJ[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) {
      qA[c] = MULTI ? par_at(T, ta[c] * A.ntypes) : tile_par<TJ, false>(A, ta[c]); // visit (i = a, j)
      qJ[c] = MULTI ? par_at(T, ta[c]) : tile_par<TJ, true>(A, ta[c]);             // visit (i = j, a)
    }
    // Per trip: the geometry of all cluster atoms first, then the records of their visits (a -> j) requested
    // together (each under its own predicate), then the arithmetic: the round trips of the cluster atoms overlap.
    // (Batching several ENTRIES was tried and lost to the occupancy it costs.)
    int li_next = kb + s < ke ? (int) row[kb + s] : nU; // the next entry's index is requested one trip ahead
    for (int k = kb + s; k < ke; k += L) {
      const double4 xj = s4[li_next];
      int tj = TJ;
      if (MULTI && TJ == 1) { // the entry's own type
        tj = s_ty[li_next];
#pragma unroll
        for (int c = 0; c < CL; c++) {
          qA[c] = par_at(T, ta[c] * A.ntypes + tj);
          qJ[c] = par_at(T, tj * A.ntypes + ta[c]);
        }
      }
      li_next = k + L < ke ? (int) row[k + L] : nU;
      double dx[CL], dy[CL], dz[CL], recip[CL], r[CL], pfa[CL];
      bool in_a[CL], in_j[CL];
      int ma[CL];
      double2 r0[CL], r1[CL], r2[CL];
#pragma unroll
      for (int c = 0; c < CL; c++) {
        dx[c] = xj.x - xa[c].x;
        dy[c] = xj.y - xa[c].y;
        dz[c] = xj.z - xa[c].z;
        const double rsq = dx[c] * dx[c] + dy[c] * dy[c] + dz[c] * dz[c];
        const bool pair = real[c] && rsq > 0.0; // (the union holds the cluster's own atoms too)
        recip[c] = rsqrt_n1(pair ? rsq : 1.0);
        r[c] = rsq * recip[c];
        in_a[c] = pair && r[c] <= qA[c].cut;
        in_j[c] = pair && r[c] <= qJ[c].cut;
        ma[c] = spline_index(r[c], qA[c].rdr, qA[c].nr, pfa[c]);
      }
#pragma unroll
      for (int c = 0; c < CL; c++) {
        r0[c] = r1[c] = r2[c] = make_double2(0.0, 0.0);
        if (in_a[c]) {
          const double2 *rec = A.pair_d6 + 3 * ((size_t) qA[c].pair * nm1 + ma[c]); // same row m for both (pair_aeam.cpp:367)
          r0[c] = rec[0];
          r1[c] = rec[1];
          r2[c] = rec[2];
        }
      }
#pragma unroll
      for (int c = 0; c < CL; c++) {
        if (!(in_a[c] || in_j[c])) continue;
        double fpair_a = 0.0, fpair_j = 0.0, dfa = 0.0;
        if (in_a[c]) {
          const double pf = pfa[c];
          dfa = (r0[c].x * pf + r0[c].y) * pf + r1[c].x;
          const double phip = (r1[c].y * pf + r2[c].x) * pf + r2[c].y;
          fpair_a = -qa[c] * dfa * recip[c] + 0.5 * (-phip * recip[c]);
          if (EV) ea[c] += 0.5 * v4_val(A.z2r_v4[(size_t) qA[c].tz2r * nm1 + ma[c]], pf); // credited to i only
        }
        if (in_j[c]) {
          const double qj = xj.w;
          if (tj == ta[c] && in_a[c]) { // same element: both visits read the same table rows
            fpair_j = fpair_a + (qa[c] - qj) * dfa * recip[c];
          } else {
            double pf;
            const int m = spline_index(r[c], qJ[c].rdr, qJ[c].nr, pf);
            const double2 *rec = A.pair_d6 + 3 * ((size_t) qJ[c].pair * nm1 + m);
            const double2 j0 = rec[0], j1 = rec[1], j2 = rec[2];
            const double dfja = (j0.x * pf + j0.y) * pf + j1.x;
            const double phip = (j1.y * pf + j2.x) * pf + j2.y;
            fpair_j = -qj * dfja * recip[c] + 0.5 * (-phip * recip[c]);
          }
        }
        const double ft = fpair_a + fpair_j;
        fx[c] -= dx[c] * ft;
        fy[c] -= dy[c] * ft;
        fz[c] -= dz[c] * ft;
        if (EV) { // ev_tally(i = a, j, ..., fpair_a, d): every rank tallies its own visits
          v0 += dx[c] * dx[c] * fpair_a;
          v1 += dy[c] * dy[c] * fpair_a;
          v2 += dz[c] * dz[c] * fpair_a;
          v3 += dx[c] * dy[c] * fpair_a;
          v4 += dx[c] * dz[c] * fpair_a;
          v5 += dy[c] * dz[c] * fpair_a;
        }
      }
    }
  };
  segment(std::integral_constant<int, 0>{});
  segment(std::integral_constant<int, 1>{});
#pragma unroll
  for (int c = 0; c < CL; c++) {
    fx[c] = lane_sum<L>(fx[c]);
    fy[c] = lane_sum<L>(fy[c]);
    fz[c] = lane_sum<L>(fz[c]);
  }
  double ev_e = 0.0;
  if (EV) {
#pragma unroll
    for (int c = 0; c < CL; c++) {
      ev_e += ea[c];
      ea[c] = lane_sum<L>(ea[c]);
    }
  }
  if (s < CL) {
#pragma unroll
    for (int c = 0; c < CL; c++)
      if (c == s && real[c]) { // plain += : only writer of owned f here (stream order); the angular kernel follows
        double *fo = f + 3 * (size_t) (kc * CL + c);
        fo[0] += fx[c];
        fo[1] += fy[c];
        fo[2] += fz[c];
        if (EV && (eflag & MDP_EFLAG_ATOM)) eatom[kc * CL + c] += ea[c];
      }
  }
  if (EV) {
    double *slot = acc + MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)));
    if (eflag & MDP_EFLAG_GLOBAL) {
      ev_e = lane_sum<64>(ev_e);
      if (lane == 0) atomicAdd(&slot[0], ev_e);
    }
    if (vflag & MDP_VFLAG_GLOBAL) {
      v0 = lane_sum<64>(v0);
      v1 = lane_sum<64>(v1);
      v2 = lane_sum<64>(v2);
      v3 = lane_sum<64>(v3);
      v4 = lane_sum<64>(v4);
      v5 = lane_sum<64>(v5);
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
}

// ------------------------------------------------------------------------------------------------------
// Persistent density kernel: the spline table of the majority pair type lives in LDS.
//
// The tile kernels above fetch one spline row per in-range pair from global memory, every lane a different row of
// a 200-300 KB table: the vector L1 misses nearly always and its fill path (one 128-byte line per ~2.3 clocks
// per CU, profiles/ubench/tcp_gather.hip) is what bounds them.  A row's four value coefficients are the cubic
// Hermite expressions of the tabulated value Y and slope S at rows m and m+1 (pair_aeam.cpp:929-935) -- so 16 bytes
// per row hold everything, and the rows that are ever addressed (r from ~2.3 A to the cutoff) of ONE function fit
// the 160 KB LDS of a CU next to the staging buffers: rho(r) of the (0,0) pair, pass 1 of the metal centres.
// (The force pass needs two functions per pair, rho' and phi'; as two such passes it measured 1.4 ms against
// 0.65 ms for the gather kernel and is not part of the library -- DESIGN.md section 4 item 10.)
// One workgroup per CU stays resident: NSUB sub-blocks of 256 threads each walk their share of the workgroup's
// contiguous run of tiles.  A sub-block's next tile (union coordinates, rows, cluster heads) is requested into
// registers BEFORE the current tile is computed and written to LDS after it; the compute phase itself reads LDS
// only, so the loads in flight never sit in front of a wait.  Pairs of other type pairs (1.5 % in sample.in),
// and rows below the window, read the same (Y,S) rows from global memory.
// ------------------------------------------------------------------------------------------------------
struct PTile {
  int nlocal, nclus, cap, capL, rowcapB, per; // per = tiles per workgroup
  int t_begin, t_end;                         // the launch walks tiles [t_begin, t_end)
  int wlo, nw, lds_table;                     // LDS window: rows [wlo, wlo + nw) of table lds_table
  double hot_rsqmax;                          // largest r^2 inside the (0,0) pair's cutoff
  const double4 *xq;
  const int *tu, *tile_nu;
  const long long *lj_off;
  const int *lj_len; // lengths of pruned rows (at the offsets of the rows as built), or null
  const int *lj_split;
  const unsigned short *lj16;
  const double2 *ys; // [table][nrmax+1] (value, slope) rows of rho(r)
  double *rho;
};

struct YsRow {
  double2 a, b; // rows m and m+1
};
// value of the row's cubic (p = position inside the row)
__device__ __forceinline__ double ys_val(const YsRow w, const double p)
{
  const double d = w.b.x - w.a.x;
  const double c4 = 3.0 * d - 2.0 * w.a.y - w.b.y, c3 = w.a.y + w.b.y - 2.0 * d;
  return ((c3 * p + c4) * p + w.a.y) * p + w.a.x;
}

// explicit address spaces for the two sources of a table row: left generic, the compiler folds "LDS or global"
// into one flat load behind a pointer select, and a flat load waits for every load in flight
// (scalars: copying a vector type goes through its copy constructor, which takes a generic reference)
typedef __attribute__((address_space(3))) const double lds_double;
typedef __attribute__((address_space(1))) const double glb_double;

template <int NSUB, int CL>
__global__ __launch_bounds__(NSUB * 256) void aeam_ptile_kernel(const AeamDev A, const PTile P)
{
  constexpr int L = 16, SK = CL == 2 ? 3 : 2, RK = 2;
  constexpr int REC = 3; // doubles per union record: x y z
  extern __shared__ double2 s_dyn[];
  double2 *__restrict__ s_tab = s_dyn; // [nw]
  const int tid = threadIdx.x, sub = tid >> 8, t8 = tid & 255, lane = tid & 63, s = lane % L, gq = t8 / L;
  char *sb = reinterpret_cast<char *>(s_tab + P.nw) + (size_t) sub * ((size_t) P.capL * REC * 8 + P.rowcapB);
  double *__restrict__ s_rec = reinterpret_cast<double *>(sb);
  unsigned short *__restrict__ s_row = reinterpret_cast<unsigned short *>(sb + (size_t) P.capL * REC * 8);
  double2 *__restrict__ s_row2 = reinterpret_cast<double2 *>(s_row); // (16 bytes = 8 entries at a time)
  const int nm1 = A.nrmax + 1;
  {
    const double2 *__restrict__ src = P.ys + (size_t) P.lds_table * nm1 + P.wlo;
    for (int i = tid; i < P.nw; i += NSUB * 256) s_tab[i] = src[i];
  }
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int t_first = P.t_begin + wg * P.per;
  const int t_last = t_first + P.per < P.t_end ? t_first + P.per : P.t_end; // exclusive
  const int rounds = (P.per + NSUB - 1) / NSUB;

  // Both stages only REQUEST: nothing loaded is touched (no select, no difference) before the next stage or
  // the commit needs it -- a use right behind a load is a wait for everything in flight.
  // ---- stage 1 of the prefetch: tile header + union member indices (two tiles ahead) ------------------
  int m_t, m_idx[SK];
  bool m_valid;
  int2 m_nu;
  long long m_rb;
  int m_re; // (low word: only differences within a tile are taken.  A register of a load in flight that is dead
            //  -- the high word of an offset that is only ever truncated -- gets reused at once, behind a wait)
  const int *__restrict__ off_lo = reinterpret_cast<const int *>(P.lj_off);
  auto load_meta = [&](const int tt) {
    m_valid = tt < t_last;
    m_t = m_valid ? tt : P.t_end - 1;
    m_nu = reinterpret_cast<const int2 *>(P.tile_nu)[m_t];
    m_rb = P.lj_off[(size_t) m_t * kTile];
    m_re = off_lo[2 * ((size_t) m_t * kTile + kTile)];
    const int *__restrict__ mem = P.tu + (size_t) m_t * P.cap;
#pragma unroll
    for (int k = 0; k < SK; k++) m_idx[k] = mem[t8 + 256 * k];
  };
  // ---- stage 2: the union's coordinates, the tile's rows, the cluster heads (one tile ahead) ----------
  int n_t, n_nU, n_rtot, n_split;
  long long n_rb;
  int n_b, n_b1;
  double4 n_sv[SK], n_xa[CL];
  double n_rv[RK][2];
  auto gather = [&]() {
    n_t = m_t;
    n_nU = m_valid ? m_nu.x : 0;
    n_rtot = m_valid ? m_re - (int) m_rb : 0;
    n_rb = m_rb;
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int j = t8 + 256 * k < n_nU ? m_idx[k] : 0;
      n_sv[k] = P.xq[j];
    }
    const double2 *__restrict__ rsrc = reinterpret_cast<const double2 *>(P.lj16 + m_rb);
#pragma unroll
    for (int k = 0; k < RK; k++) {
      const double2 v = rsrc[(t8 + 256 * k) * 8 < n_rtot ? t8 + 256 * k : 0];
      n_rv[k][0] = v.x;
      n_rv[k][1] = v.y;
    }
    const int kc = m_t * kTile + gq;
    n_b = off_lo[2 * kc];
    n_b1 = P.lj_len ? P.lj_len[kc] : off_lo[2 * kc + 2]; // (pruned rows: the length itself)
    n_split = P.lj_split[kc];
#pragma unroll
    for (int c = 0; c < CL; c++) {
      const int ia = kc * CL + c < P.nlocal ? kc * CL + c : P.nlocal - 1;
      n_xa[c] = P.xq[ia];
    }
  };
  // registers -> LDS (between two barriers)
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < SK; k++) {
      const int u = t8 + 256 * k;
      if (u < n_nU) {
        s_rec[REC * u] = n_sv[k].x;
        s_rec[REC * u + 1] = n_sv[k].y;
        s_rec[REC * u + 2] = n_sv[k].z;
      }
    }
    if (n_nU > 256 * SK) { // a union beyond the register stage: fetched here (not seen in practice)
      const int *__restrict__ mem = P.tu + (size_t) n_t * P.cap;
      for (int u = t8 + 256 * SK; u < n_nU; u += 256) {
        const int j = mem[u];
        const double4 v = P.xq[j];
        s_rec[REC * u] = v.x;
        s_rec[REC * u + 1] = v.y;
        s_rec[REC * u + 2] = v.z;
      }
    }
    if (t8 == 0) { // the dummy member every padding entry points at: outside every cutoff
      s_rec[REC * n_nU] = 1.0e30;
      s_rec[REC * n_nU + 1] = 0.0;
      s_rec[REC * n_nU + 2] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < RK; k++) {
      const int e = t8 + 256 * k;
      if (e * 8 < n_rtot) s_row2[e] = make_double2(n_rv[k][0], n_rv[k][1]);
    }
    if (n_rtot > 256 * RK * 8) {
      const double2 *__restrict__ rsrc = reinterpret_cast<const double2 *>(P.lj16 + n_rb);
      for (int e = t8 + 256 * RK; e * 8 < n_rtot; e += 256) s_row2[e] = rsrc[e];
    }
  };

  load_meta(t_first + sub);
  gather();
  load_meta(t_first + NSUB + sub);
  __syncthreads(); // (the table)
  commit();
  __syncthreads();

  for (int rd = 0; rd < rounds; rd++) {
    // ---- the tile to compute now: heads out of the stage registers ----
    const int t = n_t, nU = n_nU;
    const int cnt = __builtin_amdgcn_readfirstlane(nU ? (P.lj_len ? n_b1 : n_b1 - n_b) : 0);
    const int split = __builtin_amdgcn_readfirstlane(nU ? n_split : 0);
    const int roff = n_b - (int) n_rb;
    const int kc = t * kTile + gq;
    double4 xa[CL];
    int ta[CL];
    bool metal[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) {
      xa[c] = n_xa[c];
      ta[c] = (int) xa[c].w;
      metal[c] = nU > 0 && kc < P.nclus && kc * CL + c < P.nlocal && ta[c] < A.nnonangular;
    }
    // ---- request the next tile, and the header of the one after ----
    // (vmcnt(0): nothing is in flight here -- the commit consumed it -- but only an explicit wait lets the compiler
    //  see that on every path; otherwise its counter-based waits for "older" loads land behind the new requests)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    gather();
    load_meta(t_first + (rd + 2) * NSUB + sub);

    // ---- compute from LDS ----
    double ac0[CL];
#pragma unroll
    for (int c = 0; c < CL; c++) ac0[c] = 0.0;
    const unsigned short *__restrict__ row = s_row + roff;
    // A table row comes from the LDS window (HOT pass) or from global memory (COLD pass: other type pairs, rows
    // below the window).  The passes are separate loops because a loop that may read global memory waits for
    // every load in flight -- the prefetch of the next tile -- in each trip; the cold loops run last and only in
    // waves that met a cold pair.
    auto lds_row = [&](const int m) {
      YsRow w;
      lds_double *src = (lds_double *) (s_tab + (m - P.wlo));
      w.a.x = src[0];
      w.a.y = src[1];
      w.b.x = src[2];
      w.b.y = src[3];
      return w;
    };
    auto glb_row = [&](const int table, const int m) {
      YsRow w;
      glb_double *src = (glb_double *) (P.ys + (size_t) table * nm1 + m);
      w.a.x = src[0];
      w.a.y = src[1];
      w.b.x = src[2];
      w.b.y = src[3];
      return w;
    };
    // HOT pass: pairs of the majority type pair (0,0) -- the LDS-resident table, and every parameter is a scalar
    // (CutDec never applies to a metal centre, pair_aeam.cpp:187-190)
    auto hot_pass = [&](bool &cold_seen) {
      const int kb = 0, ke = split;
      const double rdr = A.rdr[0];
      const int nr = A.nr[0];
      const double c2max = P.hot_rsqmax; // largest r^2 whose IEEE square root is <= cut (the reference tests r, pair_aeam.cpp:192)
      bool hot[CL];
#pragma unroll
      for (int c = 0; c < CL; c++) {
        hot[c] = metal[c] && ta[c] == 0;
        if (metal[c] && ta[c] != 0) cold_seen = true;
      }
      bool anyhot = false;
#pragma unroll
      for (int c = 0; c < CL; c++) anyhot = anyhot || hot[c];
      if (!__any(anyhot)) return; // (wave-uniform)
      int li_next = kb + s < ke ? (int) row[kb + s] : nU; // the next entry's index is read one trip ahead
      for (int k = kb + s; k < ke; k += L) {
        const double *__restrict__ pj = s_rec + REC * li_next;
        const double xj = pj[0], yj = pj[1], zj = pj[2];
        li_next = k + L < ke ? (int) row[k + L] : nU;
#pragma unroll
        for (int c = 0; c < CL; c++) {
          const double dx = xj - xa[c].x, dy = yj - xa[c].y, dz = zj - xa[c].z;
          const double rsq = dx * dx + dy * dy + dz * dz;
          if (!(hot[c] && rsq > 0.0 && rsq <= c2max)) continue; // (the union holds the cluster's own atoms too)
          const double recip = rsqrt_n1(rsq);
          const double r = rsq * recip;
          double pf;
          const int m = spline_index(r, rdr, nr, pf);
          if (m < P.wlo) { // below the window: the cold pass reads the row from global memory
            cold_seen = true;
            continue;
          }
          ac0[c] += ys_val(lds_row(m), pf);
        }
      }
    };
    // COLD pass: every other pair, rows from global memory, per-lane parameters
    auto cold_pass = [&](auto tjc) {
      constexpr int TJ = decltype(tjc)::value;
      const int kb = TJ ? split : 0, ke = TJ ? cnt : split;
      TilePar qA[CL];
      double c2max[CL];
#pragma unroll
      for (int c = 0; c < CL; c++) {
        qA[c] = tile_par<TJ, false>(A, ta[c]); // visit (i = a, j)
        c2max[c] = qA[c].cut * qA[c].cut * (1.0 + 1.0e-12);
      }
      int li_next = kb + s < ke ? (int) row[kb + s] : nU;
      for (int k = kb + s; k < ke; k += L) {
        const double *__restrict__ pj = s_rec + REC * li_next;
        const double xj = pj[0], yj = pj[1], zj = pj[2];
        li_next = k + L < ke ? (int) row[k + L] : nU;
#pragma unroll
        for (int c = 0; c < CL; c++) {
          const double dx = xj - xa[c].x, dy = yj - xa[c].y, dz = zj - xa[c].z;
          const double rsq = dx * dx + dy * dy + dz * dz;
          if (!(metal[c] && rsq > 0.0 && rsq <= c2max[c])) continue;
          const double recip = rsqrt_nr(rsq);
          const double r = rsq * recip;
          const bool hotc = TJ == 0 && ta[c] == 0; // a pair the hot pass owns unless its row lies below the window
          if (!(r <= qA[c].cut)) continue;
          double pf;
          const int m = spline_index(r, qA[c].rdr, qA[c].nr, pf);
          if (hotc && m >= P.wlo) continue;
          ac0[c] += ys_val(glb_row(qA[c].trho, m), pf);
        }
      }
    };
    {
      bool cold0 = false;
      hot_pass(cold0);
      if (__any(cold0)) cold_pass(std::integral_constant<int, 0>{});
      if (cnt > split) cold_pass(std::integral_constant<int, 1>{});
    }
    // ---- results of the tile ----
#pragma unroll
    for (int c = 0; c < CL; c++) ac0[c] = lane_sum<L>(ac0[c]);
    if (s < CL) {
#pragma unroll
      for (int c = 0; c < CL; c++)
        if (c == s && metal[c]) P.rho[kc * CL + c] = ac0[c];
    }
    // ---- the next tile's data goes to LDS ----
    __syncthreads();
    commit();
    __syncthreads();
  }
}

// ---- angular centres: one wave per centre, in-range neighbours staged in LDS (list order kept) ------
// slot record: dx dy dz r rsq f df fx fy fz   (d = x_j - x_i; the flag -- inside cut - CutDec, the range of
// pass 1 and of the k loop of pass 3)
constexpr int AREC = 10; // ... plus three force accumulators (pass 3)

template <bool PASS3>
__device__ __forceinline__ int ang_stage(const AeamDev &A, const double4 *__restrict__ xq, const double4 xi,
                                         const int ti, const long long b, const long long e,
                                         const int *__restrict__ nb, double *rec, int *jdx, int *flags)
{
  const int lane = threadIdx.x & 63;
  int n = 0;
  for (long long base = b; base < e; base += 64) { // wave-uniform trip count
    const long long k = base + lane;
    bool keep = false;
    int j = 0, inj1 = 0;
    double dx = 0, dy = 0, dz = 0, r = 0, rsq = 0, fv = 0, dfv = 0;
    if (k < e) {
      j = nb[k];
      const double4 xj = xq[j];
      dx = xj.x - xi.x;
      dy = xj.y - xi.y;
      dz = xj.z - xi.z;
      rsq = dx * dx + dy * dy + dz * dz;
      r = sqrt(rsq);
      const int tj = (int) xj.w;
      const int pt = ti * A.ntypes + tj; // (per-lane pair type: the parameter block in device memory)
      const double cdec = (tj >= A.nnonangular) ? kCutDec : 0.0; // i is angular here
      const double cut_pt = A.g_cut[pt];
      inj1 = r <= cut_pt - cdec;
      keep = PASS3 ? (r <= cut_pt) : (inj1 != 0); // pass 3's j loop has no CutDec (pair_aeam.cpp:350)
      if (keep) {
        double p;
        const double *c = spline_row(A.rhor, A.g_t2rhor[pt], A.nrmax + 1, r, A.g_rdr[pt], A.g_nr[pt], p);
        fv = sp_val(c, p);
        dfv = sp_der(c, p);
      }
    }
    const unsigned long long bal = __ballot(keep);
    const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
    if (keep && pos < ANG_CAP) {
      double *q = rec + pos * AREC;
      q[0] = dx;
      q[1] = dy;
      q[2] = dz;
      q[3] = r;
      q[4] = rsq;
      q[5] = fv;
      q[6] = dfv;
      q[7] = q[8] = q[9] = 0.0;
      jdx[pos] = j | (inj1 << 30);
    }
    n += __popcll(bal);
  }
  if (n > ANG_CAP) {
    if (lane == 0) atomicOr(&flags[0], 2);
    n = ANG_CAP;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return n;
}

// pass 1, angular centres (pair_aeam.cpp:207-250)
__global__ __launch_bounds__(256) void aeam_density_ang_kernel(const AeamDev A, const int nang,
                                                               const int *__restrict__ ang_list,
                                                               const double4 *__restrict__ xq,
                                                               const long long *__restrict__ nb_off,
                                                               const int *__restrict__ nb, double *__restrict__ rho,
                                                               int *__restrict__ flags)
{
  __shared__ double s_rec[4][ANG_CAP * AREC];
  __shared__ int s_j[4][ANG_CAP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + w;
  if (g >= nang) return; // whole wave
  const int i = ang_list[g];
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  double *rec = s_rec[w];
  const int n = ang_stage<false>(A, xq, xi, ti, nb_off[i], nb_off[i + 1], nb, rec, s_j[w], flags);
  double acc = 0.0;
  const double third = 1.0 / 3.0;
  // unordered pairs a<b, flattened over b: lane takes (a,b) with b = a+1+..  (n <= 160 -> <= 12720 pairs)
  for (int a = 0; a < n; a++) {
    const double *qa = rec + a * AREC;
    for (int bq = a + 1 + lane; bq < n; bq += 64) {
      const double *qb = rec + bq * AREC;
      const double ex = qb[0] - qa[0], ey = qb[1] - qa[1], ez = qb[2] - qa[2];
      const double rsq3 = ex * ex + ey * ey + ez * ez;
      const double cs = (qa[4] + qb[4] - rsq3) / (2 * qa[3] * qb[3]);
      const double delcs = cs + third;
      acc += 2 * qa[5] * qb[5] * (delcs * delcs);
    }
  }
  acc = lane_sum<64>(acc);
  if (lane == 0) rho[i] = acc;
}

// ---- pass 2 (pair_aeam.cpp:264-303, 329-332) -----------------------------------------------------------
// nimg > 0 (resident runs): threads nlocal .. nlocal + nimg serve the periodic self-images -- the style's forward_comm
// of fp on one rank (pair_aeam.cpp:307, 946-963) -- by evaluating their OWNER's embedding once more (a table row and
// thirty flops against a kernel launch of its own); an image of another rank's atom (owner < 0) waits for the exchange.
__global__ __launch_bounds__(256) void aeam_embed_kernel(const AeamDev A, const int nlocal,
                                                         const double4 *__restrict__ xq,
                                                         const double *__restrict__ rho, double *__restrict__ fp,
                                                         double *__restrict__ eatom, double *__restrict__ acc,
                                                         const int eflag, const int accumulate, const int nimg = 0,
                                                         const int *__restrict__ img_owner = nullptr)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  double e = 0.0;
  if (i >= nlocal && i < nlocal + nimg) {
    const int o = img_owner[i - nlocal];
    if (o >= 0) {
      const int ti = (int) xq[o].w;
      const bool metal = ti < A.nnonangular;
      const double rh = rho[o];
      const double u = metal ? rh : sqrt(rh);
      const int nrho_t = A.g_nrho[ti];
      double p = u * A.g_rdrho[ti] + 1.0;
      int m = (int) p;
      m = m < nrho_t - 1 ? m : nrho_t - 1;
      m = m > 1 ? m : 1;
      p -= m;
      p = p < 1.0 ? p : 1.0;
      const double *c = A.frho + ((size_t) A.g_t2frho[ti] * (A.nrhomax + 1) + m) * 7;
      double fptmp = 0.0;
      if (rh > 0.0000000000001) fptmp = metal ? 1.0 : 0.5 / sqrt(rh);
      fp[i] = fptmp * sp_der(c, p); // (the same operations as the owner's thread below: bit-identical)
    }
  }
  if (i < nlocal) {
    const int ti = (int) xq[i].w;
    const bool metal = ti < A.nnonangular;
    const double rh = rho[i];
    const double u = metal ? rh : sqrt(rh); // pow(rho, ni), ni = 1 or 1/2
    const int nrho_t = A.g_nrho[ti];
    double p = u * A.g_rdrho[ti] + 1.0;
    int m = (int) p;
    m = m < nrho_t - 1 ? m : nrho_t - 1;
    m = m > 1 ? m : 1;
    p -= m;
    p = p < 1.0 ? p : 1.0;
    const double *c = A.frho + ((size_t) A.g_t2frho[ti] * (A.nrhomax + 1) + m) * 7;
    const double fprime = sp_der(c, p);
    // Fptmp = ni rho^(ni-1) if rho > minrho else 0 (pair_aeam.cpp:329-332); the product is all
    // pass 3 ever uses (Feam :373, FFij/FFik/FFjk :450-452)
    double fptmp = 0.0;
    if (rh > 0.0000000000001) fptmp = metal ? 1.0 : 0.5 / sqrt(rh);
    fp[i] = fptmp * fprime;
    if (eflag) {
      const double F = sp_val(c, p);
      e = F;
      if (eflag & MDP_EFLAG_ATOM) {
        const double ea = metal ? F : F * (1.0 / 3.0); // pair_aeam.cpp:294-300 (quirk kept)
        if (accumulate)
          eatom[i] += ea;
        else
          eatom[i] = ea;
      }
    }
  }
  if (eflag & MDP_EFLAG_GLOBAL) {
    e = lane_sum<64>(e);
    if ((threadIdx.x & 63) == 0) atomicAdd(&acc[MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)))], e);
  }
}

// ---- pass 3, two-body part for every owned atom (pair_aeam.cpp:337-393) ----------------------------------
template <int L, int NT, bool EV>
__global__ __launch_bounds__(256) void aeam_force_kernel(const AeamDev A, const int nlocal,
                                                         const double4 *__restrict__ xq,
                                                         const long long *__restrict__ nb_off,
                                                         const int *__restrict__ nb, const double *__restrict__ fp,
                                                         double *__restrict__ f, double *__restrict__ eatom,
                                                         double *__restrict__ vatom, double *__restrict__ acc,
                                                         const int eflag, const int vflag)
{
  constexpr int U = 2;
  const int lane = threadIdx.x & 63;
  const int s = lane % L;
  const long long a64 = (long long) blockIdx.x * (256 / L) + threadIdx.x / L;
  const bool have = a64 < nlocal;
  const int a = have ? (int) a64 : 0;
  const double4 xa = xq[a];
  const int ta = (int) xa.w;
  const bool a_metal = ta < A.nnonangular;
  const double qa = a_metal ? fp[a] : 0.0; // (1 - deli) Fptmp fp
  double fx = 0, fy = 0, fz = 0, e = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0; // per-atom virial of atom a (vflag_atom)
  if (have) {
    const PairPar<NT> qA = load_pairpar<NT>(A, ta, false); // visit (i=a, j): tables of the pair (ta,tj)
    const PairPar<NT> qJ = load_pairpar<NT>(A, ta, true);  // visit (i=j, a): tables of the pair (tj,ta)
    const long long b = nb_off[a], en = nb_off[a + 1];
    const int nm1 = A.nrmax + 1;
    for (long long k0 = b; k0 < en; k0 += U * L) {
      int jj[U];
      double4 xj[U];
      double fpj[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long k = k0 + u * L + s;
        jj[u] = k < en ? nb[k] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        xj[u] = xq[jj[u] >= 0 ? jj[u] : a];
        fpj[u] = fp[jj[u] >= 0 ? jj[u] : a];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (jj[u] < 0) continue;
        const double dx = xj[u].x - xa.x, dy = xj[u].y - xa.y, dz = xj[u].z - xa.z;
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const int tj = (int) xj[u].w;
        const bool in_a = r <= qA.cut_(tj), in_j = r <= qJ.cut_(tj);
        if (!(in_a || in_j)) continue;
        const double recip = 1.0 / r;
        double fpair_a = 0.0, fpair_j = 0.0, dfa_shared = 0.0;
        if (in_a) { // the visit (i=a, j)
          double p;
          const int m = spline_index(r, qA.rdr_(tj), qA.nr_(tj), p);
          const double dfij = d4_der(A.rhor_d4[(size_t) qA.trho_(tj) * nm1 + m], p);
          dfa_shared = dfij;
          const size_t zrow = (size_t) qA.tz2r_(tj) * nm1 + m; // same row m1 (pair_aeam.cpp:367)
          const double phip = d4_der(A.z2r_d4[zrow], p);
          fpair_a = -qa * dfij * recip + 0.5 * (-phip * recip);
          if (EV) e += 0.5 * v4_val(A.z2r_v4[zrow], p); // credited to i only (pair_aeam.cpp:386-390)
        }
        if (in_j) { // the visit (i=j, neighbour a): only its action on a
          const double qj = (tj < A.nnonangular) ? fpj[u] : 0.0;
          if (tj == ta && in_a) {
            // same element: both visits read the same table rows (99 % of the pairs of an alloy matrix)
            fpair_j = fpair_a + (qa - qj) * dfa_shared * recip;
          } else {
            double p;
            const int m = spline_index(r, qJ.rdr_(tj), qJ.nr_(tj), p);
            const double dfja = d4_der(A.rhor_d4[(size_t) qJ.trho_(tj) * nm1 + m], p);
            const double phip = d4_der(A.z2r_d4[(size_t) qJ.tz2r_(tj) * nm1 + m], p);
            fpair_j = -qj * dfja * recip + 0.5 * (-phip * recip);
          }
        }
        const double ft = fpair_a + fpair_j;
        fx -= dx * ft;
        fy -= dy * ft;
        fz -= dz * ft;
        if (EV && vflag) { // ev_tally(i=a, j, ..., fpair_a, d): this rank tallies its own visits
          v0 += dx * dx * fpair_a;
          v1 += dy * dy * fpair_a;
          v2 += dz * dz * fpair_a;
          v3 += dx * dy * fpair_a;
          v4 += dx * dz * fpair_a;
          v5 += dy * dz * fpair_a;
          if (vflag & MDP_VFLAG_ATOM) { // each visit gives half its virial to either end: a collects both
            const double h = 0.5 * ft;
            a0 += dx * dx * h;
            a1 += dy * dy * h;
            a2 += dz * dz * h;
            a3 += dx * dy * h;
            a4 += dx * dz * h;
            a5 += dy * dz * h;
          }
        }
      }
    }
  }
  fx = lane_sum<L>(fx);
  fy = lane_sum<L>(fy);
  fz = lane_sum<L>(fz);
  if (have && s == 0) {
    // plain += : this kernel is the only writer of owned f at this point (stream order); the angular
    // kernel that follows uses atomics
    double *fo = f + 3 * (size_t) a;
    fo[0] += fx;
    fo[1] += fy;
    fo[2] += fz;
  }
  if (EV) {
    if (eflag & MDP_EFLAG_ATOM) {
      const double ea = lane_sum<L>(e);
      if (have && s == 0) eatom[a] += ea;
    }
    if (vflag & MDP_VFLAG_ATOM) {
      a0 = lane_sum<L>(a0);
      a1 = lane_sum<L>(a1);
      a2 = lane_sum<L>(a2);
      a3 = lane_sum<L>(a3);
      a4 = lane_sum<L>(a4);
      a5 = lane_sum<L>(a5);
      if (have && s == 0) { // plain stores: the angular kernel (atomics) runs after this one
        double *va = vatom + 6 * (size_t) a;
        va[0] += a0;
        va[1] += a1;
        va[2] += a2;
        va[3] += a3;
        va[4] += a4;
        va[5] += a5;
      }
    }
    double *slot = acc + MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)));
    if (eflag & MDP_EFLAG_GLOBAL) {
      const double et = lane_sum<64>(e);
      if (lane == 0) atomicAdd(&slot[0], et);
    }
    if (vflag & MDP_VFLAG_GLOBAL) {
      v0 = lane_sum<64>(v0);
      v1 = lane_sum<64>(v1);
      v2 = lane_sum<64>(v2);
      v3 = lane_sum<64>(v3);
      v4 = lane_sum<64>(v4);
      v5 = lane_sum<64>(v5);
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
}

// ---- pass 3, angular three-body part (pair_aeam.cpp:395-474) ----------------------------------------------
__global__ __launch_bounds__(256) void aeam_force_ang_kernel(const AeamDev A, const int nang,
                                                             const int *__restrict__ ang_list,
                                                             const double4 *__restrict__ xq,
                                                             const long long *__restrict__ nb_off,
                                                             const int *__restrict__ nb, const double *__restrict__ fp,
                                                             double *__restrict__ f, double *__restrict__ vatom,
                                                             double *__restrict__ acc, int *__restrict__ flags,
                                                             const int vflag, const int nlocal = 0,
                                                             const int *__restrict__ img_owner = nullptr)
{
  // img_owner (resident runs): what lands on a periodic self-image is added to its owner at once -- the host's
  // reverse_comm on one rank -- instead of being folded by a kernel of its own afterwards
  __shared__ double s_rec[4][ANG_CAP * AREC];
  __shared__ int s_j[4][ANG_CAP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + w;
  if (g >= nang) return;
  const int i = ang_list[g];
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  double *rec = s_rec[w];
  int *jdx = s_j[w];
  const int n = ang_stage<true>(A, xq, xi, ti, nb_off[i], nb_off[i + 1], nb, rec, jdx, flags);
  const double K = -fp[i]; // -Fptmp fp
  const double third = 1.0 / 3.0;
  double fix = 0, fiy = 0, fiz = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  // All (j,k) pairs -- j = slot a (any staged neighbour), k = a later slot b inside cut-CutDec -- are spread
  // over the 64 lanes (the reference's double loop gives the wave only n-a-1 <= ~17 busy lanes per j).  The
  // forces on j and k are summed per slot in LDS (native ds_add_f64) and flushed with one global atomic per
  // neighbour and component afterwards.
  const int npair = n * (n - 1) / 2;
  for (int p0 = 0; p0 < npair; p0 += 64) {
    const int p = p0 + lane;
    if (p >= npair) continue;
    // triangular decode: pairs with first index < a number a (2n - a - 1) / 2
    int a = (int) (0.5 * ((2 * n - 1) - sqrt((double) ((2 * n - 1) * (2 * n - 1) - 8 * p))));
    while (a > 0 && a * (2 * n - a - 1) / 2 > p) a--;
    while ((a + 1) * (2 * n - a - 2) / 2 <= p) a++;
    const int bq = p - a * (2 * n - a - 1) / 2 + a + 1;
    if (!((unsigned) jdx[bq] >> 30)) continue;
    double *qa = rec + a * AREC, *qb = rec + bq * AREC;
    const double r1 = qa[3], fij = qa[5], dfij = qa[6];
    const double r2 = qb[3], fik = qb[5], dfik = qb[6];
    const double ex = qb[0] - qa[0], ey = qb[1] - qa[1], ez = qb[2] - qa[2]; // x_k - x_j
    const double rsq3 = ex * ex + ey * ey + ez * ez;
    const double r3 = sqrt(rsq3);
    const double cs = (qa[4] + qb[4] - rsq3) / (2 * r1 * r2);
    const double dcosij = 1 / r2 - cs / r1;
    const double dcosik = 1 / r1 - cs / r2;
    const double dcosjk = -r3 / (r1 * r2);
    const double delcs = cs + third;
    const double ftet = delcs * delcs;
    const double delcs2 = 2 * delcs;
    const double DFij = 2.0 * (fik * dfij * ftet + fij * fik * delcs2 * dcosij); // ci = 2
    const double DFik = 2.0 * (fij * dfik * ftet + fij * fik * delcs2 * dcosik);
    const double DFjk = 2.0 * fij * fik * delcs2 * dcosjk;
    const double FFij = K * DFij / r1, FFik = K * DFik / r2, FFjk = K * DFjk / r3;
    const double gjx = qa[0] * FFij - ex * FFjk, gjy = qa[1] * FFij - ey * FFjk, gjz = qa[2] * FFij - ez * FFjk;
    const double gkx = qb[0] * FFik + ex * FFjk, gky = qb[1] * FFik + ey * FFjk, gkz = qb[2] * FFik + ez * FFjk;
    atomicAdd(&qa[7], gjx);
    atomicAdd(&qa[8], gjy);
    atomicAdd(&qa[9], gjz);
    atomicAdd(&qb[7], gkx);
    atomicAdd(&qb[8], gky);
    atomicAdd(&qb[9], gkz);
    fix -= gjx + gkx;
    fiy -= gjy + gky;
    fiz -= gjz + gkz;
    if (vflag) { // ev_tally3(i,j,k,0,0,fj,fk,drji,drki)
      const double t0 = qa[0] * gjx + qb[0] * gkx, t1 = qa[1] * gjy + qb[1] * gky, t2 = qa[2] * gjz + qb[2] * gkz;
      const double t3 = qa[0] * gjy + qb[0] * gky, t4 = qa[0] * gjz + qb[0] * gkz, t5 = qa[1] * gjz + qb[1] * gkz;
      v0 += t0;
      v1 += t1;
      v2 += t2;
      v3 += t3;
      v4 += t4;
      v5 += t5;
      if (vflag & MDP_VFLAG_ATOM) { // a third each to i, j, k (angular centres are rare: atomics)
        const double tt[6] = {t0 * third, t1 * third, t2 * third, t3 * third, t4 * third, t5 * third};
        const int jj = jdx[a] & MDP_NEIGHMASK, kk = jdx[bq] & MDP_NEIGHMASK;
#pragma unroll
        for (int q6 = 0; q6 < 6; q6++) {
          atomicAdd(&vatom[6 * (size_t) i + q6], tt[q6]);
          atomicAdd(&vatom[6 * (size_t) jj + q6], tt[q6]);
          atomicAdd(&vatom[6 * (size_t) kk + q6], tt[q6]);
        }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int a = lane; a < n; a += 64) { // one flush per neighbour (ghosts included: the host's reverse comm folds them)
    const double *qa = rec + a * AREC;
    if (qa[7] != 0.0 || qa[8] != 0.0 || qa[9] != 0.0) {
      int jj = jdx[a] & MDP_NEIGHMASK;
      if (img_owner && jj >= nlocal) {
        const int o = img_owner[jj - nlocal];
        jj = o >= 0 ? o : jj;
      }
      atomicAdd(&f[3 * (size_t) jj], qa[7]);
      atomicAdd(&f[3 * (size_t) jj + 1], qa[8]);
      atomicAdd(&f[3 * (size_t) jj + 2], qa[9]);
    }
  }
  fix = lane_sum<64>(fix);
  fiy = lane_sum<64>(fiy);
  fiz = lane_sum<64>(fiz);
  if (lane == 0) {
    atomicAdd(&f[3 * (size_t) i], fix);
    atomicAdd(&f[3 * (size_t) i + 1], fiy);
    atomicAdd(&f[3 * (size_t) i + 2], fiz);
  }
  if (vflag & MDP_VFLAG_GLOBAL) {
    v0 = lane_sum<64>(v0);
    v1 = lane_sum<64>(v1);
    v2 = lane_sum<64>(v2);
    v3 = lane_sum<64>(v3);
    v4 = lane_sum<64>(v4);
    v5 = lane_sum<64>(v5);
    if (lane == 0) {
      double *slot = acc + MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)));
      atomicAdd(&slot[1], v0);
      atomicAdd(&slot[2], v1);
      atomicAdd(&slot[3], v2);
      atomicAdd(&slot[4], v3);
      atomicAdd(&slot[5], v4);
      atomicAdd(&slot[6], v5);
    }
  }
}

// re-lay the 7-coefficient rows into the aligned records the streaming kernels gather
__global__ void relay_kernel(const size_t nrows, const double *__restrict__ src, double4 *__restrict__ val4,
                             double4 *__restrict__ der4)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= nrows) return;
  const double *c = src + 7 * i;
  val4[i] = make_double4(c[3], c[4], c[5], c[6]);
  der4[i] = make_double4(c[0], c[1], c[2], 0.0);
}

// (value, slope) of every row: all the persistent tile kernels need of a table (16 bytes per row)
__global__ void ys_kernel(const size_t nrows, const double *__restrict__ src, double2 *__restrict__ ys)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= nrows) return;
  ys[i] = make_double2(src[7 * i + 6], src[7 * i + 5]);
}

// derivative coefficients of rho (c0..c2) and of phi (c0..c2) of ONE pair type side by side in a 48-byte
// record: the pair-force visit needs both at the same row.  The tile kernels are bound by the vector L1's
// lookup rate (~0.9 lookups per clock per CU measured, every lane a different row): a lookup moves at most 16
// bytes per lane, so six doubles are three lookups -- a padded 64-byte record was four
__global__ void pair_der_kernel(const int npair, const int nm1, const int *__restrict__ t2rhor,
                                const int *__restrict__ t2z2r, const double *__restrict__ rhor,
                                const double *__restrict__ z2r, double *__restrict__ out /* [npair][nm1][6] */)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t) npair * nm1) return;
  const int pt = (int) (i / nm1), m = (int) (i % nm1);
  const double *a = rhor + ((size_t) t2rhor[pt] * nm1 + m) * 7, *b = z2r + ((size_t) t2z2r[pt] * nm1 + m) * 7;
  double *o = out + 6 * i;
  o[0] = a[0];
  o[1] = a[1];
  o[2] = a[2];
  o[3] = b[0];
  o[4] = b[1];
  o[5] = b[2];
}

// owned angular centres; count[0] = their number, count[2] = 1 when one of them sits in a tile at or behind
// count[1] = the first tile whose union reaches a remote ghost (its row may then hold one: ghost forces must travel)
// the list is there already (mdp_md_build_master_list selected the same atoms for their CSR rows): only count[2]
__global__ void ang_reach_kernel(const int n, const int *__restrict__ list, int *__restrict__ count,
                                 const int atoms_per_tile)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  const bool reach = k < n && list[k] / atoms_per_tile >= count[1];
  if (__any(reach) && (threadIdx.x & 63) == 0) count[2] = 1;
}

__global__ void ang_list_kernel(const int nnonangular, int nlocal, const double4 *__restrict__ xq, int *__restrict__ list,
                                int *__restrict__ count, const int atoms_per_tile)
{
  // one atomic per wave, not per centre: 7 500 single-lane atomics on one counter took 0.2 ms per reneighboring
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool ang = i < nlocal && (int) xq[i < nlocal ? i : 0].w >= nnonangular;
  const unsigned long long b = __ballot(ang);
  if (!b) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long) b) - 1) base = atomicAdd(count, __popcll(b));
  base = __shfl(base, __ffsll((long long) b) - 1, 64);
  if (ang) list[base + __popcll(b & ((1ull << lane) - 1ull))] = i;
  // (thousands of plain stores to ONE word are as slow as atomics on it -- 0.2 ms per reneighboring: one store per wave,
  //  and none in a run without remote ghosts, where nobody reads the word)
  if (atoms_per_tile > 0) {
    const bool reach = ang && i / atoms_per_tile >= count[1];
    if (__any(reach) && lane == __ffsll((long long) b) - 1) count[2] = 1;
  }
}

// count[1] = first tile whose union holds a remote ghost (index >= remote_start); tiles before it need no halo.
// count[3], count[4] = largest union and most row entries among the tiles that reach none: the shell atoms of a brick
// lie on a thin slab, 32 consecutive ones spread wider than 32 interior atoms and their unions are the largest of the
// build -- the launches over the interior tiles size their LDS staging by their own maxima.
__global__ __launch_bounds__(256) void tile_first_remote_kernel(const int ntile, const int cap, const int remote_start,
                                                                const int *__restrict__ tile_nu,
                                                                const int *__restrict__ tu,
                                                                const long long *__restrict__ lj_off,
                                                                int *__restrict__ count)
{
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntile) return;
  const int *mem = tu + (size_t) t * cap;
  const int nU = tile_nu[2 * t];
  int hit = 0;
  for (int u = lane; u < nU; u += 64) hit |= mem[u] >= remote_start;
  hit = __any(hit); // (all 64 lanes vote: a tile that straddles the end of the interior atoms holds few remote ghosts)
  if (lane != 0) return;
  if (hit) {
    atomicMin(&count[1], t);
  } else {
    // (thousands of atomics on one word serialise: only a tile that would raise the maximum sends one)
    const int rows = (int) (lj_off[(size_t) (t + 1) * kTile] - lj_off[(size_t) t * kTile]);
    if (nU > __hip_atomic_load(&count[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&count[3], nU);
    if (rows > __hip_atomic_load(&count[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&count[4], rows);
  }
}

inline int nblk(long long n, int per) { return (int) ((n + per - 1) / per); }

} // namespace

// largest cutoff between the two classes of the tile lists (type 0 | every other type), either visit of a pair
static void aeam_class_cuts(const mdp_ctx *c, double out[4])
{
  const int nt = c->aeam.ntypes;
  for (int k = 0; k < 4; k++) out[k] = 0.0;
  for (int ti = 0; ti < nt; ti++)
    for (int tj = 0; tj < nt; tj++) {
      const double a = c->aeam_hcut[ti * nt + tj], b = c->aeam_hcut[tj * nt + ti];
      const double m = a > b ? a : b;
      const int k = (ti > 1 ? 1 : ti) * 2 + (tj > 1 ? 1 : tj);
      if (m > out[k]) out[k] = m;
    }
}

int mdp_aeam_prepare(mdp_ctx *c)
{
  if (!c->have_aeam) return mdp_fail(c, MDP_ESTATE, "aeam tables not set");
  if (!c->atoms_set || !c->neigh_set) return mdp_fail(c, MDP_ESTATE, "atoms / neighbor list not set");
  hipStream_t st = c->stream;
  MDP_HIP(c, c->rho.reserve(c->nall + 1));
  MDP_HIP(c, c->fp.reserve(c->nall + 1));
  MDP_HIP(c, c->ang_list.reserve(c->nlocal + 1));
  MDP_HIP(c, c->ang_count.reserve(8));
  // resident mode: tile lists next to the CSR list (which the angular kernels and the steps
  // that tally energy / virial keep using).  The bin grid of the list build just done is still current.
  c->aeam_tiled = false;
  c->aeam_split = 0;
  c->aeam_ang_remote = false;
  c->aeam_phase = 0;
  c->f_prezeroed = c->f_zero_remote_due = false; // (the atom arrays were rebuilt or re-ordered)
  const char *e = getenv("MDP_AEAM_TILE");
  // (tile kernels: up to MDP_AEAM_MAXT types -- their parameter block lives in the kernel arguments / in LDS)
  if ((c->md || c->aeam_device_lists) && c->nlocal > 0 && !(e && atoi(e) == 0) && c->aeam.ntypes <= MDP_AEAM_MAXT) {
    double cutsq[4];
    aeam_class_cuts(c, cutsq); // either visit of the pair may need it
    for (int k = 0; k < 4; k++) cutsq[k] = (cutsq[k] + c->cfg.skin) * (cutsq[k] + c->cfg.skin);
    const char *ecl = getenv("MDP_AEAM_CLUSTER");
    // two atoms per 16-lane group: the LDS read of a neighbour and its row index serve both, and the union staging
    // is amortised over 32 atoms (with unions sorted by atom index this beats one atom per group by 11 % at 863 K)
    c->aeam_cl = ecl && atoi(ecl) == 1 ? 1 : 2;
    bool ok = false;
    MDP_TRY(mdp_tile_lists_build(c, cutsq, c->aeam_cl, &ok));
    c->aeam_tiled = ok;
  }
  {
    // angular centres; with remote ghosts (multi-GPU): the first tile that reaches one -- the tiles before it run
    // while the halo is in flight (domain.hip stores the shell atoms of the brick behind the interior ones)
    const bool remote = c->md && c->remote_start < c->nall;
    const bool have_list = c->ang_list_n >= 0; // (selected for the CSR rows of this very build: the same atoms)
    const int init[5] = {have_list ? c->ang_list_n : 0, remote && c->aeam_tiled ? c->ntile : 0, 0, 0, 0};
    c->ang_list_n = -1;
    MDP_TRY(mdp_write_small(c, c->ang_count.p, init, sizeof init));
    if (remote && c->aeam_tiled)
      tile_first_remote_kernel<<<nblk(c->ntile, 4), 256, 0, st>>>(c->ntile, c->tile_cap, c->remote_start, c->tile_nu.p,
                                                                  c->tu.p, c->lj_off.p, c->ang_count.p);
    if (have_list) {
      if (init[0] && remote && c->aeam_tiled)
        ang_reach_kernel<<<nblk(init[0], 256), 256, 0, st>>>(init[0], c->ang_list.p, c->ang_count.p, kTile * c->aeam_cl);
    } else if (c->nlocal)
      ang_list_kernel<<<nblk(c->nlocal, 256), 256, 0, st>>>(c->aeam.nnonangular, c->nlocal, c->xq.p, c->ang_list.p, c->ang_count.p,
                                                            remote && c->aeam_tiled ? kTile * c->aeam_cl : 0);
    MDP_HIP(c, hipGetLastError());
    int h[5] = {0, 0, 0, 0, 0};
    MDP_TRY(mdp_read_one(c, c->ang_count.p, sizeof h, h));
    c->h_ang_count = h[0];
    c->aeam_split = remote && c->aeam_tiled ? h[1] : 0;
    c->tile_maxu_in = c->aeam_split > 0 ? h[3] : 0;
    c->tile_rowmax_in = c->aeam_split > 0 ? h[4] : 0;
    c->aeam_ang_remote = remote && h[0] > 0 && (c->aeam_tiled ? h[2] != 0 : true);
    if (getenv("MDP_DEBUG"))
      fprintf(stderr, "[mdp] aeam: %d angular centres; tiles [0, %d) of %d reach no remote ghost (largest union %d of %d, rows %d of %d); ghost forces %s\n",
              h[0], c->aeam_split, c->ntile, c->tile_maxu_in, c->tile_maxu, c->tile_rowmax_in, c->tile_rowmax,
              c->aeam_ang_remote ? "travel" : "stay");
  }
  if (!c->aeam_tiled && !c->csr_full) { // no tile lists after all (a union outgrew LDS): the CSR kernels need every row
    c->csr_want_full = true;
    MDP_TRY(mdp_md_build_master_list(c));
  }
  return MDP_OK;
}

// largest union / most row entries of the tiles [.., t_end): the interior tiles' own maxima when the range lies before
// the first tile that reaches a remote ghost (tile_first_remote_kernel)
static inline int aeam_range_maxu(const mdp_ctx *c, const int t_end)
{
  return c->aeam_split > 0 && t_end <= c->aeam_split && c->tile_maxu_in > 0 ? c->tile_maxu_in : c->tile_maxu;
}
static inline int aeam_range_rowmax(const mdp_ctx *c, const int t_end)
{
  return c->aeam_split > 0 && t_end <= c->aeam_split && c->tile_rowmax_in > 0 ? c->tile_rowmax_in : c->tile_rowmax;
}

// Persistent density kernel (above): geometry of the launch over the tiles [t_begin, t_end).  *done stays false when
// the window that fits LDS next to the staging buffers would cover too little of the table -- the caller then runs
// the gather kernel.
static int aeam_ptile_launch(mdp_ctx *c, const int t_begin, const int t_end, bool *done)
{
  *done = false;
  const char *epers = getenv("MDP_AEAM_PERSIST"); // unset: decided here; 0: never; 1: whenever a window fits
  if (epers && atoi(epers) == 0) return MDP_OK;
  if (c->tile_rowmax <= 0) return MDP_OK;
  if (c->aeam.ntypes != 2) return MDP_OK; // (its cold passes select between the parameters of two types)
  // the LDS-resident table is the (0,0) pair's: with many atoms of the other type most pairs would take the cold
  // pass (rows from global memory, behind the hot pass) and the gather kernels are the better choice
  if (!epers && (double) c->h_ang_count > 0.1 * (double) c->nlocal) return MDP_OK;
  if (!c->lds_max) {
    int v = 0, n = 0;
    MDP_HIP(c, hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device));
    MDP_HIP(c, hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device));
    c->lds_max = v;
    c->num_cu = n > 0 ? n : 256;
  }
  const AeamDev &A = c->aeam;
  const int capL = (aeam_range_maxu(c, t_end) + 1 + 7) & ~7;
  const int rowcapB = (2 * aeam_range_rowmax(c, t_end) + 15) & ~15;
  const size_t sub_bytes = (size_t) capL * 3 * 8 + rowcapB;
  const int nr = A.nr[0]; // rows 1..nr of the (0,0) pair's table are addressed: m in [1, nr-1] and m+1
  const char *ens = getenv("MDP_AEAM_PT_NSUB");
  const int force_nsub = ens ? atoi(ens) : 0;
  int nsub = 0, nw = 0;
  for (int ns = 4; ns >= 2; ns--) {
    if (force_nsub && ns != force_nsub) continue;
    const long long room = (long long) c->lds_max - (long long) ns * (long long) sub_bytes - 64;
    if (room <= 0) continue;
    int w = (int) (room / 16);
    if (w > nr) w = nr;
    // the window has to reach down to ~0.36 of the cutoff (2.3 A of 6.5 A: below the first neighbour shell of a hot
    // crystal), or -- with fewer waves -- at least to half of it
    if (w >= (ns == 2 ? 0.5 : 0.64) * nr || force_nsub) {
      nsub = ns;
      nw = w;
      break;
    }
  }
  if (!nsub || nw < 2) return MDP_OK;
  *done = true;
  const int nt = t_end - t_begin;
  if (nt <= 0) return MDP_OK;
  PTile P = {};
  P.nlocal = c->nlocal;
  P.nclus = c->nclus;
  P.cap = c->tile_cap;
  P.capL = capL;
  P.rowcapB = rowcapB;
  P.t_begin = t_begin;
  P.t_end = t_end;
  int grid = (nt + nsub - 1) / nsub;
  if (grid > c->num_cu) grid = c->num_cu;
  P.per = (nt + grid - 1) / grid;
  grid = (nt + P.per - 1) / P.per;
  P.nw = nw;
  P.wlo = nr + 1 - nw;
  P.lds_table = A.t2rhor[0];
  P.ys = A.rhor_ys;
  P.xq = c->xq.p;
  P.tu = c->tu.p;
  P.tile_nu = c->tile_nu.p;
  P.lj_off = c->lj_off.p;
  P.lj_len = c->prune_valid ? c->lj_len_in.p : nullptr;
  P.lj_split = c->prune_valid ? c->lj_split_in.p : c->lj_split.p;
  P.lj16 = c->prune_valid ? c->lj16_in.p : c->lj16.p;
  P.rho = c->rho.p;
  {
    // the reference skips a pair when sqrt(rsq) > cut: the largest rsq that passes, found once on the host
    const double cut = A.cut[0];
    double t = cut * cut;
    while (sqrt(t) > cut) t = nextafter(t, 0.0);
    while (sqrt(nextafter(t, INFINITY)) <= cut) t = nextafter(t, INFINITY);
    P.hot_rsqmax = t;
  }
  const size_t lds = (size_t) nw * 16 + (size_t) nsub * sub_bytes;
#define MDP_PT(NS)                                                                                                    \
  do {                                                                                                                \
    if (c->aeam_cl == 2) {                                                                                            \
      MDP_HIP(c, hipFuncSetAttribute((const void *) aeam_ptile_kernel<NS, 2>,                                         \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                         \
      aeam_ptile_kernel<NS, 2><<<grid, NS * 256, lds, c->stream>>>(c->aeam, P);                                       \
    } else {                                                                                                          \
      MDP_HIP(c, hipFuncSetAttribute((const void *) aeam_ptile_kernel<NS, 1>,                                         \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                         \
      aeam_ptile_kernel<NS, 1><<<grid, NS * 256, lds, c->stream>>>(c->aeam, P);                                       \
    }                                                                                                                 \
  } while (0)
  if (nsub == 4) MDP_PT(4);
  else if (nsub == 3) MDP_PT(3);
  else MDP_PT(2);
#undef MDP_PT
  MDP_HIP(c, hipGetLastError());
  if (getenv("MDP_DEBUG") && !c->ptile_reported[0]) {
    c->ptile_reported[0] = true;
    fprintf(stderr, "[mdp] aeam persistent density kernel: %d workgroups x %d sub-blocks, %d tiles each, window rows [%d, %d] of %d (%.1f KB), staging %zu B per sub-block\n",
            grid, nsub, P.per, P.wlo, nr, nr, nw * 16 / 1024.0, sub_bytes);
  }
  return MDP_OK;
}

// ---- one compute() in phases ---------------------------------------------------------------------------------------
// A step of a multi-GPU run hides its exchanges behind the tiles that reach no remote ghost (tiles [0, aeam_split):
// the shell atoms of a brick are stored behind the interior ones, csrc/domain.hip):
//   A  mdp_aeam_run_begin        density of the interior tiles                     | position exchange in flight
//   B  mdp_aeam_run_density      the other density tiles, angular centres, embedding (pair_aeam.cpp:158-303), and --
//                                when A ran -- the three-body forces (they need the centre's own F' only,
//                                pair_aeam.cpp:395-470), so that the forces on ghosts can travel early
//   C  mdp_aeam_run_force_begin  force tiles of the interior                       | fp forward + force reverse in flight
//   D  mdp_aeam_run_force        the other force tiles (+ three-body forces if not done), accumulators folded
// A and C are optional: without them B and D do everything (one GPU, host mode, per-atom-virial steps, CSR lists).
enum { AE_OPEN = 1, AE_DENS_INT = 2, AE_ANG_F = 4, AE_FORCE_INT = 8, AE_DENS = 16 };

// force_clear + accumulators, once per compute
static int aeam_open(mdp_ctx *c)
{
  if (c->aeam_phase & AE_OPEN) return MDP_OK;
  if (!c->have_aeam || !c->neigh_set) return mdp_fail(c, MDP_ESTATE, "aeam: tables / neighbor list not set");
  MDP_TRY(mdp_acc_begin(c, true));
  if (!c->f_prezeroed) // (resident runs on one GPU: the integrate kernel and the image refresh of this step did it)
    MDP_HIP(c, hipMemsetAsync(c->f.p, 0, sizeof(double) * 3 * c->nall, c->stream));
  c->f_prezeroed = false;
  c->aeam_phase |= AE_OPEN;
  return MDP_OK;
}

// pass 1 of the metal centres over the tiles [t_begin, t_end)
static int aeam_density_tiles(mdp_ctx *c, const int t_begin, const int t_end)
{
  hipStream_t st = c->stream;
  const int nlocal = c->nlocal;
  if (t_end <= t_begin) return MDP_OK;
  bool persistent = false;
  MDP_TRY(aeam_ptile_launch(c, t_begin, t_end, &persistent));
  if (persistent) return MDP_OK;
  const int capL = (aeam_range_maxu(c, t_end) + 1 + 7) & ~7;
  const bool multi = c->aeam.ntypes != 2; // per-entry types and an LDS parameter block (see par_fill)
  const size_t lds = (size_t) capL * 3 * sizeof(double) + (multi ? kParBytes + (size_t) capL * sizeof(int) : 0);
#define MDP_ATD(CLV, MV)                                                                                             \
  do {                                                                                                                \
    if (lds > 48 * 1024)                                                                                              \
      MDP_HIP(c, hipFuncSetAttribute((const void *) aeam_tile_density_kernel<CLV, MV>,                                \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                         \
    aeam_tile_density_kernel<CLV, MV><<<t_end - t_begin, 256, lds, st>>>(                                             \
        c->aeam, nlocal, c->nclus, t_begin, c->xq.p, c->tile_cap, capL, c->tu.p, c->tile_nu.p, c->lj_off.p,           \
        c->prune_valid ? c->lj_len_in.p : nullptr, c->prune_valid ? c->lj_split_in.p : c->lj_split.p,                 \
        c->prune_valid ? c->lj16_in.p : c->lj16.p, c->rho.p);                                                         \
  } while (0)
  if (multi) {
    if (c->aeam_cl == 1) MDP_ATD(1, true);
    else MDP_ATD(2, true);
  } else {
    if (c->aeam_cl == 1) MDP_ATD(1, false);
    else MDP_ATD(2, false);
  }
#undef MDP_ATD
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// pass 3, pair part, over the tiles [t_begin, t_end)
static int aeam_force_tiles(mdp_ctx *c, const int t_begin, const int t_end, const int eflag, const int vflag)
{
  hipStream_t st = c->stream;
  if (t_end <= t_begin) return MDP_OK;
  const int capL = (aeam_range_maxu(c, t_end) + 1 + 7) & ~7;
  const bool multi = c->aeam.ntypes != 2;
  const size_t lds = (size_t) capL * 4 * sizeof(double) + (multi ? kParBytes + (size_t) capL * sizeof(int) : 0);
  const bool ev = eflag || vflag;
#define MDP_ATF(CLV, EVV, MV)                                                                                        \
  do {                                                                                                                \
    if (lds > 48 * 1024)                                                                                              \
      MDP_HIP(c, hipFuncSetAttribute((const void *) aeam_tile_force_kernel<CLV, EVV, MV>,                             \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));                         \
    aeam_tile_force_kernel<CLV, EVV, MV><<<t_end - t_begin, 256, lds, st>>>(                                          \
        c->aeam, c->nlocal, c->nclus, t_begin, c->xq.p, c->fp.p, c->tile_cap, capL, c->tu.p, c->tile_nu.p,            \
        c->lj_off.p, c->prune_valid ? c->lj_len_in.p : nullptr, c->prune_valid ? c->lj_split_in.p : c->lj_split.p,    \
        c->prune_valid ? c->lj16_in.p : c->lj16.p, c->f.p, c->eatom.p, c->acc.p, eflag, vflag);                       \
  } while (0)
#define MDP_ATF_M(CLV, EVV)                                                                                          \
  do {                                                                                                                \
    if (multi) MDP_ATF(CLV, EVV, true);                                                                               \
    else MDP_ATF(CLV, EVV, false);                                                                                    \
  } while (0)
  if (c->aeam_cl == 1) {
    if (ev) MDP_ATF_M(1, true);
    else MDP_ATF_M(1, false);
  } else {
    if (ev) MDP_ATF_M(2, true);
    else MDP_ATF_M(2, false);
  }
#undef MDP_ATF_M
#undef MDP_ATF
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

static int aeam_ang_forces(mdp_ctx *c, const int vflag)
{
  if (c->h_ang_count)
    aeam_force_ang_kernel<<<nblk(c->h_ang_count, 4), 256, 0, c->stream>>>(c->aeam, c->h_ang_count, c->ang_list.p,
                                                                          c->xq.p, c->nb_off.p, c->nb.p, c->fp.p,
                                                                          c->f.p, c->vatom.p, c->acc.p, c->flags.p,
                                                                          vflag, c->nlocal,
                                                                          c->md ? c->ghost_owner.p : nullptr);
  MDP_HIP(c, hipGetLastError());
  c->aeam_phase |= AE_ANG_F;
  return MDP_OK;
}

static void aeam_prune_cuts(const mdp_ctx *c, double cut[4]) { aeam_class_cuts(c, cut); } // either visit may need the entry

// phase A.  Runs while this step's position exchange is in flight: the remote ghosts still hold the previous step's
// positions, which the interior tiles never read.  A step that has to (re-)prune the rows -- the pruning reads every
// atom's position -- leaves everything to phase B.
int mdp_aeam_run_begin(mdp_ctx *c, int eflag, int vflag)
{
  c->aeam_phase = 0;
  c->aeam_vflag = vflag;
  if (!c->aeam_tiled || c->aeam_split <= 0 || (vflag & MDP_VFLAG_ATOM) || !c->nlocal) return MDP_OK;
  if (const char *e = getenv("MDP_AEAM_OVERLAP"))
    if (atoi(e) == 0) return MDP_OK;
  if (c->overlap_mode == 2) return MDP_OK; // blocking order chosen for this step (MdpDomain::ov_policy)
  double cut[4];
  aeam_prune_cuts(c, cut);
  bool due = false;
  MDP_TRY(mdp_prune_upkeep(c, cut, c->cfg.skin, /*may_prune=*/false, &due));
  if (due) return MDP_OK;
  MDP_TRY(aeam_open(c));
  mdp_span_begin(c, 5);
  MDP_TRY(aeam_density_tiles(c, 0, c->aeam_split));
  mdp_span_end(c, 5);
  c->aeam_phase |= AE_DENS_INT;
  return MDP_OK;
}

// phases B: passes 1 + 2.  Leaves rho[], fp[] (= Fptmp*F') for owned atoms; embedding energy in the accumulators.
int mdp_aeam_run_density(mdp_ctx *c, int eflag)
{
  if (!c->have_aeam || !c->neigh_set) return mdp_fail(c, MDP_ESTATE, "aeam: tables / neighbor list not set");
  if (c->aeam_phase & AE_DENS) c->aeam_phase = 0; // (a compute that was never closed by the force pass)
  hipStream_t st = c->stream;
  const int nlocal = c->nlocal;
  const bool early = (c->aeam_phase & AE_DENS_INT) != 0;
  MDP_TRY(aeam_open(c));
  if (!early) {
    if (c->aeam_tiled) {
      // resident runs walk rows pruned to the pairs within reach right now (tile_prune_kernel in rebomos.hip): a third
      // of the entries of a list built with 1 A of skin on a 6.5 A cutoff are skin
      double cut[4];
      aeam_prune_cuts(c, cut);
      MDP_TRY(mdp_prune_upkeep(c, cut, c->cfg.skin, /*may_prune=*/true, nullptr));
    } else
      c->prune_valid = false;
  }
  mdp_span_begin(c, 0);
  if (nlocal && c->aeam_tiled) {
    MDP_TRY(aeam_density_tiles(c, early ? c->aeam_split : 0, c->ntile));
  } else if (nlocal) {
    const int grid = nblk(nlocal, 256 / AE_L);
    switch (c->aeam.ntypes) {
      case 1: aeam_density_kernel<AE_L, 1><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->rho.p); break;
      case 2: aeam_density_kernel<AE_L, 2><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->rho.p); break;
      case 3: aeam_density_kernel<AE_L, 3><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->rho.p); break;
      case 4: aeam_density_kernel<AE_L, 4><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->rho.p); break;
      case 5: case 6: case 7: case 8:
        aeam_density_kernel<AE_L, MDP_AEAM_MAXT><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->rho.p); break;
      default: aeam_density_kernel<AE_L, 0><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->rho.p); break;
    }
  }
  mdp_span_end(c, 0);
  mdp_span_begin(c, 1);
  if (c->h_ang_count)
    aeam_density_ang_kernel<<<nblk(c->h_ang_count, 4), 256, 0, st>>>(c->aeam, c->h_ang_count, c->ang_list.p, c->xq.p,
                                                                     c->nb_off.p, c->nb.p, c->rho.p, c->flags.p);
  MDP_HIP(c, hipGetLastError());
  mdp_span_end(c, 1);
  mdp_span_begin(c, 2);
  // resident runs: the periodic self-images [nlocal, remote_start) get their fp in the same launch
  const int nimg = c->md && c->ghost_owner.p ? (c->remote_start >= nlocal && c->remote_start <= c->nall ? c->remote_start - nlocal : c->nghost) : 0;
  if (nlocal)
    aeam_embed_kernel<<<nblk(nlocal + nimg, 256), 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->rho.p, c->fp.p, c->eatom.p,
                                                                c->acc.p, eflag, /*accumulate=*/0, nimg, c->ghost_owner.p);
  c->aeam_img_fp = nlocal > 0 && nimg > 0;
  MDP_HIP(c, hipGetLastError());
  mdp_span_end(c, 2);
  c->aeam_phase |= AE_DENS;
  if (early) { // the forces on ghosts are complete after this: their reverse exchange can start with the fp exchange
    mdp_span_begin(c, 4);
    MDP_TRY(aeam_ang_forces(c, c->aeam_vflag));
    mdp_span_end(c, 4);
  }
  return MDP_OK;
}

// phase C: needs fp of owned atoms and periodic self-images only
int mdp_aeam_run_force_begin(mdp_ctx *c, int eflag, int vflag)
{
  if (!(c->aeam_phase & AE_DENS_INT) || !(c->aeam_phase & AE_DENS) || (c->aeam_phase & AE_FORCE_INT)) return MDP_OK;
  if ((vflag & MDP_VFLAG_ATOM) || vflag != c->aeam_vflag)
    return mdp_fail(c, MDP_EINVAL, "aeam: vflag differs between the phases of one compute");
  mdp_span_begin(c, 6);
  MDP_TRY(aeam_force_tiles(c, 0, c->aeam_split, eflag, vflag));
  mdp_span_end(c, 6);
  c->aeam_phase |= AE_FORCE_INT;
  return MDP_OK;
}

// phase D: pass 3.  Owned forces complete except for the angular terms other ranks' centres put on our atoms;
// ghost forces hold our angular centres' contributions.
int mdp_aeam_run_force(mdp_ctx *c, int eflag, int vflag)
{
  hipStream_t st = c->stream;
  const int nlocal = c->nlocal;
  MDP_TRY(aeam_open(c)); // (force_clear; normally done by the density pass)
  if ((c->aeam_phase & AE_ANG_F) && vflag != c->aeam_vflag)
    return mdp_fail(c, MDP_EINVAL, "aeam: vflag differs between the phases of one compute");
  if (vflag & MDP_VFLAG_ATOM) {
    MDP_HIP(c, c->vatom.reserve((size_t) 6 * c->nall + 6));
    MDP_HIP(c, hipMemsetAsync(c->vatom.p, 0, sizeof(double) * 6 * c->nall, st));
  }
  mdp_span_begin(c, 3);
  if (nlocal && c->aeam_tiled && !(vflag & MDP_VFLAG_ATOM)) { // tile lists; per-atom virial steps keep the CSR kernel
    // (the force pass as two persistent passes like the density's -- rho' terms, phi' terms, one LDS table each -- was
    //  built and measured slower than this gather kernel, which fetches both derivatives of a pair from one record
    //  and does the geometry once: 1.4 ms against 0.65 ms)
    MDP_TRY(aeam_force_tiles(c, (c->aeam_phase & AE_FORCE_INT) ? c->aeam_split : 0, c->ntile, eflag, vflag));
  } else if (nlocal) {
    if (!c->csr_full) { // first per-atom-virial step of a tiled run: build the rows of the metal atoms now (and from now on)
      c->csr_want_full = true;
      MDP_TRY(mdp_md_build_master_list(c));
    }
    const int grid = nblk(nlocal, 256 / AE_L);
    const bool ev = eflag || vflag;
#define MDP_AF(NTV, EVV)                                                                                              \
  aeam_force_kernel<AE_L, NTV, EVV><<<grid, 256, 0, st>>>(c->aeam, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->fp.p,      \
                                                          c->f.p, c->eatom.p, c->vatom.p, c->acc.p, eflag, vflag)
    switch (c->aeam.ntypes) {
      case 1: if (ev) MDP_AF(1, true); else MDP_AF(1, false); break;
      case 2: if (ev) MDP_AF(2, true); else MDP_AF(2, false); break;
      case 3: if (ev) MDP_AF(3, true); else MDP_AF(3, false); break;
      case 4: if (ev) MDP_AF(4, true); else MDP_AF(4, false); break;
      case 5: case 6: case 7: case 8: if (ev) MDP_AF(MDP_AEAM_MAXT, true); else MDP_AF(MDP_AEAM_MAXT, false); break;
      default: if (ev) MDP_AF(0, true); else MDP_AF(0, false); break; // more than MDP_AEAM_MAXT types
    }
#undef MDP_AF
  }
  mdp_span_end(c, 3);
  if (!(c->aeam_phase & AE_ANG_F)) {
    mdp_span_begin(c, 4);
    MDP_TRY(aeam_ang_forces(c, vflag));
    mdp_span_end(c, 4);
  }
  MDP_HIP(c, hipGetLastError());
  c->aeam_phase = 0;
  return mdp_acc_end(c, eflag || vflag); // force-only steps tally nothing: no slots to fold
}

extern "C" {

int mdp_aeam_set_tables(mdp_ctx *c, const mdp_aeam_tables *t)
{
  if (!c || !t) return MDP_EINVAL;
  if (t->ntypes < 1 || t->ntypes > MDP_AEAM_MAXTYPES || t->nelements < 1 || t->nelements > MDP_AEAM_MAXTYPES ||
      t->ntypes > t->nelements)
    return mdp_fail(c, MDP_EINVAL, "aeam: 1..%d atom types/elements, ntypes <= nelements (the rho(r) tables are numbered by "
                                   "type pair inside an array sized by element pairs, pair_aeam.cpp:816-821)", MDP_AEAM_MAXTYPES);
  MDP_HIP(c, hipSetDevice(c->device));
  AeamDev &A = c->aeam;
  memset(&A, 0, sizeof A);
  A.ntypes = t->ntypes;
  A.nelements = t->nelements;
  A.nnonangular = t->nnonangular;
  A.nrhomax = t->nrhomax;
  A.nrmax = t->nrmax;
  const int nt = t->ntypes, ne = t->nelements, np = nt * nt;
  const bool small = nt <= MDP_AEAM_MAXT; // the parameter block also rides in the kernel arguments
  std::vector<double> pd((size_t) 2 * np + nt);       // cut | rdr | rdrho
  std::vector<int> pi((size_t) 3 * np + 2 * nt);      // nr | t2rhor | t2z2r | nrho | t2frho
  for (int a = 0; a < nt; a++) {
    // element of type a+1 is a (coeff() insists on file order, pair_aeam.cpp:568-572)
    pd[(size_t) 2 * np + a] = 1 / t->drho[a];
    pi[(size_t) 3 * np + a] = t->nrho[a];
    pi[(size_t) 3 * np + nt + a] = t->type2frho[a + 1];
    if (small) {
      A.rdrho[a] = 1 / t->drho[a];
      A.nrho[a] = t->nrho[a];
      A.t2frho[a] = t->type2frho[a + 1];
    }
    for (int b = 0; b < nt; b++) {
      const int k = a * nt + b;
      pd[k] = t->cut[a * ne + b];
      pd[(size_t) np + k] = 1 / t->dr[a * ne + b];
      pi[k] = t->nr[a * ne + b];
      pi[(size_t) np + k] = t->type2rhor[(size_t) (a + 1) * (nt + 1) + (b + 1)];
      pi[(size_t) 2 * np + k] = t->type2z2r[(size_t) (a + 1) * (nt + 1) + (b + 1)];
      if (small) {
        A.cut[k] = pd[k];
        A.rdr[k] = pd[(size_t) np + k];
        A.nr[k] = pi[k];
        A.t2rhor[k] = pi[(size_t) np + k];
        A.t2z2r[k] = pi[(size_t) 2 * np + k];
      }
    }
  }
  c->aeam_hcut.assign(pd.begin(), pd.begin() + np);
  MDP_HIP(c, c->aeam_par_d.reserve(pd.size() + 1));
  MDP_HIP(c, c->aeam_par_i.reserve(pi.size() + 1));
  MDP_HIP(c, hipMemcpyAsync(c->aeam_par_d.p, pd.data(), pd.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipMemcpyAsync(c->aeam_par_i.p, pi.data(), pi.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipStreamSynchronize(c->stream)); // (pd / pi go out of scope)
  A.g_cut = c->aeam_par_d.p;
  A.g_rdr = c->aeam_par_d.p + np;
  A.g_rdrho = c->aeam_par_d.p + 2 * (size_t) np;
  A.g_nr = c->aeam_par_i.p;
  A.g_t2rhor = c->aeam_par_i.p + np;
  A.g_t2z2r = c->aeam_par_i.p + 2 * (size_t) np;
  A.g_nrho = c->aeam_par_i.p + 3 * (size_t) np;
  A.g_t2frho = c->aeam_par_i.p + 3 * (size_t) np + nt;
  const size_t nf = (size_t) t->nfrho * (t->nrhomax + 1) * 7, nr = (size_t) t->nrhor * (t->nrmax + 1) * 7,
               nz = (size_t) t->nz2r * (t->nrmax + 1) * 7;
  MDP_HIP(c, c->aeam_frho.reserve(nf));
  MDP_HIP(c, c->aeam_rhor.reserve(nr));
  MDP_HIP(c, c->aeam_z2r.reserve(nz));
  MDP_HIP(c, hipMemcpyAsync(c->aeam_frho.p, t->frho_spline, nf * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipMemcpyAsync(c->aeam_rhor.p, t->rhor_spline, nr * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipMemcpyAsync(c->aeam_z2r.p, t->z2r_spline, nz * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  A.frho = c->aeam_frho.p;
  A.rhor = c->aeam_rhor.p;
  A.z2r = c->aeam_z2r.p;
  {
    const size_t rr = (size_t) t->nrhor * (t->nrmax + 1), zr = (size_t) t->nz2r * (t->nrmax + 1);
    MDP_HIP(c, c->aeam_rhor_v4.reserve(rr));
    MDP_HIP(c, c->aeam_rhor_d4.reserve(rr));
    MDP_HIP(c, c->aeam_z2r_v4.reserve(zr));
    MDP_HIP(c, c->aeam_z2r_d4.reserve(zr));
    relay_kernel<<<(int) ((rr + 255) / 256), 256, 0, c->stream>>>(rr, c->aeam_rhor.p, c->aeam_rhor_v4.p, c->aeam_rhor_d4.p);
    relay_kernel<<<(int) ((zr + 255) / 256), 256, 0, c->stream>>>(zr, c->aeam_z2r.p, c->aeam_z2r_v4.p, c->aeam_z2r_d4.p);
    MDP_HIP(c, hipGetLastError());
    MDP_HIP(c, hipStreamSynchronize(c->stream));
    MDP_HIP(c, c->aeam_rhor_ys.reserve(rr + 1));
    MDP_HIP(c, c->aeam_z2r_ys.reserve(zr + 1));
    ys_kernel<<<(int) ((rr + 255) / 256), 256, 0, c->stream>>>(rr, c->aeam_rhor.p, c->aeam_rhor_ys.p);
    ys_kernel<<<(int) ((zr + 255) / 256), 256, 0, c->stream>>>(zr, c->aeam_z2r.p, c->aeam_z2r_ys.p);
    MDP_HIP(c, hipGetLastError());
    MDP_HIP(c, hipStreamSynchronize(c->stream));
    A.rhor_ys = c->aeam_rhor_ys.p;
    A.z2r_ys = c->aeam_z2r_ys.p;
    A.rhor_v4 = c->aeam_rhor_v4.p;
    A.rhor_d4 = c->aeam_rhor_d4.p;
    A.z2r_v4 = c->aeam_z2r_v4.p;
    A.z2r_d4 = c->aeam_z2r_d4.p;
    // per pair type: {rho' coefficients | phi' coefficients} in one 64-byte record (tile force kernel)
    const int npair = A.ntypes * A.ntypes, nm1 = A.nrmax + 1;
    MDP_HIP(c, c->aeam_pair_d8.reserve((size_t) npair * nm1 * 6 + 8));
    pair_der_kernel<<<(int) (((size_t) npair * nm1 + 255) / 256), 256, 0, c->stream>>>(
        npair, nm1, A.g_t2rhor, A.g_t2z2r, c->aeam_rhor.p, c->aeam_z2r.p, c->aeam_pair_d8.p);
    MDP_HIP(c, hipGetLastError());
    MDP_HIP(c, hipStreamSynchronize(c->stream));
    A.pair_d6 = reinterpret_cast<const double2 *>(c->aeam_pair_d8.p);
  }
  c->have_aeam = true;
  return MDP_OK;
}

static int aeam_fetch(mdp_ctx *c, double *eng, double *virial)
{
  hipStream_t st = c->stream;
  MDP_HIP(c, hipMemcpyAsync(c->h_pinned, c->acc.p, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
  int *hflags = (int *) (c->h_pinned + 16);
  MDP_HIP(c, hipMemcpyAsync(hflags, c->flags.p, sizeof(int) * 5, hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  MDP_TRY(mdp_flags_check(c, hflags));
  if (eng) *eng += c->h_pinned[0];
  if (virial)
    for (int k = 0; k < 6; k++) virial[k] += c->h_pinned[1 + k];
  return MDP_OK;
}

int mdp_aeam_density_host(mdp_ctx *c, int eflag, double *fp, double *rho, double *eng_vdwl, double *eatom)
{
  if (!c) return MDP_EINVAL;
  if (!c->have_aeam) return mdp_fail(c, MDP_ESTATE, "aeam tables not set");
  const bool own_lists = c->aeam_device_lists && c->host_sort; // lists built here from the positions (as rebomos)
  if (!c->atoms_set || (!own_lists && !c->neigh_set) || (own_lists && !c->skin_set))
    return mdp_fail(c, MDP_ESTATE, "atoms / neighbor list not set");
  if (c->nlocal == 0) { // no owned atoms on this rank: nothing to tally, the host's arrays may be NULL
    c->rebo_packed = true;
    return MDP_OK;
  }
  // fp may stay on the device when the library derives the images itself (mdp_host_ghosts_derived): the style's
  // forward_comm of fp (pair_aeam.cpp:307) then happens there too
  if (!fp && !c->host_ghosts_derived)
    return mdp_fail(c, MDP_EINVAL, "mdp_aeam_density_host: fp missing for %d owned atoms", c->nlocal);
  MDP_HIP(c, hipSetDevice(c->device));
  if (!c->rebo_packed) { // reuse the flag: "style structures follow the current list"
    if (own_lists) {
      // the host's list is not read (the flattening of 86 M entries per million atoms on one host thread cost more
      // than ten steps): bins, the CSR rows of the angular centres and the tile lists come from the positions
      c->cfg.style = 2;
      c->cfg.skin = c->skin;
      c->cfg.master_list = 0;
      for (int d = 0; d < 3; d++) {
        c->cfg.bbox_lo[d] = c->bbox_lo[d];
        c->cfg.bbox_hi[d] = c->bbox_hi[d];
      }
      MDP_TRY(mdp_md_build_master_list(c));
    }
    MDP_TRY(mdp_aeam_prepare(c));
    c->rebo_packed = true;
  }
  if ((eflag & MDP_EFLAG_ATOM) && !eatom) eflag &= ~MDP_EFLAG_ATOM;
  MDP_TRY(mdp_aeam_run_density(c, eflag));
  MDP_TRY(mdp_acc_end(c, true));
  hipStream_t st = c->stream;
  const int n = c->nlocal;
  MDP_TRY(mdp_host_pinned_reserve(c, (size_t) n + 16));
  double *he = c->h_down;
  const double *dfp = c->fp.p, *drho = c->rho.p, *dea = c->eatom.p;
  if (c->host_sort) { // back to the host's atom order
    MDP_HIP(c, c->host_stage.reserve((size_t) 10 * c->nall + 16));
    double *s0 = c->host_stage.p, *s1 = s0 + n, *s2 = s1 + n;
    if (fp) MDP_TRY(mdp_to_host_order(c, n, 1, c->fp.p, s0));
    if (rho) MDP_TRY(mdp_to_host_order(c, n, 1, c->rho.p, s1));
    if (eflag & MDP_EFLAG_ATOM) MDP_TRY(mdp_to_host_order(c, n, 1, c->eatom.p, s2));
    dfp = s0;
    drho = s1;
    dea = s2;
  }
  if (fp) MDP_HIP(c, hipMemcpyAsync(fp, dfp, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  if (rho) MDP_HIP(c, hipMemcpyAsync(rho, drho, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  if (eflag & MDP_EFLAG_ATOM) MDP_HIP(c, hipMemcpyAsync(he, dea, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  // nothing for the host to wait for on a force-only step that keeps fp here: overflow bits are sticky and stop the
  // force half's read (mdp_flags_check)
  if (!fp && !rho && !(eflag & (MDP_EFLAG_GLOBAL | MDP_EFLAG_ATOM))) return MDP_OK;
  MDP_TRY(aeam_fetch(c, (eflag & MDP_EFLAG_GLOBAL) ? eng_vdwl : nullptr, nullptr));
  if (eflag & MDP_EFLAG_ATOM) mdp_host_add(eatom, he, (size_t) n);
  return MDP_OK;
}

int mdp_aeam_force_host(mdp_ctx *c, int eflag, int vflag, const double *fp_all, double *f, double *eng_vdwl,
                        double *virial, double *eatom, double *vatom)
{
  if (!c) return MDP_EINVAL;
  if (!c->have_aeam || !c->atoms_set || !c->neigh_set || !c->rebo_packed)
    return mdp_fail(c, MDP_ESTATE, "aeam: call mdp_aeam_density_host first");
  if (c->nlocal == 0) return MDP_OK; // every force term starts from an owned atom (pair_aeam.cpp:337)
  const bool local_halo = c->host_ghosts_derived; // images filled and folded here, owned atoms' results only go back
  // (f may be NULL while the integrator lives on the device too, mdp_hnve_*: needs the images kept by the library)
  if ((!fp_all && !local_halo) || (!f && !(c->hn_on && local_halo)))
    return mdp_fail(c, MDP_EINVAL, "mdp_aeam_force_host: fp / f missing for %d atoms", c->nall);
  MDP_HIP(c, hipSetDevice(c->device));
  if ((eflag & MDP_EFLAG_ATOM) && !eatom) eflag &= ~MDP_EFLAG_ATOM;
  if ((vflag & MDP_VFLAG_ATOM) && !vatom) vflag &= ~MDP_VFLAG_ATOM;
  hipStream_t st = c->stream;
  const int n = c->nlocal, nall = c->nall;
  // ghosts' fp come from the host's forward comm; owned values are already on the device
  if (!fp_all) {
    MDP_TRY(mdp_host_ghost_scalar(c, c->fp.p)); // Comm::forward_comm of fp on one periodic rank
  } else if (c->host_sort) { // the host's array is in its own atom order: whole array up, then into device order
    MDP_HIP(c, c->host_stage.reserve((size_t) 10 * nall + 16));
    MDP_HIP(c, hipMemcpyAsync(c->host_stage.p, fp_all, sizeof(double) * nall, hipMemcpyHostToDevice, st));
    MDP_TRY(mdp_to_device_order(c, nall, 1, c->host_stage.p, c->fp.p));
  } else if (nall > n)
    MDP_HIP(c, hipMemcpyAsync(c->fp.p + n, fp_all + n, sizeof(double) * (nall - n), hipMemcpyHostToDevice, st));
  MDP_TRY(mdp_acc_begin(c, true));
  MDP_HIP(c, hipMemsetAsync(c->eatom.p, 0, sizeof(double) * n, st));
  MDP_TRY(mdp_aeam_run_force(c, eflag, vflag));
  // results come back through the pinned buffer and are ADDED on the host (LAMMPS semantics; ghosts included, unless
  // the images are the library's own: then what they collected is folded onto their owners here, which is what the
  // host's reverse_comm would do with it, and the host's ghost entries receive nothing)
  const int nout = local_halo ? n : nall;
  if (local_halo) {
    MDP_TRY(mdp_host_ghost_fold(c, 3, c->f.p));
    if (vflag & MDP_VFLAG_ATOM) MDP_TRY(mdp_host_ghost_fold(c, 6, c->vatom.p));
  }
  if (!f) { // forces stay with the device integrator; totals only when asked for
    if ((eflag & MDP_EFLAG_ATOM) || (vflag & MDP_VFLAG_ATOM))
      return mdp_fail(c, MDP_EINVAL, "mdp_aeam_force_host: per-atom tallies without f");
    if (!(eflag || vflag)) return MDP_OK;
    return aeam_fetch(c, (eflag & MDP_EFLAG_GLOBAL) ? eng_vdwl : nullptr, (vflag & MDP_VFLAG_GLOBAL) ? virial : nullptr);
  }
  MDP_TRY(mdp_host_pinned_reserve(c, (size_t) 10 * nall + 16));
  double *hf = c->h_down, *he = hf + (size_t) 3 * nall, *hv = he + nall;
  const double *df = c->f.p, *de = c->eatom.p, *dv = c->vatom.p;
  if (c->host_sort) {
    MDP_HIP(c, c->host_stage.reserve((size_t) 10 * nall + 16));
    double *s0 = c->host_stage.p, *s1 = s0 + (size_t) 3 * nall, *s2 = s1 + nall;
    MDP_TRY(mdp_to_host_order(c, nout, 3, c->f.p, s0));
    df = s0;
    if (eflag & MDP_EFLAG_ATOM) {
      MDP_TRY(mdp_to_host_order(c, n, 1, c->eatom.p, s1));
      de = s1;
    }
    if (vflag & MDP_VFLAG_ATOM) {
      MDP_TRY(mdp_to_host_order(c, nout, 6, c->vatom.p, s2));
      dv = s2;
    }
  }
  if (eflag & MDP_EFLAG_ATOM) MDP_HIP(c, hipMemcpyAsync(he, de, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  if (vflag & MDP_VFLAG_ATOM)
    MDP_HIP(c, hipMemcpyAsync(hv, dv, sizeof(double) * 6 * nout, hipMemcpyDeviceToHost, st));
  MDP_TRY(mdp_host_download_add(c, f, hf, df, (size_t) 3 * nout));
  MDP_TRY(aeam_fetch(c, (eflag & MDP_EFLAG_GLOBAL) ? eng_vdwl : nullptr, (vflag & MDP_VFLAG_GLOBAL) ? virial : nullptr));
  if (vflag & MDP_VFLAG_ATOM) mdp_host_add(vatom, hv, (size_t) 6 * nout);
  if (eflag & MDP_EFLAG_ATOM) mdp_host_add(eatom, he, (size_t) n);
  return MDP_OK;
}

} // extern "C"
