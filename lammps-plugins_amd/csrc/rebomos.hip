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
//      force of E_c on each neighbour slot m is written to fnbr[c][m] -- one plain store per slot.
//   2. rebo_lj_gather_kernel<L>: L lanes per owned atom stream the trimmed, repacked Lennard-Jones
//      list (full list, both directions, no parity rule needed), then gather the REBO forces
//      F_a = -sum_m fnbr[a][m] + sum_{c in N(a)} fnbr[c][slot of a], reduce across the L lanes with
//      wave shuffles and store f[a] once.
// Nothing is written to ghost atoms; the explicit virial replaces virial_fdotr_compute.
#include "mdp_common.h"

namespace {

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

template <int W> __device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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

// G(cos) and dG/dcos, pair_rebomos.h:68-167.  cb/cg: the centre element's b0..b6 / bg0..bg6.
__device__ __forceinline__ double gspline(const double *cb, const double *cg, double c, double &dgdc)
{
  if (c < 0.5) { // caller clamps to [-1,1] (pair_rebomos.cpp:617-618)
    return poly6(cb, c, dgdc);
  }
  double dgcos, dgamma;
  const double gcos = poly6(cb, c, dgcos);
  const double gamma = poly6(cg, c, dgamma);
  double s, co;
  sincospi(2.0 * (c - 0.5), &s, &co);
  const double psi = 0.5 * (1.0 - co);
  const double dpsi = kPi * s;
  dgdc = dgcos + dpsi * (gamma - gcos) + psi * (dgamma - dgcos);
  return gcos + psi * (gamma - gcos);
}

// ------------------------------------------------------------------------------------------------
// centre kernel
// ------------------------------------------------------------------------------------------------
template <int G> struct CentreCfg {
  static constexpr int CAP = 2 * G;     // LDS slots per centre
  static constexpr int GPW = 64 / G;    // centres per wave
  static constexpr int WPB = 4;         // waves per block
  static constexpr int REC = 8;         // doubles per slot: dx dy dz r w dw C p
  static constexpr int STRIDE = CAP * REC + REC; // +1 record of padding per centre
};

template <int G>
__global__ __launch_bounds__(256) void rebo_centre_kernel(
    const RebomosDev P, const int *__restrict__ centres, const int ncent, const int nlocal,
    const double4 *__restrict__ xq, const int *__restrict__ cand_off, const int *__restrict__ cand,
    int *__restrict__ rn_num, int *__restrict__ rn_idx, double *__restrict__ fnbr, double *__restrict__ eslot,
    double *__restrict__ acc, int *__restrict__ flags, const int eflag, const int vflag)
{
  using C = CentreCfg<G>;
  __shared__ double s_rec[C::WPB * C::GPW * C::STRIDE];
  __shared__ int s_je[C::WPB * C::GPW * C::CAP];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int s = lane % G;          // lane within the centre's group
  const int glane0 = lane - s;     // first lane of the group
  const int grp_in_block = tid / G;
  const long long gid = (long long) blockIdx.x * (256 / G) + grp_in_block;
  const bool have = gid < ncent;

  double *rec = s_rec + (size_t) grp_in_block * C::STRIDE;
  int *je = s_je + grp_in_block * C::CAP;

  int c = 0, off = 0, nc = 0, tc = 0;
  double4 xc = make_double4(0, 0, 0, 0);
  if (have) {
    c = centres[gid];
    off = cand_off[c];
    nc = cand_off[c + 1] - off;
    xc = xq[c];
    tc = (int) xc.w;
  }

  // ---- phase A: filter the candidates to the current REBO set (pair_rebomos.cpp:328-344) --------
  int n = 0;
  double nsum = 0.0;
  const unsigned long long gmask = (G == 64) ? ~0ull : ((1ull << G) - 1ull);
  const int ncw = wave_max_int(nc);
  for (int base = 0; base < ncw; base += G) {
    const int t = base + s;
    const bool valid = t < nc;
    int j = c;
    bool pred = false;
    double dx = 0, dy = 0, dz = 0, rsq = 0;
    int tj = 0;
    if (valid) {
      j = cand[off + t];
      const double4 xj = xq[j];
      dx = xc.x - xj.x;
      dy = xc.y - xj.y;
      dz = xc.z - xj.z;
      rsq = dx * dx + dy * dy + dz * dz;
      tj = (int) xj.w;
      pred = rsq < P.rcmaxsq[tc * 2 + tj];
    }
    const unsigned long long bal = __ballot(pred);
    const unsigned long long gb = (bal >> glane0) & gmask;
    const int pos = n + __popcll(gb & ((1ull << s) - 1ull));
    if (pred && pos < C::CAP) {
      const int pt = tc * 2 + tj;
      const double r = sqrt(rsq);
      double dw;
      const double w = sp_switch(r, P.rcmin[pt], P.rcinv[pt], dw);
      double *q = rec + pos * C::REC;
      q[0] = dx;
      q[1] = dy;
      q[2] = dz;
      q[3] = r;
      q[4] = w;
      q[5] = dw;
      je[pos] = j | (tj << 30);
      rn_idx[off + pos] = j;
      nsum += w; // nM + nS (pair_rebomos.cpp:339-342); only their sum is ever used (:628, h:175)
    }
    n += __popcll(gb);
  }
  if (n > C::CAP) {
    if (s == 0) atomicOr(&flags[0], 1);
    n = C::CAP;
  }
  if (have && s == 0) rn_num[c] = n;
  const double Ntot = group_sum<G>(nsum);
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

  // ---- phase B: p_cm and C_m = 1/2 V_A (-1/2 p^3) for every slot (pair_rebomos.cpp:607-630) ------
  double csum_part = 0.0;
  for (int mb = 0; mb < nw; mb += G) {
    const int m = mb + s;
    const bool act = m < n;
    double mx = 0, my = 0, mz = 0, mr = 1, mw = 0;
    if (act) {
      const double *q = rec + m * C::REC;
      mx = q[0];
      my = q[1];
      mz = q[2];
      mr = q[3];
      mw = q[4];
    }
    const double mrinv = 1.0 / mr;
    double S = 0.0;
    for (int qi = 0; qi < nw; qi++) {
      if (act && qi < n && qi != m) {
        const double *q = rec + qi * C::REC;
        double cs = (mx * q[0] + my * q[1] + mz * q[2]) / (mr * q[3]);
        cs = fmin(cs, 1.0);
        cs = fmax(cs, -1.0);
        double dg;
        const double g = gspline(cb, cg, cs, dg);
        S += q[4] * g;
      }
    }
    if (act) {
      const int tm = ((unsigned) je[m]) >> 30;
      const int pt = tc * 2 + tm;
      const double p = 1.0 / sqrt(1.0 + S + PS);
      const double VA = -mw * P.B[pt] * exp(-P.beta[pt] * mr);
      const double Cm = (mw > kTol) ? 0.5 * VA * (-0.5 * p * p * p) : 0.0;
      double *q = rec + m * C::REC;
      q[6] = Cm;
      q[7] = p;
      csum_part += Cm;
    }
    (void) mrinv;
  }
  const double Csum = group_sum<G>(csum_part);
  wave_lds_fence();

  // ---- phase C: force of E_c on every neighbour slot (pair_rebomos.cpp:411-441, 634-725) ----------
  double e_acc = 0.0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  const bool owned = have && c < nlocal;
  for (int mb = 0; mb < nw; mb += G) {
    const int m = mb + s;
    const bool act = m < n;
    double mx = 0, my = 0, mz = 0, mr = 1, mw = 0, mdw = 0, mC = 0, mp = 0;
    if (act) {
      const double *q = rec + m * C::REC;
      mx = q[0];
      my = q[1];
      mz = q[2];
      mr = q[3];
      mw = q[4];
      mdw = q[5];
      mC = q[6];
      mp = q[7];
    }
    const double mrinv = 1.0 / mr;
    const double ux = mx * mrinv, uy = my * mrinv, uz = mz * mrinv; // unit vector (x_c - x_m)/r
    double fx = 0, fy = 0, fz = 0, acc1 = 0;
    for (int qi = 0; qi < nw; qi++) {
      if (act && qi < n && qi != m) {
        const double *q = rec + qi * C::REC;
        const double qrinv = 1.0 / q[3];
        double cs = (mx * q[0] + my * q[1] + mz * q[2]) * (mrinv * qrinv);
        cs = fmin(cs, 1.0);
        cs = fmax(cs, -1.0);
        double dg;
        const double g = gspline(cb, cg, cs, dg);
        // (C_m w_q + C_q w_m) G'(cos) d cos / d x_m ; d cos/d x_m = -(u_q - cos u_m)/r_m
        // force = -gradient
        const double coef = (mC * q[4] + q[6] * mw) * dg * mrinv;
        fx += coef * (q[0] * qrinv - cs * ux);
        fy += coef * (q[1] * qrinv - cs * uy);
        fz += coef * (q[2] * qrinv - cs * uz);
        acc1 += q[6] * g;
      }
    }
    if (act) {
      const int tm = ((unsigned) je[m]) >> 30;
      const int pt = tc * 2 + tm;
      double radial = (acc1 + Csum * dp) * mdw; // dw_cm [ sum_q C_q G + P'(N) sum_j C_j ]
      double ehalf = 0.0;
      if (mw > kTol) {
        // pair_rebomos.cpp:418-427
        const double ex = exp(-P.alpha[pt] * mr);
        const double pre = mw * P.A[pt] * ex;
        const double VR = pre * (1.0 + P.Q[pt] * mrinv);
        double dVR = pre * (-P.alpha[pt] - P.Q[pt] * mrinv * mrinv - P.Q[pt] * P.alpha[pt] * mrinv);
        dVR += VR / mw * mdw;
        const double VA = -mw * P.B[pt] * exp(-P.beta[pt] * mr);
        double dVA = -P.beta[pt] * VA;
        dVA += VA / mw * mdw;
        radial += 0.5 * (dVR + mp * dVA);
        ehalf = 0.5 * (VR + mp * VA);
      }
      fx += radial * ux;
      fy += radial * uy;
      fz += radial * uz;
      double *o = fnbr + 3 * (size_t) (off + m);
      o[0] = fx;
      o[1] = fy;
      o[2] = fz;
      if (eflag & MDP_EFLAG_ATOM) eslot[off + m] = 0.5 * ehalf;
      if (owned) {
        e_acc += ehalf;
        // virial of the cluster: sum_m (x_m - x_c) (x) F_m = -sum_m d_m (x) F_m
        v0 -= mx * fx;
        v1 -= my * fy;
        v2 -= mz * fz;
        v3 -= mx * fy;
        v4 -= mx * fz;
        v5 -= my * fz;
      }
    }
  }

  // partial sums go to one of MDP_ACC_SLOTS slots (by block) so that a million waves do not queue
  // on seven addresses; acc_reduce_kernel folds the slots afterwards
  double *slot = acc + MDP_ACC_STRIDE * (1 + (blockIdx.x & (MDP_ACC_SLOTS - 1)));
  if (eflag & MDP_EFLAG_GLOBAL) {
    const double e = group_sum<64>(e_acc);
    if (lane == 0) atomicAdd(&slot[0], e);
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

// ------------------------------------------------------------------------------------------------
// Lennard-Jones over the trimmed full list + gather of the REBO slot forces
// ------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void rebo_lj_gather_kernel(
    const RebomosDev P, const int nlocal, const double4 *__restrict__ xq, const long long *__restrict__ lj_off,
    const int *__restrict__ lj_cnt, const int *__restrict__ lj, const int *__restrict__ cand_off,
    const int *__restrict__ rn_num, const int *__restrict__ rn_idx, const double *__restrict__ fnbr,
    const double *__restrict__ eslot, double *__restrict__ f, double *__restrict__ eatom, double *__restrict__ acc,
    const int eflag, const int vflag, const int accumulate)
{
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int s = lane % L;
  const long long a64 = (long long) blockIdx.x * (256 / L) + tid / L;
  const bool have = a64 < nlocal;
  const int a = have ? (int) a64 : 0;

  double fx = 0, fy = 0, fz = 0, e = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  const double4 xa = xq[a];
  const int ta = (int) xa.w;

  if (have) {
    // ---- FLJ (pair_rebomos.cpp:493-556), both directions of every pair, half the energy each
    const int cnt = lj_cnt[a];
    const int *row = lj + lj_off[a];
    for (int k = s; k < cnt; k += L) {
      const int j = row[k];
      const double4 xj = xq[j];
      const double dx = xa.x - xj.x, dy = xa.y - xj.y, dz = xa.z - xj.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      const int pt = ta * 2 + (int) xj.w;
      if (rsq >= P.lj_rsq_lo[pt] && rsq <= P.lj_rsq_hi[pt]) {
        double fpair, V;
        if (rsq >= P.lj_rsq_sw[pt]) {
          const double r2inv = 1.0 / rsq;
          const double r6inv = r2inv * r2inv * r2inv;
          V = r6inv * (P.lj3[pt] * r6inv - P.lj4[pt]);
          fpair = r6inv * (P.lj1[pt] * r6inv - P.lj2[pt]) * r2inv;
        } else {
          const double rij = sqrt(rsq);
          const double drp = rij - P.rcLJmin[pt];
          V = drp * drp * (drp * P.ljc3[pt] + P.ljc2[pt]);
          fpair = -drp * (3.0 * drp * P.ljc3[pt] + 2.0 * P.ljc2[pt]) / rij;
        }
        fx += dx * fpair;
        fy += dy * fpair;
        fz += dz * fpair;
        e += 0.5 * V;
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
  }
  const double e_lj = e; // LJ part goes to both the global and the per-atom energy
  double e_rebo_atom = 0.0;

  if (have) {
    // ---- gather the REBO cluster forces: own centre (-sum of slot forces) + neighbour centres
    const int off = cand_off[a];
    const int n = rn_num[a];
    for (int m = s; m < n; m += L) {
      const double *o = fnbr + 3 * (size_t) (off + m);
      fx -= o[0];
      fy -= o[1];
      fz -= o[2];
      const int j = rn_idx[off + m];
      const int offj = cand_off[j];
      const int nj = rn_num[j];
      for (int u = 0; u < nj; u++) {
        if (rn_idx[offj + u] == a) {
          const double *oj = fnbr + 3 * (size_t) (offj + u);
          fx += oj[0];
          fy += oj[1];
          fz += oj[2];
          if (eflag & MDP_EFLAG_ATOM) e_rebo_atom += eslot[off + m] + eslot[offj + u];
          break;
        }
      }
    }
  }

  fx = group_sum<L>(fx);
  fy = group_sum<L>(fy);
  fz = group_sum<L>(fz);
  if (have && s == 0) {
    double *fo = f + 3 * (size_t) a;
    if (accumulate) {
      fo[0] += fx;
      fo[1] += fy;
      fo[2] += fz;
    } else {
      fo[0] = fx;
      fo[1] = fy;
      fo[2] = fz;
    }
  }
  if (eflag & MDP_EFLAG_ATOM) {
    const double ea = group_sum<L>(e_lj + e_rebo_atom);
    if (have && s == 0) {
      if (accumulate)
        eatom[a] += ea;
      else
        eatom[a] = ea;
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

// ------------------------------------------------------------------------------------------------
// repack at every neighbor (re)build: master CSR list -> REBO candidates + trimmed LJ list
// ------------------------------------------------------------------------------------------------
constexpr int RP_L = 16; // lanes per atom in the repack kernels

// counts per atom: cand_cnt[i] (all atoms), lj_cnt[i] (owned)
__global__ __launch_bounds__(256) void repack_count_kernel(const RebomosDev P, const int nall, const int nlocal,
                                                           const double4 *__restrict__ xq,
                                                           const long long *__restrict__ nb_off,
                                                           const int *__restrict__ nb, int *__restrict__ cand_cnt,
                                                           int *__restrict__ lj_cnt)
{
  const int s = threadIdx.x % RP_L;
  const long long i64 = (long long) blockIdx.x * (256 / RP_L) + threadIdx.x / RP_L;
  const bool have = i64 < nall;
  const int i = have ? (int) i64 : 0;
  int nc = 0, nl = 0;
  if (have) {
    const double4 xi = xq[i];
    const int ti = (int) xi.w;
    const long long b = nb_off[i], e = nb_off[i + 1];
    for (long long k = b + s; k < e; k += RP_L) {
      const int j = nb[k];
      const double4 xj = xq[j];
      const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      const int pt = ti * 2 + (int) xj.w;
      nc += rsq <= P.cand_cutsq[pt];
      nl += rsq <= P.ljlist_cutsq[pt];
    }
  }
#pragma unroll
  for (int o = RP_L / 2; o > 0; o >>= 1) {
    nc += __shfl_xor(nc, o, 64);
    nl += __shfl_xor(nl, o, 64);
  }
  if (have && s == 0) {
    cand_cnt[i] = nc;
    if (i < nlocal) lj_cnt[i] = nl;
  }
}

__global__ __launch_bounds__(256) void repack_fill_kernel(const RebomosDev P, const int nall, const int nlocal,
                                                          const double4 *__restrict__ xq,
                                                          const long long *__restrict__ nb_off,
                                                          const int *__restrict__ nb, const int *__restrict__ cand_off,
                                                          int *__restrict__ cand, const long long *__restrict__ lj_off,
                                                          int *__restrict__ lj, int *__restrict__ is_centre)
{
  const int lane = threadIdx.x & 63;
  const int s = lane % RP_L;
  const int glane0 = lane - s;
  const long long i64 = (long long) blockIdx.x * (256 / RP_L) + threadIdx.x / RP_L;
  const bool have = i64 < nall;
  const int i = have ? (int) i64 : 0;
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  const long long b = have ? nb_off[i] : 0, e = have ? nb_off[i + 1] : 0;
  const int len = (int) (e - b);
  const int lenw = wave_max_int(len);
  const bool own = have && i < nlocal;
  int nc = 0, nl = 0;
  const int coff = have ? cand_off[i] : 0;
  const long long loff = own ? lj_off[i] : 0;
  for (int base = 0; base < lenw; base += RP_L) {
    const int k = base + s;
    bool pc = false, pl = false;
    int j = 0;
    if (k < len) {
      j = nb[b + k];
      const double4 xj = xq[j];
      const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
      const double rsq = dx * dx + dy * dy + dz * dz;
      const int pt = ti * 2 + (int) xj.w;
      pc = rsq <= P.cand_cutsq[pt];
      pl = own && rsq <= P.ljlist_cutsq[pt];
    }
    const unsigned long long bc = (__ballot(pc) >> glane0) & ((1ull << RP_L) - 1ull);
    const unsigned long long bl = (__ballot(pl) >> glane0) & ((1ull << RP_L) - 1ull);
    const unsigned long long below = (1ull << s) - 1ull;
    if (pc) {
      cand[coff + nc + __popcll(bc & below)] = j;
      if (own && j >= nlocal) is_centre[j] = 1; // ghost neighbours of owned atoms are centres too
    }
    if (pl) lj[loff + nl + __popcll(bl & below)] = j;
    nc += __popcll(bc);
    nl += __popcll(bl);
  }
  if (own && s == 0) is_centre[i] = 1;
}

// current REBO coordination -> lane-group class, appended to the class lists
__global__ __launch_bounds__(256) void classify_kernel(const RebomosDev P, const int nall,
                                                       const double4 *__restrict__ xq,
                                                       const int *__restrict__ cand_off, const int *__restrict__ cand,
                                                       const int *__restrict__ is_centre, int *__restrict__ class_list,
                                                       int *__restrict__ class_count)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nall || !is_centre[i]) return;
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  int n = 0;
  for (int k = cand_off[i]; k < cand_off[i + 1]; k++) {
    const double4 xj = xq[cand[k]];
    const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
    n += (dx * dx + dy * dy + dz * dz) < P.rcmaxsq[ti * 2 + (int) xj.w];
  }
  const int ncand = cand_off[i + 1] - cand_off[i];
  if (ncand == 0) return;
  // smallest lane group that holds the current coordination with one slot to spare
  const int k = (n <= 3) ? 0 : (n <= 7) ? 1 : (n <= 15) ? 2 : 3;
  const int pos = atomicAdd(&class_count[k], 1);
  class_list[(size_t) k * nall + pos] = i;
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
}

// ------------------------------------------------------------------------------------------------
int mdp_rebomos_repack(mdp_ctx *c)
{
  if (!c->have_rebomos) return mdp_fail(c, MDP_ESTATE, "rebomos parameters not set");
  if (!c->atoms_set || !c->neigh_set) return mdp_fail(c, MDP_ESTATE, "atoms / neighbor list not set");
  mdp_rebomos_fill_dev(c, c->skin);
  const int nall = c->nall, nlocal = c->nlocal;
  hipStream_t st = c->stream;
  MDP_HIP(c, c->cand_cnt.reserve(nall + 1));
  MDP_HIP(c, c->cand_off.reserve(nall + 2));
  MDP_HIP(c, c->lj_cnt.reserve(nlocal + 1));
  MDP_HIP(c, c->lj_off.reserve(nlocal + 2));
  MDP_HIP(c, c->is_center.reserve(nall + 1));
  MDP_HIP(c, c->rn_num.reserve(nall + 1));
  MDP_HIP(c, c->class_list.reserve((size_t) 4 * nall + 4));
  MDP_HIP(c, c->class_count.reserve(4));
  MDP_HIP(c, hipMemsetAsync(c->is_center.p, 0, sizeof(int) * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->rn_num.p, 0, sizeof(int) * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->class_count.p, 0, sizeof(int) * 4, st));
  const int per_block = 256 / RP_L;
  const int grid = (nall + per_block - 1) / per_block;
  repack_count_kernel<<<grid, 256, 0, st>>>(c->rebomos, nall, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->cand_cnt.p,
                                            c->lj_cnt.p);
  MDP_HIP(c, hipGetLastError());
  MDP_TRY(mdp_scan_exclusive_int(c, c->cand_cnt.p, c->cand_off.p, nall));
  MDP_TRY(mdp_scan_exclusive_i64(c, c->lj_cnt.p, c->lj_off.p, nlocal));
  int cand_total = 0;
  long long lj_total = 0;
  MDP_HIP(c, hipMemcpyAsync(&cand_total, c->cand_off.p + nall, sizeof(int), hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipMemcpyAsync(&lj_total, c->lj_off.p + nlocal, sizeof(long long), hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  c->cand_total = cand_total;
  c->lj_total = lj_total;
  MDP_HIP(c, c->cand.reserve((size_t) cand_total + 1));
  MDP_HIP(c, c->lj.reserve((size_t) lj_total + 1));
  MDP_HIP(c, c->rn_idx.reserve((size_t) cand_total + 1));
  MDP_HIP(c, c->fnbr.reserve((size_t) 3 * cand_total + 3));
  MDP_HIP(c, c->eslot.reserve((size_t) cand_total + 1));
  repack_fill_kernel<<<grid, 256, 0, st>>>(c->rebomos, nall, nlocal, c->xq.p, c->nb_off.p, c->nb.p, c->cand_off.p,
                                           c->cand.p, c->lj_off.p, c->lj.p, c->is_center.p);
  MDP_HIP(c, hipGetLastError());
  classify_kernel<<<(nall + 255) / 256, 256, 0, st>>>(c->rebomos, nall, c->xq.p, c->cand_off.p, c->cand.p,
                                                      c->is_center.p, c->class_list.p, c->class_count.p);
  MDP_HIP(c, hipGetLastError());
  MDP_HIP(c, hipMemcpyAsync(c->h_class_count, c->class_count.p, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  c->rebo_packed = true;
  return MDP_OK;
}

template <int G>
static void launch_centre(mdp_ctx *c, int k, int eflag, int vflag)
{
  const int n = c->h_class_count[k];
  if (n <= 0) return;
  const int per_block = 256 / G;
  const int grid = (n + per_block - 1) / per_block;
  rebo_centre_kernel<G><<<grid, 256, 0, c->stream>>>(c->rebomos, c->class_list.p + (size_t) k * c->nall, n, c->nlocal,
                                                     c->xq.p, c->cand_off.p, c->cand.p, c->rn_num.p, c->rn_idx.p,
                                                     c->fnbr.p, c->eslot.p, c->acc.p, c->flags.p, eflag, vflag);
}

// force_clear (optional) + compute on the device; results stay on the device (f, eatom, acc)
int mdp_rebomos_run(mdp_ctx *c, int eflag, int vflag, bool zero_f)
{
  if (!c->rebo_packed) return mdp_fail(c, MDP_ESTATE, "rebomos: neighbor list not repacked");
  if (vflag & MDP_VFLAG_ATOM) return mdp_fail(c, MDP_ENOTIMPL, "rebomos: per-atom virial is not implemented on the device");
  hipStream_t st = c->stream;
  MDP_TRY(mdp_acc_begin(c, eflag || vflag));
  mdp_time_mark(c, 0);
  launch_centre<4>(c, 0, eflag, vflag);
  launch_centre<8>(c, 1, eflag, vflag);
  launch_centre<16>(c, 2, eflag, vflag);
  launch_centre<32>(c, 3, eflag, vflag);
  MDP_HIP(c, hipGetLastError());
  mdp_time_mark(c, 1);
  constexpr int L = 16;
  const int per_block = 256 / L;
  const int grid = (c->nlocal + per_block - 1) / per_block;
  if (grid > 0)
    rebo_lj_gather_kernel<L><<<grid, 256, 0, st>>>(c->rebomos, c->nlocal, c->xq.p, c->lj_off.p, c->lj_cnt.p, c->lj.p,
                                                   c->cand_off.p, c->rn_num.p, c->rn_idx.p, c->fnbr.p, c->eslot.p,
                                                   c->f.p, c->eatom.p, c->acc.p, eflag, vflag, zero_f ? 0 : 1);
  MDP_HIP(c, hipGetLastError());
  mdp_time_mark(c, 2);
  return mdp_acc_end(c, eflag || vflag);
}
