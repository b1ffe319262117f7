// md.hip -- resident mode: the LAMMPS host steps either side of Pair::compute(), kept on the GPU
// so that x and f never cross PCIe inside the MD loop:
//   * binned full(+ghost) neighbor-list build  (LAMMPS "pair build: full/bin/ghost",
//     USER-REBOMOS/log.rebomos-bulk.1:40-51; per-type-pair cutoffs for AEAM, USER-AEAM/sample.in:17)
//   * fix nve velocity-Verlet halves, periodic self-image ghost refresh (forward comm on one rank)
//   * thermo reductions (KE, max displacement for `neigh_modify check yes`)
//   * halo pack / unpack kernels for the multi-GPU exchange (transport is the caller's: RCCL)
#include "mdp_common.h"

#include <rocprim/rocprim.hpp>

#include <cmath>

void mdp_rebomos_fill_dev(mdp_ctx *c, double skin);

namespace {

struct CutTables {
  double owned[MDP_AEAM_MAXT * MDP_AEAM_MAXT]; // cutneighsq[ei*ne+ej]  element/type pair
  double ghost[MDP_AEAM_MAXT * MDP_AEAM_MAXT]; // 0 => no ghost lists
  const double *g_owned;                       // the owned table in device memory when ne > MDP_AEAM_MAXT (else null)
  int ne;           // table stride
  int min_type;     // owned atoms of a lower type get an empty row (AEAM with tile lists: only angular centres read the CSR list)
};

using Grid = MdpGrid;

__device__ __forceinline__ int cell_index(const Grid &g, const double4 &x, int &cx, int &cy, int &cz)
{
  cx = (int) ((x.x - g.lo[0]) * g.inv[0]);
  cy = (int) ((x.y - g.lo[1]) * g.inv[1]);
  cz = (int) ((x.z - g.lo[2]) * g.inv[2]);
  cx = cx < 0 ? 0 : (cx >= g.n[0] ? g.n[0] - 1 : cx);
  cy = cy < 0 ? 0 : (cy >= g.n[1] ? g.n[1] - 1 : cy);
  cz = cz < 0 ? 0 : (cz >= g.n[2] ? g.n[2] - 1 : cz);
  return cx + g.n[0] * (cy + g.n[1] * cz);
}

__global__ void cell_assign_kernel(const Grid g, int nall, const double4 *__restrict__ xq, unsigned *__restrict__ key,
                                   int *__restrict__ val)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nall) return;
  int cx, cy, cz;
  key[i] = (unsigned) cell_index(g, xq[i], cx, cy, cz);
  val[i] = i;
}

__global__ void cell_bounds_kernel(int nall, const unsigned *__restrict__ key_sorted, int *__restrict__ cell_start,
                                   const int *__restrict__ perm, const double4 *__restrict__ xq,
                                   double4 *__restrict__ xq_cell)
{
  // cell_start[c] .. cell_start[c+1] delimit cell c in the sorted order; filled for every cell
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nall) return;
  // positions in cell order: the list builders sweep runs of cells, i.e. runs of p -- one coalesced load of
  // xq_cell[p] instead of the dependent pair perm[p] -> xq[perm[p]] (tile_scan_kernel)
  xq_cell[p] = xq[perm[p]];
  const unsigned k = key_sorted[p];
  const unsigned kprev = p == 0 ? 0u : key_sorted[p - 1];
  if (p == 0)
    for (unsigned c = 0; c <= k; c++) cell_start[c] = 0;
  else
    for (unsigned c = kprev + 1; c <= k; c++) cell_start[c] = p;
}

__global__ void cell_tail_kernel(int nall, int ncell, const unsigned *__restrict__ key_sorted,
                                 int *__restrict__ cell_start)
{
  const unsigned last = key_sorted[nall - 1];
  for (int c = (int) last + 1 + threadIdx.x; c <= ncell; c += blockDim.x) cell_start[c] = nall;
}

// one thread per atom, threads walk the atoms in cell order so a wave shares its stencil
template <bool FILL>
__global__ __launch_bounds__(256) void nbuild_kernel(const Grid g, const CutTables ct, int nall, int nlocal,
                                                     const double4 *__restrict__ xq, const int *__restrict__ perm,
                                                     const int *__restrict__ cell_start, int *__restrict__ cnt,
                                                     const long long *__restrict__ off, int *__restrict__ nb)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nall) return;
  const int i = perm[t];
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  const double *tab = (i < nlocal) ? (ct.g_owned ? ct.g_owned : ct.owned) : ct.ghost;
  if ((i >= nlocal && ct.ghost[0] <= 0.0) || (i < nlocal && ti < ct.min_type)) {
    if (!FILL) cnt[i] = 0;
    return;
  }
  int cx, cy, cz;
  cell_index(g, xi, cx, cy, cz);
  // ghosts only need the short (rcmax+skin) range: one cell around is enough when range >= 1
  int n = 0;
  int *row = FILL ? nb + off[i] : nullptr;
  const int R = g.range;
  for (int z = max(cz - R, 0); z <= min(cz + R, g.n[2] - 1); z++)
    for (int y = max(cy - R, 0); y <= min(cy + R, g.n[1] - 1); y++) {
      const int c0 = max(cx - R, 0) + g.n[0] * (y + g.n[1] * z);
      const int c1 = min(cx + R, g.n[0] - 1) + g.n[0] * (y + g.n[1] * z);
      const int pb = cell_start[c0], pe = cell_start[c1 + 1]; // cells along x are contiguous in the sort
      for (int p = pb; p < pe; p++) {
        const int j = perm[p];
        const double4 xj = xq[j];
        const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        const double rsq = dx * dx + dy * dy + dz * dz;
        if (rsq <= tab[ti * ct.ne + (int) xj.w] && j != i) {
          if (FILL) row[n] = j;
          n++;
        }
      }
    }
  if (!FILL) cnt[i] = n;
}

// Number of (i owned, j != i) pairs with rsq <= cutsq, both atoms of a mapped (non-NULL) type: what a plain
// geometric full list of the host holds for its owned rows.  rsq is formed exactly as LAMMPS' builders form it
// (delx*delx + dely*dely + delz*delz, no fused multiply-add), so the count is comparable entry for entry.
__global__ __launch_bounds__(256) void host_list_count_kernel(const Grid g, const CutTables ct, const int nall,
                                                              const int nlocal, const double4 *__restrict__ xq,
                                                              const int *__restrict__ perm,
                                                              const int *__restrict__ cell_start,
                                                              unsigned long long *__restrict__ total)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  unsigned long long n = 0;
  if (t < nall) {
    const int i = perm[t];
    const double4 xi = xq[i];
    if (i < nlocal && xi.w >= 0.0) {
      int cx, cy, cz;
      cell_index(g, xi, cx, cy, cz);
      const int R = g.range;
      for (int z = max(cz - R, 0); z <= min(cz + R, g.n[2] - 1); z++)
        for (int y = max(cy - R, 0); y <= min(cy + R, g.n[1] - 1); y++) {
          const int c0 = max(cx - R, 0) + g.n[0] * (y + g.n[1] * z);
          const int c1 = min(cx + R, g.n[0] - 1) + g.n[0] * (y + g.n[1] * z);
          const int pb = cell_start[c0], pe = cell_start[c1 + 1];
          for (int p = pb; p < pe; p++) {
            const int j = perm[p];
            const double4 xj = xq[j];
            const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            const double rsq = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
            const int tj = xj.w >= 0.0 ? (int) xj.w : 0;
            n += (rsq <= (ct.g_owned ? ct.g_owned : ct.owned)[(int) xi.w * ct.ne + tj] && j != i && xj.w >= 0.0) ? 1u : 0u;
          }
        }
    }
  }
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

// The same list for a short list of atoms (AEAM: the angular centres, 0.75 % of the atoms in sample.in): one WAVE per
// atom, lanes stride over the stencil rows, ballot compaction -- entries come out in exactly the order the
// thread-per-atom kernel above writes them (z, y, then position in the sorted row).  With one active lane per wave
// the kernel above took 1.3 ms of a 3.5 ms reneighboring at a million atoms.
template <bool FILL>
__global__ __launch_bounds__(256) void nbuild_list_kernel(const Grid g, const CutTables ct, const int nlist,
                                                          const int *__restrict__ list, const double4 *__restrict__ xq,
                                                          const int *__restrict__ perm,
                                                          const int *__restrict__ cell_start, int *__restrict__ cnt,
                                                          const long long *__restrict__ off, int *__restrict__ nb)
{
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= nlist) return;
  const int i = list[w];
  const double4 xi = xq[i];
  const int ti = (int) xi.w;
  int cx, cy, cz;
  cell_index(g, xi, cx, cy, cz);
  int n = 0;
  int *row = FILL ? nb + off[i] : nullptr;
  const int R = g.range;
  for (int z = max(cz - R, 0); z <= min(cz + R, g.n[2] - 1); z++)
    for (int y = max(cy - R, 0); y <= min(cy + R, g.n[1] - 1); y++) {
      const int c0 = max(cx - R, 0) + g.n[0] * (y + g.n[1] * z);
      const int c1 = min(cx + R, g.n[0] - 1) + g.n[0] * (y + g.n[1] * z);
      const int pb = cell_start[c0], pe = cell_start[c1 + 1];
      for (int p0 = pb; p0 < pe; p0 += 64) { // wave-uniform trip count
        const int p = p0 + lane;
        bool hit = false;
        int j = 0;
        if (p < pe) {
          j = perm[p];
          const double4 xj = xq[j];
          const double dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
          const double rsq = dx * dx + dy * dy + dz * dz;
          hit = rsq <= (ct.g_owned ? ct.g_owned : ct.owned)[ti * ct.ne + (int) xj.w] && j != i;
        }
        const unsigned long long bal = __ballot(hit);
        if (FILL && hit) row[n + __popcll(bal & ((1ull << lane) - 1ull))] = j;
        n += __popcll(bal);
      }
    }
  if (!FILL && lane == 0) cnt[i] = n;
}

// a workgroup looks at kAngChunk atoms and sends ONE atomic for them: at 0.76 % Si one per wave still meant ~7 000
// atomics on one word, 70 us of a reneighboring (they are served one after the other)
constexpr int kAngChunk = 4096;
__global__ __launch_bounds__(256) void ang_select_kernel(const int nlocal, const int min_type,
                                                         const double4 *__restrict__ xq, int *__restrict__ list,
                                                         int *__restrict__ count)
{
  __shared__ int s_n, s_base, s_idx[kAngChunk];
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int i0 = blockIdx.x * kAngChunk;
  for (int k = threadIdx.x; k < kAngChunk; k += 256) {
    const int i = i0 + k;
    if (i < nlocal && (int) xq[i].w >= min_type) s_idx[atomicAdd(&s_n, 1)] = i;
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_n) s_base = atomicAdd(count, s_n);
  __syncthreads();
  for (int k = threadIdx.x; k < s_n; k += 256) list[s_base + k] = s_idx[k];
}

// fix nve on the device.  FINAL: the final_integrate of the step just finished and the initial_integrate of the next
// one in one pass (both half-kicks use the same forces) -- saves reading f, v, rmass and writing v once per step; the
// same operations in the same order as the two kernels, so the trajectory is bit-identical.  CHECK: `neigh_modify
// check yes` of the new positions against the positions of the last reneighboring in the same pass (what
// dd_moved_kernel does in a pass of its own): flag[0] = some atom beyond the trigger, flag[1] = beyond half the skin.
// SC: the style-level checks and the accumulator reset of the compute that follows, in the same pass (MdpStyleCheck).
template <bool FINAL, bool CHECK>
__global__ void nve_advance_kernel(int nlocal, double dtf, double dt, const double *__restrict__ rmass,
                                   double *__restrict__ f, double *__restrict__ v, double4 *__restrict__ xq,
                                   const mdp_hold_t *__restrict__ xhold, const double trigsq, const double hardsq,
                                   int *__restrict__ flag, const MdpStyleCheck SC, const int zero_f,
                                   double *__restrict__ dflag_set = nullptr, double *__restrict__ dflag_clear = nullptr)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (CHECK && dflag_clear && i == 0) *dflag_clear = 0.0; // (the word of the next step; this step's was cleared a step ago)
  bool t = false, h = false, sa = false, sah = false, sp = false, sph = false;
  if (SC.acc) { // what acc_zero_kernel does
    for (int k = i; k < SC.nacc; k += gridDim.x * 256) SC.acc[k] = 0.0;
    if (i == 0) {
      const int f0 = SC.flags[0];
      if (f0) SC.flags[4] |= f0;
      SC.flags[0] = 0;
      if (SC.ovf)
        for (int k = 0; k < MDP_NOVF_LISTS; k++) SC.ovf[(size_t) k * SC.ovf_stride] = 0;
    } else if (i < 4)
      SC.flags[i] = 0;
  }
  if (i < nlocal) {
    const double s = dtf / rmass[i];
    const double fx = f[3 * (size_t) i], fy = f[3 * (size_t) i + 1], fz = f[3 * (size_t) i + 2];
    if (zero_f) { // the forces have had their last reader: force_clear of the next compute (a style that accumulates)
      f[3 * (size_t) i] = 0.0;
      f[3 * (size_t) i + 1] = 0.0;
      f[3 * (size_t) i + 2] = 0.0;
    }
    double vx = v[3 * (size_t) i], vy = v[3 * (size_t) i + 1], vz = v[3 * (size_t) i + 2];
    if (FINAL) { // final_integrate (step n)
      vx += s * fx;
      vy += s * fy;
      vz += s * fz;
    }
    vx = vx + s * fx; // initial_integrate (step n+1)
    vy = vy + s * fy;
    vz = vz + s * fz;
    v[3 * (size_t) i] = vx;
    v[3 * (size_t) i + 1] = vy;
    v[3 * (size_t) i + 2] = vz;
    double4 x = xq[i];
    x.x += dt * vx;
    x.y += dt * vy;
    x.z += dt * vz;
    xq[i] = x;
    // The triggers are read one compute late and fire early by a fixed margin that stands for two steps of motion (0.07 /
    // 0.1 A: 35 / 50 A/ps at 1 fs).  An atom faster than that -- the tail of a 5 000 K melt -- fires by its OWN two steps
    // instead (this step's speed, a quarter on top for its acceleration): whatever it may reach before the answer is
    // read is then still inside the limit.  (profiles/prune_fuzz.py found the case.)
    const double two_steps = (CHECK || SC.xa || SC.xp) ? 2.5 * dt * sqrt(vx * vx + vy * vy + vz * vz) : 0.0;
    auto reaches = [two_steps](const double d2, const double hardsq_) { // will the atom be beyond `hard` two steps on?
      const double rem = sqrt(hardsq_) - two_steps;
      return rem <= 0.0 || d2 > rem * rem;
    };
    if (CHECK) {
      const double dx = x.x - xhold[3 * (size_t) i], dy = x.y - xhold[3 * (size_t) i + 1], dz = x.z - xhold[3 * (size_t) i + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      t = d2 > trigsq || reaches(d2, hardsq);
      h = d2 > hardsq;
    }
    if (SC.xa) {
      const double dx = x.x - SC.xa[3 * (size_t) i], dy = x.y - SC.xa[3 * (size_t) i + 1], dz = x.z - SC.xa[3 * (size_t) i + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      sa = d2 > SC.trig_a || reaches(d2, SC.hard_a);
      sah = d2 > SC.hard_a;
    }
    if (SC.xp) {
      const double dx = x.x - SC.xp[3 * (size_t) i], dy = x.y - SC.xp[3 * (size_t) i + 1], dz = x.z - SC.xp[3 * (size_t) i + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      sp = d2 > SC.trig_p || reaches(d2, SC.hard_p);
      sph = d2 > SC.hard_p;
    }
  }
  if (CHECK) { // (pinned host words zeroed by the host before the launch: plain idempotent stores)
    if (__ballot(t) && (threadIdx.x & 63) == 0) {
      flag[0] = 1;
      if (dflag_set) *dflag_set = 1.0; // the same answer for the peers: travels with this step's halo
    }
    if (__ballot(h) && (threadIdx.x & 63) == 0) flag[1] = 1;
  }
  if (SC.flag) { // (the wave votes with ALL its lanes, then lane 0 stores: a vote inside the lane-0 branch would count lane 0 alone)
    const bool wa = __any(sa), wah = __any(sah), wp = __any(sp), wph = __any(sph);
    if ((threadIdx.x & 63) == 0) {
      if (wa) SC.flag[0] = 1;
      if (wah) SC.flag[1] = 1;
      if (wp) SC.flag[2] = 1;
      if (wph) SC.flag[3] = 1;
    }
  }
}

// mass of every owned atom in the device's atom order (host mode: type[] is in the host's order, perm maps positions)
__global__ void hn_rmass_kernel(const int nlocal, const int *__restrict__ type, const int *__restrict__ perm,
                                const double *__restrict__ mass_type, double *__restrict__ rmass)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < nlocal) rmass[p] = mass_type[type[perm ? perm[p] : p] & 15];
}

__global__ void nve_final_kernel(int nlocal, double dtf, const double *__restrict__ rmass,
                                 const double *__restrict__ f, double *__restrict__ v)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nlocal) return;
  const double s = dtf / rmass[i];
  v[3 * (size_t) i] += s * f[3 * (size_t) i];
  v[3 * (size_t) i + 1] += s * f[3 * (size_t) i + 1];
  v[3 * (size_t) i + 2] += s * f[3 * (size_t) i + 2];
}

__global__ void ghost_refresh_kernel(int nlocal, int nghost, const int *__restrict__ owner,
                                     const double *__restrict__ shift, double4 *__restrict__ xq,
                                     double *__restrict__ fzero = nullptr)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nghost) return;
  if (fzero) { // force_clear of the compute that follows, for the images (the integrate kernel does the owned atoms)
    fzero[3 * (size_t) (nlocal + g)] = 0.0;
    fzero[3 * (size_t) (nlocal + g) + 1] = 0.0;
    fzero[3 * (size_t) (nlocal + g) + 2] = 0.0;
  }
  const int o = owner[g];
  if (o < 0) return;
  const double4 xo = xq[o];
  double4 x = xq[nlocal + g];
  x.x = xo.x + shift[3 * (size_t) g];
  x.y = xo.y + shift[3 * (size_t) g + 1];
  x.z = xo.z + shift[3 * (size_t) g + 2];
  xq[nlocal + g] = x;
}

__global__ void ghost_scalar_refresh_kernel(int nlocal, int nghost, const int *__restrict__ owner,
                                            double *__restrict__ a)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nghost) return;
  const int o = owner[g];
  if (o >= 0) a[nlocal + g] = a[o];
}

__global__ void fold_self_ghost_f_kernel(int nlocal, int nghost, const int *__restrict__ owner, double *__restrict__ f)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nghost) return;
  const int o = owner[g];
  if (o < 0) return;
  double *fg = f + 3 * (size_t) (nlocal + g);
  if (fg[0] != 0.0 || fg[1] != 0.0 || fg[2] != 0.0) {
    atomicAdd(&f[3 * (size_t) o], fg[0]);
    atomicAdd(&f[3 * (size_t) o + 1], fg[1]);
    atomicAdd(&f[3 * (size_t) o + 2], fg[2]);
    fg[0] = fg[1] = fg[2] = 0.0;
  }
}

__global__ void hold_kernel(int nlocal, const double4 *__restrict__ xq, mdp_hold_t *__restrict__ xhold)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nlocal) return;
  const double4 x = xq[i];
  xhold[3 * (size_t) i] = (mdp_hold_t) x.x;
  xhold[3 * (size_t) i + 1] = (mdp_hold_t) x.y;
  xhold[3 * (size_t) i + 2] = (mdp_hold_t) x.z;
}

// out[7] += KE, out[8] = max(out[8], disp^2)   (acc[7], acc[8])
__global__ __launch_bounds__(256) void thermo_kernel(int nlocal, double half_mvv2e, const double *__restrict__ rmass,
                                                     const double *__restrict__ v, const double4 *__restrict__ xq,
                                                     const mdp_hold_t *__restrict__ xhold, double *__restrict__ acc)
{
  double ke = 0.0, d2 = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nlocal; i += gridDim.x * 256) {
    const double vx = v[3 * (size_t) i], vy = v[3 * (size_t) i + 1], vz = v[3 * (size_t) i + 2];
    ke += rmass[i] * (vx * vx + vy * vy + vz * vz);
    const double4 x = xq[i];
    const double dx = x.x - xhold[3 * (size_t) i], dy = x.y - xhold[3 * (size_t) i + 1],
                 dz = x.z - xhold[3 * (size_t) i + 2];
    d2 = fmax(d2, dx * dx + dy * dy + dz * dz);
  }
  for (int o = 32; o > 0; o >>= 1) {
    ke += __shfl_xor(ke, o, 64);
    d2 = fmax(d2, __shfl_xor(d2, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&acc[7], half_mvv2e * ke);
    // d2 >= 0: compare as integers
    atomicMax((unsigned long long *) &acc[8], (unsigned long long) __double_as_longlong(d2));
  }
}

__global__ void pack_x_kernel(int n, const int *__restrict__ sendlist, const double *__restrict__ shift,
                              const double4 *__restrict__ xq, double *__restrict__ buf)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const double4 x = xq[sendlist[k]];
  buf[3 * (size_t) k] = x.x + (shift ? shift[3 * (size_t) k] : 0.0);
  buf[3 * (size_t) k + 1] = x.y + (shift ? shift[3 * (size_t) k + 1] : 0.0);
  buf[3 * (size_t) k + 2] = x.z + (shift ? shift[3 * (size_t) k + 2] : 0.0);
}

// SC: the style-level displacement checks of the arriving remote ghosts (MdpStyleCheck; flag words [4..7])
__global__ void unpack_x_kernel(int n, int first, const double *__restrict__ buf, double4 *__restrict__ xq,
                                const MdpStyleCheck SC, const double *__restrict__ gflag = nullptr, const int ngflag = 0,
                                int *__restrict__ h_glob = nullptr, double *__restrict__ fzero = nullptr)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (h_glob && k == 0) { // the ranks' "moved" words as gathered behind this halo -> one pinned word (MdpDomain::flagbuf)
    double m = 0.0;
    for (int q = 0; q < ngflag; q++) m = gflag[q] > m ? gflag[q] : m;
    *h_glob = m > 0.0 ? 1 : 0;
  }
  bool sa = false, sah = false, sp = false, sph = false;
  if (k < n) {
    const size_t i = (size_t) first + k;
    double4 x = xq[i];
    // (what the slot held is the ghost's position of the step before: its last step is its speed -- see nve_advance_kernel)
    const double ox = buf[3 * (size_t) k] - x.x, oy = buf[3 * (size_t) k + 1] - x.y, oz = buf[3 * (size_t) k + 2] - x.z;
    const double two_steps = (SC.xa || SC.xp) ? 2.5 * sqrt(ox * ox + oy * oy + oz * oz) : 0.0;
    auto reaches = [two_steps](const double d2, const double hardsq_) {
      const double rem = sqrt(hardsq_) - two_steps;
      return rem <= 0.0 || d2 > rem * rem;
    };
    x.x = buf[3 * (size_t) k];
    x.y = buf[3 * (size_t) k + 1];
    x.z = buf[3 * (size_t) k + 2];
    xq[i] = x;
    if (fzero) fzero[3 * i] = fzero[3 * i + 1] = fzero[3 * i + 2] = 0.0; // (force_clear of a style that accumulates, see mdp_md_advance)
    if (SC.xa) {
      const double dx = x.x - SC.xa[3 * i], dy = x.y - SC.xa[3 * i + 1], dz = x.z - SC.xa[3 * i + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      sa = d2 > SC.trig_a || reaches(d2, SC.hard_a);
      sah = d2 > SC.hard_a;
    }
    if (SC.xp) {
      const double dx = x.x - SC.xp[3 * i], dy = x.y - SC.xp[3 * i + 1], dz = x.z - SC.xp[3 * i + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      sp = d2 > SC.trig_p || reaches(d2, SC.hard_p);
      sph = d2 > SC.hard_p;
    }
  }
  if (SC.flag) { // (all lanes vote, lane 0 stores -- see nve_advance_kernel)
    const bool wa = __any(sa), wah = __any(sah), wp = __any(sp), wph = __any(sph);
    if ((threadIdx.x & 63) == 0) {
      if (wa) SC.flag[4] = 1;
      if (wah) SC.flag[5] = 1;
      if (wp) SC.flag[6] = 1;
      if (wph) SC.flag[7] = 1;
    }
  }
}

__global__ void pack_scalar_kernel(int n, const int *__restrict__ sendlist, const double *__restrict__ a,
                                   double *__restrict__ buf)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) buf[k] = a[sendlist[k]];
}

__global__ void copy_kernel(int n, const double *__restrict__ src, double *__restrict__ dst)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) dst[k] = src[k];
}

__global__ void unpack_add_f_kernel(int n, const int *__restrict__ sendlist, const double *__restrict__ buf,
                                    double *__restrict__ f)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int i = sendlist[k];
  // several ghosts (images) may map to one owner
  atomicAdd(&f[3 * (size_t) i], buf[3 * (size_t) k]);
  atomicAdd(&f[3 * (size_t) i + 1], buf[3 * (size_t) k + 1]);
  atomicAdd(&f[3 * (size_t) i + 2], buf[3 * (size_t) k + 2]);
}

__global__ void xq_to_x3_kernel(int n, const double4 *__restrict__ xq, double *__restrict__ x3)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double4 x = xq[i];
  x3[3 * (size_t) i] = x.x;
  x3[3 * (size_t) i + 1] = x.y;
  x3[3 * (size_t) i + 2] = x.z;
}

__global__ void x3_to_xq_kernel(int n, const double *__restrict__ x3, double4 *__restrict__ xq)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double4 x = xq[i];
  x.x = x3[3 * (size_t) i];
  x.y = x3[3 * (size_t) i + 1];
  x.z = x3[3 * (size_t) i + 2];
  xq[i] = x;
}

inline int nblk(long long n) { return (int) ((n + 255) / 256); }

} // namespace

// ---- neighbor-list cutoffs the style's init_one() would hand the host ----------------------------
// aeam: squared list cutoffs (cut[ti][tj] + skin)^2 per type pair (pair_aeam.cpp:618-620), in the kernel arguments up
// to MDP_AEAM_MAXT types, in device memory beyond
static int aeam_cut_table(mdp_ctx *c, CutTables &ct, const double skin, double &maxcut)
{
  const int nt = c->aeam.ntypes;
  ct.ne = nt;
  ct.g_owned = nullptr;
  std::vector<double> h((size_t) nt * nt);
  for (int a = 0; a < nt; a++)
    for (int b = 0; b < nt; b++) {
      const double cc = c->aeam_hcut[a * nt + b] + skin;
      h[(size_t) a * nt + b] = cc * cc;
      if (cc > maxcut) maxcut = cc;
    }
  if (nt <= MDP_AEAM_MAXT) {
    for (int k = 0; k < nt * nt; k++) ct.owned[k] = h[k];
    return MDP_OK;
  }
  MDP_HIP(c, c->cut_tab.reserve(h.size() + 1));
  MDP_HIP(c, hipMemcpyAsync(c->cut_tab.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  ct.g_owned = c->cut_tab.p;
  return MDP_OK;
}

static int md_cut_tables(mdp_ctx *c, CutTables &ct, double &maxcut)
{
  memset(&ct, 0, sizeof ct);
  maxcut = 0.0;
  const double skin = c->cfg.skin;
  if (c->cfg.style == 1) {
    if (!c->have_rebomos) return mdp_fail(c, MDP_ESTATE, "rebomos parameters not set");
    ct.ne = 2;
    const double cut3 = 3.0 * c->rebomos_host.rcmax[0][0] + skin; // pair_rebomos.cpp:257 (+ skin)
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++) {
        ct.owned[a * 2 + b] = cut3 * cut3;
        const double cg = c->rebomos_host.rcmax[a][b] + skin;     // cutghost, pair_rebomos.cpp:261
        ct.ghost[a * 2 + b] = cg * cg;
      }
    maxcut = cut3;
  } else if (c->cfg.style == 2) {
    if (!c->have_aeam) return mdp_fail(c, MDP_ESTATE, "aeam tables not set");
    const int nt = c->aeam.ntypes;
    ct.ne = nt;
    // With tile lists (resident mode) the metal atoms never touch the CSR list: force-only AND energy /
    // virial steps run the tile kernels.  Only the angular centres (0.75 % in sample.in) read it, so only their
    // rows are built -- a quarter of the reneighboring time at 1 M atoms.  A per-atom-virial step (CSR kernels)
    // asks for the full list (c->csr_want_full) and gets it rebuilt on the spot.
    const char *e = getenv("MDP_AEAM_TILE");
    const bool tiles = (c->md || c->aeam_device_lists) && !(e && atoi(e) == 0) && nt <= MDP_AEAM_MAXT;
    ct.min_type = (tiles && !c->csr_want_full && !c->cfg.master_list) ? c->aeam.nnonangular : 0;
    c->csr_full = ct.min_type == 0;
    MDP_TRY(aeam_cut_table(c, ct, skin, maxcut));
  } else
    return mdp_fail(c, MDP_EINVAL, "unknown style %d", c->cfg.style);
  return MDP_OK;
}

// bins of width >= cutoff/2 over [lo,hi], atoms sorted by bin (cell_perm), bin boundaries (cell_start)
int mdp_bin_atoms(mdp_ctx *c, double cutoff, const double lo[3], const double hi[3])
{
  const int nall = c->nall;
  hipStream_t st = c->stream;
  Grid &g = c->grid;
  if (nall <= 0) { // an empty sub-domain has nothing to bin
    for (int d = 0; d < 3; d++) {
      g.n[d] = 1;
      g.lo[d] = 0.0;
      g.inv[d] = 1.0;
    }
    g.range = 2;
    MDP_HIP(c, c->cell_start.reserve(4));
    MDP_HIP(c, c->cell_perm.reserve(4));
    MDP_HIP(c, hipMemsetAsync(c->cell_start.p, 0, sizeof(int) * 2, st));
    return MDP_OK;
  }
  const double binsize = 0.5 * cutoff; // LAMMPS default: half the cutoff (log.rebomos-bulk.1:45)
  g.range = 2;
  long long ncell = 1;
  for (int d = 0; d < 3; d++) {
    const double len = hi[d] - lo[d];
    if (!(len > 0.0)) return mdp_fail(c, MDP_EINVAL, "empty bounding box");
    int n = (int) floor(len / binsize);
    if (n < 1) n = 1;
    if (n > 1024) n = 1024; // wider cells are still correct
    g.n[d] = n;
    g.lo[d] = lo[d];
    g.inv[d] = n / len; // cells are >= binsize wide
    ncell *= n;
  }
  MDP_HIP(c, c->sort_keys_a.reserve(nall + 1));
  MDP_HIP(c, c->sort_keys_b.reserve(nall + 1));
  MDP_HIP(c, c->cell_of.reserve(nall + 1));   // values in
  MDP_HIP(c, c->cell_perm.reserve(nall + 1)); // values out
  MDP_HIP(c, c->cell_start.reserve((size_t) ncell + 2));
  cell_assign_kernel<<<nblk(nall), 256, 0, st>>>(g, nall, c->xq.p, c->sort_keys_a.p, c->cell_of.p);
  MDP_HIP(c, hipGetLastError());
  int bits = 1;
  while ((1ll << bits) < ncell) bits++;
  size_t tmp = 0;
  MDP_HIP(c, rocprim::radix_sort_pairs(nullptr, tmp, c->sort_keys_a.p, c->sort_keys_b.p, c->cell_of.p, c->cell_perm.p,
                                       (size_t) nall, 0, bits, st));
  MDP_HIP(c, c->scan_tmp.reserve(tmp + 16));
  MDP_HIP(c, rocprim::radix_sort_pairs(c->scan_tmp.p, tmp, c->sort_keys_a.p, c->sort_keys_b.p, c->cell_of.p,
                                       c->cell_perm.p, (size_t) nall, 0, bits, st));
  MDP_HIP(c, c->xq_cell.reserve((size_t) nall + 1));
  cell_bounds_kernel<<<nblk(nall), 256, 0, st>>>(nall, c->sort_keys_b.p, c->cell_start.p, c->cell_perm.p, c->xq.p,
                                                 c->xq_cell.p);
  cell_tail_kernel<<<1, 256, 0, st>>>(nall, (int) ncell, c->sort_keys_b.p, c->cell_start.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_build_master_list(mdp_ctx *c)
{
  const int nall = c->nall, nlocal = c->nlocal;
  hipStream_t st = c->stream;
  CutTables ct;
  double maxcut;
  MDP_TRY(md_cut_tables(c, ct, maxcut));
  MDP_TRY(mdp_bin_atoms(c, maxcut, c->cfg.bbox_lo, c->cfg.bbox_hi));
  const Grid g = c->grid;
  MDP_HIP(c, c->nb_cnt.reserve(nall + 2));
  MDP_HIP(c, c->nb_off.reserve(nall + 2));
  // rows for a few atoms only (AEAM with tile lists: the angular centres): a wave per listed atom
  const bool few = ct.min_type > 0 && ct.ghost[0] <= 0.0;
  int nsel = 0;
  if (few) {
    MDP_HIP(c, c->ang_list.reserve(nlocal + 1));
    MDP_HIP(c, c->ang_count.reserve(4));
    MDP_HIP(c, hipMemsetAsync(c->ang_count.p, 0, sizeof(int), st));
    MDP_HIP(c, hipMemsetAsync(c->nb_cnt.p, 0, sizeof(int) * (nall + 1), st));
    if (nlocal)
      ang_select_kernel<<<(nlocal + kAngChunk - 1) / kAngChunk, 256, 0, st>>>(nlocal, ct.min_type, c->xq.p, c->ang_list.p,
                                                                             c->ang_count.p);
    MDP_TRY(mdp_read_one(c, c->ang_count.p, sizeof(int), &nsel));
    // (aeam: these ARE the owned angular centres, min_type = nnonangular -- mdp_aeam_prepare takes the list as it is)
    c->ang_list_n = c->cfg.style == 2 && ct.min_type == c->aeam.nnonangular ? nsel : -1;
    if (nsel)
      nbuild_list_kernel<false><<<(nsel + 3) / 4, 256, 0, st>>>(g, ct, nsel, c->ang_list.p, c->xq.p, c->cell_perm.p,
                                                                c->cell_start.p, c->nb_cnt.p, nullptr, nullptr);
  } else {
    c->ang_list_n = -1;
    nbuild_kernel<false><<<nblk(nall), 256, 0, st>>>(g, ct, nall, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p,
                                                      c->nb_cnt.p, nullptr, nullptr);
  }
  MDP_HIP(c, hipGetLastError());
  MDP_TRY(mdp_scan_exclusive_i64(c, c->nb_cnt.p, c->nb_off.p, nall));
  long long tot[2] = {0, 0};
  {
    const MdpRead rd[2] = {{c->nb_off.p + nall, sizeof(long long), &tot[0]}, {c->nb_off.p + nlocal, sizeof(long long), &tot[1]}};
    MDP_TRY(mdp_read_small(c, rd, 2));
  }
  c->nb_total = tot[0];
  c->nb_owned_total = tot[1];
  MDP_HIP(c, c->nb.reserve((size_t) tot[0] + 1));
  if (few) {
    if (nsel)
      nbuild_list_kernel<true><<<(nsel + 3) / 4, 256, 0, st>>>(g, ct, nsel, c->ang_list.p, c->xq.p, c->cell_perm.p,
                                                               c->cell_start.p, nullptr, c->nb_off.p, c->nb.p);
  } else
    nbuild_kernel<true><<<nblk(nall), 256, 0, st>>>(g, ct, nall, nlocal, c->xq.p, c->cell_perm.p, c->cell_start.p,
                                                     nullptr, c->nb_off.p, c->nb.p);
  MDP_HIP(c, hipGetLastError());
  c->skin = c->cfg.skin;
  c->neigh_set = true;
  c->rebo_packed = false;
  return MDP_OK;
}

extern "C" {

int mdp_md_setup(mdp_ctx *c, const mdp_md_config *cfg, const double *x, const double *v, const int *type,
                 const int *tag, const double *mass, const int *map, const int *ghost_owner, const double *ghost_shift,
                 const int *ghost_type, const int *ghost_tag)
{
  if (!c || !cfg || !x || !type || !mass) return MDP_EINVAL;
  if (cfg->nlocal < 0 || cfg->nghost < 0 || cfg->ntypes < 1 || cfg->ntypes > 15)
    return mdp_fail(c, MDP_EINVAL, "mdp_md_setup: bad sizes");
  if (cfg->nghost > 0 && (!ghost_owner || !ghost_shift || !ghost_type))
    return mdp_fail(c, MDP_EINVAL, "mdp_md_setup: ghost arrays missing");
  MDP_HIP(c, hipSetDevice(c->device));
  c->cfg = *cfg;
  const int nlocal = cfg->nlocal, nghost = cfg->nghost, nall = nlocal + nghost;
  if ((long long) nall >= (1ll << 29)) return mdp_fail(c, MDP_EINVAL, "too many atoms for NEIGHMASK");
  // host-side assembly of the [nall] arrays, then the same upload path as host mode
  std::vector<double> xa((size_t) 3 * nall);
  std::vector<int> ta(nall), ga(nall, 0);
  memcpy(xa.data(), x, sizeof(double) * 3 * nlocal);
  memcpy(ta.data(), type, sizeof(int) * nlocal);
  if (tag) memcpy(ga.data(), tag, sizeof(int) * nlocal);
  for (int g = 0; g < nghost; g++) {
    const int o = ghost_owner[g];
    if (o >= nlocal) return mdp_fail(c, MDP_EINVAL, "ghost owner %d out of range", o);
    for (int d = 0; d < 3; d++)
      xa[3 * (size_t) (nlocal + g) + d] = (o >= 0 ? x[3 * (size_t) o + d] : 0.0) + ghost_shift[3 * (size_t) g + d];
    ta[nlocal + g] = ghost_type[g];
    if (ghost_tag) ga[nlocal + g] = ghost_tag[g];
  }
  c->md = true; // before the upload: resident atoms keep the caller's order and their staging arrays are temporaries
  MDP_TRY(mdp_set_atoms_host(c, nlocal, nghost, xa.data(), ta.data(), ga.data(), cfg->ntypes, map));
  for (int d = 0; d < 3; d++) { // resident mode bins over the caller's box
    c->bbox_lo[d] = cfg->bbox_lo[d];
    c->bbox_hi[d] = cfg->bbox_hi[d];
  }
  hipStream_t st = c->stream;
  MDP_HIP(c, c->v.reserve((size_t) 3 * nlocal + 3));
  MDP_HIP(c, c->xhold.reserve((size_t) 3 * nlocal + 3));
  MDP_HIP(c, c->rmass.reserve(nlocal + 1));
  MDP_HIP(c, c->ghost_owner.reserve(nghost + 1));
  MDP_HIP(c, c->ghost_shift.reserve((size_t) 3 * nghost + 3));
  MDP_HIP(c, c->rho.reserve(nall + 1));
  MDP_HIP(c, c->fp.reserve(nall + 1));
  for (int t = 0; t < 16; t++) c->h_mass[t] = t <= cfg->ntypes ? mass[t] : 0.0; // migrated atoms look their mass up
  std::vector<double> rm(nlocal);
  for (int i = 0; i < nlocal; i++) {
    if (type[i] < 1 || type[i] > cfg->ntypes) return mdp_fail(c, MDP_EINVAL, "atom type out of range");
    rm[i] = mass[type[i]];
  }
  if (nlocal) MDP_HIP(c, hipMemcpyAsync(c->rmass.p, rm.data(), sizeof(double) * nlocal, hipMemcpyHostToDevice, st));
  if (v && nlocal)
    MDP_HIP(c, hipMemcpyAsync(c->v.p, v, sizeof(double) * 3 * nlocal, hipMemcpyHostToDevice, st));
  else
    MDP_HIP(c, hipMemsetAsync(c->v.p, 0, sizeof(double) * 3 * nlocal, st));
  if (nghost) {
    MDP_HIP(c, hipMemcpyAsync(c->ghost_owner.p, ghost_owner, sizeof(int) * nghost, hipMemcpyHostToDevice, st));
    MDP_HIP(c, hipMemcpyAsync(c->ghost_shift.p, ghost_shift, sizeof(double) * 3 * nghost, hipMemcpyHostToDevice, st));
  }
  MDP_HIP(c, hipMemsetAsync(c->f.p, 0, sizeof(double) * 3 * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->eatom.p, 0, sizeof(double) * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->fp.p, 0, sizeof(double) * nall, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  c->remote_start = nlocal + (cfg->nghost_self >= 0 && cfg->nghost_self <= nghost ? cfg->nghost_self : nghost);
  // (ghosts below remote_start are refreshed from -- and, for rebomos, computed through -- their owners)
  for (int g = 0; g < c->remote_start - nlocal; g++)
    if (ghost_owner[g] < 0 || ghost_owner[g] >= nlocal)
      return mdp_fail(c, MDP_EINVAL, "mdp_md_setup: ghost %d is declared a periodic self-image (nghost_self = %d) but has no owner",
                      g, c->remote_start - nlocal);
  c->md = true;
  return MDP_OK;
}

int mdp_md_build_neighbors(mdp_ctx *c) { return mdp_md_build_neighbors_impl(c); }

} // extern "C"

int mdp_md_build_neighbors_impl(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  MDP_HIP(c, hipSetDevice(c->device));
  if (c->cfg.style == 2 || c->cfg.master_list) {
    MDP_TRY(mdp_md_build_master_list(c));
  } else {
    c->nb_total = c->nb_owned_total = 0;
    c->skin = c->cfg.skin;
    c->neigh_set = true; // rebomos builds its own lists from the bin grid
  }
  hold_kernel<<<nblk(c->nlocal), 256, 0, c->stream>>>(c->nlocal, c->xq.p, c->xhold.p);
  MDP_HIP(c, hipGetLastError());
  if (c->cfg.style == 1) {
    c->rebo_packed = false;
    return mdp_rebomos_repack(c);
  }
  return mdp_aeam_prepare(c);
}

// ---- style-level displacement checks fused into the integrate kernel and the halo unpack (MdpStyleCheck) -----------
void mdp_sflag_arm(mdp_ctx *c, MdpStyleCheck &sc)
{
  sc = MdpStyleCheck();
  c->sflag_set ^= 1;
  const int set = c->sflag_set;
  int *h = (int *) (c->h_pinned + 32) + 8 * set; // (the kernels that wrote this set ran two steps ago)
  for (int k = 0; k < 8; k++) h[k] = 0;
  MdpStyleCheckMeta &m = c->sflag_meta[set];
  m = MdpStyleCheckMeta();
  const double scale = mdp_margin_scale(c);
  // (rebomos: resident runs, and host mode with the integrator on the device -- mdp_hnve_initial)
  const bool rebo = c->cfg.style == 1 || (!c->md && c->have_rebomos && !c->have_aeam);
  if (rebo && c->rebo_packed && !c->check_now && c->xhold_all.p && c->skin_inner > 0.0) {
    double trig = 0.5 * c->skin_inner - kStaleMargin * scale;
    if (trig < 0.25 * c->skin_inner) trig = 0.25 * c->skin_inner;
    const double hard = 0.5 * c->skin_inner;
    sc.xa = c->xhold_all.p;
    sc.trig_a = trig * trig;
    sc.hard_a = hard * hard;
    m.has_style = true;
    m.build_epoch = c->style_builds;
  }
  if (c->prune_valid && c->xhold_prune.p && !c->check_now) {
    double ptrig = 0.5 * c->prune_buf - kPruneMargin * scale;
    if (ptrig < 0.25 * c->prune_buf) ptrig = 0.25 * c->prune_buf;
    const double phard = 0.5 * c->prune_buf;
    sc.xp = c->xhold_prune.p;
    sc.trig_p = ptrig * ptrig;
    sc.hard_p = phard * phard;
    m.has_prune = true;
    m.prune_epoch = c->prune_epoch;
  }
  sc.flag = (m.has_style || m.has_prune) ? h : nullptr;
  if (c->md && c->neigh_set && c->acc.p && c->flags.p) { // the compute that follows finds its accumulators reset
    sc.acc = c->acc.p;
    sc.nacc = MDP_ACC_STRIDE * (1 + MDP_ACC_SLOTS);
    sc.flags = c->flags.p;
    sc.ovf = c->ovf.p;
    sc.ovf_stride = c->ovf_stride;
  }
  c->sflag_chk = sc;
  c->sflag_armed = sc.flag != nullptr;
  c->sflag_committed[set] = false; // (what this set held was collected a step ago, or never will be)
}

int mdp_sflag_commit(mdp_ctx *c)
{
  if (!c->sflag_armed) return MDP_OK;
  const int set = c->sflag_set;
  if (!c->ev_sflag[set]) MDP_HIP(c, hipEventCreateWithFlags(&c->ev_sflag[set], hipEventDisableTiming));
  MDP_HIP(c, hipEventRecord(c->ev_sflag[set], c->stream));
  c->sflag_committed[set] = true;
  return MDP_OK;
}

void mdp_sflag_drop(mdp_ctx *c) { c->sflag_committed[0] = c->sflag_committed[1] = false; }

// the words of the step before (their last writer was queued a whole compute ago).  *far / *toofar: the style lists'
// trigger / half their skin, valid only if no list build happened since the check was armed; the pruning part is
// applied here (prune_stale, dangerous_prunes).
int mdp_sflag_collect(mdp_ctx *c, bool *far, bool *toofar)
{
  if (far) *far = false;
  if (toofar) *toofar = false;
  // the set of the step BEFORE the one whose integrate kernel was queued last: its writers ran a whole compute ago
  // (the set being written now is read by the next compute -- the triggers fire early by kStaleMargin / kPruneMargin
  // for exactly this one compute of delay)
  const int set = c->sflag_set ^ 1;
  if (!c->sflag_committed[set]) return MDP_OK;
  MDP_HIP(c, hipEventSynchronize(c->ev_sflag[set]));
  c->sflag_committed[set] = false;
  const int *h = (const int *) (c->h_pinned + 32) + 8 * set;
  const MdpStyleCheckMeta &m = c->sflag_meta[set];
  if (m.has_style && m.build_epoch == c->style_builds) {
    if (far) *far = (h[0] | h[4]) != 0;
    if (toofar) *toofar = (h[1] | h[5]) != 0;
  }
  if (m.has_prune && m.prune_epoch == c->prune_epoch && c->prune_valid) {
    if (h[2] | h[6]) c->prune_stale = true;
    if (h[3] | h[7]) c->dangerous_prunes++;
  }
  return MDP_OK;
}

// launches the integrate kernel of the next step (with_final: after the pending final half-kick of the finished one)
// and the refresh of the periodic self-images; flag != null: the displacement check of the new positions in the same
// pass (see mdp_md_integrate_check in domain.hip)
int mdp_md_advance(mdp_ctx *c, bool with_final, int *flag, double trigsq, double hardsq)
{
  // a deferred final half-kick (mdp_md_defer_final) that a thermo / velocity read has completed in the meantime
  // must not be applied twice
  if (c->final_deferred_seen && with_final && !c->final_pending) with_final = false;
  if (with_final) c->final_pending = false;
  const double dtf = 0.5 * c->cfg.dt * c->cfg.ftm2v;
  // aeam accumulates into f (three-body atomics, tile kernels): its force_clear rides in this kernel and in the refresh
  // of the images when every ghost is a periodic self-image (one GPU).  The flag is dropped by whatever rebuilds or
  // re-orders the atom arrays before the compute (mdp_aeam_prepare) -- the compute then clears f itself.
  const int nself = c->remote_start >= c->nlocal && c->remote_start <= c->nall ? c->remote_start - c->nlocal : c->nghost;
  // Bricks whose whole step runs in the library (mdp_dd_comm_step_begin): the forces of the remote ghosts are cleared by
  // the halo unpack of the step (mdp_md_unpack_x), which every such step runs before its first force kernel.
  const bool zero_f = c->cfg.style == 2 && c->nlocal > 0 && c->neigh_set && c->f.p &&
                      (nself == c->nghost || (c->dd.step_mode && c->dd.nccl_comm));
  // (several GPUs, library transport: this rank's "moved" word for the peers, see MdpDomain::flagbuf)
  double *dset = nullptr, *dclr = nullptr;
  if (flag && c->dd.flagbuf.p && c->dd.nccl_comm) {
    c->dd.flag_par ^= 1;
    dset = c->dd.flagbuf.p + c->dd.flag_par;
    dclr = c->dd.flagbuf.p + (c->dd.flag_par ^ 1);
  }
  if (c->nlocal) {
    const int g = nblk(c->nlocal);
    MdpStyleCheck sc;
    mdp_sflag_arm(c, sc);
#define MDP_ADV(FV, CV)                                                                                               \
  nve_advance_kernel<FV, CV><<<g, 256, 0, c->stream>>>(c->nlocal, dtf, c->cfg.dt, c->rmass.p, c->f.p, c->v.p, c->xq.p, \
                                                      c->xhold.p, trigsq, hardsq, flag, sc, zero_f ? 1 : 0, dset, dclr)
    if (with_final) {
      if (flag) MDP_ADV(true, true);
      else MDP_ADV(true, false);
    } else {
      if (flag) MDP_ADV(false, true);
      else MDP_ADV(false, false);
    }
#undef MDP_ADV
    c->acc_prezeroed = sc.acc != nullptr;
    // the flag words are complete behind this kernel unless remote ghosts arrive later in the step (mdp_md_unpack_x)
    if (!(c->remote_start < c->nall)) MDP_TRY(mdp_sflag_commit(c));
  }
  // periodic self-images come first in the ghost range; remote ghosts are refreshed by the halo exchange
  if (nself)
    ghost_refresh_kernel<<<nblk(nself), 256, 0, c->stream>>>(c->nlocal, nself, c->ghost_owner.p, c->ghost_shift.p, c->xq.p,
                                                             zero_f ? c->f.p : nullptr);
  MDP_HIP(c, hipGetLastError());
  c->f_prezeroed = zero_f;
  c->f_zero_remote_due = zero_f && nself != c->nghost; // (mdp_md_unpack_x of this step clears the remote ghosts' forces)
  return MDP_OK;
}

extern "C" {

int mdp_md_initial_integrate(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  return mdp_md_advance(c, false, nullptr, 0.0, 0.0);
}

int mdp_md_final_initial_integrate(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  return mdp_md_advance(c, true, nullptr, 0.0, 0.0);
}

int mdp_md_defer_final(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  c->final_pending = true;
  c->final_deferred_seen = true;
  return MDP_OK;
}

int mdp_md_final_integrate(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  c->final_pending = false;
  const double dtf = 0.5 * c->cfg.dt * c->cfg.ftm2v;
  if (c->nlocal) nve_final_kernel<<<nblk(c->nlocal), 256, 0, c->stream>>>(c->nlocal, dtf, c->rmass.p, c->f.p, c->v.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_aeam_density(mdp_ctx *c, int eflag)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (c->cfg.style != 2) return mdp_fail(c, MDP_EINVAL, "not an aeam sub-domain");
  MDP_TRY(mdp_aeam_run_density(c, eflag));
  // forward comm of fp on one rank: periodic self-images copy their owner's value -- normally done by the embedding
  // kernel itself (aeam_img_fp)
  if (c->nghost && !c->aeam_img_fp && !(c->dd.on && c->dd.nself == 0)) // (a brick without periodic self-images: nothing to copy)
    ghost_scalar_refresh_kernel<<<nblk(c->nghost), 256, 0, c->stream>>>(c->nlocal, c->nghost, c->ghost_owner.p,
                                                                        c->fp.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_aeam_force(mdp_ctx *c, int eflag, int vflag)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (c->cfg.style != 2) return mdp_fail(c, MDP_EINVAL, "not an aeam sub-domain");
  return mdp_aeam_run_force(c, eflag, vflag);
}

int mdp_md_fold_self_ghost_f(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  // aeam: the three-body kernel adds what belongs to a periodic self-image to its owner directly (img_owner), so
  // nothing is left on the images; rebomos writes nothing to ghosts at all
  if (c->cfg.style == 2) return MDP_OK;
  if (c->nghost)
    fold_self_ghost_f_kernel<<<nblk(c->nghost), 256, 0, c->stream>>>(c->nlocal, c->nghost, c->ghost_owner.p, c->f.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_compute(mdp_ctx *c, int eflag, int vflag)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (!c->neigh_set) return mdp_fail(c, MDP_ESTATE, "neighbor list not built");
  if (c->cfg.style == 1) return mdp_rebomos_run(c, eflag, vflag, /*zero_f=*/true);
  // single-rank AEAM: density, self-image fp refresh, force, fold angular ghost forces
  c->aeam_phase = 0;
  MDP_TRY(mdp_md_aeam_density(c, eflag));
  MDP_TRY(mdp_md_aeam_force(c, eflag, vflag));
  return mdp_md_fold_self_ghost_f(c);
}

int mdp_md_compute_begin(mdp_ctx *c, int eflag, int vflag)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (!c->neigh_set) return mdp_fail(c, MDP_ESTATE, "neighbor list not built");
  if (c->cfg.style == 1) return mdp_rebomos_run_begin(c, eflag, vflag);
  return mdp_aeam_run_begin(c, eflag, vflag);
}

int mdp_md_aeam_force_begin(mdp_ctx *c, int eflag, int vflag)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (c->cfg.style != 2) return mdp_fail(c, MDP_EINVAL, "not an aeam sub-domain");
  return mdp_aeam_run_force_begin(c, eflag, vflag);
}

int mdp_md_aeam_state(mdp_ctx *c, int out[4])
{
  if (!c || !out) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  out[0] = c->aeam_phase;
  out[1] = c->aeam_split;
  out[2] = c->ntile;
  out[3] = c->aeam_ang_remote ? 1 : 0;
  return MDP_OK;
}

int mdp_md_compute_end(mdp_ctx *c, int eflag, int vflag)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (c->cfg.style == 1) return mdp_rebomos_run_end(c, eflag, vflag);
  // one-rank aeam (or a host that does no exchange of its own): the rest of the compute
  MDP_TRY(mdp_md_aeam_density(c, eflag));
  MDP_TRY(mdp_md_aeam_force(c, eflag, vflag));
  return mdp_md_fold_self_ghost_f(c);
}

// ---- fix nve on the device for a HOST-mode context ------------------------------------------------------------------
// The plugins' `fix nve/mdp` (plugin/fix_nve_mdp.cpp; the reference repository registers a fix style from a plugin the
// same way, USER-BFIELD/bfieldplugin.cpp:15-29): between two reneighborings of the host the owned atoms' positions,
// velocities and forces stay on the device -- nothing per atom crosses the link in a step.  Needs the images kept by the
// library (mdp_set_box_host on one periodic rank), since the ghosts must follow their owners without the host.
int mdp_hnve_setup(mdp_ctx *c, double dt, double ftm2v, const double *mass_type, int ntypes)
{
  if (!c || !(dt > 0.0) || !mass_type || ntypes < 1 || ntypes > 15) return MDP_EINVAL;
  if (c->md) return mdp_fail(c, MDP_ESTATE, "mdp_hnve_setup: a resident-mode context integrates through mdp_md_*");
  c->hn_dt = dt;
  c->hn_dtf = 0.5 * dt * ftm2v;
  for (int t = 0; t < 16; t++) c->hn_mass[t] = t >= 1 && t <= ntypes ? mass_type[t] : 1.0;
  MDP_HIP(c, hipSetDevice(c->device));
  MDP_HIP(c, c->hn_mass_dev.reserve(16));
  MDP_HIP(c, hipMemcpy(c->hn_mass_dev.p, c->hn_mass, sizeof c->hn_mass, hipMemcpyHostToDevice));
  c->hn_on = true;
  c->hn_v_current = false;
  return MDP_OK;
}

int mdp_hnve_off(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  c->hn_on = false;
  c->hn_v_current = false;
  return MDP_OK;
}

// after every mdp_set_atoms_host (the host re-sorted or exchanged atoms): the owned atoms' velocities, host order
int mdp_hnve_upload_v(mdp_ctx *c, const double *v)
{
  if (!c) return MDP_EINVAL;
  if (!c->hn_on) return mdp_fail(c, MDP_ESTATE, "mdp_hnve_setup not called");
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (c->nghost > 0 && !c->host_ghosts_derived)
    return mdp_fail(c, MDP_ESTATE, "the device integrator needs the images kept by the library (mdp_set_box_host, one periodic rank)");
  const int n = c->nlocal;
  if (n > 0 && !v) return mdp_fail(c, MDP_EINVAL, "mdp_hnve_upload_v: v missing for %d owned atoms", n);
  MDP_HIP(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  MDP_HIP(c, c->v.reserve((size_t) 3 * n + 3));
  MDP_HIP(c, c->rmass.reserve(n + 1));
  MDP_HIP(c, c->xhold.reserve((size_t) 3 * n + 3));
  if (n) {
    if (c->host_sort) {
      MDP_HIP(c, c->host_stage.reserve((size_t) 10 * c->nall + 16));
      MDP_TRY(mdp_host_upload(c, c->host_stage.p, v, sizeof(double) * 3 * n));
      MDP_TRY(mdp_to_device_order(c, n, 3, c->host_stage.p, c->v.p));
    } else
      MDP_TRY(mdp_host_upload(c, c->v.p, v, sizeof(double) * 3 * n));
    hn_rmass_kernel<<<nblk(n), 256, 0, st>>>(n, c->type.p, c->host_sort ? c->host_perm.p : nullptr, c->hn_mass_dev.p, c->rmass.p);
    hold_kernel<<<nblk(n), 256, 0, st>>>(n, c->xq.p, c->xhold.p);
    MDP_HIP(c, hipGetLastError());
  }
  MDP_HIP(c, hipStreamSynchronize(st));
  c->dd.moved_pending = false;
  c->hn_v_current = true;
  return MDP_OK;
}

// initial_integrate.  *moved / *dangerous: has an owned atom moved half the host's skin (minus a margin: the answer is
// that of the PREVIOUS call, read without waiting for this one) / beyond half the skin since the last mdp_set_atoms_host
int mdp_hnve_initial(mdp_ctx *c, int *moved, int *dangerous)
{
  if (!c) return MDP_EINVAL;
  if (!c->hn_on || !c->hn_v_current) return mdp_fail(c, MDP_ESTATE, "mdp_hnve_upload_v not called for the current atoms");
  MDP_HIP(c, hipSetDevice(c->device));
  MdpDomain &D = c->dd;
  int *h = (int *) (c->h_pinned + 28);
  if (!D.ev_moved) MDP_HIP(c, hipEventCreateWithFlags(&D.ev_moved, hipEventDisableTiming));
  int m = 0, dg = 0;
  if (D.moved_pending) {
    MDP_HIP(c, hipEventSynchronize(D.ev_moved_ref ? D.ev_moved_ref : D.ev_moved));
    m = h[0];
    dg = h[1];
    D.moved_pending = false;
  }
  if (moved) *moved = m;
  if (dangerous) *dangerous = dg;
  const int n = c->nlocal;
  if (n) {
    h[0] = h[1] = 0;
    const double hard = 0.5 * c->skin;
    double trig = hard - 0.1 * mdp_margin_scale(c);
    if (trig < 0.5 * hard) trig = 0.5 * hard;
    // rebomos: the style's own displacement checks (device-built lists, pruned rows) ride in this kernel and are read
    // by the NEXT compute, as in resident runs (MdpStyleCheck; every ghost is an image that moves with its owner) --
    // a check of its own in front of every compute had the host wait for the stream once per step
    MdpStyleCheck sc;
    c->hn_deferred_check = !c->md && c->have_rebomos && !c->have_aeam && c->host_ghosts_derived;
    if (c->hn_deferred_check) mdp_sflag_arm(c, sc);
    nve_advance_kernel<false, true><<<nblk(n), 256, 0, c->stream>>>(n, c->hn_dtf, c->hn_dt, c->rmass.p, c->f.p, c->v.p, c->xq.p,
                                                                    c->xhold.p, trig * trig, hard * hard, h, sc, 0);
    MDP_HIP(c, hipGetLastError());
    if (c->hn_deferred_check && c->sflag_armed) { // one event behind the kernel serves both readers of its words
      MDP_TRY(mdp_sflag_commit(c));
      D.ev_moved_ref = c->ev_sflag[c->sflag_set];
    } else {
      MDP_HIP(c, hipEventRecord(D.ev_moved, c->stream));
      D.ev_moved_ref = D.ev_moved;
    }
    D.moved_pending = true;
  }
  MDP_TRY(mdp_host_refresh_ghosts(c));  // Comm::forward_comm of x on one periodic rank
  if (!c->hn_deferred_check) {
    MDP_TRY(mdp_rebomos_host_precheck(c)); // (the style's own displacement check of the compute that follows)
    if (c->host_check_armed) MDP_HIP(c, hipStreamSynchronize(c->stream)); // ... whose words that compute reads at once
  }
  return MDP_OK;
}

int mdp_hnve_final(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->hn_on || !c->hn_v_current) return mdp_fail(c, MDP_ESTATE, "mdp_hnve_upload_v not called for the current atoms");
  MDP_HIP(c, hipSetDevice(c->device));
  if (c->nlocal)
    nve_final_kernel<<<nblk(c->nlocal), 256, 0, c->stream>>>(c->nlocal, c->hn_dtf, c->rmass.p, c->f.p, c->v.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// owned atoms' x / v / f in the host's order (any of them NULL: not wanted); complete on return
int mdp_hnve_download(mdp_ctx *c, double *x, double *v, double *f)
{
  if (!c) return MDP_EINVAL;
  if (!c->hn_on || !c->hn_v_current) return mdp_fail(c, MDP_ESTATE, "mdp_hnve_upload_v not called for the current atoms");
  const int n = c->nlocal;
  if (!n) return MDP_OK;
  MDP_HIP(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  MDP_HIP(c, c->host_stage.reserve((size_t) 10 * c->nall + 16));
  MDP_TRY(mdp_host_pinned_reserve(c, (size_t) 9 * n + 16));
  double *s0 = c->host_stage.p, *s1 = s0 + (size_t) 3 * n, *s2 = s1 + (size_t) 3 * n;
  double *h0 = c->h_down, *h1 = h0 + (size_t) 3 * n, *h2 = h1 + (size_t) 3 * n;
  const double *dx = nullptr, *dv = c->v.p, *df = c->f.p;
  if (x) {
    xq_to_x3_kernel<<<nblk(n), 256, 0, st>>>(n, c->xq.p, c->xraw.p);
    dx = c->xraw.p;
    if (c->host_sort) {
      MDP_TRY(mdp_to_host_order(c, n, 3, c->xraw.p, s0));
      dx = s0;
    }
    MDP_HIP(c, hipMemcpyAsync(h0, dx, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  }
  if (v) {
    if (c->host_sort) {
      MDP_TRY(mdp_to_host_order(c, n, 3, c->v.p, s1));
      dv = s1;
    }
    MDP_HIP(c, hipMemcpyAsync(h1, dv, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  }
  if (f) {
    if (c->host_sort) {
      MDP_TRY(mdp_to_host_order(c, n, 3, c->f.p, s2));
      df = s2;
    }
    MDP_HIP(c, hipMemcpyAsync(h2, df, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  }
  MDP_HIP(c, hipStreamSynchronize(st));
  if (x) memcpy(x, h0, sizeof(double) * 3 * n);
  if (v) memcpy(v, h1, sizeof(double) * 3 * n);
  if (f) memcpy(f, h2, sizeof(double) * 3 * n);
  return MDP_OK;
}

int mdp_md_thermo(mdp_ctx *c, double out[9])
{
  if (!c || !out) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (c->final_pending) MDP_TRY(mdp_md_final_integrate(c)); // KE is that of full-step velocities
  hipStream_t st = c->stream;
  MDP_HIP(c, hipMemsetAsync(c->acc.p + 7, 0, sizeof(double) * 2, st));
  const int grid = c->nlocal > 0 ? (nblk(c->nlocal) < 1024 ? nblk(c->nlocal) : 1024) : 0;
  if (grid)
    thermo_kernel<<<grid, 256, 0, st>>>(c->nlocal, 0.5 * c->cfg.mvv2e, c->rmass.p, c->v.p, c->xq.p, c->xhold.p,
                                         c->acc.p);
  MDP_HIP(c, hipGetLastError());
  MDP_HIP(c, hipMemcpyAsync(c->h_pinned, c->acc.p, sizeof(double) * 9, hipMemcpyDeviceToHost, st));
  int *hflags = (int *) (c->h_pinned + 16);
  MDP_HIP(c, hipMemcpyAsync(hflags, c->flags.p, sizeof(int) * 5, hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  MDP_TRY(mdp_flags_check(c, hflags));
  out[0] = c->h_pinned[7];
  out[1] = c->h_pinned[0];
  for (int k = 0; k < 6; k++) out[2 + k] = c->h_pinned[1 + k];
  out[8] = c->h_pinned[8];
  return MDP_OK;
}

int mdp_md_download(mdp_ctx *c, double *x, double *v, double *f, double *eatom)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (v && c->final_pending) MDP_TRY(mdp_md_final_integrate(c)); // full-step velocities
  hipStream_t st = c->stream;
  const int n = c->nlocal;
  if (x && n) {
    xq_to_x3_kernel<<<nblk(n), 256, 0, st>>>(n, c->xq.p, c->xraw.p);
    MDP_HIP(c, hipMemcpyAsync(x, c->xraw.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  }
  if (v && n) MDP_HIP(c, hipMemcpyAsync(v, c->v.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  if (f && n) MDP_HIP(c, hipMemcpyAsync(f, c->f.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  if (eatom && n) MDP_HIP(c, hipMemcpyAsync(eatom, c->eatom.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  return MDP_OK;
}

int mdp_md_download_x_all(mdp_ctx *c, double *x_all)
{
  if (!c || !x_all) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  hipStream_t st = c->stream;
  const int n = c->nall;
  if (n) {
    MDP_HIP(c, c->xraw.reserve((size_t) 3 * n + 3));
    xq_to_x3_kernel<<<nblk(n), 256, 0, st>>>(n, c->xq.p, c->xraw.p);
    MDP_HIP(c, hipMemcpyAsync(x_all, c->xraw.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, st));
  }
  MDP_HIP(c, hipStreamSynchronize(st));
  return MDP_OK;
}

int mdp_md_upload_x(mdp_ctx *c, const double *x)
{
  if (!c || !x) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  hipStream_t st = c->stream;
  const int n = c->nlocal;
  if (n) {
    MDP_HIP(c, hipMemcpyAsync(c->xraw.p, x, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
    x3_to_xq_kernel<<<nblk(n), 256, 0, st>>>(n, c->xraw.p, c->xq.p);
  }
  if (c->nghost)
    ghost_refresh_kernel<<<nblk(c->nghost), 256, 0, st>>>(c->nlocal, c->nghost, c->ghost_owner.p, c->ghost_shift.p,
                                                          c->xq.p);
  MDP_HIP(c, hipGetLastError());
  MDP_HIP(c, hipStreamSynchronize(st));
  // Positions were rewritten outside the integrator: rows pruned for the old positions (buffer 0.2-0.6 A) may miss
  // pairs now, and the deferred displacement checks in flight looked at the old positions.  The next compute
  // prunes afresh and checks the style's own lists BEFORE it walks them (blocking, once).
  c->prune_valid = false;
  c->prune_stale = false;
  c->prune_epoch++;
  mdp_sflag_drop(c);
  c->check_now = true;
  return MDP_OK;
}

static int host_list_compare(mdp_ctx *c, const CutTables &ct, double maxcut, unsigned long long host_total,
                             const char *style)
{
  hipStream_t st = c->stream;
  MDP_TRY(mdp_bin_atoms(c, maxcut, c->bbox_lo, c->bbox_hi));
  MDP_HIP(c, c->scan_tmp.reserve(64));
  unsigned long long *d_total = reinterpret_cast<unsigned long long *>(c->scan_tmp.p);
  MDP_HIP(c, hipMemsetAsync(d_total, 0, sizeof(unsigned long long), st));
  host_list_count_kernel<<<nblk(c->nall), 256, 0, st>>>(c->grid, ct, c->nall, c->nlocal, c->xq.p, c->cell_perm.p,
                                                        c->cell_start.p, d_total);
  MDP_HIP(c, hipGetLastError());
  unsigned long long dev_total = 0;
  MDP_TRY(mdp_read_one(c, d_total, sizeof dev_total, &dev_total));
  if (dev_total != host_total)
    return mdp_fail(c, MDP_EINVAL,
                    "host neighbor list is not the plain geometric list (%llu entries, %llu pairs within the list cutoff "
                    "%.6g): exclusions / skip lists are not supported by the MI355X %s style, which builds its own lists",
                    host_total, dev_total, maxcut, style);
  return MDP_OK;
}

// Guard for hosts whose list is not the plain geometric one.  The REBO-MoS device path derives its lists from
// the positions (mdp_set_skin); the reference iterates the HOST's entries (pair_rebomos.cpp:304-307, 328-330,
// 490-495), so `neigh_modify exclude`, special_bonds weights or a hybrid skip list would silently change the
// reference's result but not this one.  Compare the host's owned-row entry count with the geometric count at the
// host's list cutoff and look for special-bond bits in a sample of rows; fail loudly on any difference.
int mdp_rebomos_check_host_list(mdp_ctx *c, int inum, const int *ilist, const int *numneigh, int *const *firstneigh,
                                double cutneigh)
{
  if (!c || inum < 0 || !(cutneigh > 0.0)) return MDP_EINVAL;
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (const char *e = getenv("MDP_SKIP_LIST_CHECK"))
    if (atoi(e) != 0) return MDP_OK;
  if (inum != c->nlocal) return mdp_fail(c, MDP_EINVAL, "host neighbor list has %d owned rows, expected nlocal = %d", inum, c->nlocal);
  if (inum == 0) return MDP_OK;
  if (!ilist || !numneigh || !firstneigh) return mdp_fail(c, MDP_EINVAL, "host neighbor list arrays missing");
  MDP_HIP(c, hipSetDevice(c->device));
  unsigned long long host_total = 0;
  const int stride = inum > 65536 ? inum / 65536 : 1;
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    if (i < 0 || i >= c->nlocal) return mdp_fail(c, MDP_EINVAL, "host neighbor list: owned row %d names atom %d", ii, i);
    host_total += (unsigned long long) numneigh[i];
    if (ii % stride == 0) {
      const int *row = firstneigh[i];
      for (int k = 0; k < numneigh[i]; k++)
        if (row[k] & ~MDP_NEIGHMASK)
          return mdp_fail(c, MDP_EINVAL,
                          "host neighbor list carries special-bond bits (atom %d): the MI355X rebomos style builds its "
                          "own lists from the positions and cannot honour special_bonds", i);
    }
  }
  CutTables ct;
  memset(&ct, 0, sizeof ct);
  ct.ne = 2;
  for (int k = 0; k < 4; k++) ct.owned[k] = cutneigh * cutneigh;
  return host_list_compare(c, ct, cutneigh, host_total, "rebomos");
}

// the same guard for aeam when it builds its lists on the device (mdp_aeam_device_lists): per-type-pair cutoffs
// cut[ti][tj] + skin, as the host built its list (pair_aeam.cpp:615-621 + neighbor skin)
int mdp_aeam_check_host_list(mdp_ctx *c, int inum, const int *ilist, const int *numneigh, int *const *firstneigh,
                             double skin)
{
  if (!c || inum < 0 || !(skin >= 0.0)) return MDP_EINVAL;
  if (!c->have_aeam) return mdp_fail(c, MDP_ESTATE, "aeam tables not set");
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (const char *e = getenv("MDP_SKIP_LIST_CHECK"))
    if (atoi(e) != 0) return MDP_OK;
  if (inum != c->nlocal) return mdp_fail(c, MDP_EINVAL, "host neighbor list has %d owned rows, expected nlocal = %d", inum, c->nlocal);
  if (inum == 0) return MDP_OK;
  if (!ilist || !numneigh || !firstneigh) return mdp_fail(c, MDP_EINVAL, "host neighbor list arrays missing");
  MDP_HIP(c, hipSetDevice(c->device));
  unsigned long long host_total = 0;
  const int stride = inum > 65536 ? inum / 65536 : 1;
  for (int ii = 0; ii < inum; ii++) {
    const int i = ilist[ii];
    if (i < 0 || i >= c->nlocal) return mdp_fail(c, MDP_EINVAL, "host neighbor list: owned row %d names atom %d", ii, i);
    host_total += (unsigned long long) numneigh[i];
    if (ii % stride == 0) {
      const int *row = firstneigh[i];
      for (int k = 0; k < numneigh[i]; k++)
        if (row[k] & ~MDP_NEIGHMASK)
          return mdp_fail(c, MDP_EINVAL,
                          "host neighbor list carries special-bond bits (atom %d): the MI355X aeam style builds its own "
                          "lists from the positions and cannot honour special_bonds", i);
    }
  }
  CutTables ct;
  memset(&ct, 0, sizeof ct);
  double maxcut = 0.0;
  MDP_TRY(aeam_cut_table(c, ct, skin, maxcut));
  return host_list_compare(c, ct, maxcut, host_total, "aeam");
}

void *mdp_md_ptr(mdp_ctx *c, const char *name)
{
  if (!c || !name) return nullptr;
  if (!strcmp(name, "x")) return c->xq.p; // double4 {x,y,z,element}
  if (!strcmp(name, "v")) return c->v.p;
  if (!strcmp(name, "f")) return c->f.p;
  if (!strcmp(name, "fp")) return c->fp.p;
  if (!strcmp(name, "eatom")) return c->eatom.p;
  return nullptr;
}

int mdp_rebomos_list_info(mdp_ctx *c, long long out[8])
{
  if (!c || !out) return MDP_EINVAL;
  if (!c->rebo_packed) return mdp_fail(c, MDP_ESTATE, "rebomos lists not built yet");
  out[0] = c->lj_tiled ? 1 : 0;
  out[1] = c->ntile;
  out[2] = c->tile_cap;
  out[3] = c->tile_maxu;
  out[4] = c->lj_total;
  out[5] = c->nclus;
  out[6] = (c->lj_class_base[2] - c->lj_class_base[1]) + (c->lj_class_base[4] - c->lj_class_base[3]);
  out[7] = c->style_builds;
  return MDP_OK;
}

int mdp_md_class_stats(mdp_ctx *c, long long out[32])
{
  if (!c || !out) return MDP_EINVAL;
  for (int k = 0; k < 32; k++) out[k] = 0;
  out[30] = c->tile_maxu;
  if (c->cfg.style == 2 || (c->have_aeam && !c->have_rebomos)) {
    out[0] = c->h_ang_count;
    out[1] = c->ntile;
    out[2] = c->aeam_split;
    return MDP_OK;
  }
  for (int k = 0; k < MDP_NCLASS; k++) out[k] = c->h_class_count[k];
  for (int k = 0; k < 4; k++) out[20 + k] = c->lj_class_base[k + 1] - c->lj_class_base[k];
  if (c->lj_tiled && !c->lj_ordered) out[20] = c->ntile; // (no large unions: one class, natural order)
  if (c->h_pinned) {
    const int *h = (const int *) (c->h_pinned + 40) + 4 * c->ovf_par; // (the set the last compute published)
    for (int k = 0; k < 4; k++) out[25 + k] = h[k];
    out[24] = ((const int *) (c->h_pinned + 44))[0];
  }
  out[29] = c->tile_small;
  return MDP_OK;
}

int mdp_md_neighbor_stats(mdp_ctx *c, long long out[8])
{
  if (!c || !out) return MDP_EINVAL;
  for (int k = 0; k < 8; k++) out[k] = 0;
  out[0] = c->nb_owned_total;
  out[1] = c->nb_total - c->nb_owned_total;
  out[2] = c->lj_total;
  out[3] = c->cand_total;
  out[4] = 0;
  for (int k = 0; k < MDP_NCLASS; k++) out[4] += c->h_class_count[k];
  for (int part = 0; part < 2; part++) {
    const int *h = c->h_class_count + part * MDP_NCLASS_HALF;
    out[5] += (long long) h[0] + h[1];               // 4-lane groups, both elements
    out[6] += (long long) h[4] + h[5] + h[6] + h[7]; // 12- and 16-lane groups
  }
  out[7] = c->cfg.style == 1 ? c->style_builds : c->h_ang_count;
  return MDP_OK;
}

int mdp_md_list_state(mdp_ctx *c, double out[8])
{
  if (!c || !out) return MDP_EINVAL;
  for (int k = 0; k < 8; k++) out[k] = 0.0;
  out[0] = c->cfg.style == 1 ? c->skin_inner : c->cfg.skin;
  out[1] = c->skin_inner_cap < 1.0e8 ? c->skin_inner_cap : 0.0;
  out[2] = c->prune_valid ? c->prune_buf : 0.0;
  out[3] = (double) c->dangerous_builds;
  if (c->h_pinned) {
    const int *h = (const int *) (c->h_pinned + 40) + 4 * c->ovf_par; // (the set the last compute published)
    out[4] = (double) h[0] + h[1] + h[2] + h[3];
  }
  out[5] = (c->ovf3_hot[0] > 0 || c->ovf3_hot[1] > 0 || c->ovf3_hot[2] > 0 || c->ovf3_hot[3] > 0) ? 1.0 : 0.0;
  return MDP_OK;
}

int mdp_md_prune_stats(mdp_ctx *c, long long out[4])
{
  if (!c || !out) return MDP_EINVAL;
  out[0] = c->prunes;
  out[1] = c->dangerous_prunes;
  out[2] = c->prune_valid ? 1 : 0;
  out[3] = (long long) (c->prune_buf * 1.0e6 + 0.5);
  return MDP_OK;
}

// ---- halo plumbing -----------------------------------------------------------------------------------
int mdp_md_pack_x(mdp_ctx *c, int n, const int *d_sendlist, const double *d_shift, double *d_buf)
{
  if (!c || n < 0) return MDP_EINVAL;
  if (n) pack_x_kernel<<<nblk(n), 256, 0, c->stream>>>(n, d_sendlist, d_shift, c->xq.p, d_buf);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_unpack_x(mdp_ctx *c, int first_ghost, int n, const double *d_buf)
{
  if (!c || n < 0 || first_ghost < 0 || first_ghost + n > c->nghost) return MDP_EINVAL;
  // the arriving ghosts are checked against the style's reference positions in the same pass (see MdpStyleCheck)
  MdpStyleCheck sc;
  if (c->sflag_armed) {
    sc = c->sflag_chk;
    sc.acc = nullptr;
  }
  // (library transport with the displacement words gathered behind this halo: reduced here, read by the next step)
  MdpDomain &D = c->dd;
  const bool glob = D.flagbuf.p && D.nccl_comm && D.fwd_gathered;
  int *h_glob = (int *) (c->h_pinned + 46);
  if (n || glob)
    unpack_x_kernel<<<nblk(n > 0 ? n : 1), 256, 0, c->stream>>>(n, c->nlocal + first_ghost, d_buf, c->xq.p, sc,
                                                                glob ? D.flagbuf.p + 2 : nullptr, glob ? D.G.nranks : 0,
                                                                glob ? h_glob : nullptr,
                                                                c->f_zero_remote_due ? c->f.p : nullptr);
  c->f_zero_remote_due = false;
  MDP_HIP(c, hipGetLastError());
  // one event behind this kernel serves both readers of its pinned words (an event record costs the stream ~5 us)
  if (c->sflag_armed) MDP_TRY(mdp_sflag_commit(c));
  if (glob) {
    if (c->sflag_armed) {
      D.ev_glob_ref = c->ev_sflag[c->sflag_set];
    } else {
      if (!D.ev_glob) MDP_HIP(c, hipEventCreateWithFlags(&D.ev_glob, hipEventDisableTiming));
      MDP_HIP(c, hipEventRecord(D.ev_glob, c->stream));
      D.ev_glob_ref = D.ev_glob;
    }
    D.glob_pending = true;
    D.fwd_gathered = false;
  }
  return MDP_OK;
}

int mdp_md_pack_scalar(mdp_ctx *c, int which, int n, const int *d_sendlist, double *d_buf)
{
  if (!c || n < 0 || which != 0) return MDP_EINVAL;
  if (n) pack_scalar_kernel<<<nblk(n), 256, 0, c->stream>>>(n, d_sendlist, c->fp.p, d_buf);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_unpack_scalar(mdp_ctx *c, int which, int first_ghost, int n, const double *d_buf)
{
  if (!c || n < 0 || which != 0 || first_ghost < 0 || first_ghost + n > c->nghost) return MDP_EINVAL;
  if (n) copy_kernel<<<nblk(n), 256, 0, c->stream>>>(n, d_buf, c->fp.p + c->nlocal + first_ghost);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_pack_ghost_f(mdp_ctx *c, int first_ghost, int n, double *d_buf)
{
  if (!c || n < 0 || first_ghost < 0 || first_ghost + n > c->nghost) return MDP_EINVAL;
  if (n) copy_kernel<<<nblk(3ll * n), 256, 0, c->stream>>>(3 * n, c->f.p + 3 * (size_t) (c->nlocal + first_ghost), d_buf);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_md_unpack_add_f(mdp_ctx *c, int n, const int *d_sendlist, const double *d_buf)
{
  if (!c || n < 0) return MDP_EINVAL;
  if (n) unpack_add_f_kernel<<<nblk(n), 256, 0, c->stream>>>(n, d_sendlist, d_buf, c->f.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

} // extern "C"
