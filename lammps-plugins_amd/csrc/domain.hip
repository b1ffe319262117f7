// domain.hip -- brick domain decomposition kept on the GPU: what LAMMPS' Domain::remap, Comm::exchange and
// Comm::borders do on the host for the reference plugins (the reference only works because the host gives it
// ghost atoms with their own lists: REQ_GHOST, USER-REBOMOS/pair_rebomos.cpp:218; "2 by 2 by 1 MPI processor
// grid", log.rebomos-bulk.4:22; Nlocal / Nghost of log.rebomos-bulk.1:72-75 and .4:72-75).
//
// One brick of the periodic (triclinic) box per GPU, in lamda (fractional) coordinates.  At every
// reneighboring, each rank works on ITS atoms only:
//   1. remap      owned atoms are wrapped back into the box, their new brick is looked up
//   2. exchange   atoms that left the brick are packed per destination; arrivals are appended
//   3. order      owned atoms are sorted along a Hilbert curve over the brick (tile lists want compact runs)
//   4. borders    every (atom, periodic image) that falls into the ghost shell of a brick becomes an entry:
//                 entries for the own brick are periodic self-images (refreshed on the device each step),
//                 entries for other bricks form the per-step send list; entries are ordered along a Hilbert
//                 curve over the destination's shell so that ghosts arrive spatially coherent
// The host sees O(ranks) counts.  The bytes are moved by the caller (RCCL all-to-all through torch.distributed
// in bench.py; a staged gloo group in the one-GPU rehearsal tests): the library packs and unpacks device buffers.
#include "mdp_common.h"

#include <rocprim/rocprim.hpp>

#include <cmath>

namespace {

inline int nblk(long long n) { return (int) ((n + 255) / 256); }

__device__ __forceinline__ void dd_x2lamda(const DdGeom &G, const double x, const double y, const double z, double lam[3])
{
  const double d0 = x - G.lo[0], d1 = y - G.lo[1], d2 = z - G.lo[2];
  lam[0] = G.hinv[0] * d0 + G.hinv[5] * d1 + G.hinv[4] * d2;
  lam[1] = G.hinv[1] * d1 + G.hinv[3] * d2;
  lam[2] = G.hinv[2] * d2;
}

// Cartesian displacement of s box vectors
__device__ __forceinline__ void dd_shift_cart(const DdGeom &G, const double s0, const double s1, const double s2,
                                              double out[3])
{
  out[0] = G.h[0] * s0 + G.h[5] * s1 + G.h[4] * s2;
  out[1] = G.h[1] * s1 + G.h[3] * s2;
  out[2] = G.h[2] * s2;
}

__device__ __forceinline__ unsigned dd_cell10(const double u) // u in [0,1) -> 0..1023, clamped
{
  int g = (int) (u * 1024.0);
  g = g < 0 ? 0 : (g > 1023 ? 1023 : g);
  return (unsigned) g;
}

// 1. remap + destination brick.  Positions are only rewritten for atoms that left the box (a round trip through
// lamda space would move every atom by rounding error).
__global__ __launch_bounds__(256) void dd_remap_kernel(const DdGeom G, const int n, double4 *__restrict__ xq,
                                                       int *__restrict__ dest, int *__restrict__ counts)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double4 x = xq[i];
  double lam[3];
  dd_x2lamda(G, x.x, x.y, x.z, lam);
  double s[3];
  bool moved = false;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if (G.nonper[d]) { // not periodic: the atom stays where it is, beyond the box it belongs to the brick at that end
      s[d] = 0.0;
      continue;
    }
    s[d] = floor(lam[d]);
    lam[d] -= s[d];
    if (lam[d] >= 1.0) { // lam was a tiny negative number: lam - floor(lam) rounds to 1
      lam[d] -= 1.0;
      s[d] += 1.0;
    }
    moved = moved || s[d] != 0.0;
  }
  if (moved) {
    double sh[3];
    dd_shift_cart(G, s[0], s[1], s[2], sh);
    x.x -= sh[0];
    x.y -= sh[1];
    x.z -= sh[2];
    xq[i] = x;
  }
  int b[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    b[d] = lam[d] < 0.0 ? 0 : (int) (lam[d] * G.g[d]);
    b[d] = b[d] < 0 ? 0 : (b[d] >= G.g[d] ? G.g[d] - 1 : b[d]);
  }
  const int q = (b[0] * G.g[1] + b[1]) * G.g[2] + b[2];
  dest[i] = q;
  if (q != G.rank) atomicAdd(&counts[q], 1); // leavers are rare
}

// 2. exchange: leavers -> records of 8 doubles {x, v, type, tag}, grouped by destination
__global__ __launch_bounds__(256) void dd_pack_leavers_kernel(const int n, const int rank, const int *__restrict__ dest,
                                                              const int *__restrict__ seg_off, int *__restrict__ cursor,
                                                              const double4 *__restrict__ xq, const double *__restrict__ v,
                                                              const int *__restrict__ type, const int *__restrict__ tag,
                                                              double *__restrict__ buf)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int q = dest[i];
  if (q == rank) return;
  const int pos = seg_off[q] + atomicAdd(&cursor[q], 1);
  const double4 x = xq[i];
  double *r = buf + 8 * (size_t) pos;
  r[0] = x.x;
  r[1] = x.y;
  r[2] = x.z;
  r[3] = v[3 * (size_t) i];
  r[4] = v[3 * (size_t) i + 1];
  r[5] = v[3 * (size_t) i + 2];
  r[6] = (double) type[i];
  r[7] = (double) tag[i];
}

__global__ __launch_bounds__(256) void dd_unpack_arrivals_kernel(const int n, const int first, const double *__restrict__ buf,
                                                                 const int *__restrict__ map, double4 *__restrict__ xq,
                                                                 double *__restrict__ v, int *__restrict__ type,
                                                                 int *__restrict__ tag, int *__restrict__ dest, const int rank)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const double *r = buf + 8 * (size_t) k;
  const int i = first + k, t = (int) r[6];
  xq[i] = make_double4(r[0], r[1], r[2], (double) map[t]);
  v[3 * (size_t) i] = r[3];
  v[3 * (size_t) i + 1] = r[4];
  v[3 * (size_t) i + 2] = r[5];
  type[i] = t;
  tag[i] = (int) r[7];
  dest[i] = rank;
}

// 3. storage order of the owned atoms: Hilbert key over the brick (lamda space) | tag, leavers behind everything
// shell_last: atoms within the ghost cutoff of a brick face that has a remote neighbour (the only atoms whose lists
// can reach a remote ghost, and the only ones on the send list) are stored behind all others -- so that "needs this
// step's halo" is a RANGE of atoms, tiles and clusters, and the interior can be computed while the halo travels
__global__ __launch_bounds__(256) void dd_order_key_kernel(const DdGeom G, const int n, const double4 *__restrict__ xq,
                                                           const int *__restrict__ tag, const int *__restrict__ dest,
                                                           unsigned long long *__restrict__ key, int *__restrict__ idx,
                                                           const int shell_last)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  idx[i] = i;
  if (dest[i] != G.rank) {
    key[i] = ~0ull;
    return;
  }
  const double4 x = xq[i];
  double lam[3];
  dd_x2lamda(G, x.x, x.y, x.z, lam);
  unsigned c[3];
  unsigned long long shell = 0;
  double u[3], lo[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int d = 0; d < 3; d++) u[d] = lam[d] * G.g[d] - G.me[d];
  if (shell_last) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const double w = G.cutl[d] * G.g[d];
      if (G.g[d] > 1 || G.self_remote) {
        if (u[d] < w || u[d] >= 1.0 - w) shell = 1ull << 62;
        if (w < 0.45) lo[d] = w;
      }
    }
  }
  // interior atoms: the curve fills the INTERIOR box (the brick minus its shell) -- the brick's own curve leaves it for
  // excursions through the shell, and 32 consecutive interior atoms either side of one are two clumps with a union of
  // nearly twice the size (the largest union sizes the LDS staging of every tile of a launch)
#pragma unroll
  for (int d = 0; d < 3; d++) c[d] = dd_cell10(shell ? u[d] : (u[d] - lo[d]) / (1.0 - 2.0 * lo[d]));
  key[i] = shell | ((unsigned long long) mdp_hilbert30(c[0], c[1], c[2]) << 32) | (unsigned) tag[i];
}

__global__ __launch_bounds__(256) void dd_permute_kernel(const int n, const int *__restrict__ perm,
                                                         const double4 *__restrict__ xq_in, const double *__restrict__ v_in,
                                                         const int *__restrict__ type_in, const int *__restrict__ tag_in,
                                                         const double *__restrict__ mass_type, double4 *__restrict__ xq,
                                                         double *__restrict__ v, int *__restrict__ type,
                                                         int *__restrict__ tag, double *__restrict__ rmass)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = perm[i];
  xq[i] = xq_in[o];
  v[3 * (size_t) i] = v_in[3 * (size_t) o];
  v[3 * (size_t) i + 1] = v_in[3 * (size_t) o + 1];
  v[3 * (size_t) i + 2] = v_in[3 * (size_t) o + 2];
  const int t = type_in[o];
  type[i] = t;
  tag[i] = tag_in[o];
  rmass[i] = mass_type[t];
}

// 4. borders.  Per dimension the (image shift, brick) pairs whose shell [lo_b - c, hi_b + c) holds lam + shift;
// an entry is one choice per dimension, minus the atom itself in its own brick.
struct DimHits {
  int n;
  signed char s[12], b[12];
};

__device__ __forceinline__ void dd_dim_hits(const DdGeom &G, const int d, const double lam, DimHits &H)
{
  H.n = 0;
  for (int s = -G.ns[d]; s <= G.ns[d]; s++) {
    const double ls = lam + (double) s;
    for (int b = 0; b < G.g[d]; b++) {
      double lo = (double) b / (double) G.g[d] - G.cutl[d], hi = (double) (b + 1) / (double) G.g[d] + G.cutl[d];
      if (G.nonper[d]) { // the end bricks of a non-periodic dimension reach to infinity
        if (b == 0) lo = -1.0e300;
        if (b == G.g[d] - 1) hi = 1.0e300;
      }
      if (ls >= lo && ls < hi && H.n < 12) {
        H.s[H.n] = (signed char) s;
        H.b[H.n] = (signed char) b;
        H.n++;
      }
    }
  }
}

template <bool FILL>
__global__ __launch_bounds__(256) void dd_border_kernel(const DdGeom G, const int n, const double4 *__restrict__ xq,
                                                        int *__restrict__ cnt, const int *__restrict__ off,
                                                        int *__restrict__ ent_atom, int *__restrict__ ent_code,
                                                        unsigned long long *__restrict__ key)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double4 x = xq[i];
  double lam[3];
  dd_x2lamda(G, x.x, x.y, x.z, lam);
  DimHits H0, H1, H2;
  dd_dim_hits(G, 0, lam[0], H0);
  dd_dim_hits(G, 1, lam[1], H1);
  dd_dim_hits(G, 2, lam[2], H2);
  int m = 0;
  const int base = FILL ? off[i] : 0;
  for (int a = 0; a < H0.n; a++)
    for (int b = 0; b < H1.n; b++)
      for (int c = 0; c < H2.n; c++) {
        const int q = (H0.b[a] * G.g[1] + H1.b[b]) * G.g[2] + H2.b[c];
        const int s0 = H0.s[a], s1 = H1.s[b], s2 = H2.s[c];
        if (q == G.rank && s0 == 0 && s1 == 0 && s2 == 0) continue; // the atom itself
        if (FILL) {
          const int p = base + m;
          ent_atom[p] = i;
          ent_code[p] = (q << 9) | (((s0 + 3) * 7 + (s1 + 3)) * 7 + (s2 + 3));
          // order: own brick first, then by destination rank; inside a destination along a Hilbert curve over
          // ITS extended brick, so the receiver's ghosts are spatially coherent runs
          const int bq[3] = {H0.b[a], H1.b[b], H2.b[c]};
          const double ls[3] = {lam[0] + s0, lam[1] + s1, lam[2] + s2};
          unsigned cc[3];
#pragma unroll
          for (int d = 0; d < 3; d++) {
            const double w = 1.0 / (double) G.g[d] + 2.0 * G.cutl[d];
            cc[d] = dd_cell10((ls[d] - ((double) bq[d] / (double) G.g[d] - G.cutl[d])) / w);
          }
          const unsigned long long klass = (q == G.rank && !G.self_remote) ? 0ull : (unsigned long long) (q + 1);
          key[p] = (klass << 32) | mdp_hilbert30(cc[0], cc[1], cc[2]);
        }
        m++;
      }
  if (!FILL) cnt[i] = m;
}

// first entry of every class in the sorted order (start[] preset to -1)
__global__ __launch_bounds__(256) void dd_class_start_kernel(const int n, const unsigned long long *__restrict__ key,
                                                             int *__restrict__ start)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int k = (int) (key[p] >> 32);
  if (p == 0 || (int) (key[p - 1] >> 32) != k) start[k] = p;
}

// sorted entries -> self-image ghost table (owner, Cartesian shift) and the send list of the remote ones
__global__ __launch_bounds__(256) void dd_emit_kernel(const DdGeom G, const int nent, const int nself,
                                                      const int *__restrict__ perm, const int *__restrict__ ent_atom,
                                                      const int *__restrict__ ent_code, int *__restrict__ ghost_owner,
                                                      double *__restrict__ ghost_shift, int *__restrict__ sendlist,
                                                      double *__restrict__ sendshift)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nent) return;
  const int e = perm[p];
  const int a = ent_atom[e], code = ent_code[e] & 511;
  const int s0 = code / 49 - 3, s1 = (code / 7) % 7 - 3, s2 = code % 7 - 3;
  double sh[3];
  dd_shift_cart(G, (double) s0, (double) s1, (double) s2, sh);
  if (p < nself) {
    ghost_owner[p] = a;
    ghost_shift[3 * (size_t) p] = sh[0];
    ghost_shift[3 * (size_t) p + 1] = sh[1];
    ghost_shift[3 * (size_t) p + 2] = sh[2];
  } else {
    const int k = p - nself;
    sendlist[k] = a;
    sendshift[3 * (size_t) k] = sh[0];
    sendshift[3 * (size_t) k + 1] = sh[1];
    sendshift[3 * (size_t) k + 2] = sh[2];
  }
}

// border records for the receivers: {x + shift, element, type, tag}
__global__ __launch_bounds__(256) void dd_pack_border_kernel(const int n, const int *__restrict__ sendlist,
                                                             const double *__restrict__ sendshift,
                                                             const double4 *__restrict__ xq, const int *__restrict__ type,
                                                             const int *__restrict__ tag, double *__restrict__ buf)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int a = sendlist[k];
  const double4 x = xq[a];
  double *r = buf + 6 * (size_t) k;
  r[0] = x.x + sendshift[3 * (size_t) k];
  r[1] = x.y + sendshift[3 * (size_t) k + 1];
  r[2] = x.z + sendshift[3 * (size_t) k + 2];
  r[3] = x.w;
  r[4] = (double) type[a];
  r[5] = (double) tag[a];
}

__global__ __launch_bounds__(256) void dd_unpack_border_kernel(const int n, const int first, const double *__restrict__ buf,
                                                               double4 *__restrict__ xq, int *__restrict__ type,
                                                               int *__restrict__ tag, int *__restrict__ ghost_owner,
                                                               double *__restrict__ ghost_shift, const int gfirst)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const double *r = buf + 6 * (size_t) k;
  xq[first + k] = make_double4(r[0], r[1], r[2], r[3]);
  type[first + k] = (int) r[4];
  tag[first + k] = (int) r[5];
  ghost_owner[gfirst + k] = -1; // remote: refreshed by the halo exchange, not on the device
  ghost_shift[3 * (size_t) (gfirst + k)] = 0.0;
  ghost_shift[3 * (size_t) (gfirst + k) + 1] = 0.0;
  ghost_shift[3 * (size_t) (gfirst + k) + 2] = 0.0;
}

__global__ __launch_bounds__(256) void dd_self_ghost_kernel(const int nself, const int nlocal,
                                                            const int *__restrict__ owner, const double *__restrict__ shift,
                                                            double4 *__restrict__ xq, int *__restrict__ type,
                                                            int *__restrict__ tag)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nself) return;
  const int o = owner[g];
  const double4 x = xq[o];
  xq[nlocal + g] = make_double4(x.x + shift[3 * (size_t) g], x.y + shift[3 * (size_t) g + 1], x.z + shift[3 * (size_t) g + 2], x.w);
  type[nlocal + g] = type[o];
  tag[nlocal + g] = tag[o];
}

// `neigh_modify check yes` on the host-level skin, owned atoms (Neighbor::check_distance): flag[0] = some atom is
// beyond the trigger distance, flag[1] = some atom is beyond half the skin itself (a build that came too late)
__global__ __launch_bounds__(256) void dd_moved_kernel(const int n, const double trigsq, const double hardsq,
                                                       const double4 *__restrict__ xq, const mdp_hold_t *__restrict__ xhold,
                                                       int *__restrict__ flag)
{
  bool t = false, h = false;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double4 x = xq[i];
    const double dx = x.x - xhold[3 * (size_t) i], dy = x.y - xhold[3 * (size_t) i + 1], dz = x.z - xhold[3 * (size_t) i + 2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    t = t || d2 > trigsq;
    h = h || d2 > hardsq;
  }
  if (__ballot(t) && (threadIdx.x & 63) == 0) flag[0] = 1;
  if (__ballot(h) && (threadIdx.x & 63) == 0) flag[1] = 1;
}

template <typename T> void swap_buf(DevBuf<T> &a, DevBuf<T> &b)
{
  T *p = a.p;
  a.p = b.p;
  b.p = p;
  const size_t c = a.cap;
  a.cap = b.cap;
  b.cap = c;
}

int sort_u64(mdp_ctx *c, const unsigned long long *kin, unsigned long long *kout, const int *vin, int *vout, size_t n,
             int bits)
{
  size_t tmp = 0;
  MDP_HIP(c, rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
  MDP_HIP(c, c->scan_tmp.reserve(tmp + 16));
  MDP_HIP(c, rocprim::radix_sort_pairs(c->scan_tmp.p, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
  return MDP_OK;
}

} // namespace

// sizes of everything that is indexed by atom, for `nlocal` owned atoms and `nghost` ghosts
static int dd_reserve_atoms(mdp_ctx *c, int nlocal, int nghost)
{
  const int nall = nlocal + nghost;
  hipStream_t st = c->stream;
  MDP_HIP(c, c->xq.reserve((size_t) nall + 1, true, st));
  MDP_HIP(c, c->type.reserve((size_t) nall + 32, true, st));
  MDP_HIP(c, c->tag.reserve((size_t) nall + 1, true, st));
  MDP_HIP(c, c->v.reserve((size_t) 3 * nlocal + 3, true, st));
  MDP_HIP(c, c->rmass.reserve((size_t) nlocal + 1, true, st));
  MDP_HIP(c, c->xraw.reserve((size_t) 3 * nall + 3));
  MDP_HIP(c, c->f.reserve((size_t) 3 * nall + 3));
  MDP_HIP(c, c->eatom.reserve((size_t) nall + 1));
  MDP_HIP(c, c->xhold.reserve((size_t) 3 * nlocal + 3));
  MDP_HIP(c, c->rho.reserve((size_t) nall + 1));
  MDP_HIP(c, c->fp.reserve((size_t) nall + 1));
  MDP_HIP(c, c->ghost_owner.reserve((size_t) nghost + 1, true, st)); // the self-image entries are in already
  MDP_HIP(c, c->ghost_shift.reserve((size_t) 3 * nghost + 3, true, st));
  return MDP_OK;
}

static int dd_require(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  if (!c->dd.on) return mdp_fail(c, MDP_ESTATE, "mdp_dd_setup not called");
  MDP_HIP(c, hipSetDevice(c->device));
  return MDP_OK;
}

extern "C" int mdp_dd_comm_destroy(mdp_ctx *c);

void mdp_dd_release(mdp_ctx *c)
{
  MdpDomain &D = c->dd;
  (void) mdp_dd_comm_destroy(c);
  D.dest.release();
  D.counters.release();
  D.idx_a.release();
  D.idx_b.release();
  D.ent_atom.release();
  D.ent_code.release();
  D.ent_cnt.release();
  D.ent_off.release();
  D.sendlist.release();
  D.type_tmp.release();
  D.tag_tmp.release();
  D.key_a.release();
  D.key_b.release();
  D.sendshift.release();
  D.v_tmp.release();
  D.xq_tmp.release();
  if (D.ev_moved) (void) hipEventDestroy(D.ev_moved);
  D.ev_moved = nullptr;
  D.on = false;
}

extern "C" {

int mdp_dd_setup(mdp_ctx *c, const mdp_dd_config *cfg)
{
  if (!c || !cfg) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  MdpDomain &D = c->dd;
  DdGeom &G = D.G;
  for (int d = 0; d < 3; d++) {
    if (cfg->procgrid[d] < 1 || cfg->procgrid[d] > 64) return mdp_fail(c, MDP_EINVAL, "mdp_dd_setup: bad processor grid");
    G.g[d] = cfg->procgrid[d];
    G.lo[d] = cfg->boxlo[d];
  }
  G.nranks = G.g[0] * G.g[1] * G.g[2];
  if (G.nranks > 4096 || cfg->rank < 0 || cfg->rank >= G.nranks) return mdp_fail(c, MDP_EINVAL, "mdp_dd_setup: bad rank");
  if (!(cfg->h[0] > 0.0 && cfg->h[1] > 0.0 && cfg->h[2] > 0.0) || !(cfg->cutghost > 0.0))
    return mdp_fail(c, MDP_EINVAL, "mdp_dd_setup: box edges and ghost cutoff must be positive");
  G.rank = cfg->rank;
  G.me[2] = cfg->rank % G.g[2];
  G.me[1] = (cfg->rank / G.g[2]) % G.g[1];
  G.me[0] = cfg->rank / (G.g[1] * G.g[2]);
  for (int k = 0; k < 6; k++) G.h[k] = cfg->h[k];
  // Domain::set_global_box: h_inv of the upper-triangular h
  G.hinv[0] = 1.0 / G.h[0];
  G.hinv[1] = 1.0 / G.h[1];
  G.hinv[2] = 1.0 / G.h[2];
  G.hinv[3] = -G.h[3] / (G.h[1] * G.h[2]);
  G.hinv[4] = (G.h[3] * G.h[5] - G.h[1] * G.h[4]) / (G.h[0] * G.h[1] * G.h[2]);
  G.hinv[5] = -G.h[5] / (G.h[0] * G.h[1]);
  // Comm::setup, triclinic: ghost cutoff in lamda units = cut * |row d of h_inv|
  G.cutl[0] = cfg->cutghost * sqrt(G.hinv[0] * G.hinv[0] + G.hinv[5] * G.hinv[5] + G.hinv[4] * G.hinv[4]);
  G.cutl[1] = cfg->cutghost * sqrt(G.hinv[1] * G.hinv[1] + G.hinv[3] * G.hinv[3]);
  G.cutl[2] = cfg->cutghost * G.hinv[2];
  for (int d = 0; d < 3; d++) {
    G.nonper[d] = cfg->nonperiodic[d] ? 1 : 0;
    G.ns[d] = G.nonper[d] ? 0 : (int) floor(G.cutl[d]) + 1;
    if (G.ns[d] > 3 || (2 * G.ns[d] + 1) * G.g[d] > 1000000)
      return mdp_fail(c, MDP_EINVAL, "mdp_dd_setup: the ghost cutoff spans more than three box lengths");
    // at most 12 (shift, brick) hits per dimension: (1/g + 2 cutl) * g bricks-widths, one hit each
    if ((1.0 + 2.0 * G.cutl[d] * G.g[d]) + 1.0 > 12.0)
      return mdp_fail(c, MDP_EINVAL, "mdp_dd_setup: bricks are too thin for the ghost cutoff (dimension %d)", d);
  }
  G.self_remote = cfg->self_remote ? 1 : 0;
  D.cutghost = cfg->cutghost;
  D.mig_send.assign(G.nranks, 0);
  D.bord_send.assign(G.nranks, 0);
  D.bord_recv.assign(G.nranks, 0);
  // counters: [0,n) leaver counts | [n,2n) segment offsets | [2n,3n) cursors | 16 map words | [n+1] class starts
  MDP_HIP(c, D.counters.reserve((size_t) 5 * G.nranks + 64));
  if (!D.ev_moved) MDP_HIP(c, hipEventCreateWithFlags(&D.ev_moved, hipEventDisableTiming));
  D.moved_pending = false;
  D.on = true;
  return MDP_OK;
}

// ---- reneighboring, phase 1: remap + who leaves -------------------------------------------------------------
int mdp_dd_migrate_begin(mdp_ctx *c, int *send_counts)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  const DdGeom &G = D.G;
  hipStream_t st = c->stream;
  const int n = c->nlocal;
  D.nlocal_old = n;
  D.nghost_old = c->nghost;
  MDP_HIP(c, D.dest.reserve((size_t) n + 1));
  MDP_HIP(c, hipMemsetAsync(D.counters.p, 0, sizeof(int) * (4 * G.nranks + 16), st));
  if (n) dd_remap_kernel<<<nblk(n), 256, 0, st>>>(G, n, c->xq.p, D.dest.p, D.counters.p);
  MDP_HIP(c, hipGetLastError());
  MDP_TRY(mdp_read_one(c, D.counters.p, sizeof(int) * G.nranks, D.mig_send.data()));
  D.mig_send[G.rank] = 0;
  D.mig_total = 0;
  for (int q = 0; q < G.nranks; q++) D.mig_total += D.mig_send[q];
  if (send_counts)
    for (int q = 0; q < G.nranks; q++) send_counts[q] = D.mig_send[q];
  return MDP_OK;
}

// d_buf: 8 doubles per leaving atom, segments in rank order (send_counts of mdp_dd_migrate_begin)
int mdp_dd_migrate_pack(mdp_ctx *c, double *d_buf)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  const DdGeom &G = D.G;
  hipStream_t st = c->stream;
  if (!D.mig_total) return MDP_OK;
  if (!d_buf) return mdp_fail(c, MDP_EINVAL, "mdp_dd_migrate_pack: no buffer for %d leaving atoms", D.mig_total);
  std::vector<int> off(2 * (size_t) G.nranks, 0);
  for (int q = 1; q < G.nranks; q++) off[q] = off[q - 1] + D.mig_send[q - 1];
  int *seg = D.counters.p + G.nranks, *cur = D.counters.p + 2 * G.nranks;
  MDP_TRY(mdp_write_small(c, seg, off.data(), sizeof(int) * 2 * G.nranks)); // cursors = 0
  dd_pack_leavers_kernel<<<nblk(D.nlocal_old), 256, 0, st>>>(D.nlocal_old, G.rank, D.dest.p, seg, cur, c->xq.p, c->v.p,
                                                             c->type.p, c->tag.p, d_buf);
  MDP_HIP(c, hipGetLastError());
  MDP_HIP(c, hipStreamSynchronize(st)); // `off` is a host temporary
  return MDP_OK;
}

// arrivals appended, owned atoms re-ordered along the Hilbert curve of the brick; nlocal changes here
int mdp_dd_migrate_end(mdp_ctx *c, int narrive, const double *d_buf)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  const DdGeom &G = D.G;
  hipStream_t st = c->stream;
  if (narrive < 0 || (narrive > 0 && !d_buf)) return mdp_fail(c, MDP_EINVAL, "mdp_dd_migrate_end: bad arrival buffer");
  const int nold = D.nlocal_old, ntot = nold + narrive, nnew = ntot - D.mig_total;
  if ((long long) ntot >= (1ll << 29)) return mdp_fail(c, MDP_EINVAL, "too many atoms for NEIGHMASK");
  MDP_HIP(c, c->xq.reserve((size_t) ntot + 1, true, st));
  MDP_HIP(c, c->v.reserve((size_t) 3 * ntot + 3, true, st));
  MDP_HIP(c, c->type.reserve((size_t) ntot + 32, true, st));
  MDP_HIP(c, c->tag.reserve((size_t) ntot + 1, true, st));
  MDP_HIP(c, D.dest.reserve((size_t) ntot + 1, true, st));
  MDP_HIP(c, c->mass_type.reserve(32));
  MDP_HIP(c, hipMemcpyAsync(c->mass_type.p, c->h_mass, sizeof(double) * 16, hipMemcpyHostToDevice, st));
  // the type -> element map rides behind the types (mdp_pack_xq); park a copy where the kernels below can read it
  int *d_map = D.counters.p + 3 * G.nranks;
  MDP_HIP(c, hipMemcpyAsync(d_map, c->map, sizeof(int) * 16, hipMemcpyHostToDevice, st));
  if (narrive)
    dd_unpack_arrivals_kernel<<<nblk(narrive), 256, 0, st>>>(narrive, nold, d_buf, d_map, c->xq.p, c->v.p, c->type.p,
                                                             c->tag.p, D.dest.p, G.rank);
  MDP_HIP(c, D.key_a.reserve((size_t) ntot + 1));
  MDP_HIP(c, D.key_b.reserve((size_t) ntot + 1));
  MDP_HIP(c, D.idx_a.reserve((size_t) ntot + 1));
  MDP_HIP(c, D.idx_b.reserve((size_t) ntot + 1));
  // the permuted copies become the live arrays (pointer swap): give them room for the ghosts that follow
  const size_t room = (size_t) nnew + (size_t) D.nghost_old + (size_t) D.nghost_old / 16;
  MDP_HIP(c, D.xq_tmp.reserve(room + 1));
  MDP_HIP(c, D.v_tmp.reserve((size_t) 3 * nnew + 3));
  MDP_HIP(c, D.type_tmp.reserve(room + 32));
  MDP_HIP(c, D.tag_tmp.reserve(room + 1));
  MDP_HIP(c, c->rmass.reserve((size_t) nnew + 1));
  if (ntot) {
    const int shell_last = c->cfg.style == 2;
    dd_order_key_kernel<<<nblk(ntot), 256, 0, st>>>(G, ntot, c->xq.p, c->tag.p, D.dest.p, D.key_a.p, D.idx_a.p,
                                                    shell_last);
    MDP_HIP(c, hipGetLastError());
    MDP_TRY(sort_u64(c, D.key_a.p, D.key_b.p, D.idx_a.p, D.idx_b.p, (size_t) ntot, 64));
    if (c->cfg.style == 1) { // rebomos: element-sorted runs of 32 atoms for the one-atom-row tile lists
      MDP_TRY(mdp_chunk_by_element(c, ntot, nnew, D.idx_b.p, D.idx_a.p, c->xq.p, nullptr, nullptr));
      swap_buf(D.idx_a, D.idx_b);
    }
  }
  if (nnew) {
    dd_permute_kernel<<<nblk(nnew), 256, 0, st>>>(nnew, D.idx_b.p, c->xq.p, c->v.p, c->type.p, c->tag.p, c->mass_type.p,
                                                  D.xq_tmp.p, D.v_tmp.p, D.type_tmp.p, D.tag_tmp.p, c->rmass.p);
    MDP_HIP(c, hipGetLastError());
  }
  swap_buf(c->xq, D.xq_tmp);
  swap_buf(c->v, D.v_tmp);
  swap_buf(c->type, D.type_tmp);
  swap_buf(c->tag, D.tag_tmp);
  c->nlocal = nnew;
  c->nghost = 0;
  c->nall = nnew;
  c->cfg.nlocal = nnew;
  c->neigh_set = false;
  c->rebo_packed = false;
  return MDP_OK;
}

// ---- reneighboring, phase 2: ghost entries of every brick ----------------------------------------------------
int mdp_dd_borders_begin(mdp_ctx *c, int *send_counts)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  const DdGeom &G = D.G;
  hipStream_t st = c->stream;
  const int n = c->nlocal;
  MDP_HIP(c, D.ent_cnt.reserve((size_t) n + 2));
  MDP_HIP(c, D.ent_off.reserve((size_t) n + 2));
  int nent = 0;
  if (n) {
    dd_border_kernel<false><<<nblk(n), 256, 0, st>>>(G, n, c->xq.p, D.ent_cnt.p, nullptr, nullptr, nullptr, nullptr);
    MDP_HIP(c, hipGetLastError());
    MDP_TRY(mdp_scan_exclusive_int(c, D.ent_cnt.p, D.ent_off.p, n));
    MDP_TRY(mdp_read_one(c, D.ent_off.p + n, sizeof(int), &nent));
  }
  D.nent = nent;
  const int ncls = G.nranks + 1;
  std::vector<int> start(ncls, -1);
  if (nent) {
    MDP_HIP(c, D.ent_atom.reserve((size_t) nent));
    MDP_HIP(c, D.ent_code.reserve((size_t) nent));
    MDP_HIP(c, D.key_a.reserve((size_t) nent));
    MDP_HIP(c, D.key_b.reserve((size_t) nent));
    MDP_HIP(c, D.idx_a.reserve((size_t) nent));
    MDP_HIP(c, D.idx_b.reserve((size_t) nent));
    dd_border_kernel<true><<<nblk(n), 256, 0, st>>>(G, n, c->xq.p, nullptr, D.ent_off.p, D.ent_atom.p, D.ent_code.p,
                                                    D.key_a.p);
    MDP_HIP(c, hipGetLastError());
    // idx_a = 0 .. nent-1
    MDP_HIP(c, rocprim::transform(rocprim::counting_iterator<int>(0), D.idx_a.p, (size_t) nent,
                                  [] __device__(int v) -> int { return v; }, st));
    int bits = 33;
    while ((1ll << (bits - 32)) < ncls + 1) bits++;
    MDP_TRY(sort_u64(c, D.key_a.p, D.key_b.p, D.idx_a.p, D.idx_b.p, (size_t) nent, bits));
    int *d_start = D.counters.p + 3 * G.nranks + 16; // [nranks + 1] class starts
    MDP_HIP(c, hipMemsetAsync(d_start, 0xFF, sizeof(int) * ncls, st));
    dd_class_start_kernel<<<nblk(nent), 256, 0, st>>>(nent, D.key_b.p, d_start);
    MDP_HIP(c, hipGetLastError());
    MDP_TRY(mdp_read_one(c, d_start, sizeof(int) * ncls, start.data()));
  }
  std::vector<int> cnt(ncls, 0);
  int next = nent;
  for (int k = ncls - 1; k >= 0; k--)
    if (start[k] >= 0) {
      cnt[k] = next - start[k];
      next = start[k];
    }
  D.nself = cnt[0];
  D.nsend = nent - D.nself;
  for (int q = 0; q < G.nranks; q++) D.bord_send[q] = cnt[q + 1];
  if (D.bord_send[G.rank] != 0 && !G.self_remote)
    return mdp_fail(c, MDP_EINVAL, "dd: internal error (self entries in the send list)");
  MDP_HIP(c, c->ghost_owner.reserve((size_t) D.nself + 1));
  MDP_HIP(c, c->ghost_shift.reserve((size_t) 3 * D.nself + 3));
  MDP_HIP(c, D.sendlist.reserve((size_t) D.nsend + 1));
  MDP_HIP(c, D.sendshift.reserve((size_t) 3 * D.nsend + 3));
  if (nent)
    dd_emit_kernel<<<nblk(nent), 256, 0, st>>>(G, nent, D.nself, D.idx_b.p, D.ent_atom.p, D.ent_code.p, c->ghost_owner.p,
                                               c->ghost_shift.p, D.sendlist.p, D.sendshift.p);
  MDP_HIP(c, hipGetLastError());
  if (send_counts)
    for (int q = 0; q < G.nranks; q++) send_counts[q] = D.bord_send[q];
  return MDP_OK;
}

// d_buf: 6 doubles per send-list entry {x + shift, element, type, tag}, segments in rank order
int mdp_dd_borders_pack(mdp_ctx *c, double *d_buf)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  if (!D.nsend) return MDP_OK;
  if (!d_buf) return mdp_fail(c, MDP_EINVAL, "mdp_dd_borders_pack: no buffer for %d entries", D.nsend);
  dd_pack_border_kernel<<<nblk(D.nsend), 256, 0, c->stream>>>(D.nsend, D.sendlist.p, D.sendshift.p, c->xq.p, c->type.p,
                                                              c->tag.p, d_buf);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// recv_counts[nranks]: border entries received from each rank (in rank order in d_buf); fixes nghost and every
// per-atom array, the bin-grid bounds, and leaves the sub-domain ready for mdp_md_build_neighbors
int mdp_dd_borders_end(mdp_ctx *c, const int *recv_counts, const double *d_buf)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  const DdGeom &G = D.G;
  hipStream_t st = c->stream;
  int nrecv = 0;
  for (int q = 0; q < G.nranks; q++) {
    const int r = recv_counts ? recv_counts[q] : 0;
    if (r < 0 || (q == G.rank && r != 0 && !G.self_remote))
      return mdp_fail(c, MDP_EINVAL, "mdp_dd_borders_end: bad receive counts");
    D.bord_recv[q] = r;
    nrecv += r;
  }
  if (nrecv && !d_buf) return mdp_fail(c, MDP_EINVAL, "mdp_dd_borders_end: no buffer for %d ghosts", nrecv);
  D.nrecv = nrecv;
  const int nlocal = c->nlocal, nghost = D.nself + nrecv, nall = nlocal + nghost;
  if ((long long) nall >= (1ll << 29)) return mdp_fail(c, MDP_EINVAL, "too many atoms for NEIGHMASK");
  MDP_TRY(dd_reserve_atoms(c, nlocal, nghost));
  if (D.nself)
    dd_self_ghost_kernel<<<nblk(D.nself), 256, 0, st>>>(D.nself, nlocal, c->ghost_owner.p, c->ghost_shift.p, c->xq.p,
                                                        c->type.p, c->tag.p);
  if (nrecv)
    dd_unpack_border_kernel<<<nblk(nrecv), 256, 0, st>>>(nrecv, nlocal + D.nself, d_buf, c->xq.p, c->type.p, c->tag.p,
                                                         c->ghost_owner.p, c->ghost_shift.p, D.nself);
  MDP_HIP(c, hipGetLastError());
  MDP_HIP(c, hipMemcpyAsync(c->type.p + nall, c->map, sizeof(int) * 16, hipMemcpyHostToDevice, st));
  MDP_HIP(c, hipMemsetAsync(c->f.p, 0, sizeof(double) * 3 * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->eatom.p, 0, sizeof(double) * nall, st));
  MDP_HIP(c, hipMemsetAsync(c->fp.p, 0, sizeof(double) * nall, st));
  c->nghost = nghost;
  c->nall = nall;
  c->cfg.nlocal = nlocal;
  c->cfg.nghost = nghost;
  c->cfg.nghost_self = D.nself;
  c->remote_start = nlocal + D.nself;
  c->atoms_set = true;
  // bin-grid bounds: Cartesian hull of the extended brick, padded by what atoms may drift before the next build
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int k = 0; k < 8; k++) {
    double lam[3];
    for (int d = 0; d < 3; d++)
      lam[d] = ((k >> d) & 1) ? (double) (G.me[d] + 1) / G.g[d] + G.cutl[d] : (double) G.me[d] / G.g[d] - G.cutl[d];
    const double x[3] = {G.h[0] * lam[0] + G.h[5] * lam[1] + G.h[4] * lam[2] + G.lo[0],
                         G.h[1] * lam[1] + G.h[3] * lam[2] + G.lo[1], G.h[2] * lam[2] + G.lo[2]};
    for (int d = 0; d < 3; d++) {
      lo[d] = x[d] < lo[d] ? x[d] : lo[d];
      hi[d] = x[d] > hi[d] ? x[d] : hi[d];
    }
  }
  const double pad = 1.0 + c->cfg.skin;
  for (int d = 0; d < 3; d++) {
    c->bbox_lo[d] = c->cfg.bbox_lo[d] = lo[d] - pad;
    c->bbox_hi[d] = c->cfg.bbox_hi[d] = hi[d] + pad;
  }
  D.moved_pending = false; // a check in flight refers to the old arrays
  D.reneighbors++;
  return MDP_OK;
}

// the whole sequence for a run on ONE GPU (no transport): remap, order, self-image ghosts, neighbor lists
int mdp_dd_reneighbor(mdp_ctx *c)
{
  MDP_TRY(dd_require(c));
  if (c->dd.G.nranks != 1) return mdp_fail(c, MDP_EINVAL, "mdp_dd_reneighbor: multi-rank runs use the phased calls");
  MDP_TRY(mdp_dd_migrate_begin(c, nullptr));
  MDP_TRY(mdp_dd_migrate_end(c, 0, nullptr));
  MDP_TRY(mdp_dd_borders_begin(c, nullptr));
  MDP_TRY(mdp_dd_borders_end(c, nullptr, nullptr));
  return mdp_md_build_neighbors_impl(c);
}

// out[0]=nlocal [1]=nself (periodic self-image ghosts) [2]=nsend [3]=nrecv (remote ghosts) [4]=reneighborings so far
// [5]=atoms that left at the last one; send_counts / recv_counts [nranks] of the per-step halo (may be NULL)
int mdp_dd_info(mdp_ctx *c, long long out[8], int *send_counts, int *recv_counts)
{
  MDP_TRY(dd_require(c));
  MdpDomain &D = c->dd;
  if (out) {
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[0] = c->nlocal;
    out[1] = D.nself;
    out[2] = D.nsend;
    out[3] = D.nrecv;
    out[4] = D.reneighbors;
    out[5] = D.mig_total;
    out[6] = D.G.nranks;
    out[7] = D.G.rank;
  }
  for (int q = 0; q < D.G.nranks; q++) {
    if (send_counts) send_counts[q] = D.bord_send[q];
    if (recv_counts) recv_counts[q] = D.bord_recv[q];
  }
  return MDP_OK;
}

// ---- per step: forward positions (Comm::forward_comm) with the library's own send list -------------------------
int mdp_dd_forward_pack(mdp_ctx *c, double *d_buf)
{
  MDP_TRY(dd_require(c));
  return mdp_md_pack_x(c, c->dd.nsend, c->dd.sendlist.p, c->dd.sendshift.p, d_buf);
}

int mdp_dd_forward_unpack(mdp_ctx *c, const double *d_buf)
{
  MDP_TRY(dd_require(c));
  return mdp_md_unpack_x(c, c->dd.nself, c->dd.nrecv, d_buf);
}

// AEAM: scalar forward exchange of fp, reverse exchange of the forces angular centres put on ghosts
int mdp_dd_forward_scalar_pack(mdp_ctx *c, double *d_buf)
{
  MDP_TRY(dd_require(c));
  return mdp_md_pack_scalar(c, 0, c->dd.nsend, c->dd.sendlist.p, d_buf);
}

int mdp_dd_forward_scalar_unpack(mdp_ctx *c, const double *d_buf)
{
  MDP_TRY(dd_require(c));
  return mdp_md_unpack_scalar(c, 0, c->dd.nself, c->dd.nrecv, d_buf);
}

int mdp_dd_reverse_pack(mdp_ctx *c, double *d_buf)
{
  MDP_TRY(dd_require(c));
  return mdp_md_pack_ghost_f(c, c->dd.nself, c->dd.nrecv, d_buf);
}

int mdp_dd_reverse_unpack(mdp_ctx *c, const double *d_buf)
{
  MDP_TRY(dd_require(c));
  return mdp_md_unpack_add_f(c, c->dd.nsend, c->dd.sendlist.p, d_buf);
}

// `neigh_modify every 1 delay 0 check yes` without a host round trip: *moved = result of the check launched by
// the PREVIOUS call (0 right after a reneighboring), then a new check of the current positions is launched.
// The trigger is half the skin minus a margin that covers the one step of extra motion; *dangerous counts checks
// that saw an atom beyond half the skin itself.  Call once per step, after the positions were advanced.
int mdp_md_moved_async(mdp_ctx *c, int *moved, int *dangerous)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  MdpDomain &D = c->dd;
  hipStream_t st = c->stream;
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
  if (m || !c->nlocal || !c->neigh_set) return MDP_OK; // the caller rebuilds now: nothing to check until then
  h[0] = h[1] = 0;
  const double hard = 0.5 * c->cfg.skin;
  double trig = hard - 0.1 * mdp_margin_scale(c);
  if (trig < 0.5 * hard) trig = 0.5 * hard;
  const int grid = nblk(c->nlocal) < 1024 ? nblk(c->nlocal) : 1024;
  dd_moved_kernel<<<grid, 256, 0, st>>>(c->nlocal, trig * trig, hard * hard, c->xq.p, c->xhold.p, h);
  MDP_HIP(c, hipGetLastError());
  MDP_HIP(c, hipEventRecord(D.ev_moved, st));
  D.ev_moved_ref = D.ev_moved;
  D.moved_pending = true;
  return MDP_OK;
}

// mdp_md_initial_integrate (or, with_final, mdp_md_final_initial_integrate) and mdp_md_moved_async in one call and
// one pass over the atoms: *moved / *dangerous = result of the check launched by the previous call of either kind,
// then the positions are advanced and -- unless the caller is about to reneighbor anyway -- checked in the same kernel.
int mdp_md_integrate_check(mdp_ctx *c, int with_final, int *moved, int *dangerous)
{
  if (!c) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
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
  if (m || !c->nlocal || !c->neigh_set) return mdp_md_advance(c, with_final != 0, nullptr, 0.0, 0.0);
  h[0] = h[1] = 0;
  const double hard = 0.5 * c->cfg.skin;
  double trig = hard - 0.1 * mdp_margin_scale(c);
  if (trig < 0.5 * hard) trig = 0.5 * hard;
  MDP_TRY(mdp_md_advance(c, with_final != 0, h, trig * trig, hard * hard));
  // the integrate kernel wrote h; when mdp_md_advance has recorded the style-check event behind that kernel (no remote
  // ghosts), that event serves this reader too
  if (c->sflag_armed && c->sflag_committed[c->sflag_set]) {
    D.ev_moved_ref = c->ev_sflag[c->sflag_set];
  } else {
    MDP_HIP(c, hipEventRecord(D.ev_moved, c->stream));
    D.ev_moved_ref = D.ev_moved;
  }
  D.moved_pending = true;
  return MDP_OK;
}

// owned atoms' integer properties in device order ("tag", "type"); the device re-orders atoms at every reneighboring
int mdp_md_download_int(mdp_ctx *c, const char *name, int *out)
{
  if (!c || !name || !out) return MDP_EINVAL;
  if (!c->md) return mdp_fail(c, MDP_ESTATE, "mdp_md_setup not called");
  const int *src = !strcmp(name, "tag") ? c->tag.p : (!strcmp(name, "type") ? c->type.p : nullptr);
  MDP_HIP(c, hipSetDevice(c->device));
  if (!strcmp(name, "tile_nu")) { // diagnostics: {members of the union, Mo members} of every tile, 2 * ntile <= nlocal ints
    if (2 * c->ntile > c->nlocal) return mdp_fail(c, MDP_EINVAL, "mdp_md_download_int: tile_nu needs 2 * %d ints", c->ntile);
    src = c->tile_nu.p;
  }
  // diagnostics: length / first-segment length of every cluster's row of 16-bit entries, as the Lennard-Jones (rebomos)
  // or pair (aeam) tile kernels walk them now -- the pruned rows while a pruning is valid; nclus <= nlocal ints
  const bool rows = !strcmp(name, "lj_len") || !strcmp(name, "lj_split");
  if (rows) {
    if (!c->lj_tiled || !c->nclus) return mdp_fail(c, MDP_ESTATE, "mdp_md_download_int: no tile rows");
    const bool pr = c->prune_valid;
    src = !strcmp(name, "lj_len") ? (pr ? c->lj_len_in.p : nullptr) : (pr ? c->lj_split_in.p : c->lj_split.p);
    if (!src) return mdp_fail(c, MDP_ESTATE, "mdp_md_download_int: lj_len needs pruned rows (the rows as built: mdp_rebomos_list_info)");
  }
  if (!src) return mdp_fail(c, MDP_EINVAL, "mdp_md_download_int: unknown array '%s'", name);
  // through the context's pinned buffer, complete on return (no asynchronous copy into the caller's pageable memory)
  const size_t n = !strcmp(name, "tile_nu") ? (size_t) 2 * c->ntile : (rows ? (size_t) c->nclus : (size_t) c->nlocal);
  if (n) {
    MDP_TRY(mdp_host_pinned_reserve(c, (n * sizeof(int) + 7) / 8 + 1));
    MDP_HIP(c, hipMemcpyAsync(c->h_down, src, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    MDP_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->h_down, sizeof(int) * n);
  }
  return MDP_OK;
}

} // extern "C"
