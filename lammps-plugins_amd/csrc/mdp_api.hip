// mdp_api.hip -- C-ABI entry points of libmdpair_hip.so: lifecycle, potentials, host-mode data
// movement (what a LAMMPS Pair::compute() hands over) and the host-mode compute calls.
#include "mdp_common.h"

#include <rocprim/rocprim.hpp>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>

#include <cmath>

void mdp_rebomos_fill_dev(mdp_ctx *c, double skin);

int mdp_fail(mdp_ctx *c, int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

static int small_reserve(mdp_ctx *c, size_t bytes)
{
  if (bytes <= c->h_small_cap) return MDP_OK;
  if (c->h_small) (void) hipHostFree(c->h_small);
  c->h_small = nullptr;
  c->h_small_cap = 0;
  const size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 2;
  MDP_HIP(c, hipHostMalloc((void **) &c->h_small, cap, hipHostMallocDefault));
  c->h_small_cap = cap;
  return MDP_OK;
}

int mdp_read_small(mdp_ctx *c, const MdpRead *r, int n)
{
  size_t total = 0;
  for (int k = 0; k < n; k++) total += (r[k].bytes + 15) & ~(size_t) 15;
  MDP_TRY(small_reserve(c, total));
  size_t at = 0;
  for (int k = 0; k < n; k++) {
    if (r[k].bytes) MDP_HIP(c, hipMemcpyAsync(c->h_small + at, r[k].d_src, r[k].bytes, hipMemcpyDeviceToHost, c->stream));
    at += (r[k].bytes + 15) & ~(size_t) 15;
  }
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  at = 0;
  for (int k = 0; k < n; k++) {
    if (r[k].bytes) memcpy(r[k].h_dst, c->h_small + at, r[k].bytes);
    at += (r[k].bytes + 15) & ~(size_t) 15;
  }
  return MDP_OK;
}

int mdp_read_one(mdp_ctx *c, const void *d_src, size_t bytes, void *h_dst)
{
  const MdpRead r = {d_src, bytes, h_dst};
  return mdp_read_small(c, &r, 1);
}

int mdp_write_small(mdp_ctx *c, void *d_dst, const void *h_src, size_t bytes)
{
  if (!bytes) return MDP_OK;
  MDP_HIP(c, hipStreamSynchronize(c->stream)); // (a read or write staged earlier has left the scratch buffer)
  MDP_TRY(small_reserve(c, bytes));
  memcpy(c->h_small, h_src, bytes);
  MDP_HIP(c, hipMemcpyAsync(d_dst, c->h_small, bytes, hipMemcpyHostToDevice, c->stream));
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  return MDP_OK;
}

void mdp_time_mark(mdp_ctx *c, int k)
{
  if (!c->timing) return;
  if (!c->ev_made) {
    for (int i = 0; i < 8; i++) (void) hipEventCreate(&c->ev[i]);
    c->ev_made = true;
  }
  (void) hipEventRecord(c->ev[k], c->stream);
  if (k == 0) c->ev_marks = 0;
  if (k + 1 > c->ev_marks) c->ev_marks = k + 1;
  c->timing_spans = false;
}

static void span_events(mdp_ctx *c)
{
  if (c->ev_sb[0]) return;
  for (int i = 0; i < 8; i++) {
    (void) hipEventCreate(&c->ev_sb[i]);
    (void) hipEventCreate(&c->ev_se[i]);
  }
}

void mdp_span_begin(mdp_ctx *c, int k)
{
  if (!c->timing) return;
  span_events(c);
  if (!c->timing_spans) c->span_mask = 0;
  c->timing_spans = true;
  (void) hipEventRecord(c->ev_sb[k], c->stream);
  c->span_mask &= ~(1u << k);
}

void mdp_span_end(mdp_ctx *c, int k)
{
  if (!c->timing || !c->ev_sb[0]) return;
  (void) hipEventRecord(c->ev_se[k], c->stream);
  c->span_mask |= 1u << k;
}

namespace {

__global__ void pack_xq_kernel(int n, const double *__restrict__ x3, const int *__restrict__ type,
                               const int *__restrict__ map /* [ntypes+1] on device or null */, double4 *__restrict__ xq)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double w;
  if (type) {
    const int t = type[i];
    w = (double) (map ? map[t] : t - 1);
  } else {
    w = xq[i].w;
  }
  xq[i] = make_double4(x3[3 * (size_t) i], x3[3 * (size_t) i + 1], x3[3 * (size_t) i + 2], w);
}

// ---- host mode: the device keeps its own spatial order ------------------------------------------------
// A host hands atoms over in its order; the tile lists (and every gather) want neighbours in space to be
// neighbours in memory.  Owned atoms, and separately the ghosts, are therefore sorted along a Hilbert curve
// on the device at every mdp_set_atoms_host; positions are permuted on upload and forces / per-atom
// results permuted back before download -- invisible to the host.
__global__ void hilbert_key_kernel(const int nall, const int nlocal, const double *__restrict__ x3, const double lo0,
                                   const double lo1, const double lo2, const double sc0, const double sc1,
                                   const double sc2, unsigned *__restrict__ keys, int *__restrict__ idx)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nall) return;
  constexpr int B = 10; // bits per dimension: 1024 cells over the bounding box
  const double lo[3] = {lo0, lo1, lo2}, sc[3] = {sc0, sc1, sc2};
  unsigned X[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    int g = (int) ((x3[3 * (size_t) i + d] - lo[d]) * sc[d]);
    g = g < 0 ? 0 : (g > (1 << B) - 1 ? (1 << B) - 1 : g);
    X[d] = (unsigned) g;
  }
  unsigned key = mdp_hilbert30(X[0], X[1], X[2]);
  if (i >= nlocal) key |= 1u << 30; // ghosts stay behind the owned atoms
  keys[i] = key;
  idx[i] = i;
}

// xq[n] <- atom perm[n] of the host arrays (type null: keep the element already in xq[n].w)
__global__ void pack_xq_perm_kernel(int n, const double *__restrict__ x3, const int *__restrict__ type,
                                    const int *__restrict__ map, const int *__restrict__ perm,
                                    double4 *__restrict__ xq)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = perm[i];
  double w;
  if (type) {
    const int t = type[o];
    w = (double) (map ? map[t] : t - 1);
  } else {
    w = xq[i].w;
  }
  xq[i] = make_double4(x3[3 * (size_t) o], x3[3 * (size_t) o + 1], x3[3 * (size_t) o + 2], w);
}

// dst (host order) <- src (device order), `w` doubles per atom
__global__ void unpermute_kernel(int n, int w, const int *__restrict__ perm, const double *__restrict__ src,
                                 double *__restrict__ dst)
{
  const long long k = (long long) blockIdx.x * 256 + threadIdx.x;
  if (k >= (long long) n * w) return;
  const int i = (int) (k / w), q = (int) (k % w);
  dst[(size_t) perm[i] * w + q] = src[k];
}

// dst (device order) <- src (host order)
__global__ void permute_kernel(int n, int w, const int *__restrict__ perm, const double *__restrict__ src,
                               double *__restrict__ dst)
{
  const long long k = (long long) blockIdx.x * 256 + threadIdx.x;
  if (k >= (long long) n * w) return;
  const int i = (int) (k / w), q = (int) (k % w);
  dst[k] = src[(size_t) perm[i] * w + q];
}

// ---- host mode on one periodic rank: ghosts as images of owned atoms (Comm::forward_comm done on the device) ----------
// tagmap[tag] = host index of the owned atom with that tag; flag bit 0: a tag out of range or seen twice
__global__ void host_tagmap_kernel(const int nlocal, const int *__restrict__ tag, const int maxtag, int *__restrict__ tagmap,
                                   int *__restrict__ flag)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nlocal) return;
  const int t = tag[i];
  if (t < 1 || t > maxtag || atomicExch(&tagmap[t], i) != -1) atomicOr(flag, 1);
}

__global__ void host_inv_kernel(const int nall, const int *__restrict__ perm, int *__restrict__ inv)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < nall) inv[perm[p]] = p;
}

// ghost at device position nlocal+g: owner (device position) and how many box vectors it sits away from it.  The counts
// must reproduce the uploaded image to 1e-8 A with the box the host handed over (flag bit 1 otherwise: a ghost of
// another rank's atom, a box that is not the one the positions were made with).  perm / inv null: device order = host order
__global__ void host_ghost_owner_kernel(const int nlocal, const int nall, const int *__restrict__ perm,
                                        const int *__restrict__ inv, const int *__restrict__ tag,
                                        const int *__restrict__ type, const int maxtag, const int *__restrict__ tagmap,
                                        const double *__restrict__ x3, const double h0, const double h1, const double h2,
                                        const double h3, const double h4, const double h5, int *__restrict__ owner,
                                        double *__restrict__ count, double *__restrict__ shift, int *__restrict__ flag)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nall - nlocal) return;
  const int hh = perm ? perm[nlocal + g] : nlocal + g;
  const int t = tag[hh];
  const int o = (t >= 1 && t <= maxtag) ? tagmap[t] : -1;
  owner[g] = -1;
  count[3 * (size_t) g] = count[3 * (size_t) g + 1] = count[3 * (size_t) g + 2] = 0.0;
  shift[3 * (size_t) g] = shift[3 * (size_t) g + 1] = shift[3 * (size_t) g + 2] = 0.0;
  if (hh < nlocal || o < 0 || type[o] != type[hh]) {
    atomicOr(flag, 2);
    return;
  }
  const double sx = x3[3 * (size_t) hh] - x3[3 * (size_t) o], sy = x3[3 * (size_t) hh + 1] - x3[3 * (size_t) o + 1],
               sz = x3[3 * (size_t) hh + 2] - x3[3 * (size_t) o + 2];
  const double nz = rint(sz / h2), ny = rint((sy - nz * h3) / h1), nx = rint((sx - ny * h5 - nz * h4) / h0);
  const double ex = sx - (nx * h0 + ny * h5 + nz * h4), ey = sy - (ny * h1 + nz * h3), ez = sz - nz * h2;
  if (!(fabs(ex) < 1e-8 && fabs(ey) < 1e-8 && fabs(ez) < 1e-8) || (nx == 0.0 && ny == 0.0 && nz == 0.0) ||
      fabs(nx) > 64.0 || fabs(ny) > 64.0 || fabs(nz) > 64.0) {
    atomicOr(flag, 2);
    return;
  }
  owner[g] = inv ? inv[o] : o;
  count[3 * (size_t) g] = nx;
  count[3 * (size_t) g + 1] = ny;
  count[3 * (size_t) g + 2] = nz;
  shift[3 * (size_t) g] = sx; // as uploaded: the list build matches images by position (rebomos.hip rev_kernel)
  shift[3 * (size_t) g + 1] = sy;
  shift[3 * (size_t) g + 2] = sz;
}

__global__ void host_tag_dev_kernel(const int nall, const int *__restrict__ perm, const int *__restrict__ tag,
                                    int *__restrict__ tag_dev)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < nall) tag_dev[p] = tag[perm ? perm[p] : p];
}

// x of an image = x of its owner + count . box, with the box of THIS step (the operand order of Comm::forward_comm's
// pack with pbc flags: x + pbc[0]*xprd + pbc[5]*xy + pbc[4]*xz, ...)
__global__ void host_ghost_refresh_kernel(const int nlocal, const int nghost, const int *__restrict__ owner,
                                          const double *__restrict__ count, const double h0, const double h1,
                                          const double h2, const double h3, const double h4, const double h5,
                                          double4 *__restrict__ xq)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nghost) return;
  const double4 xo = xq[owner[g]];
  const double nx = count[3 * (size_t) g], ny = count[3 * (size_t) g + 1], nz = count[3 * (size_t) g + 2];
  double4 x = xq[nlocal + g];
  x.x = xo.x + nx * h0 + ny * h5 + nz * h4;
  x.y = xo.y + ny * h1 + nz * h3;
  x.z = xo.z + nz * h2;
  xq[nlocal + g] = x;
}

__global__ void host_ghost_scalar_kernel(const int nlocal, const int nghost, const int *__restrict__ owner,
                                         double *__restrict__ a)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g < nghost) a[nlocal + g] = a[owner[g]];
}

// Comm::reverse_comm on one periodic rank: what the images collected goes to their owners (several images per owner:
// atomics; the order of the additions is not fixed, the sum is to the last bit or two)
__global__ void host_ghost_fold_kernel(const int nlocal, const int nghost, const int w, const int *__restrict__ owner,
                                       double *__restrict__ a)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nghost) return;
  const int o = owner[g];
  double *ag = a + (size_t) w * (nlocal + g);
  for (int k = 0; k < w; k++)
    if (ag[k] != 0.0) {
      atomicAdd(&a[(size_t) w * o + k], ag[k]);
      ag[k] = 0.0;
    }
}

// start of every compute: energy/virial accumulators (+ their slots), the four flag words, the overflow counter.
// flags[0] (overflow bits of the compute just finished) is folded into the STICKY word flags[4] first: force-only
// steps of a resident run never read the flags, and a truncated neighbour set must still stop the run at the next
// host read (mdp_flags_check), as "Neighbor list overflow" stops the reference (pair_rebomos.cpp:350).
__global__ void acc_zero_kernel(double *__restrict__ acc, const int n, int *__restrict__ flags, int *__restrict__ ovf,
                                const int ovf_stride)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) acc[i] = 0.0;
  if (i == 0) {
    const int f0 = flags[0];
    if (f0) flags[4] |= f0;
    flags[0] = 0;
    if (ovf)
      for (int k = 0; k < MDP_NOVF_LISTS; k++) ovf[(size_t) k * ovf_stride] = 0;
  } else if (i < 4)
    flags[i] = 0;
}

__global__ void add_f_kernel(int n3, const double *__restrict__ src, double *__restrict__ dst)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n3) dst[i] += src[i];
}

__global__ void acc_reduce_kernel(double *__restrict__ acc)
{
  // one wave per quantity k = 0..6: sum the slots
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (k >= 7) return;
  double s = 0.0;
  for (int i = lane; i < MDP_ACC_SLOTS; i += 64) s += acc[MDP_ACC_STRIDE * (1 + i) + k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) acc[k] += s;
}

} // namespace

// host order <-> device (Hilbert) order of per-atom arrays, `w` doubles per atom (host mode with host_sort)
int mdp_to_host_order(mdp_ctx *c, int n, int w, const double *d_src, double *d_dst)
{
  if (n <= 0) return MDP_OK;
  unpermute_kernel<<<(unsigned) (((long long) n * w + 255) / 256), 256, 0, c->stream>>>(n, w, c->host_perm.p, d_src, d_dst);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_to_device_order(mdp_ctx *c, int n, int w, const double *d_src, double *d_dst)
{
  if (n <= 0) return MDP_OK;
  permute_kernel<<<(unsigned) (((long long) n * w + 255) / 256), 256, 0, c->stream>>>(n, w, c->host_perm.p, d_src, d_dst);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

int mdp_acc_begin(mdp_ctx *c, bool any)
{
  // one small kernel instead of three memsets: at half a million atoms per GPU the gaps around copy-engine
  // operations were 13 % of a step
  if (c->acc_prezeroed) { // the integrate kernel of this step did it (mdp_sflag_arm)
    c->acc_prezeroed = false;
    return MDP_OK;
  }
  const int n = any ? MDP_ACC_STRIDE * (1 + MDP_ACC_SLOTS) : MDP_ACC_STRIDE;
  acc_zero_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(c->acc.p, n, c->flags.p, c->ovf.p, c->ovf_stride);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// hflags: 5 words copied from c->flags after the stream was synchronised.  Overflow bits of the last compute
// (word 0) or of any compute since the last read (sticky word 4) stop the caller; the sticky word is cleared.
int mdp_flags_check(mdp_ctx *c, const int *hflags)
{
  const int bits = hflags[0] | hflags[4];
  if (!bits) return MDP_OK;
  (void) hipMemsetAsync(c->flags.p + 4, 0, sizeof(int), c->stream);
  if (bits & 1)
    return mdp_fail(c, MDP_EOVERFLOW, "REBO neighbor count exceeds the lane-group capacity (Neighbor list overflow)");
  if (bits & 2)
    return mdp_fail(c, MDP_EOVERFLOW, "aeam: an angular atom has more in-range neighbours than the LDS tile holds "
                                      "(Neighbor list overflow)");
  return mdp_fail(c, MDP_EOVERFLOW, "Neighbor list overflow (flags %d)", bits);
}

int mdp_acc_end(mdp_ctx *c, bool any)
{
  if (any) {
    acc_reduce_kernel<<<1, 512, 0, c->stream>>>(c->acc.p);
    MDP_HIP(c, hipGetLastError());
  }
  return MDP_OK;
}

// ---- host-mode transfers --------------------------------------------------------------------------------
// The host's x array (handed over every step, 115 MB at 4 M atoms) is NOT page-locked in place by default: the
// library does not own that memory, LAMMPS re-allocates it whenever nmax grows (memory->grow at migration), and a
// registration that outlives the allocation leaves stale pinned ranges behind (or, worse, a range HIP still treats
// as pinned after the address was reused).  Instead the upload goes through two pinned staging buffers owned by
// the context: a few threads copy chunk k+1 into one while the DMA engine drains chunk k from the other.
//
// MDP_HOST_REGISTER=1 opts into in-place registration (one DMA, no staging copy) for hosts that promise to call
// mdp_host_release(ptr) before they free or re-allocate a registered array; every range is re-validated against
// the (pointer, size) of the current call and dropped when either changed.
static void host_unregister_all(mdp_ctx *c)
{
  for (auto &r : c->host_regs)
    if (r.second) (void) hipHostUnregister(const_cast<void *>(r.first));
  c->host_regs.clear();
  (void) hipGetLastError();
}

static bool host_register(mdp_ctx *c, const void *ptr, size_t bytes)
{
  const char *e = getenv("MDP_HOST_REGISTER");
  if (!e || atoi(e) == 0 || c->md || bytes < (8u << 20)) return false;
  for (auto &r : c->host_regs)
    if (r.first == ptr && r.second == bytes) return true; // same array, same extent as last time
  host_unregister_all(c); // pointer or extent changed: the old range may be gone already
  if (hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) == hipSuccess) {
    c->host_regs.emplace_back(ptr, bytes);
    return true;
  }
  (void) hipGetLastError();
  return false;
}

// Host-side copies and adds of the host-mode path (tens of MB per step) are split over a few worker threads.  The
// workers are started once per process and sleep on a condition variable between calls: starting and joining
// eight threads for each of the fifteen 12-16 MB pieces of a step cost about as much as moving the bytes.
// One parallel section at a time; a second caller (another context's thread) runs its pieces itself.
// fork(): the library is not usable in a child forked after first use (neither is the HIP runtime under it); the pool at
// least never touches the mutexes, condition variables and thread handles it inherited -- whatever state the parent's
// threads left them in -- and runs the pieces of a call serially there.
namespace {
class HostWorkers
{
 public:
  static HostWorkers &get()
  {
    static HostWorkers w;
    return w;
  }
  unsigned width() const { return nthreads; }
  // fn(k) for k = 0 .. n-1, the caller taking part
  template <typename F> void run(const unsigned n, const F &fn)
  {
    if (n <= 1 || nthreads <= 1 || (owner != 0 && getpid() != owner) || !busy.try_lock()) {
      for (unsigned k = 0; k < n; k++) fn(k);
      return;
    }
    std::function<void(unsigned)> f = fn;
    {
      std::lock_guard<std::mutex> g(mu);
      start();
      job = &f;
      njobs = n;
      next = done = 0;
      gen++;
    }
    cv.notify_all();
    work();
    {
      std::unique_lock<std::mutex> g(mu);
      cv_done.wait(g, [&] { return done == njobs; });
      job = nullptr;
    }
    busy.unlock();
  }

 private:
  HostWorkers()
  {
    unsigned hc = std::thread::hardware_concurrency(), cap = 8; // (a host usually runs more than this one rank)
    if (const char *e = getenv("MDP_HOST_THREADS")) {
      hc = (unsigned) atoi(e);
      cap = 64;
    }
    nthreads = hc > cap ? cap : (hc < 1 ? 1 : hc);
  }
  ~HostWorkers()
  {
    if (owner != 0 && getpid() != owner) return; // forked child: the handles are the parent's (and `th` is leaked on purpose)
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto &t : *th) t.join();
    delete th;
  }
  void start() // (mu held; never in a forked child: run() has sent it down the serial path)
  {
    if (!th->empty()) return;
    owner = getpid();
    for (unsigned t = 1; t < nthreads; t++)
      th->emplace_back([this] {
        unsigned long seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
          }
          work();
        }
      });
  }
  void work()
  {
    for (;;) {
      unsigned k;
      const std::function<void(unsigned)> *f;
      {
        std::lock_guard<std::mutex> g(mu);
        if (!job || next >= njobs) return;
        k = next++;
        f = job;
      }
      (*f)(k);
      {
        std::lock_guard<std::mutex> g(mu);
        if (++done == njobs) cv_done.notify_all();
      }
    }
  }
  std::mutex busy, mu;
  std::condition_variable cv, cv_done;
  std::vector<std::thread> *th = new std::vector<std::thread>(); // (on the heap: a forked child must not destroy the handles)
  const std::function<void(unsigned)> *job = nullptr;
  unsigned nthreads = 1, njobs = 0, next = 0, done = 0;
  unsigned long gen = 0;
  bool stop = false;
  pid_t owner = 0;
};
} // namespace

static void host_copy_threads(char *dst, const char *src, size_t n)
{
  HostWorkers &W = HostWorkers::get();
  unsigned nt = n >= (1u << 20) ? (unsigned) (n >> 19) : 1; // half a megabyte per thread at least
  nt = nt > W.width() ? W.width() : nt;
  if (nt <= 1) {
    memcpy(dst, src, n);
    return;
  }
  const size_t chunk = ((n + nt - 1) / nt + 4095) & ~(size_t) 4095;
  W.run(nt, [=](unsigned t) {
    const size_t b = t * chunk, e2 = b + chunk < n ? b + chunk : n;
    if (b < e2) memcpy(dst + b, src + b, e2 - b);
  });
}

// host array -> device, asynchronously on the context's stream; the host array may be reused on return
static int host_upload(mdp_ctx *c, void *d_dst, const void *h_src, size_t bytes)
{
  hipStream_t st = c->stream;
  if (!bytes) return MDP_OK;
  if (bytes < (1u << 20) || host_register(c, h_src, bytes)) { // small, or page-locked in place on request
    MDP_HIP(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
    return MDP_OK;
  }
  // 16 MB pieces through two pinned buffers.  The pipeline runs at the rate of the worker copies into the buffers (host
  // memory: 96 MB in 2.1 ms), not of the DMA behind them: a short first piece (2 MB, then 6) or 4 MB pieces throughout
  // only add worker wake-ups (2.25 and 3.1 ms, same box).
  constexpr size_t kChunk = 16u << 20;
  if (!c->h_up[0]) {
    for (int k = 0; k < MDP_UP_RING; k++) {
      MDP_HIP(c, hipHostMalloc((void **) &c->h_up[k], kChunk, hipHostMallocDefault));
      MDP_HIP(c, hipEventCreateWithFlags(&c->ev_up[k], hipEventDisableTiming));
    }
  }
  // arrays below 64 MB go in about eight pieces of at least 2 MB, so that a 24 MB array (1 M atoms) is not two pieces
  // that barely overlap (0.82 -> 0.70 ms); large arrays in 16 MB pieces (measured above)
  size_t piece = kChunk;
  if (bytes < (64u << 20)) {
    piece = ((bytes / 8 + (1u << 20) - 1) >> 20) << 20;
    piece = piece < (2u << 20) ? (2u << 20) : (piece > kChunk ? kChunk : piece);
  }
  int k = 0;
  for (size_t off = 0; off < bytes; off += piece, k = (k + 1) % MDP_UP_RING) {
    const size_t n = bytes - off < piece ? bytes - off : piece;
    if (off >= MDP_UP_RING * piece) MDP_HIP(c, hipEventSynchronize(c->ev_up[k])); // the DMA out of this buffer has finished
    host_copy_threads(c->h_up[k], (const char *) h_src + off, n);
    MDP_HIP(c, hipMemcpyAsync((char *) d_dst + off, c->h_up[k], n, hipMemcpyHostToDevice, st));
    MDP_HIP(c, hipEventRecord(c->ev_up[k], st));
  }
  return MDP_OK;
}

int mdp_host_upload(mdp_ctx *c, void *d_dst, const void *h_src, size_t bytes) { return host_upload(c, d_dst, h_src, bytes); }

int mdp_host_pinned_reserve(mdp_ctx *c, size_t ndoubles)
{
  if (ndoubles <= c->h_down_cap) return MDP_OK;
  if (c->h_down) (void) hipHostFree(c->h_down);
  c->h_down = nullptr;
  c->h_down_cap = 0;
  MDP_HIP(c, hipHostMalloc((void **) &c->h_down, sizeof(double) * (ndoubles + ndoubles / 8 + 64), hipHostMallocDefault));
  c->h_down_cap = ndoubles + ndoubles / 8 + 64;
  return MDP_OK;
}

// dst[k] += src[k], split over a few threads for arrays that take milliseconds
void mdp_host_add(double *dst, const double *src, size_t n)
{
  HostWorkers &W = HostWorkers::get();
  unsigned nt = n >= (1u << 18) ? (unsigned) (n >> 16) : 1; // half a megabyte per thread at least
  nt = nt > W.width() ? W.width() : nt;
  if (nt <= 1) {
    for (size_t k = 0; k < n; k++) dst[k] += src[k];
    return;
  }
  const size_t chunk = ((n + nt - 1) / nt + 7) & ~(size_t) 7;
  W.run(nt, [=](unsigned t) {
    const size_t b = t * chunk, e = b + chunk < n ? b + chunk : n;
    for (size_t k = b; k < e; k++) dst[k] += src[k];
  });
}

// h_dst[0..n) += d_src[0..n) through the pinned buffer h_stage.  Large arrays come down in a few chunks: while the
// DMA engine drains chunk k+1 the host threads add chunk k into the host's array (a 96 MB download followed by a 96 MB
// read-modify-write was 3 of the 8.5 ms of a 4 M-atom step).  Returns with all of it added.
int mdp_host_download_add(mdp_ctx *c, double *h_dst, double *h_stage, const double *d_src, size_t n)
{
  hipStream_t st = c->stream;
  // eight equal pieces (cutting the last one in four again, so that less of the adding is left when the last byte has
  // arrived, measured +0.08 ms: the workers' wake-ups cost more than the shorter tail saves)
  size_t cut[MDP_DOWN_CHUNKS + 1];
  int nch = 1;
  cut[0] = 0;
  cut[1] = n;
  if (n > (1u << 21)) {
    const size_t per = ((n + MDP_DOWN_CHUNKS - 1) / MDP_DOWN_CHUNKS + 7) & ~(size_t) 7;
    nch = MDP_DOWN_CHUNKS;
    for (int k = 1; k <= nch; k++) cut[k] = (size_t) k * per < n ? (size_t) k * per : n;
  }
  if (!c->ev_down[0])
    for (int k = 0; k < MDP_DOWN_CHUNKS; k++) MDP_HIP(c, hipEventCreateWithFlags(&c->ev_down[k], hipEventDisableTiming));
  for (int k = 0; k < nch; k++) {
    const size_t b = cut[k], e = cut[k + 1];
    if (b >= e) continue;
    MDP_HIP(c, hipMemcpyAsync(h_stage + b, d_src + b, sizeof(double) * (e - b), hipMemcpyDeviceToHost, st));
    MDP_HIP(c, hipEventRecord(c->ev_down[k], st));
  }
  for (int k = 0; k < nch; k++) {
    const size_t b = cut[k], e = cut[k + 1];
    if (b >= e) continue;
    MDP_HIP(c, hipEventSynchronize(c->ev_down[k]));
    mdp_host_add(h_dst + b, h_stage + b, e - b);
  }
  return MDP_OK;
}

// xraw (device [n][3]) (+ device type[]) -> xq.  d_type null: keep the element already in xq.w
int mdp_pack_xq(mdp_ctx *c, const double *d_x3, const int *d_type, int count)
{
  static_assert(sizeof(double4) == 32, "double4 layout");
  const int n = count >= 0 ? count : c->nall;
  if (n <= 0) return MDP_OK;
  int *d_map = nullptr;
  if (d_type) {
    // map lives at the tail of the type buffer
    d_map = c->type.p + c->nall;
  }
  if (c->host_sort)
    pack_xq_perm_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, d_x3, d_type, d_map, c->host_perm.p, c->xq.p);
  else
    pack_xq_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, d_x3, d_type, d_map, c->xq.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// host mode, rebomos: (re)derive the device's storage order from the positions just uploaded to xraw
static int host_sort_atoms(mdp_ctx *c)
{
  const int nall = c->nall;
  const char *e = getenv("MDP_HOST_SORT");
  // rebomos always builds its own lists; aeam does when the host asked for device lists (mdp_aeam_device_lists)
  const bool own_lists = (c->have_rebomos && !c->have_aeam) || (c->have_aeam && c->aeam_device_lists);
  // (lists from the host's rows, mdp_rebomos_host_list: the rows speak of the host's atom indices, the order stays)
  c->host_sort = !c->md && own_lists && c->nlocal > 0 && !(e && atoi(e) == 0) && !c->rebo_host_list;
  if (!c->host_sort) return MDP_OK;
  hipStream_t st = c->stream;
  MDP_HIP(c, c->sort_keys_a.reserve(nall + 1));
  MDP_HIP(c, c->sort_keys_b.reserve(nall + 1));
  MDP_HIP(c, c->cell_of.reserve(nall + 1));
  MDP_HIP(c, c->host_perm.reserve(nall + 1));
  double sc[3];
  for (int d = 0; d < 3; d++) sc[d] = 1024.0 / (c->bbox_hi[d] - c->bbox_lo[d]);
  hilbert_key_kernel<<<(nall + 255) / 256, 256, 0, st>>>(nall, c->nlocal, c->xraw.p, c->bbox_lo[0], c->bbox_lo[1],
                                                         c->bbox_lo[2], sc[0], sc[1], sc[2], c->sort_keys_a.p,
                                                         c->cell_of.p);
  MDP_HIP(c, hipGetLastError());
  size_t tmp = 0;
  MDP_HIP(c, rocprim::radix_sort_pairs(nullptr, tmp, c->sort_keys_a.p, c->sort_keys_b.p, c->cell_of.p, c->host_perm.p,
                                       (size_t) nall, 0, 31, st));
  MDP_HIP(c, c->scan_tmp.reserve(tmp + 16));
  MDP_HIP(c, rocprim::radix_sort_pairs(c->scan_tmp.p, tmp, c->sort_keys_a.p, c->sort_keys_b.p, c->cell_of.p,
                                       c->host_perm.p, (size_t) nall, 0, 31, st));
  if (c->have_rebomos && !c->have_aeam) { // element-sorted runs of 32 owned atoms (one-atom-row tile lists)
    MDP_TRY(mdp_chunk_by_element(c, nall, c->nlocal, c->host_perm.p, c->cell_of.p, nullptr, c->type.p, c->type.p + nall));
    int *p = c->host_perm.p;
    c->host_perm.p = c->cell_of.p;
    c->cell_of.p = p;
    const size_t cp = c->host_perm.cap;
    c->host_perm.cap = c->cell_of.cap;
    c->cell_of.cap = cp;
  }
  return MDP_OK;
}

// Second ordering pass of the owned atoms: inside every run of 32 consecutive positions (one tile of the one-atom-row
// lists, csrc/rebomos.hip) the atoms of element 0 come first.  The eight rows that share a wave must have equally long
// segments, so a wave of one element pads less.  idx_in[pos] = source index of the atom at position pos after the
// spatial sort; positions >= n_owned (atoms that leave / ghosts) keep their order behind the owned atoms.  Stable.
namespace {
__global__ void chunk_key_kernel(const int n, const int n_owned, const int *__restrict__ idx_in,
                                 const double4 *__restrict__ xq, const int *__restrict__ type,
                                 const int *__restrict__ map, unsigned *__restrict__ key)
{
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  if (p >= n_owned) {
    key[p] = (((unsigned) n_owned >> 5) + 1u) << 1;
    return;
  }
  const int o = idx_in[p];
  int e;
  if (xq)
    e = (int) xq[o].w;
  else {
    const int t = type[o];
    e = map ? map[t] : t - 1;
  }
  key[p] = (((unsigned) p >> 5) << 1) | (e != 0 ? 1u : 0u);
}
} // namespace

int mdp_chunk_by_element(mdp_ctx *c, int n, int n_owned, const int *d_idx_in, int *d_idx_out, const double4 *d_xq,
                         const int *d_type, const int *d_map)
{
  if (n <= 0) return MDP_OK;
  MDP_HIP(c, c->sort_keys_a.reserve((size_t) n + 1));
  MDP_HIP(c, c->sort_keys_b.reserve((size_t) n + 1));
  chunk_key_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, n_owned, d_idx_in, d_xq, d_type, d_map, c->sort_keys_a.p);
  MDP_HIP(c, hipGetLastError());
  int bits = 2;
  while ((1ull << bits) <= ((((unsigned long long) n_owned >> 5) + 2ull) << 1)) bits++;
  size_t tmp = 0;
  MDP_HIP(c, rocprim::radix_sort_pairs(nullptr, tmp, c->sort_keys_a.p, c->sort_keys_b.p, d_idx_in, d_idx_out, (size_t) n, 0,
                                       bits, c->stream));
  MDP_HIP(c, c->scan_tmp.reserve(tmp + 16));
  MDP_HIP(c, rocprim::radix_sort_pairs(c->scan_tmp.p, tmp, c->sort_keys_a.p, c->sort_keys_b.p, d_idx_in, d_idx_out,
                                       (size_t) n, 0, bits, c->stream));
  return MDP_OK;
}

int mdp_scan_exclusive_int(mdp_ctx *c, const int *d_in, int *d_out, int n)
{
  // exclusive scan over n+1 items so that d_out[n] = total (the extra input item is ignored)
  size_t tmp = 0;
  MDP_HIP(c, rocprim::exclusive_scan(nullptr, tmp, d_in, d_out, 0, (size_t) n + 1, rocprim::plus<int>(), c->stream));
  MDP_HIP(c, c->scan_tmp.reserve(tmp + 16));
  MDP_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, tmp, d_in, d_out, 0, (size_t) n + 1, rocprim::plus<int>(),
                                     c->stream));
  return MDP_OK;
}

int mdp_scan_exclusive_i64(mdp_ctx *c, const int *d_in, long long *d_out, int n)
{
  size_t tmp = 0;
  auto in = rocprim::make_transform_iterator(d_in, [] __device__(int v) -> long long { return (long long) v; });
  MDP_HIP(c, rocprim::exclusive_scan(nullptr, tmp, in, d_out, 0ll, (size_t) n + 1, rocprim::plus<long long>(),
                                     c->stream));
  MDP_HIP(c, c->scan_tmp.reserve(tmp + 16));
  MDP_HIP(c, rocprim::exclusive_scan(c->scan_tmp.p, tmp, in, d_out, 0ll, (size_t) n + 1, rocprim::plus<long long>(),
                                     c->stream));
  return MDP_OK;
}

extern "C" {

int mdp_abi_version(void) { return MDP_ABI_VERSION; }

int mdp_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mdp_create(mdp_ctx **out, int device)
{
  if (!out) return MDP_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MDP_EHIP; // no CPU fallback: fail loudly
  if (device < 0 || device >= n) return MDP_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return MDP_EHIP;
  mdp_ctx *c = new (std::nothrow) mdp_ctx();
  if (!c) return MDP_ENOMEM;
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return MDP_EHIP;
  }
  c->own_stream = true;
  if (c->acc.reserve((size_t) MDP_ACC_STRIDE * (2 + MDP_ACC_SLOTS)) != hipSuccess || c->flags.reserve(8) != hipSuccess ||
      hipHostMalloc((void **) &c->h_pinned, 64 * sizeof(double)) != hipSuccess ||
      hipMemset(c->flags.p, 0, 8 * sizeof(int)) != hipSuccess) {
    mdp_destroy(c);
    return MDP_ENOMEM;
  }
  memset(c->map, 0, sizeof c->map);
  memset(c->h_pinned, 0, 64 * sizeof(double));
  *out = c;
  return MDP_OK;
}

int mdp_destroy(mdp_ctx *c)
{
  if (!c) return MDP_OK;
  (void) hipSetDevice(c->device);
  if (c->stream) (void) hipStreamSynchronize(c->stream);
  mdp_dd_release(c);
  c->aeam_frho.release();
  c->aeam_rhor.release();
  c->aeam_z2r.release();
  c->aeam_rhor_v4.release();
  c->aeam_rhor_d4.release();
  c->aeam_z2r_v4.release();
  c->aeam_z2r_d4.release();
  c->aeam_pair_d8.release();
  c->aeam_rhor_ys.release();
  c->aeam_z2r_ys.release();
  c->aeam_maps.release();
  c->aeam_par_d.release();
  c->aeam_par_i.release();
  c->cut_tab.release();
  c->xq.release();
  c->xraw.release();
  c->tag.release();
  c->type.release();
  c->f.release();
  c->eatom.release();
  c->vatom.release();
  c->acc.release();
  c->flags.release();
  c->nb_off.release();
  c->nb.release();
  c->cand_cnt.release();
  c->cand_off.release();
  c->cand.release();
  c->lj_off.release();
  c->lj_cnt.release();
  c->lj_split.release();
  c->cl_flag.release();
  c->cl_pos.release();
  c->cl_order.release();
  c->lj.release();
  c->tu.release();
  c->tmask.release();
  c->lj16_in.release();
  c->lj_len_in.release();
  c->lj_split_in.release();
  c->cand_stage.release();
  c->lj_fixtab.release();
  c->lj_fix_stamp.release();
  c->xhold_prune.release();
  c->tile_nu.release();
  c->tile_flag.release();
  c->lj16.release();
  c->is_center.release();
  c->class_list.release();
  c->class_merged.release();
  c->class_count.release();
  c->pk_cand.release();
  c->amask.release();
  c->xhold_all.release();
  c->ovf.release();
  c->rev.release();
  c->rev16.release();
  c->host_perm.release();
  c->host_tagmap.release();
  c->host_inv.release();
  c->host_tag_dev.release();
  c->host_img.release();
  c->host_stage.release();
  host_unregister_all(c);
  for (int k = 0; k < MDP_DOWN_CHUNKS; k++)
    if (c->ev_down[k]) {
      (void) hipEventDestroy(c->ev_down[k]);
      c->ev_down[k] = nullptr;
    }
  for (int k = 0; k < MDP_UP_RING; k++) {
    if (c->h_up[k]) (void) hipHostFree(c->h_up[k]);
    if (c->ev_up[k]) (void) hipEventDestroy(c->ev_up[k]);
    c->h_up[k] = nullptr;
    c->ev_up[k] = nullptr;
  }
  if (c->h_down) (void) hipHostFree(c->h_down);
  c->h_down = nullptr;
  c->fnbr.release();
  c->fown.release();
  c->vslot.release();
  c->scan_tmp.release();
  c->rho.release();
  c->fp.release();
  c->ang_list.release();
  c->ang_count.release();
  c->v.release();
  c->xhold.release();
  c->rmass.release();
  c->ghost_owner.release();
  c->ghost_shift.release();
  c->mass_type.release();
  c->cell_of.release();
  c->cell_perm.release();
  c->cell_start.release();
  c->sort_keys_a.release();
  c->sort_keys_b.release();
  c->sort_vals_b.release();
  c->nb_cnt.release();
  for (int k = 0; k < 2; k++)
    if (c->ev_sflag[k]) (void) hipEventDestroy(c->ev_sflag[k]);
  if (c->h_pinned) (void) hipHostFree(c->h_pinned);
  if (c->h_small) (void) hipHostFree(c->h_small);
  c->h_small = nullptr;
  if (c->ev_made)
    for (int i = 0; i < 8; i++) (void) hipEventDestroy(c->ev[i]);
  if (c->ev_sb[0])
    for (int i = 0; i < 8; i++) {
      (void) hipEventDestroy(c->ev_sb[i]);
      (void) hipEventDestroy(c->ev_se[i]);
    }
  if (c->own_stream && c->stream) (void) hipStreamDestroy(c->stream);
  delete c;
  return MDP_OK;
}

const char *mdp_last_error(const mdp_ctx *c) { return c ? c->err.c_str() : "null context"; }

int mdp_set_stream(mdp_ctx *c, void *s)
{
  if (!c) return MDP_EINVAL;
  if (c->own_stream && c->stream) {
    (void) hipStreamSynchronize(c->stream);
    (void) hipStreamDestroy(c->stream);
  }
  c->stream = (hipStream_t) s;
  c->own_stream = false;
  return MDP_OK;
}

double mdp_device_bytes(const mdp_ctx *c)
{
  // this context's share: the capacities of ITS buffers (every DevBuf member of mdp_ctx and of its MdpDomain); the
  // process-wide total of all contexts is mdp_device_bytes_counter()
  if (!c) return (double) mdp_device_bytes_counter().load();
  const MdpDomain &D = c->dd;
  const size_t parts[] = {
      c->aeam_frho.bytes(), c->aeam_rhor.bytes(), c->aeam_z2r.bytes(), c->aeam_rhor_v4.bytes(),
      c->aeam_rhor_d4.bytes(), c->aeam_z2r_v4.bytes(), c->aeam_z2r_d4.bytes(), c->aeam_pair_d8.bytes(), c->cand_stage.bytes(),
      c->aeam_rhor_ys.bytes(), c->aeam_z2r_ys.bytes(), c->aeam_maps.bytes(), c->xq.bytes(), c->xraw.bytes(),
      c->host_perm.bytes(), c->host_tagmap.bytes(), c->host_inv.bytes(), c->host_tag_dev.bytes(), c->host_img.bytes(), c->host_stage.bytes(), c->tag.bytes(), c->type.bytes(), c->f.bytes(),
      c->eatom.bytes(), c->vatom.bytes(), c->acc.bytes(), c->flags.bytes(), c->nb_off.bytes(), c->nb.bytes(),
      c->cand_cnt.bytes(), c->cand_off.bytes(), c->cand.bytes(), c->lj_off.bytes(), c->lj_cnt.bytes(),
      c->lj.bytes(), c->cl_flag.bytes(), c->cl_pos.bytes(), c->cl_order.bytes(), c->lj_split.bytes(),
      c->tu.bytes(), c->tmask.bytes(), c->lj16_in.bytes(), c->lj_len_in.bytes(),
      c->lj_split_in.bytes(), c->xhold_prune.bytes(), c->tile_nu.bytes(), c->tile_flag.bytes(),
      c->lj16.bytes(), c->is_center.bytes(), c->class_list.bytes(), c->class_count.bytes(), c->pk_cand.bytes(),
      c->amask.bytes(), c->rev.bytes(), c->rev16.bytes(), c->ovf.bytes(), c->xhold_all.bytes(),
      c->fnbr.bytes(), c->fown.bytes(), c->vslot.bytes(), c->scan_tmp.bytes(), c->rho.bytes(), c->fp.bytes(),
      c->ang_list.bytes(), c->ang_count.bytes(), c->v.bytes(), c->xhold.bytes(), c->rmass.bytes(),
      c->ghost_owner.bytes(), c->ghost_shift.bytes(), c->mass_type.bytes(), c->cell_of.bytes(),
      c->cell_perm.bytes(), c->cell_start.bytes(), c->sort_keys_a.bytes(), c->sort_keys_b.bytes(),
      c->sort_vals_b.bytes(), c->nb_cnt.bytes(),
      D.dest.bytes(), D.counters.bytes(), D.idx_a.bytes(), D.idx_b.bytes(), D.ent_atom.bytes(),
      D.ent_code.bytes(), D.ent_cnt.bytes(), D.ent_off.bytes(), D.sendlist.bytes(), D.type_tmp.bytes(),
      D.tag_tmp.bytes(), D.key_a.bytes(), D.key_b.bytes(), D.sendshift.bytes(), D.v_tmp.bytes(),
      D.xq_tmp.bytes(), D.sbuf.bytes(), D.rbuf.bytes(), D.abuf.bytes(), D.cnt_dev.bytes(),
  };
  double sum = 0.0;
  for (const size_t b : parts) sum += (double) b;
  return sum;
}

int mdp_host_release(mdp_ctx *c, const void *ptr)
{
  if (!c) return MDP_EINVAL;
  for (size_t k = 0; k < c->host_regs.size(); k++)
    if (!ptr || c->host_regs[k].first == ptr) {
      if (c->host_regs[k].second) (void) hipHostUnregister(const_cast<void *>(c->host_regs[k].first));
      c->host_regs.erase(c->host_regs.begin() + k);
      k--;
    }
  (void) hipGetLastError();
  return MDP_OK;
}

int mdp_sync(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  return MDP_OK;
}

int mdp_set_timing(mdp_ctx *c, int on)
{
  if (!c) return MDP_EINVAL;
  c->timing = on != 0;
  return MDP_OK;
}

int mdp_get_timing(mdp_ctx *c, double ms[8])
{
  if (!c || !ms) return MDP_EINVAL;
  for (int i = 0; i < 8; i++) ms[i] = 0.0;
  if (!c->timing || (!c->ev_made && !c->ev_sb[0])) return MDP_OK;
  if (c->timing_spans) {
    MDP_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; i++) {
      float t = 0.f;
      if (((c->span_mask >> i) & 1u) && hipEventElapsedTime(&t, c->ev_sb[i], c->ev_se[i]) == hipSuccess) ms[i] = t;
    }
    (void) hipGetLastError();
    c->span_mask = 0; // (a span that the next compute does not record reads 0, not a stale time)
    return MDP_OK;
  }
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i + 1 < c->ev_marks; i++) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]) == hipSuccess) ms[i] = t;
  }
  (void) hipGetLastError();
  return MDP_OK;
}

// ---- potentials ---------------------------------------------------------------------------------
int mdp_rebomos_set_params(mdp_ctx *c, const mdp_rebomos_params *p)
{
  if (!c || !p) return MDP_EINVAL;
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      if (!(p->rcmax[a][b] > p->rcmin[a][b]) || !(p->sigma[a][b] > 0.0))
        return mdp_fail(c, MDP_EINVAL, "rebomos: rcmax must exceed rcmin and sigma must be positive");
  c->rebomos_host = *p;
  c->have_rebomos = true;
  c->rebo_packed = false;
  mdp_rebomos_fill_dev(c, c->skin);
  return MDP_OK;
}

// host mode: are all ghosts periodic images of owned atoms, reproducible from the host's box?  (after host_sort_atoms)
static int host_derive_ghosts(mdp_ctx *c, const int *h_tag)
{
  c->host_ghosts_derived = false;
  const char *e = getenv("MDP_HOST_GHOSTS"); // "upload": keep taking the ghosts' positions from the host
  if (c->md || !c->host_box_set || !h_tag || c->nghost <= 0 || c->nlocal <= 0 || (e && !strcmp(e, "upload"))) return MDP_OK;
  const int nlocal = c->nlocal, nall = c->nall, nghost = c->nghost;
  int maxtag = 0;
  for (int i = 0; i < nlocal; i++) maxtag = h_tag[i] > maxtag ? h_tag[i] : maxtag;
  if (maxtag < 1 || (long long) maxtag > 16ll * nall + (1ll << 20)) return MDP_OK; // no tags / a sparse tag space
  hipStream_t st = c->stream;
  MDP_HIP(c, c->host_tagmap.reserve((size_t) maxtag + 4));
  MDP_HIP(c, c->ghost_owner.reserve((size_t) nghost + 1));
  MDP_HIP(c, c->ghost_shift.reserve((size_t) 3 * nghost + 3));
  MDP_HIP(c, c->host_img.reserve((size_t) 3 * nghost + 3));
  MDP_HIP(c, c->host_tag_dev.reserve((size_t) nall + 1));
  int *flag = c->host_tagmap.p + maxtag + 1;
  MDP_HIP(c, hipMemsetAsync(c->host_tagmap.p, 0xff, sizeof(int) * ((size_t) maxtag + 1), st));
  MDP_HIP(c, hipMemsetAsync(flag, 0, sizeof(int), st));
  host_tagmap_kernel<<<(nlocal + 255) / 256, 256, 0, st>>>(nlocal, c->tag.p, maxtag, c->host_tagmap.p, flag);
  const int *perm = nullptr, *inv = nullptr;
  if (c->host_sort) {
    MDP_HIP(c, c->host_inv.reserve((size_t) nall + 1));
    host_inv_kernel<<<(nall + 255) / 256, 256, 0, st>>>(nall, c->host_perm.p, c->host_inv.p);
    perm = c->host_perm.p;
    inv = c->host_inv.p;
  }
  const double *h = c->host_h;
  host_ghost_owner_kernel<<<(nghost + 255) / 256, 256, 0, st>>>(nlocal, nall, perm, inv, c->tag.p, c->type.p, maxtag,
                                                                c->host_tagmap.p, c->xraw.p, h[0], h[1], h[2], h[3], h[4],
                                                                h[5], c->ghost_owner.p, c->host_img.p, c->ghost_shift.p,
                                                                flag);
  host_tag_dev_kernel<<<(nall + 255) / 256, 256, 0, st>>>(nall, perm, c->tag.p, c->host_tag_dev.p);
  MDP_HIP(c, hipGetLastError());
  int hflag = 1;
  MDP_TRY(mdp_read_one(c, flag, sizeof(int), &hflag));
  c->host_ghosts_derived = hflag == 0;
  return MDP_OK;
}

// positions of the images from their owners', with the box of this step
static int host_refresh_ghosts(mdp_ctx *c)
{
  const double *h = c->host_h;
  host_ghost_refresh_kernel<<<(c->nghost + 255) / 256, 256, 0, c->stream>>>(c->nlocal, c->nghost, c->ghost_owner.p,
                                                                            c->host_img.p, h[0], h[1], h[2], h[3], h[4],
                                                                            h[5], c->xq.p);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

extern "C++" int mdp_host_refresh_ghosts(mdp_ctx *c)
{
  if (!c->host_ghosts_derived || c->nghost <= 0) return MDP_OK;
  return host_refresh_ghosts(c);
}

extern "C++" int mdp_host_ghost_scalar(mdp_ctx *c, double *d_a)
{
  if (!c->host_ghosts_derived || c->nghost <= 0) return MDP_OK;
  host_ghost_scalar_kernel<<<(c->nghost + 255) / 256, 256, 0, c->stream>>>(c->nlocal, c->nghost, c->ghost_owner.p, d_a);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

extern "C++" int mdp_host_ghost_fold(mdp_ctx *c, int w, double *d_a)
{
  if (!c->host_ghosts_derived || c->nghost <= 0) return MDP_OK;
  host_ghost_fold_kernel<<<(c->nghost + 255) / 256, 256, 0, c->stream>>>(c->nlocal, c->nghost, w, c->ghost_owner.p, d_a);
  MDP_HIP(c, hipGetLastError());
  return MDP_OK;
}

// ---- host-mode atoms ----------------------------------------------------------------------------
int mdp_set_atoms_host(mdp_ctx *c, int nlocal, int nghost, const double *x, const int *type, const int *tag,
                       int ntypes, const int *map)
{
  if (!c || nlocal < 0 || nghost < 0 || ntypes < 1 || ntypes > 15)
    return mdp_fail(c, MDP_EINVAL, "mdp_set_atoms_host: bad arguments");
  const int nall = nlocal + nghost;
  // an empty sub-domain (slab / vacuum runs) hands over no arrays at all: atom->x may be NULL when nmax == 0
  if (nall > 0 && (!x || !type)) return mdp_fail(c, MDP_EINVAL, "mdp_set_atoms_host: x / type missing for %d atoms", nall);
  MDP_HIP(c, hipSetDevice(c->device));
  if ((long long) nall >= (1ll << 29)) return mdp_fail(c, MDP_EINVAL, "too many atoms for NEIGHMASK");
  c->nlocal = nlocal;
  c->nghost = nghost;
  c->nall = nall;
  c->ntypes = ntypes;
  for (int t = 1; t <= ntypes; t++) c->map[t] = map ? map[t] : t - 1;
  c->map[0] = 0;
  MDP_HIP(c, c->xq.reserve(nall + 1));
  MDP_HIP(c, c->xraw.reserve((size_t) 3 * nall + 3));
  MDP_HIP(c, c->type.reserve((size_t) nall + 32));
  MDP_HIP(c, c->tag.reserve(nall + 1));
  MDP_HIP(c, c->f.reserve((size_t) 3 * nall + 3));
  MDP_HIP(c, c->eatom.reserve(nall + 1));
  hipStream_t st = c->stream;
  if (nall) {
    MDP_TRY(host_upload(c, c->xraw.p, x, sizeof(double) * 3 * nall));
    MDP_HIP(c, hipMemcpyAsync(c->type.p, type, sizeof(int) * nall, hipMemcpyHostToDevice, st));
  }
  MDP_HIP(c, hipMemcpyAsync(c->type.p + nall, c->map, sizeof(int) * 16, hipMemcpyHostToDevice, st));
  if (tag && nall) MDP_HIP(c, hipMemcpyAsync(c->tag.p, tag, sizeof(int) * nall, hipMemcpyHostToDevice, st));
  c->atoms_set = true;
  c->host_check_armed = false;
  if (!c->md) { // host mode: bounding box for the device binning, padded so that motion inside the skin stays inside
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < nall; i++)
      for (int d = 0; d < 3; d++) {
        const double v = x[3 * (size_t) i + d];
        lo[d] = v < lo[d] ? v : lo[d];
        hi[d] = v > hi[d] ? v : hi[d];
      }
    for (int d = 0; d < 3; d++) {
      c->bbox_lo[d] = (nall ? lo[d] : 0.0) - 4.0;
      c->bbox_hi[d] = (nall ? hi[d] : 1.0) + 4.0;
    }
  }
  MDP_TRY(host_sort_atoms(c));
  MDP_TRY(mdp_pack_xq(c, c->xraw.p, c->type.p));
  MDP_TRY(host_derive_ghosts(c, tag));
  MDP_HIP(c, hipStreamSynchronize(st)); // host buffers may change after return
  c->neigh_set = false;
  c->rebo_packed = false;
  return MDP_OK;
}

int mdp_set_positions_host(mdp_ctx *c, const double *x)
{
  if (!c) return MDP_EINVAL;
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (c->nall == 0) return MDP_OK; // empty sub-domain: nothing to move (x may be NULL)
  if (!x) return mdp_fail(c, MDP_EINVAL, "mdp_set_positions_host: x missing for %d atoms", c->nall);
  MDP_HIP(c, hipSetDevice(c->device));
  if (c->host_ghosts_derived) { // owned atoms only; the images follow their owners on the device
    MDP_TRY(host_upload(c, c->xraw.p, x, sizeof(double) * 3 * c->nlocal));
    MDP_TRY(mdp_pack_xq(c, c->xraw.p, nullptr, c->nlocal));
    MDP_TRY(host_refresh_ghosts(c));
  } else {
    MDP_TRY(host_upload(c, c->xraw.p, x, sizeof(double) * 3 * c->nall));
    MDP_TRY(mdp_pack_xq(c, c->xraw.p, nullptr));
  }
  MDP_TRY(mdp_rebomos_host_precheck(c));
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  return MDP_OK;
}

int mdp_set_box_host(mdp_ctx *c, const double *h)
{
  if (!c) return MDP_EINVAL;
  if (!h) { // back to "ghost positions come from the host"
    c->host_box_set = false;
    c->host_ghosts_derived = false;
    return MDP_OK;
  }
  if (!(h[0] > 0.0 && h[1] > 0.0 && h[2] > 0.0)) return mdp_fail(c, MDP_EINVAL, "mdp_set_box_host: box lengths must be positive");
  for (int k = 0; k < 6; k++) c->host_h[k] = h[k];
  c->host_box_set = true;
  return MDP_OK;
}

int mdp_host_ghosts_derived(mdp_ctx *c) { return c && c->host_ghosts_derived ? 1 : 0; }

static int upload_csr(mdp_ctx *c, double skin)
{
  const int nall = c->nall;
  hipStream_t st = c->stream;
  const long long total = c->h_off[nall];
  MDP_HIP(c, c->nb_off.reserve(nall + 2));
  MDP_HIP(c, c->nb.reserve((size_t) total + 1));
  MDP_HIP(c, hipMemcpyAsync(c->nb_off.p, c->h_off.data(), sizeof(long long) * (nall + 1), hipMemcpyHostToDevice, st));
  if (total)
    MDP_HIP(c, hipMemcpyAsync(c->nb.p, c->h_nb.data(), sizeof(int) * total, hipMemcpyHostToDevice, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  c->nb_total = total;
  c->nb_owned_total = c->h_off[c->nlocal];
  c->skin = skin;
  c->neigh_set = true;
  c->rebo_packed = false;
  return MDP_OK;
}

int mdp_set_neighbors_host(mdp_ctx *c, int inum, int gnum, const int *ilist, const int *numneigh,
                           int *const *firstneigh, double skin)
{
  if (!c || inum < 0 || gnum < 0) return MDP_EINVAL;
  if (inum + gnum > 0 && (!ilist || !numneigh || !firstneigh))
    return mdp_fail(c, MDP_EINVAL, "mdp_set_neighbors_host: list arrays missing for %d rows", inum + gnum);
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (c->host_sort && c->have_aeam)
    return mdp_fail(c, MDP_ESTATE, "aeam builds its lists on the device (mdp_aeam_device_lists): hand over the skin, not the list");
  if (inum != c->nlocal) return mdp_fail(c, MDP_EINVAL, "inum (%d) != nlocal (%d)", inum, c->nlocal);
  MDP_HIP(c, hipSetDevice(c->device));
  const int nall = c->nall;
  // rows are stored by atom index; atoms without a row (ghosts beyond gnum) get empty rows
  std::vector<int> cnt(nall, 0);
  for (int ii = 0; ii < inum + gnum; ii++) {
    const int i = ilist[ii];
    if (i < 0 || i >= nall) return mdp_fail(c, MDP_EINVAL, "ilist entry %d out of range", i);
    cnt[i] = numneigh[i];
  }
  c->h_off.assign(nall + 1, 0);
  for (int i = 0; i < nall; i++) c->h_off[i + 1] = c->h_off[i] + cnt[i];
  c->h_nb.resize((size_t) c->h_off[nall]);
  for (int ii = 0; ii < inum + gnum; ii++) {
    const int i = ilist[ii];
    const int *src = firstneigh[i];
    int *dst = c->h_nb.data() + c->h_off[i];
    for (int k = 0; k < cnt[i]; k++) {
      const int j = src[k] & MDP_NEIGHMASK;
      if (j >= nall) return mdp_fail(c, MDP_EINVAL, "neighbor index %d out of range", j);
      dst[k] = j;
    }
  }
  return upload_csr(c, skin);
}

int mdp_set_skin(mdp_ctx *c, double skin)
{
  if (!c || !(skin >= 0.0)) return MDP_EINVAL;
  c->skin = skin;
  c->skin_set = true;
  c->rebo_packed = false;
  return MDP_OK;
}

int mdp_rebomos_host_list(mdp_ctx *c, int on)
{
  if (!c) return MDP_EINVAL;
  if (c->md) return mdp_fail(c, MDP_ESTATE, "mdp_rebomos_host_list: host mode only");
  c->rebo_host_list = on != 0;
  c->neigh_set = false;
  c->rebo_packed = false;
  c->atoms_set = false; // (the storage order depends on it: the next mdp_set_atoms_host decides)
  return MDP_OK;
}

int mdp_aeam_device_lists(mdp_ctx *c, int on)
{
  if (!c) return MDP_EINVAL;
  c->aeam_device_lists = on != 0;
  c->neigh_set = false;
  c->rebo_packed = false;
  return MDP_OK;
}

int mdp_set_neighbors_csr_host(mdp_ctx *c, int nall, const int *numneigh, const long long *offset, const int *neigh,
                               double skin)
{
  if (!c || !numneigh || !offset || (!neigh && offset[nall] > 0)) return MDP_EINVAL;
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (c->host_sort && c->have_aeam)
    return mdp_fail(c, MDP_ESTATE, "aeam builds its lists on the device (mdp_aeam_device_lists): hand over the skin, not the list");
  if (nall != c->nall) return mdp_fail(c, MDP_EINVAL, "nall mismatch");
  MDP_HIP(c, hipSetDevice(c->device));
  c->h_off.assign(nall + 1, 0);
  for (int i = 0; i < nall; i++) c->h_off[i + 1] = c->h_off[i] + numneigh[i];
  c->h_nb.resize((size_t) c->h_off[nall]);
  for (int i = 0; i < nall; i++) {
    const int *src = neigh + offset[i];
    int *dst = c->h_nb.data() + c->h_off[i];
    for (int k = 0; k < numneigh[i]; k++) {
      const int j = src[k] & MDP_NEIGHMASK;
      if (j < 0 || j >= nall) return mdp_fail(c, MDP_EINVAL, "neighbor index %d out of range", j);
      dst[k] = j;
    }
  }
  return upload_csr(c, skin);
}

// read back acc[0..6] (+flags); returns MDP_EOVERFLOW if a kernel flagged one
static int fetch_acc(mdp_ctx *c, double *eng, double *virial)
{
  hipStream_t st = c->stream;
  MDP_HIP(c, hipMemcpyAsync(c->h_pinned, c->acc.p, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
  int *hflags = (int *) (c->h_pinned + 16);
  MDP_HIP(c, hipMemcpyAsync(hflags, c->flags.p, sizeof(int) * 5, hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  MDP_TRY(mdp_flags_check(c, hflags));
  c->last_eng = c->h_pinned[0];
  for (int k = 0; k < 6; k++) c->last_virial[k] = c->h_pinned[1 + k];
  if (eng) *eng += c->h_pinned[0];
  if (virial)
    for (int k = 0; k < 6; k++) virial[k] += c->h_pinned[1 + k];
  return MDP_OK;
}

int mdp_rebomos_compute_host(mdp_ctx *c, int eflag, int vflag, double *f, double *eng_vdwl, double *virial,
                             double *eatom, double *vatom)
{
  if (!c) return MDP_EINVAL;
  if (!c->have_rebomos) return mdp_fail(c, MDP_ESTATE, "rebomos parameters not set");
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  if (c->nlocal == 0) { // a rank without owned atoms contributes nothing (f may be NULL); the reference loops over zero atoms
    c->last_eng = 0.0;
    for (int k = 0; k < 6; k++) c->last_virial[k] = 0.0;
    return MDP_OK;
  }
  // f may be NULL while the integrator lives on the device too (mdp_hnve_*): the forces then stay where their reader is
  if (!f && !c->hn_on) return mdp_fail(c, MDP_EINVAL, "mdp_rebomos_compute_host: f missing for %d owned atoms", c->nlocal);
  MDP_HIP(c, hipSetDevice(c->device));
  if ((eflag & MDP_EFLAG_ATOM) && !eatom) eflag &= ~MDP_EFLAG_ATOM;
  if ((vflag & MDP_VFLAG_ATOM) && !vatom) vflag &= ~MDP_VFLAG_ATOM;
  MDP_TRY(mdp_rebomos_run(c, eflag, vflag, /*zero_f=*/true));
  hipStream_t st = c->stream;
  const int nlocal = c->nlocal;
  if (!f) { // nothing per atom crosses the link; the totals only when asked for (overflow bits are sticky until then)
    if (eflag & MDP_EFLAG_ATOM) return mdp_fail(c, MDP_EINVAL, "mdp_rebomos_compute_host: per-atom energy without f");
    if (!(eflag || vflag)) return MDP_OK;
    return fetch_acc(c, (eflag & MDP_EFLAG_GLOBAL) ? eng_vdwl : nullptr, (vflag & MDP_VFLAG_GLOBAL) ? virial : nullptr);
  }
  // results come back through the staging buffer and are ADDED on the host (LAMMPS semantics)
  const double *df = c->f.p, *de = c->eatom.p, *dv = c->vatom.p;
  if (c->host_sort && nlocal > 0) { // back to the host's atom order (owned atoms permute among themselves)
    MDP_HIP(c, c->host_stage.reserve((size_t) 10 * nlocal + 10));
    double *sf = c->host_stage.p, *se = sf + (size_t) 3 * nlocal, *sv = se + nlocal;
    const auto blocks = [](long long n) { return (unsigned) ((n + 255) / 256); };
    unpermute_kernel<<<blocks(3ll * nlocal), 256, 0, st>>>(nlocal, 3, c->host_perm.p, c->f.p, sf);
    df = sf;
    if (eflag & MDP_EFLAG_ATOM) {
      unpermute_kernel<<<blocks(nlocal), 256, 0, st>>>(nlocal, 1, c->host_perm.p, c->eatom.p, se);
      de = se;
    }
    if (vflag & MDP_VFLAG_ATOM) {
      unpermute_kernel<<<blocks(6ll * nlocal), 256, 0, st>>>(nlocal, 6, c->host_perm.p, c->vatom.p, sv);
      dv = sv;
    }
    MDP_HIP(c, hipGetLastError());
  }
  MDP_TRY(mdp_host_pinned_reserve(c, (size_t) 10 * nlocal + 16));
  double *hf = c->h_down, *he = hf + (size_t) 3 * nlocal, *hv = he + nlocal;
  const size_t n3 = (size_t) 3 * nlocal;
  if (eflag & MDP_EFLAG_ATOM) MDP_HIP(c, hipMemcpyAsync(he, de, sizeof(double) * nlocal, hipMemcpyDeviceToHost, st));
  if (vflag & MDP_VFLAG_ATOM)
    MDP_HIP(c, hipMemcpyAsync(hv, dv, sizeof(double) * 6 * nlocal, hipMemcpyDeviceToHost, st));
  MDP_TRY(mdp_host_download_add(c, f, hf, df, n3));
  MDP_TRY(fetch_acc(c, (eflag & MDP_EFLAG_GLOBAL) ? eng_vdwl : nullptr, (vflag & MDP_VFLAG_GLOBAL) ? virial : nullptr));
  if (vflag & MDP_VFLAG_ATOM) mdp_host_add(vatom, hv, (size_t) 6 * nlocal);
  if (eflag & MDP_EFLAG_ATOM) mdp_host_add(eatom, he, (size_t) nlocal);
  return MDP_OK;
}

} // extern "C"
