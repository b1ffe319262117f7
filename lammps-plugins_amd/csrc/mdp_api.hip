// mdp_api.hip -- C-ABI entry points of libmdpair_hip.so: lifecycle, potentials, host-mode data
// movement (what a LAMMPS Pair::compute() hands over) and the host-mode compute calls.
#include "mdp_common.h"

#include <rocprim/rocprim.hpp>

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

int mdp_acc_begin(mdp_ctx *c, bool any)
{
  const size_t n = any ? (size_t) MDP_ACC_STRIDE * (1 + MDP_ACC_SLOTS) : (size_t) MDP_ACC_STRIDE;
  MDP_HIP(c, hipMemsetAsync(c->acc.p, 0, sizeof(double) * n, c->stream));
  MDP_HIP(c, hipMemsetAsync(c->flags.p, 0, sizeof(int) * 4, c->stream));
  return MDP_OK;
}

int mdp_acc_end(mdp_ctx *c, bool any)
{
  if (any) {
    acc_reduce_kernel<<<1, 512, 0, c->stream>>>(c->acc.p);
    MDP_HIP(c, hipGetLastError());
  }
  return MDP_OK;
}

// xraw (device [n][3]) (+ device type[]) -> xq.  d_type null: keep the element already in xq.w
int mdp_pack_xq(mdp_ctx *c, const double *d_x3, const int *d_type)
{
  static_assert(sizeof(double4) == 32, "double4 layout");
  const int n = c->nall;
  if (n <= 0) return MDP_OK;
  int *d_map = nullptr;
  if (d_type) {
    // map lives at the tail of the type buffer
    d_map = c->type.p + c->nall;
  }
  pack_xq_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, d_x3, d_type, d_map, c->xq.p);
  MDP_HIP(c, hipGetLastError());
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
      hipHostMalloc((void **) &c->h_pinned, 64 * sizeof(double)) != hipSuccess) {
    mdp_destroy(c);
    return MDP_ENOMEM;
  }
  memset(c->map, 0, sizeof c->map);
  *out = c;
  return MDP_OK;
}

int mdp_destroy(mdp_ctx *c)
{
  if (!c) return MDP_OK;
  (void) hipSetDevice(c->device);
  if (c->stream) (void) hipStreamSynchronize(c->stream);
  c->aeam_frho.release();
  c->aeam_rhor.release();
  c->aeam_z2r.release();
  c->aeam_rhor_v4.release();
  c->aeam_rhor_d4.release();
  c->aeam_z2r_v4.release();
  c->aeam_z2r_d4.release();
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
  c->tile_nu.release();
  c->tile_flag.release();
  c->lj16.release();
  c->is_center.release();
  c->class_list.release();
  c->class_count.release();
  c->pk_cand.release();
  c->amask.release();
  c->xhold_all.release();
  c->ovf.release();
  c->rev.release();
  c->rev16.release();
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
  if (c->ev_stale_made) (void) hipEventDestroy(c->ev_stale);
  if (c->h_pinned) (void) hipHostFree(c->h_pinned);
  if (c->ev_made)
    for (int i = 0; i < 8; i++) (void) hipEventDestroy(c->ev[i]);
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
  if (!c->timing || !c->ev_made) return MDP_OK;
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

// ---- host-mode atoms ----------------------------------------------------------------------------
int mdp_set_atoms_host(mdp_ctx *c, int nlocal, int nghost, const double *x, const int *type, const int *tag,
                       int ntypes, const int *map)
{
  if (!c || nlocal < 0 || nghost < 0 || !x || !type || ntypes < 1 || ntypes > 15)
    return mdp_fail(c, MDP_EINVAL, "mdp_set_atoms_host: bad arguments");
  MDP_HIP(c, hipSetDevice(c->device));
  const int nall = nlocal + nghost;
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
  MDP_HIP(c, hipMemcpyAsync(c->xraw.p, x, sizeof(double) * 3 * nall, hipMemcpyHostToDevice, st));
  MDP_HIP(c, hipMemcpyAsync(c->type.p, type, sizeof(int) * nall, hipMemcpyHostToDevice, st));
  MDP_HIP(c, hipMemcpyAsync(c->type.p + nall, c->map, sizeof(int) * 16, hipMemcpyHostToDevice, st));
  if (tag) MDP_HIP(c, hipMemcpyAsync(c->tag.p, tag, sizeof(int) * nall, hipMemcpyHostToDevice, st));
  c->atoms_set = true;
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
  MDP_TRY(mdp_pack_xq(c, c->xraw.p, c->type.p));
  MDP_HIP(c, hipStreamSynchronize(st)); // host buffers may change after return
  c->neigh_set = false;
  c->rebo_packed = false;
  return MDP_OK;
}

int mdp_set_positions_host(mdp_ctx *c, const double *x)
{
  if (!c || !x) return MDP_EINVAL;
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  MDP_HIP(c, hipSetDevice(c->device));
  MDP_HIP(c, hipMemcpyAsync(c->xraw.p, x, sizeof(double) * 3 * c->nall, hipMemcpyHostToDevice, c->stream));
  MDP_TRY(mdp_pack_xq(c, c->xraw.p, nullptr));
  MDP_HIP(c, hipStreamSynchronize(c->stream));
  return MDP_OK;
}

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
  if (!c || inum < 0 || gnum < 0 || !ilist || !numneigh || !firstneigh) return MDP_EINVAL;
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
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
  c->rebo_packed = false;
  return MDP_OK;
}

int mdp_set_neighbors_csr_host(mdp_ctx *c, int nall, const int *numneigh, const long long *offset, const int *neigh,
                               double skin)
{
  if (!c || !numneigh || !offset || (!neigh && offset[nall] > 0)) return MDP_EINVAL;
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
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
  MDP_HIP(c, hipMemcpyAsync(hflags, c->flags.p, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
  MDP_HIP(c, hipStreamSynchronize(st));
  if (hflags[0] & 1)
    return mdp_fail(c, MDP_EOVERFLOW, "REBO neighbor count exceeds the lane-group capacity (Neighbor list overflow)");
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
  if (!c || !f) return MDP_EINVAL;
  if (!c->have_rebomos) return mdp_fail(c, MDP_ESTATE, "rebomos parameters not set");
  if (!c->atoms_set) return mdp_fail(c, MDP_ESTATE, "atoms not set");
  MDP_HIP(c, hipSetDevice(c->device));
  if ((eflag & MDP_EFLAG_ATOM) && !eatom) eflag &= ~MDP_EFLAG_ATOM;
  if ((vflag & MDP_VFLAG_ATOM) && !vatom) vflag &= ~MDP_VFLAG_ATOM;
  MDP_TRY(mdp_rebomos_run(c, eflag, vflag, /*zero_f=*/true));
  hipStream_t st = c->stream;
  const int nlocal = c->nlocal;
  // results come back through the staging buffer and are ADDED on the host (LAMMPS semantics)
  std::vector<double> hf((size_t) 3 * nlocal), he;
  MDP_HIP(c, hipMemcpyAsync(hf.data(), c->f.p, sizeof(double) * 3 * nlocal, hipMemcpyDeviceToHost, st));
  if (eflag & MDP_EFLAG_ATOM) {
    he.resize(nlocal);
    MDP_HIP(c, hipMemcpyAsync(he.data(), c->eatom.p, sizeof(double) * nlocal, hipMemcpyDeviceToHost, st));
  }
  std::vector<double> hv;
  if (vflag & MDP_VFLAG_ATOM) {
    hv.resize((size_t) 6 * nlocal);
    MDP_HIP(c, hipMemcpyAsync(hv.data(), c->vatom.p, sizeof(double) * 6 * nlocal, hipMemcpyDeviceToHost, st));
  }
  MDP_TRY(fetch_acc(c, (eflag & MDP_EFLAG_GLOBAL) ? eng_vdwl : nullptr, (vflag & MDP_VFLAG_GLOBAL) ? virial : nullptr));
  if (vflag & MDP_VFLAG_ATOM)
    for (size_t k = 0; k < (size_t) 6 * nlocal; k++) vatom[k] += hv[k];
  for (size_t k = 0; k < (size_t) 3 * nlocal; k++) f[k] += hf[k];
  if (eflag & MDP_EFLAG_ATOM)
    for (int i = 0; i < nlocal; i++) eatom[i] += he[i];
  return MDP_OK;
}

} // extern "C"
