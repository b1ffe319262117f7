// comm_rccl.hip -- the exchanges of the domain decomposition (csrc/domain.hip) done by the library itself on RCCL,
// for hosts that are C++ (north star: "host code stays C++ ... ghost-atom halo exchange on RCCL over xGMI").
// What the reference gets from LAMMPS' Comm brick over MPI -- Comm::exchange / borders / forward_comm / reverse_comm
// (they are why a style may ask for ghost atoms at all, pair_rebomos.cpp:218, and the 5.67 % "Comm" of
// log.rebomos-bulk.4:67) -- becomes, per rank:
//     counts    one ncclAllGather of the per-rank count vector (tiny; a row of the matrix per rank)
//     records   one group of ncclSend / ncclRecv per peer with data (xGMI is point-to-point: a brick talks to its
//               <= 26 neighbours, with 8 ranks on a 2x2x2 grid to all 7 others)
// The per-step position exchange runs on its own stream between two events, so that the interior Lennard-Jones
// work launched in between overlaps it.
//
// RCCL is bound at run time (dlopen): a process that already holds a copy -- torch ships one under the same
// soname -- keeps using that copy, and hosts that never call mdp_dd_comm_* need no RCCL at all.
#include "mdp_common.h"

#include <dlfcn.h>

#include <mutex>
#include <string>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  bool ok = false;
  bool test_double = false; // bound to tests/native/libfake_rccl.so (MDP_RCCL_LIBRARY): a rehearsal, reported as one
  char path[256] = {0};
};

RcclApi *rccl()
{
  static RcclApi api;
  static std::once_flag once; // several contexts may come up on several host threads at the same time
  std::call_once(once, [] {
    void *h = nullptr;
    // MDP_RCCL_LIBRARY: the one object to bind instead of the system's RCCL (a differently named build, or the test
    // double of tests/native/fake_rccl.cpp that lets several ranks share one GPU); nothing else is tried then
    const char *forced = getenv("MDP_RCCL_LIBRARY");
    if (forced && *forced) {
      h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
      if (!h) fprintf(stderr, "mdpair_hip: MDP_RCCL_LIBRARY=%s could not be loaded: %s\n", forced, dlerror());
      else snprintf(api.path, sizeof api.path, "%s", forced);
    } else {
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) {
          snprintf(api.path, sizeof api.path, "%s", name);
          break;
        }
      }
    }
    if (!h) return;
    api.test_double = dlsym(h, "mdp_fake_rccl_marker") != nullptr;
#define MDP_SYM(field, sym)                                        \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, sym)); \
  if (!api.field) return
    MDP_SYM(GetUniqueId, "ncclGetUniqueId");
    MDP_SYM(CommInitRank, "ncclCommInitRank");
    MDP_SYM(CommDestroy, "ncclCommDestroy");
    MDP_SYM(GetErrorString, "ncclGetErrorString");
    MDP_SYM(GroupStart, "ncclGroupStart");
    MDP_SYM(GroupEnd, "ncclGroupEnd");
    MDP_SYM(Send, "ncclSend");
    MDP_SYM(Recv, "ncclRecv");
    MDP_SYM(AllGather, "ncclAllGather");
    MDP_SYM(AllReduce, "ncclAllReduce");
#undef MDP_SYM
    api.ok = true;
  });
  return api.ok ? &api : nullptr;
}

#define MDP_NCCL(c, call)                                                                                      \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess)                                                                                     \
      return mdp_fail((c), MDP_EHIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, rccl()->GetErrorString(r_)); \
  } while (0)

int comm_require(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  if (!c->md || !c->dd.on) return mdp_fail(c, MDP_ESTATE, "mdp_dd_setup not called");
  if (!c->dd.nccl_comm) return mdp_fail(c, MDP_ESTATE, "mdp_dd_comm_init not called");
  MDP_HIP(c, hipSetDevice(c->device));
  return MDP_OK;
}

// every rank's count vector -> recv[q] = what rank q sends to me
int exchange_counts(mdp_ctx *c, const std::vector<int> &send, std::vector<int> &recv)
{
  MdpDomain &D = c->dd;
  const int n = D.G.nranks;
  hipStream_t st = c->stream;
  MDP_HIP(c, D.cnt_dev.reserve((size_t) n * (n + 1) + 8));
  MDP_TRY(mdp_write_small(c, D.cnt_dev.p, send.data(), sizeof(int) * n));
  MDP_NCCL(c, rccl()->AllGather(D.cnt_dev.p, D.cnt_dev.p + n, (size_t) n, ncclInt, (ncclComm_t) D.nccl_comm, st));
  std::vector<int> all((size_t) n * n);
  MDP_TRY(mdp_read_one(c, D.cnt_dev.p + n, sizeof(int) * n * n, all.data()));
  recv.assign(n, 0);
  for (int q = 0; q < n; q++) recv[q] = all[(size_t) q * n + D.G.rank];
  return MDP_OK;
}

constexpr int kFlagP2PRanks = 16;

// ragged all-to-all of `width` doubles per record; segments in rank order on both sides
int exchange(mdp_ctx *c, const double *sbuf, const int *scnt, double *rbuf, const int *rcnt, int width, hipStream_t st)
{
  MdpDomain &D = c->dd;
  const int n = D.G.nranks;
  RcclApi *R = rccl();
  size_t so = 0, ro = 0;
  bool any = false;
  for (int q = 0; q < n; q++) any = any || scnt[q] || rcnt[q];
  if (!any) return MDP_OK;
  MDP_NCCL(c, R->GroupStart());
  // a failing Send/Recv must not leave the thread inside an open group (every later RCCL call of the thread, torch's
  // included, would queue into it): remember the first error, always close the group, report afterwards
  ncclResult_t first = ncclSuccess;
  for (int q = 0; q < n && first == ncclSuccess; q++) {
    if (scnt[q]) first = R->Send(sbuf + so, (size_t) scnt[q] * width, ncclDouble, q, (ncclComm_t) D.nccl_comm, st);
    if (rcnt[q] && first == ncclSuccess)
      first = R->Recv(rbuf + ro, (size_t) rcnt[q] * width, ncclDouble, q, (ncclComm_t) D.nccl_comm, st);
    so += (size_t) scnt[q] * width;
    ro += (size_t) rcnt[q] * width;
  }
  const ncclResult_t end = R->GroupEnd();
  if (first != ncclSuccess) return mdp_fail(c, MDP_EHIP, "ncclSend/ncclRecv -> %s", R->GetErrorString(first));
  if (end != ncclSuccess) return mdp_fail(c, MDP_EHIP, "ncclGroupEnd -> %s", R->GetErrorString(end));
  return MDP_OK;
}

// The RCCL kernel of an exchange gets a head start over the compute kernel it is to overlap: measured on one MI355X
// (profiles/r05_multi_gpu_step), a RCCL kernel that reaches the device a few microseconds AFTER a compute kernel that
// fills it ends only when that kernel's grid has drained (a compute stream whose CU mask leaves 8 or 16 compute units
// free changed nothing), one that is running already when the compute kernel arrives ends in its own time.  So the
// context's stream waits for an event the communication stream records right in front of the RCCL kernel; that costs
// the compute stream one cross-queue hand-over (about 18 us in the one-rank rehearsal).  Worth it where the kernel
// behind the exchange is long -- aeam: ONE density kernel over the interior tiles (0.27 ms at 1.0 M atoms), ONE force
// kernel (0.48 ms) -- and not for rebomos, whose first interior kernel ends after 0.04 ms with 0.12 ms of interior
// centres still to come.  MDP_COMM_LEAD=0 / 1 forces it off / on for every style.
int comm_lead(mdp_ctx *c)
{
  static const int lead = [] {
    const char *e = getenv("MDP_COMM_LEAD");
    return e ? (atoi(e) != 0 ? 1 : 0) : -1;
  }();
  MdpDomain &D = c->dd;
  // whole steps in the library (step mode): the overlap policy of the step decides; piecewise callers keep the
  // measured default (aeam yes, rebomos no)
  const bool want = D.step_mode ? D.lead : c->cfg.style == 2;
  if (lead == 0 || (lead < 0 && !want)) return MDP_OK;
  if (!D.ev_lead) MDP_HIP(c, hipEventCreateWithFlags(&D.ev_lead, hipEventDisableTiming));
  MDP_HIP(c, hipEventRecord(D.ev_lead, D.comm_stream));
  MDP_HIP(c, hipStreamWaitEvent(c->stream, D.ev_lead, 0));
  return MDP_OK;
}

// ---- overlap policy -------------------------------------------------------------------------------------------------
// Whether an exchange really travels behind the interior work depends on the machine: on one MI355X a RCCL kernel
// queued behind a compute kernel that fills the device ends when that kernel's grid has drained, and
// what real xGMI links do is not known before the first run on them.  So the first steps of a run are a trial: blocks
// of kOvBlock steps take the policies in turn (every rank the same one in the same step: the schedule depends on the
// step count only), the device time of each step is taken between two events on the context's stream -- it contains
// every wait for an exchange --, the ranks' means are reduced with MAX, and the cheapest policy stays.  The exchanges
// themselves are the same in every policy (same sends and receives, issued at the same point of the step), so ranks
// could even differ without harm; they do not, because they reduce before they choose.
// MDP_OVERLAP_POLICY = split | lead | blocking | first fixes the policy (no trial); auto (default) runs the trial.
constexpr int kOvBlock = 4, kOvRounds = 2; // steps per block (the first is the transition and is not measured), rounds

// policies in trial order; "first" (3) exists for rebomos only
int ov_npol(const mdp_ctx *c) { return c->cfg.style == 1 ? 5 : 4; }
int ov_policy_of(const mdp_ctx *c, int idx) { return c->cfg.style == 1 ? idx : (idx < 3 ? idx : 4); }

const char *ov_name(int p)
{
  static const char *n[5] = {"split", "lead", "blocking", "first", "inline"};
  return p >= 0 && p < 5 ? n[p] : "undecided";
}

void ov_read_env(mdp_ctx *c)
{
  MdpDomain &D = c->dd;
  if (D.ov_forced != -2) return;
  D.ov_forced = -1;
  if (const char *e = getenv("MDP_OVERLAP_POLICY")) {
    for (int i = 0; i < ov_npol(c); i++)
      if (!strcmp(e, ov_name(ov_policy_of(c, i)))) D.ov_forced = ov_policy_of(c, i);
    if (D.ov_forced < 0 && strcmp(e, "auto"))
      fprintf(stderr, "mdpair_hip: MDP_OVERLAP_POLICY=%s is not one of split, lead, blocking, inline%s, auto: running the trial\n", e,
              c->cfg.style == 1 ? ", first" : "");
  }
  if (D.ov_forced >= 0) D.ov_policy = D.ov_forced;
}

// completed event pairs -> sums (never waits unless `drain`)
int ov_harvest(mdp_ctx *c, bool drain)
{
  MdpDomain &D = c->dd;
  for (int k = 0; k < 4; k++) { // (event-pair slots)
    if (D.ov_slot_pol[k] < 0 || k == D.ov_slot) continue;
    if (drain) MDP_HIP(c, hipEventSynchronize(D.ov_ev[k][1]));
    else if (hipEventQuery(D.ov_ev[k][1]) != hipSuccess) continue;
    float ms = 0.0f;
    MDP_HIP(c, hipEventElapsedTime(&ms, D.ov_ev[k][0], D.ov_ev[k][1]));
    D.ov_sum[D.ov_slot_pol[k]] += ms;
    D.ov_cnt[D.ov_slot_pol[k]]++;
    D.ov_slot_pol[k] = -1;
  }
  return MDP_OK;
}

// at the head of a step: the policy this step runs under; opens the measurement of a trial step
int ov_step_begin(mdp_ctx *c, bool plain_step /* no reneighboring, no tally: comparable with its neighbours */)
{
  MdpDomain &D = c->dd;
  ov_read_env(c);
  D.ov_slot = -1;
  if (D.ov_policy < 0) {
    const int np = ov_npol(c);
    const long long total = (long long) kOvBlock * np * kOvRounds;
    MDP_TRY(ov_harvest(c, false));
    if (D.ov_step >= total) { // the trial is over: every rank reduces the same vector at the same step
      MDP_TRY(ov_harvest(c, true));
      double mean[5];
      for (int p = 0; p < 5; p++) mean[p] = D.ov_cnt[p] > 0 ? D.ov_sum[p] / D.ov_cnt[p] : 1.0e30; // (never tried: never chosen)
      MDP_TRY(mdp_dd_comm_allreduce(c, mean, 5, /*max*/ 1));
      int best = 0;
      for (int i = 1; i < np; i++) {
        const int p = ov_policy_of(c, i);
        if (mean[p] < mean[best] * 0.995) best = p; // (a later policy has to win by half a percent)
      }
      for (int p = 0; p < 5; p++) D.ov_mean[p] = mean[p] < 1.0e29 ? mean[p] : 0.0;
      D.ov_policy = best;
      if (D.G.rank == 0)
        fprintf(stderr, "mdpair_hip: overlap policy %s (device ms per step, max over %d ranks: split %.4f lead %.4f blocking %.4f "
                        "first %.4f inline %.4f)\n",
                ov_name(best), D.G.nranks, D.ov_mean[0], D.ov_mean[1], D.ov_mean[2], D.ov_mean[3], D.ov_mean[4]);
    }
  }
  if (D.ov_policy >= 0) {
    D.ov_cur = D.ov_policy;
  } else {
    D.ov_cur = ov_policy_of(c, (int) ((D.ov_step / kOvBlock) % ov_npol(c)));
    const bool measured = plain_step && D.ov_step % kOvBlock != 0;
    D.ov_step++;
    if (measured)
      for (int k = 0; k < 4 && D.ov_slot < 0; k++)
        if (D.ov_slot_pol[k] < 0) D.ov_slot = k;
    if (D.ov_slot >= 0) {
      for (int e = 0; e < 2; e++)
        if (!D.ov_ev[D.ov_slot][e]) MDP_HIP(c, hipEventCreate(&D.ov_ev[D.ov_slot][e]));
      MDP_HIP(c, hipEventRecord(D.ov_ev[D.ov_slot][0], c->stream));
    }
  }
  D.lead = D.ov_cur == 1;
  D.inline_x = D.ov_cur == 4; // the position exchange on the context's own stream (no second stream, no events)
  c->overlap_mode = (D.ov_cur == 2 || D.ov_cur == 4) ? 2 : (D.ov_cur == 3 ? 3 : 0);
  return MDP_OK;
}

int ov_step_end(mdp_ctx *c)
{
  MdpDomain &D = c->dd;
  if (D.ov_slot >= 0) {
    MDP_HIP(c, hipEventRecord(D.ov_ev[D.ov_slot][1], c->stream));
    D.ov_slot_pol[D.ov_slot] = D.ov_cur;
    D.ov_slot = -1;
  }
  c->overlap_mode = 0; // (piecewise callers of mdp_md_compute_begin / _end keep the split order)
  return MDP_OK;
}

} // namespace

extern "C" {

int mdp_dd_comm_unique_id(void *id128)
{
  if (!id128) return MDP_EINVAL;
  RcclApi *R = rccl();
  if (!R) return MDP_ENOTIMPL;
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return MDP_EHIP;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, sizeof id);
  return MDP_OK;
}

// which RCCL object the library is bound to; returns 1 when it is the test double of tests/native (several ranks on one
// GPU, host-staged: a rehearsal of the exchange schedule, never a measurement), 0 for a real RCCL, < 0 when none loads
int mdp_dd_comm_library(char *buf, int cap)
{
  RcclApi *R = rccl();
  if (buf && cap > 0) snprintf(buf, (size_t) cap, "%s", R ? R->path : "");
  if (!R) return MDP_ENOTIMPL;
  return R->test_double ? 1 : 0;
}

int mdp_dd_comm_init(mdp_ctx *c, const void *id128)
{
  if (!c || !id128) return MDP_EINVAL;
  if (!c->md || !c->dd.on) return mdp_fail(c, MDP_ESTATE, "mdp_dd_setup not called");
  RcclApi *R = rccl();
  if (!R) return mdp_fail(c, MDP_ENOTIMPL, "RCCL (librccl.so.1) could not be loaded");
  MDP_HIP(c, hipSetDevice(c->device));
  MdpDomain &D = c->dd;
  if (D.nccl_comm) {
    (void) R->CommDestroy((ncclComm_t) D.nccl_comm);
    D.nccl_comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  MDP_NCCL(c, R->CommInitRank(&comm, D.G.nranks, id, D.G.rank));
  D.nccl_comm = comm;
  if (!D.comm_stream) {
    // highest priority: the (one-workgroup) RCCL kernels must find a slot while a compute kernel fills the device
    int least = 0, greatest = 0;
    MDP_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    MDP_HIP(c, hipStreamCreateWithPriority(&D.comm_stream, hipStreamNonBlocking, greatest));
  }
  if (!D.ev_packed) MDP_HIP(c, hipEventCreateWithFlags(&D.ev_packed, hipEventDisableTiming));
  if (!D.ev_arrived) MDP_HIP(c, hipEventCreateWithFlags(&D.ev_arrived, hipEventDisableTiming));
  return MDP_OK;
}

int mdp_dd_comm_destroy(mdp_ctx *c)
{
  if (!c) return MDP_EINVAL;
  MdpDomain &D = c->dd;
  if (D.nccl_comm && rccl()) (void) rccl()->CommDestroy((ncclComm_t) D.nccl_comm);
  D.nccl_comm = nullptr;
  if (D.comm_stream) (void) hipStreamDestroy(D.comm_stream);
  D.comm_stream = nullptr;
  if (D.ev_packed) (void) hipEventDestroy(D.ev_packed);
  if (D.ev_arrived) (void) hipEventDestroy(D.ev_arrived);
  if (D.ev_lead) (void) hipEventDestroy(D.ev_lead);
  D.ev_packed = D.ev_arrived = D.ev_lead = nullptr;
  for (int k = 0; k < 4; k++)
    for (int e = 0; e < 2; e++) {
      if (D.ov_ev[k][e]) (void) hipEventDestroy(D.ov_ev[k][e]);
      D.ov_ev[k][e] = nullptr;
    }
  D.sbuf.release();
  D.rbuf.release();
  D.cnt_dev.release();
  D.abuf.release();
  return MDP_OK;
}

// Comm::exchange + Comm::borders + Neighbor::build; collective over the ranks of the communicator
int mdp_dd_comm_reneighbor(mdp_ctx *c)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  const int n = D.G.nranks;
  hipStream_t st = c->stream;
  std::vector<int> sc(n, 0), rc(n, 0);
  // -- atoms that left the brick
  MDP_TRY(mdp_dd_migrate_begin(c, sc.data()));
  MDP_TRY(exchange_counts(c, sc, rc));
  long long ns = 0, nr = 0;
  for (int q = 0; q < n; q++) {
    ns += sc[q];
    nr += rc[q];
  }
  MDP_HIP(c, D.sbuf.reserve((size_t) 8 * ns + 8));
  MDP_HIP(c, D.rbuf.reserve((size_t) 8 * nr + 8));
  MDP_TRY(mdp_dd_migrate_pack(c, D.sbuf.p));
  MDP_TRY(exchange(c, D.sbuf.p, sc.data(), D.rbuf.p, rc.data(), 8, st));
  MDP_TRY(mdp_dd_migrate_end(c, (int) nr, D.rbuf.p));
  // -- ghost entries
  MDP_TRY(mdp_dd_borders_begin(c, sc.data()));
  MDP_TRY(exchange_counts(c, sc, rc));
  ns = nr = 0;
  for (int q = 0; q < n; q++) {
    ns += sc[q];
    nr += rc[q];
  }
  MDP_HIP(c, D.sbuf.reserve((size_t) 6 * ns + 8));
  MDP_HIP(c, D.rbuf.reserve((size_t) 6 * nr + 8));
  MDP_TRY(mdp_dd_borders_pack(c, D.sbuf.p));
  MDP_TRY(exchange(c, D.sbuf.p, sc.data(), D.rbuf.p, rc.data(), 6, st));
  MDP_TRY(mdp_dd_borders_end(c, rc.data(), D.rbuf.p));
  // the per-step exchanges (3 doubles per entry either way) reuse the border buffers (6 per entry): nothing is
  // (re-)allocated -- a hipFree would wait for the whole device -- between two reneighborings
  D.fwd_pending = false;
  return mdp_md_build_neighbors_impl(c);
}

// per step: x of the send list -> remote ghosts.  _begin packs on the context's stream and starts the exchange on
// the communication stream; _end makes the context's stream wait for it and unpacks.  Work launched in between
// (mdp_md_compute_begin: the interior Lennard-Jones tiles) overlaps the exchange.
int mdp_dd_comm_forward_begin(mdp_ctx *c)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  if (!D.nsend && !D.nrecv && !D.step_mode) return MDP_OK; // (step mode: the all-gather below is collective)
  MDP_HIP(c, D.sbuf.reserve((size_t) 3 * D.nsend + 8));
  MDP_HIP(c, D.rbuf.reserve((size_t) 3 * D.nrecv + 8));
  MDP_TRY(mdp_dd_forward_pack(c, D.sbuf.p));
  // policy "inline" (whole steps only): nothing is to overlap the exchange, so it is queued on the context's stream
  // itself -- pack, RCCL kernel, unpack in stream order, without the two hand-overs between streams
  const bool inl = D.step_mode && D.inline_x;
  hipStream_t xs = inl ? c->stream : D.comm_stream;
  if (!inl) {
    MDP_HIP(c, hipEventRecord(D.ev_packed, c->stream));
    MDP_HIP(c, hipStreamWaitEvent(D.comm_stream, D.ev_packed, 0));
    MDP_TRY(comm_lead(c));
  }
  D.fwd_inline = inl;
  // the ranks' "an atom of mine has moved beyond the trigger" words of this step travel WITH the positions (the
  // integrate kernel of this step wrote this rank's; mdp_dd_forward_unpack reduces them for the next step's decision):
  // one more double to and from every rank in the same group of ncclSend/ncclRecv -- one RCCL kernel per step -- up to
  // kFlagP2PRanks ranks, an all-gather behind the positions on larger communicators
  const bool flags = D.flagbuf.p && D.step_mode;
  const int n = D.G.nranks;
  if (flags && n <= kFlagP2PRanks) {
    RcclApi *R = rccl();
    MDP_NCCL(c, R->GroupStart());
    const int rc = exchange(c, D.sbuf.p, D.bord_send.data(), D.rbuf.p, D.bord_recv.data(), 3, xs);
    ncclResult_t first = ncclSuccess; // (the group is always closed: see exchange())
    for (int q = 0; q < n && first == ncclSuccess && rc == MDP_OK; q++) {
      first = R->Send(D.flagbuf.p + D.flag_par, 1, ncclDouble, q, (ncclComm_t) D.nccl_comm, xs);
      if (first == ncclSuccess)
        first = R->Recv(D.flagbuf.p + 2 + q, 1, ncclDouble, q, (ncclComm_t) D.nccl_comm, xs);
    }
    const ncclResult_t end = R->GroupEnd();
    if (rc != MDP_OK) return rc;
    if (first != ncclSuccess) return mdp_fail(c, MDP_EHIP, "ncclSend/ncclRecv -> %s", R->GetErrorString(first));
    if (end != ncclSuccess) return mdp_fail(c, MDP_EHIP, "ncclGroupEnd -> %s", R->GetErrorString(end));
    D.fwd_gathered = true;
  } else {
    MDP_TRY(exchange(c, D.sbuf.p, D.bord_send.data(), D.rbuf.p, D.bord_recv.data(), 3, xs));
    if (flags) {
      MDP_NCCL(c, rccl()->AllGather(D.flagbuf.p + D.flag_par, D.flagbuf.p + 2, 1, ncclDouble, (ncclComm_t) D.nccl_comm,
                                    xs));
      D.fwd_gathered = true;
    }
  }
  if (!inl) MDP_HIP(c, hipEventRecord(D.ev_arrived, xs));
  D.fwd_pending = true;
  return MDP_OK;
}

int mdp_dd_comm_forward_end(mdp_ctx *c)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  if (!D.nsend && !D.nrecv && !D.step_mode) return MDP_OK;
  if (!D.fwd_inline) MDP_HIP(c, hipStreamWaitEvent(c->stream, D.ev_arrived, 0));
  D.fwd_pending = false;
  return mdp_dd_forward_unpack(c, D.rbuf.p);
}

// AEAM: fp of the send list -> remote ghosts (pair_aeam.cpp:307, 946-963)
int mdp_dd_comm_forward_scalar(mdp_ctx *c)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  if (!D.nsend && !D.nrecv) return MDP_OK;
  MDP_HIP(c, D.sbuf.reserve((size_t) D.nsend + 8));
  MDP_HIP(c, D.rbuf.reserve((size_t) D.nrecv + 8));
  MDP_TRY(mdp_dd_forward_scalar_pack(c, D.sbuf.p));
  MDP_TRY(exchange(c, D.sbuf.p, D.bord_send.data(), D.rbuf.p, D.bord_recv.data(), 1, c->stream));
  return mdp_dd_forward_scalar_unpack(c, D.rbuf.p);
}

// AEAM: forces that angular centres put on remote ghosts -> added onto their owners (Comm::reverse_comm)
int mdp_dd_comm_reverse(mdp_ctx *c)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  MDP_TRY(mdp_md_fold_self_ghost_f(c));
  if (!D.nsend && !D.nrecv) return MDP_OK;
  MDP_HIP(c, D.sbuf.reserve((size_t) 3 * D.nsend + 8));
  MDP_HIP(c, D.rbuf.reserve((size_t) 3 * D.nrecv + 8));
  MDP_TRY(mdp_dd_reverse_pack(c, D.rbuf.p));
  MDP_TRY(exchange(c, D.rbuf.p, D.bord_recv.data(), D.sbuf.p, D.bord_send.data(), 3, c->stream));
  return mdp_dd_reverse_unpack(c, D.sbuf.p);
}

// AEAM, per step, behind the interior force tiles: fp of the send list -> remote ghosts (pair_aeam.cpp:307) and --
// with_reverse -- the forces three-body terms put on remote ghosts -> their owners (Comm::reverse_comm), both legs in
// ONE group of ncclSend/ncclRecv on the communication stream.  _begin packs on the context's stream (the three-body
// forces are final there: mdp_md_aeam_density after mdp_md_compute_begin) and folds the periodic self-images;
// _end makes the context's stream wait, unpacks fp and adds the forces.  with_reverse must be the same on all ranks
// (mdp_md_aeam_state out[3], reduced with mdp_dd_comm_allreduce at every reneighboring).
int mdp_dd_comm_aeam_exchange_begin(mdp_ctx *c, int with_reverse)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  MDP_TRY(mdp_md_fold_self_ghost_f(c));
  D.aeam_pending = 0;
  if (!D.nsend && !D.nrecv) return MDP_OK;
  const int n = D.G.nranks;
  RcclApi *R = rccl();
  // sbuf: [fp out: nsend][f in: 3 nsend]
  MDP_HIP(c, D.sbuf.reserve((size_t) 4 * D.nsend + 8));
  MDP_TRY(mdp_dd_forward_scalar_pack(c, D.sbuf.p));
  // the remote ghosts lie in rank order behind the self-images: their forces leave from f itself and their fp arrives
  // in fp itself (the interior tiles that run meanwhile touch neither)
  double *gf = c->f.p + 3 * (size_t) (c->nlocal + D.nself), *gfp = c->fp.p + (size_t) (c->nlocal + D.nself);
  MDP_HIP(c, hipEventRecord(D.ev_packed, c->stream));
  MDP_HIP(c, hipStreamWaitEvent(D.comm_stream, D.ev_packed, 0));
  MDP_TRY(comm_lead(c));
  MDP_NCCL(c, R->GroupStart());
  ncclResult_t first = ncclSuccess; // (the group is always closed: see exchange())
  size_t so = 0, ro = 0;
  for (int q = 0; q < n && first == ncclSuccess; q++) {
    const size_t ns = (size_t) D.bord_send[q], nr = (size_t) D.bord_recv[q];
    if (ns) first = R->Send(D.sbuf.p + so, ns, ncclDouble, q, (ncclComm_t) D.nccl_comm, D.comm_stream);
    if (nr && first == ncclSuccess)
      first = R->Recv(gfp + ro, nr, ncclDouble, q, (ncclComm_t) D.nccl_comm, D.comm_stream);
    if (with_reverse) {
      if (nr && first == ncclSuccess)
        first = R->Send(gf + 3 * ro, 3 * nr, ncclDouble, q, (ncclComm_t) D.nccl_comm, D.comm_stream);
      if (ns && first == ncclSuccess)
        first = R->Recv(D.sbuf.p + D.nsend + 3 * so, 3 * ns, ncclDouble, q, (ncclComm_t) D.nccl_comm, D.comm_stream);
    }
    so += ns;
    ro += nr;
  }
  const ncclResult_t end = R->GroupEnd();
  if (first != ncclSuccess) return mdp_fail(c, MDP_EHIP, "ncclSend/ncclRecv -> %s", R->GetErrorString(first));
  if (end != ncclSuccess) return mdp_fail(c, MDP_EHIP, "ncclGroupEnd -> %s", R->GetErrorString(end));
  MDP_HIP(c, hipEventRecord(D.ev_arrived, D.comm_stream));
  D.fwd_pending = true;
  D.aeam_pending = with_reverse ? 2 : 1;
  return MDP_OK;
}

int mdp_dd_comm_aeam_exchange_end(mdp_ctx *c)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  if (!D.aeam_pending) return MDP_OK;
  MDP_HIP(c, hipStreamWaitEvent(c->stream, D.ev_arrived, 0));
  D.fwd_pending = false;
  const int rev = D.aeam_pending == 2;
  D.aeam_pending = 0;
  if (rev) MDP_TRY(mdp_dd_reverse_unpack(c, D.sbuf.p + D.nsend)); // (fp arrived in place)
  return MDP_OK;
}

// ---- a whole step in two calls ------------------------------------------------------------------------------------
// What a host does around Pair::compute in a step of a multi-GPU run (Verlet::run: initial_integrate, neighbor->decide,
// comm->exchange / borders or forward_comm, force, final_integrate), with the `neigh_modify every 1 check yes` decision
// taken from the word that travelled with the previous step's halo (MdpDomain::flagbuf): no blocking call, no
// collective of the host's own, identical on all ranks.
//   _begin: integrate (with the final half-kick a deferred step left, with_final), decide, reneighbor or start the
//           position exchange, then everything of the compute that needs no remote ghost of this step
//           (rebomos: interior centres; aeam: density of the interior tiles).      force_rebuild: -1 decide, 0 no, 1 yes
//   _end:   the exchange arrives, the rest of the compute -- aeam with its fp / ghost-force exchange behind the interior
//           pair forces when the step is on the phased order -- and the final half-kick (or its deferral, defer_final)
int mdp_dd_comm_step_begin(mdp_ctx *c, int with_final, int force_rebuild, int eflag, int vflag, int *reneighbored)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  if (reneighbored) *reneighbored = 0;
  if (!D.flagbuf.p) {
    MDP_HIP(c, D.flagbuf.reserve((size_t) 2 + D.G.nranks + 8));
    MDP_HIP(c, hipMemsetAsync(D.flagbuf.p, 0, sizeof(double) * ((size_t) 2 + D.G.nranks + 8), c->stream));
  }
  D.step_mode = true;
  // the answer the previous step's halo brought (complete long ago: its unpack kernel ran before that step's forces)
  int glob = 0;
  if (D.glob_pending) {
    MDP_HIP(c, hipEventSynchronize(D.ev_glob_ref));
    glob = *(int *) (c->h_pinned + 46);
    D.glob_pending = false;
  }
  int moved = 0, dangerous = 0;
  const bool rebuild = force_rebuild > 0 || (force_rebuild < 0 && glob);
  MDP_TRY(ov_step_begin(c, !rebuild && !eflag && !vflag));
  if (rebuild) {
    // (no check of the new positions: they are about to become the reference)
    MDP_TRY(mdp_md_advance(c, with_final != 0, nullptr, 0.0, 0.0));
    D.moved_pending = false;
    MDP_TRY(mdp_dd_comm_reneighbor(c));
    if (c->cfg.style == 2) { // the reverse exchange of a step is skipped by all ranks alike when nobody has ghost forces
      int st[4] = {0, 0, 0, 0};
      MDP_TRY(mdp_md_aeam_state(c, st));
      double v = st[3] ? 1.0 : 0.0;
      MDP_TRY(mdp_dd_comm_allreduce(c, &v, 1, 1));
      D.ghost_forces = v > 0.0;
    }
    D.fresh_ghosts = true;
    if (reneighbored) *reneighbored = 1;
  } else {
    MDP_TRY(mdp_md_integrate_check(c, with_final, &moved, &dangerous)); // (this rank's word for the peers; `moved` itself is not used)
    if (dangerous) D.dangerous++;
    D.fresh_ghosts = false;
    MDP_TRY(mdp_dd_comm_forward_begin(c));
  }
  return mdp_md_compute_begin(c, eflag, vflag);
}

int mdp_dd_comm_step_end(mdp_ctx *c, int eflag, int vflag, int defer_final)
{
  MDP_TRY(comm_require(c));
  MdpDomain &D = c->dd;
  if (!D.fresh_ghosts) MDP_TRY(mdp_dd_comm_forward_end(c));
  if (c->cfg.style == 1) {
    MDP_TRY(mdp_md_compute_end(c, eflag, vflag));
  } else {
    MDP_TRY(mdp_md_aeam_density(c, eflag)); // B: the rest of passes 1 + 2 (+ three-body forces when A ran)
    int st[4] = {0, 0, 0, 0};
    MDP_TRY(mdp_md_aeam_state(c, st));
    if (st[0] & 4) { // on the phased order: fp out and ghost forces back behind the interior pair forces
      D.steps_phased++;
      MDP_TRY(mdp_dd_comm_aeam_exchange_begin(c, D.ghost_forces ? 1 : 0));
      MDP_TRY(mdp_md_aeam_force_begin(c, eflag, vflag));
      MDP_TRY(mdp_dd_comm_aeam_exchange_end(c));
      MDP_TRY(mdp_md_aeam_force(c, eflag, vflag));
    } else { // this rank's rows were due for pruning (or a per-atom-virial step): the SAME exchanges, in the same order
      MDP_TRY(mdp_dd_comm_forward_scalar(c));
      MDP_TRY(mdp_md_aeam_force(c, eflag, vflag));
      if (D.ghost_forces) MDP_TRY(mdp_dd_comm_reverse(c));
      else MDP_TRY(mdp_md_fold_self_ghost_f(c));
    }
  }
  if (defer_final) MDP_TRY(mdp_md_defer_final(c));
  else MDP_TRY(mdp_md_final_integrate(c));
  return ov_step_end(c);
}

// [0] = aeam steps on the phased order so far, [1] = ghost forces travel (aeam), [2] = the last _begin reneighbored,
// [3] = reneighborings, [4] = checks that saw an owned atom beyond half the skin ("dangerous builds"),
// [5] = overlap policy (-1 while its trial runs), [6] = 1 if MDP_OVERLAP_POLICY fixed it, [7] = its trial mean in ns
int mdp_dd_comm_step_info(mdp_ctx *c, long long out[8])
{
  if (!c || !out) return MDP_EINVAL;
  for (int k = 0; k < 8; k++) out[k] = 0;
  out[0] = c->dd.steps_phased;
  out[1] = c->dd.ghost_forces ? 1 : 0;
  out[2] = c->dd.fresh_ghosts ? 1 : 0;
  out[3] = c->dd.reneighbors;
  out[4] = c->dd.dangerous;
  out[5] = c->dd.ov_policy;                  // 0 split, 1 lead, 2 blocking, 3 first; -1 while the trial runs
  out[6] = c->dd.ov_forced >= 0 ? 1 : 0;     // fixed by MDP_OVERLAP_POLICY
  out[7] = (long long) (c->dd.ov_mean[c->dd.ov_policy >= 0 ? c->dd.ov_policy : 0] * 1.0e6 + 0.5); // its trial mean, ns
  return MDP_OK;
}

// thermo sums / the collective `check yes` decision: vals[n] <- sum (op 0) or max (op 1) over the ranks
int mdp_dd_comm_allreduce(mdp_ctx *c, double *vals, int n, int op)
{
  MDP_TRY(comm_require(c));
  if (!vals || n < 1 || n > 64) return MDP_EINVAL;
  MdpDomain &D = c->dd;
  hipStream_t st = c->stream;
  // own buffer: sbuf may hold the position exchange that is in flight on the communication stream
  if (D.fwd_pending) return mdp_fail(c, MDP_ESTATE, "mdp_dd_comm_allreduce between mdp_dd_comm_forward_begin and _end");
  MDP_HIP(c, D.abuf.reserve(128));
  MDP_TRY(mdp_write_small(c, D.abuf.p, vals, sizeof(double) * n));
  MDP_NCCL(c, rccl()->AllReduce(D.abuf.p, D.abuf.p + 64, (size_t) n, ncclDouble, op ? ncclMax : ncclSum,
                                (ncclComm_t) D.nccl_comm, st));
  return mdp_read_one(c, D.abuf.p + 64, sizeof(double) * n, vals);
}

} // extern "C"
