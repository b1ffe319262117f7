// fake_rccl.cpp -- TEST DOUBLE for the ten RCCL entry points csrc/comm_rccl.hip binds with dlopen
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGetErrorString, ncclGroupStart, ncclGroupEnd, ncclSend,
// ncclRecv, ncclAllGather, ncclAllReduce).  Test infrastructure: never shipped, never linked; the library loads it
// only when MDP_RCCL_LIBRARY names it, and then says so (mdp_dd_comm_library).
//
// Why: RCCL refuses two ranks on one device, and the GPU boxes of this pool have one.  The double lets N ranks --
// threads of one process or N processes -- share ONE GPU and run the library's own transport (per-peer grouped
// send/recv schedule, the flag word riding in the halo, count all-gathers, all-reduces, migration) with peers that
// are not the rank itself.
//
// What it keeps of RCCL's contract, so that a wrong schedule FAILS here as it would hang on the wire:
//   * point-to-point operations are matched per (source, destination) in the order they were issued; the n-th
//     ncclSend of rank a to rank b meets the n-th ncclRecv of rank b from rank a, and their byte counts must agree
//     (a mismatch is an error here; on the wire it is a hang or a truncated message);
//   * a send completes only when the peer's matching receive has taken it, a receive only when the matching send was
//     issued: an unmatched operation runs into the timeout (MDP_FAKE_RCCL_TIMEOUT_S, default 120 s) and every rank of
//     the communicator then fails at its next call;
//   * collectives are matched by their order on the communicator: kind, count, datatype and reduction of call k must
//     be the same on every rank, and a point-to-point message carries the number of collectives its sender had issued,
//     which must equal the receiver's (point-to-point operations and collectives of one communicator have to be issued
//     in one order on all ranks);
//   * operations are ordered with the stream they are given: everything queued on that stream before the call has
//     finished before data leaves, and work queued behind the call sees the received data.
// What it does NOT show: anything about time -- the host blocks inside ncclGroupEnd / the
// collectives until the peers arrive, data is staged through POSIX shared memory with synchronous copies, so there is
// no overlap with compute, no xGMI link, no RCCL kernel occupying compute units, no channel or protocol limits.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 32;
constexpr int kRing = 64;            // point-to-point messages in flight per (source, destination)
constexpr size_t kCollBytes = 8192;  // largest contribution of one rank to a collective
constexpr uint64_t kMagic = 0x6d64705f66616b65ull;

struct Msg {
  uint64_t bytes, coll_epoch;
};

struct Chan {
  std::atomic<uint64_t> posted, consumed;
  Msg ring[kRing];
};

struct CollDesc {
  uint64_t kind, count, dtype, op, epoch;
};

struct Shared {
  uint64_t magic;
  std::atomic<int> nranks, attached, detached, failed;
  std::atomic<uint64_t> bar_count, bar_gen;
  CollDesc desc[kMaxRanks];
  alignas(64) char coll[kMaxRanks][kCollBytes];
  Chan chan[kMaxRanks][kMaxRanks]; // [source][destination]
};

struct Comm {
  Shared *sh = nullptr;
  int nranks = 0, rank = -1;
  uint64_t coll_epoch = 0;
  uint64_t sent[kMaxRanks] = {}, received[kMaxRanks] = {};
  std::string base;
};

struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  Comm *comm;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local ncclResult_t g_group_error = ncclSuccess;

// MDP_FAKE_RCCL_HOSTMEM=1: buffers are host memory and streams are ignored -- for the CPU tests of the double's own
// matching rules (tests/test_fake_rccl.py); the GPU tests never set it
bool hostmem()
{
  static const bool h = [] {
    const char *e = getenv("MDP_FAKE_RCCL_HOSTMEM");
    return e && atoi(e) != 0;
  }();
  return h;
}

hipError_t stream_sync(hipStream_t st) { return hostmem() ? hipSuccess : hipStreamSynchronize(st); }

hipError_t copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
  if (hostmem()) {
    memcpy(dst, src, bytes);
    return hipSuccess;
  }
  return hipMemcpy(dst, src, bytes, kind);
}

double timeout_s()
{
  static const double t = [] {
    const char *e = getenv("MDP_FAKE_RCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 120.0;
    return v > 0.0 ? v : 120.0;
  }();
  return t;
}

void say(const Comm *c, const char *fmt, ...)
{
  char msg[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof msg, fmt, ap);
  va_end(ap);
  fprintf(stderr, "[fake-rccl] rank %d of %d: %s\n", c ? c->rank : -1, c ? c->nranks : 0, msg);
  fflush(stderr);
}

using Clock = std::chrono::steady_clock;

// spin until pred() holds; false on timeout or when another rank has failed
template <class Pred> bool wait_for(Comm *c, Pred pred)
{
  const auto t0 = Clock::now();
  unsigned spins = 0;
  while (!pred()) {
    if (c->sh->failed.load(std::memory_order_acquire)) return false;
    if (++spins > 2000) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (std::chrono::duration<double>(Clock::now() - t0).count() > timeout_s()) return false;
    } else
      std::this_thread::yield();
  }
  return true;
}

ncclResult_t fail(Comm *c, ncclResult_t r)
{
  c->sh->failed.store(1, std::memory_order_release);
  return r;
}

size_t type_size(ncclDataType_t t)
{
  switch (t) {
  case ncclInt8:
  case ncclUint8: return 1;
  case ncclInt32:
  case ncclUint32:
  case ncclFloat32: return 4;
  case ncclInt64:
  case ncclUint64:
  case ncclFloat64: return 8;
  default: return 0;
  }
}

std::string msg_name(const Comm *c, int src, int dst, uint64_t seq)
{
  char b[160];
  snprintf(b, sizeof b, "%s_m_%d_%d_%llu", c->base.c_str(), src, dst, (unsigned long long) seq);
  return b;
}

bool barrier(Comm *c)
{
  Shared *s = c->sh;
  const uint64_t gen = s->bar_gen.load(std::memory_order_acquire);
  if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint64_t) c->nranks) {
    s->bar_count.store(0, std::memory_order_relaxed);
    s->bar_gen.store(gen + 1, std::memory_order_release);
    return true;
  }
  return wait_for(c, [&] { return s->bar_gen.load(std::memory_order_acquire) != gen; });
}

// the operations of one (outermost) group: all sends are issued, then the receives are matched in order, then every
// send waits for its receiver
ncclResult_t run_group(std::vector<Op> &ops)
{
  std::vector<hipStream_t> synced;
  for (const Op &o : ops) {
    bool seen = false;
    for (hipStream_t s : synced) seen = seen || s == o.stream;
    if (seen) continue;
    if (stream_sync(o.stream) != hipSuccess) return fail(o.comm, ncclUnhandledCudaError);
    synced.push_back(o.stream);
  }
  for (const Op &o : ops) {
    if (!o.send) continue;
    Comm *c = o.comm;
    if (c->sh->failed.load()) return ncclRemoteError;
    Chan &ch = c->sh->chan[c->rank][o.peer];
    const uint64_t seq = c->sent[o.peer];
    if (!wait_for(c, [&] { return seq - ch.consumed.load(std::memory_order_acquire) < (uint64_t) kRing; })) {
      say(c, "ncclSend #%llu to rank %d: %d earlier sends to that rank were never received",
          (unsigned long long) seq, o.peer, kRing);
      return fail(c, ncclSystemError);
    }
    if (o.bytes) {
      const std::string name = msg_name(c, c->rank, o.peer, seq);
      const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t) o.bytes) != 0) {
        say(c, "shm_open/ftruncate(%s, %zu bytes) failed", name.c_str(), o.bytes);
        if (fd >= 0) close(fd);
        return fail(c, ncclSystemError);
      }
      void *p = mmap(nullptr, o.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) return fail(c, ncclSystemError);
      const hipError_t e = copy(p, o.buf, o.bytes, hipMemcpyDeviceToHost);
      munmap(p, o.bytes);
      if (e != hipSuccess) {
        say(c, "copy of a send buffer failed: %s", hipGetErrorString(e));
        return fail(c, ncclUnhandledCudaError);
      }
    }
    Msg &m = ch.ring[seq % kRing];
    m.bytes = o.bytes;
    m.coll_epoch = c->coll_epoch;
    ch.posted.store(seq + 1, std::memory_order_release);
    c->sent[o.peer] = seq + 1;
  }
  for (const Op &o : ops) {
    if (o.send) continue;
    Comm *c = o.comm;
    Chan &ch = c->sh->chan[o.peer][c->rank];
    const uint64_t seq = c->received[o.peer];
    if (!wait_for(c, [&] { return ch.posted.load(std::memory_order_acquire) > seq; })) {
      if (c->sh->failed.load()) return ncclRemoteError;
      say(c, "ncclRecv #%llu from rank %d (%zu bytes) was never matched by a send: the ranks' exchange schedules differ",
          (unsigned long long) seq, o.peer, o.bytes);
      return fail(c, ncclSystemError);
    }
    const Msg m = ch.ring[seq % kRing];
    if (m.bytes != o.bytes) {
      say(c, "ncclRecv #%llu from rank %d expects %zu bytes, the matching ncclSend carries %llu",
          (unsigned long long) seq, o.peer, o.bytes, (unsigned long long) m.bytes);
      return fail(c, ncclInvalidArgument);
    }
    if (m.coll_epoch != c->coll_epoch) {
      say(c, "ncclRecv #%llu from rank %d: the sender had issued %llu collectives, this rank %llu -- point-to-point "
             "operations and collectives are not in one order on the two ranks",
          (unsigned long long) seq, o.peer, (unsigned long long) m.coll_epoch, (unsigned long long) c->coll_epoch);
      return fail(c, ncclInvalidUsage);
    }
    if (o.bytes) {
      const std::string name = msg_name(c, o.peer, c->rank, seq);
      const int fd = shm_open(name.c_str(), O_RDONLY, 0600);
      if (fd < 0) {
        say(c, "shm_open(%s) failed", name.c_str());
        return fail(c, ncclSystemError);
      }
      void *p = mmap(nullptr, o.bytes, PROT_READ, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) return fail(c, ncclSystemError);
      const hipError_t e = copy(o.buf, p, o.bytes, hipMemcpyHostToDevice);
      munmap(p, o.bytes);
      shm_unlink(name.c_str());
      if (e != hipSuccess) {
        say(c, "copy into a receive buffer failed: %s", hipGetErrorString(e));
        return fail(c, ncclUnhandledCudaError);
      }
    }
    ch.consumed.store(seq + 1, std::memory_order_release);
    c->received[o.peer] = seq + 1;
  }
  for (const Op &o : ops) {
    if (!o.send) continue;
    Comm *c = o.comm;
    Chan &ch = c->sh->chan[c->rank][o.peer];
    const uint64_t upto = c->sent[o.peer];
    if (!wait_for(c, [&] { return ch.consumed.load(std::memory_order_acquire) >= upto; })) {
      if (c->sh->failed.load()) return ncclRemoteError;
      say(c, "a ncclSend to rank %d (%zu bytes) was never received: the ranks' exchange schedules differ", o.peer, o.bytes);
      return fail(c, ncclSystemError);
    }
  }
  return ncclSuccess;
}

ncclResult_t p2p(bool send, void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || !c->sh) return ncclInvalidArgument;
  const size_t ts = type_size(dt);
  if (!ts || peer < 0 || peer >= c->nranks || (count && !buf)) {
    say(c, "%s: bad argument (peer %d, count %zu)", send ? "ncclSend" : "ncclRecv", peer, count);
    if (g_depth) g_group_error = ncclInvalidArgument;
    return ncclInvalidArgument;
  }
  Op o{send, buf, count * ts, peer, c, st};
  if (g_depth) {
    g_ops.push_back(o);
    return ncclSuccess;
  }
  std::vector<Op> one{o};
  return run_group(one);
}

// kind 1 = all-gather, 2 = all-reduce
ncclResult_t collective(int kind, const void *sb, void *rb, size_t count, ncclDataType_t dt, int op, ncclComm_t comm,
                        hipStream_t st)
{
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || !c->sh) return ncclInvalidArgument;
  if (g_depth) {
    say(c, "collectives inside ncclGroupStart/End are not part of this double");
    return ncclInvalidUsage;
  }
  const size_t ts = type_size(dt), bytes = count * ts;
  if (!ts || !bytes || bytes > kCollBytes || !sb || !rb) {
    say(c, "collective: bad argument (count %zu)", count);
    return ncclInvalidArgument;
  }
  if (kind == 2 && (dt != ncclFloat64 || (op != ncclSum && op != ncclMax))) {
    say(c, "ncclAllReduce: this double reduces doubles with sum or max");
    return ncclInvalidArgument;
  }
  if (c->sh->failed.load()) return ncclRemoteError;
  Shared *s = c->sh;
  const uint64_t epoch = ++c->coll_epoch;
  if (stream_sync(st) != hipSuccess) return fail(c, ncclUnhandledCudaError);
  if (copy(s->coll[c->rank], sb, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, ncclUnhandledCudaError);
  s->desc[c->rank] = CollDesc{(uint64_t) kind, (uint64_t) count, (uint64_t) dt, (uint64_t) op, epoch};
  const char *name = kind == 1 ? "ncclAllGather" : "ncclAllReduce";
  if (!barrier(c)) {
    if (!s->failed.load()) say(c, "%s (collective #%llu) was not joined by every rank", name, (unsigned long long) epoch);
    return fail(c, ncclSystemError);
  }
  for (int q = 0; q < c->nranks; q++) {
    const CollDesc &d = s->desc[q];
    if (d.kind != (uint64_t) kind || d.count != (uint64_t) count || d.dtype != (uint64_t) dt || d.op != (uint64_t) op ||
        d.epoch != epoch) {
      say(c, "%s #%llu (count %zu): rank %d is in collective #%llu of kind %llu, count %llu -- the ranks' collectives "
             "are not in one order",
          name, (unsigned long long) epoch, count, q, (unsigned long long) d.epoch, (unsigned long long) d.kind,
          (unsigned long long) d.count);
      return fail(c, ncclInvalidUsage);
    }
  }
  std::vector<char> out(kind == 1 ? bytes * c->nranks : bytes);
  if (kind == 1) {
    for (int q = 0; q < c->nranks; q++) memcpy(out.data() + q * bytes, s->coll[q], bytes);
  } else {
    double *o = reinterpret_cast<double *>(out.data());
    for (size_t k = 0; k < count; k++) {
      double v = reinterpret_cast<const double *>(s->coll[0])[k]; // (rank order: every rank gets the same bits)
      for (int q = 1; q < c->nranks; q++) {
        const double w = reinterpret_cast<const double *>(s->coll[q])[k];
        v = op == ncclSum ? v + w : (w > v ? w : v);
      }
      o[k] = v;
    }
  }
  if (!barrier(c)) return fail(c, ncclSystemError); // (nobody overwrites its slot before everyone has read)
  if (copy(rb, out.data(), out.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(c, ncclUnhandledCudaError);
  return ncclSuccess;
}

} // namespace

extern "C" {

// marker the library looks for: a communicator on this object is a rehearsal, and every report says so
int mdp_fake_rccl_marker(void) { return 1; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  if (!id) return ncclInvalidArgument;
  static std::atomic<unsigned> counter{0};
  memset(id, 0, sizeof *id);
  const auto now = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now().time_since_epoch()).count();
  snprintf(id->internal, sizeof id->internal, "/mdpfr_%d_%u_%llx", (int) getpid(), counter.fetch_add(1),
           (unsigned long long) now);
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t) sizeof(Shared)) != 0) {
    close(fd);
    shm_unlink(id->internal);
    return ncclSystemError;
  }
  void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  static_cast<Shared *>(p)->magic = kMagic; // (a fresh object is zero-filled: every counter starts at 0)
  munmap(p, sizeof(Shared));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[sizeof id.internal - 1] = 0;
  const int fd = shm_open(id.internal, O_RDWR, 0600);
  if (fd < 0) {
    fprintf(stderr, "[fake-rccl] rank %d: no communicator object '%s'\n", rank, id.internal);
    return ncclInvalidArgument;
  }
  void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm *c = new Comm;
  c->sh = static_cast<Shared *>(p);
  c->nranks = nranks;
  c->rank = rank;
  c->base = id.internal;
  if (c->sh->magic != kMagic) {
    say(c, "'%s' is not a communicator object of this double", id.internal);
    munmap(p, sizeof(Shared));
    delete c;
    return ncclInvalidArgument;
  }
  int expect = 0;
  if (!c->sh->nranks.compare_exchange_strong(expect, nranks) && expect != nranks) {
    say(c, "ncclCommInitRank with %d ranks, another rank said %d", nranks, expect);
    munmap(p, sizeof(Shared));
    delete c;
    return ncclInvalidArgument;
  }
  c->sh->attached.fetch_add(1);
  if (!wait_for(c, [&] { return c->sh->attached.load() >= nranks; })) {
    say(c, "only %d of %d ranks called ncclCommInitRank", c->sh->attached.load(), nranks);
    c->sh->failed.store(1);
    munmap(p, sizeof(Shared));
    delete c;
    return ncclSystemError;
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c) return ncclInvalidArgument;
  ncclResult_t r = ncclSuccess;
  for (int q = 0; q < c->nranks; q++) { // what this rank sent and nobody took (its matching receive was never issued)
    Chan &ch = c->sh->chan[c->rank][q];
    const uint64_t left = ch.posted.load() - ch.consumed.load();
    if (left && !c->sh->failed.load()) {
      say(c, "%llu sends to rank %d were never received", (unsigned long long) left, q);
      r = ncclInvalidUsage;
    }
    for (uint64_t k = ch.consumed.load(); k < ch.posted.load(); k++) shm_unlink(msg_name(c, c->rank, q, k).c_str());
  }
  if (c->sh->detached.fetch_add(1) + 1 == c->nranks) shm_unlink(c->base.c_str());
  munmap(c->sh, sizeof(Shared));
  delete c;
  return r;
}

const char *ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
  case ncclSuccess: return "no error";
  case ncclUnhandledCudaError: return "fake-rccl: HIP call failed";
  case ncclSystemError: return "fake-rccl: unmatched operation (timeout) or shared-memory failure, see stderr";
  case ncclInvalidArgument: return "fake-rccl: invalid argument or mismatched message size, see stderr";
  case ncclInvalidUsage: return "fake-rccl: operations in different orders on different ranks, see stderr";
  case ncclRemoteError: return "fake-rccl: another rank of the communicator failed";
  default: return "fake-rccl: error";
  }
}

ncclResult_t ncclGroupStart()
{
  if (g_depth++ == 0) {
    g_ops.clear();
    g_group_error = ncclSuccess;
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (g_group_error != ncclSuccess) return g_group_error;
  return ops.empty() ? ncclSuccess : run_group(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
  return p2p(true, const_cast<void *>(buf), count, dt, peer, comm, st);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
  return p2p(false, buf, count, dt, peer, comm, st);
}

ncclResult_t ncclAllGather(const void *sb, void *rb, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st)
{
  return collective(1, sb, rb, count, dt, 0, comm, st);
}

ncclResult_t ncclAllReduce(const void *sb, void *rb, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t st)
{
  return collective(2, sb, rb, count, dt, (int) op, comm, st);
}

} // extern "C"
