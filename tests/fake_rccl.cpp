// TEST INFRASTRUCTURE, not product: a transport stand-in for the ten RCCL entry points that
// epa_ng_amd/csrc/comm.hip binds with dlopen (EPA_RCCL_LIB=<this library>).  RCCL refuses two ranks on
// one device, so on a 1-GPU box the product's gather protocol (comm.hip, host/place_ranks.cpp) could only
// ever run with world == 1.  This library moves the bytes between SAME-DEVICE processes instead:
//
//   ncclSend  = D2H copy into a pinned staging buffer, then a stream-ordered host function that writes
//               the message into a file under /dev/shm (or /tmp) and publishes its sequence number in a
//               shared control block;
//   ncclRecv  = a stream-ordered host function that waits for that sequence number, reads the message
//               into a pinned staging buffer, then an H2D copy into the destination;
//   ncclAllReduce = the same through per-round slots of the control block (small counts only);
//   ncclGroupStart / End = ops are queued and issued at GroupEnd, sends before receives.
//
// Everything is ordered on the caller's stream exactly like the real calls, nothing synchronises the
// host, and every wait has a timeout (EPA_FAKE_RCCL_TIMEOUT_S, default 120) that prints, poisons the
// communicator and lets the stream continue -- a hung test must not hang the box.  Semantics covered:
// point-to-point send / recv with matching sizes, in-order per (source, destination) channel.
// Fault knob: EPA_FAKE_RCCL_HANG_RECV=1 makes every ncclRecv wait for a message that is never looked at (until
// the communicator is aborted or the stand-in's own timeout) -- "the transport cannot move data between these
// processes", the failure a first real multi-GPU run could meet; what epa_comm_probe exists for.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../epa_ng_amd/csrc/rccl_abi.hpp"

namespace {

constexpr int MAXR = 16;
constexpr int AR_SLOTS = 8;
constexpr uint64_t WINDOW = 2;   // messages a sender may run ahead of its receiver per channel

struct Ctl {   // one shared file per unique id
  std::atomic<uint32_t> arrived, departed, poisoned;
  struct Chan { std::atomic<uint64_t> sent, recvd; } chan[MAXR][MAXR];
  struct Ar {
    std::atomic<uint64_t> arrive, depart;
    unsigned char vals[MAXR][64];
  } ar[AR_SLOTS];
};

struct IdBlob {   // what travels in the 128 id bytes
  char magic[8];
  char token[24];
  char dir[96];
};
static_assert(sizeof(IdBlob) == 128, "id layout");

double timeout_s() {
  const char* e = getenv("EPA_FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

struct Staging { void* p = nullptr; size_t size = 0; std::atomic<int> busy{0}; };

}  // namespace

struct ncclComm {
  Ctl* ctl = nullptr;
  std::string dir, token;
  int rank = 0, world = 1;
  uint64_t send_seq[MAXR] = {}, recv_seq[MAXR] = {}, ar_seq = 0;
  std::mutex mu;
  std::vector<Staging*> pool;
  std::atomic<int> failed{0};

  std::string ctl_path() const { return dir + "/epa_frccl_" + token + "_ctl"; }
  std::string msg_path(int s, int d, uint64_t seq) const {
    return dir + "/epa_frccl_" + token + "_" + std::to_string(s) + "_" + std::to_string(d) + "_" + std::to_string(seq);
  }
  Staging* staging(size_t n) {
    std::lock_guard<std::mutex> g(mu);
    for (Staging* s : pool)
      if (s->size >= n && !s->busy.load(std::memory_order_acquire)) { s->busy = 1; return s; }
    Staging* s = new Staging;
    if (hipHostMalloc(&s->p, n ? n : 1) != hipSuccess) { delete s; return nullptr; }
    s->size = n; s->busy = 1;
    pool.push_back(s);
    return s;
  }
};

namespace {

struct Op {   // one queued point-to-point operation / one host-function payload
  ncclComm* c; int kind;   // 0 send, 1 recv, 2 release staging
  void* dev; size_t n; int peer; hipStream_t st; uint64_t seq; Staging* sg;
};

thread_local int g_group = 0;
thread_local std::vector<Op>* g_queue = nullptr;

template <class Pred>
bool wait_for(ncclComm* c, Pred ok, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  const double lim = timeout_s();
  unsigned spins = 0;
  while (!ok()) {
    if (c->ctl->poisoned.load(std::memory_order_acquire) || c->failed.load()) return false;
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
      fprintf(stderr, "fake_rccl: rank %d timed out after %.0f s waiting for %s\n", c->rank, lim, what);
      c->failed = 1;
      c->ctl->poisoned.store(1, std::memory_order_release);
      return false;
    }
  }
  return true;
}

void host_send(void* u) {
  Op* o = (Op*)u;
  ncclComm* c = o->c;
  Ctl::Chan& ch = c->ctl->chan[c->rank][o->peer];
  if (wait_for(c, [&] { return o->seq < ch.recvd.load(std::memory_order_acquire) + WINDOW; }, "send window")) {
    const std::string path = c->msg_path(c->rank, o->peer, o->seq);
    int fd = open(path.c_str(), O_CREAT | O_RDWR | O_TRUNC, 0600);
    bool good = fd >= 0;
    if (good && o->n) {
      good = ftruncate(fd, (off_t)o->n) == 0;
      if (good) {
        void* m = mmap(nullptr, o->n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        good = m != MAP_FAILED;
        if (good) { memcpy(m, o->sg->p, o->n); munmap(m, o->n); }
      }
    }
    if (fd >= 0) close(fd);
    if (!good) {
      fprintf(stderr, "fake_rccl: rank %d cannot write %s\n", c->rank, path.c_str());
      c->failed = 1; c->ctl->poisoned.store(1);
    }
    ch.sent.store(o->seq + 1, std::memory_order_release);
  }
  o->sg->busy.store(0, std::memory_order_release);
  delete o;
}

void host_recv(void* u) {
  Op* o = (Op*)u;
  ncclComm* c = o->c;
  Ctl::Chan& ch = c->ctl->chan[o->peer][c->rank];
  static const bool hang = getenv("EPA_FAKE_RCCL_HANG_RECV") != nullptr;
  if (wait_for(c, [&] { return !hang && ch.sent.load(std::memory_order_acquire) > o->seq; }, "a message")) {
    const std::string path = c->msg_path(o->peer, c->rank, o->seq);
    int fd = open(path.c_str(), O_RDONLY);
    bool good = fd >= 0;
    if (good && o->n) {
      struct stat sb;
      good = fstat(fd, &sb) == 0 && (size_t)sb.st_size == o->n;   // send / recv sizes must match
      if (good) {
        void* m = mmap(nullptr, o->n, PROT_READ, MAP_SHARED, fd, 0);
        good = m != MAP_FAILED;
        if (good) { memcpy(o->sg->p, m, o->n); munmap(m, o->n); }
      }
    }
    if (fd >= 0) close(fd);
    unlink(path.c_str());
    if (!good) {
      fprintf(stderr, "fake_rccl: rank %d cannot read %s (size mismatch?)\n", c->rank, path.c_str());
      c->failed = 1; c->ctl->poisoned.store(1);
    }
    ch.recvd.store(o->seq + 1, std::memory_order_release);
  }
  delete o;
}

void host_release(void* u) {
  Op* o = (Op*)u;
  o->sg->busy.store(0, std::memory_order_release);
  delete o;
}

ncclResult_t issue(const Op& q) {
  ncclComm* c = q.c;
  Staging* sg = c->staging(q.n);
  if (!sg) return ncclUnhandledCudaError;
  if (q.kind == 0) {
    if (q.n && hipMemcpyAsync(sg->p, q.dev, q.n, hipMemcpyDeviceToHost, q.st) != hipSuccess) return ncclUnhandledCudaError;
    Op* o = new Op(q); o->sg = sg;
    if (hipLaunchHostFunc(q.st, host_send, o) != hipSuccess) return ncclUnhandledCudaError;
  } else {
    Op* o = new Op(q); o->sg = sg;
    if (hipLaunchHostFunc(q.st, host_recv, o) != hipSuccess) return ncclUnhandledCudaError;
    if (q.n && hipMemcpyAsync(q.dev, sg->p, q.n, hipMemcpyHostToDevice, q.st) != hipSuccess) return ncclUnhandledCudaError;
    Op* r = new Op(q); r->sg = sg; r->kind = 2;
    if (hipLaunchHostFunc(q.st, host_release, r) != hipSuccess) return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

size_t type_size(ncclDataType_t t) {
  switch ((int)t) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: return 2;
    default: return 0;
  }
}

struct ArOp { ncclComm* c; Staging* sg; size_t count; ncclDataType_t t; ncclRedOp_t op; uint64_t round; };

template <class T>
void reduce(const Ctl::Ar& a, int world, size_t count, ncclRedOp_t op, void* out) {
  T* o = (T*)out;
  for (size_t i = 0; i < count; ++i) {
    T acc;
    memcpy(&acc, a.vals[0] + i * sizeof(T), sizeof(T));
    for (int r = 1; r < world; ++r) {
      T v;
      memcpy(&v, a.vals[r] + i * sizeof(T), sizeof(T));
      switch (op) {
        case ncclSum: case ncclAvg: acc = (T)(acc + v); break;
        case ncclProd: acc = (T)(acc * v); break;
        case ncclMax: acc = v > acc ? v : acc; break;
        case ncclMin: acc = v < acc ? v : acc; break;
      }
    }
    if (op == ncclAvg) acc = (T)(acc / (T)world);
    o[i] = acc;
  }
}

void host_allreduce(void* u) {
  ArOp* o = (ArOp*)u;
  ncclComm* c = o->c;
  Ctl::Ar& a = c->ctl->ar[o->round % AR_SLOTS];
  const uint64_t gen = o->round / AR_SLOTS;
  const size_t bytes = o->count * type_size(o->t);
  if (wait_for(c, [&] { return a.depart.load(std::memory_order_acquire) >= gen * (uint64_t)c->world; }, "an all-reduce slot")) {
    memcpy(a.vals[c->rank], o->sg->p, bytes);
    a.arrive.fetch_add(1, std::memory_order_acq_rel);
    if (wait_for(c, [&] { return a.arrive.load(std::memory_order_acquire) >= (gen + 1) * (uint64_t)c->world; }, "all-reduce peers")) {
      switch ((int)o->t) {
        case 0: reduce<int8_t>(a, c->world, o->count, o->op, o->sg->p); break;
        case 1: reduce<uint8_t>(a, c->world, o->count, o->op, o->sg->p); break;
        case 2: reduce<int32_t>(a, c->world, o->count, o->op, o->sg->p); break;
        case 3: reduce<uint32_t>(a, c->world, o->count, o->op, o->sg->p); break;
        case 4: reduce<int64_t>(a, c->world, o->count, o->op, o->sg->p); break;
        case 5: reduce<uint64_t>(a, c->world, o->count, o->op, o->sg->p); break;
        case 7: reduce<float>(a, c->world, o->count, o->op, o->sg->p); break;
        case 8: reduce<double>(a, c->world, o->count, o->op, o->sg->p); break;
        default: break;
      }
      a.depart.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  delete o;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  IdBlob b;
  memset(&b, 0, sizeof(b));
  memcpy(b.magic, "EPAFRCCL", 8);
  const char* dir = getenv("EPA_FAKE_RCCL_DIR");
  std::string d = dir ? dir : "";
  if (d.empty()) {
    struct statvfs sv;
    d = (statvfs("/dev/shm", &sv) == 0 && (double)sv.f_bavail * (double)sv.f_frsize > 2e9) ? "/dev/shm" : "/tmp";
  }
  if (d.size() >= sizeof(b.dir)) return ncclInvalidArgument;
  memcpy(b.dir, d.c_str(), d.size());
  const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(b.token, sizeof(b.token), "%x_%llx", (unsigned)getpid(), (unsigned long long)(t & 0xffffffffffffull));
  // the control block exists (zeroed) before anyone can learn the id
  const std::string path = d + "/epa_frccl_" + b.token + "_ctl";
  int fd = open(path.c_str(), O_CREAT | O_RDWR | O_EXCL, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Ctl)) != 0) { if (fd >= 0) close(fd); return ncclSystemError; }
  close(fd);
  memcpy(id, &b, sizeof(b));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  IdBlob b;
  memcpy(&b, &id, sizeof(b));
  if (memcmp(b.magic, "EPAFRCCL", 8) != 0 || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  b.token[sizeof(b.token) - 1] = 0; b.dir[sizeof(b.dir) - 1] = 0;
  ncclComm* c = new ncclComm;
  c->dir = b.dir; c->token = b.token; c->rank = rank; c->world = nranks;
  int fd = open(c->ctl_path().c_str(), O_RDWR);
  if (fd < 0) { delete c; return ncclSystemError; }
  void* m = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return ncclSystemError; }
  c->ctl = (Ctl*)m;
  c->ctl->arrived.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_for(c, [&] { return c->ctl->arrived.load(std::memory_order_acquire) >= (uint32_t)nranks; }, "the other ranks in ncclCommInitRank")) {
    munmap(m, sizeof(Ctl)); delete c;
    return ncclSystemError;
  }
  *out = c;
  return ncclSuccess;
}

static ncclResult_t teardown(ncclComm_t c, bool abort) {
  if (!c) return ncclInvalidArgument;
  if (abort) { c->failed = 1; c->ctl->poisoned.store(1, std::memory_order_release); }
  (void)hipDeviceSynchronize();   // every queued host function has run (or given up)
  const uint32_t gone = c->ctl->departed.fetch_add(1, std::memory_order_acq_rel) + 1;
  if (gone >= (uint32_t)c->world || abort) {
    unlink(c->ctl_path().c_str());
    if (abort)   // messages nobody will read
      for (int s = 0; s < c->world; ++s)
        for (int d = 0; d < c->world; ++d)
          for (uint64_t q = c->ctl->chan[s][d].recvd.load(); q < c->ctl->chan[s][d].sent.load(); ++q)
            unlink(c->msg_path(s, d, q).c_str());
  }
  munmap(c->ctl, sizeof(Ctl));
  for (Staging* s : c->pool) { (void)hipHostFree(s->p); delete s; }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { return teardown(c, false); }
ncclResult_t ncclCommAbort(ncclComm_t c) { return teardown(c, true); }

ncclResult_t ncclGroupStart() {
  if (!g_queue) g_queue = new std::vector<Op>;
  ++g_group;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_group <= 0) return ncclInvalidUsage;
  if (--g_group) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  for (int kind = 0; kind < 2; ++kind)   // sends first: a rank that sends to itself must not wait for its own message
    for (const Op& q : *g_queue)
      if (q.kind == kind && rc == ncclSuccess) rc = issue(q);
  g_queue->clear();
  return rc;
}

static ncclResult_t p2p(int kind, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  if (!c || peer < 0 || peer >= c->world || !type_size(t)) return ncclInvalidArgument;
  if (c->failed.load() || c->ctl->poisoned.load()) return ncclRemoteError;
  Op q{c, kind, buf, count * type_size(t), peer, st, kind == 0 ? c->send_seq[peer]++ : c->recv_seq[peer]++, nullptr};
  if (g_group > 0) { g_queue->push_back(q); return ncclSuccess; }
  return issue(q);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  return p2p(0, const_cast<void*>(buf), count, t, peer, c, st);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  return p2p(1, buf, count, t, peer, c, st);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
  const size_t bytes = count * type_size(t);
  if (!c || !bytes || bytes > 64 || (int)t == 6) return ncclInvalidArgument;
  if (c->failed.load() || c->ctl->poisoned.load()) return ncclRemoteError;
  Staging* sg = c->staging(bytes);
  if (!sg) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(sg->p, send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
  ArOp* o = new ArOp{c, sg, count, t, op, c->ar_seq++};
  if (hipLaunchHostFunc(st, host_allreduce, o) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync(recv, sg->p, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  Op* r = new Op{c, 2, nullptr, 0, 0, st, 0, sg};
  if (hipLaunchHostFunc(st, host_release, r) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: HIP error";
    case ncclSystemError: return "fake_rccl: system error (shared files / timeout)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    case ncclRemoteError: return "fake_rccl: a peer failed or timed out";
    default: return "fake_rccl: error";
  }
}

}  // extern "C"
