// The path's only exchange between the GPUs of a node, in the product library: the gather of every
// chunk's (pair, result) rows to rank 0 over RCCL (xGMI point-to-point links into the root, no ring).
//
// Replaces, for the one-process-per-GPU mode, what the reference's MPI build does around the chunk
// loop: queries are split into contiguous per-rank slices without any exchange
// (src/net/epa_mpi_util.cpp:10-30, local_seq_package) and the per-rank results are gathered to rank 0,
// which writes the jplace (src/io/jplace_writer.hpp:117-129 gathers the ranks' text; here the 32-byte
// numeric rows travel and rank 0 runs LWR / filter / text).
//
// Protocol (the same as epa_ng_amd/parallel.py AsyncResultGather, which remains the torch.distributed
// harness of bench.py): every gather has the SAME fixed size on every rank -- rows_cap rows of 32 bytes
// + one sentinel row carrying the valid count -- so a post only enqueues work: no size negotiation, no
// host synchronisation.  A rank whose chunk produced more rows sends the first rows_cap and carries the
// rest into its next gather; epa_comm_flush() drains what is still carried with extra rounds agreed on
// by the object's only all-reduce.  `depth` send / receive slots are used round robin.  Everything runs
// on the communicator's own stream, ordered behind the producer's stream by an event: the next chunk's
// kernels never wait for a transfer.
//
// RCCL is bound at run time (dlopen): libepa_dev.so has no link-time dependency on it, and a process
// that never creates a communicator never loads it.
#include "epa_dev_internal.hpp"

#include "rccl_abi.hpp"   // the ten entry points' shapes, declared locally: no RCCL headers needed to build

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

static_assert(sizeof(epa_row) == 32, "row layout");
static_assert(sizeof(ncclUniqueId) == EPA_COMM_ID_BYTES, "unique id size");

namespace {

struct Rccl : epa_rccl_api {
  void* h = nullptr;
  std::string err, path;
};
bool g_rccl_loaded = false;

// Which library carries the ten entry points, in this order:
//   1. the path handed to epa_comm_set_library() (an API call: the host program decides),
//   2. EPA_RCCL_LIB (the same as an environment override: how the CLI and the tests point at
//      tests/fake_rccl.cpp, a same-device transport stand-in, so that world > 1 runs on a 1-GPU box),
//   3. a librccl that is ALREADY MAPPED in this process (/proc/self/maps) -- a host program that brought
//      its own RCCL (PyTorch ships torch/lib/librccl.so) must not end up with a second copy of the
//      library, with its own topology detection and its own IPC state, beside the one it already uses,
//   4. the loader's search path: librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1.
std::string g_rccl_pref;          // epa_comm_set_library
std::mutex g_rccl_mu;

std::string mapped_rccl() {
  std::string found;
  if (FILE* f = fopen("/proc/self/maps", "r")) {
    char line[4352];
    while (fgets(line, sizeof line, f)) {
      const char* sl = strchr(line, '/');
      if (!sl) continue;
      std::string path(sl);
      while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
      const size_t b = path.rfind('/');
      if (path.compare(b + 1, 10, "librccl.so") == 0) { found = path; break; }
    }
    fclose(f);
  }
  return found;
}

void load_rccl(Rccl& r) {
  std::vector<std::string> names;
  {
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (!g_rccl_pref.empty()) names.push_back(g_rccl_pref);
  }
  const bool pinned = !names.empty();
  if (const char* e = getenv("EPA_RCCL_LIB")) if (*e && !pinned) names.push_back(e);
  if (names.empty()) {   // an explicit choice is final: no silent fallback to another library
    const std::string m = mapped_rccl();
    if (!m.empty()) names.push_back(m);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
  }
  for (const std::string& n : names) {
    r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (r.h) { r.path = n; break; }
    r.err = dlerror();
  }
  if (!r.h) return;
#define SYM(field, name)                                              \
  *(void**)(&r.field) = dlsym(r.h, name);                             \
  if (!r.field) { r.err = std::string("missing symbol ") + name; r.h = nullptr; return; }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(CommAbort, "ncclCommAbort")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(AllReduce, "ncclAllReduce")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  Dl_info di;   // the file the loader really bound (a bare soname resolves through the search path)
  if (dladdr((void*)r.CommInitRank, &di) && di.dli_fname && *di.dli_fname) r.path = di.dli_fname;
}

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] { load_rccl(r); std::lock_guard<std::mutex> g(g_rccl_mu); g_rccl_loaded = true; });
  return &r;
}

// the wait of a collective's host side: an event that a healthy job completes in milliseconds.  A peer
// that died leaves it pending for ever (the reference's MPI build would abort the job): give up after the
// communicator's timeout -- epa_comm_set_timeout(), else the process default (epa_comm_set_timeout(NULL, s)),
// else EPA_COMM_TIMEOUT_S, else 600 s -- so that the caller can epa_comm_abort() and exit non-zero.
double g_default_timeout = 0.0;   // 0: not set through the API
double comm_timeout_s() {
  if (g_default_timeout > 0) return g_default_timeout;
  const char* e = getenv("EPA_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 600.0;
  return v > 0 ? v : 600.0;
}

hipError_t wait_event(hipEvent_t ev, bool* timed_out, double lim) {
  *timed_out = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (spin > 4000) std::this_thread::sleep_for(std::chrono::microseconds(spin > 40000 ? 200 : 20));
    if ((spin & 255) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
      *timed_out = true;
      return hipSuccess;
    }
  }
}

// rows [0, n) of dst <- (pairs, results)[src_off + i], sequence ids made global
__global__ void __launch_bounds__(256) k_pack_rows(const epa_pair* __restrict__ pairs, const epa_result* __restrict__ res,
                                                   uint64_t src_off, uint64_t n, uint32_t seq_offset,
                                                   epa_row* __restrict__ dst) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const epa_pair p = pairs[src_off + i];
  const epa_result r = res[src_off + i];
  epa_row o;
  o.branch_id = p.branch_id;
  o.seq_id = p.seq_id + seq_offset;
  o.lnl = r.lnl;
  o.pendant_length = r.pendant_length;
  o.distal_length = r.distal_length;
  dst[i] = o;
}

__global__ void k_sentinel(epa_row* row, uint32_t count, double pending) {
  epa_row o;
  o.branch_id = count;
  o.seq_id = 0xE9A0C0DEu;   // marks a sentinel row
  o.lnl = (double)count;
  o.pendant_length = pending;   // rows this rank still carries after this gather
  o.distal_length = 0.0;
  *row = o;
}

}  // namespace

struct epa_comm {
  epa_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, depth = 2;
  uint32_t cap = 0;
  bool self_send = false;       // rank 0's own rows travel through ncclSend / ncclRecv to itself (test switch)
  hipStream_t cs = nullptr;
  hipEvent_t ev_src = nullptr, ev_packed = nullptr;
  struct GSlot {
    epa_row* send = nullptr;    // [cap + 1]
    epa_row* recv = nullptr;    // rank 0: [world][cap + 1]
    epa_row* h_cnt = nullptr;   // rank 0, pinned: [world] sentinel rows
    epa_row* h_rows = nullptr;  // rank 0, pinned: [world][cap]
    hipEvent_t ev_gather = nullptr, ev_host = nullptr;
    uint64_t ticket = ~0ull;
    std::vector<uint32_t> counts;
    std::vector<const epa_row*> ptrs;
  };
  std::vector<GSlot> gs;
  epa_row* carry[2] = {nullptr, nullptr};
  size_t carry_cap[2] = {0, 0};
  uint64_t carry_n = 0;
  int carry_cur = 0;
  uint64_t carried_rows = 0, next_ticket = 0;
  unsigned long long* d_pend = nullptr;   // all-reduce scratch (2 words)
  unsigned long long* h_pend = nullptr;   // pinned
  epa_row* d_probe = nullptr;             // one row: this rank's identity in epa_comm_probe
  double timeout = 0.0;                   // seconds a host-side wait may take; 0 = the process default
  double lim() const { return timeout > 0 ? timeout : comm_timeout_s(); }
};

#define EPA_NCCL(ctx, call)                                                                               \
  do {                                                                                                    \
    ncclResult_t r__ = (call);                                                                            \
    if (r__ != ncclSuccess)                                                                               \
      return epa_fail(ctx, EPA_ERR_HIP, std::string(#call) + ": " + rccl()->GetErrorString(r__));        \
  } while (0)

extern "C" int epa_comm_set_library(const char* path) {
  std::lock_guard<std::mutex> g(g_rccl_mu);
  if (g_rccl_loaded) return epa_fail(nullptr, EPA_ERR_INVALID_ARG, "comm_set_library: the transport library is already loaded");
  g_rccl_pref = path ? path : "";
  return EPA_OK;
}

extern "C" const char* epa_comm_library_path(void) {
  Rccl* R = rccl();
  return R->h ? R->path.c_str() : "";
}

extern "C" int epa_comm_set_timeout(epa_comm* c, double seconds) {
  if (!(seconds >= 0)) return EPA_ERR_INVALID_ARG;
  if (c) c->timeout = seconds; else g_default_timeout = seconds;
  return EPA_OK;
}

extern "C" int epa_comm_set_self_send(epa_comm* c, int on) {
  if (!c || c->next_ticket) return EPA_ERR_INVALID_ARG;
  c->self_send = on != 0;
  return EPA_OK;
}

extern "C" int epa_comm_get_unique_id(void* id128) {
  Rccl* R = rccl();
  if (!R->h) return epa_fail(nullptr, EPA_ERR_UNSUPPORTED, "RCCL not available: " + R->err);
  ncclUniqueId id;
  const ncclResult_t rc = R->GetUniqueId(&id);
  if (rc != ncclSuccess) return epa_fail(nullptr, EPA_ERR_HIP, std::string("ncclGetUniqueId: ") + R->GetErrorString(rc));
  std::memcpy(id128, &id, sizeof(id));
  return EPA_OK;
}

extern "C" void epa_comm_destroy(epa_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  if (c->cs) (void)hipStreamSynchronize(c->cs);
  for (auto& g : c->gs) {
    if (g.send) (void)hipFree(g.send);
    if (g.recv) (void)hipFree(g.recv);
    if (g.h_cnt) (void)hipHostFree(g.h_cnt);
    if (g.h_rows) (void)hipHostFree(g.h_rows);
    if (g.ev_gather) (void)hipEventDestroy(g.ev_gather);
    if (g.ev_host) (void)hipEventDestroy(g.ev_host);
  }
  for (int i = 0; i < 2; ++i)
    if (c->carry[i]) (void)hipFree(c->carry[i]);
  if (c->d_pend) (void)hipFree(c->d_pend);
  if (c->d_probe) (void)hipFree(c->d_probe);
  if (c->h_pend) (void)hipHostFree(c->h_pend);
  if (c->ev_src) (void)hipEventDestroy(c->ev_src);
  if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
  if (c->comm) (void)rccl()->CommDestroy(c->comm);
  if (c->cs) (void)hipStreamDestroy(c->cs);
  delete c;
}

extern "C" int epa_comm_create(epa_ctx* ctx, const void* id128, int rank, int world, uint32_t rows_cap, int depth,
                               epa_comm** out) {
  if (!ctx || !out || !id128) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_create: null argument");
  if (world < 1 || rank < 0 || rank >= world || rows_cap == 0 || depth < 1 || depth > 8)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_create: need 0 <= rank < world, rows_cap > 0, 1 <= depth <= 8");
  Rccl* R = rccl();
  if (!R->h) return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "RCCL not available: " + R->err);
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  epa_comm* c = new epa_comm;
  c->ctx = ctx; c->rank = rank; c->world = world; c->cap = rows_cap; c->depth = depth;
  auto fail = [&](int rc) { epa_comm_destroy(c); return rc; };
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  {
    // ncclCommInitRank blocks until every rank has joined: a rank that never arrives would hold this one for ever.
    // It runs on a helper thread; past the timeout the caller gets an error (the helper, still inside RCCL's
    // bootstrap, is left behind -- the process is expected to fall back or exit)
    struct Init { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t rc = ncclSuccess; ncclComm_t comm = nullptr; };
    auto st = std::make_shared<Init>();
    const int dev = ctx->device;
    std::thread([st, R, world, id, rank, dev] {
      (void)hipSetDevice(dev);
      ncclComm_t cm = nullptr;
      const ncclResult_t rc = R->CommInitRank(&cm, world, id, rank);
      std::lock_guard<std::mutex> g(st->mu);
      st->rc = rc; st->comm = cm; st->done = true;
      st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(st->mu);
    const double lim = comm_timeout_s();
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(lim), [&] { return st->done; }))
      return fail(epa_fail(ctx, EPA_ERR_HIP, "ncclCommInitRank: not all " + std::to_string(world) + " ranks joined within " +
                                                 std::to_string((int)lim) + " s"));
    if (st->rc != ncclSuccess) {
      c->comm = nullptr;
      return fail(epa_fail(ctx, EPA_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(st->rc)));
    }
    c->comm = st->comm;
  }
#define TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(epa_fail(ctx, EPA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__))); } while (0)
  {
    // the exchange runs beside the next chunk's kernels, which fill the device: the highest stream priority lets
    // the (small) pack / send / receive kernels take the first slot that frees up instead of queueing behind them
    int lo = 0, hi = 0;
    TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    TRY(hipStreamCreateWithPriority(&c->cs, hipStreamNonBlocking, hi));
  }
  TRY(hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming));
  TRY(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
  TRY(hipMalloc((void**)&c->d_pend, 16));
  TRY(hipMalloc((void**)&c->d_probe, sizeof(epa_row)));
  TRY(hipHostMalloc((void**)&c->h_pend, 16));
  const size_t msg = sizeof(epa_row) * ((size_t)rows_cap + 1);
  c->gs.resize(depth);
  for (auto& g : c->gs) {
    TRY(hipMalloc((void**)&g.send, msg));
    TRY(hipEventCreateWithFlags(&g.ev_gather, hipEventDisableTiming));
    TRY(hipEventCreateWithFlags(&g.ev_host, hipEventDisableTiming));
    if (rank == 0) {
      TRY(hipMalloc((void**)&g.recv, msg * world));
      TRY(hipHostMalloc((void**)&g.h_cnt, sizeof(epa_row) * world));
      TRY(hipHostMalloc((void**)&g.h_rows, sizeof(epa_row) * (size_t)rows_cap * world));
      g.counts.assign(world, 0);
      g.ptrs.assign(world, nullptr);
    }
  }
#undef TRY
  *out = c;
  return EPA_OK;
}

static int grow_carry(epa_comm* c, int which, uint64_t rows) {
  if (c->carry_cap[which] >= rows) return EPA_OK;
  epa_ctx* ctx = c->ctx;
  // the other half may still be read by queued work on cs; this half is only written by work queued
  // below, so replacing it needs no synchronisation beyond hipFree's own
  if (c->carry[which]) EPA_HIP(ctx, hipFree(c->carry[which]));
  c->carry[which] = nullptr; c->carry_cap[which] = 0;
  const uint64_t want = rows + rows / 2 + 1024;
  EPA_HIP(ctx, hipMalloc((void**)&c->carry[which], sizeof(epa_row) * want));
  c->carry_cap[which] = want;
  return EPA_OK;
}

// d_rows != nullptr: the rows are already packed (epa_dev_place_all_rows); else they are made from (pairs, results)
static int gather_post(epa_ctx* ctx, epa_comm* c, const epa_pair* d_pairs, const epa_result* d_results, const epa_row* d_rows,
                       uint64_t n, uint32_t seq_offset, uint64_t* ticket);

extern "C" int epa_dev_gather_results(epa_ctx* ctx, epa_comm* c, const epa_pair* d_pairs, const epa_result* d_results,
                                      uint64_t n, uint32_t seq_offset, uint64_t* ticket) {
  if (!ctx || !c || c->ctx != ctx) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_results: communicator of another context");
  if (n && (!epa_is_device_ptr(d_pairs) || !epa_is_device_ptr(d_results)))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_results: pairs / results must be device memory (EPA_CHUNK_NO_D2H)");
  return gather_post(ctx, c, d_pairs, d_results, nullptr, n, seq_offset, ticket);
}

extern "C" int epa_dev_gather_rows(epa_ctx* ctx, epa_comm* c, const epa_row* d_rows, uint64_t n, uint64_t* ticket) {
  if (!ctx || !c || c->ctx != ctx) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_rows: communicator of another context");
  if (n && !epa_is_device_ptr(d_rows)) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_rows: the rows must be device memory");
  return gather_post(ctx, c, nullptr, nullptr, n ? d_rows : nullptr, n, 0, ticket);
}

static int gather_post(epa_ctx* ctx, epa_comm* c, const epa_pair* d_pairs, const epa_result* d_results, const epa_row* d_rows,
                       uint64_t n, uint32_t seq_offset, uint64_t* ticket) {
  Rccl* R = rccl();
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t tk = c->next_ticket;
  epa_comm::GSlot& g = c->gs[tk % c->depth];
  const uint32_t cap = c->cap;
  hipStream_t cs = c->cs;
  // behind the producer (the chunk's kernels on the context's stream)
  EPA_HIP(ctx, hipEventRecord(c->ev_src, ctx->stream));
  EPA_HIP(ctx, hipStreamWaitEvent(cs, c->ev_src, 0));
  // rank 0 packs its own rows straight into its receive block
  epa_row* dst = (c->rank == 0 && !c->self_send) ? g.recv : g.send;
  const uint64_t total = c->carry_n + n;
  const uint64_t m = total < cap ? total : cap;
  const uint64_t from_carry = c->carry_n < m ? c->carry_n : m;
  const uint64_t from_new = m - from_carry;
  const uint64_t left_carry = c->carry_n - from_carry, left_new = n - from_new;
  epa_row* old = c->carry[c->carry_cur];
  if (from_carry)
    EPA_HIP(ctx, hipMemcpyAsync(dst, old, sizeof(epa_row) * from_carry, hipMemcpyDeviceToDevice, cs));
  if (from_new) {
    if (d_rows) EPA_HIP(ctx, hipMemcpyAsync(dst + from_carry, d_rows, sizeof(epa_row) * from_new, hipMemcpyDeviceToDevice, cs));
    else
      hipLaunchKernelGGL(k_pack_rows, dim3((uint32_t)((from_new + 255) / 256)), dim3(256), 0, cs, d_pairs, d_results,
                         (uint64_t)0, from_new, seq_offset, dst + from_carry);
  }
  if (left_carry + left_new) {
    const int nx = c->carry_cur ^ 1;
    int rc = grow_carry(c, nx, left_carry + left_new);
    if (rc) return rc;
    if (left_carry)
      EPA_HIP(ctx, hipMemcpyAsync(c->carry[nx], old + from_carry, sizeof(epa_row) * left_carry, hipMemcpyDeviceToDevice, cs));
    if (left_new) {
      if (d_rows)
        EPA_HIP(ctx, hipMemcpyAsync(c->carry[nx] + left_carry, d_rows + from_new, sizeof(epa_row) * left_new, hipMemcpyDeviceToDevice, cs));
      else
        hipLaunchKernelGGL(k_pack_rows, dim3((uint32_t)((left_new + 255) / 256)), dim3(256), 0, cs, d_pairs, d_results,
                           from_new, left_new, seq_offset, c->carry[nx] + left_carry);
    }
    c->carry_cur = nx;
    c->carried_rows += left_new;
  }
  c->carry_n = left_carry + left_new;
  hipLaunchKernelGGL(k_sentinel, dim3(1), dim3(1), 0, cs, dst + cap, (uint32_t)m, (double)c->carry_n);
  EPA_HIP(ctx, hipGetLastError());
  // the source buffers are free once the rows are packed: whatever the caller queues next on the context's
  // stream (the next chunk into the same buffers) is ordered behind the packing, not behind the transfer
  EPA_HIP(ctx, hipEventRecord(c->ev_packed, cs));
  EPA_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_packed, 0));
  // the exchange: point-to-point into the root (7 xGMI links in parallel at 8 ranks), one group
  const size_t msg = sizeof(epa_row) * ((size_t)cap + 1);
  if (c->world > 1 || c->self_send) {
    EPA_NCCL(ctx, R->GroupStart());
    // a failure inside the group still closes it: an open group would swallow every later call of the thread
    ncclResult_t gr = ncclSuccess;
    const char* what = "";
    if (c->rank == 0) {
      for (int r = c->self_send ? 0 : 1; r < c->world && gr == ncclSuccess; ++r) {
        gr = R->Recv((char*)g.recv + msg * r, msg, ncclChar, r, c->comm, cs);
        what = "ncclRecv";
      }
      if (c->self_send && gr == ncclSuccess) { gr = R->Send(g.send, msg, ncclChar, 0, c->comm, cs); what = "ncclSend"; }
    } else {
      gr = R->Send(g.send, msg, ncclChar, 0, c->comm, cs);
      what = "ncclSend";
    }
    const ncclResult_t ge = R->GroupEnd();
    if (gr != ncclSuccess) return epa_fail(ctx, EPA_ERR_HIP, std::string(what) + ": " + R->GetErrorString(gr));
    if (ge != ncclSuccess) return epa_fail(ctx, EPA_ERR_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(ge));
  }
  if (c->rank == 0)   // the sentinel rows (valid counts) to pinned memory: what collect() reads first
    EPA_HIP(ctx, hipMemcpy2DAsync(g.h_cnt, sizeof(epa_row), g.recv + cap, msg, sizeof(epa_row), c->world,
                                  hipMemcpyDeviceToHost, cs));
  EPA_HIP(ctx, hipEventRecord(g.ev_gather, cs));
  g.ticket = tk;
  c->next_ticket = tk + 1;
  if (ticket) *ticket = tk;
  return EPA_OK;
}

extern "C" int epa_dev_gather_slot(epa_ctx* ctx, epa_comm* c, int slot, uint32_t seq_offset, uint64_t* ticket) {
  if (!ctx || slot < 0 || slot >= epa_ctx::N_SLOTS) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_slot: slot out of range");
  ChunkSlot* s = &ctx->slots[slot];
  if (s->state != 2 || !(s->l_flags & EPA_CHUNK_NO_D2H))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_slot: the slot needs a finished launch with EPA_CHUNK_NO_D2H");
  if (s->leader >= 0)   // its rows are a range of the group's regrouped buffers, known to the host only at finish
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "gather_slot: the slot is part of a group launch: epa_dev_chunk_finish it and post its rows with epa_dev_gather_results");
  return epa_dev_gather_results(ctx, c, s->l_pairs, s->l_res, s->n, seq_offset, ticket);
}

extern "C" int epa_comm_collect(epa_comm* c, uint64_t ticket, const epa_row** rows, uint32_t* counts, uint64_t* pending) {
  if (!c) return EPA_ERR_INVALID_ARG;
  epa_ctx* ctx = c->ctx;
  if (c->rank != 0) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_collect: only rank 0 receives rows");
  epa_comm::GSlot& g = c->gs[ticket % c->depth];
  if (g.ticket != ticket) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_collect: the gather's slot has been reused (collect within `depth` posts)");
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  bool late = false;
  EPA_HIP(ctx, wait_event(g.ev_gather, &late, c->lim()));
  if (late) return epa_fail(ctx, EPA_ERR_HIP, "comm_collect: gather " + std::to_string(ticket) + " did not complete within the communicator's timeout (a peer failed?): epa_comm_abort and exit");
  const size_t msg_rows = (size_t)c->cap + 1;
  for (int r = 0; r < c->world; ++r) {
    const epa_row& s = g.h_cnt[r];
    if (s.seq_id != 0xE9A0C0DEu || s.branch_id > c->cap)
      return epa_fail(ctx, EPA_ERR_HIP, "comm_collect: rank " + std::to_string(r) + " sent no sentinel row");
    g.counts[r] = s.branch_id;
    g.ptrs[r] = g.h_rows + (size_t)r * c->cap;
    // rows == NULL: the caller only wants the counts; the rows stay in HBM (epa_comm_device_rows)
    if (s.branch_id && rows)
      EPA_HIP(ctx, hipMemcpyAsync(g.h_rows + (size_t)r * c->cap, g.recv + msg_rows * r, sizeof(epa_row) * s.branch_id,
                                  hipMemcpyDeviceToHost, c->cs));
  }
  if (rows) {
    EPA_HIP(ctx, hipEventRecord(g.ev_host, c->cs));
    EPA_HIP(ctx, wait_event(g.ev_host, &late, c->lim()));
    if (late) return epa_fail(ctx, EPA_ERR_HIP, "comm_collect: the copy of the rows to the host did not complete");
  }
  for (int r = 0; r < c->world; ++r) {
    if (rows) rows[r] = g.ptrs[r];
    if (counts) counts[r] = g.counts[r];
    if (pending) pending[r] = (uint64_t)g.h_cnt[r].pendant_length;
  }
  return EPA_OK;
}

extern "C" int epa_comm_flush(epa_ctx* ctx, epa_comm* c, uint64_t* first_extra_ticket, uint32_t* n_extra) {
  if (!ctx || !c || c->ctx != ctx) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_flush: communicator of another context");
  Rccl* R = rccl();
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  c->h_pend[0] = c->carry_n;
  EPA_HIP(ctx, hipMemcpyAsync(c->d_pend, c->h_pend, 8, hipMemcpyHostToDevice, c->cs));
  if (c->world > 1)
    EPA_NCCL(ctx, R->AllReduce(c->d_pend, c->d_pend + 1, 1, ncclUint64, ncclMax, c->comm, c->cs));
  else
    EPA_HIP(ctx, hipMemcpyAsync(c->d_pend + 1, c->d_pend, 8, hipMemcpyDeviceToDevice, c->cs));
  EPA_HIP(ctx, hipMemcpyAsync(c->h_pend + 1, c->d_pend + 1, 8, hipMemcpyDeviceToHost, c->cs));
  EPA_HIP(ctx, hipEventRecord(c->ev_src, c->cs));
  bool late = false;
  EPA_HIP(ctx, wait_event(c->ev_src, &late, c->lim()));
  if (late) return epa_fail(ctx, EPA_ERR_HIP, "comm_flush: the all-reduce did not complete within the communicator's timeout (a peer failed?): epa_comm_abort and exit");
  const uint64_t pend = c->h_pend[1];
  // every rank computed the same maximum (the all-reduce), so every rank posts the same number of rounds;
  // at most `depth` per call: rank 0 collects them before their slots are posted again
  uint64_t extra = (pend + c->cap - 1) / c->cap;
  if (extra > (uint64_t)c->depth) extra = (uint64_t)c->depth;
  if (first_extra_ticket) *first_extra_ticket = c->next_ticket;
  if (n_extra) *n_extra = (uint32_t)extra;
  for (uint64_t i = 0; i < extra; ++i) {
    int rc = gather_post(ctx, c, nullptr, nullptr, nullptr, 0, 0, nullptr);
    if (rc) return rc;
  }
  return EPA_OK;
}

extern "C" uint64_t epa_comm_carried_rows(const epa_comm* c) { return c ? c->carried_rows : 0; }

extern "C" const epa_row* epa_comm_device_rows(const epa_comm* c, uint64_t ticket, int rank) {
  if (!c || c->rank != 0 || rank < 0 || rank >= c->world) return nullptr;
  const epa_comm::GSlot& g = c->gs[ticket % c->depth];
  if (g.ticket != ticket) return nullptr;
  return g.recv + ((size_t)c->cap + 1) * rank;
}

// One round trip through everything a real gather uses, before any work depends on it: every rank posts a gather of
// ONE row that names it (rank, PCI id of its device), rank 0 collects the world's rows, then all ranks meet in an
// all-reduce (a sender's ncclSend may complete eagerly: the all-reduce is what tells every rank that every other rank
// got this far).  Every wait is bounded by timeout_s.  On success the communicator is as created (tickets start at 0
// again); on failure the caller epa_comm_abort()s it -- all ranks fail within about timeout_s of each other, since
// none can pass the all-reduce alone.
__global__ void k_probe_row(epa_row* row, uint32_t rank, double dev_id) {
  epa_row o;
  o.branch_id = rank;
  o.seq_id = 0x9B0BE000u;
  o.lnl = dev_id;
  o.pendant_length = 0.0;
  o.distal_length = 0.0;
  *row = o;
}

extern "C" int epa_comm_probe(epa_ctx* ctx, epa_comm* c, double timeout_s, uint64_t* device_ids) {
  if (!ctx || !c || c->ctx != ctx) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_probe: communicator of another context");
  if (c->next_ticket || c->carry_n) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "comm_probe: only on a communicator that has not gathered yet");
  Rccl* R = rccl();
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  int dom = 0, bus = 0, devn = 0;
  EPA_HIP(ctx, hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, ctx->device));
  EPA_HIP(ctx, hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, ctx->device));
  EPA_HIP(ctx, hipDeviceGetAttribute(&devn, hipDeviceAttributePciDeviceId, ctx->device));
  const uint64_t my_id = ((uint64_t)(uint32_t)dom << 16) | ((uint64_t)(bus & 0xff) << 8) | (uint64_t)(devn & 0xff);
  const double keep = c->timeout;
  if (timeout_s > 0) c->timeout = timeout_s;
  struct Restore { epa_comm* c; double t; ~Restore() { c->timeout = t; } } restore{c, keep};
  hipLaunchKernelGGL(k_probe_row, dim3(1), dim3(1), 0, c->cs, c->d_probe, (uint32_t)c->rank, (double)my_id);
  EPA_HIP(ctx, hipGetLastError());
  uint64_t tk = 0;   // (the row is made on the communicator's stream, where gather_post queues its copy)
  int rc = gather_post(ctx, c, nullptr, nullptr, c->d_probe, 1, 0, &tk);
  if (rc) return rc;
  if (c->rank == 0) {
    std::vector<const epa_row*> rows(c->world);
    std::vector<uint32_t> counts(c->world);
    rc = epa_comm_collect(c, tk, rows.data(), counts.data(), nullptr);
    if (rc) return rc;
    for (int r = 0; r < c->world; ++r) {
      if (counts[r] != 1 || rows[r][0].branch_id != (uint32_t)r || rows[r][0].seq_id != 0x9B0BE000u)
        return epa_fail(ctx, EPA_ERR_HIP, "comm_probe: rank " + std::to_string(r) + " did not deliver its probe row");
      if (device_ids) device_ids[r] = (uint64_t)rows[r][0].lnl;
    }
  }
  // all ranks: sum of ones == world
  c->h_pend[0] = 1;
  EPA_HIP(ctx, hipMemcpyAsync(c->d_pend, c->h_pend, 8, hipMemcpyHostToDevice, c->cs));
  if (c->world > 1)
    EPA_NCCL(ctx, R->AllReduce(c->d_pend, c->d_pend + 1, 1, ncclUint64, ncclSum, c->comm, c->cs));
  else
    EPA_HIP(ctx, hipMemcpyAsync(c->d_pend + 1, c->d_pend, 8, hipMemcpyDeviceToDevice, c->cs));
  EPA_HIP(ctx, hipMemcpyAsync(c->h_pend + 1, c->d_pend + 1, 8, hipMemcpyDeviceToHost, c->cs));
  EPA_HIP(ctx, hipEventRecord(c->ev_src, c->cs));
  bool late = false;
  EPA_HIP(ctx, wait_event(c->ev_src, &late, c->lim()));
  if (late) return epa_fail(ctx, EPA_ERR_HIP, "comm_probe: the all-reduce did not complete within " + std::to_string((int)c->lim()) + " s");
  if (c->h_pend[1] != (unsigned long long)c->world)
    return epa_fail(ctx, EPA_ERR_HIP, "comm_probe: the all-reduce saw " + std::to_string(c->h_pend[1]) + " ranks, not " + std::to_string(c->world));
  // as created
  c->next_ticket = 0;
  for (auto& g : c->gs) g.ticket = ~0ull;
  return EPA_OK;
}

extern "C" void epa_comm_abort(epa_comm* c) {
  // the failing rank's way out (the reference's MPI build would MPI_Abort): tears the communicator down
  // without waiting for the peers, which then see errors / timeouts instead of waiting for ever
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  if (c->comm) {
    Rccl* R = rccl();
    if (R->CommAbort) (void)R->CommAbort(c->comm);
    c->comm = nullptr;
  }
  // an aborted communicator's kernels leave the stream; should they not (a transport that ignores the abort), the
  // buffers they may still touch are LEAKED rather than freed under them, and nobody waits for ever
  bool idle = !c->cs;
  const auto t0 = std::chrono::steady_clock::now();
  while (!idle && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 10.0) {
    idle = hipStreamQuery(c->cs) != hipErrorNotReady;
    if (!idle) std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  if (idle) epa_comm_destroy(c);
}
