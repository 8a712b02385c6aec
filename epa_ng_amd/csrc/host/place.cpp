// The chunk loop and its two hot loops, host side: the reference's place() / place_thorough() /
// simple_mpi() (src/core/place.cpp) with the per-branch Tiny_Tree evaluator replaced by calls
// into the C-ABI of libepa_dev.so.  Heuristics, LWR and filters are the reference's host-side
// set operations (src/core/heuristics.hpp, src/set_manipulators.cpp) on flat arrays.
#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <fstream>
#include <mutex>
#include <thread>
#include <limits>
#include <numeric>

#include <omp.h>
#include <sched.h>

#include "epa_host.hpp"

namespace epa {

namespace { int g_thread_limit = 0; }   // -T/--threads of the command line; 0 = no user limit
void set_host_thread_limit(int n) {
  g_thread_limit = n > 0 ? n : 0;
  epa_encode_set_threads((unsigned)g_thread_limit);   // the query encoder's std::threads honour -T too
}

namespace { thread_local int tl_team = 0; }   // set_thread_team: OpenMP team of the calling host thread's stages; 0 = all
void set_thread_team(int n) {
  tl_team = n > 0 ? n : 0;
  configure_host_threads();
}

int configure_host_threads() {
  static const int n = [] {
    int v = omp_get_max_threads();
    if (g_thread_limit > 0) v = std::min(v, g_thread_limit);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) v = std::min(v, std::max(1, CPU_COUNT(&set)));
    std::ifstream f("/sys/fs/cgroup/cpu.max");  // cgroup v2: "<quota> <period>" or "max <period>"
    std::string quota;
    double period = 0;
    if (f >> quota >> period && quota != "max" && period > 0)
      v = std::min(v, std::max(1, (int)std::ceil(std::stod(quota) / period)));
    epa_encode_set_threads((unsigned)v);   // affinity mask / cgroup quota apply to the query encoder as well
    return v;
  }();
  const int team = tl_team > 0 ? std::min(tl_team, n) : n;
  omp_set_num_threads(team);  // per calling thread (the device workers are separate host threads)
  return team;
}

namespace {
const double kDefaultBranchLength = -std::log(0.9);

[[noreturn]] void throw_dev(epa_ctx* ctx, int rc) {
  throw std::runtime_error{std::string(epa_dev_last_error(ctx)) + " (epa_dev status " +
                           std::to_string(rc) + ")"};
}
}  // namespace

Device_Evaluator::Device_Evaluator(const Tree& tree, const Options& options, int device) {
  const bool rs = options.rate_scalers(tree.nums().tip_nodes);
  const uint32_t flags = (options.sliding_blo ? EPA_FLAG_SLIDING_BLO : EPA_FLAG_RAXML_BLO) |
                         (rs ? EPA_FLAG_RATE_SCALERS : 0u);
  if (rs && !options.device_precompute)
    throw std::runtime_error{"per-rate scalers need the device-side reference precompute (the host CLV "
                             "path keeps per-site scalers): drop --host-precompute or pass --rate-scalers off"};
  int rc;
  if (options.device_precompute) {
    // tree + tip sequences go to the device, all directional CLVs are computed there
    epa_tree_desc d;
    Tree::Tree_Desc_Storage store;
    tree.fill_tree_desc(d, store);
    d.ref.flags = flags;
    d.ref.aa_x_as_n = options.aa_x_as_n ? 1 : 0;
    rc = epa_dev_create_from_tree(&d, device, &ctx_);
  } else {
    epa_ref_desc d;
    std::vector<const double*> pc, dc;
    std::vector<const uint32_t*> ps, ds;
    std::vector<const uint8_t*> dt;
    std::vector<double> bl;
    tree.fill_desc(d, pc, ps, dc, dt, ds, bl);
    d.flags = flags;
    d.aa_x_as_n = options.aa_x_as_n ? 1 : 0;
    rc = epa_dev_create(&d, device, &ctx_);
  }
  if (rc != EPA_OK) throw_dev(nullptr, rc);
}

double Device_Evaluator::ref_tree_logl(size_t branch) const {
  double v = 0.0;
  const int rc = epa_dev_tree_logl(ctx_, (uint32_t)branch, &v);
  if (rc != EPA_OK) throw_dev(ctx_, rc);
  return v;
}

Device_Evaluator::~Device_Evaluator() { epa_dev_destroy(ctx_); }

Encoded_Chunk encode_chunk(const MSA& chunk, const Tree& tree, const Options& options) {
  const size_t Q = chunk.size(), W = tree.num_sites();
  std::vector<const char*> rows(Q);
  for (size_t q = 0; q < Q; ++q) {
    if (chunk[q].sequence().size() != W)  // Tiny_Tree.cpp:145-147
      throw std::runtime_error{"Query sequence length not same as reference alignment!"};
    rows[q] = chunk[q].sequence().c_str();
  }
  return encode_rows(rows, chunk, tree, options);
}

Encoded_Chunk encode_rows(const std::vector<const char*>& rows, const MSA& chunk, const Tree& tree,
                          const Options& options) {
  Encoded_Chunk e;
  const size_t Q = rows.size(), W = tree.num_sites();
  e.win_begin.resize(Q);
  e.win_span.resize(Q);
  uint32_t bad = 0;
  auto check = [&](int rc) {
    if (rc == EPA_ERR_QUERY_ALL_GAP)  // Tiny_Tree.cpp:153-156
      throw std::runtime_error{std::string() + "Sequence with header '" + chunk[bad].header() +
                               "' does not appear to have any non-gap sites!"};
    if (rc == EPA_ERR_INVALID_CHAR)  // Lookup_Store.hpp:100-108
      throw std::runtime_error{"char is invalid! (sequence '" + chunk[bad].header() + "')"};
    if (rc != EPA_OK) throw std::runtime_error{"query encoding failed"};
  };
  // compact wire format: pass 1 finds the windows, pass 2 writes only the window columns
  check(epa_encode_queries_compact((uint32_t)tree.model().num_states(), (uint32_t)W, (uint32_t)Q,
                                   rows.data(), options.premasking, options.aa_x_as_n, 0, nullptr,
                                   e.win_begin.data(), e.win_span.data(), &bad));
  uint32_t mx = 1;
  for (uint32_t s : e.win_span) mx = std::max(mx, s);
  e.stride = (mx + 15) / 16 * 16;
  e.codes.resize(Q * (size_t)e.stride);
  check(epa_encode_queries_compact((uint32_t)tree.model().num_states(), (uint32_t)W, (uint32_t)Q,
                                   rows.data(), options.premasking, options.aa_x_as_n, e.stride,
                                   e.codes.data(), e.win_begin.data(), e.win_span.data(), &bad));
  if (tree.model().num_states() == 4) {
    // the reference's 4-bit packing (src/io/encoding.hpp) as the H2D wire format: half the bytes
    std::vector<uint8_t> packed(Q * (size_t)((e.stride + 1) / 2));
    if (epa_pack_codes_4bit(e.codes.data(), (uint32_t)Q, e.stride, packed.data()) != EPA_OK)
      throw std::runtime_error{"query packing failed"};
    e.codes.swap(packed);
    e.bits = 4;
  }
  return e;
}

void place(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
           std::vector<double>& lnl, const Options&) {
  const size_t Q = chunk.size(), B = tree.num_branches();
  lnl.resize(Q * B);
  epa_dev_set_query_layout(dev.ctx(), enc.stride);
  epa_dev_set_query_packing(dev.ctx(), enc.bits);
  const int rc = epa_dev_preplace(dev.ctx(), enc.codes.data(), enc.win_begin.data(),
                                  enc.win_span.data(), (uint32_t)Q, lnl.data());
  if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
}

// ---- candidate selection ------------------------------------------------------------------
static void row_lwr(const double* row, size_t B, std::vector<double>& lwr) {
  // compute_and_set_lwr (src/set_manipulators.cpp:43-69)
  const double mx = *std::max_element(row, row + B);
  lwr.resize(B);
  double total = 0.0;
  for (size_t i = 0; i < B; ++i) { lwr[i] = std::exp(row[i] - mx); total += lwr[i]; }
  for (size_t i = 0; i < B; ++i) lwr[i] /= total;
}

Work apply_heuristic(const std::vector<double>& lnl, size_t Q, size_t B, const Options& options) {
  configure_host_threads();
  std::vector<std::vector<uint32_t>> keep(Q);
#pragma omp parallel
  {
    std::vector<double> lwr;
    std::vector<uint32_t> order(B);
#pragma omp for schedule(dynamic)
    for (long q = 0; q < (long)Q; ++q) {
      const double* row = &lnl[(size_t)q * B];
      std::iota(order.begin(), order.end(), 0u);
      size_t n_keep = 0;
      if (options.baseball) {
        // baseball_heuristic (src/core/heuristics.hpp:70-117); quirk D7: clamp at B
        const double strike_box = 3, best = *std::max_element(row, row + B);
        const size_t max_strikes = 6, max_pitches = 40;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return row[a] > row[b]; });
        size_t hits = 0;
        while (hits < B && !(row[order[hits]] < best - strike_box)) ++hits;
        // size_t arithmetic of heuristics.hpp:107: the difference wraps when hits > max_pitches
        const size_t to_add = hits > max_pitches ? max_strikes : std::min(max_pitches - hits, max_strikes);
        n_keep = std::min(B, hits + to_add);
      } else {
        row_lwr(row, B, lwr);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lwr[a] > lwr[b]; });
        if (options.prescoring_by_percentage) {
          // until_top_percent (src/set_manipulators.cpp:82-88)
          n_keep = std::min(B, (size_t)std::ceil(options.prescoring_threshold * (double)B));
        } else {
          // until_accumulated_reached(pq, thresh, 1, inf) (:90-114)
          double sum = 0.0;
          while (n_keep < B && sum < options.prescoring_threshold) sum += lwr[order[n_keep++]];
          if (n_keep < 1) n_keep = 1;
        }
      }
      keep[q].assign(order.begin(), order.begin() + n_keep);
    }
  }
  // Work is a map<branch, vector<seq>> in the reference: iteration order is branch-major
  std::vector<size_t> per_branch(B + 1, 0);
  for (const auto& k : keep)
    for (uint32_t b : k) ++per_branch[b + 1];
  for (size_t b = 0; b < B; ++b) per_branch[b + 1] += per_branch[b];
  Work work(per_branch[B]);
  std::vector<size_t> cur(per_branch.begin(), per_branch.end() - 1);
  for (size_t q = 0; q < Q; ++q)
    for (uint32_t b : keep[q]) work[cur[b]++] = Work_Pair{b, q};
  return work;
}

// one PQuery per query, in query order (quirk D8: the reference's order is thread-dependent)
static void build_sample(const Work& to_place, const epa_result* res, const MSA& chunk,
                         Sample& sample, size_t seq_id_offset) {
  // pairs arrive branch-major; a counting sort by query keeps that order inside every pquery, then
  // the pqueries (one small vector each) are built in parallel
  const size_t n = to_place.size(), Q = chunk.size();
  std::vector<uint32_t> first(Q + 1, 0);
  for (size_t i = 0; i < n; ++i) ++first[to_place[i].sequence_id + 1];
  std::vector<long> slot(Q, -1);
  size_t used = 0;
  for (size_t q = 0; q < Q; ++q) {
    if (first[q + 1]) slot[q] = (long)used++;
    first[q + 1] += first[q];
  }
  std::vector<uint32_t> idx(n), cur(first.begin(), first.end() - 1);
  for (size_t i = 0; i < n; ++i) idx[cur[to_place[i].sequence_id]++] = (uint32_t)i;
  sample.clear();
  sample.resize(used);
#pragma omp parallel for schedule(static)
  for (long q = 0; q < (long)Q; ++q) {
    if (slot[q] < 0) continue;
    PQuery pq(seq_id_offset + (size_t)q, chunk[q].header());
    pq.placements().reserve(first[q + 1] - first[q]);
    for (uint32_t k = first[q]; k < first[q + 1]; ++k) {
      const uint32_t i = idx[k];
      pq.emplace_back(to_place[i].branch_id, res[i].lnl, res[i].pendant_length, res[i].distal_length);
    }
    sample[slot[q]] = std::move(pq);
  }
}

void place_thorough(const Work& to_place, const MSA& chunk, const Encoded_Chunk& enc,
                    const Tree& tree, Device_Evaluator& dev, Sample& sample, const Options&,
                    size_t seq_id_offset) {
  const size_t n = to_place.size(), Q = chunk.size();
  std::vector<epa_pair> pairs(n);
  for (size_t i = 0; i < n; ++i)
    pairs[i] = epa_pair{(uint32_t)to_place[i].branch_id, (uint32_t)to_place[i].sequence_id};
  std::vector<epa_result> res(n);
  epa_dev_set_query_layout(dev.ctx(), enc.stride);
  epa_dev_set_query_packing(dev.ctx(), enc.bits);
  const int rc = epa_dev_thorough(dev.ctx(), pairs.data(), n, enc.codes.data(), enc.win_begin.data(),
                                  enc.win_span.data(), (uint32_t)Q, res.data(), nullptr);
  if (rc == EPA_ERR_NEG_INF)  // Tiny_Tree.cpp:209-212
    throw std::runtime_error{epa_dev_last_error(dev.ctx())};
  if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
  build_sample(to_place, res.data(), chunk, sample, seq_id_offset);
  (void)tree;
}

void place_chunk(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
                 Work& work, Sample& sample, const Options& options, size_t seq_id_offset) {
  const size_t Q = chunk.size();
  // the dynamic heuristic keeps a handful of branches per read: start with room for 8 per read,
  // grow on overflow (the buffers live in the evaluator and are reused by the next chunk)
  uint64_t cap = std::max<uint64_t>((uint64_t)Q * 8, dev.pair_buffer().size()), n = 0;
  // apply_heuristic's three rules all run on the device (src/core/heuristics.hpp:119-127)
  const size_t nb = tree.num_branches();
  const int mode = options.baseball ? EPA_HEUR_BASEBALL
                                    : options.prescoring_by_percentage ? EPA_HEUR_FIXED : EPA_HEUR_DYNAMIC;
  if (epa_dev_set_heuristic(dev.ctx(), mode, mode == EPA_HEUR_FIXED ? options.prescoring_threshold : 0.0) != EPA_OK)
    throw std::runtime_error{epa_dev_last_error(dev.ctx())};
  if (mode == EPA_HEUR_FIXED)
    cap = std::max<uint64_t>(cap, (uint64_t)Q * std::min<size_t>(nb, (size_t)std::ceil(options.prescoring_threshold * (double)nb)));
  else if (mode == EPA_HEUR_BASEBALL)
    cap = std::max<uint64_t>(cap, (uint64_t)Q * std::min<size_t>(nb, 46));
  std::vector<epa_pair>& pairs = dev.pair_buffer();
  std::vector<epa_result>& res = dev.result_buffer();
  uint32_t max_span = 0;
  for (uint32_t s : enc.win_span) max_span = std::max(max_span, s);
  for (;;) {
    if (pairs.size() < cap) pairs.resize(cap);
    if (res.size() < cap) res.resize(cap);
    epa_dev_set_query_layout(dev.ctx(), enc.stride);
  epa_dev_set_query_packing(dev.ctx(), enc.bits);
    const int rc = epa_dev_place_chunk(dev.ctx(), enc.codes.data(), enc.win_begin.data(),
                                       enc.win_span.data(), (uint32_t)Q, max_span,
                                       options.prescoring_threshold, pairs.data(), res.data(), cap, &n,
                                       nullptr);
    if (rc == EPA_ERR_PAIR_OVERFLOW && cap < (uint64_t)Q * tree.num_branches()) {
      cap = std::min<uint64_t>(cap * 8, (uint64_t)Q * tree.num_branches());  // candidate overflow
      continue;
    }
    if (rc == EPA_ERR_NEG_INF) throw std::runtime_error{epa_dev_last_error(dev.ctx())};
    if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
    break;
  }
  work.resize(n);
  for (size_t i = 0; i < n; ++i) work[i] = Work_Pair{pairs[i].branch_id, pairs[i].seq_id};
  build_sample(work, res.data(), chunk, sample, seq_id_offset);
}

// ---- the fused chunk body as a two-slot pipeline (epa_dev_chunk_stage / _launch / _finish): the
// H2D of this chunk and the D2H + host post-processing of the previous one overlap the kernels
// (the device-side counterpart of the read-ahead in src/seq/MSA_Stream.cpp:79-82)
void chunk_launch(const Encoded_Chunk& enc, size_t Q, const Tree& tree, Device_Evaluator& dev, int slot,
                  const Options& options) {
  const size_t nb = tree.num_branches();
  const int mode = options.baseball ? EPA_HEUR_BASEBALL
                                    : options.prescoring_by_percentage ? EPA_HEUR_FIXED : EPA_HEUR_DYNAMIC;
  if (epa_dev_set_heuristic(dev.ctx(), mode, mode == EPA_HEUR_FIXED ? options.prescoring_threshold : 0.0) != EPA_OK)
    throw std::runtime_error{epa_dev_last_error(dev.ctx())};
  uint64_t cap = std::max<uint64_t>((uint64_t)Q * 8, dev.pair_capacity());
  if (mode == EPA_HEUR_FIXED)
    cap = std::max<uint64_t>(cap, (uint64_t)Q * std::min<size_t>(nb, (size_t)std::ceil(options.prescoring_threshold * (double)nb)));
  else if (mode == EPA_HEUR_BASEBALL)
    cap = std::max<uint64_t>(cap, (uint64_t)Q * std::min<size_t>(nb, 46));
  uint32_t max_span = 0;
  for (uint32_t s : enc.win_span) max_span = std::max(max_span, s);
  epa_dev_set_query_layout(dev.ctx(), enc.stride);
  epa_dev_set_query_packing(dev.ctx(), enc.bits);
  int rc = epa_dev_chunk_stage(dev.ctx(), slot, enc.codes.data(), enc.win_begin.data(), enc.win_span.data(),
                               (uint32_t)Q);
  if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
  for (;;) {
    // host arrays in, host rows out, read after finish: no stream-level ordering against the context's stream needed
    rc = epa_dev_chunk_launch(dev.ctx(), slot, max_span, options.prescoring_threshold, nullptr, nullptr, cap, EPA_CHUNK_HOST_ORDERED);
    if (rc == EPA_ERR_PAIR_OVERFLOW && cap < (uint64_t)Q * nb) {
      cap = std::min<uint64_t>(cap * 8, (uint64_t)Q * nb);  // candidate overflow: the slot stays staged
      continue;
    }
    if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
    break;
  }
  dev.pair_capacity() = cap;
}

size_t place_all(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
                 Sample& sample, const Options& options, size_t seq_id_offset) {
  const size_t Q = chunk.size(), fm = options.filter_max;
  std::vector<epa_pair> pairs(Q * fm);
  std::vector<epa_result> res(Q * fm);
  std::vector<double> lwr(Q * fm);
  std::vector<uint32_t> counts(Q);
  uint32_t max_span = 0;
  for (uint32_t s : enc.win_span) max_span = std::max(max_span, s);
  epa_dev_set_query_layout(dev.ctx(), enc.stride);
  epa_dev_set_query_packing(dev.ctx(), enc.bits);
  const int rc = epa_dev_place_all(dev.ctx(), enc.codes.data(), enc.win_begin.data(), enc.win_span.data(),
                                   (uint32_t)Q, max_span, options.support_threshold,
                                   options.acc_threshold ? 1 : 0, options.filter_min, options.filter_max,
                                   pairs.data(), res.data(), lwr.data(), counts.data(), nullptr);
  if (rc == EPA_ERR_NEG_INF) throw std::runtime_error{epa_dev_last_error(dev.ctx())};
  if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
  sample.clear();
  sample.reserve(Q);
  for (size_t q = 0; q < Q; ++q) {
    sample.emplace_back(seq_id_offset + q, chunk[q].header());
    auto& pq = sample.back();
    for (uint32_t k = 0; k < counts[q]; ++k) {
      const size_t o = q * fm + k;
      pq.emplace_back(pairs[o].branch_id, res[o].lnl, res[o].pendant_length, res[o].distal_length);
      pq[pq.size() - 1].lwr(lwr[o]);
    }
  }
  return Q * tree.num_branches();
}

void compute_and_set_lwr(Sample& sample) {
  configure_host_threads();
#pragma omp parallel for schedule(static)
  for (long j = 0; j < (long)sample.size(); ++j) {
    auto& pq = sample[j];
    double mx = -std::numeric_limits<double>::infinity();
    for (auto& p : pq) mx = std::max(mx, p.likelihood());
    double total = 0.0;
    // exp(lnl - max) is parked in the lwr field, then normalised: no scratch vector per pquery
    for (size_t i = 0; i < pq.size(); ++i) { const double e = std::exp(pq[i].likelihood() - mx); pq[i].lwr(e); total += e; }
    for (size_t i = 0; i < pq.size(); ++i) pq[i].lwr(pq[i].lwr() / total);
  }
}

static void sort_by_lwr(PQuery& pq) {
  std::stable_sort(pq.begin(), pq.end(),
                   [](const Placement& a, const Placement& b) { return a.lwr() > b.lwr(); });
}

void filter(Sample& sample, const Options& options) {
  configure_host_threads();
  const double thresh = options.support_threshold;
  if (thresh < 0.0 || thresh > 1.0)
    throw std::range_error{"thresh is not a valid likelihood weight ratio (outside of [0,1])"};
  if (options.filter_min < 1) throw std::range_error{"Filter min cannot be smaller than 1!"};
  const size_t mn = options.filter_min, mx = options.filter_max;
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)sample.size(); ++i) {
    auto& pq = sample[i];
    sort_by_lwr(pq);
    size_t n_keep = 0;
    if (options.acc_threshold) {
      // discard_by_accumulated_threshold (src/set_manipulators.cpp:165-190)
      double sum = 0.0;
      // (the reference tops up to min-1, not min: distance(pq_iter, begin + min - 1), :104-107)
      while (n_keep < pq.size() && n_keep < mx && sum < thresh) sum += pq[n_keep++].lwr();
      if (n_keep + 1 < mn) n_keep = std::min(pq.size(), mn - 1);
    } else {
      // discard_by_support_threshold (:131-163)
      while (n_keep < pq.size() && pq[n_keep].lwr() > thresh) ++n_keep;
      if (n_keep < mn) n_keep = std::min(pq.size(), mn);
      if (mx && n_keep > mx) n_keep = mx;
    }
    pq.placements().resize(n_keep);
  }
}

namespace {
struct Chunk_Timing { double place = 0, thorough = 0, post = 0; size_t pairs = 0; };

// the body of the reference's chunk loop (src/core/place.cpp:219-246) for one chunk on one device
Sample process_chunk(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
                     const Options& options, size_t seq_id_offset, Chunk_Timing& tm) {
  using clk = std::chrono::steady_clock;
  const size_t B = tree.num_branches(), n = chunk.size();
  Work blo_work;
  Sample blo_sample;
  auto t0 = clk::now();
  auto t1 = t0;
  // --no-heur: pairs, thorough placement, LWR and filter all on the device (epa_dev_place_all)
  const bool all_on_device = !options.prescoring && options.filter_min >= 1 &&
                             options.filter_max >= options.filter_min && options.filter_max <= 64 &&
                             (uint64_t)n * B <= 0xffffffffull;
  // --host-heuristic: keep the Q x B table round trip and the host heuristics (cross-check)
  const bool host_heur = options.host_heuristic;
  const bool fused = options.prescoring && options.device_select && B <= 65536 && !host_heur;
  if (all_on_device) {
    tm.pairs = place_all(chunk, enc, tree, dev, blo_sample, options, seq_id_offset);
    t1 = clk::now();
  } else if (fused) {
    // the whole chunk body runs on the GPU (epa_dev_place_chunk, any of the three heuristics),
    // the Q x B table never crosses PCIe
    place_chunk(chunk, enc, tree, dev, blo_work, blo_sample, options, seq_id_offset);
    t1 = clk::now();
  } else {
    if (options.prescoring) {
      std::vector<double> lnl;
      place(chunk, enc, tree, dev, lnl, options);
      blo_work = apply_heuristic(lnl, n, B, options);
    } else {  // --no-heur: all B x Q pairs (src/core/place.cpp:189,228)
      blo_work.reserve(n * B);
      for (size_t b = 0; b < B; ++b)
        for (size_t q = 0; q < n; ++q) blo_work.push_back(Work_Pair{b, q});
    }
    t1 = clk::now();
    place_thorough(blo_work, chunk, enc, tree, dev, blo_sample, options, seq_id_offset);
  }
  auto t2 = clk::now();
  if (!all_on_device) {
    compute_and_set_lwr(blo_sample);
    filter(blo_sample, options);
    tm.pairs = blo_work.size();
  }
  tm.place = std::chrono::duration<double>(t1 - t0).count();
  tm.thorough = std::chrono::duration<double>(t2 - t1).count();
  tm.post = std::chrono::duration<double>(clk::now() - t2).count();
  return blo_sample;
}
}  // namespace

Run_Stats simple_mpi(const Tree& tree, const std::string& query_file, const std::string& outdir,
                     const Options& options, const std::string& invocation, int device) {
  return simple_mpi(tree, query_file, outdir, options, invocation, std::vector<int>{device});
}

// Chunks are read and encoded by one staging thread (the reference prefetches its next chunk the
// same way, src/core/place.cpp:190-215) and handed, in file order, to one worker per GPU: every
// (query, branch) pair is independent, the reference shards the query file across MPI ranks the
// same way (src/net/epa_mpi_util.cpp:10-30).  Results are written in chunk order, so the jplace
// does not depend on the number of devices.
Run_Stats simple_mpi(const Tree& tree, const std::string& query_file, const std::string& outdir,
                     const Options& options, const std::string& invocation, const std::vector<int>& devices) {
  return simple_mpi(tree, query_file, MSA_Info{}, outdir, options, invocation, devices);
}

Run_Stats simple_mpi(const Tree& tree, const std::string& query_file, const MSA_Info& msa_info,
                     const std::string& outdir, const Options& options, const std::string& invocation,
                     const std::vector<int>& devices) {
  using clk = std::chrono::steady_clock;
  // premasking: masked columns are dropped from every query as it is read (src/seq/MSA_Stream.cpp:26)
  const bool premask = options.premasking && msa_info.gap_count() > 0;
  if (devices.empty()) throw std::runtime_error{"no device given"};
  Run_Stats st;
  st.host_threads = configure_host_threads();
  auto ts = clk::now();
  std::vector<std::unique_ptr<Device_Evaluator>> devs;
  for (int d : devices) devs.emplace_back(new Device_Evaluator(tree, options, d));
  st.ref_tree_logl = devs[0]->ref_tree_logl(0);
  st.seconds_setup = std::chrono::duration<double>(clk::now() - ts).count();
  const auto t_loop = clk::now();

  struct Staged { size_t index = 0, offset = 0; MSA chunk; Encoded_Chunk enc; };
  std::mutex mu;
  std::condition_variable cv_put, cv_get;
  std::deque<Staged> queue;
  bool eof = false;
  std::exception_ptr failure;
  const size_t depth = devices.size() + 1;  // staged chunks in flight
  std::vector<std::string> results;  // jplace text per chunk, in chunk order
  // the jplace is written while the run goes on: a chunk's text leaves as soon as every chunk before
  // it has (any worker that finds the file free drains what is ready, in chunk order)
  std::string dir = outdir;
  if (!dir.empty() && dir.back() != '/') dir += "/";
  const std::string out_path = dir + "epa_result.jplace";
  std::ofstream os(out_path);
  if (!os) throw std::runtime_error{"cannot open " + out_path};
  os << "{\n  \"tree\": \"" << tree.numbered_newick(options.precision) << "\",\n  \"placements\": \n  [\n";
  std::mutex wmu;
  std::vector<char> ready;
  size_t next_write = 0;
  bool first_chunk = true;
  auto drain = [&] {   // caller holds wmu
    for (;;) {
      std::string t;
      {
        std::lock_guard<std::mutex> lk(mu);
        if (next_write >= ready.size() || !ready[next_write]) return;
        t = std::move(results[next_write]);
        results[next_write] = std::string();
        ++next_write;
      }
      if (t.empty()) continue;
      const auto tw = clk::now();
      if (!first_chunk) os << ",\n";   // chunks separated by ",\n" (jplace_writer.hpp:141)
      first_chunk = false;
      os.write(t.data(), (std::streamsize)t.size());
      const double secs = std::chrono::duration<double>(clk::now() - tw).count();
      std::lock_guard<std::mutex> lk(mu);
      st.seconds_write += secs;
    }
  };

  // Reads per device chunk.  The device_min_chunk floor lifts the DEFAULT chunk size only: an explicit
  // --chunk-size is the user's memory knob (src/main.cpp:234-238) and is honoured as given unless
  // --device-min-chunk is given as well.  Either way the chunk is clamped to what the smallest device
  // has room for: two pipeline slots x Q x pitch(B) x 8 bytes of preplacement table (+ a quarter for
  // codes, bitmap, candidates) within half of its free memory.
  size_t device_chunk = options.chunk_size;
  if (!options.chunk_size_given || options.device_min_chunk_given)
    device_chunk = std::max<size_t>(device_chunk, options.device_min_chunk);
  {
    const size_t pitch = (tree.num_branches() * 8 + 63) / 64 * 64;
    for (auto& d : devs) {
      uint64_t fr = 0, tot = 0;
      if (epa_dev_mem_info(d->ctx(), &fr, &tot) != EPA_OK) continue;
      const size_t room = (size_t)(fr / 2) / (pitch * 2 * 5 / 4);
      if (room < device_chunk) device_chunk = std::max<size_t>(std::min<size_t>(options.chunk_size, device_chunk), std::max<size_t>(room, 1));
    }
  }
  // Post-processing pool.  The reference hands a finished chunk to an asynchronous writer (src/io/jplace_writer.hpp:58-69);
  // here everything behind the device calls -- pquery building, LWR, filter, jplace text -- leaves the device worker
  // with the chunk: the pool's threads take a finished chunk in kParts query-range jobs and run those stages
  // single-threaded, a dozen jobs at a time.  Round 6 measured the alternative (every stage an OpenMP region over all cores, entered from the stager, the
  // worker and a writer thread at once): three teams of 16 on 16 cores ran each stage 2 - 3 x slower than alone and the
  // 1 M-read run got slower, not faster (bench.py cli_e2e: 2.9 -> 2.5 M reads/s).  Text leaves in chunk order.
  const int host_cores = st.host_threads;
  // the stager's share of the cores: a FASTA file is 1.5 KB of text per 150 bp read of a 1500-column alignment (gap scan +
  // encoding: the largest host stage), a binary fasta file needs next to nothing
  Fasta_Stream reader(query_file);
  const int stager_team = reader.is_bfast() ? std::max(1, std::min(3, host_cores / 5)) : std::max(1, std::min(8, host_cores / 2));
  const int n_post = std::max(2, std::min(12, host_cores - (int)devices.size() - stager_team));
  constexpr int kSlots = 4;   // pipeline slots per device: a finished chunk's pinned rows stay valid for two more chunks
  constexpr size_t kParts = 4;   // post-processing jobs per chunk (query ranges): a 50 000-read chunk is ~35 ms of serial work,
                                 // which is what the run's tail -- and the pool's load balance -- is made of
  struct Finished {
    size_t index = 0, offset = 0;   // index: position of the text in the jplace (chunk index x kParts + part)
    std::shared_ptr<const MSA> chunk;   // headers (shared by the chunk's parts)
    size_t q_lo = 0, q_hi = 0;      // this job's queries of the chunk (raw rows only)
    bool have_sample = false;       // the worker already built (and filtered) the sample: text only
    Sample smp;
    const epa_pair* pairs = nullptr;   // else: the chunk's rows in the slot's pinned buffer, as the device returned them
    const epa_result* res = nullptr;
    size_t n = 0;
    std::atomic<int>* slot_busy = nullptr;   // cleared once the rows have been consumed (the worker re-stages the slot then)
  };
  std::vector<std::unique_ptr<std::atomic<int>[]>> slot_busy;
  for (size_t d = 0; d < devices.size(); ++d) {
    slot_busy.emplace_back(new std::atomic<int>[kSlots]);
    for (int i = 0; i < kSlots; ++i) slot_busy.back()[i].store(0);
  }
  std::deque<Finished> wq;
  std::mutex wq_mu;
  std::condition_variable wq_put, wq_get;
  bool workers_done = false, writer_failed = false;
  auto post_thread = [&] {
    set_thread_team(1);   // this thread's stages run serially: the parallelism is across chunks
    for (;;) {
      Finished f;
      {
        std::unique_lock<std::mutex> lk(wq_mu);
        wq_get.wait(lk, [&] { return !wq.empty() || workers_done || writer_failed; });
        if (writer_failed || wq.empty()) return;
        f = std::move(wq.front());
        wq.pop_front();
        wq_put.notify_all();
      }
      try {
        const auto t0 = clk::now();
        if (!f.have_sample) {
          Work work;
          std::vector<epa_result> res;
          work.reserve(f.n / kParts + f.n / 16 + 16);
          res.reserve(f.n / kParts + f.n / 16 + 16);
          for (size_t i = 0; i < f.n; ++i) {   // branch-major rows: this job's queries keep their order
            const uint32_t q = f.pairs[i].seq_id;
            if (q >= f.q_lo && q < f.q_hi) { work.push_back(Work_Pair{f.pairs[i].branch_id, q}); res.push_back(f.res[i]); }
          }
          f.slot_busy->fetch_sub(1, std::memory_order_acq_rel);   // the pinned rows are free again once every part has read them
          f.slot_busy = nullptr;
          build_sample(work, res.data(), *f.chunk, f.smp, f.offset);
        }
        const auto t1 = clk::now();
        if (!f.have_sample) {
          compute_and_set_lwr(f.smp);
          filter(f.smp, options);
        }
        const auto t2 = clk::now();
        std::string text = jplace_chunk_text(f.smp, options.precision, &tree.mapper());
        const auto t3 = clk::now();
        {
          std::lock_guard<std::mutex> lk(mu);
          if (results.size() <= f.index) { results.resize(f.index + 1); ready.resize(f.index + 1, 0); }
          results[f.index] = std::move(text);
          ready[f.index] = 1;
          st.seconds_sample += std::chrono::duration<double>(t1 - t0).count();
          st.seconds_post += std::chrono::duration<double>(t2 - t1).count();
          st.seconds_text += std::chrono::duration<double>(t3 - t2).count();
        }
        std::unique_lock<std::mutex> wl(wmu, std::try_to_lock);   // whoever finds the file free writes what is ready, in order
        if (wl.owns_lock()) drain();
      } catch (...) {
        {
          std::lock_guard<std::mutex> lk(mu);
          if (!failure) failure = std::current_exception();
          cv_put.notify_all();
          cv_get.notify_all();
        }
        if (f.slot_busy) f.slot_busy->store(0, std::memory_order_release);
        std::lock_guard<std::mutex> lk(wq_mu);
        writer_failed = true;
        for (auto& q : wq) if (q.slot_busy) q.slot_busy->store(0, std::memory_order_release);
        wq.clear();
        wq_put.notify_all();
        wq_get.notify_all();
        return;
      }
    }
  };
  std::vector<std::thread> post_pool;
  for (int i = 0; i < n_post; ++i) post_pool.emplace_back(post_thread);
  auto hand_over = [&](Finished&& f) {
    std::unique_lock<std::mutex> lk(wq_mu);
    wq_put.wait(lk, [&] { return wq.size() < (size_t)n_post + 2 * kParts || writer_failed; });
    if (writer_failed) { if (f.slot_busy) f.slot_busy->store(0, std::memory_order_release); return; }
    wq.push_back(std::move(f));
    wq_get.notify_one();
  };

  // The stager is two threads: the READER locates the next chunk's records (serial by nature: one record's end is the next
  // one's start; binary fasta: reads and re-aligns the nibbles) while the ENCODER turns the previous chunk's rows into wire
  // codes with its threads -- for a FASTA query file the two were 0.08 + 0.15 s of a 0.25-s run, one behind the other.
  struct Read_Chunk { Staged s; std::vector<const char*> rows; bool wire = false; double rd = 0; };
  std::deque<Read_Chunk> rq;
  std::mutex rq_mu;
  std::condition_variable rq_put, rq_get;
  bool read_done = false;
  std::thread reader_thread([&] {
    try {
      set_thread_team(std::max(1, std::min(3, stager_team)));
      size_t index = 0, offset = 0;
      for (;;) {
        Read_Chunk rc;
        Staged& s = rc.s;
        const auto r0 = clk::now();
        // one-line records of a mapped file are encoded straight from the mapping (no sequence strings)
        const size_t per_chunk = device_chunk;
        // binary fasta: the file's nibbles ARE the device's codes -- no ASCII stage (nucleotide data, no column mask)
        rc.wire = !premask && reader.is_bfast() && tree.model().num_states() == 4 &&
                  reader.read_next_wire(s.chunk, s.enc, tree.num_sites(), per_chunk, options.premasking) > 0;
        if (!rc.wire && !premask) reader.read_next_views(s.chunk, rc.rows, tree.num_sites(), per_chunk);
        if (!rc.wire && rc.rows.empty()) reader.read_next(s.chunk, per_chunk);
        rc.rd = std::chrono::duration<double>(clk::now() - r0).count();
        if (s.chunk.empty()) break;
        if (premask) s.chunk = subset_msa(s.chunk, msa_info.gap_mask());
        s.index = index++;
        s.offset = offset;
        offset += s.chunk.size();
        std::unique_lock<std::mutex> lk(rq_mu);
        rq_put.wait(lk, [&] { return rq.size() < 2 || read_done; });
        if (read_done) break;   // the encoder gave up (failure)
        rq.push_back(std::move(rc));
        rq_get.notify_one();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failure) failure = std::current_exception();
      cv_put.notify_all();
    }
    std::lock_guard<std::mutex> lk(rq_mu);
    read_done = true;
    rq_get.notify_all();
  });
  std::thread stager([&] {
    try {
      set_thread_team(stager_team);
      epa_encode_set_threads((unsigned)stager_team);
      for (;;) {
        Read_Chunk rc;
        {
          std::unique_lock<std::mutex> lk(rq_mu);
          rq_get.wait(lk, [&] { return !rq.empty() || read_done; });
          if (rq.empty()) break;
          rc = std::move(rq.front());
          rq.pop_front();
          rq_put.notify_one();
        }
        Staged& s = rc.s;
        const auto e0 = clk::now();
        if (!rc.wire) s.enc = rc.rows.empty() ? encode_chunk(s.chunk, tree, options) : encode_rows(rc.rows, s.chunk, tree, options);
        const double en = std::chrono::duration<double>(clk::now() - e0).count();
        std::unique_lock<std::mutex> lk(mu);
        st.seconds_read += rc.rd;
        st.seconds_encode += en;
        cv_put.wait(lk, [&] { return queue.size() < depth || failure; });
        if (failure) break;
        queue.push_back(std::move(s));
        cv_get.notify_one();
      }
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failure) failure = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lk(rq_mu);   // (on a failure: release a reader waiting for room)
      read_done = true;
      rq.clear();
      rq_put.notify_all();
    }
    std::lock_guard<std::mutex> lk(mu);
    eof = true;
    cv_get.notify_all();
  });

  // --host-heuristic: keep the Q x B table round trip and the host heuristics (cross-check); --no-pipeline: one chunk at a time
  const bool pipelined = options.prescoring && options.device_select && tree.num_branches() <= 65536 &&
                         !options.host_heuristic && !options.no_pipeline;
  auto worker = [&](size_t k) {
    try {
      auto take = [&](Staged& cur) -> bool {
        std::unique_lock<std::mutex> lk(mu);
        const auto tw = clk::now();
        cv_get.wait(lk, [&] { return !queue.empty() || eof || failure; });
        st.seconds_stage_wait += std::chrono::duration<double>(clk::now() - tw).count() / devices.size();
        if (failure || queue.empty()) return false;
        cur = std::move(queue.front());
        queue.pop_front();
        cv_put.notify_one();
        return true;
      };
      auto account = [&](size_t queries, const Chunk_Timing& tm) {
        std::lock_guard<std::mutex> lk(mu);
        st.queries += queries;
        st.pairs += tm.pairs;
        st.seconds_place += tm.place;
        st.seconds_thorough += tm.thorough;
      };
      if (!pipelined) {
        for (;;) {
          Staged cur;
          if (!take(cur)) return;
          Chunk_Timing tm;
          Finished f;
          f.smp = process_chunk(cur.chunk, cur.enc, tree, *devs[k], options, cur.offset, tm);
          {
            std::lock_guard<std::mutex> lk(mu);
            st.seconds_post += tm.post;
          }
          account(cur.chunk.size(), tm);
          f.index = cur.index * kParts; f.offset = cur.offset; f.have_sample = true;
          f.chunk = std::make_shared<const MSA>(std::move(cur.chunk));
          {
            std::lock_guard<std::mutex> lk(mu);   // the other part positions of this chunk stay empty
            if (results.size() < (cur.index + 1) * kParts) { results.resize((cur.index + 1) * kParts); ready.resize((cur.index + 1) * kParts, 0); }
            for (size_t pp = 1; pp < kParts; ++pp) ready[cur.index * kParts + pp] = 1;
          }
          hand_over(std::move(f));
        }
      }
      // two-slot pipeline: launch chunk i (its upload and kernels are queued, the call returns once
      // the candidate count is known), then finish chunk i-1 and hand its rows to the post-processing pool:
      // the worker goes straight back to its GPU
      Staged prev;
      bool have_prev = false;
      int slot = 0;
      for (;;) {
        Staged cur;
        const bool have_cur = take(cur);
        Chunk_Timing tm;
        if (have_cur) {
          // the slot's pinned result buffer may still be read by a post-processing thread (its chunk before last)
          while (slot_busy[k][slot].load(std::memory_order_acquire)) {
            { std::lock_guard<std::mutex> lk(wq_mu); if (writer_failed) return; }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
          }
          const auto t0 = clk::now();
          chunk_launch(cur.enc, cur.chunk.size(), tree, *devs[k], slot, options);
          tm.place = std::chrono::duration<double>(clk::now() - t0).count();
        }
        if (have_prev) {
          const auto t1 = clk::now();
          Finished f;
          const epa_pair* pairs = nullptr;
          const epa_result* res = nullptr;
          uint64_t n = 0;
          const int pslot = (slot + kSlots - 1) % kSlots;
          const int rc = epa_dev_chunk_finish(devs[k]->ctx(), pslot, &pairs, &res, &n, nullptr);
          if (rc == EPA_ERR_NEG_INF) throw std::runtime_error{epa_dev_last_error(devs[k]->ctx())};  // Tiny_Tree.cpp:209-212
          if (rc != EPA_OK) throw_dev(devs[k]->ctx(), rc);
          tm.pairs = n;
          tm.thorough = std::chrono::duration<double>(clk::now() - t1).count();
          account(prev.chunk.size(), tm);
          // rows read in place by kParts jobs: valid until the slot is staged again (guarded by slot_busy = parts still reading)
          slot_busy[k][pslot].store((int)kParts, std::memory_order_release);
          const auto headers = std::make_shared<const MSA>(std::move(prev.chunk));
          const size_t Qp = headers->size();
          for (size_t pp = 0; pp < kParts; ++pp) {
            Finished part;
            part.pairs = pairs; part.res = res; part.n = n;
            part.slot_busy = &slot_busy[k][pslot];
            part.index = prev.index * kParts + pp; part.offset = prev.offset;
            part.chunk = headers;
            part.q_lo = Qp * pp / kParts; part.q_hi = Qp * (pp + 1) / kParts;
            hand_over(std::move(part));
          }
        } else if (have_cur) {
          std::lock_guard<std::mutex> lk(mu);
          st.seconds_place += tm.place;
        }
        if (!have_cur) break;
        prev = std::move(cur);
        have_prev = true;
        slot = (slot + 1) % kSlots;
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failure) {
        const std::string what = e.what();
        if (what.find("hipMalloc") != std::string::npos)   // the chunk did not fit: name the knobs
          failure = std::make_exception_ptr(std::runtime_error{what + " (device chunks of " + std::to_string(device_chunk) +
                    " queries; lower --chunk-size, and pass --device-min-chunk 0 if it was raised)"});
        else failure = std::current_exception();
      }
      cv_put.notify_all();
      cv_get.notify_all();
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failure) failure = std::current_exception();
      cv_put.notify_all();
      cv_get.notify_all();
    }
  };
  std::vector<std::thread> workers;
  for (size_t k = 1; k < devices.size(); ++k) workers.emplace_back(worker, k);
  worker(0);
  for (auto& w : workers) w.join();
  {
    std::lock_guard<std::mutex> lk(mu);
    cv_put.notify_all();
  }
  stager.join();
  reader_thread.join();
  {
    std::lock_guard<std::mutex> lk(wq_mu);
    workers_done = true;
    wq_get.notify_all();
  }
  for (auto& t : post_pool) t.join();
  if (failure) {
    os.close();
    std::remove(out_path.c_str());   // no half-written result file
    std::rethrow_exception(failure);
  }
  {
    std::lock_guard<std::mutex> wl(wmu);
    drain();
  }
  ts = clk::now();
  os << "  ],\n  \"metadata\": {\"invocation\": \"" << invocation << "\"},\n  \"version\": 3,\n"
     << "  \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\""
     << ", \"pendant_length\"]\n}\n";
  os.flush();
  if (!os) throw std::runtime_error{"writing " + out_path + " failed"};
  st.seconds_write += std::chrono::duration<double>(clk::now() - ts).count();
  st.seconds_loop = std::chrono::duration<double>(clk::now() - t_loop).count();
  return st;
}

}  // namespace epa
