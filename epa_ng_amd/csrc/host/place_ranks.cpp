// The chunk loop in one-process-per-GPU mode (the reference's MPI build of simple_mpi,
// src/core/place.cpp:173-251 with src/net/epa_mpi_util.cpp:10-30 and src/io/jplace_writer.hpp:117-129):
// every rank places the contiguous slice of the query file that local_seq_package() assigns to it on
// its own GPU; after each chunk the (pair, result) rows are posted to the product library's RCCL gather
// (epa_dev_gather_slot: asynchronous, fixed-size, carry-over -- include/epa_dev.h); rank 0 collects them
// a chunk behind, runs compute_and_set_lwr + filter on complete queries and writes the jplace.
//
// Launch: N processes (mpirun / torchrun / a shell loop) with --rank r --world N --comm-file F, or the
// environment torchrun sets (RANK, WORLD_SIZE, LOCAL_RANK) plus EPA_COMM_FILE.  Rank 0 writes the 128-byte
// RCCL id to F (atomically), the others wait for it: no MPI, no torch in the product.
#include "epa_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <thread>
#include <unordered_map>

namespace epa {

namespace {
[[noreturn]] void throw_dev(epa_ctx* ctx, int rc) {
  throw std::runtime_error{std::string(epa_dev_last_error(ctx)) + " (epa_dev status " + std::to_string(rc) + ")"};
}
}  // namespace

std::pair<size_t, size_t> local_seq_package(size_t num_sequences, int rank, int world) {
  // src/net/epa_mpi_util.cpp:10-30: part size ceil(n / world), trailing ranks may come out empty
  const size_t part = (num_sequences + (size_t)world - 1) / (size_t)world;
  const size_t offset = std::min(part * (size_t)rank, num_sequences);
  return {offset, std::min(part, num_sequences - offset)};
}

// The 128-byte RCCL id travels through a file.  A file left behind by an earlier run (a crashed one: a
// healthy rank 0 removes it once the communicator exists) must never be taken for this run's: the record
// carries a nonce (Options::comm_nonce: --comm-nonce / EPA_COMM_NONCE / torchrun's TORCHELASTIC_RUN_ID, hashed) and rank 0's
// wall-clock time of writing; a reader accepts a record only with its own nonce.  Without a nonce (plain mpirun /
// srun / a shell loop) what keeps an older run's file out is rank 0 itself -- it REPLACES the file before anything
// slow and removes it once the communicator exists -- plus a coarse age test: a record written more than 15 minutes
// before the reader started is ignored (the ranks of one job start within that, and their clocks agree to within it;
// a tighter bound would reject a valid id on a late rank or a skewed node: ADVICE round 5).  Give a nonce
// (--comm-nonce) whenever the launcher can.
namespace {
struct Id_Record {
  char magic[8];
  uint64_t nonce;
  int64_t written_unix_ms;
  unsigned char id[EPA_COMM_ID_BYTES];
};

uint64_t comm_nonce(const std::string& text) {   // Options::comm_nonce, hashed; 0 = none given
  if (text.empty()) return 0;
  uint64_t h = 1469598103934665603ull;   // FNV-1a
  for (const char ch : text) h = (h ^ (unsigned char)ch) * 1099511628211ull;
  return h ? h : 1;
}

int64_t unix_ms() {
  return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}
}  // namespace

static void publish_id(const std::string& path, uint64_t nonce, unsigned char (&id)[EPA_COMM_ID_BYTES]) {
  if (path.empty()) throw std::runtime_error{"--world > 1 needs --comm-file (or EPA_COMM_FILE): where rank 0 leaves the RCCL id"};
  std::remove(path.c_str());   // whatever an earlier run left there
  if (epa_comm_get_unique_id(id) != EPA_OK) throw std::runtime_error{epa_dev_last_error(nullptr)};
  Id_Record rec{};
  std::memcpy(rec.magic, "EPACOMM1", 8);
  rec.nonce = nonce;
  rec.written_unix_ms = unix_ms();
  std::memcpy(rec.id, id, sizeof(id));
  const std::string tmp = path + ".tmp";
  std::FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f || std::fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec)) throw std::runtime_error{"cannot write " + tmp};
  std::fclose(f);
  if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error{"cannot rename " + tmp};
}

static void await_id(const std::string& path, uint64_t nonce, int rank, int64_t started_unix_ms, unsigned char (&id)[EPA_COMM_ID_BYTES]) {
  if (path.empty()) throw std::runtime_error{"--world > 1 needs --comm-file (or EPA_COMM_FILE): where rank 0 leaves the RCCL id"};
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (std::FILE* f = std::fopen(path.c_str(), "rb")) {
      Id_Record rec{};
      const size_t n = std::fread(&rec, 1, sizeof(rec), f);
      std::fclose(f);
      if (n == sizeof(rec) && std::memcmp(rec.magic, "EPACOMM1", 8) == 0 && rec.nonce == nonce &&
          (nonce != 0 || rec.written_unix_ms >= started_unix_ms - 15 * 60000)) {
        std::memcpy(id, rec.id, sizeof(id));
        return;
      }
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300))
      throw std::runtime_error{"rank " + std::to_string(rank) + ": no RCCL id of this run in " + path + " after 300 s"};
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

Run_Stats simple_mpi_ranks(const Tree& tree, const std::string& query_file, const MSA_Info& msa_info,
                           const std::string& outdir, const Options& options, const std::string& invocation,
                           int device, int rank, int world, const std::string& comm_file) {
  using clk = std::chrono::steady_clock;
  // --no-heur (prescoring == false, src/core/place.cpp:219-231): every branch is placed thoroughly, LWR over all of them
  // and the filter run on the device (epa_dev_place_all_rows); the kept placements travel WITH their like-weight ratios
  const bool no_heur = !options.prescoring;
  if (no_heur && !(options.filter_min >= 1 && options.filter_max >= options.filter_min && options.filter_max <= 64))
    throw std::runtime_error{"--no-heur in the one-process-per-GPU mode needs 1 <= --filter-min <= --filter-max <= 64"};
  const int64_t started = unix_ms();
  unsigned char id[EPA_COMM_ID_BYTES] = {};
  // rank 0 replaces the id file before anything slow (the reference precompute, the scan of the query
  // file): the other ranks, which only look for it after their own setup, never meet an older run's
  const uint64_t nonce = comm_nonce(options.comm_nonce);
  if (rank == 0) publish_id(comm_file, nonce, id);
  const bool premask = options.premasking && msa_info.gap_count() > 0;
  Run_Stats st;
  configure_host_threads();
  auto ts = clk::now();
  Device_Evaluator dev(tree, options, device);
  st.ref_tree_logl = dev.ref_tree_logl(0);
  // one pass over the query file: the record count every rank needs for its slice, and on rank 0 the
  // headers of ALL queries (it writes every rank's placements)
  std::vector<std::string> headers;
  size_t total = 0;
  {
    Fasta_Stream all(query_file);
    MSA blk;
    for (;;) {
      blk.clear();
      const size_t got = all.read_next(blk, 65536);
      if (!got) break;
      if (rank == 0)
        for (size_t i = 0; i < got; ++i) headers.push_back(blk[i].header());
      total += got;
    }
  }
  const auto slice = local_seq_package(total, rank, world);
  size_t per_chunk = options.chunk_size;
  if (!options.chunk_size_given || options.device_min_chunk_given) per_chunk = std::max<size_t>(per_chunk, options.device_min_chunk);
  if (no_heur) per_chunk = std::max<size_t>(1, std::min<size_t>(per_chunk, 0xffffffffull / std::max<size_t>(1, tree.num_branches())));   // B x Q pairs per call < 2^32
  const size_t part = (total + (size_t)world - 1) / (size_t)world;
  const size_t nchunks = (part + per_chunk - 1) / per_chunk;   // the SAME on every rank: posts are collective
  // rows per rank and gather: candidates per read average 2 .. 3 under the default heuristic; beyond that
  // the carry path takes over (--comm-rows-per-read)
  const size_t rows_per_read = (size_t)std::max(1, options.comm_rows_per_read);
  if (rank != 0) await_id(comm_file, nonce, rank, started, id);
  if (options.comm_timeout_seconds > 0) epa_comm_set_timeout(nullptr, options.comm_timeout_seconds);
  epa_comm* comm = nullptr;
  int rc = epa_comm_create(dev.ctx(), id, rank, world, (uint32_t)std::min<size_t>(per_chunk * rows_per_read, 0x7fffffffu), 2, &comm);
  if (rank == 0) std::remove(comm_file.c_str());   // collective: every rank has read it by now
  if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
  // a rank that fails from here on aborts the communicator instead of leaving its peers blocked in a
  // send / receive for ever (they fail or time out: epa_comm_set_timeout / EPA_COMM_TIMEOUT_S); the reference's MPI build aborts the job
  struct Guard {
    epa_comm* c;
    int live = std::uncaught_exceptions();
    ~Guard() { if (std::uncaught_exceptions() > live) epa_comm_abort(c); else epa_comm_destroy(c); }
  } guard{comm};
  if (options.comm_self_send) epa_comm_set_self_send(comm, 1);
  {
    // handshake before any work depends on the transport: a one-row gather + an all-reduce, every wait bounded
    std::vector<uint64_t> dev_ids((size_t)world);
    rc = epa_comm_probe(dev.ctx(), comm, options.comm_probe_seconds, dev_ids.data());
    if (rc != EPA_OK)
      throw std::runtime_error{std::string("the ranks cannot exchange data (") + epa_dev_last_error(dev.ctx()) + "); transport library: " +
                               epa_comm_library_path() + " -- see --help on EPA_RCCL_LIB / LD_LIBRARY_PATH"};
    if (rank == 0) {
      std::string devs;
      for (int r = 0; r < world; ++r) {
        char b[32];
        std::snprintf(b, sizeof b, "%s%04x:%02x:%02x", r ? " " : "", (unsigned)(dev_ids[r] >> 16), (unsigned)((dev_ids[r] >> 8) & 0xff),
                      (unsigned)(dev_ids[r] & 0xff));
        devs += b;
      }
      std::fprintf(stderr, "epa-ng-amd: %d ranks on devices [%s], transport %s\n", world, devs.c_str(), epa_comm_library_path());
    }
  }
  st.seconds_setup = std::chrono::duration<double>(clk::now() - ts).count();

  std::ofstream os;
  std::string out_path;
  bool first_text = true;
  if (rank == 0) {
    std::string dir = outdir;
    if (!dir.empty() && dir.back() != '/') dir += "/";
    out_path = dir + "epa_result.jplace";
    os.open(out_path);
    if (!os) throw std::runtime_error{"cannot open " + out_path};
    os << "{\n  \"tree\": \"" << tree.numbered_newick(options.precision) << "\",\n  \"placements\": \n  [\n";
  }
  // rank 0: rows of every rank accumulate until that rank reports nothing carried (a chunk's rows are
  // branch-major, so a query is complete only when all rows posted so far have arrived)
  std::vector<std::vector<epa_row>> acc(rank == 0 ? world : 0);
  auto emit = [&](std::vector<epa_row>& rows) {
    if (rows.empty()) return;
    const auto t0 = clk::now();
    std::unordered_map<uint32_t, size_t> at;
    Sample smp;
    size_t n_placements = 0;
    for (const epa_row& r : rows) {   // arrival order = (chunk, branch) order: kept inside every pquery
      auto it = at.find(r.seq_id);
      if (r.branch_id == EPA_ROW_LWR) {   // --no-heur: the like-weight ratio of the placement row just before it
        if (it == at.end() || smp[it->second].size() == 0) throw std::runtime_error{"LWR row without its placement"};
        PQuery& pq = smp[it->second];
        pq[pq.size() - 1].lwr(r.lnl);
        continue;
      }
      if (it == at.end()) {
        it = at.emplace(r.seq_id, smp.size()).first;
        smp.emplace_back((size_t)r.seq_id, headers.at(r.seq_id));
      }
      smp[it->second].emplace_back((size_t)r.branch_id, r.lnl, r.pendant_length, r.distal_length);
      ++n_placements;
    }
    std::stable_sort(smp.begin(), smp.end(), [](const PQuery& a, const PQuery& b) { return a.sequence_id() < b.sequence_id(); });
    st.pairs += no_heur ? smp.size() * tree.num_branches() : n_placements;
    st.queries += smp.size();
    rows.clear();
    if (!no_heur) {   // (--no-heur: LWR and filter ran on the device, over all branches)
      compute_and_set_lwr(smp);
      filter(smp, options);
    }
    const auto t1 = clk::now();
    const std::string text = jplace_chunk_text(smp, options.precision, &tree.mapper());
    if (!text.empty()) {
      if (!first_text) os << ",\n";
      first_text = false;
      os.write(text.data(), (std::streamsize)text.size());
    }
    st.seconds_post += std::chrono::duration<double>(t1 - t0).count();
    st.seconds_write += std::chrono::duration<double>(clk::now() - t1).count();
  };
  auto collect = [&](uint64_t ticket) {
    std::vector<const epa_row*> rows(world);
    std::vector<uint32_t> counts(world);
    std::vector<uint64_t> pend(world);
    const int r2 = epa_comm_collect(comm, ticket, rows.data(), counts.data(), pend.data());
    if (r2 != EPA_OK) throw_dev(dev.ctx(), r2);
    for (int r = 0; r < world; ++r) {
      acc[r].insert(acc[r].end(), rows[r], rows[r] + counts[r]);
      if (pend[r] == 0) emit(acc[r]);
    }
  };

  Fasta_Stream reader(query_file);
  {  // skip to this rank's slice
    MSA skip;
    size_t left = slice.first;
    while (left) {
      skip.clear();
      const size_t got = reader.read_next(skip, std::min<size_t>(left, 65536));
      if (!got) break;
      left -= got;
    }
  }
  const int mode = options.baseball ? EPA_HEUR_BASEBALL : options.prescoring_by_percentage ? EPA_HEUR_FIXED : EPA_HEUR_DYNAMIC;
  if (epa_dev_set_heuristic(dev.ctx(), mode, mode == EPA_HEUR_FIXED ? options.prescoring_threshold : 0.0) != EPA_OK)
    throw std::runtime_error{epa_dev_last_error(dev.ctx())};
  const size_t nb = tree.num_branches();
  size_t done = 0;
  std::vector<uint64_t> tickets(nchunks);
  int prev_slot = -1;
  for (size_t k = 0; k < nchunks; ++k) {
    const int slot = (int)(k & 1);
    MSA chunk;
    const auto r0 = clk::now();
    if (done < slice.second) reader.read_next(chunk, std::min(per_chunk, slice.second - done));
    st.seconds_read += std::chrono::duration<double>(clk::now() - r0).count();
    const auto t0 = clk::now();
    if (!chunk.empty() && no_heur) {
      if (premask) chunk = subset_msa(chunk, msa_info.gap_mask());
      const Encoded_Chunk enc = encode_chunk(chunk, tree, options);
      const size_t Q = chunk.size();
      uint32_t max_span = 0;
      for (uint32_t s : enc.win_span) max_span = std::max(max_span, s);
      epa_dev_set_query_layout(dev.ctx(), enc.stride);
      epa_dev_set_query_packing(dev.ctx(), enc.bits);
      const epa_row* d_rows = nullptr;
      uint64_t n_rows = 0;
      rc = epa_dev_place_all_rows(dev.ctx(), enc.codes.data(), enc.win_begin.data(), enc.win_span.data(), (uint32_t)Q, max_span,
                                  options.support_threshold, options.acc_threshold ? 1 : 0, options.filter_min, options.filter_max,
                                  (uint32_t)(slice.first + done), &d_rows, &n_rows, nullptr);
      if (rc == EPA_ERR_NEG_INF) throw std::runtime_error{epa_dev_last_error(dev.ctx())};
      if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
      rc = epa_dev_gather_rows(dev.ctx(), comm, d_rows, n_rows, &tickets[k]);
      if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
      done += Q;
    } else if (!chunk.empty()) {
      if (premask) chunk = subset_msa(chunk, msa_info.gap_mask());
      const Encoded_Chunk enc = encode_chunk(chunk, tree, options);
      const size_t Q = chunk.size();
      uint64_t cap = std::max<uint64_t>((uint64_t)Q * 8, dev.pair_capacity());
      if (mode == EPA_HEUR_FIXED)
        cap = std::max<uint64_t>(cap, (uint64_t)Q * std::min<size_t>(nb, (size_t)std::ceil(options.prescoring_threshold * (double)nb)));
      else if (mode == EPA_HEUR_BASEBALL)
        cap = std::max<uint64_t>(cap, (uint64_t)Q * std::min<size_t>(nb, 46));
      uint32_t max_span = 0;
      for (uint32_t s : enc.win_span) max_span = std::max(max_span, s);
      epa_dev_set_query_layout(dev.ctx(), enc.stride);
      epa_dev_set_query_packing(dev.ctx(), enc.bits);
      rc = epa_dev_chunk_stage(dev.ctx(), slot, enc.codes.data(), enc.win_begin.data(), enc.win_span.data(), (uint32_t)Q);
      if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
      for (;;) {   // results stay in HBM: the gather reads them there
        rc = epa_dev_chunk_launch(dev.ctx(), slot, max_span, options.prescoring_threshold, nullptr, nullptr, cap, EPA_CHUNK_NO_D2H);
        if (rc == EPA_ERR_PAIR_OVERFLOW && cap < (uint64_t)Q * nb) { cap = std::min<uint64_t>(cap * 8, (uint64_t)Q * nb); continue; }
        if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
        break;
      }
      dev.pair_capacity() = cap;
      rc = epa_dev_gather_slot(dev.ctx(), comm, slot, (uint32_t)(slice.first + done), &tickets[k]);
      if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
      done += Q;
    } else {   // this rank's slice is exhausted: the post is collective all the same
      rc = epa_dev_gather_results(dev.ctx(), comm, nullptr, nullptr, 0, 0, &tickets[k]);
      if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
    }
    // retire the previous chunk's slot (its kernels ran while this chunk was read, encoded and queued)
    if (prev_slot >= 0) {
      rc = epa_dev_chunk_finish(dev.ctx(), prev_slot, nullptr, nullptr, nullptr, nullptr);
      if (rc == EPA_ERR_NEG_INF) throw std::runtime_error{epa_dev_last_error(dev.ctx())};
      if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
    }
    prev_slot = (chunk.empty() || no_heur) ? -1 : slot;
    st.seconds_place += std::chrono::duration<double>(clk::now() - t0).count();
    if (rank == 0 && k >= 1) collect(tickets[k - 1]);
  }
  if (prev_slot >= 0) {
    rc = epa_dev_chunk_finish(dev.ctx(), prev_slot, nullptr, nullptr, nullptr, nullptr);
    if (rc == EPA_ERR_NEG_INF) throw std::runtime_error{epa_dev_last_error(dev.ctx())};
    if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
  }
  if (rank == 0 && nchunks) collect(tickets[nchunks - 1]);
  for (;;) {   // drain carried rows: rounds of at most `depth` extra gathers, agreed on by an all-reduce each
    uint64_t first_extra = 0;
    uint32_t n_extra = 0;
    rc = epa_comm_flush(dev.ctx(), comm, &first_extra, &n_extra);
    if (rc != EPA_OK) throw_dev(dev.ctx(), rc);
    if (!n_extra) break;
    if (rank == 0)
      for (uint32_t i = 0; i < n_extra; ++i) collect(first_extra + i);
  }
  if (rank == 0) {
    for (auto& a : acc)
      if (!a.empty()) throw std::runtime_error{"rows left incomplete after the last gather"};
    os << "  ],\n  \"metadata\": {\"invocation\": \"" << invocation << "\"},\n  \"version\": 3,\n"
       << "  \"fields\": [\"edge_num\", \"likelihood\", \"like_weight_ratio\", \"distal_length\""
       << ", \"pendant_length\"]\n}\n";
    os.flush();
    if (!os) throw std::runtime_error{"writing " + out_path + " failed"};
  } else {
    st.queries = done;
  }
  return st;
}

}  // namespace epa
