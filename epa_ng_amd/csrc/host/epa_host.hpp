// Host side of the MI355X placement evaluator: the C++ mirror of the reference's operator
// interface for the placement path, sitting directly above the C-ABI of include/epa_dev.h.
// Class / function names follow the reference (paths relative to the reference checkout):
//   Options            src/util/Options.hpp:6-35
//   Sequence / MSA     src/seq/Sequence.hpp, src/seq/MSA.hpp
//   Placement/PQuery/Sample  src/sample/
//   Work               src/core/Work.hpp
//   Model              src/core/raxml/Model.hpp (subset: GTR / PROTGTR + FU/FE + G, see model.cpp)
//   Tree               src/tree/Tree.hpp (owns all directional CLVs of the reference tree)
//   place / place_thorough / simple_mpi   src/core/place.cpp
//   apply_heuristic    src/core/heuristics.hpp
//   compute_and_set_lwr / filter          src/set_manipulators.cpp
//   jplace output      src/io/jplace_util.cpp, src/io/jplace_writer.hpp
#pragma once

#include <cstddef>
#include <cstdio>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "epa_dev.h"

namespace epa {

struct Options {  // defaults: src/util/Options.hpp:13-34
  bool prescoring = true;
  bool sliding_blo = true;
  double support_threshold = 0.01;
  bool acc_threshold = false;
  unsigned int filter_min = 1;
  unsigned int filter_max = 7;
  bool prescoring_by_percentage = false;
  double prescoring_threshold = 0.99999;
  unsigned int chunk_size = 50000;  // the reference's CPU default is 5000 (Options.hpp:26); a GPU wants larger chunks
  // `--chunk-size` is "number of query sequences to be read in at a time; may influence performance"
  // (src/main.cpp:234-238): a memory / speed knob without an effect on the result.  The device
  // runs 5000-read chunks at 42 % of its 100k-read rate, so the chunk loop reads
  // at least this many sequences per device chunk, whatever smaller value was asked for (0: never
  // more than chunk_size)
  unsigned int device_min_chunk = 40000;
  // an explicit --chunk-size is the user's memory knob and is honoured as given: the floor above
  // applies to the DEFAULT chunk size only (or when --device-min-chunk itself is given)
  bool chunk_size_given = false, device_min_chunk_given = false;
  unsigned int num_threads = 0;
  bool premasking = true;
  bool baseball = false;
  unsigned int precision = 10;
  bool aa_x_as_n = false;   // quirk D4 switch
  bool device_select = true;  // run the dynamic heuristic on the GPU (default heuristic only)
  bool device_precompute = true;  // reference CLVs computed on the GPU from the tree (no host CLVs)
  bool preserve_rooting = true;   // rooted input: jplace on the rooted tree (Options.hpp:34)
  // per-rate scalers (Options.hpp:33, src/main.cpp:248-250,399-407): auto = on above 2000 tips
  // (Tree_Numbers::large_tree, src/io/file_io.cpp:211-214)
  enum class NumericalScaling { kOn, kOff, kAuto };
  NumericalScaling scaling = NumericalScaling::kAuto;
  bool rate_scalers(size_t tips) const {
    return scaling == NumericalScaling::kOn || (scaling == NumericalScaling::kAuto && tips > 2000);
  }
  // one-process-per-GPU mode (place_ranks.cpp; the reference's MPI build has no counterpart of these: mpirun is its launcher)
  std::string comm_nonce;            // --comm-nonce / EPA_COMM_NONCE / TORCHELASTIC_RUN_ID: marks this run's id record in --comm-file
  int comm_rows_per_read = 8;        // --comm-rows-per-read: rows per rank and gather = chunk x this (beyond it: the carry path)
  double comm_probe_seconds = 120;   // --comm-probe-seconds: bound of the handshake after the communicator is created
  double comm_timeout_seconds = 0;   // --comm-timeout: bound of every later wait for a peer (0: EPA_COMM_TIMEOUT_S, else 600)
  bool comm_self_send = false;       // --comm-self-send (test hook): rank 0's own rows travel through ncclSend / ncclRecv
  bool host_heuristic = false;       // --host-heuristic (diagnostic): candidate selection on the host from the Q x B table
  bool no_pipeline = false;          // --no-pipeline (diagnostic): one chunk at a time, no read-ahead / staged slots
};

class Sequence {
public:
  Sequence() = default;
  Sequence(std::string header, std::string sequence)
      : header_(std::move(header)), sequence_(std::move(sequence)) {}
  const std::string& header() const { return header_; }
  const std::string& sequence() const { return sequence_; }
private:
  std::string header_, sequence_;
};

using MSA = std::vector<Sequence>;

// Column premasking (src/seq/MSA_Info.hpp:13-111, src/main.cpp:470-494): a column that is a gap in
// EVERY sequence of the reference file, or in every sequence of the query file, is removed from both
// alignments before anything else happens (`or_mask`, "like masking in pplacer").  It changes no
// LWR (such a column multiplies every branch's likelihood of a query by the same factor) but it is
// what makes the absolute `likelihood` field of the jplace equal to the reference's.
// Gap characters: genesis' gap_sites() default, the "undetermined" nucleic-acid codes N O X . - ?
// in either case -- also for protein data, as the reference calls it (a recollection of
// genesis/sequence/functions/functions.hpp; genesis is an empty submodule in the reference checkout).
class MSA_Info {
public:
  using mask_type = std::vector<uint8_t>;   // 1 = gap in every sequence seen
  MSA_Info() = default;
  explicit MSA_Info(const MSA& msa);                 // one pass over sequences already in memory
  // one streamed pass over a (binary) fasta file; stops early once no column can stay masked
  static MSA_Info from_file(const std::string& path);
  size_t sites() const { return sites_; }
  size_t sequences() const { return sequences_; }    // sequences looked at (a lower bound after an early stop)
  const mask_type& gap_mask() const { return gap_mask_; }
  size_t gap_count() const;
  void add(const std::string& seq);                  // gap_mask &= gap sites of seq
  static void or_mask(MSA_Info& lhs, MSA_Info& rhs); // throws on unequal widths
private:
  size_t sites_ = 0, sequences_ = 0;
  mask_type gap_mask_;
};
std::string subset_sequence(const std::string& seq, const MSA_Info::mask_type& mask);
MSA subset_msa(const MSA& msa, const MSA_Info::mask_type& mask);

class Placement {  // src/sample/Placement.hpp:7-54
public:
  Placement() = default;
  Placement(size_t branch_id, double likelihood, double pendant_length, double distal_length)
      : branch_id_(branch_id), likelihood_(likelihood), lwr_(0.0),
        pendant_length_(pendant_length), distal_length_(distal_length) {}
  double lwr() const { return lwr_; }
  double likelihood() const { return likelihood_; }
  double pendant_length() const { return pendant_length_; }
  double distal_length() const { return distal_length_; }
  size_t branch_id() const { return branch_id_; }
  void lwr(double v) { lwr_ = v; }
private:
  size_t branch_id_ = 0;
  double likelihood_ = 0, lwr_ = 0, pendant_length_ = 0, distal_length_ = 0;
};

class PQuery {  // src/sample/PQuery.hpp:12-92
public:
  PQuery() = default;
  PQuery(size_t seq_id, std::string header) : sequence_id_(seq_id), header_(std::move(header)) {}
  size_t sequence_id() const { return sequence_id_; }
  const std::string& header() const { return header_; }
  std::vector<Placement>& placements() { return placements_; }
  const std::vector<Placement>& placements() const { return placements_; }
  auto begin() { return placements_.begin(); }
  auto end() { return placements_.end(); }
  auto begin() const { return placements_.begin(); }
  auto end() const { return placements_.end(); }
  size_t size() const { return placements_.size(); }
  Placement& operator[](size_t i) { return placements_[i]; }
  template <class... A> void emplace_back(A&&... a) { placements_.emplace_back(std::forward<A>(a)...); }
private:
  size_t sequence_id_ = 0;
  std::string header_;
  std::vector<Placement> placements_;
};

using Sample = std::vector<PQuery>;

struct Work_Pair {  // src/core/Work.hpp:31-34
  size_t branch_id;
  size_t sequence_id;
};
using Work = std::vector<Work_Pair>;  // kept branch-major sorted (the reference's map order)

// named empirical amino-acid matrix (aa_models.cpp): 190 exchangeabilities (upper triangle,
// row-major, libpll state order) + 20 equilibrium frequencies
struct Named_AA_Model {
  const char* name;
  const double* rates;
  const double* freqs;
};
const Named_AA_Model* named_aa_model(const std::string& upper_name);

class Model {
public:
  Model() = default;
  // raxml-ng style descriptor (src/core/raxml/Model.cpp:123-538):
  //   NAME  [{r1/r2/...}]  [+FU{f1/..} | +FE | +FO(->equal) | +F | +FC (empirical, from the
  //         reference MSA)]  [+I{p}]  [+G[n][{alpha}] | +R[n]{rates}{weights}]
  //   NAME: GTR, the named nucleotide models with their rate symmetries (JC K80 F81 HKY TN93[ef]
  //         K81[uf] TPM2[uf] TPM3[uf] TIM1[uf] TIM2[uf] TIM3[uf] TVM[ef] SYM), PROTGTR, or a named
  //         empirical amino-acid matrix (LG WAG JTT DAYHOFF: rates + frequencies from the table)
  explicit Model(const std::string& descriptor);
  Model(int states, std::vector<double> subst, std::vector<double> freqs, std::vector<double> rates,
        std::vector<double> weights, double pinv = 0.0);
  int num_states() const { return states_; }
  int num_ratecats() const { return (int)rates_.size(); }
  const std::vector<double>& subst_rates() const { return subst_; }
  const std::vector<double>& base_freqs() const { return freqs_; }
  const std::vector<double>& ratecat_rates() const { return rates_; }
  const std::vector<double>& ratecat_weights() const { return weights_; }
  double alpha() const { return alpha_; }
  double pinv() const { return pinv_; }  // +I: proportion of invariant sites
  // +F / +FC: the frequencies are counted on the reference MSA (link_tree_msa,
  // src/core/pll/epa_pll_util.cpp:55-57); Tree's constructor does that and calls set_base_freqs
  bool empirical_base_freqs() const { return empirical_freqs_; }
  void set_base_freqs(std::vector<double> freqs);
  const std::vector<double>& eigenvals() const { return eigenvals_; }
  const std::vector<double>& eigenvecs_u() const { return u_; }       // libpll inv_eigenvecs
  const std::vector<double>& eigenvecs_uinv() const { return uinv_; }  // libpll eigenvecs
  // state bitmask of a character (pll_map_nt / pll_map_aa); 0 = invalid
  uint32_t char_mask(char c) const;
  // P(t) for rate category k, row-major states x states
  void pmatrix(double t, int k, double* P) const;
  std::string to_string() const;
private:
  void update_eigen();
  int states_ = 4;
  double alpha_ = 1.0, pinv_ = 0.0;
  bool free_rates_ = false, empirical_freqs_ = false;
  std::string name_ = "GTR";
  std::vector<double> subst_, freqs_, rates_, weights_, eigenvals_, u_, uinv_;
};

std::vector<double> compute_gamma_cats(double alpha, int k, bool median = false);  // pll_compute_gamma_cats
// model descriptor from a RAxML 8 info / RAxML-NG .bestModel / IQ-TREE report file
// (src/util/parse_model.hpp)
std::string parse_model(const std::string& file);

struct Tree_Numbers {  // src/tree/Tree_Numbers.hpp
  unsigned int tip_nodes = 0, inner_nodes = 0, nodes = 0, branches = 0;
};

// Edge-number translation from the unrooted working tree back to a rooted input tree
// (src/core/pll/rtree_mapper.hpp:9-110; built like determine_edge_num_translation,
// src/io/file_io.cpp:46-116).  Inactive (operator bool false) for unrooted input.
class Rtree_Mapper {
public:
  explicit operator bool() const { return !map_.empty(); }
  // (branch_id, distal_length) of a placement on the unrooted tree -> the same on the rooted tree;
  // the former root edge splits into a distal part (kept) and a proximal part (direction flipped)
  std::pair<unsigned int, double> in_rtree(unsigned int branch_id, double distal_length) const;
  unsigned int map_at(size_t i) const;  // throws for the root edge
  bool uroot_is_left() const { return left_; }
  unsigned int utree_root_edge = 0, rtree_proximal_edge = 0, rtree_distal_edge = 0;
  double proximal_edge_length = -1.0, distal_edge_length = -1.0;
  bool left_ = true;
  std::string root_label;
  std::vector<unsigned int> map_;
};

// Reference tree with all 3(n-2) directional CLVs precomputed (Tree::Tree src/tree/Tree.cpp:16-56,
// precompute_clvs src/core/pll/epa_pll_util.cpp:62-107).
class Tree {
public:
  Tree(const std::string& newick, const MSA& ref_msa, const Model& model, const Options& options);
  const Tree_Numbers& nums() const { return nums_; }
  const Model& model() const { return model_; }
  size_t num_sites() const { return sites_; }
  size_t num_branches() const { return nums_.branches; }
  bool rooted_input() const { return rooted_input_; }  // the newick had a bifurcating root (removed)
  double ref_tree_logl(size_t branch = 0) const;  // edge lnL (Tree::ref_tree_logl :119-131)
  // numbered newick, edge ids in utree_query_branches order (pll_util.cpp:182-259)
  // Rooted input with preserve_rooting: the rooted tree with its own edge numbering
  // (pll_util.cpp:227-352; literals test/src/pll_util.cpp:159-186).
  std::string numbered_newick(unsigned int precision) const;
  const Rtree_Mapper& mapper() const { return mapper_; }

  // the two sides of branch b after the tip-is-distal orientation of Tiny_Tree.cpp:64-74
  struct Branch {
    const double* prox_clv;
    const uint32_t* prox_scaler;
    const double* dist_clv;        // nullptr when the distal end is a tip
    const uint8_t* dist_tipchars;  // tip codes (state bitmask index into tipmap) or nullptr
    const uint32_t* dist_scaler;   // nullptr for a tip
    double length;
  };
  Branch branch(size_t b) const;
  const std::vector<uint32_t>& tipmap() const { return tipmap_; }
  // +I: state of the sites invariant over the reference tips, -1 otherwise (pll_update_invariant_sites)
  const std::vector<int8_t>& invariant_state() const { return invariant_; }
  // fills an epa_ref_desc that borrows this tree's buffers (valid while *this lives)
  void fill_desc(epa_ref_desc& d, std::vector<const double*>& pc, std::vector<const uint32_t*>& ps,
                 std::vector<const double*>& dc, std::vector<const uint8_t*>& dt,
                 std::vector<const uint32_t*>& ds, std::vector<double>& bl) const;
  // tree + tip sequences for the device-side reference precompute (epa_dev_create_from_tree);
  // needs no host CLVs.  `store` owns the arrays the descriptor points to.
  struct Tree_Desc_Storage {
    std::vector<uint8_t> tipchars;
    std::vector<uint32_t> child_a, child_b, prox, dist;
    std::vector<double> len_a, len_b, blen;
  };
  void fill_tree_desc(epa_tree_desc& d, Tree_Desc_Storage& store) const;
  // host CLVs are computed on first use (Tree::branch / ref_tree_logl / fill_desc): the device
  // path of the product never needs them
  void ensure_host_clvs() const;

private:
  struct Rec { int next = -1, back = -1, tip = -1; double length = 0.0; std::string label; };
  int new_rec();
  int parse_subtree(const char*& p, double& len);
  void compute_all_clvs();
  mutable bool host_clvs_ready_ = false;
  void side(int rec, const double*& clv, const uint8_t*& tip, const uint32_t*& sc) const;

  Model model_;
  Tree_Numbers nums_;
  size_t sites_ = 0;
  std::vector<Rec> recs_;
  std::vector<std::string> labels_;
  int vroot_ = -1;
  bool rooted_input_ = false;
  Rtree_Mapper mapper_;
  std::string root_label_;
  std::vector<int> branch_rec_;
  std::vector<std::vector<uint8_t>> tipchars_;   // per tip
  std::vector<std::vector<double>> clv_;         // per record (empty for tips)
  std::vector<std::vector<uint32_t>> scaler_;    // per record
  std::vector<uint32_t> tipmap_;
  std::vector<int8_t> invariant_;
};

// Caps OpenMP at the CPUs this process may really use (affinity mask and cgroup cpu.max quota:
// a container that sees 256 CPUs but is limited to 16 crawls with 256 spinning threads).
// Called once by the entry points; returns the thread count in effect.
int configure_host_threads();
// OpenMP team size of the CALLING host thread's stages from now on (0 = all usable CPUs): the chunk loop gives its
// stager a small team and runs its post-processing threads serially, one whole chunk each
void set_thread_team(int n);
// user limit on the host threads (-T/--threads, src/main.cpp:256-258); call before the first host stage
void set_host_thread_limit(int n);

// ---- readers / writers
MSA read_fasta(const std::string& path);
// chunked reader of the query file: FASTA (the reference's MSA_Stream, src/seq/MSA_Stream.hpp) or
// binary fasta (.bfast, Binary_Fasta_Reader), detected by the magic
class Fasta_Stream {
public:
  explicit Fasta_Stream(const std::string& path);
  ~Fasta_Stream();
  Fasta_Stream(const Fasta_Stream&) = delete;
  // appends up to max_seqs sequences to `out`; returns how many (0 = end of file)
  size_t read_next(MSA& out, size_t max_seqs);
  // zero-copy variant (mapped files, one sequence line of `sites` characters per record): headers in
  // `out`, pointers into the mapping in `rows`; 0 = not applicable for the next chunk, use read_next()
  size_t read_next_views(MSA& out, std::vector<const char*>& rows, size_t sites, size_t max_seqs);
  // binary fasta only: the next records go STRAIGHT to the compact 4-bit wire rows the device reads (Encoded_Chunk) --
  // the file's nibbles are the device's codes (src/io/encoding.hpp:11-134), so there is no ASCII stage at all: window =
  // first / last non-zero nibble (get_valid_range, src/util/Range.hpp:34-49), row = the window's nibbles re-aligned.
  // Headers in `out` (empty sequences).  0 = not a mapped bfast file, or its end.
  size_t read_next_wire(MSA& out, struct Encoded_Chunk& enc, size_t sites, size_t max_seqs, bool premasking);
  bool is_bfast() const { return bfast_; }
private:
  size_t index_records(size_t max_seqs, bool& done, bool predict = false);
  void consume(size_t m);
  bool refill();
  bool open_bfast();                                  // binary fasta (src/io/Binary_Fasta.hpp)?
  size_t read_next_bfast(MSA& out, size_t max_seqs);
  bool bfast_ = false;
  std::vector<uint64_t> bfast_offsets_;
  size_t bfast_next_ = 0;
  std::FILE* f_ = nullptr;
  const char* map_ = nullptr;             // regular fasta files are mapped (no copy through a read buffer)
  size_t map_len_ = 0;
  std::vector<char> buf_;
  size_t pos_ = 0, len_ = 0, scan_ = 0;  // unparsed region [pos_, len_), record index built up to scan_
  std::vector<size_t> starts_;            // offsets of the '>' of the records found so far
  bool eof_ = false, first_block_ = true;
  // read_next_views: records of one header line + one sequence line all have the same distance from the end of the
  // header line to the next record; once learned, the next '>' is LOOKED UP there instead of searched for (the index
  // was a serial memchr over the whole file: 1.5 GB per million reads of a 1500-column alignment).  Every record found
  // that way is validated by read_next_views (length, no line break inside); a miss falls back to the search for good.
  size_t seq_part_ = 0;
  bool predict_ok_ = true;
  char up_[256];
};
// v in fixed notation with `precision` digits, correctly rounded (= printf %.*f); returns the length written to buf
size_t format_fixed(char* buf, size_t cap, double v, unsigned int precision);
void write_jplace(std::ostream& os, const std::vector<Sample>& chunks, const std::string& newick,
                  const std::string& invocation, unsigned int precision,
                  const Rtree_Mapper* mapper = nullptr);
// the same in two steps: every chunk is turned into text as soon as it is done (by the device
// worker that produced it), the file is assembled at the end
std::string jplace_chunk_text(const Sample& sample, unsigned int precision,
                              const Rtree_Mapper* mapper = nullptr);
void write_jplace_text(std::ostream& os, const std::vector<std::string>& chunk_texts, const std::string& newick,
                       const std::string& invocation);

// ---- device-backed evaluator: one epa_ctx, RAII
class Device_Evaluator {
public:
  Device_Evaluator(const Tree& tree, const Options& options, int device = 0);
  ~Device_Evaluator();
  Device_Evaluator(const Device_Evaluator&) = delete;
  epa_ctx* ctx() const { return ctx_; }
  // host staging of the candidate placements of a chunk, reused from chunk to chunk (a fresh
  // zero-filled 100 MB vector per chunk costs more than the device work)
  std::vector<epa_pair>& pair_buffer() { return pairs_buf_; }
  std::vector<epa_result>& result_buffer() { return res_buf_; }
  uint64_t& pair_capacity() { return pair_cap_; }  // candidate capacity the chunk pipeline last needed
  double ref_tree_logl(size_t branch = 0) const;  // Tree::ref_tree_logl evaluated on the device
private:
  epa_ctx* ctx_ = nullptr;
  std::vector<epa_pair> pairs_buf_;
  std::vector<epa_result> res_buf_;
  uint64_t pair_cap_ = 0;
};

// encoded chunk of queries (what crosses the C-ABI)
struct Encoded_Chunk {
  std::vector<uint8_t> codes;  // compact layout: row q = the window of query q, `stride` codes;
                               // nucleotides travel in the 4-bit wire format (bits == 4: two codes per byte)
  int bits = 8;
  std::vector<uint32_t> win_begin, win_span;
  uint32_t stride = 0;
};
Encoded_Chunk encode_chunk(const MSA& chunk, const Tree& tree, const Options& options);
// the same from `sites`-long character rows (not NUL-terminated); `chunk` supplies the headers for the messages
Encoded_Chunk encode_rows(const std::vector<const char*>& rows, const MSA& chunk, const Tree& tree,
                          const Options& options);

// Hot loop 1: dense Q x B preplacement table (src/core/place.cpp:41-95).  lnl is Q x B row-major;
// sample[q][b] = Placement{b, lnl, pendant = -ln 0.9, distal = len/2} is materialised lazily by
// the heuristics below, not here (the reference's 40-byte-per-cell Sample is never built).
void place(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
           std::vector<double>& lnl, const Options& options);

// Candidate selection (src/core/heuristics.hpp:119-127): dynamic (default), fixed-%, baseball.
Work apply_heuristic(const std::vector<double>& lnl, size_t num_sequences, size_t num_branches,
                     const Options& options);

// Hot loop 2 (src/core/place.cpp:97-171): one PQuery per query with its candidate placements.
void place_thorough(const Work& to_place, const MSA& chunk, const Encoded_Chunk& enc,
                    const Tree& tree, Device_Evaluator& dev, Sample& sample, const Options& options,
                    size_t seq_id_offset = 0);

// place() + apply_heuristic(dynamic) + place_thorough() in one device-side pass
// (epa_dev_place_chunk); fills `work` (branch-major) and `sample`.
void place_chunk(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
                 Work& work, Sample& sample, const Options& options, size_t seq_id_offset = 0);

// --no-heur with the chunk's post-processing on the device (epa_dev_place_all): thorough
// placement on every branch, compute_and_set_lwr over all of them and filter(); `sample` comes
// back final (LWR set, filtered, best first).  Returns the number of pairs evaluated.
size_t place_all(const MSA& chunk, const Encoded_Chunk& enc, const Tree& tree, Device_Evaluator& dev,
                 Sample& sample, const Options& options, size_t seq_id_offset = 0);

void compute_and_set_lwr(Sample& sample);            // src/set_manipulators.cpp:43-69
void filter(Sample& sample, const Options& options);  // :192-204

// The chunk loop (src/core/place.cpp:173-251): reads `query_file` in chunks, writes
// <outdir>/epa_result.jplace.  rank/world: contiguous query sharding as local_seq_package
// (src/net/epa_mpi_util.cpp:10-30); every rank returns its own samples, rank 0 writes.
struct Run_Stats {
  size_t queries = 0, pairs = 0;
  double ref_tree_logl = 0;  // of the reference tree, evaluated on the device at branch 0
  double seconds_place = 0, seconds_thorough = 0;  // device calls (fused path: all in seconds_place)
  double seconds_setup = 0, seconds_read = 0, seconds_stage_wait = 0, seconds_post = 0, seconds_write = 0;
  // finer stages of the chunk loop (busy time of the stage, summed over chunks; the stages run on different threads
  // and overlap, so they do not add up to seconds_loop = wall time from the first read to the closed jplace)
  double seconds_encode = 0, seconds_sample = 0, seconds_text = 0, seconds_loop = 0;
  int host_threads = 0;
};
Run_Stats simple_mpi(const Tree& tree, const std::string& query_file, const std::string& outdir,
                     const Options& options, const std::string& invocation, int device = 0);
// same, one worker thread per listed GPU; the jplace does not depend on the device count
Run_Stats simple_mpi(const Tree& tree, const std::string& query_file, const std::string& outdir,
                     const Options& options, const std::string& invocation, const std::vector<int>& devices);
// as the reference's signature (src/core/place.cpp:173): with the premasking column mask the query
// reader applies to every sequence (src/seq/MSA_Stream.cpp:26); an empty mask = no column removed
Run_Stats simple_mpi(const Tree& tree, const std::string& query_file, const MSA_Info& msa_info,
                     const std::string& outdir, const Options& options, const std::string& invocation,
                     const std::vector<int>& devices);

// one process per GPU: rank's contiguous slice of the query file on `device`, results gathered to rank 0
// over the product library's RCCL gather (place_ranks.cpp; src/net/epa_mpi_util.cpp:10-30)
std::pair<size_t, size_t> local_seq_package(size_t num_sequences, int rank, int world);
Run_Stats simple_mpi_ranks(const Tree& tree, const std::string& query_file, const MSA_Info& msa_info,
                           const std::string& outdir, const Options& options, const std::string& invocation,
                           int device, int rank, int world, const std::string& comm_file);

}  // namespace epa
