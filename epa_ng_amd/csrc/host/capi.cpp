// C API of libepa_host.so for the Python harness (tests / bench): reference precompute and the
// full chunk loop, without any likelihood code in Python.
#include <cstring>
#include <fstream>
#include <sstream>

#include "epa_host.hpp"

using namespace epa;

namespace {
thread_local std::string g_err;
struct Ref {
  std::unique_ptr<Tree> tree;
  Options opt;
};
template <class F>
int guarded(F&& f) {
  try { f(); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}
}  // namespace

extern "C" {

const char* epa_host_last_error() { return g_err.c_str(); }

// model given either as descriptor string (model_desc != NULL) or explicit arrays
void* epa_host_ref_create_ex2(const char* newick, int n_seqs, const char* const* labels,
                              const char* const* seqs, const char* model_desc, int states,
                              const double* subst, const double* freqs, int cats, const double* rates,
                              const double* weights, double pinv, int preserve_rooting) {
  Ref* r = nullptr;
  if (guarded([&] {
        MSA msa;
        for (int i = 0; i < n_seqs; ++i) msa.emplace_back(labels[i], seqs[i]);
        Model m = model_desc ? Model(std::string(model_desc))
                             : Model(states, std::vector<double>(subst, subst + states * (states - 1) / 2),
                                     std::vector<double>(freqs, freqs + states),
                                     std::vector<double>(rates, rates + cats),
                                     weights ? std::vector<double>(weights, weights + cats)
                                             : std::vector<double>(),
                                     pinv);
        r = new Ref();
        r->opt.preserve_rooting = preserve_rooting != 0;
        r->tree.reset(new Tree(newick, msa, m, r->opt));
      }))
    return nullptr;
  return r;
}

void* epa_host_ref_create_ex(const char* newick, int n_seqs, const char* const* labels,
                             const char* const* seqs, const char* model_desc, int states,
                             const double* subst, const double* freqs, int cats, const double* rates,
                             const double* weights, double pinv) {
  return epa_host_ref_create_ex2(newick, n_seqs, labels, seqs, model_desc, states, subst, freqs, cats, rates,
                                 weights, pinv, 1);
}

void* epa_host_ref_create(const char* newick, int n_seqs, const char* const* labels,
                          const char* const* seqs, const char* model_desc, int states,
                          const double* subst, const double* freqs, int cats, const double* rates,
                          const double* weights) {
  return epa_host_ref_create_ex(newick, n_seqs, labels, seqs, model_desc, states, subst, freqs, cats, rates,
                                weights, 0.0);
}

void epa_host_ref_destroy(void* h) { delete static_cast<Ref*>(h); }

// OpenMP thread cap of this process (affinity + cgroup quota); returns the count in effect
int epa_host_configure_threads() { return epa::configure_host_threads(); }

void epa_host_ref_dims(void* h, uint32_t* states, uint32_t* cats, uint32_t* sites, uint32_t* branches) {
  const Tree& t = *static_cast<Ref*>(h)->tree;
  *states = t.model().num_states(); *cats = t.model().num_ratecats();
  *sites = (uint32_t)t.num_sites(); *branches = (uint32_t)t.num_branches();
}

double epa_host_ref_tree_logl(void* h, uint32_t branch) {
  return static_cast<Ref*>(h)->tree->ref_tree_logl(branch);
}

int epa_host_ref_numbered_newick(void* h, unsigned precision, char* out, size_t cap) {
  const std::string s = static_cast<Ref*>(h)->tree->numbered_newick(precision);
  if (s.size() + 1 > cap) return -(int)s.size();
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// model descriptor scraped from a model file (src/util/parse_model.hpp); returns the length, or
// -needed if `cap` is too small, or INT_MIN on error (epa_host_last_error)
int epa_host_parse_model(const char* file, char* out, size_t cap) {
  std::string s;
  if (guarded([&] { s = epa::parse_model(file); })) return -2147483647 - 1;
  if (s.size() + 1 > cap) return -(int)s.size();
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// premasking column mask of a (reference file, query file) pair: MSA_Info of both + or_mask
// (src/main.cpp:470-490).  mask[i] = 1: column i is removed.  Returns the number of sites, -1 on error.
long epa_host_premask(const char* ref_file, const char* query_file, uint8_t* mask, size_t cap) {
  long sites = -1;
  if (guarded([&] {
        epa::MSA_Info r = epa::MSA_Info::from_file(ref_file), q = epa::MSA_Info::from_file(query_file);
        epa::MSA_Info::or_mask(r, q);
        if (r.sites() > cap) throw std::runtime_error{"epa_host_premask: mask buffer too small"};
        std::memcpy(mask, r.gap_mask().data(), r.sites());
        sites = (long)r.sites();
      }))
    return -1;
  return sites;
}

// rooted input tree: placement (edge, distal) on the unrooted working tree -> on the rooted tree
// (rtree_mapper::in_rtree).  Returns 1 when no mapping is active (unrooted input / preserve off).
int epa_host_ref_in_rtree(void* h, uint32_t branch, double distal, uint32_t* out_branch, double* out_distal) {
  const auto& m = static_cast<Ref*>(h)->tree->mapper();
  if (!m) return 1;
  return guarded([&] {
    const auto p = m.in_rtree(branch, distal);
    *out_branch = p.first;
    *out_distal = p.second;
  });
}

void epa_host_ref_model(void* h, double* eigenvals, double* u, double* uinv, double* freqs,
                        double* rates, double* weights) {
  const Model& m = static_cast<Ref*>(h)->tree->model();
  const int s = m.num_states(), c = m.num_ratecats();
  std::memcpy(eigenvals, m.eigenvals().data(), sizeof(double) * s);
  std::memcpy(u, m.eigenvecs_u().data(), sizeof(double) * s * s);
  std::memcpy(uinv, m.eigenvecs_uinv().data(), sizeof(double) * s * s);
  std::memcpy(freqs, m.base_freqs().data(), sizeof(double) * s);
  std::memcpy(rates, m.ratecat_rates().data(), sizeof(double) * c);
  std::memcpy(weights, m.ratecat_weights().data(), sizeof(double) * c);
}

// exchangeabilities (upper triangle, row-major) and the model string the tree really uses
void epa_host_ref_subst(void* h, double* subst) {
  const Model& m = static_cast<Ref*>(h)->tree->model();
  std::memcpy(subst, m.subst_rates().data(), sizeof(double) * m.subst_rates().size());
}
int epa_host_ref_model_string(void* h, char* out, size_t cap) {
  const std::string s = static_cast<Ref*>(h)->tree->model().to_string();
  if (s.size() + 1 > cap) return -1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// borrowed pointers into the tree for branch b (for parity tests against the oracle)
void epa_host_ref_branch(void* h, uint32_t b, const double** prox_clv, const uint32_t** prox_sc,
                         const double** dist_clv, const uint8_t** dist_tip, const uint32_t** dist_sc,
                         double* length) {
  const Tree::Branch br = static_cast<Ref*>(h)->tree->branch(b);
  *prox_clv = br.prox_clv; *prox_sc = br.prox_scaler; *dist_clv = br.dist_clv;
  *dist_tip = br.dist_tipchars; *dist_sc = br.dist_scaler; *length = br.length;
}

uint32_t epa_host_ref_tipmap(void* h, uint32_t* out, uint32_t cap) {
  const auto& tm = static_cast<Ref*>(h)->tree->tipmap();
  for (uint32_t i = 0; i < tm.size() && i < cap; ++i) out[i] = tm[i];
  return (uint32_t)tm.size();
}

// creates the device context straight from the host tree (what simple_mpi does internally).
// device_precompute != 0: reference CLVs computed on the GPU from the tree; 0: host CLVs uploaded.
// flags: EPA_FLAG_* of include/epa_dev.h (per-rate scalers, --raxml-blo); 0 = defaults
int epa_host_dev_create_flags(void* h, int device, int aa_x_as_n, int device_precompute, uint32_t flags,
                              epa_ctx** out);
int epa_host_dev_create_ex(void* h, int device, int aa_x_as_n, int device_precompute, epa_ctx** out) {
  return epa_host_dev_create_flags(h, device, aa_x_as_n, device_precompute, 0, out);
}
int epa_host_dev_create_opts(void* h, int device, int aa_x_as_n, int device_precompute, uint32_t flags,
                             double blo_min_branch, epa_ctx** out);
int epa_host_dev_create_flags(void* h, int device, int aa_x_as_n, int device_precompute, uint32_t flags,
                              epa_ctx** out) {
  return epa_host_dev_create_opts(h, device, aa_x_as_n, device_precompute, flags, 0.0, out);
}
// blo_min_branch: PLLMOD_OPT_MIN_BRANCH_LEN (0 = the library default)
int epa_host_dev_create_opts(void* h, int device, int aa_x_as_n, int device_precompute, uint32_t flags,
                             double blo_min_branch, epa_ctx** out) {
  const Tree& t = *static_cast<Ref*>(h)->tree;
  int rc;
  if (device_precompute) {
    epa_tree_desc d;
    Tree::Tree_Desc_Storage store;
    t.fill_tree_desc(d, store);
    d.ref.aa_x_as_n = aa_x_as_n;
    d.ref.flags = flags;
    d.ref.blo_min_branch = blo_min_branch;
    rc = epa_dev_create_from_tree(&d, device, out);
  } else {
    if (flags & EPA_FLAG_RATE_SCALERS) {   // as Device_Evaluator: the host CLV path keeps per-site scalers only
      g_err = "per-rate scalers need the device-side reference precompute (the host CLV path keeps per-site scalers)";
      return EPA_ERR_UNSUPPORTED;
    }
    epa_ref_desc d;
    std::vector<const double*> pc, dc;
    std::vector<const uint32_t*> ps, ds;
    std::vector<const uint8_t*> dt;
    std::vector<double> bl;
    t.fill_desc(d, pc, ps, dc, dt, ds, bl);
    d.aa_x_as_n = aa_x_as_n;
    d.flags = flags;
    d.blo_min_branch = blo_min_branch;
    rc = epa_dev_create(&d, device, out);
  }
  if (rc) g_err = epa_dev_last_error(nullptr);
  return rc;
}
int epa_host_dev_create(void* h, int device, int aa_x_as_n, epa_ctx** out) {
  return epa_host_dev_create_ex(h, device, aa_x_as_n, 1, out);
}

// the whole pipeline: query fasta -> <outdir>/epa_result.jplace
int epa_host_place_file(void* h, const char* query_file, const char* outdir, uint32_t chunk_size,
                        int prescoring, double prescoring_threshold, int premasking, int device,
                        const char* invocation, uint64_t* n_queries, uint64_t* n_pairs) {
  return guarded([&] {
    Ref* r = static_cast<Ref*>(h);
    Options o = r->opt;
    if (chunk_size) { o.chunk_size = chunk_size; o.chunk_size_given = true; o.device_min_chunk = 0; }   // the harness asks for exact chunks
    o.prescoring = prescoring != 0;
    if (prescoring_threshold > 0) o.prescoring_threshold = prescoring_threshold;
    o.premasking = premasking != 0;
    const Run_Stats st = simple_mpi(*r->tree, query_file, outdir, o, invocation ? invocation : "epa_host_place_file", device);
    if (n_queries) *n_queries = st.queries;
    if (n_pairs) *n_pairs = st.pairs;
  });
}

// host-side set operations exposed for unit tests with literal vectors
// (the reference's test/src/set_manipulators.cpp:340-443)
// the C++ chunk loop's query slice of a rank (place_ranks.cpp; src/net/epa_mpi_util.cpp:10-30)
void epa_host_local_seq_package(uint64_t num_sequences, int rank, int world, uint64_t* offset, uint64_t* count) {
  const auto p = epa::local_seq_package((size_t)num_sequences, rank, world);
  *offset = p.first;
  *count = p.second;
}

int epa_host_filter(const double* lwr, uint32_t n, double thresh, int acc, uint32_t mn, uint32_t mx,
                    uint32_t* kept_branch_ids, uint32_t* n_kept) {
  return guarded([&] {
    Sample s(1);
    for (uint32_t i = 0; i < n; ++i) { s[0].emplace_back(i, 0.0, 0.0, 0.0); s[0][i].lwr(lwr[i]); }
    Options o;
    o.support_threshold = thresh; o.acc_threshold = acc != 0; o.filter_min = mn; o.filter_max = mx;
    filter(s, o);
    *n_kept = (uint32_t)s[0].size();
    for (uint32_t i = 0; i < *n_kept; ++i) kept_branch_ids[i] = (uint32_t)s[0][i].branch_id();
  });
}

int epa_host_heuristic(const double* lnl, uint32_t Q, uint32_t B, int mode, double thresh,
                       uint32_t* pair_branch, uint32_t* pair_seq, uint64_t cap, uint64_t* n) {
  return guarded([&] {
    Options o;
    o.prescoring_threshold = thresh;
    o.prescoring_by_percentage = mode == 1;
    o.baseball = mode == 2;
    std::vector<double> v(lnl, lnl + (size_t)Q * B);
    const Work w = apply_heuristic(v, Q, B, o);
    *n = w.size();
    for (size_t i = 0; i < w.size() && i < cap; ++i) {
      pair_branch[i] = (uint32_t)w[i].branch_id;
      pair_seq[i] = (uint32_t)w[i].sequence_id;
    }
  });
}

// test hook: the jplace number formatter on an array of values, one text of `width` bytes per value (NUL-padded)
int epa_host_format_fixed(const double* v, size_t n, unsigned int precision, char* out, size_t width) {
  for (size_t i = 0; i < n; ++i) {
    char buf[512];
    const size_t len = epa::format_fixed(buf, sizeof(buf), v[i], precision);
    if (len >= width) return -1;
    std::memset(out + i * width, 0, width);
    std::memcpy(out + i * width, buf, len);
  }
  return 0;
}

}  // extern "C"
