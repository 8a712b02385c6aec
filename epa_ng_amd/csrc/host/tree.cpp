// Reference tree: newick -> unrooted record structure, branch numbering, tip encoding and the
// one-off precompute of every directional CLV (what Tree::Tree does through libpll,
// src/tree/Tree.cpp:16-56; src/core/pll/epa_pll_util.cpp:10-107).  Runs once per run on the
// host; its buffers are what epa_dev_create uploads to HBM.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <unordered_map>

#include "epa_host.hpp"

namespace epa {

namespace {
const double kDefaultBranchLength = -std::log(0.9);  // src/util/constants.hpp:12
const double kScaleThreshold = std::ldexp(1.0, -256);
const double kScaleFactor = std::ldexp(1.0, 256);

void skip_ws(const char*& p) { while (*p && std::isspace((unsigned char)*p)) ++p; }

void label_and_length(const char*& p, std::string* label, double& len) {
  skip_ws(p);
  const char* b = p;
  while (*p && !std::strchr(":,();", *p) && !std::isspace((unsigned char)*p)) ++p;
  if (label) label->assign(b, p);
  skip_ws(p);
  len = 0.0;
  if (*p == ':') {
    ++p;
    char* e = nullptr;
    len = std::strtod(p, &e);
    p = e;
  }
  skip_ws(p);
}
}  // namespace

int Tree::new_rec() {
  recs_.emplace_back();
  return (int)recs_.size() - 1;
}

// returns the record facing the parent
int Tree::parse_subtree(const char*& p, double& len) {
  skip_ws(p);
  if (*p == '(') {
    ++p;
    double l1, l2;
    const int k1 = parse_subtree(p, l1);
    skip_ws(p);
    if (*p != ',') throw std::invalid_argument{"Input Tree contains a unary node or is malformed!"};
    ++p;
    const int k2 = parse_subtree(p, l2);
    skip_ws(p);
    if (*p == ',') throw std::invalid_argument{"Input Tree contains multifurcations (polytomies)!"};
    if (*p != ')') throw std::runtime_error{"Treeparsing failed! expected ')'"};
    ++p;
    const int a = new_rec(), b = new_rec(), c = new_rec();
    recs_[a].next = b; recs_[b].next = c; recs_[c].next = a;
    recs_[b].back = k1; recs_[k1].back = b; recs_[b].length = recs_[k1].length = l1;
    recs_[c].back = k2; recs_[k2].back = c; recs_[c].length = recs_[k2].length = l2;
    label_and_length(p, &recs_[a].label, len);  // inner label (printed by numbered_newick)
    return a;
  }
  const int a = new_rec();
  recs_[a].tip = (int)labels_.size();
  std::string lab;
  label_and_length(p, &lab, len);
  if (lab.empty()) throw std::runtime_error{"Treeparsing failed! empty tip label"};
  labels_.push_back(lab);
  return a;
}

Tree::Tree(const std::string& newick, const MSA& ref_msa, const Model& model, const Options& options)
    : model_(model) {
  const char* p = newick.c_str();
  skip_ws(p);
  if (*p != '(') throw std::runtime_error{"Treeparsing failed! expected '('"};
  ++p;
  int kids[3];
  double kl[3];
  int nk = 0;
  for (;;) {
    if (nk == 3) throw std::invalid_argument{"Input Tree contains multifurcations (polytomies)!"};
    kids[nk] = parse_subtree(p, kl[nk]);
    ++nk;
    skip_ws(p);
    if (*p == ',') { ++p; continue; }
    if (*p == ')') { ++p; break; }
    throw std::runtime_error{"Treeparsing failed! expected ',' or ')'"};
  }
  double top_len = 0.0;
  {  // label (and an ignored length) after the top-level ')'
    std::string lab;
    label_and_length(p, &lab, top_len);
    root_label_ = lab;
  }
  if (nk == 2) {
    // Rooted input (src/io/file_io.cpp:129-171): the root is removed and its two edges become
    // one, as libpll's pll_rtree_unroot does -- the first root child that has descendants becomes
    // the top-level trifurcation, its ring = {other root child, its left child, its right child}.
    // When that is the LEFT root child the reference moves the virtual root one ring step on
    // (file_io.cpp:147-153), so that the branch order is left child, right child, other subtree
    // and the former root edge is the last branch.
    const bool left = recs_[kids[0]].next >= 0;
    const int top = left ? kids[0] : kids[1], other = left ? kids[1] : kids[0];
    if (recs_[top].next < 0) throw std::runtime_error{"Number of tip nodes too small"};
    recs_[top].back = other;
    recs_[other].back = top;
    recs_[top].length = recs_[other].length = kl[0] + kl[1];
    rooted_input_ = true;
    vroot_ = left ? recs_[top].next : top;
    const std::string rtree_root_label = root_label_;
    root_label_ = recs_[top].label;  // the utree root carries the label of the promoted child
    if (options.preserve_rooting) {
      // determine_edge_num_translation (file_io.cpp:60-116)
      const unsigned n_tips = (unsigned)labels_.size();
      const unsigned nb = 2 * n_tips - 3;
      mapper_.left_ = left;
      mapper_.root_label = rtree_root_label;
      mapper_.map_.resize(nb);
      if (left) {
        // branches of the promoted child's two subtrees keep their numbers, the rooted edge above
        // the promoted child takes the next number, everything in the other subtree shifts by one
        struct Count { static unsigned edges(const std::vector<Rec>& r, int rec) {
          // edges of the subtree hanging below `rec` including the edge above it
          unsigned cnt = 0; std::vector<int> st{rec};
          while (!st.empty()) { const int x = st.back(); st.pop_back(); ++cnt;
            if (r[x].next >= 0) { st.push_back(r[r[x].next].back); st.push_back(r[r[r[x].next].next].back); } }
          return cnt; } };
        const unsigned below = Count::edges(recs_, recs_[vroot_].back) +
                               Count::edges(recs_, recs_[recs_[vroot_].next].back);
        for (unsigned i = 0; i < nb; ++i) mapper_.map_[i] = i < below ? i : i + 1;
        mapper_.rtree_proximal_edge = below;
        mapper_.rtree_distal_edge = mapper_.map_.back();
        mapper_.utree_root_edge = nb - 1;
        mapper_.proximal_edge_length = kl[0];
        mapper_.distal_edge_length = kl[1];
      } else {
        // the left root child is a tip: it is branch 0 on both trees, every other branch keeps
        // its number, the rooted edge above the promoted (right) child is the extra last one
        for (unsigned i = 0; i < nb; ++i) mapper_.map_[i] = i;
        mapper_.rtree_distal_edge = 0;
        mapper_.utree_root_edge = 0;
        mapper_.rtree_proximal_edge = nb;
        mapper_.proximal_edge_length = kl[1];
        mapper_.distal_edge_length = kl[0];
      }
    }
  } else {
  const int a = new_rec(), b = new_rec(), c = new_rec();
  recs_[a].next = b; recs_[b].next = c; recs_[c].next = a;
  const int ring[3] = {a, b, c};
  for (int i = 0; i < 3; ++i) {
    recs_[ring[i]].back = kids[i];
    recs_[kids[i]].back = ring[i];
    recs_[ring[i]].length = recs_[kids[i]].length = kl[i];
  }
  vroot_ = a;
  }
  if (labels_.size() < 3) throw std::runtime_error{"Number of tip nodes too small"};
  // set_missing_branch_lengths (src/core/pll/pll_util.cpp:13-39): a zero length counts as missing
  for (auto& r : recs_)
    if (r.length == 0.0) r.length = kDefaultBranchLength;
  const unsigned n = (unsigned)labels_.size();
  nums_.tip_nodes = n;
  nums_.inner_nodes = n - 2;
  nums_.nodes = 2 * n - 2;
  nums_.branches = 2 * n - 3;

  // utree_query_branches (src/core/pll/pll_util.cpp:182-205): post-order from the three
  // subtrees of the virtual root; the visited record is the distal end of its branch
  branch_rec_.reserve(nums_.branches);
  {
    struct Frame { int rec; int state; };
    for (int start : {recs_[vroot_].back, recs_[recs_[vroot_].next].back,
                      recs_[recs_[recs_[vroot_].next].next].back}) {
      std::vector<Frame> st{{start, 0}};
      while (!st.empty()) {
        Frame& f = st.back();
        const Rec& r = recs_[f.rec];
        if (r.next < 0 || f.state == 2) {
          branch_rec_.push_back(f.rec);
          st.pop_back();
        } else if (f.state == 0) {
          f.state = 1;
          st.push_back({recs_[r.next].back, 0});
        } else {
          f.state = 2;
          st.push_back({recs_[recs_[r.next].next].back, 0});
        }
      }
    }
  }
  if (branch_rec_.size() != nums_.branches)
    throw std::runtime_error{"Traversing the utree went wrong during pipeline startup!"};

  // link_tree_msa (src/core/pll/epa_pll_util.cpp:10-60)
  if (ref_msa.empty()) throw std::runtime_error{"empty reference MSA"};
  sites_ = ref_msa[0].sequence().size();
  std::unordered_map<std::string, size_t> by_label;
  for (size_t i = 0; i < ref_msa.size(); ++i) by_label.emplace(ref_msa[i].header(), i);
  // convenience beyond the reference: a reference sequence whose header carries a description
  // ("taxon1 len=1500") is also found under its first word
  for (size_t i = 0; i < ref_msa.size(); ++i) {
    const std::string& h = ref_msa[i].header();
    const size_t sp = h.find_first_of(" \t");
    if (sp != std::string::npos) by_label.emplace(h.substr(0, sp), i);
  }
  const int s = model_.num_states();
  std::unordered_map<uint32_t, uint8_t> code_of;
  if (s == 4) {
    tipmap_.resize(16);
    for (uint32_t m = 0; m < 16; ++m) { tipmap_[m] = m; code_of[m] = (uint8_t)m; }
  }
  tipchars_.resize(n);
  for (unsigned t = 0; t < n; ++t) {
    auto it = by_label.find(labels_[t]);
    if (it == by_label.end())
      throw std::invalid_argument{"Bad tree/ref msa combination: failed to find " + labels_[t]};
    const std::string& sq = ref_msa[it->second].sequence();
    if (sq.size() != sites_) throw std::runtime_error{"reference MSA rows differ in length"};
    tipchars_[t].resize(sites_);
    for (size_t w = 0; w < sites_; ++w) {
      const uint32_t m = model_.char_mask(sq[w]);
      if (!m) throw std::invalid_argument{"Bad sequence: illegal character in " + labels_[t]};
      auto c = code_of.find(m);
      if (c == code_of.end()) {
        if (tipmap_.size() >= 255) throw std::runtime_error{"too many distinct tip state sets"};
        c = code_of.emplace(m, (uint8_t)tipmap_.size()).first;
        tipmap_.push_back(m);
      }
      tipchars_[t][w] = c->second;
    }
  }
  if (model_.empirical_base_freqs()) {
    // +F / +FC: compute_and_set_empirical_frequencies (src/core/pll/epa_pll_util.cpp:55-57 ->
    // pllmod_msa_empirical_frequencies, pll-modules, restated): every tip character adds
    // 1 / |state set| to each state it allows (gaps and fully ambiguous characters spread
    // uniformly), normalised to sum 1
    std::vector<double> f(s, 0.0);
    for (unsigned t = 0; t < n; ++t)
      for (size_t w = 0; w < sites_; ++w) {
        const uint32_t m = tipmap_[tipchars_[t][w]];
        const double share = 1.0 / (double)__builtin_popcount(m);
        for (int i = 0; i < s; ++i)
          if ((m >> i) & 1u) f[i] += share;
      }
    model_.set_base_freqs(f);
  }
  // pll_update_invariant_sites over the reference tips (the array the tiny partition borrows,
  // src/tree/tiny_util.cpp:153-156): AND of the tips' state sets is a single state -> that state
  invariant_.assign(sites_, (int8_t)-1);
  for (size_t w = 0; w < sites_; ++w) {
    uint32_t all = s == 4 ? 15u : ((1u << 20) - 1);
    for (unsigned t = 0; t < n; ++t) all &= tipmap_[tipchars_[t][w]];
    if (all && !(all & (all - 1))) {
      int st = 0;
      while (!((all >> st) & 1u)) ++st;
      invariant_[w] = (int8_t)st;
    }
  }
  // precompute_clvs (src/core/pll/epa_pll_util.cpp:62-107): all three directions per inner node
  clv_.resize(recs_.size());
  scaler_.resize(recs_.size());
}

void Tree::ensure_host_clvs() const {
  if (host_clvs_ready_) return;
#pragma omp critical(epa_host_clvs)
  {
    if (!host_clvs_ready_) {
      const_cast<Tree*>(this)->compute_all_clvs();
      host_clvs_ready_ = true;
    }
  }
}

void Tree::fill_tree_desc(epa_tree_desc& d, Tree_Desc_Storage& st) const {
  std::memset(&d, 0, sizeof(d));
  const size_t B = nums_.branches, n = nums_.tip_nodes;
  // dense ids of the inner records
  std::vector<uint32_t> id(recs_.size(), 0xffffffffu);
  uint32_t R = 0;
  for (size_t r = 0; r < recs_.size(); ++r)
    if (recs_[r].tip < 0) id[r] = R++;
  auto operand = [&](int rec) -> uint32_t {
    return recs_[rec].tip >= 0 ? (EPA_TIP | (uint32_t)recs_[rec].tip) : id[rec];
  };
  st.child_a.assign(R, 0); st.child_b.assign(R, 0); st.len_a.assign(R, 0.0); st.len_b.assign(R, 0.0);
  for (size_t r = 0; r < recs_.size(); ++r) {
    if (recs_[r].tip >= 0) continue;
    const int n1 = recs_[r].next, n2 = recs_[n1].next;
    const int c1 = recs_[n1].back, c2 = recs_[n2].back;
    st.child_a[id[r]] = operand(c1); st.len_a[id[r]] = recs_[c1].length;
    st.child_b[id[r]] = operand(c2); st.len_b[id[r]] = recs_[c2].length;
  }
  st.prox.resize(B); st.dist.resize(B); st.blen.resize(B);
  for (size_t b = 0; b < B; ++b) {
    int dd = branch_rec_.at(b), p = recs_[dd].back;
    st.blen[b] = recs_[dd].length;
    if (recs_[dd].tip < 0 && recs_[p].tip >= 0) std::swap(dd, p);  // the tip is always DISTAL
    st.prox[b] = operand(p);
    st.dist[b] = operand(dd);
  }
  st.tipchars.resize(n * sites_);
  for (size_t t = 0; t < n; ++t) std::memcpy(&st.tipchars[t * sites_], tipchars_[t].data(), sites_);
  epa_ref_desc& rd = d.ref;
  rd.states = (uint32_t)model_.num_states();
  rd.rate_cats = (uint32_t)model_.num_ratecats();
  rd.sites = (uint32_t)sites_;
  rd.branches = (uint32_t)B;
  rd.eigenvals = model_.eigenvals().data();
  rd.eigenvecs_u = model_.eigenvecs_u().data();
  rd.eigenvecs_uinv = model_.eigenvecs_uinv().data();
  rd.freqs = model_.base_freqs().data();
  rd.rates = model_.ratecat_rates().data();
  rd.rate_weights = model_.ratecat_weights().data();
  rd.branch_length = st.blen.data();
  rd.prop_invar = model_.pinv();
  rd.invariant_state = model_.pinv() > 0.0 ? invariant_.data() : nullptr;
  rd.tipmap = tipmap_.data();
  rd.tipmap_size = (uint32_t)tipmap_.size();
  d.tips = (uint32_t)n;
  d.inner_records = R;
  d.tipchars = st.tipchars.data();
  d.rec_child_a = st.child_a.data(); d.rec_child_b = st.child_b.data();
  d.rec_length_a = st.len_a.data(); d.rec_length_b = st.len_b.data();
  d.branch_prox = st.prox.data(); d.branch_dist = st.dist.data();
}

void Tree::side(int rec, const double*& clv, const uint8_t*& tip, const uint32_t*& sc) const {
  const Rec& r = recs_[rec];
  if (r.tip >= 0) { clv = nullptr; tip = tipchars_[r.tip].data(); sc = nullptr; }
  else { clv = clv_[rec].data(); tip = nullptr; sc = scaler_[rec].data(); }
}

// All directional CLVs in ONE parallel region: the recursion is independent per alignment site,
// so the records are ordered once (children before parents), their P-matrices are formed once, and
// every thread walks the whole order for its own block of sites.  (A parallel loop per node costs
// a fork/join for 1500 sites of work, 3(n-2) times: seconds on a busy many-core host.)
void Tree::compute_all_clvs() {
  configure_host_threads();
  const int s = model_.num_states(), c = model_.num_ratecats();
  const size_t cs = (size_t)c * s;
  const int nrec = (int)recs_.size();
  // post-order over the dependency graph (explicit stack: caterpillar trees of thousands of tips
  // would overflow recursion)
  std::vector<int> order;
  order.reserve(nrec);
  std::vector<char> done(nrec, 0);
  for (int root = 0; root < nrec; ++root) {
    if (recs_[root].tip >= 0 || done[root]) continue;
    std::vector<int> st{root};
    while (!st.empty()) {
      const int rec = st.back();
      if (done[rec]) { st.pop_back(); continue; }
      const int c1 = recs_[recs_[rec].next].back, c2 = recs_[recs_[recs_[rec].next].next].back;
      const bool need1 = recs_[c1].tip < 0 && !done[c1];
      const bool need2 = recs_[c2].tip < 0 && !done[c2];
      if (need1 || need2) {
        if (need1) st.push_back(c1);
        if (need2) st.push_back(c2);
        continue;
      }
      st.pop_back();
      done[rec] = 1;
      order.push_back(rec);
    }
  }
  const size_t pm = cs * s;  // one P-matrix set (all categories)
  std::vector<double> P(order.size() * 2 * pm);
  for (size_t o = 0; o < order.size(); ++o) {
    const int rec = order[o];
    const int c1 = recs_[recs_[rec].next].back, c2 = recs_[recs_[recs_[rec].next].next].back;
    for (int k = 0; k < c; ++k) {
      model_.pmatrix(recs_[c1].length, k, &P[(o * 2) * pm + (size_t)k * s * s]);
      model_.pmatrix(recs_[c2].length, k, &P[(o * 2 + 1) * pm + (size_t)k * s * s]);
    }
    clv_[rec].assign(sites_ * cs, 0.0);
    scaler_[rec].assign(sites_, 0u);
  }
  const long block = 32;
  const long nblocks = ((long)sites_ + block - 1) / block;
#pragma omp parallel for schedule(dynamic)
  for (long blk = 0; blk < nblocks; ++blk) {
    const long w0 = blk * block, w1 = std::min<long>((long)sites_, w0 + block);
    for (size_t o = 0; o < order.size(); ++o) {
      const int rec = order[o];
      const int c1 = recs_[recs_[rec].next].back, c2 = recs_[recs_[recs_[rec].next].next].back;
      const double* P1 = &P[(o * 2) * pm];
      const double* P2 = &P[(o * 2 + 1) * pm];
      const double *v1, *v2;
      const uint8_t *t1, *t2;
      const uint32_t *s1, *s2;
      side(c1, v1, t1, s1);
      side(c2, v2, t2, s2);
      double* out = clv_[rec].data();
      uint32_t* sc = scaler_[rec].data();
      for (long w = w0; w < w1; ++w) {
        bool all_small = true;
        double* o_ = &out[(size_t)w * cs];
        const uint32_t m1 = t1 ? tipmap_[t1[w]] : 0, m2 = t2 ? tipmap_[t2[w]] : 0;
        for (int k = 0; k < c; ++k)
          for (int i = 0; i < s; ++i) {
            const double* r1 = &P1[((size_t)k * s + i) * s];
            const double* r2 = &P2[((size_t)k * s + i) * s];
            double a = 0.0, b = 0.0;
            if (t1) { for (int j = 0; j < s; ++j) if ((m1 >> j) & 1u) a += r1[j]; }
            else { const double* x = v1 + (size_t)w * cs + (size_t)k * s; for (int j = 0; j < s; ++j) a += r1[j] * x[j]; }
            if (t2) { for (int j = 0; j < s; ++j) if ((m2 >> j) & 1u) b += r2[j]; }
            else { const double* x = v2 + (size_t)w * cs + (size_t)k * s; for (int j = 0; j < s; ++j) b += r2[j] * x[j]; }
            const double v = a * b;
            o_[(size_t)k * s + i] = v;
            all_small = all_small && v < kScaleThreshold;
          }
        uint32_t cnt = (s1 ? s1[w] : 0) + (s2 ? s2[w] : 0);
        if (all_small) {
          for (size_t x = 0; x < cs; ++x) o_[x] *= kScaleFactor;
          ++cnt;
        }
        sc[w] = cnt;
      }
    }
  }
}

Tree::Branch Tree::branch(size_t b) const {
  ensure_host_clvs();
  int d = branch_rec_.at(b), p = recs_[d].back;
  const double len = recs_[d].length;
  // the reference tip is always DISTAL (src/tree/Tiny_Tree.cpp:64-74)
  if (recs_[d].tip < 0 && recs_[p].tip >= 0) std::swap(d, p);
  Branch br{};
  const uint8_t* ptip;
  side(p, br.prox_clv, ptip, br.prox_scaler);
  side(d, br.dist_clv, br.dist_tipchars, br.dist_scaler);
  br.length = len;
  return br;
}

void Tree::fill_desc(epa_ref_desc& d, std::vector<const double*>& pc,
                     std::vector<const uint32_t*>& ps, std::vector<const double*>& dc,
                     std::vector<const uint8_t*>& dt, std::vector<const uint32_t*>& ds,
                     std::vector<double>& bl) const {
  std::memset(&d, 0, sizeof(d));
  const size_t B = nums_.branches;
  pc.resize(B); ps.resize(B); dc.resize(B); dt.resize(B); ds.resize(B); bl.resize(B);
  for (size_t b = 0; b < B; ++b) {
    const Branch br = branch(b);
    pc[b] = br.prox_clv; ps[b] = br.prox_scaler;
    dc[b] = br.dist_clv; dt[b] = br.dist_tipchars; ds[b] = br.dist_scaler;
    bl[b] = br.length;
  }
  d.states = (uint32_t)model_.num_states();
  d.rate_cats = (uint32_t)model_.num_ratecats();
  d.sites = (uint32_t)sites_;
  d.branches = (uint32_t)B;
  d.eigenvals = model_.eigenvals().data();
  d.eigenvecs_u = model_.eigenvecs_u().data();
  d.eigenvecs_uinv = model_.eigenvecs_uinv().data();
  d.freqs = model_.base_freqs().data();
  d.rates = model_.ratecat_rates().data();
  d.rate_weights = model_.ratecat_weights().data();
  d.prop_invar = model_.pinv();
  d.invariant_state = model_.pinv() > 0.0 ? invariant_.data() : nullptr;
  d.prox_clv = pc.data(); d.prox_scaler = ps.data();
  d.dist_clv = dc.data(); d.dist_tipchars = dt.data(); d.dist_scaler = ds.data();
  d.branch_length = bl.data();
  d.tipmap = tipmap_.data();
  d.tipmap_size = (uint32_t)tipmap_.size();
}

double Tree::ref_tree_logl(size_t b) const {
  ensure_host_clvs();
  const int d = branch_rec_.at(b), p = recs_[d].back;
  const int s = model_.num_states(), c = model_.num_ratecats();
  const size_t cs = (size_t)c * s;
  std::vector<double> P(cs * s);
  for (int k = 0; k < c; ++k) model_.pmatrix(recs_[d].length, k, &P[(size_t)k * s * s]);
  const double *vp, *vd;
  const uint8_t *tp, *td;
  const uint32_t *sp, *sd;
  side(p, vp, tp, sp);
  side(d, vd, td, sd);
  const double log_thr = std::log(kScaleThreshold);
  double logl = 0.0;
  for (size_t w = 0; w < sites_; ++w) {
    double site = 0.0;
    for (int k = 0; k < c; ++k) {
      double cat = 0.0;
      for (int i = 0; i < s; ++i) {
        const double pv = tp ? (((tipmap_[tp[w]] >> i) & 1u) ? 1.0 : 0.0) : vp[w * cs + (size_t)k * s + i];
        if (pv == 0.0) continue;
        double t = 0.0;
        for (int j = 0; j < s; ++j) {
          const double dv = td ? (((tipmap_[td[w]] >> j) & 1u) ? 1.0 : 0.0) : vd[w * cs + (size_t)k * s + j];
          t += P[((size_t)k * s + i) * s + j] * dv;
        }
        cat += pv * model_.base_freqs()[i] * t;
      }
      site += cat * model_.ratecat_weights()[k];
    }
    if (model_.pinv() > 0.0)  // +I: (1 - p) L + p pi_inv (libpll, unscaled invariant term)
      site = site * (1.0 - model_.pinv()) +
             (invariant_[w] >= 0 ? model_.pinv() * model_.base_freqs()[invariant_[w]] : 0.0);
    const uint32_t cnt = (sp ? sp[w] : 0) + (sd ? sd[w] : 0);
    logl += std::log(site) + cnt * log_thr;
  }
  return logl;
}

std::pair<unsigned int, double> Rtree_Mapper::in_rtree(unsigned int branch_id, double distal_length) const {
  // rtree_mapper::in_rtree (src/core/pll/rtree_mapper.hpp:38-61; literals test/src/rtree_mapper.cpp:58-102)
  if (branch_id >= map_.size()) throw std::out_of_range{"Rtree_Mapper: branch id out of range"};
  if (branch_id == utree_root_edge) {
    if (distal_length > distal_edge_length) {
      // past the former root: the placement lands on the proximal rooted edge, whose direction
      // towards the root is flipped
      const double carryover = distal_length - distal_edge_length;
      return {rtree_proximal_edge, proximal_edge_length - carryover};
    }
    return {rtree_distal_edge, distal_length};
  }
  return {map_[branch_id], distal_length};
}

unsigned int Rtree_Mapper::map_at(size_t i) const {
  if (i == utree_root_edge)
    throw std::invalid_argument{"Edge " + std::to_string(i) + " is the root edge! Please handle separately"};
  return map_.at(i);
}

std::string Tree::numbered_newick(unsigned int precision) const {
  // get_numbered_newick_string (src/core/pll/pll_util.cpp:207-352); literal expectations in the
  // reference's test/src/pll_util.cpp:134-186 (unrooted, inner labels, rooted x3, rooted + labels).
  std::ostringstream ss;
  ss.precision(precision);
  ss.setf(std::ios::fixed, std::ios::floatfield);
  unsigned idx = 0;  // edge id on the UNROOTED tree
  const bool rooted = (bool)mapper_;
  auto edge_id = [&](unsigned i) { return rooted ? mapper_.map_at(i) : i; };
  struct Frame { int rec; int state; };
  auto emit = [&](int start) {
    std::vector<Frame> st{{start, 0}};
    while (!st.empty()) {
      Frame& f = st.back();
      const Rec& r = recs_[f.rec];
      if (r.next < 0) {
        ss << labels_[r.tip] << ":" << r.length << "{" << edge_id(idx) << "}";
        ++idx;
        st.pop_back();
      } else if (f.state == 0) {
        ss << "(";
        f.state = 1;
        st.push_back({recs_[r.next].back, 0});
      } else if (f.state == 1) {
        ss << ",";
        f.state = 2;
        st.push_back({recs_[recs_[r.next].next].back, 0});
      } else {
        ss << ")" << r.label << ":" << r.length << "{" << edge_id(idx) << "}";
        ++idx;
        st.pop_back();
      }
    }
  };
  const int r0 = recs_[vroot_].back, r1 = recs_[recs_[vroot_].next].back,
            r2 = recs_[recs_[recs_[vroot_].next].next].back;
  if (!rooted) {
    ss << "(";
    emit(r0);
    ss << ",";
    emit(r1);
    ss << ",";
    emit(r2);
    ss << ")" << root_label_ << ";";
    return ss.str();
  }
  // the rooted tree, simulated on the unrooted one
  ss << "(";
  if (mapper_.uroot_is_left()) {
    ss << "(";
    emit(r0);
    ss << ",";
    emit(r1);
    ss << ")" << root_label_ << ":" << mapper_.proximal_edge_length << "{" << mapper_.rtree_proximal_edge << "},";
    const Rec& right = recs_[r2];
    if (right.next < 0) {
      ss << labels_[right.tip];
    } else {
      ss << "(";
      emit(recs_[right.next].back);
      ss << ",";
      emit(recs_[recs_[right.next].next].back);
      ss << ")" << right.label;
    }
    ss << ":" << mapper_.distal_edge_length << "{" << mapper_.rtree_distal_edge << "}";
  } else {
    const Rec& lf = recs_[r0];  // the left root child, a tip: branch 0 of both trees
    ss << labels_[lf.tip] << ":" << mapper_.distal_edge_length << "{" << mapper_.rtree_distal_edge << "},";
    idx = 1;
    ss << "(";
    emit(r1);
    ss << ",";
    emit(r2);
    ss << ")" << root_label_ << ":" << mapper_.proximal_edge_length << "{" << mapper_.rtree_proximal_edge << "}";
  }
  ss << ")" << mapper_.root_label << ";";
  return ss.str();
}

}  // namespace epa
