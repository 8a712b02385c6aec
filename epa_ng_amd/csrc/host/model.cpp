// Substitution model: descriptor parsing (subset), eigen system, discrete-Gamma rates.
// Mirrors what raxml::Model + raxml::assign hand to libpll (src/core/raxml/Model.cpp:123-538,
// 711-733): the device only ever sees eigenvalues, U, U^-1, frequencies, rates and weights.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <sstream>

#include "epa_host.hpp"

namespace epa {

namespace {

const char* AA_STATE_ORDER = "ARNDCQEGHILKMFPSTWYV";  // libpll / PAML state order

// regularised lower incomplete gamma P(a, x)
double gamma_p(double a, double x) {
  if (x <= 0) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1.0) {  // power series
    double term = 1.0 / a, sum = term, ap = a;
    for (int i = 0; i < 100000; ++i) {
      ap += 1.0;
      term *= x / ap;
      sum += term;
      if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
    }
    return sum * std::exp(a * std::log(x) - x - lg);
  }
  // modified Lentz continued fraction for Q(a, x)
  const double tiny = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (std::fabs(d) < tiny) d = tiny;
    c = b + an / c;
    if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - std::exp(a * std::log(x) - x - lg) * h;
}

// symmetric eigen-decomposition, cyclic Jacobi; v columns are eigenvectors
void jacobi(int n, std::vector<double>& a, std::vector<double>& w, std::vector<double>& v) {
  v.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 200; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * x - s * y;
          a[(size_t)k * n + q] = s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
          const double x = a[(size_t)p * n + k], y = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * x - s * y;
          a[(size_t)q * n + k] = s * x + c * y;
        }
        for (int k = 0; k < n; ++k) {
          const double x = v[(size_t)k * n + p], y = v[(size_t)k * n + q];
          v[(size_t)k * n + p] = c * x - s * y;
          v[(size_t)k * n + q] = s * x + c * y;
        }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
}

std::vector<double> parse_list(const std::string& s) {
  std::vector<double> out;
  std::string tok;
  std::stringstream ss(s);
  while (std::getline(ss, tok, '/')) out.push_back(std::stod(tok));
  return out;
}

}  // namespace

std::vector<double> compute_gamma_cats(double alpha, int k, bool median) {
  // Yang 1994.  Quantiles of Gamma(alpha, rate alpha) by bisection on the regularised P.
  auto quantile = [&](double p) {
    double lo = 0.0, hi = 1.0;
    while (gamma_p(alpha, hi * alpha) < p) hi *= 2.0;
    for (int it = 0; it < 300 && hi - lo > 1e-17 * hi; ++it) {
      const double mid = 0.5 * (lo + hi);
      (gamma_p(alpha, mid * alpha) < p ? lo : hi) = mid;
    }
    return 0.5 * (lo + hi);
  };
  std::vector<double> cuts(k + 1, 0.0), rates(k);
  if (median) {
    // PLL_GAMMA_RATES_MEDIAN ("+G4a"): the medians of the k equal-probability slices, rescaled to
    // mean 1 (libpll's published algorithm, restated; not covered by a reference literal)
    double sum = 0.0;
    for (int i = 0; i < k; ++i) { rates[i] = quantile((2.0 * i + 1.0) / (2.0 * k)); sum += rates[i]; }
    for (double& r : rates) r *= k / sum;
    return rates;
  }
  // category means: the mean of a slice is K * [P(alpha+1, alpha*hi) - P(alpha+1, alpha*lo)]
  for (int i = 1; i < k; ++i) cuts[i] = quantile((double)i / k);
  double prev = 0.0;
  for (int i = 0; i < k; ++i) {
    const double cur = (i == k - 1) ? 1.0 : gamma_p(alpha + 1.0, cuts[i + 1] * alpha);
    rates[i] = (cur - prev) * k;
    prev = cur;
  }
  return rates;
}

Model::Model(int states, std::vector<double> subst, std::vector<double> freqs,
             std::vector<double> rates, std::vector<double> weights, double pinv)
    : states_(states), pinv_(pinv), subst_(std::move(subst)), freqs_(std::move(freqs)),
      rates_(std::move(rates)), weights_(std::move(weights)) {
  if (!(pinv_ >= 0.0 && pinv_ < 1.0)) throw std::runtime_error{"Model: p-inv must be in [0, 1)"};
  if ((int)subst_.size() != states * (states - 1) / 2 || (int)freqs_.size() != states)
    throw std::runtime_error{"Model: wrong number of substitution rates / frequencies"};
  if (weights_.empty()) weights_.assign(rates_.size(), 1.0 / rates_.size());
  update_eigen();
}

namespace {
// named nucleotide models: rate symmetries over AC AG AT CG CT GT and whether the model fixes equal
// base frequencies (the pll-modules model list the reference queries at Model.cpp:148-183)
struct Named_DNA { const char* name; const char* sym; bool equal_freqs; bool fixed_rates; };
const Named_DNA DNA_MODELS[] = {
    {"JC", "000000", true, true},     {"K80", "010010", true, false},    {"F81", "000000", false, true},
    {"HKY", "010010", false, false},  {"TN93EF", "010020", true, false}, {"TN93", "010020", false, false},
    {"K81", "012210", true, false},   {"K81UF", "012210", false, false}, {"TPM2", "010212", true, false},
    {"TPM2UF", "010212", false, false}, {"TPM3", "012012", true, false}, {"TPM3UF", "012012", false, false},
    {"TIM1", "012230", true, false},  {"TIM1UF", "012230", false, false}, {"TIM2", "010232", true, false},
    {"TIM2UF", "010232", false, false}, {"TIM3", "012032", true, false}, {"TIM3UF", "012032", false, false},
    {"TVMEF", "012314", true, false}, {"TVM", "012314", false, false},   {"SYM", "012345", true, false},
    {"GTR", "012345", false, false},  {"DNA", "012345", false, false},
};
}  // namespace

void Model::set_base_freqs(std::vector<double> freqs) {
  if ((int)freqs.size() != states_) throw std::runtime_error{"Model: wrong number of frequencies"};
  double sum = 0;
  for (double f : freqs) sum += f;
  for (double& f : freqs) f /= sum;
  freqs_ = std::move(freqs);
  update_eigen();
}

Model::Model(const std::string& descriptor_in) {
  // RAxML compatibility alias of the reference (Model.cpp:113-116)
  const std::string descriptor = descriptor_in == "DNA" ? "GTR+G+F" : descriptor_in;
  // name up to the first of "+{["
  size_t pos = descriptor.find_first_of("+{[");
  std::string name = descriptor.substr(0, pos);
  std::transform(name.begin(), name.end(), name.begin(), ::toupper);
  const Named_DNA* dna = nullptr;
  for (const auto& d : DNA_MODELS)
    if (name == d.name) dna = &d;
  const Named_AA_Model* aa = dna ? nullptr : named_aa_model(name);
  if (dna) states_ = 4;
  else if (aa || name == "PROTGTR") states_ = 20;
  else
    throw std::runtime_error{"Invalid model name: " + name};
  name_ = name;
  const int nr = states_ * (states_ - 1) / 2;
  // defaults of raxml::Model (Model.cpp:190-193,470,487-488): parameters a model does not fix
  // start at rates 0.5.. / 1.0 and equal frequencies (the optimiser's starting values; EPA-ng
  // never optimises them in the placement flow)
  subst_.assign(nr, 0.5);
  subst_.back() = 1.0;
  freqs_.assign(states_, 1.0 / states_);
  if (dna && dna->fixed_rates) subst_.assign(nr, 1.0);
  if (aa) {
    subst_.assign(aa->rates, aa->rates + 190);
    freqs_.assign(aa->freqs, aa->freqs + 20);
    double sum = 0;
    for (double f : freqs_) sum += f;
    for (double& f : freqs_) f /= sum;
  }
  int cats = 1;
  alpha_ = 1.0;
  bool gamma = false, free_rates = false, gamma_median = false;
  std::string rest = pos == std::string::npos ? "" : descriptor.substr(pos);
  size_t i = 0;
  auto braces = [&](std::string& out) {
    out.clear();
    if (i < rest.size() && rest[i] == '{') {
      size_t e = rest.find('}', i);
      if (e == std::string::npos) throw std::runtime_error{"Model: unbalanced '{'"};
      out = rest.substr(i + 1, e - i - 1);
      i = e + 1;
      return true;
    }
    return false;
  };
  std::string arg;
  if (braces(arg)) {
    // user rates: one value per class of the model's rate symmetry (Model.cpp:218-241)
    const std::vector<double> user = parse_list(arg);
    if (dna) {
      int nuniq = 0;
      for (const char* c = dna->sym; *c; ++c) nuniq = std::max(nuniq, *c - '0' + 1);
      if ((int)user.size() != nuniq)
        throw std::runtime_error{"Invalid number of substitution rates specified: " + std::to_string(user.size()) +
                                 " (expected: " + std::to_string(nuniq) + ")"};
      for (int r = 0; r < nr; ++r) subst_[r] = user[dna->sym[r] - '0'];
    } else {
      if ((int)user.size() != nr)
        throw std::runtime_error{"Invalid number of substitution rates specified: " + std::to_string(user.size()) +
                                 " (expected: " + std::to_string(nr) + ")"};
      subst_ = user;
    }
  }
  while (i < rest.size()) {
    if (rest[i] != '+') throw std::runtime_error{"Model: cannot parse '" + rest.substr(i) + "'"};
    ++i;
    size_t b = i;
    while (i < rest.size() && std::isalpha((unsigned char)rest[i])) ++i;
    std::string opt = rest.substr(b, i - b);
    std::transform(opt.begin(), opt.end(), opt.begin(), ::toupper);
    if (opt == "FU" || (opt == "F" && i < rest.size() && rest[i] == '{')) {
      if (!braces(arg)) throw std::runtime_error{"Model: +FU needs explicit frequencies, e.g. +FU{0.25/0.25/0.25/0.25}"};
      freqs_ = parse_list(arg);
      if ((int)freqs_.size() != states_)
        throw std::runtime_error{"Invalid number of user frequencies specified: " + std::to_string(freqs_.size())};
      double sum = 0;
      for (double f : freqs_) sum += f;
      for (double& f : freqs_) f /= sum;
      empirical_freqs_ = false;
    } else if (opt == "F" || opt == "FC") {
      // empirical (Model.cpp:302-306): counted on the reference MSA when the tree is linked to it
      // (link_tree_msa, src/core/pll/epa_pll_util.cpp:55-57); until then the current values stand
      empirical_freqs_ = true;
    } else if (opt == "FE" || opt == "FO") {
      // FE: equal; FO = ML estimate, whose starting value (all EPA-ng ever uses in the placement
      // flow) is equal frequencies as well (Model.cpp:466-471)
      freqs_.assign(states_, 1.0 / states_);
      empirical_freqs_ = false;
    } else if (opt == "G") {
      gamma = true;
      cats = 4;
      size_t nb = i;
      while (i < rest.size() && std::isdigit((unsigned char)rest[i])) ++i;
      if (i > nb) cats = std::stoi(rest.substr(nb, i - nb));
      // rate mode suffix of RAxML-NG descriptors (Model.cpp:390-396): m = category means
      // (default), a = medians
      if (i < rest.size() && (rest[i] == 'm' || rest[i] == 'M')) ++i;
      else if (i < rest.size() && (rest[i] == 'a' || rest[i] == 'A')) { ++i; gamma_median = true; }
      if (braces(arg)) alpha_ = std::stod(arg);
    } else if (opt == "R") {
      // free rates: +R<n>{r1/../rn}{w1/../wn} (Model.cpp:405-455): weights normalised to sum 1
      // (equal if omitted), then rates divided by sum_k w_k r_k
      free_rates = true;
      cats = 4;
      size_t nb = i;
      while (i < rest.size() && std::isdigit((unsigned char)rest[i])) ++i;
      if (i > nb) cats = std::stoi(rest.substr(nb, i - nb));
      if (!braces(arg)) throw std::runtime_error{"Model: +R needs explicit rates, e.g. +R4{0.1/0.5/1/3}{0.25/0.25/0.25/0.25}"};
      rates_ = parse_list(arg);
      if ((int)rates_.size() != cats)
        throw std::runtime_error{"Invalid number of free rates specified: " + std::to_string(rates_.size()) +
                                 " (expected: " + std::to_string(cats) + ")"};
      if (braces(arg)) {
        weights_ = parse_list(arg);
        if ((int)weights_.size() != cats)
          throw std::runtime_error{"Invalid number of rate weights specified: " + std::to_string(weights_.size()) +
                                   " (expected: " + std::to_string(cats) + ")"};
        double sum = 0;
        for (double w : weights_) sum += w;
        for (double& w : weights_) w /= sum;
      } else {
        weights_.assign(cats, 1.0 / cats);
      }
      double swr = 0;
      for (int k = 0; k < cats; ++k) swr += rates_[k] * weights_[k];
      for (double& r : rates_) r /= swr;
    } else if (opt == "I" || opt == "IU" || opt == "IO" || opt == "IC") {
      // +I{p} / +IU{p}: user-defined proportion of invariant sites (Model.cpp:355-375); +IO / +IC
      // (ML / empirical estimate) need the optimiser EPA-ng never runs: a value must be given
      if (!braces(arg)) throw std::runtime_error{"Model: +" + opt + " needs an explicit value, e.g. +I{0.1}"};
      pinv_ = std::stod(arg);
      if (!(pinv_ >= 0.0 && pinv_ < 1.0)) throw std::runtime_error{"Model: p-inv must be in [0, 1)"};
    } else {
      throw std::runtime_error{"Model: option +" + opt + " is not supported by this build"};
    }
  }
  if (gamma && free_rates) throw std::runtime_error{"Model: +G and +R are mutually exclusive"};
  if (!free_rates) {
    if (gamma) rates_ = compute_gamma_cats(alpha_, cats, gamma_median);
    else rates_.assign(1, 1.0);
    weights_.assign(rates_.size(), 1.0 / rates_.size());
  }
  free_rates_ = free_rates;
  update_eigen();
}

void Model::update_eigen() {
  const int s = states_;
  std::vector<double> q((size_t)s * s, 0.0);
  int k = 0;
  for (int i = 0; i < s; ++i)
    for (int j = i + 1; j < s; ++j) {
      q[(size_t)i * s + j] = subst_[k] * freqs_[j];
      q[(size_t)j * s + i] = subst_[k] * freqs_[i];
      ++k;
    }
  double mean = 0.0;
  for (int i = 0; i < s; ++i) {
    double d = 0.0;
    for (int j = 0; j < s; ++j)
      if (j != i) d += q[(size_t)i * s + j];
    q[(size_t)i * s + i] = -d;
    mean += freqs_[i] * d;
  }
  for (auto& v : q) v /= mean;
  std::vector<double> a((size_t)s * s), sq(s), w, v;
  for (int i = 0; i < s; ++i) sq[i] = std::sqrt(freqs_[i]);
  for (int i = 0; i < s; ++i)
    for (int j = 0; j < s; ++j) a[(size_t)i * s + j] = q[(size_t)i * s + j] * sq[i] / sq[j];
  for (int i = 0; i < s; ++i)
    for (int j = i + 1; j < s; ++j)
      a[(size_t)i * s + j] = a[(size_t)j * s + i] = 0.5 * (a[(size_t)i * s + j] + a[(size_t)j * s + i]);
  jacobi(s, a, w, v);
  eigenvals_ = w;
  u_.assign((size_t)s * s, 0.0);
  uinv_.assign((size_t)s * s, 0.0);
  for (int i = 0; i < s; ++i)
    for (int j = 0; j < s; ++j) {
      u_[(size_t)i * s + j] = v[(size_t)i * s + j] / sq[i];
      uinv_[(size_t)i * s + j] = v[(size_t)j * s + i] * sq[j];
    }
}

void Model::pmatrix(double t, int k, double* P) const {
  const int s = states_;
  std::vector<double> e(s);
  // +I: libpll divides the rate by (1 - p-inv) in every P-matrix
  for (int x = 0; x < s; ++x) e[x] = std::exp(eigenvals_[x] * rates_[k] * t / (1.0 - pinv_));
  for (int i = 0; i < s; ++i)
    for (int j = 0; j < s; ++j) {
      double acc = 0.0;
      for (int x = 0; x < s; ++x) acc += u_[(size_t)i * s + x] * e[x] * uinv_[(size_t)x * s + j];
      P[(size_t)i * s + j] = acc;
    }
}

uint32_t Model::char_mask(char ch) const {
  const int c = std::toupper((unsigned char)ch);
  if (states_ == 4) {
    switch (c) {
      case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
      case 'R': return 5; case 'Y': return 10; case 'S': return 6; case 'W': return 9;
      case 'K': return 12; case 'M': return 3; case 'B': return 14; case 'D': return 13;
      case 'H': return 11; case 'V': return 7;
      case 'N': case 'O': case 'X': case '-': case '?': case '.': return 15;
      default: return 0;
    }
  }
  if (c) {
    const char* p = std::strchr(AA_STATE_ORDER, c);
    if (p) return 1u << (p - AA_STATE_ORDER);
  }
  if (c == 'B') return (1u << 2) | (1u << 3);
  if (c == 'Z') return (1u << 5) | (1u << 6);
  if (c == 'X' || c == '-' || c == '?' || c == '*') return (1u << 20) - 1;
  return 0;
}

std::string Model::to_string() const {
  std::ostringstream s;
  s.precision(6);
  const bool generic = name_ == "GTR" || name_ == "DNA" || name_ == "PROTGTR";
  s << std::fixed << (generic ? (states_ == 4 ? "GTR" : "PROTGTR") : name_) << "{";
  if (generic || states_ == 20) {
    for (size_t i = 0; i < subst_.size(); ++i) s << (i ? "/" : "") << subst_[i];
  } else {  // named nucleotide model: one value per class of its rate symmetry
    for (const auto& d : DNA_MODELS)
      if (name_ == d.name) {
        int seen = 0;
        for (int r = 0; r < 6; ++r)
          if (d.sym[r] - '0' == seen) { s << (seen ? "/" : "") << subst_[r]; ++seen; }
      }
  }
  s << "}+FU{";
  for (size_t i = 0; i < freqs_.size(); ++i) s << (i ? "/" : "") << freqs_[i];
  s << "}";
  if (pinv_ > 0.0) s << "+IU{" << pinv_ << "}";
  if (free_rates_) {
    s << "+R" << rates_.size() << "{";
    for (size_t k = 0; k < rates_.size(); ++k) s << (k ? "/" : "") << rates_[k];
    s << "}{";
    for (size_t k = 0; k < weights_.size(); ++k) s << (k ? "/" : "") << weights_[k];
    s << "}";
  } else if (rates_.size() > 1) s << "+G" << rates_.size() << "{" << alpha_ << "}";
  return s.str();
}

}  // namespace epa
