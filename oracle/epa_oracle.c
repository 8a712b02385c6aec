/*
 * oracle/epa_oracle.c -- CPU restatement (plain C, fp64) of EPA-ng's placement hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library, and
 * there only as the checker / the timed CPU baseline -- never as a fallback for the HIP path.
 *
 * PARITY UNPINNED (vs. the real reference binary).  EPA-ng's arithmetic lives in two un-vendored
 * submodules that are absent from /root/reference: xflouris/libpll-2 @ 69411e4b and
 * ddarriba/pll-modules @ d46415e0 (reference CMakeLists.txt:130-131).  The reference therefore
 * cannot be built here and none of its tests holds a literal log-likelihood for this path
 * (SURVEY.md section 8c).  What this file does instead:
 *   - follows the *control flow and constants* of the reference files cited at each function
 *     (paths relative to /root/reference), and
 *   - restates the libpll/pll-modules routines from their published algorithms (Felsenstein
 *     pruning with 2^256 per-site scaling, eigen-decomposed GTR P-matrices, sumtable-based
 *     branch-length derivatives, rtsafe-style safeguarded Newton);
 *   - is pinned against an independent brute-force numpy/scipy evaluator (tests/gen_golden.py,
 *     scipy.linalg.expm on Q, whole-tree pruning with the query inserted) whose outputs are the
 *     committed fixtures tests/golden/*.json, and against the reference's own literal test
 *     vectors that do exist for this path (edge numbering: test/src/pll_util.cpp:134-143);
 *   - is anchored (edge numbers, winning edge, lnL to a few 1e-1 in 4400, lengths to a few 1e-3 --
 *     not a 1e-6 pin) on the one program output the reference checkout holds for this path:
 *     test/data/raxml_output.jplace, RAxML 8.2.4's EPA on the 10-taxon fixture
 *     (tests/test_external_anchor.py, tests/fit_raxml_anchor.py).
 *
 * Layout conventions (libpll's): CLV [site][cat][state], P-matrix
 * [cat][from][to], scaler uint32[site].  DNA state order A,C,G,T; tip codes are state bitmasks.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_S 20
#define ORC_MAX_C 16

/* constants that live in the missing pll headers (runtime parameters of the product's shim;
 * fixed here to the recollected libpll / pll-modules values) */
#define ORC_OPT_MIN_BRANCH_LEN 1.0e-4  /* PLLMOD_OPT_MIN_BRANCH_LEN */
#define ORC_OPT_MAX_BRANCH_LEN 100.0   /* PLLMOD_OPT_MAX_BRANCH_LEN */
#define ORC_OPT_DEFAULT_BRANCH_LEN 0.1 /* PLLMOD_OPT_DEFAULT_BRANCH_LEN */
#define ORC_OPT_BRANCH_EPSILON 1e-1    /* src/core/pll/optimize.hpp:9 */
#define ORC_DEFAULT_BRANCH_LENGTH (-log(0.9)) /* src/util/constants.hpp:12 */

static double orc_scale_factor(void) { return ldexp(1.0, 256); }     /* PLL_SCALE_FACTOR */
static double orc_scale_threshold(void) { return ldexp(1.0, -256); } /* PLL_SCALE_THRESHOLD */

/* ------------------------------------------------------------------------------------------
 * Model: GTR-family rate matrix -> eigen system (restates libpll pll_update_eigen; call sites
 * src/core/raxml/Model.cpp:711-733 `assign`).  P(t) = U diag(exp(lambda r t)) Uinv.
 * libpll calls U "inv_eigenvecs" and Uinv "eigenvecs".
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int s, c;
  double freqs[ORC_MAX_S];
  double evals[ORC_MAX_S];
  double u[ORC_MAX_S * ORC_MAX_S];    /* row-major s x s */
  double uinv[ORC_MAX_S * ORC_MAX_S]; /* row-major s x s */
  double rates[ORC_MAX_C];
  double weights[ORC_MAX_C];
  double pinv;
  int rate_scalers; /* PLL_ATTRIB_RATE_SCALERS: scaler arrays are [site][cat] (src/tree/tiny_util.cpp:37-44,
                     * auto-on for > 2000 tips, src/io/file_io.cpp:211-214) */
  uint64_t rounding_variant; /* 0 = this file's arithmetic as written.  != 0: a "faithfully rounded
                     * sibling" of it (orc_set_rounding_variant): every TERM of the sumtable's dot
                     * products and of the derivative contractions (what another summation order
                     * amounts to) and every site likelihood of the edge score is moved by
                     * -a / 0 / +a ulp (hash of the seed and the operand bits; a = 2^(bits 16..19 of the
                     * seed): sums of s products in another order differ by a few ulp, not one), and odd
                     * seeds sum the sites of f, f' from the far end.  It stands for ANY other correct evaluation order (another SIMD width,
                     * libm, reduction tree): tests use it to show that a pair on which the device and
                     * this oracle end at different lengths is one on which the oracle disagrees with
                     * itself. */
} orc_model;

/* -1 / 0 / +1 from a hash of (seed, salt, bits of v); v * (1 + sigma 2^-52) */
static inline double orc_ulp_noise(uint64_t seed, uint64_t salt, double v) {
  if (!seed) return v;
  uint64_t h;
  memcpy(&h, &v, 8);
  h ^= seed * 0x9E3779B97F4A7C15ull + salt * 0xBF58476D1CE4E5B9ull;
  h ^= h >> 31; h *= 0x94D049BB133111EBull; h ^= h >> 29;
  const int sg = (int)(h % 3) - 1;
  const int amp = 1 << ((seed >> 16) & 15); /* bits 16..19 of the seed: amplitude 2^0 .. 2^15 ulp */
  return sg ? v * (1.0 + sg * amp * 0x1p-52) : v;
}

/* libpll PLL_SCALE_RATE_MAXDIFF and the scale_minlh table (recollection, like the other pll
 * constants): with per-rate scalers a category whose count exceeds the site minimum by d is
 * multiplied by 2^(-256 min(d, 4)) when the categories are combined */
#define ORC_SCALE_RATE_MAXDIFF 4
static double orc_rate_scale_factor(uint32_t diff) {
  if (diff == 0) return 1.0;
  if (diff > ORC_SCALE_RATE_MAXDIFF) diff = ORC_SCALE_RATE_MAXDIFF;
  return ldexp(1.0, -256 * (int)diff);
}
/* scaler count of (site, cat) of one side: per-site array or per-rate array */
#define ORC_SC(sd, m, site, k) ((sd)->scaler ? ((m)->rate_scalers ? (sd)->scaler[(site) * (m)->c + (k)] : (sd)->scaler[(site)]) : 0u)

/* cyclic Jacobi for a symmetric n x n matrix; v columns = eigenvectors */
static void jacobi_sym(int n, double* a, double* w, double* v) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) v[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = a[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k) {
          double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = cs * akp - sn * akq;
          a[k * n + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; ++k) {
          double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = cs * apk - sn * aqk;
          a[q * n + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; ++k) {
          double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = cs * vkp - sn * vkq;
          v[k * n + q] = sn * vkp + cs * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = a[i * n + i];
}

/* subst: upper-triangular exchangeabilities, row-major (DNA: AC AG AT CG CT GT) */
int orc_model_init(orc_model* m, int s, const double* subst, const double* freqs, int c,
                   const double* rates, const double* weights, double pinv) {
  if (s > ORC_MAX_S || c > ORC_MAX_C) return -1;
  m->s = s;
  m->c = c;
  m->pinv = pinv;
  m->rate_scalers = 0;
  double q[ORC_MAX_S * ORC_MAX_S];
  int k = 0;
  for (int i = 0; i < s; ++i) {
    m->freqs[i] = freqs[i];
    for (int j = i + 1; j < s; ++j) {
      q[i * s + j] = subst[k] * freqs[j];
      q[j * s + i] = subst[k] * freqs[i];
      ++k;
    }
  }
  double mean = 0.0;
  for (int i = 0; i < s; ++i) {
    double d = 0.0;
    for (int j = 0; j < s; ++j)
      if (j != i) d += q[i * s + j];
    q[i * s + i] = -d;
    mean += freqs[i] * d;
  }
  for (int i = 0; i < s * s; ++i) q[i] /= mean;
  /* symmetrise with sqrt(pi): A = pi^1/2 Q pi^-1/2 */
  double a[ORC_MAX_S * ORC_MAX_S], v[ORC_MAX_S * ORC_MAX_S], sq[ORC_MAX_S];
  for (int i = 0; i < s; ++i) sq[i] = sqrt(freqs[i]);
  for (int i = 0; i < s; ++i)
    for (int j = 0; j < s; ++j) a[i * s + j] = q[i * s + j] * sq[i] / sq[j];
  for (int i = 0; i < s; ++i)
    for (int j = i + 1; j < s; ++j) {
      double av = 0.5 * (a[i * s + j] + a[j * s + i]);
      a[i * s + j] = a[j * s + i] = av;
    }
  jacobi_sym(s, a, m->evals, v);
  for (int i = 0; i < s; ++i)
    for (int j = 0; j < s; ++j) {
      m->u[i * s + j] = v[i * s + j] / sq[i];    /* U    = pi^-1/2 V   */
      m->uinv[i * s + j] = v[j * s + i] * sq[j]; /* Uinv = V^T pi^1/2  */
    }
  for (int i = 0; i < c; ++i) {
    m->rates[i] = rates[i];
    m->weights[i] = weights[i];
  }
  return 0;
}

/* regularised lower incomplete gamma P(a,x) (series / continued fraction) */
static double inc_gamma_p(double a, double x) {
  if (x <= 0.0) return 0.0;
  double gln = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 10000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 10000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c;
    if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - exp(-x + a * log(x) - gln) * h;
}

/* Yang-1994 discrete gamma, category means (restates libpll pll_compute_gamma_cats with
 * PLL_GAMMA_RATES_MEAN; call site src/core/raxml/Model.cpp:514). alpha = beta. */
void orc_gamma_rates(double alpha, int K, double* rates) {
  double cut[ORC_MAX_C + 1];
  cut[0] = 0.0;
  for (int i = 1; i < K; ++i) {
    double p = (double)i / K, lo = 0.0, hi = 1.0;
    while (inc_gamma_p(alpha, hi * alpha) < p) hi *= 2.0;
    for (int it = 0; it < 200; ++it) {
      double mid = 0.5 * (lo + hi);
      if (inc_gamma_p(alpha, mid * alpha) < p) lo = mid; else hi = mid;
    }
    cut[i] = 0.5 * (lo + hi); /* quantile of Gamma(shape alpha, rate alpha) */
  }
  double prev = 0.0;
  for (int i = 0; i < K; ++i) {
    double cur = (i == K - 1) ? 1.0 : inc_gamma_p(alpha + 1.0, cut[i + 1] * alpha);
    rates[i] = (cur - prev) * K;
    prev = cur;
  }
}

/* restates libpll pll_update_prob_matrices for one branch (call sites
 * src/tree/Tiny_Tree.cpp:105-109, src/core/pll/optimize.cpp:34-37,165,209,
 * src/core/pll/pll_util.cpp:381-384).  P[k][i][j], t==0 -> identity. */
void orc_pmatrix(const orc_model* m, double t, double* P) {
  const int s = m->s;
  for (int k = 0; k < m->c; ++k) {
    double* Pk = P + (size_t)k * s * s;
    if (t == 0.0) {
      for (int i = 0; i < s; ++i)
        for (int j = 0; j < s; ++j) Pk[i * s + j] = (i == j) ? 1.0 : 0.0;
      continue;
    }
    double e[ORC_MAX_S], tmp[ORC_MAX_S * ORC_MAX_S];
    for (int j = 0; j < s; ++j) {
      if (m->pinv > 1e-12)
        e[j] = exp(m->evals[j] * m->rates[k] * t / (1.0 - m->pinv));
      else
        e[j] = exp(m->evals[j] * m->rates[k] * t);
    }
    for (int i = 0; i < s; ++i)
      for (int j = 0; j < s; ++j) tmp[i * s + j] = m->u[i * s + j] * e[j];
    for (int i = 0; i < s; ++i)
      for (int j = 0; j < s; ++j) {
        double acc = 0.0;
        for (int x = 0; x < s; ++x) acc += tmp[i * s + x] * m->uinv[x * s + j];
        Pk[i * s + j] = acc;
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * Character maps.  Tip codes are state bitmasks (restates libpll pll_map_nt / pll_map_aa,
 * selected at src/core/raxml/Model.cpp:9-22).  Lookup-column order = NT_MAP / AA_MAP
 * (src/util/maps.hpp:9-31); ASCII -> column normalisation = Lookup_Store ctor
 * (src/core/Lookup_Store.hpp:33-68).
 * ---------------------------------------------------------------------------------------- */
static const char NT_MAP[16] = {'-', 'T', 'G', 'K', 'C', 'Y', 'S', 'B',
                                'A', 'W', 'R', 'D', 'M', 'H', 'V', 'N'};
static const char AA_MAP[24] = {'A', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'K', 'L', 'M', 'N',
                                'P', 'Q', 'R', 'S', 'T', 'V', 'W', 'Y', '-', 'X', 'B', 'Z'};
/* libpll AA state order */
static const char AA_STATES[21] = "ARNDCQEGHILKMFPSTWYV";

uint32_t orc_char_mask(int s, char ch) {
  ch = (char)toupper((unsigned char)ch);
  if (s == 4) {
    switch (ch) {
      case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': case 'U': return 8;
      case 'R': return 5; case 'Y': return 10; case 'S': return 6; case 'W': return 9;
      case 'K': return 12; case 'M': return 3; case 'B': return 14; case 'D': return 13;
      case 'H': return 11; case 'V': return 7;
      case 'N': case 'O': case 'X': case '-': case '?': case '.': return 15;
      default: return 0;
    }
  }
  const char* p = strchr(AA_STATES, ch);
  if (ch && p) return 1u << (p - AA_STATES);
  if (ch == 'B') return (1u << 2) | (1u << 3); /* N | D */
  if (ch == 'Z') return (1u << 5) | (1u << 6); /* Q | E */
  if (ch == 'X' || ch == '-' || ch == '?' || ch == '*') return (1u << 20) - 1;
  return 0;
}

int orc_num_columns(int s) { return s == 4 ? 16 : 24; }
char orc_column_char(int s, int col) { return s == 4 ? NT_MAP[col] : AA_MAP[col]; }

/* ASCII -> lookup column, or -1 (src/core/Lookup_Store.hpp:33-68,100-108).
 * aa_x_quirk: reproduce quirk D4 (AA 'X' scored in the 'N' column). */
int orc_char_column(int s, char ch, int aa_x_quirk) {
  const int dna = (s == 4);
  int up = toupper((unsigned char)ch);
  if (dna) {
    if (up == 'U') up = 'T';
    if (up == 'X' || up == 'O' || up == '.') up = '-';
  } else if (up == 'X' && aa_x_quirk) {
    up = 'N';
  }
  if (up == '?') up = '-';
  const char* map = dna ? NT_MAP : AA_MAP;
  const int n = dna ? 16 : 24;
  for (int i = 0; i < n; ++i)
    if (map[i] == up) return i;
  return -1;
}

/* ------------------------------------------------------------------------------------------
 * Unrooted tree as libpll-style unode records (tips: 1 record, inner nodes: ring of 3).
 * Parsing order restates pll_utree_parse_newick: "(X,Y,Z);" -> vroot->back=X,
 * vroot->next->back=Y, vroot->next->next->back=Z; inner "(X,Y)": next->back=X,
 * next->next->back=Y.  Branch enumeration = utree_query_branches
 * (src/core/pll/pll_util.cpp:182-205).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int next, back;
  double length;
  int tip; /* tip index or -1 */
  int clv; /* directional clv id: tips = tip index, inner = n_tips + running */
} orc_rec;

typedef struct {
  int n_tips, n_recs, cap;
  orc_rec* r;
  char** labels; /* per tip */
  int vroot;
  int B;
  int* branch_rec; /* distal-side record per branch id */
} orc_tree;

static int new_rec(orc_tree* t) {
  if (t->n_recs == t->cap) {
    t->cap = t->cap ? t->cap * 2 : 64;
    t->r = (orc_rec*)realloc(t->r, sizeof(orc_rec) * t->cap);
  }
  orc_rec* x = &t->r[t->n_recs];
  x->next = -1; x->back = -1; x->length = 0.0; x->tip = -1; x->clv = -1;
  return t->n_recs++;
}

static void skip_ws(const char** p) { while (**p && isspace((unsigned char)**p)) ++*p; }

static void parse_label_len(const char** p, char** label_out, double* len) {
  skip_ws(p);
  const char* b = *p;
  while (**p && !strchr(":,();", **p) && !isspace((unsigned char)**p)) ++*p;
  if (label_out) {
    size_t n = (size_t)(*p - b);
    *label_out = (char*)malloc(n + 1);
    memcpy(*label_out, b, n);
    (*label_out)[n] = 0;
  }
  skip_ws(p);
  *len = 0.0;
  if (**p == ':') {
    ++*p;
    char* e;
    *len = strtod(*p, &e);
    *p = e;
  }
  skip_ws(p);
}

static void connect(orc_tree* t, int a, int b, double len) {
  t->r[a].back = b; t->r[b].back = a;
  t->r[a].length = len; t->r[b].length = len;
}

/* returns the record that faces the parent; its branch length is returned via *len */
static int parse_subtree(orc_tree* t, const char** p, double* len, int* err) {
  skip_ws(p);
  if (**p == '(') {
    ++*p;
    int kids[3]; double kl[3]; int nk = 0;
    for (;;) {
      if (nk == 3) { *err = 1; return -1; } /* multifurcation */
      kids[nk] = parse_subtree(t, p, &kl[nk], err);
      if (*err) return -1;
      ++nk;
      skip_ws(p);
      if (**p == ',') { ++*p; continue; }
      if (**p == ')') { ++*p; break; }
      *err = 2; return -1;
    }
    if (nk != 2) { *err = 1; return -1; }
    int a = new_rec(t), b = new_rec(t), c = new_rec(t);
    t->r[a].next = b; t->r[b].next = c; t->r[c].next = a;
    connect(t, b, kids[0], kl[0]);
    connect(t, c, kids[1], kl[1]);
    parse_label_len(p, NULL, len);
    return a;
  }
  int a = new_rec(t);
  t->r[a].tip = t->n_tips;
  t->labels = (char**)realloc(t->labels, sizeof(char*) * (t->n_tips + 1));
  parse_label_len(p, &t->labels[t->n_tips], len);
  t->n_tips++;
  return a;
}

static void query_branches_rec(orc_tree* t, int node, int* idx) {
  if (t->r[node].next >= 0) {
    query_branches_rec(t, t->r[t->r[node].next].back, idx);
    query_branches_rec(t, t->r[t->r[t->r[node].next].next].back, idx);
  }
  t->branch_rec[(*idx)++] = node;
}

void orc_tree_free(orc_tree* t) {
  if (!t) return;
  for (int i = 0; i < t->n_tips; ++i) free(t->labels[i]);
  free(t->labels); free(t->r); free(t->branch_rec); free(t);
}

/* Only unrooted (top-level trifurcation) input; rooted input is SURVEY section 8f-4. */
orc_tree* orc_tree_parse(const char* newick) {
  orc_tree* t = (orc_tree*)calloc(1, sizeof(orc_tree));
  const char* p = newick;
  int err = 0;
  skip_ws(&p);
  if (*p != '(') { orc_tree_free(t); return NULL; }
  ++p;
  int kids[3]; double kl[3];
  for (int i = 0; i < 3; ++i) {
    kids[i] = parse_subtree(t, &p, &kl[i], &err);
    if (err) { orc_tree_free(t); return NULL; }
    skip_ws(&p);
    if (i < 2) { if (*p != ',') { orc_tree_free(t); return NULL; } ++p; }
  }
  if (*p != ')') { orc_tree_free(t); return NULL; }
  int a = new_rec(t), b = new_rec(t), c = new_rec(t);
  t->r[a].next = b; t->r[b].next = c; t->r[c].next = a;
  connect(t, a, kids[0], kl[0]);
  connect(t, b, kids[1], kl[1]);
  connect(t, c, kids[2], kl[2]);
  t->vroot = a;
  /* set_missing_branch_lengths (src/core/pll/pll_util.cpp:13-39): zero == missing */
  for (int i = 0; i < t->n_recs; ++i)
    if (t->r[i].length == 0.0) t->r[i].length = ORC_DEFAULT_BRANCH_LENGTH;
  int ninner = 0;
  for (int i = 0; i < t->n_recs; ++i)
    t->r[i].clv = (t->r[i].tip >= 0) ? t->r[i].tip : t->n_tips + ninner++;
  t->B = 2 * t->n_tips - 3;
  t->branch_rec = (int*)malloc(sizeof(int) * t->B);
  int idx = 0;
  query_branches_rec(t, t->r[t->vroot].back, &idx);
  query_branches_rec(t, t->r[t->r[t->vroot].next].back, &idx);
  query_branches_rec(t, t->r[t->r[t->r[t->vroot].next].next].back, &idx);
  if (idx != t->B) { orc_tree_free(t); return NULL; }
  return t;
}

/* numbered newick (restates get_numbered_newick_string, src/core/pll/pll_util.cpp:207-259;
 * literal expectations test/src/pll_util.cpp:134-143).  No inner labels kept. */
static void numbered_rec(const orc_tree* t, int node, char** w, int* idx, int prec) {
  if (t->r[node].next >= 0) {
    *w += sprintf(*w, "(");
    numbered_rec(t, t->r[t->r[node].next].back, w, idx, prec);
    *w += sprintf(*w, ",");
    numbered_rec(t, t->r[t->r[t->r[node].next].next].back, w, idx, prec);
    *w += sprintf(*w, "):%.*f{%d}", prec, t->r[node].length, *idx);
  } else {
    *w += sprintf(*w, "%s:%.*f{%d}", t->labels[t->r[node].tip], prec, t->r[node].length, *idx);
  }
  ++*idx;
}

int orc_tree_numbered_newick(const orc_tree* t, int prec, char* out, int cap) {
  char* buf = (char*)malloc((size_t)t->n_recs * 96 + 64);
  char* w = buf;
  int idx = 0;
  w += sprintf(w, "(");
  numbered_rec(t, t->r[t->vroot].back, &w, &idx, prec);
  w += sprintf(w, ",");
  numbered_rec(t, t->r[t->r[t->vroot].next].back, &w, &idx, prec);
  w += sprintf(w, ",");
  numbered_rec(t, t->r[t->r[t->r[t->vroot].next].next].back, &w, &idx, prec);
  w += sprintf(w, ");");
  int n = (int)(w - buf);
  if (n + 1 > cap) { free(buf); return -n; }
  memcpy(out, buf, (size_t)n + 1);
  free(buf);
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Likelihood kernels (restated libpll routines)
 * ---------------------------------------------------------------------------------------- */
typedef struct {            /* one side of an operation: inner CLV or tip chars */
  const double* clv;        /* [W][c][s] or NULL */
  const uint32_t* tipmask;  /* [W] state bitmasks or NULL */
  const uint32_t* scaler;   /* [W] or NULL */
} orc_side;

/* sum_j P[i][j] * x_j for a side at (site, cat) */
static inline double side_term(const orc_side* sd, const double* Prow, int s, size_t site, int k,
                               int c) {
  double acc = 0.0;
  if (sd->clv) {
    const double* x = sd->clv + (site * c + k) * s;
    for (int j = 0; j < s; ++j) acc += Prow[j] * x[j];
  } else {
    uint32_t mk = sd->tipmask[site];
    for (int j = 0; j < s; ++j) {
      if (mk & 1u) acc += Prow[j];
      mk >>= 1;
    }
  }
  return acc;
}


/* ------------------------------------------------------------------------------------------
 * ORC_FAST_KERNELS (the timed `cpu_baseline` build of bench.py only; the parity build never defines it):
 * the four hot loops specialised for 4 states x 4 rate categories, per-site scalers, no +I, written the
 * way libpll's AVX / AVX2 kernels are -- the 4 states of a (site, category) are one vector, a
 * matrix-vector product is four broadcast-multiply-adds over it, matrices are kept transposed for that --
 * with fixed trip counts and `omp simd` so that -O3 -march=native turns them into vector code.  Same
 * algorithm and data layout as the generic loops below them; sums are associated per state instead of
 * per row, so results agree with the parity build to rounding (bench.py checks 1e-6 on the sample).
 * ---------------------------------------------------------------------------------------- */
#ifdef ORC_FAST_KERNELS
static inline int orc_fast_ok(const orc_model* m) {
  return m->s == 4 && m->c == 4 && !m->rate_scalers && !m->rounding_variant && m->pinv == 0.0;
}
typedef struct { double v[4]; } orc_v4;
static inline orc_v4 side_vec(const orc_side* sd, size_t site, int k) {
  orc_v4 x;
  if (sd->clv) {
    const double* p = sd->clv + (site * 4 + k) * 4;
#pragma omp simd
    for (int j = 0; j < 4; ++j) x.v[j] = p[j];
  } else {
    const uint32_t mk = sd->tipmask[site];
#pragma omp simd
    for (int j = 0; j < 4; ++j) x.v[j] = (double)((mk >> j) & 1u);
  }
  return x;
}
/* y[i] = sum_j MT[j][i] x[j]  (MT = the matrix transposed: four broadcast FMAs over a 4-vector) */
static inline orc_v4 matvec_t(const double* MT, orc_v4 x) {
  orc_v4 y;
#pragma omp simd
  for (int i = 0; i < 4; ++i) y.v[i] = MT[i] * x.v[0];
  for (int j = 1; j < 4; ++j) {
#pragma omp simd
    for (int i = 0; i < 4; ++i) y.v[i] += MT[j * 4 + i] * x.v[j];
  }
  return y;
}
static inline void transpose_p(const double* P, double* PT) {   /* P[k][i][j] -> PT[k][j][i] */
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) PT[(k * 4 + j) * 4 + i] = P[(k * 4 + i) * 4 + j];
}
static void update_partial_fast(const orc_side* l, const double* Pl, const orc_side* r, const double* Pr,
                                double* parent, uint32_t* parent_sc, size_t b, size_t n) {
  const double thr = orc_scale_threshold(), fac = orc_scale_factor();
  double PlT[64], PrT[64];
  transpose_p(Pl, PlT);
  transpose_p(Pr, PrT);
  for (size_t site = b; site < b + n; ++site) {
    double* p = parent + site * 16;
    double mx = 0.0;
    for (int k = 0; k < 4; ++k) {
      const orc_v4 ta = matvec_t(PlT + k * 16, side_vec(l, site, k)), tb = matvec_t(PrT + k * 16, side_vec(r, site, k));
#pragma omp simd reduction(max : mx)
      for (int i = 0; i < 4; ++i) {
        const double v = ta.v[i] * tb.v[i];
        p[k * 4 + i] = v;
        mx = v > mx ? v : mx;
      }
    }
    uint32_t sc = (l->scaler ? l->scaler[site] : 0) + (r->scaler ? r->scaler[site] : 0);
    if (mx < thr) {
#pragma omp simd
      for (int x = 0; x < 16; ++x) p[x] *= fac;
      sc += 1;
    }
    parent_sc[site] = sc;
  }
}
static double edge_lnl_fast(const orc_model* m, const orc_side* par, const orc_side* ch, const double* P,
                            double* persite, size_t b, size_t n) {
  const double log_thr = log(orc_scale_threshold());
  double PT[64], logl = 0.0;
  transpose_p(P, PT);
  for (size_t site = b; site < b + n; ++site) {
    double terma = 0.0;
    for (int k = 0; k < 4; ++k) {
      const orc_v4 tb = matvec_t(PT + k * 16, side_vec(ch, site, k)), pv = side_vec(par, site, k);
      double acc = 0.0;
#pragma omp simd reduction(+ : acc)
      for (int i = 0; i < 4; ++i) acc += pv.v[i] * m->freqs[i] * tb.v[i];
      terma += acc * m->weights[k];
    }
    const uint32_t sc = (par->scaler ? par->scaler[site] : 0) + (ch->scaler ? ch->scaler[site] : 0);
    double site_lk = log(terma);
    if (sc) site_lk += sc * log_thr;
    if (persite) persite[site] = site_lk;
    logl += site_lk;
  }
  return logl;
}
static void update_sumtable_fast(const orc_model* m, const orc_side* A, const orc_side* Bs, double* S, size_t b, size_t n) {
  double piU[16], UiT[16];   /* piU[i][j] = pi_i U[i][j];  UiT[i][j] = Uinv[j][i] */
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { piU[i * 4 + j] = m->freqs[i] * m->u[i * 4 + j]; UiT[i * 4 + j] = m->uinv[j * 4 + i]; }
  for (size_t site = b; site < b + n; ++site)
    for (int k = 0; k < 4; ++k) {
      const orc_v4 lt = matvec_t(piU, side_vec(A, site, k)), rt = matvec_t(UiT, side_vec(Bs, site, k));
      double* o = S + ((site - b) * 4 + k) * 4;
#pragma omp simd
      for (int j = 0; j < 4; ++j) o[j] = lt.v[j] * rt.v[j];
    }
}
static void lk_derivatives_fast(const orc_model* m, const double* S, size_t n, double t, double* d1, double* d2) {
  double d0[16], dg1[16], dg2[16];   /* weights folded in */
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 4; ++j) {
      const double lr = m->evals[j] * m->rates[k], e0 = exp(lr * t) * m->weights[k];
      d0[k * 4 + j] = e0; dg1[k * 4 + j] = lr * e0; dg2[k * 4 + j] = lr * lr * e0;
    }
  double f = 0.0, df = 0.0;
  for (size_t x = 0; x < n; ++x) {
    const double* sm = S + x * 16;
    double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma omp simd reduction(+ : l0, l1, l2)
    for (int j = 0; j < 16; ++j) { l0 += sm[j] * d0[j]; l1 += sm[j] * dg1[j]; l2 += sm[j] * dg2[j]; }
    const double dv1 = -l1 / l0;
    f += dv1;
    df += dv1 * dv1 - l2 / l0;
  }
  *d1 = f;
  *d2 = df;
}
/* ---- the same four loops for 20 states x 4 categories (the cfg3 CPU baseline): libpll's AVX2 shape for amino
 * acids -- a matrix-vector product accumulates whole columns of the TRANSPOSED matrix scaled by one entry of
 * the vector (`y[0..19] += MT[j][0..19] * x[j]`: five 4-wide FMAs per j, no horizontal adds), a tip child
 * contributes only the columns of its set states. */
/* (explicit alignment: gcc 11 at -O3 -march=native stores 32-byte vectors to these locals with aligned moves) */
#define ORC_AL __attribute__((aligned(64)))
static inline int orc_fast20_ok(const orc_model* m) {
  return m->s == 20 && m->c == 4 && !m->rate_scalers && !m->rounding_variant && m->pinv == 0.0;
}
static inline void matvec_t20(const double* MT, const orc_side* sd, size_t site, int k, double* y) {
#pragma omp simd
  for (int i = 0; i < 20; ++i) y[i] = 0.0;
  if (sd->clv) {
    const double* x = sd->clv + (site * 4 + k) * 20;
    for (int j = 0; j < 20; ++j) {
      const double xj = x[j];
#pragma omp simd
      for (int i = 0; i < 20; ++i) y[i] += MT[j * 20 + i] * xj;
    }
  } else {
    uint32_t mk = sd->tipmask[site] & 0xfffffu;
    for (int j = 0; mk; ++j, mk >>= 1)
      if (mk & 1u) {
#pragma omp simd
        for (int i = 0; i < 20; ++i) y[i] += MT[j * 20 + i];
      }
  }
}
static inline void transpose_p20(const double* P, double* PT) {   /* P[k][i][j] -> PT[k][j][i] */
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 20; ++i)
      for (int j = 0; j < 20; ++j) PT[(k * 20 + j) * 20 + i] = P[(k * 20 + i) * 20 + j];
}
static void update_partial_fast20(const orc_side* l, const double* Pl, const orc_side* r, const double* Pr,
                                  double* parent, uint32_t* parent_sc, size_t b, size_t n) {
  const double thr = orc_scale_threshold(), fac = orc_scale_factor();
  ORC_AL double PlT[1600]; ORC_AL double PrT[1600];
  transpose_p20(Pl, PlT);
  transpose_p20(Pr, PrT);
  for (size_t site = b; site < b + n; ++site) {
    double* p = parent + site * 80;
    double mx = 0.0;
    for (int k = 0; k < 4; ++k) {
      ORC_AL double ta[20]; ORC_AL double tb[20];
      matvec_t20(PlT + k * 400, l, site, k, ta);
      matvec_t20(PrT + k * 400, r, site, k, tb);
#pragma omp simd reduction(max : mx)
      for (int i = 0; i < 20; ++i) {
        const double v = ta[i] * tb[i];
        p[k * 20 + i] = v;
        mx = v > mx ? v : mx;
      }
    }
    uint32_t sc = (l->scaler ? l->scaler[site] : 0) + (r->scaler ? r->scaler[site] : 0);
    if (mx < thr) {
#pragma omp simd
      for (int x = 0; x < 80; ++x) p[x] *= fac;
      sc += 1;
    }
    parent_sc[site] = sc;
  }
}
static double edge_lnl_fast20(const orc_model* m, const orc_side* par, const orc_side* ch, const double* P,
                              double* persite, size_t b, size_t n) {
  const double log_thr = log(orc_scale_threshold());
  ORC_AL double PT[1600]; double logl = 0.0;
  transpose_p20(P, PT);
  for (size_t site = b; site < b + n; ++site) {
    double terma = 0.0;
    for (int k = 0; k < 4; ++k) {
      ORC_AL double tb[20]; double acc = 0.0;
      matvec_t20(PT + k * 400, ch, site, k, tb);
      if (par->clv) {
        const double* pv = par->clv + (site * 4 + k) * 20;
#pragma omp simd reduction(+ : acc)
        for (int i = 0; i < 20; ++i) acc += pv[i] * m->freqs[i] * tb[i];
      } else {
        uint32_t mk = par->tipmask[site] & 0xfffffu;
        for (int i = 0; mk; ++i, mk >>= 1)
          if (mk & 1u) acc += m->freqs[i] * tb[i];
      }
      terma += acc * m->weights[k];
    }
    const uint32_t sc = (par->scaler ? par->scaler[site] : 0) + (ch->scaler ? ch->scaler[site] : 0);
    double site_lk = log(terma);
    if (sc) site_lk += sc * log_thr;
    if (persite) persite[site] = site_lk;
    logl += site_lk;
  }
  return logl;
}
static void update_sumtable_fast20(const orc_model* m, const orc_side* A, const orc_side* Bs, double* S, size_t b, size_t n) {
  ORC_AL double piU[400]; ORC_AL double UiT[400];   /* piU[i][j] = pi_i U[i][j];  UiT[i][j] = Uinv[j][i] */
  for (int i = 0; i < 20; ++i)
    for (int j = 0; j < 20; ++j) { piU[i * 20 + j] = m->freqs[i] * m->u[i * 20 + j]; UiT[i * 20 + j] = m->uinv[j * 20 + i]; }
  for (size_t site = b; site < b + n; ++site)
    for (int k = 0; k < 4; ++k) {
      ORC_AL double lt[20]; ORC_AL double rt[20];
      matvec_t20(piU, A, site, k, lt);
      matvec_t20(UiT, Bs, site, k, rt);
      double* o = S + ((site - b) * 4 + k) * 20;
#pragma omp simd
      for (int j = 0; j < 20; ++j) o[j] = lt[j] * rt[j];
    }
}
static void lk_derivatives_fast20(const orc_model* m, const double* S, size_t n, double t, double* d1, double* d2) {
  ORC_AL double d0[80]; ORC_AL double dg1[80]; ORC_AL double dg2[80];   /* weights folded in */
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 20; ++j) {
      const double lr = m->evals[j] * m->rates[k], e0 = exp(lr * t) * m->weights[k];
      d0[k * 20 + j] = e0; dg1[k * 20 + j] = lr * e0; dg2[k * 20 + j] = lr * lr * e0;
    }
  double f = 0.0, df = 0.0;
  for (size_t x = 0; x < n; ++x) {
    const double* sm = S + x * 80;
    double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma omp simd reduction(+ : l0, l1, l2)
    for (int j = 0; j < 80; ++j) { l0 += sm[j] * d0[j]; l1 += sm[j] * dg1[j]; l2 += sm[j] * dg2[j]; }
    const double dv1 = -l1 / l0;
    f += dv1;
    df += dv1 * dv1 - l2 / l0;
  }
  *d1 = f;
  *d2 = df;
}
#endif /* ORC_FAST_KERNELS */

/* restates pll_update_partials for one op over sites [b, b+n)
 * (call sites src/tree/Tiny_Tree.cpp:112,203; src/core/pll/optimize.cpp:41,173,217;
 *  src/core/pll/epa_pll_util.cpp:105).  Per-site scaling: all c*s entries < 2^-256. */
static void update_partial(const orc_model* m, const orc_side* l, const double* Pl,
                           const orc_side* r, const double* Pr, double* parent,
                           uint32_t* parent_sc, size_t b, size_t n) {
#ifdef ORC_FAST_KERNELS
  if (orc_fast_ok(m)) { update_partial_fast(l, Pl, r, Pr, parent, parent_sc, b, n); return; }
  if (orc_fast20_ok(m)) { update_partial_fast20(l, Pl, r, Pr, parent, parent_sc, b, n); return; }
#endif
  const int s = m->s, c = m->c;
  const double thr = orc_scale_threshold(), fac = orc_scale_factor();
  for (size_t site = b; site < b + n; ++site) {
    int scaling = 1;
    double* p = parent + site * c * s;
    for (int k = 0; k < c; ++k) {
      int rate_scaling = 1;
      for (int i = 0; i < s; ++i) {
        double ta = side_term(l, Pl + ((size_t)k * s + i) * s, s, site, k, c);
        double tb = side_term(r, Pr + ((size_t)k * s + i) * s, s, site, k, c);
        double v = ta * tb;
        p[k * s + i] = v;
        rate_scaling = rate_scaling && (v < thr);
      }
      scaling = scaling && rate_scaling;
      if (m->rate_scalers) { /* per-rate scaling: every category on its own */
        uint32_t sc = ORC_SC(l, m, site, k) + ORC_SC(r, m, site, k);
        if (rate_scaling) {
          for (int i = 0; i < s; ++i) p[k * s + i] *= fac;
          sc += 1;
        }
        parent_sc[site * c + k] = sc;
      }
    }
    if (m->rate_scalers) continue;
    uint32_t sc = (l->scaler ? l->scaler[site] : 0) + (r->scaler ? r->scaler[site] : 0);
    if (scaling) {
      for (int x = 0; x < c * s; ++x) p[x] *= fac;
      sc += 1;
    }
    parent_sc[site] = sc;
  }
}

/* per-rate scalers: the site's minimum count over the categories and the alignment factor of
 * every category (restates the rate_scalings logic of libpll's core_edge_loglikelihood /
 * core_update_sumtable) */
static uint32_t rate_alignment(const orc_model* m, const orc_side* a, const orc_side* b2, size_t site,
                               double* factor) {
  uint32_t mn = 0xffffffffu, cnt[ORC_MAX_C];
  for (int k = 0; k < m->c; ++k) {
    cnt[k] = ORC_SC(a, m, site, k) + ORC_SC(b2, m, site, k);
    if (cnt[k] < mn) mn = cnt[k];
  }
  for (int k = 0; k < m->c; ++k) factor[k] = orc_rate_scale_factor(cnt[k] - mn);
  return mn;
}

/* restates pll_compute_edge_loglikelihood over sites [b, b+n) (call sites
 * src/tree/Tiny_Tree.cpp:39-41, src/core/pll/optimize.cpp:111-113,219-222, src/tree/Tree.cpp:126).
 * parent side carries pi; P applies to the child side.  persite may be NULL. */
static double edge_lnl(const orc_model* m, const orc_side* par, const orc_side* ch,
                       const double* P, const int8_t* invariant, double* persite, size_t b,
                       size_t n) {
#ifdef ORC_FAST_KERNELS
  if (orc_fast_ok(m)) return edge_lnl_fast(m, par, ch, P, persite, b, n);
  if (orc_fast20_ok(m)) return edge_lnl_fast20(m, par, ch, P, persite, b, n);
#endif
  const int s = m->s, c = m->c;
  const double log_thr = log(orc_scale_threshold());
  double logl = 0.0;
  for (size_t site = b; site < b + n; ++site) {
    double terma = 0.0;
    double rfac[ORC_MAX_C];
    uint32_t rmin = 0;
    if (m->rate_scalers) rmin = rate_alignment(m, par, ch, site, rfac);
    for (int k = 0; k < c; ++k) {
      double terma_r = 0.0;
      for (int i = 0; i < s; ++i) {
        double pv;
        if (par->clv) pv = par->clv[(site * c + k) * s + i];
        else pv = ((par->tipmask[site] >> i) & 1u) ? 1.0 : 0.0;
        if (pv == 0.0) continue;
        double termb = side_term(ch, P + ((size_t)k * s + i) * s, s, site, k, c);
        terma_r += pv * m->freqs[i] * termb;
      }
      if (m->rate_scalers) terma_r *= rfac[k];
      if (m->pinv > 0.0) {
        double inv = (invariant && invariant[site] >= 0) ? m->freqs[invariant[site]] : 0.0;
        terma += m->weights[k] * (terma_r * (1.0 - m->pinv) + inv * m->pinv);
      } else {
        terma += terma_r * m->weights[k];
      }
    }
    uint32_t sc = m->rate_scalers ? rmin
                                  : (par->scaler ? par->scaler[site] : 0) + (ch->scaler ? ch->scaler[site] : 0);
    if (m->rounding_variant) terma = orc_ulp_noise(m->rounding_variant, site, terma);
    double site_lk = log(terma);
    if (sc) site_lk += sc * log_thr;
    if (persite) persite[site] = site_lk;
    logl += site_lk;
  }
  return logl;
}

/* restates pll_update_sumtable (call sites src/core/pll/optimize.cpp:147-149,190-192):
 * S[site][k][j] = (sum_i pi_i A_i U[i][j]) * (sum_m Uinv[j][m] B_m) */
static void update_sumtable(const orc_model* m, const orc_side* A, const orc_side* Bs, double* S,
                            size_t b, size_t n) {
#ifdef ORC_FAST_KERNELS
  if (orc_fast_ok(m)) { update_sumtable_fast(m, A, Bs, S, b, n); return; }
  if (orc_fast20_ok(m)) { update_sumtable_fast20(m, A, Bs, S, b, n); return; }
#endif
  const int s = m->s, c = m->c;
  for (size_t site = b; site < b + n; ++site) {
    double rfac[ORC_MAX_C];
    if (m->rate_scalers) (void)rate_alignment(m, A, Bs, site, rfac);
    for (int k = 0; k < c; ++k)
      for (int j = 0; j < s; ++j) {
        double lt = 0.0, rt = 0.0;
        for (int i = 0; i < s; ++i) {
          double av = A->clv ? A->clv[(site * c + k) * s + i]
                             : (((A->tipmask[site] >> i) & 1u) ? 1.0 : 0.0);
          double bv = Bs->clv ? Bs->clv[(site * c + k) * s + i]
                              : (((Bs->tipmask[site] >> i) & 1u) ? 1.0 : 0.0);
          if (m->rounding_variant) { /* noise on every TERM: what another summation order amounts to */
            lt += orc_ulp_noise(m->rounding_variant, (site * 64 + k * 20 + j) * 41 + i, av * m->freqs[i] * m->u[i * s + j]);
            rt += orc_ulp_noise(m->rounding_variant, (site * 64 + k * 20 + j) * 43 + i, m->uinv[j * s + i] * bv);
          } else {
            lt += av * m->freqs[i] * m->u[i * s + j];
            rt += m->uinv[j * s + i] * bv;
          }
        }
        double sv = m->rate_scalers ? lt * rt * rfac[k] : lt * rt;
        S[((site - b) * c + k) * s + j] = sv;
      }
  }
}

/* restates pll_compute_likelihood_derivatives (via utree_derivative_func,
 * src/core/pll/optimize.cpp:44-49): first/second derivative of -lnL w.r.t. t */
static void lk_derivatives(const orc_model* m, const double* S, size_t n, double t,
                           const int8_t* invariant, size_t b, double* d1, double* d2) {
#ifdef ORC_FAST_KERNELS
  if (orc_fast_ok(m)) { lk_derivatives_fast(m, S, n, t, d1, d2); return; }
  if (orc_fast20_ok(m)) { lk_derivatives_fast20(m, S, n, t, d1, d2); return; }
#endif
  const int s = m->s, c = m->c;
  double dg[ORC_MAX_C * ORC_MAX_S * 3];
  /* rounding variants with bit 11 / bit 12 set: the eigenvalue of the stationary mode -- 0 in exact
   * arithmetic, a few 1e-17 of either sign out of any numerical eigen-solver -- taken as exactly 0 /
   * with the opposite sign.  At saturated lengths (every other exp() underflown) its residue IS f. */
  int jz = -1;
  if (m->rounding_variant & 0x1800) {
    jz = 0;
    for (int j = 1; j < s; ++j) if (fabs(m->evals[j]) < fabs(m->evals[jz])) jz = j;
  }
  for (int k = 0; k < c; ++k) {
    double ki = m->rates[k] / (1.0 - m->pinv);
    for (int j = 0; j < s; ++j) {
      double ev = m->evals[j];
      if (j == jz) ev = (m->rounding_variant & 0x800) ? 0.0 : -ev;
      double e0 = exp(ev * ki * t);
      dg[(k * s + j) * 3 + 0] = e0;
      dg[(k * s + j) * 3 + 1] = ev * ki * e0;
      dg[(k * s + j) * 3 + 2] = ev * ki * ev * ki * e0;
    }
  }
  double f = 0.0, df = 0.0;
  for (size_t xi = 0; xi < n; ++xi) {
    const size_t x = (m->rounding_variant & 1) ? n - 1 - xi : xi;
    double l0 = 0, l1 = 0, l2 = 0;
    for (int k = 0; k < c; ++k) {
      double c0 = 0, c1 = 0, c2 = 0;
      const double* sm = S + (x * c + k) * s;
      if (m->rounding_variant) { /* noise on every term of the three contractions */
        for (int j = 0; j < s; ++j) {
          c0 += orc_ulp_noise(m->rounding_variant, 3 * j, sm[j] * dg[(k * s + j) * 3 + 0]);
          c1 += orc_ulp_noise(m->rounding_variant, 3 * j + 1, sm[j] * dg[(k * s + j) * 3 + 1]);
          c2 += orc_ulp_noise(m->rounding_variant, 3 * j + 2, sm[j] * dg[(k * s + j) * 3 + 2]);
        }
      } else
      for (int j = 0; j < s; ++j) {
        c0 += sm[j] * dg[(k * s + j) * 3 + 0];
        c1 += sm[j] * dg[(k * s + j) * 3 + 1];
        c2 += sm[j] * dg[(k * s + j) * 3 + 2];
      }
      if (m->pinv > 0.0) {
        double inv = (invariant && invariant[b + x] >= 0) ? m->freqs[invariant[b + x]] * m->pinv : 0.0;
        c0 = c0 * (1.0 - m->pinv) + inv;
        c1 *= (1.0 - m->pinv);
        c2 *= (1.0 - m->pinv);
      }
      l0 += c0 * m->weights[k];
      l1 += c1 * m->weights[k];
      l2 += c2 * m->weights[k];
    }
    double dv1 = -l1 / l0;
    double dv2 = dv1 * dv1 - l2 / l0;
    f += dv1;
    df += dv2;
  }
  *d1 = f;
  *d2 = df;
}

/* restates pll-modules pllmod_opt_minimize_newton (source not in tree; recollected rtsafe-style
 * safeguarded Newton; call sites src/core/pll/optimize.cpp:157-158,200-201).  Returns the new
 * branch length, NAN on non-finite derivatives / iteration overflow. */
typedef struct {
  const orc_model* m; const double* S; size_t n; const int8_t* inv; size_t b;
  long n_evals;
  int variant;
} nr_ctx;

/* optional trace of one pair's optimisation (orc_trace_pair): rows of 4 doubles
 *   {1 | 2, t, f, f'}   a derivative evaluation of the pendant (1) / distal (2) solve
 *   {3, new -lnL, old -lnL, reverted}   the score at the end of a round (optimize.cpp:217-232) */
typedef struct { double* rows; long cap, n; int phase; } orc_trace;
static __thread orc_trace* g_trace = NULL;
static inline void trace_row(double kind, double a, double b, double c) {
  if (!g_trace) return;
  if (g_trace->n < g_trace->cap) {
    double* r = g_trace->rows + 4 * g_trace->n;
    r[0] = kind; r[1] = a; r[2] = b; r[3] = c;
  }
  g_trace->n++;
}
#define TRACE_EVAL() do { if (g_trace) trace_row(g_trace->phase, rts, f, df); } while (0)

static double minimize_newton(double x1, double xguess, double x2, double tol, int max_iters,
                              nr_ctx* cx) {
  double rts = xguess, f, df, xl, xh, dx, dxold;
  if (rts < x1) rts = x1;
  if (rts > x2) rts = x2;
  lk_derivatives(cx->m, cx->S, cx->n, rts, cx->inv, cx->b, &f, &df); cx->n_evals++;
  TRACE_EVAL();
  if (!isfinite(f) || !isfinite(df)) return NAN;
  if (((cx->variant & 2) ? df > 0.0 : df >= 0.0) && fabs(f) < tol) return rts;
  if (f < 0.0) { xl = rts; xh = x2; } else { xh = rts; xl = x1; }
  dx = dxold = fabs(xh - xl);
  for (int i = 1; i <= max_iters; ++i) {
    const int slow = (cx->variant & 1) && fabs(2.0 * f) > fabs(dxold * df); /* Numerical Recipes rtsafe */
    dxold = dx;
    if (df <= 0.0 || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0) || slow) {
      dx = 0.5 * (xh - xl);
      rts = xl + dx;
      if (xl == rts) return rts;
    } else {
      dx = f / df;
      double temp = rts;
      rts -= dx;
      if (temp == rts) return rts;
    }
    if (fabs(dx) < tol || i == max_iters) return rts;
    if (rts < x1) rts = x1;
    lk_derivatives(cx->m, cx->S, cx->n, rts, cx->inv, cx->b, &f, &df); cx->n_evals++;
    TRACE_EVAL();
    if (!isfinite(f) || !isfinite(df)) return NAN;
    if (df > 0.0 && fabs(f) < tol) return rts;
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return NAN;
}

/* ------------------------------------------------------------------------------------------
 * Reference tree with all directional CLVs (restates Tree::Tree src/tree/Tree.cpp:16-56,
 * precompute_clvs src/core/pll/epa_pll_util.cpp:62-107)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  orc_model m;
  orc_tree* t;
  size_t W;
  uint32_t** tipmask; /* per tip [W] */
  double** clv;       /* per record (NULL for tips) [W][c][s] */
  uint32_t** scaler;  /* per record (NULL for tips) */
  int8_t* invariant;  /* [W] or NULL */
  int aa_x_quirk;
  /* constants of the branch-length optimiser: recollected pll-modules values by default, runtime
   * parameters so that their blast radius can be measured (orc_set_blo, tests/sensitivity.py) */
  double blo_min, blo_max, blo_default, blo_eps;
  int raxml_blo;      /* 1: --raxml-blo (optimize.cpp:274-279) instead of the sliding rule */
  int newton_variant; /* bit 0: rtsafe's "slow convergence -> bisect" clause; bit 1: strict df > 0
                       * in the first convergence test (0 = the recollected routine) */
  double** store; /* Lookup_Store: per-branch [W][C], lazily filled */
#ifdef _OPENMP
  omp_lock_t* locks;
#endif
} orc_ctx;

static void side_of(const orc_ctx* x, int rec, orc_side* sd) {
  const orc_rec* r = &x->t->r[rec];
  if (r->tip >= 0) { sd->clv = NULL; sd->tipmask = x->tipmask[r->tip]; sd->scaler = NULL; }
  else { sd->clv = x->clv[rec]; sd->tipmask = NULL; sd->scaler = x->scaler[rec]; }
}

static void compute_clv(orc_ctx* x, int rec) {
  orc_rec* r = &x->t->r[rec];
  if (r->tip >= 0 || x->clv[rec]) return;
  int c1 = x->t->r[r->next].back, c2 = x->t->r[x->t->r[r->next].next].back;
  compute_clv(x, c1);
  compute_clv(x, c2);
  const int s = x->m.s, c = x->m.c;
  double* P1 = (double*)malloc(sizeof(double) * c * s * s * 2);
  double* P2 = P1 + (size_t)c * s * s;
  orc_pmatrix(&x->m, x->t->r[c1].length, P1);
  orc_pmatrix(&x->m, x->t->r[c2].length, P2);
  x->clv[rec] = (double*)malloc(sizeof(double) * x->W * c * s);
  x->scaler[rec] = (uint32_t*)malloc(sizeof(uint32_t) * x->W * (x->m.rate_scalers ? c : 1));
  orc_side a, b;
  side_of(x, c1, &a);
  side_of(x, c2, &b);
  update_partial(&x->m, &a, P1, &b, P2, x->clv[rec], x->scaler[rec], 0, x->W);
  free(P1);
}

void orc_destroy(orc_ctx* x) {
  if (!x) return;
  if (x->t) {
    for (int i = 0; i < x->t->n_tips; ++i) if (x->tipmask) free(x->tipmask[i]);
    for (int i = 0; i < x->t->n_recs; ++i) { if (x->clv) free(x->clv[i]); if (x->scaler) free(x->scaler[i]); }
  }
  if (x->store && x->t) for (int b = 0; b < x->t->B; ++b) free(x->store[b]);
  free(x->store);
#ifdef _OPENMP
  free(x->locks);
#endif
  free(x->tipmask); free(x->clv); free(x->scaler); free(x->invariant);
  orc_tree_free(x->t);
  free(x);
}

/* per-rate scalers (PLL_ATTRIB_RATE_SCALERS) for the NEXT orc_create: the reference turns them on
 * for trees with more than 2000 tips (src/io/file_io.cpp:211-214) or with --rate-scalers */
static int g_next_rate_scalers = 0;
void orc_next_create_rate_scalers(int on) { g_next_rate_scalers = on; }

/* labels/seqs: the reference MSA (n_seqs rows of width W, ASCII).  Returns NULL on error. */
orc_ctx* orc_create(const char* newick, int n_seqs, const char** labels, const char** seqs,
                    size_t W, int s, const double* subst, const double* freqs, int c,
                    const double* rates, const double* weights, double pinv) {
  orc_ctx* x = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  if (orc_model_init(&x->m, s, subst, freqs, c, rates, weights, pinv)) { free(x); return NULL; }
  x->t = orc_tree_parse(newick);
  if (!x->t) { free(x); return NULL; }
  x->W = W;
  x->blo_min = ORC_OPT_MIN_BRANCH_LEN;
  x->blo_max = ORC_OPT_MAX_BRANCH_LEN;
  x->blo_default = ORC_OPT_DEFAULT_BRANCH_LEN;
  x->blo_eps = ORC_OPT_BRANCH_EPSILON;
  x->m.rate_scalers = g_next_rate_scalers;
  x->tipmask = (uint32_t**)calloc(x->t->n_tips, sizeof(uint32_t*));
  x->clv = (double**)calloc(x->t->n_recs, sizeof(double*));
  x->scaler = (uint32_t**)calloc(x->t->n_recs, sizeof(uint32_t*));
  /* link_tree_msa (src/core/pll/epa_pll_util.cpp:10-60) */
  for (int i = 0; i < x->t->n_tips; ++i) {
    int found = -1;
    for (int k = 0; k < n_seqs; ++k)
      if (!strcmp(labels[k], x->t->labels[i])) { found = k; break; }
    if (found < 0) { orc_destroy(x); return NULL; }
    x->tipmask[i] = (uint32_t*)malloc(sizeof(uint32_t) * W);
    for (size_t w = 0; w < W; ++w) {
      uint32_t mk = orc_char_mask(s, seqs[found][w]);
      if (!mk) { orc_destroy(x); return NULL; }
      x->tipmask[i][w] = mk;
    }
  }
  if (pinv > 0.0) { /* restates libpll pll_update_invariant_sites over the reference tips */
    x->invariant = (int8_t*)malloc(W);
    for (size_t w = 0; w < W; ++w) {
      uint32_t all = (s == 4) ? 15u : ((1u << 20) - 1);
      for (int i = 0; i < x->t->n_tips; ++i) all &= x->tipmask[i][w];
      int st = -1;
      if (all && !(all & (all - 1))) { st = 0; while (!((all >> st) & 1u)) ++st; }
      x->invariant[w] = (int8_t)st;
    }
  }
  for (int i = 0; i < x->t->n_recs; ++i) compute_clv(x, i);
  return x;
}

int orc_num_branches(const orc_ctx* x) { return x->t->B; }
int orc_num_tips(const orc_ctx* x) { return x->t->n_tips; }
size_t orc_width(const orc_ctx* x) { return x->W; }
void orc_set_aa_x_quirk(orc_ctx* x, int on) { x->aa_x_quirk = on; }
int orc_rate_scalers(const orc_ctx* x) { return x->m.rate_scalers; }
void orc_set_raxml_blo(orc_ctx* x, int on) { x->raxml_blo = on; }
/* optimiser constants / Newton variant (0 keeps a value); see the fields of orc_ctx */
void orc_set_blo(orc_ctx* x, double mn, double mx, double def, double eps, int newton_variant) {
  if (mn > 0) x->blo_min = mn;
  if (mx > 0) x->blo_max = mx;
  if (def > 0) x->blo_default = def;
  if (eps > 0) x->blo_eps = eps;
  if (newton_variant >= 0) x->newton_variant = newton_variant;
}
const double* orc_model_evals(const orc_ctx* x) { return x->m.evals; }
const double* orc_model_u(const orc_ctx* x) { return x->m.u; }
const double* orc_model_uinv(const orc_ctx* x) { return x->m.uinv; }
int orc_numbered_newick(const orc_ctx* x, int prec, char* out, int cap) {
  return orc_tree_numbered_newick(x->t, prec, out, cap);
}

/* edge lnL of the reference tree at branch b (restates Tree::ref_tree_logl,
 * src/tree/Tree.cpp:119-131, generalised to any edge: invariant (i) of SURVEY 8c) */
double orc_tree_lnl(const orc_ctx* x, int b) {
  int d = x->t->branch_rec[b], p = x->t->r[d].back;
  const int s = x->m.s, c = x->m.c;
  double* P = (double*)malloc(sizeof(double) * c * s * s);
  orc_pmatrix(&x->m, x->t->r[d].length, P);
  orc_side sp, sd;
  side_of(x, p, &sp);
  side_of(x, d, &sd);
  double l = edge_lnl(&x->m, &sp, &sd, P, x->invariant, NULL, 0, x->W);
  free(P);
  return l;
}

/* ------------------------------------------------------------------------------------------
 * Tiny tree (restates Tiny_Tree::Tiny_Tree src/tree/Tiny_Tree.cpp:48-129 and the slot logic of
 * src/tree/tiny_util.cpp:234-306): proximal / distal sides borrowed from the reference tree,
 * inner CLV + 3 P-matrices owned.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const orc_ctx* x;
  int branch;
  orc_side prox, dist;
  double orig;
  double len_prox, len_dist, len_pend;
  double *P_prox, *P_dist, *P_pend;
  double* inner;
  uint32_t* inner_sc;
  double* sumtable;
} orc_tiny;

/* reset_triplet_lengths (src/core/pll/pll_util.cpp:354-386) */
static void tiny_reset_lengths(orc_tiny* tt) {
  tt->len_pend = ORC_DEFAULT_BRANCH_LENGTH;
  tt->len_prox = tt->orig / 2.0;
  tt->len_dist = tt->orig / 2.0;
  orc_pmatrix(&tt->x->m, tt->len_prox, tt->P_prox);
  orc_pmatrix(&tt->x->m, tt->len_dist, tt->P_dist);
  orc_pmatrix(&tt->x->m, tt->len_pend, tt->P_pend);
}

static void tiny_free(orc_tiny* tt) {
  if (!tt) return;
  free(tt->P_prox); free(tt->inner); free(tt->inner_sc); free(tt->sumtable); free(tt);
}

static orc_tiny* tiny_create(const orc_ctx* x, int b) {
  orc_tiny* tt = (orc_tiny*)calloc(1, sizeof(orc_tiny));
  const int s = x->m.s, c = x->m.c;
  tt->x = x;
  tt->branch = b;
  int d = x->t->branch_rec[b], p = x->t->r[d].back;
  tt->orig = x->t->r[d].length;
  /* tip-tip detection: the reference tip is always DISTAL (src/tree/Tiny_Tree.cpp:64-74) */
  if (x->t->r[d].tip < 0 && x->t->r[p].tip >= 0) { int tmp = d; d = p; p = tmp; }
  side_of(x, p, &tt->prox);
  side_of(x, d, &tt->dist);
  size_t psz = (size_t)c * s * s;
  tt->P_prox = (double*)malloc(sizeof(double) * psz * 3);
  tt->P_dist = tt->P_prox + psz;
  tt->P_pend = tt->P_dist + psz;
  tt->inner = (double*)malloc(sizeof(double) * x->W * c * s);
  tt->inner_sc = (uint32_t*)malloc(sizeof(uint32_t) * x->W * (x->m.rate_scalers ? c : 1));
  tt->sumtable = (double*)malloc(sizeof(double) * x->W * c * s);
  tiny_reset_lengths(tt);
  /* inner <- distal (x) proximal (src/tree/Tiny_Tree.cpp:88-112) */
  update_partial(&x->m, &tt->dist, tt->P_dist, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc, 0,
                 x->W);
  return tt;
}

/* per-branch lookup table T[site][col] (restates precompute_sites_static +
 * Lookup_Store::init_branch, src/tree/Tiny_Tree.cpp:18-46,114-128,
 * src/core/Lookup_Store.hpp:73-81) */
static void tiny_build_lookup(const orc_tiny* tt, double* T) {
  const orc_ctx* x = tt->x;
  const int s = x->m.s, C = orc_num_columns(s);
  uint32_t* mk = (uint32_t*)malloc(sizeof(uint32_t) * x->W);
  double* ps = (double*)malloc(sizeof(double) * x->W);
  orc_side tip = {NULL, mk, NULL};
  orc_side in = {tt->inner, NULL, tt->inner_sc};
  for (int col = 0; col < C; ++col) {
    uint32_t v = orc_char_mask(s, orc_column_char(s, col));
    for (size_t w = 0; w < x->W; ++w) mk[w] = v;
    edge_lnl(&x->m, &tip, &in, tt->P_pend, x->invariant, ps, 0, x->W);
    for (size_t w = 0; w < x->W; ++w) T[w * C + col] = ps[w];
  }
  free(mk); free(ps);
}

/* get_valid_range (src/util/Range.hpp:34-49): only the literal '-' counts */
static void valid_range(const char* q, size_t W, size_t* begin, size_t* span) {
  size_t lo = 0, hi = W;
  while (lo < hi && q[lo] == '-') ++lo;
  while (hi > lo && q[hi - 1] == '-') --hi;
  *begin = lo;
  *span = hi - lo;
}

/* sum_precomputed_sitelk (src/core/Lookup_Store.hpp:110-141), same association order */
static double lookup_sum(const double* T, int C, const int* cols, size_t begin, size_t span) {
  double sum = 0.0;
  size_t site = begin, end = begin + span;
  for (; site + 3 < end; site += 4) {
    double s1 = T[site * C + cols[site]] + T[(site + 1) * C + cols[site + 1]];
    double s2 = T[(site + 2) * C + cols[site + 2]] + T[(site + 3) * C + cols[site + 3]];
    s1 += s2;
    sum += s1;
  }
  while (site < end) { sum += T[site * C + cols[site]]; ++site; }
  return sum;
}

/* restates opt_branch_lengths_pplacer (src/core/pll/optimize.cpp:60-248) over the window
 * [b, b+n) with the query as `tip`.  Returns the NEGATIVE log-likelihood like the reference. */
static double opt_pplacer(orc_tiny* tt, const orc_side* tip, size_t b, size_t n, int smoothings,
                          double tolerance, long* stat) {
  const orc_ctx* x = tt->x;
  const orc_model* m = &x->m;
  const int max_iters = 30;
  orc_side in = {tt->inner, NULL, tt->inner_sc};
  const double original_length = tt->len_dist * 2;
  double loglikelihood = -edge_lnl(m, tip, &in, tt->P_pend, x->invariant, NULL, b, n);
  nr_ctx cx = {m, tt->sumtable, n, x->invariant, b, 0, x->newton_variant};
  while (smoothings) {
    const double old_dist = tt->len_dist, old_pend = tt->len_pend;
    /* ---- NR for pendant (:135-166) ---- */
    double xmin = x->blo_min, xmax = x->blo_max, xtol = xmin / 10.0;
    double xguess = tt->len_pend;
    if (xguess < xmin || xguess > xmax) xguess = x->blo_default;
    update_sumtable(m, &in, tip, tt->sumtable, b, n);
    if (g_trace) g_trace->phase = 1;
    double xres = minimize_newton(xmin, xguess, xmax, xtol, max_iters, &cx);
    if (xres > 0.0) {
      tt->len_pend = xres;
      orc_pmatrix(m, xres, tt->P_pend);
    }
    /* ---- NR for distal, proximal := orig - distal (:170-211) ---- */
    update_partial(m, tip, tt->P_pend, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc, b, n);
    xguess = tt->len_dist;
    xmin = fmin(x->blo_min / 2.0, original_length / 2.0);
    xtol = xmin / 10.0;
    xmax = original_length - xtol;
    if (xguess < xmin || xguess > xmax) xguess = original_length / 2.0;
    update_sumtable(m, &tt->dist, &in, tt->sumtable, b, n);
    if (g_trace) g_trace->phase = 2;
    xres = minimize_newton(xmin, xguess, xmax, xtol, max_iters, &cx);
    if (xres > 0.0) {
      tt->len_dist = xres;
      tt->len_prox = original_length - xres;
      orc_pmatrix(m, tt->len_dist, tt->P_dist);
      orc_pmatrix(m, tt->len_prox, tt->P_prox);
    }
    /* ---- score (:217-222) ---- */
    update_partial(m, &tt->dist, tt->P_dist, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc, b, n);
    double new_ll = -edge_lnl(m, tip, &in, tt->P_pend, x->invariant, NULL, b, n);
    if (stat) stat[0]++;
    trace_row(3, new_ll, loglikelihood, (new_ll - loglikelihood > new_ll * 1e-14) ? 1.0 : 0.0);
    if (new_ll - loglikelihood > new_ll * 1e-14) { /* worse: restore lengths, keep old lnL */
      tt->len_pend = old_pend;
      tt->len_dist = old_dist;
      tt->len_prox = original_length - old_dist;
      if (stat) stat[2]++;
      break;
    }
    --smoothings;
    if (fabs(new_ll - loglikelihood) < tolerance) smoothings = 0;
    loglikelihood = new_ll;
  }
  if (stat) stat[1] += cx.n_evals;
  return loglikelihood;
}

/* restates pll-modules pllmod_opt_optimize_branch_lengths_local(radius 1, keep_update 1) on the
 * triplet (call site src/core/pll/optimize.cpp:274-279, `--raxml-blo`).  The source is not in the
 * tree; recollected structure (recomp_iterative): per smoothing round
 *   NR on the edge inner--query; CLV of the inner node re-aimed at the distal node, NR on that edge;
 *   re-aimed at the proximal node (with the new distal length), NR on that edge; CLV re-aimed at
 *   the query; NR on the query's edge once more (from the tip's side); edge lnL; stop when the lnL
 *   moved by less than the tolerance.  A length is replaced when the solver moved it by more than
 *   1e-10 (keep_update).  Bounds [MIN, MAX], tolerance MIN / 10, guess reset to DEFAULT when out
 *   of bounds; the three lengths are independent (Tiny_Tree::place rescales distal afterwards,
 *   src/tree/Tiny_Tree.cpp:183-185).  Returns the NEGATIVE log-likelihood. */
static double opt_local(orc_tiny* tt, const orc_side* tip, size_t b, size_t n, int smoothings,
                        double tolerance, long* stat) {
  const orc_ctx* x = tt->x;
  const orc_model* m = &x->m;
  const int max_iters = 30;
  orc_side in = {tt->inner, NULL, tt->inner_sc};
  nr_ctx cx = {m, tt->sumtable, n, x->invariant, b, 0, x->newton_variant};
  const double xmin = x->blo_min, xmax = x->blo_max, xtol = xmin / 10.0;
  double loglikelihood = -edge_lnl(m, tip, &in, tt->P_pend, x->invariant, NULL, b, n);
#define ORC_SOLVE(len, P)                                                            \
  do {                                                                               \
    double g_ = (len);                                                               \
    if (g_ < xmin || g_ > xmax) g_ = x->blo_default;                                 \
    const double r_ = minimize_newton(xmin, g_, xmax, xtol, max_iters, &cx);         \
    if (isfinite(r_) && fabs((len) - r_) > 1e-10) { (len) = r_; orc_pmatrix(m, r_, (P)); } \
  } while (0)
  while (smoothings) {
    update_sumtable(m, &in, tip, tt->sumtable, b, n);
    ORC_SOLVE(tt->len_pend, tt->P_pend);
    update_partial(m, tip, tt->P_pend, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc, b, n);
    update_sumtable(m, &tt->dist, &in, tt->sumtable, b, n);
    ORC_SOLVE(tt->len_dist, tt->P_dist);
    update_partial(m, tip, tt->P_pend, &tt->dist, tt->P_dist, tt->inner, tt->inner_sc, b, n);
    update_sumtable(m, &tt->prox, &in, tt->sumtable, b, n);
    ORC_SOLVE(tt->len_prox, tt->P_prox);
    update_partial(m, &tt->dist, tt->P_dist, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc, b, n);
    update_sumtable(m, tip, &in, tt->sumtable, b, n);
    ORC_SOLVE(tt->len_pend, tt->P_pend);
    const double new_ll = -edge_lnl(m, tip, &in, tt->P_pend, x->invariant, NULL, b, n);
    if (stat) stat[0]++;
    --smoothings;
    if (fabs(new_ll - loglikelihood) < tolerance) smoothings = 0;
    loglikelihood = new_ll;
  }
#undef ORC_SOLVE
  if (stat) stat[1] += cx.n_evals;
  return loglikelihood;
}

/* restates Tiny_Tree::place, opt_branches_ == true (src/tree/Tiny_Tree.cpp:159-204) with
 * call_focused / shift_partition_focus expressed as the window [begin, begin+span)
 * (src/core/pll/pll_util.cpp:388-418) and optimize_branch_triplet (optimize.cpp:253-286).
 * Returns 0 ok, 1 invalid char, 2 all-gap query, 3 -inf. */
static int tiny_place_thorough(orc_tiny* tt, const char* q, int premask, double* lnl,
                               double* pendant, double* distal, long* stat) {
  const orc_ctx* x = tt->x;
  const int s = x->m.s;
  size_t begin = 0, span = x->W;
  if (premask) {
    valid_range(q, x->W, &begin, &span);
    if (!span) return 2;
  }
  uint32_t* mk = (uint32_t*)malloc(sizeof(uint32_t) * x->W);
  for (size_t w = 0; w < x->W; ++w) {
    mk[w] = orc_char_mask(s, q[w]);
    if (!mk[w]) { free(mk); return 1; }
  }
  orc_side tip = {NULL, mk, NULL};
  /* traverse_update_partials (optimize.cpp:15-42): 3 P-matrices + inner CLV over the window */
  orc_pmatrix(&x->m, tt->len_prox, tt->P_prox);
  orc_pmatrix(&x->m, tt->len_dist, tt->P_dist);
  orc_pmatrix(&x->m, tt->len_pend, tt->P_pend);
  update_partial(&x->m, &tt->dist, tt->P_dist, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc,
                 begin, span);
  double ll = x->raxml_blo ? -opt_local(tt, &tip, begin, span, 32, x->blo_eps, stat)
                           : -opt_pplacer(tt, &tip, begin, span, 32, x->blo_eps, stat);
  free(mk);
  const double total = tt->len_dist + tt->len_prox;
  *distal = (tt->orig / total) * tt->len_dist;
  *pendant = tt->len_pend;
  *lnl = ll;
  /* state restored for the next query (Tiny_Tree.cpp:187-203) */
  tiny_reset_lengths(tt);
  update_partial(&x->m, &tt->dist, tt->P_dist, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc, 0,
                 x->W);
  if (ll == -INFINITY) return 3;
  return 0;
}

/* branch facts the product's descriptor is compared against */
void orc_branch_info(const orc_ctx* x, int b, double* orig, int* dist_is_tip) {
  int d = x->t->branch_rec[b], p = x->t->r[d].back;
  *orig = x->t->r[d].length;
  *dist_is_tip = (x->t->r[d].tip >= 0) || (x->t->r[p].tip >= 0);
}

/* copies of the two reference sides of branch b after the tip-is-distal orientation;
 * tips are exported as 0/1 CLVs and zero scalers.  clv_* [W][c][s], sc_* [W]. */
void orc_branch_sides(const orc_ctx* x, int b, double* clv_prox, uint32_t* sc_prox,
                      double* clv_dist, uint32_t* sc_dist) {
  orc_tiny* tt = tiny_create(x, b);
  const int s = x->m.s, c = x->m.c;
  const orc_side* sd[2] = {&tt->prox, &tt->dist};
  double* oc[2] = {clv_prox, clv_dist};
  uint32_t* os[2] = {sc_prox, sc_dist};
  for (int z = 0; z < 2; ++z)
    for (size_t w = 0; w < x->W; ++w) {
      if (x->m.rate_scalers)
        for (int k = 0; k < c; ++k) os[z][w * c + k] = sd[z]->scaler ? sd[z]->scaler[w * c + k] : 0;
      else
        os[z][w] = sd[z]->scaler ? sd[z]->scaler[w] : 0;
      for (int k = 0; k < c; ++k)
        for (int i = 0; i < s; ++i)
          oc[z][(w * c + k) * s + i] =
              sd[z]->clv ? sd[z]->clv[(w * c + k) * s + i]
                         : (((sd[z]->tipmask[w] >> i) & 1u) ? 1.0 : 0.0);
    }
  tiny_free(tt);
}

/* lookup table of one branch: T [W][C] */
void orc_branch_lookup(const orc_ctx* x, int b, double* T) {
  orc_tiny* tt = tiny_create(x, b);
  tiny_build_lookup(tt, T);
  tiny_free(tt);
}

/* ------------------------------------------------------------------------------------------
 * Hot loop 1: place() (src/core/place.cpp:41-95).  lnl [Q][B] row-major.  Same loop shape as
 * the reference: branch-major flattened index, thread-local tiny tree rebuilt on branch change,
 * schedule(guided,10000); the per-branch lookup is built once (Lookup_Store).
 * Returns 0 or the first error code (1 invalid char, 2 all-gap).
 * ---------------------------------------------------------------------------------------- */
int orc_preplace(orc_ctx* x, int Q, const char** queries, int premask, double* lnl) {
  const int s = x->m.s, C = orc_num_columns(s), B = x->t->B;
  const size_t W = x->W;
  int* cols = (int*)malloc(sizeof(int) * (size_t)Q * W);
  size_t* rng = (size_t*)malloc(sizeof(size_t) * 2 * Q);
  int err = 0;
  for (int q = 0; q < Q; ++q) {
    for (size_t w = 0; w < W; ++w) {
      int col = orc_char_column(s, queries[q][w], x->aa_x_quirk);
      if (col < 0) err = err ? err : 1;
      cols[(size_t)q * W + w] = col < 0 ? 0 : col;
    }
    rng[2 * q] = 0; rng[2 * q + 1] = W;
    if (premask) {
      valid_range(queries[q], W, &rng[2 * q], &rng[2 * q + 1]);
      if (!rng[2 * q + 1]) err = err ? err : 2;
    }
  }
  if (err) { free(cols); free(rng); return err; }
  /* Lookup_Store: lives as long as the tree, one lazily-filled matrix + mutex per branch */
  if (!x->store) {
    x->store = (double**)calloc(B, sizeof(double*));
#ifdef _OPENMP
    x->locks = (omp_lock_t*)malloc(sizeof(omp_lock_t) * B);
    for (int b = 0; b < B; ++b) omp_init_lock(&x->locks[b]);
#endif
  }
  double** store = x->store;
  const long total = (long)Q * B;
#pragma omp parallel
  {
    long prev = -1;
#pragma omp for schedule(guided, 10000)
    for (long i = 0; i < total; ++i) {
      const long b = i / Q, q = i % Q;
      if (b != prev) {
        /* the reference rebuilds the whole Tiny_Tree here; only its lookup is shared */
#ifdef _OPENMP
        omp_set_lock(&x->locks[b]);
#endif
        if (!store[b]) {
          double* T = (double*)malloc(sizeof(double) * W * C);
          orc_tiny* tt = tiny_create(x, (int)b);
          tiny_build_lookup(tt, T);
          tiny_free(tt);
          store[b] = T;
        }
#ifdef _OPENMP
        omp_unset_lock(&x->locks[b]);
#endif
      }
      lnl[(size_t)q * B + b] = lookup_sum(store[b], C, cols + (size_t)q * W, rng[2 * q], rng[2 * q + 1]);
      prev = b;
    }
  }
  free(cols); free(rng);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Hot loop 2: place_thorough() (src/core/place.cpp:97-171) over (branch, seq) pairs that the
 * caller supplies branch-sorted (Work iteration order, src/core/Work.hpp:115-207).
 * out: lnl/pendant/distal per pair.  stats[3] = {rounds, newton evals, reverts} (may be NULL).
 * ---------------------------------------------------------------------------------------- */
int orc_thorough(orc_ctx* x, long n_pairs, const int* pair_branch, const int* pair_seq,
                 const char** queries, int premask, double* lnl, double* pendant, double* distal,
                 long* stats) {
  int err = 0;
  long st0 = 0, st1 = 0, st2 = 0;
#pragma omp parallel reduction(+ : st0, st1, st2)
  {
    orc_tiny* tt = NULL;
    long st[3] = {0, 0, 0};
#pragma omp for schedule(dynamic)
    for (long i = 0; i < n_pairs; ++i) {
      if (!tt || tt->branch != pair_branch[i]) {
        tiny_free(tt);
        tt = tiny_create(x, pair_branch[i]);
      }
      int e = tiny_place_thorough(tt, queries[pair_seq[i]], premask, &lnl[i], &pendant[i],
                                  &distal[i], st);
      if (e) {
#pragma omp critical(orc_err)
        if (!err) err = e;
      }
    }
    tiny_free(tt);
    st0 += st[0]; st1 += st[1]; st2 += st[2];
  }
  if (stats) { stats[0] = st0; stats[1] = st1; stats[2] = st2; }
  return err;
}

/* direct (non-lookup) edge lnL of query q on branch b at default lengths over its window:
 * the quantity the lookup sum must reproduce (SURVEY 8c invariant "lookup-sum == direct") */
double orc_direct_default_lnl(const orc_ctx* x, int b, const char* q, int premask) {
  orc_tiny* tt = tiny_create(x, b);
  size_t begin = 0, span = x->W;
  if (premask) valid_range(q, x->W, &begin, &span);
  uint32_t* mk = (uint32_t*)malloc(sizeof(uint32_t) * x->W);
  for (size_t w = 0; w < x->W; ++w) mk[w] = orc_char_mask(x->m.s, q[w]);
  orc_side tip = {NULL, mk, NULL};
  orc_side in = {tt->inner, NULL, tt->inner_sc};
  double l = edge_lnl(&x->m, &tip, &in, tt->P_pend, x->invariant, NULL, begin, span);
  free(mk);
  tiny_free(tt);
  return l;
}

/* f, df of -lnL w.r.t. the pendant length at t for query q on branch b (default other lengths):
 * used by the finite-difference KAT */
void orc_pendant_derivatives(const orc_ctx* x, int b, const char* q, double t, double* f,
                             double* df, double* lnl_at_t) {
  orc_tiny* tt = tiny_create(x, b);
  uint32_t* mk = (uint32_t*)malloc(sizeof(uint32_t) * x->W);
  for (size_t w = 0; w < x->W; ++w) mk[w] = orc_char_mask(x->m.s, q[w]);
  orc_side tip = {NULL, mk, NULL};
  orc_side in = {tt->inner, NULL, tt->inner_sc};
  update_sumtable(&x->m, &in, &tip, tt->sumtable, 0, x->W);
  lk_derivatives(&x->m, tt->sumtable, x->W, t, x->invariant, 0, f, df);
  orc_pmatrix(&x->m, t, tt->P_pend);
  *lnl_at_t = edge_lnl(&x->m, &tip, &in, tt->P_pend, x->invariant, NULL, 0, x->W);
  free(mk);
  tiny_free(tt);
}

/* ------------------------------------------------------------------------------------------
 * Evaluator without the optimiser: the edge log-likelihood of query q on branch b at GIVEN
 * lengths -- what Tiny_Tree::place computes when the lengths are fixed: the three P-matrices and
 * the inner CLV over the window (traverse_update_partials, src/core/pll/optimize.cpp:15-42) and
 * pll_compute_edge_loglikelihood (:281-283), i.e. the quantity the reference's own sanity test
 * looks at for a returned Placement (test/src/Tiny_Tree.cpp:39-48).  proximal == NULL: the
 * sliding rule's proximal = original - distal (optimize.cpp:206-210).  Tests use it to check the
 * DEVICE's log-likelihood at the DEVICE's lengths, whatever path its optimiser took.
 * ---------------------------------------------------------------------------------------- */
int orc_score_at(orc_ctx* x, long n_pairs, const int* pair_branch, const int* pair_seq,
                 const char** queries, int premask, const double* pendant, const double* distal,
                 const double* proximal, double* lnl) {
  int err = 0;
#pragma omp parallel
  {
    orc_tiny* tt = NULL;
#pragma omp for schedule(dynamic)
    for (long i = 0; i < n_pairs; ++i) {
      if (!tt || tt->branch != pair_branch[i]) {
        tiny_free(tt);
        tt = tiny_create(x, pair_branch[i]);
      }
      const char* q = queries[pair_seq[i]];
      size_t begin = 0, span = x->W;
      if (premask) valid_range(q, x->W, &begin, &span);
      uint32_t* mk = (uint32_t*)malloc(sizeof(uint32_t) * x->W);
      int bad = span == 0;
      for (size_t w = 0; w < x->W; ++w) {
        mk[w] = orc_char_mask(x->m.s, q[w]);
        if (!mk[w]) bad = 1;
      }
      if (bad) {
#pragma omp critical(orc_err)
        err = 1;
        lnl[i] = NAN;
      } else {
        orc_side tip = {NULL, mk, NULL};
        orc_side in = {tt->inner, NULL, tt->inner_sc};
        const double lp = proximal ? proximal[i] : tt->orig - distal[i];
        orc_pmatrix(&x->m, lp, tt->P_prox);
        orc_pmatrix(&x->m, distal[i], tt->P_dist);
        orc_pmatrix(&x->m, pendant[i], tt->P_pend);
        update_partial(&x->m, &tt->dist, tt->P_dist, &tt->prox, tt->P_prox, tt->inner, tt->inner_sc,
                       begin, span);
        lnl[i] = edge_lnl(&x->m, &tip, &in, tt->P_pend, x->invariant, NULL, begin, span);
      }
      free(mk);
    }
    tiny_free(tt);
  }
  return err;
}

/* see orc_model.rounding_variant; 0 = off.  Affects the optimiser's path (derivative tables, edge
 * score), not the reference CLVs (those are computed at create time). */
void orc_set_rounding_variant(orc_ctx* x, uint64_t seed) { x->m.rounding_variant = seed; }

/* one pair through tiny_place_thorough with the trace on (rows documented at orc_trace).
 * Returns the number of rows the run produced (may exceed cap; only cap are stored). */
long orc_trace_pair(orc_ctx* x, int b, const char* q, int premask, double* rows, long cap,
                    double* lnl, double* pendant, double* distal) {
  orc_trace tr = {rows, cap, 0, 0};
  orc_tiny* tt = tiny_create(x, b);
  g_trace = &tr;
  tiny_place_thorough(tt, q, premask, lnl, pendant, distal, NULL);
  g_trace = NULL;
  tiny_free(tt);
  return tr.n;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
