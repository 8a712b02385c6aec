// Hot loop 2 on device, 4-state / 4-category models: one wavefront per (branch, query) pair.
//
// k_thorough_dna replaces, per pair, Tiny_Tree::place with opt_branches (src/tree/Tiny_Tree.cpp:
// 159-204) -> call_focused (src/core/pll/pll_util.hpp:53-65) -> optimize_branch_triplet
// (src/core/pll/optimize.cpp:253-286) -> opt_branch_lengths_pplacer (:60-248) and the libpll /
// pll-modules calls underneath: pll_update_prob_matrices, pll_update_partials,
// pll_update_sumtable, pll_compute_likelihood_derivatives, pll_compute_edge_loglikelihood,
// pllmod_opt_minimize_newton.  The control flow (round structure, bounds, guesses, the
// "worse -> restore lengths, keep old lnL" exit, the 0.1 lnL stop) is the reference's; the
// arithmetic is reorganised for the machine:
//
//  * everything lives in the eigenbasis of Q.  HBM holds Xt = U^-1 X and Dt = U^-1 D for the
//    proximal / distal reference CLVs (component-major, so a wave reads 64 consecutive sites of
//    one component = one 512 B coalesced segment).  A branch's transition matrix is never
//    formed:  P(t) v = U (exp(lambda r t) o U^-1 v).
//  * with S = sumtable, lnL_site(t) = log sum_k w_k sum_x S_kx exp(lambda_x r_k t) is the SAME
//    contraction as the Newton derivatives (order 0), so one register-resident S per site
//    serves pll_compute_edge_loglikelihood and every Newton iteration:
//        pendant:  S_kx = (U^-1 I)_kx * (U^-1 q)_x          I = inner CLV toward the query
//        distal :  S_kx = Dt_kx * (U^-1 I')_kx              I' = inner CLV toward distal
//  * lane = alignment site of the query's window (NCH sites per lane, S in VGPRs); f, f' and
//    lnL are wave-wide butterfly reductions; the 16 exp() of a Newton proposal are computed by
//    16 lanes and broadcast with v_readlane (SGPR operands); U / U^-1 / eigenvalues come in as
//    kernel arguments (scalar registers).
//  * no P-matrix, no sumtable, no inner CLV ever touches memory.
#include "epa_dev_internal.hpp"

namespace {

constexpr double LOG_THR = -256.0 * 0.6931471805599453094;  // log(2^-256)

struct ThArgs {
  ModelDNA m;
  BloConsts blo;
  const double* refT;      // [2B][16][W]
  const uint32_t* scSum;   // [B][W]
  const double* blen;      // [B]
  const double* qt;        // [16 columns][4]   U^-1 image of each column's tip vector
  const epa_pair* pairs;
  const uint8_t* codes;    // [Q][W]
  const uint32_t* win_begin;
  const uint32_t* win_span;
  epa_result* out;
  unsigned long long* stats;  // [0] rounds [1] newton evals [2] reverts [3] non-finite [4] first bad
  double* sscratch;        // NCH == 0 only: [waves][16][Wpad]
  uint64_t n_pairs;
  uint32_t W;
  uint32_t Wpad;
};

__device__ __forceinline__ double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return readlane_d(v, 0);
}

// exp(lam_x * r_k * t) for the 16 (k,x) pairs: lane l < 16 computes pair l, all lanes receive all
struct E16 { double v[16]; };
__device__ __forceinline__ void exp_table(const ModelDNA& m, double lr_lane, double t, double scale_lane,
                                          E16& e) {
  const double mine = exp(lr_lane * t) * scale_lane;
#pragma unroll
  for (int i = 0; i < 16; ++i) e.v[i] = readlane_d(mine, i);
}

// one site: I_ki = (U (ea_k o A_k))_i * (U (eb_k o Bv_k))_i, per-site rescale, return U^-1 I
// A, Bv: eigen-space vectors [k][x]; ea/eb uniform exp tables.  resc: 1 if rescaled.
__device__ __forceinline__ void inner_site(const ModelDNA& m, const double (&A)[16], const E16& ea,
                                           const double (&Bv)[16], const E16& eb, double (&It)[16],
                                           uint32_t& resc) {
  double I[16];
  double mx = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double av[4], bv[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      av[x] = A[k * 4 + x] * ea.v[k * 4 + x];
      bv[x] = Bv[k * 4 + x] * eb.v[k * 4 + x];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double a = m.U[i * 4] * av[0], b = m.U[i * 4] * bv[0];
#pragma unroll
      for (int x = 1; x < 4; ++x) {
        a = fma(m.U[i * 4 + x], av[x], a);
        b = fma(m.U[i * 4 + x], bv[x], b);
      }
      const double v = a * b;
      I[k * 4 + i] = v;
      mx = fmax(mx, v);
    }
  }
  // pll_update_partials per-site scaling: all c*s entries < 2^-256 -> * 2^256, scaler + 1
  resc = (mx < 0x1p-256) ? 1u : 0u;
  const double mult = resc ? 0x1p+256 : 1.0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      double acc = m.Ui[x * 4] * I[k * 4];
#pragma unroll
      for (int i = 1; i < 4; ++i) acc = fma(m.Ui[x * 4 + i], I[k * 4 + i], acc);
      It[k * 4 + x] = acc * mult;
    }
}

template <int NCH>
struct SiteState {
  double S[NCH][16];   // sumtable of the branch currently being optimised
  uint32_t sc[NCH];    // proximal + distal scaler counts
  uint32_t resc[NCH];  // rescale flag of the last inner CLV toward the query
  uint32_t code[NCH];  // query column code
  bool valid[NCH];
};

// sum over the wave's sites of  -l1/l0  and  (l1/l0)^2 - l2/l0   (pll_compute_likelihood_
// derivatives).  ew = w_k * exp(lam_x r_k t).
template <int NCH>
__device__ __forceinline__ void derivatives(const ModelDNA& m, const SiteState<NCH>& st,
                                            const E16& ew, double& f, double& df) {
  double fl = 0.0, dfl = 0.0;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double u0 = 0.0, u1 = 0.0, u2 = 0.0;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const double t = st.S[ch][k * 4 + x] * ew.v[k * 4 + x];
        const double lx = m.lam[x];
        u0 += t;
        u1 = fma(t, lx, u1);
        u2 = fma(t, lx * lx, u2);
      }
      const double r = m.rate[k];
      l0 += u0;
      l1 = fma(u1, r, l1);
      l2 = fma(u2, r * r, l2);
    }
    const double inv = 1.0 / l0;
    const double d1 = -l1 * inv;
    const double d2 = fma(d1, d1, -l2 * inv);
    if (st.valid[ch]) { fl += d1; dfl += d2; }
  }
  f = wave_sum(fl);
  df = wave_sum(dfl);
}

// sum over the window of log L_site(t) + scalers * log(2^-256)  (pll_compute_edge_loglikelihood)
template <int NCH>
__device__ __forceinline__ double window_lnl(const SiteState<NCH>& st, const E16& ew) {
  double acc = 0.0;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    double l0 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double u0 = 0.0;
#pragma unroll
      for (int x = 0; x < 4; ++x) u0 = fma(st.S[ch][k * 4 + x], ew.v[k * 4 + x], u0);
      l0 += u0;
    }
    const double v = log(l0) + (double)(st.sc[ch] + st.resc[ch]) * LOG_THR;
    if (st.valid[ch]) acc += v;
  }
  return wave_sum(acc);
}

// pllmod_opt_minimize_newton (pll-modules; rtsafe-style safeguarded Newton).  Wave-uniform.
template <int NCH>
__device__ __forceinline__ double newton(const ModelDNA& m, const SiteState<NCH>& st, double lr_lane,
                                         double w_lane, double x1, double xguess, double x2,
                                         double tol, int max_iters, uint32_t& evals) {
  double rts = xguess, f, df, xl, xh, dx;
  if (rts < x1) rts = x1;
  if (rts > x2) rts = x2;
  E16 ew;
  exp_table(m, lr_lane, rts, w_lane, ew);
  derivatives<NCH>(m, st, ew, f, df);
  ++evals;
  if (!isfinite(f) || !isfinite(df)) return NAN;
  if (df >= 0.0 && fabs(f) < tol) return rts;
  if (f < 0.0) { xl = rts; xh = x2; } else { xh = rts; xl = x1; }
  for (int i = 1; i <= max_iters; ++i) {
    if (df <= 0.0 || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0)) {
      dx = 0.5 * (xh - xl);
      rts = xl + dx;
      if (xl == rts) return rts;
    } else {
      dx = f / df;
      const double temp = rts;
      rts -= dx;
      if (temp == rts) return rts;
    }
    if (fabs(dx) < tol || i == max_iters) return rts;
    if (rts < x1) rts = x1;
    exp_table(m, lr_lane, rts, w_lane, ew);
    derivatives<NCH>(m, st, ew, f, df);
    ++evals;
    if (!isfinite(f) || !isfinite(df)) return NAN;
    if (df > 0.0 && fabs(f) < tol) return rts;
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return NAN;
}

template <int NCH>
__device__ __forceinline__ void process_pair(const ThArgs& a, const uint64_t pid, const int lane) {
  const ModelDNA& m = a.m;
  const epa_pair pr = a.pairs[pid];
  const uint32_t b = pr.branch_id, q = pr.seq_id;
  const uint32_t begin = a.win_begin[q], n = a.win_span[q];
  const size_t cW = a.W;
  const double* Xt = a.refT + (size_t)(2 * b) * 16 * cW + begin;
  const double* Dt = a.refT + (size_t)(2 * b + 1) * 16 * cW + begin;
  const uint32_t* scp = a.scSum + (size_t)b * cW + begin;
  const uint8_t* qc = a.codes + (size_t)q * cW + begin;
  const double orig = a.blen[b];

  // lane-private constants for the broadcast exp tables: pair (k,x) = lane & 15
  const int lk = (lane >> 2) & 3, lx = lane & 3;
  const double lr_lane = m.lam[lx] * m.rate[lk];
  const double w_lane = m.w[lk];

  SiteState<NCH> st;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const uint32_t s = ch * 64 + lane;
    st.valid[ch] = s < n;
    const uint32_t sc = st.valid[ch] ? s : 0;  // clamp: inactive lanes recompute site 0
    st.sc[ch] = scp[sc];
    st.code[ch] = qc[sc];
    st.resc[ch] = 0;
  }

  double tp = a.blo.pendant_default, td = orig * 0.5, tx = orig * 0.5;
  uint32_t evals = 0, rounds = 0, reverted = 0;

  // inner CLV toward the query at the current (td, tx), folded with the query: S = (U^-1 I) o qt
  auto score_sumtable = [&](double td_, double tx_) {
    E16 ed, ex;
    exp_table(m, lr_lane, td_, 1.0, ed);
    exp_table(m, lr_lane, tx_, 1.0, ex);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const uint32_t s = st.valid[ch] ? ch * 64 + lane : 0;
      double D[16], X[16], It[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) { D[c] = Dt[(size_t)c * cW + s]; X[c] = Xt[(size_t)c * cW + s]; }
      inner_site(m, D, ed, X, ex, It, st.resc[ch]);
      const double* qv = a.qt + st.code[ch] * 4;
      const double q0 = qv[0], q1 = qv[1], q2 = qv[2], q3 = qv[3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        st.S[ch][k * 4 + 0] = It[k * 4 + 0] * q0;
        st.S[ch][k * 4 + 1] = It[k * 4 + 1] * q1;
        st.S[ch][k * 4 + 2] = It[k * 4 + 2] * q2;
        st.S[ch][k * 4 + 3] = It[k * 4 + 3] * q3;
      }
    }
  };
  // inner CLV toward distal: I' = (P_pend q) o (P_prox X); S = Dt o (U^-1 I')
  auto distal_sumtable = [&](double tp_, double tx_) {
    E16 ep, ex;
    exp_table(m, lr_lane, tp_, 1.0, ep);
    exp_table(m, lr_lane, tx_, 1.0, ex);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const uint32_t s = st.valid[ch] ? ch * 64 + lane : 0;
      double Qv[16], X[16], It[16];
      const double* qv = a.qt + st.code[ch] * 4;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const double v = qv[x];
        Qv[x] = v; Qv[4 + x] = v; Qv[8 + x] = v; Qv[12 + x] = v;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) X[c] = Xt[(size_t)c * cW + s];
      uint32_t r;
      inner_site(m, Qv, ep, X, ex, It, r);
#pragma unroll
      for (int c = 0; c < 16; ++c) st.S[ch][c] = Dt[(size_t)c * cW + s] * It[c];
    }
  };
  auto lnl_at = [&](double tp_) {
    E16 ew;
    exp_table(m, lr_lane, tp_, w_lane, ew);
    return window_lnl<NCH>(st, ew);
  };

  // traverse_update_partials + initial score (optimize.cpp:15-42,111-113)
  score_sumtable(td, tx);
  double loglikelihood = -lnl_at(tp);

  uint32_t smoothings = a.blo.max_rounds;
  while (smoothings) {
    const double old_td = td, old_tp = tp;
    // ---- NR for the pendant length (optimize.cpp:135-166); S already holds the pendant sumtable
    double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
    double xguess = tp;
    if (xguess < xmin || xguess > xmax) xguess = a.blo.default_branch;
    double xres = newton<NCH>(m, st, lr_lane, w_lane, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, evals);
    if (xres > 0.0) tp = xres;
    // ---- NR for the distal length with the proximal P-matrix held fixed (:170-211)
    distal_sumtable(tp, tx);
    xguess = td;
    xmin = fmin(a.blo.min_branch / 2.0, orig / 2.0);
    xtol = xmin / 10.0;
    xmax = orig - xtol;
    if (xguess < xmin || xguess > xmax) xguess = orig / 2.0;
    xres = newton<NCH>(m, st, lr_lane, w_lane, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, evals);
    if (xres > 0.0) { td = xres; tx = orig - xres; }
    // ---- score (:217-222)
    score_sumtable(td, tx);
    const double new_ll = -lnl_at(tp);
    ++rounds;
    if (new_ll - loglikelihood > new_ll * 1e-14) {  // worse: restore lengths, keep the old lnL
      tp = old_tp; td = old_td; tx = orig - old_td;
      reverted = 1;
      break;
    }
    --smoothings;
    if (fabs(new_ll - loglikelihood) < a.blo.epsilon) smoothings = 0;
    loglikelihood = new_ll;
  }

  if (lane == 0) {
    const double lnl = -loglikelihood;
    epa_result r;
    r.lnl = lnl;
    r.pendant_length = tp;
    r.distal_length = (orig / (td + tx)) * td;  // Tiny_Tree.cpp:183-185
    a.out[pid] = r;
    atomicAdd(&a.stats[0], (unsigned long long)rounds);
    atomicAdd(&a.stats[1], (unsigned long long)evals);
    atomicAdd(&a.stats[2], (unsigned long long)reverted);
    if (!isfinite(lnl)) {
      if (atomicAdd(&a.stats[3], 1ull) == 0) a.stats[4] = ((unsigned long long)b << 32) | q;
    }
  }
}

// Persistent single-wave workgroups.  Workgroup g is observed to run on XCD g % 8 (used for
// speed only): XCD x owns the x-th eighth of the branch-sorted pair list, so one branch's CLV
// windows are served by one 4 MiB L2.  Inside its XCD slice a wave takes pairs round-robin
// (wave, wave + stride, ...): neighbouring waves work on neighbouring pairs = the same branch,
// and every wave gets a ~25-pair random sample of the 10x cost spread (1..32 NR rounds).
template <int NCH>
__global__ void __launch_bounds__(64) k_thorough_dna(const ThArgs a) {
  const int lane = threadIdx.x;
  const uint32_t x = blockIdx.x & 7;
  const uint32_t w = blockIdx.x >> 3, stride = gridDim.x >> 3;
  const uint64_t per = (a.n_pairs + 7) / 8;
  const uint64_t lo = (uint64_t)x * per;
  const uint64_t hi = lo + per < a.n_pairs ? lo + per : a.n_pairs;
  for (uint64_t p = lo + w; p < hi; p += stride) process_pair<NCH>(a, p, lane);
}

}  // namespace

int launch_thorough_aa(epa_ctx* ctx, const epa_pair* d_pairs, uint64_t n_pairs, const uint8_t* d_codes,
                       const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                       epa_result* d_out, unsigned long long* d_stats);

int launch_thorough(epa_ctx* ctx, const epa_pair* d_pairs, uint64_t n_pairs, const uint8_t* d_codes,
                    const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                    epa_result* d_out, unsigned long long* d_stats) {
  if (ctx->s == 20)
    return launch_thorough_aa(ctx, d_pairs, n_pairs, d_codes, d_begin, d_span, max_span, d_out, d_stats);
  ThArgs a;
  a.m = ctx->dna;
  a.blo = ctx->blo;
  a.refT = ctx->refT;
  a.scSum = ctx->scSum;
  a.blen = ctx->blen;
  a.qt = ctx->dmodel->qt;
  a.pairs = d_pairs;
  a.codes = d_codes;
  a.win_begin = d_begin;
  a.win_span = d_span;
  a.out = d_out;
  a.stats = d_stats;
  a.sscratch = nullptr;
  a.n_pairs = n_pairs;
  a.W = ctx->W;
  a.Wpad = 0;
  const uint32_t nch = (max_span + 63) / 64;
  // persistent grid: 8 single-wave workgroups per CU (2 per SIMD at this kernel's VGPR budget)
  uint32_t nwg = 256 * 8;
  if ((uint64_t)nwg > n_pairs) nwg = (uint32_t)((n_pairs + 7) / 8 * 8);
  epa_timer_start(ctx, ctx->t_thorough);
#define LAUNCH(N) hipLaunchKernelGGL(k_thorough_dna<N>, dim3(nwg), dim3(64), 0, ctx->stream, a)
  if (nch <= 1) LAUNCH(1);
  else if (nch <= 2) LAUNCH(2);
  else if (nch <= 3) LAUNCH(3);
  else if (nch <= 4) LAUNCH(4);
  else if (nch <= 6) LAUNCH(6);
  else if (nch <= 8) LAUNCH(8);
  else if (nch <= 12) LAUNCH(12);
  else if (nch <= 16) LAUNCH(16);
  else if (nch <= 24) LAUNCH(24);
  else {
    epa_timer_stop(ctx, ctx->t_thorough);
    return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "thorough: query windows longer than 1536 sites");
  }
#undef LAUNCH
  epa_timer_stop(ctx, ctx->t_thorough);
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}
