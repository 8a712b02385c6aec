// Hot loop 2 on device, general form: any number of rate categories (1 .. EPA_MAX_CATS), 4 or 20
// states, sliding ("pplacer") or radius-1 local ("--raxml-blo") branch-length optimisation.
//
// Same mapping to the reference as k_thorough_dna (thorough_dna.hip): Tiny_Tree::place with
// opt_branches (src/tree/Tiny_Tree.cpp:159-204) -> optimize_branch_triplet
// (src/core/pll/optimize.cpp:253-286) -> opt_branch_lengths_pplacer (:60-248) or
// pllmod_opt_optimize_branch_lengths_local (:274-279), everything in the eigenbasis of Q.
// One wavefront per (branch, query) pair, lane = site of the window (runtime loop over 64-site
// chunks), the sumtable of the branch being optimised lives in an HBM slab ([c*s + 1][Wpad]
// doubles per resident wave: c*s components + the site's scaler count), category and state loops
// are runtime / unrolled-by-S loops.  The tuned kernels (k_thorough_dna, k_thorough_aa) cover the
// default shape (4 categories, per-site scalers, sliding); every other valid model lands here, so
// no model string the parser accepts is refused by the thorough step.
//
// Per-rate scalers (PLL_ATTRIB_RATE_SCALERS) never reach the thorough kernels: the reference
// precompute keeps one count per category and k_align_rates (epa_dev.hip) folds libpll's
// evaluation-time alignment of the categories into the stored vectors, so scSum is per site here.
#include "epa_dev_internal.hpp"
#include "wave_util.hpp"

#include <algorithm>

namespace {

using namespace epa_wave;

struct ThArgsG {
  const ModelDev* m;
  BloConsts blo;
  const double* refT;      // [2B][c*s][W]
  const uint32_t* scSum;   // [B][W] proximal + distal scaler counts
  const double* cinv;      // +I: [W] p * pi_inv per site, or null
  double inv_w0;
  const double* blen;
  const epa_pair* pairs;
  const uint32_t* order;
  const uint8_t* codes;
  uint32_t cstride, crel;
  const uint32_t* win_begin;
  const uint32_t* win_span;
  epa_result* out;
  unsigned long long* stats;
  double* slab;            // [gridDim.x][c*s + 1][Wpad]
  uint64_t n_pairs;
  uint32_t W, Wpad;
};

constexpr double LN2 = 0.6931471805599453094;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// pllmod_opt_minimize_newton (rtsafe-style safeguarded Newton), wave-uniform; see newton() in
// thorough_dna.hip and minimize_newton() in oracle/epa_oracle.c
template <class Deriv>
__device__ __forceinline__ double newton_g(Deriv&& deriv, double x1, double xguess, double x2, double tol,
                                           int max_iters, int nv, uint32_t& evals) {
  double rts = xguess, f, df, xl, xh, dx;
  if (rts < x1) rts = x1;
  if (rts > x2) rts = x2;
  deriv(rts, f, df);
  ++evals;
  if (!isfinite(f) || !isfinite(df)) return NAN;
  if (((nv & 2) ? df > 0.0 : df >= 0.0) && fabs(f) < tol) return rts;
  if (f < 0.0) { xl = rts; xh = x2; } else { xh = rts; xl = x1; }
  double dxold = fabs(xh - xl);
  dx = dxold;
  for (int i = 1; i <= max_iters; ++i) {
    const bool slow = (nv & 1) && fabs(2.0 * f) > fabs(dxold * df);   // Numerical Recipes rtsafe clause
    dxold = dx;
    if (df <= 0.0 || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0) || slow) {
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
    deriv(rts, f, df);
    ++evals;
    if (!isfinite(f) || !isfinite(df)) return NAN;
    if (df > 0.0 && fabs(f) < tol) return rts;
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return NAN;
}

template <int S>
__global__ void __launch_bounds__(64) k_thorough_generic(const ThArgsG a) {
  __shared__ double U[S * S], Ui[S * S];
  __shared__ double tab[3][EPA_MAX_CATS * S];
  const int lane = threadIdx.x;
  const ModelDev* __restrict__ m = a.m;
  const int c = m->c, cs = c * S;
  for (int i = lane; i < S * S; i += 64) { U[i] = m->U[i]; Ui[i] = m->Ui[i]; }
  __syncthreads();
  double* slab = a.slab + (size_t)blockIdx.x * (size_t)(cs + 1) * a.Wpad;
  const size_t cW = a.W, Wp = a.Wpad;
  uint32_t wrounds = 0, wevals = 0, wreverts = 0;

  for (uint64_t pidx = blockIdx.x; pidx < a.n_pairs; pidx += gridDim.x) {
    const uint64_t pid = a.order ? a.order[pidx] : pidx;
    const epa_pair pr = a.pairs[pid];
    const uint32_t b = pr.branch_id, q = pr.seq_id;
    const uint32_t begin = a.win_begin[q], n = a.win_span[q];
    const uint32_t nch = (n + 63) / 64;
    const double* Xt = a.refT + (size_t)(2 * b) * cs * cW + begin;       // proximal side
    const double* Dt = a.refT + (size_t)(2 * b + 1) * cs * cW + begin;   // distal side
    const uint32_t* scp = a.scSum + (size_t)b * cW + begin;
    const uint8_t* qc = a.codes + (size_t)q * a.cstride + (a.crel ? 0u : begin);
    const double orig = a.blen[b];

    // wave-uniform tables: slot j entry i = (k, x) -> f_j(lam_x r_k) ; mode 0: exp(lr t_j) for
    // j = 0, 1 and w_k exp(lr t_2); mode 1 (Newton): w e, w lr e, w lr^2 e at t0
    auto tables = [&](double t0, double t1, double t2, int mode) {
      wave_sync();
      for (int i = lane; i < cs; i += 64) {
        const int k = i / S, x = i - k * S;
        const double lr = m->lam[x] * m->rate[k], w = m->w[k];
        if (mode == 0) {
          tab[0][i] = exp(lr * t0);
          tab[1][i] = exp(lr * t1);
          tab[2][i] = w * exp(lr * t2);
        } else {
          const double e = w * exp(lr * t0);
          tab[0][i] = e;
          tab[1][i] = lr * e;
          tab[2][i] = lr * lr * e;
        }
      }
      wave_sync();
    };

    // One site of "two operands -> inner CLV -> sumtable": I_ki = (U (ea o A_k))_i (U (eb o B_k))_i,
    // It = U^-1 I, sumtable entry = It_x * F_kx written to the slab; returns sum_kx entry * tab[2]
    // (the site likelihood at the pendant length of tab[2], used by score only).
    // Operands: 0 = the query's tip vector (category independent), 1 = distal CLV, 2 = proximal CLV;
    // A is propagated with tab[0], B with tab[1], F is the other end of the branch being optimised.
    auto site_pass = [&](uint32_t site, bool valid, int opA, int opB, int opF, uint32_t& count) -> double {
      const uint32_t s = valid ? site : 0;
      const double* qv = m->qt + (size_t)qc[s] * S;
      auto operand = [&](int op, int k, int x) -> double {
        return op == 0 ? qv[x] : (op == 1 ? Dt : Xt)[(size_t)(k * S + x) * cW + s];
      };
      double lk[EPA_MAX_CATS];
      bool all_small = true;
      for (int k = 0; k < c; ++k) {
        double av[S], bv[S], I[S];
#pragma unroll
        for (int x = 0; x < S; ++x) {
          av[x] = operand(opA, k, x) * tab[0][k * S + x];
          bv[x] = operand(opB, k, x) * tab[1][k * S + x];
        }
        double mx = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          double p = U[i * S] * av[0], r = U[i * S] * bv[0];
#pragma unroll
          for (int x = 1; x < S; ++x) {
            p = fma(U[i * S + x], av[x], p);
            r = fma(U[i * S + x], bv[x], r);
          }
          I[i] = p * r;
          mx = fmax(mx, I[i]);
        }
        all_small = all_small && mx < 0x1p-256;   // pll_update_partials: every entry below 2^-256
        double l = 0.0;
#pragma unroll
        for (int x = 0; x < S; ++x) {
          double acc = Ui[x * S] * I[0];
#pragma unroll
          for (int i = 1; i < S; ++i) acc = fma(Ui[x * S + i], I[i], acc);
          const double sv = acc * operand(opF, k, x);
          if (valid) slab[(size_t)(k * S + x) * Wp + site] = sv;
          l = fma(sv, tab[2][k * S + x], l);
        }
        lk[k] = l;
      }
      // per-site scaling: all c * s entries below the threshold -> * 2^256, count + 1
      const double mult = all_small ? 0x1p+256 : 1.0;
      count = scp[s] + (all_small ? 1u : 0u);
      double l0 = 0.0;
      for (int k = 0; k < c; ++k) l0 = fma(mult, lk[k], l0);
      if (all_small && valid)
        for (int i = 0; i < cs; ++i) slab[(size_t)i * Wp + site] *= mult;
      if (a.cinv) {   // +I: p * pi_inv enters L_0 only (eigenvalue 0 is exactly 0, see thorough_dna.hip)
        const double add = a.cinv[begin + s] * a.inv_w0;
        if (valid) slab[site] += add;
        l0 = fma(add, tab[2][0], l0);
      }
      return l0;
    };

    // inner CLV toward the query at (td, tx), sumtable = (U^-1 I) o q -> slab; returns the window lnL
    auto score = [&](double td_, double tx_, double tp_) -> double {
      tables(td_, tx_, tp_, 0);
      double mant = 1.0;
      int ex = 0;
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        const bool valid = site < n;
        uint32_t count;
        double l0 = site_pass(site, valid, 1, 2, 0, count);
        if (valid) slab[(size_t)cs * Wp + site] = (double)count;
        if (!valid) { l0 = 1.0; count = 0; }
        mant *= __builtin_amdgcn_frexp_mant(l0);
        ex += __builtin_amdgcn_frexp_exp(l0) - 256 * (int)count;
        ex += __builtin_amdgcn_frexp_exp(mant);
        mant = __builtin_amdgcn_frexp_mant(mant);
      }
      __threadfence_block();
      return wave_sum(log(mant) + (double)ex * LN2);
    };
    // inner CLV toward distal: I' = (P_pend q) o (P_prox X); sumtable = Dt o (U^-1 I') -> slab
    // (toward_proximal: the mirror image, I'' = (P_pend q) o (P_dist D), sumtable = Xt o (U^-1 I''))
    auto side_sumtable = [&](double tp_, double tother, bool toward_proximal) {
      tables(tp_, tother, 0.0, 0);
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        uint32_t count;
        (void)site_pass(site, site < n, 0, toward_proximal ? 1 : 2, toward_proximal ? 2 : 1, count);
      }
      __threadfence_block();
    };
    // window lnL at pendant length t from the pendant sumtable in the slab (pll_compute_edge_
    // loglikelihood on the query's edge without touching the inner CLV)
    auto lnl_from_slab = [&](double t) -> double {
      tables(0.0, 0.0, t, 0);
      double mant = 1.0;
      int ex = 0;
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        const bool valid = site < n;
        const uint32_t s = valid ? site : 0;
        double l0 = 0.0;
        for (int i = 0; i < cs; ++i) l0 = fma(slab[(size_t)i * Wp + s], tab[2][i], l0);
        int count = (int)slab[(size_t)cs * Wp + s];
        if (!valid) { l0 = 1.0; count = 0; }
        mant *= __builtin_amdgcn_frexp_mant(l0);
        ex += __builtin_amdgcn_frexp_exp(l0) - 256 * count;
        ex += __builtin_amdgcn_frexp_exp(mant);
        mant = __builtin_amdgcn_frexp_mant(mant);
      }
      return wave_sum(log(mant) + (double)ex * LN2);
    };
    auto deriv = [&](double t, double& f, double& df) {
      tables(t, 0.0, 0.0, 1);
      double fl = 0.0, dfl = 0.0;
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        const bool valid = site < n;
        const uint32_t s = valid ? site : 0;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0;
        for (int i = 0; i < cs; ++i) {
          const double sv = slab[(size_t)i * Wp + s];
          l0 = fma(sv, tab[0][i], l0);
          l1 = fma(sv, tab[1][i], l1);
          l2 = fma(sv, tab[2][i], l2);
        }
        const double inv = fast_rcp(l0);
        const double d1 = -l1 * inv;
        const double d2 = fma(d1, d1, -l2 * inv);
        if (valid) { fl += d1; dfl += d2; }
      }
      f = wave_sum(fl);
      df = wave_sum(dfl);
    };

    double tp = a.blo.pendant_default, td = orig * 0.5, tx = orig * 0.5;
    uint32_t evals = 0, rounds = 0, reverted = 0;
    double loglikelihood = -score(td, tx, tp);
    uint32_t smoothings = a.blo.max_rounds;
    // --raxml-blo: pllmod_opt_optimize_branch_lengths_local(radius 1, keep_update 1) on the triplet
    // (optimize.cpp:274-279; pll-modules source absent: restated in oracle/epa_oracle.c opt_local).
    // Per smoothing round: NR on the pendant edge, on the distal edge (inner CLV re-aimed at it),
    // on the proximal edge (with the new distal length), the inner CLV re-aimed at the query, NR
    // on the pendant edge once more from the tip's side, then the edge lnL.  The three lengths are
    // independent (no sliding): Tiny_Tree::place rescales distal by orig / (distal + proximal).
    while (!a.blo.sliding && smoothings) {
      const double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
      auto solve = [&](double cur) -> double {
        double g = cur;
        if (g < xmin || g > xmax) g = a.blo.default_branch;
        const double r = newton_g(deriv, xmin, g, xmax, xtol, (int)a.blo.max_newton, (int)a.blo.newton_variant, evals);
        // keep_update: the length is replaced when the solver moved it (recomp_iterative)
        return (isfinite(r) && fabs(cur - r) > 1e-10) ? r : cur;
      };
      tp = solve(tp);
      side_sumtable(tp, tx, false);
      td = solve(td);
      side_sumtable(tp, td, true);
      tx = solve(tx);
      (void)score(td, tx, tp);          // inner CLV back toward the query + the pendant sumtable
      tp = solve(tp);
      const double new_ll = -lnl_from_slab(tp);
      ++rounds;
      --smoothings;
      if (fabs(new_ll - loglikelihood) < a.blo.epsilon) smoothings = 0;
      loglikelihood = new_ll;
    }
    while (a.blo.sliding && smoothings) {
      const double old_td = td, old_tp = tp;
      // ---- NR for the pendant length (optimize.cpp:135-166); the slab holds the pendant sumtable
      double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
      double xguess = tp;
      if (xguess < xmin || xguess > xmax) xguess = a.blo.default_branch;
      double xres = newton_g(deriv, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, (int)a.blo.newton_variant, evals);
      if (xres > 0.0) tp = xres;
      // ---- NR for the distal length with the proximal P-matrix held fixed (:170-211)
      side_sumtable(tp, tx, false);
      xguess = td;
      xmin = fmin(a.blo.min_branch / 2.0, orig / 2.0);
      xtol = xmin / 10.0;
      xmax = orig - xtol;
      if (xguess < xmin || xguess > xmax) xguess = orig / 2.0;
      xres = newton_g(deriv, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, (int)a.blo.newton_variant, evals);
      if (xres > 0.0) { td = xres; tx = orig - xres; }
      // ---- score (:217-222)
      const double new_ll = -score(td, tx, tp);
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
      if (!isfinite(lnl)) {
        if (atomicAdd(&a.stats[3], 1ull) == 0) a.stats[4] = ((unsigned long long)b << 32) | q;
      }
    }
    wrounds += rounds; wevals += evals; wreverts += reverted;
  }
  if (lane == 0) {
    atomicAdd(&a.stats[0], (unsigned long long)wrounds);
    atomicAdd(&a.stats[1], (unsigned long long)wevals);
    atomicAdd(&a.stats[2], (unsigned long long)wreverts);
  }
}

}  // namespace

int launch_thorough_generic(epa_ctx* ctx, const epa_pair* d_pairs, uint64_t n_pairs, const uint8_t* d_codes,
                            const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                            epa_result* d_out, unsigned long long* d_stats, const uint32_t* d_order, bool caller_times) {
  ThArgsG a;
  a.m = ctx->dmodel;
  a.blo = ctx->blo;
  a.refT = ctx->refT;
  a.scSum = ctx->scSum;
  a.cinv = ctx->cinv;
  a.inv_w0 = ctx->inv_w0;
  a.blen = ctx->blen;
  a.pairs = d_pairs;
  a.order = d_order;   // pair indices of this launch (one span class of a mixed call), or all pairs
  a.codes = d_codes;
  a.crel = ctx->code_stride ? 1u : 0u;
  a.cstride = a.crel ? ctx->code_stride : ctx->W;
  a.win_begin = d_begin;
  a.win_span = d_span;
  a.out = d_out;
  a.stats = d_stats;
  a.n_pairs = n_pairs;
  a.W = ctx->W;
  a.Wpad = (std::max(max_span, 1u) + 63) / 64 * 64;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(n_pairs, (uint64_t)ctx->n_cu * 8);
  const size_t per = (size_t)(ctx->c * ctx->s + 1) * a.Wpad;
  a.slab = (double*)epa_scratch(ctx, 7, sizeof(double) * per * grid);
  if (!a.slab) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(generic sumtable scratch)");
  const bool timed = !d_order && !caller_times;   // a per-class launch is timed by its caller
  if (timed) epa_timer_start(ctx, epa_t(ctx, epa_ctx::T_THOROUGH));
  if (ctx->s == 4) hipLaunchKernelGGL(k_thorough_generic<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_thorough_generic<20>, dim3(grid), dim3(64), 0, ctx->stream, a);
  if (timed) epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_THOROUGH));
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}
