// Hot loop 2 on device, 20-state models with 4 rate categories.
//
// Same algorithm and control flow as k_thorough_dna (see thorough_dna.hip for the mapping to
// the reference: Tiny_Tree::place -> optimize_branch_triplet -> opt_branch_lengths_pplacer,
// src/core/pll/optimize.cpp:60-286, and the eigenbasis reformulation), different geometry:
//
//   workgroup = one (branch, query) pair, 4 wavefronts = the 4 rate categories,
//   lane      = alignment site of the query's window (64 sites per pass),
//   U, U^-1   = wave-uniform operands fetched through the scalar cache (s_load -> SGPR operands
//               of v_fma_f64); the 20x20 contraction is 400 FMAs per (site, category),
//   sumtable  = 20 doubles per (site, category): too large for registers at AA window lengths,
//               it lives in a per-workgroup slab of HBM scratch (component-major, L2 resident:
//               80 x n x 8 B = 64 KB at n = 100) and is streamed once per Newton evaluation,
//   cross-category sums (site likelihood and derivatives, the "all 80 entries < 2^-256"
//   rescale test) go through LDS with one workgroup barrier.
//
// Why not v_mfma_f64_16x16x4_f64 for the 20x20 contraction: on MI355X the fp64 matrix peak equals
// the fp64 vector peak (78.6 TFLOP/s), and M = 20 has to be padded to 32 (two 16-row tiles), so
// the MFMA formulation needs 1.6x the cycles of the VALU one (30 MFMA x 64 cycles vs 1200 FMA x
// 4 cycles per 16 sites and category).  DESIGN.md section 4.2 keeps the arithmetic.
#include "epa_dev_internal.hpp"
#include "wave_util.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

using namespace epa_wave;

constexpr int S = 20;
constexpr int C = 4;
constexpr double LOG_2 = 0.6931471805599453094;

struct ThArgsAA {
  const ModelDev* m;
  BloConsts blo;
  const double* refT;      // [2B][80][W]
  const double* refI;      // [B][80][W] U^-1 inner CLV at the starting lengths (k_build_lookup), or null
  const uint8_t* resc0;    // [B][W]     its per-site rescale flag
  const double* cinv;      // +I: [W] p * pi_inv per site, or null
  double inv_w0;           // +I: 1 / w_0 (folded into sumtable entry (category 0, eigen index 0))
  const uint32_t* scSum;   // [B][W]
  const double* blen;      // [B]
  const epa_pair* pairs;
  const uint32_t* order;   // pair indices of this launch (span class), or null = 0..n_pairs-1
  const uint8_t* codes;    // [Q][cstride]; window at +begin (crel == 0, cstride == W) or at 0 (compact)
  uint32_t cstride, crel;
  const uint32_t* win_begin;
  const uint32_t* win_span;
  epa_result* out;
  unsigned long long* stats;
  double* sscratch;        // [gridDim.x][80][Wpad]
  uint64_t n_pairs;
  uint32_t W;
  uint32_t Wpad;
};


struct Shared {
  double tab[3][80];      // wave-uniform exp tables, [slot][k*20 + x]
  double red[2][3][C][64];  // per-site per-category partial sums of (up to) two passes
  double bc[4];             // broadcast scalars (lnL; f, f' partials of the two ratio waves)
  double lnl_acc;
};

// y[i] = sum_x M[i*20 + x] * v[x]  with M wave-uniform (scalar loads)
typedef const __attribute__((address_space(4))) double* ConstD;

// 0 in an SGPR that the optimiser cannot see through: added to a constant-space pointer it pins
// the scalar loads behind it to this point of the program (they are invariant, and LICM would
// otherwise hoist all 800 matrix words of U / U^-1 to kernel entry and spill them).
__device__ __forceinline__ uint32_t szero(uint32_t dep) {
  uint32_t z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z) : "s"(dep));
  return z;
}
// same, data-dependent on a vector value: a scalar load addressed through it cannot be issued
// before `dep` has been computed (one v_readfirstlane per use)
__device__ __forceinline__ uint32_t szero_after(double dep) {
  return szero((uint32_t)__builtin_amdgcn_readfirstlane(__double2loint(dep)));
}
__device__ __forceinline__ void matvec20(ConstD M, const double (&v)[S], double (&y)[S]) {
#pragma unroll
  for (int i = 0; i < S; ++i) {
    double acc = M[i * S] * v[0];
#pragma unroll
    for (int x = 1; x < S; ++x) acc = fma(M[i * S + x], v[x], acc);
    y[i] = acc;
  }
}

// The model block is never written by any kernel: reading it through the constant address space
// lets hipcc use scalar loads (s_load -> SGPR operands) for the wave-uniform U / U^-1 rows
// instead of 64-lane vector loads of one address.
typedef const __attribute__((address_space(4))) ModelDev* ConstModel;

// LDS_SLAB: the pair's sumtable slab lives in dynamic LDS (80 x Wpad doubles, Wpad = window
// length rounded up to 2, + 2; two workgroups per CU fit up to Wpad = 104); otherwise in HBM
// scratch.
template <bool LDS_SLAB>
__global__ void __launch_bounds__(256, 2) k_thorough_aa(const ThArgsAA a) {
  __shared__ Shared sh;
  extern __shared__ double dyn_slab[];
  ConstModel m = (ConstModel)(a.m);
  // U / U^-1 base pointers as stand-alone SGPR pairs: taken straight from the kernel-argument
  // block they stay part of an 8-register tuple that is spilled and reloaded whole (16
  // v_readlane per matrix row).
  ConstD Ub, Uib;
  {
    auto uniform64 = [](uint64_t v) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
      return ((uint64_t)hi << 32) | lo;
    };
    uint64_t r0 = uniform64((uint64_t)(uintptr_t)m->U), r1 = uniform64((uint64_t)(uintptr_t)m->Ui);
    asm volatile("" : "+s"(r0), "+s"(r1));
    Ub = (ConstD)r0;
    Uib = (ConstD)r1;
  }
  const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6;  // k = rate category of this wave
  // this wave's category plane of the slab, [x][site]
  double* Sglob = LDS_SLAB ? nullptr : a.sscratch + (size_t)blockIdx.x * 80 * a.Wpad + (size_t)k * S * a.Wpad;
  const uint32_t slab_k = k * S * a.Wpad;
  auto slab = [&](uint32_t idx) -> double& {
    if constexpr (LDS_SLAB) return dyn_slab[slab_k + idx];
    else return Sglob[idx];
  };
  // per-thread table constants: thread t < 240 owns (slot = t / 80, kx = t % 80)
  const int tslot = tid / 80, tkx = tid % 80;
  double t_lr = 0.0, t_w = 0.0;
  if (tid < 240) {
    t_lr = m->lam[tkx % S] * m->rate[tkx / S];
    t_w = m->w[tkx / S];
  }
  // Newton tables of this wave's own category: lane = slot * 20 + x (lanes 60..63 idle)
  double tl_lr = 0.0, tl_c = 0.0;
  if (lane < 60) {
    const int x = lane % S, slot = lane / S;
    tl_lr = m->lam[x] * m->rate[k];
    tl_c = slot == 0 ? m->w[k] : (slot == 1 ? m->w[k] * tl_lr : m->w[k] * tl_lr * tl_lr);
  }

  uint32_t wrounds = 0, wevals = 0, wreverts = 0;
  for (uint64_t pidx = blockIdx.x; pidx < a.n_pairs; pidx += gridDim.x) {
    const uint64_t pid = a.order ? a.order[pidx] : pidx;
    const epa_pair pr = a.pairs[pid];
    const uint32_t b = pr.branch_id, q = pr.seq_id;
    const uint32_t begin = a.win_begin[q], n = a.win_span[q];
    const size_t cW = a.W;
    const double* Xt = a.refT + ((size_t)(2 * b) * 80 + (size_t)k * S) * cW + begin;
    const double* Dt = a.refT + ((size_t)(2 * b + 1) * 80 + (size_t)k * S) * cW + begin;
    const uint32_t* scp = a.scSum + (size_t)b * cW + begin;
    const uint8_t* qc = a.codes + (size_t)q * a.cstride + (a.crel ? 0u : begin);
    const double orig = a.blen[b];
    const uint32_t npass = (n + 63) / 64;

    // ---- table publication: every thread < 240 computes one exp()
    auto publish = [&](double t0, double t1, double t2) {
      __syncthreads();  // previous readers of sh.tab are done
      if (tid < 240) {
        const double t = tslot == 0 ? t0 : (tslot == 1 ? t1 : t2);
        const double e = exp(t_lr * t);
        sh.tab[tslot][tkx] = tslot == 2 ? e * t_w : e;
      }
      __syncthreads();
    };

    // Inner CLV toward the query (mode 0: A = distal, lnL computed) or toward distal (mode 1:
    // A = query tip, sumtable folded with D).  Writes this category's sumtable slab.
    auto phase = [&](int mode, double& lnl_out) {
      double mant = 1.0;
      int ex = 0;
      // Reference operands of a pass (X = proximal, Dv = distal, 20 + 20 doubles per lane) are
      // requested as one batch: left alone hipcc interleaves load / wait / multiply one value at
      // a time.  (Requesting the next pass's batch under the U^-1 loop was tried: the extra 80
      // live registers spill and the kernel gets 25% slower.)
      double Xv[S], Dv[S];
      auto request = [&](uint32_t pp, bool with_d) {
        const uint32_t st = pp * 64 + lane;
        const uint32_t sx = st < n ? st : 0;
#pragma unroll
        for (int x = 0; x < S; ++x) Xv[x] = Xt[(size_t)x * cW + sx];
        if (with_d) {
#pragma unroll
          for (int x = 0; x < S; ++x) Dv[x] = Dt[(size_t)x * cW + sx];
        }
        asm volatile("" ::: "memory");
      };
      for (uint32_t p = 0; p < npass; ++p) {
        const uint32_t site = p * 64 + lane;
        const bool valid = site < n;
        const uint32_t s = valid ? site : 0;
        double E[S], It[S];
        bool resc;
        double mult;
        const uint32_t code = qc[s];  // its latency hides under the requests below
        if (mode == 2) {
          // starting lengths: U^-1 inner CLV of this (branch, site) from the per-branch precompute
          const double* It0 = a.refI + ((size_t)b * 80 + (size_t)k * S) * cW + begin;
#pragma unroll
          for (int x = 0; x < S; ++x) It[x] = It0[(size_t)x * cW + s];
#pragma unroll
          for (int x = 0; x < S; ++x) E[x] = m->qt[code * S + x];
          resc = a.resc0[(size_t)b * cW + begin + s] != 0;
          asm volatile("" ::: "memory");
          mult = 1.0;  // already applied
        } else {
        double A[S], Q[S], I[S];
        request(p, true);
#pragma unroll
        for (int x = 0; x < S; ++x) Q[x] = m->qt[code * S + x];
        asm volatile("" ::: "memory");
        // mode 0: A = distal, the query tip is folded in at the end; mode 1: the other way round
#pragma unroll
        for (int x = 0; x < S; ++x) {
          A[x] = (mode == 0 ? Dv[x] : Q[x]) * sh.tab[0][k * S + x];
          E[x] = mode == 0 ? Q[x] : Dv[x];
          Xv[x] *= sh.tab[1][k * S + x];
        }
        // I_i = (U A)_i (U X)_i : one pass over the rows of U feeds both products (each row is
        // fetched once through the scalar cache and used for 40 FMAs)
        // Row i+1 of U is fetched (s_load_dwordx16 x2.5 -> 40 SGPRs) while row i is being used;
        // the sched_barrier keeps hipcc from hoisting all 20 rows (800 SGPRs -> spills to VGPR
        // lanes).
        double mx = 0.0;
        double ur[2][S];
        {
          ConstD row = Ub + szero(p);
#pragma unroll
          for (int x = 0; x < S; ++x) ur[0][x] = row[x];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
          // scalar loads complete out of order, so the only wait is lgkmcnt(0): the first use of
          // row i (which waits) has to come BEFORE row i+1 is requested, or every row would wait
          // for the request issued right in front of it.
          double pa = ur[i & 1][0] * A[0], pb = ur[i & 1][0] * Xv[0];
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < S) {
            ConstD row = Ub + (i + 1) * S + szero_after(pa);
#pragma unroll
            for (int x = 0; x < S; ++x) ur[(i + 1) & 1][x] = row[x];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int x = 1; x < S; ++x) {
            pa = fma(ur[i & 1][x], A[x], pa);
            pb = fma(ur[i & 1][x], Xv[x], pb);
          }
          I[i] = pa * pb;
          mx = fmax(mx, I[i]);
          asm volatile("" ::"v"(I[i]));  // pins row i's FMAs here (LLVM would sink them to their use)
          __builtin_amdgcn_sched_barrier(0);
        }
        // per-site rescale: ALL 80 entries (4 categories = 4 waves) below 2^-256.  sh.red is
        // double-buffered by pass parity, so one barrier per exchange is enough.
        sh.red[p & 1][0][k][lane] = mx;
        __syncthreads();
        const double mall = fmax(fmax(sh.red[p & 1][0][0][lane], sh.red[p & 1][0][1][lane]),
                                 fmax(sh.red[p & 1][0][2][lane], sh.red[p & 1][0][3][lane]));
        resc = mall < 0x1p-256;
        mult = resc ? 0x1p+256 : 1.0;
        {
          ConstD row = Uib + szero(p);
#pragma unroll
          for (int x = 0; x < S; ++x) ur[0][x] = row[x];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
          double acc = ur[i & 1][0] * I[0];
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < S) {
            ConstD row = Uib + (i + 1) * S + szero_after(acc);
#pragma unroll
            for (int x = 0; x < S; ++x) ur[(i + 1) & 1][x] = row[x];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int x = 1; x < S; ++x) acc = fma(ur[i & 1][x], I[x], acc);
          It[i] = acc;
          asm volatile("" ::"v"(It[i]));
          __builtin_amdgcn_sched_barrier(0);
        }
        }  // mode != 2
        double l0 = 0.0;
        // LDS slab: lanes past the window write to the spare last column (an `if (valid)` around
        // the stores would let LLVM sink the whole U^-1 product into the branch and wreck the
        // scalar-load pipeline above)
        const uint32_t wsite = (LDS_SLAB && !valid) ? a.Wpad - 1 : site;
#pragma unroll
        for (int x = 0; x < S; ++x) {
          double sv = It[x] * mult * E[x];
          // +I: p * pi_inv enters L_0 only: eigenvalue 0 is exactly 0, its order-1/2 table entries
          // vanish, so the term is folded into the sumtable entry (category 0, eigen index 0)
          if (x == 0 && k == 0 && a.cinv) sv += a.cinv[begin + s] * a.inv_w0;
          slab(x * a.Wpad + wsite) = sv;
          if (mode != 1) l0 = fma(sv, sh.tab[2][k * S + x], l0);
        }
        if (mode != 1) {
          sh.red[p & 1][1][k][lane] = l0;
          __syncthreads();
          if (k == 0) {
            double ls = (sh.red[p & 1][1][0][lane] + sh.red[p & 1][1][1][lane]) +
                        (sh.red[p & 1][1][2][lane] + sh.red[p & 1][1][3][lane]);
            if (!valid) ls = 1.0;
            const int sc = valid ? (int)(scp[s] + (resc ? 1u : 0u)) : 0;
            mant *= __builtin_amdgcn_frexp_mant(ls);
            ex += __builtin_amdgcn_frexp_exp(ls) - 256 * sc;
            ex += __builtin_amdgcn_frexp_exp(mant);
            mant = __builtin_amdgcn_frexp_mant(mant);
          }
        }
      }
      if (mode != 1) {
        if (k == 0) {
          const double tot = wave_sum(log(mant) + (double)ex * LOG_2);
          if (lane == 0) sh.bc[2] = tot;
        }
        __syncthreads();
        lnl_out = sh.bc[2];
      }
      __syncthreads();  // sh.red / sh.bc / the slab are consistent for whoever comes next
    };

    // f, f' at proposal t from this pair's sumtable slab.  Each wave builds the 60 table entries
    // of its own category in one VGPR pair (lane = slot * 20 + x, one exp per lane) and feeds them
    // to the FMAs as SGPR operands through v_readlane: no LDS traffic for the tables and no
    // workgroup barrier to publish them.  Two 64-site passes are handled per barrier; the
    // per-site ratio of pass j is done by wave j.
    auto derivatives = [&](double t, double& f, double& df) {
      const double ev = exp(tl_lr * t) * tl_c;
      double fl = 0.0, dfl = 0.0;
      constexpr uint32_t PG = LDS_SLAB ? 2 : 1;  // passes per barrier (HBM slab: register budget)
      for (uint32_t p0 = 0; p0 < npass; p0 += PG) {
        const bool two = PG == 2 && p0 + 1 < npass;
        const uint32_t site0 = p0 * 64 + lane, site1 = site0 + 64;
        const uint32_t i0 = site0 < n ? site0 : 0, i1 = site1 < n ? site1 : 0;
        double sv0[S], sv1[S];
#pragma unroll
        for (int x = 0; x < S; ++x) sv0[x] = slab(x * a.Wpad + i0);
        if (two) {
#pragma unroll
          for (int x = 0; x < S; ++x) sv1[x] = slab(x * a.Wpad + i1);
        } else {
#pragma unroll
          for (int x = 0; x < S; ++x) sv1[x] = 0.0;
        }
        asm volatile("" ::: "memory");
        double l00 = 0.0, l01 = 0.0, l02 = 0.0, l10 = 0.0, l11 = 0.0, l12 = 0.0;
#pragma unroll
        for (int x = 0; x < S; ++x) {
          const double e0 = readlane_d(ev, x), e1 = readlane_d(ev, S + x), e2 = readlane_d(ev, 2 * S + x);
          l00 = fma(sv0[x], e0, l00);
          l01 = fma(sv0[x], e1, l01);
          l02 = fma(sv0[x], e2, l02);
          if (PG == 2) {
            l10 = fma(sv1[x], e0, l10);
            l11 = fma(sv1[x], e1, l11);
            l12 = fma(sv1[x], e2, l12);
          }
        }
        sh.red[0][0][k][lane] = l00; sh.red[0][1][k][lane] = l01; sh.red[0][2][k][lane] = l02;
        if (PG == 2) { sh.red[1][0][k][lane] = l10; sh.red[1][1][k][lane] = l11; sh.red[1][2][k][lane] = l12; }
        __syncthreads();
        if (k < (int)PG && (k == 0 ? site0 : site1) < n) {
          const double s0 = (sh.red[k][0][0][lane] + sh.red[k][0][1][lane]) + (sh.red[k][0][2][lane] + sh.red[k][0][3][lane]);
          const double s1 = (sh.red[k][1][0][lane] + sh.red[k][1][1][lane]) + (sh.red[k][1][2][lane] + sh.red[k][1][3][lane]);
          const double s2 = (sh.red[k][2][0][lane] + sh.red[k][2][1][lane]) + (sh.red[k][2][2][lane] + sh.red[k][2][3][lane]);
          const double inv = fast_rcp(s0);
          const double d1 = -s1 * inv;
          fl += d1;
          dfl += fma(d1, d1, -s2 * inv);
        }
        if (p0 + PG < npass) __syncthreads();  // sh.red is rewritten by the next group of passes
      }
      if (k < 2) {  // wave 1 holds zeros when PG == 1
        const double ft = wave_sum(fl), dft = wave_sum(dfl);
        if (lane == 0) { sh.bc[k] = ft; sh.bc[2 + k] = dft; }
      }
      __syncthreads();
      f = sh.bc[0] + sh.bc[1];
      df = sh.bc[2] + sh.bc[3];
      __syncthreads();  // sh.bc / sh.red are free again
    };

    // pllmod_opt_minimize_newton (rtsafe-style), uniform across the workgroup
    uint32_t evals = 0;
    auto newton = [&](double x1, double xguess, double x2, double tol, int max_iters) -> double {
      double rts = xguess, f, df, xl, xh, dx;
      if (rts < x1) rts = x1;
      if (rts > x2) rts = x2;
      derivatives(rts, f, df);
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
        derivatives(rts, f, df);
        ++evals;
        if (!isfinite(f) || !isfinite(df)) return NAN;
        if (df > 0.0 && fabs(f) < tol) return rts;
        if (f < 0.0) xl = rts; else xh = rts;
      }
      return NAN;
    };

    double tp = a.blo.pendant_default, td = orig * 0.5, tx = orig * 0.5;
    uint32_t rounds = 0, reverted = 0;
    double lnl_now = 0.0;
    publish(td, tx, tp);
    if (a.refI) phase(2, lnl_now); else phase(0, lnl_now);
    double loglikelihood = -lnl_now;
    uint32_t smoothings = a.blo.max_rounds;
    while (smoothings) {
      const double old_td = td, old_tp = tp;
      double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
      double xguess = tp;
      if (xguess < xmin || xguess > xmax) xguess = a.blo.default_branch;
      double xres = newton(xmin, xguess, xmax, xtol, (int)a.blo.max_newton);
      if (xres > 0.0) tp = xres;
      publish(tp, tx, tp);
      double dummy;
      phase(1, dummy);
      xguess = td;
      xmin = fmin(a.blo.min_branch / 2.0, orig / 2.0);
      xtol = xmin / 10.0;
      xmax = orig - xtol;
      if (xguess < xmin || xguess > xmax) xguess = orig / 2.0;
      xres = newton(xmin, xguess, xmax, xtol, (int)a.blo.max_newton);
      if (xres > 0.0) { td = xres; tx = orig - xres; }
      publish(td, tx, tp);
      phase(0, lnl_now);
      const double new_ll = -lnl_now;
      ++rounds;
      if (new_ll - loglikelihood > new_ll * 1e-14) {
        tp = old_tp; td = old_td; tx = orig - old_td;
        reverted = 1;
        break;
      }
      --smoothings;
      if (fabs(new_ll - loglikelihood) < a.blo.epsilon) smoothings = 0;
      loglikelihood = new_ll;
    }
    if (tid == 0) {
      const double lnl = -loglikelihood;
      epa_result r;
      r.lnl = lnl;
      r.pendant_length = tp;
      r.distal_length = (orig / (td + tx)) * td;
      a.out[pid] = r;
      if (!isfinite(lnl)) {
        if (atomicAdd(&a.stats[3], 1ull) == 0) a.stats[4] = ((unsigned long long)b << 32) | q;
      }
    }
    wrounds += rounds; wevals += evals; wreverts += reverted;
  }
  if (tid == 0) {
    atomicAdd(&a.stats[0], (unsigned long long)wrounds);
    atomicAdd(&a.stats[1], (unsigned long long)wevals);
    atomicAdd(&a.stats[2], (unsigned long long)wreverts);
  }
}

}  // namespace

int launch_thorough_aa(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* d_order, uint64_t n_pairs,
                       const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span,
                       uint32_t max_span, bool want_lds, epa_result* d_out, unsigned long long* d_stats) {
  if (ctx->c != 4)
    return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "thorough (20 states): needs 4 rate categories");
  ThArgsAA a;
  a.m = ctx->dmodel;
  a.blo = ctx->blo;
  a.refT = ctx->refT;
  a.refI = ctx->lookup_built ? ctx->refI : nullptr;
  a.resc0 = ctx->resc0;
  a.cinv = ctx->cinv;
  a.inv_w0 = ctx->inv_w0;
  a.scSum = ctx->scSum;
  a.blen = ctx->blen;
  a.pairs = d_pairs;
  a.order = d_order;
  a.codes = d_codes;
  a.crel = ctx->code_stride ? 1u : 0u;
  a.cstride = a.crel ? ctx->code_stride : ctx->W;
  a.win_begin = d_begin;
  a.win_span = d_span;
  a.out = d_out;
  a.stats = d_stats;
  a.n_pairs = n_pairs;
  a.W = ctx->W;
  // 2 resident workgroups per CU; the grid is oversubscribed so that the hardware dispatcher
  // balances the 10x cost spread of the pairs (a finished workgroup's slot is refilled at once)
  // (measured at 25.7k pairs: x1 6.47 ms, x4 6.2, x8 6.04, x16 6.0); the HBM-slab variant owns
  // 80 x Wpad doubles of scratch per workgroup and stays at x2
  const bool lds_slab = want_lds && max_span <= EPA_AA_LDS_MAX_SPAN;
  uint32_t per_slot = lds_slab ? 16 : 2;
  uint32_t nwg = (uint32_t)std::min<uint64_t>(n_pairs, (uint64_t)512 * per_slot);
  const uint32_t wpad_lds = (max_span + 1) / 2 * 2 + 2;  // + spare column for lanes past the window
  // two workgroups per CU: static + dynamic LDS <= 80 KB each
  static_assert(sizeof(Shared) + sizeof(double) * 80 * ((EPA_AA_LDS_MAX_SPAN + 1) / 2 * 2 + 2) <= 80 * 1024,
                "EPA_AA_LDS_MAX_SPAN does not fit two workgroups per CU");
  if (lds_slab) {
    a.Wpad = wpad_lds;
    a.sscratch = nullptr;
  } else {
    a.Wpad = (max_span + 63) / 64 * 64;
    a.sscratch = (double*)epa_scratch(ctx, 7, sizeof(double) * (size_t)nwg * 80 * a.Wpad);
    if (!a.sscratch) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(AA sumtable scratch)");
  }
  if (lds_slab) {
    const int dyn = (int)(sizeof(double) * 80 * a.Wpad);
    EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_thorough_aa<true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
  }
  if (lds_slab)
    hipLaunchKernelGGL(k_thorough_aa<true>, dim3(nwg), dim3(256), sizeof(double) * 80 * a.Wpad, ctx->stream, a);
  else
    hipLaunchKernelGGL(k_thorough_aa<false>, dim3(nwg), dim3(256), 0, ctx->stream, a);
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}
