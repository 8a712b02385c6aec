// Hot loop 2 on device, 4-state models (3 .. 16 rate categories): one wavefront per (branch, query)
// pair and group of four rate categories.
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
//  * lane = alignment site of the query's window (NCH 64-site chunks per lane, S in VGPRs); f, f'
//    and lnL are DPP wave reductions; the wave-uniform numbers of a Newton proposal or a phase are
//    computed one per lane (table-driven exp, wave_util.hpp), published in a 64-entry LDS table
//    and read back with uniform-address ds_read.
//  * a phase (inner vector + sumtable of a window) is ONE software pipeline over (chunk, category)
//    steps: the 8 operand loads of step i + TH_STREAM_DEPTH are requested before step i is computed.
//    U / U^-1 / weights are re-read per phase from the kernarg segment (scalar loads) instead of
//    occupying 72 scalar registers across the Newton loops.
//  * no P-matrix, no sumtable, no inner CLV ever touches memory.
#include "epa_dev_internal.hpp"
#include "wave_util.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

// rocprim's radix sort falls back to a merge sort (block sort + log2 n merge passes: ~19 launches for
// the 262k candidate keys of a chunk) below one million items; the sorts here use a few key bits only
// (branch id, window start, span class), where Onesweep digit passes do: 4 - 5 launches.
using epa_radix_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                 rocprim::default_config, 0>;

#include <algorithm>
#include <vector>
#include <cstdio>
#include <type_traits>
#include <cstdlib>

namespace {


struct ThArgs {
  ModelDNA m;
  BloConsts blo;
  const double* refT;      // [2B][16][W]
  const double* refI;      // [B][16][W]  U^-1 inner CLV at the starting lengths (k_build_lookup)
  const uint8_t* resc0;    // [B][W]      its per-site rescale flag
  const double* cinv;      // +I: [W] p * pi_inv per site, or null
  double inv_w0;           // +I: 1 / w_0
  const uint32_t* scSum;   // [B][W]
  const double* blen;      // [B]
  const double* qt;        // [16 columns][4]   U^-1 image of each column's tip vector
  const epa_pair* pairs;
  const uint32_t* order;   // pair indices of this launch (span class), or null = 0..n_pairs-1
  const uint8_t* codes;    // [Q][cstride]; window at +begin (crel == 0, cstride == W) or at 0 (compact)
  uint32_t cstride, crel;
  const uint32_t* win_begin;
  const uint32_t* win_span;
  epa_result* out;
  unsigned long long* stats;  // [0] rounds [1] newton evals [2] reverts [3] non-finite [4] first bad
  uint32_t* qctr;          // single-wave classes: one work counter per XCD slice (dynamic pair fetch), or null
  double* sscratch;        // NCH == 0 only: [waves][16][Wpad]
  uint64_t n_pairs;
  uint32_t W;
  uint32_t Wpad;
  // queued launch (launch_thorough_queued): the pair count is not known on the host yet.  spec = the selection's
  // packed read-back block ON THE DEVICE (k_pack_readback: [0] total, [9 + c] pairs of span class c, [32], [33]
  // window-validation errors); the kernel runs only if the host, once it has read the same block, will find
  // that this launch was the right one (launch_select_end / chunk_body_end apply the same test), else it exits
  const uint32_t* spec;
  uint32_t spec_cls, spec_max;
  // single-wave classes: XCD x takes the pairs [n * xcum[x], n * xcum[x + 1]) >> 20 of the branch-sorted list.  The
  // shares follow the speed each XCD showed in the context's previous launches (epa_xcd_feedback): in-kernel stamps
  // showed the eight equal slices of a 262k-pair launch draining up to 250 us apart, the same XCDs early / late from
  // launch to launch and other ones on another box.  xstamp != 0: the launch records stats[7] = start and
  // stats[8 + x] = the last exit of XCD x's waves (s_memrealtime, 100 MHz, << 21) | the XCD's share in this launch
  // for that feedback; stats[5] / stats[6] = shader cycles / 100 MHz ticks of workgroup 0's first wave (the clock
  // the launch ran at: epa_dev_last_sclk_mhz).
  uint32_t xcum[9];
  uint32_t xstamp;
};

using namespace epa_wave;

// Orders the NEXT chunk's loads after THIS chunk's arithmetic: returns 0 but the compiler must
// assume it depends on `v`.  Without it the 32 loads of every chunk (and of every later phase)
// are all issued up front and live ranges explode into scratch.
__device__ __forceinline__ uint32_t zero_after(double v) {
  uint32_t z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(__double2loint(v)));
  return z;
}

// The kernel's argument block, re-read through an opaque pointer: U, U^-1 and the weights (72 scalar
// registers) are needed inside the phases only.  Held as ordinary kernel arguments they stay live
// across the Newton loops, and the compiler parks ~270 scalar registers in vector-register lanes
// (v_writelane / v_readlane around every use).  A phase instead reloads them from the kernarg
// segment with a few s_load_dwordx16 (scalar cache hits) -- the asm makes the pointer unknown to the
// optimiser, so the loads cannot be merged with earlier ones or kept alive after the phase.
using KArgs = const __attribute__((address_space(4))) ThArgs*;
__device__ __forceinline__ KArgs kargs_fresh() {
  KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
struct PhaseModel {   // what a phase needs of ModelDNA, in scalar registers for the phase's duration
  double U[16], Ui[16], w[4];
  __device__ __forceinline__ void load(uint32_t grp) {
    KArgs p = kargs_fresh();
#pragma unroll
    for (int i = 0; i < 16; ++i) { U[i] = p->m.U[i]; Ui[i] = p->m.Ui[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = p->m.w[grp * 4 + i];
  }
};

// ---- wave-uniform tables through LDS: every lane computes ONE exp(), writes it to the
// workgroup's (= wave's) 64-entry table, then all lanes read the entries they need with
// uniform-address ds_read (broadcast, no bank conflicts).  Lane l holds pair kx = l & 15
// (k = rate category, x = eigenvalue index) and "slot" l >> 4.
struct LaneConst {
  double lr;   // lambda_x * r_k
  double cN;   // Newton table coefficient: w_k * lr^slot (slot 0,1,2), 0 for slot 3
  double w;    // w_k
  int slot;
  int npos;    // ZERO0: this lane's entry of the COMPACT Newton table [order][category][x = 1..3] (36
               // doubles, read back as 18 aligned ds_read_b128); lanes of eigen index 0 / slot 3: a dump slot
  const double* e2t;  // LDS: 2^(j/64), j = 0..63 (exp_tab)
};

// 16-byte LDS reads of table entries: ds_read_b128 moves 256 B/clk/CU, the ds_read2_b64 the compiler
// picks for unaligned pairs 128 (MI355X_MICROARCH.md, LDS table) -- and the tables are what keeps the
// LDS busy in this kernel (36 wave-uniform doubles per Newton evaluation)
__device__ __forceinline__ void lds_pair(const double* p, double& a, double& b) {
  const double2 v = *reinterpret_cast<const double2*>(__builtin_assume_aligned(p, 16));
  a = v.x;
  b = v.y;
}

__device__ __forceinline__ void table_publish(double* tab, int lane, double v) {
  tab[lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one site: I_ki = (U (ea_k o A_k))_i * (U (eb_k o Bv_k))_i, per-site rescale, return U^-1 I
// A, Bv: eigen-space vectors [k][x]; ea/eb: exp tables.  resc: 1 if rescaled.
__device__ __forceinline__ void inner_site(const ModelDNA& m, const double (&A)[16], const double* ea,
                                           const double (&Bv)[16], const double* eb, double (&It)[16],
                                           uint32_t& resc) {
  double I[16];
  double mx = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double av[4], bv[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      av[x] = A[k * 4 + x] * ea[k * 4 + x];
      bv[x] = Bv[k * 4 + x] * eb[k * 4 + x];
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

// ---- "tail half-chunk" (TAILH instantiations): the last 64-lane chunk of a window that ends within
// 32 sites of a chunk boundary (150-site reads: 64 + 64 + 22) is laid out as lane = (site, half) --
// lanes 0..31 carry categories 0, 1 of sites 0..31 of the chunk, lanes 32..63 categories 2, 3 of the
// SAME sites -- so the chunk costs half the loads and products of a full one.  What has to see all
// four categories (the rescale test, the site likelihood, l0 / l1 / l2 of a Newton evaluation) is
// combined across the halves with v_permlane32_swap.
__device__ __forceinline__ double xhalf_add(double v) {   // v[l] + v[l ^ 32], in every lane
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// lanes 0..31: x[l] + x[l + 32];  lanes 32..63: y[l - 32] + y[l]
__device__ __forceinline__ double xhalf_add2(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double xhalf_max(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return fmax(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
}
// (A matrix-core form of the phases -- the three 4 x 4 products of a (site, category) step on v_mfma_f64_4x4x4_4b_f64,
// reference rows loaded in the MFMA B layout, the sumtable transposed back with v_permlane16/32_swap -- was built in
// round 4, is parity-green and SLOWER: 5.73 against 5.17 ms per 262k-pair launch; on gfx950 the fp64 matrix instruction
// runs on the vector ALU's fp64 datapath and blocks its issue.  Source: profiles/variants/r5_thorough_dna_mfma_phase.hip
// (-DTH_MFMA_PHASE=1), measurements: profiles/r4_mfma_phase_go_nogo.txt)
constexpr int QA_STRIDE = 20;   // doubles per column code in the per-wave query table: codes A C G T 160 B
                                // apart fall on disjoint LDS banks for the lane = (state, site) gathers
// one category of one site: I_i = (U (e0 o F))_i (U (e1 o G))_i, It = U^-1 I (unscaled; the caller
// keeps the running maximum for the per-site rescale test).  The streamed phases (TH_STREAM_DEPTH)
// walk a window category by category with this.
// mx: running maximum of the HIGH WORDS of the inner vector's entries, as signed integers.  The
// per-site rescale test of pll_update_partials -- every entry < 2^-256 -- is  max_hi < 0x2ff00000
// (2^-256 has a zero low word; negative rounding residues compare below it either way, NaN above):
// two entries per v_max3_i32 instead of one per v_max_f64.
template <class Model>
__device__ __forceinline__ void cat_inner(const Model& m, const double (&F)[4], const double* e0,
                                          const double (&G)[4], const double* e1, double (&It)[4], int& mx) {
  double av[4], bv[4], I[4], ev0[4], ev1[4];
  lds_pair(e0, ev0[0], ev0[1]); lds_pair(e0 + 2, ev0[2], ev0[3]);   // table rows are 32-byte aligned
  lds_pair(e1, ev1[0], ev1[1]); lds_pair(e1 + 2, ev1[2], ev1[3]);
#pragma unroll
  for (int x = 0; x < 4; ++x) { av[x] = F[x] * ev0[x]; bv[x] = G[x] * ev1[x]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double a = m.U[i * 4] * av[0], b = m.U[i * 4] * bv[0];
#pragma unroll
    for (int x = 1; x < 4; ++x) {
      a = fma(m.U[i * 4 + x], av[x], a);
      b = fma(m.U[i * 4 + x], bv[x], b);
    }
    I[i] = a * b;
  }
  mx = max(max(mx, __double2hiint(I[0])), __double2hiint(I[1]));
  mx = max(max(mx, __double2hiint(I[2])), __double2hiint(I[3]));
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    double acc = m.Ui[x * 4] * I[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) acc = fma(m.Ui[x * 4 + i], I[i], acc);
    It[x] = acc;
  }
}

// the same with the first factor given: a_i = (U (e0 o q))_i depends on the query's column code and the
// category only, so the phases toward the distal / proximal node read it from a per-wave table of
// 16 codes x 16 (category, state) entries (built once per phase by the wave's 64 lanes) instead of
// forming it per site: 2 LDS reads replace 20 of the ~66 vector instructions of a category step.
template <class Model>
__device__ __forceinline__ void cat_inner_pre(const Model& m, const double (&Ap)[4], const double (&G)[4],
                                              const double* e1, double (&It)[4], int& mx) {
  double bv[4], I[4], ev1[4];
  lds_pair(e1, ev1[0], ev1[1]); lds_pair(e1 + 2, ev1[2], ev1[3]);
#pragma unroll
  for (int x = 0; x < 4; ++x) bv[x] = G[x] * ev1[x];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double b = m.U[i * 4] * bv[0];
#pragma unroll
    for (int x = 1; x < 4; ++x) b = fma(m.U[i * 4 + x], bv[x], b);
    I[i] = Ap[i] * b;
  }
  mx = max(max(mx, __double2hiint(I[0])), __double2hiint(I[1]));
  mx = max(max(mx, __double2hiint(I[2])), __double2hiint(I[3]));
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    double acc = m.Ui[x * 4] * I[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) acc = fma(m.Ui[x * 4 + i], I[i], acc);
    It[x] = acc;
  }
}

// Prefetch depth of the streamed phases: a phase is the sequence of (64-site chunk, rate category)
// steps of the window; the 8 operand loads of step i + DEPTH are requested before step i is
// computed, ACROSS chunk boundaries, so that one load latency is exposed per phase instead of one
// per chunk.  0 = the batch form (32 loads per chunk up front, chunks serialised).
// TH_EXP(x): the table-driven exp of wave_util.hpp (default) or the library exp (A/B: -DTH_LIBM_EXP)
#ifdef TH_LIBM_EXP
#define TH_EXP(x) exp(x)
#else
#define TH_EXP(x) exp_tab((x), lc.e2t)
#endif
// resident waves per SIMD the register allocation leaves room for (2: 256 VGPRs per lane)
#ifndef TH_WAVES
#define TH_WAVES 2
#endif
#ifndef TH_STREAM_DEPTH
#define TH_STREAM_DEPTH 2
#endif
#ifndef TH_SPEC_STEP
#define TH_SPEC_STEP 0
#endif

template <int NCH>
struct SiteState {
  double S[NCH][16];   // sumtable of the branch currently being optimised.  ZERO0: entry [0] holds
                       // sum_k w_k S_k0 (the zero-eigenvalue terms, constant in t), [4] [8] [12] are dead
  uint32_t sc[NCH];    // proximal + distal scaler counts
  uint32_t resc[NCH];  // rescale flag of the last inner CLV toward the query
  uint32_t code[NCH];  // query column code
  bool valid[NCH];
};

// Windows longer than 3 x 64 sites are spread over the NW wavefronts of a workgroup (wave w owns
// sites [w * NCH * 64, (w + 1) * NCH * 64) of the window, its sumtable stays in ITS registers);
// the per-wave sums of f, f' and lnL are combined through LDS with one barrier per Newton
// evaluation (double-buffered: a wave can be at most one barrier ahead of the slowest one).
// All waves then hold identical scalars and walk the optimiser's control flow in lock step.
// NG > 1 (8 / 12 / 16 rate categories): the workgroup's NG waves are the model's GROUPS OF FOUR
// CATEGORIES instead -- wave g runs the four-category code on categories 4 g .. 4 g + 3 of ALL sites of
// the window (its rows of the reference data, its rates and weights) -- and what a site needs from
// all of its categories crosses the waves through LDS: l0 / l1 / l2 of a Newton evaluation, the
// site likelihood of the window lnL, the maximum of the inner vector for the per-site rescale test.
// Every wave then holds the same per-site values and the same scalars; the sums over the groups are
// formed in group order by all of them.
template <int NW, int NG = 1>
struct Comb {
  double* red;  // LDS [2][NW][2]
  int wv;       // wave of the workgroup: site block (NW > 1) or category group (NG > 1)
  int phase;
  double* xch;  // NG > 1: LDS [2][NG][XCH_MAX][64] per-lane exchange slots
  int xphase;
  static constexpr int XCH_MAX = 12;
  __device__ __forceinline__ void sum2(double& a, double& b, int lane) {
    if constexpr (NW > 1) {
      double* r = red + phase * NW * 2;
      if (lane == 0) { r[wv * 2] = a; r[wv * 2 + 1] = b; }
      __syncthreads();
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sa += r[w * 2]; sb += r[w * 2 + 1]; }
      a = sa;
      b = sb;
      phase ^= 1;
    }
  }
  // v[i] <- sum over the category groups of v[i] (MAXI: maximum of integers carried as doubles),
  // per lane; one workgroup barrier (double-buffered like sum2)
  template <int N, bool MAXI = false>
  __device__ __forceinline__ void groups(double (&v)[N], int lane) {
    if constexpr (NG > 1) {
      static_assert(N <= XCH_MAX, "exchange slots");
      double* x = xch + (size_t)xphase * NG * XCH_MAX * 64;
#pragma unroll
      for (int i = 0; i < N; ++i) x[(wv * XCH_MAX + i) * 64 + lane] = v[i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double acc = x[i * 64 + lane];
#pragma unroll
        for (int g = 1; g < NG; ++g) {
          const double o = x[(g * XCH_MAX + i) * 64 + lane];
          acc = MAXI ? fmax(acc, o) : acc + o;
        }
        v[i] = acc;
      }
      xphase ^= 1;
    }
  }
};

// half-chunk contraction of a Newton evaluation: this half's two categories (2 h, 2 h + 1) against
// the table (ZERO0: compact layout, entries m * 12 + h * 6 + kk * 3 + x - 1, three aligned pairs per order)
template <bool ZERO0>
__device__ __forceinline__ void half_contract(const double (&S)[16], const double* tab, int lane, double& l0,
                                              double& l1, double& l2) {
  if constexpr (ZERO0) {
    const double* th = tab + (lane >> 5) * 6;
    double c[18];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int i = 0; i < 3; ++i) lds_pair(th + 12 * m + 2 * i, c[6 * m + 2 * i], c[6 * m + 2 * i + 1]);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int x = 1; x < 4; ++x) {
        const double sv = S[kk * 4 + x];
        l0 = fma(sv, c[kk * 3 + x - 1], l0);
        l1 = fma(sv, c[6 + kk * 3 + x - 1], l1);
        l2 = fma(sv, c[12 + kk * 3 + x - 1], l2);
      }
  } else {
    const double* th = tab + ((lane >> 5) << 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      l0 = fma(S[i], th[i], l0);
      l1 = fma(S[i], th[16 + i], l1);
      l2 = fma(S[i], th[32 + i], l2);
    }
  }
}

// Newton tables for proposal t: e = w exp(lr t), e1 = w lr exp(lr t), e2 = w lr^2 exp(lr t).
// ZERO0: eigenvalue 0 is exactly 0 (stationary mode) -> its e1/e2 columns vanish.
// Then  f = sum_sites -l1/l0,  f' = sum_sites (l1/l0)^2 - l2/l0   (pll_compute_likelihood_
// derivatives): 40 (48) FMAs per site.
template <int NCH, bool ZERO0, int NW, bool TAILH = false, int NG = 1>
__device__ __forceinline__ void derivatives(const SiteState<NCH>& st, double* tab, int lane,
                                            const LaneConst& lc, Comb<NW, NG>& cb, double t, double& f,
                                            double& df) {
  double e[16], e1[16], e2[16];
  if constexpr (ZERO0) {
    // compact table: entry (order m, category k, eigen index x >= 1) at m * 12 + k * 3 + x - 1
    table_publish(tab, lc.npos, TH_EXP(lc.lr * t) * lc.cN);
    if (TH_WAVES < 3 || NG > 1) {
      double c[36];
#pragma unroll
      for (int i = 0; i < 18; ++i) lds_pair(tab + 2 * i, c[2 * i], c[2 * i + 1]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int x = 1; x < 4; ++x) {
          e[k * 4 + x] = c[k * 3 + x - 1];
          e1[k * 4 + x] = c[12 + k * 3 + x - 1];
          e2[k * 4 + x] = c[24 + k * 3 + x - 1];
        }
    }
  } else {
    table_publish(tab, lane, TH_EXP(lc.lr * t) * lc.cN);
    if (TH_WAVES < 3 || NG > 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { e[i] = tab[i]; e1[i] = tab[16 + i]; e2[i] = tab[32 + i]; }
    }
  }
  double fl = 0.0, dfl = 0.0;
#if TH_WAVES >= 3
  static_assert(!ZERO0 || TH_WAVES < 3, "the three-pass form reads the uncompacted table");
  if constexpr (NG == 1) {
    // three-waves-per-SIMD register budget: the contraction order by order (12 table entries live
    // instead of 36), partial sums of all chunks kept
    double lm[3][NCH];
#pragma unroll
    for (int m3 = 0; m3 < 3; ++m3) {
      double em[16];
      const int tokm = (int)zero_after(m3 == 0 ? t : lm[m3 - 1][NCH - 1]);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (!(ZERO0 && (i & 3) == 0)) em[i] = tab[16 * m3 + i + tokm];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        double l = (ZERO0 && m3 == 0) ? st.S[ch][0] : 0.0;
        if (TAILH && ch == NCH - 1) {
          const double* th = tab + ((lane >> 5) << 3) + 16 * m3 + tokm;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (!(ZERO0 && (i & 3) == 0)) l = fma(st.S[ch][i], th[i], l);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (!(ZERO0 && (i & 3) == 0)) l = fma(st.S[ch][i], em[i], l);
        }
        lm[m3][ch] = l;
      }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (TAILH && ch == NCH - 1) {
        const double l0 = xhalf_add(lm[0][ch]);
        const double l12 = xhalf_add2(lm[1][ch], lm[2][ch]);
        const double qv = -l12 * fast_rcp(l0);
        const bool lower = lane < 32;
        if (st.valid[ch]) { fl += lower ? qv : 0.0; dfl += lower ? qv * qv : qv; }
        continue;
      }
      const double inv = fast_rcp(lm[0][ch]);
      const double d1 = -lm[1][ch] * inv;
      const double d2 = fma(d1, d1, -lm[2][ch] * inv);
      if (st.valid[ch]) { fl += d1; dfl += d2; }
    }
    wave_sum2(fl, dfl, f, df);
    cb.sum2(f, df, lane);
    return;
  }
#endif
  if constexpr (NG > 1) {
    // this group's share of l0 / l1 / l2 of every site, summed over the groups, then the ratios
    constexpr int NV = TAILH ? 3 * (NCH - 1) + 2 : 3 * NCH;
    double lv[NV];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (TAILH && ch == NCH - 1) {
        double l0 = ZERO0 ? st.S[ch][0] : 0.0, l1 = 0.0, l2 = 0.0;
        half_contract<ZERO0>(st.S[ch], tab, lane, l0, l1, l2);
        lv[3 * ch] = xhalf_add(l0);
        lv[3 * ch + 1] = xhalf_add2(l1, l2);
        continue;
      }
      double l0 = ZERO0 ? st.S[ch][0] : 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (!(ZERO0 && (i & 3) == 0)) {
          l0 = fma(st.S[ch][i], e[i], l0);
          l1 = fma(st.S[ch][i], e1[i], l1);
          l2 = fma(st.S[ch][i], e2[i], l2);
        }
      }
      lv[3 * ch] = l0; lv[3 * ch + 1] = l1; lv[3 * ch + 2] = l2;
    }
    cb.template groups<NV>(lv, lane);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (TAILH && ch == NCH - 1) {
        const double qv = -lv[3 * ch + 1] * fast_rcp(lv[3 * ch]);
        const bool lower = lane < 32;
        if (st.valid[ch]) { fl += lower ? qv : 0.0; dfl += lower ? qv * qv : qv; }
        continue;
      }
      const double inv = fast_rcp(lv[3 * ch]);
      const double d1 = -lv[3 * ch + 1] * inv;
      const double d2 = fma(d1, d1, -lv[3 * ch + 2] * inv);
      if (st.valid[ch]) { fl += d1; dfl += d2; }
    }
    wave_sum2(fl, dfl, f, df);   // every wave holds every site's full values: no cross-wave sum
    return;
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (TAILH && ch == NCH - 1) {
      // half-chunk: 8 sumtable entries per lane against this half's table entries; l0 is needed
      // in both halves, l1 lands in the lower and l2 in the upper one
      double l0 = ZERO0 ? st.S[ch][0] : 0.0, l1 = 0.0, l2 = 0.0;
      half_contract<ZERO0>(st.S[ch], tab, lane, l0, l1, l2);
      l0 = xhalf_add(l0);
      const double l12 = xhalf_add2(l1, l2);
      const double qv = -l12 * fast_rcp(l0);   // lower half: -l1 / l0 = d1;  upper half: -l2 / l0
      const bool lower = lane < 32;
      if (st.valid[ch]) { fl += lower ? qv : 0.0; dfl += lower ? qv * qv : qv; }
      continue;
    }
    double l0 = ZERO0 ? st.S[ch][0] : 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (!(ZERO0 && (i & 3) == 0)) {
        l0 = fma(st.S[ch][i], e[i], l0);
        l1 = fma(st.S[ch][i], e1[i], l1);
        l2 = fma(st.S[ch][i], e2[i], l2);
      }
    }
    const double inv = fast_rcp(l0);
    const double d1 = -l1 * inv;
    const double d2 = fma(d1, d1, -l2 * inv);
    if (st.valid[ch]) { fl += d1; dfl += d2; }
  }
  wave_sum2(fl, dfl, f, df);
  cb.sum2(f, df, lane);
}

// sum over the window of log L_site + scalers * log(2^-256)  (pll_compute_edge_loglikelihood);
// ew = w_k exp(lam_x r_k t_pendant).  Per lane the NCH site likelihoods are split into mantissa
// and exponent (v_frexp_*), mantissas multiplied, exponents and scaler counts added as integers:
//   sum_ch log(L_ch) + sc_ch log 2^-256  =  log(prod mant) + ln2 * (sum exp - 256 sum sc)
// -> ONE log() per lane instead of NCH.
template <int NCH, bool ZERO0, int NW, bool TAILH = false, int NG = 1>
__device__ __forceinline__ double window_lnl(const SiteState<NCH>& st, const double (&ew)[16], Comb<NW, NG>& cb,
                                             int lane, const double* tab = nullptr) {
  double mant = 1.0;
  int ex = 0;
  double lsite[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    double l0 = ZERO0 ? st.S[ch][0] : 0.0;
    if (TAILH && ch == NCH - 1) {   // half-chunk: this half's two categories, then both halves
      const double* ewh = tab + 32 + ((lane >> 5) << 3);
      double eh[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) lds_pair(ewh + 2 * i, eh[2 * i], eh[2 * i + 1]);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!(ZERO0 && (i & 3) == 0)) l0 = fma(st.S[ch][i], eh[i], l0);
      l0 = xhalf_add(l0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (!(ZERO0 && (i & 3) == 0)) l0 = fma(st.S[ch][i], ew[i], l0);
    }
    lsite[ch] = l0;
  }
  cb.template groups<NCH>(lsite, lane);   // NG > 1: the site likelihood over all category groups
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    double l0 = lsite[ch];
    bool mine = st.valid[ch];
    if (TAILH && ch == NCH - 1) mine = mine && lane < 32;     // a site is counted once
    if (!mine) l0 = 1.0;
    const int sc = mine ? (int)(st.sc[ch] + st.resc[ch]) : 0;
    mant *= __builtin_amdgcn_frexp_mant(l0);
    ex += __builtin_amdgcn_frexp_exp(l0) - 256 * sc;
    if ((ch & 7) == 7) {  // long windows: keep the running product normalised
      ex += __builtin_amdgcn_frexp_exp(mant);
      mant = __builtin_amdgcn_frexp_mant(mant);
    }
  }
  double v = wave_sum(log(mant) + (double)ex * 0.6931471805599453094), z = 0.0;
  cb.sum2(v, z, lane);
  return v;
}

// pllmod_opt_minimize_newton (pll-modules; rtsafe-style safeguarded Newton).  Wave-uniform.
template <int NCH, bool ZERO0, int NW, bool TAILH = false, int NG = 1>
__device__ __forceinline__ double newton(const SiteState<NCH>& st, double* tab, int lane,
                                         const LaneConst& lc, Comb<NW, NG>& cb, double x1, double xguess,
                                         double x2, double tol, int max_iters, uint32_t& evals) {
  double rts = xguess, f, df, xl, xh, dx;
  if (rts < x1) rts = x1;
  if (rts > x2) rts = x2;
  derivatives<NCH, ZERO0, NW, TAILH, NG>(st, tab, lane, lc, cb, rts, f, df);
  ++evals;
  if (!isfinite(f) || !isfinite(df)) return NAN;
  if (df >= 0.0 && fabs(f) < tol) return rts;
  if (f < 0.0) { xl = rts; xh = x2; } else { xh = rts; xl = x1; }
  for (int i = 1; i <= max_iters; ++i) {
#if TH_SPEC_STEP
    // Both candidate steps are formed before the branch decision (the same IEEE operations on the same operands: the
    // step taken has the bits of the branchy form; the quotient of the step NOT taken may be inf / nan and is dropped).
    // The decision chain of an evaluation -- bracket test, division, comparisons: ~35 dependent fp64 instructions on
    // wave-uniform values -- is pure latency for the wave; the division (10 dependent instructions) now runs beside
    // the bracket test instead of behind it.
    const double dxn = f / df;
    const double dxb = 0.5 * (xh - xl);
    const bool bis = df <= 0.0 || (((rts - xh) * df - f) * ((rts - xl) * df - f) >= 0.0);
    const double nxt = bis ? xl + dxb : rts - dxn;
    const bool same = bis ? (xl == nxt) : (rts == nxt);
    dx = bis ? dxb : dxn;
    rts = nxt;
    if (same) return rts;
#else
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
#endif
    if (fabs(dx) < tol || i == max_iters) return rts;
    if (rts < x1) rts = x1;
    derivatives<NCH, ZERO0, NW, TAILH, NG>(st, tab, lane, lc, cb, rts, f, df);
    ++evals;
    if (!isfinite(f) || !isfinite(df)) return NAN;
    if (df > 0.0 && fabs(f) < tol) return rts;
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return NAN;
}

template <int NCH, bool ZERO0, bool INV, int NW, bool LOCAL, bool TAILH = false, int NG = 1>
__device__ __forceinline__ void process_pair(const ThArgs& a, const uint64_t pidx, const int lane,
                                             double* tab, const double* qts, double* qa, const LaneConst& lc,
                                             Comb<NW, NG>& cb, uint32_t (&wstat)[3]) {
  static_assert(NG == 1 || NW == 1, "category groups and site blocks do not combine");
  const uint32_t site0 = NW > 1 ? (uint32_t)cb.wv * NCH * 64 : 0u;  // first window site of this wave
  const uint32_t grp = NG > 1 ? (uint32_t)cb.wv : 0u;               // this wave's category group
  const uint64_t pid = a.order ? a.order[pidx] : pidx;
  const epa_pair pr = a.pairs[pid];
  // wave-uniform by construction; said explicitly so that every base address below is scalar
  const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)pr.branch_id);
  const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)pr.seq_id);
  const uint32_t begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.win_begin[q]);
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.win_span[q]);
  const size_t cW = a.W;
  // one uniform base (SGPR pair) + 32-bit per-lane byte offsets: component c of the proximal
  // CLV sits at c*W*8, of the distal CLV at (16+c)*W*8 (saddr + voffset addressing, no
  // per-stream 64-bit pointers in VGPRs)
  // component rows of this wave's category group: proximal block [16 NG rows], then the distal block
  const char* ref = reinterpret_cast<const char*>(a.refT + ((size_t)(2 * b) * NG + grp) * 16 * cW + begin);
  const uint32_t W8 = a.W * 8u;
  const uint32_t* scp = a.scSum + (size_t)b * cW + begin;
  const uint8_t* qc = a.codes + (size_t)q * a.cstride + (a.crel ? 0u : begin);
  const double orig = a.blen[b];

  SiteState<NCH> st;
  // TAILH: the last chunk is a half-chunk, lane = (site = lane & 31, half = lane >> 5)
  const uint32_t half = TAILH ? (uint32_t)lane >> 5 : 0u;
  const uint32_t hoff = half * 8u * W8;            // rows of categories 2 h, 2 h + 1
  double* const tabh = tab + (half << 3);          // this half's entries of a 16-entry table block
  auto lane_site = [&](int ch) -> uint32_t {        // window site of this lane in chunk ch
    return site0 + ch * 64 + ((TAILH && ch == NCH - 1) ? (uint32_t)lane & 31u : (uint32_t)lane);
  };
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const uint32_t s = lane_site(ch);
    // single-wave classes: a window of class NCH is longer than (NCH - 1) x 64 sites, so only its LAST
    // chunk can have idle lanes (known at compile time: no masks on the other chunks)
    st.valid[ch] = (NW == 1 && ch < NCH - 1) ? true : s < n;
    const uint32_t sc = st.valid[ch] ? s : 0;  // clamp: inactive lanes recompute site 0
    st.sc[ch] = scp[sc];
    st.code[ch] = qc[sc];
    st.resc[ch] = 0;
  }

  double tp = a.blo.pendant_default, td = orig * 0.5, tx = orig * 0.5;
  uint32_t evals = 0, rounds = 0, reverted = 0;
  uint32_t chain = 0;  // zero_after() token: serialises load batches behind the preceding math

  // +I: the site's invariant term p * pi_inv enters the site likelihood L_0 only.  Eigenvalue 0
  // is exactly 0 (ZERO0), so its table entries are w_0 (order 0) and 0 (orders 1, 2): adding
  // c / w_0 to sumtable entry (category 0, eigen index 0) adds c to L_0 and nothing to L_1, L_2.
  auto cinv_of = [&](int ch) -> double {
    const uint32_t s = st.valid[ch] ? lane_site(ch) : 0;
    const double v = a.cinv[begin + s] * a.inv_w0;
    if (NG > 1 && grp != 0) return 0.0;                  // ... and in the first category group
    return (TAILH && ch == NCH - 1 && half) ? 0.0 : v;   // category 0 lives in the lower half
  };
  // ---- a phase = all (chunk, category) steps of the window in one
  // software pipeline.  The table of the phase must have been published before the call.
  //   MODE 0: inner vector toward the query from (distal e0, proximal e1), folded with the query
  //   MODE 1: toward distal from (query e0, proximal e1), folded with the distal vector
  //   MODE 3: toward proximal from (query e0, distal e1), folded with the proximal vector
  //   MODE 2: the precomputed inner vector of the starting lengths (refI), folded with the query
  // The per-site rescale (pll_update_partials: all c * s entries < 2^-256) is applied to the finished
  // sumtable entries of the chunk instead of to the inner vector (the same 16 multiplications).
  const char* refi_s = reinterpret_cast<const char*>(a.refI + ((size_t)b * NG + grp) * 16 * cW + begin);
  auto stream_phase = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
    constexpr int DEPTH = TH_STREAM_DEPTH;
    constexpr int NKL = TAILH ? 2 : 4;             // categories per lane in the last chunk
    constexpr int NS = 4 * (NCH - 1) + NKL;
    constexpr int R = DEPTH + 1;
    PhaseModel pm;
    pm.load(grp);
    uint32_t W8p = W8;
    asm volatile("" : "+s"(W8p));
    double An[R][4], Bn[R][4];
    uint32_t soff[NCH], sidx[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      sidx[ch] = st.valid[ch] ? lane_site(ch) : 0;
      soff[ch] = sidx[ch] * 8u + ((TAILH && ch == NCH - 1) ? hoff : 0u);
    }
    uint32_t tok = chain;
    if constexpr (MODE == 1 || MODE == 3) {
      // lane = (code = lane >> 2, category = lane & 3): the four entries a_i of (code, category)
      const int kq = lane & 3;
      const double* qv = qts + (lane >> 2) * 4;
      const double* e0 = tab + kq * 4;
      double av[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) av[x] = qv[x] * e0[x];
      double* dst = qa + (lane >> 2) * QA_STRIDE + kq * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double v = pm.U[i * 4] * av[0];
#pragma unroll
        for (int x = 1; x < 4; ++x) v = fma(pm.U[i * 4 + x], av[x], v);
        dst[i] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    auto issue = [&](int stp, uint32_t tk) {
      const int ch = stp >> 2, k = stp & 3, sl = stp % R;
      const uint32_t s0 = soff[ch] + tk;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        // one scalar base + 32-bit lane offsets (saddr + voffset loads).  The row stride is opaque per
        // phase (W8p): with the full chunks' lane offsets now pair-invariant the compiler otherwise
        // hoists all 32 row offsets out of the pair loop and parks them in scratch
        if (MODE == 2) An[sl][x] = *reinterpret_cast<const double*>(refi_s + (s0 + (uint32_t)(k * 4 + x) * W8p));
        else {
          An[sl][x] = *reinterpret_cast<const double*>(ref + (s0 + (uint32_t)(16 * NG + k * 4 + x) * W8p));
          Bn[sl][x] = *reinterpret_cast<const double*>(ref + (s0 + (uint32_t)(k * 4 + x) * W8p));
        }
      }
    };
#pragma unroll
    for (int p = 0; p < DEPTH; ++p)
      if (p < NS) issue(p, tok);
    int mx = 0;
    double qf[4];
#pragma unroll
    for (int stp = 0; stp < NS; ++stp) {
      const int ch = stp >> 2, k = stp & 3, sl = stp % R;
      const bool halfc = TAILH && ch == NCH - 1;
      const int NK = halfc ? 2 : 4;
      const double* tb = halfc ? tabh : tab;
      if (k == 0) {
        const double* qv = qts + st.code[ch] * 4;
        qf[0] = qv[0]; qf[1] = qv[1]; qf[2] = qv[2]; qf[3] = qv[3];
      }
      if (stp + DEPTH < NS) issue(stp + DEPTH, tok);
      asm volatile("" ::: "memory");   // the prefetch is issued here, not at its use
      double It[4];
      if (MODE == 2) {
#pragma unroll
        for (int x = 0; x < 4; ++x) It[x] = An[sl][x];
      } else if (MODE == 0) {
        cat_inner(pm, An[sl], tb + k * 4 + 2 * tok, Bn[sl], tb + 16 + k * 4 + 2 * tok, It, mx);   // (token in units of 16 bytes)
      } else {
        // this half's categories are 2 h, 2 h + 1 in a half-chunk
        const double* ap = qa + st.code[ch] * QA_STRIDE + ((halfc ? (int)half * 2 : 0) + k) * 4 + 2 * tok;
        double Ap[4];
        lds_pair(ap, Ap[0], Ap[1]); lds_pair(ap + 2, Ap[2], Ap[3]);
        cat_inner_pre(pm, Ap, MODE == 1 ? Bn[sl] : An[sl], tb + 16 + k * 4 + 2 * tok, It, mx);
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
        st.S[ch][k * 4 + x] = It[x] * ((MODE == 0 || MODE == 2) ? qf[x] : (MODE == 1 ? An[sl][x] : Bn[sl][x]));
      tok = zero_after(st.S[ch][k * 4 + 3]);
      if (k == NK - 1) {   // chunk complete
        if (MODE == 2) {
          st.resc[ch] = (a.resc0 + (size_t)b * cW + begin)[sidx[ch]];
        } else {
          int mxs = mx;
          if (halfc) {   // both halves of the site
            const auto sw = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mxs = max((int)sw[0], (int)sw[1]);
          }
          if constexpr (NG > 1) {   // the test spans all categories of the site: maximum over the groups
            double mv[1] = {(double)mxs};
            cb.template groups<1, true>(mv, lane);
            mxs = (int)mv[0];
          }
          const uint32_t resc = (mxs < 0x2ff00000) ? 1u : 0u;
          if (__builtin_amdgcn_ballot_w64(resc != 0) != 0) {   // rare: some site of the chunk underflowed
            const double mult = resc ? 0x1p+256 : 1.0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (i < NK * 4) st.S[ch][i] *= mult;
          }
          if (MODE == 0) st.resc[ch] = resc;
          mx = 0;
        }
        if constexpr (INV) st.S[ch][0] += cinv_of(ch);
        if constexpr (ZERO0) {   // fold0 with the phase's copy of the weights
          if (halfc) st.S[ch][0] = fma(half ? pm.w[3] : pm.w[1], st.S[ch][4], (half ? pm.w[2] : pm.w[0]) * st.S[ch][0]);
          else st.S[ch][0] = fma(pm.w[3], st.S[ch][12], fma(pm.w[2], st.S[ch][8], fma(pm.w[1], st.S[ch][4], pm.w[0] * st.S[ch][0])));
        }
        tok = zero_after(st.S[ch][halfc ? 7 : 15]);
      }
    }
    chain = tok;
  };
  auto product_phase = [&](auto mode_c) { stream_phase(mode_c); };
  // Inner CLV toward the query at (td, tx) folded with the query, S = (U^-1 I) o qt, and the
  // window lnL at pendant length tp_.  One table pass: slot 0 -> exp(lr td), slot 1 ->
  // exp(lr tx), slot 2 -> w exp(lr tp).
  auto window_score = [&]() -> double {
    double ew[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) lds_pair(tab + 32 + 2 * i, ew[2 * i], ew[2 * i + 1]);
    return window_lnl<NCH, ZERO0, NW, TAILH, NG>(st, ew, cb, lane, tab);
  };
  auto score = [&](double td_, double tx_, double tp_) -> double {
    const double tl = lc.slot == 0 ? td_ : (lc.slot == 1 ? tx_ : tp_);
    table_publish(tab, lane, TH_EXP(lc.lr * tl) * (lc.slot == 2 ? lc.w : 1.0));
    product_phase(std::integral_constant<int, 0>{});
    return window_score();
  };
  // inner CLV toward distal: I' = (P_pend q) o (P_prox X); S = Dt o (U^-1 I').  TOWARD_PROX (the
  // --raxml-blo loop only): the same toward the proximal node, I' = (P_pend q) o (P_dist D), S = Xt o ...
  auto side_sumtable = [&](double tp_, double tother_, auto toward_prox) {
    constexpr bool TOWARD_PROX = decltype(toward_prox)::value;
    table_publish(tab, lane, TH_EXP(lc.lr * (lc.slot == 0 ? tp_ : tother_)));
    product_phase(std::integral_constant<int, TOWARD_PROX ? 3 : 1>{});
  };
  auto distal_sumtable = [&](double tp_, double tx_) { side_sumtable(tp_, tx_, std::false_type{}); };

  // Initial score at the starting lengths (orig/2, orig/2, default pendant): the inner CLV does
  // not depend on the query, k_build_lookup stored its U^-1 image per (branch, site) -> 16 loads
  // and the fold with the query instead of 32 loads and the two 4x4 products per category.
  auto score_first = [&](double tp_) -> double {
    table_publish(tab, lane, TH_EXP(lc.lr * tp_) * (lc.slot == 2 ? lc.w : 1.0));
    stream_phase(std::integral_constant<int, 2>{});
    return window_score();
  };

  // traverse_update_partials + initial score (optimize.cpp:15-42,111-113)
  double loglikelihood = a.refI ? -score_first(tp) : -score(td, tx, tp);

  uint32_t smoothings = a.blo.max_rounds;
  if constexpr (LOCAL) {
    // --raxml-blo: pllmod_opt_optimize_branch_lengths_local(radius 1, keep_update 1) on the triplet
    // (optimize.cpp:274-279; pll-modules source absent: restated in oracle/epa_oracle.c opt_local).
    // Per smoothing round: NR on the pendant edge, on the distal edge (inner CLV re-aimed at it), on
    // the proximal edge (with the new distal length), the inner CLV re-aimed at the query, NR on the
    // pendant edge once more, then the edge lnL from the sumtable in registers.  The three lengths
    // are independent (no sliding): the result rescales distal by orig / (distal + proximal).
    const double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
    auto solve = [&](double cur) -> double {
      double g = cur;
      if (g < xmin || g > xmax) g = a.blo.default_branch;
      const double r = newton<NCH, ZERO0, NW, TAILH, NG>(st, tab, lane, lc, cb, xmin, g, xmax, xtol, (int)a.blo.max_newton, evals);
      chain = zero_after(r);
      // keep_update: the length is replaced when the solver moved it
      return (isfinite(r) && fabs(cur - r) > 1e-10) ? r : cur;
    };
    while (smoothings) {
      tp = solve(tp);
      side_sumtable(tp, tx, std::false_type{});
      td = solve(td);
      side_sumtable(tp, td, std::true_type{});
      tx = solve(tx);
      (void)score(td, tx, tp);   // inner CLV back toward the query + the pendant sumtable
      tp = solve(tp);
      table_publish(tab, lane, TH_EXP(lc.lr * tp) * (lc.slot == 2 ? lc.w : 1.0));
      double ew[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) lds_pair(tab + 32 + 2 * i, ew[2 * i], ew[2 * i + 1]);
      const double new_ll = -window_lnl<NCH, ZERO0, NW, TAILH, NG>(st, ew, cb, lane, tab);
      ++rounds;
      --smoothings;
      if (fabs(new_ll - loglikelihood) < a.blo.epsilon) smoothings = 0;
      loglikelihood = new_ll;
    }
  }
  while (!LOCAL && smoothings) {
    const double old_td = td, old_tp = tp;
    // ---- NR for the pendant length (optimize.cpp:135-166); S already holds the pendant sumtable
    double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
    double xguess = tp;
    if (xguess < xmin || xguess > xmax) xguess = a.blo.default_branch;
    double xres = newton<NCH, ZERO0, NW, TAILH, NG>(st, tab, lane, lc, cb, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, evals);
    if (xres > 0.0) tp = xres;
    chain = zero_after(tp);
    // ---- NR for the distal length with the proximal P-matrix held fixed (:170-211)
    distal_sumtable(tp, tx);
    xguess = td;
    xmin = fmin(a.blo.min_branch / 2.0, orig / 2.0);
    xtol = xmin / 10.0;
    xmax = orig - xtol;
    if (xguess < xmin || xguess > xmax) xguess = orig / 2.0;
    xres = newton<NCH, ZERO0, NW, TAILH, NG>(st, tab, lane, lc, cb, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, evals);
    if (xres > 0.0) { td = xres; tx = orig - xres; }
    chain = zero_after(td);
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

  if (lane == 0 && ((NW == 1 && NG == 1) || cb.wv == 0)) {
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
  // counters stay in registers: one atomic per WAVE at kernel end, not three per pair (a single
  // L2 word retires ~1 atomic / 12 ns: 3 x 50k pairs on shared words would cost ~2 ms)
  wstat[0] += rounds;
  wstat[1] += evals;
  wstat[2] += reverted;
}

// Persistent single-wave workgroups.  Workgroup g is observed to run on XCD g % 8 (used for
// speed only): XCD x owns the x-th eighth of the branch-sorted pair list, so one branch's CLV
// windows are served by one 4 MiB L2.  Inside its XCD slice a wave takes pairs round-robin
// (wave, wave + stride, ...): neighbouring waves work on neighbouring pairs = the same branch,
// and every wave gets a ~25-pair random sample of the 10x cost spread (1..32 NR rounds).
// INV: the model has +I (instantiated for ZERO0 only; a separate instantiation so that the
// default kernel's register allocation is untouched: the runtime-flag version cost 40 more spills)
template <int NCH, bool ZERO0, bool INV, int NW, bool LOCAL = false, bool TAILH = false, int NG = 1>
__global__ void __launch_bounds__(64 * NW * NG, TH_WAVES) k_thorough_dna(const ThArgs a) {
  constexpr int NWV = NW * NG;     // waves of the workgroup: site blocks (NW) or category groups (NG)
  __shared__ __attribute__((aligned(16))) double tab[64 * NWV];  // broadcast table of each wave
  __shared__ __attribute__((aligned(16))) double qts[64];   // U^-1 image of the 16 query column codes
  __shared__ double red[2 * NW * 2];
  __shared__ double e2t[64];
  __shared__ __attribute__((aligned(16))) double qa[(16 * QA_STRIDE) * NWV];  // per wave: (U (e o q_code))_i for 16 codes x 16 (category, state)
  __shared__ double xch[NG > 1 ? 2 * NG * Comb<NW, NG>::XCH_MAX * 64 : 1];
  const int lane = threadIdx.x & 63;
  Comb<NW, NG> cb{red, (int)(threadIdx.x >> 6), 0, xch, 0};
  if (threadIdx.x < 64) {
    qts[lane] = a.qt[lane];
    e2t[lane] = exp2((double)lane * 0.015625);
  }
  LaneConst lc;
  lc.e2t = e2t;
  {
    const int lk = (lane >> 2) & 3, lx = lane & 3;
    const int kg = (NG > 1 ? cb.wv * 4 : 0) + lk;   // category of this lane's table entry
    lc.slot = lane >> 4;
    lc.lr = a.m.lam[lx] * a.m.rate[kg];
    lc.w = a.m.w[kg];
    lc.cN = lc.slot == 0 ? lc.w : (lc.slot == 1 ? lc.w * lc.lr : (lc.slot == 2 ? lc.w * lc.lr * lc.lr : 0.0));
    // compact Newton table position; the 28 lanes without an entry (eigen index 0, slot 3) get the dump slots 36..63
    const int dump = lc.slot == 3 ? 36 + (lane & 15) : 52 + lc.slot * 4 + lk;
    lc.npos = (lc.slot < 3 && lx != 0) ? lc.slot * 12 + lk * 3 + lx - 1 : dump;
  }
  __syncthreads();
  const uint32_t x = blockIdx.x & 7;
  const uint32_t w = blockIdx.x >> 3, stride = gridDim.x >> 3;
  uint64_t n_pairs = a.n_pairs;
  if constexpr (NW == 1 && NG == 1) {
    if (a.spec) {   // queued launch: count and validity come from the selection's read-back block
      const uint32_t n = a.spec[0];
      if (n > a.spec_max || a.spec[9 + a.spec_cls] != n || a.spec[32] || a.spec[33]) return;
      n_pairs = n;
    }
  }
  const uint64_t per = (n_pairs + 7) / 8;
  uint64_t lo = (uint64_t)x * per;
  uint64_t hi = lo + per < n_pairs ? lo + per : n_pairs;
  if constexpr (NW == 1 && NG == 1) {
    lo = (n_pairs * a.xcum[x]) >> 20;
    hi = (n_pairs * a.xcum[x + 1]) >> 20;
    if (a.xstamp && blockIdx.x == 0 && threadIdx.x == 0) a.stats[7] = __builtin_amdgcn_s_memrealtime() & EPA_XSTAMP_MASK;
  }
  // clock-true roofline: workgroup 0's first wave lives as long as the launch (resident waves); its s_memtime
  // (shader cycles) over s_memrealtime (100 MHz) is the shader clock the launch really ran at
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  uint32_t wstat[3] = {0, 0, 0};
  if constexpr (NW == 1 && NG == 1) {
    // resident waves fetch the next pair of their XCD slice from a counter: no relaunches, no
    // tail of unlucky waves.  The next index is requested before the current pair is processed.
    // (The single-wave classes are always launched with the counters; one loop, not two copies of
    // the pair body: the second copy cost 16 spilled registers in the dominant instantiation.)
    // (Measured and dropped, round 4: waves that find their slice exhausted fetching from the next XCD's slice in
    // short launches -- 13k pairs, six per resident wave: 0.563 -> 0.655 ms per 5000-read chunk; the stolen pairs'
    // reference rows are in another XCD's L2.)
    // (Measured and dropped, round 4: XCD x working through eight interleaved blocks of the list instead of one
    // contiguous eighth -- 5.19 - 5.22 ms per launch either way.  In-kernel stamps (-DTH_TIMING) show the slices
    // draining up to 250 us apart in a 5.5 ms launch, but the SAME XCDs are early / late for both block layouts and
    // other ones on another box: the XCDs differ in speed by a few per cent, the slices do not differ in cost.)
    uint32_t* ctr = a.qctr + x;
    const uint32_t cnt = (uint32_t)(hi > lo ? hi - lo : 0);
    uint32_t nxt = 0;
#ifdef TH_TIMING
    unsigned long long* tlog = reinterpret_cast<unsigned long long*>(a.sscratch) + (size_t)blockIdx.x * 8;
    unsigned long long t_first = 0, t_second = 0;
    uint32_t npr = 0;
    if (lane == 0) tlog[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (lane == 0) nxt = atomicAdd(ctr, 1u);
    nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
    while (nxt < cnt) {
      const uint32_t cur = nxt;
      uint32_t f = 0;
      if (lane == 0) f = atomicAdd(ctr, 1u);
      process_pair<NCH, ZERO0, INV, NW, LOCAL, TAILH, NG>(a, lo + cur, lane, tab + cb.wv * 64, qts, qa + cb.wv * (16 * QA_STRIDE), lc, cb, wstat);
      nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
#ifdef TH_TIMING
      ++npr;
      if (npr == 1) t_first = __builtin_amdgcn_s_memrealtime();
      if (npr == 2) t_second = __builtin_amdgcn_s_memrealtime();
#endif
    }
#ifdef TH_TIMING
    if (lane == 0) { tlog[1] = t_first; tlog[2] = t_second; tlog[3] = __builtin_amdgcn_s_memrealtime(); tlog[4] = npr; }
#endif
  } else {
    for (uint64_t p = lo + w; p < hi; p += stride)
      process_pair<NCH, ZERO0, INV, NW, LOCAL, TAILH, NG>(a, p, lane, tab + cb.wv * 64, qts, qa + cb.wv * (16 * QA_STRIDE), lc, cb, wstat);
  }
  if (threadIdx.x == 0) {
    atomicAdd(&a.stats[0], (unsigned long long)wstat[0]);
    atomicAdd(&a.stats[1], (unsigned long long)wstat[1]);
    atomicAdd(&a.stats[2], (unsigned long long)wstat[2]);
    if constexpr (NW == 1 && NG == 1) {
      // last exit of XCD x's waves, with the share of the pair list the XCD had in THIS launch in the low bits
      if (a.xstamp) atomicMax(&a.stats[8 + x], ((unsigned long long)(__builtin_amdgcn_s_memrealtime() & EPA_XSTAMP_MASK) << 21) |
                                                   (unsigned long long)(a.xcum[x + 1] - a.xcum[x]));
    }
    if (a.xstamp && blockIdx.x == 0) {
      a.stats[5] = __builtin_amdgcn_s_memtime() - clk0;
      a.stats[6] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
  }
}

#ifdef TH_ONLY_MAIN   // experiment builds (exp/): the dominant instantiation alone
template __global__ void k_thorough_dna<3, true, false, 1, false, true, 1>(const ThArgs a);
}  // namespace
#else
// ---------------------------------------------------------------------------------------------
// Long windows (more than 24 x 64 sites): same algorithm, the sumtable lives in an HBM slab
// ([17][Wpad] doubles per resident wave: 16 components + the per-site scaler count) instead of
// registers, the site chunks are a runtime loop.  Throughput is secondary here: the path exists
// so that full-length queries of long alignments are placed at all.
// ---------------------------------------------------------------------------------------------
template <class Deriv>
__device__ __forceinline__ double newton_fn(Deriv&& deriv, double x1, double xguess, double x2, double tol,
                                            int max_iters, uint32_t& evals) {
  double rts = xguess, f, df, xl, xh, dx;
  if (rts < x1) rts = x1;
  if (rts > x2) rts = x2;
  deriv(rts, f, df);
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
    deriv(rts, f, df);
    ++evals;
    if (!isfinite(f) || !isfinite(df)) return NAN;
    if (df > 0.0 && fabs(f) < tol) return rts;
    if (f < 0.0) xl = rts; else xh = rts;
  }
  return NAN;
}

template <bool ZERO0>
__global__ void __launch_bounds__(64, 2) k_thorough_dna_long(const ThArgs a) {
  __shared__ double tab[64];
  __shared__ double qts[64];
  __shared__ double e2t[64];
  const int lane = threadIdx.x;
  qts[lane] = a.qt[lane];
  e2t[lane] = exp2((double)lane * 0.015625);
  LaneConst lc;
  lc.e2t = e2t;
  {
    const int lk = (lane >> 2) & 3, lx = lane & 3;
    lc.slot = lane >> 4;
    lc.lr = a.m.lam[lx] * a.m.rate[lk];
    lc.w = a.m.w[lk];
    lc.cN = lc.slot == 0 ? lc.w : (lc.slot == 1 ? lc.w * lc.lr : (lc.slot == 2 ? lc.w * lc.lr * lc.lr : 0.0));
  }
  __syncthreads();
  const ModelDNA& m = a.m;
  double* slab = a.sscratch + (size_t)blockIdx.x * 17 * a.Wpad;  // [17][Wpad]
  const size_t cW = a.W;
  uint32_t wrounds = 0, wevals = 0, wreverts = 0;
  for (uint64_t pidx = blockIdx.x; pidx < a.n_pairs; pidx += gridDim.x) {
    const uint64_t pid = a.order ? a.order[pidx] : pidx;
    const epa_pair pr = a.pairs[pid];
    const uint32_t b = pr.branch_id, q = pr.seq_id;
    const uint32_t begin = a.win_begin[q], n = a.win_span[q];
    const uint32_t nch = (n + 63) / 64;
    const double* Xt = a.refT + (size_t)(2 * b) * 16 * cW + begin;
    const double* Dt = a.refT + (size_t)(2 * b + 1) * 16 * cW + begin;
    const uint32_t* scp = a.scSum + (size_t)b * cW + begin;
    const uint8_t* qc = a.codes + (size_t)q * a.cstride + (a.crel ? 0u : begin);
    const double orig = a.blen[b];

    // inner CLV toward the query at (td, tx), S = (U^-1 I) o qt -> slab; returns the window lnL
    auto score = [&](double td_, double tx_, double tp_, bool first) -> double {
      const double tl = lc.slot == 0 ? td_ : (lc.slot == 1 ? tx_ : tp_);
      table_publish(tab, lane, TH_EXP(lc.lr * tl) * (lc.slot == 2 ? lc.w : 1.0));
      double ew[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) ew[i] = tab[32 + i];
      double mant = 1.0;
      int ex = 0;
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        const bool valid = site < n;
        const uint32_t s = valid ? site : 0;
        double It[16];
        uint32_t resc;
        if (first) {
          const double* It0 = a.refI + (size_t)b * 16 * cW + begin;
#pragma unroll
          for (int c = 0; c < 16; ++c) It[c] = It0[(size_t)c * cW + s];
          resc = a.resc0[(size_t)b * cW + begin + s];
        } else {
          double D[16], X[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) { D[c] = Dt[(size_t)c * cW + s]; X[c] = Xt[(size_t)c * cW + s]; }
          inner_site(m, D, tab, X, tab + 16, It, resc);
        }
        const double* qv = qts + qc[s] * 4;
        double l0 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            double sv = It[k * 4 + x] * qv[x];
            if (a.cinv && k == 0 && x == 0) sv += a.cinv[begin + s] * a.inv_w0;  // +I, see process_pair
            if (valid) slab[(size_t)(k * 4 + x) * a.Wpad + site] = sv;
            l0 = fma(sv, ew[k * 4 + x], l0);
          }
        const uint32_t sc = scp[s] + resc;
        if (valid) slab[(size_t)16 * a.Wpad + site] = (double)sc;
        if (!valid) l0 = 1.0;
        mant *= __builtin_amdgcn_frexp_mant(l0);
        ex += __builtin_amdgcn_frexp_exp(l0) - (valid ? 256 * (int)sc : 0);
        ex += __builtin_amdgcn_frexp_exp(mant);
        mant = __builtin_amdgcn_frexp_mant(mant);
      }
      __threadfence_block();
      return wave_sum(log(mant) + (double)ex * 0.6931471805599453094);
    };
    // inner CLV toward distal: I' = (P_pend q) o (P_prox X); S = Dt o (U^-1 I') -> slab
    auto distal_sumtable = [&](double tp_, double tx_) {
      table_publish(tab, lane, TH_EXP(lc.lr * (lc.slot == 0 ? tp_ : tx_)));
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        const bool valid = site < n;
        const uint32_t s = valid ? site : 0;
        double Qv[16], X[16], D[16], It[16];
        const double* qv = qts + qc[s] * 4;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const double v = qv[x];
          Qv[x] = v; Qv[4 + x] = v; Qv[8 + x] = v; Qv[12 + x] = v;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) { X[c] = Xt[(size_t)c * cW + s]; D[c] = Dt[(size_t)c * cW + s]; }
        uint32_t r;
        inner_site(m, Qv, tab, X, tab + 16, It, r);
        if (valid) {
#pragma unroll
          for (int c = 0; c < 16; ++c)
            slab[(size_t)c * a.Wpad + site] = D[c] * It[c] + ((a.cinv && c == 0) ? a.cinv[begin + s] * a.inv_w0 : 0.0);
        }
      }
      __threadfence_block();
    };
    auto deriv = [&](double t, double& f, double& df) {
      table_publish(tab, lane, TH_EXP(lc.lr * t) * lc.cN);
      double e[16], e1[16], e2[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { e[i] = tab[i]; e1[i] = tab[16 + i]; e2[i] = tab[32 + i]; }
      double fl = 0.0, dfl = 0.0;
      for (uint32_t ch = 0; ch < nch; ++ch) {
        const uint32_t site = ch * 64 + lane;
        const bool valid = site < n;
        const uint32_t s = valid ? site : 0;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const double sv = slab[(size_t)i * a.Wpad + s];
          l0 = fma(sv, e[i], l0);
          if (!(ZERO0 && (i & 3) == 0)) { l1 = fma(sv, e1[i], l1); l2 = fma(sv, e2[i], l2); }
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
    double loglikelihood = -score(td, tx, tp, a.refI != nullptr);
    uint32_t smoothings = a.blo.max_rounds;
    while (smoothings) {
      const double old_td = td, old_tp = tp;
      double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
      double xguess = tp;
      if (xguess < xmin || xguess > xmax) xguess = a.blo.default_branch;
      double xres = newton_fn(deriv, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, evals);
      if (xres > 0.0) tp = xres;
      distal_sumtable(tp, tx);
      xguess = td;
      xmin = fmin(a.blo.min_branch / 2.0, orig / 2.0);
      xtol = xmin / 10.0;
      xmax = orig - xtol;
      if (xguess < xmin || xguess > xmax) xguess = orig / 2.0;
      xres = newton_fn(deriv, xmin, xguess, xmax, xtol, (int)a.blo.max_newton, evals);
      if (xres > 0.0) { td = xres; tx = orig - xres; }
      const double new_ll = -score(td, tx, tp, false);
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
    if (lane == 0) {
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
  if (lane == 0) {
    atomicAdd(&a.stats[0], (unsigned long long)wrounds);
    atomicAdd(&a.stats[1], (unsigned long long)wevals);
    atomicAdd(&a.stats[2], (unsigned long long)wreverts);
  }
}

}  // namespace

namespace {

__global__ void __launch_bounds__(256) k_pair_class(const epa_pair* __restrict__ pairs, uint64_t n,
                                                   const uint32_t* __restrict__ win_span, int states,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                   uint32_t* __restrict__ hist) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const int cls = i < n ? epa_span_class(states, win_span[pairs[i].seq_id]) : -1;
  if (i < n) {
    if (keys) { keys[i] = (uint32_t)cls; idx[i] = (uint32_t)i; }
  }
  if (hist) {
    for (int c = 0; c < EPA_N_CLS; ++c) {
      const unsigned long long bal = __ballot(cls == c);
      if (bal && (threadIdx.x & 63) == 0) atomicAdd(&hist[c], (uint32_t)__popcll(bal));
    }
  }
}

}  // namespace

// One launch per span class present (see epa_span_class).  `order` = this class's pair indices
// in their original (branch-major) order, null when the launch covers all pairs.
static int launch_thorough_dna_class(epa_ctx* ctx, ThArgs a, int cls, uint32_t max_span) {
  const uint64_t n_pairs = a.n_pairs;
  // Grid: 2048 wavefronts are resident (8 per CU, 2 per SIMD at this kernel's VGPR budget), i.e.
  // 2048 / NW workgroups.  Oversubscribing the resident set lets the hardware dispatcher do the
  // load balancing (pairs differ 10x in cost): a finished workgroup's slot is refilled at once.
  // About four pairs per wave balance best (measured: 131k pairs 8 vs 16, 262k pairs 8 / 16 / 32
  // / 64 -> 6.64 / 6.66 / 6.50 / 6.58 ms; one pair per wave: 10.2 ms).
  uint32_t per_slot = (uint32_t)std::min<uint64_t>(32, std::max<uint64_t>(8, n_pairs / (2048 * 4)));
  constexpr uint64_t th_grid_waves = 0;   // (resident waves of the single-wave classes: 2048 = every slot; fewer were measured and lose)
  // class -> (wavefronts per pair NW, 64-site chunks per wavefront NCH): windows up to 192 sites
  // are one wave's job; longer ones are spread over 2 / 4 / 8 waves of a workgroup, each keeping
  // its part of the sumtable in registers (NCH stays <= 3: the kernel's register budget)
  // single-wave classes: resident waves + a work counter per XCD slice (round 1: the oversubscribed
  // static grid instead cost 262k pairs 6.56 vs 6.47 ms)
#ifdef TH_TIMING
#define TH_TIMING_ALLOC a.sscratch = (double*)epa_scratch(ctx, 9, 2048 * 64); (void)hipMemsetAsync(a.sscratch, 0, 2048 * 64, ctx->stream);
#define TH_TIMING_DUMP th_timing_dump(ctx, a.sscratch);
#else
#define TH_TIMING_ALLOC
#define TH_TIMING_DUMP
#endif
#define LAUNCH(N, NW_)                                                                            \
  do {                                                                                            \
    uint64_t want = (uint64_t)256 * 8 * per_slot / (NW_);                                          \
    a.qctr = nullptr;                                                                              \
    if ((NW_) == 1) {                                                                              \
      TH_TIMING_ALLOC                                                                              \
      { const int zr_ = epa_th_ctr_reset(ctx); if (zr_) return zr_; }                                  \
      a.qctr = epa_th_ctr(ctx);                                                                        \
      want = th_grid_waves ? th_grid_waves : 1024 * TH_WAVES;                                                   \
    }                                                                                              \
    if (want > n_pairs) want = n_pairs;                                                            \
    const uint32_t nwg = (uint32_t)((want + 7) / 8 * 8);                                           \
    constexpr bool TH_ = (NW_) == 1 && ((N) == 2 || (N) == 3);   /* half-chunk tail instantiations exist for these */  \
    if (tailh && TH_ && ctx->dna_zero0 && !ctx->blo.sliding && !a.cinv)                                                \
      hipLaunchKernelGGL((k_thorough_dna<N, true, false, 1, true, TH_>), dim3(nwg), dim3(64), 0, ctx->stream, a);      \
    else if (tailh && TH_ && ctx->dna_zero0 && !ctx->blo.sliding)   /* --raxml-blo with +I */                          \
      hipLaunchKernelGGL((k_thorough_dna<N, true, true, 1, true, TH_>), dim3(nwg), dim3(64), 0, ctx->stream, a);       \
    else if (tailh && TH_ && ctx->dna_zero0 && ctx->blo.sliding && a.cinv)                                            \
      hipLaunchKernelGGL((k_thorough_dna<N, true, true, 1, false, TH_>), dim3(nwg), dim3(64), 0, ctx->stream, a);      \
    else if (tailh && TH_ && ctx->dna_zero0 && ctx->blo.sliding)                                                       \
      hipLaunchKernelGGL((k_thorough_dna<N, true, false, 1, false, TH_>), dim3(nwg), dim3(64), 0, ctx->stream, a);     \
    else if (!ctx->blo.sliding && a.cinv) hipLaunchKernelGGL((k_thorough_dna<N, true, true, NW_, true>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a); \
    else if (!ctx->blo.sliding) hipLaunchKernelGGL((k_thorough_dna<N, true, false, NW_, true>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a); \
    else if (a.cinv) hipLaunchKernelGGL((k_thorough_dna<N, true, true, NW_>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a); \
    else if (ctx->dna_zero0) hipLaunchKernelGGL((k_thorough_dna<N, true, false, NW_>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a); \
    else hipLaunchKernelGGL((k_thorough_dna<N, false, false, NW_>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a); \
  } while (0)
  // half-chunk tail (TAILH), classes 10 / 11: every window of the class ends within 32 sites of its
  // last chunk's start (150-site reads: 64 + 64 + 22).  Same-box A/B on the cfg2 bench: 6.57 - 6.69
  // -> 6.42 ms per launch.  EPA_TH_TAIL=0 runs these classes on the full-chunk kernels.
  // 8 / 12 / 16 rate categories (ctx->dna.ng = 2 / 3 / 4 groups of four): workgroup = NG waves = the
  // category groups of one pair (see Comb); single-wave window classes only, longer windows, --raxml-blo
  // and rate matrices without an exact zero eigenvalue take the general kernel
  if (ctx->dna.ng > 1) {
    const bool ok = ctx->blo.sliding && ctx->dna_zero0 && (cls <= 2 || cls == 10 || cls == 11);
    if (!ok) {
      a.Wpad = (std::max(max_span, 1u) + 63) / 64 * 64;
      return launch_thorough_generic(ctx, a.pairs, n_pairs, a.codes, a.win_begin, a.win_span, max_span, a.out,
                                     a.stats, a.order ? a.order : nullptr, true);
    }
    const int nch = cls == 0 ? 1 : (cls == 1 || cls == 10) ? 2 : 3;
    uint64_t want = (uint64_t)256 * 8 * per_slot / (uint64_t)ctx->dna.ng;
    if (want > n_pairs) want = n_pairs;
    const uint32_t nwg = (uint32_t)((want + 7) / 8 * 8);
    a.qctr = nullptr;
#define LAUNCH_G(N, G)                                                                                          \
    do {                                                                                                        \
      if (a.cinv) hipLaunchKernelGGL((k_thorough_dna<N, true, true, 1, false, false, G>), dim3(nwg), dim3(64 * (G)), 0, ctx->stream, a); \
      else hipLaunchKernelGGL((k_thorough_dna<N, true, false, 1, false, false, G>), dim3(nwg), dim3(64 * (G)), 0, ctx->stream, a);       \
    } while (0)
#define LAUNCH_GN(G) do { if (nch == 1) LAUNCH_G(1, G); else if (nch == 2) LAUNCH_G(2, G); else LAUNCH_G(3, G); } while (0)
    if (ctx->dna.ng == 2) LAUNCH_GN(2); else if (ctx->dna.ng == 3) LAUNCH_GN(3); else LAUNCH_GN(4);
#undef LAUNCH_GN
#undef LAUNCH_G
    return EPA_OK;
  }
  const bool tailh = cls == 10 || cls == 11;
  constexpr bool n4_off = false;
  switch (cls) {
    case 0: LAUNCH(1, 1); break;
    case 1: case 10: LAUNCH(2, 1); break;
    case 2: case 11: LAUNCH(3, 1); break;
    // <= 256 sites: ONE wave with four chunks beats two waves with two (one barrier per Newton
    // evaluation and 2-wave workgroups cost more than the 82 spilled registers of NCH = 4): len 224
    // 7.60 -> 6.11 ms, len 256 6.88 -> 5.68 ms per launch.  NCH = 5 / 6 lose (9.6 / 11.4 vs 7.8 / 7.4 ms).
    case 3: if (n4_off) LAUNCH(2, 2); else LAUNCH(4, 1); break;
    case 4: LAUNCH(3, 2); break;   // <= 384
    case 5: LAUNCH(2, 4); break;   // <= 512
    case 6: LAUNCH(3, 4); break;   // <= 768
    case 7: LAUNCH(2, 8); break;   // <= 1024
    case 8: LAUNCH(3, 8); break;   // <= 1536
    default: {
      // long windows: sumtable slab in HBM, one resident wave per slab
      const uint32_t nlong = (uint32_t)std::min<uint64_t>(n_pairs, 2048);
      a.Wpad = (max_span + 63) / 64 * 64;
      a.sscratch = (double*)epa_scratch(ctx, 7, sizeof(double) * (size_t)nlong * 17 * a.Wpad);
      if (!a.sscratch) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(long-window sumtable scratch)");
      if (ctx->dna_zero0) hipLaunchKernelGGL((k_thorough_dna_long<true>), dim3(nlong), dim3(64), 0, ctx->stream, a);
      else hipLaunchKernelGGL((k_thorough_dna_long<false>), dim3(nlong), dim3(64), 0, ctx->stream, a);
    }
  }
#undef LAUNCH
#ifdef TH_TIMING
  if (a.sscratch && (cls <= 2 || cls == 10 || cls == 11)) {
    std::vector<unsigned long long> h(2048 * 8);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(h.data(), a.sscratch, 2048 * 64, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < 2048; ++w) if (h[w * 8]) { t0 = std::min(t0, h[w * 8]); t1 = std::max(t1, h[w * 8 + 3]); }
    std::vector<double> st, fp, sp, ex, np_;
    for (int w = 0; w < 2048; ++w) {
      if (!h[w * 8]) continue;
      st.push_back((h[w * 8] - t0) * 0.01);
      if (h[w * 8 + 1]) fp.push_back((h[w * 8 + 1] - h[w * 8]) * 0.01);
      if (h[w * 8 + 2]) sp.push_back((h[w * 8 + 2] - h[w * 8 + 1]) * 0.01);
      ex.push_back((h[w * 8 + 3] - t0) * 0.01);
      np_.push_back((double)h[w * 8 + 4]);
    }
    for (int x = 0; x < 8; ++x) {
      double e0 = 1e30, e1 = 0, fsum = 0; unsigned long long np2 = 0; int nw = 0;
      for (int w = x; w < 2048; w += 8) {
        if (!h[w * 8]) continue;
        const double e = (h[w * 8 + 3] - t0) * 0.01;
        e0 = std::min(e0, e); e1 = std::max(e1, e); np2 += h[w * 8 + 4]; ++nw;
        if (h[w * 8 + 1]) fsum += (h[w * 8 + 1] - h[w * 8]) * 0.01;
      }
      fprintf(stderr, "  x %d: waves %d pairs %llu exit %.1f .. %.1f us, mean first pair %.1f\n", x, nw, np2, e0, e1, nw ? fsum / nw : 0.0);
    }
    auto pct = [](std::vector<double>& v, double p) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    fprintf(stderr, "TH_TIMING n_pairs %llu: span %.1f us | wave start p50 %.1f p99 %.1f max %.1f | first pair p50 %.1f p90 %.1f max %.1f | second pair p50 %.1f p90 %.1f | exit p1 %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | pairs/wave p50 %.0f max %.0f\n",
            (unsigned long long)a.n_pairs, (t1 - t0) * 0.01, pct(st, 0.5), pct(st, 0.99), pct(st, 1.0), pct(fp, 0.5), pct(fp, 0.9), pct(fp, 1.0), pct(sp, 0.5), pct(sp, 0.9),
            pct(ex, 0.01), pct(ex, 0.1), pct(ex, 0.5), pct(ex, 0.9), pct(ex, 1.0), pct(np_, 0.5), pct(np_, 1.0));
  }
#endif
  return EPA_OK;
}

static ThArgs dna_args(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* ord, const uint8_t* d_codes,
                       const uint32_t* d_begin, const uint32_t* d_span, epa_result* d_out, unsigned long long* d_stats) {
  ThArgs a;
  a.m = ctx->dna;
  a.blo = ctx->blo;
  a.refT = ctx->refT;
  // filled by k_build_lookup; until then every pair computes its own starting vector
  a.refI = ctx->lookup_built ? ctx->refI : nullptr;
  a.resc0 = ctx->resc0;
  a.cinv = ctx->cinv;
  a.inv_w0 = ctx->inv_w0;
  a.scSum = ctx->scSum;
  a.blen = ctx->blen;
  a.qt = ctx->dmodel->qt;
  a.pairs = d_pairs;
  a.order = ord;
  a.codes = d_codes;
  a.crel = ctx->code_stride ? 1u : 0u;
  a.cstride = a.crel ? ctx->code_stride : ctx->W;
  a.win_begin = d_begin;
  a.win_span = d_span;
  a.out = d_out;
  a.stats = d_stats;
  a.sscratch = nullptr;
  a.qctr = nullptr;
  a.n_pairs = 0;
  a.W = ctx->W;
  a.Wpad = 0;
  a.spec = nullptr;
  a.spec_cls = 0;
  a.spec_max = 0;
  for (int i = 0; i < 9; ++i) a.xcum[i] = ctx->xcd_cum[i];
  a.xstamp = 0;
  return a;
}

// The thorough launch of a fused chunk body queued BEHIND the selection, before the host has seen the candidate
// count (what place_thorough() starts from, src/core/place.cpp:97-171, decided on the device): possible when the
// chunk's windows can only be of ONE span class that a single-wave instantiation serves -- the class of max_span,
// i.e. reads of one length, the BASELINE workloads -- because then the launch needs no class partition and the
// resident grid is the same for every pair count.  The kernel takes the count from d_spec and exits at once when
// the block says that this was not the right launch (pairs of another class, candidate overflow, a window error);
// the host applies the same test to its copy of the block and queues the ordinary launches in that case.
// Returns the class (>= 0) when queued, -1 when the configuration is not eligible.
int launch_thorough_queued(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* d_spec, uint64_t max_pairs,
                           const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                           epa_result* d_out, unsigned long long* d_stats) {
  if (ctx->s != 4 || ctx->generic_thorough || ctx->dna.ng != 1 || !epa_th_ctr(ctx) || !d_span || !d_spec) return -1;
  if (max_pairs > 0xffffffffull) return -1;
  const int cls = epa_span_class(4, max_span);
  if (!(cls <= 2 || cls == 10 || cls == 11)) return -1;   // single-wave classes (resident waves + work counters)
  static const uint32_t bound[12] = {64, 128, 192, 0, 0, 0, 0, 0, 0, 0, 96, 160};
  ThArgs a = dna_args(ctx, d_pairs, nullptr, d_codes, d_begin, d_span, d_out, d_stats);
  a.n_pairs = 0xffffffffull;   // sizes the (full resident) grid only
  a.spec = d_spec;
  a.spec_cls = (uint32_t)cls;
  a.spec_max = (uint32_t)max_pairs;
  a.xstamp = 1u;
  epa_timer_start(ctx, epa_t(ctx, epa_ctx::T_THOROUGH));
  const int rc = launch_thorough_dna_class(ctx, a, cls, std::min(max_span, bound[cls]));
  epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_THOROUGH));
  if (rc) return -2;
  if (hipGetLastError() != hipSuccess) { (void)epa_fail(ctx, EPA_ERR_HIP, "queued thorough launch"); return -2; }
  return cls;
}

int launch_thorough(epa_ctx* ctx, const epa_pair* d_pairs, uint64_t n_pairs, const uint8_t* d_codes,
                    const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                    epa_result* d_out, unsigned long long* d_stats) {
  if (n_pairs > 0xffffffffull) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "thorough: more than 2^32 pairs per call");
  // any category count, Newton variants, --raxml-blo outside the tuned instantiation (20 states, +I,
  // windows beyond the multi-wave classes)
  if (ctx->generic_thorough || (!ctx->blo.sliding && ctx->s == 4 && max_span > 1536u)) {
    ctx->cls_hist_pairs = 0;
    return launch_thorough_generic(ctx, d_pairs, n_pairs, d_codes, d_begin, d_span, max_span, d_out, d_stats);
  }
  // ---- span classes present in this call
  uint32_t hist[EPA_N_CLS] = {};
  const int cmax = epa_span_class(ctx->s, max_span);
  const bool cached = ctx->cls_hist_pairs == n_pairs && n_pairs != 0;
  if (cached) {
    for (int c = 0; c < EPA_N_CLS; ++c) hist[c] = ctx->cls_hist[c];
  } else if (cmax == 0) {
    hist[0] = (uint32_t)n_pairs;  // every window is in the smallest class
  }
  ctx->cls_hist_pairs = 0;
  // scratch 8: [hist 64 B | keys n | idx n | keys_out n | order n | rocprim temp]
  size_t temp_bytes = 0;
  (void)rocprim::radix_sort_pairs<epa_radix_cfg>(nullptr, temp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n_pairs, 0, 4, ctx->stream);
  const size_t nb = (sizeof(uint32_t) * n_pairs + 255) & ~(size_t)255;
  uint32_t* d_hist = nullptr;
  uint32_t *d_keys = nullptr, *d_idx = nullptr, *d_keys2 = nullptr, *d_order = nullptr;
  auto carve = [&]() -> int {
    char* base = (char*)epa_scratch(ctx, 8, 256 + 4 * nb + temp_bytes);
    if (!base) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(span-class scratch)");
    d_hist = (uint32_t*)base;
    d_keys = (uint32_t*)(base + 256); d_idx = (uint32_t*)(base + 256 + nb);
    d_keys2 = (uint32_t*)(base + 256 + 2 * nb); d_order = (uint32_t*)(base + 256 + 3 * nb);
    return EPA_OK;
  };
  const dim3 cgrid((uint32_t)((n_pairs + 255) / 256));
  bool have_keys = false;
  if (!cached && cmax != 0) {  // pairs from the caller: histogram (and keys) in one pass, one round trip
    int rc = carve();
    if (rc) return rc;
    EPA_HIP(ctx, hipMemsetAsync(d_hist, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_pair_class, cgrid, dim3(256), 0, ctx->stream, d_pairs, n_pairs, d_span, ctx->s, d_keys,
                       d_idx, d_hist);
    EPA_HIP(ctx, hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, ctx->stream));
    EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    have_keys = true;
  }
  int present = 0;
  for (int c = 0; c < EPA_N_CLS; ++c) present += hist[c] != 0;
  const uint32_t* order = nullptr;
  if (present > 1) {  // stable partition by class: branch-major order survives inside a class
    if (!have_keys) {
      int rc = carve();
      if (rc) return rc;
      hipLaunchKernelGGL(k_pair_class, cgrid, dim3(256), 0, ctx->stream, d_pairs, n_pairs, d_span, ctx->s,
                         d_keys, d_idx, (uint32_t*)nullptr);
    }
    void* temp = (char*)d_order + nb;
    EPA_HIP(ctx, rocprim::radix_sort_pairs<epa_radix_cfg>(temp, temp_bytes, d_keys, d_keys2, d_idx, d_order, (size_t)n_pairs, 0, 4,
                                           ctx->stream));
    order = d_order;
  }
  epa_timer_start(ctx, epa_t(ctx, epa_ctx::T_THOROUGH));
  int rc = EPA_OK;
  uint64_t off = 0;
  ctx->xstamp_ok = present == 1;
  // window bound of a class (slab sizing of the 20-state LDS kernel, the long-window kernel)
  static const uint32_t dna_bound[EPA_N_CLS] = {64, 128, 192, 256, 384, 512, 768, 1024, 1536, 0xffffffffu, 96, 160};
  for (int c = 0; c < EPA_N_CLS && rc == EPA_OK; ++c) {
    if (!hist[c]) continue;
    const uint32_t* ord = order ? order + off : nullptr;
    off += hist[c];
    if (ctx->s == 20) {
      // matrix-core kernel (sumtable in registers): windows up to 384 sites with 4 rate categories, up to 256
      // with 8 (+G8, +R5 ..); longer ones, or all of them with EPA_AA_VALU=1 (A/B switch, 4 categories), the
      // lane = site VALU kernel / the general kernel
      static const uint32_t aa_bound[4] = {64, 128, 192, 0xffffffffu};
      const bool aa_valu = ctx->opt.aa_valu != 0;
      const uint32_t bound = std::min(max_span, aa_bound[c < 4 ? c : 3]);
      const uint32_t mfma_max = ctx->c == 4 ? 384u : 256u;
      // (the VALU kernel has no --raxml-blo instantiation: the A/B switch applies to the sliding rule only)
      if (bound <= mfma_max && (!aa_valu || !ctx->blo.sliding || ctx->c != 4))
        rc = launch_thorough_aa_mfma(ctx, d_pairs, ord, hist[c], d_codes, d_begin, d_span, bound, d_out, d_stats);
      else if (ctx->c == 4 && ctx->blo.sliding)
        rc = launch_thorough_aa(ctx, d_pairs, ord, hist[c], d_codes, d_begin, d_span, bound,
                                bound <= EPA_AA_LDS_MAX_SPAN, d_out, d_stats);
      else
        rc = launch_thorough_generic(ctx, d_pairs, hist[c], d_codes, d_begin, d_span, bound, d_out, d_stats, ord, true);
      continue;
    }
    ThArgs a = dna_args(ctx, d_pairs, ord, d_codes, d_begin, d_span, d_out, d_stats);
    a.n_pairs = hist[c];
    a.xstamp = (present == 1 && ctx->dna.ng == 1 && (c <= 2 || c == 10 || c == 11)) ? 1u : 0u;
    rc = launch_thorough_dna_class(ctx, a, c, std::min(max_span, dna_bound[c]));
  }
  epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_THOROUGH));
  if (rc) return rc;
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}
#endif  // TH_ONLY_MAIN
