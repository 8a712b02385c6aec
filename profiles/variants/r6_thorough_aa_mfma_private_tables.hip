// Hot loop 2 on device, 20-state models with 4 rate categories, windows up to 192 sites: the
// states x states contractions on the matrix cores.
//
// Same algorithm and control flow as k_thorough_aa / k_thorough_dna (Tiny_Tree::place ->
// optimize_branch_triplet -> opt_branch_lengths_pplacer, src/core/pll/optimize.cpp:60-286, in the
// eigenbasis of Q); what changes is the geometry.  Per (site, category) a phase needs three
// 20 x 20 matrix-vector products (U (e o A), U (e o B), U^-1 I: 83 % of the pair's flops) and a
// Newton evaluation is a 3 x 80 contraction of the site's sumtable with the tables
// w lr^i exp(lr t).  Both map onto v_mfma_f64_4x4x4_4b_f64 WITHOUT padding (20 = 5 x 4): measured
// on this chip (profiles/aa_contraction_ab.hip) the 20 x 20 . 20 x 64 contraction runs at 68.8
// TFLOP/s in that form, 45.8 useful TFLOP/s with v_mfma_f64_16x16x4_f64 (20 rows padded to 32) and
// 26.7 TFLOP/s in the lane = site VALU form with the matrix behind the scalar cache.
//
// Layout (found by an impulse-response probe of the instruction, exp/mfma_layout.hip): a lane of
// v_mfma_f64_4x4x4_4b_f64 is (k = lane / 16, block = lane / 4 % 4, r = lane % 4); A holds
// A_block[i = r][k], B holds B_block[k][j = r], D returns D_block[i = lane / 16][j = r].  With the
// four blocks = four groups of four sites and the same matrix tile in every block:
//     vector registers:  reg t, lane (q = lane / 16, s = lane % 16) = v[4 t + q] of site s
//     D of a product  :  reg rt, lane (q, s)                        = y[4 rt + q] of site s
// i.e. a product's output IS the next product's input, element-wise operations are lane-wise, and
// nothing is ever transposed.  A tile of 16 sites x 4 categories is 20 vector registers.
//
//   workgroup = one (branch, query) pair, wave w owns the 16-site tiles w, w + 4, w + 8 (NT <= 3),
//   all four categories of a tile live in one wave: the per-site sums over the categories and the
//   "all 80 entries < 2^-256" rescale test need no workgroup barrier;
//   U / U^-1 sit in LDS as 25 A-operand tiles each (a read per MFMA, shared by the a and b products);
//   the sumtable of the wave's tiles stays in REGISTERS (NT x 20 doubles), the Newton evaluation is
//   20 MFMAs per tile whose D rows are l0, l1, l2 of the 16 sites.
//   Only f, f', lnL cross the waves (LDS + one barrier), and the exp tables are published per phase.
#include "epa_dev_internal.hpp"
#include "wave_util.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

using namespace epa_wave;

constexpr int S = 20;
constexpr int NTS = 5;   // tiles of 4 along the state axis
constexpr double LOG_2 = 0.6931471805599453094;

struct ThArgsAM {
  const ModelDev* m;
  BloConsts blo;
  const double* refT;      // [2B][80][W]
  const double* refI;      // [B][80][W] U^-1 inner CLV at the starting lengths (k_build_lookup), or null
  const uint8_t* resc0;    // [B][W]     its per-site rescale flag
  const double* cinv;      // +I: [W] p * pi_inv per site, or null
  double inv_w0;
  const uint32_t* scSum;   // [B][W]
  const double* blen;
  const epa_pair* pairs;
  const uint32_t* order;
  const uint8_t* codes;
  uint32_t cstride, crel;
  const uint32_t* win_begin;
  const uint32_t* win_span;
  epa_result* out;
  unsigned long long* stats;
  uint64_t n_pairs;
  uint32_t W;
  uint32_t* qctr;          // one work counter per XCD slice of the pair list (resident workgroups), or null
  uint32_t xcum[9];        // XCD shares of the pair list and the drain stamps behind them: see thorough_dna.hip ThArgs
  uint32_t xstamp;
};

// LDS is read with ds_read_b128 wherever two neighbouring doubles go to the same lane: 256 B/clk from
// one wave per SIMD, where the 8-byte reads need four waves per SIMD for half of that -- and this
// kernel runs at two.
// doubles per table: NC * 4 * 6 entries + 8 of padding = 64 B mod 256 B for NC = 4 and 8: the three Newton rows
// hit different banks
template <int NC> constexpr int tab_stride() { return NC * 24 + 8; }
// AAM_PRIVATE_TABLES (round 6): a Newton evaluation used to cost two workgroup barriers -- one behind the table every
// thread had written one entry of, one behind the per-wave partial sums.  Now every wave writes the whole table for
// itself (80 / 160 exponentials over 64 lanes = 2 / 3 per lane instead of one, + the three orders' coefficients: ~25
// vector instructions more per wave) into its own LDS block, ordered by the wave's own LDS queue -- no workgroup
// barrier -- and the partial sums alternate between two slots, so the ONE remaining barrier per evaluation also covers a
// wave that is already writing the next evaluation's sums.  Same entries, same products, same summation order: the bits
// of the two-barrier form (0: that form, for the A/B).
#ifndef AAM_PRIVATE_TABLES
#define AAM_PRIVATE_TABLES 1
#endif
template <int NC>
struct SharedM {
  // A-operand tiles of U in the order the products use them, q = 5 t + rt, two tiles interleaved:
  // Ua[((q / 2) * 16 + k * 4 + i) * 2 + q % 2] = U[4 rt + i][4 t + k]
  double Ua[13 * 32];
  double Uia[13 * 32];             // the same of U^-1
  // wave-uniform exp tables, entry (category c, eigen index 4 t + kq) at [(c * 4 + kq) * 6 + t]:
  // the five values a lane needs are consecutive (the sixth is padding)
  double tab[4][tab_stride<NC>()];  // [3]: zeros -- the B operand's fourth column in a Newton evaluation
  double bc[24];                   // cross-wave sums: f [0..8), f' [8..16), lnL [16..24)
  double bc2[2][16];               // AAM_PRIVATE_TABLES: f / f' partial sums, alternating slot per evaluation
  double ncst[NC * S][4];          // ... per (category, eigen index): lr, w, w lr, (w lr) lr -- the table coefficients
  uint32_t next_pair;              // work-queue hand-out of the workgroup
  double e2t[64];                  // 2^(j/64): table of exp_tab (wave_util.hpp)
};
// the per-wave partial sums of a workgroup in a fixed order (4 waves: (b0 + b1) + (b2 + b3), as ever)
template <int NW>
__device__ __forceinline__ double sum_waves(const double* b) {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per pair");
  const double lo = (b[0] + b[1]) + (b[2] + b[3]);
  if constexpr (NW == 4) return lo;
  else return lo + ((b[4] + b[5]) + (b[6] + b[7]));
}


// 0 that the optimiser cannot see through, ordered after `v`: added to an LDS index it keeps the
// (loop-invariant) matrix-tile and table reads where they are written -- hoisted out of the
// category / tile loops they would pin 150 VGPRs and the kernel would live in scratch
__device__ __forceinline__ int zero_after(double v) {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(__double2loint(v)));
  return z;
}
__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
// value of lane (quad base + Q) in every lane of the quad (DPP quad_perm, no LDS)
template <int Q>
__device__ __forceinline__ double quad_bcast(double v) {
  constexpr int CTRL = Q | (Q << 2) | (Q << 4) | (Q << 6);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// five consecutive doubles at a 16-byte aligned LDS address: two 16-byte reads and one of 8
__device__ __forceinline__ void lds5(const double* p, double (&v)[5]) {
  const double2 x = *reinterpret_cast<const double2*>(p);
  const double2 y = *reinterpret_cast<const double2*>(p + 2);
  v[0] = x.x; v[1] = x.y; v[2] = y.x; v[3] = y.y; v[4] = p[4];
}
// tiles q = 2 p, 2 p + 1 of an interleaved A-operand array (tile 24 has no partner)
__device__ __forceinline__ void lds_tiles(const double* U2, int p, int ao, double& u0, double& u1) {
  if (p < 12) {
    const double2 v = *reinterpret_cast<const double2*>(U2 + (p * 16 + ao) * 2);
    u0 = v.x;
    u1 = v.y;
  } else {
    u0 = U2[(12 * 16 + ao) * 2];
    u1 = 0.0;
  }
}
// (wave-uniform pointer) + 32-bit byte offset: global_load with a scalar base
__device__ __forceinline__ double ldg_off(const double* base, uint32_t byte_off) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}
// The reference rows of a pair through a buffer descriptor: address = base + (scalar row offset) + (lane offset).
// The row offset of every (category, state block) is formed on the SCALAR unit and the lane offset is one VGPR
// per tile -- no vector instruction per load.  (global_load kept the first row of a category in the scalar-base
// form and chained a 64-bit v_lshl_add_u64 per further row: 8 vector instructions per category step, each paid in
// full between the fp64 MFMAs, profiles/r4_mfma_fill_microbench.txt.)
#ifndef AAM_BUFFER_LOADS
#define AAM_BUFFER_LOADS 1
#endif
typedef unsigned int aam_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double ldb_off(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t row_off) {
  const aam_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_off, (int)row_off, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
// combine the four component rows (lanes l, l ^ 16, l ^ 32, l ^ 48) of a site: v_permlane16_swap /
// v_permlane32_swap of a register with itself put row r ^ 1 (half h ^ 1) beside row r -- four
// cross-lane instructions and two adds, no LDS round trip (__shfl_xor is ds_bpermute)
__device__ __forceinline__ void rows_pair16(double v, double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void rows_pair32(double v, double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double rows_sum(double v) {
  double a, b;
  rows_pair16(v, a, b);
  v = a + b;
  rows_pair32(v, a, b);
  return a + b;
}
__device__ __forceinline__ double rows_max(double v) {
  double a, b;
  rows_pair16(v, a, b);
  v = fmax(a, b);
  rows_pair32(v, a, b);
  return fmax(a, b);
}

// NC: rate categories (4, or 8 = two groups of four: +G8, +R5 .. +R8 padded with weight-0 copies; the
// sumtable of a tile is NC x 5 registers, so NT x NC <= 16 to stay near the register budget);
// NW: waves per pair (4, or 8 for windows of up to 8 x NT x 16 sites: one workgroup of 512 threads per CU)
template <int NT, bool LOCAL = false, int NC = 4, int NW = 4>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 2) k_thorough_aa_mfma(const ThArgsAM a) {
  constexpr int NTHR = 64 * NW;
  constexpr int CS = NC * S;          // component rows of a reference vector
  __shared__ alignas(16) SharedM<NC> sh;
#if AAM_PRIVATE_TABLES
  __shared__ alignas(16) double ntab_all[NW][3][tab_stride<NC>()];   // every wave's own Newton table
#endif
  const ModelDev* __restrict__ m = a.m;
  // wave-uniform values are told to the compiler as such (v_readfirstlane): the pair's ids, window and row
  // pointers then live in SGPRs, the reference rows are loaded as (scalar base)[lane offset] without a 64-bit
  // vector add per load, and the per-tile "is this tile inside the window" tests are scalar branches
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4;            // component residue of this lane (state 4 t + kq of vector register t)
  const int sl = lane & 15;            // site inside a 16-site tile
  const int aoff = kq * 4 + (lane & 3);  // this lane's element of an A-operand tile
  for (int idx = tid; idx < 400; idx += NTHR) {
    const int qq = idx >> 4, e = idx & 15, t = qq / NTS, rt = qq % NTS, kk = e >> 2, i = e & 3;
    const int dst = ((qq >> 1) * 16 + e) * 2 + (qq & 1);
    sh.Ua[dst] = m->U[(4 * rt + i) * S + 4 * t + kk];
    sh.Uia[dst] = m->Ui[(4 * rt + i) * S + 4 * t + kk];
  }
  if (tid < 64) sh.e2t[tid] = exp2((double)tid * 0.015625);
  for (int i = tid; i < tab_stride<NC>(); i += NTHR) sh.tab[3][i] = 0.0;
  // per-thread table constants: entry e = tid + i NTHR < 3 CS is (slot = e / CS, kx = e % CS)
  constexpr int NENT = 3 * CS, NE = (NENT + NTHR - 1) / NTHR;
  int tslot[NE], tpos[NE];
  double t_lr[NE], t_w[NE], t_c[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = tid + i * NTHR;
    const int kx = e % CS;
    tslot[i] = e / CS;
    tpos[i] = ((kx / S) * 4 + (kx % S) % 4) * 6 + (kx % S) / 4;
    t_lr[i] = 0.0; t_w[i] = 0.0; t_c[i] = 0.0;
    if (e < NENT) {
      t_lr[i] = m->lam[kx % S] * m->rate[kx / S];
      t_w[i] = m->w[kx / S];
      t_c[i] = tslot[i] == 0 ? t_w[i] : (tslot[i] == 1 ? t_w[i] * t_lr[i] : t_w[i] * t_lr[i] * t_lr[i]);
    }
  }
#if AAM_PRIVATE_TABLES
  for (int kx = tid; kx < CS; kx += NTHR) {
    const double lr = m->lam[kx % S] * m->rate[kx / S], w = m->w[kx / S];
    sh.ncst[kx][0] = lr; sh.ncst[kx][1] = w; sh.ncst[kx][2] = w * lr; sh.ncst[kx][3] = w * lr * lr;
  }
  for (int i = tid; i < NW * 3 * tab_stride<NC>(); i += NTHR) (&ntab_all[0][0][0])[i] = 0.0;   // incl. the padding entries
  constexpr int NE2 = (CS + 63) / 64;
  int npos[NE2];
#pragma unroll
  for (int i = 0; i < NE2; ++i) {
    const int kx = (lane + 64 * i) % CS;
    npos[i] = ((kx / S) * 4 + (kx % S) % 4) * 6 + (kx % S) / 4;
  }
  int bcpar = 0;
#endif
  __syncthreads();

  uint32_t wrounds = 0, wevals = 0, wreverts = 0;
#ifdef AAM_PROFILE
  long long cyc_phase = 0, cyc_newton = 0, cyc_pub = 0, cyc_total = 0, n_phase = 0, n_pairs_done = 0;
  long long cyc_d[5] = {0, 0, 0, 0, 0};   // per evaluation: exp + table write | barrier | table reads + MFMAs + ratios | wave sum + exchange write | barrier + cross-wave sum
  const long long cyc_begin = clock64();
#endif
  // Workgroup g runs on XCD g % 8 (observed; used for speed only): XCD x owns the x-th eighth of the
  // branch-sorted pair list, so one branch's windows are served by one L2.  Resident workgroups
  // take the pairs of their slice from a counter (requested one pair ahead); without a counter the
  // slice is dealt round-robin to an oversubscribed grid.
  const uint32_t xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const uint64_t slice_lo = (a.n_pairs * a.xcum[xcd]) >> 20;
  const uint64_t slice_hi = (a.n_pairs * a.xcum[xcd + 1]) >> 20;
  if (a.xstamp && blockIdx.x == 0 && tid == 0) a.stats[7] = __builtin_amdgcn_s_memrealtime() & EPA_XSTAMP_MASK;
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  const uint32_t slice_n = (uint32_t)(slice_hi > slice_lo ? slice_hi - slice_lo : 0);
  uint32_t* const ctr = a.qctr ? a.qctr + xcd : nullptr;
  uint32_t cur = wg_in_xcd;
  if (ctr) {
    if (tid == 0) sh.next_pair = atomicAdd(ctr, 1u);
    __syncthreads();
    cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.next_pair);
  }
  auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  while (cur < slice_n) {
    uint32_t ahead = 0;
    if (ctr && tid == 0) ahead = atomicAdd(ctr, 1u);
    const uint64_t pidx = slice_lo + cur;
    const uint64_t pid = a.order ? (uint64_t)uni(a.order[pidx]) : pidx;
    const epa_pair pr = a.pairs[pid];
    const uint32_t b = uni(pr.branch_id), q = uni(pr.seq_id);
    const uint32_t begin = uni(a.win_begin[q]), n = uni(a.win_span[q]);
    const size_t cW = a.W;
    const double* Xt = a.refT + (size_t)(2 * b) * CS * cW + begin;       // proximal
    const double* Dt = a.refT + (size_t)(2 * b + 1) * CS * cW + begin;   // distal
#if AAM_BUFFER_LOADS
    // one descriptor for both sides of the branch (distal = + CS W doubles), one for the precomputed inner rows
    const __amdgpu_buffer_rsrc_t rsT = row_rsrc(Xt), rsI = row_rsrc(a.refI ? a.refI + (size_t)b * CS * cW + begin : Xt);
    const uint32_t side_off = (uint32_t)(CS * cW * 8);
#endif
    const uint32_t* scp = a.scSum + (size_t)b * cW + begin;
    const uint8_t* qc = a.codes + (size_t)q * a.cstride + (a.crel ? 0u : begin);
    double orig = a.blen[b];
    orig = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(orig)), __builtin_amdgcn_readfirstlane(__double2loint(orig)));

    // this wave's tiles: tile g = wv + NW j covers sites 16 g .. 16 g + 15 of the window
    // loads are (wave-uniform row pointer)[32-bit lane offset]: scalar base + one offset VGPR per tile,
    // instead of a 64-bit address pair per row kept live across the whole optimisation
    uint32_t sscl[NT], lo[NT];      // site of this lane in tile j, clamped for the loads; kq W + site
    bool valid[NT], tile_on[NT];
    double qv[NT][NTS];             // query tip vector in the eigenbasis, vector layout
    double Sm[NT][NC][NTS];         // sumtable of the branch being optimised: [tile][category][reg]
    bool resc_keep[NT];             // LOCAL: rescale flag of the last inner vector toward the query
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint32_t site = 16u * (uint32_t)(wv + NW * j) + (uint32_t)sl;
      valid[j] = site < n;
      tile_on[j] = 16u * (uint32_t)(wv + NW * j) < n;
      sscl[j] = valid[j] ? site : 0;
      lo[j] = ((uint32_t)kq * a.W + sscl[j]) * 8u;   // bytes
      const uint32_t code = qc[sscl[j]];
#pragma unroll
      for (int t = 0; t < NTS; ++t) qv[j][t] = m->qt[code * S + 4 * t + kq];
    }
    // ---- table publication: every thread < 240 computes one exp()
    // (no barrier in front: whatever ran before ended with a workgroup barrier behind its last
    // table read -- the phases and the Newton evaluations below keep that invariant)
    auto publish = [&](double t0, double t1, double t2) {
#pragma unroll
      for (int i = 0; i < NE; ++i)
        if (tid + i * NTHR < NENT) {
          const double t = tslot[i] == 0 ? t0 : (tslot[i] == 1 ? t1 : t2);
          const double e = exp_tab(t_lr[i] * t, sh.e2t);
          sh.tab[tslot[i]][tpos[i]] = tslot[i] == 2 ? e * t_w[i] : e;
        }
      __syncthreads();
    };

    // mode 0: inner CLV toward the query from (distal, proximal), sumtable folded with the query,
    //         window lnL; mode 1: toward distal from (query, proximal), sumtable folded with the
    //         distal vector; mode 2: as 0 from the per-branch precomputed inner CLV (refI); mode 3
    //         (--raxml-blo only): the mirror image of 1, toward proximal from (query, distal).
    auto phase = [&](int mode, double& lnl_out) {
      const bool side = mode == 1 || mode == 3;   // sumtable of a side branch: no window lnL
      double mant = 1.0;
      int ex = 0;
      const double* I0 = a.refI + (size_t)b * CS * cW + begin;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!tile_on[j]) continue;   // wave-uniform
        const uint32_t s = sscl[j];
        double l0 = 0.0, mx = 0.0;
        bool resc = false;
        // the operands of category `cat + 1` are requested while category `cat` is on the matrix cores
        double Bn[NTS], Dn[NTS];
        auto fetch = [&](int cat, int zf) {
          const uint32_t lof = lo[j] + (uint32_t)zf;   // the token keeps the (read-only, phase-invariant) loads in this phase
#if AAM_BUFFER_LOADS
          // (the row pitch re-derived behind the token: twenty loop-invariant row offsets per side would otherwise be
          // hoisted out of the pair loop into SGPRs the kernel does not have -- spilled, each use a v_readlane)
          const uint32_t cw8 = uni((uint32_t)cW * 8u + (uint32_t)zf);
          const uint32_t r0 = (uint32_t)(cat * S) * cw8, rstep = 4u * cw8;   // scalar
          if (mode == 2) {
#pragma unroll
            for (int t = 0; t < NTS; ++t) Bn[t] = ldb_off(rsI, lof, r0 + (uint32_t)t * rstep);
            return;
          }
          const uint32_t ob = mode == 3 ? side_off : 0u, of = mode == 3 ? 0u : side_off;
#pragma unroll
          for (int t = 0; t < NTS; ++t) {
            Bn[t] = ldb_off(rsT, lof, ob + r0 + (uint32_t)t * rstep);
            Dn[t] = ldb_off(rsT, lof, of + r0 + (uint32_t)t * rstep);
          }
#else
          const size_t c0 = (size_t)(cat * S) * cW;   // uniform; component 4 t + kq: + 4 t cW + lo
          if (mode == 2) {
#pragma unroll
            for (int t = 0; t < NTS; ++t) Bn[t] = ldg_off(I0 + c0 + (size_t)(4 * t) * cW, lof);
            return;
          }
          const double* Pb = mode == 3 ? Dt : Xt;   // the side that goes through the matrix products
          const double* Pf = mode == 3 ? Xt : Dt;   // the side the result is folded with (modes 1, 3) / the first factor (mode 0)
#pragma unroll
          for (int t = 0; t < NTS; ++t) {
            Bn[t] = ldg_off(Pb + c0 + (size_t)(4 * t) * cW, lof);
            Dn[t] = ldg_off(Pf + c0 + (size_t)(4 * t) * cW, lof);
          }
#endif
        };
        fetch(0, zero_after(mant));
#pragma unroll
        for (int cat = 0; cat < NC; ++cat) {
          __builtin_amdgcn_sched_barrier(0);
          // ordering token of this category's matrix-tile / table reads: behind the arrival of its
          // operands (requested a category ago).  Without it the loop-invariant tile reads are
          // hoisted out of the category loop (150 VGPRs of them) and the kernel lives in scratch.
          const int zt = zero_after(Bn[NTS - 1]);
          const int ao = aoff + zt;
          const int tp5 = (cat * 4 + kq) * 6;   // this lane's five table entries of the category
          double It[NTS], Dv[NTS];
          if (mode == 2) {
#pragma unroll
            for (int t = 0; t < NTS; ++t) It[t] = Bn[t];
            if (cat < NC - 1) fetch(cat + 1, zt);
          } else {
            double Av[NTS], Bv[NTS];
            {
              double e0[NTS], e1[NTS];
              lds5(&sh.tab[0][tp5 + zt], e0);
              lds5(&sh.tab[1][tp5 + zt], e1);
#pragma unroll
              for (int t = 0; t < NTS; ++t) {
                Dv[t] = Dn[t];
                Av[t] = (mode == 0 ? Dn[t] : qv[j][t]) * e0[t];
                Bv[t] = Bn[t] * e1[t];
              }
            }
            if (cat < NC - 1) fetch(cat + 1, zt);
            // a = U (e0 o A), b = U (e1 o B): one read of a U tile pair feeds four MFMAs
            double ya[NTS], yb[NTS];
#pragma unroll
            for (int rt = 0; rt < NTS; ++rt) { ya[rt] = 0.0; yb[rt] = 0.0; }
            // the tile pair of step p + 1 is requested before the MFMAs of step p are issued: with one
            // register pair the read could only be issued once its predecessor had been consumed, and
            // every group of MFMAs waited for a full LDS round trip
            {
              double u0, u1;
              lds_tiles(sh.Ua, 0, ao, u0, u1);
#pragma unroll
              for (int p = 0; p < 13; ++p) {
                double n0 = 0.0, n1 = 0.0;
                if (p < 12) lds_tiles(sh.Ua, p + 1, ao, n0, n1);
                __builtin_amdgcn_sched_barrier(0);
                ya[(2 * p) % NTS] = mfma4(u0, Av[(2 * p) / NTS], ya[(2 * p) % NTS]);
                yb[(2 * p) % NTS] = mfma4(u0, Bv[(2 * p) / NTS], yb[(2 * p) % NTS]);
                if (p < 12) {
                  ya[(2 * p + 1) % NTS] = mfma4(u1, Av[(2 * p + 1) / NTS], ya[(2 * p + 1) % NTS]);
                  yb[(2 * p + 1) % NTS] = mfma4(u1, Bv[(2 * p + 1) / NTS], yb[(2 * p + 1) % NTS]);
                }
                __builtin_amdgcn_sched_barrier(0);
                u0 = n0; u1 = n1;
              }
            }
            double Iv[NTS];
            {
              // the first U^-1 tile pair is requested before the element-wise product it will meet
              double u0, u1;
              lds_tiles(sh.Uia, 0, ao, u0, u1);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int rt = 0; rt < NTS; ++rt) {
                Iv[rt] = ya[rt] * yb[rt];
                mx = fmax(mx, Iv[rt]);
                It[rt] = 0.0;
              }
#pragma unroll
              for (int p = 0; p < 13; ++p) {
                double n0 = 0.0, n1 = 0.0;
                if (p < 12) lds_tiles(sh.Uia, p + 1, ao, n0, n1);
                __builtin_amdgcn_sched_barrier(0);
                It[(2 * p) % NTS] = mfma4(u0, Iv[(2 * p) / NTS], It[(2 * p) % NTS]);
                if (p < 12) It[(2 * p + 1) % NTS] = mfma4(u1, Iv[(2 * p + 1) / NTS], It[(2 * p + 1) % NTS]);
                __builtin_amdgcn_sched_barrier(0);
                u0 = n0; u1 = n1;
              }
            }
          }
          double e2[NTS];
          if (!side) lds5(&sh.tab[2][tp5 + zt], e2);
#pragma unroll
          for (int t = 0; t < NTS; ++t) {
            const double sv = It[t] * (side ? Dv[t] : qv[j][t]);
            Sm[j][cat][t] = sv;
            if (!side) l0 = fma(sv, e2[t], l0);
          }
        }
        // pll_update_partials per-site scaling: ALL 80 entries of the site below 2^-256
        if (mode == 2) {
          resc = a.resc0[(size_t)b * cW + begin + s] != 0;   // already applied to refI
        } else {
          resc = rows_max(mx) < 0x1p-256;
          if (__any(resc)) {
            const double mult = resc ? 0x1p+256 : 1.0;
#pragma unroll
            for (int cat = 0; cat < NC; ++cat)
#pragma unroll
              for (int t = 0; t < NTS; ++t) Sm[j][cat][t] *= mult;
            l0 *= mult;
          }
        }
        // +I: p * pi_inv enters L_0 only (eigenvalue 0 is exactly 0): folded into sumtable entry
        // (category 0, eigen index 0) = register 0 of the lanes with kq == 0
        if (a.cinv && kq == 0) {
          const double add = a.cinv[begin + s] * a.inv_w0;
          Sm[j][0][0] += add;
          if (!side) l0 = fma(add, sh.tab[2][0], l0);
        }
        if (LOCAL && !side) resc_keep[j] = resc;
        if (!side) {
          double ls = rows_sum(l0);
          const bool mine = valid[j] && kq == 0;
          if (!mine) ls = 1.0;
          const int sc = mine ? (int)(scp[s] + (resc ? 1u : 0u)) : 0;
          mant *= __builtin_amdgcn_frexp_mant(ls);
          ex += __builtin_amdgcn_frexp_exp(ls) - 256 * sc;
        }
      }
      if (!side) {
        const double tot = wave_sum(log(mant) + (double)ex * LOG_2);
        if (lane == 0) sh.bc[16 + wv] = tot;
        __syncthreads();   // also: every wave is past its last table read
        lnl_out = sum_waves<NW>(sh.bc + 16);
      } else {
        __syncthreads();   // every wave is past its last table read
      }
    };

    // f, f' at proposal t: per tile 20 MFMAs contract the register-resident sumtable (A operand:
    // rows = the four sites of a block) with the Newton tables (B operand: column i = table i,
    // column 3 zeros); D puts l_0, l_1, l_2 of a site in lanes 0, 1, 2 of one quad
    uint32_t evals = 0;
    auto derivatives = [&](double t, double& f, double& df) {
#ifdef AAM_PROFILE
      const long long d0_ = clock64();
#endif
#if AAM_PRIVATE_TABLES
      double* const mytab = &ntab_all[wv][0][0];
#pragma unroll
      for (int i = 0; i < NE2; ++i) {
        const int kx = lane + 64 * i;
        if (kx < CS) {
          const double2 c01 = *reinterpret_cast<const double2*>(&sh.ncst[kx][0]);
          const double2 c23 = *reinterpret_cast<const double2*>(&sh.ncst[kx][2]);
          const double e = exp_tab(c01.x * t, sh.e2t);
          mytab[npos[i]] = e * c01.y;
          mytab[tab_stride<NC>() + npos[i]] = e * c23.x;
          mytab[2 * tab_stride<NC>() + npos[i]] = e * c23.y;
        }
      }
#ifdef AAM_PROFILE
      const long long d1_ = clock64();
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
#pragma unroll
      for (int i = 0; i < NE; ++i)
        if (tid + i * NTHR < NENT) sh.tab[tslot[i]][tpos[i]] = exp_tab(t_lr[i] * t, sh.e2t) * t_c[i];
#ifdef AAM_PROFILE
      const long long d1_ = clock64();
#endif
      __syncthreads();
#endif
#ifdef AAM_PROFILE
      const long long d2_ = clock64();
#endif
      const int row = lane & 3;
      double fl = 0.0, dfl = 0.0;
      // the B operands (this lane's five table entries of each category) are the same for every tile:
      // read once per evaluation; row 3 reads the zero table: no per-value select
      double av[NC][NTS];
      {
        const int zt = zero_after(t);
#if AAM_PRIVATE_TABLES
        const double* const rowp = row < 3 ? mytab + row * tab_stride<NC>() : &sh.tab[3][0];   // row 3: the zero table
#else
        const double* const rowp = &sh.tab[row][0];
#endif
#pragma unroll
        for (int cat = 0; cat < NC; ++cat) lds5(rowp + (cat * 4 + kq) * 6 + zt, av[cat]);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!tile_on[j]) continue;
        // four independent accumulation chains (one per category), issued round robin: a chain's next
        // MFMA never waits for its predecessor's result (two chains of ten back to back did)
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int tt = 0; tt < NTS; ++tt)
#pragma unroll
          for (int cat = 0; cat < NC; ++cat) acc[cat & 3] = mfma4(Sm[j][cat][tt], av[cat][tt], acc[cat & 3]);
        // lane (i = lane / 16, block, r): l_r of site 4 block + i of the tile
        const double lr = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        const double l0 = lr, l1 = quad_bcast<1>(lr), l2 = quad_bcast<2>(lr);
        const uint32_t dsite = 16u * (uint32_t)(wv + NW * j) + (uint32_t)(((lane >> 2) & 3) * 4 + kq);
        if (row == 0 && dsite < n) {
          const double inv = fast_rcp(l0);
          const double d1 = -l1 * inv;
          fl += d1;
          dfl += fma(d1, d1, -l2 * inv);
        }
      }
#ifdef AAM_PROFILE
      const long long d3_ = clock64() + (long long)(fl != 12345.678 ? 0 : 1);
#endif
      double ft, dft;
      wave_sum2(fl, dfl, ft, dft);
#if AAM_PRIVATE_TABLES
      double* const bcp = sh.bc2[bcpar];
      bcpar ^= 1;
#else
      double* const bcp = sh.bc;
#endif
      if (lane == 0) { bcp[wv] = ft; bcp[8 + wv] = dft; }
#ifdef AAM_PROFILE
      const long long d4_ = clock64() + (long long)(ft != 12345.678 ? 0 : 1);
#endif
      __syncthreads();   // the evaluation's one barrier (two-barrier form: also "every wave is past its table reads")
      f = sum_waves<NW>(bcp);
      df = sum_waves<NW>(bcp + 8);
      ++evals;
#ifdef AAM_PROFILE
      const long long d5_ = clock64() + (long long)(f != 12345.678 ? 0 : 1);
      cyc_d[0] += d1_ - d0_; cyc_d[1] += d2_ - d1_; cyc_d[2] += d3_ - d2_; cyc_d[3] += d4_ - d3_; cyc_d[4] += d5_ - d4_;
#endif
    };

    // pllmod_opt_minimize_newton (rtsafe-style), uniform across the workgroup
    auto newton_body = [&](double x1, double xguess, double x2, double tol, int max_iters) -> double {
      double rts = xguess, f, df, xl, xh, dx;
      if (rts < x1) rts = x1;
      if (rts > x2) rts = x2;
      derivatives(rts, f, df);
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
        if (!isfinite(f) || !isfinite(df)) return NAN;
        if (df > 0.0 && fabs(f) < tol) return rts;
        if (f < 0.0) xl = rts; else xh = rts;
      }
      return NAN;
    };

    // A Newton solve is a chain of short dependent steps (exp, table round trip, 40 MFMAs, ratios, reductions, two
    // barriers, the scalar decisions): while it runs its wave is raised above the OTHER workgroup's wave on the same
    // SIMD, which is typically streaming the 75-MFMA tile bodies of a phase -- the latency-bound wave issues as soon as
    // the matrix pipe frees up instead of queueing behind an equally old stream (s_setprio; AAM_SETPRIO=0: off)
#ifndef AAM_SETPRIO
#define AAM_SETPRIO 1
#endif
    auto newton = [&](double x1, double xguess, double x2, double tol, int max_iters) -> double {
      if (AAM_SETPRIO) __builtin_amdgcn_s_setprio(3);
      const double r = newton_body(x1, xguess, x2, tol, max_iters);
      if (AAM_SETPRIO) __builtin_amdgcn_s_setprio(0);
      return r;
    };
    double tp = a.blo.pendant_default, td = orig * 0.5, tx = orig * 0.5;
    uint32_t rounds = 0, reverted = 0;
    // window lnL at the pendant length of table slot 2 from the sumtable in registers (LOCAL only:
    // the edge lnL after the last pendant solve of a round)
    auto lnl_from_sumtable = [&](double& lnl_out) {
      double mant = 1.0;
      int ex = 0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!tile_on[j]) continue;
        double l0 = 0.0;
        const int zt = zero_after(mant);
#pragma unroll
        for (int cat = 0; cat < NC; ++cat) {
          double e2[NTS];
          lds5(&sh.tab[2][(cat * 4 + kq) * 6 + zt], e2);
#pragma unroll
          for (int t = 0; t < NTS; ++t) l0 = fma(Sm[j][cat][t], e2[t], l0);
        }
        double ls = rows_sum(l0);
        const bool mine = valid[j] && kq == 0;
        if (!mine) ls = 1.0;
        const int sc = mine ? (int)(scp[sscl[j]] + (resc_keep[j] ? 1u : 0u)) : 0;
        mant *= __builtin_amdgcn_frexp_mant(ls);
        ex += __builtin_amdgcn_frexp_exp(ls) - 256 * sc;
      }
      const double tot = wave_sum(log(mant) + (double)ex * LOG_2);
      if (lane == 0) sh.bc[16 + wv] = tot;
      __syncthreads();   // also: every wave is past its last table read
      lnl_out = sum_waves<NW>(sh.bc + 16);
    };

    double lnl_now = 0.0;
#ifdef AAM_PROFILE
#define AAM_T0 const long long t0_ = clock64();
#define AAM_T1(x) x += clock64() - t0_;
#else
#define AAM_T0
#define AAM_T1(x)
#endif
    publish(td, tx, tp);
    if (a.refI) phase(2, lnl_now); else phase(0, lnl_now);
    double loglikelihood = -lnl_now;
    uint32_t smoothings = a.blo.max_rounds;
    if constexpr (LOCAL) {
      // --raxml-blo: pllmod_opt_optimize_branch_lengths_local(radius 1, keep_update 1) on the triplet
      // (optimize.cpp:274-279; restated in oracle/epa_oracle.c opt_local): NR on the pendant edge, on
      // the distal edge (inner vector re-aimed at it), on the proximal edge (with the new distal
      // length), the inner vector re-aimed at the query, NR on the pendant edge once more, edge lnL.
      const double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
      auto solve = [&](double cur) -> double {
        double g = cur;
        if (g < xmin || g > xmax) g = a.blo.default_branch;
        const double r = newton(xmin, g, xmax, xtol, (int)a.blo.max_newton);
        return (isfinite(r) && fabs(cur - r) > 1e-10) ? r : cur;   // keep_update
      };
      double dummy;
      while (smoothings) {
        tp = solve(tp);
        publish(tp, tx, tp);
        phase(1, dummy);
        td = solve(td);
        publish(tp, td, tp);
        phase(3, dummy);
        tx = solve(tx);
        publish(td, tx, tp);
        phase(0, lnl_now);
        tp = solve(tp);
        publish(td, tx, tp);
        lnl_from_sumtable(lnl_now);
        const double new_ll = -lnl_now;
        ++rounds;
        --smoothings;
        if (fabs(new_ll - loglikelihood) < a.blo.epsilon) smoothings = 0;
        loglikelihood = new_ll;
      }
    }
    while (!LOCAL && smoothings) {
      const double old_td = td, old_tp = tp;
      double xmin = a.blo.min_branch, xmax = a.blo.max_branch, xtol = xmin / 10.0;
      double xguess = tp;
      if (xguess < xmin || xguess > xmax) xguess = a.blo.default_branch;
      double xres;
      { AAM_T0 xres = newton(xmin, xguess, xmax, xtol, (int)a.blo.max_newton); AAM_T1(cyc_newton) }
      if (xres > 0.0) tp = xres;
      { AAM_T0 publish(tp, tx, tp); AAM_T1(cyc_pub) }
      double dummy;
      { AAM_T0 phase(1, dummy); AAM_T1(cyc_phase) }
      xguess = td;
      xmin = fmin(a.blo.min_branch / 2.0, orig / 2.0);
      xtol = xmin / 10.0;
      xmax = orig - xtol;
      if (xguess < xmin || xguess > xmax) xguess = orig / 2.0;
      { AAM_T0 xres = newton(xmin, xguess, xmax, xtol, (int)a.blo.max_newton); AAM_T1(cyc_newton) }
      if (xres > 0.0) { td = xres; tx = orig - xres; }
      { AAM_T0 publish(td, tx, tp); AAM_T1(cyc_pub) }
      { AAM_T0 phase(0, lnl_now); AAM_T1(cyc_phase) }
#ifdef AAM_PROFILE
      n_phase += 2;
#endif
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
#ifdef AAM_PROFILE
    ++n_pairs_done;
#endif
    if (ctr) {
      // the last phase of the pair ended with a workgroup barrier: next_pair has been read by all
      if (tid == 0) sh.next_pair = ahead;
      __syncthreads();
      cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.next_pair);
    } else {
      cur += wgs_per_xcd;
    }
  }
#ifdef AAM_PROFILE
  cyc_total = clock64() - cyc_begin;
  if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 777))
    printf("AAM blk %d: pairs %lld total %lld phase %lld (%lld phases) newton %lld (%u evals) publish %lld | per eval: exp+write %lld barrier %lld reads+mfma+ratio %lld wavesum+write %lld barrier+sum %lld\n", (int)blockIdx.x,
           n_pairs_done, cyc_total, cyc_phase, n_phase, cyc_newton, wevals, cyc_pub, cyc_d[0] / (wevals ? wevals : 1), cyc_d[1] / (wevals ? wevals : 1),
           cyc_d[2] / (wevals ? wevals : 1), cyc_d[3] / (wevals ? wevals : 1), cyc_d[4] / (wevals ? wevals : 1));
#endif
  if (tid == 0) {
    atomicAdd(&a.stats[0], (unsigned long long)wrounds);
    atomicAdd(&a.stats[1], (unsigned long long)wevals);
    atomicAdd(&a.stats[2], (unsigned long long)wreverts);
    if (a.xstamp) atomicMax(&a.stats[8 + xcd], ((unsigned long long)(__builtin_amdgcn_s_memrealtime() & EPA_XSTAMP_MASK) << 21) |
                                                     (unsigned long long)(a.xcum[xcd + 1] - a.xcum[xcd]));
    if (a.xstamp && blockIdx.x == 0) {   // the shader clock this launch ran at (epa_dev_last_sclk_mhz)
      a.stats[5] = __builtin_amdgcn_s_memtime() - clk0;
      a.stats[6] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
  }
}

}  // namespace

// windows up to 384 sites (4 rate categories) / 256 sites (8); the caller routes longer windows elsewhere
int launch_thorough_aa_mfma(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* d_order, uint64_t n_pairs,
                            const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span,
                            uint32_t max_span, epa_result* d_out, unsigned long long* d_stats) {
  ThArgsAM a;
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
  // (an LDS cache of the pair's reference windows was measured and dropped: a third fewer HBM reads,
  // 1 - 6 % slower -- profiles/r2_aa_mfma_ab.txt)
  const uint32_t tiles = (max_span + 15) / 16;
  const int nc = ctx->c;
  if (!(nc == 4 || nc == 8)) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "thorough_aa_mfma: 4 or 8 rate categories");
  if (tiles > (nc == 4 ? 24u : 16u))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "thorough_aa_mfma: window longer than 384 (4 categories) / 256 (8) sites");
  // waves per pair: 4 (two workgroups per CU) or 8 (one); tiles per wave NT
  const int nw = nc == 4 ? (tiles <= 12 ? 4 : 8) : (tiles <= 4 ? 4 : 8);
  const uint32_t wg_per_cu = nw == 4 ? 2u : 1u;
  // resident workgroups + a work counter per XCD slice (EPA_TH_QUEUE=0: an oversubscribed static grid,
  // the dispatcher balances the cost spread of the pairs)
  uint32_t per_slot = 16;
  uint32_t nwg = (uint32_t)std::min<uint64_t>(n_pairs, (uint64_t)ctx->n_cu * wg_per_cu * per_slot);
  a.qctr = nullptr;
  if (ctx->th_ctr) {
    { const int zr = epa_th_ctr_reset(ctx); if (zr) return zr; }
    a.qctr = epa_th_ctr(ctx);
    nwg = (uint32_t)std::min<uint64_t>(n_pairs, (uint64_t)ctx->n_cu * wg_per_cu);
  }
  nwg = (nwg + 7) / 8 * 8;
  for (int i = 0; i < 9; ++i) a.xcum[i] = ctx->xcd_cum[i];
  a.xstamp = (a.qctr && ctx->xstamp_ok) ? 1u : 0u;
#define AAM(NT_, NC_, NW_)                                                                                        \
  do {                                                                                                            \
    if (!ctx->blo.sliding) hipLaunchKernelGGL((k_thorough_aa_mfma<NT_, true, NC_, NW_>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a); \
    else hipLaunchKernelGGL((k_thorough_aa_mfma<NT_, false, NC_, NW_>), dim3(nwg), dim3(64 * (NW_)), 0, ctx->stream, a);                 \
  } while (0)
  if (nc == 4) {
    if (tiles <= 4) AAM(1, 4, 4);
    else if (tiles <= 8) AAM(2, 4, 4);
    else if (tiles <= 12) AAM(3, 4, 4);
    else if (tiles <= 16) AAM(2, 4, 8);
    else AAM(3, 4, 8);
  } else {
    if (tiles <= 4) AAM(1, 8, 4);
    else if (tiles <= 8) AAM(1, 8, 8);
    else AAM(2, 8, 8);
  }
#undef AAM
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}
