// Internal declarations shared by the HIP translation units of libepa_dev.so.
// gfx950 (MI355X) only: wave64, 160 KiB LDS/CU, fp64 VALU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "epa_dev.h"

#define EPA_MAX_STATES 20
#define EPA_MAX_CATS 16
#define EPA_MAX_COLS 24

// Model constants handed to kernels by value (kernarg -> SGPRs via s_load).
struct ModelDNA {
  double U[16];     // [i][x]
  double Ui[16];    // [x][i]
  double lam[4];
  double rate[16];  // r_k (prop_invar folded in); ng groups of four categories
  double w[16];
  double pi[4];
  int ng;           // category groups: 1 (4 categories) .. 4 (16)
};

struct BloConsts {
  double min_branch, max_branch, default_branch, epsilon, pendant_default;
  uint32_t max_rounds, max_newton;
  uint32_t sliding;
  uint32_t newton_variant;  // EPA_FLAG_NEWTON_* bits: which pllmod_opt_minimize_newton is replicated
};

// Generic model block in HBM (used by the setup kernels and the 20-state path)
struct ModelDev {
  int s, c, ncols;
  double U[EPA_MAX_STATES * EPA_MAX_STATES];
  double Ui[EPA_MAX_STATES * EPA_MAX_STATES];
  double lam[EPA_MAX_STATES];
  double pi[EPA_MAX_STATES];
  double rate[EPA_MAX_CATS];
  double w[EPA_MAX_CATS];
  uint32_t colmask[EPA_MAX_COLS];  // lookup column -> state set
  // eigen-space image of each column's 0/1 tip vector: qt[col][x] = sum_{m in mask} Ui[x][m]
  double qt[EPA_MAX_COLS * EPA_MAX_STATES];
};

struct EvTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool valid = false;
};

// One slot of the double-buffered chunk pipeline (epa_dev_chunk_stage / _launch / _finish): the
// query upload of chunk k+1 and the result download of chunk k-1 run on the context's copy stream
// while the kernels of chunk k run on its compute stream.
// state of a candidate selection between launch_select_begin and launch_select_end (preplace.hip)
struct SelectPending {
  const double* d_lnl = nullptr;
  uint32_t Q = 0, cap = 0;
  double threshold = 0.0;
  epa_pair* d_pairs = nullptr;
  uint64_t max_pairs = 0;
  const uint32_t* d_span = nullptr;
  unsigned long long *stage = nullptr, *keys_a = nullptr, *keys_b = nullptr;
  uint32_t *counts = nullptr, *offsets = nullptr;
  uint32_t *bitmap = nullptr, *boffs = nullptr;   // bitmap form of the selection (preplace.hip, SelOut)
  uint32_t wpr = 0;
  void* temp = nullptr;
  size_t sort_bytes = 0;
  uint32_t* rb = nullptr;   // host read-back block, 64 words
  bool have_status = false;
  bool rerun = false;
  hipEvent_t ev_rb = nullptr;      // recorded behind the read-back copy: launch_select_end waits for this, not for the stream
  const uint32_t* d_rb = nullptr;  // the packed read-back block on the device (bitmap form)
  bool emitted = false;            // k_emit_pairs (guarded by the total) is already queued behind the read-back
  int queued_cls = -1;             // >= 0: a thorough launch for this span class is queued too (launch_thorough_queued)
  const uint32_t* pre_status = nullptr;   // window-validation words of THIS chunk's preplacement (ctx->d_status may belong
                                          // to another pipeline slot by the time a widened re-run packs the read-back)
};

struct ChunkSlot {
  void* h_in = nullptr;       // pinned bounce buffer: codes | win_begin | win_span
  size_t h_in_sz = 0;
  void* d_in = nullptr;       // the same three arrays in HBM
  const uint8_t* x_codes = nullptr;   // HBM-resident chunk staged in place (caller's arrays), else null
  const uint32_t* x_begin = nullptr;
  const uint32_t* x_span = nullptr;
  size_t d_in_sz = 0;
  uint8_t* d_unpacked = nullptr;  // one-byte codes when the chunk arrived in the 4-bit wire format
  size_t d_unpacked_sz = 0;
  size_t codes_bytes = 0;
  uint32_t Q = 0, stride = 0;
  bool packed4 = false;
  epa_pair* d_pairs = nullptr;    // slot-owned result buffers (used when the caller passes none)
  epa_result* d_res = nullptr;
  size_t cap = 0;
  void* h_out = nullptr;      // pinned: pairs | results
  size_t h_out_sz = 0;
  unsigned long long* h_stats = nullptr;  // pinned, 16 words
  unsigned long long* d_stats = nullptr;  // 16 words in HBM
  hipEvent_t ev_up = nullptr, ev_done = nullptr, ev_down = nullptr, ev_base = nullptr;
  hipStream_t stream = nullptr;  // the slot's own compute stream: the kernels of the two chunks in flight overlap
  const epa_pair* out_pairs = nullptr;    // what finish() hands out
  const epa_result* out_res = nullptr;
  uint64_t n = 0;
  int state = 0;              // 0 free, 1 staged, 3 launch begun (selection in flight), 2 launched, 4 member of a group whose launch was begun
  // between launch_begin and launch_end
  SelectPending sel;
  uint32_t* h_sel = nullptr;  // pinned, 64 words: the selection's read-back block
  const uint8_t* l_codes = nullptr;
  const uint32_t *l_begin = nullptr, *l_span = nullptr;
  epa_pair* l_pairs = nullptr;
  epa_result* l_res = nullptr;
  uint32_t l_max_span = 0, l_flags = 0;
  // group launches (epa_dev_chunk_launch_many): ONE chunk body over the concatenated queries of up to
  // EPA_MAX_GROUP staged slots -- one preplacement, one selection, one Newton launch instead of one each per
  // small chunk -- run on the first slot (the leader); a kernel regroups the (pair, result) rows by member
  // afterwards (stable: every member keeps the branch-major order of its own chunk, sequence ids local again).
  int leader = -1;            // >= 0: this slot's chunk is member g_index of slot `leader`'s group (the leader: itself)
  int g_index = 0;
  int g_n = 0;                // leader: members of the group in flight (0: an ordinary launch)
  int g_left = 0;             // leader: members not finished yet (their results live in the leader's buffers)
  int g_slots[8] = {};
  uint32_t g_qoff[9] = {};    // leader: first merged query index of every member
  uint32_t own_Q = 0;         // leader: its own chunk (restored after the group launch / on a recoverable error)
  const uint8_t* own_x_codes = nullptr;
  const uint32_t *own_x_begin = nullptr, *own_x_span = nullptr;
  void* d_merge = nullptr;    // leader: merged codes | win_begin | win_span
  size_t d_merge_sz = 0;
  epa_pair* d_gpairs = nullptr;   // leader: rows regrouped by member
  epa_result* d_gres = nullptr;
  size_t g_cap = 0;
  uint32_t* d_goff = nullptr;     // [9] member offsets into the regrouped rows (device / pinned)
  uint32_t* h_goff = nullptr;
};
constexpr int EPA_MAX_GROUP = 8;

// Diagnostic switches of a context (epa_dev_set_option; include/epa_dev.h lists them).  They select between code
// paths that return the same results -- cross-check kernels for the parity tests, A/B switches of the bench -- and
// replace the environment variables earlier rounds read inside the library.
struct EpaOptions {
  int thorough_generic = 0;   // every Newton launch on k_thorough_generic (the reference-shaped kernel)
  int preplace_generic = 0;   // preplacement on k_preplace (per-site gathers) instead of the pair / site fast paths
  int select_full_rows = 0;   // candidate selection from whole table rows instead of the segment maxima
  int select_sort = 0;        // candidate list through the sorted staging path instead of the bitmap
  int queued_thorough = 0;    // the Newton launch queued behind the selection, guarded by the device-side count
  int xcd_balance = 1;        // XCD shares of a Newton launch follow the measured speeds (epa_xcd_feedback)
  int aa_valu = 0;            // 20-state windows on the lane = site VALU kernel instead of the matrix-core kernel
  int timers = 1;             // hipEvent records around the kernel families (epa_dev_last_kernel_ms)
};

struct epa_ctx {
  int device = 0;
  EpaOptions opt;
  int n_cu = 256;  // compute units of the device (persistent-grid sizing)
  hipStream_t stream = nullptr;
  std::string err;

  int s = 0, c = 0, ncols = 0;
  int c_in = 0;  // rate categories of the caller's CLVs (1 or 2 are replicated to c = 4)
  uint32_t W = 0, B = 0;
  ModelDev hmodel;          // host copy
  ModelDev* dmodel = nullptr;
  ModelDNA dna;             // valid when s == 4 && c == 4
  bool dna_zero0 = false;   // eigenvalue 0 is the (exactly) zero one after the create-time reorder
  bool rate_scalers = false;  // per-rate scalers (EPA_FLAG_RATE_SCALERS): scSum is [B][c][W]
  // the tuned thorough kernels serve 4 categories + per-site scalers + sliding BLO; everything
  // else runs on k_thorough_generic (thorough_generic.hip)
  bool generic_thorough = false;
  bool generic_native = false;   // ... by the context's shape (generic_thorough may also be the option thorough_generic)
  BloConsts blo;
  int aa_x_as_n = 0;
  uint32_t code_stride = 0;  // 0: query codes are Q x W rows; S: compact rows of S bytes (window only)

  // HBM-resident reference data
  //   refT   [2B][c*s][W]  eigen-transformed CLVs, component-major (side 0 proximal, 1 distal)
  //   scSum  [B][W]        prox + dist per-site scaler counts ([B][c][W] with per-rate scalers)
  //   blen   [B]
  //   lookup [B][W][ncols]
  //   lookup2 [B][2][ceil(W/2)][36]  (start parity, site >> 1)  DNA: lookup[s][c0] + lookup[s+1][c1] over {A,C,G,T,N,none}^2
  double* refT = nullptr;
  uint32_t* scSum = nullptr;
  double* blen = nullptr;
  double* lookup = nullptr;
  double* refI = nullptr;     // c == 4: [B][c*s][W] U^-1 image of the inner CLV toward the query at
                              // the starting lengths (orig/2, orig/2), rescaled; resc0 [B][W] its flag
  uint8_t* resc0 = nullptr;
  double* cinv = nullptr;     // +I only: [W] p * pi[invariant state of the site] (0 where not invariant)
  double inv_w0 = 0.0;        // 1 / w_0: folds cinv into the zero-eigenvalue sumtable entry (thorough)
  double* lookup2 = nullptr;  // DNA only: [B][2][ceil(W/2)][36] site-pair sums (preplace.hip, k_preplace_pairs)
  bool lookup_built = false;
  std::vector<double> h_blen;

  // per-call scratch (grown on demand)
  // Banks: 0 = the direct entry points (caller's stream), 1 .. N_SLOTS = the slots of the chunk
  // pipeline, whose kernels run concurrently on their own streams and therefore share no scratch
  // (allocated on first use: a two-slot caller pays for two).
  static constexpr int N_SCRATCH = 13;
  static constexpr int N_SLOTS = 24;
  static constexpr int N_BANKS = 1 + N_SLOTS;
  int bank = 0;
  void* scratch[N_BANKS * N_SCRATCH] = {};
  size_t scratch_sz[N_BANKS * N_SCRATCH] = {};

  uint32_t* d_status = nullptr;   // window-validation words of the last preplace (in scratch 6)
  // span-class histogram of the candidate pairs of the last select (valid for the thorough call
  // that follows it in the fused path: saves one device round trip)
  uint32_t cls_hist[16] = {};
  uint64_t cls_hist_pairs = 0;  // 0 = not valid
  uint32_t select_cap = 64;       // staging slots per query of the candidate selection
  uint32_t* th_ctr = nullptr;  // work counters of the thorough kernel (one per XCD slice): 256 B per bank,
                               // [0, 64) the counters, [128, 256) the fused chunk's statistics
  uint32_t lnl_pitch = 0;  // row pitch (doubles) of the table handed to launch_preplace / launch_select; 0 = B
  // fused chunk body only: [Q][segp] order-preserving keys of the per-segment maxima of the table rows
  // (written by the preplacement fast paths, read by k_select_seg; preplace.hip), or null
  unsigned long long* segmax = nullptr;
  uint32_t segp = 0;
  size_t segmax_zero_bytes = 0;   // chunk_body_begin -> launch_preplace: clear segmax with the status / key-count fill
  bool code_packed4 = false;  // q_codes arrive in the 4-bit wire format (epa_dev_set_query_packing)
  int heur_mode = 0;        // EPA_HEUR_* (epa_dev_set_heuristic)
  double heur_param = 0.0;  // fixed: fraction of the branches

  // non-blocking copy streams of the chunk pipeline: uploads and downloads each have their own, so
  // the H2D of chunk k+1 is not queued behind the D2H of chunk k (which waits for k's kernels)
  hipStream_t copy_stream = nullptr, down_stream = nullptr;
  ChunkSlot slots[N_SLOTS];

  // kernel-family timers (epa_dev_last_kernel_ms): one set per scratch bank -- the pipeline slots run
  // concurrently on their own streams, a single set would have its start event re-recorded by slot k + 1
  // before slot k records its stop; t_last = the bank whose timer was stopped last
  enum { T_PREPLACE = 0, T_THOROUGH = 1, T_SELECT = 2 };
  EvTimer t_lookup;
  EvTimer t_bank[N_BANKS][3];
  // shares of the branch-sorted pair list the eight XCDs take in the single-wave Newton launches (cumulative, 20-bit
  // fixed point; thorough_dna.hip ThArgs::xcum), adapted to the speeds the XCDs showed: epa_xcd_feedback
  uint32_t xcd_cum[9] = {0u, 1u << 17, 2u << 17, 3u << 17, 4u << 17, 5u << 17, 6u << 17, 7u << 17, 1u << 20};
  double xcd_w[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
  double last_sclk_mhz = 0.0;   // shader clock of the last stamped Newton launch (s_memtime / s_memrealtime)
  bool xstamp_ok = false;   // launch_thorough: this call is ONE kernel launch (one span class): its stamps mean something
  hipEvent_t ev_rb[N_BANKS] = {};   // per bank: behind the selection's read-back copy (SelectPending::ev_rb)
  // per bank: the statistics block / the work counters of the NEXT Newton launch were already zeroed behind the
  // selection (chunk_body_begin: in the shadow of the host's round trip); the launch that uses them clears the mark
  const void* clean_stats[N_BANKS] = {};
  bool clean_ctr[N_BANKS] = {};
  int t_last[3] = {0, 0, 0};
  epa_thorough_stats last_stats{};
};

// ---- helpers (epa_dev.hip)
int epa_fail(epa_ctx* ctx, int code, const std::string& msg);
void* epa_scratch(epa_ctx* ctx, int slot, size_t bytes);
inline uint32_t* epa_th_ctr(epa_ctx* ctx) { return ctx->th_ctr ? ctx->th_ctr + 64 * ctx->bank : nullptr; }
// Zero `bytes` at p (and optionally a second range) with ONE plain kernel on the context's stream.  A hipMemsetAsync is
// a blit launch of its own with ~6 us of idle queue before and after it between kernels (profiles/r5_step_timeline.txt:
// five of them per chunk body); a plain kernel follows its predecessor without a gap.  Ranges must be 16-byte aligned
// multiples of 16 bytes (else: hipMemsetAsync).
int epa_zero_async(epa_ctx* ctx, void* p, size_t bytes, void* p2 = nullptr, size_t bytes2 = 0);
// the work counters of this bank's next Newton launch: zero them unless chunk_body_begin already did
inline int epa_th_ctr_reset(epa_ctx* ctx) {
  if (ctx->clean_ctr[ctx->bank]) { ctx->clean_ctr[ctx->bank] = false; return 0; }
  return epa_zero_async(ctx, epa_th_ctr(ctx), 64);
}
bool epa_is_device_ptr(const void* p);
// returns a device pointer holding `bytes` of *p (copying into scratch slot if p is on the host)
const void* epa_to_device(epa_ctx* ctx, int slot, const void* p, size_t bytes);
// query code rows -> device in the one-byte layout the kernels read (unpacks the 4-bit wire format)
const uint8_t* epa_codes_to_device(epa_ctx* ctx, const uint8_t* q_codes, uint32_t Q);
// query code rows -> device in the one-byte layout the kernels read (unpacks the 4-bit wire format)
const uint8_t* epa_codes_to_device(epa_ctx* ctx, const uint8_t* q_codes, uint32_t Q);
// hst: the 16 statistics words of a thorough launch ([7] start, [8 + x] last exit of XCD x: ThArgs::xstamp)
void epa_xcd_feedback(epa_ctx* ctx, uint64_t n_pairs, const unsigned long long* hst);
// in-kernel s_memrealtime stamps are kept to 43 bits (24 h at 100 MHz) so that the XCD's share fits below them
#define EPA_XSTAMP_MASK 0x7ffffffffffull
// Kernel-family timers (hipEventRecord on the context's stream).  Each record costs ~5.5 us of idle queue between two
// kernels (profiles/r5_step_timeline.txt: six per chunk body); events riding on the launches themselves
// (hipExtLaunchKernelGGL start / stop events) were built and cost exactly the same on this runtime: not kept.
void epa_timer_start(epa_ctx* ctx, EvTimer& t);
void epa_timer_stop(epa_ctx* ctx, EvTimer& t);
// the current bank's timer of a kernel family (epa_ctx::T_*)
inline EvTimer& epa_t(epa_ctx* ctx, int which) { ctx->t_last[which] = ctx->bank; return ctx->t_bank[ctx->bank][which]; }

#define EPA_HIP(ctx, call)                                                               \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess)                                                                \
      return epa_fail(ctx, EPA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// ---- span classes of the thorough kernels: a launch covers pairs whose window needs the same
// kernel instantiation, so one long window does not drag a whole chunk onto the big-register
// variant.  DNA: sites per lane NCH in {1,2,3,4,6,8,12,16,24} (class 0..8), 9 = HBM-slab kernel;
// 20 states: 0 / 1 / 2 = windows up to 64 / 128 / 192 sites (k_thorough_aa_mfma<1/2/3>: sumtable
// in registers), 3 = longer (k_thorough_aa with the HBM slab).
constexpr int EPA_N_CLS = 12;
constexpr uint32_t EPA_AA_LDS_MAX_SPAN = 102;
__host__ __device__ inline int epa_span_class(int states, uint32_t span) {
  if (states != 4) return span <= 64 ? 0 : span <= 128 ? 1 : span <= 192 ? 2 : 3;
  // DNA classes 10 / 11: windows of 65..96 / 129..160 sites, whose last 64-lane chunk is at most half
  // full: the half-chunk instantiations of k_thorough_dna (thorough_dna.hip, TAILH)
  if (span > 64 && span <= 96) return 10;
  if (span > 128 && span <= 160) return 11;
  const uint32_t nch = (span + 63) / 64;
  return nch <= 4 ? (nch ? (int)nch - 1 : 0) : nch <= 6 ? 4 : nch <= 8 ? 5 : nch <= 12 ? 6 : nch <= 16 ? 7 : nch <= 24 ? 8 : 9;
}

// ---- kernel launchers
int launch_transform(epa_ctx* ctx, const double* d_clv_or_null, const uint8_t* d_tip_or_null,
                     const uint32_t* d_tipmap, uint32_t tipmap_size, double* dst);
int launch_build_lookup(epa_ctx* ctx);
int launch_build_lookup2(epa_ctx* ctx);  // DNA site-pair table from `lookup` (preplace.hip)
int launch_preplace(epa_ctx* ctx, const uint8_t* d_codes, const uint32_t* d_begin,
                    const uint32_t* d_span, uint32_t Q, double* d_lnl, uint32_t max_span);
int preplace_check_status(epa_ctx* ctx);
int launch_thorough(epa_ctx* ctx, const epa_pair* d_pairs, uint64_t n_pairs, const uint8_t* d_codes,
                    const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                    epa_result* d_out, unsigned long long* d_stats);
int launch_thorough_queued(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* d_spec, uint64_t max_pairs,
                           const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                           epa_result* d_out, unsigned long long* d_stats);
int launch_select_emit(epa_ctx* ctx, SelectPending* sp);   // queues the guarded k_emit_pairs behind launch_select_begin
int launch_thorough_aa(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* d_order, uint64_t n_pairs,
                       const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span,
                       uint32_t max_span, bool want_lds, epa_result* d_out, unsigned long long* d_stats);
int launch_thorough_generic(epa_ctx* ctx, const epa_pair* d_pairs, uint64_t n_pairs, const uint8_t* d_codes,
                            const uint32_t* d_begin, const uint32_t* d_span, uint32_t max_span,
                            epa_result* d_out, unsigned long long* d_stats,
                            const uint32_t* d_order = nullptr, bool caller_times = false);   // caller_times: a per-class launch inside launch_thorough's timed region
int launch_thorough_aa_mfma(epa_ctx* ctx, const epa_pair* d_pairs, const uint32_t* d_order, uint64_t n_pairs,
                            const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span,
                            uint32_t max_span, epa_result* d_out, unsigned long long* d_stats);
int launch_select(epa_ctx* ctx, const double* d_lnl, uint32_t Q, double threshold,
                  epa_pair* d_pairs, uint64_t max_pairs, uint64_t* n_pairs,
                  const uint32_t* d_span = nullptr);  // d_span: also histogram the span classes
int launch_select_begin(epa_ctx* ctx, const double* d_lnl, uint32_t Q, double threshold, epa_pair* d_pairs,
                        uint64_t max_pairs, const uint32_t* d_span, uint32_t* rb, SelectPending* sp);
int launch_select_end(epa_ctx* ctx, SelectPending* sp, uint64_t* n_pairs);
int select_check_status(epa_ctx* ctx, const SelectPending* sp);
