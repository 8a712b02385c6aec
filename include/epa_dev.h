/*
 * epa_dev.h -- C-ABI of the MI355X placement evaluator (libepa_dev.so).
 *
 * This is the drop-in boundary for EPA-ng's placement hot path.  The reference has no FFI for
 * this path; the seam is the pair of static templates called from the chunk loop
 *     place(chunk, tree, branches, preplace, options, lookups)            src/core/place.cpp:41-95,221
 *     place_thorough(blo_work, chunk, tree, branches, blo_sample, ...)    src/core/place.cpp:97-171,234
 * and, below them, the per-branch evaluator
 *     Tiny_Tree::Tiny_Tree(edge, branch_id, Tree&, opt_branches, ...)     src/tree/Tiny_Tree.cpp:48-129
 *     Placement Tiny_Tree::place(const Sequence&)                         src/tree/Tiny_Tree.cpp:131-218
 * together with the libpll / pll-modules calls they wrap.
 * Paths are relative to the reference checkout.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *  - plain C, no torch / HIP types; every `const T*` data argument may be a host pointer or a
 *    device (HBM) pointer -- the library inspects it (hipPointerGetAttributes) and copies only
 *    when it lives on the host.  Outputs follow the same rule.
 *  - all likelihood arithmetic is fp64.
 *  - every entry point returns EPA_OK or a negative epa_status; epa_dev_last_error() returns
 *    the message the reference would have thrown (std::runtime_error text) where one exists.
 *  - one epa_ctx per GPU; a ctx is thread-compatible (one caller at a time), entry points are
 *    called once per chunk, outside any OpenMP region.
 */
#ifndef EPA_DEV_H
#define EPA_DEV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct epa_ctx epa_ctx;

typedef enum {
  EPA_OK = 0,
  EPA_ERR_INVALID_ARG = -1,   /* bad descriptor / null pointer / unsupported shape            */
  EPA_ERR_HIP = -2,           /* HIP runtime failure (message carries hipGetErrorString)       */
  EPA_ERR_NO_DEVICE = -3,     /* no gfx950 device visible: there is NO CPU fallback            */
  EPA_ERR_QUERY_WIDTH = -4,   /* "Query sequence length not same as reference alignment!"      *
                               *   Tiny_Tree.cpp:145-147                                       */
  EPA_ERR_QUERY_ALL_GAP = -5, /* "... does not appear to have any non-gap sites!" :153-156     */
  EPA_ERR_INVALID_CHAR = -6,  /* "char is invalid!" Lookup_Store.hpp:100-108 (quirk D5: the     *
                               *   reference does not validate; this library does on ingest)   */
  EPA_ERR_NEG_INF = -7,       /* "-INF logl at branch ..." Tiny_Tree.cpp:209-212               */
  EPA_ERR_UNSUPPORTED = -8,   /* feature marked "next" in SURVEY.md section 8f                 */
  EPA_ERR_PAIR_OVERFLOW = -9  /* the candidate selection found more pairs than max_pairs: call    *
                               *   again with larger buffers (a staged chunk stays staged)       */
} epa_status;

/* flags of epa_ref_desc.flags */
#define EPA_FLAG_SLIDING_BLO 0x1u  /* Options::sliding_blo (default on, src/util/Options.hpp:16) */
#define EPA_FLAG_RAXML_BLO 0x4u    /* --raxml-blo: pllmod_opt_optimize_branch_lengths_local, radius 1 *
                                    * (src/core/pll/optimize.cpp:274-279) instead of the sliding rule */
/* pllmod_opt_minimize_newton is not in the reference's tree; the routine this library replicates
 * by default is a recollection (see oracle/epa_oracle.c minimize_newton).  Two details differ
 * between published variants of it and are switchable (profiles/r2_sensitivity.md measures what
 * they change): */
#define EPA_FLAG_NEWTON_SLOW_BISECT 0x8u  /* also bisect when |2 f| > |dx_old f'| (Numerical Recipes rtsafe) */
#define EPA_FLAG_NEWTON_STRICT_DF 0x10u   /* first convergence test needs f' > 0 instead of f' >= 0          */
#define EPA_FLAG_KEEP_EIGENVALUES 0x20u   /* use the caller's eigenvalues verbatim: by default the stationary one
                                          * (a few 1e-17 of either sign out of libpll's eigen-solver, 0 in exact
                                          * arithmetic) is set to exactly 0 and the kernels drop its derivative terms.
                                          * With this flag nothing is snapped and the general kernel evaluates every
                                          * term as libpll does: the switch to A/B the snap against a real libpll
                                          * build (at saturated lengths the residue's sign decides a bisection,
                                          * DESIGN section 2).  Not available with prop_invar > 0. */
#define EPA_FLAG_RATE_SCALERS 0x2u /* PLL_ATTRIB_RATE_SCALERS: every rate category is rescaled on its *
                                    * own (src/tree/tiny_util.cpp:37-44; the reference turns it on    *
                                    * above 2000 tips, src/io/file_io.cpp:211-214, or with            *
                                    * --rate-scalers).  prox_scaler / dist_scaler rows are then      *
                                    * [W][rate_cats] (libpll layout)                                  */

/*
 * Reference-side inputs: what Tiny_Tree's constructor pulls out of `Tree` per branch
 * (Tree::get_clv for the proximal and distal node, their scalers, the branch length:
 * src/tree/tiny_util.cpp:72-199) plus the model arrays the tiny partition shares by pointer
 * (:109-163).  Arrays use libpll's layouts: CLV [site][rate_cat][state] fp64, scaler
 * uint32[site] (or uint32[site][rate_cat] with EPA_FLAG_RATE_SCALERS).
 * Branch b is EPA-ng's branch_id == jplace edge_num (utree_query_branches order,
 * src/core/pll/pll_util.cpp:182-205).  Orientation is the caller's job exactly as in
 * Tiny_Tree.cpp:64-74: when one end of the branch is a tip it is passed as the DISTAL side.
 */
typedef struct {
  uint32_t states;     /* 4 (DNA) or 20 (AA)                                                   */
  uint32_t rate_cats;  /* number of rate categories c                                          */
  uint32_t sites;      /* alignment width W after column pre-masking                           */
  uint32_t branches;   /* B = 2n-3                                                             */

  /* eigen system of the (single) rate matrix:  P(t) = U diag(exp(eigenvals r_k t)) U^-1.
   * libpll stores U as `inv_eigenvecs` and U^-1 as `eigenvecs` (pll_update_eigen).           */
  const double* eigenvals;   /* [states]                                                       */
  const double* eigenvecs_u; /* [states*states] row-major U                                    */
  const double* eigenvecs_uinv; /* [states*states] row-major U^-1                              */
  const double* freqs;       /* [states] stationary frequencies                                */
  const double* rates;       /* [rate_cats]                                                    */
  const double* rate_weights;/* [rate_cats]                                                    */
  double prop_invar;         /* proportion of invariant sites (+I), 0 <= p < 1; see invariant_state */

  /* per branch: B pointers each */
  const double* const* prox_clv;       /* [B] -> [W][c][s]                                     */
  const uint32_t* const* prox_scaler;  /* [B] -> [W], entry or whole array may be NULL (= 0)   */
  const double* const* dist_clv;       /* [B] -> [W][c][s]; NULL where dist_tipchars[b] is set */
  const uint8_t* const* dist_tipchars; /* [B] -> [W] tip codes (index into tipmap); may be NULL */
  const uint32_t* const* dist_scaler;  /* [B] -> [W], NULL for tips                            */
  const double* branch_length;         /* [B] original branch length                           */

  /* tip code -> state set (pll_map_nt / pll_map_aa semantics): bit i = state i allowed.
   * Used for dist_tipchars only; query codes are lookup columns, see epa_dev_preplace.        */
  const uint32_t* tipmap;
  uint32_t tipmap_size;

  /* constants of the branch-length optimiser.  They live in headers absent from the reference
   * tree (pll_optimize.h), hence runtime parameters; 0 selects the default in brackets.       */
  double blo_min_branch;     /* PLLMOD_OPT_MIN_BRANCH_LEN      [1e-4]  optimize.hpp:11          */
  double blo_max_branch;     /* PLLMOD_OPT_MAX_BRANCH_LEN      [100]   optimize.hpp:12          */
  double blo_default_branch; /* PLLMOD_OPT_DEFAULT_BRANCH_LEN  [0.1]   optimize.cpp:143         */
  double blo_epsilon;        /* OPT_BRANCH_EPSILON             [0.1]   optimize.hpp:9           */
  double pendant_default;    /* DEFAULT_BRANCH_LENGTH          [-ln 0.9] util/constants.hpp:12  */
  uint32_t blo_max_rounds;   /* smoothings                     [32]    optimize.cpp:269         */
  uint32_t blo_max_newton;   /* max_iters                      [30]    optimize.cpp:62          */
  uint32_t flags;            /* EPA_FLAG_*; neither BLO bit set selects EPA_FLAG_SLIDING_BLO     */
  uint32_t aa_x_as_n;        /* 1: reproduce quirk D4 (AA 'X' preplaced in the 'N' column,      *
                              *    Lookup_Store.hpp:63-66); 0 (default): 'X' = any             */
  /* +I: state of every site that is invariant over the REFERENCE tips (libpll
   * pll_update_invariant_sites: AND of the tips' state sets is a single state), -1 otherwise;
   * the array the tiny partition borrows at src/tree/tiny_util.cpp:153-156.  [sites]; required
   * by epa_dev_create when prop_invar > 0, optional for epa_dev_create_from_tree (derived from
   * the tip sequences when NULL).  With +I the library follows libpll: rates are divided by
   * (1 - p) in every P-matrix, a site's likelihood is (1-p) L + p pi_state (the second term is
   * added unscaled and regardless of the query's character, as libpll does).                  */
  const int8_t* invariant_state;
} epa_ref_desc;

/*
 * Reference precompute ON the device: instead of 2B host CLVs the caller hands over the tree and
 * the tip sequences and the library computes all 3(n-2) directional CLVs itself (the job of
 * precompute_clvs, src/core/pll/epa_pll_util.cpp:62-107 / Tree::Tree, src/tree/Tree.cpp:16-56),
 * level by level, directly in the eigen-transformed layout the kernels use (nothing but n x W
 * tip codes crosses PCIe).  `ref` carries the model, sizes, tip map, branch lengths and optimiser
 * constants; its per-branch CLV / scaler pointer arrays are ignored.
 * A directional CLV record r is the partial likelihood of the subtree seen from one end of a
 * branch; it is the product of its two children propagated over their branches.  Operand ids:
 * value < inner_records = record, EPA_TIP | t = tip t.  Every inner record must be the proximal
 * or distal side of exactly one branch; a tip end is always passed as the DISTAL side.
 */
#define EPA_TIP 0x80000000u
typedef struct {
  epa_ref_desc ref;
  uint32_t tips;                  /* n                                                          */
  uint32_t inner_records;         /* 3 (n - 2)                                                  */
  const uint8_t* tipchars;        /* [n][sites] tip codes (index into ref.tipmap)               */
  const uint32_t* rec_child_a;    /* [inner_records] operand ids of the two children            */
  const uint32_t* rec_child_b;
  const double* rec_length_a;     /* [inner_records] branch length toward each child            */
  const double* rec_length_b;
  const uint32_t* branch_prox;    /* [B] operand id of the proximal side (an inner record)      */
  const uint32_t* branch_dist;    /* [B] operand id of the distal side (record or tip)          */
} epa_tree_desc;

/* one unit of Work (src/core/Work.hpp:31-34 Work_Pair) */
typedef struct {
  uint32_t branch_id;
  uint32_t seq_id; /* index into the chunk's queries */
} epa_pair;

/* one Placement minus the LWR (src/sample/Placement.hpp:48-53) */
typedef struct {
  double lnl;
  double pendant_length;
  double distal_length;
} epa_result;

/* Counters of the last epa_dev_thorough call (optional diagnostics). */
typedef struct {
  uint64_t pairs;
  uint64_t rounds;       /* outer pplacer rounds over all pairs                               */
  uint64_t newton_evals; /* derivative evaluations over all pairs                             */
  uint64_t reverts;      /* pairs that hit the "worse -> restore" exit, optimize.cpp:224-232  */
} epa_thorough_stats;

/* Number of GPUs visible (0 => every other entry point fails with EPA_ERR_NO_DEVICE). */
int epa_dev_device_count(void);

/*
 * Replaces: Tree::get_clv + make_tiny_partition + make_tiny_tree_structure for every branch
 * (src/tree/tiny_util.cpp:72-306) -- copies the reference data to HBM once, transformed into the
 * eigenbasis of the model -- and the per-branch part of Tiny_Tree::Tiny_Tree + Lookup_Store
 * (src/tree/Tiny_Tree.cpp:88-128, src/core/Lookup_Store.hpp:73-81): the per-branch per-site
 * lookup tables are built on device by epa_dev_build_lookup (called lazily by preplace).
 */
int epa_dev_create(const epa_ref_desc* desc, int device, epa_ctx** out);
void epa_dev_destroy(epa_ctx* ctx);
const char* epa_dev_last_error(const epa_ctx* ctx); /* ctx may be NULL: last create() error */

/* Same context, reference CLVs computed on the device from the tree (see epa_tree_desc). */
int epa_dev_create_from_tree(const epa_tree_desc* tree, int device, epa_ctx** out);

/* Log-likelihood of the reference tree evaluated at branch `branch` (Tree::ref_tree_logl,
 * src/tree/Tree.cpp:119-131; equal on every branch up to rounding). */
int epa_dev_tree_logl(epa_ctx* ctx, uint32_t branch, double* lnl);

/* Optional: run all kernels of `ctx` on this hipStream_t (passed as void*). Default stream 0. */
int epa_dev_set_stream(epa_ctx* ctx, void* hip_stream);

/*
 * Diagnostic switches of a context, no reference counterpart.  Every switch selects between code paths that return
 * the same results (cross-check kernels of the parity tests, A/B switches of the bench); the library reads NO
 * environment variable for its compute path (RCCL's location and timeout excepted: EPA_RCCL_LIB, EPA_COMM_TIMEOUT_S).
 *   "thorough_generic"  1: every Newton launch on the general kernel (k_thorough_generic, the reference-shaped loop)
 *   "preplace_generic"  1: preplacement on k_preplace (one gather per site) instead of the pair / site fast paths
 *   "select_full_rows"  1: candidate selection from whole table rows instead of the segment maxima
 *   "select_sort"       1: candidate list through staging rows + a stable sort instead of the [B][Q] bitmap
 *   "queued_thorough"   1: the Newton launch queued behind the selection, guarded by the device-side count
 *   "xcd_balance"       0: the eight XCDs keep equal shares of a Newton launch (default 1: epa_dev_xcd_shares)
 *   "aa_valu"           1: 20-state windows on the lane = site VALU kernel instead of the matrix-core kernel
 *   "timers"            0: no hipEvent records around the kernel families (epa_dev_last_kernel_ms returns < 0)
 * Unknown key: EPA_ERR_INVALID_ARG.
 */
int epa_dev_set_option(epa_ctx* ctx, const char* key, int value);

/* Builds T[b][site][col] for all branches (idempotent).  Replaces precompute_sites_static x C
 * + Lookup_Store::init_branch (Tiny_Tree.cpp:18-46,114-128). */
int epa_dev_build_lookup(epa_ctx* ctx);

/*
 * Query encoding (one byte per site, Q x W row-major): the lookup-column index of the
 * character, i.e. the position in NT_MAP / AA_MAP (src/util/maps.hpp:9-31; for DNA this is the
 * 4-bit code of the .bfast format, src/io/encoding.hpp) after the normalisation of
 * Lookup_Store's constructor (Lookup_Store.hpp:33-68).  epa_encode_queries() produces it from
 * ASCII.  win_begin/win_span = get_valid_range() per query (src/util/Range.hpp:34-49), or
 * (0, W) when premasking is off.
 */

/* ASCII -> column codes + windows on the host.  seqs: Q pointers to W characters each.
 * aa_x_as_n is accepted for ABI stability and ignored: quirk D4 lives in the lookup table of a
 * context created with epa_ref_desc.aa_x_as_n (preplacement only, as in the reference), the
 * code of 'X' is the same either way and the thorough placement always reads it as "any".
 * Returns EPA_OK, EPA_ERR_INVALID_CHAR or EPA_ERR_QUERY_ALL_GAP (first offender in *bad_query). */
int epa_encode_queries(uint32_t states, uint32_t sites, uint32_t Q, const char* const* seqs,
                       int premasking, int aa_x_as_n, uint8_t* codes, uint32_t* win_begin,
                       uint32_t* win_span, uint32_t* bad_query);

/*
 * Compact query layout (the wire format for real runs: a 150-column read of a 1500-column
 * alignment is 150 bytes instead of 1500).  Row q holds ONLY the window of query q:
 * codes[q*stride + i] = code of alignment column win_begin[q] + i, i < win_span[q] <= stride;
 * the rest of the row is padding.  Two passes: call with stride == 0 (codes may be NULL) to get
 * the windows, pick stride >= max(win_span) (a multiple of 16 keeps rows aligned), call again.
 * epa_dev_set_query_layout() tells a context which layout the q_codes of the following
 * preplace / thorough / place_chunk calls use: 0 (default) = Q x W rows as produced by
 * epa_encode_queries(), S > 0 = compact rows of S bytes.  Results are identical.
 */
/* Upper limit on the host threads the encoders use (0 = the hardware concurrency, at most 32). */
void epa_encode_set_threads(unsigned n);
int epa_encode_queries_compact(uint32_t states, uint32_t sites, uint32_t Q, const char* const* seqs,
                               int premasking, int aa_x_as_n, uint32_t stride, uint8_t* codes,
                               uint32_t* win_begin, uint32_t* win_span, uint32_t* bad_query);
int epa_dev_set_query_layout(epa_ctx* ctx, uint32_t code_stride);

/*
 * 4-bit wire format for nucleotide queries (the packing of the reference's FourBit,
 * src/io/encoding.hpp:18-27,81-97): two codes per byte, the earlier site in the high nibble, an
 * odd row padded with code 0 ('-').  The DNA column codes of epa_encode_queries*() are the 4-bit
 * state sets of NT_MAP (src/util/maps.hpp:9-26), so a row of `stride` codes packs into
 * (stride + 1) / 2 bytes.  epa_dev_set_query_packing(ctx, 4) tells a context that the q_codes of
 * the following calls are packed rows (row pitch (stride + 1) / 2, stride = the query layout's
 * row length); they are expanded on the device.  8 (default) = one byte per code.
 */
int epa_pack_codes_4bit(const uint8_t* codes, uint32_t Q, uint32_t stride, uint8_t* packed);
int epa_unpack_codes_4bit(const uint8_t* packed, uint32_t Q, uint32_t stride, uint8_t* codes);
int epa_dev_set_query_packing(epa_ctx* ctx, int bits);

/*
 * Replaces place() (src/core/place.cpp:41-95): lnl[q*B + b] = sum over the query's window of
 * T[b][site][code(q,site)]  (Lookup_Store::sum_precomputed_sitelk, Lookup_Store.hpp:110-141,
 * same summation order).  pendant = pendant_default and distal = branch_length/2 are implied.
 */
int epa_dev_preplace(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin,
                     const uint32_t* win_span, uint32_t Q, double* lnl);

/*
 * Replaces place_thorough() (src/core/place.cpp:97-171) = Tiny_Tree::place with opt_branches
 * (Tiny_Tree.cpp:159-204) -> call_focused(optimize_branch_triplet) -> opt_branch_lengths_pplacer
 * (src/core/pll/optimize.cpp:60-286).  One result per pair, in pair order.  `stats` may be NULL.
 */
int epa_dev_thorough(epa_ctx* ctx, const epa_pair* pairs, uint64_t n_pairs,
                     const uint8_t* q_codes, const uint32_t* win_begin, const uint32_t* win_span,
                     uint32_t Q, epa_result* out, epa_thorough_stats* stats);

/*
 * Replaces apply_heuristic() for the default dynamic heuristic (src/core/heuristics.hpp:119-127,
 * until_accumulated_reached src/set_manipulators.cpp:90-114) on device, so the Q x B table never
 * leaves HBM: per query, branches in descending LWR order until the accumulated LWR reaches
 * `threshold` (the crossing element included).  lnl is the Q x B table of epa_dev_preplace.
 * Writes at most max_pairs pairs, branch-major sorted (Work iteration order); *n_pairs receives
 * the number selected (if larger than max_pairs the call fails with EPA_ERR_INVALID_ARG).
 */
int epa_dev_select_candidates(epa_ctx* ctx, const double* lnl, uint32_t Q, double threshold,
                              epa_pair* pairs, uint64_t max_pairs, uint64_t* n_pairs);

/*
 * Selection rule used by epa_dev_select_candidates() and epa_dev_place_chunk() of this context
 * (apply_heuristic, src/core/heuristics.hpp:119-127):
 *   EPA_HEUR_DYNAMIC   (default) accumulated LWR >= the call's `threshold`   (--dyn-heur)
 *   EPA_HEUR_FIXED     the ceil(param * B) best branches per query           (--fix-heur,
 *                      until_top_percent src/set_manipulators.cpp:82-88); `threshold` is ignored
 *   EPA_HEUR_BASEBALL  every branch within 3.0 lnL of the best, plus min(40 - hits, 6) more (6 more
 *                      when over 40 were hit: the reference's size_t wrap-around, heuristics.hpp:107)
 *                      (--baseball-heur, heuristics.hpp:70-117, clamped at B)
 */
#define EPA_HEUR_DYNAMIC 0
#define EPA_HEUR_FIXED 1
#define EPA_HEUR_BASEBALL 2
int epa_dev_set_heuristic(epa_ctx* ctx, int mode, double param);

/*
 * The body of the reference's chunk loop for the default configuration (src/core/place.cpp:
 * 219-235): place() -> apply_heuristic() [dynamic, `threshold`] -> place_thorough(), fused so
 * that the Q x B preplacement table never leaves HBM and the host is only consulted once (the
 * candidate count sizes the thorough launch).  pairs / results: caller buffers of max_pairs
 * entries (host or device), filled branch-major; *n_pairs = number of candidate placements.
 * max_span: an upper bound of win_span[] if the caller knows it, 0 to let the library find it.
 */
int epa_dev_place_chunk(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin,
                        const uint32_t* win_span, uint32_t Q, uint32_t max_span, double threshold,
                        epa_pair* pairs, epa_result* results, uint64_t max_pairs,
                        uint64_t* n_pairs, epa_thorough_stats* stats);

/*
 * The same chunk body as a double-buffered pipeline, the device-side counterpart of the
 * reference's read-ahead of the next chunk (src/seq/MSA_Stream.cpp:79-82, the prefetch in
 * src/core/place.cpp:190-215): the query upload of chunk k+1 and the result download of chunk k-1
 * run on a copy stream while the kernels of chunk k run.  Slots 0 .. 23 (the loops below use two; a
 * caller of many small chunks keeps more in flight, see launch_begin), each walks
 *     stage -> launch -> finish -> stage -> ...
 *   stage   copies the caller's HOST arrays (layout / packing as set for the context) into the
 *           slot's pinned buffer and starts the H2D transfer; returns at once.  DEVICE arrays (all
 *           three) are read in place instead: no copy; they must be complete on the context's
 *           stream at launch and stay untouched until finish.
 *   launch  preplace -> heuristic -> thorough on the compute stream as soon as the upload has
 *           landed; blocks only until the candidate count is known (it sizes the thorough launch),
 *           then queues the thorough kernels and the D2H of pairs / results and returns.
 *           d_pairs / d_results: optional DEVICE buffers of max_pairs entries that receive the
 *           results (e.g. the send buffers of a collective); NULL = buffers owned by the slot.
 *           EPA_CHUNK_NO_D2H: results stay in HBM (finish() hands out the device pointers).
 *           EPA_CHUNK_HOST_ORDERED: the caller orders this chunk through the HOST only -- device arrays staged in
 *           place were complete when launch was called, and device-resident results are touched only after finish()
 *           returned.  Without it the library orders the slot's stream behind the context's stream at launch and
 *           the context's stream behind the chunk at the end of launch (stream-ordered consumers); those two event
 *           hops put ~20 us of idle queue between one chunk's Newton kernel and the next chunk's first kernel.
 *   finish  waits for the slot's download; *pairs / *results point into the slot's pinned host
 *           buffer (or HBM with EPA_CHUNK_NO_D2H), valid until the slot is staged again.
 *   launch_begin / launch_end: launch in two halves for callers that keep two chunks in flight --
 *           begin queues preplacement + candidate selection and returns WITHOUT waiting; end waits
 *           for the candidate count (normally long there by then) and queues the rest.  Each slot
 *           has its own stream and scratch, so chunk k + 1's begin-half overlaps chunk k's Newton
 *           kernel on the device:
 *               for k in 0..2: { stage(k, c[k]); begin(k); }
 *               for k: { end(k%5); if (k >= 2) finish((k-2)%5); stage((k+3)%5, c[k+3]); begin((k+3)%5); }
 *           (five slots: chunk k's Newton kernel is queued while chunk k-1's still runs and fills
 *           its tail wave by wave -- what small chunks need: 6.6 -> 8.3 M placements/s at 5000 reads
 *           per chunk against the two-slot order; bench.py's chunk5000 leg).  For large chunks two
 *           slots do: begin(k&1); finish((k-1)&1); stage((k+1)&1); end(k&1).
 * A typical loop:  stage(0, c0); for k: { launch(k&1); finish((k-1)&1); stage((k+1)&1, c[k+1]); }
 * Candidate overflow (EPA_ERR_PAIR_OVERFLOW from launch, as epa_dev_place_chunk) leaves the slot
 * staged: launch again with a larger max_pairs.
 */
#define EPA_CHUNK_NO_D2H 0x1u
#define EPA_CHUNK_HOST_ORDERED 0x2u
int epa_dev_chunk_stage(epa_ctx* ctx, int slot, const uint8_t* q_codes, const uint32_t* win_begin,
                        const uint32_t* win_span, uint32_t Q);
int epa_dev_chunk_launch(epa_ctx* ctx, int slot, uint32_t max_span, double threshold,
                         epa_pair* d_pairs, epa_result* d_results, uint64_t max_pairs,
                         uint32_t flags);
int epa_dev_chunk_launch_begin(epa_ctx* ctx, int slot, uint32_t max_span, double threshold,
                               epa_pair* d_pairs, epa_result* d_results, uint64_t max_pairs,
                               uint32_t flags);
int epa_dev_chunk_launch_end(epa_ctx* ctx, int slot);
int epa_dev_chunk_finish(epa_ctx* ctx, int slot, const epa_pair** pairs, const epa_result** results,
                         uint64_t* n_pairs, epa_thorough_stats* stats);

/*
 * Small chunks (the reference's default --chunk-size is 5000, src/util/Options.hpp:26; the chunk body of
 * src/core/place.cpp:207-246 then runs once per 5000 reads): every launch of the Newton kernel costs a fixed
 * ~0.11 ms, a preplacement over a handful of query groups another ~0.15 ms, and a chunk body is ~23 dispatches.
 * A GROUP LAUNCH runs ONE chunk body -- one preplacement, one selection, one Newton launch -- over the
 * concatenated queries of up to 8 STAGED slots and regroups the rows by chunk afterwards:
 *   launch_many_begin  slots[0] is the group's leader; every listed slot must be staged (same query layout).
 *                      Queues merge + preplacement + selection on the leader's stream, returns without waiting.
 *                      max_pairs bounds the candidates of the WHOLE group; results live in library buffers.
 *   epa_dev_chunk_launch_end(ctx, slots[0])   as for a single slot: waits for the candidate count, queues the
 *                      Newton kernel, the regrouping and the D2H (one copy per group)
 *   epa_dev_chunk_finish(ctx, slot) for EVERY member, in any order: that chunk's rows, bit-identical to a launch
 *                      of its own -- branch-major, queries ascending, sequence ids local to the chunk -- as a
 *                      range of the leader's buffers: valid until the LEADER is staged again, which is refused
 *                      while members are unfinished.  The Newton counters of the one launch are reported with
 *                      slots[0] (the other members report their pair count only).
 * EPA_ERR_PAIR_OVERFLOW and the window errors leave every member staged.  A loop for 5000-read chunks, groups
 * of four on twelve slots, two groups begun ahead (bench.py's chunk5000 leg):
 *     stage + many_begin(g0); stage + many_begin(g1);
 *     for g: { launch_end(leader(g)); finish(members of g-1); stage + many_begin(g+2); }
 */
int epa_dev_chunk_launch_many_begin(epa_ctx* ctx, const int* slots, int n_slots, uint32_t max_span,
                                    double threshold, uint64_t max_pairs, uint32_t flags);
int epa_dev_chunk_launch_many(epa_ctx* ctx, const int* slots, int n_slots, uint32_t max_span, double threshold,
                              uint64_t max_pairs, uint32_t flags);

/*
 * --no-heur (src/core/place.cpp:189,228: every branch gets a thorough placement) with the
 * post-processing of the chunk loop fused in: thorough optimisation of all B x Q pairs (pairs are
 * generated on the device), then per query compute_and_set_lwr over all B placements and filter()
 * (src/set_manipulators.cpp:43-69,131-204) on the device, so that only the surviving placements
 * cross PCIe.  acc_threshold == 0: discard_by_support_threshold (keep lwr > min_lwr, at least
 * filter_min, at most filter_max); != 0: discard_by_accumulated_threshold (the reference's
 * min - 1 top-up quirk included).  1 <= filter_min <= filter_max <= 64.
 * Outputs (host or device buffers): query q owns slots [q*filter_max, q*filter_max + counts[q]) of
 * pairs / results / lwr, best placement first.
 */
int epa_dev_place_all(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin,
                      const uint32_t* win_span, uint32_t Q, uint32_t max_span, double min_lwr,
                      int acc_threshold, uint32_t filter_min, uint32_t filter_max, epa_pair* pairs,
                      epa_result* results, double* lwr, uint32_t* counts,
                      epa_thorough_stats* stats);

/*
 * Multi-GPU: one process per GPU.  Queries are split into contiguous per-rank slices without any
 * exchange (src/net/epa_mpi_util.cpp:10-30, local_seq_package); the path's ONLY collective is the
 * gather of every chunk's (pair, result) rows to rank 0, which writes the jplace
 * (src/io/jplace_writer.hpp:117-129 gathers the ranks' text with MPI_Gatherv; here the 32-byte numeric
 * rows travel over RCCL: point-to-point xGMI links into the root, one group per chunk).
 *   epa_comm_get_unique_id  rank 0: 128 bytes to hand to the other ranks out of band (a file, an
 *                           environment variable, MPI_Bcast -- whatever launched the processes)
 *   epa_comm_create         collective (ncclCommInitRank).  rows_cap: rows per rank and gather (size it
 *                           from the expected candidates per chunk, e.g. 4 x reads); depth: send /
 *                           receive slots used round robin (2: gather k travels while chunk k + 1 runs)
 *   epa_dev_gather_results  collective, asynchronous, every rank once per chunk in the same order:
 *                           packs n rows (DEVICE pairs / results, complete on the context's stream;
 *                           sequence ids + seq_offset = global ids) and posts the fixed-size exchange on
 *                           the communicator's own stream.  Never blocks the host; rows beyond rows_cap
 *                           are carried into the rank's next gather.  *ticket names the gather.
 *   epa_dev_gather_slot     the same for a chunk slot launched with EPA_CHUNK_NO_D2H (after launch_end)
 *   epa_comm_collect        rank 0: waits for gather `ticket` (at most the communicator's timeout), copies its VALID rows to pinned host
 *                           memory and hands out one row block and count per rank; the blocks stay
 *                           valid until gather ticket + depth is posted.  pending[r] (optional): rows rank
 *                           r still carried after this gather -- 0 means every row it has posted so far
 *                           has arrived (a chunk's rows are branch-major: a query is complete only then)
 *   epa_comm_flush          collective, at the end, REPEATED until it reports *n_extra == 0: agrees (one
 *                           all-reduce) on whether any rank still carries rows and posts up to `depth` extra
 *                           gathers (tickets *first .. *first + *n_extra - 1) that drain them; rank 0
 *                           collects those before the next call
 *   epa_comm_probe          collective, optional, right after create: one round trip through everything a gather
 *                           uses (a one-row gather naming each rank and the PCI id of its device, then an all-reduce),
 *                           every wait bounded by timeout_s -- so that a transport that cannot move data between THESE
 *                           processes shows up in seconds, before any work depends on it, instead of as a gather that
 *                           never completes.  Rank 0 receives device_ids[world] = (pci domain << 16 | bus << 8 | device)
 *                           of every rank's GPU ("RCCL saw N ranks on N devices").  On failure: epa_comm_abort.
 *   epa_comm_set_timeout    seconds a host-side wait of this communicator may take (collect, flush, probe); comm ==
 *                           NULL sets the process default, which also bounds epa_comm_create (ncclCommInitRank waits
 *                           for all ranks).  Default: EPA_COMM_TIMEOUT_S, else 600.
 * RCCL is loaded at run time (dlopen): EPA_ERR_UNSUPPORTED without it.  Which library: the path given to
 * epa_comm_set_library() (before the first communicator); else the environment's EPA_RCCL_LIB; else a librccl ALREADY
 * MAPPED in the process (a host program that brought its own RCCL -- PyTorch ships torch/lib/librccl.so -- must not get a
 * second copy beside it); else librccl.so.1 / librccl.so on the loader's search path (LD_LIBRARY_PATH), /opt/rocm/lib.
 * epa_comm_library_path() names the file that was bound ("" if none): log it.
 */
#define EPA_COMM_ID_BYTES 128
typedef struct epa_comm epa_comm;
typedef struct {
  uint32_t branch_id, seq_id;   /* seq_id is GLOBAL (rank's offset added) */
  double lnl, pendant_length, distal_length;
} epa_row;
int epa_comm_set_library(const char* path);
const char* epa_comm_library_path(void);
int epa_comm_set_timeout(epa_comm* comm_or_null, double seconds);
int epa_comm_get_unique_id(void* id128);
int epa_comm_create(epa_ctx* ctx, const void* id128, int rank, int world, uint32_t rows_cap, int depth,
                    epa_comm** out);
void epa_comm_destroy(epa_comm* comm);
int epa_comm_probe(epa_ctx* ctx, epa_comm* comm, double timeout_s, uint64_t* device_ids);
/* test hook: rank 0's own rows travel through ncclSend / ncclRecv to itself instead of a local copy (before the
 * first gather only) */
int epa_comm_set_self_send(epa_comm* comm, int on);
int epa_dev_gather_results(epa_ctx* ctx, epa_comm* comm, const epa_pair* d_pairs, const epa_result* d_results,
                           uint64_t n, uint32_t seq_offset, uint64_t* ticket);
int epa_dev_gather_slot(epa_ctx* ctx, epa_comm* comm, int slot, uint32_t seq_offset, uint64_t* ticket);
/* --no-heur (src/core/place.cpp:219-231 with prescoring == false) in the one-process-per-GPU mode: epa_dev_place_all with
 * its filtered placements left in HBM as gather rows -- per kept placement, best first, its row followed by a row
 * {branch_id = EPA_ROW_LWR, seq_id, lnl = its like-weight ratio} (normalised over ALL branches on the device: it cannot
 * be recomputed from the kept rows) -- and epa_dev_gather_rows posts such rows (collective, asynchronous, carry-over:
 * exactly as epa_dev_gather_results).  *d_rows stays valid until the context's next place_all_rows call. */
#define EPA_ROW_LWR 0xfffffffeu
int epa_dev_place_all_rows(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin, const uint32_t* win_span,
                           uint32_t Q, uint32_t max_span, double min_lwr, int acc_threshold, uint32_t filter_min,
                           uint32_t filter_max, uint32_t seq_offset, const epa_row** d_rows, uint64_t* n_rows,
                           epa_thorough_stats* stats);
int epa_dev_gather_rows(epa_ctx* ctx, epa_comm* comm, const epa_row* d_rows, uint64_t n, uint64_t* ticket);
int epa_comm_collect(epa_comm* comm, uint64_t ticket, const epa_row** rows, uint32_t* counts, uint64_t* pending);
/* rows == NULL in epa_comm_collect: counts / pending only, the rows stay in HBM; this is rank r's block
 * of gather `ticket` there (valid until gather ticket + depth is posted; NULL if the slot was reused) */
const epa_row* epa_comm_device_rows(const epa_comm* comm, uint64_t ticket, int rank);
int epa_comm_flush(epa_ctx* ctx, epa_comm* comm, uint64_t* first_extra_ticket, uint32_t* n_extra);
uint64_t epa_comm_carried_rows(const epa_comm* comm);
/* a rank that fails mid-job: ncclCommAbort + release, without waiting for the peers (their pending
 * collect / flush calls then fail or time out after the communicator's timeout, default 600 s, instead of
 * blocking for ever; the reference's MPI build aborts the whole job, src/main.cpp's MPI_Abort path) */
void epa_comm_abort(epa_comm* comm);

/* free / total bytes of the context's device (hipMemGetInfo): the chunk loop sizes its device chunks
 * against it -- a chunk of Q queries keeps 2 pipeline slots x Q x pitch(B) x 8 bytes of preplacement
 * table live (`--chunk-size` is the user's memory knob, src/main.cpp:234-238). */
int epa_dev_mem_info(epa_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);

/* Diagnostics, no reference counterpart: the shares of a Newton launch's (branch-sorted) pair list that the device's
 * eight XCDs currently take.  0.125 each on a new context; every launch of one span class that ran for >= 1 ms reports
 * when each XCD ran out of pairs, and the next launches' shares move half way toward share / time (the XCDs of one
 * MI355X differ in speed by a few per cent: DESIGN.md 4.1).  Results do not depend on the shares. */
int epa_dev_xcd_shares(const epa_ctx* ctx, double shares[8]);

/* Diagnostics, no reference counterpart: the shader clock (MHz) the last single-class Newton launch really ran
 * at -- shader cycles (s_memtime) over 100 MHz ticks (s_memrealtime) of the launch's first wave, which lives as
 * long as the launch.  0 before the first such launch.  The bench line prices the kernel's fp64 rate against the
 * peak AT THIS CLOCK next to the spec-sheet peak at 2400 MHz (roofline.sclk_mhz, frac_at_measured_clock). */
double epa_dev_last_sclk_mhz(const epa_ctx* ctx);

/* duration in milliseconds of the last launch of the named kernel family on ctx's stream,
 * measured with HIP events ("preplace", "thorough", "lookup", "select"); < 0 if never run. */
double epa_dev_last_kernel_ms(const epa_ctx* ctx, const char* which);

#ifdef __cplusplus
}
#endif
#endif /* EPA_DEV_H */
