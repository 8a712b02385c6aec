// libepa_dev.so -- host side of the C-ABI (include/epa_dev.h) plus the reference-data setup
// kernels.  MI355X / gfx950 only; there is deliberately no CPU fallback anywhere in this file:
// with no device every entry point fails with EPA_ERR_NO_DEVICE.
#include "epa_dev_internal.hpp"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <atomic>
#include <thread>

static thread_local std::string g_create_err;

int epa_fail(epa_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg; else g_create_err = msg;
  return code;
}

namespace {
__global__ void __launch_bounds__(256) k_zero(uint4* __restrict__ a, size_t na, uint4* __restrict__ b, size_t nb) {
  const size_t stride = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = i0; i < na; i += stride) a[i] = z;
  for (size_t i = i0; i < nb; i += stride) b[i] = z;
}
}  // namespace

int epa_zero_async(epa_ctx* ctx, void* p, size_t bytes, void* p2, size_t bytes2) {
  if (!p) bytes = 0;
  if (!p2) bytes2 = 0;
  if (bytes + bytes2 == 0) return EPA_OK;
  if (((bytes | bytes2) & 15u) || (((uintptr_t)p | (uintptr_t)p2) & 15u)) {
    if (bytes) EPA_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
    if (bytes2) EPA_HIP(ctx, hipMemsetAsync(p2, 0, bytes2, ctx->stream));
    return EPA_OK;
  }
  const size_t na = bytes / 16, nb = bytes2 / 16;
  const uint32_t grid = (uint32_t)std::min<size_t>((std::max(na, nb) + 255) / 256, (size_t)ctx->n_cu * 8);
  hipLaunchKernelGGL(k_zero, dim3(grid), dim3(256), 0, ctx->stream, (uint4*)p, na, (uint4*)p2, nb);
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}

void* epa_scratch(epa_ctx* ctx, int slot, size_t bytes) {
  slot += ctx->bank * epa_ctx::N_SCRATCH;
  if (bytes <= ctx->scratch_sz[slot]) return ctx->scratch[slot];
  if (ctx->scratch[slot]) (void)hipFree(ctx->scratch[slot]);
  ctx->scratch[slot] = nullptr;
  ctx->scratch_sz[slot] = 0;
  size_t want = bytes + bytes / 4 + 256;
  if (hipMalloc(&ctx->scratch[slot], want) != hipSuccess) return nullptr;
  ctx->scratch_sz[slot] = want;
  return ctx->scratch[slot];
}

bool epa_is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

const void* epa_to_device(epa_ctx* ctx, int slot, const void* p, size_t bytes) {
  if (epa_is_device_ptr(p)) return p;
  void* d = epa_scratch(ctx, slot, bytes + 1024);  // slack: kernels may over-read a few words
  if (!d) return nullptr;
  if (hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
  return d;
}

// 4-bit wire format (the packing of the reference's FourBit, src/io/encoding.hpp:18-27,81-97: two
// codes per byte, the earlier site in the high nibble, an odd row padded with code 0).  The DNA
// column codes ARE the 4-bit state sets, so a row of `stride` codes travels as (stride + 1) / 2
// bytes and is expanded on the device into the one-byte layout every kernel reads.
__global__ void __launch_bounds__(256) k_unpack4(const uint8_t* __restrict__ packed, uint8_t* __restrict__ codes,
                                                 uint32_t Q, uint32_t stride, uint32_t pstride) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // one packed byte each
  if (i >= (uint64_t)Q * pstride) return;
  const uint32_t q = (uint32_t)(i / pstride), p = (uint32_t)(i - (uint64_t)q * pstride);
  const uint8_t b = packed[i];
  uint8_t* row = codes + (size_t)q * stride;
  row[2 * p] = b >> 4;
  if (2 * p + 1 < stride) row[2 * p + 1] = b & 15u;
}

// Even row length: row q of the packed stream starts at byte q * stride / 2, the expansion is one
// flat stream -- 4 packed bytes in, 8 codes out per thread, both coalesced (the byte-per-thread form
// took 160 us for a 100k x 150 chunk, this one the time of moving 22 MB).
__global__ void __launch_bounds__(256) k_unpack4_flat(const uint32_t* __restrict__ packed, uint2* __restrict__ codes,
                                                      uint64_t n_bytes) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (4 * i >= n_bytes) return;
  if (4 * i + 4 > n_bytes) {   // the last, partial word byte by byte: nothing beyond the arrays is touched
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(packed);
    uint8_t* cb = reinterpret_cast<uint8_t*>(codes);
    for (uint64_t k = 4 * i; k < n_bytes; ++k) { cb[2 * k] = pb[k] >> 4; cb[2 * k + 1] = pb[k] & 15u; }
    return;
  }
  const uint32_t w = packed[i];
  auto two = [](uint32_t b) { return (b >> 4) | ((b & 15u) << 8); };   // earlier site = high nibble
  uint2 o;
  o.x = two(w & 0xffu) | (two((w >> 8) & 0xffu) << 16);
  o.y = two((w >> 16) & 0xffu) | (two(w >> 24) << 16);
  codes[i] = o;
}
static void launch_unpack4(hipStream_t st, const uint8_t* d_packed, uint8_t* d_codes, uint32_t Q, uint32_t stride) {
  const size_t pstride = ((size_t)stride + 1) / 2;
  const uint64_t n = (uint64_t)Q * pstride;
  if ((stride & 1u) == 0 && ((uintptr_t)d_packed & 3u) == 0 && ((uintptr_t)d_codes & 7u) == 0) {
    const uint64_t nw = (n + 3) / 4;
    hipLaunchKernelGGL(k_unpack4_flat, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, st,
                       (const uint32_t*)d_packed, (uint2*)d_codes, n);
  } else {
    hipLaunchKernelGGL(k_unpack4, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, d_packed, d_codes, Q,
                       stride, (uint32_t)pstride);
  }
}

const uint8_t* epa_codes_to_device(epa_ctx* ctx, const uint8_t* q_codes, uint32_t Q) {
  const size_t stride = ctx->code_stride ? ctx->code_stride : ctx->W;
  if (!ctx->code_packed4) return (const uint8_t*)epa_to_device(ctx, 0, q_codes, (size_t)Q * stride);
  const size_t pstride = (stride + 1) / 2;
  const uint8_t* d_packed = (const uint8_t*)epa_to_device(ctx, 10, q_codes, (size_t)Q * pstride);
  uint8_t* d = (uint8_t*)epa_scratch(ctx, 0, (size_t)Q * stride + 1024);
  if (!d_packed || !d) return nullptr;
  launch_unpack4(ctx->stream, d_packed, d, Q, (uint32_t)stride);
  return d;
}

// The eight XCDs of one device do not run the Newton kernel equally fast (a few per cent, stable from launch to
// launch, different from box to box): a launch that stamped its XCDs' drain times (ThArgs::xstamp) moves the shares of
// the next ones half way toward share_x ~ pairs_x / time_x.  Launches shorter than 1 ms (start-up and the last
// pairs dominate) and implausible stamps are ignored; a share stays within 0.85 .. 1.15 of an eighth.  Option xcd_balance = 0: off.
void epa_xcd_feedback(epa_ctx* ctx, uint64_t n_pairs, const unsigned long long* hst) {
  const bool off = !ctx->opt.xcd_balance;
  // the shader clock of the launch, whatever the shares do: cycles / (10 ns ticks) of workgroup 0's first wave
  if (hst[6] > 1000) ctx->last_sclk_mhz = 100.0 * (double)hst[5] / (double)hst[6];
  if (off || hst[7] == 0) return;
  double sp[8], tot = 0.0;
  for (int x = 0; x < 8; ++x) {
    const unsigned long long tx = hst[8 + x] >> 21;
    if (tx <= hst[7]) return;
    const double t = (double)(tx - hst[7]);                        // 10 ns ticks
    // the share the XCD had IN THE MEASURED LAUNCH (stamped by the kernel), not the context's current one: with
    // several slots in flight a later launch may already have been issued with updated shares
    const double share = (double)(hst[8 + x] & 0x1fffffu) / (double)(1u << 20);
    if (share <= 0.0) return;
    if (t < 1e5 || t > 1e9) return;                                // < 1 ms (start-up and the last pairs dominate) or > 10 s
    sp[x] = share / t;
    tot += sp[x];
  }
  double norm = 0.0;
  for (int x = 0; x < 8; ++x) {
    double w = 0.5 * ctx->xcd_w[x] + 0.5 * sp[x] / tot;
    w = std::min(0.125 * 1.15, std::max(0.125 * 0.85, w));
    ctx->xcd_w[x] = w;
    norm += w;
  }
  double c = 0.0;
  for (int x = 0; x < 8; ++x) {
    ctx->xcd_w[x] /= norm;
    ctx->xcd_cum[x] = (uint32_t)std::llround(c * (double)(1u << 20));
    c += ctx->xcd_w[x];
  }
  ctx->xcd_cum[8] = 1u << 20;
}

void epa_timer_start(epa_ctx* ctx, EvTimer& t) {
  if (!ctx->opt.timers) return;   // (option timers = 0: what the event records cost)
  if (!t.a) { (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b); }
  (void)hipEventRecord(t.a, ctx->stream);
}
void epa_timer_stop(epa_ctx* ctx, EvTimer& t) {
  if (!ctx->opt.timers) return;
  (void)hipEventRecord(t.b, ctx->stream);
  t.valid = true;
}

// =============================================================================================
// Setup kernel 1: eigen-transform one reference CLV (or tip row) into component-major layout.
//   dst[(k*s + x) * W + site] = sum_i Ui[x][i] * clv[site][k][i]
// Replaces the borrow of Tree::get_clv() pointers in make_tiny_partition
// (src/tree/tiny_util.cpp:165-181); the transform is exact-arithmetic-equivalent to what
// pll_update_sumtable / pll_update_prob_matrices apply per use (U^-1 is folded in once).
// Coalescing: reads are [site][c*s] rows (one 128 B / 640 B row per thread, consecutive threads
// consecutive rows); writes are lane-consecutive per component.
// =============================================================================================
__global__ void __launch_bounds__(256) k_transform(const ModelDev* __restrict__ m,
                                                   const double* __restrict__ clv,
                                                   const uint8_t* __restrict__ tip,
                                                   const uint32_t* __restrict__ tipmap,
                                                   uint32_t W, int c_in, double* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* Ui = reinterpret_cast<double*>(smem);
  const int s = m->s, c = m->c;
  for (int i = threadIdx.x; i < s * s; i += blockDim.x) Ui[i] = m->Ui[i];
  __syncthreads();
  const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= W) return;
  uint32_t mask = 0;
  if (tip) mask = tipmap[tip[site]];
  for (int k = 0; k < c; ++k) {
    for (int x = 0; x < s; ++x) {
      double acc = 0.0;
      if (tip) {
        for (int i = 0; i < s; ++i)
          if ((mask >> i) & 1u) acc += Ui[x * s + i];
      } else {
        const int ksrc = c_in <= 2 ? k % c_in : (k < c_in ? k : c_in - 1);       // c_in < c: replicated / padded categories
        const double* v = clv + ((size_t)site * c_in + ksrc) * s;
        for (int i = 0; i < s; ++i) acc = fma(Ui[x * s + i], v[i], acc);
      }
      dst[(size_t)(k * s + x) * W + site] = acc;
    }
  }
}

int launch_transform(epa_ctx* ctx, const double* d_clv, const uint8_t* d_tip,
                     const uint32_t* d_tipmap, uint32_t, double* dst) {
  dim3 grid((ctx->W + 255) / 256);
  size_t lds = sizeof(double) * ctx->s * ctx->s;
  hipLaunchKernelGGL(k_transform, grid, dim3(256), lds, ctx->stream, ctx->dmodel, d_clv, d_tip,
                     d_tipmap, ctx->W, ctx->c_in, dst);
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}

// =============================================================================================
// Setup kernel 2: per-branch per-site lookup table (K1 + K2 of SURVEY.md section 7.1).
// Replaces, for every branch at once, the Tiny_Tree constructor's
//   pll_update_prob_matrices(3) + pll_update_partials(inner <- distal (x) proximal)
//   (src/tree/Tiny_Tree.cpp:88-112) and precompute_sites_static() x |char map| +
//   Lookup_Store::init_branch (:18-46,114-128; src/core/Lookup_Store.hpp:73-81).
// One thread per (branch, site); the inner CLV never leaves registers.
// HBM-bound: reads 2*c*s*8 B + 4 B, writes ncols*8 B per site.
// =============================================================================================
template <int S>
__global__ void __launch_bounds__(256) k_build_lookup(const ModelDev* __restrict__ m,
                                                      const double* __restrict__ refT,
                                                      const uint32_t* __restrict__ scSum,
                                                      const double* __restrict__ blen,
                                                      double pendant, uint32_t W,
                                                      double* __restrict__ lookup,
                                                      double* __restrict__ refI,
                                                      uint8_t* __restrict__ resc0,
                                                      const double* __restrict__ cinv) {
  __shared__ double U[S * S], Ui[S * S];
  __shared__ double Eh[EPA_MAX_CATS * S], Ep[EPA_MAX_CATS * S];  // exp tables: half branch, pendant
  const int c = m->c, ncols = m->ncols;
  const uint32_t b = blockIdx.y;
  const double half = blen[b] * 0.5;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) { U[i] = m->U[i]; Ui[i] = m->Ui[i]; }
  for (int i = threadIdx.x; i < c * S; i += blockDim.x) {
    const int k = i / S, x = i % S;
    Eh[i] = exp(m->lam[x] * m->rate[k] * half);     // proximal == distal == orig/2
    Ep[i] = exp(m->lam[x] * m->rate[k] * pendant);  // reset_triplet_lengths, pll_util.cpp:354-374
  }
  __syncthreads();
  const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= W) return;
  const double* Xt = refT + (size_t)(2 * b) * c * S * W + site;
  const double* Dt = refT + (size_t)(2 * b + 1) * c * S * W + site;
  double I[EPA_MAX_CATS][S];
  double mx = 0.0;
  for (int k = 0; k < c; ++k) {
    double dv[S], xv[S];
#pragma unroll
    for (int x = 0; x < S; ++x) {
      dv[x] = Dt[(size_t)(k * S + x) * W] * Eh[k * S + x];
      xv[x] = Xt[(size_t)(k * S + x) * W] * Eh[k * S + x];
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
      double a = 0.0, bb = 0.0;
#pragma unroll
      for (int x = 0; x < S; ++x) {
        a = fma(U[i * S + x], dv[x], a);
        bb = fma(U[i * S + x], xv[x], bb);
      }
      const double v = a * bb;
      I[k][i] = v;
      mx = fmax(mx, v);
    }
  }
  uint32_t sc = scSum[(size_t)b * W + site];
  // per-site scaling of pll_update_partials: every entry below 2^-256 -> multiply by 2^256
  const bool resc = mx < 0x1p-256;
  if (resc) sc += 1;
  if (resc0) resc0[(size_t)b * W + site] = resc ? 1 : 0;
  const double mult = resc ? 0x1p+256 : 1.0;
  // g[k][i] = pi_i * (P_pendant I)_i  via the eigenbasis: P I = U (e o (Ui I))
  double g[EPA_MAX_CATS][S];
  for (int k = 0; k < c; ++k) {
    double it[S];
#pragma unroll
    for (int x = 0; x < S; ++x) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < S; ++i) acc = fma(Ui[x * S + i], I[k][i] * mult, acc);
      // the thorough kernel starts every pair from exactly this vector (same lengths): keep it
      if (refI) refI[((size_t)b * c * S + (size_t)(k * S + x)) * W + site] = acc;
      it[x] = acc * Ep[k * S + x];
    }
#pragma unroll
    for (int i = 0; i < S; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int x = 0; x < S; ++x) acc = fma(U[i * S + x], it[x], acc);
      g[k][i] = m->pi[i] * acc;
    }
  }
  const double log_thr = -256.0 * 0.6931471805599453094;  // log(2^-256)
  double* out = lookup + ((size_t)b * W + site) * ncols;
  for (int col = 0; col < ncols; ++col) {
    const uint32_t mask = m->colmask[col];
    double terma = 0.0;
    for (int k = 0; k < c; ++k) {
      double tr = 0.0;
#pragma unroll
      for (int i = 0; i < S; ++i)
        if ((mask >> i) & 1u) tr += g[k][i];
      terma += tr * m->w[k];
    }
    if (cinv) terma += cinv[site];  // +I: p * pi_inv, unscaled and independent of the query char
    double v = log(terma);
    if (sc) v += sc * log_thr;
    out[col] = v;
  }
}

int launch_build_lookup(epa_ctx* ctx) {
  dim3 grid((ctx->W + 255) / 256, ctx->B);
  epa_timer_start(ctx, ctx->t_lookup);
  if (ctx->s == 4)
    hipLaunchKernelGGL(k_build_lookup<4>, grid, dim3(256), 0, ctx->stream, ctx->dmodel, ctx->refT,
                       ctx->scSum, ctx->blen, ctx->blo.pendant_default, ctx->W, ctx->lookup, ctx->refI,
                       ctx->resc0, ctx->cinv);
  else
    hipLaunchKernelGGL(k_build_lookup<20>, grid, dim3(256), 0, ctx->stream, ctx->dmodel, ctx->refT,
                       ctx->scSum, ctx->blen, ctx->blo.pendant_default, ctx->W, ctx->lookup, ctx->refI,
                       ctx->resc0, ctx->cinv);
  EPA_HIP(ctx, hipGetLastError());
  if (ctx->s == 4) {
    int rc = launch_build_lookup2(ctx);
    if (rc != EPA_OK) return rc;
  }
  epa_timer_stop(ctx, ctx->t_lookup);
  return EPA_OK;
}

// src: the caller's scaler row, [W] or (per-rate scalers) libpll's [W][cdim]; dst: [cdim][W]
__global__ void k_add_scaler(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                             uint32_t W, uint32_t cdim) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // i = k * W + site
  if (i >= W * cdim) return;
  const uint32_t k = i / W, w = i - k * W;
  dst[i] += src[(size_t)w * cdim + k];
}

// =============================================================================================
// character maps (src/util/maps.hpp:9-31, src/core/Lookup_Store.hpp:33-68, pll_map_nt/_aa)
// =============================================================================================
static const char NT_COLS[16] = {'-', 'T', 'G', 'K', 'C', 'Y', 'S', 'B',
                                 'A', 'W', 'R', 'D', 'M', 'H', 'V', 'N'};
static const char AA_COLS[24] = {'A', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'K', 'L', 'M', 'N',
                                 'P', 'Q', 'R', 'S', 'T', 'V', 'W', 'Y', '-', 'X', 'B', 'Z'};
static const char AA_STATE_ORDER[21] = "ARNDCQEGHILKMFPSTWYV";

static uint32_t column_mask(int s, int col) {
  if (s == 4) {  // column index bits: A=8 C=4 G=2 T=1 -> state bits A=1 C=2 G=4 T=8
    if (col == 0) return 15u;  // '-' == any
    uint32_t m = 0;
    if (col & 8) m |= 1u;
    if (col & 4) m |= 2u;
    if (col & 2) m |= 4u;
    if (col & 1) m |= 8u;
    return m;
  }
  const char ch = AA_COLS[col];
  const char* p = strchr(AA_STATE_ORDER, ch);
  if (p) return 1u << (p - AA_STATE_ORDER);
  if (ch == 'B') return (1u << 2) | (1u << 3);
  if (ch == 'Z') return (1u << 5) | (1u << 6);
  return (1u << 20) - 1;  // '-' and 'X'
}

// upper limit on the host threads of the query encoder (0 = hardware concurrency, at most 32)
static std::atomic<unsigned> g_encode_threads{0};
extern "C" void epa_encode_set_threads(unsigned n) { g_encode_threads.store(n, std::memory_order_relaxed); }

static int encode_impl(uint32_t states, uint32_t sites, uint32_t Q, const char* const* seqs,
                       int premasking, int aa_x_as_n, bool compact, uint32_t stride, uint8_t* codes,
                       uint32_t* win_begin, uint32_t* win_span, uint32_t* bad_query) {
  int8_t map[256];
  memset(map, -1, sizeof(map));
  const bool dna = states == 4;
  const char* cols = dna ? NT_COLS : AA_COLS;
  const int n = dna ? 16 : 24;
  for (int i = 0; i < n; ++i) {
    map[(unsigned char)cols[i]] = (int8_t)i;
    map[(unsigned char)tolower(cols[i])] = (int8_t)i;
  }
  if (dna) {
    map['U'] = map['u'] = map['T'];
    map['X'] = map['x'] = map['O'] = map['o'] = map['.'] = map['-'];
  }
  // aa_x_as_n (quirk D4) is NOT applied here: the reference remaps 'X' to the 'N' column only in
  // Lookup_Store's index map (preplacement), the thorough placement reads 'X' as "any" through
  // pll_map_aa.  A context created with epa_ref_desc.aa_x_as_n builds its 'X' lookup column as a
  // copy of the 'N' column instead, so the same code rows serve both steps.
  (void)aa_x_as_n;
  map['?'] = map['-'];
  // queries are independent: a few host threads (the first offender in query order is reported);
  // epa_encode_set_threads() caps them (the CLI's -T, a container's CPU quota)
  const unsigned cap = g_encode_threads.load(std::memory_order_relaxed);
  const unsigned hw = cap ? cap : std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = (unsigned)std::min<uint64_t>(std::min(hw, 32u), std::max<uint64_t>(1, (uint64_t)Q * sites >> 20));
  std::vector<uint32_t> bad(nt, 0xffffffffu);
  std::vector<int> code(nt, EPA_OK);
  auto work = [&](unsigned t) {
    const uint32_t q0 = (uint32_t)((uint64_t)Q * t / nt), q1 = (uint32_t)((uint64_t)Q * (t + 1) / nt);
    for (uint32_t q = q0; q < q1; ++q) {
      const char* sq = seqs[q];
      uint32_t lo = 0, hi = sites;
      if (premasking) {  // get_valid_range, src/util/Range.hpp:34-49: only the literal '-'
        // a short read of a wide alignment is ~90 % leading / trailing gaps: eight characters per compare
        // (1 M reads x 1500 columns are 1.5 GB of '-' that the byte loop walked twice per chunk)
        constexpr uint64_t GAPS = 0x2d2d2d2d2d2d2d2dull;
        while (lo + 8 <= hi) { uint64_t w8; memcpy(&w8, sq + lo, 8); if (w8 != GAPS) break; lo += 8; }
        while (lo < hi && sq[lo] == '-') ++lo;
        while (hi >= lo + 8) { uint64_t w8; memcpy(&w8, sq + hi - 8, 8); if (w8 != GAPS) break; hi -= 8; }
        while (hi > lo && sq[hi - 1] == '-') --hi;
      }
      // every character of the row is validated, as the reference does (Lookup_Store.hpp:100-108); outside the
      // window all of them are '-' (just compared), which is valid
      for (uint32_t w = lo; w < hi; ++w)
        if (map[(unsigned char)sq[w]] < 0) { bad[t] = q; code[t] = EPA_ERR_INVALID_CHAR; return; }
      if (hi == lo) { bad[t] = q; code[t] = EPA_ERR_QUERY_ALL_GAP; return; }
      win_begin[q] = lo;
      win_span[q] = hi - lo;
      if (!compact) {
        uint8_t* out = codes + (size_t)q * sites;
        for (uint32_t w = 0; w < sites; ++w) out[w] = (uint8_t)map[(unsigned char)sq[w]];
      } else if (stride) {
        if (hi - lo > stride) { bad[t] = q; code[t] = EPA_ERR_INVALID_ARG; return; }
        uint8_t* out = codes + (size_t)q * stride;
        for (uint32_t w = lo; w < hi; ++w) out[w - lo] = (uint8_t)map[(unsigned char)sq[w]];
        memset(out + (hi - lo), 0, stride - (hi - lo));
      }
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < nt; ++t)
    if (code[t] != EPA_OK) { if (bad_query) *bad_query = bad[t]; return code[t]; }
  return EPA_OK;
}

extern "C" int epa_encode_queries(uint32_t states, uint32_t sites, uint32_t Q,
                                  const char* const* seqs, int premasking, int aa_x_as_n,
                                  uint8_t* codes, uint32_t* win_begin, uint32_t* win_span,
                                  uint32_t* bad_query) {
  return encode_impl(states, sites, Q, seqs, premasking, aa_x_as_n, false, 0, codes, win_begin, win_span,
                     bad_query);
}

extern "C" int epa_encode_queries_compact(uint32_t states, uint32_t sites, uint32_t Q,
                                          const char* const* seqs, int premasking, int aa_x_as_n,
                                          uint32_t stride, uint8_t* codes, uint32_t* win_begin,
                                          uint32_t* win_span, uint32_t* bad_query) {
  if (stride && !codes) return EPA_ERR_INVALID_ARG;
  return encode_impl(states, sites, Q, seqs, premasking, aa_x_as_n, true, stride, codes, win_begin,
                     win_span, bad_query);
}

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" int epa_dev_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" const char* epa_dev_last_error(const epa_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

extern "C" int epa_dev_set_query_layout(epa_ctx* ctx, uint32_t code_stride) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  ctx->code_stride = code_stride;
  return EPA_OK;
}

extern "C" int epa_dev_set_option(epa_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return EPA_ERR_INVALID_ARG;
  struct Entry { const char* name; int EpaOptions::*field; };
  static const Entry table[] = {
      {"thorough_generic", &EpaOptions::thorough_generic}, {"preplace_generic", &EpaOptions::preplace_generic},
      {"select_full_rows", &EpaOptions::select_full_rows}, {"select_sort", &EpaOptions::select_sort},
      {"queued_thorough", &EpaOptions::queued_thorough},   {"xcd_balance", &EpaOptions::xcd_balance},
      {"aa_valu", &EpaOptions::aa_valu},                   {"timers", &EpaOptions::timers}};
  for (const Entry& e : table)
    if (strcmp(key, e.name) == 0) {
      ctx->opt.*e.field = value;
      // a context whose shape only the general kernel serves stays there whatever the switch says
      ctx->generic_thorough = ctx->generic_native || ctx->opt.thorough_generic != 0;
      return EPA_OK;
    }
  return epa_fail(ctx, EPA_ERR_INVALID_ARG, std::string("set_option: unknown key '") + key + "'");
}

extern "C" int epa_dev_set_heuristic(epa_ctx* ctx, int mode, double param) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  if (mode < EPA_HEUR_DYNAMIC || mode > EPA_HEUR_BASEBALL)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "set_heuristic: unknown mode");
  if (mode == EPA_HEUR_FIXED && !(param >= 0.0 && param <= 1.0))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "set_heuristic: the fraction must lie in [0, 1]");
  ctx->heur_mode = mode;
  ctx->heur_param = param;
  return EPA_OK;
}

extern "C" int epa_dev_set_query_packing(epa_ctx* ctx, int bits) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  if (bits != 8 && bits != 4) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "set_query_packing: 8 or 4 bits per code");
  if (bits == 4 && ctx->s != 4)
    return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "set_query_packing: the 4-bit format holds nucleotide codes only");
  ctx->code_packed4 = bits == 4;
  return EPA_OK;
}

extern "C" int epa_pack_codes_4bit(const uint8_t* codes, uint32_t Q, uint32_t stride, uint8_t* packed) {
  if (!codes || !packed) return EPA_ERR_INVALID_ARG;
  const size_t ps = ((size_t)stride + 1) / 2;
  // rows are independent: the encoder's host threads (an 80 MB pass per million 150 bp reads)
  const unsigned cap = g_encode_threads.load(std::memory_order_relaxed);
  const unsigned hw = cap ? cap : std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = (unsigned)std::min<uint64_t>(std::min(hw, 32u), std::max<uint64_t>(1, (uint64_t)Q * stride >> 20));
  std::vector<int> bad(nt, 0);
  auto work = [&](unsigned t) {
    const uint32_t q0 = (uint32_t)((uint64_t)Q * t / nt), q1 = (uint32_t)((uint64_t)Q * (t + 1) / nt);
    unsigned over = 0;
    for (uint32_t q = q0; q < q1; ++q) {
      const uint8_t* r = codes + (size_t)q * stride;
      uint8_t* o = packed + (size_t)q * ps;
      for (uint32_t p = 0; p < stride / 2; ++p) {
        const uint8_t hi = r[2 * p], lo = r[2 * p + 1];
        over |= hi | lo;
        o[p] = (uint8_t)((hi << 4) | lo);
      }
      if (stride & 1) { const uint8_t hi = r[stride - 1]; over |= hi; o[ps - 1] = (uint8_t)(hi << 4); }
    }
    bad[t] = over > 15;
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (unsigned t = 0; t < nt; ++t) if (bad[t]) return EPA_ERR_INVALID_ARG;
  return EPA_OK;
}

extern "C" int epa_unpack_codes_4bit(const uint8_t* packed, uint32_t Q, uint32_t stride, uint8_t* codes) {
  if (!codes || !packed) return EPA_ERR_INVALID_ARG;
  const size_t ps = ((size_t)stride + 1) / 2;
  for (uint32_t q = 0; q < Q; ++q)
    for (uint32_t i = 0; i < stride; ++i) {
      const uint8_t b = packed[(size_t)q * ps + i / 2];
      codes[(size_t)q * stride + i] = (i & 1) ? (b & 15u) : (b >> 4);
    }
  return EPA_OK;
}

extern "C" int epa_dev_set_stream(epa_ctx* ctx, void* s) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  ctx->stream = (hipStream_t)s;
  return EPA_OK;
}

extern "C" void epa_dev_destroy(epa_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < epa_ctx::N_BANKS * epa_ctx::N_SCRATCH; ++i) if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
  if (ctx->refT) (void)hipFree(ctx->refT);
  if (ctx->scSum) (void)hipFree(ctx->scSum);
  if (ctx->blen) (void)hipFree(ctx->blen);
  if (ctx->lookup) (void)hipFree(ctx->lookup);
  if (ctx->lookup2) (void)hipFree(ctx->lookup2);
  if (ctx->th_ctr) (void)hipFree(ctx->th_ctr);
  for (hipEvent_t e : ctx->ev_rb) if (e) (void)hipEventDestroy(e);
  if (ctx->refI) (void)hipFree(ctx->refI);
  if (ctx->cinv) (void)hipFree(ctx->cinv);
  if (ctx->resc0) (void)hipFree(ctx->resc0);
  if (ctx->dmodel) (void)hipFree(ctx->dmodel);
  for (ChunkSlot& sl : ctx->slots) {
    if (sl.h_in) (void)hipHostFree(sl.h_in);
    if (sl.h_out) (void)hipHostFree(sl.h_out);
    if (sl.h_stats) (void)hipHostFree(sl.h_stats);
    if (sl.h_sel) (void)hipHostFree(sl.h_sel);
    if (sl.d_in) (void)hipFree(sl.d_in);
    if (sl.d_unpacked) (void)hipFree(sl.d_unpacked);
    if (sl.d_pairs) (void)hipFree(sl.d_pairs);
    if (sl.d_res) (void)hipFree(sl.d_res);
    if (sl.d_stats) (void)hipFree(sl.d_stats);
    if (sl.d_merge) (void)hipFree(sl.d_merge);
    if (sl.d_gpairs) (void)hipFree(sl.d_gpairs);
    if (sl.d_gres) (void)hipFree(sl.d_gres);
    if (sl.d_goff) (void)hipFree(sl.d_goff);
    if (sl.h_goff) (void)hipHostFree(sl.h_goff);
    for (hipEvent_t e : {sl.ev_up, sl.ev_done, sl.ev_down, sl.ev_base}) if (e) (void)hipEventDestroy(e);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
  std::vector<EvTimer*> ts = {&ctx->t_lookup};
  for (auto& bk : ctx->t_bank) for (auto& t : bk) ts.push_back(&t);
  for (auto* t : ts) { if (t->a) (void)hipEventDestroy(t->a); if (t->b) (void)hipEventDestroy(t->b); }
  delete ctx;
}

static int precompute_from_tree(epa_ctx* ctx, const epa_tree_desc* t, const uint32_t* d_tipmap);
__global__ void k_align_rates(const ModelDev* __restrict__ m, double* __restrict__ refT,
                              const uint32_t* __restrict__ sc_side, uint32_t* __restrict__ scSum, uint32_t W);

static int create_impl(const epa_ref_desc* d, int device, epa_ctx* ctx, const epa_tree_desc* tree = nullptr) {
  const int s = (int)d->states, c_in = (int)d->rate_cats;
  if (!(s == 4 || s == 20)) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "states must be 4 or 20");
  if (c_in < 1 || c_in > EPA_MAX_CATS) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "rate_cats out of range");
  // The tuned thorough kernels work on groups of 4 rate categories.  1 or 2 categories (no +G, +G2)
  // are replicated to 4 with the weights divided accordingly: sum_k w_k L_k is unchanged.  Nucleotide
  // models with 3 or 5 .. 16 categories (+G8, +R5, ...) are padded to the next multiple of 4 with copies
  // of the last category at weight 0: every weighted sum gains exact zeros, the per-site rescale
  // test (all entries < 2^-256) sees values it has already seen -- results are those of the
  // unpadded model, and k_thorough_dna serves them with one wave per group of four.
  int c = (c_in == 1 || c_in == 2) ? 4 : c_in;
  if (s == 4 && c_in >= 3 && (c_in & 3)) c = (c_in + 3) & ~3;
  // 20 states: 3 -> 4, 5 .. 7 -> 8 (k_thorough_aa_mfma serves 4 and 8 categories; more go to the general kernel)
  if (s == 20 && (c_in == 3 || (c_in >= 5 && c_in <= 7))) c = (c_in + 3) & ~3;
  // per-rate scaler rows from the caller ([W][c_in]) cannot be padded here: such a context keeps its category
  // count and runs on the general kernel (as before the category groups existed)
  if (c_in >= 3 && c != c_in && (d->flags & EPA_FLAG_RATE_SCALERS) && !tree) c = c_in;
  if (c_in != c && (d->flags & EPA_FLAG_RATE_SCALERS) && !tree)
    return epa_fail(ctx, EPA_ERR_UNSUPPORTED,
                    "per-rate scaler arrays of a 1- or 2-category model (replicated to four): use epa_dev_create_from_tree");
  ctx->c_in = c_in;
  if (!d->sites || !d->branches) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "empty reference");
  const double pinv = d->prop_invar;
  if (!(pinv >= 0.0 && pinv < 1.0)) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "prop_invar must be in [0, 1)");
  if (pinv > 0.0 && !tree && !d->invariant_state)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "prop_invar > 0 needs invariant_state");
  if (!d->eigenvals || !d->eigenvecs_u || !d->eigenvecs_uinv || !d->freqs || !d->rates ||
      !d->rate_weights || (!tree && !d->prox_clv) || !d->branch_length)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null model / reference pointer in descriptor");
  if (!tree && !d->dist_clv && !d->dist_tipchars)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "descriptor has neither dist_clv nor dist_tipchars");
  ctx->device = device;
  ctx->s = s; ctx->c = c; ctx->W = d->sites; ctx->B = d->branches;
  ctx->ncols = s == 4 ? 16 : 24;
  ctx->aa_x_as_n = (int)d->aa_x_as_n;
  EPA_HIP(ctx, hipSetDevice(device));
  {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0)
      ctx->n_cu = n;
  }

  ModelDev& m = ctx->hmodel;
  memset(&m, 0, sizeof(m));
  m.s = s; m.c = c; m.ncols = ctx->ncols;
  for (int i = 0; i < s * s; ++i) { m.U[i] = d->eigenvecs_u[i]; m.Ui[i] = d->eigenvecs_uinv[i]; }
  for (int i = 0; i < s; ++i) { m.lam[i] = d->eigenvals[i]; m.pi[i] = d->freqs[i]; }
  for (int k = 0; k < c; ++k) {
    // +I (libpll): every P-matrix uses r / (1 - p); the (1 - p) factor of the site likelihood
    // (1-p) sum_k w_k L_k + p pi_inv is folded into the weights
    const int src = c_in <= 2 ? k % c_in : std::min(k, c_in - 1);
    m.rate[k] = d->rates[src] / (1.0 - pinv);
    m.w[k] = (c_in <= 2 ? d->rate_weights[src] * (double)c_in / (double)c : (k < c_in ? d->rate_weights[k] : 0.0)) * (1.0 - pinv);
  }
  {
    // Internal convention: eigenvalue 0 is the stationary (zero) one.  Swap the largest
    // eigenvalue to index 0 (columns of U, rows of U^-1); when it is numerically zero it is
    // set to exactly 0 so the kernels can drop its derivative terms.
    int imax = 0;
    double amax = 0.0;
    for (int i = 0; i < s; ++i) {
      if (m.lam[i] > m.lam[imax]) imax = i;
      amax = std::max(amax, fabs(m.lam[i]));
    }
    if (imax != 0) {
      std::swap(m.lam[0], m.lam[imax]);
      for (int i = 0; i < s; ++i) std::swap(m.U[i * s], m.U[i * s + imax]);
      for (int j = 0; j < s; ++j) std::swap(m.Ui[j], m.Ui[imax * s + j]);
    }
    ctx->dna_zero0 = fabs(m.lam[0]) <= 1e-9 * amax && !(d->flags & EPA_FLAG_KEEP_EIGENVALUES);
    if (ctx->dna_zero0) m.lam[0] = 0.0;
    if (pinv > 0.0 && !ctx->dna_zero0)
      return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "+I needs a rate matrix with a zero eigenvalue (any proper GTR)");
  }
  for (int col = 0; col < ctx->ncols; ++col) {
    m.colmask[col] = column_mask(s, col);
    for (int x = 0; x < s; ++x) {
      double acc = 0.0;
      for (int i = 0; i < s; ++i)
        if ((m.colmask[col] >> i) & 1u) acc += m.Ui[x * s + i];
      m.qt[col * s + x] = acc;
    }
  }
  if (s == 20 && ctx->aa_x_as_n)  // quirk D4: the lookup (preplacement) column of 'X' holds asparagine's numbers
    m.colmask[21] = m.colmask[11];  // AA_COLS[21] == 'X', AA_COLS[11] == 'N'
  const bool dna_groups = s == 4 && (c & 3) == 0 && c <= 16;   // 1 .. 4 groups of four categories
  if (dna_groups) {
    for (int i = 0; i < 16; ++i) { ctx->dna.U[i] = m.U[i]; ctx->dna.Ui[i] = m.Ui[i]; }
    for (int i = 0; i < 4; ++i) { ctx->dna.lam[i] = m.lam[i]; ctx->dna.pi[i] = m.pi[i]; }
    for (int k = 0; k < 16; ++k) { ctx->dna.rate[k] = k < c ? m.rate[k] : 1.0; ctx->dna.w[k] = k < c ? m.w[k] : 0.0; }
    ctx->dna.ng = c / 4;
  }
  ctx->blo.min_branch = d->blo_min_branch > 0 ? d->blo_min_branch : 1e-4;
  ctx->blo.max_branch = d->blo_max_branch > 0 ? d->blo_max_branch : 100.0;
  ctx->blo.default_branch = d->blo_default_branch > 0 ? d->blo_default_branch : 0.1;
  ctx->blo.epsilon = d->blo_epsilon > 0 ? d->blo_epsilon : 0.1;
  ctx->blo.pendant_default = d->pendant_default > 0 ? d->pendant_default : -log(0.9);
  ctx->blo.max_rounds = d->blo_max_rounds ? d->blo_max_rounds : 32;
  ctx->blo.max_newton = d->blo_max_newton ? d->blo_max_newton : 30;
  if ((d->flags & EPA_FLAG_SLIDING_BLO) && (d->flags & EPA_FLAG_RAXML_BLO))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "flags: sliding and raxml BLO are mutually exclusive");
  ctx->blo.sliding = (d->flags & EPA_FLAG_RAXML_BLO) ? 0 : 1;
  ctx->blo.newton_variant = ((d->flags & EPA_FLAG_NEWTON_SLOW_BISECT) ? 1u : 0u) |
                            ((d->flags & EPA_FLAG_NEWTON_STRICT_DF) ? 2u : 0u);
  ctx->rate_scalers = (d->flags & EPA_FLAG_RATE_SCALERS) != 0;
  // the tuned thorough kernels are built for 4 categories (the Newton-variant switches exist to pin
  // parity once a reference build is at hand, not for production runs: they are served by the general
  // kernel too, the tuned kernels cost 1.7 % with them).  --raxml-blo has tuned instantiations for
  // nucleotide models with an exact zero eigenvalue, with or without +I (k_thorough_dna<.., INV, .., LOCAL>),
  // and for 20-state models (k_thorough_aa_mfma<.., LOCAL>); everything else local goes general.
  const bool tuned_local = (s == 4 && ctx->dna_zero0) || s == 20;
  // (more than 4 categories: tuned for nucleotide models, sliding rule, zero eigenvalue -- the class
  // launcher of thorough_dna.hip sends what it does not serve to the general kernel itself)
  const bool tuned_cats = c == 4 || (dna_groups && ctx->blo.sliding && ctx->dna_zero0) ||
                          (s == 20 && c == 8);   // k_thorough_aa_mfma<.., NC = 8>
  ctx->generic_thorough = !tuned_cats || (!ctx->blo.sliding && !tuned_local) || ctx->blo.newton_variant != 0 ||
                          (d->flags & EPA_FLAG_KEEP_EIGENVALUES) != 0;     // every term as libpll: the general kernel
  ctx->generic_native = ctx->generic_thorough;
  // (epa_dev_set_option "thorough_generic" sends a context that has tuned kernels to the general one as well)

  EPA_HIP(ctx, hipMalloc(&ctx->dmodel, sizeof(ModelDev)));
  EPA_HIP(ctx, hipMemcpy(ctx->dmodel, &m, sizeof(ModelDev), hipMemcpyHostToDevice));

  const size_t W = ctx->W, B = ctx->B, cs = (size_t)c * s;
  EPA_HIP(ctx, hipMalloc(&ctx->refT, sizeof(double) * 2 * B * cs * W));
  EPA_HIP(ctx, hipMalloc(&ctx->th_ctr, 256 * epa_ctx::N_BANKS));
  EPA_HIP(ctx, hipMalloc(&ctx->scSum, sizeof(uint32_t) * B * W));
  EPA_HIP(ctx, hipMemset(ctx->scSum, 0, sizeof(uint32_t) * B * W));
  EPA_HIP(ctx, hipMalloc(&ctx->blen, sizeof(double) * B));
  EPA_HIP(ctx, hipMalloc(&ctx->lookup, sizeof(double) * B * W * ctx->ncols));
  if (!ctx->generic_thorough) {  // starting vectors of the tuned thorough kernels
    EPA_HIP(ctx, hipMalloc(&ctx->refI, sizeof(double) * B * cs * W));
    EPA_HIP(ctx, hipMalloc(&ctx->resc0, B * W));
  }
  ctx->h_blen.assign(d->branch_length, d->branch_length + B);
  EPA_HIP(ctx, hipMemcpy(ctx->blen, d->branch_length, sizeof(double) * B, hipMemcpyHostToDevice));

  uint32_t* d_tipmap = nullptr;
  if (d->tipmap && d->tipmap_size) {
    d_tipmap = (uint32_t*)epa_scratch(ctx, 2, sizeof(uint32_t) * d->tipmap_size);
    if (!d_tipmap) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(tipmap)");
    EPA_HIP(ctx, hipMemcpy(d_tipmap, d->tipmap, sizeof(uint32_t) * d->tipmap_size,
                           hipMemcpyHostToDevice));
  }
  if (pinv > 0.0) {
    // per-site constant of the +I term: p * pi[state] for sites invariant over the reference tips
    std::vector<int8_t> inv_buf;
    const int8_t* inv = d->invariant_state;
    if (!inv) {  // tree path: derive it from the tip sequences (pll_update_invariant_sites)
      if (!tree->tipchars || !d->tipmap)
        return epa_fail(ctx, EPA_ERR_INVALID_ARG, "prop_invar > 0: no tip sequences to derive invariant sites from");
      inv_buf.resize(W);
      for (size_t w = 0; w < W; ++w) {
        uint32_t all = s == 4 ? 15u : ((1u << 20) - 1);
        for (uint32_t t = 0; t < tree->tips; ++t) {
          const uint8_t code = tree->tipchars[(size_t)t * W + w];
          if (code >= d->tipmap_size) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tip code outside the tip map");
          all &= d->tipmap[code];
        }
        int st = -1;
        if (all && !(all & (all - 1))) { st = 0; while (!((all >> st) & 1u)) ++st; }
        inv_buf[w] = (int8_t)st;
      }
      inv = inv_buf.data();
    }
    std::vector<double> cinv(W);
    for (size_t w = 0; w < W; ++w) {
      if (inv[w] >= s) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "invariant_state out of range");
      cinv[w] = inv[w] >= 0 ? pinv * d->freqs[inv[w]] : 0.0;
    }
    EPA_HIP(ctx, hipMalloc(&ctx->cinv, sizeof(double) * W));
    EPA_HIP(ctx, hipMemcpy(ctx->cinv, cinv.data(), sizeof(double) * W, hipMemcpyHostToDevice));
    ctx->inv_w0 = 1.0 / m.w[0];
  }
  if (tree) return precompute_from_tree(ctx, tree, d_tipmap);
  // per-rate scaler rows of the caller ([W][c], libpll layout): collected per side, aligned at the end
  const size_t cdim = ctx->rate_scalers ? (size_t)c : 1;
  uint32_t* d_sc_side = nullptr;
  if (ctx->rate_scalers) {
    if (hipMalloc(&d_sc_side, sizeof(uint32_t) * 2 * B * cdim * W) != hipSuccess)
      return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(per-rate scaler staging)");
    (void)hipMemset(d_sc_side, 0, sizeof(uint32_t) * 2 * B * cdim * W);
  }
  // upload + transform, one CLV at a time through two alternating staging buffers
  for (size_t b = 0; b < B; ++b) {
    for (int side = 0; side < 2; ++side) {
      const double* clv = side == 0 ? d->prox_clv[b] : (d->dist_clv ? d->dist_clv[b] : nullptr);
      const uint8_t* tip = (side == 1 && d->dist_tipchars) ? d->dist_tipchars[b] : nullptr;
      const uint32_t* sc = side == 0 ? (d->prox_scaler ? d->prox_scaler[b] : nullptr)
                                     : (d->dist_scaler ? d->dist_scaler[b] : nullptr);
      if (side == 0 && !clv) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "prox_clv[b] is NULL");
      if (side == 1 && !clv && !tip)
        return epa_fail(ctx, EPA_ERR_INVALID_ARG, "branch has neither dist_clv nor dist_tipchars");
      if (tip && !clv && !d_tipmap)
        return epa_fail(ctx, EPA_ERR_INVALID_ARG, "dist_tipchars given without tipmap");
      double* dst = ctx->refT + (2 * b + side) * cs * W;
      const int slot = side;  // staging slots 0/1
      if (clv) {
        const double* dclv = (const double*)epa_to_device(ctx, slot, clv, sizeof(double) * W * (size_t)c_in * s);
        if (!dclv) return epa_fail(ctx, EPA_ERR_HIP, "staging copy of CLV failed");
        int rc = launch_transform(ctx, dclv, nullptr, nullptr, 0, dst);
        if (rc) return rc;
      } else {
        const uint8_t* dtip = (const uint8_t*)epa_to_device(ctx, slot, tip, W);
        if (!dtip) return epa_fail(ctx, EPA_ERR_HIP, "staging copy of tipchars failed");
        int rc = launch_transform(ctx, nullptr, dtip, d_tipmap, d->tipmap_size, dst);
        if (rc) return rc;
      }
      if (sc) {
        const uint32_t* dsc = (const uint32_t*)epa_to_device(ctx, 3 + side, sc, sizeof(uint32_t) * W * cdim);
        if (!dsc) return epa_fail(ctx, EPA_ERR_HIP, "staging copy of scaler failed");
        hipLaunchKernelGGL(k_add_scaler, dim3((uint32_t)((W * cdim + 255) / 256)), dim3(256), 0, ctx->stream, dsc,
                           d_sc_side ? d_sc_side + (2 * b + side) * cdim * W : ctx->scSum + b * W, (uint32_t)W,
                           (uint32_t)cdim);
      }
      // staging buffers are reused by the next branch: pageable H2D copies are synchronous
      // with respect to the host buffer but the kernel must finish before the slot is reused
      EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  if (d_sc_side) {
    hipLaunchKernelGGL(k_align_rates, dim3((uint32_t)((W + 255) / 256), (uint32_t)B), dim3(256), 0, ctx->stream,
                       ctx->dmodel, ctx->refT, d_sc_side, ctx->scSum, (uint32_t)W);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_sc_side);
  }
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}

// =============================================================================================
// Reference precompute on device (epa_dev_create_from_tree).
// One directional CLV = (P(la) child_a) o (P(lb) child_b), computed in the eigenbasis:
//   I_ki = (U (e^{lam r_k la} o a~_k))_i (U (e^{lam r_k lb} o b~_k))_i,  out~ = U^-1 I
// with the per-site 2^256 rescale of pll_update_partials; children are read from, and the result
// is written to, the refT slot of the branch side the record belongs to (every record is the
// proximal or distal side of exactly one branch), so no state-space CLV is ever stored.
// Thread per site, blockIdx.y = record inside a dependency level.
// =============================================================================================
struct RecDev {
  uint32_t side;    // destination: branch side index (2b or 2b+1)
  uint32_t ca, cb;  // operands: side index of an inner child, or EPA_TIP | tip
  uint32_t pad;
  double la, lb;
};

template <int S>
__global__ void __launch_bounds__(256) k_clv_level(const ModelDev* __restrict__ m,
                                                   const RecDev* __restrict__ recs,
                                                   const uint8_t* __restrict__ tipchars,
                                                   const uint32_t* __restrict__ tipmap, uint32_t W,
                                                   double* __restrict__ refT,
                                                   uint32_t* __restrict__ sc_side, int rate_scalers) {
  __shared__ double U[S * S], Ui[S * S];
  __shared__ double Ea[EPA_MAX_CATS * S], Eb[EPA_MAX_CATS * S];
  const int c = m->c;
  const RecDev r = recs[blockIdx.y];
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) { U[i] = m->U[i]; Ui[i] = m->Ui[i]; }
  for (int i = threadIdx.x; i < c * S; i += blockDim.x) {
    const int k = i / S, x = i % S;
    Ea[i] = exp(m->lam[x] * m->rate[k] * r.la);
    Eb[i] = exp(m->lam[x] * m->rate[k] * r.lb);
  }
  __syncthreads();
  const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= W) return;
  const size_t cs = (size_t)c * S;
  const bool tip_a = r.ca & EPA_TIP, tip_b = r.cb & EPA_TIP;
  const double* A = tip_a ? nullptr : refT + (size_t)r.ca * cs * W + site;
  const double* Bv = tip_b ? nullptr : refT + (size_t)r.cb * cs * W + site;
  double ta[S], tb[S];  // U^-1 image of a tip's state set (category independent)
  if (tip_a) {
    const uint32_t mask = tipmap[tipchars[(size_t)(r.ca & ~EPA_TIP) * W + site]];
    for (int x = 0; x < S; ++x) {
      double acc = 0.0;
      for (int i = 0; i < S; ++i) if ((mask >> i) & 1u) acc += Ui[x * S + i];
      ta[x] = acc;
    }
  }
  if (tip_b) {
    const uint32_t mask = tipmap[tipchars[(size_t)(r.cb & ~EPA_TIP) * W + site]];
    for (int x = 0; x < S; ++x) {
      double acc = 0.0;
      for (int i = 0; i < S; ++i) if ((mask >> i) & 1u) acc += Ui[x * S + i];
      tb[x] = acc;
    }
  }
  double I[EPA_MAX_CATS][S];
  bool ksmall[EPA_MAX_CATS];
  double mx = 0.0;
  for (int k = 0; k < c; ++k) {
    double av[S], bv[S];
#pragma unroll
    for (int x = 0; x < S; ++x) {
      av[x] = (tip_a ? ta[x] : A[(size_t)(k * S + x) * W]) * Ea[k * S + x];
      bv[x] = (tip_b ? tb[x] : Bv[(size_t)(k * S + x) * W]) * Eb[k * S + x];
    }
    double mxk = 0.0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int x = 0; x < S; ++x) {
        a = fma(U[i * S + x], av[x], a);
        b = fma(U[i * S + x], bv[x], b);
      }
      const double v = a * b;
      I[k][i] = v;
      mxk = fmax(mxk, v);
    }
    ksmall[k] = mxk < 0x1p-256;
    mx = fmax(mx, mxk);
  }
  // per-site scaling: all c * s entries below 2^-256; per-rate: every category on its own
  const bool resc = mx < 0x1p-256;
  double* out = refT + (size_t)r.side * cs * W + site;
  for (int k = 0; k < c; ++k) {
    const double mult = (rate_scalers ? ksmall[k] : resc) ? 0x1p+256 : 1.0;
#pragma unroll
    for (int x = 0; x < S; ++x) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < S; ++i) acc = fma(Ui[x * S + i], I[k][i] * mult, acc);
      out[(size_t)(k * S + x) * W] = acc;
    }
  }
  if (rate_scalers) {   // sc_side [2B][c][W]
    for (int k = 0; k < c; ++k) {
      const uint32_t sa = tip_a ? 0u : sc_side[((size_t)r.ca * c + k) * W + site];
      const uint32_t sb = tip_b ? 0u : sc_side[((size_t)r.cb * c + k) * W + site];
      sc_side[((size_t)r.side * c + k) * W + site] = sa + sb + (ksmall[k] ? 1u : 0u);
    }
    return;
  }
  const uint32_t sa = tip_a ? 0u : sc_side[(size_t)r.ca * W + site];
  const uint32_t sb = tip_b ? 0u : sc_side[(size_t)r.cb * W + site];
  sc_side[(size_t)r.side * W + site] = sa + sb + (resc ? 1u : 0u);
}

// tip sides of the branches: U^-1 image of the tip's state set, replicated over the categories
__global__ void __launch_bounds__(256) k_tip_sides(const ModelDev* __restrict__ m,
                                                   const uint32_t* __restrict__ tip_of_branch,
                                                   const uint8_t* __restrict__ tipchars,
                                                   const uint32_t* __restrict__ tipmap, uint32_t W,
                                                   double* __restrict__ refT) {
  const uint32_t b = blockIdx.y;
  const uint32_t t = tip_of_branch[b];
  if (t == 0xffffffffu) return;
  const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= W) return;
  const int s = m->s, c = m->c;
  const uint32_t mask = tipmap[tipchars[(size_t)t * W + site]];
  double* out = refT + (size_t)(2 * b + 1) * c * s * W + site;
  for (int x = 0; x < s; ++x) {
    double acc = 0.0;
    for (int i = 0; i < s; ++i) if ((mask >> i) & 1u) acc += m->Ui[x * s + i];
    for (int k = 0; k < c; ++k) out[(size_t)(k * s + x) * W] = acc;
  }
}

__global__ void k_scaler_sum(const uint32_t* __restrict__ sc_side, uint32_t* __restrict__ scSum, size_t n,
                             uint32_t W) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // i = b * W + site
  if (i >= n) return;
  const size_t b = i / W, w = i % W;
  scSum[i] = sc_side[(2 * b) * W + w] + sc_side[(2 * b + 1) * W + w];
}

// Per-rate scalers -> the per-site form every other kernel reads.  The two sides of a branch only
// ever meet category by category (inner CLV, edge lnL and sumtables are bilinear in (X_k, D_k)), so
// libpll's alignment of the categories at evaluation time -- category k, whose total count
// t_k = prox_k + dist_k exceeds the site's minimum by d, is multiplied by 2^(-256 min(d, 4))
// (rate_scalings / scale_minlh, restated in oracle/epa_oracle.c rate_alignment) -- can be applied
// ONCE to the stored proximal vector; the site then carries the minimum count alone.  What per-rate
// scaling protects against (a category that underflows while another keeps the site from being
// rescaled) happens during the level-by-level precompute, which keeps one count per category
// (k_clv_level); from here on the tuned per-site kernels run unchanged.
// sc_side [2B][c][W] per-rate counts of the branch sides; thread per (branch, site).
__global__ void __launch_bounds__(256) k_align_rates(const ModelDev* __restrict__ m, double* __restrict__ refT,
                                                     const uint32_t* __restrict__ sc_side,
                                                     uint32_t* __restrict__ scSum, uint32_t W) {
  const uint32_t b = blockIdx.y;
  const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= W) return;
  const int c = m->c, s = m->s;
  const uint32_t* sp = sc_side + (size_t)(2 * b) * c * W + site;
  const uint32_t* sd = sc_side + (size_t)(2 * b + 1) * c * W + site;
  uint32_t mn = 0xffffffffu;
  for (int k = 0; k < c; ++k) mn = min(mn, sp[(size_t)k * W] + sd[(size_t)k * W]);
  double* X = refT + (size_t)(2 * b) * c * s * W + site;
  for (int k = 0; k < c; ++k) {
    const uint32_t d = min(sp[(size_t)k * W] + sd[(size_t)k * W] - mn, 4u);
    if (!d) continue;
    const double f = d == 1 ? 0x1p-256 : d == 2 ? 0x1p-512 : d == 3 ? 0x1p-768 : 0x1p-1024;
    for (int x = 0; x < s; ++x) X[(size_t)(k * s + x) * W] *= f;
  }
  scSum[(size_t)b * W + site] = mn;
}

static int precompute_from_tree(epa_ctx* ctx, const epa_tree_desc* t, const uint32_t* d_tipmap) {
  const uint32_t n = t->tips, R = t->inner_records, B = ctx->B, W = ctx->W;
  if (!t->tipchars || !t->rec_child_a || !t->rec_child_b || !t->rec_length_a || !t->rec_length_b ||
      !t->branch_prox || !t->branch_dist || !d_tipmap)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null pointer in tree descriptor");
  if (n < 3 || R != 3 * (n - 2) || B != 2 * n - 3)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: sizes are not those of an unrooted binary tree");
  // record -> branch side
  std::vector<uint32_t> side_of(R, 0xffffffffu), tip_of_branch(B, 0xffffffffu);
  for (uint32_t b = 0; b < B; ++b) {
    const uint32_t p = t->branch_prox[b], d = t->branch_dist[b];
    if ((p & EPA_TIP) || p >= R || side_of[p] != 0xffffffffu)
      return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: bad proximal record");
    side_of[p] = 2 * b;
    if (d & EPA_TIP) {
      if ((d & ~EPA_TIP) >= n) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: bad tip id");
      tip_of_branch[b] = d & ~EPA_TIP;
    } else {
      if (d >= R || side_of[d] != 0xffffffffu)
        return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: bad distal record");
      side_of[d] = 2 * b + 1;
    }
  }
  for (uint32_t r = 0; r < R; ++r)
    if (side_of[r] == 0xffffffffu)
      return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: a record is not a side of any branch");
  // dependency levels (explicit stack: the dependency chain of a caterpillar tree is n long)
  std::vector<int> level(R, -1);
  auto child_level = [&](uint32_t op) { return (op & EPA_TIP) ? 0 : level[op]; };
  for (uint32_t root = 0; root < R; ++root) {
    if (level[root] >= 0) continue;
    std::vector<uint32_t> st{root};
    while (!st.empty()) {
      const uint32_t r = st.back();
      if (level[r] >= 0) { st.pop_back(); continue; }
      const uint32_t a = t->rec_child_a[r], b = t->rec_child_b[r];
      if ((!(a & EPA_TIP) && a >= R) || (!(b & EPA_TIP) && b >= R) || ((a & EPA_TIP) && (a & ~EPA_TIP) >= n) ||
          ((b & EPA_TIP) && (b & ~EPA_TIP) >= n))
        return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: bad child operand");
      if (st.size() > (size_t)R + 1) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "tree descriptor: cyclic records");
      const bool na = !(a & EPA_TIP) && level[a] < 0, nb = !(b & EPA_TIP) && level[b] < 0;
      if (na) st.push_back(a);
      if (nb) st.push_back(b);
      if (na || nb) continue;
      level[r] = 1 + std::max(child_level(a), child_level(b));
      st.pop_back();
    }
  }
  int nlev = 0;
  for (uint32_t r = 0; r < R; ++r) nlev = std::max(nlev, level[r]);
  std::vector<std::vector<uint32_t>> by_level(nlev + 1);
  for (uint32_t r = 0; r < R; ++r) by_level[level[r]].push_back(r);
  std::vector<RecDev> recs;
  recs.reserve(R);
  std::vector<uint32_t> lev_begin;
  auto opnd = [&](uint32_t op) { return (op & EPA_TIP) ? op : side_of[op]; };
  for (int l = 1; l <= nlev; ++l) {
    lev_begin.push_back((uint32_t)recs.size());
    for (uint32_t r : by_level[l])
      recs.push_back(RecDev{side_of[r], opnd(t->rec_child_a[r]), opnd(t->rec_child_b[r]), 0, t->rec_length_a[r],
                            t->rec_length_b[r]});
  }
  lev_begin.push_back((uint32_t)recs.size());
  // device buffers (freed at the end: only refT / scSum persist)
  uint8_t* d_tips = nullptr;
  RecDev* d_recs = nullptr;
  uint32_t *d_sc = nullptr, *d_tob = nullptr;
  auto cleanup = [&]() {
    if (d_tips) (void)hipFree(d_tips);
    if (d_recs) (void)hipFree(d_recs);
    if (d_sc) (void)hipFree(d_sc);
    if (d_tob) (void)hipFree(d_tob);
  };
#define TREE_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return epa_fail(ctx, EPA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
  TREE_HIP(hipMalloc(&d_tips, (size_t)n * W));
  TREE_HIP(hipMemcpy(d_tips, t->tipchars, (size_t)n * W, hipMemcpyHostToDevice));
  TREE_HIP(hipMalloc(&d_recs, sizeof(RecDev) * recs.size()));
  TREE_HIP(hipMemcpy(d_recs, recs.data(), sizeof(RecDev) * recs.size(), hipMemcpyHostToDevice));
  const size_t cdim = ctx->rate_scalers ? (size_t)ctx->c : 1;
  TREE_HIP(hipMalloc(&d_sc, sizeof(uint32_t) * 2 * (size_t)B * cdim * W));
  TREE_HIP(hipMemset(d_sc, 0, sizeof(uint32_t) * 2 * (size_t)B * cdim * W));
  TREE_HIP(hipMalloc(&d_tob, sizeof(uint32_t) * B));
  TREE_HIP(hipMemcpy(d_tob, tip_of_branch.data(), sizeof(uint32_t) * B, hipMemcpyHostToDevice));
  const dim3 blk(256);
  hipLaunchKernelGGL(k_tip_sides, dim3((W + 255) / 256, B), blk, 0, ctx->stream, ctx->dmodel, d_tob, d_tips,
                     d_tipmap, W, ctx->refT);
  for (size_t l = 0; l + 1 < lev_begin.size(); ++l) {
    const uint32_t cnt = lev_begin[l + 1] - lev_begin[l];
    if (!cnt) continue;
    for (uint32_t off = 0; off < cnt; off += 65535) {  // grid.y limit
      const dim3 grid((W + 255) / 256, std::min<uint32_t>(65535, cnt - off));
      if (ctx->s == 4)
        hipLaunchKernelGGL(k_clv_level<4>, grid, blk, 0, ctx->stream, ctx->dmodel, d_recs + lev_begin[l] + off,
                           d_tips, d_tipmap, W, ctx->refT, d_sc, ctx->rate_scalers ? 1 : 0);
      else
        hipLaunchKernelGGL(k_clv_level<20>, grid, blk, 0, ctx->stream, ctx->dmodel, d_recs + lev_begin[l] + off,
                           d_tips, d_tipmap, W, ctx->refT, d_sc, ctx->rate_scalers ? 1 : 0);
    }
  }
  if (ctx->rate_scalers) {
    hipLaunchKernelGGL(k_align_rates, dim3((W + 255) / 256, B), blk, 0, ctx->stream, ctx->dmodel, ctx->refT, d_sc,
                       ctx->scSum, W);
  } else {
    const size_t nbw = (size_t)B * W;
    hipLaunchKernelGGL(k_scaler_sum, dim3((uint32_t)((nbw + 255) / 256)), blk, 0, ctx->stream, d_sc, ctx->scSum, nbw, W);
  }
  TREE_HIP(hipStreamSynchronize(ctx->stream));
  TREE_HIP(hipGetLastError());
#undef TREE_HIP
  cleanup();
  return EPA_OK;
}

// edge log-likelihood of the reference tree at branch b from the two stored sides
template <int S>
__global__ void __launch_bounds__(256) k_tree_logl(const ModelDev* __restrict__ m,
                                                  const double* __restrict__ refT,
                                                  const uint32_t* __restrict__ scSum, uint32_t b, double len,
                                                  uint32_t W, const double* __restrict__ cinv,
                                                  double* __restrict__ partial) {
  __shared__ double U[S * S];
  __shared__ double E[EPA_MAX_CATS * S];
  __shared__ double red[256];
  const int c = m->c;
  for (int i = threadIdx.x; i < S * S; i += blockDim.x) U[i] = m->U[i];
  for (int i = threadIdx.x; i < c * S; i += blockDim.x) E[i] = exp(m->lam[i % S] * m->rate[i / S] * len);
  __syncthreads();
  const uint32_t site = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (site < W) {
    const size_t cs = (size_t)c * S;
    const double* X = refT + (size_t)(2 * b) * cs * W + site;
    const double* D = refT + (size_t)(2 * b + 1) * cs * W + site;
    double L = 0.0;
    for (int k = 0; k < c; ++k) {
      double xv[S], dv[S];
#pragma unroll
      for (int x = 0; x < S; ++x) { xv[x] = X[(size_t)(k * S + x) * W]; dv[x] = D[(size_t)(k * S + x) * W] * E[k * S + x]; }
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        double a = 0.0, d = 0.0;
#pragma unroll
        for (int x = 0; x < S; ++x) { a = fma(U[i * S + x], xv[x], a); d = fma(U[i * S + x], dv[x], d); }
        t = fma(m->pi[i] * a, d, t);
      }
      L = fma(m->w[k], t, L);
    }
    if (cinv) L += cinv[site];
    v = log(L) + (double)scSum[(size_t)b * W + site] * (-256.0 * 0.6931471805599453094);
  }
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

extern "C" int epa_dev_tree_logl(epa_ctx* ctx, uint32_t branch, double* lnl) {
  if (!ctx || !lnl || branch >= ctx->B) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "bad argument");
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  const uint32_t nblk = (ctx->W + 255) / 256;
  double* d_part = (double*)epa_scratch(ctx, 9, sizeof(double) * nblk);
  if (!d_part) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(tree lnL partials)");
  const double len = ctx->h_blen[branch];
  if (ctx->s == 4)
    hipLaunchKernelGGL(k_tree_logl<4>, dim3(nblk), dim3(256), 0, ctx->stream, ctx->dmodel, ctx->refT, ctx->scSum,
                       branch, len, ctx->W, ctx->cinv, d_part);
  else
    hipLaunchKernelGGL(k_tree_logl<20>, dim3(nblk), dim3(256), 0, ctx->stream, ctx->dmodel, ctx->refT, ctx->scSum,
                       branch, len, ctx->W, ctx->cinv, d_part);
  std::vector<double> part(nblk);
  EPA_HIP(ctx, hipMemcpyAsync(part.data(), d_part, sizeof(double) * nblk, hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double s = 0.0;
  for (double p : part) s += p;  // fixed order: deterministic
  *lnl = s;
  return EPA_OK;
}

extern "C" int epa_dev_create_from_tree(const epa_tree_desc* tree, int device, epa_ctx** out) {
  if (!tree || !out) return epa_fail(nullptr, EPA_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (epa_dev_device_count() <= 0)
    return epa_fail(nullptr, EPA_ERR_NO_DEVICE,
                    "no HIP device visible: libepa_dev has no CPU fallback by design");
  epa_ctx* ctx = new epa_ctx();
  int rc = create_impl(&tree->ref, device, ctx, tree);
  if (rc != EPA_OK) {
    g_create_err = ctx->err;
    epa_dev_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return EPA_OK;
}

extern "C" int epa_dev_create(const epa_ref_desc* desc, int device, epa_ctx** out) {
  if (!desc || !out) return epa_fail(nullptr, EPA_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  if (epa_dev_device_count() <= 0)
    return epa_fail(nullptr, EPA_ERR_NO_DEVICE,
                    "no HIP device visible: libepa_dev has no CPU fallback by design");
  epa_ctx* ctx = new epa_ctx();
  int rc = create_impl(desc, device, ctx);
  if (rc != EPA_OK) {
    g_create_err = ctx->err;
    epa_dev_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return EPA_OK;
}

extern "C" int epa_dev_build_lookup(epa_ctx* ctx) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  if (ctx->lookup_built) return EPA_OK;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = launch_build_lookup(ctx);
  if (rc) return rc;
  // once per context: the tables are COMPLETE when this returns.  The build runs on the context's stream, the
  // chunk pipeline's slots on non-blocking streams of their own, and an EPA_CHUNK_HOST_ORDERED launch skips the
  // event hop that would order a slot behind the context's stream -- a lazily built lookup must not be something
  // any later launch on any stream has to be ordered against
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->lookup_built = true;
  return EPA_OK;
}

// fetch a small u32 array to the host if it lives on the device
static const uint32_t* host_view(const uint32_t* p, size_t n, std::vector<uint32_t>& buf,
                                 hipStream_t st) {
  if (!epa_is_device_ptr(p)) return p;
  buf.resize(n);
  (void)hipStreamSynchronize(st);
  if (hipMemcpy(buf.data(), p, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
    return nullptr;
  return buf.data();
}

static int check_windows(epa_ctx* ctx, const uint32_t* hb, const uint32_t* hs, uint32_t Q,
                         uint32_t* max_span) {
  uint32_t mx = 0;
  for (uint32_t q = 0; q < Q; ++q) {
    if (hs[q] == 0)
      return epa_fail(ctx, EPA_ERR_QUERY_ALL_GAP,
                      "Sequence " + std::to_string(q) + " does not appear to have any non-gap sites!");
    if ((uint64_t)hb[q] + hs[q] > ctx->W)
      return epa_fail(ctx, EPA_ERR_QUERY_WIDTH,
                      "Query sequence length not same as reference alignment!");
    mx = std::max(mx, hs[q]);
  }
  if (ctx->code_stride && mx > ctx->code_stride)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "a window is longer than the compact code stride");
  *max_span = mx;
  return EPA_OK;
}

extern "C" int epa_dev_preplace(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin,
                                const uint32_t* win_span, uint32_t Q, double* lnl) {
  if (!ctx || !q_codes || !win_begin || !win_span || !lnl)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null argument");
  if (Q == 0) return EPA_OK;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = epa_dev_build_lookup(ctx);
  if (rc) return rc;
  const uint8_t* d_codes = epa_codes_to_device(ctx, q_codes, Q);
  const uint32_t* d_begin = (const uint32_t*)epa_to_device(ctx, 1, win_begin, sizeof(uint32_t) * Q);
  const uint32_t* d_span = (const uint32_t*)epa_to_device(ctx, 2, win_span, sizeof(uint32_t) * Q);
  if (!d_codes || !d_begin || !d_span) return epa_fail(ctx, EPA_ERR_HIP, "query upload failed");
  const bool out_dev = epa_is_device_ptr(lnl);
  double* d_lnl = out_dev ? lnl : (double*)epa_scratch(ctx, 3, sizeof(double) * (size_t)Q * ctx->B);
  if (!d_lnl) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(lnl)");
  rc = launch_preplace(ctx, d_codes, d_begin, d_span, Q, d_lnl, 0);
  if (rc) return rc;
  if (!out_dev)
    EPA_HIP(ctx, hipMemcpyAsync(lnl, d_lnl, sizeof(double) * (size_t)Q * ctx->B,
                                hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return preplace_check_status(ctx);  // windows are validated on device (k_make_groups)
}

extern "C" int epa_dev_thorough(epa_ctx* ctx, const epa_pair* pairs, uint64_t n_pairs,
                                const uint8_t* q_codes, const uint32_t* win_begin,
                                const uint32_t* win_span, uint32_t Q, epa_result* out,
                                epa_thorough_stats* stats) {
  if (!ctx || !q_codes || !win_begin || !win_span || (n_pairs && (!pairs || !out)))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null argument");
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_pairs == 0) return EPA_OK;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->refI) {  // per-branch starting vectors of the DNA kernel come with the lookup build
    int brc = epa_dev_build_lookup(ctx);
    if (brc) return brc;
  }
  std::vector<uint32_t> hb_buf, hs_buf;
  const uint32_t* hb = host_view(win_begin, Q, hb_buf, ctx->stream);
  const uint32_t* hs = host_view(win_span, Q, hs_buf, ctx->stream);
  if (!hb || !hs) return epa_fail(ctx, EPA_ERR_HIP, "cannot read window arrays");
  uint32_t max_span = 0;
  int rc = check_windows(ctx, hb, hs, Q, &max_span);
  if (rc) return rc;
  if (!epa_is_device_ptr(pairs)) {
    for (uint64_t i = 0; i < n_pairs; ++i)
      if (pairs[i].branch_id >= ctx->B || pairs[i].seq_id >= Q)
        return epa_fail(ctx, EPA_ERR_INVALID_ARG, "pair index out of range");
  }
  const uint8_t* d_codes = epa_codes_to_device(ctx, q_codes, Q);
  const uint32_t* d_begin = (const uint32_t*)epa_to_device(ctx, 1, win_begin, sizeof(uint32_t) * Q);
  const uint32_t* d_span = (const uint32_t*)epa_to_device(ctx, 2, win_span, sizeof(uint32_t) * Q);
  const epa_pair* d_pairs = (const epa_pair*)epa_to_device(ctx, 4, pairs, sizeof(epa_pair) * n_pairs);
  if (!d_codes || !d_begin || !d_span || !d_pairs)
    return epa_fail(ctx, EPA_ERR_HIP, "thorough input upload failed");
  const bool out_dev = epa_is_device_ptr(out);
  epa_result* d_out = out_dev ? out : (epa_result*)epa_scratch(ctx, 5, sizeof(epa_result) * n_pairs);
  unsigned long long* d_stats = (unsigned long long*)epa_scratch(ctx, 6, 256);
  if (!d_out || !d_stats) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(thorough out)");
  rc = epa_zero_async(ctx, d_stats, 128);
  if (rc) return rc;
  rc = launch_thorough(ctx, d_pairs, n_pairs, d_codes, d_begin, d_span, max_span, d_out, d_stats);
  if (rc) return rc;
  unsigned long long hst[16];
  if (!out_dev)
    EPA_HIP(ctx, hipMemcpyAsync(out, d_out, sizeof(epa_result) * n_pairs, hipMemcpyDeviceToHost,
                                ctx->stream));
  if (!out_dev || stats) {
    EPA_HIP(ctx, hipMemcpyAsync(hst, d_stats, 128, hipMemcpyDeviceToHost, ctx->stream));
    EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    epa_xcd_feedback(ctx, n_pairs, hst);
    ctx->last_stats.pairs = n_pairs;
    ctx->last_stats.rounds = hst[0];
    ctx->last_stats.newton_evals = hst[1];
    ctx->last_stats.reverts = hst[2];
    if (stats) *stats = ctx->last_stats;
    if (hst[3])  // pairs whose lnL came out -inf / NaN
      return epa_fail(ctx, EPA_ERR_NEG_INF,
                      "-INF logl at branch " + std::to_string((uint32_t)(hst[4] >> 32)) +
                          " with sequence " + std::to_string((uint32_t)(hst[4] & 0xffffffffu)));
  }
  return EPA_OK;
}

extern "C" int epa_dev_select_candidates(epa_ctx* ctx, const double* lnl, uint32_t Q,
                                         double threshold, epa_pair* pairs, uint64_t max_pairs,
                                         uint64_t* n_pairs) {
  if (!ctx || !lnl || !pairs || !n_pairs) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null argument");
  *n_pairs = 0;
  if (Q == 0) return EPA_OK;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  const double* d_lnl = (const double*)epa_to_device(ctx, 3, lnl, sizeof(double) * (size_t)Q * ctx->B);
  if (!d_lnl) return epa_fail(ctx, EPA_ERR_HIP, "lnl upload failed");
  const bool out_dev = epa_is_device_ptr(pairs);
  epa_pair* d_pairs = out_dev ? pairs : (epa_pair*)epa_scratch(ctx, 4, sizeof(epa_pair) * max_pairs);
  if (!d_pairs) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(pairs)");
  int rc = launch_select(ctx, d_lnl, Q, threshold, d_pairs, max_pairs, n_pairs);
  if (rc) return rc;
  if (!out_dev && *n_pairs) {
    EPA_HIP(ctx, hipMemcpy(pairs, d_pairs, sizeof(epa_pair) * (*n_pairs), hipMemcpyDeviceToHost));
  }
  return EPA_OK;
}

// The body of the reference's chunk loop (src/core/place.cpp:219-235) on device buffers:
// place() -> apply_heuristic() -> place_thorough().  The Q x B table never leaves HBM; ONE host
// sync in the middle (the candidate count sizes the thorough launch); the thorough kernels are
// queued on return, nothing is waited for after them.
// begin: preplacement + first half of the selection, nothing waited for; end: the wait for the
// candidate count (and the window-validation words, read back in the same copy), compaction / sort,
// the thorough launches.  rb: 64-word host block (pinned in the chunk pipeline).
static int chunk_body_begin(epa_ctx* ctx, const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span,
                            uint32_t Q, uint32_t max_span, double threshold, epa_pair* d_pairs,
                            uint64_t max_pairs, uint32_t* rb, SelectPending* sp, epa_result* d_res = nullptr,
                            unsigned long long* d_stats = nullptr) {
  // internal table: rows padded to whole 64-byte sectors (the preplacement kernels write 8
  // consecutive branches per burst; with rows of B doubles every burst straddled two sectors)
  const uint32_t pitch = (ctx->B + 7u) & ~7u;
  double* d_lnl = (double*)epa_scratch(ctx, 3, sizeof(double) * (size_t)Q * pitch);
  if (!d_lnl) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(chunk table)");
  ctx->lnl_pitch = pitch;
  // per-(query, 64-branch segment) maxima of the table: a by-product of the preplacement fast paths that lets
  // the dynamic rule read only the segments near a row's maximum (preplace.hip: seg_key, k_select_seg)
  const uint32_t nseg = (ctx->B + 63) / 64;
  ctx->segmax = nullptr;
  if (nseg <= 64 && ctx->heur_mode == 0) {
    ctx->segp = (nseg + 7u) & ~7u;
    ctx->segmax = (unsigned long long*)epa_scratch(ctx, 10, sizeof(unsigned long long) * (size_t)Q * ctx->segp);
    if (!ctx->segmax) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(segment maxima)");
    // zeroed by launch_preplace together with its status words and key counts (one kernel: ctx->segmax_zero_bytes)
    ctx->segmax_zero_bytes = sizeof(unsigned long long) * (size_t)Q * ctx->segp;
  }
  int rc = launch_preplace(ctx, d_codes, d_begin, d_span, Q, d_lnl, max_span);
  if (ctx->segmax_zero_bytes) {   // launch_preplace returned before its fill
    ctx->segmax_zero_bytes = 0;
    if (!rc) rc = epa_fail(ctx, EPA_ERR_HIP, "segment maxima not cleared");
  }
  if (!rc) rc = launch_select_begin(ctx, d_lnl, Q, threshold, d_pairs, max_pairs, d_span, rb, sp);
  ctx->lnl_pitch = 0;
  ctx->segmax = nullptr;
  // One read length (the windows can only be of max_span's class): pair list and Newton kernel can be queued right
  // here, guarded on the device by the same read-back block the host will look at in chunk_body_end -- no host round
  // trip between the selection and the thorough kernel.  OPT-IN (EPA_QUEUED_THOROUGH=1):
  // measured (profiles/r4_queued_thorough.txt) it buys nothing -- a chunk's preplacement and its Newton kernel each
  // need whole CUs, so they serialise on the device whichever is queued first -- and with HIP's default four
  // hardware queues the early Newton kernel blocks the other slots' chains that share its queue (8.5 -> 7.7 M/s).
  // Bitmap form: the pair list is written behind the read-back at once (k_emit_pairs refuses a list that would not fit),
  // and the statistics block + work counters of the chunk's Newton launch are cleared behind it -- both in the shadow
  // of the host's round trip for the candidate count (15 - 25 us) instead of after it (profiles/r5_step_timeline.txt)
  if (!rc && d_stats && sp->d_rb && sp->bitmap && max_pairs <= 0xffffffffull) {
    rc = launch_select_emit(ctx, sp);
    if (!rc) rc = epa_zero_async(ctx, d_stats, 128, epa_th_ctr(ctx), epa_th_ctr(ctx) ? 64 : 0);
    if (!rc) {
      ctx->clean_stats[ctx->bank] = d_stats;
      ctx->clean_ctr[ctx->bank] = epa_th_ctr(ctx) != nullptr;
    }
  }
  if (!rc && d_res && d_stats && sp->d_rb && sp->bitmap && max_pairs <= 0xffffffffull) {
    const bool off = !ctx->opt.queued_thorough;
    const int cls = epa_span_class(ctx->s, max_span);
    const bool eligible = !off && ctx->s == 4 && !ctx->generic_thorough && ctx->dna.ng == 1 && epa_th_ctr(ctx) &&
                          (cls <= 2 || cls == 10 || cls == 11);
    if (eligible) {
      rc = launch_select_emit(ctx, sp);
      if (!rc && ctx->clean_stats[ctx->bank] == d_stats) ctx->clean_stats[ctx->bank] = nullptr;
      else if (!rc) rc = epa_zero_async(ctx, d_stats, 128);
      if (!rc) {
        const int q = launch_thorough_queued(ctx, d_pairs, sp->d_rb, max_pairs, d_codes, d_begin, d_span, max_span, d_res, d_stats);
        if (q == -2) return EPA_ERR_HIP;
        sp->queued_cls = q;
      }
    }
  }
  return rc;
}

static int chunk_body_end(epa_ctx* ctx, SelectPending* sp, const uint8_t* d_codes, const uint32_t* d_begin,
                          const uint32_t* d_span, uint32_t max_span, epa_pair* d_pairs, epa_result* d_res,
                          unsigned long long* d_stats, uint64_t* n_out) {
  const uint32_t pitch = (ctx->B + 7u) & ~7u;
  ctx->lnl_pitch = pitch;   // a widened re-run of the selection reads the same table
  uint64_t n = 0;
  int rc = launch_select_end(ctx, sp, &n);  // syncs once
  ctx->lnl_pitch = 0;
  if (rc) return rc;
  rc = select_check_status(ctx, sp);
  if (rc) return rc;
  *n_out = n;
  // the queued launch ran iff all n pairs are of its class (no overflow / window error: checked above) -- the test
  // k_thorough_dna applied to the same block
  if (sp->queued_cls >= 0 && sp->rb[9 + sp->queued_cls] == (uint32_t)n) { ctx->cls_hist_pairs = 0; return EPA_OK; }
  if (ctx->clean_stats[ctx->bank] == d_stats && sp->queued_cls < 0) ctx->clean_stats[ctx->bank] = nullptr;   // cleared behind the selection
  else {
    ctx->clean_stats[ctx->bank] = nullptr;
    rc = epa_zero_async(ctx, d_stats, 128);
    if (rc) return rc;
  }
  if (n == 0) return EPA_OK;
  return launch_thorough(ctx, d_pairs, n, d_codes, d_begin, d_span, max_span, d_res, d_stats);
}

static int chunk_body(epa_ctx* ctx, const uint8_t* d_codes, const uint32_t* d_begin, const uint32_t* d_span,
                      uint32_t Q, uint32_t max_span, double threshold, epa_pair* d_pairs, epa_result* d_res,
                      uint64_t max_pairs, unsigned long long* d_stats, uint64_t* n_out) {
  uint32_t rb[64] = {};
  SelectPending sp;
  int rc = chunk_body_begin(ctx, d_codes, d_begin, d_span, Q, max_span, threshold, d_pairs, max_pairs, rb, &sp, d_res, d_stats);
  if (rc) return rc;
  return chunk_body_end(ctx, &sp, d_codes, d_begin, d_span, max_span, d_pairs, d_res, d_stats, n_out);
}

static int chunk_stats(epa_ctx* ctx, uint64_t n, const unsigned long long* hst, epa_thorough_stats* stats) {
  epa_xcd_feedback(ctx, n, hst);
  ctx->last_stats.pairs = n;
  ctx->last_stats.rounds = hst[0];
  ctx->last_stats.newton_evals = hst[1];
  ctx->last_stats.reverts = hst[2];
  if (stats) *stats = ctx->last_stats;
  if (hst[3])
    return epa_fail(ctx, EPA_ERR_NEG_INF,
                    "-INF logl at branch " + std::to_string((uint32_t)(hst[4] >> 32)) +
                        " with sequence " + std::to_string((uint32_t)(hst[4] & 0xffffffffu)));
  return EPA_OK;
}

static int thorough_supported(epa_ctx*) { return EPA_OK; }  // every model has a thorough kernel

extern "C" int epa_dev_place_chunk(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin,
                                   const uint32_t* win_span, uint32_t Q, uint32_t max_span,
                                   double threshold, epa_pair* pairs, epa_result* results,
                                   uint64_t max_pairs, uint64_t* n_pairs, epa_thorough_stats* stats) {
  if (!ctx || !q_codes || !win_begin || !win_span || !pairs || !results || !n_pairs)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null argument");
  *n_pairs = 0;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (Q == 0) return EPA_OK;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = thorough_supported(ctx);
  if (rc) return rc;
  rc = epa_dev_build_lookup(ctx);
  if (rc) return rc;
  if (max_span == 0) {  // not supplied: look at the spans (host array, or one small D2H copy)
    std::vector<uint32_t> buf;
    const uint32_t* hs = host_view(win_span, Q, buf, ctx->stream);
    if (!hs) return epa_fail(ctx, EPA_ERR_HIP, "cannot read window spans");
    for (uint32_t q = 0; q < Q; ++q) max_span = std::max(max_span, hs[q]);
  }
  if (max_span > ctx->W) max_span = ctx->W;
  const uint8_t* d_codes = epa_codes_to_device(ctx, q_codes, Q);
  const uint32_t* d_begin = (const uint32_t*)epa_to_device(ctx, 1, win_begin, sizeof(uint32_t) * Q);
  const uint32_t* d_span = (const uint32_t*)epa_to_device(ctx, 2, win_span, sizeof(uint32_t) * Q);
  if (!d_codes || !d_begin || !d_span) return epa_fail(ctx, EPA_ERR_HIP, "query upload failed");
  const bool pairs_dev = epa_is_device_ptr(pairs), res_dev = epa_is_device_ptr(results);
  epa_pair* d_pairs = pairs_dev ? pairs : (epa_pair*)epa_scratch(ctx, 4, sizeof(epa_pair) * max_pairs);
  epa_result* d_res = res_dev ? results : (epa_result*)epa_scratch(ctx, 5, sizeof(epa_result) * max_pairs);
  if (!d_pairs || !d_res) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(chunk buffers)");
  // thorough counters: second half of the context's 256-byte counter block
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(epa_th_ctr(ctx)) + 128);
  uint64_t n = 0;
  rc = chunk_body(ctx, d_codes, d_begin, d_span, Q, max_span, threshold, d_pairs, d_res, max_pairs, d_stats, &n);
  if (rc) return rc;
  *n_pairs = n;
  if (n == 0) return EPA_OK;
  unsigned long long hst[16];
  if (!pairs_dev)
    EPA_HIP(ctx, hipMemcpyAsync(pairs, d_pairs, sizeof(epa_pair) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (!res_dev)
    EPA_HIP(ctx, hipMemcpyAsync(results, d_res, sizeof(epa_result) * n, hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipMemcpyAsync(hst, d_stats, 128, hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return chunk_stats(ctx, n, hst, stats);
}

// ---------------------------------------------------------------------------------------------
// Double-buffered chunk pipeline (see include/epa_dev.h): stage -> launch -> finish per slot.
// ---------------------------------------------------------------------------------------------
static int slot_of(epa_ctx* ctx, int slot, ChunkSlot** out) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  if (slot < 0 || slot >= epa_ctx::N_SLOTS) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk pipeline: slot must be 0 .. 23");
  ChunkSlot& s = ctx->slots[slot];
  if (!ctx->copy_stream) EPA_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!ctx->down_stream) EPA_HIP(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
  if (!s.ev_up) {
    EPA_HIP(ctx, hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
    EPA_HIP(ctx, hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    EPA_HIP(ctx, hipEventCreateWithFlags(&s.ev_down, hipEventDisableTiming));
    EPA_HIP(ctx, hipEventCreateWithFlags(&s.ev_base, hipEventDisableTiming));
    EPA_HIP(ctx, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    EPA_HIP(ctx, hipHostMalloc((void**)&s.h_stats, 128, hipHostMallocDefault));
    EPA_HIP(ctx, hipHostMalloc((void**)&s.h_sel, 256, hipHostMallocDefault));
    EPA_HIP(ctx, hipMalloc((void**)&s.d_stats, 128));
  }
  *out = &s;
  return EPA_OK;
}

template <class T>
static int grow_dev(epa_ctx* ctx, T** p, size_t* have, size_t want_bytes) {
  if (want_bytes <= *have) return EPA_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *have = 0;
  const size_t sz = want_bytes + want_bytes / 4 + 1024;
  EPA_HIP(ctx, hipMalloc((void**)p, sz));
  *have = sz;
  return EPA_OK;
}
static int grow_pinned(epa_ctx* ctx, void** p, size_t* have, size_t want_bytes) {
  if (want_bytes <= *have) return EPA_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr; *have = 0;
  const size_t sz = want_bytes + want_bytes / 4 + 1024;
  EPA_HIP(ctx, hipHostMalloc(p, sz, hipHostMallocDefault));
  *have = sz;
  return EPA_OK;
}

extern "C" int epa_dev_chunk_stage(epa_ctx* ctx, int slot, const uint8_t* q_codes, const uint32_t* win_begin,
                                   const uint32_t* win_span, uint32_t Q) {
  ChunkSlot* s;
  int rc = slot_of(ctx, slot, &s);
  if (rc) return rc;
  if (!q_codes || !win_begin || !win_span || Q == 0) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_stage: null / empty chunk");
  if (s->state == 2 || s->state == 3 || s->state == 4) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_stage: the slot has an unfinished launch");
  if (s->g_left > 0) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_stage: members of this slot's group launch are not finished yet (their results live in its buffers)");
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  s->stride = ctx->code_stride ? ctx->code_stride : ctx->W;
  s->packed4 = ctx->code_packed4;
  const size_t row = s->packed4 ? ((size_t)s->stride + 1) / 2 : s->stride;
  s->codes_bytes = ((size_t)Q * row + 255) & ~(size_t)255;
  const size_t total = s->codes_bytes + 2 * sizeof(uint32_t) * (size_t)Q;
  const bool dev_in = epa_is_device_ptr(q_codes);
  if (dev_in != epa_is_device_ptr(win_begin) || dev_in != epa_is_device_ptr(win_span))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_stage: the three arrays must be all host or all device memory");
  if (dev_in) {
    // HBM-resident chunk: the slot reads the caller's arrays in place (no copy, no copy-stream
    // work); they must stay untouched until chunk_finish and be complete on the context's stream
    s->x_codes = q_codes; s->x_begin = win_begin; s->x_span = win_span;
    s->Q = Q;
    s->state = 1;
    return EPA_OK;
  }
  s->x_codes = nullptr;
  rc = grow_pinned(ctx, &s->h_in, &s->h_in_sz, total);
  if (!rc) rc = grow_dev(ctx, (char**)&s->d_in, &s->d_in_sz, total + 1024);
  if (rc) return rc;
  char* h = (char*)s->h_in;
  memcpy(h, q_codes, (size_t)Q * row);
  memcpy(h + s->codes_bytes, win_begin, sizeof(uint32_t) * (size_t)Q);
  memcpy(h + s->codes_bytes + sizeof(uint32_t) * (size_t)Q, win_span, sizeof(uint32_t) * (size_t)Q);
  EPA_HIP(ctx, hipMemcpyAsync(s->d_in, s->h_in, total, hipMemcpyHostToDevice, ctx->copy_stream));
  EPA_HIP(ctx, hipEventRecord(s->ev_up, ctx->copy_stream));
  s->Q = Q;
  s->state = 1;
  return EPA_OK;
}

namespace {
// the context's stream / scratch bank are the slot's for the duration of a pipeline call only
struct SlotScope {
  epa_ctx* c; hipStream_t st; uint32_t stride; bool packed;
  SlotScope(epa_ctx* ctx, ChunkSlot* s, int slot) : c(ctx), st(ctx->stream), stride(ctx->code_stride), packed(ctx->code_packed4) {
    ctx->stream = s->stream;
    ctx->bank = 1 + slot;
    ctx->code_stride = s->stride == ctx->W ? 0 : s->stride;   // the kernels read the layout the chunk was staged with
    ctx->code_packed4 = false;
  }
  ~SlotScope() { c->stream = st; c->bank = 0; c->code_stride = stride; c->code_packed4 = packed; }
};
}  // namespace

static int chunk_launch_begin_impl(epa_ctx* ctx, int slot, ChunkSlot* s, uint32_t max_span, double threshold,
                                   epa_pair* d_pairs, epa_result* d_results, uint64_t max_pairs, uint32_t flags) {
  int rc = EPA_OK;
  if (s->state != 1) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch: the slot holds no staged chunk");
  if ((d_pairs == nullptr) != (d_results == nullptr))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch: pass both result buffers or neither");
  if (d_pairs && (!epa_is_device_ptr(d_pairs) || !epa_is_device_ptr(d_results)))
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch: result buffers must be device memory");
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  rc = thorough_supported(ctx);
  if (!rc) rc = epa_dev_build_lookup(ctx);
  if (rc) return rc;
  const uint32_t Q = s->Q;
  if (max_span == 0 || max_span > ctx->W) max_span = ctx->W;
  if (!d_pairs) {
    if (s->cap < max_pairs) {
      if (s->d_pairs) (void)hipFree(s->d_pairs);
      if (s->d_res) (void)hipFree(s->d_res);
      s->d_pairs = nullptr; s->d_res = nullptr; s->cap = 0;
      EPA_HIP(ctx, hipMalloc((void**)&s->d_pairs, sizeof(epa_pair) * max_pairs));
      EPA_HIP(ctx, hipMalloc((void**)&s->d_res, sizeof(epa_result) * max_pairs));
      s->cap = max_pairs;
    }
    d_pairs = s->d_pairs;
    d_results = s->d_res;
  }
  // The slot's kernels go to the slot's own stream and scratch bank, so that the two chunks in
  // flight overlap on the device (chunk k + 1's preplacement and selection fill the tail of chunk
  // k's Newton kernel) and the wait for chunk k + 1's candidate count does not wait for chunk k's
  // kernels.  The slot stream starts behind what the caller's stream has queued so far (lookup
  // build; with device-resident results, the caller's reads of the result buffers).
  if (!(flags & EPA_CHUNK_HOST_ORDERED)) {
    EPA_HIP(ctx, hipEventRecord(s->ev_base, ctx->stream));
    EPA_HIP(ctx, hipStreamWaitEvent(s->stream, s->ev_base, 0));
  }
  if (!s->x_codes) EPA_HIP(ctx, hipStreamWaitEvent(s->stream, s->ev_up, 0));
  SlotScope scope(ctx, s, slot);
  const char* d = (const char*)s->d_in;
  const uint8_t* d_codes = s->x_codes ? s->x_codes : (const uint8_t*)d;
  const uint32_t* d_begin = s->x_codes ? s->x_begin : (const uint32_t*)(d + s->codes_bytes);
  const uint32_t* d_span = s->x_codes ? s->x_span : d_begin + Q;
  if (s->packed4) {
    size_t have = s->d_unpacked_sz;
    rc = grow_dev(ctx, &s->d_unpacked, &have, (size_t)Q * s->stride + 1024);
    s->d_unpacked_sz = have;
    if (rc) return rc;
    launch_unpack4(ctx->stream, d_codes, s->d_unpacked, Q, s->stride);
    d_codes = s->d_unpacked;
  }
  s->l_codes = d_codes; s->l_begin = d_begin; s->l_span = d_span;
  s->l_pairs = d_pairs; s->l_res = d_results; s->l_max_span = max_span; s->l_flags = flags;
  rc = chunk_body_begin(ctx, d_codes, d_begin, d_span, Q, max_span, threshold, d_pairs, max_pairs, s->h_sel, &s->sel,
                        d_results, s->d_stats);
  if (rc) return rc;   // the slot stays staged
  s->state = 3;
  return EPA_OK;
}

extern "C" int epa_dev_chunk_launch_begin(epa_ctx* ctx, int slot, uint32_t max_span, double threshold,
                                          epa_pair* d_pairs, epa_result* d_results, uint64_t max_pairs,
                                          uint32_t flags) {
  ChunkSlot* s;
  int rc = slot_of(ctx, slot, &s);
  if (rc) return rc;
  return chunk_launch_begin_impl(ctx, slot, s, max_span, threshold, d_pairs, d_results, max_pairs, flags);
}

// ---- group launches ---------------------------------------------------------------------------------
namespace {
struct GroupDesc {
  const uint8_t* codes[EPA_MAX_GROUP];
  const uint32_t* begin[EPA_MAX_GROUP];
  const uint32_t* span[EPA_MAX_GROUP];
  uint32_t qoff[EPA_MAX_GROUP + 1];   // first merged query index of every member; [n] = total
  uint32_t n, row;                    // members; bytes per code row (the same for all of them)
};

__device__ __forceinline__ int group_member(const uint32_t* qoff, int n, uint32_t q) {
  int m = 0;
  for (int i = 1; i < n; ++i) m += q >= qoff[i];
  return m;
}

// the members' staged arrays -> one chunk: codes rows, window begins, window spans back to back
__global__ void __launch_bounds__(256) k_concat_chunks(const GroupDesc g, uint8_t* __restrict__ codes, uint32_t* __restrict__ begin,
                                                       uint32_t* __restrict__ span) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t total_bytes = (uint64_t)g.qoff[g.n] * g.row;
  const uint64_t nwords = (total_bytes + 3) / 4;
  if (i < nwords) {
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
      const uint64_t b = i * 4 + k;
      if (b >= total_bytes) break;
      const uint32_t q = (uint32_t)(b / g.row);
      const int m = group_member(g.qoff, (int)g.n, q);
      v |= (uint32_t)g.codes[m][b - (uint64_t)g.qoff[m] * g.row] << (8 * k);
    }
    if (i * 4 + 4 <= total_bytes) reinterpret_cast<uint32_t*>(codes)[i] = v;
    else for (int k = 0; i * 4 + k < total_bytes; ++k) codes[i * 4 + k] = (uint8_t)(v >> (8 * k));
    return;
  }
  const uint64_t q = i - nwords;
  if (q >= g.qoff[g.n]) return;
  const int m = group_member(g.qoff, (int)g.n, (uint32_t)q);
  begin[q] = g.begin[m][q - g.qoff[m]];
  span[q] = g.span[m][q - g.qoff[m]];
}

// Stable partition of the group's n (pair, result) rows by member: member m's rows keep their order (branch-major,
// queries ascending -- the order of its own chunk's Work, src/core/Work.hpp:21-113) and get their chunk-local
// sequence ids back; goff[m] = first row of member m, goff[n_members] = n.  One 1024-thread workgroup (group
// launches exist for SMALL chunks: tens of thousands of rows; correct for any n).
__global__ void __launch_bounds__(1024) k_split_group(const epa_pair* __restrict__ pairs, const epa_result* __restrict__ res,
                                                      uint32_t n, const GroupDesc g, epa_pair* __restrict__ o_pairs,
                                                      epa_result* __restrict__ o_res, uint32_t* __restrict__ goff) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t base[EPA_MAX_GROUP + 1];
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
  uint32_t cnt[EPA_MAX_GROUP] = {};
  for (uint32_t i = lo; i < hi; ++i) {
    const int m = group_member(g.qoff, (int)g.n, pairs[i].seq_id);
#pragma unroll
    for (int k = 0; k < EPA_MAX_GROUP; ++k) cnt[k] += (k == m);
  }
  // per member: exclusive prefix of the threads' counts (wave scan + wave totals), then the members' bases
  uint32_t pre[EPA_MAX_GROUP];
  uint32_t run = 0;
  for (int k = 0; k < EPA_MAX_GROUP; ++k) {
    uint32_t v = cnt[k];
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(v, d, 64);
      if ((int)lane >= d) v += u;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < 16; ++w) { const uint32_t x = wsum[w]; total += x; if (w < (int)wv) before += x; }
    pre[k] = run + before + v - cnt[k];
    if (t == 0) base[k] = run;
    run += total;
    __syncthreads();
  }
  if (t == 0) {
    for (int k = 0; k <= (int)g.n && k < EPA_MAX_GROUP; ++k) goff[k] = base[k];
    goff[g.n] = n;
  }
  for (uint32_t i = lo; i < hi; ++i) {
    epa_pair p = pairs[i];
    const int m = group_member(g.qoff, (int)g.n, p.seq_id);
    uint32_t dst = 0;
#pragma unroll
    for (int k = 0; k < EPA_MAX_GROUP; ++k)
      if (k == m) dst = pre[k]++;
    p.seq_id -= g.qoff[m];
    o_pairs[dst] = p;
    o_res[dst] = res[i];
  }
}

void group_restore_leader(ChunkSlot* L) {
  L->Q = L->own_Q;
  L->x_codes = L->own_x_codes; L->x_begin = L->own_x_begin; L->x_span = L->own_x_span;
}

GroupDesc group_desc(epa_ctx* ctx, const ChunkSlot* L) {
  GroupDesc g{};
  g.n = (uint32_t)L->g_n;
  for (int i = 0; i <= L->g_n; ++i) g.qoff[i] = L->g_qoff[i];
  (void)ctx;
  return g;
}
}  // namespace

extern "C" int epa_dev_chunk_launch_many_begin(epa_ctx* ctx, const int* slots, int n_slots, uint32_t max_span,
                                               double threshold, uint64_t max_pairs, uint32_t flags) {
  if (!ctx || !slots || n_slots < 1 || n_slots > EPA_MAX_GROUP)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_many: 1 .. 8 slots");
  ChunkSlot* m[EPA_MAX_GROUP];
  for (int i = 0; i < n_slots; ++i) {
    int rc = slot_of(ctx, slots[i], &m[i]);
    if (rc) return rc;
    for (int j = 0; j < i; ++j)
      if (slots[j] == slots[i]) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_many: a slot is listed twice");
    if (m[i]->state != 1) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_many: slot " + std::to_string(slots[i]) + " holds no staged chunk");
    if (m[i]->stride != m[0]->stride || m[i]->packed4 != m[0]->packed4)
      return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_many: the chunks were staged with different query layouts");
  }
  ChunkSlot* L = m[0];
  if (L->g_left > 0) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_many: members of the leader's previous group are not finished yet");
  if (n_slots == 1) return chunk_launch_begin_impl(ctx, slots[0], L, max_span, threshold, nullptr, nullptr, max_pairs, flags);
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  GroupDesc g{};
  g.n = (uint32_t)n_slots;
  g.row = L->packed4 ? (L->stride + 1) / 2 : L->stride;
  uint64_t total = 0;
  for (int i = 0; i < n_slots; ++i) {
    const char* d = (const char*)m[i]->d_in;
    g.codes[i] = m[i]->x_codes ? m[i]->x_codes : (const uint8_t*)d;
    g.begin[i] = m[i]->x_codes ? m[i]->x_begin : (const uint32_t*)(d + m[i]->codes_bytes);
    g.span[i] = m[i]->x_codes ? m[i]->x_span : g.begin[i] + m[i]->Q;
    g.qoff[i] = (uint32_t)total;
    total += m[i]->Q;
  }
  if (total > 0x7fffffffull) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_many: too many queries in one group");
  g.qoff[n_slots] = (uint32_t)total;
  const size_t cbytes = ((size_t)total * g.row + 255) & ~(size_t)255;
  int rc = grow_dev(ctx, (char**)&L->d_merge, &L->d_merge_sz, cbytes + 2 * sizeof(uint32_t) * (size_t)total + 1024);
  if (rc) return rc;
  if (!L->d_goff) {
    EPA_HIP(ctx, hipMalloc((void**)&L->d_goff, sizeof(uint32_t) * (EPA_MAX_GROUP + 1)));
    EPA_HIP(ctx, hipHostMalloc((void**)&L->h_goff, sizeof(uint32_t) * (EPA_MAX_GROUP + 1), hipHostMallocDefault));
  }
  // the merge runs on the leader's stream, behind the caller's stream (device-resident chunks) and every member's upload
  if (!(flags & EPA_CHUNK_HOST_ORDERED)) {
    EPA_HIP(ctx, hipEventRecord(L->ev_base, ctx->stream));
    EPA_HIP(ctx, hipStreamWaitEvent(L->stream, L->ev_base, 0));
  }
  for (int i = 0; i < n_slots; ++i)
    if (!m[i]->x_codes) EPA_HIP(ctx, hipStreamWaitEvent(L->stream, m[i]->ev_up, 0));
  uint8_t* mc = (uint8_t*)L->d_merge;
  uint32_t* mb = (uint32_t*)(mc + cbytes);
  uint32_t* ms = mb + total;
  const uint64_t threads = ((uint64_t)total * g.row + 3) / 4 + total;
  hipLaunchKernelGGL(k_concat_chunks, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, L->stream, g, mc, mb, ms);
  EPA_HIP(ctx, hipGetLastError());
  // the leader now stands for the merged chunk (read in place, like an HBM-resident one)
  L->own_Q = L->Q; L->own_x_codes = L->x_codes; L->own_x_begin = L->x_begin; L->own_x_span = L->x_span;
  L->Q = (uint32_t)total; L->x_codes = mc; L->x_begin = mb; L->x_span = ms;
  rc = chunk_launch_begin_impl(ctx, slots[0], L, max_span, threshold, nullptr, nullptr, max_pairs, flags);
  if (rc) { group_restore_leader(L); return rc; }   // every member stays staged
  L->g_n = n_slots;
  for (int i = 0; i <= n_slots; ++i) L->g_qoff[i] = g.qoff[i];
  for (int i = 0; i < n_slots; ++i) {
    L->g_slots[i] = slots[i];
    m[i]->leader = slots[0];
    m[i]->g_index = i;
    if (i) m[i]->state = 4;
  }
  return EPA_OK;
}

extern "C" int epa_dev_chunk_launch_many(epa_ctx* ctx, const int* slots, int n_slots, uint32_t max_span, double threshold,
                                         uint64_t max_pairs, uint32_t flags) {
  int rc = epa_dev_chunk_launch_many_begin(ctx, slots, n_slots, max_span, threshold, max_pairs, flags);
  if (rc) return rc;
  return epa_dev_chunk_launch_end(ctx, slots[0]);
}

extern "C" int epa_dev_chunk_launch_end(epa_ctx* ctx, int slot) {
  ChunkSlot* s;
  int rc = slot_of(ctx, slot, &s);
  if (rc) return rc;
  if (s->state == 4) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_end: the slot is a member of a group launch: end the group's first slot");
  if (s->state != 3) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_launch_end: no launch was begun on the slot");
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t base = ctx->stream;
  uint64_t n = 0;
  epa_pair* d_pairs = s->l_pairs;
  epa_result* d_results = s->l_res;
  SlotScope scope(ctx, s, slot);
  rc = chunk_body_end(ctx, &s->sel, s->l_codes, s->l_begin, s->l_span, s->l_max_span, d_pairs, d_results, s->d_stats, &n);
  // Candidate overflow and the window-validation errors are found before any thorough kernel is queued: the
  // slot stays staged (launch again with a larger max_pairs).  Any other failure may leave kernels of this
  // chunk in flight on the slot's stream: wait for them and free the slot, so that it cannot be re-staged or
  // re-launched (its result buffers re-allocated) under running kernels.
  const int gn = s->g_n;
  auto group_release = [&](int member_state) {   // the group is over (error): members back to `member_state`
    if (gn <= 1) return;
    group_restore_leader(s);
    for (int i = 0; i < gn; ++i) {
      ChunkSlot& mm = ctx->slots[s->g_slots[i]];
      mm.leader = -1;
      if (i) mm.state = member_state;
    }
    s->g_n = 0;
    s->g_left = 0;   // (a failure after the members were marked must not leave the leader refusing chunk_stage for ever)
  };
  auto abandon = [&](int code) {
    (void)hipStreamSynchronize(s->stream);
    s->state = 0;
    group_release(0);
    return code;
  };
  if (rc == EPA_ERR_PAIR_OVERFLOW || rc == EPA_ERR_QUERY_ALL_GAP || rc == EPA_ERR_QUERY_WIDTH) {
    s->state = 1;
    group_release(1);   // every member stays staged: launch the group again (larger max_pairs) or one by one
    return rc;
  }
  if (rc) return abandon(rc);
  s->n = n;
  if (gn > 1) {
    // regroup the rows by member (stable), chunk-local sequence ids back
    if (s->g_cap < n) {
      if (s->d_gpairs) (void)hipFree(s->d_gpairs);
      if (s->d_gres) (void)hipFree(s->d_gres);
      s->d_gpairs = nullptr; s->d_gres = nullptr; s->g_cap = 0;
      const size_t want = n + n / 4 + 1024;
      if (hipMalloc((void**)&s->d_gpairs, sizeof(epa_pair) * want) != hipSuccess ||
          hipMalloc((void**)&s->d_gres, sizeof(epa_result) * want) != hipSuccess)
        return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(group rows)"));
      s->g_cap = want;
    }
    GroupDesc g = group_desc(ctx, s);
    hipLaunchKernelGGL(k_split_group, dim3(1), dim3(1024), 0, ctx->stream, d_pairs, d_results, (uint32_t)n, g, s->d_gpairs,
                       s->d_gres, s->d_goff);
    if (hipGetLastError() != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "k_split_group launch"));
    d_pairs = s->d_gpairs;
    d_results = s->d_gres;
  }
  if (hipEventRecord(s->ev_done, ctx->stream) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipEventRecord(chunk done)"));
  if (hipStreamWaitEvent(ctx->down_stream, s->ev_done, 0) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipStreamWaitEvent(chunk done)"));
  if (gn > 1) {
    if (hipMemcpyAsync(s->h_goff, s->d_goff, sizeof(uint32_t) * (EPA_MAX_GROUP + 1), hipMemcpyDeviceToHost, ctx->down_stream) != hipSuccess)
      return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipMemcpyAsync(group offsets)"));
    group_restore_leader(s);
    s->g_left = gn;
    for (int i = 1; i < gn; ++i) {
      ChunkSlot& mm = ctx->slots[s->g_slots[i]];
      mm.state = 2;
      mm.l_flags = s->l_flags;
    }
  }
  if (s->l_flags & EPA_CHUNK_NO_D2H) {
    s->out_pairs = d_pairs;
    s->out_res = d_results;
    // device-resident results are consumed on the caller's stream
    if (!(s->l_flags & EPA_CHUNK_HOST_ORDERED)) if (hipStreamWaitEvent(base, s->ev_done, 0) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipStreamWaitEvent (chunk_launch_end)"));
  } else {
    const size_t off_r = (sizeof(epa_pair) * n + 255) & ~(size_t)255;
    rc = grow_pinned(ctx, &s->h_out, &s->h_out_sz, off_r + sizeof(epa_result) * n);
    if (rc) return abandon(rc);
    if (n) {
      if (hipMemcpyAsync(s->h_out, d_pairs, sizeof(epa_pair) * n, hipMemcpyDeviceToHost, ctx->down_stream) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipMemcpyAsync (chunk_launch_end)"));
      if (hipMemcpyAsync((char*)s->h_out + off_r, d_results, sizeof(epa_result) * n, hipMemcpyDeviceToHost,
                                  ctx->down_stream) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipMemcpyAsync (chunk_launch_end)"));
    }
    s->out_pairs = (const epa_pair*)s->h_out;
    s->out_res = (const epa_result*)((char*)s->h_out + off_r);
  }
  if (hipMemcpyAsync(s->h_stats, s->d_stats, 128, hipMemcpyDeviceToHost, ctx->down_stream) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipMemcpyAsync (chunk_launch_end)"));
  if (hipEventRecord(s->ev_down, ctx->down_stream) != hipSuccess) return abandon(epa_fail(ctx, EPA_ERR_HIP, "hipEventRecord (chunk_launch_end)"));
  s->state = 2;
  return EPA_OK;
}

extern "C" int epa_dev_chunk_launch(epa_ctx* ctx, int slot, uint32_t max_span, double threshold,
                                    epa_pair* d_pairs, epa_result* d_results, uint64_t max_pairs,
                                    uint32_t flags) {
  int rc = epa_dev_chunk_launch_begin(ctx, slot, max_span, threshold, d_pairs, d_results, max_pairs, flags);
  if (rc) return rc;
  return epa_dev_chunk_launch_end(ctx, slot);
}

extern "C" int epa_dev_chunk_finish(epa_ctx* ctx, int slot, const epa_pair** pairs, const epa_result** results,
                                    uint64_t* n_pairs, epa_thorough_stats* stats) {
  ChunkSlot* s;
  int rc = slot_of(ctx, slot, &s);
  if (rc) return rc;
  if (s->state != 2) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "chunk_finish: the slot has no launch in flight");
  if (s->leader >= 0) {   // member of a group launch: its rows are a range of the leader's regrouped buffers
    ChunkSlot* L = &ctx->slots[s->leader];
    EPA_HIP(ctx, hipEventSynchronize(L->ev_down));
    const uint32_t lo = L->h_goff[s->g_index], hi = L->h_goff[s->g_index + 1];
    const uint64_t group_rows = L->n;   // (the leader's own n is replaced by its member count below)
    s->n = hi - lo;
    s->out_pairs = L->out_pairs + lo;
    s->out_res = L->out_res + lo;
    s->state = 0;
    s->leader = -1;
    if (--L->g_left == 0) L->g_n = 0;
    if (pairs) *pairs = s->out_pairs;
    if (results) *results = s->out_res;
    if (n_pairs) *n_pairs = s->n;
    // the Newton launch was ONE launch for the group: its counters are reported with the group's first slot
    if (s->g_index == 0) {
      const int rc2 = chunk_stats(ctx, group_rows, L->h_stats, stats);
      ctx->last_stats.pairs = s->n;
      if (stats) stats->pairs = s->n;
      return rc2;
    }
    ctx->last_stats = epa_thorough_stats{};
    ctx->last_stats.pairs = s->n;
    if (stats) *stats = ctx->last_stats;
    return EPA_OK;
  }
  EPA_HIP(ctx, hipEventSynchronize(s->ev_down));
  s->state = 0;
  if (pairs) *pairs = s->out_pairs;
  if (results) *results = s->out_res;
  if (n_pairs) *n_pairs = s->n;
  return chunk_stats(ctx, s->n, s->h_stats, stats);
}

// =============================================================================================
// --no-heur: all B x Q pairs on the device, then LWR + filter per query on the device.
// =============================================================================================
__global__ void __launch_bounds__(256) k_all_pairs(epa_pair* __restrict__ pairs, uint32_t B, uint32_t Q) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)B * Q) return;
  const uint32_t b = (uint32_t)(i / Q);
  pairs[i].branch_id = b;                          // branch-major: Work's iteration order
  pairs[i].seq_id = (uint32_t)(i - (uint64_t)b * Q);
}

// One 256-thread workgroup per query.  res is branch-major: the placement of (b, q) is res[b*Q+q].
// compute_and_set_lwr over all B placements, then the best placements one by one (largest lnL
// first, ties: lowest branch id) until filter() would stop.
__global__ void __launch_bounds__(256) k_lwr_filter(const epa_result* __restrict__ res, uint32_t B, uint32_t Q,
                                                   double min_lwr, int acc, uint32_t keep_min, uint32_t keep_max,
                                                   epa_pair* __restrict__ out_pairs,
                                                   epa_result* __restrict__ out_res,
                                                   double* __restrict__ out_lwr,
                                                   uint32_t* __restrict__ counts) {
  __shared__ double s_val[2][4];
  __shared__ uint32_t s_idx[2][4];
  __shared__ uint32_t s_taken[64];
  const uint32_t q = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  auto lnl_of = [&](uint32_t b) { return res[(size_t)b * Q + q].lnl; };
  auto wmax = [](double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
  };
  auto wadd = [](double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  int ph = 0;
  double mx = -INFINITY;
  for (uint32_t b = t; b < B; b += 256) mx = fmax(mx, lnl_of(b));
  mx = wmax(mx);
  if (lane == 0) s_val[ph][wv] = mx;
  __syncthreads();
  mx = fmax(fmax(s_val[ph][0], s_val[ph][1]), fmax(s_val[ph][2], s_val[ph][3]));
  ph ^= 1;
  double tot = 0.0;
  for (uint32_t b = t; b < B; b += 256) tot += exp(lnl_of(b) - mx);
  tot = wadd(tot);
  if (lane == 0) s_val[ph][wv] = tot;
  __syncthreads();
  tot = (s_val[ph][0] + s_val[ph][1]) + (s_val[ph][2] + s_val[ph][3]);
  ph ^= 1;
  uint32_t taken = 0;
  double sum = 0.0;
  const uint32_t limit = min(B, keep_max);
  for (;;) {
    if (taken >= limit) break;
    if (acc && !(sum < min_lwr) && taken + 1 >= keep_min) break;  // accumulated mode: quota reached
    double best = -INFINITY;
    uint32_t bi = 0xffffffffu;
    for (uint32_t b = t; b < B; b += 256) {
      bool used = false;
      for (uint32_t k = 0; k < taken; ++k) used |= s_taken[k] == b;
      const double v = lnl_of(b);
      if (!used && v > best) { best = v; bi = b; }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const uint32_t oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { s_val[ph][wv] = best; s_idx[ph][wv] = bi; }
    __syncthreads();
    best = s_val[ph][0];
    bi = s_idx[ph][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const double ob = s_val[ph][w];
      const uint32_t oi = s_idx[ph][w];
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    ph ^= 1;
    if (bi == 0xffffffffu) break;
    const double lw = exp(best - mx) / tot;
    if (!acc) {
      // discard_by_support_threshold: keep lwr > thresh, but at least keep_min
      if (!(lw > min_lwr) && taken >= keep_min) break;
    } else {
      // discard_by_accumulated_threshold: while sum < thresh take; top-up to keep_min - 1 (quirk)
      if (!(sum < min_lwr) && taken + 1 >= keep_min) break;
      sum += lw;
    }
    if (t == 0) {
      const size_t o = (size_t)q * keep_max + taken;
      out_pairs[o].branch_id = bi;
      out_pairs[o].seq_id = q;
      out_res[o] = res[(size_t)bi * Q + q];
      out_lwr[o] = lw;
      s_taken[taken] = bi;
    }
    ++taken;
    __syncthreads();  // s_taken visible to the next scan
  }
  if (t == 0) counts[q] = taken;
}

extern "C" int epa_dev_place_all(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin,
                                 const uint32_t* win_span, uint32_t Q, uint32_t max_span, double min_lwr,
                                 int acc_threshold, uint32_t filter_min, uint32_t filter_max,
                                 epa_pair* pairs, epa_result* results, double* lwr, uint32_t* counts,
                                 epa_thorough_stats* stats) {
  if (!ctx || !q_codes || !win_begin || !win_span || !pairs || !results || !lwr || !counts)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null argument");
  if (filter_min < 1 || filter_max < filter_min || filter_max > 64)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG, "place_all: need 1 <= filter_min <= filter_max <= 64");
  if (stats) memset(stats, 0, sizeof(*stats));
  if (Q == 0) return EPA_OK;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = epa_dev_build_lookup(ctx);  // starting vectors of the thorough kernels
  if (rc) return rc;
  std::vector<uint32_t> hb_buf, hs_buf;
  const uint32_t* hb = host_view(win_begin, Q, hb_buf, ctx->stream);
  const uint32_t* hs = host_view(win_span, Q, hs_buf, ctx->stream);
  if (!hb || !hs) return epa_fail(ctx, EPA_ERR_HIP, "cannot read window arrays");
  uint32_t ms = 0;
  rc = check_windows(ctx, hb, hs, Q, &ms);
  if (rc) return rc;
  if (max_span == 0 || max_span < ms) max_span = ms;
  const uint64_t n = (uint64_t)ctx->B * Q;
  if (n > 0xffffffffull) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "place_all: more than 2^32 pairs per chunk");
  const uint8_t* d_codes = epa_codes_to_device(ctx, q_codes, Q);
  const uint32_t* d_begin = (const uint32_t*)epa_to_device(ctx, 1, win_begin, sizeof(uint32_t) * Q);
  const uint32_t* d_span = (const uint32_t*)epa_to_device(ctx, 2, win_span, sizeof(uint32_t) * Q);
  if (!d_codes || !d_begin || !d_span) return epa_fail(ctx, EPA_ERR_HIP, "query upload failed");
  epa_pair* d_all = (epa_pair*)epa_scratch(ctx, 4, sizeof(epa_pair) * n);
  epa_result* d_res = (epa_result*)epa_scratch(ctx, 5, sizeof(epa_result) * n);
  unsigned long long* d_stats = (unsigned long long*)epa_scratch(ctx, 6, 256);
  if (!d_all || !d_res || !d_stats) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(all-pairs buffers)");
  ctx->d_status = nullptr;
  EPA_HIP(ctx, hipMemsetAsync(d_stats, 0, 256, ctx->stream));
  hipLaunchKernelGGL(k_all_pairs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_all, ctx->B, Q);
  ctx->cls_hist_pairs = 0;
  rc = launch_thorough(ctx, d_all, n, d_codes, d_begin, d_span, max_span, d_res, d_stats);
  if (rc) return rc;
  // filtered output: Q x filter_max slots, staged in scratch 3 when the caller's buffers are on the host
  const size_t slots = (size_t)Q * filter_max;
  const bool p_dev = epa_is_device_ptr(pairs), r_dev = epa_is_device_ptr(results), l_dev = epa_is_device_ptr(lwr),
             c_dev = epa_is_device_ptr(counts);
  const size_t off_r = (sizeof(epa_pair) * slots + 255) & ~(size_t)255;
  const size_t off_l = off_r + ((sizeof(epa_result) * slots + 255) & ~(size_t)255);
  const size_t off_c = off_l + ((sizeof(double) * slots + 255) & ~(size_t)255);
  char* stage = (char*)epa_scratch(ctx, 3, off_c + sizeof(uint32_t) * Q);
  if (!stage) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(filter output)");
  epa_pair* o_p = p_dev ? pairs : (epa_pair*)stage;
  epa_result* o_r = r_dev ? results : (epa_result*)(stage + off_r);
  double* o_l = l_dev ? lwr : (double*)(stage + off_l);
  uint32_t* o_c = c_dev ? counts : (uint32_t*)(stage + off_c);
  hipLaunchKernelGGL(k_lwr_filter, dim3(Q), dim3(256), 0, ctx->stream, d_res, ctx->B, Q, min_lwr, acc_threshold,
                     filter_min, filter_max, o_p, o_r, o_l, o_c);
  if (!p_dev) EPA_HIP(ctx, hipMemcpyAsync(pairs, o_p, sizeof(epa_pair) * slots, hipMemcpyDeviceToHost, ctx->stream));
  if (!r_dev) EPA_HIP(ctx, hipMemcpyAsync(results, o_r, sizeof(epa_result) * slots, hipMemcpyDeviceToHost, ctx->stream));
  if (!l_dev) EPA_HIP(ctx, hipMemcpyAsync(lwr, o_l, sizeof(double) * slots, hipMemcpyDeviceToHost, ctx->stream));
  if (!c_dev) EPA_HIP(ctx, hipMemcpyAsync(counts, o_c, sizeof(uint32_t) * Q, hipMemcpyDeviceToHost, ctx->stream));
  unsigned long long hst[16];
  EPA_HIP(ctx, hipMemcpyAsync(hst, d_stats, 128, hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  EPA_HIP(ctx, hipGetLastError());
  epa_xcd_feedback(ctx, n, hst);
  ctx->last_stats.pairs = n;
  ctx->last_stats.rounds = hst[0];
  ctx->last_stats.newton_evals = hst[1];
  ctx->last_stats.reverts = hst[2];
  if (stats) *stats = ctx->last_stats;
  if (hst[3])
    return epa_fail(ctx, EPA_ERR_NEG_INF,
                    "-INF logl at branch " + std::to_string((uint32_t)(hst[4] >> 32)) +
                        " with sequence " + std::to_string((uint32_t)(hst[4] & 0xffffffffu)));
  return EPA_OK;
}

// --no-heur in the one-process-per-GPU mode: the filtered placements of epa_dev_place_all as gather rows.  Query q
// (count k) owns rows [off_q, off_q + 2 k): per kept placement, best first, its (branch, GLOBAL sequence id, lnL,
// pendant, distal) row followed by a row {EPA_ROW_LWR, sequence id, lnl = its like-weight ratio} -- the LWR of a
// --no-heur placement is normalised over ALL B branches (src/core/place.cpp:238 on the full Work), so it cannot be
// recomputed from the rows that travel.  One 1024-thread workgroup: counts -> offsets (wave scan) -> rows.
__global__ void __launch_bounds__(1024) k_all_rows(const epa_pair* __restrict__ pairs, const epa_result* __restrict__ res,
                                                   const double* __restrict__ lwr, const uint32_t* __restrict__ counts, uint32_t Q,
                                                   uint32_t fmax, uint32_t seq_offset, epa_row* __restrict__ rows,
                                                   unsigned long long* __restrict__ n_rows) {
  __shared__ uint32_t wsum[16];
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t per = (Q + 1023) / 1024;
  const uint32_t lo = t * per < Q ? t * per : Q, hi = lo + per < Q ? lo + per : Q;
  uint32_t mine = 0;
  for (uint32_t q = lo; q < hi; ++q) mine += 2u * counts[q];
  uint32_t v = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d, 64);
    if ((int)lane >= d) v += u;
  }
  if (lane == 63) wsum[wv] = v;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < 16; ++w) { const uint32_t x = wsum[w]; total += x; if (w < (int)wv) before += x; }
  uint32_t o = before + v - mine;
  if (t == 0) *n_rows = total;
  for (uint32_t q = lo; q < hi; ++q) {
    const uint32_t k = counts[q];
    for (uint32_t i = 0; i < k; ++i) {
      const size_t src = (size_t)q * fmax + i;
      epa_row r;
      r.branch_id = pairs[src].branch_id;
      r.seq_id = pairs[src].seq_id + seq_offset;
      r.lnl = res[src].lnl; r.pendant_length = res[src].pendant_length; r.distal_length = res[src].distal_length;
      rows[o++] = r;
      epa_row w;
      w.branch_id = EPA_ROW_LWR; w.seq_id = r.seq_id; w.lnl = lwr[src]; w.pendant_length = 0.0; w.distal_length = 0.0;
      rows[o++] = w;
    }
  }
}

extern "C" int epa_dev_place_all_rows(epa_ctx* ctx, const uint8_t* q_codes, const uint32_t* win_begin, const uint32_t* win_span,
                                      uint32_t Q, uint32_t max_span, double min_lwr, int acc_threshold, uint32_t filter_min,
                                      uint32_t filter_max, uint32_t seq_offset, const epa_row** d_rows, uint64_t* n_rows,
                                      epa_thorough_stats* stats) {
  if (!ctx || !d_rows || !n_rows) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "null argument");
  *d_rows = nullptr;
  *n_rows = 0;
  if (Q == 0) return EPA_OK;
  if (filter_max < 1 || filter_max > 64) return epa_fail(ctx, EPA_ERR_INVALID_ARG, "place_all: need 1 <= filter_min <= filter_max <= 64");
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  const size_t slots = (size_t)Q * filter_max;
  const size_t off_r = (sizeof(epa_pair) * slots + 255) & ~(size_t)255;
  const size_t off_l = off_r + ((sizeof(epa_result) * slots + 255) & ~(size_t)255);
  const size_t off_c = off_l + ((sizeof(double) * slots + 255) & ~(size_t)255);
  const size_t off_n = off_c + ((sizeof(uint32_t) * Q + 255) & ~(size_t)255);
  char* out = (char*)epa_scratch(ctx, 11, off_n + 256);
  epa_row* rows = (epa_row*)epa_scratch(ctx, 12, sizeof(epa_row) * 2 * slots);
  if (!out || !rows) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(place_all rows)");
  int rc = epa_dev_place_all(ctx, q_codes, win_begin, win_span, Q, max_span, min_lwr, acc_threshold, filter_min, filter_max,
                             (epa_pair*)out, (epa_result*)(out + off_r), (double*)(out + off_l), (uint32_t*)(out + off_c), stats);
  if (rc) return rc;
  unsigned long long* d_n = (unsigned long long*)(out + off_n);
  hipLaunchKernelGGL(k_all_rows, dim3(1), dim3(1024), 0, ctx->stream, (const epa_pair*)out, (const epa_result*)(out + off_r),
                     (const double*)(out + off_l), (const uint32_t*)(out + off_c), Q, filter_max, seq_offset, rows, d_n);
  EPA_HIP(ctx, hipGetLastError());
  unsigned long long h_n = 0;
  EPA_HIP(ctx, hipMemcpyAsync(&h_n, d_n, sizeof(h_n), hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *d_rows = rows;
  *n_rows = h_n;
  return EPA_OK;
}

extern "C" int epa_dev_mem_info(epa_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!ctx) return EPA_ERR_INVALID_ARG;
  EPA_HIP(ctx, hipSetDevice(ctx->device));
  size_t fr = 0, tot = 0;
  EPA_HIP(ctx, hipMemGetInfo(&fr, &tot));
  if (free_bytes) *free_bytes = fr;
  if (total_bytes) *total_bytes = tot;
  return EPA_OK;
}

extern "C" int epa_dev_xcd_shares(const epa_ctx* ctx, double shares[8]) {
  if (!ctx || !shares) return EPA_ERR_INVALID_ARG;
  for (int x = 0; x < 8; ++x) shares[x] = (double)(ctx->xcd_cum[x + 1] - ctx->xcd_cum[x]) / (double)(1u << 20);
  return EPA_OK;
}

extern "C" double epa_dev_last_sclk_mhz(const epa_ctx* ctx) { return ctx ? ctx->last_sclk_mhz : 0.0; }

extern "C" double epa_dev_last_kernel_ms(const epa_ctx* ctx, const char* which) {
  if (!ctx || !which) return -1.0;
  const EvTimer* t = nullptr;
  if (!strcmp(which, "preplace")) t = &ctx->t_bank[ctx->t_last[epa_ctx::T_PREPLACE]][epa_ctx::T_PREPLACE];
  else if (!strcmp(which, "thorough")) t = &ctx->t_bank[ctx->t_last[epa_ctx::T_THOROUGH]][epa_ctx::T_THOROUGH];
  else if (!strcmp(which, "lookup")) t = &ctx->t_lookup;
  else if (!strcmp(which, "select")) t = &ctx->t_bank[ctx->t_last[epa_ctx::T_SELECT]][epa_ctx::T_SELECT];
  if (!t || !t->valid) return -1.0;
  if (hipEventSynchronize(t->b) != hipSuccess) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, t->a, t->b) != hipSuccess) return -1.0;
  return (double)ms;
}
