// Hot loop 1 on device: preplacement gather-sum + candidate selection.
//
// k_preplace replaces Lookup_Store::sum_precomputed_sitelk inside place()
// (src/core/Lookup_Store.hpp:110-141, src/core/place.cpp:65-91):
//     lnl[q][b] = sum_{site in window(q)} T[b][site][code(q, site)]
// with the reference's association order ((a0+a1)+(a2+a3) per group of 4, then singles).
//
// Mapping (MI355X): queries are sorted by window start on the host and cut into groups of
// <= 256 whose window starts lie within SPREAD sites.  A workgroup = one query group x one
// tile of NB branches.  Per (branch, 160-site chunk) the 256 x ncols slice of T that the whole
// group can touch is staged through LDS once (coalesced 16 B/lane loads), every thread then
// gathers its own query's values with ds_read_b64.  Query codes live in registers (packed
// byte offsets), partial sums in LDS, so the only HBM traffic is T-slices in (L2/MALL resident:
// T is 196 MB at cfg2), codes in, and the Q x B table out.
#include "epa_dev_internal.hpp"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <numeric>

namespace {

constexpr int GQ = 256;      // queries (threads) per group
constexpr int CH = 160;      // sites per chunk (multiple of 4)
constexpr int SPREAD = 96;   // max spread of window starts inside a group
constexpr int TROWS = CH + SPREAD;  // rows of T staged per (branch, chunk)
constexpr int NB = 16;       // branches per workgroup
constexpr int CW = CH / 4;   // packed code words per chunk

struct Group {
  uint32_t start;     // first index into perm[]
  uint32_t count;
  uint32_t min_begin;
  uint32_t max_span;
};

template <int NCOLS>
__global__ void __launch_bounds__(GQ) k_preplace(const double* __restrict__ lookup,
                                                 const uint8_t* __restrict__ codes,
                                                 const uint32_t* __restrict__ win_begin,
                                                 const uint32_t* __restrict__ win_span,
                                                 const uint32_t* __restrict__ perm,
                                                 const Group* __restrict__ groups, uint32_t W,
                                                 uint32_t B, size_t codes_bytes,
                                                 double* __restrict__ lnl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* tile = reinterpret_cast<double*>(smem);                  // [TROWS][NCOLS]
  double* accs = tile + (size_t)TROWS * NCOLS;                     // [NB][GQ]
  const Group g = groups[blockIdx.x];
  const uint32_t b0 = blockIdx.y * NB;
  const uint32_t nb = min((uint32_t)NB, B - b0);
  const int t = threadIdx.x;
  const bool active = t < (int)g.count;
  uint32_t qi = 0, begin = 0, span = 0;
  if (active) {
    qi = perm[g.start + t];
    begin = win_begin[qi];
    span = win_span[qi];
  }
  const uint32_t rel = begin - g.min_begin;  // < SPREAD for active threads
  const uint32_t nchunks = (g.max_span + CH - 1) / CH;
  for (uint32_t j = 0; j < nb; ++j) accs[j * GQ + t] = 0.0;

  for (uint32_t c = 0; c < nchunks; ++c) {
    const uint32_t cbase = c * CH;           // chunk offset inside every thread's own window
    const bool mine = active && cbase < span;
    const uint32_t rem = mine ? span - cbase : 0;  // sites of this thread in this chunk (cap CH)
    // ---- my CH codes as byte offsets (code * 8), packed 4 per register
    uint32_t cw[CW];
    if (mine) {
      const size_t addr = (size_t)qi * W + begin + cbase;
      const size_t a0 = addr & ~(size_t)3;
      const uint32_t sh = (uint32_t)(addr & 3) * 8;
      const size_t last = (codes_bytes - 1) & ~(size_t)3;
      const uint32_t* p = reinterpret_cast<const uint32_t*>(codes);
      uint32_t prev = p[min(a0, last) >> 2];
#pragma unroll
      for (int i = 0; i < CW; ++i) {
        const uint32_t nxt = p[min(a0 + 4 * (i + 1), last) >> 2];
        const uint32_t v = __funnelshift_r(prev, nxt, sh);
        cw[i] = v << 3;  // each byte < 32 -> *8 stays inside the byte
        prev = nxt;
      }
    } else {
#pragma unroll
      for (int i = 0; i < CW; ++i) cw[i] = 0;
    }
    const uint32_t row0 = g.min_begin + cbase;  // first alignment site of the staged slice
    for (uint32_t j = 0; j < nb; ++j) {
      __syncthreads();  // previous consumers of `tile` are done
      {
        const uint32_t rows = (row0 < W) ? min((uint32_t)TROWS, W - row0) : 0;
        const double2* src = reinterpret_cast<const double2*>(
            lookup + ((size_t)(b0 + j) * W + row0) * NCOLS);
        double2* dst = reinterpret_cast<double2*>(tile);
        const uint32_t n2 = rows * NCOLS / 2;
        for (uint32_t i = t; i < n2; i += GQ) dst[i] = src[i];
      }
      __syncthreads();
      if (mine) {
        double sum = accs[j * GQ + t];
        const char* base = reinterpret_cast<const char*>(tile) + (size_t)rel * NCOLS * 8;
#pragma unroll
        for (int i = 0; i < CW; ++i) {
          const uint32_t s0 = 4 * i;
          if (s0 + 3 < rem) {
            const uint32_t w = cw[i];
            const double v0 = *reinterpret_cast<const double*>(base + (s0 + 0) * NCOLS * 8 + (w & 0xff));
            const double v1 = *reinterpret_cast<const double*>(base + (s0 + 1) * NCOLS * 8 + ((w >> 8) & 0xff));
            const double v2 = *reinterpret_cast<const double*>(base + (s0 + 2) * NCOLS * 8 + ((w >> 16) & 0xff));
            const double v3 = *reinterpret_cast<const double*>(base + (s0 + 3) * NCOLS * 8 + (w >> 24));
            double s1 = v0 + v1;
            const double s2 = v2 + v3;
            s1 += s2;
            sum += s1;
          } else if (s0 < rem) {  // tail of the window: singles, in order
            const uint32_t w = cw[i];
#pragma unroll
            for (int k = 0; k < 3; ++k)
              if (s0 + k < rem)
                sum += *reinterpret_cast<const double*>(base + (s0 + k) * NCOLS * 8 + ((w >> (8 * k)) & 0xff));
          }
        }
        accs[j * GQ + t] = sum;
      }
    }
  }
  if (active) {
    double* out = lnl + (size_t)qi * B + b0;
    for (uint32_t j = 0; j < nb; ++j) out[j] = accs[j * GQ + t];
  }
}

// ---------------------------------------------------------------------------------------------
// k_select: dynamic heuristic on device (apply_heuristic -> dynamic_heuristic,
// src/core/heuristics.hpp:40-68; compute_and_set_lwr src/set_manipulators.cpp:43-69;
// until_accumulated_reached :90-114).  One wave per query; the row of B log-likelihoods is
// cached in LDS, then the largest remaining LWR is extracted until the running sum reaches
// `threshold` (the crossing element is included, min 1).  Ties: lowest branch id first
// (the reference's std::sort is unstable).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_select(const double* __restrict__ lnl, uint32_t B,
                                               double threshold,
                                               unsigned long long* __restrict__ keys,
                                               unsigned long long max_pairs,
                                               unsigned long long* __restrict__ counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* row = reinterpret_cast<double*>(smem);
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
  const double* src = lnl + (size_t)q * B;
  double mx = -INFINITY;
  for (uint32_t i = lane; i < B; i += 64) {
    const double v = src[i];
    row[i] = v;
    mx = fmax(mx, v);
  }
  for (int o = 32; o; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  double tot = 0.0;
  for (uint32_t i = lane; i < B; i += 64) tot += exp(row[i] - mx);
  for (int o = 32; o; o >>= 1) tot += __shfl_xor(tot, o);
  __builtin_amdgcn_wave_barrier();
  double sum = 0.0;
  uint32_t taken = 0;
  while (taken < B && sum < threshold) {
    double best = -INFINITY;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = lane; i < B; i += 64) {
      const double v = row[i];
      if (v > best) { best = v; bi = i; }  // first (lowest index) maximum per lane
    }
    for (int o = 32; o; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const uint32_t oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0xffffffffu) break;
    sum += exp(best - mx) / tot;
    if (lane == 0) {
      row[bi] = -INFINITY;
      const unsigned long long slot = atomicAdd(counter, 1ull);
      if (slot < max_pairs) keys[slot] = ((unsigned long long)bi << 32) | q;
    }
    ++taken;
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void k_keys_to_pairs(const unsigned long long* __restrict__ keys, uint64_t n,
                                epa_pair* __restrict__ pairs) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    pairs[i].branch_id = (uint32_t)(keys[i] >> 32);
    pairs[i].seq_id = (uint32_t)(keys[i] & 0xffffffffu);
  }
}

}  // namespace

int launch_preplace(epa_ctx* ctx, const uint8_t* d_codes, const uint32_t* hb, const uint32_t* hs,
                    const uint32_t* d_begin, const uint32_t* d_span, uint32_t Q, double* d_lnl) {
  // sort by window start, cut into groups (host: Q is a few thousand..1e5 keys)
  std::vector<uint32_t> perm(Q);
  std::iota(perm.begin(), perm.end(), 0u);
  std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return hb[a] < hb[b]; });
  std::vector<Group> groups;
  uint32_t i = 0;
  while (i < Q) {
    Group g{i, 0, hb[perm[i]], 0};
    while (i < Q && g.count < (uint32_t)GQ && hb[perm[i]] - g.min_begin < (uint32_t)SPREAD) {
      g.max_span = std::max(g.max_span, hs[perm[i]]);
      ++g.count;
      ++i;
    }
    groups.push_back(g);
  }
  uint32_t* d_perm = (uint32_t*)epa_scratch(ctx, 6, sizeof(uint32_t) * Q + sizeof(Group) * groups.size() + 64);
  if (!d_perm) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(perm)");
  Group* d_groups = reinterpret_cast<Group*>(d_perm + ((Q + 3) & ~3u));
  EPA_HIP(ctx, hipMemcpyAsync(d_perm, perm.data(), sizeof(uint32_t) * Q, hipMemcpyHostToDevice, ctx->stream));
  EPA_HIP(ctx, hipMemcpyAsync(d_groups, groups.data(), sizeof(Group) * groups.size(),
                              hipMemcpyHostToDevice, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // perm/groups are stack-owned host buffers
  dim3 grid((uint32_t)groups.size(), (ctx->B + NB - 1) / NB);
  const size_t lds = sizeof(double) * ((size_t)TROWS * ctx->ncols + (size_t)NB * GQ);
  const size_t codes_bytes = (size_t)Q * ctx->W;
  epa_timer_start(ctx, ctx->t_preplace);
  if (ctx->ncols == 16) {
    EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_preplace<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_preplace<16>, grid, dim3(GQ), lds, ctx->stream, ctx->lookup, d_codes, d_begin,
                       d_span, d_perm, d_groups, ctx->W, ctx->B, codes_bytes, d_lnl);
  } else {
    EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_preplace<24>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_preplace<24>, grid, dim3(GQ), lds, ctx->stream, ctx->lookup, d_codes, d_begin,
                       d_span, d_perm, d_groups, ctx->W, ctx->B, codes_bytes, d_lnl);
  }
  epa_timer_stop(ctx, ctx->t_preplace);
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}

int launch_select(epa_ctx* ctx, const double* d_lnl, uint32_t Q, double threshold,
                  epa_pair* d_pairs, uint64_t max_pairs, uint64_t* n_pairs) {
  // scratch 7: [counter | keys_in[max] | keys_out[max] | rocprim temp]
  size_t temp_bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, temp_bytes, (unsigned long long*)nullptr,
                                 (unsigned long long*)nullptr, max_pairs, 0, 64, ctx->stream);
  const size_t need = 256 + 2 * sizeof(unsigned long long) * max_pairs + temp_bytes;
  char* base = (char*)epa_scratch(ctx, 7, need);
  if (!base) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(select scratch)");
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(base);
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(base + 256);
  unsigned long long* keys_out = keys_in + max_pairs;
  void* temp = keys_out + max_pairs;
  EPA_HIP(ctx, hipMemsetAsync(counter, 0, 8, ctx->stream));
  epa_timer_start(ctx, ctx->t_select);
  hipLaunchKernelGGL(k_select, dim3(Q), dim3(64), sizeof(double) * ctx->B, ctx->stream, d_lnl, ctx->B,
                     threshold, keys_in, (unsigned long long)max_pairs, counter);
  unsigned long long n = 0;
  EPA_HIP(ctx, hipMemcpyAsync(&n, counter, 8, hipMemcpyDeviceToHost, ctx->stream));
  EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n > max_pairs)
    return epa_fail(ctx, EPA_ERR_INVALID_ARG,
                    "select_candidates: " + std::to_string(n) + " candidates exceed max_pairs");
  if (n) {
    // branch-major order == Work iteration order (std::map<branch, vector<seq>>)
    EPA_HIP(ctx, rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_keys_to_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       keys_out, (uint64_t)n, d_pairs);
  }
  epa_timer_stop(ctx, ctx->t_select);
  EPA_HIP(ctx, hipGetLastError());
  *n_pairs = n;
  return EPA_OK;
}
