// Hot loop 1 on device: preplacement gather-sum + candidate selection, no host in the loop.
//
// k_preplace replaces Lookup_Store::sum_precomputed_sitelk inside place()
// (src/core/Lookup_Store.hpp:110-141, src/core/place.cpp:65-91):
//     lnl[q][b] = sum_{site in window(q)} T[b][site][code(q, site)]
// with the reference's association order ((a0+a1)+(a2+a3) per group of 4, then singles).
//
// Pipeline (all on the stream; the host reads back one 36-word block: candidate total + status):
//   1. queries are sorted by window start -- the fast paths by a counting sort over (class, start
//      parity, start) (k_count_keys / k_scan_counts / k_scatter_sorted), the generic path by a device
//      radix sort -- and k_make_groups cuts the sorted list into groups whose starts fall into one
//      SPREAD-site bucket (<= 1024 queries on the fast paths, <= 256 on the generic one);
//   2. k_preplace (generic): workgroup = one query group x one tile of NB branches.  Per (branch,
//      160-site chunk) the slice of T the whole group can touch is staged through LDS once
//      (coalesced 16 B/lane loads), then every thread gathers its own query's values with
//      ds_read_b64.  Query codes sit in registers as packed byte offsets, partial sums in LDS.  For
//      the 16-column DNA table the LDS columns are XOR-swizzled by the row ((row >> 1) & 15).
//   2b. DNA fast path (k_preplace_pairs): queries whose windows hold only A/C/G/T/N-or-gap use a
//      second table T2[b][s][36] = T[b][s][c0] + T[b][s+1][c1] over {A,C,G,T,N,none}^2: one
//      ds_read_b64 per TWO sites, and the value is exactly the (a0 + a1) the reference forms
//      first, so the association order ((a0+a1)+(a2+a3)) and the results stay bit-identical to
//      the generic path.  Queries are additionally split by window-start parity (a group stages
//      only the pair rows of its own parity) and their per-pair LDS offsets are precomputed once
//      per chunk of queries (k_pack_pairs); queries with other ambiguity codes take the generic
//      kernel.  Persistent grid, XCD-contiguous walk of the (tile, group) items (item_walk),
//      branches per item chosen per launch (choose_tile), pair rows in LDS-bank order
//      (pair_entry, ROWL_NARROW), results written as whole 64-byte sectors.
//   2c. 20-state fast path (k_preplace_sites): the same machinery over the ordinary [W][24] table.
//   3. k_select (the three selection rules) works on the table in HBM: one wave per query, the row
//      lives in registers; a selected (query, branch) is one bit of a [B][Q] bitmap + a per-branch
//      counter, the candidate list in Work's branch-major order is one scan over the counters and
//      one pass over the bitmap (k_emit_pairs).  Bitmaps over 64 MB: per-query staging rows, an
//      exclusive scan of the counts, compaction and a stable radix sort on the branch bits.
#include "epa_dev_internal.hpp"
#include "wave_util.hpp"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

// rocprim's radix sort falls back to a merge sort (block sort + log2 n merge passes: ~19 launches for
// the 262k candidate keys of a chunk) below one million items; the sorts here use a few key bits only
// (branch id, window start, span class), where Onesweep digit passes do: 4 - 5 launches.
using epa_radix_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                 rocprim::default_config, 0>;
#include <rocprim/device/device_scan.hpp>

#include <algorithm>

namespace {

constexpr int GQ = 256;      // queries (threads) per group
constexpr int CH = 160;      // sites per chunk (multiple of 4)
constexpr int SPREAD = 96;   // window starts of a group lie in one bucket of SPREAD sites
constexpr int TROWS = CH + SPREAD;  // rows of T staged per (branch, chunk)
constexpr int NB = 16;       // branches per workgroup
constexpr int CW = CH / 4;   // packed code words per chunk

struct Group {
  uint32_t start;      // first index into the sorted order
  uint32_t count;      // 0 = unused slot of the over-provisioned group table
  uint32_t min_begin;  // bucket start (key space)
  uint32_t cls;        // 0 = site-pair fast path (or the only class), 1 = generic kernel
};

// window validation (error words read back by preplace_check_status)
__device__ __forceinline__ void validate_window(uint32_t i, uint32_t begin, uint32_t span, uint32_t W,
                                                uint32_t cmax, uint32_t* status) {
  if (span == 0) atomicMax(&status[0], 0x80000000u | i);                          // all-gap query
  else if ((uint64_t)begin + span > W || span > cmax) atomicMax(&status[1], 0x80000000u | i);  // width
}
__global__ void k_validate(const uint32_t* __restrict__ win_begin, const uint32_t* __restrict__ win_span,
                           uint32_t Q, uint32_t W, uint32_t cmax, uint32_t* __restrict__ status) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Q) validate_window(i, win_begin[i], win_span[i], W, cmax, status);
}

__global__ void k_iota(uint32_t* v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

// Counting sort of the queries by their grouping key (the fast paths; K = key space, a few thousand
// values): a kernel counts every key and keeps each query's arrival rank, one workgroup turns
// the counts into offsets, a scatter writes (sorted_keys, perm).  Three short kernels instead of the
// ten launches of a two-pass device radix sort with its histogram fills (~75 us per chunk, the same for
// 5 000 reads as for 100 000).  The order of equal keys is arrival order: which lane of a group a
// query lands on varies from run to run, its sums do not.
// one thread per query: its key's count, the old value = the query's rank among equal keys (in the
// pack kernels, one returning atomic per WAVE, the same 100k atomics cost 50 us instead of 5)
__global__ void __launch_bounds__(256) k_count_keys(const uint32_t* __restrict__ keys, uint32_t Q,
                                                    uint32_t* __restrict__ cnt, uint32_t* __restrict__ rank) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q < Q) rank[q] = atomicAdd(&cnt[keys[q]], 1u);
}
__global__ void __launch_bounds__(1024) k_scan_counts(uint32_t* __restrict__ cnt, uint32_t K) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  constexpr uint32_t PER = 8;
  for (uint32_t base = 0; base < K; base += 1024 * PER) {
    uint32_t v[PER], tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
      const uint32_t k = base + t * PER + i;
      v[i] = k < K ? cnt[k] : 0u;
      tot += v[i];
    }
    uint32_t inc = tot;                       // inclusive scan of the thread totals inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o);
      if ((int)lane >= o) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t before = carry_s;
    for (uint32_t w = 0; w < wv; ++w) before += wsum[w];
    uint32_t run = before + inc - tot;
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
      const uint32_t k = base + t * PER + i;
      if (k < K) cnt[k] = run;
      run += v[i];
    }
    __syncthreads();
    if (t == 1023) carry_s = run;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_scatter_sorted(const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ rank,
                                                        const uint32_t* __restrict__ offs, uint32_t Q,
                                                        uint32_t* __restrict__ sorted_keys,
                                                        uint32_t* __restrict__ perm) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  const uint32_t k = keys[q], pos = offs[k] + rank[q];
  sorted_keys[pos] = k;
  perm[pos] = q;
}

// One workgroup.  sorted_keys ascending (window start, or class / parity / start composed by
// k_pack_pairs: block = key / Wp, blocks >= class_blocks are class 1).  Every block is cut into
// runs of gq0 (class 0) / gq1 (class 1) consecutive queries; a run whose window starts spread over
// SPREAD sites or more (sparse data) is subdivided along the fixed SPREAD grid, so every group
// fits the staged slice.
__global__ void __launch_bounds__(256) k_make_groups(const uint32_t* __restrict__ sorted_keys, uint32_t Q,
                                                     uint32_t Wp, uint32_t n_blocks,
                                                     uint32_t class_blocks, uint32_t gq0, uint32_t gq1,
                                                     uint32_t spr0, uint32_t max_runs, Group* __restrict__ groups,
                                                     uint32_t max_groups, uint32_t* __restrict__ status) {
  extern __shared__ uint32_t sh[];
  uint32_t* lo = sh;                       // [n_blocks + 1] first sorted index of each block
  uint32_t* runbase = lo + n_blocks + 1;   // [n_blocks + 1] first run of each block
  uint32_t* run_start = runbase + n_blocks + 1;  // [max_runs]
  uint32_t* run_end = run_start + max_runs;
  uint32_t* run_off = run_end + max_runs;        // first group slot of the run
  auto lower_bound = [&](uint32_t a, uint32_t b, uint64_t key) {
    while (a < b) {
      const uint32_t m = (a + b) >> 1;
      if ((uint64_t)sorted_keys[m] < key) a = m + 1; else b = m;
    }
    return a;
  };
  for (uint32_t k = threadIdx.x; k <= n_blocks; k += blockDim.x)
    lo[k] = k == n_blocks ? Q : lower_bound(0, Q, (uint64_t)k * Wp);
  for (uint32_t i = threadIdx.x; i < max_groups; i += blockDim.x) groups[i] = Group{0, 0, 0, 0};
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
      const uint32_t gq = b >= class_blocks ? gq1 : gq0;
      runbase[b] = r;
      r += (lo[b + 1] - lo[b] + gq - 1) / gq;
    }
    runbase[n_blocks] = r;
  }
  __syncthreads();
  const uint32_t nruns = min(runbase[n_blocks], max_runs);
  for (uint32_t r = threadIdx.x; r < nruns; r += blockDim.x) {
    uint32_t b = 0;
    while (b + 1 < n_blocks && r >= runbase[b + 1]) ++b;
    const uint32_t gq = b >= class_blocks ? gq1 : gq0;
    const uint32_t start = lo[b] + (r - runbase[b]) * gq, end = min(lo[b + 1], start + gq);
    const uint32_t k0 = sorted_keys[start], k1 = sorted_keys[end - 1];
    const uint32_t spr = b >= class_blocks ? (uint32_t)SPREAD : spr0;   // class 0 may stage wider slices
    run_start[r] = start;
    run_end[r] = end;
    run_off[r] = (k1 - k0 < spr) ? 1u : (k1 / spr - k0 / spr + 1);
  }
  __syncthreads();
  // exclusive scan of the group counts: a few hundred runs, every thread sums its own prefix
  uint32_t* cnt = run_off + max_runs + 1;  // [max_runs] copy of the counts
  for (uint32_t r = threadIdx.x; r < nruns; r += blockDim.x) cnt[r] = run_off[r];
  __syncthreads();
  for (uint32_t r = threadIdx.x; r <= nruns; r += blockDim.x) {
    uint32_t g = 0;
    for (uint32_t i = 0; i < r; ++i) g += cnt[i];
    run_off[r] = g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // group ranges for the persistent kernels: class 0 = [0, status[5]), class 1 = [status[5], status[4])
    const uint32_t r1 = class_blocks < n_blocks ? min(runbase[class_blocks], nruns) : nruns;
    status[4] = min(run_off[nruns], max_groups);
    status[5] = min(run_off[r1], max_groups);
  }
  for (uint32_t r = threadIdx.x; r < nruns; r += blockDim.x) {
    uint32_t b = 0;
    while (b + 1 < n_blocks && r >= runbase[b + 1]) ++b;
    const uint32_t cls = b >= class_blocks ? 1u : 0u;
    const uint32_t start = run_start[r], end = run_end[r], g0 = run_off[r], n = run_off[r + 1] - g0;
    if (n == 1) {
      if (g0 < max_groups) groups[g0] = Group{start, end - start, 0, cls};
    } else {
      const uint32_t spr = cls ? (uint32_t)SPREAD : spr0;
      const uint32_t kb = sorted_keys[start] / spr;
      uint32_t a = start;
      for (uint32_t k = 0; k < n; ++k) {
        const uint32_t e = k + 1 == n ? end : lower_bound(a, end, (uint64_t)(kb + k + 1) * spr);
        if (g0 + k < max_groups) groups[g0 + k] = Group{a, e - a, 0, cls};
        a = e;
      }
    }
  }
}

// ACC: partial sums carried across 160-site chunks in LDS (windows longer than CH); the
// short-window instantiation keeps no accumulators and stages only the slice (32 KB of LDS,
// 4 workgroups per CU).
template <int NCOLS, bool ACC>
__global__ void __launch_bounds__(GQ, ACC ? 2 : 4) k_preplace(const double* __restrict__ lookup,
                                                 const uint8_t* __restrict__ codes,
                                                 const uint32_t* __restrict__ win_begin,
                                                 const uint32_t* __restrict__ win_span,
                                                 const uint32_t* __restrict__ perm,
                                                 const Group* __restrict__ groups, uint32_t W,
                                                 uint32_t cstride, uint32_t crel,
                                                 uint32_t B, uint32_t pitch, size_t codes_bytes, uint32_t want_cls,
                                                 const uint32_t* __restrict__ status,
                                                 double* __restrict__ lnl) {
  constexpr bool SWZ = NCOLS == 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* tile = reinterpret_cast<double*>(smem);                  // [TROWS][NCOLS]
  double* accs = tile + (size_t)TROWS * NCOLS;                     // [NB][GQ] (ACC only)
  __shared__ uint32_t s_maxspan;
  // Persistent grid: work item = (group, tile of NB branches), groups fastest (neighbouring groups
  // overlap in the rows of T they stage: L2 reuse).  The group range of this kernel's class comes
  // from k_make_groups (status[4], status[5]); no workgroup is launched for an empty slot.
  const uint32_t gbeg = want_cls ? status[5] : 0, gend = want_cls ? status[4] : status[5];
  const uint32_t ng = gend - gbeg, ntiles = (B + NB - 1) / NB;
  const int t = threadIdx.x;
  for (uint32_t item = blockIdx.x; item < ng * ntiles; item += gridDim.x) {
  const Group g = groups[gbeg + item % ng];
  const uint32_t b0 = (item / ng) * NB;
  const uint32_t nb = min((uint32_t)NB, B - b0);
  __syncthreads();  // the previous item's readers of s_maxspan / accs / tile are done
  if (g.count == 0) continue;
  const bool active = t < (int)g.count;
  uint32_t qi = 0, begin = 0, span = 0;
  if (active) {
    qi = perm[g.start + t];
    begin = win_begin[qi];
    span = win_span[qi];
    if ((uint64_t)begin + span > W || (crel && span > cstride)) span = 0;  // invalid window: flagged by the validation pass
  }
  if (t == 0) s_maxspan = 0;
  __syncthreads();
  atomicMax(&s_maxspan, span);
  if (ACC) for (uint32_t j = 0; j < nb; ++j) accs[j * GQ + t] = 0.0;
  __syncthreads();
  // the group is a run of the begin-sorted order: its first / last member bound the rows of T
  // any member can touch in a chunk -> stage only those (spread + CH <= TROWS rows)
  const uint32_t gmin = win_begin[perm[g.start]];
  const uint32_t gspread = win_begin[perm[g.start + g.count - 1]] - gmin;  // < SPREAD
  const uint32_t rel = begin - gmin;
  const uint32_t nchunks = ACC ? (s_maxspan + CH - 1) / CH : 1;

  for (uint32_t c = 0; c < nchunks; ++c) {
    const uint32_t cbase = c * CH;           // chunk offset inside every thread's own window
    const bool mine = active && cbase < span;
    const uint32_t rem = mine ? span - cbase : 0;  // sites of this thread in this chunk (cap CH)
    // ---- my CH codes as byte offsets (code * 8), packed 4 per register
    uint32_t cw[CW];
    if (mine) {
      // row of query qi: Q x W layout (window at +begin) or compact (window at 0)
      const size_t addr = (size_t)qi * cstride + (crel ? 0u : begin) + cbase;
      const size_t a0 = addr & ~(size_t)3;
      const uint32_t sh = (uint32_t)(addr & 3) * 8;
      const size_t last = (codes_bytes - 1) & ~(size_t)3;
      const uint32_t* p = reinterpret_cast<const uint32_t*>(codes);
      uint32_t prev = p[min(a0, last) >> 2];
#pragma unroll
      for (int i = 0; i < CW; ++i) {
        const uint32_t nxt = p[min(a0 + 4 * (i + 1), last) >> 2];
        const uint32_t v = __funnelshift_r(prev, nxt, sh);
        uint32_t o = v << 3;  // each byte < 32 -> *8 stays inside the byte
        if (SWZ) {  // fold the row swizzle into the offsets once per chunk, not once per gather
          const uint32_t r = rel + 4 * i;  // tile row of the word's first site
          o ^= (((r >> 1) & 15) << 3) | ((((r + 1) >> 1) & 15) << 11) | ((((r + 2) >> 1) & 15) << 19) |
               ((((r + 3) >> 1) & 15) << 27);
        }
        cw[i] = o;
        prev = nxt;
      }
    } else {
#pragma unroll
      for (int i = 0; i < CW; ++i) cw[i] = 0;
    }
    const uint32_t row0 = gmin + cbase;  // first alignment site of the staged slice
    const uint32_t need = min((uint32_t)TROWS, gspread + min((uint32_t)CH, s_maxspan - cbase));
    for (uint32_t j = 0; j < nb; ++j) {
      __syncthreads();  // previous consumers of `tile` are done
      {
        const uint32_t rows = (row0 < W) ? min(need, W - row0) : 0;
        const double2* src = reinterpret_cast<const double2*>(
            lookup + ((size_t)(b0 + j) * W + row0) * NCOLS);
        double2* dst = reinterpret_cast<double2*>(tile);
        const uint32_t n2 = rows * NCOLS / 2;
        constexpr int PF = (TROWS * NCOLS / 2 + GQ - 1) / GQ;
        double2 pf[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {  // all loads of the slice in flight at once
          const uint32_t i = u * GQ + t;
          if (i < n2) pf[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const uint32_t i = u * GQ + t;
          if (i < n2) {
            double2 v = pf[u];
            uint32_t o = i;
            if (SWZ) {  // column pair (2p, 2p+1) of row r goes to columns (2p ^ f, (2p+1) ^ f)
              const uint32_t r = i >> 3, f = (r >> 1) & 15;
              o = (r << 3) | ((i & 7) ^ (f >> 1));
              if (f & 1) { const double x = v.x; v.x = v.y; v.y = x; }
            }
            dst[o] = v;
          }
        }
      }
      __syncthreads();
      if (mine) {
        double sum = ACC ? accs[j * GQ + t] : 0.0;
        // the gather addresses (row base + per-site byte offset) are loop-invariant over the
        // branches: left alone hipcc hoists all 160 of them into VGPRs; the zero token ties
        // them to j so they are rebuilt (2 VALU per gather) and the kernel fits 4 workgroups/CU
        uint32_t zj;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zj) : "v"(j));
        const char* base = reinterpret_cast<const char*>(tile);
        const char* mybase = base + (size_t)(rel * (NCOLS * 8) + zj);
        auto at = [&](uint32_t s, uint32_t boff) -> double {
          return *reinterpret_cast<const double*>(mybase + s * (NCOLS * 8) + boff);
        };
        // Branch-free main loop: every group of 4 is read and summed in the reference's
        // association order, groups past the window contribute an exact +0.0.  (A branch per
        // group would end the basic block after 4 reads and expose the full LDS latency 40
        // times per slice; this way up to 15 reads stay in flight.)
        const uint32_t cur = min(rem, (uint32_t)CH);
        const uint32_t nfull = cur >> 2, ntail = cur & 3;
        // Software pipeline: batch k+1 (2 groups = 8 ds_read_b64) is issued before batch k is
        // summed; sched_barriers pin that order (left alone, hipcc sinks every read next to its
        // use and each group of 4 waits out a full LDS round trip: measured 183 cycles/group).
        constexpr int GPB = 2, NBATCH = CW / GPB;
        double rb[2][GPB * 4];
        auto issue = [&](int bt) {
#pragma unroll
          for (int g = 0; g < GPB; ++g) {
            const int i = bt * GPB + g;
            const uint32_t w = cw[i], s0 = 4 * i;
            rb[bt & 1][g * 4 + 0] = at(s0 + 0, w & 0xff);
            rb[bt & 1][g * 4 + 1] = at(s0 + 1, (w >> 8) & 0xff);
            rb[bt & 1][g * 4 + 2] = at(s0 + 2, (w >> 16) & 0xff);
            rb[bt & 1][g * 4 + 3] = at(s0 + 3, w >> 24);
          }
        };
        issue(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
          if (bt + 1 < NBATCH) issue(bt + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < GPB; ++g) {
            const double* v = &rb[bt & 1][g * 4];
            double s1 = v[0] + v[1];
            const double s2 = v[2] + v[3];
            s1 += s2;
            sum += ((uint32_t)(bt * GPB + g) < nfull) ? s1 : 0.0;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ntail) {  // tail of the window: singles, in order
          uint32_t w = 0;
#pragma unroll
          for (int i = 0; i < CW; ++i) w = ((uint32_t)i == nfull) ? cw[i] : w;
          const uint32_t s0 = 4 * nfull;
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if ((uint32_t)k < ntail) sum += at(s0 + k, (w >> (8 * k)) & 0xff);
        }
        if (ACC) accs[j * GQ + t] = sum;
        else lnl[(size_t)qi * pitch + b0 + j] = sum;
      }
    }
  }
  if (ACC && active) {
    double* out = lnl + (size_t)qi * pitch + b0;
    for (uint32_t j = 0; j < nb; ++j) out[j] = accs[j * GQ + t];
  }
  }  // work items
}

// ---------------------------------------------------------------------------------------------
// Site-pair fast path (DNA).
// ---------------------------------------------------------------------------------------------
constexpr int NSYM = 6;                 // A C G T N(or gap) none
constexpr int PE = NSYM * NSYM;         // entries per pair row, in pair_entry() order
constexpr int PROWB = PE * 8;           // 288 bytes per pair row in HBM
// Entry order and LDS row stride are chosen for the LDS banks (64 dwords; a ds_read_b64 serves 32
// lanes per cycle when their 8-byte words sit on different bank pairs or coincide).  32 neighbouring
// queries of the sorted order start on one or two adjacent pair rows and, site pair by site pair, read
// one of the 16 plain-nucleotide entries of "their" row.  Those 16 entries come FIRST in a row (128
// contiguous bytes = 32 banks) and rows are ROWL_NARROW = 384 bytes apart in LDS (96 dwords = 32 mod
// 64): the hot halves of adjacent rows cover all 64 banks once, the gathers of a lane group are
// conflict-free.  exp/lds_gather.hip, 37 queries per start: 198 B/clk/CU against 143 for the old
// e = s0 * 6 + s1 order at 288 bytes (row r on bank 8 r: every second symbol of adjacent rows collides),
// which is why padding 288 -> 296 alone never helped.  Wide (small-chunk) and multi-chunk variants keep
// 288-byte rows: their slices would not fit the 16-bit offsets / the LDS at 384, and their lane groups
// span many rows anyway (12 queries per start: 125 vs 120 B/clk).
constexpr int ROWL_NARROW = 384;
constexpr int ROWL_PACKED = PROWB;
constexpr int CP = CH / 2;              // pair slots per chunk
constexpr int PW = CP / 2;              // packed words (2 x 16-bit LDS offsets) per chunk
constexpr int TROWS2 = TROWS / 2;       // pair rows staged per (branch, chunk)
constexpr int GQ2 = 1024;               // queries (threads) per group of the pair path: one
                                        // workgroup per CU, the staged slice is shared by 1024 queries
                                        // branches per item of the single-chunk fast paths: choose_tile()
constexpr int NB2_ACC = 15;             // branches per work item when partial sums live in LDS (multi-chunk
                                        // windows): 15 x 1024 doubles + the 36 KB slice fill the CU's 160 KB
constexpr int NB2_ACC_S = 14;           // the same for the 20-state site path (43 KB slice)
constexpr int NB2_BURST = 8;            // single-chunk variants: result rows staged per 64-byte burst
constexpr int NB2_RING = 12;            // ... with staggered bursts (k_preplace_pairs STAG): a ring of rows, a burst may leave up to 3 branches late
constexpr uint32_t ZERO_OFF = (PE - 1) * 8;  // (none, none) of the thread's own first row: exact +0.0

// (symbol of site, symbol of site + 1) -> entry of the pair row: the 16 plain pairs first
__device__ __forceinline__ uint32_t pair_entry(uint32_t s0, uint32_t s1) {
  if (s0 < 4 && s1 < 4) return s0 * 4 + s1;
  if (s0 < 4) return 16 + s0 * 2 + (s1 - 4);
  return 24 + (s0 - 4) * NSYM + s1;      // (5, 5) = PE - 1
}
__device__ __forceinline__ void pair_symbols(uint32_t e, uint32_t& s0, uint32_t& s1) {
  if (e < 16) { s0 = e >> 2; s1 = e & 3u; }
  else if (e < 24) { s0 = (e - 16) >> 1; s1 = 4 + ((e - 16) & 1u); }
  else { s0 = 4 + (e - 24) / NSYM; s1 = (e - 24) % NSYM; }
}

// state-set code (4-bit mask) -> symbol; 6 = any other ambiguity code (generic kernel)
__device__ __forceinline__ uint32_t dna_sym(uint32_t code) {
  if (code == 15) return 4;
  if (code == 1 || code == 2 || code == 4 || code == 8) return (uint32_t)__ffs((int)code) - 1;
  return 6;
}

// T2[b][s][e] = (i0 < 5 ? T[b][s][code(i0)] : 0) + (i1 < 5 ? T[b][s+1][code(i1)] : 0).  The sum is
// the very `a0 + a1` the reference's 4-way unrolled loop forms first (Lookup_Store.hpp:120-131),
// x + 0.0 == x covers the singles of the tail.
__global__ void __launch_bounds__(256) k_build_lookup2(const double* __restrict__ lookup, uint32_t W,
                                                       double* __restrict__ lookup2) {
  const uint32_t b = blockIdx.y;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * PE) return;
  const uint32_t s = i / PE, e = i - s * PE;
  uint32_t i0, i1;
  pair_symbols(e, i0, i1);
  const uint32_t code[5] = {1, 2, 4, 8, 15};
  const double* T = lookup + (size_t)b * W * 16;
  const double v0 = i0 < 5 ? T[(size_t)s * 16 + code[i0]] : 0.0;
  const double v1 = (i1 < 5 && s + 1 < W) ? T[(size_t)(s + 1) * 16 + code[i1]] : 0.0;
  // rows of one start parity are contiguous: [b][s & 1][s >> 1][PE].  A group of queries uses the
  // rows of ONE parity; interleaved, its slice was 288 useful bytes out of every 576
  const uint32_t Wh = (W + 1) / 2;
  lookup2[(((size_t)b * 2 + (s & 1u)) * Wh + (s >> 1)) * PE + e] = v0 + v1;
}

// Per-pair LDS offsets relative to the query's first pair row, 16 bit each (pair p of the window sits in
// chunk p / CP at row p % CP), the three tail singles, the sort key (class, window-start parity, window
// start) for the grouping.
// (code of site, code of site + 1) -> 8 * pair entry, bit 15 = one of the two is a rare ambiguity code
// (the query then goes to the generic kernel); codes above 15 are mapped to code 0 first (same symbol).
struct PairTab {
  uint16_t v[256];
  constexpr PairTab() : v{} {
    for (uint32_t i = 0; i < 256; ++i) {
      const uint32_t c[2] = {i >> 4, i & 15u};
      uint32_t s[2] = {0, 0};
      for (int k = 0; k < 2; ++k)
        s[k] = c[k] == 15 ? 4u : c[k] == 1 ? 0u : c[k] == 2 ? 1u : c[k] == 4 ? 2u : c[k] == 8 ? 3u : 6u;   // = dna_sym
      const bool rare = s[0] > 4 || s[1] > 4;
      const uint32_t s0 = s[0] > 5 ? 5u : s[0], s1 = s[1] > 5 ? 5u : s[1];
      const uint32_t e = (s0 < 4 && s1 < 4) ? s0 * 4 + s1 : s0 < 4 ? 16 + s0 * 2 + (s1 - 4) : 24 + (s0 - 4) * NSYM + s1;   // = pair_entry
      v[i] = (uint16_t)(e * 8 | (rare ? 0x8000u : 0u));
    }
  }
};
__device__ const PairTab PAIR_TAB{};

// Sixteen lanes per query, four queries per wave, a persistent grid (grid-stride over blocks of sixteen
// queries).  Round 5: a lane forms one 32-bit WORD (two pairs = four sites) per step -- four code bytes, two
// table reads (LDS copy of PAIR_TAB), one add of the two row offsets, one 4-byte store (sixteen lanes: 64
// contiguous bytes) -- where the first form walked dna_sym / pair_entry per site (branches: 352 vector + 640
// scalar instructions per wave, 81 us per 100k reads: instruction-bound, profiles/r5_pmc_summary.txt).
// Offsets stay below 2^15 ((CP - 1) * rowl + 35 * 8 <= 30616), so the table's rare bit survives the add in
// bit 15 / 31 and is masked out of the stored word.
__global__ void __launch_bounds__(256) k_pack_pairs(const uint8_t* __restrict__ codes,
                                                    const uint32_t* __restrict__ win_begin,
                                                    const uint32_t* __restrict__ win_span, uint32_t Q,
                                                    uint32_t W, uint32_t cstride, uint32_t crel,
                                                    uint32_t span_bound, uint32_t Wp, uint32_t NP16,
                                                    uint32_t rowl, uint16_t* __restrict__ packed,
                                                    uint16_t* __restrict__ tails,
                                                    uint32_t* __restrict__ keys, uint32_t K,
                                                    uint32_t* __restrict__ status) {
  __shared__ uint16_t s_tab[256];
  s_tab[threadIdx.x] = PAIR_TAB.v[threadIdx.x];
  __syncthreads();
  const uint32_t l16 = threadIdx.x & 15, grp = (threadIdx.x >> 4) & 3u;
  // a window longer than the caller's max_span (or than a compact row) is an input error: the
  // kernel variant and the packed rows were sized by it
  const uint32_t cmax = min(span_bound, crel ? cstride : 0xffffffffu);
  constexpr uint32_t ZERO_W = ZERO_OFF | (ZERO_OFF << 16);
  constexpr int NWI = (PW + 15) / 16;     // word steps per chunk of CP pair slots
  const uint32_t lane_base = (2u * l16 * rowl) * 0x10001u + (rowl << 16);   // rows 2 l16 and 2 l16 + 1
  for (uint32_t q0 = blockIdx.x * 16; q0 < Q; q0 += gridDim.x * 16) {
    const uint32_t q = q0 + (threadIdx.x >> 4);
    const bool live = q < Q;
    const uint32_t begin = live ? win_begin[q] : 0u;
    uint32_t span = live ? win_span[q] : 0u;
    if (live && l16 == 0) validate_window(q, begin, span, W, cmax, status);
    if ((uint64_t)begin + span > W || span > cmax) span = 0;  // invalid window (flagged above)
    const uint8_t* c = codes + (size_t)(live ? q : 0u) * cstride + (crel ? 0u : begin);
    const uint32_t nfull = span >> 2, ntail = span & 3, npairs = 2 * nfull;
    uint32_t* prow = reinterpret_cast<uint32_t*>(packed + (size_t)(live ? q : 0u) * NP16);   // NP16 is even, the base 256-byte aligned
    uint32_t racc = 0;
    // the tail singles' codes are requested with the first chunk's
    uint32_t ctail = 0;
    if (l16 < ntail) ctail = c[4 * nfull + l16];
    for (uint32_t pb = 0; pb < NP16; pb += CP) {   // one chunk of CP = 80 pair slots = 40 words at a time
      uint32_t cb[NWI][4];
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const uint32_t gw = pb / 2 + (uint32_t)i * 16 + l16;   // word = pairs 2 gw, 2 gw + 1 = sites 4 gw .. 4 gw + 3
        const bool on = (i * 16 + (int)l16 < PW) && gw < nfull;
#pragma unroll
        for (int k = 0; k < 4; ++k) cb[i][k] = on ? c[4 * gw + k] : 0u;
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const uint32_t wl = (uint32_t)i * 16 + l16, gw = pb / 2 + wl;
        if (i * 16 + (int)l16 >= PW) continue;
        uint32_t c0 = cb[i][0], c1 = cb[i][1], c2 = cb[i][2], c3 = cb[i][3];
        if (__builtin_expect(((c0 | c1 | c2 | c3) >> 4) != 0u, 0)) {   // not a 4-bit state set: the symbol of code 0
          c0 = c0 > 15u ? 0u : c0; c1 = c1 > 15u ? 0u : c1; c2 = c2 > 15u ? 0u : c2; c3 = c3 > 15u ? 0u : c3;
        }
        const uint32_t e0 = s_tab[(c0 << 4) | c1], e1 = s_tab[(c2 << 4) | c3];
        const uint32_t word = ((e1 << 16) | e0) + lane_base + (uint32_t)i * ((32u * rowl) * 0x10001u);
        const bool on = gw < nfull;
        if (on) racc |= word;
        if (live) prow[gw] = on ? (word & 0x7fff7fffu) : ZERO_W;
      }
    }
    bool rare = (racc & 0x80008000u) != 0u;
    if (l16 < 4) {
      uint32_t v = ZERO_OFF;
      if (l16 < ntail) {
        uint32_t sy = dna_sym(ctail);
        rare |= sy > 4;
        sy = min(sy, 5u);
        const uint32_t kt = npairs % CP;  // row of the first tail site inside its chunk (even)
        const uint32_t e = l16 == 1 ? pair_entry(5, sy) : pair_entry(sy, 5);
        v = (kt + (l16 == 2 ? 1u : 0u)) * rowl + e * 8;
      }
      if (live) tails[(size_t)q * 4 + l16] = (uint16_t)v;
    }
    const bool any_rare = ((__ballot(rare) >> (16 * grp)) & 0xffffull) != 0ull;
    if (live && l16 == 0) {   // an invalid window start (flagged above) must not leave the key space
      const uint32_t key = min((any_rare ? 2 * Wp : 0) + (begin & 1u) * Wp + min(begin, Wp - 1), K - 1);
      keys[q] = key;
    }
  }
}

// Walk of the (branch tile, group) work items by a persistent grid.  Workgroups are dealt to the
// eight XCDs round robin (workgroup w runs on XCD w % 8, each with its own 4 MB L2), and the groups
// are sorted by window start: neighbouring groups stage overlapping slices of the same branch rows
// (a site lies in the slices of ~6 groups at cfg2).  XCD x therefore takes the CONTIGUOUS eighth
// [x T / 8, (x + 1) T / 8) of the items in (tile, group) order and its workgroups stride through
// it, so that at any time one L2 serves neighbouring groups of one tile instead of every eighth:
// slice re-reads that reach the fabric 3.5 GB -> 1.5 GB per 100k-read launch (PMC, round 3).
#ifndef PP_XCD
#define PP_XCD 1
#endif
// ---- per-(query, 64-branch segment) maxima of the preplacement table, written by the single-chunk fast
// paths as a by-product (one fmax per branch, one atomic per segment and work item) and read by the
// candidate selection (k_select_seg): a query's candidates and every term that can move its LWR
// denominator lie in the one or two segments whose maximum is within a few dozen lnL units of the row
// maximum (cfg2: 1.5 of 16 segments on average), so the selection reads 128 bytes of maxima + those
// segments instead of the whole 8 KB row.  Stored as order-preserving 64-bit keys (atomicMax on unsigned
// integers), 0 = nothing written (a query served by another kernel: the selection then reads its row).
__device__ __forceinline__ unsigned long long seg_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double seg_val(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}
struct SegTrack {   // running maximum of the current segment of one query
  double m = -INFINITY;
  __device__ __forceinline__ void add(unsigned long long* __restrict__ segmax, uint32_t segp, uint32_t qi, uint32_t b,
                                      bool last, double v) {
    m = fmax(m, v);
    if ((b & 63u) == 63u || last) {
      atomicMax(&segmax[(size_t)qi * segp + (b >> 6)], seg_key(m));
      m = -INFINITY;
    }
  }
};

// Progress-based wave priority in the gather loops (round 5).  The SIMD's arbiter is oldest-first, so the sixteen waves
// of the workgroup get through a branch's gathers at different speeds and the average wave waited 1200 of a branch's
// 5200 clocks at the barrier for the slowest one (profiles/r4_preplace_cycles.txt).  A wave now starts its gathers at
// priority 3 and steps down to 0 a quarter of its batches at a time: whoever is behind issues first, the waves reach
// the barrier together.  Same box, cfg2: 0.965 -> 0.910 ms per launch (PP_PRIO=0: off; two levels 0.929, thresholds at
// 40 / 70 / 90 % 0.914, at 10 / 30 / 60 % 0.930; the double-buffered and two-branch loops still lose with it:
// 0.949 / 1.011; profiles/r5_preplace_prio_ab.txt).
#ifndef PP_PRIO
#define PP_PRIO 1
#endif
template <int NBATCH>
__device__ __forceinline__ void gather_prio(int bt) {
#if PP_PRIO == 1
  if (bt == 0) __builtin_amdgcn_s_setprio(3);
  else if (bt == NBATCH / 4) __builtin_amdgcn_s_setprio(2);
  else if (bt == NBATCH / 2) __builtin_amdgcn_s_setprio(1);
  else if (bt == (3 * NBATCH) / 4) __builtin_amdgcn_s_setprio(0);
#elif PP_PRIO == 2   // two levels
  if (bt == 0) __builtin_amdgcn_s_setprio(3);
  else if (bt == NBATCH / 2) __builtin_amdgcn_s_setprio(0);
#elif PP_PRIO == 3   // late thresholds
  if (bt == 0) __builtin_amdgcn_s_setprio(3);
  else if (bt == (2 * NBATCH) / 5) __builtin_amdgcn_s_setprio(2);
  else if (bt == (7 * NBATCH) / 10) __builtin_amdgcn_s_setprio(1);
  else if (bt == (9 * NBATCH) / 10) __builtin_amdgcn_s_setprio(0);
#elif PP_PRIO == 4   // early thresholds
  if (bt == 0) __builtin_amdgcn_s_setprio(3);
  else if (bt == NBATCH / 10) __builtin_amdgcn_s_setprio(2);
  else if (bt == (3 * NBATCH) / 10) __builtin_amdgcn_s_setprio(1);
  else if (bt == (6 * NBATCH) / 10) __builtin_amdgcn_s_setprio(0);
#endif
}

struct ItemWalk { uint32_t pos, end, step, lo; };
__device__ __forceinline__ ItemWalk item_walk(uint32_t total) {
  if (PP_XCD && (gridDim.x & 7u) == 0) {
    const uint32_t x = blockIdx.x & 7u, local = blockIdx.x >> 3;
    const uint32_t lo = (uint32_t)((uint64_t)total * x / 8), hi = (uint32_t)((uint64_t)total * (x + 1) / 8);
    return {lo + local, hi, gridDim.x >> 3, lo};
  }
  return {blockIdx.x, total, gridDim.x, 0u};
}

// Which (group, tile) a position v of an XCD's contiguous range [lo, hi) stands for.  The RANGES are cut in group-major
// order (u = group * ntiles + tile): an XCD owns ~ng / 8 whole groups (a boundary group is shared by two), so the
// packed offsets of a group -- 160 B per query, read again by every one of its tiles -- come out of ONE L2 instead of
// a different one per tile (0.2 GB of the launch's HBM-side traffic at cfg2).  INSIDE its range an XCD still walks
// tile-major (tile t: the range's groups ascending), so the workgroups running at the same time stage overlapping
// slices of the same branches.  With lo = 0, hi = total this is the plain tile-major order (tile = v / ng).
#ifndef PP_XCD_GROUPS
#define PP_XCD_GROUPS 1
#endif
__device__ __forceinline__ void item_group_tile(uint32_t v, uint32_t lo, uint32_t hi, uint32_t ng, uint32_t ntiles,
                                                uint32_t& grp, uint32_t& tile) {
  if (!PP_XCD_GROUPS) { grp = v % ng; tile = v / ng; return; }
  const uint32_t gl = lo / ntiles, tl = lo - gl * ntiles, gh = (hi - 1) / ntiles, th = (hi - 1) - gh * ntiles;
  uint32_t r = v - lo;
  for (uint32_t t = 0; t < ntiles; ++t) {
    const uint32_t first = gl + (t < tl ? 1u : 0u), last1 = gh + 1u - (t > th ? 1u : 0u);   // [first, last1)
    const uint32_t c = last1 > first ? last1 - first : 0u;
    if (r < c) { grp = first + r; tile = t; return; }
    r -= c;
  }
  grp = 0; tile = ntiles;   // not reached for lo <= v < hi
}

// Branches per work item of the single-chunk fast paths (ng groups x ceil(B / n) items on nwg
// persistent workgroups): the multiple of 8 (whole result bursts) in 16 .. 96 with the smallest
// rounds x (branches + per-item setup).  Per item a workgroup fetches its queries' offsets (160 B per
// query) and runs a chain of dependent loads before the first slice is staged, ~3 branch times: at
// cfg2 (98 groups, 1021 branches, 256 workgroups) 13 tiles of 80 are 5 rounds = 415 branch times
// where 32 tiles of 32 were 13 rounds = 455, and the offsets are read 13 times instead of 32; a
// 5000-read chunk (10 groups) gets 43 tiles of 24 instead of 1.25 rounds of 32.
__device__ __forceinline__ uint32_t choose_tile(uint32_t ng, uint32_t B, uint32_t nwg) {
  uint32_t best = 32, best_cost = 0xffffffffu;
  for (uint32_t n = 16; n <= 96; n += 8) {
    const uint32_t items = ng * ((B + n - 1) / n);
    const uint32_t cost = ((items + nwg - 1) / nwg) * (n + 3);
    if (cost <= best_cost) { best_cost = cost; best = n; }
  }
  return best;
}

// SPR: window starts of a group lie within SPR sites.  96 for large chunks (many reads per window
// start: 1024 consecutive reads of the sorted order span few starts); 288 for small ones (the
// reference's default --chunk-size 5000 puts ~2 reads on a start: a 96-site bucket holds ~180 reads,
// a workgroup's 1024 lanes would be 18 % full) -- the 16-bit LDS offsets still fit:
// (288 / 2 + 79) * 288 + 35 * 8 < 65536.  ROWL: byte stride of the staged pair rows in LDS (see above).
// (the batch structure was once pinned with sched_barriers; with two-word batches the compiler's own
// schedule is as good or better: -DPP_SCHED_PIN restores them for an A/B)
#ifdef PP_SCHED_PIN
#define PP_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define PP_SCHED_BARRIER()
#endif
// Two other loop forms were built, measured and dropped (source kept under profiles/variants/r5_preplace_db_duo_v1.hip,
// A/Bs in profiles/r5_preplace_duo_ab.txt, r5_preplace_prio_ab.txt): slices of consecutive branches alternating between
// two LDS buffers with one barrier per branch (0.966 / 0.949 against 0.930 / 0.910 ms), and two branches staged and
// gathered between one pair of barriers (1.018 / 1.011 ms) -- the gather phase is bound by the LDS itself.
template <bool ACC, int SPR, int ROWL, bool STAG = false>
__global__ void __launch_bounds__(GQ2, 4) k_preplace_pairs(
    const double* __restrict__ lookup2, const uint16_t* __restrict__ packed,
    const uint16_t* __restrict__ tails, const uint32_t* __restrict__ win_begin,
    const uint32_t* __restrict__ win_span, const uint32_t* __restrict__ perm,
    const Group* __restrict__ groups, uint32_t W, uint32_t B, uint32_t pitch, uint32_t NP16,
    const uint32_t* __restrict__ status, double* __restrict__ lnl,
    unsigned long long* __restrict__ segmax, uint32_t segp) {
  constexpr int TR2 = (CH + SPR) / 2;   // pair rows staged per (branch, chunk)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [TR2][PE] doubles, then accs
  static_assert(ROWL % 16 == 0 && ROWL >= PROWB && ((TR2 - 1) * ROWL + PROWB) < 65536, "16-bit LDS offsets");
  double* accs = reinterpret_cast<double*>(smem + (size_t)TR2 * ROWL);  // [NB2_ACC][GQ2] / burst rows
  __shared__ uint32_t s_maxspan;
  __shared__ uint32_t s_qi[GQ2];   // query of thread t (burst write-out), ~0 = nothing to write
  constexpr uint32_t BSTR = GQ2 + 4;    // burst staging rows 4 doubles apart in the banks: conflict-free
  constexpr uint32_t RING = STAG ? NB2_RING : NB2_BURST;
  static_assert(!STAG || !ACC, "staggered bursts: single-chunk variant only");
  // persistent grid over (group, branch tile) items of class 0, see k_preplace
  const uint32_t ng = status[5];
  const uint32_t NBP = ACC ? NB2_ACC : choose_tile(ng, B, gridDim.x);
  const uint32_t ntiles = (B + NBP - 1) / NBP;
  const int t = threadIdx.x;
  const ItemWalk iw = item_walk(ng * ntiles);
  for (uint32_t item = iw.pos; item < iw.end; item += iw.step) {
  uint32_t item_g, item_t;
  item_group_tile(item, iw.lo, iw.end, ng, ntiles, item_g, item_t);
  const Group g = groups[item_g];
  const uint32_t b0 = item_t * NBP;
  const uint32_t nb = min(NBP, B - b0);
  __syncthreads();  // the previous item's readers of s_maxspan / accs / the tile are done
  if (g.count == 0) continue;
  const bool active = t < (int)g.count;
  uint32_t qi = 0, begin = 0, span = 0;
  if (active) {
    qi = perm[g.start + t];
    begin = win_begin[qi];
    span = win_span[qi];
    if ((uint64_t)begin + span > W) span = 0;
  }
  if (t == 0) s_maxspan = 0;
  const uint32_t my_q = (active && span > 0) ? qi : 0xffffffffu;
  s_qi[t] = my_q;
  __syncthreads();
  {   // one LDS atomic per wave, not per thread (1024 atomics on one word cost ~16k cycles per item)
    uint32_t m = span;
#pragma unroll
    for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((t & 63) == 0) atomicMax(&s_maxspan, m);
  }
  if (ACC) for (uint32_t j = 0; j < nb; ++j) accs[j * GQ2 + t] = 0.0;
  __syncthreads();
  // all members share the parity of their window start: rel is even, pair rows line up
  const uint32_t gmin = win_begin[perm[g.start]];
  const uint32_t gspread = win_begin[perm[g.start + g.count - 1]] - gmin;  // < SPREAD
  const uint32_t rowoff = ((begin - gmin) >> 1) * ROWL;
  const uint32_t rowoff2 = rowoff | (rowoff << 16);
  const uint32_t nchunks = ACC ? (s_maxspan + CH - 1) / CH : 1;
  uint32_t t0 = 0, t1 = 0, t2 = 0, tailchunk = 0xffffffffu;
  if (active && (span & 3)) {
    const uint16_t* tq = tails + (size_t)qi * 4;
    t0 = tq[0] + rowoff; t1 = tq[1] + rowoff; t2 = tq[2] + rowoff;
    tailchunk = ((span >> 2) * 2) / CP;
  }
  auto at = [&](uint32_t off) -> double { return *reinterpret_cast<const double*>(smem + off); };
  SegTrack seg;

  for (uint32_t c = 0; c < nchunks; ++c) {
    const uint32_t cbase = c * CH;
    const bool mine = active && cbase < span;
    uint32_t cw[PW];  // two 16-bit LDS byte offsets per word = one group of 4 sites
    if (mine) {
      // 160 B per (query, chunk), 16 B aligned (NP16 is a multiple of CP = 80)
      const uint4* p = reinterpret_cast<const uint4*>(packed + (size_t)qi * NP16 + (size_t)c * CP);
#pragma unroll
      for (int i = 0; i < PW / 4; ++i) {  // no carry between the halves: offsets stay below 2^16
        const uint4 v = p[i];
        cw[4 * i] = v.x + rowoff2;
        cw[4 * i + 1] = v.y + rowoff2;
        cw[4 * i + 2] = v.z + rowoff2;
        cw[4 * i + 3] = v.w + rowoff2;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PW; ++i) cw[i] = 0;
    }
    const uint32_t row0 = gmin + cbase;  // alignment site of pair row 0 of the staged slice
    const uint32_t need = min((uint32_t)TR2, (gspread >> 1) + (min((uint32_t)CH, s_maxspan - cbase) + 1) / 2);
    const uint32_t rows = (row0 < W) ? min(need, (W - row0 + 1) / 2) : 0;
    const uint32_t n2 = rows * (PE / 2);
    constexpr int PF = (TR2 * (PE / 2) + GQ2 - 1) / GQ2;
    auto request = [&](uint32_t j, double2 (&pfs)[PF]) {
      // pair row r = table row (row0 + 2r); the rows of one parity are contiguous in lookup2
      const double2* src = reinterpret_cast<const double2*>(
          lookup2 + (((size_t)(b0 + j) * 2 + (row0 & 1u)) * ((W + 1) / 2) + (row0 >> 1)) * PE);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const uint32_t i = u * GQ2 + t;
        pfs[u] = make_double2(0.0, 0.0);
        if (i < n2) pfs[u] = src[i];
      }
    };
    auto stage = [&](const double2 (&pfs)[PF], uint32_t boff) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const uint32_t i = u * GQ2 + t;   // double2 index in the compact [rows][PE / 2] slice
        if (i < n2) {
          if (ROWL == PROWB) {
            reinterpret_cast<double2*>(smem + boff)[i] = pfs[u];
          } else {                         // padded rows
            const uint32_t r = i / (PE / 2), c2 = i - r * (PE / 2);
            *reinterpret_cast<double2*>(smem + boff + r * ROWL + c2 * 16) = pfs[u];
          }
        }
      }
    };
    // the gathers of branch j from the slice staged at byte boff; the sum goes to the staging rows
    auto gather = [&](uint32_t j, uint32_t boff) {
      if (mine) {
        double sum = ACC ? accs[j * GQ2 + t] : 0.0;
        // the offsets are invariant over the branches: without this hipcc hoists the 80 extracted
        // addresses out of the j loop into VGPRs (the kernel has to fit 4 workgroups per CU)
#pragma unroll
        for (int i = 0; i < PW; ++i) asm volatile("" : "+v"(cw[i]));
        // Software pipeline as in k_preplace: batch k+1 (4 words = 8 ds_read_b64) is issued
        // before batch k is summed.  Slots past the window read an exact +0.0.
#ifndef PP_WPB
#define PP_WPB 2   // round 4 A/B, ms per 100k-read launch: 1 -> 0.91, 2 -> 0.905 - 0.914, 4 -> 0.93 - 0.95, 5 -> 0.95, 8 -> 1.35
#endif
        constexpr int WPB = PP_WPB, NBATCH = PW / WPB;
        double rb[2][WPB * 2];
        auto issue = [&](int bt) {
#pragma unroll
          for (int w = 0; w < WPB; ++w) {
            const uint32_t v = cw[bt * WPB + w];
            rb[bt & 1][2 * w] = at((v & 0xffffu) + boff);
            rb[bt & 1][2 * w + 1] = at((v >> 16) + boff);
          }
        };
        issue(0);
        PP_SCHED_BARRIER();
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
          gather_prio<NBATCH>(bt);
          if (bt + 1 < NBATCH) issue(bt + 1);
          PP_SCHED_BARRIER();
#pragma unroll
          for (int w = 0; w < WPB; ++w) {
            const double s1 = rb[bt & 1][2 * w] + rb[bt & 1][2 * w + 1];  // (a0+a1) + (a2+a3)
            sum += s1;
          }
          PP_SCHED_BARRIER();
        }
        if (c == tailchunk) {  // singles of the window tail, in order
          sum += at(t0 + boff);
          sum += at(t1 + boff);
          sum += at(t2 + boff);
        }
        if (ACC) {
          accs[j * GQ2 + t] = sum;
        } else {
          accs[(STAG ? j % RING : (j & 7u)) * BSTR + t] = sum;
          if (segmax) seg.add(segmax, segp, qi, b0 + j, j + 1 == nb, sum);
        }
      }
    };
    // Results leave in bursts of 8 consecutive branches, 64 B per query, and the eight lanes
    // l, l+1 .. l+7 write the eight doubles of ONE query: every store instruction hands the
    // memory system whole 64-byte sectors (8 queries per wave and instruction).  History: single
    // 8-byte stores spread over the item were evicted sector by sector (1.76 GB written for a
    // 0.41 GB table); 8 stores per lane back to back, each lane its own sector, relied on L2
    // merging them before eviction (1.19 GB for 0.83 GB; 2.15 GB under the XCD-contiguous walk).
    // Burst kb = branches b0 + 8 kb .. of this item, staged in rows (8 kb + c) % RING.
    auto emit_burst = [&](uint32_t kb) {
      __builtin_amdgcn_wave_barrier();
      // four lanes per query, 16 bytes (two branches) per lane: a store instruction still hands over whole
      // 64-byte sectors (16 queries per wave and instruction) with half as many instructions as 8-byte
      // stores -- the write-out is store-issue bound (in-kernel cycle counters: 3.7k cycles per burst)
      const uint32_t lane = (uint32_t)t & 63u, wbase = (uint32_t)t & ~63u, c4 = lane & 3u;
      const uint32_t ncol = min(8u, nb - 8u * kb), bcol = b0 + 8u * kb + 2u * c4;
      const uint32_t r0 = STAG ? (8u * kb + 2u * c4) % RING : 2u * c4, r1 = STAG ? (8u * kb + 2u * c4 + 1u) % RING : 2u * c4 + 1u;
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t tq = wbase + k * 16 + (lane >> 2);
        const uint32_t q = s_qi[tq];
        if (q != 0xffffffffu && 2u * c4 < ncol) {
          const double a0 = accs[r0 * BSTR + tq];
          double* dst = lnl + (size_t)q * pitch + bcol;
          if (2u * c4 + 1u < ncol) *reinterpret_cast<double2*>(dst) = make_double2(a0, accs[r1 * BSTR + tq]);
          else *dst = a0;
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    // STAG: all sixteen waves used to issue their bursts behind the same barrier, every eighth branch: 64 KB per CU
    // through one write path while the LDS gather path idled (535 of a branch's 5200 clocks).  With four more staging
    // rows (a ring of 12) wave w writes its burst (w & 3) branches late -- four waves per branch on four of eight
    // branches -- beside the other waves' gathers.  A wave only ever reads its own 64 columns of the staging rows.
    const uint32_t stag_d = STAG ? (uint32_t)__builtin_amdgcn_readfirstlane(t >> 6) & 3u : 0u;
    auto flush = [&](uint32_t j) {
      if (ACC) return;
      uint32_t done = 0;   // bursts of this item already written by this wave
      if (j >= 7u + stag_d) {
        done = ((j - 7u - stag_d) >> 3) + 1u;
        if (((j - 7u - stag_d) & 7u) == 0u) emit_burst(done - 1u);
      }
      if (j + 1 == nb)
        for (uint32_t kb = done; 8u * kb < nb; ++kb) emit_burst(kb);
    };
    {
      // The slice of branch j+1 is requested (into registers) before the gathers of branch j start
      // and written to LDS after them: its HBM latency hides under the gather phase.
      double2 pf[PF];
      request(0, pf);
      for (uint32_t j = 0; j < nb; ++j) {
        __syncthreads();  // previous consumers of the tile are done
        stage(pf, 0);
        __syncthreads();
        if (j + 1 < nb) request(j + 1, pf);
        PP_SCHED_BARRIER();
        gather(j, 0);
        flush(j);
      }
    }
  }
  if (ACC && active) {
    double* out = lnl + (size_t)qi * pitch + b0;
    for (uint32_t j = 0; j < nb; ++j) out[j] = accs[j * GQ2 + t];
  }
  }  // work items
}

// ---------------------------------------------------------------------------------------------
// Single-site fast path (20 states): the machinery of k_preplace_pairs -- per-site LDS offsets
// precomputed once per chunk of queries (16 bit each), 1024-query groups, one workgroup per CU,
// register prefetch of the next slice, persistent grid -- over the ordinary [W][24] table.  Sums in
// the reference's order: ((a0+a1)+(a2+a3)) per group of four sites, then the singles.
// ---------------------------------------------------------------------------------------------
constexpr int CHS = 128;              // sites per chunk
constexpr int PWS = CHS / 2;          // packed words per chunk (two 16-bit offsets each)
constexpr int TROWS_S = CHS + SPREAD; // table rows staged per (branch, chunk)

template <int NCOLS>
__global__ void __launch_bounds__(256) k_pack_sites(const uint8_t* __restrict__ codes,
                                                    const uint32_t* __restrict__ win_begin,
                                                    const uint32_t* __restrict__ win_span, uint32_t Q,
                                                    uint32_t W, uint32_t cstride, uint32_t crel,
                                                    uint32_t span_bound, uint32_t NP16,
                                                    uint16_t* __restrict__ packed,
                                                    uint16_t* __restrict__ tails,
                                                    uint32_t* __restrict__ keys, uint32_t K,
                                                    uint32_t* __restrict__ status) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (q >= Q) return;
  const uint32_t begin = win_begin[q];
  uint32_t span = win_span[q];
  const uint32_t cmax = min(span_bound, crel ? cstride : 0xffffffffu);
  if (lane == 0) validate_window(q, begin, span, W, cmax, status);
  if ((uint64_t)begin + span > W || span > cmax) span = 0;
  const uint8_t* c = codes + (size_t)q * cstride + (crel ? 0u : begin);
  const uint32_t nsum = span & ~3u;  // sites summed in groups of four
  for (uint32_t p = lane; p < NP16; p += 64) {
    // slots past the last full group are masked by the kernel: any finite entry of the own row
    uint32_t v = 0;
    if (p < nsum) v = (p % CHS) * (NCOLS * 8) + min((uint32_t)c[p], (uint32_t)NCOLS - 1) * 8;
    packed[(size_t)q * NP16 + p] = (uint16_t)v;
  }
  if (lane < 4) {
    uint32_t v = 0;
    if (lane < (span & 3))
      v = ((nsum + lane) % CHS) * (NCOLS * 8) + min((uint32_t)c[nsum + lane], (uint32_t)NCOLS - 1) * 8;
    tails[(size_t)q * 4 + lane] = (uint16_t)v;
  }
  if (lane == 0) {
    const uint32_t key = min(begin, K - 1);
    keys[q] = key;
  }
}

template <int NCOLS, bool ACC>
__global__ void __launch_bounds__(GQ2, 4) k_preplace_sites(
    const double* __restrict__ lookup, const uint16_t* __restrict__ packed,
    const uint16_t* __restrict__ tails, const uint32_t* __restrict__ win_begin,
    const uint32_t* __restrict__ win_span, const uint32_t* __restrict__ perm,
    const Group* __restrict__ groups, uint32_t W, uint32_t B, uint32_t pitch, uint32_t NP16,
    const uint32_t* __restrict__ status, double* __restrict__ lnl,
    unsigned long long* __restrict__ segmax, uint32_t segp) {
  constexpr uint32_t ROWB = NCOLS * 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [TROWS_S][NCOLS] doubles, then accs
  double* accs = reinterpret_cast<double*>(smem + (size_t)TROWS_S * ROWB);  // [NB2_ACC][GQ2] / burst rows
  __shared__ uint32_t s_maxspan;
  __shared__ uint32_t s_qi[GQ2];        // query of thread t (burst write-out), ~0 = nothing to write
  constexpr uint32_t BSTR = GQ2 + 4;
  const uint32_t ng = status[5];
  const uint32_t NBP = ACC ? NB2_ACC_S : choose_tile(ng, B, gridDim.x);
  const uint32_t ntiles = (B + NBP - 1) / NBP;
  const int t = threadIdx.x;
  auto at = [&](uint32_t off) -> double { return *reinterpret_cast<const double*>(smem + off); };
  const ItemWalk iw = item_walk(ng * ntiles);
  for (uint32_t item = iw.pos; item < iw.end; item += iw.step) {
    const Group g = groups[item % ng];
    const uint32_t b0 = (item / ng) * NBP;
    const uint32_t nb = min(NBP, B - b0);
    __syncthreads();  // the previous item's readers of s_maxspan / accs / the tile are done
    SegTrack seg;
    if (g.count == 0) continue;
    const bool active = t < (int)g.count;
    uint32_t qi = 0, begin = 0, span = 0;
    if (active) {
      qi = perm[g.start + t];
      begin = win_begin[qi];
      span = win_span[qi];
      if ((uint64_t)begin + span > W) span = 0;
    }
    if (t == 0) s_maxspan = 0;
    s_qi[t] = (active && span > 0) ? qi : 0xffffffffu;
    __syncthreads();
    {   // one LDS atomic per wave, not per thread
      uint32_t m = span;
#pragma unroll
      for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
      if ((t & 63) == 0) atomicMax(&s_maxspan, m);
    }
    if (ACC) for (uint32_t j = 0; j < nb; ++j) accs[j * GQ2 + t] = 0.0;
    __syncthreads();
    const uint32_t gmin = win_begin[perm[g.start]];
    const uint32_t gspread = win_begin[perm[g.start + g.count - 1]] - gmin;  // < SPREAD
    const uint32_t rowoff = (begin - gmin) * ROWB;
    const uint32_t rowoff2 = rowoff | (rowoff << 16);
    const uint32_t nchunks = ACC ? (s_maxspan + CHS - 1) / CHS : 1;
    const uint32_t nsum = span & ~3u, ntail = span & 3;
    uint32_t t0 = 0, t1 = 0, t2 = 0, tailchunk = 0xffffffffu;
    if (active && ntail) {
      const uint16_t* tq = tails + (size_t)qi * 4;
      t0 = tq[0] + rowoff; t1 = tq[1] + rowoff; t2 = tq[2] + rowoff;
      tailchunk = nsum / CHS;
    }
    for (uint32_t c = 0; c < nchunks; ++c) {
      const uint32_t cbase = c * CHS;
      const bool mine = active && cbase < span;
      // full groups of four sites of this thread inside the chunk
      const uint32_t gfull = mine ? (min(nsum, cbase + CHS) > cbase ? (min(nsum, cbase + CHS) - cbase) >> 2 : 0) : 0;
      uint32_t cw[PWS];
      if (mine) {
        const uint4* p = reinterpret_cast<const uint4*>(packed + (size_t)qi * NP16 + (size_t)c * CHS);
#pragma unroll
        for (int i = 0; i < PWS / 4; ++i) {
          const uint4 v = p[i];
          cw[4 * i] = v.x + rowoff2;
          cw[4 * i + 1] = v.y + rowoff2;
          cw[4 * i + 2] = v.z + rowoff2;
          cw[4 * i + 3] = v.w + rowoff2;
        }
      } else {
#pragma unroll
        for (int i = 0; i < PWS; ++i) cw[i] = 0;
      }
      const uint32_t row0 = gmin + cbase;
      const uint32_t need = min((uint32_t)TROWS_S, gspread + min((uint32_t)CHS, s_maxspan - cbase));
      const uint32_t rows = (row0 < W) ? min(need, W - row0) : 0;
      const uint32_t n2 = rows * (NCOLS / 2);
      constexpr int PF = (TROWS_S * (NCOLS / 2) + GQ2 - 1) / GQ2;
      double2 pf[PF];
      auto request = [&](uint32_t j) {
        const double2* src = reinterpret_cast<const double2*>(lookup + ((size_t)(b0 + j) * W + row0) * NCOLS);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const uint32_t i = u * GQ2 + t;
          pf[u] = make_double2(0.0, 0.0);
          if (i < n2) pf[u] = src[i];
        }
      };
      request(0);
      for (uint32_t j = 0; j < nb; ++j) {
        __syncthreads();
        {
          double2* dst = reinterpret_cast<double2*>(smem);
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const uint32_t i = u * GQ2 + t;
            if (i < n2) dst[i] = pf[u];
          }
        }
        __syncthreads();
        if (j + 1 < nb) request(j + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (mine) {
          double sum = ACC ? accs[j * GQ2 + t] : 0.0;
#pragma unroll
          for (int i = 0; i < PWS; ++i) asm volatile("" : "+v"(cw[i]));
          constexpr int WPB = 4, NBATCH = PWS / WPB;  // 4 words = 8 sites = 2 groups per batch
          double rb[2][WPB * 2];
          auto issue = [&](int bt) {
#pragma unroll
            for (int w = 0; w < WPB; ++w) {
              const uint32_t v = cw[bt * WPB + w];
              rb[bt & 1][2 * w] = at(v & 0xffffu);
              rb[bt & 1][2 * w + 1] = at(v >> 16);
            }
          };
          issue(0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int bt = 0; bt < NBATCH; ++bt) {
            gather_prio<NBATCH>(bt);
            if (bt + 1 < NBATCH) issue(bt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
              const double* v = &rb[bt & 1][gq * 4];
              double s1 = v[0] + v[1];
              const double s2 = v[2] + v[3];
              s1 += s2;
              sum += ((uint32_t)(bt * 2 + gq) < gfull) ? s1 : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (c == tailchunk) {  // singles of the window tail, in order
            sum += at(t0);
            if (ntail > 1) sum += at(t1);
            if (ntail > 2) sum += at(t2);
          }
          if (ACC) {
            accs[j * GQ2 + t] = sum;
          } else {
            accs[(j & 7u) * BSTR + t] = sum;
            if (segmax) seg.add(segmax, segp, qi, b0 + j, j + 1 == nb, sum);
          }
        }
        if (!ACC && ((j & 7u) == 7u || j + 1 == nb)) {
          // bursts of 8 branches, eight lanes per query: whole 64-byte sectors (see k_preplace_pairs)
          __builtin_amdgcn_wave_barrier();
          const uint32_t lane = (uint32_t)t & 63u, wbase = (uint32_t)t & ~63u, col = lane & 7u;
          const uint32_t ncol = (j & 7u) + 1u, bcol = b0 + (j & ~7u) + col;
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t tq = wbase + k * 8 + (lane >> 3);
            const uint32_t q = s_qi[tq];
            if (q != 0xffffffffu && col < ncol) lnl[(size_t)q * pitch + bcol] = accs[col * BSTR + tq];
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (ACC && active) {
      double* out = lnl + (size_t)qi * pitch + b0;
      for (uint32_t j = 0; j < nb; ++j) out[j] = accs[j * GQ2 + t];
    }
  }  // work items
}

// ---------------------------------------------------------------------------------------------
// k_select: dynamic heuristic on device (apply_heuristic -> dynamic_heuristic,
// src/core/heuristics.hpp:40-68; compute_and_set_lwr src/set_manipulators.cpp:43-69;
// until_accumulated_reached :90-114).  One wave per query, the row of B log-likelihoods in
// registers (NR values per lane); the largest remaining LWR is extracted until the running sum
// reaches `threshold` (the crossing element is included, min 1).  Ties: lowest branch id first
// (the reference's std::sort is unstable).  Selections go to stage[q][0..cap).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_add(double v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Stop rule of the per-query extraction loop (descending lnL == descending LWR, ties by branch id):
//   0 dynamic   until the accumulated LWR reaches thr, the crossing element included
//               (until_accumulated_reached, src/set_manipulators.cpp:90-114)
//   1 fixed     the ceil(x * B) best (until_top_percent, :82-88); limit precomputed by the host
//   2 baseball  every branch within 3.0 lnL of the best ("strike box"), plus min(40 - hits, 6)
//               more -- 6 when more than 40 were hit: the reference's unsigned wrap-around
//               (src/core/heuristics.hpp:70-117; quirk D7: clamped at B)
// Where a selected (query, branch) goes.  Bitmap form (the usual one): bit q of row b of a
// [B][ceil(Q / 32)] bitmap plus a per-branch counter -- the candidate list in the reference's Work
// order (branch-major, queries ascending: src/core/Work.hpp:21-113) is then one scan over the B
// counters and one pass over the bitmap (k_emit_pairs), no per-query staging rows, no compaction, no
// device sort.  Staging form (bitmap over 64 MB): key (branch << 32 | query) in stage[q][0..cap).
struct SelOut {
  unsigned long long* stage;
  uint32_t cap;
  uint32_t* bitmap;
  uint32_t* bcount;
  uint32_t wpr;     // bitmap words per branch row
  __device__ __forceinline__ void put(uint32_t q, uint32_t bi, uint32_t taken, uint32_t* status) const {
    if (bitmap) {
      atomicOr(&bitmap[(size_t)bi * wpr + (q >> 5)], 1u << (q & 31u));
      atomicAdd(&bcount[bi], 1u);
    } else if (taken < cap) {
      stage[(size_t)q * cap + taken] = ((unsigned long long)bi << 32) | q;
    } else {
      atomicMax(&status[2], taken + 1);  // staging row too short: caller retries wider
    }
  }
  __device__ __forceinline__ uint32_t count(uint32_t taken) const { return bitmap ? taken : min(taken, cap); }
};

struct SelRule {
  int mode;
  double thr;
  uint32_t limit;
  double sum = 0.0;
  uint32_t hits = 0;
  bool striking = true;
  const double* e2t = nullptr;   // LDS table of exp_tab (wave_util.hpp), or null: library exp
  __device__ __forceinline__ SelRule(int m, double t, uint32_t l, const double* tab = nullptr) : mode(m), thr(t), limit(l), e2t(tab) {}
  // exp(x), x <= 0; far below the underflow threshold either way once x < -800
  __device__ __forceinline__ double ex(double x) const {
    if (!e2t) return exp(x);
    return x == 0.0 ? 1.0 : epa_wave::exp_tab(fmax(x, -800.0), e2t);   // exp_tab(0) == 1 as well: the branch only saves the work
  }
  __device__ __forceinline__ bool more(uint32_t taken, uint32_t B) const {
    if (mode == 0) return taken < B && sum < thr;
    if (mode == 1) return taken < limit;
    return taken < B && (striking || taken < limit);
  }
  // the next best value: take it?  (wave / workgroup uniform)
  __device__ __forceinline__ bool accept(double best, double mx, double tot, uint32_t taken) {
    if (mode == 0) { sum += ex(best - mx) / tot; return true; }
    if (mode == 1) return true;
    if (striking) {
      if (!(best < mx - 3.0)) { ++hits; return true; }
      striking = false;
      // std::min(max_pitches - hits, max_strikes) in size_t arithmetic (heuristics.hpp:107): with
      // more than 40 hits the difference wraps around and 6 more are added, with exactly 40 none
      limit = hits + (hits > 40u ? 6u : min(40u - hits, 6u));
    }
    return taken < limit;
  }
};

template <int NR>
__global__ void __launch_bounds__(256) k_select(const double* __restrict__ lnl, uint32_t Q, uint32_t B, uint32_t pitch,
                                                double threshold, int mode, uint32_t limit, SelOut so,
                                                uint32_t* __restrict__ counts,
                                                uint32_t* __restrict__ status) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= Q) return;
  const double* src = lnl + (size_t)q * pitch;
  double v[NR];
  double mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint32_t i = r * 64 + lane;
    v[r] = i < B ? src[i] : -INFINITY;
    mx = fmax(mx, v[r]);
  }
  // wave reductions on the DPP ladder (no LDS round trips: the extraction loop below was 18
  // ds_bpermute per candidate), results wave-uniform
  mx = epa_wave::wave_max_d(mx);
  // (the table-driven exp_tab instead of the library exp: 0.283 -> 0.296 ms, the kernel is not bound by them)
  SelRule rule(mode, threshold, limit);
  double tot = 0.0;
#pragma unroll
  for (int r = 0; r < NR; ++r) tot += exp(v[r] - mx);  // exp(-inf) == 0 for the padding
  tot = epa_wave::wave_sum(tot);
  uint32_t taken = 0;
  while (rule.more(taken, B)) {
    double lbest = -INFINITY;
    uint32_t lbi = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (v[r] > lbest) { lbest = v[r]; lbi = r * 64 + lane; }
    // the largest value, ties by the lowest branch id
    const double best = epa_wave::wave_max_d(lbest);
    const uint32_t bi = epa_wave::wave_min_u((lbest == best && lbest > -INFINITY) ? lbi : 0xffffffffu);
    if (bi == 0xffffffffu) break;
    if (!rule.accept(best, mx, tot, taken)) break;
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if ((uint32_t)(r * 64 + lane) == bi) v[r] = -INFINITY;
    if (lane == 0) so.put(q, bi, taken, status);
    ++taken;
  }
  if (lane == 0) counts[q] = so.count(taken);
}

// The dynamic rule from the per-segment maxima the preplacement left behind (seg_key above): wave per query,
// lane = 64-branch segment.  With tot >= 1 (the maximum itself) a candidate has lnL >= mx + log((1 - thr) / B)
// -- everything below sums to less than 1 - thr -- and a term below mx - (log B + 38) is less than 2^-54 / B of
// tot: all B of them together cannot move its double.  So only the segments whose maximum lies inside those
// bands are read: 1.5 of 16 on average at cfg2 (tests/...) -- 0.13 GB instead of the 0.82 GB row
// read per 100k-read chunk.  The candidate segments are held in registers (2 -- the usual case -- or up to NRN);
// more (or a row without published maxima: its query went through another preplacement kernel) are streamed per
// extraction.
// Round 5 (the kernel was bound by its own vector instructions: 820 per query, 82 M per 100k reads = the 147 us it
// took, profiles/r5_pmc_summary.txt): the two band widths come from the host (two library logs per wave), the
// exponentials are exp_tab (16 instructions instead of ~45; exp_tab(0) = 1 exactly and the first candidate -- the
// row maximum -- skips it), rows with at most two candidate segments run a two-register extraction loop, a wave
// walks several queries (grid-stride, the next query's maxima requested ahead), and the span-class histogram of
// the candidate counts (k_class_hist: 1563 atomics on one word, 20 us) is summed per workgroup in LDS.
template <int N>
__device__ __forceinline__ uint32_t seg_extract(const double* __restrict__ src, uint32_t lane, uint32_t B,
                                                unsigned long long mask_x, double mx, double tot, SelRule& rule,
                                                const SelOut& so, uint32_t q, uint32_t* __restrict__ status) {
  double v[N];
  uint32_t base[N];
  unsigned long long mm = mask_x;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    if (mm) {
      const uint32_t s = (uint32_t)__builtin_ctzll(mm);
      mm &= mm - 1;
      base[r] = s * 64;
      const uint32_t i = s * 64 + lane;
      v[r] = i < B ? src[i] : -INFINITY;
    } else {
      base[r] = 0;
      v[r] = -INFINITY;
    }
  }
  uint32_t taken = 0, cand = 0;
  while (rule.more(taken, B)) {
    double lbest = -INFINITY;
    uint32_t lbi = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < N; ++r)   // segments ascend with r: the first maximum has the lowest branch id
      if (v[r] > lbest) { lbest = v[r]; lbi = base[r] + lane; }
    const double best = epa_wave::wave_max_d(lbest);
    const uint32_t bi = epa_wave::wave_min_u((lbest == best && lbest > -INFINITY) ? lbi : 0xffffffffu);
    if (bi == 0xffffffffu) break;
    if (!rule.accept(best, mx, tot, taken)) break;
#pragma unroll
    for (int r = 0; r < N; ++r)
      if (base[r] + lane == bi) v[r] = -INFINITY;
    // candidate k waits in lane k: the atomics of a query leave together, one instruction per kind, after its loop
    // (one pair of them per candidate from lane 0 sat between the next query's loads and their wait)
    if (taken < 64u) { if (lane == taken) cand = bi; }
    else if (lane == 0) so.put(q, bi, taken, status);
    ++taken;
  }
  if (lane < min(taken, 64u)) so.put(q, cand, lane, status);
  return taken;
}

template <int NRN>
__global__ void __launch_bounds__(256) k_select_seg(const double* __restrict__ lnl, const unsigned long long* __restrict__ segmax,
                                                    uint32_t segp, uint32_t Q, uint32_t B, uint32_t pitch, double threshold,
                                                    double band_x, double band_t,
                                                    SelOut so, uint32_t* __restrict__ counts, uint32_t* __restrict__ status,
                                                    const uint32_t* __restrict__ win_span, int states,
                                                    uint32_t* __restrict__ hist) {
  __shared__ double s_e2t[64];
  __shared__ uint32_t s_hist[EPA_N_CLS];
  if (threadIdx.x < 64) s_e2t[threadIdx.x] = exp2((double)threadIdx.x * 0.015625);
  if (threadIdx.x < EPA_N_CLS) s_hist[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t nseg = (B + 63) >> 6;   // <= 64
  const uint32_t stride = gridDim.x * 4;
  uint32_t q = blockIdx.x * 4 + wv;
  unsigned long long key_next = (q < Q && lane < nseg) ? segmax[(size_t)q * segp + lane] : ~0ull;
  for (; q < Q; q += stride) {
    const unsigned long long key = key_next;
    {
      const uint32_t qn = q + stride;
      key_next = (qn < Q && lane < nseg) ? segmax[(size_t)qn * segp + lane] : ~0ull;
    }
    const double* src = lnl + (size_t)q * pitch;
    auto value = [&](uint32_t s) -> double {   // this lane's element of segment s (-inf past the row)
      const uint32_t i = s * 64 + lane;
      return i < B ? src[i] : -INFINITY;
    };
    double m = -INFINITY;
    if (__ballot(key == 0ull) == 0ull) {
      if (lane < nseg) m = seg_val(key);
    } else {   // no maxima for this row: build them from the row itself
      for (uint32_t s = 0; s < nseg; ++s) {
        const double xm = epa_wave::wave_max_d(value(s));
        if (lane == s) m = xm;
      }
    }
    const double mx = epa_wave::wave_max_d(m);
    const unsigned long long mask_t = __ballot(m >= mx - band_t), mask_x = __ballot(m >= mx - band_x);
    double tot = 0.0;
    for (unsigned long long mm = mask_t; mm; mm &= mm - 1)
      tot += epa_wave::exp_tab(fmax(value((uint32_t)__builtin_ctzll(mm)) - mx, -800.0), s_e2t);
    tot = epa_wave::wave_sum(tot);
    SelRule rule(0, threshold, 0u, s_e2t);
    uint32_t taken = 0;
    const int ncs = __popcll(mask_x);
    if (ncs <= 2) {
      taken = seg_extract<2>(src, lane, B, mask_x, mx, tot, rule, so, q, status);
    } else if (ncs <= NRN) {
      taken = seg_extract<NRN>(src, lane, B, mask_x, mx, tot, rule, so, q, status);
    } else {
      // many candidate segments: every extraction streams them again and takes the next element in
      // (lnL descending, branch ascending) order after the previous one
      double pbest = INFINITY;
      uint32_t pbi = 0, cand = 0;
      while (rule.more(taken, B)) {
        double lbest = -INFINITY;
        uint32_t lbi = 0xffffffffu;
        for (unsigned long long mm = mask_x; mm; mm &= mm - 1) {
          const uint32_t s = (uint32_t)__builtin_ctzll(mm), i = s * 64 + lane;
          const double x = value(s);
          const bool after = x < pbest || (x == pbest && i > pbi);
          if (after && x > lbest) { lbest = x; lbi = i; }
        }
        const double best = epa_wave::wave_max_d(lbest);
        const uint32_t bi = epa_wave::wave_min_u((lbest == best && lbest > -INFINITY) ? lbi : 0xffffffffu);
        if (bi == 0xffffffffu) break;
        if (!rule.accept(best, mx, tot, taken)) break;
        pbest = best;
        pbi = bi;
        if (taken < 64u) { if (lane == taken) cand = bi; }
        else if (lane == 0) so.put(q, bi, taken, status);
        ++taken;
      }
      if (lane < min(taken, 64u)) so.put(q, cand, lane, status);
    }
    if (lane == 0) {
      const uint32_t n = so.count(taken);
      counts[q] = n;
      if (hist && n) atomicAdd(&s_hist[epa_span_class(states, win_span[q])], n);
    }
  }
  __syncthreads();
  if (hist && threadIdx.x < EPA_N_CLS && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

// (the round-4 forms of k_select_seg / k_pack_pairs: profiles/variants/r5_preplace_db_duo_v1.hip; A/Bs in profiles/r5_select_pack_ab*.txt)

// Same selection, workgroup per query (4 waves): the row of up to 256 x NRT branches lives in the
// registers of the whole workgroup (element i in thread i % 256, slot i / 256), so the table is
// read exactly once; wave-level (max, argmax, sum) results are combined across the four waves
// through LDS.  Used for 4096 < B <= 16384.
template <int NRT>
__global__ void __launch_bounds__(256) k_select_wg(const double* __restrict__ lnl, uint32_t Q, uint32_t B, uint32_t pitch,
                                                   double threshold, int mode, uint32_t limit, SelOut so,
                                                   uint32_t* __restrict__ counts,
                                                   uint32_t* __restrict__ status) {
  __shared__ double s_val[2][4];
  __shared__ uint32_t s_idx[2][4];
  const uint32_t q = blockIdx.x;
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const double* src = lnl + (size_t)q * pitch;
  double v[NRT];
  double mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < NRT; ++r) {
    const uint32_t i = r * 256 + t;
    v[r] = i < B ? src[i] : -INFINITY;
    mx = fmax(mx, v[r]);
  }
  int ph = 0;
  auto block_max = [&](double x) {
    x = wave_max(x);
    if (lane == 0) s_val[ph][wv] = x;
    __syncthreads();
    x = fmax(fmax(s_val[ph][0], s_val[ph][1]), fmax(s_val[ph][2], s_val[ph][3]));
    ph ^= 1;
    return x;
  };
  mx = block_max(mx);
  double tot = 0.0;
#pragma unroll
  for (int r = 0; r < NRT; ++r) tot += exp(v[r] - mx);  // exp(-inf) == 0 for the padding
  tot = wave_add(tot);
  if (lane == 0) s_val[ph][wv] = tot;
  __syncthreads();
  tot = (s_val[ph][0] + s_val[ph][1]) + (s_val[ph][2] + s_val[ph][3]);
  ph ^= 1;
  SelRule rule(mode, threshold, limit);
  uint32_t taken = 0;
  while (rule.more(taken, B)) {
    double best = -INFINITY;
    uint32_t bi = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < NRT; ++r)
      if (v[r] > best) { best = v[r]; bi = r * 256 + t; }
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
    if (!rule.accept(best, mx, tot, taken)) break;
#pragma unroll
    for (int r = 0; r < NRT; ++r)
      if ((uint32_t)(r * 256) + t == bi) v[r] = -INFINITY;
    if (t == 0) so.put(q, bi, taken, status);
    ++taken;
  }
  if (t == 0) counts[q] = so.count(taken);
}

// Same selection for references with more than 16384 branches: the row does not fit the register
// file, so it is streamed from HBM/L2 once per pass (max, total, then one pass per selected
// branch: 3-4 on typical data); taken branches are remembered in a per-lane bitmask
// (element i lives in lane i % 64, bit i / 64; up to 64 x 64 x NW branches).
template <int NW>
__global__ void __launch_bounds__(256) k_select_big(const double* __restrict__ lnl, uint32_t Q, uint32_t B, uint32_t pitch,
                                                    double threshold, int mode, uint32_t limit, SelOut so,
                                                    uint32_t* __restrict__ counts,
                                                    uint32_t* __restrict__ status) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (q >= Q) return;
  const double* src = lnl + (size_t)q * pitch;
  unsigned long long takenmask[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) takenmask[w] = 0ull;
  double mx = -INFINITY;
  for (uint32_t i = lane; i < B; i += 64) mx = fmax(mx, src[i]);
  mx = wave_max(mx);
  double tot = 0.0;
  for (uint32_t i = lane; i < B; i += 64) tot += exp(src[i] - mx);
  tot = wave_add(tot);
  SelRule rule(mode, threshold, limit);
  uint32_t taken = 0;
  while (rule.more(taken, B)) {
    double best = -INFINITY;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = lane, r = 0; i < B; i += 64, ++r) {
      const double v = src[i];
      const bool free_ = !((takenmask[r >> 6] >> (r & 63)) & 1ull);
      if (free_ && v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const uint32_t oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0xffffffffu) break;
    if (!rule.accept(best, mx, tot, taken)) break;
    if ((bi & 63u) == lane) {
      const uint32_t r = bi >> 6;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if ((uint32_t)w == (r >> 6)) takenmask[w] |= 1ull << (r & 63);
    }
    if (lane == 0) so.put(q, bi, taken, status);
    ++taken;
  }
  if (lane == 0) counts[q] = so.count(taken);
}

// span-class histogram of the selected pairs: sum of the per-query candidate counts by the class
// of the query's window (one atomic per wave and class present)
__global__ void __launch_bounds__(256) k_class_hist(const uint32_t* __restrict__ counts,
                                                   const uint32_t* __restrict__ win_span, uint32_t Q,
                                                   int states, uint32_t* __restrict__ hist) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  const int cls = q < Q ? epa_span_class(states, win_span[q]) : -1;
  const uint32_t n = q < Q ? counts[q] : 0;
  for (int c = 0; c < EPA_N_CLS; ++c) {
    if (__ballot(cls == c) == 0ull) continue;
    uint32_t v = cls == c ? n : 0;
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&hist[c], v);
  }
}

// everything the host reads after the selection, in one block: out[0] = total, out[1 ..] = the
// selection's status words + class histogram, out[32 .. 35] = the preplacement's window validation
__global__ void __launch_bounds__(64) k_pack_readback(const uint32_t* __restrict__ status,
                                                     const uint32_t* __restrict__ total,
                                                     const uint32_t* __restrict__ pre_status,
                                                     uint32_t* __restrict__ out) {
  const uint32_t t = threadIdx.x;
  if (t == 0) out[0] = *total;
  if (t < 8 + EPA_N_CLS) out[1 + t] = status[t];
  if (t < 4) out[32 + t] = pre_status ? pre_status[t] : 0u;
}

__global__ void __launch_bounds__(256) k_compact(const unsigned long long* __restrict__ stage,
                                                 const uint32_t* __restrict__ counts,
                                                 const uint32_t* __restrict__ offsets, uint32_t Q,
                                                 uint32_t cap, unsigned long long* __restrict__ keys) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= Q) return;
  const uint32_t n = counts[q], o = offsets[q];
  for (uint32_t i = lane; i < n; i += 64) keys[o + i] = stage[(size_t)q * cap + i];
}

// Bitmap form of the selection: workgroup b walks row b of the bitmap in chunks of 256 words and
// writes the set bits -- queries ascending -- as pairs behind boffs[b] (exclusive scan of the per-
// branch counters): the candidate list in (branch, query) order.
__global__ void __launch_bounds__(256) k_emit_pairs(const uint32_t* __restrict__ bitmap, uint32_t wpr,
                                                    const uint32_t* __restrict__ boffs,
                                                    epa_pair* __restrict__ pairs, uint32_t nb, uint32_t max_pairs) {
  __shared__ uint32_t wsum[4];
  const uint32_t b = blockIdx.x, t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  if (boffs[nb] > max_pairs) return;   // queued ahead of the host's overflow check (launch_select_emit): the list would not fit
  uint32_t base = boffs[b];
  const uint32_t end = boffs[b + 1];
  if (base == end) return;
  const uint32_t* row = bitmap + (size_t)b * wpr;
  for (uint32_t w0 = 0; w0 < wpr && base < end; w0 += 256) {
    const uint32_t wi = w0 + t;
    uint32_t bits = wi < wpr ? row[wi] : 0u;
    const uint32_t c = (uint32_t)__popc(bits);
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o);
      if ((int)lane >= o) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      if (w < wv) before += wsum[w];
      total += wsum[w];
    }
    uint32_t o = base + before + inc - c;
    while (bits) {
      const uint32_t k = (uint32_t)__ffs((int)bits) - 1u;
      bits &= bits - 1u;
      pairs[o].branch_id = b;
      pairs[o].seq_id = wi * 32u + k;
      ++o;
    }
    base += total;
    __syncthreads();
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

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

int launch_build_lookup2(epa_ctx* ctx) {
  if (!ctx->lookup2)
    EPA_HIP(ctx, hipMalloc(&ctx->lookup2, sizeof(double) * (size_t)ctx->B * 2 * ((ctx->W + 1) / 2) * PE));
  dim3 grid((ctx->W * PE + 255) / 256, ctx->B);
  hipLaunchKernelGGL(k_build_lookup2, grid, dim3(256), 0, ctx->stream, ctx->lookup, ctx->W, ctx->lookup2);
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}

int launch_preplace(epa_ctx* ctx, const uint8_t* d_codes, const uint32_t* d_begin,
                    const uint32_t* d_span, uint32_t Q, double* d_lnl, uint32_t max_span) {
  const uint32_t pitch = ctx->lnl_pitch ? ctx->lnl_pitch : ctx->B;  // row pitch of d_lnl in doubles
  unsigned long long* const segmax = ctx->segmax;   // per-(query, 64-branch segment) maxima wanted by the fused chunk body, or null
  const uint32_t segp = ctx->segp;
  const bool pairs = ctx->s == 4 && ctx->lookup2 && !ctx->opt.preplace_generic;
  const bool sites = ctx->s == 20 && ctx->ncols == 24 && !ctx->opt.preplace_generic;
  const uint32_t crel = ctx->code_stride ? 1u : 0u, cstride = crel ? ctx->code_stride : ctx->W;
  const uint32_t n_buckets = (ctx->W + SPREAD - 1) / SPREAD;
  const uint32_t Wp = n_buckets * SPREAD;  // key space of one (class, parity) block
  const uint32_t n_blocks = pairs ? 4 : 1, class_blocks = pairs ? 2 : 1;
  const uint32_t gq0 = (pairs || sites) ? GQ2 : GQ, gq1 = GQ;
  const uint32_t max_runs = (Q + GQ - 1) / GQ + n_blocks;
  const uint32_t max_groups = max_runs + n_blocks * n_buckets;  // + one split per SPREAD boundary
  // pair offsets per query: whole chunks of CP, enough for the longest window
  const uint32_t span_bound = (max_span == 0 || max_span > ctx->W) ? ctx->W : max_span;
  const uint32_t NP16 = pairs ? ((span_bound + 1) / 2 + CP - 1) / CP * CP
                        : sites ? (span_bound + CHS - 1) / CHS * CHS : 0;
  // scratch 6: [status 256 B | key counts K | iota / rank Q | sorted_keys Q | perm Q | keys Q | groups | packed | tails | rocprim temp]
  const uint32_t K = pairs ? n_blocks * Wp : sites ? ctx->W + 1 : 0;   // key space of the counting sort (fast paths)
  const size_t hb = align256(sizeof(uint32_t) * (size_t)K);
  size_t temp_bytes = 0;
  (void)rocprim::radix_sort_pairs<epa_radix_cfg>(nullptr, temp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, (uint32_t*)nullptr, Q, 0, 32, ctx->stream);
  const size_t qb = align256(sizeof(uint32_t) * Q);
  const size_t gb = align256(sizeof(Group) * max_groups);
  const size_t pb = align256(sizeof(uint16_t) * (size_t)Q * NP16);
  const size_t tb = (pairs || sites) ? align256(sizeof(uint16_t) * 4 * (size_t)Q) : 0;
  const size_t need = 256 + hb + 4 * qb + gb + pb + tb + temp_bytes;
  char* base0 = (char*)epa_scratch(ctx, 6, need);
  if (!base0) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(preplace scratch)");
  uint32_t* status = reinterpret_cast<uint32_t*>(base0);
  uint32_t* key_cnt = reinterpret_cast<uint32_t*>(base0 + 256);
  char* base = base0 + hb;
  uint32_t* iota = reinterpret_cast<uint32_t*>(base + 256);   // the counting sort keeps the arrival ranks here
  uint32_t* sorted_keys = reinterpret_cast<uint32_t*>(base + 256 + qb);
  uint32_t* perm = reinterpret_cast<uint32_t*>(base + 256 + 2 * qb);
  uint32_t* keys = reinterpret_cast<uint32_t*>(base + 256 + 3 * qb);
  Group* groups = reinterpret_cast<Group*>(base + 256 + 4 * qb);
  uint16_t* packed = reinterpret_cast<uint16_t*>(base + 256 + 4 * qb + gb);
  uint16_t* tails = reinterpret_cast<uint16_t*>(base + 256 + 4 * qb + gb + pb);
  void* temp = base + 256 + 4 * qb + gb + pb + tb;
  ctx->d_status = status;
  {   // status words + key counts, and the fused chunk body's segment maxima: one kernel
    const size_t zb = ctx->segmax_zero_bytes;
    ctx->segmax_zero_bytes = 0;
    const int zr = epa_zero_async(ctx, status, 256 + hb, zb ? ctx->segmax : nullptr, zb);
    if (zr) return zr;
  }
  // wide slices for the pair path when a chunk puts few reads on a window start (see k_preplace_pairs)
  const bool acc = max_span == 0 || max_span > (uint32_t)CH;
  const bool wide = pairs && !acc && (uint64_t)Q * SPREAD < (uint64_t)1400 * ctx->W;   // < ~700 reads per 96-site bucket and parity
  const uint32_t rowl = (wide || acc) ? ROWL_PACKED : ROWL_NARROW;  // LDS row stride the 16-bit offsets are built for
  if (pairs) {
    constexpr uint32_t pack_grid = 64;   // workgroups per CU of the persistent packing grid
    hipLaunchKernelGGL(k_pack_pairs, dim3(std::min<uint32_t>((Q + 15) / 16, (uint32_t)ctx->n_cu * pack_grid)), dim3(256), 0, ctx->stream, d_codes, d_begin,
                       d_span, Q, ctx->W, cstride, crel, span_bound, Wp, NP16, rowl, packed, tails, keys, K, status);
  } else if (sites) {
    hipLaunchKernelGGL(k_pack_sites<24>, dim3((Q + 3) / 4), dim3(256), 0, ctx->stream, d_codes, d_begin,
                       d_span, Q, ctx->W, cstride, crel, span_bound, NP16, packed, tails, keys, K, status);
  }
  if (pairs || sites) {
    hipLaunchKernelGGL(k_count_keys, dim3((Q + 255) / 256), dim3(256), 0, ctx->stream, keys, Q, key_cnt, iota);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, ctx->stream, key_cnt, K);
    hipLaunchKernelGGL(k_scatter_sorted, dim3((Q + 255) / 256), dim3(256), 0, ctx->stream, keys, iota, key_cnt, Q,
                       sorted_keys, perm);
  } else {
    hipLaunchKernelGGL(k_iota, dim3((Q + 255) / 256), dim3(256), 0, ctx->stream, iota, Q);
    hipLaunchKernelGGL(k_validate, dim3((Q + 255) / 256), dim3(256), 0, ctx->stream, d_begin, d_span, Q,
                       ctx->W, std::min(span_bound, crel ? cstride : 0xffffffffu), status);
    int wbits = 1;
    while (wbits < 32 && (1ull << wbits) <= (uint64_t)ctx->W) ++wbits;
    EPA_HIP(ctx, rocprim::radix_sort_pairs<epa_radix_cfg>(temp, temp_bytes, d_begin, sorted_keys, iota, perm, Q, 0, wbits,
                                           ctx->stream));
  }
  constexpr int SPREAD_WIDE = 288;
  hipLaunchKernelGGL(k_make_groups, dim3(1), dim3(256),
                     sizeof(uint32_t) * (2 * (n_blocks + 1) + 4 * (size_t)max_runs + 2), ctx->stream,
                     sorted_keys, Q, pairs ? Wp : 0xffffffffu, n_blocks, class_blocks, gq0, gq1,
                     (uint32_t)(wide ? SPREAD_WIDE : SPREAD), max_runs, groups, max_groups, status);
  // persistent grids: every resident workgroup slot of the device, work items strided over them
  const uint32_t ntiles = (ctx->B + NB - 1) / NB;
  // max_span: upper bound of the window spans when the caller knows it (0 = unknown) -> acc, above
  const size_t lds = sizeof(double) * ((size_t)TROWS * ctx->ncols + (acc ? (size_t)NB * GQ : 0));
  const size_t lds2 = (size_t)TROWS2 * rowl + sizeof(double) * (acc ? NB2_ACC * GQ2 : NB2_BURST * (GQ2 + 4));  // accs / result staging
  const uint32_t ntiles2 = (ctx->B + (acc ? NB2_ACC : 16) - 1) / (acc ? NB2_ACC : 16);   // single chunk: choose_tile, >= 16
  const dim3 grid2((uint32_t)std::min<uint64_t>((uint64_t)max_groups * ntiles2, (uint64_t)ctx->n_cu));  // 1 per CU
  // generic kernel: with the pair path on it only sees the few groups of queries with rare
  // ambiguity codes -> persistent grid; as the only kernel (20 states) one workgroup per item,
  // dispatched dynamically (items differ in cost, partial groups are cheaper)
  const dim3 grid(pairs ? (uint32_t)std::min<uint64_t>((uint64_t)max_groups * ntiles, (uint64_t)ctx->n_cu * (acc ? 2 : 4))
                        : max_groups * ntiles);
  const size_t codes_bytes = (size_t)Q * cstride;
  const uint32_t want_cls = pairs ? 1u : 0u;
  epa_timer_start(ctx, epa_t(ctx, epa_ctx::T_PREPLACE));
#define PRE2(A, SP, RL, LDSB)                                                                        \
  do {                                                                                               \
    EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_preplace_pairs<A, SP, RL>,                       \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDSB)));      \
    hipLaunchKernelGGL((k_preplace_pairs<A, SP, RL>), grid2, dim3(GQ2), (uint32_t)(LDSB), ctx->stream, ctx->lookup2, (const uint16_t*)packed, \
                       (const uint16_t*)tails, d_begin, d_span, perm, groups, ctx->W, ctx->B, pitch, NP16, status, d_lnl, (A) ? nullptr : segmax, segp);    \
  } while (0)
  if (pairs && wide) {
    const size_t lds2w = (size_t)((CH + SPREAD_WIDE) / 2) * ROWL_PACKED + sizeof(double) * NB2_BURST * (GQ2 + 4);
    PRE2(false, SPREAD_WIDE, ROWL_PACKED, lds2w);
  } else if (pairs) {
    const size_t lds_stag = (size_t)TROWS2 * ROWL_NARROW + sizeof(double) * NB2_RING * (GQ2 + 4);
    if (acc) PRE2(true, SPREAD, ROWL_PACKED, lds2);
    else if (ctx->opt.preplace_stagger && lds_stag + 4352 <= 163840) {
      EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_preplace_pairs<false, SPREAD, ROWL_NARROW, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stag));
      hipLaunchKernelGGL((k_preplace_pairs<false, SPREAD, ROWL_NARROW, true>), grid2, dim3(GQ2), (uint32_t)lds_stag, ctx->stream, ctx->lookup2,
                         (const uint16_t*)packed, (const uint16_t*)tails, d_begin, d_span, perm, groups, ctx->W, ctx->B, pitch, NP16, status,
                         d_lnl, segmax, segp);
    } else PRE2(false, SPREAD, ROWL_NARROW, lds2);
  }
#undef PRE2
  if (sites) {
    const bool acc_s = max_span == 0 || max_span > (uint32_t)CHS;
    const size_t lds_s = (size_t)TROWS_S * 24 * 8 + sizeof(double) * (acc_s ? NB2_ACC_S * GQ2 : NB2_BURST * (GQ2 + 4));  // accs / result staging
    const uint32_t ntiles_s = (ctx->B + (acc_s ? NB2_ACC_S : 16) - 1) / (acc_s ? NB2_ACC_S : 16);
    const dim3 grid_s((uint32_t)std::min<uint64_t>((uint64_t)max_groups * ntiles_s, (uint64_t)ctx->n_cu));
#define PRES(A)                                                                                      \
  do {                                                                                               \
    EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_preplace_sites<24, A>,                           \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));       \
    hipLaunchKernelGGL((k_preplace_sites<24, A>), grid_s, dim3(GQ2), (uint32_t)lds_s, ctx->stream, ctx->lookup, \
                       packed, tails, d_begin, d_span, perm, groups, ctx->W, ctx->B, pitch, NP16, status, d_lnl, (A) ? nullptr : segmax, segp); \
  } while (0)
    if (acc_s) PRES(true); else PRES(false);
#undef PRES
    epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_PREPLACE));
    EPA_HIP(ctx, hipGetLastError());
    return EPA_OK;
  }
#define PRE(NC, A)                                                                                  \
  do {                                                                                              \
    EPA_HIP(ctx, hipFuncSetAttribute((const void*)k_preplace<NC, A>,                                \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
    hipLaunchKernelGGL((k_preplace<NC, A>), grid, dim3(GQ), (uint32_t)lds, ctx->stream, ctx->lookup, d_codes, \
                       d_begin, d_span, perm, groups, ctx->W, cstride, crel, ctx->B, pitch, codes_bytes, want_cls, status, d_lnl); \
  } while (0)
  if (ctx->ncols == 16) { if (acc) PRE(16, true); else PRE(16, false); }
  else { if (acc) PRE(24, true); else PRE(24, false); }
#undef PRE
  epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_PREPLACE));
  EPA_HIP(ctx, hipGetLastError());
  return EPA_OK;
}

// reads the window-validation words written by k_make_groups (call after a stream sync)
int preplace_check_status(epa_ctx* ctx) {
  if (!ctx->d_status) return EPA_OK;
  uint32_t st[4];
  EPA_HIP(ctx, hipMemcpy(st, ctx->d_status, sizeof(st), hipMemcpyDeviceToHost));
  if (st[0])
    return epa_fail(ctx, EPA_ERR_QUERY_ALL_GAP, "Sequence " + std::to_string(st[0] & 0x7fffffffu) +
                                                    " does not appear to have any non-gap sites!");
  if (st[1])
    return epa_fail(ctx, EPA_ERR_QUERY_WIDTH, "Query sequence length not same as reference alignment!");
  return EPA_OK;
}

// The candidate selection in two halves, so that a caller with something else to queue (the chunk
// pipeline: the other slot's work) does not sit in the wait for the candidate count:
//   begin  queues k_select, the offsets scan, the span-class histogram and the read-back of
//          {total, status words, class histogram, window-validation words of the preplacement} into
//          `rb` (pinned host memory for a truly asynchronous copy; 64 words) -- no synchronisation;
//   end    waits for the stream, widens the staging rows and repeats on overflow (rare), then queues
//          compaction, the stable sort into Work order and the key -> pair conversion.
int launch_select_begin(epa_ctx* ctx, const double* d_lnl, uint32_t Q, double threshold, epa_pair* d_pairs,
                        uint64_t max_pairs, const uint32_t* d_span, uint32_t* rb, SelectPending* sp) {
  ctx->cls_hist_pairs = 0;
  const uint32_t B = ctx->B;
  const uint32_t pitch = ctx->lnl_pitch ? ctx->lnl_pitch : B;  // row pitch of d_lnl in doubles
  if (!sp->rerun) sp->pre_status = (const uint32_t*)ctx->d_status;
  if (B > 64 * 64 * 16)
    return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "select_candidates: more than 65536 branches");
  // selection rule of the context (epa_dev_set_heuristic); fixed: ceil(x * B) best, at least... none:
  // until_top_percent keeps ceil(x * B) elements, 0 for x == 0 (src/set_manipulators.cpp:82-88)
  const int mode = ctx->heur_mode;
  uint32_t limit = 0;
  if (mode == 1) {
    limit = (uint32_t)std::min<double>((double)B, std::ceil(ctx->heur_param * (double)B));
    ctx->select_cap = std::max(ctx->select_cap, std::max(limit, 1u));
  } else if (mode == 2) {
    ctx->select_cap = std::max(ctx->select_cap, std::min(B, 48u));
  }
  const uint32_t cap = ctx->select_cap;
  size_t scan_bytes = 0, sort_bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, Q,
                                rocprim::plus<uint32_t>(), ctx->stream);
  const size_t worst = std::min<uint64_t>(max_pairs, (uint64_t)Q * cap);
  (void)rocprim::radix_sort_keys<epa_radix_cfg>(nullptr, sort_bytes, (unsigned long long*)nullptr,
                                 (unsigned long long*)nullptr, worst, 0, 64, ctx->stream);
  // Bitmap form (SelOut): [status 512 B | branch counters B+1 | bitmap B x wpr | counts Q+1]
  const uint32_t wpr = (Q + 31) / 32;
  const bool force_sort = ctx->opt.select_sort != 0;
  const bool bm = !force_sort && (size_t)B * wpr * sizeof(uint32_t) <= ((size_t)64 << 20);
  sp->bitmap = nullptr;
  if (bm) {
    const size_t cb = align256(sizeof(uint32_t) * ((size_t)B + 1));
    const size_t mb = align256(sizeof(uint32_t) * (size_t)B * wpr);
    const size_t qb2 = align256(sizeof(uint32_t) * (Q + 1));
    char* base = (char*)epa_scratch(ctx, 7, 512 + cb + mb + qb2);
    if (!base) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(select scratch)");
    uint32_t* status = reinterpret_cast<uint32_t*>(base);
    uint32_t* bcount = reinterpret_cast<uint32_t*>(base + 512);
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(base + 512 + cb);
    uint32_t* counts = reinterpret_cast<uint32_t*>(base + 512 + cb + mb);
    sp->bitmap = bitmap; sp->boffs = bcount; sp->wpr = wpr;
    sp->counts = counts; sp->offsets = nullptr;
    sp->d_lnl = d_lnl; sp->Q = Q; sp->threshold = threshold; sp->d_pairs = d_pairs; sp->max_pairs = max_pairs;
    sp->d_span = d_span; sp->cap = cap; sp->rb = rb;
    { const int zr = epa_zero_async(ctx, base, 512 + cb + mb); if (zr) return zr; }   // status, counters, bitmap: one fill
    epa_timer_start(ctx, epa_t(ctx, epa_ctx::T_SELECT));
    const SelOut so{nullptr, 0u, bitmap, bcount, wpr};
    const dim3 grid((Q + 3) / 4);
    const int nr = (int)((B + 63) / 64);
#define SEL(N) hipLaunchKernelGGL(k_select<N>, grid, dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status)
    const bool seg_off = ctx->opt.select_full_rows != 0;   // A/B and test switch: the full-row kernels
    bool seg_hist = false;   // k_select_seg sums the span-class histogram itself
    if (ctx->segmax && !seg_off && mode == 0 && threshold < 1.0 && nr <= 64)
    {
      // band widths of the segment test (see k_select_seg): conservative margins, so the host's log is as good as the device's
      const double band_x = 1.0 - std::log((1.0 - threshold) / (double)B);
      const double band_t = std::max(band_x, std::log((double)B) + 38.0);
      constexpr uint32_t sel_grid = 16;   // workgroups per CU of the persistent selection grid
      const dim3 grid_seg(std::min<uint32_t>((Q + 3) / 4, (uint32_t)ctx->n_cu * sel_grid));
      hipLaunchKernelGGL(k_select_seg<8>, grid_seg, dim3(256), 0, ctx->stream, d_lnl, ctx->segmax, ctx->segp, Q, B, pitch, threshold,
                         band_x, band_t, so, counts, status, d_span, ctx->s, d_span ? status + 8 : nullptr);
      seg_hist = d_span != nullptr;
    }
    else if (nr <= 2) SEL(2); else if (nr <= 4) SEL(4); else if (nr <= 8) SEL(8); else if (nr <= 16) SEL(16);
    else if (nr <= 32) SEL(32); else if (nr <= 64) SEL(64);
    else if (nr <= 128) hipLaunchKernelGGL(k_select_wg<32>, dim3(Q), dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status);
    else if (nr <= 256) hipLaunchKernelGGL(k_select_wg<64>, dim3(Q), dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status);
    else hipLaunchKernelGGL(k_select_big<16>, grid, dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status);
#undef SEL
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, ctx->stream, bcount, B + 1);   // bcount[B] = total
    if (d_span && !seg_hist)
      hipLaunchKernelGGL(k_class_hist, dim3((Q + 255) / 256), dim3(256), 0, ctx->stream, counts, d_span, Q,
                         ctx->s, status + 8);
    hipLaunchKernelGGL(k_pack_readback, dim3(1), dim3(64), 0, ctx->stream, status, bcount + B,
                       sp->pre_status, status + 64);
    EPA_HIP(ctx, hipMemcpyAsync(rb, status + 64, sizeof(uint32_t) * 36, hipMemcpyDeviceToHost, ctx->stream));
    sp->have_status = sp->pre_status != nullptr;
    if (!ctx->ev_rb[ctx->bank]) EPA_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_rb[ctx->bank], hipEventDisableTiming));
    EPA_HIP(ctx, hipEventRecord(ctx->ev_rb[ctx->bank], ctx->stream));
    sp->ev_rb = ctx->ev_rb[ctx->bank];
    sp->d_rb = status + 64;
    sp->emitted = false;
    sp->queued_cls = -1;
    return EPA_OK;
  }
  sp->ev_rb = nullptr; sp->d_rb = nullptr; sp->emitted = false; sp->queued_cls = -1;
  // scratch 7: [status 512 B | counts Q+1 | offsets Q+1 | stage Q*cap | keys_a worst | keys_b worst | temp]
  // status words [0, 8 + EPA_N_CLS): selection status + class histogram; [64, 128): the packed read-back block
  const size_t qb = align256(sizeof(uint32_t) * (Q + 1));
  const size_t sb = align256(sizeof(unsigned long long) * (size_t)Q * cap);
  const size_t kb = align256(sizeof(unsigned long long) * worst);
  const size_t tb = std::max(scan_bytes, sort_bytes);
  char* base = (char*)epa_scratch(ctx, 7, 512 + 2 * qb + sb + 2 * kb + tb);
  if (!base) return epa_fail(ctx, EPA_ERR_HIP, "hipMalloc(select scratch)");
  uint32_t* status = reinterpret_cast<uint32_t*>(base);
  uint32_t* counts = reinterpret_cast<uint32_t*>(base + 512);
  uint32_t* offsets = reinterpret_cast<uint32_t*>(base + 512 + qb);
  sp->stage = reinterpret_cast<unsigned long long*>(base + 512 + 2 * qb);
  sp->keys_a = reinterpret_cast<unsigned long long*>(base + 512 + 2 * qb + sb);
  sp->keys_b = reinterpret_cast<unsigned long long*>(base + 512 + 2 * qb + sb + kb);
  sp->temp = base + 512 + 2 * qb + sb + 2 * kb;
  sp->sort_bytes = sort_bytes;
  sp->counts = counts; sp->offsets = offsets;
  sp->d_lnl = d_lnl; sp->Q = Q; sp->threshold = threshold; sp->d_pairs = d_pairs; sp->max_pairs = max_pairs;
  sp->d_span = d_span; sp->cap = cap; sp->rb = rb;
  // status words and the trailing count in one fill: counts[Q] sits in the same allocation
  EPA_HIP(ctx, hipMemsetAsync(status, 0, 512, ctx->stream));
  EPA_HIP(ctx, hipMemsetAsync(counts + Q, 0, sizeof(uint32_t), ctx->stream));
  epa_timer_start(ctx, epa_t(ctx, epa_ctx::T_SELECT));
  const dim3 grid((Q + 3) / 4);
  const int nr = (int)((B + 63) / 64);
  const SelOut so{sp->stage, cap, nullptr, nullptr, 0u};
#define SEL(N) hipLaunchKernelGGL(k_select<N>, grid, dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status)
#define SELBIG(N) hipLaunchKernelGGL(k_select_big<N>, grid, dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status)
  if (nr <= 2) SEL(2); else if (nr <= 4) SEL(4); else if (nr <= 8) SEL(8); else if (nr <= 16) SEL(16);
  else if (nr <= 32) SEL(32); else if (nr <= 64) SEL(64);
  else if (nr <= 128) hipLaunchKernelGGL(k_select_wg<32>, dim3(Q), dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status);
  else if (nr <= 256) hipLaunchKernelGGL(k_select_wg<64>, dim3(Q), dim3(256), 0, ctx->stream, d_lnl, Q, B, pitch, threshold, mode, limit, so, counts, status);
  else SELBIG(16);
#undef SELBIG
#undef SEL
  EPA_HIP(ctx, rocprim::exclusive_scan(sp->temp, scan_bytes, counts, offsets, 0u, Q + 1,
                                       rocprim::plus<uint32_t>(), ctx->stream));
  if (d_span)  // class histogram in status[8 ..]: read back with the total, no extra round trip
    hipLaunchKernelGGL(k_class_hist, dim3((Q + 255) / 256), dim3(256), 0, ctx->stream, counts, d_span, Q,
                       ctx->s, status + 8);
  // rb: [0] total, [1 .. 8 + EPA_N_CLS] status words + class histogram, [32 .. 35] window validation:
  // packed on the device so that ONE small copy brings everything the host waits for
  hipLaunchKernelGGL(k_pack_readback, dim3(1), dim3(64), 0, ctx->stream, status, offsets + Q,
                     sp->pre_status, status + 64);
  EPA_HIP(ctx, hipMemcpyAsync(rb, status + 64, sizeof(uint32_t) * 36, hipMemcpyDeviceToHost, ctx->stream));
  sp->have_status = sp->pre_status != nullptr;
  return EPA_OK;
}

// Bitmap form: the pair list written behind the read-back without waiting for the host (the kernel itself refuses a
// list that would not fit); the selection's timer stops here.
int launch_select_emit(epa_ctx* ctx, SelectPending* sp) {
  if (!sp->bitmap || !sp->d_rb || sp->emitted) return EPA_OK;
  hipLaunchKernelGGL(k_emit_pairs, dim3(ctx->B), dim3(256), 0, ctx->stream, sp->bitmap, sp->wpr, sp->boffs, sp->d_pairs, ctx->B, (uint32_t)std::min<uint64_t>(sp->max_pairs, 0xffffffffu));
  epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_SELECT));
  EPA_HIP(ctx, hipGetLastError());
  sp->emitted = true;
  return EPA_OK;
}

int launch_select_end(epa_ctx* ctx, SelectPending* sp, uint64_t* n_pairs) {
  const uint32_t B = ctx->B, Q = sp->Q;
  for (;;) {
    // everything up to the read-back copy; kernels queued behind it (launch_select_emit, launch_thorough_queued) run on
    if (sp->ev_rb) EPA_HIP(ctx, hipEventSynchronize(sp->ev_rb));
    else EPA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t total = sp->rb[0];
    const uint32_t* hst = sp->rb + 1;
    if (hst[2] && !sp->bitmap) {  // some query selected more candidates than the staging row holds: widen, redo
      if (sp->cap >= B) return epa_fail(ctx, EPA_ERR_HIP, "select_candidates: staging overflow");
      ctx->select_cap = std::min<uint32_t>(B, std::max(sp->cap * 4, hst[2]));
      uint32_t* rb = sp->rb;
      sp->rerun = true;    // keeps this chunk's preplacement status pointer
      int rc = launch_select_begin(ctx, sp->d_lnl, Q, sp->threshold, sp->d_pairs, sp->max_pairs, sp->d_span, rb, sp);
      sp->rerun = false;
      if (rc) return rc;
      continue;
    }
    if (total > sp->max_pairs)
      return epa_fail(ctx, EPA_ERR_PAIR_OVERFLOW,
                      "select_candidates: " + std::to_string(total) + " candidates exceed max_pairs");
    if (total && sp->bitmap) {
      if (!sp->emitted)
        hipLaunchKernelGGL(k_emit_pairs, dim3(B), dim3(256), 0, ctx->stream, sp->bitmap, sp->wpr, sp->boffs, sp->d_pairs, B, (uint32_t)std::min<uint64_t>(sp->max_pairs, 0xffffffffu));
    } else if (total) {
      const dim3 grid((Q + 3) / 4);
      hipLaunchKernelGGL(k_compact, grid, dim3(256), 0, ctx->stream, sp->stage, sp->counts, sp->offsets, Q, sp->cap,
                         sp->keys_a);
      // branch-major order == Work iteration order (std::map<branch, vector<seq>>).  The compacted
      // list is already ascending in the query id (and a query names a branch at most once), so a
      // STABLE sort on the branch bits alone gives (branch, query) order: 2 digit passes, not 6.
      int bits = 33;
      while ((1ull << (bits - 32)) <= B && bits < 64) ++bits;
      size_t sort_bytes = sp->sort_bytes;
      EPA_HIP(ctx, rocprim::radix_sort_keys<epa_radix_cfg>(sp->temp, sort_bytes, sp->keys_a, sp->keys_b, (size_t)total,
                                                           32, bits, ctx->stream));
      hipLaunchKernelGGL(k_keys_to_pairs, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, sp->keys_b,
                         (uint64_t)total, sp->d_pairs);
    }
    if (!sp->emitted) epa_timer_stop(ctx, epa_t(ctx, epa_ctx::T_SELECT));   // (else stopped behind the queued k_emit_pairs)
    EPA_HIP(ctx, hipGetLastError());
    *n_pairs = total;
    if (sp->d_span) {
      for (int c = 0; c < EPA_N_CLS; ++c) ctx->cls_hist[c] = hst[8 + c];
      ctx->cls_hist_pairs = total;
    }
    return EPA_OK;
  }
}

// what the preplacement's window validation found (read back by launch_select_begin into rb[32..])
int select_check_status(epa_ctx* ctx, const SelectPending* sp) {
  if (!sp->have_status) return EPA_OK;
  const uint32_t* st = sp->rb + 32;
  if (st[0])
    return epa_fail(ctx, EPA_ERR_QUERY_ALL_GAP, "Sequence " + std::to_string(st[0] & 0x7fffffffu) +
                                                    " does not appear to have any non-gap sites!");
  if (st[1])
    return epa_fail(ctx, EPA_ERR_QUERY_WIDTH, "Query sequence length not same as reference alignment!");
  return EPA_OK;
}

int launch_select(epa_ctx* ctx, const double* d_lnl, uint32_t Q, double threshold,
                  epa_pair* d_pairs, uint64_t max_pairs, uint64_t* n_pairs, const uint32_t* d_span) {
  uint32_t rb[64] = {};
  SelectPending sp;
  int rc = launch_select_begin(ctx, d_lnl, Q, threshold, d_pairs, max_pairs, d_span, rb, &sp);
  if (rc) return rc;
  return launch_select_end(ctx, &sp, n_pairs);
}
