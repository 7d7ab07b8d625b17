// mrk_kprune.h -- device helpers shared by the packed-doclist kernels: the pruning histogram
// (bin of a match, threshold bin, histogram flush) and the tf-exception lookup.
#pragma once
#include "mrk_kcommon.h"

namespace mrk {

// Pruning bin of a match: monotone non-decreasing in the sorter's order (weight, then lower
// rowid), so "K matches already sit in higher bins" proves a match cannot reach the top K.
__device__ __forceinline__ uint32_t bin_of(uint32_t mode, int32_t lo, uint32_t shift, int32_t weight, uint32_t grow) {
  if (mode == BIN_WEIGHT) {
    if (weight < lo) return 0u;
    const uint32_t b = (uint32_t)(weight - lo) >> shift;
    return b < (uint32_t)NBINS ? b : (uint32_t)NBINS - 1u;
  }
  const uint32_t b = grow >> shift;
  return (uint32_t)NBINS - 1u - (b < (uint32_t)NBINS ? b : (uint32_t)NBINS - 1u);
}

// Largest bin b with sum(hist[b..]) >= k (0 if the whole histogram holds fewer than k).
// The histogram is read quarter by quarter from the top, lane l taking 4 consecutive bins of
// each quarter (coalesced 1 KiB per quarter, L1 bypassed so other CUs' adds are seen).
static __device__ uint32_t threshold_bin(const uint32_t* __restrict__ gh, uint32_t k) {
  const uint32_t lane = lane_id();
  uint32_t acc = 0;
  for (int qd = NBINS / 256 - 1; qd >= 0; --qd) {
    const uint32_t* p = gh + 256 * qd + 4 * lane;
    const uint32_t g0 = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t sum = g0 + g1 + g2 + g3;
    const uint32_t pre = wave_incl_scan(sum);
    const uint32_t tot = rdlane(pre, 63);
    if (acc + tot >= k) { // the threshold lies in this quarter
      const uint32_t above = acc + tot - pre; // everything above my 4 bins
      const uint64_t okl = __ballot(above + sum >= k);
      const uint32_t L = 63u - (uint32_t)__builtin_clzll(okl | 1ull);
      uint32_t run = above + g3, bi = 3;
      if (run < k) run += g2, bi = 2;
      if (run < k && bi == 2) run += g1, bi = 1;
      if (run < k && bi == 1) bi = 0;
      return 256u * (uint32_t)qd + 4u * L + rdlane(bi, L);
    }
    acc += tot;
  }
  return 0u;
}

// add this wave's per-bin counts to the query's global histogram and clear them
__device__ __forceinline__ void flush_hist(uint32_t* lh, uint32_t* gh) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t b = 16 * lane + (uint32_t)i;
    const uint32_t c = lh[b];
    if (c) atomicAdd(gh + b, c);
  }
}

// add the bins of the wave's buffered candidate keys to the query's global histogram: per 64 keys one atomic per DISTINCT
// bin (candidates crowd around the threshold: a handful of bins), straight to memory -- no LDS histogram to clear, fill and
// flush, no lock around one
__device__ __forceinline__ void hist_add_keys(uint32_t* __restrict__ gh, const uint64_t* cbuf, uint32_t cn, uint32_t mode, int32_t lo, uint32_t shift) {
  const uint32_t lane = lane_id();
  for (uint32_t i0 = 0; i0 < cn; i0 += 64) {
    const bool v = i0 + lane < cn;
    uint32_t bin = 0xFFFFFFFFu;
    if (v) {
      const uint64_t key = cbuf[i0 + lane];
      bin = bin_of(mode, lo, shift, key_weight(key), key_rowid(key));
    }
    uint64_t left = __ballot(v);
    while (left) {
      const uint32_t l = (uint32_t)__builtin_ctzll(left);
      const uint32_t b = rdlane(bin, l);
      const uint64_t same = __ballot(bin == b);
      if (lane == l) atomicAdd(gh + b, (uint32_t)__popcll(same));
      left &= ~same;
    }
  }
}

// the same for one bin per lane (v = the lane holds one)
__device__ __forceinline__ void hist_add_bins(uint32_t* __restrict__ gh, bool v, uint32_t bin) {
  const uint32_t lane = lane_id();
  const uint32_t mine = v ? bin : 0xFFFFFFFFu;
  uint64_t left = __ballot(v);
  while (left) {
    const uint32_t l = (uint32_t)__builtin_ctzll(left);
    const uint32_t b = rdlane(mine, l);
    const uint64_t same = __ballot(mine == b);
    if (lane == l) atomicAdd(gh + b, (uint32_t)__popcll(same));
    left &= ~same;
  }
}

// ---- second level of the lower-bound histogram (round 3): the matches INSIDE the threshold bin, by exact weight and rowid ----
// BM25 without document length is tie-heavy: thousands of matches share the very weight the K-th best has, and the sorter
// breaks the tie by rowid (lower wins).  The first level only knows "K matches reach bin T"; the second level counts the
// lower bounds that fall INTO bin T by (weight offset inside the bin, 2^rbits rowid slices, lower rowids in higher slots): once
// K matches are known to sit above slot S, a match whose UPPER bound lies in bin T below slot S cannot enter the top K either.
// The threshold bin moves (upwards only) while the kernels run, so every counter carries the bin it counts for in its upper
// ten bits: adds for an older bin are dropped, the first add for a newer bin restarts the counter -- a counter never holds a
// match that is not in the bin it names, and an undercount only lowers the threshold.
constexpr uint32_t H2_CNT_BITS = 22, H2_CNT_MASK = (1u << H2_CNT_BITS) - 1u;

__device__ __forceinline__ void hist2_add(uint32_t* __restrict__ gh2, bool v, uint32_t slot, uint32_t ver) {
  const uint32_t lane = lane_id();
  const uint32_t mine = v ? slot : 0xFFFFFFFFu;
  uint64_t left = __ballot(v);
  while (left) {
    const uint32_t l = (uint32_t)__builtin_ctzll(left);
    const uint32_t b = rdlane(mine, l);
    const uint64_t same = __ballot(mine == b);
    if (lane == l) {
      const uint32_t cnt = (uint32_t)__popcll(same);
      uint32_t old = __hip_atomic_load(gh2 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        const uint32_t over = old >> H2_CNT_BITS;
        if (over > ver) break; // the threshold bin has moved on: this match is below it now
        uint32_t c = (over == ver ? (old & H2_CNT_MASK) : 0u) + cnt;
        if (c > H2_CNT_MASK) c = H2_CNT_MASK;
        if (__hip_atomic_compare_exchange_strong(gh2 + b, &old, (ver << H2_CNT_BITS) | c, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    left &= ~same;
  }
}

// threshold_bin over the first level, plus the number of matches in the bins strictly above the one returned
static __device__ uint32_t threshold_bin_above(const uint32_t* __restrict__ gh, uint32_t k, uint32_t& n_above) {
  const uint32_t lane = lane_id();
  uint32_t acc = 0;
  n_above = 0;
  for (int qd = NBINS / 256 - 1; qd >= 0; --qd) {
    const uint32_t* p = gh + 256 * qd + 4 * lane;
    const uint32_t g0 = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t g3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t sum = g0 + g1 + g2 + g3;
    const uint32_t pre = wave_incl_scan(sum);
    const uint32_t tot = rdlane(pre, 63);
    if (acc + tot >= k) {
      const uint32_t above = acc + tot - pre;
      const uint64_t okl = __ballot(above + sum >= k);
      const uint32_t L = 63u - (uint32_t)__builtin_clzll(okl | 1ull);
      uint32_t run = above + g3, bi = 3, ab = above;
      if (run < k) ab = run, run += g2, bi = 2;
      if (run < k && bi == 2) ab = run, run += g1, bi = 1;
      if (run < k && bi == 1) ab = run, bi = 0;
      n_above = rdlane(ab, L);
      return 256u * (uint32_t)qd + 4u * L + rdlane(bi, L);
    }
    acc += tot;
  }
  n_above = acc;
  return 0u;
}

// threshold_bin over the second level: only the counters that name bin `ver` count
static __device__ uint32_t threshold_slot(const uint32_t* __restrict__ gh2, uint32_t k, uint32_t ver) {
  const uint32_t lane = lane_id();
  uint32_t acc = 0;
  if (!k) return (uint32_t)NBINS - 1u;
  for (int qd = NBINS / 256 - 1; qd >= 0; --qd) {
    const uint32_t* p = gh2 + 256 * qd + 4 * lane;
    uint32_t g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t x = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      g[i] = (x >> H2_CNT_BITS) == ver ? (x & H2_CNT_MASK) : 0u;
    }
    const uint32_t sum = g[0] + g[1] + g[2] + g[3];
    const uint32_t pre = wave_incl_scan(sum);
    const uint32_t tot = rdlane(pre, 63);
    if (acc + tot >= k) {
      const uint32_t above = acc + tot - pre;
      const uint64_t okl = __ballot(above + sum >= k);
      const uint32_t L = 63u - (uint32_t)__builtin_clzll(okl | 1ull);
      uint32_t run = above + g[3], bi = 3;
      if (run < k) run += g[2], bi = 2;
      if (run < k && bi == 2) run += g[1], bi = 1;
      if (run < k && bi == 1) bi = 0;
      return 256u * (uint32_t)qd + 4u * L + rdlane(bi, L);
    }
    acc += tot;
  }
  return 0u;
}

// position of the n-th (0-based) set bit of a wave-uniform 64-bit mask, n per lane (n < popcount)
__device__ __forceinline__ uint32_t nth_set_bit64(uint64_t m, uint32_t n) {
  uint32_t pos = 0, w = (uint32_t)m;
  uint32_t c = (uint32_t)__popc(w);
  if (n >= c) n -= c, pos = 32u, w = (uint32_t)(m >> 32);
  c = (uint32_t)__popc(w & 0xFFFFu);
  if (n >= c) n -= c, pos += 16u, w >>= 16;
  c = (uint32_t)__popc(w & 0xFFu);
  if (n >= c) n -= c, pos += 8u, w >>= 8;
  c = (uint32_t)__popc(w & 0xFu);
  if (n >= c) n -= c, pos += 4u, w >>= 4;
  c = (uint32_t)__popc(w & 0x3u);
  if (n >= c) n -= c, pos += 2u, w >>= 2;
  if (n >= (w & 1u)) pos += 1u;
  return pos;
}

// Bounds of a match's weight under the proximity rankers BEFORE its hits are read (round 3).  RankerState_Proximity_fn
// (sphinxsearch.cpp:1320-1438) adds 1000 x sum_f LCS[f] x w[f] to the BM25 part; LCS[f] is the longest run of hits of
// field f whose (position - query position) stays the same.  A single hit already makes a run of one, and every hit adds at
// most one to a run, so 1 <= LCS[f] <= h_f wherever the field has a hit at all, h_f = the hits the emitting keywords can have
// in field f.  The doclists say, per keyword, its hit count tf and the fields F it occurs in -- each of those holds at least one
// of its hits, so at most tf - |F| + 1 of them lie in any one field: h_f = sum over the emitting keywords with f in F of
// (tf - |F| + 1).
// A tighter end holds where hits at one (field, position) reach the ranker in query-position order: the run's position groups then
// have strictly increasing query positions (p' - qmin' = p - qmax with p' > p), so a run holds every keyword at most once and
// LCS[f] <= the number of emitting keywords with f in F.  The merges of AND / OR / MAYBE / ANDNOT order by (Hitpos_t, query position)
// (IsHitLess, searchnode.cpp:2611) and Hitpos_t carries the field-END flag: the order is by query position iff that flag belongs to the
// POSITION -- the reference's indexer marks every hit of a field's tail position (sphinx.cpp:22424-22430) -- or no hit carries one.
// A writer that flags each keyword's own last hit (the synthetic corpora with end_markers do) breaks it: 't2@2, t0@2|END, t2@3' runs
// to three with two keywords (found by a soak of the fuzz test with a fresh seed, MRK_FUZZ_SEED=777, after the second level below
// began to cut INSIDE the threshold bin; until then the tight end was the only one).  So the tight end is the caller's statement
// about the index (ctx key prox_bound_keywords -> TF_LCS_BY_KEYWORDS); the default is the bound by hits.
// The scan keeps a histogram of the LOWER bounds: K docs whose lower bound reaches bin T prove that the K-th best weight
// reaches it, and a doc whose UPPER bound stays below T cannot enter the top K -- it is counted (total_found) and never
// travels to the hit pass.
__device__ __forceinline__ void prox_bounds(uint32_t ranker, float acc, uint32_t emit, const uint32_t* kf, const uint32_t* ktf, int nk, const int32_t* fw, uint32_t nw,
                                            uint32_t index_weight, bool by_keywords, uint32_t& wlo, uint32_t& whi) {
  int lo = 0, hi = 0;
  int room[MAX_PROX_TERMS]; // hits of keyword k that one field's run can take at most
#pragma unroll
  for (int k = 0; k < MAX_PROX_TERMS; ++k) room[k] = k < nk && ((emit >> k) & 1u) ? (by_keywords ? 1 : (int)ktf[k] - (int)__popc(kf[k]) + 1) : 0;
  for (uint32_t f = 0; f < nw; ++f) {
    int h = 0;
#pragma unroll
    for (int k = 0; k < MAX_PROX_TERMS; ++k) h += ((kf[k] >> f) & 1u) ? room[k] : 0;
    const int w = fw[f], one = h > 0 ? 1 : 0;
    lo += w >= 0 ? w * one : w * h;
    hi += w >= 0 ? w * h : w * one;
  }
  const int32_t bm = (int32_t)((acc + 0.5f) * 1000.0f);
  if (ranker == MRK_RANK_PROXIMITY_BM25)
    wlo = (uint32_t)bm + (uint32_t)lo * 1000u, whi = (uint32_t)bm + (uint32_t)hi * 1000u;
  else
    wlo = (uint32_t)lo, whi = (uint32_t)hi;
  wlo *= index_weight, whi *= index_weight;
}

// exact hit count of a doc whose packed tf saturated (>= 255)
static __device__ uint32_t exc_tf(const DevSegment& seg, const DevTerm& T, uint32_t rowid) {
  const uint64_t* __restrict__ e = seg.pk_exc + T.exc_first;
  uint32_t lo = 0, hi = T.exc_n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint32_t)(e[mid] >> 32) < rowid)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo < T.exc_n && (uint32_t)(e[lo] >> 32) == rowid) return (uint32_t)e[lo];
  return 255u;
}

} // namespace mrk
