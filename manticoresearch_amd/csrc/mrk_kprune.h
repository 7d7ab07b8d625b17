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
