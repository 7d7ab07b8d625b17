// mrk_kcommon.h -- device helpers shared by the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "mrk_dev.h"

namespace mrk {

constexpr uint32_t NOBLK = 0xFFFFFFFFu;
constexpr uint32_t INF_ROWID = 0xFFFFFFFFu;
constexpr int CHUNK = 63; // usable block-index entries per 64-lane metadata chunk (one spare for "next")

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// ---- wave-wide primitives on DPP (row_shr within rows of 16, then row_bcast15 / row_bcast31)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += dpp_u32<0x111, 0xf>(0, v); // row_shr:1
  v += dpp_u32<0x112, 0xf>(0, v); // row_shr:2
  v += dpp_u32<0x114, 0xf>(0, v); // row_shr:4
  v += dpp_u32<0x118, 0xf>(0, v); // row_shr:8
  v += dpp_u32<0x142, 0xa>(0, v); // row_bcast:15 -> rows 1,3
  v += dpp_u32<0x143, 0xc>(0, v); // row_bcast:31 -> rows 2,3
  return v;
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
  v = min(v, dpp_u32<0x111, 0xf>(0xFFFFFFFFu, v));
  v = min(v, dpp_u32<0x112, 0xf>(0xFFFFFFFFu, v));
  v = min(v, dpp_u32<0x114, 0xf>(0xFFFFFFFFu, v));
  v = min(v, dpp_u32<0x118, 0xf>(0xFFFFFFFFu, v));
  v = min(v, dpp_u32<0x142, 0xa>(0xFFFFFFFFu, v));
  v = min(v, dpp_u32<0x143, 0xc>(0xFFFFFFFFu, v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, uint32_t l) {
  return ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l);
}

// LDS hand-off between lanes of ONE wave: make earlier ds_writes visible and keep the
// compiler from moving LDS accesses across this point.
// (MRK_WAVE_FENCE=1: wavefront-scope fences -- ordering for the compiler only.  One wave's DS instructions execute in issue
// order, so another lane's earlier ds_write / ds_or is seen by a later ds_read without any s_waitcnt; the workgroup-scope
// form also drains vmcnt, i.e. stalls on every global load and store still in flight.)
#ifndef MRK_WAVE_FENCE
#define MRK_WAVE_FENCE 0
#endif
__device__ __forceinline__ void wave_lds_fence() {
#if MRK_WAVE_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

// bit k of the result = byte k of w is a varint terminator (bit 7 clear)
__device__ __forceinline__ uint32_t term4(uint32_t w) { return (((~w & 0x80808080u) >> 7) * 0x10204080u) >> 28; }

__device__ __forceinline__ float term_tfidf(uint32_t tf, float idf) {
  // float(hits) / float(hits + 1.2f) * idf  -- searchnode.cpp:2828; no contraction (-ffp-contract=off)
  const float fh = (float)tf;
  const float den = fh + 1.2f;
  const float q = fh / den;
  return q * idf;
}

// bitonic sort of c[0..len) descending, all WG threads; len = a power of two <= CAND (entries past the keys are zeros: a
// short list sorts a short network -- 15 stages for 32 keys instead of the 66 of the full buffer)
__device__ void sort_cand_desc(uint64_t* c, uint32_t len = (uint32_t)CAND) {
  for (uint32_t k = 2; k <= len; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < len / 2; t += WG) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // lower index of the pair
        const uint32_t p = i | j;
        const uint64_t x = c[i], y = c[p];
        const bool desc = (i & k) == 0;
        if ((x < y) == desc) {
          c[i] = y;
          c[p] = x;
        }
      }
      __syncthreads();
    }
  }
}

// exclusive positions of flags laid out as i = tid + r*WG; returns total
__device__ __forceinline__ uint32_t block_scan2(uint32_t* wave_cnt, bool f0, bool f1, uint32_t& pos0, uint32_t& pos1) {
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
  __syncthreads(); // wave_cnt reuse
  if (lane == 0) {
    wave_cnt[wave] = __popcll(b0);
    wave_cnt[WAVES + wave] = __popcll(b1);
  }
  __syncthreads();
  uint32_t base0 = 0, base1 = 0, tot0 = 0, tot1 = 0;
#pragma unroll
  for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
    const uint32_t c0 = wave_cnt[w], c1 = wave_cnt[WAVES + w];
    if (w < wave) base0 += c0, base1 += c1;
    tot0 += c0;
    tot1 += c1;
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  pos0 = base0 + __popcll(b0 & lt);
  pos1 = tot0 + base1 + __popcll(b1 & lt);
  return tot0 + tot1;
}


// wave-cooperative search: largest i in [lo, n) with base[i] <= r, given base[lo] <= r.
// 64-ary: every step probes 64 evenly spaced entries.
__device__ uint32_t wave_find_block(const uint32_t* __restrict__ base, uint32_t lo, uint32_t n, uint32_t r) {
  const uint32_t lane = lane_id();
  uint32_t hi = n;
  while (hi - lo > 1) {
    const uint32_t len = hi - lo;
    const uint32_t stride = (len + 63) / 64;
    const uint32_t idx = lo + lane * stride;
    const bool le = idx < hi && base[idx] <= r;
    const uint64_t bal = __ballot(le);
    const uint32_t p = 63u - (uint32_t)__builtin_clzll(bal | 1ull); // lane 0 always qualifies
    const uint32_t nlo = lo + p * stride;
    const uint32_t nhi = nlo + stride < hi ? nlo + stride : hi;
    lo = nlo;
    hi = nhi;
  }
  return lo;
}

// keep the best min(n, k) keys; once k keys are held their worst one is a valid lower bound of
// the query's final K-th best key: raise the shared threshold with it
template <typename S>
__device__ uint32_t compact_cand(S& s, uint32_t k, uint64_t* gtau) {
  __syncthreads();
  const uint32_t n = s.cand_n;
  uint32_t len = 64; // the smallest power of two that holds the keys
  while (len < n) len <<= 1;
  if (len > (uint32_t)CAND) len = (uint32_t)CAND;
  for (uint32_t i = n + threadIdx.x; i < len; i += WG) s.cand[i] = 0;
  __syncthreads();
  sort_cand_desc(s.cand, len);
  const uint32_t keep = n < k ? n : k;
  if (threadIdx.x == 0) {
    s.cand_n = keep;
    if (keep == k && s.cand[k - 1] > s.tau) {
      s.tau = s.cand[k - 1];
      atomicMax((unsigned long long*)gtau, (unsigned long long)s.tau);
    }
  }
  __syncthreads();
  return keep;
}


} // namespace mrk
