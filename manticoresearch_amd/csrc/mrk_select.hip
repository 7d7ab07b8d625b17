// mrk_select.hip -- exact top-K of the queries' candidate lists (CSphMatchQueue's result: the K greatest matches under
// MatchRelevanceLt_fn, sphinxsort.cpp:583-812, 4541-4547), gfx950 / wave64.
//
// The scan kernels leave, per query, a pruning histogram over all its matches and a candidate list that holds every match
// whose bin could still reach the top K when it was found (mrk_kprune.h).  Selection = (1) the final threshold bin -- the
// largest bin with K matches at or above it --, (2) drop the candidates below it, (3) sort what is left.  Until round 3 one
// workgroup per query did all three, walking the whole list: 0.4 ms per launch whatever the shard size, and the tail of the
// single-query latency (a list of 200 K candidates = 200 dependent round trips).  Now (2) is a grid over 2048-key SLICES of
// all lists at once -- the chip's whole bandwidth on 30-50 MB -- and (3) sees about K + the threshold bin's population.
#include "mrk_kcommon.h"
#include "mrk_kprune.h"

namespace mrk {

// ---------------------------------------------------------------------------------------
// (1) one wave per query
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void sel_tau_kernel(SelectArgs a) {
  const uint32_t q = blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (q >= a.n_queries) return;
  const DevQuery* __restrict__ Q = a.queries + q;
  const uint32_t K = Q->k ? Q->k : 1u;
  const uint32_t tb = threshold_bin(a.q_hist + (uint64_t)q * NBINS, K);
  uint32_t n = a.q_cand_n[(size_t)q * QSTRIDE];
  if (n > Q->cand_cap) n = Q->cand_cap;
  if (lane_id() == 0) {
    a.sel_tau[q] = tb;
    a.sel_nslice[q] = (n + (uint32_t)SEL_SLICE - 1u) / (uint32_t)SEL_SLICE;
  }
}

// ---------------------------------------------------------------------------------------
// (2) a persistent grid over the slices of queries [q0, q0 + nq)
// ---------------------------------------------------------------------------------------
struct __align__(16) FilterSmem {
  uint32_t pre[SEL_MAXQ + 1]; // pre[i] = slices of the queries before i
  uint32_t wave_cnt[WAVES];
};

__global__ __launch_bounds__(WG) void sel_filter_kernel(SelectArgs a, uint32_t q0, uint32_t nq) {
  __shared__ FilterSmem s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  { // slice prefix over the group's queries: 16 consecutive queries per thread
    constexpr int PER = SEL_MAXQ / WG;
    uint32_t loc[PER], sum = 0;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const uint32_t qi = tid * PER + r;
      sum += qi < nq ? a.sel_nslice[q0 + qi] : 0u;
      loc[r] = sum;
    }
    const uint32_t incl = wave_incl_scan(sum);
    if (lane == 63) s.wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WAVES; ++w)
      if (w < wave) wbase += s.wave_cnt[w];
    const uint32_t excl = wbase + incl - sum;
#pragma unroll
    for (int r = 0; r < PER; ++r) s.pre[tid * PER + r + 1] = excl + loc[r];
    if (tid == 0) s.pre[0] = 0;
    __syncthreads();
  }
  const uint32_t total = s.pre[nq];
  for (uint32_t sl = blockIdx.x; sl < total; sl += gridDim.x) {
    // the query that owns slice sl: largest qi with pre[qi] <= sl
    uint32_t lo = 0, hi = nq;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s.pre[mid] <= sl)
        lo = mid;
      else
        hi = mid;
    }
    const uint32_t q = q0 + lo, j = sl - s.pre[lo];
    const DevQuery* __restrict__ Q = a.queries + q;
    const uint32_t bin_mode = Q->bin_mode, bin_shift = Q->bin_shift;
    const int32_t bin_lo = Q->bin_lo;
    const uint64_t cand_off = Q->cand_off;
    uint32_t n = a.q_cand_n[(size_t)q * QSTRIDE];
    if (n > Q->cand_cap) n = Q->cand_cap;
    const uint32_t tau_bin = a.sel_tau[q];
    const uint32_t first = j * (uint32_t)SEL_SLICE;
    const uint32_t cnt = n - first < (uint32_t)SEL_SLICE ? n - first : (uint32_t)SEL_SLICE;
    uint64_t* __restrict__ base = a.cand + cand_off + first;
    constexpr int R = SEL_SLICE / WG;
    uint64_t key[R];
    uint32_t keep = 0, c = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = (uint32_t)r * WG + tid;
      key[r] = i < cnt ? base[i] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t i = (uint32_t)r * WG + tid;
      const bool k = i < cnt && bin_of(bin_mode, bin_lo, bin_shift, key_weight(key[r]), key_rowid(key[r])) >= tau_bin;
      keep |= (k ? 1u : 0u) << r;
      c += k ? 1u : 0u;
    }
    const uint32_t incl = wave_incl_scan(c);
    __syncthreads(); // every key of the slice is in registers (and wave_cnt is free again): the slice may be rewritten
    if (lane == 63) s.wave_cnt[wave] = incl;
    __syncthreads();
    uint32_t pos = incl - c, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
      if (w < wave) pos += s.wave_cnt[w];
      tot += s.wave_cnt[w];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((keep >> r) & 1u) base[pos++] = key[r];
    if (tid == 0) a.slice_cnt[(cand_off >> 11) + q + j] = tot;
  }
}

// ---------------------------------------------------------------------------------------
// (3) one workgroup per query over the survivors of its slices
// ---------------------------------------------------------------------------------------
// The threshold bin is a weight interval, and BM25 weights of a two-keyword query take a few hundred distinct values: the
// K-th best match usually lies inside a weight class of thousands of docs that only the rowid tells apart (the sorter's
// tie order).  All of them survive the bin test.  So the survivors of the threshold bin go through one more histogram --
// SUB sub-bins over (weight offset inside the bin, rowid from the top of the segment's range), still monotone in the
// sorter's order -- and only the sub-bins that can reach the top K are gathered and sorted: about K + one sub-bin's share.
constexpr int SCH = 1024; // slices handled per chunk (a list of 2^20 candidates has 512)
constexpr int SUB = 2048; // sub-bins of the threshold bin
constexpr int SUB_BITS = 11;

struct __align__(16) SortSmem {
  uint64_t cand[CAND];
  uint32_t lpre[SCH]; // inclusive prefix of the slices' survivor counts
  uint32_t hist2[SUB];
  uint32_t wave_cnt[2 * WAVES];
  uint32_t cand_n, tau2, above;
  uint64_t tau;
};

// order of a key INSIDE its pruning bin, bigger = better, as a number of `totbits` bits
struct SubKey {
  uint32_t mode, shift, rbits, rhi, totbits;
  int32_t lo;
  __device__ __forceinline__ uint32_t sub(uint64_t key) const {
    const uint32_t grow = key_rowid(key);
    uint64_t ok;
    if (mode == BIN_WEIGHT) {
      const uint32_t woff = shift ? ((uint32_t)(key_weight(key) - lo) & ((1u << shift) - 1u)) : 0u;
      ok = ((uint64_t)woff << rbits) | (uint64_t)(rhi - grow);
    } else
      ok = (uint64_t)(((1u << shift) - 1u) - (grow & ((1u << shift) - 1u)));
    return totbits > (uint32_t)SUB_BITS ? (uint32_t)(ok >> (totbits - SUB_BITS)) : (uint32_t)ok;
  }
};

__global__ __launch_bounds__(WG) void sel_sort_kernel(SelectArgs a) {
  __shared__ SortSmem s;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (q >= a.n_queries) return;
  const DevQuery* __restrict__ Q = a.queries + q;
  const uint32_t K = Q->k ? Q->k : 1u;
  const uint64_t cand_off = Q->cand_off;
  const uint32_t ns = a.sel_nslice[q];
  const uint32_t* __restrict__ scnt = a.slice_cnt + (cand_off >> 11) + q;
  const uint64_t* __restrict__ src = a.cand + cand_off;
  const uint32_t bin_mode = Q->bin_mode, bin_shift = Q->bin_shift, tau_bin = a.sel_tau[q];
  const int32_t bin_lo = Q->bin_lo;
  if (tid == 0) {
    s.cand_n = 0;
    s.tau = 0;
    s.tau2 = 0;
    s.above = 0;
  }
  __syncthreads();
  for (uint32_t sbase = 0; sbase < ns; sbase += SCH) {
    const uint32_t nsc = ns - sbase < (uint32_t)SCH ? ns - sbase : (uint32_t)SCH;
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t l = tid * 4 + r;
      sum += l < nsc ? scnt[sbase + l] : 0u;
      c[r] = sum;
    }
    const uint32_t inc = wave_incl_scan(sum);
    __syncthreads();
    if (lane == 63) s.wave_cnt[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
      if (w < wave) wbase += s.wave_cnt[w];
      total += s.wave_cnt[w];
    }
    const uint32_t excl = wbase + inc - sum;
#pragma unroll
    for (int r = 0; r < 4; ++r) s.lpre[tid * 4 + r] = excl + c[r];
    __syncthreads();

    auto key_at = [&](uint32_t f) -> uint64_t { // survivor f of this chunk: slice = first l with lpre[l] > f
      uint32_t lo = 0, hi = nsc - 1;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s.lpre[mid] > f)
          hi = mid;
        else
          lo = mid + 1;
      }
      const uint32_t idx = f - (lo ? s.lpre[lo - 1] : 0u);
      return src[(uint64_t)(sbase + lo) * SEL_SLICE + idx];
    };

    // ---- the sub-bin pass: only when the whole list is this one chunk, does not fit one sort, and the threshold bin is
    // an interior one (the two edge bins are open-ended: no finite sub-order)
    SubKey sk{bin_mode, bin_shift, a.rowid_bits, a.rowid_hi, 0u, bin_lo};
    sk.totbits = bin_mode == BIN_WEIGHT ? bin_shift + a.rowid_bits : bin_shift;
    bool use_sub = ns <= (uint32_t)SCH && total > (uint32_t)CAND && tau_bin > 0 && sk.totbits > 0 && sk.totbits <= 56u &&
                   (bin_mode == BIN_ROWID || tau_bin < (uint32_t)NBINS - 1u);
    if (use_sub) {
      for (uint32_t i = tid; i < (uint32_t)SUB; i += WG) s.hist2[i] = 0;
      __syncthreads();
      uint32_t above = 0;
      for (uint32_t f = tid; f < total; f += WG) {
        const uint64_t key = key_at(f);
        const uint32_t b = bin_of(bin_mode, bin_lo, bin_shift, key_weight(key), key_rowid(key));
        if (b > tau_bin)
          ++above;
        else
          atomicAdd(&s.hist2[sk.sub(key)], 1u);
      }
      for (int dlt = 32; dlt; dlt >>= 1) above += __shfl_down(above, dlt, 64);
      if (lane == 0 && above) atomicAdd(&s.above, above);
      __syncthreads();
      // largest sub-bin t with above + sum(hist2[t..]) >= K: 8 consecutive sub-bins per thread, suffix sums via a prefix scan
      const uint32_t need = K > s.above ? K - s.above : 0u;
      uint32_t h[SUB / WG], mine = 0;
#pragma unroll
      for (int r = 0; r < SUB / WG; ++r) {
        h[r] = s.hist2[tid * (SUB / WG) + r];
        mine += h[r];
      }
      const uint32_t pin = wave_incl_scan(mine);
      if (lane == 63) s.wave_cnt[WAVES + wave] = pin;
      __syncthreads();
      uint32_t before = pin - mine, all = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
        if (w < wave) before += s.wave_cnt[WAVES + w];
        all += s.wave_cnt[WAVES + w];
      }
      const uint32_t after = all - before - mine; // survivors in the sub-bins above this thread's
      if (need && after < need && after + mine >= need) {
        uint32_t run = after;
        int r = SUB / WG - 1;
        for (; r > 0; --r) {
          run += h[r];
          if (run >= need) break;
        }
        s.tau2 = tid * (SUB / WG) + (uint32_t)r;
      }
      __syncthreads();
    }
    const uint32_t tau2 = s.tau2;

    for (uint32_t f0 = 0; f0 < total; f0 += 4 * WG) {
      if (s.cand_n > (uint32_t)(CAND - 4 * WG)) compact_cand(s, K, &s.tau);
      const uint64_t tau = s.tau;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t f = f0 + r * WG + tid;
        uint64_t key = 0;
        bool push = false;
        if (f < total) {
          key = key_at(f);
          push = key >= tau;
          if (use_sub && push)
            push = bin_of(bin_mode, bin_lo, bin_shift, key_weight(key), key_rowid(key)) > tau_bin || sk.sub(key) >= tau2;
        }
        const uint64_t bal = __ballot(push);
        if (bal) {
          uint32_t basep = 0;
          if (lane == 0) basep = atomicAdd(&s.cand_n, (uint32_t)__popcll(bal));
          basep = rdlane(basep, 0);
          if (push) s.cand[basep + __popcll(bal & ((1ull << lane) - 1ull))] = key;
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }
  const uint32_t m = compact_cand(s, K, &s.tau); // sorted best-first
  for (uint32_t i = tid; i < m; i += WG) a.out_keys[(uint64_t)q * KCAP + i] = s.cand[i];
  if (a.h_keys)
    for (uint32_t i = tid; i < m; i += WG) a.h_keys[(uint64_t)q * KCAP + i] = s.cand[i];
  if (a.rows_dst) { // the row of the shard exchange: a query whose candidate list overflowed leaves empty with MRK_ROW_RERUN (pack_rows_kernel's rule)
    const bool bad = (a.q_flags[q] & (QF_OVERFLOW | QF_FSM)) != 0;
    const uint32_t nr = bad ? 0u : m;
    uint64_t* __restrict__ row = a.rows_dst + (uint64_t)q * ROW_WORDS;
    for (uint32_t i = tid; i < (uint32_t)KCAP; i += WG) row[i] = i < nr ? s.cand[i] : 0ull;
    if (tid == 0) {
      row[KCAP] = nr;
      row[KCAP + 1] = bad ? ROW_RERUN : (a.q_total[q] & ~ROW_FLAG_MASK);
    }
  }
  if (tid == 0) {
    a.out_cnt[q] = m;
    if (a.h_cnt) a.h_cnt[q] = m;
    if (a.h_total) a.h_total[q] = a.q_total[q];
    if (a.h_flags) a.h_flags[q] = a.q_flags[q];
    if (a.h_cand_n) a.h_cand_n[q] = a.q_cand_n[(size_t)q * QSTRIDE];
  }
}

// ---------------------------------------------------------------------------------------
// descriptors up, scan state cleared: in front of every scan, on the scan stream
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void prep_kernel(PrepArgs a) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x, nt = gridDim.x * WG;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t n16 = a.n4[r] >> 2;
    const uint4* __restrict__ s4 = (const uint4*)a.src[r];
    uint4* __restrict__ d4 = (uint4*)a.dst[r];
    for (uint32_t i = t; i < n16; i += nt) d4[i] = s4[i];
    for (uint32_t i = (n16 << 2) + t; i < a.n4[r]; i += nt) a.dst[r][i] = a.src[r][i];
  }
  {
    const uint32_t n16 = a.zero_n4 >> 2;
    uint4* __restrict__ z4 = (uint4*)a.zero;
    for (uint32_t i = t; i < n16; i += nt) z4[i] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = (n16 << 2) + t; i < a.zero_n4; i += nt) a.zero[i] = 0;
  }
}

// ---------------------------------------------------------------------------------------
// merge of <= 8 sorted result rows per query (the shard exchange's merge step; CSphMatchQueue::MoveTo across chunks,
// sphinxsort.cpp:681-710): one workgroup per query, all lists in LDS
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void merge_rows_kernel(MergeRowsArgs a, uint32_t P) { // P = power of two >= n_lists
  extern __shared__ uint64_t mk[]; // [P][KCAP]
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  if (q >= a.n_queries) return;
  uint64_t total = 0, flags = 0, have = 0;
  for (uint32_t l = 0; l < P; ++l) {
    uint32_t cnt = 0;
    const uint64_t* __restrict__ row = nullptr;
    if (l < a.n_lists) {
      row = a.in_rows + ((uint64_t)l * a.list_stride + q) * ROW_WORDS;
      cnt = (uint32_t)row[KCAP];
      if (cnt > (uint32_t)KCAP) cnt = KCAP;
      const uint64_t t = row[KCAP + 1];
      total += t & ~ROW_FLAG_MASK, flags |= t & ROW_FLAG_MASK, have += cnt;
    }
    for (uint32_t i = tid; i < (uint32_t)KCAP; i += WG) mk[l * KCAP + i] = i < cnt ? row[i] : 0ull;
  }
  __syncthreads();
  for (uint32_t step = 1; step < P; step <<= 1) { // this round merges list slot 2 p step with slot (2 p + 1) step
    const uint32_t pairs = P / (2 * step);
    for (uint32_t t = tid; t < pairs * KCAP; t += WG) { // top K of A and B as a bitonic sequence, in A's place
      const uint32_t p = t / KCAP, i = t % KCAP;
      uint64_t* A = mk + (size_t)(2 * p * step) * KCAP;
      const uint64_t* B = mk + (size_t)((2 * p + 1) * step) * KCAP;
      const uint64_t x = A[i], y = B[KCAP - 1 - i];
      A[i] = x > y ? x : y;
    }
    __syncthreads();
    for (uint32_t j = KCAP / 2; j > 0; j >>= 1) { // ... sorted descending by half-cleaners
      for (uint32_t t = tid; t < pairs * (KCAP / 2); t += WG) {
        const uint32_t p = t / (KCAP / 2), i0 = t % (KCAP / 2);
        uint64_t* A = mk + (size_t)(2 * p * step) * KCAP;
        const uint32_t i = ((i0 & ~(j - 1)) << 1) | (i0 & (j - 1));
        const uint64_t x = A[i], y = A[i | j];
        if (x < y) A[i] = y, A[i | j] = x;
      }
      __syncthreads();
    }
  }
  const uint32_t n = have < a.k ? (uint32_t)have : a.k;
  uint64_t* __restrict__ out = a.out_rows + (uint64_t)(a.out_first + q) * ROW_WORDS;
  for (uint32_t i = tid; i < (uint32_t)KCAP; i += WG) out[i] = i < n ? mk[i] : 0ull;
  if (tid == 0) {
    out[KCAP] = n;
    out[KCAP + 1] = (total & ~ROW_FLAG_MASK) | flags; // CSphMatchQueue::MoveTo adds the totals up; the shards' flag bits are OR-ed through
    if (a.flags_any) {
      if (flags & ROW_RERUN) a.flags_any[0] = 1u;
      if (flags & ROW_DECLINED) a.flags_any[1] = 1u;
    }
  }
}

void launch_merge_rows(const MergeRowsArgs& a, void* stream) {
  if (!a.n_queries) return;
  uint32_t P = 1;
  while (P < a.n_lists) P <<= 1;
  const size_t lds = (size_t)P * KCAP * sizeof(uint64_t);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)merge_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * KCAP * (int)sizeof(uint64_t));
    attr_set = true;
  }
  hipLaunchKernelGGL(merge_rows_kernel, dim3(a.n_queries), dim3(WG), lds, (hipStream_t)stream, a, P);
}

void launch_prep(const PrepArgs& a, void* stream) {
  const uint32_t work = (a.n4[0] + a.n4[1] + a.zero_n4) / 4;
  uint32_t grid = (work + WG * 4 - 1) / (WG * 4);
  grid = grid < 1u ? 1u : grid > 256u ? 256u : grid;
  hipLaunchKernelGGL(prep_kernel, dim3(grid), dim3(WG), 0, (hipStream_t)stream, a);
}

void launch_select(const SelectArgs& a, void* stream) {
  if (!a.n_queries) return;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sel_tau_kernel, dim3((a.n_queries + WAVES - 1) / WAVES), dim3(WG), 0, st, a);
  uint32_t grid = a.max_slices < 2048u ? a.max_slices : 2048u;
  if (grid < 1u) grid = 1u;
  for (uint32_t q0 = 0; q0 < a.n_queries; q0 += SEL_MAXQ) {
    const uint32_t nq = a.n_queries - q0 < (uint32_t)SEL_MAXQ ? a.n_queries - q0 : (uint32_t)SEL_MAXQ;
    hipLaunchKernelGGL(sel_filter_kernel, dim3(grid), dim3(WG), 0, st, a, q0, nq);
  }
  hipLaunchKernelGGL(sel_sort_kernel, dim3(a.n_queries), dim3(WG), 0, st, a);
}

} // namespace mrk
