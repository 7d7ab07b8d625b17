// mrk_kernels.hip -- gfx950 (MI355X, wave64) kernels of the match -> rank -> top-K path.
//
// scan_kernel  : one workgroup per work item = (query, range of driver-term blocks).
//                VLB doclist blocks are pulled from HBM 16 B/lane, staged in LDS, split
//                into entries with wave-wide terminator-bit prefix sums, decoded one
//                entry per lane, intersected N-way (driver docs probe the other terms'
//                decoded blocks), scored (fp32 BM25 in reference op order, int weight)
//                and filtered into a per-workgroup top-K buffer in LDS.
// merge_kernel : one workgroup per query: top-K of the candidates the scan produced
//                (or of per-shard partial top-K lists), sorted best-first.
//
// What each piece restates (reference file:line, Manticore 3.6.1 src/):
//   decode_block      DiskIndexQword_c::ReadNext         sphinx.cpp:511-549
//   block lookup      DiskIndexQword_c::HintRowID        sphinx.cpp:407-451
//   intersection      ExtMultiAnd_T::AdvanceQwords       searchnode.cpp:2865-2889
//   field filter      NodeInfo_t::FitsFields             searchnode.cpp:2727-2747
//   tfidf             ExtMultiAnd_T::GetTFIDF            searchnode.cpp:2821-2832
//   weights           GetFilteredDocs / WeightSum / None sphinxsearch.cpp:1070, 1097-1169
//   top-K order       MatchRelevanceLt_fn                sphinxsort.cpp:4541-4547
// Integer / byte work on HBM-bound streams: no MFMA by design.
#include <hip/hip_runtime.h>

#include "mrk_dev.h"

namespace mrk {

constexpr int RNG_CAP = 512; // other-term block bases cached in LDS per tile
constexpr uint32_t NOBLK = 0xFFFFFFFFu;

struct __align__(16) Smem {
  uint8_t stage[WAVES][STAGE_BYTES + 16];
  uint64_t cand[CAND];
  uint32_t t0_rowid[TILE];
  float t0_acc[TILE];
  uint32_t t0_fields[TILE];
  uint32_t blk_of[TILE];
  uint32_t need_list[TILE];
  uint32_t tj_rowid[SLOTS][DEVBLK];
  uint32_t tj_tf[SLOTS][DEVBLK];
  uint32_t tj_fields[SLOTS][DEVBLK];
  uint32_t rng_base[RNG_CAP];
  uint16_t need_idx[TILE];
  uint16_t docstart[WAVES][DEVBLK + 8];
  int32_t weights[32];
  uint32_t wave_cnt[2 * WAVES];
  uint32_t cand_n;
  uint32_t rng_lo, rng_hi;
  uint64_t tau;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += t;
  }
  return v;
}

// LDS hand-off between lanes of ONE wave: make earlier ds_writes visible and keep the
// compiler from moving LDS accesses across this point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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

struct Dec {
  uint32_t rowid[2], tf[2], fields[2];
  bool ok[2];
};

// One wave decodes one device block (<= 128 doclist entries): lane l returns entries l and l+64.
__device__ void decode_block(const DevSegment& seg, const DevTerm& T, uint32_t b, uint32_t nd, uint8_t* stage,
                             uint16_t* docstart, Dec& out) {
  const uint32_t lane = lane_id();
  const uint32_t gb = T.blk_first + b;
  const uint64_t p0 = seg.blk_off[gb];
  const uint64_t p1 = (b + 1 < T.nblocks) ? seg.blk_off[gb + 1] : T.spd_end;
  const uint32_t base = seg.blk_base[gb] - 1u; // decoder rowid before the block's first entry (sphinx.cpp:447)
  const uint64_t a = p0 & ~15ull;
  const uint32_t lead = (uint32_t)(p0 - a);
  uint64_t span = p1 > a ? p1 - a : 0;
  const uint32_t nbytes = span < (uint64_t)STAGE_BYTES ? (uint32_t)span : (uint32_t)STAGE_BYTES;

  docstart[lane] = 0xFFFF;
  docstart[lane + 64] = 0xFFFF;
  if (lane == 0) docstart[0] = (uint16_t)lead;

  // ---- phase A: stage bytes, number the varints, record where every 4th one ends
  uint32_t carry = 0;
  for (uint32_t off = 0; off < nbytes; off += 1024) {
    const uint32_t my = off + lane * 16;
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    if (my < nbytes) {
      v = *reinterpret_cast<const uint4*>(seg.spd + a + my);
      *reinterpret_cast<uint4*>(stage + my) = v;
    }
    uint32_t t = term4(v.x) | (term4(v.y) << 4) | (term4(v.z) << 8) | (term4(v.w) << 12);
    if (my + 16 > nbytes) t = (my < nbytes) ? (t & ((1u << (nbytes - my)) - 1u)) : 0u;
    if (my < lead) t &= ~((1u << (lead - my)) - 1u);
    const uint32_t cnt = __popc(t);
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t s = carry + inc - cnt; // index of my first varint
    // every doclist entry is exactly 4 varints (sphinx.cpp:8456-8490): entry e ends with varint 4e+3
    uint32_t m = t;
    const uint32_t r = (3u - s) & 3u;
    for (uint32_t i = 0; i < r; ++i) m &= m - 1;
    uint32_t idx = s + r;
    while (m) {
      const uint32_t pos = __builtin_ctz(m);
      const uint32_t doc = (idx >> 2) + 1;
      if (doc < nd) docstart[doc] = (uint16_t)(my + pos + 1);
      m &= m - 1;
      m &= m - 1;
      m &= m - 1;
      m &= m - 1;
      idx += 4;
    }
    carry += __shfl(inc, 63, 64);
  }
  wave_lds_fence();

  // ---- phase B: one entry per lane
  uint32_t delta[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t d = lane + 64 * r;
    bool ok = d < nd;
    uint32_t st = ok ? docstart[d] : 0xFFFFu;
    ok = ok && st != 0xFFFFu && st < nbytes;
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (ok) {
      const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
      const uint32_t wi = st >> 2, sh = st & 3u;
      const uint32_t w0 = s32[wi], w1 = s32[wi + 1], w2 = s32[wi + 2];
      const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
      const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
      uint64_t x = ((uint64_t)hi << 32) | lo;
      const uint64_t tm = ~x & 0x8080808080808080ull;
      if (__popcll(tm) >= 4) {
        uint32_t vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t val = 0, bb;
          do {
            bb = (uint32_t)x & 0xffu;
            x >>= 8;
            val = (val << 7) | (bb & 0x7fu);
          } while (bb & 0x80u);
          vv[k] = val;
        }
        v0 = vv[0], v1 = vv[1], v2 = vv[2], v3 = vv[3];
      } else {
        // long entry (rare): byte loop out of LDS
        uint32_t pos = st;
        const uint32_t lim = STAGE_BYTES + 15;
        uint32_t vv[4];
        for (int k = 0; k < 4; ++k) {
          uint32_t val = 0, bb;
          do {
            bb = stage[pos < lim ? pos : lim];
            ++pos;
            val = (val << 7) | (bb & 0x7fu);
          } while ((bb & 0x80u) && pos <= lim);
          vv[k] = val;
        }
        v0 = vv[0], v1 = vv[1], v2 = vv[2], v3 = vv[3];
      }
    }
    // ReadNext (sphinx.cpp:511-549)
    uint32_t tf, fields;
    if (seg.inline_hits) {
      tf = v1;
      if (tf == 1) { // lone hit inlined: v2 = position, v3 = field<<1 | end
        const uint32_t f = (v3 >> 1) & 255u;
        fields = f < 32 ? (1u << f) : 0u;
      } else
        fields = v2; // v3 = hitlist offset delta
    } else { // plain: delta, hitlist offset delta, fieldmask, hits
      fields = v2;
      tf = v3;
    }
    delta[r] = ok ? v0 : 0u;
    out.tf[r] = tf;
    out.fields[r] = fields;
    out.ok[r] = ok;
  }
  const uint32_t s0 = wave_incl_scan(delta[0]);
  const uint32_t tot0 = __shfl(s0, 63, 64);
  const uint32_t s1 = wave_incl_scan(delta[1]);
  out.rowid[0] = base + s0;
  out.rowid[1] = base + tot0 + s1;
  wave_lds_fence(); // stage/docstart are reused by this wave's next block
}

// largest i in [0,n) with base[i] <= r, given base[0] <= r
template <typename P>
__device__ __forceinline__ uint32_t find_block(P base, uint32_t n, uint32_t r) {
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (base[mid] <= r)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// bitonic sort of s.cand[0..CAND) descending, all WG threads
__device__ void sort_cand_desc(uint64_t* c) {
  for (uint32_t k = 2; k <= (uint32_t)CAND; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < (uint32_t)CAND / 2; t += WG) {
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

// keep the best min(n, k) keys; returns new count; once k keys are held their worst one is a
// valid lower bound of the query's final K-th best key: raise the shared threshold with it
__device__ uint32_t compact_cand(Smem& s, uint32_t k, uint64_t* gtau) {
  __syncthreads();
  const uint32_t n = s.cand_n;
  for (uint32_t i = n + threadIdx.x; i < (uint32_t)CAND; i += WG) s.cand[i] = 0;
  __syncthreads();
  sort_cand_desc(s.cand);
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

// exclusive positions of flags laid out as i = tid + r*WG; returns total
__device__ __forceinline__ uint32_t block_scan2(Smem& s, bool f0, bool f1, uint32_t& pos0, uint32_t& pos1) {
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
  __syncthreads(); // wave_cnt reuse
  if (lane == 0) {
    s.wave_cnt[wave] = __popcll(b0);
    s.wave_cnt[WAVES + wave] = __popcll(b1);
  }
  __syncthreads();
  uint32_t base0 = 0, base1 = 0, tot0 = 0, tot1 = 0;
#pragma unroll
  for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
    const uint32_t c0 = s.wave_cnt[w], c1 = s.wave_cnt[WAVES + w];
    if (w < wave) base0 += c0, base1 += c1;
    tot0 += c0;
    tot1 += c1;
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  pos0 = base0 + __popcll(b0 & lt);
  pos1 = tot0 + base1 + __popcll(b1 & lt);
  return tot0 + tot1;
}

__global__ __launch_bounds__(WG) void scan_kernel(ScanArgs a) {
  __shared__ Smem s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= a.n_items) return;
  const DevItem item = a.items[blockIdx.x];
  const DevQuery* __restrict__ Q = a.queries + item.query;
  const uint32_t nterms = Q->n_terms, K = Q->k, ranker = Q->ranker;
  const uint32_t nw = Q->n_weights < 32u ? Q->n_weights : 32u;
  const uint32_t index_weight = Q->index_weight;
  const DevTerm T0 = Q->t[0];
  if (tid < 32) s.weights[tid] = Q->weights[tid];
  if (tid == 0) {
    s.cand_n = 0;
    s.tau = 0;
  }
  uint32_t total = 0; // matches seen by this thread's slots
  __syncthreads();

  for (uint32_t g = item.blk_begin; g < item.blk_end; g += T0_BLOCKS) {
    // refresh the shared threshold (any value <= the true K-th best key is safe)
    if (tid == 0) {
      const uint64_t gt = __hip_atomic_load(a.q_tau + item.query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gt > s.tau) s.tau = gt;
    }
    // ---- driver term: one block per wave
    Dec d;
    d.ok[0] = d.ok[1] = false;
    d.rowid[0] = d.rowid[1] = d.tf[0] = d.tf[1] = d.fields[0] = d.fields[1] = 0;
    const uint32_t b = g + wave;
    if (b < item.blk_end) {
      const uint32_t left = T0.docs - b * DEVBLK;
      decode_block(a.seg, T0, b, left < (uint32_t)DEVBLK ? left : (uint32_t)DEVBLK, s.stage[wave], s.docstart[wave], d);
    }
    // field filter (FitsFields) + compaction into the tile arrays
    uint32_t mf[2];
    bool live[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      mf[r] = d.fields[r] & T0.queried32;
      live[r] = d.ok[r] && mf[r] != 0;
    }
    // tile order must be rowid order: wave w's entries l, l+64 sit at w*128 + l, w*128 + 64 + l;
    // scan them as two sub-tiles per wave
    uint32_t n0;
    {
      const uint64_t b0 = __ballot(live[0]), b1 = __ballot(live[1]);
      __syncthreads();
      if (lane == 0) {
        s.wave_cnt[2 * wave] = __popcll(b0);
        s.wave_cnt[2 * wave + 1] = __popcll(b1);
      }
      __syncthreads();
      uint32_t basew = 0, tot = 0;
#pragma unroll
      for (uint32_t w = 0; w < 2u * WAVES; ++w) {
        const uint32_t c = s.wave_cnt[w];
        if (w < 2 * wave) basew += c;
        tot += c;
      }
      n0 = tot;
      const uint64_t lt = (1ull << lane) - 1ull;
      const uint32_t p0 = basew + __popcll(b0 & lt);
      const uint32_t p1 = basew + __popcll(b0) + __popcll(b1 & lt);
      if (live[0]) {
        s.t0_rowid[p0] = d.rowid[0];
        s.t0_acc[p0] = 0.0f + term_tfidf(d.tf[0], T0.idf);
        s.t0_fields[p0] = mf[0];
      }
      if (live[1]) {
        s.t0_rowid[p1] = d.rowid[1];
        s.t0_acc[p1] = 0.0f + term_tfidf(d.tf[1], T0.idf);
        s.t0_fields[p1] = mf[1];
      }
    }
    __syncthreads();

    // ---- the other terms, in ascending-docs order
    for (uint32_t j = 1; j < nterms && n0 > 0; ++j) {
      const DevTerm Tj = Q->t[j];
      const uint32_t* __restrict__ gbase = a.seg.blk_base + Tj.blk_first;
      if (tid == 0) s.rng_lo = find_block(gbase, Tj.nblocks, s.t0_rowid[0]);
      if (tid == 64) s.rng_hi = find_block(gbase, Tj.nblocks, s.t0_rowid[n0 - 1]);
      __syncthreads();
      const uint32_t rlo = s.rng_lo, rn = s.rng_hi - rlo + 1;
      const bool in_lds = rn <= (uint32_t)RNG_CAP;
      if (in_lds)
        for (uint32_t i = tid; i < rn; i += WG) s.rng_base[i] = gbase[rlo + i];
      __syncthreads();
      // block of every driver doc (HintRowID's FindSpan)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t i = tid + r * WG;
        if (i < n0) {
          const uint32_t rowid = s.t0_rowid[i];
          s.blk_of[i] = rlo + (in_lds ? find_block(s.rng_base, rn, rowid) : find_block(gbase + rlo, rn, rowid));
        }
      }
      __syncthreads();
      // distinct blocks, in order
      bool head[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t i = tid + r * WG;
        head[r] = i < n0 && (i == 0 || s.blk_of[i - 1] != s.blk_of[i]);
      }
      uint32_t hp0, hp1;
      const uint32_t n_need = block_scan2(s, head[0], head[1], hp0, hp1);
      {
        // need_idx[i] = (#heads at or before i) - 1
        const uint32_t i0 = tid, i1 = tid + WG;
        if (i0 < n0) s.need_idx[i0] = (uint16_t)(hp0 + (head[0] ? 1 : 0) - 1);
        if (i1 < n0) s.need_idx[i1] = (uint16_t)(hp1 + (head[1] ? 1 : 0) - 1);
        if (head[0]) s.need_list[hp0] = s.blk_of[i0];
        if (head[1]) s.need_list[hp1] = s.blk_of[i1];
      }
      __syncthreads();

      for (uint32_t pass = 0; pass < n_need; pass += SLOTS) {
        for (uint32_t sl = wave; sl < (uint32_t)SLOTS && pass + sl < n_need; sl += WAVES) {
          const uint32_t bj = s.need_list[pass + sl];
          const uint32_t left = Tj.docs - bj * DEVBLK;
          Dec e;
          decode_block(a.seg, Tj, bj, left < (uint32_t)DEVBLK ? left : (uint32_t)DEVBLK, s.stage[wave], s.docstart[wave], e);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const uint32_t o = lane + 64 * r;
            s.tj_rowid[sl][o] = e.ok[r] ? e.rowid[r] : MRK_INVALID_ROWID;
            s.tj_tf[sl][o] = e.tf[r];
            s.tj_fields[sl][o] = e.fields[r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint32_t i = tid + r * WG;
          if (i < n0) {
            const uint32_t sl = (uint32_t)s.need_idx[i] - pass;
            if (sl < (uint32_t)SLOTS) {
              const uint32_t rowid = s.t0_rowid[i];
              const uint32_t* arr = s.tj_rowid[sl];
              uint32_t pos = 0;
#pragma unroll
              for (uint32_t step = DEVBLK / 2; step; step >>= 1)
                if (arr[pos + step - 1] < rowid) pos += step;
              bool hit = arr[pos] == rowid;
              if (hit) {
                const uint32_t f = s.tj_fields[sl][pos] & Tj.queried32;
                if (f) {
                  s.t0_acc[i] = s.t0_acc[i] + term_tfidf(s.tj_tf[sl][pos], Tj.idf);
                  s.t0_fields[i] |= f;
                } else
                  hit = false;
              }
              if (!hit) s.blk_of[i] = NOBLK;
            }
          }
        }
        __syncthreads();
      }
      // drop the docs term j rejected
      {
        uint32_t rr[2], ff[2];
        float aa[2];
        bool keep[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint32_t i = tid + r * WG;
          keep[r] = i < n0 && s.blk_of[i] != NOBLK;
          if (keep[r]) rr[r] = s.t0_rowid[i], aa[r] = s.t0_acc[i], ff[r] = s.t0_fields[i];
        }
        uint32_t q0, q1;
        n0 = block_scan2(s, keep[0], keep[1], q0, q1);
        if (keep[0]) s.t0_rowid[q0] = rr[0], s.t0_acc[q0] = aa[0], s.t0_fields[q0] = ff[0];
        if (keep[1]) s.t0_rowid[q1] = rr[1], s.t0_acc[q1] = aa[1], s.t0_fields[q1] = ff[1];
        __syncthreads();
      }
    }

    // ---- n0 matches: weight, threshold, candidates
    if (s.cand_n > (uint32_t)(CAND - TILE)) compact_cand(s, K, a.q_tau + item.query);
    const uint64_t tau = s.tau;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const uint32_t i = tid + r * WG;
      bool push = false;
      uint64_t key = 0;
      if (i < n0) {
        ++total;
        uint32_t weight;
        if (ranker == MRK_RANK_NONE)
          weight = 1u; // ExtRanker_None_c, sphinxsearch.cpp:1160
        else {
          // ExtRanker_WeightSum_c<BM25>, sphinxsearch.cpp:1070, 1112-1129
          const int32_t bm = (int32_t)((s.t0_acc[i] + 0.5f) * 1000.0f);
          const uint32_t mask = s.t0_fields[i];
          uint32_t rank = 0;
          if (!mask)
            rank = 1;
          else
            for (uint32_t f = 0; f < nw; ++f)
              if (mask & (1u << f)) rank += (uint32_t)s.weights[f];
          weight = (uint32_t)bm + rank * 1000u;
        }
        weight *= index_weight; // MatchExtended, sphinx.cpp:12220
        key = make_key((int32_t)weight, a.seg.rowid_base + s.t0_rowid[i]);
        push = key >= tau;
      }
      const uint64_t bal = __ballot(push);
      if (bal) {
        uint32_t basep = 0;
        if (lane == 0) basep = atomicAdd(&s.cand_n, (uint32_t)__popcll(bal));
        basep = __shfl(basep, 0, 64);
        if (push) s.cand[basep + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      }
    }
    __syncthreads();
  }

  // ---- item epilogue
  if (s.cand_n > K) compact_cand(s, K, a.q_tau + item.query);
  // total_found: every match counts (CSphMatchQueue::PushT ++m_iTotal, sphinxsort.cpp:724)
  {
    uint32_t t = total;
    for (int dlt = 32; dlt; dlt >>= 1) t += __shfl_down(t, dlt, 64);
    if (lane == 0 && t) atomicAdd((unsigned long long*)(a.q_total + item.query), (unsigned long long)t);
  }
  __syncthreads();
  // publish candidates that can still make the query's top-K
  if (tid == 0) {
    const uint64_t gt = __hip_atomic_load(a.q_tau + item.query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gt > s.tau) s.tau = gt;
  }
  __syncthreads();
  {
    const uint64_t tau = s.tau;
    const uint32_t n = s.cand_n;
    uint64_t* dst = a.item_cand + (uint64_t)blockIdx.x * KCAP;
    uint32_t written = 0;
    for (uint32_t base_i = 0; base_i < n; base_i += WG) {
      const uint32_t i = base_i + tid;
      const bool w = i < n && s.cand[i] >= tau;
      uint32_t p0, p1;
      const uint32_t c = block_scan2(s, w, false, p0, p1);
      if (w) dst[written + p0] = s.cand[i];
      written += c;
    }
    if (tid == 0) a.item_cnt[blockIdx.x] = written;
  }
}

// ---------------------------------------------------------------------------------------
// merge: top-K of candidate lists, one workgroup per query
// ---------------------------------------------------------------------------------------
constexpr int LCH = 1024; // lists handled per chunk

struct __align__(16) MergeSmem {
  uint64_t cand[CAND];
  uint32_t lpre[LCH]; // inclusive prefix of list lengths
  uint32_t wave_cnt[2 * WAVES];
  uint32_t cand_n;
  uint64_t tau;
};

__device__ uint32_t merge_compact(MergeSmem& s, uint32_t k) {
  __syncthreads();
  const uint32_t n = s.cand_n;
  for (uint32_t i = n + threadIdx.x; i < (uint32_t)CAND; i += WG) s.cand[i] = 0;
  __syncthreads();
  sort_cand_desc(s.cand);
  const uint32_t keep = n < k ? n : k;
  if (threadIdx.x == 0) {
    s.cand_n = keep;
    if (keep == k) s.tau = s.cand[k - 1];
  }
  __syncthreads();
  return keep;
}

__global__ __launch_bounds__(WG) void merge_kernel(MergeArgs a) {
  __shared__ MergeSmem s;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (q >= a.n_queries) return;
  const uint32_t k = a.k_per_query ? a.k_per_query[q] : a.k;
  if (tid == 0) {
    s.cand_n = 0;
    s.tau = 0;
  }
  __syncthreads();
  const uint32_t nl = a.list_first ? a.list_n[q] : a.n_lists;
  for (uint32_t lbase = 0; lbase < nl; lbase += LCH) {
    const uint32_t nlc = nl - lbase < (uint32_t)LCH ? nl - lbase : (uint32_t)LCH;
    // list lengths -> inclusive prefix sums in LDS (4 lists per thread)
    uint32_t c[4], sum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t l = tid * 4 + r;
      uint32_t cnt = 0;
      if (l < nlc) {
        const uint32_t li = a.list_first ? a.list_first[q] + lbase + l : (lbase + l) * a.n_queries + q;
        cnt = a.in_cnt[li];
        if (cnt > (uint32_t)KCAP) cnt = KCAP;
      }
      sum += cnt;
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

    for (uint32_t f0 = 0; f0 < total; f0 += 4 * WG) {
      if (s.cand_n > (uint32_t)(CAND - 4 * WG)) merge_compact(s, k);
      const uint64_t tau = s.tau;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t f = f0 + r * WG + tid;
        uint64_t key = 0;
        bool push = false;
        if (f < total) {
          // list = first l with lpre[l] > f
          uint32_t lo = 0, hi = nlc - 1;
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s.lpre[mid] > f)
              hi = mid;
            else
              lo = mid + 1;
          }
          const uint32_t idx = f - (lo ? s.lpre[lo - 1] : 0u);
          const uint32_t li = a.list_first ? a.list_first[q] + lbase + lo : (lbase + lo) * a.n_queries + q;
          key = a.in_keys[(uint64_t)li * KCAP + idx];
          push = key >= tau && key != 0;
        }
        const uint64_t bal = __ballot(push);
        if (bal) {
          uint32_t basep = 0;
          if (lane == 0) basep = atomicAdd(&s.cand_n, (uint32_t)__popcll(bal));
          basep = __shfl(basep, 0, 64);
          if (push) s.cand[basep + __popcll(bal & ((1ull << lane) - 1ull))] = key;
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }
  const uint32_t n = merge_compact(s, k);
  for (uint32_t i = tid; i < n; i += WG) a.out_keys[(uint64_t)q * KCAP + i] = s.cand[i];
  if (tid == 0) a.out_cnt[q] = n;
}

void launch_scan(const ScanArgs& a, void* stream) {
  if (!a.n_items) return;
  hipLaunchKernelGGL(scan_kernel, dim3(a.n_items), dim3(WG), 0, (hipStream_t)stream, a);
}

void launch_merge(const MergeArgs& a, void* stream) {
  if (!a.n_queries) return;
  hipLaunchKernelGGL(merge_kernel, dim3(a.n_queries), dim3(WG), 0, (hipStream_t)stream, a);
}

} // namespace mrk
