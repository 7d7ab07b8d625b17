// mrk_kernels.hip -- gfx950 (MI355X, wave64) kernels of the match -> rank -> top-K path.
//
// scan_kernel  : one workgroup per work item = (query, range of driver-term blocks); each of
//                its 4 waves owns a contiguous run of driver blocks and works on its own:
//                it streams the driver term's VLB doclist blocks and, for every other term,
//                the blocks its driver docs can fall into (2-stream merge when the lists
//                are dense, skiplist-style jumps when they are not), decodes them in LDS
//                (terminator-bit prefix sums split the byte stream into entries, one entry
//                per lane), probes, scores in fp32 in the reference's op order and pushes
//                candidates above the running threshold into the workgroup's top-K buffer.
//                Waves only meet at one barrier per driver block (buffer compaction).
// merge_kernel : one workgroup per query: top-K of the candidates the scan produced
//                (or of per-shard partial top-K lists), sorted best-first.
//
// What each piece restates (reference file:line, Manticore 3.6.1 src/):
//   decode_block      DiskIndexQword_c::ReadNext         sphinx.cpp:511-549
//   block seek        DiskIndexQword_c::HintRowID        sphinx.cpp:407-451
//   intersection      ExtMultiAnd_T::AdvanceQwords       searchnode.cpp:2865-2889
//   field filter      NodeInfo_t::FitsFields             searchnode.cpp:2727-2747
//   tfidf             ExtMultiAnd_T::GetTFIDF            searchnode.cpp:2821-2832
//   weights           GetFilteredDocs / WeightSum / None sphinxsearch.cpp:1070, 1097-1169
//   top-K order       MatchRelevanceLt_fn                sphinxsort.cpp:4541-4547
// Integer / byte work on HBM-bound streams: no MFMA by design.
#include "mrk_kcommon.h"

namespace mrk {

struct __align__(16) WaveLds {
  uint8_t stage[STAGE_BYTES + 16];
  uint32_t tj_rowid[DEVBLK];
  uint32_t tj_tf[DEVBLK];
  uint32_t tj_fields[DEVBLK];
  uint16_t docstart[DEVBLK + 8];
};

struct __align__(16) Smem {
  WaveLds w[WAVES];
  uint64_t cand[CAND];
  int32_t weights[32];
  uint32_t wave_cnt[2 * WAVES];
  uint32_t cand_n;
  uint64_t tau;
};

// ---- block metadata: 64 consecutive entries of a term's block index, one per lane
struct Chunk {
  uint32_t first; // block index (within the term) held by lane 0
  uint32_t bp1;   // SkiplistEntry_t::m_tBaseRowIDPlus1 of block first+lane (INF past the end)
  uint64_t off;   // .spd offset of block first+lane (doclist end past the end)
};

__device__ __forceinline__ void load_chunk(Chunk& c, const DevSegment& seg, const DevTerm& T, uint32_t first) {
  const uint32_t i = first + lane_id();
  c.first = first;
  if (i < T.nblocks) {
    c.bp1 = seg.blk_base[T.blk_first + i];
    c.off = seg.blk_off[T.blk_first + i];
  } else {
    c.bp1 = INF_ROWID;
    c.off = T.spd_end;
  }
}

struct BlkMeta {
  uint64_t p0, p1; // byte range in .spd
  uint32_t base;   // decoder rowid before the block's first entry (sphinx.cpp:447)
  uint32_t nd;     // entries in the block
};

// i = index into the chunk (wave-uniform, < CHUNK)
__device__ __forceinline__ BlkMeta chunk_meta(const Chunk& c, const DevTerm& T, uint32_t i) {
  BlkMeta m;
  m.p0 = rdlane64(c.off, i);
  m.p1 = rdlane64(c.off, i + 1);
  m.base = rdlane(c.bp1, i) - 1u;
  const uint32_t left = T.docs - (c.first + i) * DEVBLK;
  m.nd = left < (uint32_t)DEVBLK ? left : (uint32_t)DEVBLK;
  return m;
}

// first 1 KiB of a block's byte run, 16 B per lane, straight into registers (issued early, used late)
__device__ __forceinline__ uint4 issue_strip(const uint8_t* __restrict__ spd, const BlkMeta& m) {
  const uint64_t a = m.p0 & ~15ull;
  const uint64_t span = m.p1 > a ? m.p1 - a : 0;
  const uint32_t my = lane_id() * 16;
  uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
  if ((uint64_t)my < span) v = *reinterpret_cast<const uint4*>(spd + a + my);
  return v;
}

struct Dec {
  uint32_t rowid[2], tf[2], fields[2];
  bool ok[2];
};

// One wave decodes one device block (<= 128 doclist entries): lane l returns entries l and l+64.
__device__ void decode_block(const uint8_t* __restrict__ spd, const bool inline_hits, const BlkMeta& m, uint4 first_strip,
                             WaveLds& L, Dec& out) {
  const uint32_t lane = lane_id();
  uint8_t* stage = L.stage;
  uint16_t* docstart = L.docstart;
  const uint64_t a = m.p0 & ~15ull;
  const uint32_t lead = (uint32_t)(m.p0 - a);
  const uint64_t span = m.p1 > a ? m.p1 - a : 0;
  const uint32_t nbytes = span < (uint64_t)STAGE_BYTES ? (uint32_t)span : (uint32_t)STAGE_BYTES;
  const uint32_t nd = m.nd;

  docstart[lane] = 0xFFFF;
  docstart[lane + 64] = 0xFFFF;
  if (lane == 0) docstart[0] = (uint16_t)lead;

  // ---- phase A: stage bytes, number the varints, record where every 4th one ends
  uint32_t carry = 0;
  for (uint32_t off = 0; off < nbytes; off += 1024) {
    const uint32_t my = off + lane * 16;
    uint4 v = first_strip;
    if (off) {
      v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
      if (my < nbytes) v = *reinterpret_cast<const uint4*>(spd + a + my);
    }
    if (my < nbytes) *reinterpret_cast<uint4*>(stage + my) = v;
    uint32_t t = term4(v.x) | (term4(v.y) << 4) | (term4(v.z) << 8) | (term4(v.w) << 12);
    if (my + 16 > nbytes) t = (my < nbytes) ? (t & ((1u << (nbytes - my)) - 1u)) : 0u;
    if (my < lead) t &= ~((1u << (lead - my)) - 1u);
    const uint32_t cnt = __popc(t);
    const uint32_t inc = wave_incl_scan(cnt);
    const uint32_t s = carry + inc - cnt; // index of my first varint
    // every doclist entry is exactly 4 varints (sphinx.cpp:8456-8490): entry e ends with varint 4e+3
    uint32_t mm = t;
    const uint32_t r = (3u - s) & 3u;
    for (uint32_t i = 0; i < r; ++i) mm &= mm - 1;
    uint32_t idx = s + r;
    while (mm) {
      const uint32_t pos = __builtin_ctz(mm);
      const uint32_t doc = (idx >> 2) + 1;
      if (doc < nd) docstart[doc] = (uint16_t)(my + pos + 1);
      mm &= mm - 1;
      mm &= mm - 1;
      mm &= mm - 1;
      mm &= mm - 1;
      idx += 4;
    }
    carry += rdlane(inc, 63);
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
    uint32_t v0 = 0, v1 = 0, vf = 0;
    bool lone = false; // inline format: hits == 1
    bool fast = false;
    uint64_t x = 0;
    if (ok) {
      const uint32_t* s32 = reinterpret_cast<const uint32_t*>(stage);
      const uint32_t wi = st >> 2, sh = st & 3u;
      const uint32_t w0 = s32[wi], w1 = s32[wi + 1], w2 = s32[wi + 2];
      const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
      const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
      x = ((uint64_t)hi << 32) | lo;
      // fast path: the entry fits bytes 0..6 of the window and no varint is longer than 2 bytes
      const uint32_t T = (term4(lo) | (term4(hi) << 4)) & 0x7Fu;
      const uint32_t T1 = T & (T - 1), T2 = T1 & (T1 - 1), T3 = T2 & (T2 - 1);
      if (T3) {
        const uint32_t e0 = __builtin_ctz(T), e1 = __builtin_ctz(T1), e2 = __builtin_ctz(T2), e3 = __builtin_ctz(T3);
        const uint32_t C = ~T & ((2u << e3) - 1u); // continuation bytes inside the entry
        if (!(C & (C >> 1))) {
          fast = true;
          const uint64_t xs = x << 8; // byte e-1 (or 0) lands at bits 0..7 of (xs >> 8e), byte e at 8..15
          auto val2 = [&](uint32_t e) -> uint32_t {
            const uint32_t w = (uint32_t)(xs >> (8 * e));
            const uint32_t last = (w >> 8) & 0x7fu;
            const uint32_t prev = (w & 0x80u) ? ((w & 0x7fu) << 7) : 0u;
            return prev | last;
          };
          v0 = val2(e0);
          v1 = val2(e1);
          if (inline_hits) {
            lone = v1 == 1;
            vf = val2(lone ? e3 : e2);
          } else {
            // plain: delta, hitlist offset delta, fieldmask, hits
            vf = val2(e2);
            v1 = val2(e3);
          }
        }
      }
    }
    if (ok && !fast) {
      // general path (rare): byte loop out of LDS, any varint length
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
      v0 = vv[0];
      if (inline_hits) {
        v1 = vv[1];
        lone = v1 == 1;
        vf = lone ? vv[3] : vv[2];
      } else {
        vf = vv[2];
        v1 = vv[3];
      }
    }
    // ReadNext (sphinx.cpp:511-549)
    uint32_t fields;
    if (lone) { // lone hit inlined: vf = field<<1 | end
      const uint32_t f = (vf >> 1) & 255u;
      fields = f < 32 ? (1u << f) : 0u;
    } else
      fields = vf;
    delta[r] = ok ? v0 : 0u;
    out.tf[r] = v1;
    out.fields[r] = fields;
    out.ok[r] = ok;
  }
  const uint32_t s0 = wave_incl_scan(delta[0]);
  const uint32_t tot0 = rdlane(s0, 63);
  const uint32_t s1 = wave_incl_scan(delta[1]);
  out.rowid[0] = m.base + s0;
  out.rowid[1] = m.base + tot0 + s1;
  wave_lds_fence(); // stage/docstart are reused by this wave's next block
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
  const uint8_t* __restrict__ spd = a.seg.spd;
  const bool inline_hits = a.seg.inline_hits != 0;
  WaveLds& L = s.w[wave];
  if (tid < 32) s.weights[tid] = Q->weights[tid];
  if (tid == 0) {
    s.cand_n = 0;
    s.tau = 0;
  }
  // this wave's run of driver blocks
  const uint32_t nb = item.blk_end - item.blk_begin;
  const uint32_t per = (nb + WAVES - 1) / WAVES;
  const uint32_t wb0 = item.blk_begin + wave * per;
  const uint32_t wb1 = wb0 + per < item.blk_end ? wb0 + per : item.blk_end;

  uint32_t total = 0; // matches seen by this lane
  Chunk c0;           // driver-term block index chunk
  Chunk cj;           // other-term chunk (valid for term cj_term)
  uint32_t cj_term = 0, slot_blk = NOBLK, slot_term = 0;
  uint32_t kj = 0; // cursor into the other term's blocks (2-term queries: persists across driver blocks)
  BlkMeta m0{};
  uint4 strip0 = make_uint4(0, 0, 0, 0);
  if (wb0 < wb1) {
    load_chunk(c0, a.seg, T0, wb0);
    m0 = chunk_meta(c0, T0, 0);
    strip0 = issue_strip(spd, m0);
  }
  __syncthreads();

  for (uint32_t it = 0; it < per; ++it) {
    const uint32_t b = wb0 + it;
    if (b < wb1) { // wave-uniform
      if (lane == 0) { // refresh the shared threshold (any value <= the final K-th best key is safe)
        const uint64_t gt = __hip_atomic_load(a.q_tau + item.query, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gt > s.tau) atomicMax((unsigned long long*)&s.tau, (unsigned long long)gt);
      }
      // ---- driver block b (strip already in flight); issue the next one before decoding
      BlkMeta m_next{};
      uint4 strip_next = make_uint4(0, 0, 0, 0);
      if (b + 1 < wb1) {
        if (b + 1 - c0.first >= (uint32_t)CHUNK) load_chunk(c0, a.seg, T0, b + 1);
        m_next = chunk_meta(c0, T0, b + 1 - c0.first);
        strip_next = issue_strip(spd, m_next);
      }
      Dec d;
      decode_block(spd, inline_hits, m0, strip0, L, d);
      m0 = m_next;
      strip0 = strip_next;

      uint32_t fld[2];
      float acc[2];
      bool live[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        fld[r] = d.fields[r] & T0.queried32; // FitsFields
        live[r] = d.ok[r] && fld[r] != 0;
        acc[r] = 0.0f + term_tfidf(d.tf[r], T0.idf);
      }

      // ---- the other terms, in ascending-docs order
      for (uint32_t j = 1; j < nterms; ++j) {
        if (!__ballot(live[0] || live[1])) break;
        const DevTerm Tj = Q->t[j];
        const uint32_t* __restrict__ gbase = a.seg.blk_base + Tj.blk_first;
        bool done[2] = {!live[0], !live[1]};
        bool hit[2] = {false, false};
        if (cj_term != j) { // (re)bind the chunk/cursor to this term
          cj_term = j;
          kj = 0;
          cj.first = NOBLK;
        }
        for (;;) {
          // smallest driver rowid still waiting for this term
          const uint32_t r_min = wave_min(min(done[0] ? INF_ROWID : d.rowid[0], done[1] ? INF_ROWID : d.rowid[1]));
          if (r_min == INF_ROWID) break;
          // block of r_min: the last block whose base <= r_min (HintRowID's FindSpan), never behind the cursor
          if (cj.first == NOBLK || kj < cj.first || kj - cj.first >= (uint32_t)CHUNK) load_chunk(cj, a.seg, Tj, kj);
          {
            const uint64_t le = __ballot(cj.bp1 <= r_min);
            uint32_t p = le ? 63u - (uint32_t)__builtin_clzll(le) : 0u; // chunk entries are ascending
            if (p >= (uint32_t)CHUNK) { // beyond this chunk: jump with a wave-wide search, then reload
              kj = wave_find_block(gbase, cj.first + CHUNK - 1, Tj.nblocks, r_min);
              load_chunk(cj, a.seg, Tj, kj);
              p = 0;
            }
            const uint32_t k_new = cj.first + p;
            if (k_new > kj) kj = k_new;
          }
          const uint32_t ci = kj - cj.first;
          const uint32_t bp1_k = rdlane(cj.bp1, ci), bp1_n = rdlane(cj.bp1, ci + 1);
          if (slot_blk != kj || slot_term != j) {
            const BlkMeta mj = chunk_meta(cj, Tj, ci);
            const uint4 sj = issue_strip(spd, mj);
            Dec e;
            decode_block(spd, inline_hits, mj, sj, L, e);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const uint32_t o = lane + 64 * r;
              L.tj_rowid[o] = e.ok[r] ? e.rowid[r] : MRK_INVALID_ROWID;
              L.tj_tf[o] = e.tf[r];
              L.tj_fields[o] = e.fields[r];
            }
            slot_blk = kj;
            slot_term = j;
            wave_lds_fence();
          }
          // probe: driver docs that fall into [bp1_k, bp1_n)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            if (!done[r] && d.rowid[r] >= bp1_k && d.rowid[r] < bp1_n) {
              done[r] = true;
              const uint32_t rowid = d.rowid[r];
              uint32_t pos = 0;
#pragma unroll
              for (uint32_t step = DEVBLK / 2; step; step >>= 1)
                if (L.tj_rowid[pos + step - 1] < rowid) pos += step;
              if (L.tj_rowid[pos] == rowid) {
                const uint32_t f = L.tj_fields[pos] & Tj.queried32;
                if (f) {
                  hit[r] = true;
                  acc[r] = acc[r] + term_tfidf(L.tj_tf[pos], Tj.idf);
                  fld[r] |= f;
                }
              }
            }
          }
          // docs below bp1_k cannot exist (blocks only move forward); guard against a stall anyway
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (!done[r] && d.rowid[r] < bp1_k) done[r] = true;
        }
        live[0] = live[0] && hit[0];
        live[1] = live[1] && hit[1];
      }
      if (a.seg.dead) { // MatchExtended drops dead rows before they reach the sorter (sphinx.cpp:12213-12217)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (live[r] && row_is_dead(a.seg, d.rowid[r])) live[r] = false;
      }

      // ---- matches: weight, threshold, candidates
      const uint64_t tau = s.tau;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        bool push = false;
        uint64_t key = 0;
        if (live[r]) {
          ++total;
          uint32_t weight;
          if (ranker == MRK_RANK_NONE)
            weight = 1u; // ExtRanker_None_c, sphinxsearch.cpp:1160
          else {
            // ExtRanker_WeightSum_c<BM25>, sphinxsearch.cpp:1070, 1112-1129
            const int32_t bm = (int32_t)((acc[r] + 0.5f) * 1000.0f);
            const uint32_t mask = fld[r];
            uint32_t rank = 0;
            if (!mask)
              rank = 1;
            else
              for (uint32_t f = 0; f < nw; ++f)
                if (mask & (1u << f)) rank += (uint32_t)s.weights[f];
            weight = (uint32_t)bm + rank * 1000u;
          }
          weight *= index_weight; // MatchExtended, sphinx.cpp:12220
          key = make_key((int32_t)weight, a.seg.rowid_base + d.rowid[r]);
          push = key >= tau;
        }
        const uint64_t bal = __ballot(push);
        if (bal) {
          uint32_t basep = 0;
          if (lane == 0) basep = atomicAdd(&s.cand_n, (uint32_t)__popcll(bal));
          basep = rdlane(basep, 0);
          if (push) s.cand[basep + __popcll(bal & ((1ull << lane) - 1ull))] = key;
        }
      }
    }
    __syncthreads();
    if (s.cand_n > (uint32_t)(CAND - WAVES * DEVBLK)) compact_cand(s, K, a.q_tau + item.query);
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
  // publish the candidates that can still make the query's top-K
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
      const uint32_t c = block_scan2(s.wave_cnt, w, false, p0, p1);
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
        cnt = a.in_rows ? (uint32_t)a.in_rows[(uint64_t)li * ROW_WORDS + KCAP] : a.in_cnt[li];
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
          key = a.in_rows ? a.in_rows[(uint64_t)li * ROW_WORDS + idx] : a.in_keys[(uint64_t)li * KCAP + idx];
          push = key >= tau && key != 0;
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
  const uint32_t n = merge_compact(s, k);
  if (a.out_rows) {
    uint64_t* __restrict__ row = a.out_rows + (uint64_t)q * ROW_WORDS;
    for (uint32_t i = tid; i < (uint32_t)KCAP; i += WG) row[i] = i < n ? s.cand[i] : 0ull;
    if (tid == 0) {
      // CSphMatchQueue::MoveTo adds the totals up (sphinxsort.cpp:681-710); the two flag bits of a shard's row
      // (ROW_RERUN / ROW_DECLINED) are OR-ed through so that the receiver sees them whatever the other shards sent
      uint64_t total = 0, flags = 0;
      for (uint32_t l = 0; l < nl; ++l) {
        const uint64_t t = a.in_rows[((uint64_t)l * a.n_queries + q) * ROW_WORDS + KCAP + 1];
        total += t & ~ROW_FLAG_MASK;
        flags |= t & ROW_FLAG_MASK;
      }
      row[KCAP] = n;
      row[KCAP + 1] = (total & ~ROW_FLAG_MASK) | flags;
    }
    return;
  }
  for (uint32_t i = tid; i < n; i += WG) a.out_keys[(uint64_t)q * KCAP + i] = s.cand[i];
  if (tid == 0) a.out_cnt[q] = n;
}

// a batch's results as exchange rows: KCAP keys | count | total_found
__global__ __launch_bounds__(WG) void pack_rows_kernel(PackRowsArgs a) {
  const uint32_t q = blockIdx.x;
  if (q >= a.n) return;
  uint64_t* __restrict__ row = a.rows + (uint64_t)q * ROW_WORDS;
  // a query whose candidate list overflowed has no trustworthy list on the device until the host reran it: its row goes
  // out empty with ROW_RERUN set in the total_found word; a query this shard declined (MRK_E_UNSUPPORTED) goes out empty
  // with ROW_DECLINED.  The merge ORs both bits through: the receiver reruns / fails the query, never a partial answer.
  const bool declined = a.declined && a.declined[q] != 0;
  const bool bad = declined || (a.flags && (a.flags[q] & (QF_OVERFLOW | QF_FSM)) != 0);
  const uint32_t n = bad ? 0u : a.cnt[q] < (uint32_t)KCAP ? a.cnt[q] : (uint32_t)KCAP;
  for (uint32_t i = threadIdx.x; i < (uint32_t)KCAP; i += WG) row[i] = i < n ? a.keys[(uint64_t)q * KCAP + i] : 0ull;
  if (threadIdx.x == 0) {
    row[KCAP] = n;
    row[KCAP + 1] = declined ? ROW_DECLINED : bad ? ROW_RERUN : (a.total[q] & ~ROW_FLAG_MASK);
  }
}

void launch_pack_rows(const PackRowsArgs& a, void* stream) {
  if (!a.n) return;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(a.n), dim3(WG), 0, (hipStream_t)stream, a);
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
