// mrk_scan_bm.hip -- 2-keyword AND over doc-set BITMAPS (dense keywords), gfx950 / wave64.
//
// A keyword found in >= 1/bitmap_inv of a segment's docs carries, next to its packed doclist, a
// bitmap of its doc set (mrk_pack.cpp).  ExtMultiAnd_T's leap-frog over two such keywords
// (searchnode.cpp:2836-2981) then collapses to one AND per 32 rowids:
//
//   window = 2048 rowids: lane l holds word l of both bitmaps (two coalesced 256 B loads),
//   m = a & b (& ~dead) are the window's matches.  A match's slot in each keyword's packed arrays
//   is its RANK in that keyword's doc set: rank at the window start (running count, seeded from the
//   rank directory) + popcounts of the lower lanes' words (one packed wave prefix sum for both
//   keywords) + popcount of the lower bits of the lane's own word.
//
// Matches are queued per wave in LDS (rowid, rankA, rankB) and scored 64 at a time with every lane
// busy: tf / field bytes come from the packed attr words (pk_attr, gathered by rank), BM25 and the
// field-weight sum from the same per-workgroup tables and with the same fp32 operation order as the
// packed scan kernel, then the pruning histogram / candidate list shared with it (mrk_kprune.h).
// Field-limited keywords are honoured at scoring time (a doc counts for a keyword only if
// fields & queried != 0, searchnode.cpp:1925-1939), so totals are counted there too.
#include "mrk_kcommon.h"
#include "mrk_kprune.h"

#ifndef MRK_BMEXP
#define MRK_BMEXP 0 // ablations for profiling: 1 no per-burst threshold load, 2 no scoring, 3 no match extraction
#endif

namespace mrk {

constexpr int BM_CBUF = 112; // candidates a wave collects before it publishes them
constexpr int BM_WORDS = 64; // words per window
#ifndef MRK_BM_BURST
#define MRK_BM_BURST 4 // windows requested back to back
#endif
constexpr int BM_BURST = MRK_BM_BURST;
constexpr int BM_WQCAP = 128; // word queue entries per wave (unpacked in batches of 64)

// (4 x (896 + 3072) B of wave queues + 3 KB of tables)
struct __align__(16) BmWaveLds {
  uint64_t cbuf[BM_CBUF];
  // word queue: the windows' non-empty match words wait here until 64 of them can be unpacked with every lane busy
  uint4 wq[BM_WQCAP];  // match word, keyword A's word, keyword B's word, rowid of bit 0
  uint2 wqr[BM_WQCAP]; // ranks of the words' first bits in A and in B
  uint16_t wpre[BM_WQCAP]; // final drain: matches held by the queue entries before this one (<= 128 x 32)
};

struct __align__(16) BmSmem {
  BmWaveLds w[WAVES];
  uint32_t rank[256];
  float tfidf[2][256];
};

// SEQ: the segment holds the dense keywords' tf / field bytes in slot order (DevSegment::pk_attr2) and the gathers read those
template <bool SEQ>
__global__ __launch_bounds__(WG) void scan_bm_kernel(ScanArgs a) {
  __shared__ BmSmem s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= a.n_items) return;
  const DevItem item = a.items[blockIdx.x];
  const DevQuery* __restrict__ Q = a.queries + item.query;
  const uint32_t K = Q->k, ranker = Q->ranker, oq = Q->out_q;
  const uint32_t nw = Q->n_weights < 8u ? Q->n_weights : 8u;
  const uint32_t index_weight = Q->index_weight;
  const DevTerm TA = Q->t[0], TB = Q->t[1]; // ExtMultiAnd_T node order: tfidf = 0 + A + B
  BmWaveLds& L = s.w[wave];
  s.tfidf[0][tid] = term_tfidf(tid, TA.idf);
  s.tfidf[1][tid] = term_tfidf(tid, TB.idf);
  {
    uint32_t rk = 0;
    if (!tid)
      rk = 1; // empty mask: "just fake it" (sphinxsearch.cpp:1114-1118)
    else
      for (uint32_t f = 0; f < nw; ++f)
        if (tid & (1u << f)) rk += (uint32_t)Q->weights[f];
    s.rank[tid] = rk;
  }
  const uint32_t bin_mode = Q->bin_mode, bin_shift = Q->bin_shift;
  const int32_t bin_lo = Q->bin_lo;
  const uint32_t cand_cap = Q->cand_cap;
  uint64_t* __restrict__ cand = a.cand + Q->cand_off;
  uint32_t* __restrict__ ghist = a.q_hist + (uint64_t)oq * NBINS;
  uint32_t* __restrict__ gcount = a.q_cand_n + (size_t)oq * QSTRIDE;
  uint32_t* __restrict__ gtaubin = a.q_tau_bin + (size_t)oq * QSTRIDE;
  const uint32_t* __restrict__ bmA = a.seg.bm + TA.bm_off;
  const uint32_t* __restrict__ bmB = a.seg.bm + TB.bm_off;
  const uint32_t* __restrict__ dead = a.seg.dead;
  const uint32_t* __restrict__ attr = SEQ ? nullptr : a.seg.pk_attr;
  const uint8_t* __restrict__ attr1 = SEQ ? nullptr : a.seg.pk_attr1; // (the SEQ instance is only launched without the nibble plane)
  const uint16_t* __restrict__ attr2 = SEQ ? a.seg.pk_attr2 : nullptr;
  // SPH_RANK_NONE without field limits: every common doc matches with weight 1 and the sorter keeps the lowest
  // rowids, so matches are counted straight off the match words, tf / field bytes are never fetched, and windows
  // that lie wholly behind the pruning threshold are not even unpacked
  const uint32_t field_all = nw >= 32u ? 0xFFFFFFFFu : (1u << nw) - 1u;
  const bool none_fast = ranker == MRK_RANK_NONE && (TA.queried32 & field_all) == field_all && (TB.queried32 & field_all) == field_all;

  const uint32_t nwin = item.blk_end - item.blk_begin;
  const uint32_t per = (nwin + WAVES - 1) / WAVES;
  const uint32_t w0 = item.blk_begin + wave * per;
  const uint32_t w1 = w0 + per < item.blk_end ? w0 + per : item.blk_end;
  __syncthreads(); // tables ready; the waves never meet again

  uint32_t total = 0, tau_bin = 0, cn = 0;
  // running ranks at the start of the next window (uniform)
  uint32_t baseA = 0, baseB = 0;
  if (w0 < w1) {
    baseA = a.seg.bm_dir[TA.dir_off + (uint64_t)w0 * (BM_WORDS / 8)];
    baseB = a.seg.bm_dir[TB.dir_off + (uint64_t)w0 * (BM_WORDS / 8)];
  }

  auto publish = [&]() {
    if (cn) {
      uint32_t basep = 0;
      if (lane == 0) basep = atomicAdd(gcount, cn);
      basep = rdlane(basep, 0);
      const bool fits = basep + cn <= cand_cap;
      const uint32_t npub = cn;
      wave_lds_fence();
      if (fits)
        for (uint32_t i = lane; i < cn; i += 64) cand[basep + i] = L.cbuf[i];
      else if (lane == 0)
        atomicOr(a.q_flags + oq, QF_OVERFLOW);
      hist_add_keys(ghist, L.cbuf, cn, bin_mode, bin_lo, bin_shift);
      wave_lds_fence();
      cn = 0;
      // only the publisher whose slice crosses a 2048-candidate boundary recomputes the threshold
      if ((basep >> 11) != ((basep + npub) >> 11) || basep == 0) {
        const uint32_t tb = threshold_bin(ghist, K);
        if (tb > tau_bin) {
          tau_bin = tb;
          if (lane == 0) atomicMax(gtaubin, tb);
        }
      }
    }
    const uint32_t gt = __hip_atomic_load(gtaubin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gt > tau_bin) tau_bin = gt;
  };

  // Score one match per lane (its rowid, its slots in the two keywords' packed arrays) -- in two halves a round apart: a call
  // first finishes the round the call before it started (weights, pruning test, candidate buffer), then only REQUESTS its own
  // matches' tf / field words.  The gathers are the kernel's longest waits (a line per lane); this way they are in flight
  // while the wave unpacks further words and works through the next windows, instead of being waited for on the spot.
  bool p_any = false, p_valid = false; // a round is pending; the lane holds a match of it
  uint32_t p_row = 0, p_ra = 0, p_rb = 0, p_wa = 0, p_wb = 0;
  auto score = [&](const bool valid_new, const uint32_t row_new, const uint32_t ra_new, const uint32_t rb_new) __attribute__((always_inline)) {
    if (p_any) {
    const bool valid = p_valid;
    const uint32_t row = p_row, ra = p_ra, rb = p_rb;
    // packed attr word of slot r: block r >> 7, word r & 63, byte pair (r >> 6) & 1
    uint32_t tfa, tfb, fa, fb;
    if (attr1 && !none_fast) {
      // nibble plane: one byte per doc (tf | fields << 4); tf 15 escapes to the attr word
      const uint32_t ba = p_wa, bb = p_wb;
      tfa = ba & 15u, tfb = bb & 15u;
      fa = (ba >> 4) & TA.queried32, fb = (bb >> 4) & TB.queried32; // FitsFields
      if (tfa == 15u) tfa = (attr[(uint64_t)(TA.blk_first + (ra >> 7)) * 64 + (ra & 63u)] >> (((ra >> 6) & 1u) * 8u)) & 0xffu;
      if (tfb == 15u) tfb = (attr[(uint64_t)(TB.blk_first + (rb >> 7)) * 64 + (rb & 63u)] >> (((rb >> 6) & 1u) * 8u)) & 0xffu;
    } else {
      const uint32_t wa = p_wa, wb = p_wb;
      const uint32_t sa = SEQ ? 0u : ((ra >> 6) & 1u) * 8u, sb = SEQ ? 0u : ((rb >> 6) & 1u) * 8u;
      tfa = (wa >> sa) & 0xffu, tfb = (wb >> sb) & 0xffu;
      fa = (wa >> ((SEQ ? 8u : 16u) + sa)) & 0xffu & TA.queried32, fb = (wb >> ((SEQ ? 8u : 16u) + sb)) & 0xffu & TB.queried32; // FitsFields
    }
    const bool live = valid && (none_fast || (fa != 0 && fb != 0));
    float ta = s.tfidf[0][tfa], tb = s.tfidf[1][tfb];
    if (tfa == 255u && live) ta = term_tfidf(exc_tf(a.seg, TA, row), TA.idf);
    if (tfb == 255u && live) tb = term_tfidf(exc_tf(a.seg, TB, row), TB.idf);
    float acc = 0.0f + ta; // ExtMultiAnd_T::GetTFIDF: nodes in ascending-docs order
    acc = acc + tb;
    bool push = false;
    uint64_t key = 0;
    if (live) {
      if (!none_fast) ++total; // (counted off the match words otherwise)
      uint32_t weight;
      if (ranker == MRK_RANK_NONE)
        weight = 1u; // ExtRanker_None_c, sphinxsearch.cpp:1160
      else {
        // ExtRanker_WeightSum_c<BM25>, sphinxsearch.cpp:1070, 1112-1129
        const int32_t bm = (int32_t)((acc + 0.5f) * 1000.0f);
        weight = (uint32_t)bm + s.rank[fa | fb] * 1000u;
      }
      weight *= index_weight; // MatchExtended, sphinx.cpp:12220
      const uint32_t grow = a.seg.rowid_base + row;
      if (bin_of(bin_mode, bin_lo, bin_shift, (int32_t)weight, grow) >= tau_bin) {
        push = true;
        key = make_key((int32_t)weight, grow);
      }
    }
    const uint64_t bal = __ballot(push);
    if (bal) {
      const uint32_t np = (uint32_t)__popcll(bal);
      if (cn + np > (uint32_t)BM_CBUF) publish();
      if (push) L.cbuf[cn + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      cn += np;
      // (a wave that knows no threshold yet publishes its first 16 candidates at once: the sooner the query's histogram holds
      // K matches the sooner everybody prunes -- a launch whose waves all start together otherwise floods the list with ~100
      // candidates per wave before the first threshold exists)
      if (cn >= (tau_bin ? (uint32_t)BM_CBUF - 64u : 16u)) publish();
    }
    } // (the pending round)
    // ... and this round's requests
    p_any = true, p_valid = valid_new, p_row = row_new, p_ra = ra_new, p_rb = rb_new;
    if (attr1 && !none_fast)
      p_wa = attr1[(uint64_t)TA.blk_first * 128 + ra_new], p_wb = attr1[(uint64_t)TB.blk_first * 128 + rb_new];
    else {
#if MRK_BMEXP == 4 // ablation: scoring without the two gathers
      p_wa = 0x01010101u + (ra_new & 3u), p_wb = 0x01010101u + (rb_new & 1u);
#else
      if (SEQ) { // slot order: the matches of a line arrive together
        p_wa = none_fast ? 0x0101u : attr2[(uint64_t)TA.blk_first * 128 + ra_new];
        p_wb = none_fast ? 0x0101u : attr2[(uint64_t)TB.blk_first * 128 + rb_new];
      } else {
        p_wa = none_fast ? 0x01010101u : attr[(uint64_t)(TA.blk_first + (ra_new >> 7)) * 64 + (ra_new & 63u)];
        p_wb = none_fast ? 0x01010101u : attr[(uint64_t)(TB.blk_first + (rb_new >> 7)) * 64 + (rb_new & 63u)];
      }
#endif
    }
  };

  // Unpack the word queue entries [from, from + n), n <= 64, one word per lane: each lane's lowest match bit is a match to
  // score -- all n lanes have one -- and the words that hold more bits go back on the queue, at `from`, for a later round.
  // (A window holds a dozen matches in its 64 words: unpacking it on the spot would run the bit loop, and the scoring, with a
  // fifth of the lanes busy.)
  uint32_t wqn = 0;
  auto unpack = [&](uint32_t from, uint32_t n) -> uint32_t {
    wave_lds_fence();
    const bool valid = lane < n;
    const uint32_t e = from + (valid ? lane : 0u);
    const uint4 we = L.wq[e];
    const uint2 wr = L.wqr[e];
    const uint32_t m = valid ? we.x : 0u, aw = we.y, bw = we.z, rowbase = we.w;
    const uint32_t bit = valid ? (uint32_t)__builtin_ctz(m) : 0u; // (m != 0 for every queued word)
    const uint32_t below = (1u << bit) - 1u;
    const uint32_t rest = m & (m - 1u);
    const uint64_t bal = __ballot(rest != 0);
    wave_lds_fence(); // every lane holds its entry: the slots may be rewritten
    if (rest) {
      const uint32_t pos = from + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      L.wq[pos] = make_uint4(rest, aw, bw, rowbase);
      L.wqr[pos] = wr;
    }
#if MRK_BMEXP != 2
    score(valid, rowbase + bit, wr.x + (uint32_t)__popc(aw & below), wr.y + (uint32_t)__popc(bw & below));
#endif
    return (uint32_t)__popcll(bal);
  };

  for (uint32_t wb = w0; wb < w1; wb += BM_BURST) {
    // BM_BURST windows requested back to back (one memory round trip per burst)
    const uint32_t nb = w1 - wb < (uint32_t)BM_BURST ? w1 - wb : (uint32_t)BM_BURST;
    uint32_t av[BM_BURST], bv[BM_BURST], dv[BM_BURST];
#pragma unroll
    for (int i = 0; i < BM_BURST; ++i) {
      av[i] = bv[i] = dv[i] = 0;
      if ((uint32_t)i < nb) {
        const uint64_t o = (uint64_t)(wb + i) * BM_WORDS + lane;
        av[i] = bmA[o];
        bv[i] = bmB[o];
        if (dead) dv[i] = dead[o];
      }
    }
#if MRK_BMEXP != 1
    // the query's shared pruning threshold rides along with every 4th burst (every wave of the query reads the one word: with
    // every burst the reads cost 0.18 of the launch's 2.76 ms, without any the candidate lists grow by a quarter)
    if (((wb - w0) / BM_BURST & 3u) == 0) {
      const uint32_t gt = __hip_atomic_load(gtaubin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gt > tau_bin) tau_bin = gt;
    }
#endif
#pragma unroll
    for (int i = 0; i < BM_BURST; ++i) {
      if ((uint32_t)i < nb) {
        const uint32_t aw = av[i], bw = bv[i];
        uint32_t m = aw & bw & ~dv[i];
        if (none_fast) {
          total += (uint32_t)__popc(m);
          // the window's lowest rowid has its best bin: behind the threshold, nothing in it can enter the top K
          if (bin_of(bin_mode, bin_lo, bin_shift, 1, a.seg.rowid_base + (wb + i) * 2048u) < tau_bin) m = 0;
        }
        // ranks of the lane's first bit: one prefix sum carries both keywords' popcounts
        const uint32_t pc = (uint32_t)__popc(aw) | ((uint32_t)__popc(bw) << 16);
        const uint32_t incl = wave_incl_scan(pc);
        const uint32_t excl = incl - pc;
        const uint32_t ra0 = baseA + (excl & 0xFFFFu), rb0 = baseB + (excl >> 16);
        const uint32_t tot = rdlane(incl, 63);
        baseA += tot & 0xFFFFu;
        baseB += tot >> 16;
        const uint32_t rowbase = (wb + i) * 2048u + lane * 32u;
#if MRK_BMEXP == 3
        total += __popc(m);
        m = 0;
#endif
        // A window holds a dozen matches in its 64 words: unpacking it on the spot would run the bit loop with a fifth of
        // the lanes busy.  Its non-empty words are queued instead (one ballot, no loop) and unpacked 64 words at a time.
        const bool has = m != 0;
        const uint64_t bal = __ballot(has);
        if (bal) {
          if (has) {
            const uint32_t pos = wqn + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            L.wq[pos] = make_uint4(m, aw, bw, rowbase);
            L.wqr[pos] = make_uint2(ra0, rb0);
          }
          wqn += (uint32_t)__popcll(bal);
          while (wqn >= 64u) wqn = wqn - 64u + unpack(wqn - 64u, 64u);
        }
      }
    }
  }
  // Final drain.  One bit per word per round (unpack) would take as many rounds as the fullest word has bits -- six to eight
  // gather round trips with a handful of lanes busy at the end of EVERY wave, which is what a wave of a one-eighth shard (64
  // windows) mostly consisted of.  Here the queue's matches are numbered through (prefix sum of the words' popcounts) and
  // lane l of round r takes match 64 r + l: its word by binary search, its bit by rank.
  if (wqn) {
    wave_lds_fence();
    const uint32_t e0 = lane, e1 = lane + 64u;
    const uint32_t c0 = e0 < wqn ? (uint32_t)__popc(L.wq[e0].x) : 0u, c1 = e1 < wqn ? (uint32_t)__popc(L.wq[e1].x) : 0u;
    const uint32_t i0 = wave_incl_scan(c0), t0 = rdlane(i0, 63);
    const uint32_t i1 = wave_incl_scan(c1) + t0, nmatch = rdlane(i1, 63);
    L.wpre[e0] = (uint16_t)(i0 - c0);
    L.wpre[e1] = (uint16_t)(i1 - c1);
    wave_lds_fence();
    for (uint32_t r0 = 0; r0 < nmatch; r0 += 64u) {
      const uint32_t j = r0 + lane;
      const bool valid = j < nmatch;
      uint32_t lo = 0, hi = wqn; // entry = last e with wpre[e] <= j
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (L.wpre[mid] <= j)
          lo = mid;
        else
          hi = mid;
      }
      const uint4 we = L.wq[lo];
      const uint2 wr = L.wqr[lo];
      uint32_t k = valid ? j - L.wpre[lo] : 0u, m = we.x, bit = 0; // the k-th set bit of m (k < popc(m))
      uint32_t c = (uint32_t)__popc(m & 0xFFFFu);
      if (k >= c) k -= c, bit += 16u, m >>= 16;
      c = (uint32_t)__popc(m & 0xFFu);
      if (k >= c) k -= c, bit += 8u, m >>= 8;
      c = (uint32_t)__popc(m & 0xFu);
      if (k >= c) k -= c, bit += 4u, m >>= 4;
      c = (uint32_t)__popc(m & 0x3u);
      if (k >= c) k -= c, bit += 2u, m >>= 2;
      if (k >= (m & 1u)) bit += 1u;
      const uint32_t below = (1u << bit) - 1u;
#if MRK_BMEXP != 2
      score(valid, we.w + bit, wr.x + (uint32_t)__popc(we.y & below), wr.y + (uint32_t)__popc(we.z & below));
#endif
    }
    wqn = 0;
  }
#if MRK_BMEXP != 2
  if (p_any) score(false, 0u, 0u, 0u); // finishes the last round (its own, empty one stays unfinished)
#endif
  if (cn) publish();
  {
    uint32_t t = total;
    for (int dlt = 32; dlt; dlt >>= 1) t += __shfl_down(t, dlt, 64);
    if (lane == 0 && t) atomicAdd((unsigned long long*)(a.q_total + oq), (unsigned long long)t);
  }
}

void launch_scan_bm(const ScanArgs& a, void* stream) {
  if (!a.n_items) return;
  if (a.seg.pk_attr2 && !a.seg.pk_attr1)
    hipLaunchKernelGGL(scan_bm_kernel<true>, dim3(a.n_items), dim3(WG), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(scan_bm_kernel<false>, dim3(a.n_items), dim3(WG), 0, (hipStream_t)stream, a);
}

} // namespace mrk
