// mrk_scan_bt.hip -- boolean trees evaluated over doc-set BITMAP WORDS, gfx950 / wave64.
//
// The block-scan kernel walks the docs of a tree's candidate cover one by one.  When that cover is a common keyword --
// '(a | b) c' with b and c in millions of docs -- it decodes tens of thousands of driver blocks per query and probes
// the other keywords once per doc.  Here the same ExtAnd_c / ExtOr_c / ExtMaybe_c / ExtAndNot_c tree
// (searchnode.cpp:2570-2706, 3465-3694) is evaluated 32 rowids at a time:
//
//   window = 2048 rowids, lane l holds rowids [32 l, 32 l + 32) of it as one word per keyword:
//     dense keyword   word l of its doc-set bitmap (one coalesced 256-B load per window, as in scan_bm_kernel);
//     sparse keyword  assembled on the fly: a cursor walks the keyword's packed blocks (2 docs per lane in registers),
//                     docs that fall into the window set their bit through an LDS word per lane;
//   the tree program runs on those words (AND = &, OR = |, ANDNOT = & ~, MAYBE = its left side) and leaves the window's
//   matches as bits.  A match's slot in each present keyword's packed arrays is its RANK in that keyword's doc list:
//   docs before the window (rank directory / cursor, then a running count) + popcounts of the lower lanes' words (one
//   packed wave prefix sum per two keywords) + popcount of the lower bits of the lane's own word.
//
// Matches are queued per wave in LDS (rowid, present keywords, one rank per keyword) and scored 64 at a time with
// every lane busy: tf / field bytes gathered by rank, the tree's VALUE rules (tfidf sums in the reference's fp32
// order, field masks, which keywords' hits a match emits) on a per-lane stack exactly as in scan_pk_kernel, then either
// the weight-sum rankers + pruning histogram right here, or -- hit-reading rankers -- a 64-doc chunk written straight
// from registers into the HBM match queue for rank_kernel.
//
// Only queries whose keywords are unrestricted in fields come here (a bitmap bit then IS "the keyword holds the doc",
// searchnode.cpp:1925-1939); the planner keeps everything else on the block path.
#include "mrk_kcommon.h"
#include "mrk_kprune.h"
#include "mrk_kpk.h"
#include "mrk_kmq.h"

#ifndef MRK_BT_WAVES
#define MRK_BT_WAVES 4 // waves per SIMD the register allocation is made for
#endif
#ifndef MRK_BTEXP
#define MRK_BTEXP 0 // ablations for profiling: 1 no match-queue write, 2 no scoring, 3 no match extraction, 4 no sparse keywords, 6 no lower-bound histogram, 7 bounds but no queue write, 8 no second level of the lower-bound histogram
#endif

namespace mrk {

constexpr int BT_KW = MAX_PROX_TERMS; // keywords per query on this path
constexpr int BT_CBUF = 128;          // candidates a wave collects before it publishes them
constexpr int BT_QCAP = 128;          // match queue entries per wave (scored in batches of 64)
constexpr int BT_WORDS = 64;          // words per window
constexpr uint32_t BT_NOTES = 128;    // match notes a wave holds between the bit walk and the queue (uint16 each, in the abm words)
constexpr int BT_SPAN = 4;            // windows per step: lane l holds 128 consecutive rowids = four bitmap words per keyword

struct __align__(16) BtWaveLds {
  union { // a pass either weighs its matches itself (candidate buffer) or hands them to the hit pass (pending queue chunk)
    uint64_t cbuf[BT_CBUF];
    uint32_t pend[MQ_PLANES * 64]; // survivors of the pruning in front of the hit pass, plane-major, until 64 make a chunk
  };
  uint32_t q_row[BT_QCAP];
  uint32_t q_pm[BT_QCAP]; // keywords present in the doc
  uint32_t q_rank[BT_KW][BT_QCAP];
  uint32_t abm[BT_SPAN * BT_WORDS]; // the step's words of ONE sparse keyword while they are assembled
};

struct __align__(16) BtSmem {
  BtWaveLds w[WAVES];
  uint32_t hist[NBINS]; // publishing scratch, one per workgroup behind hist_lock
  uint32_t hist_lock;
  uint32_t rank[256];
  float tfidf[BT_KW][256];
  int32_t fw[8]; // the field weights (prox_bounds)
};

// the two docs lane l owns of block kj of a keyword (rowids, INF past the block's end) and the first possible rowid of the
// NEXT block (= this block's last rowid + 1; INF after the last block)
__device__ __forceinline__ void bt_load_block(const DevSegment& seg, const DevTerm& T, uint32_t kj, uint32_t& e0, uint32_t& e1, uint32_t& bnext) {
  const uint32_t lane = lane_id();
  const uint32_t g = T.blk_first + kj;
  const uint32_t bp1 = seg.pk_base[g], w = seg.pk_w[g];
  const uint32_t* __restrict__ dp = seg.pk_delta + seg.pk_doff[g];
  PkRaw raw;
  raw.attr = 0;
  if (w == PK_WIDE) {
    raw.lo = dp[lane];
    raw.hi = dp[64 + lane];
  } else {
    const uint32_t wi = (lane * 2 * w) >> 5;
    raw.lo = dp[wi];
    raw.hi = dp[wi + 1];
  }
  bnext = kj + 1 < T.nblocks ? seg.pk_base[g + 1] : INF_ROWID;
  const uint32_t left = T.docs - kj * DEVBLK;
  uint32_t r0, r1, o0, o1;
  bool ok0, ok1;
  decode_pk(raw, w, bp1, left < (uint32_t)DEVBLK ? left : (uint32_t)DEVBLK, r0, r1, o0, o1, ok0, ok1);
  e0 = ok0 ? r0 : INF_ROWID;
  e1 = ok1 ? r1 : INF_ROWID;
}

// PRUNE: the instance that bounds hit-ranked matches' weights and keeps the hopeless ones out of the match queue (launched when the
// batch has the lower-bound histograms: ScanArgs::q_hist_lb)
template <bool PRUNE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(MRK_BT_WAVES))) void scan_bt_kernel(ScanArgs a) {
  __shared__ BtSmem s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= a.n_items) return;
  const DevItem item = a.items[blockIdx.x];
  const DevQuery* __restrict__ Q = a.queries + item.query;
  if (__hip_atomic_load(a.q_flags + Q->out_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & QF_OVERFLOW) return; // (rerun alone by the host anyway: see scan_pk_kernel)
  const uint32_t nterms = Q->n_terms < (uint32_t)BT_KW ? Q->n_terms : (uint32_t)BT_KW;
  const uint32_t K = Q->k, ranker = Q->ranker, oq = Q->out_q, n_nodes = Q->n_nodes;
  const uint32_t nw = Q->n_weights < 8u ? Q->n_weights : 8u;
  const uint32_t index_weight = Q->index_weight;
  const bool inline_hits = a.seg.inline_hits != 0;
  // a root PHRASE / PROXIMITY: the docs that hold all its words are CANDIDATES, the word state machine of the hit pass decides
  // (queue 1, rank_kernel<1>; the planner sends such a query here when its words are common: mrk_plan.cpp)
  const bool fat_q = (Q->tree_flags & TF_FAT) != 0;
  const bool need_hits = fat_q || ((ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY)
                                       ? nterms > 1
                                       : (ranker == MRK_RANK_WORDCOUNT || ranker == MRK_RANK_MATCHANY || ranker == MRK_RANK_FIELDMASK ||
                                          ranker == MRK_RANK_SPH04));
  BtWaveLds& L = s.w[wave];
  for (uint32_t j = 0; j < nterms; ++j) s.tfidf[j][tid] = term_tfidf(tid, Q->t[j].idf);
  {
    uint32_t rk = 0;
    if (!tid)
      rk = 1; // empty mask: "just fake it" (sphinxsearch.cpp:1114-1118)
    else
      for (uint32_t f = 0; f < nw; ++f)
        if (tid & (1u << f)) rk += (uint32_t)Q->weights[f];
    s.rank[tid] = rk;
    if (!tid) s.hist_lock = 0;
    if (tid < 8) s.fw[tid] = tid < nw ? Q->weights[tid] : 0;
  }
  // prune in front of the hit pass (mrk_kprune.h, prox_bounds): proximity rankers over distinct keywords, nothing that needs the exact weight of every match
  const bool ph_lone = fat_q && (Q->tree_flags & TF_FAT) == TF_PHRASE && Q->px_dist == 0 && inline_hits && nterms >= 2; // (see score())
  // the bounds are 32-bit like the weights: a run of up to 4 x 255 hits times the field weights, x 1000, x index_weight must fit (absurd
  // weights: no pruning -- the weights themselves wrap there as the reference's int does)
  uint64_t wabs = 0;
  for (uint32_t f = 0; f < nw; ++f) wabs += (uint64_t)(Q->weights[f] < 0 ? -(int64_t)Q->weights[f] : (int64_t)Q->weights[f]);
  const bool bounds_fit = (wabs * 1020ull * 1000ull + 1000ull) * (uint64_t)index_weight < (1ull << 32);
  const bool prune_prox = PRUNE && need_hits && !fat_q && bounds_fit && a.q_hist_lb && (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY) && !(Q->tree_flags & TF_DUPES) &&
                          Q->n_wfilters == 0 && Q->bin_mode == BIN_WEIGHT;
  uint32_t* __restrict__ ghist_lb = prune_prox ? a.q_hist_lb + (uint64_t)oq * NBINS : nullptr;
  uint32_t* __restrict__ ghist_lb2 = prune_prox ? a.q_hist_lb2 + (uint64_t)oq * NBINS : nullptr;
  uint32_t* __restrict__ gtau_lb = prune_prox ? a.q_tau_lb + (size_t)oq * QSTRIDE : nullptr; // (threshold bin << 10) | threshold slot inside it
  const uint32_t bin_mode = Q->bin_mode, bin_shift = Q->bin_shift;
  const int32_t bin_lo = Q->bin_lo;
  const uint32_t cand_cap = Q->cand_cap;
  uint64_t* __restrict__ cand = a.cand + Q->cand_off;
  uint32_t* __restrict__ ghist = a.q_hist + (uint64_t)oq * NBINS;
  uint32_t* __restrict__ gcount = a.q_cand_n + (size_t)oq * QSTRIDE;
  uint32_t* __restrict__ gtaubin = a.q_tau_bin + (size_t)oq * QSTRIDE;
  const uint32_t* __restrict__ dead = a.seg.dead;
  const uint32_t* __restrict__ attr = a.seg.pk_attr;

  // the tree program, once (uniform values)
  uint32_t prog[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) prog[i] = (uint32_t)i < n_nodes ? Q->prog[i] : 0u;

  const uint32_t nwin = item.blk_end - item.blk_begin;
  const uint32_t per = (nwin + WAVES - 1) / WAVES;
  const uint32_t w0 = item.blk_begin + wave * per;
  const uint32_t w1 = w0 + per < item.blk_end ? w0 + per : item.blk_end;
  __syncthreads(); // tables ready; the waves never meet again

  // per keyword: dense (bitmap) or sparse (block cursor); docs before the next window (uniform)
  bool dense[BT_KW];
  uint32_t base[BT_KW];
  uint32_t kj[BT_KW], e0[BT_KW], e1[BT_KW], bnext[BT_KW]; // sparse cursors: block in registers
  const uint32_t lo_first = w0 * 2048u;
#pragma unroll
  for (int k = 0; k < BT_KW; ++k) {
    dense[k] = false, base[k] = 0, kj[k] = 0, e0[k] = e1[k] = INF_ROWID, bnext[k] = INF_ROWID;
    if ((uint32_t)k < nterms && w0 < w1) {
      const DevTerm& T = Q->t[k];
      dense[k] = T.nblocks != 0 && T.bm_off != ~0ull; // (a keyword without postings has an all-zero descriptor)
      if (dense[k])
        base[k] = a.seg.bm_dir[T.dir_off + (uint64_t)w0 * (BT_WORDS / 8)];
      else if (T.nblocks) {
        // the block that holds the first doc >= lo_first: the last one whose first possible rowid is <= lo_first
        kj[k] = wave_find_block(a.seg.pk_base + T.blk_first, 0, T.nblocks, lo_first);
        bt_load_block(a.seg, T, kj[k], e0[k], e1[k], bnext[k]);
        base[k] = kj[k] * DEVBLK + (uint32_t)__popcll(__ballot(e0[k] < lo_first)) + (uint32_t)__popcll(__ballot(e1[k] < lo_first));
      } else
        kj[k] = 0xFFFFFFFFu; // keyword without postings
    }
  }

  uint32_t total = 0, tau_bin = 0, cn = 0, qn = 0;
  MqWriter mqw;
  // pruned hit-ranked matches: the survivors of a scoring round are compacted into the wave's pending chunk (LDS) until 64 are there
  uint32_t tau_lb = 0, tau2 = 0, lb_rounds = 0, lb_added = 0, pend_n = 0;
  // second level of the lower-bound histogram: 10 bits = the weight's offset inside its bin (bin_shift bits) + rowid slices
#if MRK_BTEXP == 8 // (ablation: first level only)
  const uint32_t l2_rbits = 0u;
#else
  const uint32_t l2_rbits = bin_shift < 10u ? 10u - bin_shift : 0u;
#endif
  const uint32_t l2_rmax = a.seg.n_windows * 2048u - 1u; // >= every row of the segment
  const uint32_t l2_rshift = (32u - (uint32_t)__builtin_clz(l2_rmax | 1u)) > l2_rbits ? (32u - (uint32_t)__builtin_clz(l2_rmax | 1u)) - l2_rbits : 0u;
  auto write_chunk = [&](const uint32_t* v, uint32_t n) { // one chunk of the HBM match queue from registers: lane l = entry l, n entries
    const MatchQueue& MQ = a.mq[fat_q ? 1 : 0];
    const uint32_t c = mq_take(MQ, mqw);
    if (c != 0xFFFFFFFFu) {
      uint32_t* __restrict__ d = MQ.data + (uint64_t)c * (MQ_PLANES * 64) + lane;
#pragma unroll
      for (int i = 0; i < MQ_PLANES; ++i) d[64 * i] = v[i];
      if (lane == 0) MQ.hdr[c] = item.query | (n << 24);
    } else if (lane == 0)
      atomicOr(a.q_flags + oq, QF_OVERFLOW);
  };

  auto publish = [&]() {
    if (cn) {
      uint32_t basep = 0;
      if (lane == 0) basep = atomicAdd(gcount, cn);
      basep = rdlane(basep, 0);
      const bool fits = basep + cn <= cand_cap;
      const uint32_t npub = cn;
      if (lane == 0) {
        uint32_t expected = 0;
        while (!__hip_atomic_compare_exchange_strong(&s.hist_lock, &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          expected = 0;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      wave_lds_fence();
      for (uint32_t i = lane; i < (uint32_t)NBINS; i += 64) s.hist[i] = 0;
      wave_lds_fence();
      for (uint32_t i = lane; i < cn; i += 64) {
        const uint64_t key = L.cbuf[i];
        if (fits) cand[basep + i] = key;
        atomicAdd(&s.hist[bin_of(bin_mode, bin_lo, bin_shift, key_weight(key), key_rowid(key))], 1u);
      }
      if (!fits && lane == 0) atomicOr(a.q_flags + oq, QF_OVERFLOW);
      wave_lds_fence();
      flush_hist(s.hist, ghist);
      wave_lds_fence();
      if (lane == 0) __hip_atomic_store(&s.hist_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      cn = 0;
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

  // score the queue entries [from, from + n), n <= 64, one per lane
  auto score = [&](uint32_t from, uint32_t n) {
    wave_lds_fence();
    const bool valid = lane < n;
    const uint32_t e = from + (valid ? lane : 0u);
    const uint32_t row = L.q_row[e], pm = valid ? L.q_pm[e] : 0u;
    float kv[BT_KW];
    uint32_t kf[BT_KW], ktf[BT_KW], href[BT_KW];
#pragma unroll
    for (int k = 0; k < BT_KW; ++k) {
      kv[k] = 0.0f, kf[k] = 0, ktf[k] = 0, href[k] = 0;
      if ((uint32_t)k < nterms) {
        const DevTerm& T = Q->t[k];
        const bool pres = ((pm >> k) & 1u) != 0;
        const uint32_t r = pres ? L.q_rank[k][e] : 0u;
        // packed attr word of slot r: block r >> 7, word r & 63, byte pair (r >> 6) & 1
        const uint32_t wd = attr[(uint64_t)(T.blk_first + (r >> 7)) * 64 + (r & 63u)];
        const uint32_t sh = ((r >> 6) & 1u) * 8u;
        const uint32_t tf = (wd >> sh) & 0xffu;
        if (pres) {
          kf[k] = (wd >> (16u + sh)) & 0xffu & T.queried32; // (all fields queried: the doc's own field bits)
          kv[k] = tf == 255u ? term_tfidf(exc_tf(a.seg, T, row), T.idf) : s.tfidf[k][tf];
          ktf[k] = tf; // (255 = "255 or more": a run is counted in a byte, RankState::lcs, so 255 already bounds it)
          href[k] = ((inline_hits && tf == 1u) ? 0x80000000u : 0u) | r;
        }
      }
    }
    // the tree's value rules on a per-lane stack (ExtAnd_c / ExtOr_c / ExtMaybe_c / ExtAndNot_c, searchnode.cpp:2585-2594,
    // 3494-3540, 3587-3600, 3650-3680): tfidf adds left + right where both sides hold the doc, fields OR together
    bool m0 = false, m1 = false, m2 = false, m3 = false;
    float v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    uint32_t f0 = 0, f1 = 0, f2 = 0, f3 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if ((uint32_t)i < n_nodes) {
        const uint32_t ins = prog[i], op = ins & 0xffu, kw = ins >> 24;
        if (op == PN_TERM) {
          m3 = m2, v3 = v2, f3 = f2, a3 = a2;
          m2 = m1, v2 = v1, f2 = f1, a2 = a1;
          m1 = m0, v1 = v0, f1 = f0, a1 = a0;
          const bool m = ((pm >> kw) & 1u) != 0;
          const float v = kw == 0 ? kv[0] : kw == 1 ? kv[1] : kw == 2 ? kv[2] : kv[3];
          const uint32_t f = kw == 0 ? kf[0] : kw == 1 ? kf[1] : kw == 2 ? kf[2] : kf[3];
          m0 = m, v0 = m ? v : 0.0f, f0 = m ? f : 0u, a0 = m ? 1u << kw : 0u;
        } else {
          bool m;
          if (op == PN_AND)
            m = m1 && m0;
          else if (op == PN_OR)
            m = m1 || m0;
          else if (op == PN_MAYBE)
            m = m1;
          else
            m = m1 && !m0;
          const bool both = op != PN_ANDNOT; // ANDNOT passes its left side through
          const float v = m ? (both ? v1 + v0 : v1) : 0.0f; // x + 0.0f == x: an absent side adds nothing
          const uint32_t f = m ? (both ? f1 | f0 : f1) : 0u, av = m ? (both ? a1 | a0 : a1) : 0u;
          m0 = m, v0 = v, f0 = f, a0 = av;
          m1 = m2, v1 = v2, f1 = f2, a1 = a2;
          m2 = m3, v2 = v3, f2 = f3, a2 = a3;
        }
      }
    }
    const bool live = valid && m0; // (every queued doc matched bitwise: m0 holds for all valid lanes)
    if (need_hits) {
      // a 64-doc chunk of the HBM match queue straight from registers (rank_kernel does the hit pass and the ranking)
#if MRK_BTEXP == 1
      total += valid ? 1u : 0u;
      return;
#endif
      uint32_t vals[MQ_PLANES];
      vals[0] = row, vals[1] = __float_as_uint(v0), vals[2] = (f0 & 0xffu) | ((fat_q ? 0xffu : a0) << 8); // (a phrase's words all emit: scan_pk_kernel's non-tree instance says 0xff)
#pragma unroll
      for (int t = 0; t < MAX_PROX_TERMS; ++t) vals[3 + t] = href[t];
      bool keep = valid;
      if (ph_lone) {
        // A plain PHRASE over candidates whose words all have ONE hit each (it sits in the doclist entry: pk_hit): the phrase
        // occurs iff the hits share the field and lie exactly their query positions apart (FSMphrase_c, searchnode.cpp:3884-3953,
        // with one hit per word) -- a candidate that fails is no match and never travels to the hit pass.  Candidates with a
        // longer hit list, and the ones that pass (their weight is the hit pass's business), go on as before.
        bool lone = valid;
#pragma unroll
        for (int k = 0; k < BT_KW; ++k)
          if ((uint32_t)k < nterms) lone = lone && (href[k] >> 31) != 0;
        uint32_t d0 = 0;
        bool same = true;
#pragma unroll
        for (int k = 0; k < BT_KW; ++k) {
          if ((uint32_t)k < nterms) {
            const DevTerm& T = Q->t[k];
            const uint32_t r = href[k] & 0x7FFFFFFFu;
            const uint32_t hv = lone ? a.seg.pk_hit[(uint64_t)(T.blk_first + (r >> 7)) * DEVBLK + (r & 127u)] : 0u;
            const uint32_t d = (hv & ~(1u << 23)) - T.qpos; // (field << 24 | position) - query position: equal <=> same field, in step
            if (k == 0)
              d0 = d;
            else
              same = same && d == d0;
          }
        }
        keep = valid && !(lone && !same);
      }
      if (!prune_prox && !ph_lone) {
        write_chunk(vals, n);
        return;
      }
      if (prune_prox) {
      // bounds of the weight from what the doclists say; the lower bounds feed the query's histograms, the upper bound is tested
      uint32_t wlo, whi;
      prox_bounds(ranker, v0, a0, kf, ktf, BT_KW, s.fw, nw, index_weight, (Q->tree_flags & TF_LCS_BY_KEYWORDS) != 0, wlo, whi);
      // (only lower bounds that reach the current threshold can raise it: the others are never counted -- an undercounted
      // histogram only makes the threshold lower than it could be -- and once the threshold stands almost no round adds anything)
      const uint32_t blo = bin_of(BIN_WEIGHT, bin_lo, bin_shift, (int32_t)wlo, 0u);
      const uint32_t bhi = bin_of(BIN_WEIGHT, bin_lo, bin_shift, (int32_t)whi, 0u);
      // second level (mrk_kprune.h): inside the threshold bin, (weight offset in the bin, rowid slice) -- a bin in the middle of the
      // range holds exactly the weights [bin_lo + (T << shift), + 2^shift), so the offset orders them exactly
      const bool lvl2 = tau_lb > 0 && tau_lb < (uint32_t)NBINS - 1u && l2_rbits > 0;
      const uint32_t rsl = (l2_rmax - row) >> l2_rshift; // (row <= l2_rmax; lower rowids in higher slices)
      const uint32_t off_mask = (1u << bin_shift) - 1u;
      const uint32_t slo = (((wlo - (uint32_t)bin_lo) & off_mask) << l2_rbits) | rsl;
      const uint32_t shi = (((whi - (uint32_t)bin_lo) & off_mask) << l2_rbits) | rsl;
#if MRK_BTEXP == 6
      const bool counts = false, counts2 = false;
#else
      const bool counts = valid && blo >= tau_lb;
      const bool counts2 = valid && lvl2 && blo == tau_lb && slo >= tau2;
#endif
      lb_added += (uint32_t)__popcll(__ballot(counts)) + (uint32_t)__popcll(__ballot(counts2));
      hist_add_bins(ghist_lb, counts, blo);
      if (lvl2) hist2_add(ghist_lb2, counts2, slo, tau_lb);
      if (lb_added >= 256u) { // the wave added enough to matter: recompute
        uint32_t n_above = 0;
        const uint32_t tb = threshold_bin_above(ghist_lb, K, n_above);
        bool moved = false;
        if (tb > tau_lb) tau_lb = tb, tau2 = 0, moved = true;
        if (tb == tau_lb && tau_lb > 0 && tau_lb < (uint32_t)NBINS - 1u && l2_rbits > 0 && n_above < K) {
          const uint32_t t2 = threshold_slot(ghist_lb2, K - n_above, tau_lb);
          if (t2 > tau2) tau2 = t2, moved = true;
        }
        if (moved && lane == 0) atomicMax(gtau_lb, (tau_lb << 10) | tau2);
        lb_added = 0;
      }
      if ((lb_rounds++ & 3u) == 0u) {
        const uint32_t gt = __hip_atomic_load(gtau_lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gt > ((tau_lb << 10) | tau2)) tau_lb = gt >> 10, tau2 = gt & 1023u;
      }
      keep = valid && (bhi > tau_lb || (bhi == tau_lb && (!(tau_lb > 0 && tau_lb < (uint32_t)NBINS - 1u && l2_rbits > 0) || shi >= tau2)));
      total += (valid && !keep) ? 1u : 0u; // a match all the same (CSphMatchQueue::PushT counts every push): rank_kernel counts the ones it sees
#ifdef MRK_BT_DEBUG_ROW
      if (valid && row == MRK_BT_DEBUG_ROW)
        printf("bt debug: q %u row %u keep %d wlo %u whi %u blo %u bhi %u T %u tau2 %u slo %u shi %u shift %u lo %d rbits %u rshift %u acc %f emit %x kf %x %x %x %x pm %x K %u\n", oq, row, (int)keep, wlo, whi, blo, bhi,
               tau_lb, tau2, slo, shi, bin_shift, bin_lo, l2_rbits, l2_rshift, v0, a0, kf[0], kf[1], kf[2], kf[3], pm, K);
#endif
      }
#if MRK_BTEXP == 7
      total += keep ? 1u : 0u;
      return;
#endif
      const uint64_t km = __ballot(keep);
      const uint32_t kcnt = (uint32_t)__popcll(km);
      if (!kcnt) return;
      // survivors -> slots [pend_n, pend_n + kcnt) of the pending chunk; a full chunk leaves for the queue
      const uint32_t slot = pend_n + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
      if (keep && slot < 64u) {
#pragma unroll
        for (int i = 0; i < MQ_PLANES; ++i) L.pend[i * 64 + slot] = vals[i];
      }
      if (pend_n + kcnt >= 64u) {
        wave_lds_fence();
        uint32_t full[MQ_PLANES];
#pragma unroll
        for (int i = 0; i < MQ_PLANES; ++i) full[i] = L.pend[i * 64 + lane];
        write_chunk(full, 64u);
        wave_lds_fence();
        if (keep && slot >= 64u) {
#pragma unroll
          for (int i = 0; i < MQ_PLANES; ++i) L.pend[i * 64 + slot - 64u] = vals[i];
        }
        pend_n = pend_n + kcnt - 64u;
      } else
        pend_n += kcnt;
      return;
    }
    bool push = false;
    uint64_t key = 0;
    if (live) {
      ++total;
      uint32_t weight;
      if (ranker == MRK_RANK_NONE)
        weight = 1u; // ExtRanker_None_c, sphinxsearch.cpp:1160
      else if (ranker == MRK_RANK_PROXIMITY)
        weight = s.rank[f0 & 0xffu]; // single keyword: ExtRanker_WeightSum_c<> without BM25
      else {
        // ExtRanker_WeightSum_c<BM25>, sphinxsearch.cpp:1070, 1112-1129
        const int32_t bm = (int32_t)((v0 + 0.5f) * 1000.0f);
        weight = (uint32_t)bm + s.rank[f0 & 0xffu] * 1000u;
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
      if (cn + np > (uint32_t)BT_CBUF) publish();
      if (push) L.cbuf[cn + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      cn += np;
      if (cn >= (uint32_t)BT_CBUF - 64u) publish();
    }
  };

  // the dense keywords' words (and the dead-row words) of one step: lane l holds rowids [128 l, 128 l + 128) of it, FOUR consecutive
  // bitmap words per keyword in one 16-byte load
  auto load_step = [&](uint32_t wb, uint4 (&kw)[BT_KW], uint4& dw) {
    const uint32_t wl = wb + (lane >> 4);   // the window the lane's words belong to
    const bool inr = wl < w1;               // (the wave's range need not be a multiple of BT_SPAN windows)
    const uint64_t woff = (uint64_t)(inr ? wl : w0) * BT_WORDS + (lane & 15u) * 4u;
#pragma unroll
    for (int k = 0; k < BT_KW; ++k) {
      kw[k] = make_uint4(0, 0, 0, 0);
      if ((uint32_t)k < nterms && dense[k]) {
        const uint4 v = *(const uint4*)(a.seg.bm + Q->t[k].bm_off + woff);
        if (inr) kw[k] = v;
      }
    }
    dw = make_uint4(0, 0, 0, 0);
    if (dead) dw = *(const uint4*)(dead + woff);
  };
  for (uint32_t wb = w0; wb < w1; wb += BT_SPAN) {
    // one step = BT_SPAN windows = 8192 rowids (a step used to be one window, one word per lane: the tree program, the prefix sums
    // and the loop around them cost the same per step whatever the lane holds, and at 14 matches per window they, not the matches,
    // were the kernel).  Requesting the NEXT step's words ahead of this one's work bought nothing, twice: into registers (1.42 vs
    // 1.39 ms, 60 B of scratch) and by LDS DMA into landing rows of their own (global_load_lds_dwordx4; 53 KB of LDS = three
    // workgroups per CU: config 3 4.85 -> 5.2 ms, config 5's two-word ANDs 12.9 -> 13.9 ms) -- the step is not what the waves wait for.
    uint4 nkw[BT_KW], ndw;
    load_step(wb, nkw, ndw);
    uint64_t klo[BT_KW], khi[BT_KW]; // rowids [0, 64) and [64, 128) of the lane, per keyword
#pragma unroll
    for (int k = 0; k < BT_KW; ++k) klo[k] = nkw[k].x | ((uint64_t)nkw[k].y << 32), khi[k] = nkw[k].z | ((uint64_t)nkw[k].w << 32);
    const uint64_t dlo = ndw.x | ((uint64_t)ndw.y << 32), dhi = ndw.z | ((uint64_t)ndw.w << 32);
    // the query's shared pruning threshold: only a pass that weighs its matches itself prunes, and every 4th step is often enough
    // (all waves of the query read the one word; see scan_bm_kernel)
    if (!need_hits && ((wb - w0) / BT_SPAN & 3u) == 0) {
      const uint32_t gt = __hip_atomic_load(gtaubin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gt > tau_bin) tau_bin = gt;
    }
    const uint32_t lo = wb * 2048u;
    const uint32_t hi_m1 = (wb + BT_SPAN < w1 ? wb + BT_SPAN : w1) * 2048u - 1u; // the step = rowids [lo, hi_m1]
    // sparse keywords: the cursor's docs that fall into the step, assembled in LDS (one keyword at a time) and read back as the lane's four words
#pragma unroll
    for (int k = 0; k < BT_KW; ++k) {
      if (MRK_BTEXP != 4 && (uint32_t)k < nterms && !dense[k] && kj[k] != 0xFFFFFFFFu) {
        const DevTerm& T = Q->t[k];
        bool zeroed = false;
        for (;;) {
          const bool in0 = e0[k] >= lo && e0[k] <= hi_m1, in1 = e1[k] >= lo && e1[k] <= hi_m1;
          if (__ballot(in0 || in1)) {
            if (!zeroed) {
              *(uint4*)(L.abm + lane * 4u) = make_uint4(0, 0, 0, 0);
              wave_lds_fence();
              zeroed = true;
            }
            if (in0) atomicOr(&L.abm[(e0[k] - lo) >> 5], 1u << (e0[k] & 31u));
            if (in1) atomicOr(&L.abm[(e1[k] - lo) >> 5], 1u << (e1[k] & 31u));
          }
          // the next block starts at bnext: beyond the step -> this block served it
          if (bnext[k] > hi_m1) break;
          ++kj[k];
          bt_load_block(a.seg, T, kj[k], e0[k], e1[k], bnext[k]);
        }
        if (zeroed) {
          wave_lds_fence();
          const uint4 v = *(const uint4*)(L.abm + lane * 4u);
          klo[k] = v.x | ((uint64_t)v.y << 32), khi[k] = v.z | ((uint64_t)v.w << 32);
          wave_lds_fence(); // (the next sparse keyword zeroes the same words)
        }
      }
    }
    // the tree on 128 rowids per lane
    uint64_t s0l = 0, s1l = 0, s2l = 0, s3l = 0, s0h = 0, s1h = 0, s2h = 0, s3h = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if ((uint32_t)i < n_nodes) {
        const uint32_t ins = prog[i], op = ins & 0xffu, kw = ins >> 24;
        if (op == PN_TERM) {
          s3l = s2l, s2l = s1l, s1l = s0l;
          s3h = s2h, s2h = s1h, s1h = s0h;
          s0l = kw == 0 ? klo[0] : kw == 1 ? klo[1] : kw == 2 ? klo[2] : klo[3];
          s0h = kw == 0 ? khi[0] : kw == 1 ? khi[1] : kw == 2 ? khi[2] : khi[3];
        } else {
          const uint64_t rl = op == PN_AND ? (s1l & s0l) : op == PN_OR ? (s1l | s0l) : op == PN_MAYBE ? s1l : (s1l & ~s0l);
          const uint64_t rh = op == PN_AND ? (s1h & s0h) : op == PN_OR ? (s1h | s0h) : op == PN_MAYBE ? s1h : (s1h & ~s0h);
          s0l = rl, s1l = s2l, s2l = s3l;
          s0h = rh, s1h = s2h, s2h = s3h;
        }
      }
    }
    uint64_t mlo = s0l & ~dlo, mhi = s0h & ~dhi;
#if MRK_BTEXP == 3
    total += (uint32_t)__popcll(mlo) + (uint32_t)__popcll(mhi);
    mlo = mhi = 0;
#endif
    // ranks of the lane's first rowid: one prefix sum carries two keywords' popcounts (a step holds at most 8192 docs of a keyword)
    uint32_t r0[BT_KW], plo[BT_KW];
#pragma unroll
    for (int k = 0; k < BT_KW; ++k) plo[k] = (uint32_t)__popcll(klo[k]);
#pragma unroll
    for (int k = 0; k < BT_KW; k += 2) {
      r0[k] = r0[k + 1] = 0;
      if ((uint32_t)k < nterms) {
        const uint32_t pc = (plo[k] + (uint32_t)__popcll(khi[k])) | ((plo[k + 1] + (uint32_t)__popcll(khi[k + 1])) << 16);
        const uint32_t incl = wave_incl_scan(pc);
        const uint32_t excl = incl - pc;
        r0[k] = base[k] + (excl & 0xFFFFu);
        r0[k + 1] = base[k + 1] + (excl >> 16);
        const uint32_t tot = rdlane(incl, 63);
        base[k] += tot & 0xFFFFu;
        base[k + 1] += tot >> 16;
      }
    }
    // The step's matches, in two moves (round 3).  (1) Every lane walks its 128 match bits and only NOTES where they are: one
    // 16-bit (lane, bit) per match into a small ring in LDS -- the walk runs as many rounds as the busiest lane has matches, at
    // ~ 1/4 of the lanes busy, so it has to be cheap.  (2) 64 notes at a time, one per lane and every lane busy, become queue
    // entries: the noting lane's words come over by ds_bpermute, the match's rank in every keyword is the lane's first rank
    // plus the popcount of the keyword's bits below the match.  (Both used to be one loop: ~ 80 instructions per round.)
    uint16_t* __restrict__ note = (uint16_t*)L.abm; // (free again: the sparse keywords' words are in registers)
    uint32_t th = 0, tn = 0;                        // ring of BT_NOTES notes: [th, th + tn)
    for (;;) {
      const uint64_t bal = __ballot((mlo | mhi) != 0);
      if (bal) {
        const bool has = (mlo | mhi) != 0;
        const bool up = mlo == 0; // the lane's next match is in its upper 64 rowids
        const uint64_t cur = up ? mhi : mlo;
        const uint32_t bit = has ? (uint32_t)__builtin_ctzll(cur) : 0u;
        if (has) note[(th + tn + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))) & (BT_NOTES - 1u)] = (uint16_t)((lane << 7) | (up ? 64u : 0u) | bit);
        tn += (uint32_t)__popcll(bal);
        const uint64_t nx = cur & (cur - 1ull);
        if (up)
          mhi = nx;
        else
          mlo = nx;
        if (tn < 64u) continue; // (< 64 notes before a round, <= 64 more in it: the ring holds them)
      } else if (!tn)
        break;
      // notes [th, th + n) -> queue entries [qn, qn + n)
      const uint32_t n = tn < 64u ? tn : 64u;
      wave_lds_fence();
      const bool valid = lane < n;
      const uint32_t e = note[(th + lane) & (BT_NOTES - 1u)];
      const uint32_t src = valid ? e >> 7 : lane, b7 = e & 127u, bit = e & 63u;
      const bool up = (e & 64u) != 0;
      const uint64_t below = (1ull << bit) - 1ull;
      const uint64_t msk_lo = up ? ~0ull : below, msk_hi = up ? below : 0ull; // the lane's rowids below the match
      uint32_t pm = 0;
#pragma unroll
      for (int k = 0; k < BT_KW; ++k) {
        if ((uint32_t)k < nterms) {
          const uint64_t sl = (uint64_t)(uint32_t)__shfl((int)(uint32_t)klo[k], (int)src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(klo[k] >> 32), (int)src, 64) << 32);
          const uint64_t sh = (uint64_t)(uint32_t)__shfl((int)(uint32_t)khi[k], (int)src, 64) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(khi[k] >> 32), (int)src, 64) << 32);
          const uint32_t rb = (uint32_t)__shfl((int)r0[k], (int)src, 64);
          pm |= (uint32_t)(((up ? sh : sl) >> bit) & 1ull) << k;
          if (valid) L.q_rank[k][qn + lane] = rb + (uint32_t)__popcll(sl & msk_lo) + (uint32_t)__popcll(sh & msk_hi);
        }
      }
      if (valid) {
        L.q_row[qn + lane] = lo + src * 128u + b7;
        L.q_pm[qn + lane] = pm;
      }
      th += n, tn -= n, qn += n;
      if (qn >= 64u) {
#if MRK_BTEXP != 2
        score(qn - 64u, 64u);
#else
        total += 64u;
#endif
        qn -= 64u;
      }
      wave_lds_fence(); // the scored entries' slots and the notes just read may be rewritten
    }
  }
  if (qn) score(0, qn);
  if (pend_n) {
    wave_lds_fence();
    uint32_t part[MQ_PLANES];
#pragma unroll
    for (int i = 0; i < MQ_PLANES; ++i) part[i] = L.pend[i * 64 + lane];
    write_chunk(part, pend_n);
  }
  if (need_hits) mq_close(a.mq[fat_q ? 1 : 0], mqw, item.query);
  if (cn) publish();
  {
    uint32_t t = total;
    for (int dlt = 32; dlt; dlt >>= 1) t += __shfl_down(t, dlt, 64);
    if (lane == 0 && t) atomicAdd((unsigned long long*)(a.q_total + oq), (unsigned long long)t);
  }
}

void launch_scan_bt(const ScanArgs& a, void* stream) {
  if (!a.n_items) return;
  if (a.q_hist_lb)
    hipLaunchKernelGGL(scan_bt_kernel<true>, dim3(a.n_items), dim3(WG), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(scan_bt_kernel<false>, dim3(a.n_items), dim3(WG), 0, (hipStream_t)stream, a);
}

} // namespace mrk
