// mrk_scan_pk.hip -- scan kernel over PACKED doclists (mrk_pack.cpp), gfx950 / wave64.
//
// Same work decomposition and semantics as scan_kernel (mrk_kernels.hip): one workgroup per
// (query, range of driver-term blocks), each wave streams its own run of driver blocks against
// the other terms' blocks, results go through the workgroup's top-K buffer.  What changes is the
// per-block cost: a block's rowids come out of one bit-field extract + two wave prefix sums
// (no byte-stream parsing), tf / field bits arrive pre-split in one word per lane, probing uses
// a direct rowid -> slot map in LDS, and per-term tfidf(tf) and per-mask field-weight sums are
// table lookups filled once per workgroup with the reference's exact fp32 ops
// (searchnode.cpp:2828, sphinxsearch.cpp:1112-1129).  No MFMA: integer streaming work.
#include "mrk_kcommon.h"
#include "mrk_kprune.h"

#ifndef MRK_EXP
#define MRK_EXP 0
#endif
#ifndef MRK_BURST
#define MRK_BURST 4 // blocks requested back to back per stream (4 or 8); 4 keeps VGPRs <= 96 => 5 waves/SIMD
#endif

namespace mrk {

#ifndef MRK_CBUF
#define MRK_CBUF 128
#endif
constexpr int CBUF = MRK_CBUF; // candidates a wave collects before it publishes them
constexpr int MQCAP = 128;     // matched docs a wave queues for the hit pass (processed 64 at a time)

template <bool PROX, bool TREE>
struct __align__(16) PkWaveLds {
  uint64_t cbuf[CBUF];  // candidates not yet published to the query's global list
  // proximity rankers: where each matched doc sits in the other terms' blocks (block<<7 | slot, bit 31 = lone hit)
  uint32_t href[PROX ? MAX_PROX_TERMS - 1 : 1][PROX ? DEVBLK : 1];
  uint32_t tj_rowid[DEVBLK];
  uint32_t tj_attr[64];
  // boolean trees: what each keyword contributes to each doc of the driver block (tfidf term, field bits)
  // hit rankers / PHRASE: matched docs wait here until 64 of them can go through the hit pass with every lane busy
  // (row, tfidf sum, fields | contributing keywords << 8, one hit reference per keyword)
  uint32_t mq_row[PROX ? MQCAP : 1];
  float mq_acc[PROX ? MQCAP : 1];
  uint32_t mq_fa[PROX ? MQCAP : 1];
  uint32_t mq_ref[PROX ? MAX_PROX_TERMS : 1][PROX ? MQCAP : 1];
  float kv[TREE ? MRK_MAX_AND_TERMS : 1][TREE ? DEVBLK : 4];
  uint8_t kf[TREE ? MRK_MAX_AND_TERMS : 1][TREE ? DEVBLK : 16];
  union {
    uint8_t map[MAPCAP];  // rowid offset -> slot of the decoded other-term block
    uint32_t hist[NBINS]; // publishing scratch: per-bin counts of the candidates being flushed
  };                      // (publishing invalidates the decoded block: it is simply decoded again)
};
static_assert(NBINS * 4 <= MAPCAP, "hist must fit the map area");

template <bool PROX, bool TREE>
struct __align__(16) PkSmem {
  PkWaveLds<PROX, TREE> w[WAVES];
  uint32_t rank[256];
  float tfidf[1][256]; // really [n_terms][256]: the tail lives in dynamic LDS right behind this struct
};

struct PkChunk {
  uint32_t first; // block index (within the term) held by lane 0
  uint32_t bp1;   // first possible rowid of block first+lane (INF past the end)
  uint32_t doff;  // word offset of its deltas
  uint32_t w;     // bits per delta / PK_WIDE
};

__device__ __forceinline__ void load_pk_chunk(PkChunk& c, const DevSegment& seg, const DevTerm& T, uint32_t first) {
  const uint32_t i = first + lane_id();
  c.first = first;
  if (i < T.nblocks) {
    const uint32_t g = T.blk_first + i;
    c.bp1 = seg.pk_base[g];
    c.doff = seg.pk_doff[g];
    c.w = seg.pk_w[g];
  } else {
    c.bp1 = INF_ROWID;
    c.doff = 0;
    c.w = 0;
  }
}

struct PkRaw {
  uint32_t lo, hi, attr;
};

// a block's words for this lane, straight into registers (issued early, used late)
__device__ __forceinline__ PkRaw issue_pk(const DevSegment& seg, const DevTerm& T, const PkChunk& c, uint32_t ci) {
  const uint32_t lane = lane_id();
  const uint32_t w = rdlane(c.w, ci);
  const uint32_t* __restrict__ dp = seg.pk_delta + rdlane(c.doff, ci);
  PkRaw r;
  if (w == PK_WIDE) {
    r.lo = dp[lane];
    r.hi = dp[64 + lane];
  } else {
    const uint32_t wi = (lane * 2 * w) >> 5;
    r.lo = dp[wi];
    r.hi = dp[wi + 1];
  }
  r.attr = seg.pk_attr[(uint64_t)(T.blk_first + c.first + ci) * 64 + lane];
  return r;
}

// rowids of the block's docs lane and lane+64 (o0/o1: their offsets from the block base)
__device__ __forceinline__ void decode_pk(const PkRaw& raw, uint32_t w, uint32_t bp1, uint32_t nd, uint32_t& r0, uint32_t& r1,
                                          uint32_t& o0, uint32_t& o1, bool& ok0, bool& ok1) {
  const uint32_t lane = lane_id();
  if (w == PK_WIDE) {
    o0 = raw.lo;
    o1 = raw.hi;
  } else {
    const uint32_t f = __builtin_amdgcn_alignbit(raw.hi, raw.lo, (lane * 2 * w) & 31u);
    const uint32_t mask = (1u << w) - 1u;
    o0 = f & mask;
    o1 = (f >> w) & mask;
  }
  ok0 = lane < nd;
  ok1 = lane + 64 < nd;
  r0 = bp1 + o0;
  r1 = bp1 + o1;
}

// one VLB-coded u32 out of the hitlist file at byte position p (GetHitlistEntry, sphinx.cpp:374-388):
// two aligned dwords + a funnel shift give the 4-byte window; a 5th byte is fetched when needed
__device__ __forceinline__ uint32_t read_vlb32(const uint8_t* __restrict__ spp, uint64_t& p) {
  const uint64_t al = p & ~3ull;
  const uint32_t w0 = *reinterpret_cast<const uint32_t*>(spp + al);
  const uint32_t w1 = *reinterpret_cast<const uint32_t*>(spp + al + 4);
  const uint32_t x = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)p & 3u);
  // branch-free for 1..4 bytes: the first byte with a clear top bit ends the value; the 7-bit groups are MSB first
  const uint32_t stop = ~x & 0x80808080u;
  const uint32_t n = stop ? ((uint32_t)__builtin_ctz(stop) >> 3) + 1u : 4u;
  const uint32_t all = ((x & 0x7Fu) << 21) | ((x & 0x7F00u) << 6) | ((x >> 9) & 0x3F80u) | ((x >> 24) & 0x7Fu); // 4 groups
  uint32_t val = all >> (7u * (4u - n));
  uint32_t len = n;
  if (!stop) { // 5-byte varint
    val = (val << 7) | (spp[p + 4] & 0x7fu);
    len = 5;
  }
  p += len;
  return val;
}

// next hit of one keyword in one doc (GetNextHit, sphinx.cpp:479-501); 0 (EMPTY_HIT) when exhausted
__device__ __forceinline__ void hit_advance(const uint8_t* __restrict__ spp, uint64_t& p, uint32_t& cur) {
  if (!p) {
    cur = 0;
    return;
  }
  const uint32_t d = read_vlb32(spp, p);
  if (!d) {
    p = 0;
    cur = 0;
  } else
    cur += d;
}

__device__ __forceinline__ bool field_queried(uint32_t qmask, uint32_t hitpos) {
  const uint32_t f = hitpos >> 24;
  return f < 32 ? ((qmask >> f) & 1u) != 0 : qmask == 0xFFFFFFFFu;
}

// TermAcceptor_T<>::IsAcceptableHit (searchnode.cpp:2264-2285): '^word' / 'word$' / '^word$' / '@field[N] word'
__device__ __forceinline__ bool tp_accept(uint32_t kind, uint32_t max_pos, uint32_t hitpos) {
  const uint32_t pos = hitpos & 0x7FFFFFu;
  const bool end = ((hitpos >> 23) & 1u) != 0;
  return kind == MRK_TERMPOS_START      ? pos == 1u
         : kind == MRK_TERMPOS_END      ? end
         : kind == MRK_TERMPOS_STARTEND ? (pos == 1u && end)
         : kind == MRK_TERMPOS_LIMIT    ? pos <= max_pos
                                        : true;
}

// FSMphrase_c (searchnode.cpp:3884-3947): live states = (index of the last word read, expected position of the
// next one).  A first-word hit opens a state; states whose expected position was passed die; a state that reads
// its last word completes an occurrence and resets the machine.
struct PhraseFsm {
  uint32_t fexp[PHRASE_STATES];
  uint32_t ftag, fvalid;
  bool over; // more live states than we keep: the query is failed loudly

  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i) fexp[i] = 0;
    ftag = 0, fvalid = 0, over = false;
  }
  // one hit (position with field, no end bit; query position) of the merged word streams; true = occurrence complete
  __device__ __forceinline__ bool step(uint32_t hp, uint32_t hq, uint32_t nph, uint32_t ap0, uint32_t ap1, uint32_t ap2,
                                       uint32_t ap3) {
    bool emit = false;
    if (hq == (ap0 & 0xFFFFu)) {
      const uint32_t freeb = ~fvalid & ((1u << PHRASE_STATES) - 1u);
      if (!freeb)
        over = true;
      else {
        const uint32_t idx = (uint32_t)__builtin_ctz(freeb);
#pragma unroll
        for (int i = 0; i < PHRASE_STATES; ++i)
          if ((uint32_t)i == idx) fexp[i] = hp + (ap1 - ap0);
        ftag &= ~(3u << (2 * idx));
        fvalid |= 1u << idx;
      }
    }
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i) {
      if (!emit && ((fvalid >> i) & 1u)) {
        if (fexp[i] < hp)
          fvalid &= ~(1u << i);
        else {
          uint32_t tg = (ftag >> (2 * i)) & 3u;
          const uint32_t nextq = tg == 0 ? ap1 : tg == 1 ? ap2 : ap3;
          if (fexp[i] == hp && tg + 1 < nph && (nextq & 0xFFFFu) == hq) {
            ++tg;
            const uint32_t cq = tg == 1 ? ap1 : tg == 2 ? ap2 : ap3, nq = tg == 1 ? ap2 : ap3;
            fexp[i] = tg + 1 < nph ? hp + (nq - cq) : hp - 0x7FFFFFFFu; // FSMphrase_c: -INT_MAX past the last word
            ftag = (ftag & ~(3u << (2 * i))) | (tg << (2 * i));
          }
          if (tg == nph - 1) emit = true;
        }
      }
    }
    if (emit) fvalid = 0; // ResetFSM
    return emit;
  }

  // FSMproximity_c (searchnode.cpp:3958-4075), '"a b c"~N', on the same storage: fexp[i] = m_dProx[i] (last position
  // of the word at query offset i, ~0 = none), ftag = m_uWords, fvalid = m_iMinQindex + 1, exp = m_uExpPos.
  // A hit that completes "all words within qlen + dist" folds them into one hit: position / spanlen of the words
  // gathered, weight from how many of them keep the query's relative offsets; the earliest word is then dropped.
  uint32_t exp;
  __device__ __forceinline__ void reset_prox() {
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i) fexp[i] = 0xFFFFFFFFu;
    ftag = 0, fvalid = 0, over = false, exp = 0;
  }
  __device__ __forceinline__ bool step_prox(uint32_t hp, uint32_t hq, uint32_t nph, uint32_t min_qpos, uint32_t qlen, uint32_t dist,
                                            uint32_t& out_pos, uint32_t& out_w, uint32_t& out_span) {
    const uint32_t qi = hq - min_qpos;
    int min_q = (int)fvalid - 1;
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i)
      if ((uint32_t)i == qi) {
        if (fexp[i] == 0xFFFFFFFFu) ++ftag;
        fexp[i] = hp;
      }
    if (hp >= exp || (int)qi == min_q) {
      min_q = (int)qi;
      uint32_t h = hp;
      const int min_pos = (int)(hp - qlen - dist);
#pragma unroll
      for (int i = 0; i < PHRASE_STATES; ++i)
        if ((uint32_t)i <= qlen && fexp[i] != 0xFFFFFFFFu) {
          if ((int)fexp[i] <= min_pos) {
            fexp[i] = 0xFFFFFFFFu;
            --ftag;
          } else if (fexp[i] < h) {
            min_q = i;
            h = fexp[i];
          }
        }
      uint32_t pm = 0;
#pragma unroll
      for (int i = 0; i < PHRASE_STATES; ++i)
        if (i == min_q) pm = fexp[i];
      exp = pm + qlen + dist;
    }
    fvalid = (uint32_t)(min_q + 1);
    if (ftag != nph) return false;
    // weight: sort the words' (position - query offset); runs of equal values are words in the query's order
    int d[PHRASE_STATES];
    uint32_t umax = 0;
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i) {
      const bool have = (uint32_t)i <= qlen && fexp[i] != 0xFFFFFFFFu;
      d[i] = have ? (int)(fexp[i] - (uint32_t)i) : 0x7FFFFFFF;
      if (have && fexp[i] > umax) umax = fexp[i];
    }
#pragma unroll
    for (int pass = 0; pass < PHRASE_STATES; ++pass) // odd-even transposition sort, 8 elements
#pragma unroll
      for (int i = pass & 1; i + 1 < PHRASE_STATES; i += 2) {
        const int lo = d[i] < d[i + 1] ? d[i] : d[i + 1], hi = d[i] < d[i + 1] ? d[i + 1] : d[i];
        d[i] = lo;
        d[i + 1] = hi;
      }
    uint32_t cur_w = 0, w = 0;
    int last = -0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i)
      if (d[i] != 0x7FFFFFFF) {
        if (d[i] == last)
          ++cur_w;
        else {
          w += cur_w ? 1u + cur_w : 0u;
          cur_w = 0;
        }
        last = d[i];
      }
    w += cur_w ? 1u + cur_w : 0u;
    if (!w) w = 1;
    uint32_t pm = 0;
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i)
      if (i == min_q) pm = fexp[i];
    out_pos = pm;
    out_w = w;
    out_span = umax - pm; // spanlen - 1
    // drop the earliest word and force a recompute on the next hit
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i)
      if (i == min_q) fexp[i] = 0xFFFFFFFFu;
    fvalid = 0;
    --ftag;
    exp = 0;
    return true;
  }
};

// The state rankers (ExtRanker_State_T<STATE>, sphinxsearch.cpp:1198-1315), one doc at a time.  All of them see the
// same hit stream; `ranker` picks the Update / Finalize pair:
//   PROXIMITY_BM25 / PROXIMITY  RankerState_Proximity_fn<.., false>      :1351-1437  LCS per field (BYTE arithmetic)
//   SPH04                       RankerState_ProximityBM25Exact_fn        :1443-1530  LCS + head hit + exact hit
//   MATCHANY                    RankerState_MatchAny_fn                  :1577-1616  LCS + matched query positions per field
//   WORDCOUNT                   RankerState_Wordcount_fn                 :1620-1643  field weight per hit
//   FIELDMASK                   RankerState_Fieldmask_fn                 :1647-1668  fields that hold a hit
// m_uLCS[field] / m_uMatchMask[field] are one byte per field in a u64 (<= 8 fields on this path).
// A doc starts from the post-Finalize state; SPH04's m_uMinExpPos survives Finalize in the reference, which cannot
// change a doc's first hit (it either fails the delta test or both branches agree), so every doc starts it afresh.
struct RankState {
  uint64_t lcs, mmask;
  uint32_t cur_lcs, min_exp_pos, head, exact, fmask;
  int exp_delta, last_pwf, wc;
  bool first;
  __device__ __forceinline__ void reset() {
    lcs = 0, mmask = 0, cur_lcs = 0, min_exp_pos = 0, head = 0, exact = 0, fmask = 0;
    exp_delta = -1, last_pwf = -1, wc = 0, first = true;
  }
  // one hit: hp = position with field (no end bit), is_end = its end-of-field marker, hq = query position,
  // hw = weight (1; word count for a folded phrase hit), hspan = spanlen - 1; w_of = field weight table
  __device__ __forceinline__ void update(uint32_t ranker, bool dupes, uint32_t hp, bool is_end, uint32_t hq, uint32_t hw,
                                         uint32_t hspan, const uint32_t* w_of, int max_qpos) {
    const uint32_t f = hp >> 24;
    const int pwf = (int)hp;
    const int delta = pwf - (int)hq;
    if (ranker == MRK_RANK_WORDCOUNT) {
      wc += f < 8 ? (int)w_of[1u << f] : 0;
      return;
    }
    if (ranker == MRK_RANK_FIELDMASK) {
      fmask |= 1u << (f & 31u);
      return;
    }
    if (ranker == MRK_RANK_SPH04) {
      const int pos = (int)(hp & 0x7FFFFFu);
      if (!first && delta == exp_delta && hp >= min_exp_pos) {
        if (pwf > last_pwf) cur_lcs = (cur_lcs + hw) & 0xffu;
        if (is_end && (int)hq == max_qpos && pos == max_qpos) exact |= 1u << (f & 31u);
      } else {
        if (pwf > last_pwf) cur_lcs = hw & 0xffu;
        if (pos == 1) {
          head |= 1u << (f & 31u);
          if (is_end && max_qpos == 1) exact |= 1u << (f & 31u);
        }
      }
      min_exp_pos = hp + 1u;
      first = false;
    } else if (dupes) {
      // RankerState_Proximity_fn<.., true>::Update (:1370-1412): repeated query keywords -- several query positions
      // may share a hit position.  min_exp_pos / head / exact / fmask double as m_uLcsTailPos / m_uLcsTailQposMask /
      // m_uCurQposMask / m_uCurPos (SPH04 and FIELDMASK have no dupes variant).
      if ((fmask >> 24) != f) exact = 0;
      if (hp != fmask) {
        if (cur_lcs < 2) {
          min_exp_pos = fmask;
          head = exact;
          cur_lcs = 1;
        }
        exact = 0;
        fmask = hp;
        if (f < 8 && (uint32_t)((lcs >> (8 * f)) & 0xffu) < hw) lcs = (lcs & ~(0xffull << (8 * f))) | ((uint64_t)(hw & 0xffu) << (8 * f));
      }
      exact |= (uint32_t)(1ull << (hq & 63u)); // 1UL << qpos stored into a DWORD: positions 32..63 add no bit
      const int dd = (int)(fmask - min_exp_pos);
      if (dd && dd < 32 && ((exact >> (dd & 31)) & head)) {
        head = (uint32_t)(1ull << (hq & 63u));
        min_exp_pos = fmask;
        cur_lcs = (cur_lcs + hw) & 0xffu;
        exact = 0;
        if (f < 8 && cur_lcs > (uint32_t)((lcs >> (8 * f)) & 0xffu)) lcs = (lcs & ~(0xffull << (8 * f))) | ((uint64_t)cur_lcs << (8 * f));
      }
      return;
    } else { // the proximity family
      if (pwf > last_pwf) cur_lcs = (((delta == exp_delta) ? cur_lcs : 0u) + hw) & 0xffu;
      if (ranker == MRK_RANK_MATCHANY && f < 8) mmask |= (uint64_t)((1u << ((hq - 1u) & 31u)) & 0xffu) << (8 * f);
    }
    if (f < 8 && cur_lcs > (uint32_t)((lcs >> (8 * f)) & 0xffu)) lcs = (lcs & ~(0xffull << (8 * f))) | ((uint64_t)cur_lcs << (8 * f));
    last_pwf = pwf;
    exp_delta = delta + (int)hspan;
  }
  __device__ __forceinline__ int finalize(uint32_t ranker, uint32_t nw, const int32_t* weights, const uint32_t* w_of, int n_qwords) const {
    if (ranker == MRK_RANK_WORDCOUNT) return wc;
    if (ranker == MRK_RANK_FIELDMASK) return (int)fmask;
    int rk = 0;
    if (ranker == MRK_RANK_MATCHANY) {
      const int phrase_k = (int)w_of[(1u << nw) - 1u] * n_qwords; // sum of the field weights x query words
      for (uint32_t f = 0; f < nw; ++f) {
        const uint32_t mm = (uint32_t)(mmask >> (8 * f)) & 0xffu;
        if (mm) rk += (int)(__popc(mm) + ((int)((lcs >> (8 * f)) & 0xffu) - 1) * phrase_k) * weights[f];
      }
      return rk;
    }
    for (uint32_t f = 0; f < nw; ++f) {
      const int l = (int)((lcs >> (8 * f)) & 0xffu);
      rk += (ranker == MRK_RANK_SPH04 ? 4 * l + 2 * (int)((head >> f) & 1u) + (int)((exact >> f) & 1u) : l) * weights[f];
    }
    return rk;
  }
};

// one value of the boolean-tree evaluation stack, for the two docs a lane owns
struct TreeEnt {
  bool m[2];     // subtree matches the doc
  float v[2];    // its tfidf sum (0 when unmatched)
  uint32_t f[2]; // its matched-fields bits
  uint32_t a[2]; // keywords whose hits it emits
};

// what one doc's hit pass needs from the kernel (plain values: no reference to the kernel's locals survives)
struct HitCtx {
  const DevQuery* Q;
  const uint8_t* spp;
  const uint32_t* hit;    // DevSegment::pk_hit
  const uint64_t* hbase;  // DevSegment::pk_hbase
  uint32_t* flags;        // the query's flag word
  uint32_t nterms, nw;
  uint32_t ap0, ap1, ap2, ap3;
  uint32_t nph, span;     // the query's phrase: words, distance between its first and last query position
  uint32_t px_dist;       // 0 = exact PHRASE, else the PROXIMITY operator's distance ('"a b"~N')
  uint32_t ranker;        // MRK_RANK_* of the state ranker fed by the pass
  const uint32_t* w_of;   // LDS table: field-weight sum per field mask (w_of[1 << f] = weight of field f)
  int max_qpos, n_qwords; // ExtRanker_c::m_iMaxQpos / m_iQwords
  bool inline_hits, multi_and;
  bool dupes;             // repeated query keywords under a proximity ranker: RankerState_Proximity_fn<.., true>
  uint64_t apack;         // ap0..ap3, 16 bits each: indexed by shifting (a select over the four fields would be
                          // turned into an indexed load and push the whole struct to scratch)
  bool order;             // the keywords of pmask form a BEFORE node (ExtOrder_c), not a PHRASE
  bool termpos;           // some keyword carries a position modifier: its stream yields acceptable hits only
  bool quorum_hits;       // the root is an ExtQuorum_c: hits order by position without the end flag (QuorumCmpHitPos_fn)
};

// One doc's hit pass.  ref0..ref3 = where the doc sits in each keyword's packed arrays (block within the keyword << 7 |
// slot, bit 31 = its one hit was inlined), smask = keyword slots whose hits take part, pmask = slots forming the
// phrase (0 = none), rank = feed the state ranker (else: stop at the first phrase occurrence).
__device__ __forceinline__ void hit_pass(const HitCtx& C, uint32_t ref0, uint32_t ref1, uint32_t ref2, uint32_t ref3, uint32_t smask,
                                         uint32_t pmask, bool rank, bool& ph_found, uint32_t& ph_field, int& rk_out) {
  // .spp cursor (0 = inlined hit / exhausted), current Hitpos_t (0 = exhausted), query position, field limit
  uint64_t sp[MAX_PROX_TERMS];
  uint32_t sc[MAX_PROX_TERMS], sq[MAX_PROX_TERMS], sm[MAX_PROX_TERMS];
  uint32_t tpk[MAX_PROX_TERMS], tpm[MAX_PROX_TERMS]; // ExtTermPos_T: the keyword's acceptor
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t) {
    sp[t] = 0, sc[t] = 0, sq[t] = 0, sm[t] = 0, tpk[t] = 0, tpm[t] = 0;
    if ((uint32_t)t < C.nterms && ((smask >> t) & 1u)) {
      const DevTerm& Tt = C.Q->t[t];
      if (C.termpos) tpk[t] = Tt.tp_kind, tpm[t] = Tt.tp_max;
      const uint32_t h = t == 0 ? ref0 : t == 1 ? ref1 : t == 2 ? ref2 : ref3;
      const uint32_t gblk = Tt.blk_first + ((h >> 7) & 0xFFFFFFu), idx = h & 127u;
      const bool lone = (h >> 31) != 0;
      sq[t] = Tt.qpos;
      sm[t] = Tt.queried32;
      const uint32_t hv = C.hit[(uint64_t)gblk * DEVBLK + idx];
      if (lone) // the hit travelled in the doclist entry (SeekHitlist state 1, sphinx.cpp:461-464)
        sc[t] = hv;
      else {
        sp[t] = C.hbase[gblk] + hv;
        hit_advance(C.spp, sp[t], sc[t]);
      }
      if (C.termpos)
        while (sc[t] && !tp_accept(tpk[t], tpm[t], sc[t])) hit_advance(C.spp, sp[t], sc[t]);
    }
  }
  // the phrase as a stream of folded hits: position = first word's, weight = word count, spanlen = span + 1
  const uint32_t nph = C.nph, span = C.span; // the query's one phrase: word count, last - first query position
  PhraseFsm F;
  if (C.px_dist)
    F.reset_prox();
  else
    F.reset();
  bool phave = false, pdone = pmask == 0, first = true;
  uint32_t pcur = 0, pfield = 0, pw = 0, pspan = 0;
  // BEFORE (ExtOrder_c::GetMatchingHits, searchnode.cpp:4734-4829): the longest in-order run of the children's hits so far
  // and the most recently started one (entry i = child i's hit); a full run is flushed to the ranker hit by hit
  uint32_t ol0 = 0, ol1 = 0, ol2 = 0, ol3 = 0, or0 = 0, or1 = 0, or2 = 0, or3 = 0; // trackers
  uint32_t oe0 = 0, oe1 = 0, oe2 = 0, oe3 = 0;                                     // flushed run waiting for the ranker
  uint32_t olen_l = 0, olen_r = 0, opos_l = 0, opos_r = 0, ofield = 0xFFFFFFFFu, opend_i = 0, opend_n = 0, pq = C.ap0 & 0xFFFFu;
  bool pend_is_end = false;
  RankState X;
  X.reset();
  const uint32_t dmask = smask & ~pmask; // keywords whose hits reach the ranker as they are
  const uint32_t cmpmask = C.quorum_hits ? ~(1u << 23) : 0xFFFFFFFFu; // ExtQuorum_c sorts its hits without the end flag
  // MergeHits3 quirk (searchnode.cpp:3072-3077 + 3052-3054): once one of three streams runs dry the
  // 2-stream merge tests fields against nodes 0 and 1, whichever streams are left, until one more is dry
  int phase = (C.multi_and && !pmask && C.nterms == 3 && (sm[0] & sm[1] & sm[2]) != 0xFFFFFFFFu) ? 0 : 2, tl = 0, tr = 1;
  for (;;) {
    if (!pdone && !phave && C.order && opend_i < opend_n) { // the next hit of the run flushed last
      const uint32_t h = opend_i == 1 ? oe1 : opend_i == 2 ? oe2 : oe3;
      pcur = h & ~(1u << 23), pend_is_end = ((h >> 23) & 1u) != 0;
      pq = (uint32_t)(C.apack >> (16u * opend_i)) & 0xFFFFu;
      pw = 1u, pspan = 0u;
      ++opend_i;
      phave = true;
    }
    if (!pdone && !phave && C.order) {
      for (;;) {
        int best = -1;
        uint32_t bh = 0, bkey = 0, bci = 0, bmask = 0;
#pragma unroll
        for (int t = 0; t < MAX_PROX_TERMS; ++t) { // GetChildIdWithNextHit (:4706-4731): least position, ties to the first child
          const uint32_t q16 = sq[t] & 0xFFFFu;
          const uint32_t ci = q16 == ((uint32_t)C.apack & 0xFFFFu)           ? 0u
                              : q16 == ((uint32_t)(C.apack >> 16) & 0xFFFFu) ? 1u
                              : q16 == ((uint32_t)(C.apack >> 32) & 0xFFFFu) ? 2u
                                                                             : 3u;
          const uint32_t key = sc[t] & ~(1u << 23);
          if (((pmask >> t) & 1u) && sc[t] && (best < 0 || key < bkey || (key == bkey && ci < bci)))
            best = t, bh = sc[t], bkey = key, bci = ci, bmask = sm[t];
        }
        if (best < 0) {
          pdone = true;
          break;
        }
        bool flushed = false;
        if (field_queried(bmask, bh)) {
          const uint32_t hf = bh >> 24, hpos = bh & 0x7FFFFFu;
          if (hf != ofield) { // new field: both trackers start over
            olen_l = olen_r = 0;
            if (bci == 0) {
              ol0 = bh, olen_l = 1, opos_l = hpos + 1u;
              ofield = hf;
            }
          } else if (bci == olen_l && hpos >= opos_l) { // it extends the longest run
            if (bci == 0) ol0 = bh;
            if (bci == 1) ol1 = bh;
            if (bci == 2) ol2 = bh;
            if (bci == 3) ol3 = bh;
            ++olen_l, opos_l = hpos + 1u;
            if (olen_l == C.nph) {
              oe0 = ol0, oe1 = ol1, oe2 = ol2, oe3 = ol3;
              opend_n = olen_l, opend_i = 1;
              olen_l = olen_r = 0;
              opos_r = opos_l;
              flushed = true;
            }
          } else if (bci == 0) { // it restarts the most recent run
            or0 = bh, olen_r = 1, opos_r = hpos + 1u;
            if (!olen_l) ol0 = bh, olen_l = 1, opos_l = hpos + 1u;
          } else if (bci == olen_r && hpos >= opos_r) { // it extends the most recent run
            if (bci == 1) or1 = bh;
            if (bci == 2) or2 = bh;
            if (bci == 3) or3 = bh;
            ++olen_r, opos_r = hpos + 1u;
            if (olen_r == olen_l) { // which just became the longest
              ol0 = or0, ol1 = or1, ol2 = or2, ol3 = or3;
              olen_r = 0;
              opos_l = opos_r;
            }
          }
        }
        {
          uint64_t ap = best == 0 ? sp[0] : best == 1 ? sp[1] : best == 2 ? sp[2] : sp[3];
          uint32_t ac = best == 0 ? sc[0] : best == 1 ? sc[1] : best == 2 ? sc[2] : sc[3];
          hit_advance(C.spp, ap, ac);
          if (C.termpos) {
            const uint32_t ak = best == 0 ? tpk[0] : best == 1 ? tpk[1] : best == 2 ? tpk[2] : tpk[3];
            const uint32_t am = best == 0 ? tpm[0] : best == 1 ? tpm[1] : best == 2 ? tpm[2] : tpm[3];
            while (ac && !tp_accept(ak, am, ac)) hit_advance(C.spp, ap, ac);
          }
#pragma unroll
          for (int t = 0; t < MAX_PROX_TERMS; ++t)
            if (t == best) sp[t] = ap, sc[t] = ac;
        }
        if (flushed) {
          phave = true;
          pcur = oe0 & ~(1u << 23), pend_is_end = ((oe0 >> 23) & 1u) != 0;
          pq = C.ap0 & 0xFFFFu;
          pw = 1u, pspan = 0u;
          pfield = oe0 >> 24;
          break;
        }
      }
      if (first) {
        ph_found = phave;
        ph_field = pfield;
        first = false;
      }
    }
    if (!pdone && !phave && !C.order) { // pull the next occurrence out of the phrase's word streams
      for (;;) {
        int best = -1;
        uint32_t bh = 0, bq = 0, bmask = 0;
#pragma unroll
        for (int t = 0; t < MAX_PROX_TERMS; ++t) // the phrase's top ExtAnd_c orders equal positions by DESCENDING qpos
          if (((pmask >> t) & 1u) && sc[t] && (best < 0 || sc[t] < bh || (sc[t] == bh && sq[t] > bq)))
            best = t, bh = sc[t], bq = sq[t], bmask = sm[t];
        if (best < 0) {
          pdone = true;
          break;
        }
        const uint32_t hp = bh & ~(1u << 23);
        bool emit = false;
        uint32_t e_pos = hp - span, e_w = nph, e_span = span; // exact phrase: first word's position, word count, span
        if (field_queried(bmask, bh))
          emit = C.px_dist ? F.step_prox(hp, bq & 0xFFFFu, nph, C.ap0 & 0xFFFFu, span, C.px_dist, e_pos, e_w, e_span)
                           : F.step(hp, bq & 0xFFFFu, nph, C.ap0, C.ap1, C.ap2, C.ap3);
        { // advance the chosen stream: one decode for the wave, whatever stream each lane picked
          uint64_t ap = best == 0 ? sp[0] : best == 1 ? sp[1] : best == 2 ? sp[2] : sp[3];
          uint32_t ac = best == 0 ? sc[0] : best == 1 ? sc[1] : best == 2 ? sc[2] : sc[3];
          hit_advance(C.spp, ap, ac);
#pragma unroll
          for (int t = 0; t < MAX_PROX_TERMS; ++t)
            if (t == best) sp[t] = ap, sc[t] = ac;
        }
        if (emit) {
          phave = true;
          pcur = e_pos, pw = e_w, pspan = e_span;
          pfield = (bh >> 24) & 31u;
          break;
        }
      }
      if (first) {
        ph_found = phave;
        ph_field = pfield;
        first = false;
      }
    }
    if (!rank) break;
    if (phase == 0 && !(sc[0] && sc[1] && sc[2])) {
      if (!sc[0])
        tl = 1, tr = 2;
      else if (!sc[1])
        tl = 0, tr = 2;
      else
        tl = 0, tr = 1;
      phase = 1;
    }
    if (phase == 1) {
      const uint32_t cl = tl == 0 ? sc[0] : sc[1], cr = tr == 1 ? sc[1] : sc[2];
      if (!(cl && cr)) phase = 2;
    }
    int best = -1;
    uint32_t bh = 0, bq = 0, bmask = 0;
#pragma unroll
    for (int t = 0; t < MAX_PROX_TERMS; ++t)
      if (((dmask >> t) & 1u) && sc[t] &&
          (best < 0 || (sc[t] & cmpmask) < (bh & cmpmask) || ((sc[t] & cmpmask) == (bh & cmpmask) && sq[t] < bq)))
        best = t, bh = sc[t], bq = sq[t], bmask = sm[t];
    // (a BEFORE node hands on plain hits: against its siblings they order by the raw position, end flag included)
    const uint32_t pkey = pcur | (pend_is_end ? 1u << 23 : 0u);
    if (phave && (best < 0 || pkey < bh || (pkey == bh && pq < (bq & 0xFFFFu)))) {
      X.update(C.ranker, C.dupes, pcur, pend_is_end, pq, pw, pspan, C.w_of, C.max_qpos);
      phave = false;
      continue;
    }
    if (best < 0) break;
    if (phase == 1) bmask = best == tl ? sm[0] : sm[1];
    // hits outside the keyword's own field limit never reach the ranker (AddHit, searchnode.cpp:3032-3043)
    if (field_queried(bmask, bh))
      X.update(C.ranker, C.dupes, bh & ~(1u << 23), ((bh >> 23) & 1u) != 0, bq & 0xFFFFu, 1u, 0u, C.w_of, C.max_qpos);
    { // advance the chosen stream: one decode for the wave, whatever stream each lane picked
      uint64_t ap = best == 0 ? sp[0] : best == 1 ? sp[1] : best == 2 ? sp[2] : sp[3];
      uint32_t ac = best == 0 ? sc[0] : best == 1 ? sc[1] : best == 2 ? sc[2] : sc[3];
      hit_advance(C.spp, ap, ac);
      if (C.termpos) {
        const uint32_t ak = best == 0 ? tpk[0] : best == 1 ? tpk[1] : best == 2 ? tpk[2] : tpk[3];
        const uint32_t am = best == 0 ? tpm[0] : best == 1 ? tpm[1] : best == 2 ? tpm[2] : tpm[3];
        while (ac && !tp_accept(ak, am, ac)) hit_advance(C.spp, ap, ac);
      }
#pragma unroll
      for (int t = 0; t < MAX_PROX_TERMS; ++t)
        if (t == best) sp[t] = ap, sc[t] = ac;
    }
  }
  if (rank) rk_out = X.finalize(C.ranker, C.nw, C.Q->weights, C.w_of, C.n_qwords);
  if (F.over) atomicOr(C.flags, QF_FSM);
}

// ExtConditional_T::GetDocsChunk (searchnode.cpp:2332-2405): the keyword holds the doc iff one of its hits -- inside the
// keyword's field limit -- is acceptable.  ref = where the doc sits in the keyword's packed arrays (as for hit_pass).
__device__ __forceinline__ bool termpos_any(const HitCtx& C, const DevTerm& Tt, uint32_t ref) {
  const uint32_t gblk = Tt.blk_first + ((ref >> 7) & 0xFFFFFFu), idx = ref & 127u;
  const uint32_t hv = C.hit[(uint64_t)gblk * DEVBLK + idx];
  uint64_t sp = 0;
  uint32_t sc = 0;
  if (ref >> 31)
    sc = hv;
  else {
    sp = C.hbase[gblk] + hv;
    hit_advance(C.spp, sp, sc);
  }
  while (sc) {
    if (field_queried(Tt.queried32, sc) && tp_accept(Tt.tp_kind, Tt.tp_max, sc)) return true;
    hit_advance(C.spp, sp, sc);
  }
  return false;
}

// EXT: the batch holds queries with position modifiers, a BEFORE node or attribute filters; batches without them run the
// leaner instance (the extra code costs the three-keyword proximity mixes ~7 % even when it never executes)
template <bool PROX, bool TREE, bool EXT = false>
__global__ __launch_bounds__(WG) void scan_pk_kernel(ScanArgs a) {
  extern __shared__ __align__(16) uint8_t smem_raw[];
  PkSmem<PROX, TREE>& s = *reinterpret_cast<PkSmem<PROX, TREE>*>(smem_raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= a.n_items) return;
  const DevItem item = a.items[blockIdx.x];
  const DevQuery* __restrict__ Q = a.queries + item.query;
  const uint32_t nterms = Q->n_terms, K = Q->k, ranker = Q->ranker;
  const uint32_t nw = Q->n_weights < 8u ? Q->n_weights : 8u;
  const uint32_t index_weight = Q->index_weight;
  const DevTerm T0 = Q->t[0];
  const DevTerm T1 = Q->t[nterms > 1 ? 1 : 0];
  const bool inline_hits = a.seg.inline_hits != 0;
  // state rankers read hits; a single keyword under PROXIMITY_BM25 / PROXIMITY was mapped to the weight-sum rankers
  const bool prox_ranker = PROX && ((ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY)
                                        ? nterms > 1
                                        : (ranker == MRK_RANK_WORDCOUNT || ranker == MRK_RANK_MATCHANY ||
                                           ranker == MRK_RANK_FIELDMASK || ranker == MRK_RANK_SPH04));
  PkWaveLds<PROX, TREE>& L = s.w[wave];
  const uint32_t oq = Q->out_q; // logical query: several passes (driver keywords) may feed one result
  const uint32_t req_mask = TREE ? Q->req_mask : 0u, excl_mask = TREE ? Q->excl_mask : 0u;
  const uint32_t n_nodes = TREE ? Q->n_nodes : 0u;
  const bool phrase = PROX && (Q->tree_flags & TF_PHRASE) != 0;                // the whole query is one PHRASE
  const bool ph_leaf = TREE && PROX && (Q->tree_flags & TF_PHRASE_LEAF) != 0; // a PHRASE below other operators
  const uint32_t ph_mask = ph_leaf ? Q->ph_mask : 0u;                         // its words' keyword slots
  const uint32_t ph_n = !PROX ? 0u : phrase ? nterms : (uint32_t)__popc(ph_mask);
  const uint32_t ph_span = !PROX || ph_n < 2 ? 0u : Q->ph_atoms[ph_n - 1] - Q->ph_atoms[0];
  const bool need_hits = PROX && (prox_ranker || phrase); // matches go through the hit pass before they are weighed
  const bool multi_and = (!TREE || (Q->tree_flags & TF_MULTIAND) != 0) && !phrase;
  // PHRASE: query positions of its words in phrase order (FSMphrase_c::m_dAtomPos, searchnode.cpp:3884-3899)
  const uint32_t ap0 = PROX ? Q->ph_atoms[0] : 0u, ap1 = PROX ? Q->ph_atoms[1] : 0u, ap2 = PROX ? Q->ph_atoms[2] : 0u,
                 ap3 = PROX ? Q->ph_atoms[3] : 0u;
  // per-workgroup tables: tfidf(tf) per term, field-weight sum per mask
  for (uint32_t j = 0; j < nterms; ++j) s.tfidf[j][tid] = term_tfidf(tid, Q->t[j].idf);
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
  uint32_t* __restrict__ gcount = a.q_cand_n + oq;
  uint32_t* __restrict__ gtaubin = a.q_tau_bin + oq;
  const uint32_t nb = item.blk_end - item.blk_begin;
  const uint32_t per = (nb + WAVES - 1) / WAVES;
  const uint32_t wb0 = item.blk_begin + wave * per;
  const uint32_t wb1 = wb0 + per < item.blk_end ? wb0 + per : item.blk_end;

  uint32_t total = 0;
  PkChunk c0, cj;
  cj.first = NOBLK;
  uint32_t cj_term = 0, slot_blk = NOBLK, slot_term = 0, kj = 0;
  bool slot_map = false;
  // Blocks are requested in bursts of up to 8 (hipcc drains the whole VMEM queue at the first use of
  // any load, so one memory round trip is paid per burst, not per block).
  const PkRaw zraw{0, 0, 0};
  PkRaw t0r = zraw, t1r = zraw, t2r = zraw, t3r = zraw;
#if MRK_BURST > 4
  PkRaw t4r = zraw, t5r = zraw, t6r = zraw, t7r = zraw;
#endif
  uint32_t tb_first = 0, tb_n = 0; // driver-term burst: blocks tb_first .. tb_first+tb_n-1
  c0.first = NOBLK;
  // other term: a burst of up to 8 consecutive blocks requested back to back (hipcc drains the
  // whole VMEM queue at the first use, so one round trip is paid per burst, not per block)
  PkRaw q0r = zraw, q1r = zraw, q2r = zraw, q3r = zraw;
#if MRK_BURST > 4
  PkRaw q4r = zraw, q5r = zraw, q6r = zraw, q7r = zraw;
#endif
  uint32_t bq_first = 0, bq_n = 0, last_dec = NOBLK;
  uint32_t gt_new = 0;
  __syncthreads(); // tables ready; from here on the waves never meet again
  uint32_t tau_bin = 0, cn = 0, flush_at = 64;

  // publish the wave's buffered candidates: reserve a slice of the query's list with ONE atomic,
  // write it coalesced, add the per-bin counts to the global histogram, re-read the threshold
  auto publish = [&]() {
    if (cn) {
      uint32_t basep = 0;
      if (lane == 0) basep = atomicAdd(gcount, cn);
      basep = rdlane(basep, 0);
      const bool fits = basep + cn <= cand_cap;
      const uint32_t npub = cn;
      for (uint32_t i = lane; i < (uint32_t)NBINS; i += 64) L.hist[i] = 0;
      slot_blk = NOBLK; // the map area is being reused
      wave_lds_fence();
      for (uint32_t i = lane; i < cn; i += 64) {
        const uint64_t key = L.cbuf[i];
        if (fits) cand[basep + i] = key;
        atomicAdd(&L.hist[bin_of(bin_mode, bin_lo, bin_shift, key_weight(key), key_rowid(key))], 1u);
      }
      if (!fits && lane == 0) atomicOr(a.q_flags + oq, QF_OVERFLOW);
      wave_lds_fence();
#if MRK_EXP != 4
      flush_hist(L.hist, ghist);
#endif
      cn = 0;
      // recompute the query's threshold from the merged histogram and share it (one word)
      // Recomputing the threshold reads the whole (hot) histogram: only the publisher whose slice
      // crosses a 2048-candidate boundary of the query's list does it, and shares the result.
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

  const uint32_t rowid_base = a.seg.rowid_base;
  // one match: weight, pruning bin, candidate buffer (all lanes call it; live = this lane holds a match)
  auto emit_match = [&](bool is_live, uint32_t rowid, float tfidf, uint32_t fields, int rk) {
    bool push = false;
    uint64_t key = 0;
    if (is_live) {
      ++total;
      uint32_t weight;
      if (ranker == MRK_RANK_NONE)
        weight = 1u; // ExtRanker_None_c, sphinxsearch.cpp:1160
      else if (PROX && prox_ranker) {
        // Finalize() of the state rankers, e.g. RankerState_Proximity_fn sphinxsearch.cpp:1415-1437
        const int32_t bm = (int32_t)((tfidf + 0.5f) * 1000.0f);
        weight = (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_SPH04) ? (uint32_t)bm + (uint32_t)rk * 1000u : (uint32_t)rk;
      } else if (ranker == MRK_RANK_PROXIMITY) {
        weight = s.rank[fields]; // single keyword: ExtRanker_WeightSum_c<> without BM25 (sphinxsearch.cpp:4216-4217, 1131)
      } else {
        // ExtRanker_WeightSum_c<BM25>, sphinxsearch.cpp:1070, 1112-1129
        const int32_t bm = (int32_t)((tfidf + 0.5f) * 1000.0f);
        weight = (uint32_t)bm + s.rank[fields] * 1000u;
      }
      weight *= index_weight; // MatchExtended, sphinx.cpp:12220
      const uint32_t grow = rowid_base + rowid;
      const uint32_t bin = bin_of(bin_mode, bin_lo, bin_shift, (int32_t)weight, grow);
#if MRK_EXP != 1 && MRK_EXP != 6 && MRK_EXP != 7
      if (bin >= tau_bin) {
        push = true;
        key = make_key((int32_t)weight, grow);
      }
#endif
    }
    const uint64_t bal = __ballot(push);
    if (bal) {
      const uint32_t n = (uint32_t)__popcll(bal);
      if (cn + n > (uint32_t)CBUF) publish(); // keys pushed under the older threshold stay valid candidates
      if (push) L.cbuf[cn + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      cn += n;
      if (cn >= flush_at) {
        publish();
        flush_at = CBUF - 64;
      }
    }
  };

  // hit pass over queued matches [from, from + n), n <= 64, one per lane
  uint32_t mqn = 0;
  HitCtx HC;
  HC.Q = Q;
  HC.spp = a.seg.spp;
  HC.hit = a.seg.pk_hit;
  HC.hbase = a.seg.pk_hbase;
  HC.flags = a.q_flags + oq;
  HC.nterms = nterms, HC.nw = nw;
  HC.ap0 = ap0, HC.ap1 = ap1, HC.ap2 = ap2, HC.ap3 = ap3;
  HC.nph = ph_n, HC.span = ph_span;
  HC.px_dist = PROX ? Q->px_dist : 0u;
  HC.ranker = ranker;
  HC.w_of = s.rank;
  HC.max_qpos = (int)Q->max_qpos, HC.n_qwords = (int)Q->n_qwords;
  HC.inline_hits = inline_hits, HC.multi_and = multi_and;
  HC.quorum_hits = (Q->tree_flags & TF_QUORUM_HITS) != 0;
  HC.termpos = EXT && PROX && TREE && (Q->tree_flags & TF_TERMPOS) != 0; // both only occur in tree programs
  HC.order = EXT && PROX && TREE && (Q->tree_flags & TF_ORDER) != 0;
  HC.apack = (uint64_t)(ap0 & 0xFFFFu) | ((uint64_t)(ap1 & 0xFFFFu) << 16) | ((uint64_t)(ap2 & 0xFFFFu) << 32) | ((uint64_t)(ap3 & 0xFFFFu) << 48);
  HC.dupes = (Q->tree_flags & TF_DUPES) != 0 && (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY);
  auto drain_hits = [&](uint32_t from, uint32_t n) __attribute__((always_inline)) {
    if (!PROX) return;
    wave_lds_fence();
    const bool valid = lane < n;
    const uint32_t e = from + (valid ? lane : 0u);
    const uint32_t rowid = L.mq_row[e], fa = L.mq_fa[e];
    const float tfidf = L.mq_acc[e];
    const uint32_t r0 = L.mq_ref[0][e], r1 = L.mq_ref[1][e], r2 = L.mq_ref[2][e], r3 = L.mq_ref[3][e];
    wave_lds_fence(); // the slots may be refilled from here on
    bool is_live = valid;
    uint32_t fields = fa & 0xffu;
    int rk = 0;
    if (valid) {
      const uint32_t all_slots = (1u << (nterms < (uint32_t)MAX_PROX_TERMS ? nterms : (uint32_t)MAX_PROX_TERMS)) - 1u;
      const uint32_t smask = (fa >> 8) & all_slots;
      const uint32_t pmask = phrase ? all_slots : (ph_leaf && (smask & ph_mask) == ph_mask) ? ph_mask : 0u;
      bool found = false;
      uint32_t ffield = 0;
      hit_pass(HC, r0, r1, r2, r3, smask, pmask, prox_ranker, found, ffield, rk);
      if (phrase) {
        is_live = found;
        fields = 1u << ffield; // the doc's field mask comes from its first occurrence (searchnode.cpp:3836)
      }
    }
    emit_match(is_live, rowid, tfidf, fields, rk);
  };

  for (uint32_t b = wb0; b < wb1; ++b) {
    {

      // ---- driver block b, out of the current burst (request the next 8 when it runs dry)
      if (!(tb_n && b - tb_first < tb_n)) {
        if (c0.first == NOBLK || b < c0.first || b - c0.first + MRK_BURST > (uint32_t)CHUNK) load_pk_chunk(c0, a.seg, T0, b);
        const uint32_t ci = b - c0.first;
        uint32_t nbq = wb1 - b;
        if (nbq > MRK_BURST) nbq = MRK_BURST;
        tb_first = b;
        tb_n = nbq;
        t0r = issue_pk(a.seg, T0, c0, ci);
        if (nbq > 1) t1r = issue_pk(a.seg, T0, c0, ci + 1);
        if (nbq > 2) t2r = issue_pk(a.seg, T0, c0, ci + 2);
        if (nbq > 3) t3r = issue_pk(a.seg, T0, c0, ci + 3);
#if MRK_BURST > 4
        if (nbq > 4) t4r = issue_pk(a.seg, T0, c0, ci + 4);
        if (nbq > 5) t5r = issue_pk(a.seg, T0, c0, ci + 5);
        if (nbq > 6) t6r = issue_pk(a.seg, T0, c0, ci + 6);
        if (nbq > 7) t7r = issue_pk(a.seg, T0, c0, ci + 7);
#endif
      }
      const uint32_t w0 = rdlane(c0.w, b - c0.first), bp0 = rdlane(c0.bp1, b - c0.first);
      const uint32_t left0 = T0.docs - b * DEVBLK;
      PkRaw cur0;
      switch (b - tb_first) {
        case 0: cur0 = t0r; break;
        case 1: cur0 = t1r; break;
        case 2: cur0 = t2r; break;
#if MRK_BURST > 4
        case 3: cur0 = t3r; break;
        case 4: cur0 = t4r; break;
        case 5: cur0 = t5r; break;
        case 6: cur0 = t6r; break;
        default: cur0 = t7r; break;
#else
        default: cur0 = t3r; break;
#endif
      }
      uint32_t row[2], off0[2];
      bool ok[2];
      decode_pk(cur0, w0, bp0, left0 < (uint32_t)DEVBLK ? left0 : (uint32_t)DEVBLK, row[0], row[1], off0[0], off0[1], ok[0], ok[1]);

      uint32_t fld[2];
      float acc[2];
      bool live[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const uint32_t tf = (cur0.attr >> (8 * r)) & 0xffu;
        fld[r] = (cur0.attr >> (16 + 8 * r)) & 0xffu & T0.queried32; // FitsFields
        live[r] = ok[r] && fld[r] != 0;
        float t = s.tfidf[0][tf];
        if (tf == 255u && live[r]) t = term_tfidf(exc_tf(a.seg, T0, row[r]), T0.idf);
        acc[r] = 0.0f + t;
      }
      // trees: live = "still a possible match of this pass"; pres = keywords present in the doc
      uint32_t pres[2] = {0u, 0u}, act[2] = {0u, 0u};
      if (TREE) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          pres[r] = live[r] ? 1u : 0u;
          L.kv[0][lane + 64 * r] = acc[r];
          L.kf[0][lane + 64 * r] = (uint8_t)fld[r];
        }
      }

      // ---- the other terms, in ascending-docs order
#if MRK_EXP == 2 || MRK_EXP == 6
      for (uint32_t j = 1; j < 1; ++j) {
#else
      for (uint32_t j = 1; j < nterms; ++j) {
#endif
        if (!__ballot(live[0] || live[1])) break;
        const DevTerm Tj = j == 1 ? T1 : Q->t[j];
        if (TREE && Tj.nblocks == 0) { // keyword without postings: present nowhere
          if ((req_mask >> j) & 1u) live[0] = live[1] = false;
          continue;
        }
        bool done[2] = {!live[0], !live[1]};
        bool hit[2] = {false, false};
        if (Tj.bm_off != ~0ull) {
          // Dense keyword: instead of decoding its blocks, test each waiting doc's bit in the keyword's doc-set
          // bitmap; a set bit's RANK (directory count of the 256-rowid group + popcounts inside it) is the doc's
          // slot in the keyword's packed arrays (block = rank >> 7, slot = rank & 127).
          const uint32_t* __restrict__ bmj = a.seg.bm + Tj.bm_off;
          const uint32_t* __restrict__ dirj = a.seg.bm_dir + Tj.dir_off;
          const uint32_t row_end = a.seg.n_windows * 2048u;
          uint4 g0[2], g1[2];
          uint32_t db[2];
          bool inb[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            inb[r] = !done[r] && row[r] < row_end;
            const uint32_t grp = inb[r] ? row[r] >> 8 : 0u;
            const uint4* __restrict__ gp = reinterpret_cast<const uint4*>(bmj + (uint64_t)grp * 8);
            g0[r] = gp[0];
            g1[r] = gp[1];
            db[r] = dirj[grp];
          }
          uint32_t rk[2];
          bool present[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const uint32_t wi = (row[r] >> 5) & 7u, bit = row[r] & 31u;
            const uint32_t w[8] = {g0[r].x, g0[r].y, g0[r].z, g0[r].w, g1[r].x, g1[r].y, g1[r].z, g1[r].w};
            uint32_t word = w[0], cnt = db[r];
#pragma unroll
            for (uint32_t i = 0; i < 7; ++i) {
              if (i < wi) cnt += (uint32_t)__popc(w[i]);
              if (i + 1 == wi) word = w[i + 1];
            }
            if (wi == 0) word = w[0];
            present[r] = inb[r] && ((word >> bit) & 1u);
            rk[r] = cnt + (uint32_t)__popc(word & ((1u << bit) - 1u));
          }
          uint32_t aw[2];
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const uint32_t q = present[r] ? rk[r] : 0u;
            aw[r] = a.seg.pk_attr[(uint64_t)(Tj.blk_first + (q >> 7)) * 64 + (q & 63u)];
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const uint32_t sh = ((rk[r] >> 6) & 1u) * 8u;
            const uint32_t tfq = (aw[r] >> sh) & 0xffu;
            const uint32_t fq = (aw[r] >> (16u + sh)) & 0xffu & Tj.queried32;
            if (present[r] && fq != 0) {
              hit[r] = true;
              const float tvx = tfq == 255u ? term_tfidf(exc_tf(a.seg, Tj, row[r]), Tj.idf) : s.tfidf[j][tfq];
              if (TREE) {
                L.kv[j][lane + 64 * r] = tvx;
                L.kf[j][lane + 64 * r] = (uint8_t)fq;
              } else {
                acc[r] = acc[r] + tvx;
                fld[r] |= fq;
              }
              if (PROX && j < (uint32_t)MAX_PROX_TERMS)
                L.href[j - 1][lane + 64 * r] = ((inline_hits && tfq == 1u) ? 0x80000000u : 0u) | rk[r];
            }
          }
        } else {
          if (cj_term != j) {
            cj_term = j;
            kj = 0;
            cj.first = NOBLK;
            bq_n = 0;
            last_dec = NOBLK;
          }
          for (;;) {
            // smallest driver rowid still waiting for this term (docs are in rowid order lane by lane)
            const uint64_t p0 = __ballot(!done[0]), p1 = __ballot(!done[1]);
            if (!(p0 | p1)) break;
            const uint32_t r_min = p0 ? rdlane(row[0], (uint32_t)__builtin_ctzll(p0)) : rdlane(row[1], (uint32_t)__builtin_ctzll(p1));
            // its block: the last one whose base <= r_min (HintRowID's FindSpan), never behind the cursor
            if (cj.first == NOBLK || kj < cj.first || kj - cj.first >= (uint32_t)CHUNK) load_pk_chunk(cj, a.seg, Tj, kj);
            {
              const uint64_t le = __ballot(cj.bp1 <= r_min);
              uint32_t p = le ? 63u - (uint32_t)__builtin_clzll(le) : 0u;
              if (p >= (uint32_t)CHUNK) { // beyond this chunk: wave-wide 64-ary search, then reload
                kj = wave_find_block(a.seg.pk_base + Tj.blk_first, cj.first + CHUNK - 1, Tj.nblocks, r_min);
                load_pk_chunk(cj, a.seg, Tj, kj);
                p = 0;
              }
              const uint32_t k_new = cj.first + p;
              if (k_new > kj) kj = k_new;
            }
            const uint32_t ci = kj - cj.first;
            const uint32_t bp1_k = rdlane(cj.bp1, ci), bp1_n = rdlane(cj.bp1, ci + 1);
            if (slot_blk != kj || slot_term != j) {
              if (!(bq_n && kj >= bq_first && kj - bq_first < bq_n)) {
                // sequential access bets on 8 blocks, a jump on 2
                const uint32_t want = (last_dec == NOBLK || last_dec + 1 == kj) ? (uint32_t)MRK_BURST : 2u;
                uint32_t nbq = Tj.nblocks - kj;
                if (nbq > want) nbq = want;
                if (nbq > (uint32_t)CHUNK - ci) nbq = (uint32_t)CHUNK - ci;
                bq_first = kj;
                bq_n = nbq;
                q0r = issue_pk(a.seg, Tj, cj, ci);
                if (nbq > 1) q1r = issue_pk(a.seg, Tj, cj, ci + 1);
                if (nbq > 2) q2r = issue_pk(a.seg, Tj, cj, ci + 2);
                if (nbq > 3) q3r = issue_pk(a.seg, Tj, cj, ci + 3);
  #if MRK_BURST > 4
                if (nbq > 4) q4r = issue_pk(a.seg, Tj, cj, ci + 4);
                if (nbq > 5) q5r = issue_pk(a.seg, Tj, cj, ci + 5);
                if (nbq > 6) q6r = issue_pk(a.seg, Tj, cj, ci + 6);
                if (nbq > 7) q7r = issue_pk(a.seg, Tj, cj, ci + 7);
  #endif
                // the shared threshold word rides along with the burst
                gt_new = __hip_atomic_load(gtaubin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              last_dec = kj;
              PkRaw rj;
              switch (kj - bq_first) {
                case 0: rj = q0r; break;
                case 1: rj = q1r; break;
                case 2: rj = q2r; break;
  #if MRK_BURST > 4
                case 3: rj = q3r; break;
                case 4: rj = q4r; break;
                case 5: rj = q5r; break;
                case 6: rj = q6r; break;
                default: rj = q7r; break;
  #else
                default: rj = q3r; break;
  #endif
              }
              const uint32_t wj = rdlane(cj.w, ci);
              const uint32_t leftj = Tj.docs - kj * DEVBLK;
              const uint32_t ndj = leftj < (uint32_t)DEVBLK ? leftj : (uint32_t)DEVBLK;
              uint32_t e0, e1, f0, f1;
              bool k0, k1;
              decode_pk(rj, wj, bp1_k, ndj, e0, e1, f0, f1, k0, k1);
              // offsets of a w-bit block stay below 2^w: small enough => direct map
              slot_map = wj != PK_WIDE && (1u << wj) <= (uint32_t)MAPCAP;
              L.tj_rowid[lane] = k0 ? e0 : INF_ROWID;
              L.tj_rowid[lane + 64] = k1 ? e1 : INF_ROWID;
              L.tj_attr[lane] = rj.attr;
              if (slot_map) {
                if (k0) L.map[f0] = (uint8_t)lane;
                if (k1) L.map[f1] = (uint8_t)(lane + 64);
              }
              slot_blk = kj;
              slot_term = j;
              wave_lds_fence();
            }
            // probe: driver docs that fall into [bp1_k, bp1_n).  Both docs of a lane go through the
            // dependent LDS reads side by side (map -> {rowid, attr} -> tfidf) instead of one after the other.
            {
              bool inr[2];
              uint32_t pos[2];
  #pragma unroll
              for (int r = 0; r < 2; ++r) {
                inr[r] = !done[r] && row[r] >= bp1_k && row[r] < bp1_n;
                done[r] = done[r] || inr[r];
              }
              if (slot_map) {
                uint32_t mb[2];
  #pragma unroll
                for (int r = 0; r < 2; ++r) {
                  const uint32_t o = row[r] - bp1_k;
                  mb[r] = L.map[inr[r] && o < (uint32_t)MAPCAP ? o : 0u]; // stale bytes are caught by the rowid check
                }
  #pragma unroll
                for (int r = 0; r < 2; ++r) pos[r] = mb[r] & 127u;
              } else {
  #pragma unroll
                for (int r = 0; r < 2; ++r) {
                  uint32_t p = 0;
                  const uint32_t rowid = inr[r] ? row[r] : 0u;
  #pragma unroll
                  for (uint32_t step = DEVBLK / 2; step; step >>= 1)
                    if (L.tj_rowid[p + step - 1] < rowid) p += step;
                  pos[r] = p;
                }
              }
              uint32_t rid[2], aw[2];
  #pragma unroll
              for (int r = 0; r < 2; ++r) {
                rid[r] = L.tj_rowid[pos[r]];
                aw[r] = L.tj_attr[pos[r] & 63u];
              }
              uint32_t tfj[2], fj[2];
              bool hp[2];
  #pragma unroll
              for (int r = 0; r < 2; ++r) {
                const uint32_t sh = (pos[r] >> 6) * 8;
                tfj[r] = (aw[r] >> sh) & 0xffu;
                fj[r] = (aw[r] >> (16 + sh)) & 0xffu & Tj.queried32;
                hp[r] = inr[r] && rid[r] == row[r] && fj[r] != 0;
              }
              float tv[2];
  #pragma unroll
              for (int r = 0; r < 2; ++r) tv[r] = s.tfidf[j][tfj[r]];
  #pragma unroll
              for (int r = 0; r < 2; ++r) {
                if (hp[r]) {
                  hit[r] = true;
                  const float tvx = tfj[r] == 255u ? term_tfidf(exc_tf(a.seg, Tj, row[r]), Tj.idf) : tv[r];
                  if (TREE) {
                    L.kv[j][lane + 64 * r] = tvx;
                    L.kf[j][lane + 64 * r] = (uint8_t)fj[r];
                  } else {
                    acc[r] = acc[r] + tvx;
                    fld[r] |= fj[r];
                  }
                  if (PROX && j < (uint32_t)MAX_PROX_TERMS)
                    L.href[j - 1][lane + 64 * r] = ((inline_hits && tfj[r] == 1u) ? 0x80000000u : 0u) | (kj << 7) | pos[r];
                }
              }
            }
  #pragma unroll
            for (int r = 0; r < 2; ++r)
              if (!done[r] && row[r] < bp1_k) done[r] = true; // cannot happen (cursor only moves forward)
          }
        }
        if (TREE) {
          const bool rq = ((req_mask >> j) & 1u) != 0, ex = ((excl_mask >> j) & 1u) != 0;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            if (hit[r]) pres[r] |= 1u << j;
            if (rq) live[r] = live[r] && hit[r];
            if (ex) live[r] = live[r] && !hit[r]; // an earlier pass owns this doc
          }
        } else {
          live[0] = live[0] && hit[0];
          live[1] = live[1] && hit[1];
        }
      }

      // ---- keywords with a position modifier: the doc holds them only through an acceptable hit
      if (TREE && PROX && HC.termpos && __ballot(live[0] || live[1])) {
        wave_lds_fence();
        for (uint32_t j = 0; j < nterms && j < (uint32_t)MAX_PROX_TERMS; ++j) {
          const DevTerm& Tj = Q->t[j];
          if (!Tj.tp_kind) continue;
          const bool rq = ((req_mask >> j) & 1u) != 0;
#pragma unroll
          for (int r = 0; r < 2; ++r)
            if (live[r] && ((pres[r] >> j) & 1u)) {
              const uint32_t dref = ((inline_hits && ((cur0.attr >> (8 * r)) & 0xffu) == 1u) ? 0x80000000u : 0u) | (b << 7) | (lane + 64 * r);
              const uint32_t ref = j == 0 ? dref : L.href[j - 1][lane + 64 * r];
              if (!termpos_any(HC, Tj, ref)) {
                pres[r] &= ~(1u << j);
                if (rq) live[r] = false;
              }
            }
        }
      }

      // ---- a PHRASE below other operators: whether it occurs has to be known before the tree is evaluated
      bool ph_ok[2] = {false, false};
      uint32_t ph_fld[2] = {0u, 0u};
      if (TREE && PROX && ph_leaf && __ballot(live[0] || live[1])) {
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (live[r] && (pres[r] & ph_mask) == ph_mask) {
            int unused = 0;
            const uint32_t dref = ((inline_hits && ((cur0.attr >> (8 * r)) & 0xffu) == 1u) ? 0x80000000u : 0u) | (b << 7) | (lane + 64 * r);
            hit_pass(HC, dref, L.href[0][lane + 64 * r], L.href[1][lane + 64 * r], L.href[2][lane + 64 * r], ph_mask, ph_mask, false,
                     ph_ok[r], ph_fld[r], unused);
          }
      }

      // ---- boolean tree: post-order program over the keywords' presence bits.  Value rules restate
      // ExtAnd_c / ExtOr_c / ExtMaybe_c / ExtAndNot_c (searchnode.cpp:2585-2594, 3494-3540, 3587-3600,
      // 3650-3680): tfidf adds left + right where both sides hold the doc, fields OR together.
      if (TREE && __ballot(live[0] || live[1])) {
        TreeEnt s0{}, s1{}, s2{}, s3{};
        for (uint32_t i = 0; i < n_nodes; ++i) {
          const uint32_t ins = Q->prog[i];
          const uint32_t op = ins & 0xffu, kw = ins >> 24;
          if (op == PN_TERM) {
            s3 = s2;
            s2 = s1;
            s1 = s0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const bool m = live[r] && ((pres[r] >> kw) & 1u);
              const float v = L.kv[kw][lane + 64 * r];
              const uint32_t f = L.kf[kw][lane + 64 * r];
              s0.m[r] = m;
              s0.v[r] = m ? v : 0.0f;
              s0.f[r] = m ? f : 0u;
              s0.a[r] = m ? 1u << kw : 0u;
            }
          } else if (op == PN_QUORUM) {
            // ExtQuorum_c (searchnode.cpp:4466-4545): at least qr_thr of its keywords hold the doc; tfidf adds up in the
            // order m_dChildren has at this rowid (keywords whose doclists ended before it have left by RemoveFast)
            s3 = s2;
            s2 = s1;
            s1 = s0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const uint32_t pm = pres[r] & Q->qr_mask;
              const bool m = live[r] && (uint32_t)__popc(pm) >= Q->qr_thr;
              uint32_t ord = Q->qr_ord[0];
              for (uint32_t e = 0; e < Q->qr_n; ++e)
                if (row[r] > Q->qr_row[e]) ord = Q->qr_ord[e + 1];
              float v = 0.0f;
              uint32_t f = 0;
              bool first = true;
#pragma unroll
              for (int i = 0; i < QUORUM_EVENTS; ++i) {
                const uint32_t sl = (ord >> (4 * i)) & 15u;
                if (sl != 15u && ((pm >> sl) & 1u)) {
                  const float x = L.kv[sl & 7u][lane + 64 * r];
                  v = first ? x : v + x;
                  first = false;
                  f |= L.kf[sl & 7u][lane + 64 * r];
                }
              }
              s0.m[r] = m;
              s0.v[r] = m ? v : 0.0f;
              s0.f[r] = m ? f : 0u;
              s0.a[r] = m ? pm : 0u;
            }
          } else if (op == PN_ORDERFIX) {
            // ExtOrder_c over the AND chain of its keywords just evaluated: the doc stays only if their hits line up in
            // order; it is the FIRST child's doc -- that keyword's tfidf and fields alone (searchnode.cpp:4907-4908)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const bool m = s0.m[r] && ph_ok[r];
              s0.m[r] = m;
              s0.v[r] = m ? L.kv[kw][lane + 64 * r] : 0.0f;
              s0.f[r] = m ? (uint32_t)L.kf[kw][lane + 64 * r] : 0u;
              s0.a[r] = m ? s0.a[r] : 0u;
            }
          } else if (op == PN_PHRASEFIX) {
            // ExtNWay_T<FSMphrase_c> over the AND chain of its words just evaluated: the doc stays only if the
            // words line up; its field mask is the field of the first occurrence (searchnode.cpp:3806-3848)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const bool m = s0.m[r] && ph_ok[r];
              s0.m[r] = m;
              s0.v[r] = m ? s0.v[r] : 0.0f;
              s0.f[r] = m ? 1u << ph_fld[r] : 0u;
              s0.a[r] = m ? s0.a[r] : 0u;
            }
          } else {
            TreeEnt o;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              bool m;
              if (op == PN_AND)
                m = s1.m[r] && s0.m[r];
              else if (op == PN_OR)
                m = s1.m[r] || s0.m[r];
              else if (op == PN_MAYBE)
                m = s1.m[r];
              else
                m = s1.m[r] && !s0.m[r];
              const bool both = op != PN_ANDNOT; // ANDNOT passes its left side through
              o.m[r] = m;
              o.v[r] = m ? (both ? s1.v[r] + s0.v[r] : s1.v[r]) : 0.0f; // x + 0.0f == x: an absent side adds nothing
              o.f[r] = m ? (both ? s1.f[r] | s0.f[r] : s1.f[r]) : 0u;
              o.a[r] = m ? (both ? s1.a[r] | s0.a[r] : s1.a[r]) : 0u;
            }
            s0 = o;
            s1 = s2;
            s2 = s3;
          }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          live[r] = live[r] && s0.m[r];
          acc[r] = s0.v[r];
          fld[r] = s0.f[r];
          act[r] = s0.a[r];
        }
      }

      if (gt_new > tau_bin) tau_bin = gt_new;
      if (EXT && Q->n_filters) { // EarlyReject: filtered rows never reach the ranker (sphinxsearch.cpp:1055-1064)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (live[r] && !row_passes_filters(a.seg, Q->filters, Q->n_filters, row[r])) live[r] = false;
      }
      if (a.seg.dead) { // MatchExtended drops dead rows before they reach the sorter (sphinx.cpp:12213-12217)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (live[r] && row_is_dead(a.seg, row[r])) live[r] = false;
      }
      // ---- matches.  Hit rankers / PHRASE: per matched doc the keywords' hit streams are merged by (hitpos, qpos)
      // (MergeHits2/3/N, searchnode.cpp:3047-3181; ExtAnd_c / ExtOr_c::CollectHits) and fed to the state ranker
      // (sphinxsearch.cpp:1351-1668); the words of a PHRASE go through their state machine first -- to the ranker the
      // phrase is one more stream, of folded hits.  That pass is a chain of dependent loads per doc, so matched docs
      // are queued and go through it 64 at a time, one per lane, instead of the few a single driver block holds.
      if (PROX && need_hits) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint64_t bal = __ballot(live[r]);
          if (bal) {
            if (live[r]) {
              const uint32_t pos = mqn + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
              L.mq_row[pos] = row[r];
              L.mq_acc[pos] = acc[r];
              L.mq_fa[pos] = (fld[r] & 0xffu) | ((TREE ? act[r] : 0xffu) << 8);
              L.mq_ref[0][pos] = ((inline_hits && ((cur0.attr >> (8 * r)) & 0xffu) == 1u) ? 0x80000000u : 0u) | (b << 7) | (lane + 64 * r);
#pragma unroll
              for (int t = 1; t < MAX_PROX_TERMS; ++t) L.mq_ref[t][pos] = L.href[t - 1][lane + 64 * r];
            }
            mqn += (uint32_t)__popcll(bal);
            if (mqn >= 64u) {
              drain_hits(mqn - 64u, 64u);
              mqn -= 64u;
            }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) emit_match(live[r], row[r], acc[r], fld[r], 0);
      }
    }
  }
  if (PROX && mqn) drain_hits(0u, mqn);

  // ---- wave epilogue
  if (cn) publish();
  {
    uint32_t t = total;
    for (int dlt = 32; dlt; dlt >>= 1) t += __shfl_down(t, dlt, 64);
    if (lane == 0 && t) atomicAdd((unsigned long long*)(a.q_total + oq), (unsigned long long)t);
  }
}

// ---------------------------------------------------------------------------------------
// select: exact top-K of a query's candidate list, one workgroup per query
// ---------------------------------------------------------------------------------------
struct __align__(16) SelSmem {
  uint64_t cand[CAND];
  uint32_t wave_cnt[2 * WAVES];
  uint32_t cand_n;
  uint32_t tau_bin;
  uint64_t tau;
};

__global__ __launch_bounds__(WG) void select_kernel(SelectArgs a) {
  __shared__ SelSmem s;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  if (q >= a.n_queries) return;
  const DevQuery* __restrict__ Q = a.queries + q;
  const uint32_t K = Q->k ? Q->k : 1u;
  uint32_t n = a.q_cand_n[q];
  if (n > Q->cand_cap) n = Q->cand_cap;
  if (tid == 0) {
    s.cand_n = 0;
    s.tau = 0;
  }
  if (tid < 64) {
    const uint32_t tb = threshold_bin(a.q_hist + (uint64_t)q * NBINS, K);
    if (lane == 0) s.tau_bin = tb;
  }
  __syncthreads();
  const uint32_t tau_bin = s.tau_bin;
  const uint32_t bin_mode = Q->bin_mode, bin_shift = Q->bin_shift;
  const int32_t bin_lo = Q->bin_lo;
  const uint64_t* __restrict__ src = a.cand + Q->cand_off;
  for (uint32_t f0 = 0; f0 < n; f0 += 4 * WG) {
    if (s.cand_n > (uint32_t)(CAND - 4 * WG)) compact_cand(s, K, &s.tau);
    const uint64_t tau = s.tau;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t f = f0 + r * WG + tid;
      uint64_t key = 0;
      bool push = false;
      if (f < n) {
        key = src[f];
        push = key >= tau && bin_of(bin_mode, bin_lo, bin_shift, key_weight(key), key_rowid(key)) >= tau_bin;
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
  const uint32_t m = compact_cand(s, K, &s.tau); // sorted best-first
  for (uint32_t i = tid; i < m; i += WG) a.out_keys[(uint64_t)q * KCAP + i] = s.cand[i];
  if (tid == 0) a.out_cnt[q] = m;
}

template <bool PROX, bool TREE, bool EXT = false>
static void launch_pk(const ScanArgs& a, size_t tail, hipStream_t st) {
  hipLaunchKernelGGL((scan_pk_kernel<PROX, TREE, EXT>), dim3(a.n_items), dim3(WG), sizeof(PkSmem<PROX, TREE>) + tail, st, a);
}

void launch_scan_pk(const ScanArgs& a, uint32_t max_terms, bool prox, bool tree, bool ext, void* stream) {
  if (!a.n_items) return;
  if (max_terms < 1) max_terms = 1;
  const size_t tail = (size_t)(max_terms - 1) * 256 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (ext) // filters alone may come with a plain AND: the EXT instance is the full tree + hit-stream kernel
    launch_pk<true, true, true>(a, tail, st);
  else if (tree)
    prox ? launch_pk<true, true>(a, tail, st) : launch_pk<false, true>(a, tail, st);
  else
    prox ? launch_pk<true, false>(a, tail, st) : launch_pk<false, false>(a, tail, st);
}

void launch_select(const SelectArgs& a, void* stream) {
  if (!a.n_queries) return;
  hipLaunchKernelGGL(select_kernel, dim3(a.n_queries), dim3(WG), 0, (hipStream_t)stream, a);
}

} // namespace mrk
