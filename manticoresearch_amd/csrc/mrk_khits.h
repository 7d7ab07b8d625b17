// mrk_khits.h -- the per-doc hit pass shared by the block-scan kernel (in-scan decisions: PHRASE below other operators,
// position modifiers, BEFORE) and rank_kernel (final ranking of queued matches): hitlist VLB decode, the k-way hit merge
// (MergeHits2/3/N, searchnode.cpp:3047-3181), the PHRASE / PROXIMITY / BEFORE state machines and the state rankers
// (sphinxsearch.cpp:1198-1668).  gfx950 / wave64, one doc per lane.
#pragma once
#include "mrk_kcommon.h"

namespace mrk {

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
  // FSMmultinear_c (searchnode.cpp:4096-4288) for exactly TWO keyword operands ('a NEAR/N b'), on the same storage:
  // fexp[0] = m_uLastP, [1] = m_uFirstHit, [2] = m_uFirstNpos, [3] = m_uFirstQpos, [4] = m_uWeight.  Keyword hits have
  // weight, match length and span 1, which leaves the pre-last roll-back and the overlap special case of HitFSM dead.  (With
  // three or more operands the reference's emitted query position depends on the docs seen before -- m_uFirstQpos is never
  // reset for the ring form -- so only the twofer is offered on the device.)
  __device__ __forceinline__ void reset_near() {
#pragma unroll
    for (int i = 0; i < PHRASE_STATES; ++i) fexp[i] = 0;
    ftag = 0, fvalid = 0, over = false, exp = 0;
  }
  __device__ __forceinline__ bool step_near2(uint32_t hp, uint32_t npos, uint32_t qpos, uint32_t dist, uint32_t& out_pos, uint32_t& out_w, uint32_t& out_q) {
    if (fexp[0] == hp) { // dupe hit: the leftmost (in the query) of the words at this position leads ('a NEAR/2 a')
      if (npos < fexp[2]) fexp[3] = qpos, fexp[2] = npos;
      return false;
    }
    if (fexp[0] == 0 || fexp[0] + 1u + dist <= hp) { // probably a new chain
      fexp[1] = fexp[0] = hp, fexp[4] = 1, fexp[3] = qpos, fexp[2] = npos;
      return false;
    }
    if (npos == fexp[2]) { // the same operand again: the chain restarts from it
      if (fexp[0] < hp) fexp[1] = fexp[0] = hp, fexp[4] = 1, fexp[3] = qpos, fexp[2] = npos;
      return false;
    }
    out_pos = fexp[1];
    out_w = fexp[4] + 1u;
    out_q = fexp[3] < qpos ? fexp[3] : qpos;
    fexp[1] = fexp[0] = hp, fexp[4] = 1, fexp[3] = qpos; // two operands may overlap: shift the chain, do not reset it
    return true;
  }

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
  // hw = weight (1; word count for a folded phrase hit), hspan = spanlen - 1; fw = the nw per-field weights
  __device__ __forceinline__ void update(uint32_t ranker, bool dupes, uint32_t hp, bool is_end, uint32_t hq, uint32_t hw,
                                         uint32_t hspan, const int32_t* fw, uint32_t nw, int max_qpos) {
    const uint32_t f = hp >> 24;
    const int pwf = (int)hp;
    const int delta = pwf - (int)hq;
    if (ranker == MRK_RANK_WORDCOUNT) {
      wc += f < nw ? fw[f] : 0;
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
  __device__ __forceinline__ int finalize(uint32_t ranker, uint32_t nw, const int32_t* weights, int n_qwords) const {
    if (ranker == MRK_RANK_WORDCOUNT) return wc;
    if (ranker == MRK_RANK_FIELDMASK) return (int)fmask;
    int rk = 0;
    if (ranker == MRK_RANK_MATCHANY) {
      int wsum = 0;
      for (uint32_t f = 0; f < nw; ++f) wsum += weights[f];
      const int phrase_k = wsum * n_qwords; // sum of the field weights x query words
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
  const uint8_t* spp;
  const uint32_t* hit;    // DevSegment::pk_hit
  const uint64_t* hbase;  // DevSegment::pk_hbase
  uint32_t* flags;        // the query's flag word
  uint32_t nterms, nw;
  uint32_t ap0, ap1, ap2, ap3;
  uint32_t nph, span;     // the query's phrase: words, distance between its first and last query position
  uint32_t px_dist;       // 0 = exact PHRASE, else the PROXIMITY operator's distance ('"a b"~N'); bit 31: a NEAR/N node, N below it
  uint32_t ranker;        // MRK_RANK_* of the state ranker fed by the pass
  const int32_t* fw;      // the nw per-field weights (an LDS copy where the pass ranks; WORDCOUNT reads one per hit)
  // per keyword slot, by value (uniform): first block in the packed arrays, query position, queried fields, position modifier
  uint32_t tb[MAX_PROX_TERMS], tq[MAX_PROX_TERMS], tm[MAX_PROX_TERMS], tpk[MAX_PROX_TERMS], tpm[MAX_PROX_TERMS];
  int max_qpos, n_qwords; // ExtRanker_c::m_iMaxQpos / m_iQwords
  bool inline_hits, multi_and;
  bool dupes;             // repeated query keywords under a proximity ranker: RankerState_Proximity_fn<.., true>
  uint64_t apack;         // ap0..ap3, 16 bits each: indexed by shifting (a select over the four fields would be
                          // turned into an indexed load and push the whole struct to scratch)
  bool order;             // the keywords of pmask form a BEFORE node (ExtOrder_c), not a PHRASE
  bool termpos;           // some keyword carries a position modifier: its stream yields acceptable hits only
  bool quorum_hits;       // the root is an ExtQuorum_c: hits order by position without the end flag (QuorumCmpHitPos_fn)
  // NOTNEAR (ExtNotNear_c): keyword slot nn_a's hits survive only if no hit of slot nn_b at or behind them comes within nn_dist
  uint32_t nn_a, nn_b, nn_dist; // nn_dist = 0: no such node
};

// ExtNotNear_c::FilterHits (searchnode.cpp:5352-5380) for two keywords: (ap, ac) = the must side's cursor on a candidate hit,
// (np, nc) = the not side's; drops candidates until one survives -- no not-hit left at or behind it, or the next one farther
// than the distance (keyword hits are one position long) -- or the must side runs dry.  Not-hits outside that keyword's
// field limit do not exist for the node.
__device__ __forceinline__ void notnear_filter(const uint8_t* __restrict__ spp, uint64_t& ap, uint32_t& ac, uint64_t& np, uint32_t& nc, uint32_t mask_b,
                                               uint32_t dist) {
  while (ac) {
    const uint32_t pm = ac & ~(1u << 23);
    while (nc && ((nc & ~(1u << 23)) < pm || !field_queried(mask_b, nc))) hit_advance(spp, np, nc);
    if (!nc || pm + dist < (nc & ~(1u << 23))) break;
    hit_advance(spp, ap, ac);
  }
}

// One doc's hit pass.  ref0..ref3 = where the doc sits in each keyword's packed arrays (block within the keyword << 7 |
// slot, bit 31 = its one hit was inlined), smask = keyword slots whose hits take part, pmask = slots forming the
// phrase (0 = none), rank = feed the state ranker (else: stop at the first phrase occurrence).
__device__ __forceinline__ void hit_pass(const HitCtx& C, uint32_t ref0, uint32_t ref1, uint32_t ref2, uint32_t ref3, uint32_t smask,
                                         uint32_t pmask, bool rank, bool& ph_found, uint32_t& ph_field, int& rk_out, bool nn_partner = false) {
  // .spp cursor (0 = inlined hit / exhausted), current Hitpos_t (0 = exhausted), query position, field limit
  uint64_t sp[MAX_PROX_TERMS];
  uint32_t sc[MAX_PROX_TERMS], sq[MAX_PROX_TERMS], sm[MAX_PROX_TERMS];
  uint32_t tpk[MAX_PROX_TERMS], tpm[MAX_PROX_TERMS]; // ExtTermPos_T: the keyword's acceptor
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t) {
    sp[t] = 0, sc[t] = 0, sq[t] = 0, sm[t] = 0, tpk[t] = 0, tpm[t] = 0;
    if ((uint32_t)t < C.nterms && ((smask >> t) & 1u)) {
      if (C.termpos) tpk[t] = C.tpk[t], tpm[t] = C.tpm[t];
      const uint32_t h = t == 0 ? ref0 : t == 1 ? ref1 : t == 2 ? ref2 : ref3;
      const uint32_t gblk = C.tb[t] + ((h >> 7) & 0xFFFFFFu), idx = h & 127u;
      const bool lone = (h >> 31) != 0;
      sq[t] = C.tq[t];
      sm[t] = C.tm[t];
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
  // NOTNEAR: the not side's own cursor (its hits never reach the ranker), and the must side's first surviving hit
  const bool nn = C.nn_dist != 0 && nn_partner;
  uint64_t nnp = 0;
  uint32_t nnc = 0;
  if (nn) {
    const uint32_t h = C.nn_b == 0 ? ref0 : C.nn_b == 1 ? ref1 : C.nn_b == 2 ? ref2 : ref3;
    const uint32_t tbb = C.nn_b == 0 ? C.tb[0] : C.nn_b == 1 ? C.tb[1] : C.nn_b == 2 ? C.tb[2] : C.tb[3];
    const uint32_t gblk = tbb + ((h >> 7) & 0xFFFFFFu), idx = h & 127u;
    const uint32_t hv = C.hit[(uint64_t)gblk * DEVBLK + idx];
    if (h >> 31)
      nnc = hv;
    else {
      nnp = C.hbase[gblk] + hv;
      hit_advance(C.spp, nnp, nnc);
    }
#pragma unroll
    for (int t = 0; t < MAX_PROX_TERMS; ++t)
      if ((uint32_t)t == C.nn_a) notnear_filter(C.spp, sp[t], sc[t], nnp, nnc, (C.nn_b == 0 ? C.tm[0] : C.nn_b == 1 ? C.tm[1] : C.nn_b == 2 ? C.tm[2] : C.tm[3]), C.nn_dist);
  }
  // the phrase as a stream of folded hits: position = first word's, weight = word count, spanlen = span + 1
  const uint32_t nph = C.nph, span = C.span; // the query's one phrase: word count, last - first query position
  PhraseFsm F;
  const bool near2 = (C.px_dist >> 31) != 0; // 'a NEAR/N b' (two keyword operands)
  const uint32_t px_dist = C.px_dist & 0x7FFFFFFFu;
  if (near2)
    F.reset_near();
  else if (px_dist)
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
        uint32_t e_pos = hp - span, e_w = nph, e_span = span, e_q = C.ap0 & 0xFFFFu; // exact phrase: first word's position, word count, span
        if (field_queried(bmask, bh)) {
          if (near2) { // the operand's place in the node = the hit's m_uNodepos (SetNodePos, searchnode.cpp:3767-3786)
            e_span = 1u;
            emit = F.step_near2(hp, (bq & 0xFFFFu) == (C.ap0 & 0xFFFFu) ? 1u : 2u, bq & 0xFFFFu, px_dist, e_pos, e_w, e_q);
          } else
            emit = px_dist ? F.step_prox(hp, bq & 0xFFFFu, nph, C.ap0 & 0xFFFFu, span, px_dist, e_pos, e_w, e_span)
                           : F.step(hp, bq & 0xFFFFu, nph, C.ap0, C.ap1, C.ap2, C.ap3);
        }
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
          pcur = e_pos, pw = e_w, pspan = e_span, pq = e_q;
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
      X.update(C.ranker, C.dupes, pcur, pend_is_end, pq, pw, pspan, C.fw, C.nw, C.max_qpos);
      phave = false;
      continue;
    }
    if (best < 0) break;
    if (phase == 1) bmask = best == tl ? sm[0] : sm[1];
    // hits outside the keyword's own field limit never reach the ranker (AddHit, searchnode.cpp:3032-3043)
    if (field_queried(bmask, bh))
      X.update(C.ranker, C.dupes, bh & ~(1u << 23), ((bh >> 23) & 1u) != 0, bq & 0xFFFFu, 1u, 0u, C.fw, C.nw, C.max_qpos);
    { // advance the chosen stream: one decode for the wave, whatever stream each lane picked
      uint64_t ap = best == 0 ? sp[0] : best == 1 ? sp[1] : best == 2 ? sp[2] : sp[3];
      uint32_t ac = best == 0 ? sc[0] : best == 1 ? sc[1] : best == 2 ? sc[2] : sc[3];
      hit_advance(C.spp, ap, ac);
      if (C.termpos) {
        const uint32_t ak = best == 0 ? tpk[0] : best == 1 ? tpk[1] : best == 2 ? tpk[2] : tpk[3];
        const uint32_t am = best == 0 ? tpm[0] : best == 1 ? tpm[1] : best == 2 ? tpm[2] : tpm[3];
        while (ac && !tp_accept(ak, am, ac)) hit_advance(C.spp, ap, ac);
      }
      if (nn && (uint32_t)best == C.nn_a) notnear_filter(C.spp, ap, ac, nnp, nnc, (C.nn_b == 0 ? C.tm[0] : C.nn_b == 1 ? C.tm[1] : C.nn_b == 2 ? C.tm[2] : C.tm[3]), C.nn_dist);
#pragma unroll
      for (int t = 0; t < MAX_PROX_TERMS; ++t)
        if (t == best) sp[t] = ap, sc[t] = ac;
    }
  }
  if (rank) rk_out = X.finalize(C.ranker, C.nw, C.fw, C.n_qwords);
  if (F.over) atomicOr(C.flags, QF_FSM);
}

// One keyword's hit stream of one doc with a 16-byte read-ahead window: the hitlist (GetNextHit, sphinx.cpp:479-501 --
// delta VLB, 0-terminated) is decoded out of registers, and .spp is touched once per 12+ consumed bytes instead of once
// per hit.  A doc's hitlist is 2-3 bytes per hit, so for most docs ONE memory round trip serves the whole list -- the
// pass is a chain of dependent loads per doc, and the length of that chain is its cost.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// (plain scalars by reference, not a struct: the streams live in register arrays indexed by unrolled loops only)
__device__ __forceinline__ void hs_fill(const uint8_t* __restrict__ spp, uint64_t p, uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3, uint32_t& off) {
  const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(spp + (p & ~3ull)); // (.spp carries 64 zero bytes of slack)
  w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
  off = (uint32_t)p & 3u; // p - base, base = p rounded down to a dword
}
// p = .spp position of the next undecoded byte (0 = inlined hit / exhausted), cur = current Hitpos_t (0 = exhausted),
// w0..w3 = .spp bytes [base, base + 16), off = p - base
__device__ __forceinline__ void hs_advance(const uint8_t* __restrict__ spp, uint64_t& p, uint32_t& cur, uint32_t& w0, uint32_t& w1, uint32_t& w2,
                                           uint32_t& w3, uint32_t& off) {
  if (!p) {
    cur = 0;
    return;
  }
  if (off > 11u) hs_fill(spp, p, w0, w1, w2, w3, off); // fewer than 5 bytes left in the window
  const uint32_t i = off >> 2;                         // 0..2
  const uint32_t lo = i == 0 ? w0 : i == 1 ? w1 : w2, hi = i == 0 ? w1 : i == 1 ? w2 : w3;
  const uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
  // as read_vlb32: the first byte with a clear top bit ends the value; 7-bit groups, most significant first
  const uint32_t stop = ~x & 0x80808080u;
  const uint32_t n = stop ? ((uint32_t)__builtin_ctz(stop) >> 3) + 1u : 4u;
  const uint32_t all = ((x & 0x7Fu) << 21) | ((x & 0x7F00u) << 6) | ((x >> 9) & 0x3F80u) | ((x >> 24) & 0x7Fu);
  uint32_t d = all >> (7u * (4u - n)), len = n;
  if (!stop) { // 5-byte varint: its last byte is byte off + 4 <= 15 of the window
    const uint32_t j = (off + 4u) >> 2;
    const uint32_t wj = j == 1 ? w1 : j == 2 ? w2 : w3;
    d = (d << 7) | ((wj >> (8u * ((off + 4u) & 3u))) & 0x7Fu);
    len = 5;
  }
  p += len;
  off += len;
  if (!d) {
    p = 0;
    cur = 0;
  } else
    cur += d;
}

// The same pass for queries without PHRASE / PROXIMITY / BEFORE nodes and without position modifiers -- plain boolean trees
// under a state ranker, the bulk of hit-ranked traffic: the keywords' streams merged by (hitpos, qpos) straight into the
// ranker.  No word state machines, run trackers or acceptors: about half the registers of hit_pass, which is what
// lets rank_kernel keep many docs in flight per SIMD.
__device__ __forceinline__ int hit_rank_plain(const HitCtx& C, uint32_t ref0, uint32_t ref1, uint32_t ref2, uint32_t ref3, uint32_t smask) {
  uint64_t sp[MAX_PROX_TERMS];
  uint32_t sc[MAX_PROX_TERMS], sq[MAX_PROX_TERMS], sm[MAX_PROX_TERMS];
  uint32_t w0[MAX_PROX_TERMS], w1[MAX_PROX_TERMS], w2[MAX_PROX_TERMS], w3[MAX_PROX_TERMS], so[MAX_PROX_TERMS];
  // three rounds of independent loads: hit references + hitlist bases, then the windows, then decode
  uint32_t hv[MAX_PROX_TERMS];
  uint64_t hb[MAX_PROX_TERMS];
  bool lone[MAX_PROX_TERMS], on[MAX_PROX_TERMS];
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t) {
    on[t] = (uint32_t)t < C.nterms && ((smask >> t) & 1u);
    sq[t] = 0, sm[t] = 0, hv[t] = 0, hb[t] = 0, lone[t] = true;
    if (on[t]) {
      const uint32_t h = t == 0 ? ref0 : t == 1 ? ref1 : t == 2 ? ref2 : ref3;
      const uint32_t gblk = C.tb[t] + ((h >> 7) & 0xFFFFFFu), idx = h & 127u;
      sq[t] = C.tq[t];
      sm[t] = C.tm[t];
      lone[t] = (h >> 31) != 0; // the hit travelled in the doclist entry (SeekHitlist state 1, sphinx.cpp:461-464)
      hv[t] = C.hit[(uint64_t)gblk * DEVBLK + idx];
      if (!lone[t]) hb[t] = C.hbase[gblk];
    }
  }
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t) {
    sp[t] = 0, sc[t] = 0, w0[t] = w1[t] = w2[t] = w3[t] = 0, so[t] = 0;
    if (on[t]) {
      if (lone[t])
        sc[t] = hv[t];
      else {
        sp[t] = hb[t] + hv[t];
        hs_fill(C.spp, sp[t], w0[t], w1[t], w2[t], w3[t], so[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t)
    if (on[t] && !lone[t]) hs_advance(C.spp, sp[t], sc[t], w0[t], w1[t], w2[t], w3[t], so[t]);
  RankState X;
  X.reset();
  const uint32_t cmpmask = C.quorum_hits ? ~(1u << 23) : 0xFFFFFFFFu; // ExtQuorum_c sorts its hits without the end flag
  // MergeHits3 quirk (searchnode.cpp:3072-3077 + 3052-3054), as in hit_pass
  int phase = (C.multi_and && C.nterms == 3 && (sm[0] & sm[1] & sm[2]) != 0xFFFFFFFFu) ? 0 : 2, tl = 0, tr = 1;
  for (;;) {
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
      if (sc[t] && (best < 0 || (sc[t] & cmpmask) < (bh & cmpmask) || ((sc[t] & cmpmask) == (bh & cmpmask) && sq[t] < bq)))
        best = t, bh = sc[t], bq = sq[t], bmask = sm[t];
    if (best < 0) break;
    if (phase == 1) bmask = best == tl ? sm[0] : sm[1];
    // hits outside the keyword's own field limit never reach the ranker (AddHit, searchnode.cpp:3032-3043)
    if (field_queried(bmask, bh))
      X.update(C.ranker, C.dupes, bh & ~(1u << 23), ((bh >> 23) & 1u) != 0, bq & 0xFFFFu, 1u, 0u, C.fw, C.nw, C.max_qpos);
    // advance the chosen stream: one decode for the wave, whatever stream each lane picked
#define MRK_PICK(a) (best == 0 ? a[0] : best == 1 ? a[1] : best == 2 ? a[2] : a[3])
    uint64_t ap = MRK_PICK(sp);
    uint32_t ac = MRK_PICK(sc), a0 = MRK_PICK(w0), a1 = MRK_PICK(w1), a2 = MRK_PICK(w2), a3 = MRK_PICK(w3), ao = MRK_PICK(so);
#undef MRK_PICK
    hs_advance(C.spp, ap, ac, a0, a1, a2, a3, ao);
#pragma unroll
    for (int t = 0; t < MAX_PROX_TERMS; ++t)
      if (t == best) sp[t] = ap, sc[t] = ac, w0[t] = a0, w1[t] = a1, w2[t] = a2, w3[t] = a3, so[t] = ao;
  }
  return X.finalize(C.ranker, C.nw, C.fw, C.n_qwords);
}

// hit_rank_plain for the common case, specialized at compile time: NT keyword streams, the proximity family
// (SPH_RANK_PROXIMITY_BM25 / SPH_RANK_PROXIMITY: RankerState_Proximity_fn<.., false>::Update, sphinxsearch.cpp:1351-1367 --
// LCS per field in BYTE arithmetic), no repeated keywords, no quorum hit order, no MergeHits3 field-test quirk.  Same
// results, about half the instructions per hit: the stream state is a window + a 32-bit offset (the 64-bit .spp position
// is only touched at a refill), the ranker update is the three lines it is for this family, and the loop carries no
// per-ranker branches.  rank_kernel is bound by instruction issue (vector and scalar), not by memory.
#ifndef MRK_RKEXP
#define MRK_RKEXP 0
#endif
template <int NT>
__device__ __forceinline__ int hit_rank_prox(const HitCtx& C, uint32_t ref0, uint32_t ref1, uint32_t ref2, uint32_t ref3, uint32_t smask) {
  uint64_t sb[NT];                                     // .spp position of the window's first byte
  uint32_t so[NT], sc[NT], w0[NT], w1[NT], w2[NT], w3[NT]; // offset of the next undecoded byte (~0 = no hitlist left), current hit
  uint32_t hv[NT];
  uint64_t hb[NT];
  bool lone[NT], on[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    on[t] = ((smask >> t) & 1u) != 0;
    hv[t] = 0, hb[t] = 0, lone[t] = true;
    if (on[t]) {
      const uint32_t h = t == 0 ? ref0 : t == 1 ? ref1 : t == 2 ? ref2 : ref3;
      const uint32_t gblk = C.tb[t] + ((h >> 7) & 0xFFFFFFu), idx = h & 127u;
      lone[t] = (h >> 31) != 0;
      hv[t] = C.hit[(uint64_t)gblk * DEVBLK + idx];
      if (!lone[t]) hb[t] = C.hbase[gblk];
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    sb[t] = 0, so[t] = 0xFFFFFFFFu, sc[t] = 0xFFFFFFFFu, w0[t] = w1[t] = w2[t] = w3[t] = 0; // (current hit ~0: the stream is dry)
    if (on[t]) {
      if (lone[t])
        sc[t] = hv[t];
      else {
        const uint64_t p = hb[t] + hv[t];
        sc[t] = 0;
        sb[t] = p & ~3ull;
        const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(C.spp + sb[t]);
        w0[t] = v.x, w1[t] = v.y, w2[t] = v.z, w3[t] = v.w;
        so[t] = (uint32_t)p & 3u;
      }
    }
  }
  // next hit of a stream out of its window (GetNextHit, sphinx.cpp:479-501); refills when fewer than 5 bytes are left.
  // Written without branches but for the two rare ones (refill, 5-byte varint): the lanes of a wave sit on different
  // streams and offsets, and every divergent branch costs the wave both sides plus the exec-mask bookkeeping.
  auto next = [&](uint64_t& b, uint32_t& o, uint32_t& cur, uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    const bool dry = o == 0xFFFFFFFFu; // (a lone hit's stream, consumed)
    if (!dry && o > 11u) {
      b += o & ~3u;
      const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(C.spp + b);
      x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
      o &= 3u;
    }
    const uint32_t i = (o >> 2) & 3u;
    const uint32_t lo = i == 0 ? x0 : i == 1 ? x1 : x2, hi = i == 0 ? x1 : i == 1 ? x2 : x3;
    const uint32_t x = __builtin_amdgcn_alignbyte(hi, lo, o & 3u);
    const uint32_t stop = ~x & 0x80808080u;
    const uint32_t n = stop ? ((uint32_t)__builtin_ctz(stop) >> 3) + 1u : 4u;
    const uint32_t all = ((x & 0x7Fu) << 21) | ((x & 0x7F00u) << 6) | ((x >> 9) & 0x3F80u) | ((x >> 24) & 0x7Fu);
    uint32_t d = all >> (7u * (4u - n)), len = n;
    if (!dry && !stop) {
      const uint32_t j = (o + 4u) >> 2;
      const uint32_t wj = j == 1 ? x1 : j == 2 ? x2 : x3;
      d = (d << 7) | ((wj >> (8u * ((o + 4u) & 3u))) & 0x7Fu);
      len = 5;
    }
    const bool end = dry || d == 0;
    o = end ? 0xFFFFFFFFu : o + len;
    cur = end ? 0xFFFFFFFFu : cur + d;
  };
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (on[t] && !lone[t]) next(sb[t], so[t], sc[t], w0[t], w1[t], w2[t], w3[t]);
  // m_uLCS[field]: the merged stream is ordered by (field, position), so one field is open at a time -- its best LCS is kept in
  // a register and weighed in when the field changes (fields past the weights' count add nothing, as in Finalize)
  uint32_t cur_lcs = 0, cur_f = 0xFFFFFFFFu, f_best = 0;
  int exp_delta = -1, last_pwf = -1, rk = 0;
  for (;;) {
    // the least (hitpos, qpos) among the live streams
    uint32_t bh = 0xFFFFFFFFu, bq = 0xFFFFFFFFu;
    int best = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint32_t h = sc[t];
      const bool less = h < bh || (h == bh && C.tq[t] < bq);
      if (t == 0 || less) best = t, bh = h, bq = C.tq[t];
    }
    if (bh == 0xFFFFFFFFu) break;
#if MRK_RKEXP == 1 // ablation: the loop stops after one hit
    if (cur_f != 0xFFFFFFFFu) break;
#endif
    const uint32_t bm = best == 0 ? C.tm[0] : best == 1 ? C.tm[NT > 1 ? 1 : 0] : best == 2 ? C.tm[NT > 2 ? 2 : 0] : C.tm[NT > 3 ? 3 : 0];
    { // hits outside the keyword's own field limit never reach the ranker (AddHit, searchnode.cpp:3032-3043): predicated, not branched
      const uint32_t hp = bh & ~(1u << 23), f = hp >> 24;
      const bool q = f < 32u ? ((bm >> (f & 31u)) & 1u) != 0 : bm == 0xFFFFFFFFu;
      const int pwf = (int)hp, delta = pwf - (int)(bq & 0xFFFFu);
      const uint32_t grown = (((delta == exp_delta) ? cur_lcs : 0u) + 1u) & 0xffu;
      cur_lcs = (q && pwf > last_pwf) ? grown : cur_lcs;
      if (q && f != cur_f) { // (rare: once per field of the doc)
        if (cur_f < C.nw) rk += (int)f_best * C.fw[cur_f];
        cur_f = f, f_best = 0;
      }
      f_best = (q && cur_lcs > f_best) ? cur_lcs : f_best;
      last_pwf = q ? pwf : last_pwf;
      exp_delta = q ? delta : exp_delta;
    }
    // advance the stream the hit came from: its state is selected into temporaries, stepped ONCE and selected back (the
    // lanes of a wave sit on different streams: NT predicated copies of the decoder cost NT times its instructions)
    {
      uint64_t xb = sb[0];
      uint32_t xo = so[0], xc = sc[0], x0 = w0[0], x1 = w1[0], x2 = w2[0], x3 = w3[0];
#pragma unroll
      for (int t = 1; t < NT; ++t) { // (one select per value: a block of assignments under `if` comes out as a branch)
        const bool me = t == best;
        xb = me ? sb[t] : xb, xo = me ? so[t] : xo, xc = me ? sc[t] : xc;
        x0 = me ? w0[t] : x0, x1 = me ? w1[t] : x1, x2 = me ? w2[t] : x2, x3 = me ? w3[t] : x3;
      }
      next(xb, xo, xc, x0, x1, x2, x3);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool me = t == best;
        sb[t] = me ? xb : sb[t], so[t] = me ? xo : so[t], sc[t] = me ? xc : sc[t];
        w0[t] = me ? x0 : w0[t], w1[t] = me ? x1 : w1[t], w2[t] = me ? x2 : w2[t], w3[t] = me ? x3 : w3[t];
      }
    }
  }
  if (cur_f < C.nw) rk += (int)f_best * C.fw[cur_f];
  return rk;
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

// ExtNotNear_c::GetDocsChunk for a doc both keywords hold: it stays iff a must-hit (inside the must keyword's field limit)
// survives the filter.  ref_a / ref_b = where the doc sits in the two keywords' packed arrays (as for hit_pass).
__device__ __forceinline__ bool notnear_any(const HitCtx& C, uint32_t ref_a, uint32_t ref_b) {
  uint64_t p[2] = {0, 0};
  uint32_t c[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t slot = i ? C.nn_b : C.nn_a, ref = i ? ref_b : ref_a;
    const uint32_t tbs = slot == 0 ? C.tb[0] : slot == 1 ? C.tb[1] : slot == 2 ? C.tb[2] : C.tb[3];
    const uint32_t gblk = tbs + ((ref >> 7) & 0xFFFFFFu), idx = ref & 127u;
    const uint32_t hv = C.hit[(uint64_t)gblk * DEVBLK + idx];
    if (ref >> 31)
      c[i] = hv;
    else {
      p[i] = C.hbase[gblk] + hv;
      hit_advance(C.spp, p[i], c[i]);
    }
  }
  const uint32_t mask_a = C.nn_a == 0 ? C.tm[0] : C.nn_a == 1 ? C.tm[1] : C.nn_a == 2 ? C.tm[2] : C.tm[3];
  const uint32_t mask_b = C.nn_b == 0 ? C.tm[0] : C.nn_b == 1 ? C.tm[1] : C.nn_b == 2 ? C.tm[2] : C.tm[3];
  for (;;) {
    notnear_filter(C.spp, p[0], c[0], p[1], c[1], mask_b, C.nn_dist);
    if (!c[0]) return false;
    if (field_queried(mask_a, c[0])) return true;
    hit_advance(C.spp, p[0], c[0]);
  }
}

} // namespace mrk
