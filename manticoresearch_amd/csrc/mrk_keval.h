// mrk_keval.h -- the generic per-doc evaluator: query shapes the specialised hit pass (mrk_khits.h) does not take -- more
// than four keywords under a hit ranker, phrases of five and more words, BEFORE / NEAR / NOTNEAR over phrases, groups and
// quorums, several such nodes in one query.  gfx950 / wave64, one candidate doc per lane.
//
// The scan kernel has already intersected the posting lists: a candidate is a doc the tree CAN match (every PHRASE /
// PROXIMITY / NEAR / BEFORE node taken as the AND of its words, NOTNEAR / ANDNOT as their left side), with one reference
// per keyword into the packed arrays.  Here each node of the reference's evaluation tree (ExtNode_i::Create,
// searchnode.cpp:1599-1811) is evaluated bottom-up for that one doc: does it hold the doc, its field mask, its tfidf, and
// its hit list -- what GetDocsChunk / GetHits of that node would hand its parent -- materialised in a per-lane arena in
// HBM (a shared spill area takes lists the lane's slice cannot).  The root's hits feed the state ranker.  Slow next to the
// specialised passes (every list makes a round trip through memory), complete in exchange, and only used for the shapes
// they decline.
//   ExtTerm_T / ExtTermPos_T   :1876-2020, 2259-2405     ExtMultiAnd_T (hits: MergeHits2/3/N)   :2716-3223
//   ExtAnd_c                   :2570-2706                ExtOr_c / ExtMaybe_c / ExtAndNot_c     :3465-3694
//   ExtNWay_T<FSMphrase_c>     :3792-3953                <FSMproximity_c> :3958-4075             <FSMmultinear_c> :4080-4318
//   ExtQuorum_c                :4342-4617                ExtOrder_c :4657-4936                   ExtNotNear_c :5325-5478
#pragma once
#include "mrk_khits.h"
#include "mrk_kprune.h"

namespace mrk {

constexpr uint32_t GEN_NOREF = 0xFFFFFFFFu; // the doc does not hold the keyword
constexpr int GEN_FSM_STATES = 32;          // live FSMphrase_c states / FSMproximity_c slots per doc

struct GenAlloc {
  GenHit* lane;
  uint32_t cap, used;
  GenHit* spill;
  unsigned long long spill_cap;
  unsigned long long* spill_used;
  bool failed;
  __device__ __forceinline__ GenHit* take(uint32_t n) {
    if (used + n <= cap) {
      GenHit* p = lane + used;
      used += n;
      return p;
    }
    const unsigned long long off = atomicAdd(spill_used, (unsigned long long)n);
    if (off + n > spill_cap) {
      failed = true;
      return nullptr;
    }
    return spill + off;
  }
};

struct GenRes { // one node, one doc
  GenHit* p;
  uint32_t n;
  float tfidf;
  uint32_t fields;
  bool ok;
};

__device__ __forceinline__ uint32_t gen_pwf(uint32_t hitpos) { return hitpos & ~(1u << 23); } // HITMAN::GetPosWithField

// the doc's entry in a keyword's packed arrays: hit count, field mask, its first hit
__device__ __forceinline__ void gen_open(const DevSegment& seg, const DevTerm& T, uint32_t ref, uint32_t row, uint64_t& sp, uint32_t& sc, uint32_t& tf,
                                         uint32_t& fl) {
  const uint32_t gblk = T.blk_first + ((ref >> 7) & 0xFFFFFFu), idx = ref & 127u;
  const uint32_t aw = seg.pk_attr[(uint64_t)gblk * 64 + (idx & 63u)], sh = (idx >> 6) * 8u;
  tf = (aw >> sh) & 0xffu;
  fl = (aw >> (16u + sh)) & 0xffu;
  if (tf == 255u) tf = exc_tf(seg, T, row);
  const uint32_t hv = seg.pk_hit[(uint64_t)gblk * DEVBLK + idx];
  sp = 0, sc = 0;
  if (ref >> 31) // the hit travelled in the doclist entry
    sc = hv;
  else {
    sp = seg.pk_hbase[gblk] + hv;
    hit_advance(seg.spp, sp, sc);
  }
}

__device__ __forceinline__ GenHit gen_hit(uint32_t hitpos, uint32_t qpos, uint32_t nodepos, uint32_t spanlen, uint32_t matchlen, uint32_t weight) {
  GenHit h;
  h.hitpos = hitpos, h.qpos = (uint16_t)qpos, h.nodepos = (uint16_t)nodepos, h.spanlen = (uint16_t)spanlen, h.matchlen = (uint16_t)matchlen, h.weight = weight;
  return h;
}

// ExtTerm_T (+ ExtConditional_T for a position modifier: the doc stays as the term emitted it -- fields, tfidf of ALL its
// hits -- once one hit is acceptable; only the acceptable hits travel on)
__device__ inline void gen_term(const DevSegment& seg, const DevTerm& T, uint32_t ref, uint32_t row, GenAlloc& A, GenRes& R) {
  R.p = nullptr, R.n = 0, R.tfidf = 0.0f, R.fields = 0, R.ok = false;
  if (ref == GEN_NOREF) return;
  uint64_t sp;
  uint32_t sc, tf, fl;
  gen_open(seg, T, ref, row, sp, sc, tf, fl);
  GenHit* out = A.take(tf);
  if (!out) return;
  uint32_t n = 0;
  while (sc) {
    if (n < tf && field_queried(T.queried32, sc) && tp_accept(T.tp_kind, T.tp_max, sc)) out[n++] = gen_hit(sc, T.qpos, 0, 1, 1, 1);
    hit_advance(seg.spp, sp, sc);
  }
  if (T.tp_kind && !n) return;
  R.p = out, R.n = n;
  R.tfidf = term_tfidf(tf, T.idf);
  R.fields = fl & T.queried32;
  R.ok = true;
}

// IsHitLess (searchnode.cpp:2611-2615)
__device__ __forceinline__ bool gen_hit_less(const GenHit& a, const GenHit& b) { return a.hitpos < b.hitpos || (a.hitpos == b.hitpos && a.qpos <= b.qpos); }

// ExtAnd_c / ExtOr_c::CollectHits: two position-sorted lists into one; a non-zero nodepos relabels that side's hits
__device__ inline void gen_merge(GenHit* out, uint32_t& n_out, const GenHit* L, uint32_t nl, const GenHit* Rr, uint32_t nr, uint32_t npl, uint32_t npr) {
  uint32_t i = 0, j = 0, n = 0;
  while (i < nl || j < nr) {
    const bool left = j >= nr || (i < nl && gen_hit_less(L[i], Rr[j]));
    GenHit h = left ? L[i++] : Rr[j++];
    const uint32_t np = left ? npl : npr;
    if (np) h.nodepos = (uint16_t)np;
    out[n++] = h;
  }
  n_out = n;
}

// One doc through the program.  refs[k] = the doc's place in keyword slot k's packed arrays (GEN_NOREF: not there).
// Returns false when the root does not hold the doc; else tfidf / fields of the root and, for the state rankers, rk.
__device__ inline bool gen_eval(const DevSegment& seg, const DevQuery* __restrict__ Q, const GenProg* __restrict__ P, const uint32_t* refs, uint32_t row,
                                GenAlloc& A, bool state_ranker, bool dupes, const int32_t* fw, uint32_t nw, uint32_t* qflags, float& tfidf_out,
                                uint32_t& fields_out, int& rk_out, uint32_t near_fq_in = 65535u, uint32_t* near_m_out = nullptr) {
  GenRes res[GEN_MAX_NODES];
  A.used = 0;
  const uint32_t nn = P->n_nodes;
  for (uint32_t ni = 0; ni < nn; ++ni) {
    const GenNode N = P->nodes[ni];
    GenRes R;
    R.p = nullptr, R.n = 0, R.tfidf = 0.0f, R.fields = 0, R.ok = false;
    switch (N.kind) {
      case GN_TERM: gen_term(seg, Q->t[N.kid[0]], refs[N.kid[0]], row, A, R); break;
      case GN_MULTIAND: {
        // ExtMultiAnd_T: every keyword holds the doc; tfidf adds up in node order; the hits are the keywords' streams merged by
        // (position, query position).  The MergeHits3 quirk is kept: once one of three streams runs dry the two-stream merge tests
        // fields against nodes 0 and 1, whichever streams are left (:3072-3077, 3052-3054)
        const uint32_t k = N.n_kids;
        uint64_t sp[MRK_MAX_AND_TERMS];
        uint32_t sc[MRK_MAX_AND_TERMS];
        bool all = true;
        for (uint32_t i = 0; i < k; ++i) all = all && refs[N.kid[i]] != GEN_NOREF;
        if (!all) break;
        uint32_t total = 0, mask = 0;
        float t = 0.0f;
        for (uint32_t i = 0; i < k; ++i) {
          const DevTerm& T = Q->t[N.kid[i]];
          uint32_t tf, fl;
          gen_open(seg, T, refs[N.kid[i]], row, sp[i], sc[i], tf, fl);
          total += tf;
          mask |= fl & T.queried32;
          t += term_tfidf(tf, T.idf);
        }
        GenHit* out = A.take(total);
        if (!out) break;
        const bool test_fields = (N.flags & 2u) != 0;
        int phase = (k == 3 && test_fields) ? 0 : 2;
        uint32_t tl = 0, tr = 1, n = 0;
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
          if (phase == 1 && !(sc[tl] && sc[tr])) phase = 2;
          int best = -1;
          for (uint32_t i = 0; i < k; ++i) {
            if (!sc[i]) continue;
            if (best < 0 || sc[i] < sc[best] || (sc[i] == sc[best] && (Q->t[N.kid[i]].qpos & 0xFFFFu) < (Q->t[N.kid[best]].qpos & 0xFFFFu))) best = (int)i;
          }
          if (best < 0) break;
          const DevTerm& T = Q->t[N.kid[best]];
          uint32_t fmask = T.queried32;
          if (phase == 1) fmask = Q->t[N.kid[(uint32_t)best == tl ? 0 : 1]].queried32;
          if ((!test_fields || field_queried(fmask, sc[best])) && n < total) out[n++] = gen_hit(sc[best], T.qpos, N.aux[best], 1, 1, 1);
          hit_advance(seg.spp, sp[best], sc[best]);
        }
        R.p = out, R.n = n, R.tfidf = t, R.fields = mask, R.ok = true;
        break;
      }
      case GN_AND: {
        const GenRes &L = res[N.kid[0]], &Rr = res[N.kid[1]];
        if (!(L.ok && Rr.ok)) break;
        R.ok = true;
        R.fields = L.fields | Rr.fields;
        R.tfidf = L.tfidf + Rr.tfidf;
        if (!L.n || !Rr.n) break; // CollectHits only emits a doc's hits once both sides have one (:2639-2702)
        GenHit* out = A.take(L.n + Rr.n);
        if (!out) break;
        gen_merge(out, R.n, L.p, L.n, Rr.p, Rr.n, N.npl, N.npr);
        R.p = out;
        if (N.flags & 1u) { // m_bQPosReverse: the hits sorted again, by position and DESCENDING query position (CmpAndHitReverse_fn
          // :2618-2624).  A sort, not a fix-up of ties: a PROXIMITY operand may hand over hits out of position order
          for (uint32_t i = 1; i < R.n; ++i) {
            const GenHit x = out[i];
            uint32_t j = i;
            while (j > 0 && (out[j - 1].hitpos > x.hitpos || (out[j - 1].hitpos == x.hitpos && out[j - 1].qpos < x.qpos))) {
              out[j] = out[j - 1];
              --j;
            }
            if (j != i) out[j] = x;
          }
        }
        break;
      }
      case GN_OR:
      case GN_MAYBE: {
        const GenRes &L = res[N.kid[0]], &Rr = res[N.kid[1]];
        if (N.kind == GN_MAYBE ? !L.ok : !(L.ok || Rr.ok)) break;
        R.ok = true;
        if (L.ok && Rr.ok) {
          R.fields = L.fields | Rr.fields;
          R.tfidf = L.tfidf + Rr.tfidf;
          GenHit* out = A.take(L.n + Rr.n);
          if (!out) break;
          gen_merge(out, R.n, L.p, L.n, Rr.p, Rr.n, 0, 0);
          R.p = out;
        } else {
          const GenRes& S = L.ok ? L : Rr;
          R.fields = S.fields, R.tfidf = S.tfidf, R.p = S.p, R.n = S.n;
        }
        break;
      }
      case GN_ANDNOT: {
        const GenRes &L = res[N.kid[0]], &Rr = res[N.kid[1]];
        if (!L.ok || Rr.ok) break;
        R = L;
        break;
      }
      case GN_NOTNEAR: {
        // ExtNotNear_c: the must side's doc; where the not side holds it too only the must hits that no later not-hit comes
        // within the distance of survive (FilterHits :5352-5380), and the doc stays iff one does
        const GenRes &L = res[N.kid[0]], &Rr = res[N.kid[1]];
        if (!L.ok) break;
        R = L;
        if (!Rr.ok) break;
        GenHit* out = A.take(L.n);
        if (!out) {
          R.ok = false;
          break;
        }
        uint32_t n = 0, j = 0;
        for (uint32_t i = 0; i < L.n; ++i) {
          const uint32_t pm = gen_pwf(L.p[i].hitpos);
          while (j < Rr.n && gen_pwf(Rr.p[j].hitpos) < pm) ++j;
          if (j >= Rr.n || pm + L.p[i].matchlen - 1u + (uint32_t)N.opt < gen_pwf(Rr.p[j].hitpos)) out[n++] = L.p[i];
        }
        R.p = out, R.n = n;
        R.ok = n != 0;
        break;
      }
      case GN_UNIT: {
        // ExtUnit_c, SENTENCE / PARAGRAPH (searchnode.cpp:4983-5310): both arguments hold the doc; where it also holds boundary hits
        // ("dots", the index_sp keyword, limited to the node's fields), only hit pairs no dot separates match and the hits of
        // every matching unit are copied (FilterHits :5082-5166); positions compare raw, as Hitpos_t does
        const GenRes &L = res[N.kid[0]], &Rr = res[N.kid[1]];
        if (!(L.ok && Rr.ok)) break;
        GenRes D;
        D.p = nullptr, D.n = 0, D.ok = false;
        if (N.aux[0] != 0xFFu) gen_term(seg, Q->t[N.aux[0]], refs[N.aux[0]], row, A, D);
        if (A.failed) break;
        GenHit* out = A.take(L.n + Rr.n);
        if (!out) break;
        const uint32_t nd = D.ok ? D.n : 0u;
        uint32_t i1 = 0, i2 = 0, id = 0, n = 0, end = nd ? 0u : 0xFFFFFFFFu;
        for (;;) {
          if (end) { // in a matched unit: copy hits up to the next dot
            const bool v1 = i1 < L.n && L.p[i1].hitpos < end, v2 = i2 < Rr.n && Rr.p[i2].hitpos < end;
            if (!v1 && !v2) {
              end = 0;
              if (i1 < L.n && i2 < Rr.n) continue;
              break;
            }
            out[n++] = (v1 && (!v2 || gen_hit_less(L.p[i1], Rr.p[i2]))) ? L.p[i1++] : Rr.p[i2++];
          } else {
            if (i1 >= L.n || i2 >= Rr.n) break;
            const uint32_t a = L.p[i1].hitpos, b = Rr.p[i2].hitpos, umin = a < b ? a : b, umax = a < b ? b : a;
            while (id < nd && D.p[id].hitpos <= umin) ++id;
            if (id >= nd) { // no dot past the pair's start: a match, copy to the doc's end
              end = 0xFFFFFFFFu;
              continue;
            }
            const uint32_t dp = D.p[id].hitpos;
            if (dp < umax) { // "A dot B": both sides move past this dot
              while (i1 < L.n && L.p[i1].hitpos <= dp) ++i1;
              if (i1 >= L.n) break;
              while (i2 < Rr.n && Rr.p[i2].hitpos <= dp) ++i2;
              if (i2 >= Rr.n) break;
              continue;
            }
            while (id < nd && D.p[id].hitpos <= umax) ++id;
            end = id >= nd ? 0xFFFFFFFFu : D.p[id].hitpos;
          }
        }
        if (!n) break;
        R.p = out, R.n = n, R.ok = true, R.fields = L.fields | Rr.fields, R.tfidf = L.tfidf + Rr.tfidf;
        break;
      }
      case GN_PHRASE: {
        // ExtNWay_T<FSMphrase_c>: the words' AND chain (kid 0) through the phrase state machine (HitFSM :3901-3947)
        const GenRes& I = res[N.kid[0]];
        if (!I.ok || !I.n) break;
        GenHit* out = A.take(I.n);
        if (!out) break;
        const uint32_t k = N.n_words;
        uint32_t atoms[MRK_MAX_AND_TERMS];
        for (uint32_t i = 0; i < k; ++i) atoms[i] = Q->t[N.aux[i]].qpos & 0xFFFFu;
        uint32_t st_exp[GEN_FSM_STATES], st_tag[GEN_FSM_STATES], ns = 0, n = 0, ffield = 0;
        for (uint32_t hi = 0; hi < I.n; ++hi) {
          const GenHit h = I.p[hi];
          const uint32_t hpf = gen_pwf(h.hitpos);
          if (h.qpos == atoms[0]) {
            if (ns == (uint32_t)GEN_FSM_STATES) {
              atomicOr(qflags, QF_FSM);
              break;
            }
            st_tag[ns] = 0, st_exp[ns] = hpf + (atoms[1] - atoms[0]);
            ++ns;
          }
          for (int i = (int)ns - 1; i >= 0; --i) {
            if (st_exp[i] < hpf) {
              --ns;
              st_exp[i] = st_exp[ns], st_tag[i] = st_tag[ns]; // RemoveFast
              continue;
            }
            if (st_exp[i] == hpf && atoms[st_tag[i] + 1] == h.qpos) {
              const uint32_t tg = ++st_tag[i];
              st_exp[i] = tg + 1 < k ? hpf + (atoms[tg + 1] - atoms[tg]) : hpf - 0x7FFFFFFFu;
            }
            if (st_tag[i] == k - 1) {
              const uint32_t span = atoms[k - 1] - atoms[0];
              if (!n) ffield = h.hitpos >> 24;
              out[n++] = gen_hit(hpf - span, atoms[0], 0, span + 1, span + 1, k);
              ns = 0; // ResetFSM
              break;
            }
          }
        }
        if (!n) break;
        R.p = out, R.n = n, R.ok = true, R.tfidf = I.tfidf, R.fields = 1u << (ffield & 31u);
        break;
      }
      case GN_PROX: {
        // ExtNWay_T<FSMproximity_c> (HitFSM :3973-4065): all words within qlen + distance fold into one hit; its weight
        // counts the words that keep the query's relative offsets
        const GenRes& I = res[N.kid[0]];
        if (!I.ok || !I.n) break;
        GenHit* out = A.take(I.n);
        if (!out) break;
        const uint32_t k = N.n_words;
        const uint32_t min_qpos = Q->t[N.aux[0]].qpos & 0xFFFFu, qlen = (Q->t[N.aux[k - 1]].qpos & 0xFFFFu) - min_qpos, dist = (uint32_t)N.opt;
        if (qlen + 1 > (uint32_t)GEN_FSM_STATES) {
          atomicOr(qflags, QF_FSM);
          break;
        }
        uint32_t prox[GEN_FSM_STATES];
        int deltas[GEN_FSM_STATES];
        for (uint32_t i = 0; i <= qlen; ++i) prox[i] = 0xFFFFFFFFu;
        uint32_t exp_pos = 0, words = 0, n = 0, ffield = 0;
        int min_qindex = -1;
        const int nq = (int)qlen + 1;
        for (uint32_t hi = 0; hi < I.n; ++hi) {
          const GenHit h = I.p[hi];
          const int qindex = (int)h.qpos - (int)min_qpos;
          uint32_t hpf = gen_pwf(h.hitpos);
          if (qindex < 0 || qindex >= nq) continue; // (cannot happen: the inner chain holds the phrase's words only)
          if (prox[qindex] == 0xFFFFFFFFu) ++words;
          prox[qindex] = hpf;
          if (hpf >= exp_pos || qindex == min_qindex) {
            min_qindex = qindex;
            const int min_pos = (int)(hpf - qlen - dist);
            for (int i = 0; i < nq; ++i)
              if (prox[i] != 0xFFFFFFFFu) {
                if ((int)prox[i] <= min_pos) {
                  prox[i] = 0xFFFFFFFFu;
                  --words;
                  continue;
                }
                if (prox[i] < hpf) {
                  min_qindex = i;
                  hpf = prox[i];
                }
              }
            exp_pos = prox[min_qindex] + qlen + dist;
          }
          if (words != k) continue;
          uint32_t umax = 0;
          for (int i = 0; i < nq; ++i)
            if (prox[i] != 0xFFFFFFFFu) {
              deltas[i] = (int)(prox[i] - (uint32_t)i);
              if (prox[i] > umax) umax = prox[i];
            } else
              deltas[i] = 0x7FFFFFFF;
          for (int i = 1; i < nq; ++i)
            for (int j = i; j > 0 && deltas[j - 1] > deltas[j]; --j) {
              const int x = deltas[j];
              deltas[j] = deltas[j - 1];
              deltas[j - 1] = x;
            }
          uint32_t cur_w = 0, w = 0;
          int last = -0x7FFFFFFF;
          for (int i = 0; i < nq && deltas[i] != 0x7FFFFFFF; ++i) {
            if (deltas[i] == last)
              ++cur_w;
            else {
              w += cur_w ? 1u + cur_w : 0u;
              cur_w = 0;
            }
            last = deltas[i];
          }
          w += cur_w ? 1u + cur_w : 0u;
          if (!w) w = 1;
          const uint32_t pm = prox[min_qindex], span = umax - pm + 1u;
          if (!n) ffield = h.hitpos >> 24;
          out[n++] = gen_hit(pm, min_qpos, 0, span, span, w);
          prox[min_qindex] = 0xFFFFFFFFu;
          min_qindex = -1;
          --words;
          exp_pos = 0;
        }
        if (!n) break;
        R.p = out, R.n = n, R.ok = true, R.tfidf = I.tfidf, R.fields = 1u << (ffield & 31u);
        break;
      }
      case GN_NEAR:
        if (N.n_words > 2) {
          // ExtNWay_T<FSMmultinear_c>, three and more operands (HitFSM :4096-4288, the ring branches): m_dNpos = the operand
          // numbers gathered so far (sorted), m_dRing = their hits in arrival order; a complete chain resets.  m_uFirstQpos is
          // never reset in the reference -- it is the least query position any EARLIER doc of the node inserted -- so it
          // comes in from outside (near_fq_in: the probe launch found it) and this doc's own least goes out (near_m_out).
          const GenRes& I = res[N.kid[0]];
          if (!I.ok) break;
          GenHit* out = A.take(I.n);
          if (!out) break;
          constexpr int NR = 16;
          const uint32_t k = N.n_words, dist = (uint32_t)N.opt;
          uint32_t last_p = 0, last_ml = 0, first_hit = 0, weight = 0, first_qpos = near_fq_in, m_ins = 65535u, n = 0, ffield = 0;
          uint32_t np[NR], nnp = 0, iring = 0;
          GenHit ring[MRK_MAX_AND_TERMS];
          for (uint32_t i = 0; i < k && i < (uint32_t)MRK_MAX_AND_TERMS; ++i) ring[i] = gen_hit(0, 0, 0, 0, 0, 0);
          auto tail = [&]() -> uint32_t { return (iring + nnp - 1u) % k; };
          auto find = [&](uint32_t v) -> int {
            for (uint32_t i = 0; i < nnp; ++i)
              if (np[i] == v) return (int)i;
            return -1;
          };
          auto insert = [&](uint32_t at, uint32_t v) -> bool {
            if (nnp >= (uint32_t)NR) return false;
            for (uint32_t i = nnp; i > at; --i) np[i] = np[i - 1];
            np[at] = v;
            ++nnp;
            return true;
          };
          auto seen_q = [&](uint32_t q) {
            if (q < first_qpos) first_qpos = q;
            if (q < m_ins) m_ins = q;
          };
          bool bad = false;
          for (uint32_t hi = 0; hi < I.n && !bad; ++hi) {
            const GenHit h = I.p[hi];
            const uint32_t hpf = gen_pwf(h.hitpos), npos = h.nodepos, qpos = h.qpos;
            if (last_p == hpf) { // a dupe hit (an OR operand, 'a NEAR/2 a NEAR/2 a'): the leftmost operand of the dupes takes the ring slot
              if (npos < ring[tail()].nodepos && find(npos) < 0) {
                const int at = find(ring[tail()].nodepos);
                if (at >= 0) np[at] = npos;
                for (uint32_t i = 1; i < nnp; ++i)
                  for (uint32_t j = i; j > 0 && np[j - 1] > np[j]; --j) {
                    const uint32_t x = np[j];
                    np[j] = np[j - 1];
                    np[j - 1] = x;
                  }
                ring[tail()].nodepos = (uint16_t)npos, ring[tail()].qpos = (uint16_t)qpos;
              }
              continue; // (the pre-last roll-back only exists for two operands)
            }
            if (last_p == 0 || last_p + last_ml + dist <= hpf) { // probably a new chain
              first_hit = last_p = hpf;
              last_ml = h.matchlen;
              weight = h.weight;
              nnp = 1, np[0] = npos;
              ring[tail()] = h;
              continue;
            }
            if (npos < np[0]) {
              seen_q(qpos);
              bad = !insert(0, npos);
            } else if (npos > np[nnp - 1]) {
              seen_q(qpos);
              bad = !insert(nnp, npos);
            } else if (npos != np[0] && npos != np[nnp - 1]) {
              int end = (int)nnp, start = 0;
              bool drop = false;
              while (end - start > 1) {
                const int mid = (start + end) / 2;
                if (npos == np[mid]) {
                  const GenHit rh = ring[iring];
                  if (npos == rh.nodepos) { // the last addition is the same operand as the first: shift
                    weight -= rh.weight;
                    first_hit = gen_pwf(rh.hitpos);
                    if (++iring == k) iring = 0;
                  } else if (npos == ring[tail()].nodepos)
                    weight -= ring[tail()].weight;
                  else {
                    drop = true;
                    break;
                  }
                }
                if (npos < np[mid])
                  end = mid;
                else
                  start = mid;
              }
              if (drop) continue;
              bad = !insert((uint32_t)end, npos);
              seen_q(qpos);
            } else if (npos == ring[iring].nodepos) { // the same operand as the head: shift
              weight -= ring[iring].weight;
              first_hit = gen_pwf(ring[iring].hitpos);
              if (++iring == k) iring = 0;
            } else if (npos == ring[tail()].nodepos) // ... as the tail: the tail moves onto it
              weight -= ring[tail()].weight;
            else
              continue;
            if (bad) break;
            weight += h.weight;
            last_ml = h.matchlen;
            ring[tail()] = h;
            if (k == nnp) { // the whole chain: emit it (no overlapping in generic chains)
              if (!n) ffield = h.hitpos >> 24;
              out[n++] = gen_hit(first_hit, first_qpos < qpos ? first_qpos : qpos, 0, nnp, hpf - first_hit + last_ml, weight);
              last_p = 0;
              continue;
            }
            last_p = hpf;
          }
          if (bad) atomicOr(qflags, QF_FSM);
          if (near_m_out) *near_m_out = m_ins;
          if (!n) break;
          R.p = out, R.n = n, R.ok = true, R.tfidf = I.tfidf, R.fields = 1u << (ffield & 31u);
          break;
        } else {
        // ExtNWay_T<FSMmultinear_c> for TWO operands of any kind (HitFSM :4096-4288, the twofer branches): the operands' AND
        // chain (kid 0) carries each hit's operand number in nodepos; chains may overlap, so a complete one shifts, not resets
        const GenRes& I = res[N.kid[0]];
        if (!I.ok || !I.n) break;
        GenHit* out = A.take(I.n);
        if (!out) break;
        const uint32_t dist = (uint32_t)N.opt;
        uint32_t last_p = 0, last_ml = 0, last_sl = 0, last_w = 0, prelast_p = 0, prelast_ml = 0, prelast_sl = 0, prelast_w = 0, first_hit = 0, weight = 0;
        uint32_t first_npos = 0, first_qpos = 65535, n = 0, ffield = 0;
        for (uint32_t hi = 0; hi < I.n; ++hi) {
          const GenHit h = I.p[hi];
          const uint32_t hpf = gen_pwf(h.hitpos), npos = h.nodepos, qpos = h.qpos;
          if (last_p == hpf) {
            if (npos < first_npos) { // leftmost (in the query) of all dupes: 'a NEAR/2 a'
              first_qpos = qpos, first_npos = npos;
              continue;
            } else if (prelast_p && last_ml < h.matchlen) { // the hit is a subset of another one: roll back
              last_ml = prelast_ml, last_sl = prelast_sl;
              first_hit = last_p = prelast_p;
              weight = weight - last_w + prelast_w;
            } else
              continue;
          }
          if (last_p == 0 || last_p + last_ml + dist <= hpf) { // probably a new chain
            first_hit = last_p = hpf;
            last_ml = h.matchlen, last_sl = h.spanlen;
            weight = last_w = h.weight;
            first_qpos = qpos, first_npos = npos;
            continue;
          }
          if (first_hit + last_ml > hpf && first_hit + last_ml < hpf + h.matchlen && last_ml != h.matchlen) { // hold the overlapping
            first_hit = last_p = hpf;
            last_ml = h.matchlen, last_sl = h.spanlen;
            weight = last_w = h.weight;
            first_qpos = qpos, first_npos = npos;
            continue;
          }
          if (npos == first_npos) {
            if (last_p < hpf) {
              prelast_ml = last_ml, prelast_sl = last_sl, prelast_p = last_p, prelast_w = h.weight;
              first_hit = last_p = hpf;
              last_ml = h.matchlen, last_sl = h.spanlen;
              weight = last_w = prelast_w;
              first_qpos = qpos, first_npos = npos;
            }
            continue;
          }
          weight += h.weight;
          last_ml = h.matchlen, last_sl = h.spanlen;
          if (!n) ffield = h.hitpos >> 24;
          out[n++] = gen_hit(first_hit, first_qpos < qpos ? first_qpos : qpos, 0, 2, hpf - first_hit + last_ml, weight);
          prelast_p = 0;
          first_hit = last_p = hpf;
          weight = h.weight;
          first_qpos = qpos;
        }
        (void)last_sl, (void)prelast_sl;
        if (!n) break;
        R.p = out, R.n = n, R.ok = true, R.tfidf = I.tfidf, R.fields = 1u << (ffield & 31u);
        }
        break;
      case GN_QUORUM: {
        // ExtQuorum_c over plain keywords (kid[] = keyword slots in query-position order): at least `opt` of them hold the doc;
        // tfidf adds up in the order m_dChildren has at this rowid (a keyword whose doclist ran dry has left by RemoveFast);
        // the hits of those present merge by (position without the end flag, query position) (QuorumCmpHitPos_fn :4548-4564)
        const uint32_t k = N.n_kids;
        GenRes kr[MRK_MAX_AND_TERMS];
        uint32_t have = 0, total = 0;
        bool bad = false;
        for (uint32_t i = 0; i < k; ++i) {
          gen_term(seg, Q->t[N.kid[i]], refs[N.kid[i]], row, A, kr[i]);
          if (kr[i].ok) ++have, total += kr[i].n;
          bad = bad || A.failed;
        }
        if (bad || have < (uint32_t)N.opt) break;
        uint32_t ord = Q->qr_ord[0];
        for (uint32_t e = 0; e < Q->qr_n; ++e)
          if (row > Q->qr_row[e]) ord = Q->qr_ord[e + 1];
        float t = 0.0f;
        uint32_t f = 0;
        bool first = true;
        for (int i = 0; i < QUORUM_EVENTS; ++i) {
          const uint32_t sl = (ord >> (4 * i)) & 15u;
          if (sl == 15u) continue;
          for (uint32_t j = 0; j < k; ++j)
            if (N.kid[j] == sl && kr[j].ok) {
              t = first ? kr[j].tfidf : t + kr[j].tfidf;
              first = false;
              f |= kr[j].fields;
            }
        }
        GenHit* out = A.take(total);
        if (!out) break;
        uint32_t cur[MRK_MAX_AND_TERMS], n = 0;
        for (uint32_t i = 0; i < k; ++i) cur[i] = 0;
        for (;;) {
          int best = -1;
          for (uint32_t i = 0; i < k; ++i) {
            if (!kr[i].ok || cur[i] >= kr[i].n) continue;
            if (best < 0) {
              best = (int)i;
              continue;
            }
            const GenHit &a = kr[i].p[cur[i]], &b = kr[best].p[cur[best]];
            if (gen_pwf(a.hitpos) < gen_pwf(b.hitpos) || (gen_pwf(a.hitpos) == gen_pwf(b.hitpos) && a.qpos < b.qpos)) best = (int)i;
          }
          if (best < 0) break;
          out[n++] = kr[best].p[cur[best]++];
        }
        R.p = out, R.n = n, R.ok = true, R.tfidf = t, R.fields = f;
        break;
      }
      case GN_ORDER: {
        // ExtOrder_c, the BEFORE operator: every child holds the doc and their hits line up in order inside one field
        // (GetMatchingHits :4734-4829: the longest in-order run so far and the most recently started one; a full run is
        // flushed); the doc is the FIRST child's doc as it is (:4907-4908)
        const uint32_t k = N.n_kids;
        bool all = true;
        uint32_t total = 0;
        for (uint32_t i = 0; i < k; ++i) all = all && res[N.kid[i]].ok, total += res[N.kid[i]].n;
        if (!all) break;
        GenHit* out = A.take(total);
        if (!out) break;
        GenHit acc_l[MRK_MAX_AND_TERMS], acc_r[MRK_MAX_AND_TERMS];
        uint32_t cur[MRK_MAX_AND_TERMS], n = 0;
        int len_l = 0, len_r = 0, pos_l = 0, pos_r = 0, field = -1;
        for (uint32_t i = 0; i < k; ++i) cur[i] = 0;
        for (;;) {
          uint32_t best_pos = 0xFFFFFFFFu; // GetChildIdWithNextHit (:4706-4731): least position, ties to the first child
          int c = -1;
          for (uint32_t i = 0; i < k; ++i) {
            const GenRes& K = res[N.kid[i]];
            if (cur[i] < K.n && gen_pwf(K.p[cur[i]].hitpos) < best_pos) best_pos = gen_pwf(K.p[cur[i]].hitpos), c = (int)i;
          }
          if (c < 0) break;
          const GenHit h = res[N.kid[c]].p[cur[c]];
          const int hfield = (int)(h.hitpos >> 24), hpos = (int)(h.hitpos & 0x7FFFFFu);
          if (hfield != field) { // new field: both trackers start over
            len_l = len_r = 0;
            if (c == 0) {
              acc_l[len_l++] = h;
              pos_l = hpos + h.spanlen;
              field = hfield;
            }
          } else if (c == len_l && hpos >= pos_l) {
            acc_l[len_l++] = h;
            pos_l = hpos + h.spanlen;
            if (len_l == (int)k) {
              for (int i = 0; i < len_l && n < total; ++i) out[n++] = acc_l[i];
              len_l = len_r = 0;
              pos_r = pos_l;
            }
          } else if (c == 0) {
            len_r = 0;
            acc_r[len_r++] = h;
            pos_r = hpos + h.spanlen;
            if (!len_l) {
              acc_l[len_l++] = h;
              pos_l = hpos + h.spanlen;
            }
          } else if (c == len_r && hpos >= pos_r) {
            acc_r[len_r++] = h;
            pos_r = hpos + h.spanlen;
            if (len_r == len_l) {
              for (int i = 0; i < len_r; ++i) acc_l[i] = acc_r[i];
              len_r = 0;
              pos_l = pos_r;
            }
          }
          ++cur[c];
        }
        if (!n) break;
        const GenRes& F = res[N.kid[0]];
        R.p = out, R.n = n, R.ok = true, R.tfidf = F.tfidf, R.fields = F.fields;
        break;
      }
      default: break;
    }
    if (A.failed) return false;
    res[ni] = R;
  }
  const GenRes& root = res[nn - 1];
  if (!root.ok) return false;
  tfidf_out = root.tfidf;
  fields_out = root.fields;
  rk_out = 0;
  if (state_ranker) {
    if (!root.n) return false; // ExtRanker_State_T::GetMatches: a doc without hits is never flushed (sphinxsearch.cpp:1198-1315)
    RankState X;
    X.reset();
    for (uint32_t i = 0; i < root.n; ++i) {
      const GenHit h = root.p[i];
      X.update(Q->ranker, dupes, gen_pwf(h.hitpos), ((h.hitpos >> 23) & 1u) != 0, h.qpos, h.weight, (uint32_t)h.spanlen - 1u, fw, nw, (int)Q->max_qpos);
    }
    rk_out = X.finalize(Q->ranker, nw, fw, (int)Q->n_qwords);
  }
  return true;
}

} // namespace mrk
