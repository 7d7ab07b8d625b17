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
#include "mrk_khits.h"
#include "mrk_kpk.h"
#include "mrk_kmq.h"

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

template <bool PROX, bool TREE, int NREF = MAX_PROX_TERMS>
struct __align__(16) PkWaveLds {
  uint64_t cbuf[CBUF];  // candidates not yet published to the query's global list
  // proximity rankers: where each matched doc sits in the other terms' blocks (block<<7 | slot, bit 31 = lone hit)
  uint32_t href[PROX ? NREF - 1 : 1][PROX ? DEVBLK : 1];
  uint32_t tj_rowid[DEVBLK];
  uint32_t tj_attr[64];
  // boolean trees: what each keyword contributes to each doc of the driver block (tfidf term, field bits)
  // hit rankers / PHRASE: matched docs wait here until 64 of them can go through the hit pass with every lane busy
  // (row, tfidf sum, fields | contributing keywords << 8, one hit reference per keyword)
  uint32_t mq_row[PROX ? MQCAP : 1];
  float mq_acc[PROX ? MQCAP : 1];
  uint32_t mq_fa[PROX ? MQCAP : 1];
  uint32_t mq_ref[PROX ? NREF : 1][PROX ? MQCAP : 1];
  float kv[TREE ? MRK_MAX_AND_TERMS : 1][TREE ? DEVBLK : 4];
  uint8_t kf[TREE ? MRK_MAX_AND_TERMS : 1][TREE ? DEVBLK : 16];
  union {
    uint8_t map[MAPCAP];  // rowid offset -> slot of the decoded other-term block
    uint32_t hist[NBINS]; // publishing scratch: per-bin counts of the candidates being flushed
  };                      // (publishing invalidates the decoded block: it is simply decoded again)
};
static_assert(NBINS * 4 <= MAPCAP, "hist must fit the map area");

template <bool PROX, bool TREE, int NREF = MAX_PROX_TERMS>
struct __align__(16) PkSmem {
  PkWaveLds<PROX, TREE, NREF> w[WAVES];
  uint32_t rank[256];
  float tfidf[1][256]; // really [n_terms][256]: the tail lives in dynamic LDS right behind this struct
};

// EXT: the batch holds queries with position modifiers, a BEFORE node or attribute filters; batches without them run the
// leaner instance (the extra code costs the three-keyword proximity mixes ~7 % even when it never executes)
// NREF: keyword slots whose packed-array references travel with a match (more than MAX_PROX_TERMS: the instance that
// feeds the generic evaluator's queue -- candidates, not matches, with one reference per keyword)
template <bool PROX, bool TREE, bool EXT = false, int NREF = MAX_PROX_TERMS>
__global__ __launch_bounds__(WG) void scan_pk_kernel(ScanArgs a) {
  constexpr bool GEN = NREF > MAX_PROX_TERMS;
  extern __shared__ __align__(16) uint8_t smem_raw[];
  PkSmem<PROX, TREE, NREF>& s = *reinterpret_cast<PkSmem<PROX, TREE, NREF>*>(smem_raw);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= a.n_items) return;
  const DevItem item = a.items[blockIdx.x];
  const DevQuery* __restrict__ Q = a.queries + item.query;
  // a query that already overflowed (candidate list or match queue) is rerun alone by the host whatever else this launch finds for it:
  // its remaining work items are not worth their time -- least of all when the queue they would write to is full
  if (__hip_atomic_load(a.q_flags + Q->out_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & QF_OVERFLOW) return;
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
  PkWaveLds<PROX, TREE, NREF>& L = s.w[wave];
  const uint32_t oq = Q->out_q; // logical query: several passes (driver keywords) may feed one result
  const uint32_t req_mask = TREE ? Q->req_mask : 0u, excl_mask = TREE ? Q->excl_mask : 0u;
  const uint32_t n_nodes = TREE ? Q->n_nodes : 0u;
  const bool phrase = PROX && (Q->tree_flags & TF_PHRASE) != 0;                // the whole query is one PHRASE
  const bool ph_leaf = EXT && TREE && PROX && (Q->tree_flags & TF_PHRASE_LEAF) != 0; // a PHRASE below other operators
  const uint32_t ph_mask = ph_leaf ? Q->ph_mask : 0u;                         // its words' keyword slots
  const uint32_t ph_n = !PROX ? 0u : phrase ? nterms : (uint32_t)__popc(ph_mask);
  const uint32_t ph_span = !PROX || ph_n < 2 ? 0u : Q->ph_atoms[ph_n - 1] - Q->ph_atoms[0];
  const bool need_hits = PROX && (prox_ranker || phrase || GEN); // matches go through the hit pass before they are weighed
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
  uint32_t* __restrict__ gcount = a.q_cand_n + (size_t)oq * QSTRIDE;
  uint32_t* __restrict__ gtaubin = a.q_tau_bin + (size_t)oq * QSTRIDE;
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
    uint32_t weight = 0;
    if (is_live) {
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
      if (EXT && Q->n_wfilters && !weight_passes_filters(Q->wfilters, Q->n_wfilters, (int32_t)weight)) is_live = false; // m_pWeightFilter (:12223-12227)
    }
    if (is_live) {
      ++total;
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

  // Matched docs of hit-ranked queries leave through the HBM match queue (rank_kernel, mrk_rank.hip): 64 entries per
  // chunk, one coalesced 256-B row per plane.  Queries with PHRASE / PROXIMITY / BEFORE nodes or position modifiers go to
  // the second queue, whose consumer carries the word state machines.
  uint32_t mqn = 0;
  MqWriter mqw;
  const bool fat_q = (Q->tree_flags & TF_FAT) != 0;
  auto flush_matches = [&](uint32_t from, uint32_t n) __attribute__((always_inline)) {
    if (!PROX) return;
    wave_lds_fence();
    const MatchQueue& MQ = a.mq[GEN ? 2 : fat_q ? 1 : 0];
    const uint32_t c = mq_take(MQ, mqw);
    if (c != 0xFFFFFFFFu) {
      uint32_t* __restrict__ d = MQ.data + (uint64_t)c * ((GEN ? MQ_GEN_PLANES : MQ_PLANES) * 64) + lane;
      const uint32_t e = from + lane; // (entries past n are stale slots; the header's count masks them)
      d[0] = L.mq_row[e];
      if (GEN) {
#pragma unroll
        for (int t = 0; t < NREF; ++t) d[64 + 64 * t] = L.mq_ref[t][e];
      } else {
        d[64] = __float_as_uint(L.mq_acc[e]);
        d[128] = L.mq_fa[e];
#pragma unroll
        for (int t = 0; t < MAX_PROX_TERMS; ++t) d[192 + 64 * t] = L.mq_ref[t][e];
      }
      if (lane == 0) MQ.hdr[c] = item.query | (n << 24);
    } else if (lane == 0)
      atomicOr(a.q_flags + oq, QF_OVERFLOW); // the host reruns the query alone with a queue sized for all its driver docs
    wave_lds_fence(); // the slots may be refilled from here on
  };
  // in-scan hit reading (EXT instance only): does a PHRASE / BEFORE node below other operators occur in the doc, does a
  // keyword with a position modifier hold it
  HitCtx HC;
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t) {
    const DevTerm& Tt = Q->t[t];
    HC.tb[t] = Tt.blk_first, HC.tq[t] = Tt.qpos, HC.tm[t] = Tt.queried32, HC.tpk[t] = Tt.tp_kind, HC.tpm[t] = Tt.tp_max;
  }
  HC.spp = a.seg.spp;
  HC.hit = a.seg.pk_hit;
  HC.hbase = a.seg.pk_hbase;
  HC.flags = a.q_flags + oq;
  HC.nterms = nterms, HC.nw = nw;
  HC.ap0 = ap0, HC.ap1 = ap1, HC.ap2 = ap2, HC.ap3 = ap3;
  HC.nph = ph_n, HC.span = ph_span;
  HC.px_dist = PROX ? Q->px_dist : 0u;
  HC.ranker = ranker;
  HC.fw = Q->weights; // (never read here: the in-scan passes do not rank)
  HC.max_qpos = (int)Q->max_qpos, HC.n_qwords = (int)Q->n_qwords;
  HC.inline_hits = inline_hits, HC.multi_and = multi_and;
  HC.quorum_hits = (Q->tree_flags & TF_QUORUM_HITS) != 0;
  HC.termpos = EXT && PROX && TREE && (Q->tree_flags & TF_TERMPOS) != 0; // both only occur in tree programs
  HC.order = EXT && PROX && TREE && (Q->tree_flags & TF_ORDER) != 0;
  HC.apack = (uint64_t)(ap0 & 0xFFFFu) | ((uint64_t)(ap1 & 0xFFFFu) << 16) | ((uint64_t)(ap2 & 0xFFFFu) << 32) | ((uint64_t)(ap3 & 0xFFFFu) << 48);
  HC.dupes = (Q->tree_flags & TF_DUPES) != 0 && (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY);
  const bool notnear = EXT && PROX && TREE && (Q->tree_flags & TF_NOTNEAR) != 0;
  HC.nn_a = notnear ? Q->nn_a : 0u, HC.nn_b = notnear ? Q->nn_b : 0u, HC.nn_dist = notnear ? Q->nn_dist : 0u;

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
              if (PROX && j < (uint32_t)NREF)
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
                  if (PROX && j < (uint32_t)NREF)
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
      if (EXT && TREE && PROX && HC.termpos && __ballot(live[0] || live[1])) {
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
      if (EXT && TREE && PROX && ph_leaf && __ballot(live[0] || live[1])) {
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

      // ---- a NOTNEAR node: where both keywords hold the doc, whether a must-hit survives has to be known before the tree is evaluated
      bool nn_ok[2] = {true, true};
      if (EXT && TREE && PROX && notnear && __ballot(live[0] || live[1])) {
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (live[r] && ((pres[r] >> HC.nn_a) & 1u) && ((pres[r] >> HC.nn_b) & 1u)) {
            const uint32_t dref = ((inline_hits && ((cur0.attr >> (8 * r)) & 0xffu) == 1u) ? 0x80000000u : 0u) | (b << 7) | (lane + 64 * r);
            const uint32_t ra = HC.nn_a == 0 ? dref : L.href[(HC.nn_a - 1u) & 3u][lane + 64 * r];
            const uint32_t rb = HC.nn_b == 0 ? dref : L.href[(HC.nn_b - 1u) & 3u][lane + 64 * r];
            nn_ok[r] = notnear_any(HC, ra, rb);
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
          } else if (op == PN_NOTNEAR) {
            // ExtNotNear_c over (must, not): the must side's doc, values and hits; dropped where the not side holds the doc
            // too and no must-hit survived (searchnode.cpp:5383-5470)
            TreeEnt o;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const bool m = s1.m[r] && (!s0.m[r] || nn_ok[r]);
              o.m[r] = m;
              o.v[r] = m ? s1.v[r] : 0.0f;
              o.f[r] = m ? s1.f[r] : 0u;
              o.a[r] = m ? s1.a[r] : 0u;
            }
            s0 = o;
            s1 = s2;
            s2 = s3;
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
      if (EXT) { // cutoff: the scan stopped at that row (MatchExtended, sphinx.cpp:12261-12267)
#pragma unroll
        for (int r = 0; r < 2; ++r) live[r] = live[r] && row[r] <= Q->rowid_max;
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
              // (bit 16: the doc also holds the NOTNEAR node's not-keyword -- its hits filter the must side's in the final pass)
              L.mq_fa[pos] = (fld[r] & 0xffu) | ((TREE ? act[r] & 0xffu : 0xffu) << 8) | ((notnear && ((pres[r] >> HC.nn_b) & 1u)) ? 1u << 16 : 0u);
              L.mq_ref[0][pos] = ((inline_hits && ((cur0.attr >> (8 * r)) & 0xffu) == 1u) ? 0x80000000u : 0u) | (b << 7) | (lane + 64 * r);
#pragma unroll
              for (int t = 1; t < NREF; ++t) // (the generic evaluator is told which keywords the doc does not hold)
                L.mq_ref[t][pos] = (GEN && !((pres[r] >> t) & 1u)) ? 0xFFFFFFFFu : L.href[t - 1][lane + 64 * r];
            }
            mqn += (uint32_t)__popcll(bal);
            if (mqn >= 64u) {
              flush_matches(mqn - 64u, 64u);
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
  if (PROX && mqn) flush_matches(0u, mqn);
  if (PROX) mq_close(a.mq[GEN ? 2 : fat_q ? 1 : 0], mqw, item.query);

  // ---- wave epilogue
  if (cn) publish();
  {
    uint32_t t = total;
    for (int dlt = 32; dlt; dlt >>= 1) t += __shfl_down(t, dlt, 64);
    if (lane == 0 && t) atomicAdd((unsigned long long*)(a.q_total + oq), (unsigned long long)t);
  }
}

template <bool PROX, bool TREE, bool EXT = false, int NREF = MAX_PROX_TERMS>
static void launch_pk(const ScanArgs& a, size_t tail, hipStream_t st) {
  hipLaunchKernelGGL((scan_pk_kernel<PROX, TREE, EXT, NREF>), dim3(a.n_items), dim3(WG), sizeof(PkSmem<PROX, TREE, NREF>) + tail, st, a);
}

void launch_scan_pk(const ScanArgs& a, uint32_t max_terms, bool prox, bool tree, bool ext, void* stream, bool gen) {
  if (!a.n_items) return;
  if (max_terms < 1) max_terms = 1;
  const size_t tail = (size_t)(max_terms - 1) * 256 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (gen) // candidates of the generic evaluator: the tree kernel without the in-scan hit passes, a reference per keyword
    launch_pk<true, true, true, MRK_MAX_AND_TERMS>(a, tail, st);
  else if (ext) // filters alone may come with a plain AND: the EXT instance is the full tree + hit-stream kernel
    launch_pk<true, true, true>(a, tail, st);
  else if (tree)
    prox ? launch_pk<true, true>(a, tail, st) : launch_pk<false, true>(a, tail, st);
  else
    prox ? launch_pk<true, false>(a, tail, st) : launch_pk<false, false>(a, tail, st);
}

} // namespace mrk
