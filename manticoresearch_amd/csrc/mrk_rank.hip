// mrk_rank.hip -- rank_kernel: the hit pass + state rankers over the HBM match queue, gfx950 / wave64.
//
// The scan kernels decide WHICH docs match (posting intersection, boolean tree); for the hit-reading rankers
// (ExtRanker_State_T<...>::GetMatches, sphinxsearch.cpp:1198-1315) and whole-query PHRASE each match then needs its
// keywords' hit streams merged and fed to the ranker -- a chain of dependent loads per doc.  Doing that inside the scan
// kernel pinned it at 2 waves per SIMD (178 VGPRs) and left waves idle behind the items with the most matches.  Here the
// matches arrive as 64-entry chunks in HBM and a persistent grid drains them: one doc per lane, any wave takes any
// chunk, and the common case
// (no PHRASE / BEFORE node, no position modifier) runs the lean hit_rank_plain.
#include "mrk_khits.h"
#include "mrk_kprune.h"
#include "mrk_keval.h"

namespace mrk {

constexpr int RK_CBUF = 128; // candidates a wave collects before it publishes them

struct __align__(16) RkWaveLds {
  uint64_t cbuf[RK_CBUF];
  int32_t fw[8]; // the current query's per-field weights
};

struct __align__(16) RkSmem {
  RkWaveLds w[WAVES];
  uint32_t hist[NBINS]; // publishing scratch, one per workgroup behind hist_lock (as in scan_bm_kernel)
  uint32_t hist_lock;
  uint32_t pre[MQ_SHARDS + 1]; // chunks in the shards before shard s (filled prefixes laid end to end)
};

// MODE 0: lean (hit_rank_plain / hit_rank_prox), 1: FAT (hit_pass with the word state machines), 2: GEN (gen_eval)
template <int MODE>
__global__ __launch_bounds__(WG) void rank_kernel(ScanArgs a) {
  constexpr bool FAT = MODE == 1, GEN = MODE == 2;
  constexpr int NP = GEN ? MQ_GEN_PLANES : MQ_PLANES; // planes of a queue entry
  __shared__ RkSmem s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const MatchQueue MQ = a.mq[MODE];
  RkWaveLds& L = s.w[wave];
  if (!tid) s.hist_lock = 0;
  static_assert(MQ_SHARDS == 64, "one shard per lane of the prefix sum");
  if (wave == 0) {
    uint32_t v = __hip_atomic_load(MQ.count + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v > MQ.cap) v = MQ.cap;
    const uint32_t inc = wave_incl_scan(v);
    s.pre[lane + 1] = inc;
    if (!lane) s.pre[0] = 0;
  }
  __syncthreads(); // the waves never meet again
  // chunks are dealt out statically: wave w of the grid takes runs of MQ_BATCH virtual chunks (one producer wave's
  // reservation, i.e. one query: the per-query state below changes hands less often), w-th run of every n_waves runs, of
  // the shards' filled prefixes laid end to end.  64 docs of work per chunk, hundreds of chunks per wave: the spread evens
  // out, and a shared cursor would be one atomic address hit once per chunk by every wave of the chip.
  const uint32_t n_chunks = s.pre[MQ_SHARDS];
  const uint32_t n_waves = gridDim.x * WAVES;
  const bool inline_hits = a.seg.inline_hits != 0;

  // the logical query whose candidates / total the wave is holding
  uint32_t cur_oq = 0xFFFFFFFFu, cn = 0, total = 0, tau_bin = 0;
  uint32_t K = 1, bin_mode = 0, bin_shift = 0, cand_cap = 0;
  int32_t bin_lo = 0;
  uint64_t* cand = nullptr;
  uint32_t *ghist = nullptr, *gcount = nullptr, *gtaubin = nullptr;

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
      if (!fits && lane == 0) atomicOr(a.q_flags + cur_oq, QF_OVERFLOW);
      wave_lds_fence();
      flush_hist(s.hist, ghist);
      wave_lds_fence();
      if (lane == 0) __hip_atomic_store(&s.hist_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      cn = 0;
      if ((basep >> 11) != ((basep + npub) >> 11) || basep == 0) { // (see scan_pk_kernel: who recomputes the threshold)
        const uint32_t tb = threshold_bin(ghist, K);
        if (tb > tau_bin) {
          tau_bin = tb;
          if (lane == 0) atomicMax(gtaubin, tb);
        }
      }
    }
  };
  auto leave_query = [&]() {
    if (cur_oq == 0xFFFFFFFFu) return;
    publish();
    uint32_t t = total;
    for (int dlt = 32; dlt; dlt >>= 1) t += __shfl_down(t, dlt, 64);
    if (lane == 0 && t) atomicAdd((unsigned long long*)(a.q_total + cur_oq), (unsigned long long)t);
    total = 0;
  };

  // what the current pass's descriptor says, cached as uniform values (see the loop)
  uint32_t cur_pass = 0xFFFFFFFFu, qc_nterms = 0, qc_ranker = 0, qc_flags = 0, qc_index_weight = 1, qc_nw = 0, qc_max_qpos = 0, qc_n_qwords = 0,
           qc_ph_mask = 0;
  HitCtx HC;
  HC.spp = a.seg.spp;
  HC.hit = a.seg.pk_hit;
  HC.hbase = a.seg.pk_hbase;
  HC.fw = L.fw;
  HC.inline_hits = inline_hits;
#pragma unroll
  for (int t = 0; t < MAX_PROX_TERMS; ++t) HC.tb[t] = HC.tq[t] = HC.tm[t] = HC.tpk[t] = HC.tpm[t] = 0;
  HC.ap0 = HC.ap1 = HC.ap2 = HC.ap3 = 0, HC.px_dist = 0;
  HC.nn_a = HC.nn_b = HC.nn_dist = 0;

  GenAlloc GA;
  GA.lane = nullptr, GA.cap = 0, GA.used = 0, GA.spill = nullptr, GA.spill_cap = 0, GA.spill_used = nullptr, GA.failed = false;
  if (GEN) {
    const uint32_t gl = blockIdx.x * WG + tid;
    GA.lane = a.gen.lane_arena + (uint64_t)(gl < a.gen.n_lanes ? gl : 0u) * a.gen.lane_hits;
    GA.cap = gl < a.gen.n_lanes ? a.gen.lane_hits : 0u;
    GA.spill = a.gen.spill, GA.spill_cap = a.gen.spill_cap, GA.spill_used = a.gen.spill_used;
  }
  uint32_t qc_gen_prog = 0, qc_nwf = 0;

  // virtual -> physical chunk: the shard whose prefix range holds it (uniform binary search over 65 LDS words)
  auto chunk_of = [&](uint32_t vc) -> uint32_t {
    uint32_t lo = 0, hi = MQ_SHARDS; // pre[lo] <= vc < pre[hi]
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s.pre[mid] <= vc)
        lo = mid;
      else
        hi = mid;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(lo * MQ.cap + (vc - s.pre[lo])));
  };
  // run r of the wave = virtual chunks [ (r * n_waves + gw) * MQ_BATCH, + MQ_BATCH )
  const uint32_t gw = blockIdx.x * WAVES + wave;
  auto vchunk = [&](uint32_t i) -> uint32_t { return ((i / MQ_BATCH) * n_waves + gw) * MQ_BATCH + (i % MQ_BATCH); };
  // the chunk after the one being ranked is already on its way (header + the 7 planes): one memory round trip less in
  // every chunk's chain of dependent loads
  uint32_t it = 0, vc = vchunk(0);
  uint32_t nx_hdr = 0, nx[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) nx[i] = 0;
  constexpr bool PREFETCH = MODE == 0; // (the PHRASE & co instance would drop from 3 to 2 waves per SIMD for the 8 registers)
  if (PREFETCH && vc < n_chunks) {
    const uint32_t c = chunk_of(vc);
    nx_hdr = MQ.hdr[c];
    const uint32_t* __restrict__ d = MQ.data + (uint64_t)c * (NP * 64) + lane;
#pragma unroll
    for (int i = 0; i < NP; ++i) nx[i] = d[64 * i];
  }
  for (; vc < n_chunks; vc = vchunk(++it)) {
    if (!PREFETCH) {
      const uint32_t c = chunk_of(vc);
      nx_hdr = MQ.hdr[c];
      const uint32_t* __restrict__ d = MQ.data + (uint64_t)c * (NP * 64) + lane;
#pragma unroll
      for (int i = 0; i < NP; ++i) nx[i] = d[64 * i];
    }
    const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx_hdr);
    uint32_t cur[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) cur[i] = nx[i];
    if (PREFETCH && vchunk(it + 1) < n_chunks) {
      const uint32_t c = chunk_of(vchunk(it + 1));
      nx_hdr = MQ.hdr[c];
      const uint32_t* __restrict__ d = MQ.data + (uint64_t)c * (NP * 64) + lane;
#pragma unroll
      for (int i = 0; i < NP; ++i) nx[i] = d[64 * i];
    }
    const uint32_t n = hdr >> 24, pass = hdr & 0xFFFFFFu;
    if (pass != cur_pass) {
      // another pass's chunks: fetch what the hit pass and the weighing read of its descriptor ONCE, as uniform values
      // (one batch of independent loads; a producer wave's reservation of MQ_BATCH chunks arrives as one run).  Loads
      // behind the kernel's own stores are not scalarized by the compiler, hence the explicit readfirstlane.
      cur_pass = pass;
      const DevQuery* __restrict__ Q = a.queries + pass;
      auto U = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
      const uint32_t oq = U(Q->out_q);
      qc_nterms = U(Q->n_terms), qc_ranker = U(Q->ranker), qc_flags = U(Q->tree_flags), qc_index_weight = U(Q->index_weight);
      qc_nw = U(Q->n_weights < 8u ? Q->n_weights : 8u);
      qc_max_qpos = U(Q->max_qpos), qc_n_qwords = U(Q->n_qwords);
      qc_gen_prog = GEN ? U(Q->gen_prog) : 0u;
      qc_nwf = U(Q->n_wfilters);
#pragma unroll
      for (int t = 0; t < MAX_PROX_TERMS; ++t) {
        HC.tb[t] = U(Q->t[t].blk_first), HC.tq[t] = U(Q->t[t].qpos), HC.tm[t] = U(Q->t[t].queried32);
        HC.tpk[t] = FAT ? U(Q->t[t].tp_kind) : 0u, HC.tpm[t] = FAT ? U(Q->t[t].tp_max) : 0u;
      }
      HC.ap0 = FAT ? U(Q->ph_atoms[0]) : 0u, HC.ap1 = FAT ? U(Q->ph_atoms[1]) : 0u, HC.ap2 = FAT ? U(Q->ph_atoms[2]) : 0u, HC.ap3 = FAT ? U(Q->ph_atoms[3]) : 0u;
      qc_ph_mask = FAT ? U(Q->ph_mask) : 0u;
      HC.px_dist = FAT ? U(Q->px_dist) : 0u;
      {
        const bool nnq = FAT && (U(Q->tree_flags) & TF_NOTNEAR) != 0;
        HC.nn_a = nnq ? U(Q->nn_a) : 0u, HC.nn_b = nnq ? U(Q->nn_b) : 0u, HC.nn_dist = nnq ? U(Q->nn_dist) : 0u;
      }
      const uint32_t k_ = U(Q->k), bm_ = U(Q->bin_mode), bs_ = U(Q->bin_shift), bl_ = U((uint32_t)Q->bin_lo), cc_ = U(Q->cand_cap);
      const uint64_t co_ = ((uint64_t)U((uint32_t)(Q->cand_off >> 32)) << 32) | U((uint32_t)Q->cand_off);
      const int32_t wl = lane < 8 ? Q->weights[lane] : 0;
      if (oq != cur_oq) {
        leave_query();
        cur_oq = oq;
        K = k_, bin_mode = bm_, bin_shift = bs_, bin_lo = (int32_t)bl_, cand_cap = cc_;
        cand = a.cand + co_;
        ghist = a.q_hist + (uint64_t)oq * NBINS;
        gcount = a.q_cand_n + (size_t)oq * QSTRIDE;
        gtaubin = a.q_tau_bin + (size_t)oq * QSTRIDE;
        tau_bin = 0;
      }
      wave_lds_fence();
      if (lane < 8) L.fw[lane] = wl;
      wave_lds_fence();
    }
    {
      const uint32_t gt = __hip_atomic_load(gtaubin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gt > tau_bin) tau_bin = gt;
    }
    const uint32_t nterms = qc_nterms, ranker = qc_ranker, flags = qc_flags, nw = qc_nw;
    const bool phrase = FAT && (flags & TF_PHRASE) != 0;
    const bool ph_leaf = FAT && (flags & TF_PHRASE_LEAF) != 0;
    const bool prox_ranker = (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY)
                                 ? (GEN || nterms > 1) // (a generic-path tree is never a single keyword)
                                 : (ranker == MRK_RANK_WORDCOUNT || ranker == MRK_RANK_MATCHANY || ranker == MRK_RANK_FIELDMASK ||
                                    ranker == MRK_RANK_SPH04);
    HC.flags = a.q_flags + cur_oq;
    HC.nterms = nterms, HC.nw = nw;
    const uint32_t ph_mask = ph_leaf ? qc_ph_mask : 0u;
    HC.nph = !FAT ? 0u : phrase ? nterms : (uint32_t)__popc(ph_mask);
    {
      const uint32_t last = HC.nph == 2 ? HC.ap1 : HC.nph == 3 ? HC.ap2 : HC.ap3;
      HC.span = !FAT || HC.nph < 2 ? 0u : last - HC.ap0;
    }
    HC.ranker = ranker;
    HC.max_qpos = (int)qc_max_qpos, HC.n_qwords = (int)qc_n_qwords;
    HC.multi_and = (flags & TF_MULTIAND) != 0 && !phrase;
    HC.quorum_hits = (flags & TF_QUORUM_HITS) != 0;
    HC.termpos = FAT && (flags & TF_TERMPOS) != 0;
    HC.order = FAT && (flags & TF_ORDER) != 0;
    HC.apack = (uint64_t)(HC.ap0 & 0xFFFFu) | ((uint64_t)(HC.ap1 & 0xFFFFu) << 16) | ((uint64_t)(HC.ap2 & 0xFFFFu) << 32) | ((uint64_t)(HC.ap3 & 0xFFFFu) << 48);
    HC.dupes = (flags & TF_DUPES) != 0 && (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY);
    // the specialized pass: proximity family, distinct keywords, hits ordered by the raw position, no MergeHits3 field quirk
    const bool fast_prox = !FAT && (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY) && !HC.dupes && !HC.quorum_hits && nterms >= 2 &&
                           !(HC.multi_and && nterms == 3 && (HC.tm[0] & HC.tm[1] & HC.tm[2]) != 0xFFFFFFFFu);

    const bool valid = lane < n;
    const uint32_t rowid = cur[0], fa = GEN ? 0u : cur[2];
    float tfidf = GEN ? 0.0f : __uint_as_float(cur[1]);
    const uint32_t r0 = GEN ? 0u : cur[3], r1 = GEN ? 0u : cur[4], r2 = GEN ? 0u : cur[5], r3 = GEN ? 0u : cur[6];
    bool is_live = valid;
    uint32_t fields = fa & 0xffu;
    int rk = 0;
    if (GEN) {
      if (valid) {
        uint32_t refs[MRK_MAX_AND_TERMS];
#pragma unroll
        for (int t = 0; t < MRK_MAX_AND_TERMS; ++t) refs[t] = cur[NP - MRK_MAX_AND_TERMS + t];
        const bool nearn = (flags & TF_GEN_NEARN) != 0, probe = a.gen.phase == 1;
        uint32_t fq = 65535u, m_ins = 65535u;
        uint32_t* __restrict__ tab = a.gen.near_tab + (uint64_t)cur_oq * 64;
        if (nearn && !probe) // the least query position an earlier doc inserted (the table is final: the probe launch ran before)
          for (uint32_t v = 0; v < 64; ++v)
            if (tab[v] < rowid) {
              fq = v;
              break;
            }
        if (probe && !nearn)
          is_live = false;
        else
          is_live = gen_eval(a.seg, a.queries + cur_pass, a.gen.progs + qc_gen_prog, refs, rowid, GA, prox_ranker, HC.dupes, L.fw, nw, a.q_flags + cur_oq, tfidf,
                             fields, rk, fq, &m_ins);
        if (probe) {
          if (nearn && m_ins < 64u) atomicMin(tab + m_ins, rowid);
          is_live = false;
        }
        if (GA.failed) {
          atomicOr(a.q_flags + cur_oq, QF_ARENA);
          GA.failed = false;
        }
      }
    } else if (valid) {
      const uint32_t all_slots = (1u << (nterms < (uint32_t)MAX_PROX_TERMS ? nterms : (uint32_t)MAX_PROX_TERMS)) - 1u;
      const uint32_t smask = (fa >> 8) & all_slots;
      if (FAT) {
        const uint32_t pmask = phrase ? all_slots : (ph_leaf && (smask & ph_mask) == ph_mask) ? ph_mask : 0u;
        bool found = false;
        uint32_t ffield = 0;
        hit_pass(HC, r0, r1, r2, r3, smask, pmask, prox_ranker, found, ffield, rk, ((fa >> 16) & 1u) != 0);
        if (phrase) {
          is_live = found;
          fields = 1u << ffield; // the doc's field mask comes from its first occurrence (searchnode.cpp:3836)
        }
      } else
        if (fast_prox) // (uniform: decided per pass)
          rk = nterms == 2 ? hit_rank_prox<2>(HC, r0, r1, r2, r3, smask) : nterms == 3 ? hit_rank_prox<3>(HC, r0, r1, r2, r3, smask)
                                                                                       : hit_rank_prox<4>(HC, r0, r1, r2, r3, smask);
        else
          rk = hit_rank_plain(HC, r0, r1, r2, r3, smask);
    }
    // the match: weight, pruning bin, candidate buffer (emit_match of scan_pk_kernel)
    bool push = false;
    uint64_t key = 0;
    uint32_t weight = 0;
    if (is_live) {
      if (ranker == MRK_RANK_NONE)
        weight = 1u; // ExtRanker_None_c, sphinxsearch.cpp:1160
      else if (prox_ranker) {
        // Finalize() of the state rankers, e.g. RankerState_Proximity_fn sphinxsearch.cpp:1415-1437
        const int32_t bm = (int32_t)((tfidf + 0.5f) * 1000.0f);
        weight = (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_SPH04) ? (uint32_t)bm + (uint32_t)rk * 1000u : (uint32_t)rk;
      } else {
        // a whole-query PHRASE under the weight-sum rankers: ExtRanker_WeightSum_c (sphinxsearch.cpp:1070, 1112-1131) over the
        // occurrence's field
        uint32_t rsum = 0;
        if (!fields)
          rsum = 1; // empty mask: "just fake it" (sphinxsearch.cpp:1114-1118)
        else
          for (uint32_t f = 0; f < nw; ++f)
            if (fields & (1u << f)) rsum += (uint32_t)L.fw[f];
        const int32_t bm = (int32_t)((tfidf + 0.5f) * 1000.0f);
        weight = ranker == MRK_RANK_PROXIMITY ? rsum : (uint32_t)bm + rsum * 1000u;
      }
      weight *= qc_index_weight; // MatchExtended, sphinx.cpp:12220
      if (qc_nwf && !weight_passes_filters((a.queries + cur_pass)->wfilters, qc_nwf, (int32_t)weight)) is_live = false; // m_pWeightFilter (:12223-12227)
    }
    if (is_live) {
      ++total;
      const uint32_t grow = a.seg.rowid_base + rowid;
      if (bin_of(bin_mode, bin_lo, bin_shift, (int32_t)weight, grow) >= tau_bin) {
        push = true;
        key = make_key((int32_t)weight, grow);
      }
    }
    const uint64_t bal = __ballot(push);
    if (bal) {
      const uint32_t np = (uint32_t)__popcll(bal);
      if (cn + np > (uint32_t)RK_CBUF) publish(); // keys pushed under the older threshold stay valid candidates
      if (push) L.cbuf[cn + __popcll(bal & ((1ull << lane) - 1ull))] = key;
      cn += np;
      if (cn >= (uint32_t)RK_CBUF - 64u) publish();
    }
  }
  leave_query();
}

void launch_rank(const ScanArgs& a, int which, void* stream) {
  // persistent grid: enough workgroups to fill every CU at the kernel's occupancy; late ones find the cursor past the
  // count and leave at once
  const dim3 grid(256 * 8), block(WG);
  if (which == 2)
    hipLaunchKernelGGL(rank_kernel<2>, dim3(GEN_GRID), block, 0, (hipStream_t)stream, a);
  else if (which)
    hipLaunchKernelGGL(rank_kernel<1>, grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(rank_kernel<0>, grid, block, 0, (hipStream_t)stream, a);
}

} // namespace mrk
