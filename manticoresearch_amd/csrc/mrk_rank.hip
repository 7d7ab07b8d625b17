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
};

template <bool FAT>
__global__ __launch_bounds__(WG) void rank_kernel(ScanArgs a) {
  __shared__ RkSmem s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const MatchQueue MQ = a.mq[FAT ? 1 : 0];
  RkWaveLds& L = s.w[wave];
  if (!tid) s.hist_lock = 0;
  __syncthreads(); // the waves never meet again
  // chunks are dealt out statically: wave w of the grid takes the virtual chunks w, w + W, w + 2W ... of the shards'
  // filled prefixes laid end to end (64 docs of work each, hundreds per wave: the spread evens out, and a shared cursor
  // would be one atomic address hit once per chunk by every wave of the chip)
  uint32_t shard_n[MQ_SHARDS], n_chunks = 0;
#pragma unroll
  for (int i = 0; i < MQ_SHARDS; ++i) {
    uint32_t v = __hip_atomic_load(MQ.count + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    shard_n[i] = v < MQ.cap ? v : MQ.cap;
    n_chunks += shard_n[i];
  }
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

  for (uint32_t vc = blockIdx.x * WAVES + wave; vc < n_chunks; vc += n_waves) {
    uint32_t rem = vc, c = 0; // virtual -> physical chunk
    bool found = false;
#pragma unroll
    for (int i = 0; i < MQ_SHARDS; ++i)
      if (!found) {
        if (rem < shard_n[i])
          c = (uint32_t)i * MQ.cap + rem, found = true;
        else
          rem -= shard_n[i];
      }
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    const uint32_t hdr = (uint32_t)__builtin_amdgcn_readfirstlane((int)MQ.hdr[c]);
    const uint32_t n = hdr >> 24;
    const DevQuery* __restrict__ Q = a.queries + (hdr & 0xFFFFFFu);
    const uint32_t oq = Q->out_q;
    if (oq != cur_oq) {
      leave_query();
      cur_oq = oq;
      K = Q->k, bin_mode = Q->bin_mode, bin_shift = Q->bin_shift, bin_lo = Q->bin_lo, cand_cap = Q->cand_cap;
      cand = a.cand + Q->cand_off;
      ghist = a.q_hist + (uint64_t)oq * NBINS;
      gcount = a.q_cand_n + oq;
      gtaubin = a.q_tau_bin + oq;
      tau_bin = 0;
      wave_lds_fence();
      if (lane < 8) L.fw[lane] = Q->weights[lane];
      wave_lds_fence();
    }
    {
      const uint32_t gt = __hip_atomic_load(gtaubin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gt > tau_bin) tau_bin = gt;
    }
    const uint32_t nterms = Q->n_terms, ranker = Q->ranker, flags = Q->tree_flags;
    const uint32_t nw = Q->n_weights < 8u ? Q->n_weights : 8u;
    const bool phrase = FAT && (flags & TF_PHRASE) != 0;
    const bool ph_leaf = FAT && (flags & TF_PHRASE_LEAF) != 0;
    const bool prox_ranker = (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY)
                                 ? nterms > 1
                                 : (ranker == MRK_RANK_WORDCOUNT || ranker == MRK_RANK_MATCHANY || ranker == MRK_RANK_FIELDMASK ||
                                    ranker == MRK_RANK_SPH04);
    HitCtx HC;
    HC.Q = Q;
    HC.spp = a.seg.spp;
    HC.hit = a.seg.pk_hit;
    HC.hbase = a.seg.pk_hbase;
    HC.flags = a.q_flags + oq;
    HC.nterms = nterms, HC.nw = nw;
    HC.ap0 = FAT ? Q->ph_atoms[0] : 0u, HC.ap1 = FAT ? Q->ph_atoms[1] : 0u, HC.ap2 = FAT ? Q->ph_atoms[2] : 0u, HC.ap3 = FAT ? Q->ph_atoms[3] : 0u;
    const uint32_t ph_mask = ph_leaf ? Q->ph_mask : 0u;
    HC.nph = !FAT ? 0u : phrase ? nterms : (uint32_t)__popc(ph_mask);
    HC.span = !FAT || HC.nph < 2 ? 0u : Q->ph_atoms[HC.nph - 1] - Q->ph_atoms[0];
    HC.px_dist = FAT ? Q->px_dist : 0u;
    HC.ranker = ranker;
    HC.fw = L.fw;
    HC.max_qpos = (int)Q->max_qpos, HC.n_qwords = (int)Q->n_qwords;
    HC.inline_hits = inline_hits;
    HC.multi_and = (flags & TF_MULTIAND) != 0 && !phrase;
    HC.quorum_hits = (flags & TF_QUORUM_HITS) != 0;
    HC.termpos = FAT && (flags & TF_TERMPOS) != 0;
    HC.order = FAT && (flags & TF_ORDER) != 0;
    HC.apack = (uint64_t)(HC.ap0 & 0xFFFFu) | ((uint64_t)(HC.ap1 & 0xFFFFu) << 16) | ((uint64_t)(HC.ap2 & 0xFFFFu) << 32) | ((uint64_t)(HC.ap3 & 0xFFFFu) << 48);
    HC.dupes = (flags & TF_DUPES) != 0 && (ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_PROXIMITY);

    const uint32_t* __restrict__ d = MQ.data + (uint64_t)c * (MQ_PLANES * 64) + lane;
    const bool valid = lane < n;
    const uint32_t rowid = d[0], fa = d[128];
    const float tfidf = __uint_as_float(d[64]);
    const uint32_t r0 = d[192], r1 = d[256], r2 = d[320], r3 = d[384];
    bool is_live = valid;
    uint32_t fields = fa & 0xffu;
    int rk = 0;
    if (valid) {
      const uint32_t all_slots = (1u << (nterms < (uint32_t)MAX_PROX_TERMS ? nterms : (uint32_t)MAX_PROX_TERMS)) - 1u;
      const uint32_t smask = (fa >> 8) & all_slots;
      if (FAT) {
        const uint32_t pmask = phrase ? all_slots : (ph_leaf && (smask & ph_mask) == ph_mask) ? ph_mask : 0u;
        bool found = false;
        uint32_t ffield = 0;
        hit_pass(HC, r0, r1, r2, r3, smask, pmask, prox_ranker, found, ffield, rk);
        if (phrase) {
          is_live = found;
          fields = 1u << ffield; // the doc's field mask comes from its first occurrence (searchnode.cpp:3836)
        }
      } else
        rk = hit_rank_plain(HC, r0, r1, r2, r3, smask);
    }
    // the match: weight, pruning bin, candidate buffer (emit_match of scan_pk_kernel)
    bool push = false;
    uint64_t key = 0;
    if (is_live) {
      ++total;
      uint32_t weight;
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
            if (fields & (1u << f)) rsum += (uint32_t)Q->weights[f];
        const int32_t bm = (int32_t)((tfidf + 0.5f) * 1000.0f);
        weight = ranker == MRK_RANK_PROXIMITY ? rsum : (uint32_t)bm + rsum * 1000u;
      }
      weight *= Q->index_weight; // MatchExtended, sphinx.cpp:12220
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
  if (which)
    hipLaunchKernelGGL(rank_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(rank_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
}

} // namespace mrk
