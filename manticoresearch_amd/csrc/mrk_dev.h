// mrk_dev.h -- plain structs shared by the host planner and the gfx950 kernels.
#pragma once
#include <stdint.h>

#include "../../include/mrk.h"

namespace mrk {

constexpr int DEVBLK = 128;      // docs per device block (= 1..4 skiplist blocks)
constexpr int WAVES = 4;         // waves per workgroup
constexpr int WG = 64 * WAVES;   // threads per workgroup
constexpr int T0_BLOCKS = WAVES; // driver-term blocks per tile (one per wave)
constexpr int TILE = T0_BLOCKS * DEVBLK;
constexpr int SLOTS = 16;        // other-term blocks decoded per pass
constexpr int KCAP = MRK_MAX_K;  // top-K capacity
constexpr int CAND = 2 * KCAP;   // candidate buffer (keys) per workgroup
// worst-case doclist entry: 5 (rowid delta) + 5 (hits) + 5 (fieldmask) + 10 (hitlist offset delta)
constexpr int MAX_DOC_BYTES = 25;
constexpr int STAGE_BYTES = ((DEVBLK * MAX_DOC_BYTES + 1 + 15 + 15) / 16) * 16; // 3232

// Device-resident segment (all pointers are HBM)
struct DevSegment {
  const uint8_t* spd;
  const uint8_t* spp;
  const uint32_t* blk_base; // per device block: SkiplistEntry_t::m_tBaseRowIDPlus1
  const uint64_t* blk_off;  // absolute .spd offset of the block's first doclist entry
  const uint64_t* blk_hit;  // SkiplistEntry_t::m_iBaseHitlistPos
  uint64_t spd_len;
  uint64_t spp_len;
  uint32_t rowid_base;
  uint32_t inline_hits;
  // packed doclists (lossless load-time transcode of .spd, see mrk_pack.cpp); NULL if absent
  const uint32_t* pk_base;  // per block: first possible rowid (= last rowid of the previous block + 1)
  const uint32_t* pk_doff;  // per block: word offset of its bit-packed deltas in pk_delta
  const uint8_t* pk_w;      // per block: bits per delta (0..16), or PK_WIDE for 32-bit deltas
  const uint32_t* pk_delta; // delta arena
  const uint32_t* pk_attr;  // per block 64 words: tf[l] | tf[l+64]<<8 | fields[l]<<16 | fields[l+64]<<24
  const uint8_t* pk_attr1;  // per doc slot (block*128 + i): tf nibble (15 = see pk_attr) | fields nibble; NULL if > 4 fields
  const uint16_t* pk_attr2; // per doc slot (block*128 + i) in slot order: tf | fields << 8, filled for keywords with a bitmap, cut after the last of them; NULL = not built
  const uint64_t* pk_exc;   // tf exceptions (tf >= 255): rowid<<32 | tf, sorted per term
  const uint32_t* pk_hit;   // per doc slot (block*128 + i): inlined hit, or hitlist offset from pk_hbase[block]
  const uint64_t* pk_hbase; // per block: .spp position of the block's first hitlist
  const uint32_t* dead;     // dead-row bitmap (DeadRowMap_c layout, padded to whole windows) or NULL
  // dense terms: doc-set bitmaps (one 2048-rowid window = 64 words) + rank directory (docs before each
  // 256-rowid group); a doc's rank is its slot in the packed arrays (pk_attr / pk_hit)
  const uint32_t* bm;
  const uint32_t* bm_dir;
  uint32_t n_windows;       // windows per bitmap = ceil(total_docs / 2048)
  const uint32_t* attrs;    // row-wise attributes (.spa rows, attr_stride dwords each) or NULL
  uint32_t attr_stride;
  const uint8_t* blobs;     // blob pool (.spb / m_dBlobs) for MVA filters or NULL; every row's blob row was bounds-checked at load
};

struct DevFilter { // mrk_filter with the values inline
  uint32_t kind;                  // MRK_FILTER_* | exclude << 8 | has_equal_min << 9 | has_equal_max << 10 | open_left << 11 | open_right << 12
  uint32_t item, shift, bits;     // dword of the row, bit offset inside it, width (64 = two dwords)
  uint32_t n_values;
  uint32_t mva;                   // 0 = a row attribute; else MVA width in bits | all << 8 | blob attribute id << 16 | blob attributes of the row << 24
  int64_t lo, hi;                 // RANGE
  int64_t values[MRK_MAX_FILTER_VALUES];
};

// ISphFilter::Eval over the row's attributes: Filter_Values (binary search in the reference, <= 8 values here),
// Filter_Range (EvalRange, sphinxfilter.h:130-143), FilterNot for m_bExclude; Filter_And over the query's filters
__device__ __forceinline__ bool row_passes_filters(const DevSegment& seg, const DevFilter* __restrict__ fl, uint32_t n, uint32_t rowid) {
  const uint32_t* __restrict__ row = seg.attrs + (uint64_t)rowid * seg.attr_stride;
  bool ok = true;
  for (uint32_t i = 0; i < n; ++i) {
    const DevFilter& F = fl[i];
    if (F.mva) { // a multi-value attribute: sorted values in the row's blob row (attribute.cpp:495-513)
      const uint32_t w = (F.mva & 0xffu) >> 3, all = (F.mva >> 8) & 1u, id = (F.mva >> 16) & 0xffu, nblob = F.mva >> 24;
      const uint8_t* __restrict__ br = seg.blobs + ((uint64_t)row[2] | ((uint64_t)row[3] << 32)); // sphGetBlobRowOffset: the 2nd attribute
      const uint32_t sz = br[0] == 0 ? 1u : br[0] == 1 ? 2u : 4u;
      auto rd = [](const uint8_t* p, uint32_t nb) {
        uint64_t x = 0;
        for (uint32_t i = 0; i < nb; ++i) x |= (uint64_t)p[i] << (8 * i);
        return x;
      };
      const uint64_t l1 = rd(br + 1 + id * sz, sz), l0 = id ? rd(br + 1 + (id - 1) * sz, sz) : 0;
      const uint8_t* __restrict__ data = br + 1 + nblob * sz + l0;
      const uint32_t nv = (uint32_t)((l1 - l0) / w);
      auto val = [&](uint32_t i) -> int64_t { return w == 4 ? (int64_t)(uint32_t)rd(data + 4 * i, 4) : (int64_t)rd(data + 8ull * i, 8); };
      bool pass = false;
      const bool eq_min = (F.kind >> 9) & 1u, eq_max = (F.kind >> 10) & 1u;
      if (nv) {
        if ((F.kind & 0xffu) == MRK_FILTER_VALUES) {
          if (!all) { // MvaEval_Any
            for (uint32_t i = 0; i < nv && !pass; ++i) {
              const int64_t x = val(i);
              for (uint32_t k = 0; k < F.n_values; ++k) pass = pass || x == F.values[k];
            }
          } else { // MvaEval_All
            pass = true;
            for (uint32_t i = 0; i < nv && pass; ++i) {
              const int64_t x = val(i);
              bool in = false;
              for (uint32_t k = 0; k < F.n_values; ++k) in = in || x == F.values[k];
              pass = in;
            }
          }
        } else if (!all) { // MvaEval_RangeAny (sphinxfilter.h:203-233): binary search for the minimum, then the value at / after it
          int L = 0, R = (int)nv - 1;
          bool decided = false;
          while (L <= R) {
            const int mid = L + (R - L) / 2;
            const int64_t x = val((uint32_t)mid);
            if (F.lo > x)
              L = mid + 1;
            else if (F.lo < x)
              R = mid - 1;
            else {
              pass = eq_min || mid + 1 < (int)nv; // (the reference does not test that next value against the maximum)
              decided = true;
              break;
            }
          }
          if (!decided && L != (int)nv) {
            const int64_t x = val((uint32_t)L);
            pass = eq_max ? x <= F.hi : x < F.hi;
          }
        } else { // MvaEval_RangeAll: the least and the largest value inside the range
          const int64_t a = val(0), b = val(nv - 1);
          const int64_t lo = w == 4 ? (int64_t)(uint32_t)F.lo : F.lo, hi = w == 4 ? (int64_t)(uint32_t)F.hi : F.hi; // ( (T)m_iMinValue, (T)m_iMaxValue )
          pass = (eq_min ? a >= lo : a > lo) && (eq_max ? b <= hi : b < hi);
        }
      }
      if ((F.kind >> 8) & 1u) pass = !pass;
      ok = ok && pass;
      continue;
    }
    int64_t v; // sphGetRowAttr (sphinx.h:993-1014)
    if (F.bits == 64)
      v = (int64_t)((uint64_t)row[F.item] | ((uint64_t)row[F.item + 1] << 32));
    else if (F.bits == 32)
      v = (int64_t)row[F.item];
    else
      v = (int64_t)((row[F.item] >> F.shift) & ((1u << F.bits) - 1u));
    bool pass;
    if ((F.kind & 0xffu) == MRK_FILTER_VALUES) {
      pass = false;
      for (uint32_t k = 0; k < F.n_values; ++k) pass = pass || v == F.values[k];
    } else if ((F.kind & 0xffu) == MRK_FILTER_FLOATRANGE) { // Filter_FloatRange::Eval (sphinxfilter.cpp:286-289): both bounds, always
      const float fv = __uint_as_float((uint32_t)v), flo = __uint_as_float((uint32_t)F.lo), fhi = __uint_as_float((uint32_t)F.hi);
      const bool eq_min = (F.kind >> 9) & 1u, eq_max = (F.kind >> 10) & 1u;
      pass = (eq_min ? fv >= flo : fv > flo) && (eq_max ? fv <= fhi : fv < fhi);
    } else {
      const bool eq_min = (F.kind >> 9) & 1u, eq_max = (F.kind >> 10) & 1u, open_l = (F.kind >> 11) & 1u, open_r = (F.kind >> 12) & 1u;
      const bool min_ok = eq_min ? v >= F.lo : v > F.lo, max_ok = eq_max ? v <= F.hi : v < F.hi;
      pass = open_l ? max_ok : open_r ? min_ok : (min_ok && max_ok);
    }
    if ((F.kind >> 8) & 1u) pass = !pass;
    ok = ok && pass;
  }
  return ok;
}

// CSphQueryContext::m_pWeightFilter: Filter_WeightValues / Filter_WeightRange over the match weight (sphinxfilter.cpp:304-320)
__device__ __forceinline__ bool weight_passes_filters(const DevFilter* __restrict__ fl, uint32_t n, int32_t weight) {
  bool ok = true;
  const int64_t v = (int64_t)weight;
  for (uint32_t i = 0; i < n; ++i) {
    const DevFilter& F = fl[i];
    bool pass;
    if ((F.kind & 0xffu) == MRK_FILTER_VALUES) {
      pass = false;
      for (uint32_t k = 0; k < F.n_values; ++k) pass = pass || v == F.values[k];
    } else {
      const bool eq_min = (F.kind >> 9) & 1u, eq_max = (F.kind >> 10) & 1u;
      pass = (eq_min ? v >= F.lo : v > F.lo) && (eq_max ? v <= F.hi : v < F.hi);
    }
    if ((F.kind >> 8) & 1u) pass = !pass;
    ok = ok && pass;
  }
  return ok;
}

__device__ __forceinline__ bool row_is_dead(const DevSegment& seg, uint32_t rowid) {
  return (seg.dead[rowid >> 5] >> (rowid & 31u)) & 1u; // DeadRowMap_c::IsSet, killlist.h:39-46
}

constexpr uint32_t PK_WIDE = 0xFFu;
// A query's candidate counter and its threshold word are the two addresses every wave of the query hits (a returning atomic per
// publish, a read per few bursts): each sits in a 64-byte line of its own -- sixteen queries' counters in one line made the
// atomics of DIFFERENT queries serialize on that line
constexpr int QSTRIDE = 16; // dwords between consecutive queries' entries in ScanArgs::q_cand_n / q_tau_bin
constexpr int NBINS = 1024; // pruning histogram bins per query
constexpr uint32_t BIN_WEIGHT = 0, BIN_ROWID = 1;
constexpr uint32_t PN_TERM = 0, PN_AND = 1, PN_OR = 2, PN_MAYBE = 3, PN_ANDNOT = 4, PN_PHRASEFIX = 5, PN_QUORUM = 6, PN_ORDERFIX = 7, PN_NOTNEAR = 8; // prog[] opcodes
constexpr int QUORUM_EVENTS = 8; // keywords of a quorum node = doclists that can run dry and reorder its children
constexpr uint32_t TF_MULTIAND = 1; // the whole query is one ExtMultiAnd_T (or a single keyword)
constexpr uint32_t TF_BITMAP = 4;   // 2-keyword AND answered by the bitmap kernel (items are window ranges)
constexpr uint32_t TF_BTREE = 256;  // boolean tree evaluated on bitmap words by the window-driven tree kernel (items are window ranges)
constexpr uint32_t TF_ORDER = 128;      // the TF_PHRASE_LEAF node is a BEFORE operator (ExtOrder_c) over the keywords of ph_mask
constexpr uint32_t TF_TERMPOS = 64;     // some keyword carries a position modifier (ExtTermPos_T)
constexpr uint32_t TF_NOTNEAR = 512;    // the tree holds a NOTNEAR node (ExtNotNear_c) over keywords nn_a (must) and nn_b (not)
constexpr uint32_t TF_QUORUM_HITS = 32; // the root is an ExtQuorum_c: its hits sort by position WITHOUT the end flag
constexpr uint32_t TF_DUPES = 16;      // a keyword occurs more than once in the query (HasQwordDupes, sphinxsearch.cpp:4178)
constexpr uint32_t TF_PHRASE_LEAF = 8; // one PHRASE below other operators: ph_mask = its words' slots in t[]
constexpr uint32_t TF_PHRASE = 2;   // the whole query is one PHRASE: ph_atoms[] = atom positions in phrase order
constexpr uint32_t TF_FAT = TF_PHRASE | TF_PHRASE_LEAF | TF_TERMPOS | TF_ORDER | TF_NOTNEAR; // final ranking needs the full hit pass (rank_kernel<true>)
constexpr uint32_t TF_GEN = 1024;   // answered by the generic per-doc evaluator (mrk_keval.h): matches go to queue 2 with one reference per keyword
constexpr uint32_t TF_LCS_BY_KEYWORDS = 4096; // the weight bounds may take "one hit per keyword" for a proximity run's length (ctx key prox_bound_keywords; mrk_kprune.h, prox_bounds)
constexpr uint32_t TF_GEN_NEARN = 2048; // ... whose root is a NEAR over three and more operands: its folded hits' query position needs the probe launch
constexpr int PHRASE_STATES = 8;    // live FSMphrase_c states per doc (>= phrase span + 1)
constexpr int TREE_STACK = 4;       // evaluation stack depth of the tree program
constexpr int MAX_PASSES = 8; // driver keywords per query (size of the tree's candidate cover)
constexpr uint32_t QF_OVERFLOW = 1; // candidate list overflowed: the host reruns the query with a list that holds every doc
constexpr uint32_t QF_FSM = 2;      // more live phrase states than the kernel keeps: the query fails
constexpr uint32_t QF_ARENA = 4;    // the generic evaluator ran out of hit-list memory (ctx tunable gen_spill_mb): the query fails
constexpr int MAX_PROX_TERMS_ = 4;
constexpr int MAPCAP = 4096; // direct-map probe window (rowids) per decoded block

struct DevTerm {
  uint32_t blk_first; // index of the term's first block in blk_*[]
  uint32_t nblocks;
  uint32_t docs;
  uint32_t queried32; // queried-fields mask (low dword)
  float idf;
  uint32_t qpos;      // atom position
  uint64_t spd_end;   // doclist_off + doclist_len
  uint32_t exc_first; // tf exceptions of this term in pk_exc
  uint32_t exc_n;
  uint64_t bm_off;    // word offset of the term's bitmap in DevSegment::bm (~0 = none)
  uint64_t dir_off;   // word offset of its rank directory in DevSegment::bm_dir
  uint32_t tp_kind;   // MRK_TERMPOS_*: only docs with an acceptable hit hold the keyword, only acceptable hits travel on
  uint32_t tp_max;    // MRK_TERMPOS_LIMIT: largest acceptable position within the field
};


struct DevQuery {
  uint32_t n_terms; // terms sorted ascending by docs (ExtMultiAnd_T node order); [0] drives
  uint32_t ranker;
  uint32_t k;
  uint32_t n_weights;
  uint32_t index_weight;
  uint32_t item_first;
  uint32_t n_items;
  uint32_t bin_mode;  // BIN_WEIGHT / BIN_ROWID: what the pruning histogram is keyed on
  int32_t bin_lo;     // weight (or rowid) that maps to the edge of bin 0
  uint32_t bin_shift;
  uint32_t cand_cap;  // capacity of this query's candidate list
  uint64_t cand_off;  // its offset in the candidate arena
  // boolean trees (ExtAnd_c / ExtOr_c / ExtMaybe_c / ExtAndNot_c): a query may run as several
  // passes, one per driver keyword; all passes feed the logical query out_q
  uint32_t out_q;     // logical query slot (histogram, candidate list, totals)
  uint32_t n_nodes;   // 0 = plain N-way AND of t[]; else post-order program in prog[]
  uint32_t req_mask;  // keywords (bit = index in t[]) without which the tree cannot match
  uint32_t excl_mask; // keywords whose presence hands the doc to an earlier pass
  uint32_t tree_flags; // TF_*
  uint32_t prog[16];  // op | left node << 8 | right node << 16 | keyword << 24
  uint32_t ph_atoms[MAX_PROX_TERMS_]; // PHRASE: query positions of its words, in phrase order
  uint32_t n_filters;                 // attribute filters: all must pass (EarlyReject)
  DevFilter filters[MRK_MAX_FILTERS];
  uint32_t ph_mask;                   // TF_PHRASE_LEAF: keyword slots of the phrase's words
  // ExtQuorum_c: keyword slots, threshold, and the order of its children (4 bits per slot, 0xF = end) as a
  // function of the rowid: qr_ord[0] up to qr_row[0], qr_ord[i + 1] for rowids beyond qr_row[i] (the keyword whose
  // doclist ends there leaves m_dChildren by RemoveFast, searchnode.cpp:4478, 4531)
  uint32_t qr_mask, qr_thr, qr_n;
  uint32_t qr_row[QUORUM_EVENTS];
  uint32_t qr_ord[QUORUM_EVENTS + 1];
  uint32_t px_dist;                   // 0 = exact PHRASE; else PROXIMITY ('"a b"~N'): XQNode_t::m_iOpArg; bit 31: 'a NEAR/N b'
  uint32_t nn_a, nn_b, nn_dist;       // TF_NOTNEAR: keyword slots of the must / not side, the distance
  uint32_t max_qpos, n_qwords;        // ExtRanker_c::m_iMaxQpos (largest query position) / m_iQwords (distinct words)
  uint32_t gen_prog;                  // TF_GEN: index of the pass's GenProg
  uint32_t rowid_max;                 // cutoff: rows past this one (segment-local) never reach the ranker; 0xFFFFFFFF = no bound
  uint32_t n_wfilters;                // filters on the match weight (m_pWeightFilter): all must pass, else the match is not one
  DevFilter wfilters[MRK_MAX_FILTERS];
  int32_t weights[32];
  DevTerm t[MRK_MAX_AND_TERMS];
};

struct DevItem {
  uint32_t query;
  uint32_t blk_begin; // driver-term block range [begin, end); window range for the bitmap-driven kernels
  uint32_t blk_end;
  uint32_t kind;      // window-range items: 0 = scan_bm_kernel, 1 = scan_bt_kernel
};

// candidate key: bigger = better under MatchRelevanceLt_fn (weight desc, rowid asc)
__host__ __device__ inline uint64_t make_key(int32_t weight, uint32_t rowid) {
  return ((uint64_t)((uint32_t)weight ^ 0x80000000u) << 32) | (uint32_t)(~rowid);
}
__host__ __device__ inline int32_t key_weight(uint64_t k) { return (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u); }
__host__ __device__ inline uint32_t key_rowid(uint64_t k) { return ~(uint32_t)k; }

// Matched docs of hit-ranked queries travel from the scan kernels to rank_kernel through an HBM queue of 64-entry chunks
// (structure of arrays: per chunk MQ_PLANES rows of 64 dwords -- rowid, tfidf sum, fields | contributing keywords << 8,
// one packed-array reference per keyword -- each written and read as one coalesced 256-B row)
constexpr int MQ_PLANES = 3 + MAX_PROX_TERMS_;
constexpr int MQ_SHARDS = 64; // chunk allocators per queue (workgroup b uses shard b % 64): a hot atomic address serializes the producers
constexpr int MQ_BATCH = 4;   // chunks a wave reserves per atomic
struct MatchQueue {
  uint32_t* data;  // [MQ_SHARDS * cap][MQ_PLANES][64]
  uint32_t* hdr;   // [MQ_SHARDS * cap] pass index | valid entries << 24
  uint32_t* count; // [MQ_SHARDS] chunks handed out per shard (past cap: never written, the query flagged QF_OVERFLOW)
  uint32_t cap;    // chunks per shard; shard s owns chunks [s * cap, (s + 1) * cap)
};

// ---- the generic per-doc evaluator (mrk_keval.h): the reference's evaluation tree, one node per entry, post-order
constexpr int GEN_MAX_NODES = 24;
constexpr int MQ_GEN_PLANES = 1 + MRK_MAX_AND_TERMS; // queue 2: rowid + one packed-array reference per keyword slot
constexpr uint32_t GN_TERM = 0, GN_MULTIAND = 1, GN_AND = 2, GN_OR = 3, GN_MAYBE = 4, GN_ANDNOT = 5, GN_PHRASE = 6, GN_PROX = 7, GN_NEAR = 8, GN_QUORUM = 9,
                   GN_ORDER = 10, GN_NOTNEAR = 11, GN_UNIT = 12;
struct GenNode {
  uint8_t kind;     // GN_*
  uint8_t n_kids;
  uint8_t flags;    // AND: 1 = m_bQPosReverse; MULTIAND: 2 = some keyword is field-limited (hits are tested against the masks)
  uint8_t n_words;  // PHRASE / PROX: words
  uint16_t npl, npr; // AND: node positions its sides' hits are relabelled with (0 = keep)
  uint8_t kid[8];   // child node indices; TERM: [0] = keyword slot; MULTIAND / QUORUM: keyword slots; PHRASE / PROX / NEAR: [0] = the operands' AND chain
  uint8_t aux[8];   // MULTIAND: node position per keyword; PHRASE / PROX: the words' keyword slots in phrase order; UNIT: [0] = the boundary keyword's slot (0xFF: none)
  int32_t opt;      // distance / threshold
  uint32_t pad;
};
struct GenProg {
  uint32_t n_nodes, pad[7];
  GenNode nodes[GEN_MAX_NODES];
};
struct GenHit { // ExtHit_t without the rowid (sphinxint.h:725-743)
  uint32_t hitpos;
  uint16_t qpos, nodepos, spanlen, matchlen;
  uint32_t weight;
};
struct GenArgs {
  const GenProg* progs;
  GenHit* lane_arena;  // [lanes of the evaluator's grid][lane_hits]
  uint32_t lane_hits, n_lanes;
  GenHit* spill;       // lists that do not fit the lane's slice; handed out by atomicAdd, released with the launch
  unsigned long long spill_cap;
  unsigned long long* spill_used;
  // NEAR over 3+ operands (FSMmultinear_c::m_uFirstQpos is never reset): per logical query, for each query position v < 64 the
  // first rowid whose evaluation inserted an operand hit with that position; phase 1 = the probe launch that fills it
  uint32_t* near_tab;
  uint32_t phase, pad2;
};

struct ScanArgs {
  DevSegment seg;
  const DevQuery* queries;
  const DevItem* items;
  uint64_t* item_cand; // [n_items][KCAP]
  uint32_t* item_cnt;  // [n_items]
  uint64_t* q_total;   // [n_queries]
  uint64_t* q_tau;     // [n_queries] running K-th best key (lower bound), atomicMax
  uint32_t n_items;
  // packed path: global pruning histograms + per-query candidate lists
  uint32_t* q_hist;    // [n_queries][NBINS]
  uint32_t* q_cand_n;  // [n_queries * QSTRIDE]
  uint32_t* q_flags;   // [n_queries]
  uint32_t* q_tau_bin; // [n_queries * QSTRIDE] running pruning threshold (bin index), atomicMax
  uint64_t* cand;      // candidate arena
  // hit-ranked queries: histogram of the matches' LOWER weight bounds and its threshold bin (mrk_kprune.h, prox_bounds); NULL = no pruning in front of the hit pass
  uint32_t* q_hist_lb; // [n_queries][NBINS]
  uint32_t* q_hist_lb2; // [n_queries][NBINS]: the matches inside the threshold bin by (weight offset, rowid slice), mrk_kprune.h
  uint32_t* q_tau_lb;  // [n_queries * QSTRIDE]: (threshold bin << 10) | threshold slot of the second level
  MatchQueue mq[3];    // [0] plain boolean trees, [1] queries with PHRASE / PROXIMITY / BEFORE nodes or position modifiers, [2] TF_GEN
  GenArgs gen;
};

// Top-K selection over the candidate lists (mrk_select.hip), three launches on the batch's stream:
//   sel_tau_kernel    one wave per query: final threshold bin of its histogram, number of 2048-key slices of its list
//   sel_filter_kernel a grid over ALL slices of ALL queries: keys whose bin reaches the threshold are compacted to the
//                     front of their own slice (in place: a slice belongs to one workgroup), the count goes to slice_cnt
//   sel_sort_kernel   one workgroup per query: gathers the slices' survivors (about K + the threshold bin's population),
//                     sorts, emits the best K
// A query's selection cost no longer depends on the length of its candidate list (one workgroup used to walk it all).
constexpr int SEL_SLICE = 2048;  // keys per slice
constexpr int SEL_MAXQ = 4096;   // queries per launch group (the filter kernel keeps their slice prefix in LDS)
struct SelectArgs {
  const DevQuery* queries;
  const uint32_t* q_hist;
  const uint32_t* q_cand_n;
  uint64_t* cand;     // (the filter pass compacts each slice in place)
  uint32_t n_queries;
  uint32_t rowid_base;
  uint64_t* out_keys; // [n_queries][KCAP], sorted descending
  uint32_t* out_cnt;
  uint32_t* sel_tau;    // [n_queries] final threshold bin
  uint32_t* sel_nslice; // [n_queries] slices of the query's list in use
  uint32_t* slice_cnt;  // survivors per slice; slice j of query q sits at (cand_off >> 11) + q + j
  uint32_t max_slices;  // upper bound of the batch's slices in use (sizes the filter grid)
  uint32_t rowid_hi;    // largest global rowid of the segment (rowid_base + docs - 1) and the bits of docs - 1: the sort pass
  uint32_t rowid_bits;  // orders the survivors of the threshold bin by (weight, rowid_hi - rowid) in sub-bins
  // The sort pass also hands the batch's results to the host: it writes them straight into pinned host memory (stores over
  // PCIe from the kernel).  A batch used to queue five device-to-host copies behind its selection; on MI355X those go through
  // the SDMA engine, which serves the copies of ALL streams in order -- the next batch's descriptor upload waited behind the
  // previous batch's result copies, and a 12.5 M-doc step spent a third of its time with the GPU idle.
  const uint64_t* q_total; // [n_queries] matches counted by the scan
  const uint32_t* q_flags; // [n_queries] QF_*
  uint64_t* h_keys;        // pinned host [n_queries][KCAP] or NULL (the batch feeds a shard exchange: rows stay on the device)
  uint32_t* h_cnt;         // pinned host [n_queries] or NULL
  uint64_t* h_total;       // pinned host [n_queries] or NULL
  uint32_t* h_flags;       // pinned host [n_queries] or NULL
  uint32_t* h_cand_n;      // pinned host [n_queries] or NULL
  uint64_t* rows_dst;      // device [n_queries][ROW_WORDS] or NULL: the exchange rows (keys zero-padded | count | total_found or MRK_ROW_RERUN)
  const uint32_t* declined; // unused by the kernel (declined queries' rows are rewritten by pack_rows_kernel)
};

// One launch instead of two copies and a memset in front of every scan: query and work-item descriptors are read from the
// batch's pinned staging memory by the kernel itself (no SDMA engine between two scans), the per-query scan state is cleared.
struct PrepArgs {
  uint32_t* dst[2];
  const uint32_t* src[2]; // pinned host memory
  uint32_t n4[2];         // dwords
  uint32_t* zero;
  uint32_t zero_n4;
};
void launch_prep(const PrepArgs& a, void* stream);
// slice_cnt entries a batch of n queries with cand_total candidate slots needs
inline uint64_t sel_slice_slots(uint64_t cand_total, uint64_t n) { return (cand_total >> 11) + n + 2; }

struct MergeArgs {
  const uint64_t* in_keys;  // lists of <= KCAP keys
  const uint32_t* in_cnt;
  const uint32_t* list_first; // per query: first list index   (NULL => strided layout below)
  const uint32_t* list_n;     // per query: number of lists
  uint32_t n_lists;           // strided layout: list l of query q is at l*n_queries + q
  uint32_t n_queries;
  const uint32_t* k_per_query; // NULL => k
  uint32_t k;
  uint64_t* out_keys;          // [n_queries][KCAP], sorted descending
  uint32_t* out_cnt;
  // row layout (shard exchange): one row of ROW_WORDS u64 per list = KCAP keys | count | total_found;
  // in_rows replaces in_keys / in_cnt, out_rows replaces out_keys / out_cnt (totals are summed)
  const uint64_t* in_rows;
  uint64_t* out_rows;
};
// Merge of the shards' result rows, up to 8 lists per query (mrk_select.hip): every list arrives sorted, so the lists are merged
// pairwise by bitonic MERGES in LDS (top-K of two sorted K-lists = elementwise max of one against the other reversed, then ten
// half-cleaner stages) -- three rounds for eight shards instead of sorting 8 K keys from scratch.
struct MergeRowsArgs {
  const uint64_t* in_rows; // list l of query q: in_rows[(l * list_stride + q) * ROW_WORDS]
  uint32_t n_lists;        // <= 8
  uint32_t list_stride;    // rows between consecutive lists (>= n_queries: a rank's receive buffer is sized for the largest slice)
  uint32_t n_queries;
  uint32_t k;
  uint64_t* out_rows;      // merged row of query q -> out_rows[(out_first + q) * ROW_WORDS] (device or pinned host memory)
  uint32_t out_first;
  uint32_t* flags_any;     // device dword pair or NULL: [0] = 1 if any merged row carries MRK_ROW_RERUN, [1] = ... MRK_ROW_DECLINED
};
void launch_merge_rows(const MergeRowsArgs& a, void* stream);
constexpr int ROW_WORDS = MRK_ROW_WORDS;
constexpr uint64_t ROW_RERUN = MRK_ROW_RERUN, ROW_DECLINED = MRK_ROW_DECLINED, ROW_FLAG_MASK = MRK_ROW_RERUN | MRK_ROW_DECLINED;

struct PackRowsArgs {
  const uint64_t* keys;   // [n][KCAP]
  const uint32_t* cnt;    // [n]
  const uint64_t* total;  // [n]
  uint64_t* rows;         // [n][ROW_WORDS]
  const uint32_t* flags;  // [n] QF_* of the scan (NULL = none): a flagged query's row is poisoned, not trusted
  const uint32_t* declined; // [n] != 0: the planner declined the query on this segment (NULL = none declined)
  uint32_t n;
};
void launch_pack_rows(const PackRowsArgs& a, void* stream);

void launch_scan(const ScanArgs& a, void* stream);
void launch_scan_pk(const ScanArgs& a, uint32_t max_terms, bool prox, bool tree, bool ext, void* stream, bool gen = false);
constexpr int MAX_PROX_TERMS = MAX_PROX_TERMS_; // keywords whose hit streams the hit kernel merges per doc
void launch_scan_bm(const ScanArgs& a, void* stream); // a.items: (query, window range) work items
void launch_scan_bt(const ScanArgs& a, void* stream); // the same for TF_BTREE passes (mrk_scan_bt.hip)
// final ranking of the queued matches (mrk_rank.hip): a persistent grid drains queue `which` of a.mq (2 = the generic evaluator)
constexpr int GEN_GRID = 512; // its workgroups: lane arenas are sized for GEN_GRID * WG lanes
void launch_rank(const ScanArgs& a, int which, void* stream);
void launch_select(const SelectArgs& a, void* stream);
void launch_merge(const MergeArgs& a, void* stream);

} // namespace mrk
