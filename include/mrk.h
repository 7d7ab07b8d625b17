/*
 * mrk.h -- C-ABI of the MI355X-native match -> rank -> top-K path ("mrk").
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point names the reference interface it stands in for
 * (paths relative to the Manticore 3.6.1 tree, src/...).
 *
 *   mrk_segment_create   <- what DiskIndexQwordSetup_c binds per index: .spd/.spp/.spe
 *                           readers + skiplist_block_size + the dictionary lookup result
 *                           CSphDictEntry (sphinx.cpp:326-354, 12953-13080; sphinx.h:542-552)
 *   mrk_query (tree)     <- XQNode_t / XQKeyword_t as handed to sphCreateRanker
 *                           (sphinxquery.h:134-286; sphinxsearch.cpp:4167)
 *   mrk_batch_submit     <- sphCreateRanker + the MatchExtended loop
 *                           (sphinxsearch.cpp:4167-4380; sphinx.cpp:12190-12269):
 *                           ExtNode_i::Create, IDF setup, GetMatches() until exhausted
 *   mrk_result           <- what the caller reads back from ISphMatchSorter:
 *                           Flatten() order + GetTotalCount() (sphinxsort.h:39-133,
 *                           sphinxsort.cpp:627-641, 724) and CSphQueryStats (sphinx.h:2697-2705)
 *   mrk_topk_merge       <- CSphMatchQueue::MoveTo across chunks / shards
 *                           (sphinxsort.cpp:681-710; sphinxrt.cpp:5945-5981)
 *   mrk_idf              <- the IDF block of sphCreateRanker (sphinxsearch.cpp:4317-4361)
 *
 * Threading: a mrk_ctx owns one HIP device, its streams and ONE internal submission thread: every
 * entry point that reaches the HIP runtime posts its work there and sleeps until it has run, so
 * nothing of HIP executes on the caller's stack -- the reference runs rankers on 128 KB coroutine
 * stacks (coroutine.cpp:47) -- and all of a context's host state is touched by one thread only.
 * A mrk_batch is used by one thread at a time; different batches (and segments) of one context
 * may be driven from different threads concurrently.  mrk_batch_wait polls the batch's stream
 * and steps aside for other threads' submits instead of parking the submission thread.
 * MRK_INLINE_HIP=1 in the environment runs everything on the calling thread instead.
 *
 * Errors: every function returns MRK_OK (0) or a negative code; mrk_last_error() gives
 * the message for the calling thread (the reference's convention: no exceptions, error
 * string on the side -- sphinxsearch.cpp:4377-4378).
 */
#ifndef MRK_H
#define MRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRK_OK 0
#define MRK_E_INVAL (-1)       /* bad argument */
#define MRK_E_UNSUPPORTED (-2) /* query shape / ranker the device path does not cover (yet) */
#define MRK_E_HIP (-3)         /* HIP runtime failure */
#define MRK_E_NOMEM (-4)
#define MRK_E_FORMAT (-5)      /* malformed index bytes */

#define MRK_INVALID_ROWID 0xFFFFFFFFu
#define MRK_MAX_AND_TERMS 8    /* device N-way AND width */
#define MRK_MAX_K 1024         /* device top-K capacity (max_matches default is 1000) */
#define MRK_ALL_FIELDS 0xFFFFFFFFu

/* ESphRankMode order (sphinx.h) */
enum {
  MRK_RANK_PROXIMITY_BM25 = 0,
  MRK_RANK_BM25 = 1,
  MRK_RANK_NONE = 2,
  MRK_RANK_WORDCOUNT = 3,
  MRK_RANK_PROXIMITY = 4,
  MRK_RANK_MATCHANY = 5,
  MRK_RANK_FIELDMASK = 6,
  MRK_RANK_SPH04 = 7
};

/* XQOperator_e subset (sphinxquery.h) */
enum { MRK_OP_TERM = 0, MRK_OP_AND = 1, MRK_OP_OR = 2, MRK_OP_MAYBE = 3, MRK_OP_ANDNOT = 4, MRK_OP_PHRASE = 5,
       MRK_OP_PROXIMITY = 6, /* '"a b c"~N': opt = N */
       MRK_OP_BEFORE = 8, /* 'a << b << c' (ExtOrder_c): the children occur in this order inside one field */
       MRK_OP_QUORUM = 7, /* '"a b c"/N': opt = N (ExtQuorum_c; N = 1 / N >= words as the reference rewrites them) */
       MRK_OP_NEAR = 9,   /* 'a NEAR/N b' (ExtNWay_T<FSMmultinear_c>): opt = N; operands: keywords, phrases, nested NEARs; 3+ operands at the query root only */
       MRK_OP_NOTNEAR = 10, /* 'a NOTNEAR/N b' (ExtNotNear_c): opt = N */
       MRK_OP_SENTENCE = 11, /* 'a SENTENCE b' (ExtUnit_c): both sides inside one sentence of an index_sp = 1 index; the node's term_id = the
                                dictionary slot of the boundary keyword MAGIC_WORD_SENTENCE "\3sentence" (< 0: the index holds none: plain AND) */
       MRK_OP_PARAGRAPH = 12 /* 'a PARAGRAPH b': the same over MAGIC_WORD_PARAGRAPH "\3paragraph" */ };

enum { MRK_HITFMT_PLAIN = 0, MRK_HITFMT_INLINE = 1 }; /* ESphHitFormat */

typedef struct mrk_ctx mrk_ctx;
typedef struct mrk_segment mrk_segment;
typedef struct mrk_batch mrk_batch;

/* CSphDictEntry (sphinx.h:542-552): result of the dictionary lookup for one keyword */
typedef struct {
  uint64_t wordid;       /* m_uWordID (informational) */
  uint64_t doclist_off;  /* m_iDoclistOffset into .spd */
  uint64_t doclist_len;  /* m_iDoclistLength, bytes incl. the 0 terminator */
  uint64_t skiplist_off; /* m_iSkiplistOffset into .spe (valid iff docs > skiplist_block_size) */
  uint32_t docs;         /* m_iDocs */
  uint32_t hits;         /* m_iHits */
} mrk_dict_entry;

/* One index segment (plain index / RT disk chunk) in the reference's v62 byte format */
typedef struct {
  const uint8_t* spd; /* doclists  (host memory; copied to HBM) */
  uint64_t spd_len;
  const uint8_t* spp; /* hitlists */
  uint64_t spp_len;
  const uint8_t* spe; /* skiplists */
  uint64_t spe_len;
  const mrk_dict_entry* dict; /* flat term table indexed by term id (stands in for .spi) */
  uint32_t n_terms;
  uint64_t total_docs;          /* m_iTotalDocuments */
  uint32_t skiplist_block_size; /* index setting; 32 (default) or 128 */
  uint32_t hit_format;          /* MRK_HITFMT_* */
  uint32_t n_fields;            /* schema full-text fields (<= 32 on the device path) */
  uint32_t rowid_base;          /* global docid = rowid_base + rowid for shard merges */
} mrk_segment_desc;

/* XQNode_t flattened: nodes[] + children[] (indices into nodes[]) */
typedef struct {
  int32_t op;          /* MRK_OP_* */
  int32_t n_children;
  int32_t first_child; /* offset into children[] */
  int32_t term_id;     /* leaf: dictionary slot; < 0 = keyword not in the dictionary.  SENTENCE / PARAGRAPH node: the boundary keyword's slot */
  int32_t atom_pos;    /* XQKeyword_t::m_iAtomPos */
  uint32_t field_mask; /* XQLimitSpec_t::m_dFieldMask, low dword */
  float boost;         /* XQKeyword_t::m_fBoost */
  int32_t opt;         /* XQNode_t::m_iOpArg */
  int32_t not_weighted;
  int32_t term_pos;      /* leaf: MRK_TERMPOS_* position modifier (ExtTermPos_T, searchnode.cpp:2259-2405) */
  int32_t field_max_pos; /* leaf: XQLimitSpec_t::m_iFieldMaxPos for MRK_TERMPOS_LIMIT ('@field[N] word') */
} mrk_node;

/* TermPosFilter_e as ExtNode_i::Create derives it (searchnode.cpp:875-878, 1145-1146): '^word' = START, 'word$' = END,
   both = STARTEND, a field position limit = LIMIT (and wins over the anchors) */
enum { MRK_TERMPOS_NONE = 0, MRK_TERMPOS_START = 1, MRK_TERMPOS_END = 2, MRK_TERMPOS_STARTEND = 3, MRK_TERMPOS_LIMIT = 4 };

/* One attribute filter (CSphFilterSettings, sphinx.h:2461-2496) over an integer attribute of the row-wise .spa storage,
   already resolved to the attribute's locator.  All filters of a query must pass (Filter_And); rejected rows never reach
   the ranker or the sorter and are not counted (ExtRanker_c::GetMatches -> CSphIndex_VLN::EarlyReject,
   sphinxsearch.cpp:1055-1064, sphinx.cpp:11903-11917). */
enum { MRK_FILTER_VALUES = 0, MRK_FILTER_RANGE = 1, MRK_FILTER_FLOATRANGE = 2 }; /* SPH_FILTER_VALUES, SPH_FILTER_RANGE, SPH_FILTER_FLOATRANGE */
#define MRK_MAX_FILTERS 2
#define MRK_MAX_FILTER_VALUES 8
typedef struct {
  int32_t kind;                 /* MRK_FILTER_* */
  int32_t bit_offset, bit_count; /* CSphAttrLocator::m_iBitOffset / m_iBitCount (1..32 inside one dword, or 64 dword-aligned) */
  int32_t exclude;              /* m_bExclude: the row passes iff the test fails */
  int32_t has_equal_min, has_equal_max, open_left, open_right; /* RANGE: m_bHasEqualMin/Max, m_bOpenLeft/Right */
  int64_t min_value, max_value; /* RANGE (SphAttr_t is signed 64-bit) */
  const int64_t* values;        /* VALUES: ascending (IFilter_Values::SetValues), <= MRK_MAX_FILTER_VALUES on the device */
  int32_t n_values;
  /* FLOATRANGE over a 32-bit float attribute (Filter_FloatRange, sphinxfilter.cpp:275-300): the row's dword read as a float
     (sphDW2F) against [fmin, fmax] with m_bHasEqualMin / Max; the reference's float filter has no open-sided form */
  float fmin, fmax;
  /* A multi-value attribute (SPH_ATTR_UINT32SET / INT64SET): mva_bits = 32 / 64 (0 = a plain row attribute).  Its values live,
     sorted, in the blob pool (mrk_segment_set_blobs) as blob attribute blob_attr_id of the row's n_blob_attrs
     (CSphAttrLocator::m_iBlobAttrId / m_nBlobAttrs; the row's blob offset is its second attribute, sphGetBlobRowOffset).
     VALUES: any of the doc's values is in the set (Filter_MVAValues_Any_c) or, mva_all, every one is (.._All_c);
     RANGE: some value inside [min, max] (MvaEval_RangeAny) or all of them (MvaEval_RangeAll); sphinxfilter.h:160-240,
     sphinxfilter.cpp:340-383.  A doc without values fails both forms. */
  int32_t mva_bits, mva_all, blob_attr_id, n_blob_attrs;
} mrk_filter;

/* CSphQuery fields that reach the ranker + the query tree */
typedef struct {
  const mrk_node* nodes;
  int32_t n_nodes;
  const int32_t* children;
  int32_t root;
  int32_t ranker;               /* MRK_RANK_* (CSphQuery::m_eRanker) */
  int32_t max_matches;          /* K (CSphQuery::m_iMaxMatches) */
  const int32_t* field_weights; /* CSphQueryContext::m_dWeights, NULL => 1 per field */
  int32_t n_weights;
  int32_t index_weight;         /* MatchExtended iIndexWeight, 0 => 1 */
  int32_t plain_idf;            /* CSphQuery::m_bPlainIDF */
  int32_t normalized_tfidf;     /* CSphQuery::m_bNormalizedTFIDF */
  int64_t total_docs_override;  /* CSphQueryContext::m_iTotalDocs (local_df), <= 0: segment's */
  const int64_t* local_docs;    /* per node: m_pLocalDocs override of term docs, < 0 none; or NULL */
  int32_t cutoff;               /* CSphQuery::m_iCutoff (0 = none): the first `cutoff` matches in rowid order are the result set.
                                   <= MRK_MAX_K, packed path, not next to a weight filter; costs a probe launch inside mrk_batch_submit.
                                   Per segment, as MatchExtended counts per disk index; an RT index counts ON across its RAM segments
                                   (PerformFullTextSearch, sphinxrt.cpp:6302-6380): the caller lowers the cutoff by each segment's total_found */
  const mrk_filter* filters;    /* CSphQuery::m_dFilters resolved against the schema; needs mrk_segment_set_attrs */
  int32_t n_filters;            /* <= MRK_MAX_FILTERS on the device */
  /* CSphQueryContext::m_pWeightFilter: filters on the match weight ('WHERE weight() >= N'; Filter_WeightValues /
     Filter_WeightRange, sphinxfilter.cpp:304-320), applied after index_weight and before the sorter (MatchExtended,
     sphinx.cpp:12220-12227); matches they drop are not counted.  kind VALUES / RANGE, exclude, has_equal_min / max;
     the locator fields are ignored.  All must pass. */
  const mrk_filter* weight_filters;
  int32_t n_weight_filters;     /* <= MRK_MAX_FILTERS on the device */
} mrk_query;

typedef struct {
  int32_t n;              /* matches returned, best first (<= max_matches) */
  int64_t total_found;    /* ISphMatchSorter::GetTotalCount() */
  const uint32_t* rowid;  /* n entries, valid until the batch is resubmitted/destroyed */
  const int32_t* weight;
  int32_t status;         /* MRK_OK or MRK_E_UNSUPPORTED for this query */
} mrk_result;

typedef struct {
  float scan_ms;          /* decode+intersect+score kernel, HIP-event time on the batch stream */
  float merge_ms;         /* top-K merge kernel */
  uint64_t algo_bytes;    /* sum over queries of doclist bytes of their terms (reference format) */
  uint64_t n_items;       /* work items (workgroups) launched by the scan kernel */
  uint64_t dev_bytes;     /* the same sum over the format the kernel actually read (packed or .spd) */
  uint32_t packed;        /* 1 = packed-doclist path, 0 = VLB-direct path */
  uint64_t n_cands;       /* packed path: candidates that survived in-scan pruning (all queries) */
  uint64_t n_items_bm;    /* of n_items: work items of the two-bitmap AND kernel (dense keywords) */
  float plan_ms;          /* host: query planning inside mrk_batch_submit */
  float submit_ms;        /* host: whole mrk_batch_submit call */
} mrk_batch_stats;

const char* mrk_last_error(void);

int mrk_ctx_create(int device, mrk_ctx** out);
/* Destroy a context's segments, batches and batchers BEFORE the context: their destructors run on its submission thread.
   The context counts them: with any still alive the call destroys nothing and returns MRK_E_INVAL (the message says how many);
   destroy them and call again.  (Until round 3 this was a comment only: a segment or batch destroyed after its context posted
   its destructor to a thread that no longer existed and waited forever.) */
int mrk_ctx_destroy(mrk_ctx* ctx);
/* tunables: "item_bytes" (work-item size target); "pack" (1 = build packed doclists at segment
   load, default); "path" (0 = packed when present, 1 = VLB-direct, 2 = packed only);
   "bitmap_inv" (keywords found in >= 1/bitmap_inv of a segment's docs also get a doc-set bitmap,
   used by the two-bitmap AND kernel; default 64, 0 = off; read at segment load and at submit);
   "attr_seq" (1 = keywords with a bitmap also get their tf / field bytes as two bytes per posting in slot order, which the two-bitmap
   AND kernel gathers from instead of the block decoder's interleaved words: the docs of a 128-byte line are then scored together and
   the line is fetched once, -2.7 % on the headline launch; +2 bytes per posting up to the last such keyword; default 1; read at
   segment load);
   "attr_nibbles" (1 = segments with <= 4 fields also get a one-byte tf/field plane for the bitmap kernel's gathers:
   28 % fewer bytes per dense x dense query, +14 % queries/s on the 100 M-doc bench, +1 byte per posting; default 0 --
   see DESIGN.md section 6; read at segment load);
   "bm_target_items" / "bt_target_items" (cap of the work items per launch the window ranges of the two-bitmap AND kernel / of the tree
   kernel over bitmap words are cut into, defaults 2^20 / 6144) and "bm_min_windows" (windows per work item of the AND kernel, default
   128: since work items of different queries interleave -- "item_order", a mask: 1 block scan, 2 bitmap AND, 4 bitmap trees, 8 also in
   batches that feed the hit pass; default 7 -- small items are the fast ones); "pk_min_items" (a batch with fewer block-scan work items
   has its block ranges cut finer, default 2048);
   "prox_prune" (1 = proximity rankers: matches whose weight upper bound cannot reach the top K skip the hit pass, default);
   "prox_bound_keywords" (0 = that bound takes a proximity run to be as long as the field's hits allow: always sound, default;
   1 = as long as the number of keywords in the field -- tighter, and sound where hits of different keywords at ONE position reach
   the ranker in query-position order: indexes whose field-end flag belongs to the position, which is what the reference's indexer
   writes (sphinx.cpp:22424-22430), or that carry no field-end flags; the caller's statement about the context's indexes, read at submit);
   "exchange_part" (1 = mrk_shard_exchange partitions the merge by query, default), "exchange_self_rccl" (1 = a one-rank exchange
   still goes through RCCL, default 0);
   "bt_cover_inv" (boolean trees whose candidate cover -- the keywords whose doc lists together hold every possible match --
   names >= 1/bt_cover_inv of the segment's docs are evaluated on doc-set bitmap words, 8192 rowids per step, instead of
   block by block; default 1024 -- a pure AND of keywords only from 1/32 --, 0 = never; read at submit);
   "bt_phrase" (1 = a root PHRASE / PROXIMITY whose rarest word holds >= 1/bt_cover_inv of the docs takes its candidates -- the AND of
   its words -- from the bitmap words too, default; 0 = block walk);
   "mq_max_chunks" (cap of a batch's match queue -- matched docs of hit-ranked queries on their way to the ranking kernel -- in
   chunks of 64 docs / 1792 bytes, default 2^20; queries that outgrow it are rerun one by one by mrk_batch_wait);
   "gen_lane_hits" / "gen_spill_mb" (the generic per-doc evaluator -- query shapes the specialised hit passes do not take: more
   than four keywords under a hit ranker, phrases of five and more words, BEFORE / NEAR / NOTNEAR over phrases, groups and
   quorums, several such nodes -- keeps each node's hit list of the doc at hand in HBM: 16-byte hits per evaluator lane
   (default 256 x 131072 lanes = 512 MB) and a shared area for longer lists (default 1024 MB); allocated with the first
   batch that holds such a query; a query that outgrows them fails with MRK_E_UNSUPPORTED, it is never truncated);
   returns MRK_E_INVAL for unknown keys */
int mrk_ctx_set(mrk_ctx* ctx, const char* key, int64_t value);

int mrk_segment_create(mrk_ctx* ctx, const mrk_segment_desc* desc, mrk_segment** out);
/* The host-only half of mrk_segment_create, no device needed: the descriptor's limits, every skiplist, and a walk of
   every doclist (entries decode, rowids ascend and stay below total_docs, hitlist offsets stay inside .spp, the
   terminator sits where the dictionary's doc count says).  MRK_OK, or the error mrk_segment_create would return --
   it runs the same checks on whatever the load-time transcode did not already walk, so that untrusted bytes never
   become an out-of-bounds device read. */
int mrk_segment_validate(const mrk_segment_desc* desc);
void mrk_segment_destroy(mrk_segment* seg);
/* Dead-row map of the segment (DeadRowMap_c, killlist.h:22-46: bit rowid&31 of DWORD rowid>>5), copied
   to the device; dead rows are dropped right after ranking, before the sorter, and do not count as found
   (CSphIndex_VLN::MatchExtended, sphinx.cpp:12213-12217).  bitmap = NULL clears the map.  The call waits
   for the context's running batches, so a search sees either the old or the new map. */
int mrk_segment_set_dead_rows(mrk_segment* seg, const uint32_t* bitmap, uint64_t n_rows);
/* Row-wise attribute storage (.spa: CSphRowitem rows[n_rows][stride], sphinx.cpp GetDocinfoByRowID) copied to HBM for the
   filters of mrk_query; rows = NULL drops it.  Waits for the context's running batches like mrk_segment_set_dead_rows. */
int mrk_segment_set_attrs(mrk_segment* seg, const uint32_t* rows, uint32_t stride_dwords, uint64_t n_rows);
/* The blob pool of the attribute storage (.spb file / RtSegment_t::m_dBlobs: per row a blob row = length-size byte, cumulative
   lengths, data; attribute.cpp:495-513) copied to HBM for MVA filters.  rows / stride / n_rows = the rows given to
   mrk_segment_set_attrs: every row's blob row is walked here once (offset, header and lengths inside the pool), so that
   untrusted bytes never become an out-of-bounds device read.  pool = NULL drops it. */
int mrk_segment_set_blobs(mrk_segment* seg, const uint8_t* pool, uint64_t pool_len, uint32_t n_blob_attrs, const uint32_t* rows,
                          uint32_t stride_dwords, uint64_t n_rows);
/* device bytes held, and the reference-format doclist bytes of one term */
uint64_t mrk_segment_device_bytes(const mrk_segment* seg);

/* ---- the batching front: ONE query per caller, common launches ----------------------------------------------------------
   searchd runs one ranker per (query x index) on a pool of workers (SearchHandler_c::RunLocalSearches, searchd.cpp:5596-5797);
   a ranker that submits a batch of one pays a launch chain per query.  A batcher (one per index / context) lets the workers'
   queries meet: mrk_batcher_search enqueues the caller's query and sleeps (mutex + condition variable: nothing of HIP on the
   caller's stack) until a driver thread has sent it down -- together with everything else that arrived while the device was
   busy with the launch before, up to max_batch queries of the same segment per mrk_batch_submit, three launches in flight --
   and copied its rows into the caller's own buffers (rowid_out / weight_out, cap entries; res->rowid / res->weight point
   there).  max_wait_us: how long an IDLE device waits for a batch to fill up (0 = launch at once; under load arrivals queue
   behind the running launch anyway).  A query the planner refuses with MRK_E_INVAL fails alone: the others of its launch are
   rerun one by one.  Returns MRK_OK with res->status = the query's own status (MRK_E_UNSUPPORTED = keep the CPU ranker). */
typedef struct mrk_batcher mrk_batcher;
typedef struct {
  uint64_t launches;   /* mrk_batch_submit calls */
  uint64_t queries;    /* queries answered */
  uint32_t max_batch;  /* most queries one launch carried */
  double submit_ms;    /* driver thread: time inside mrk_batch_submit, in all */
  double collect_ms;   /* ... inside mrk_batch_wait + handing the rows out */
  double flight_ms;    /* sum over launches of submit-return -> found complete */
} mrk_batcher_stats;
int mrk_batcher_create(mrk_ctx* ctx, uint32_t max_batch, uint32_t max_wait_us, mrk_batcher** out);
void mrk_batcher_destroy(mrk_batcher* b);
int mrk_batcher_search(mrk_batcher* b, mrk_segment* seg, const mrk_query* q, uint32_t* rowid_out, int32_t* weight_out, int32_t cap, mrk_result* res);
int mrk_batcher_stats_get(mrk_batcher* b, mrk_batcher_stats* out);

int mrk_batch_create(mrk_ctx* ctx, uint32_t max_queries, mrk_batch** out);
void mrk_batch_destroy(mrk_batch* b);
/* plan on host, copy descriptors, launch kernels, start the result copy; returns at once */
int mrk_batch_submit(mrk_batch* b, mrk_segment* seg, const mrk_query* queries, uint32_t n_queries);
/* block until the results of the last submit are in host memory */
int mrk_batch_wait(mrk_batch* b);
/* has the device finished the last submit?  MRK_OK = yes (mrk_batch_wait returns without blocking), 1 = not yet.  A stream
   query on the CALLING thread (no hop to the submission thread: a polling loop must not queue behind other callers'
   submits): call it from an ordinary thread, not from a coroutine stack. */
int mrk_batch_test(mrk_batch* b);
int mrk_batch_result(mrk_batch* b, uint32_t q, mrk_result* out);
int mrk_batch_stats_get(mrk_batch* b, mrk_batch_stats* out);
/* device-resident partial top-K of the last submit, for shard merges without a host hop:
   keys[q*MRK_MAX_K + i] = ((weight ^ 0x80000000) << 32) | ~(rowid_base + rowid), sorted
   descending; counts[q]; totals[q].  Valid after mrk_batch_wait. */
int mrk_batch_device_results(mrk_batch* b, const uint64_t** keys, const uint32_t** counts, const uint64_t** totals);

/* copy those three arrays into caller-owned device buffers (e.g. tensors handed to RCCL);
   any pointer may be NULL; synchronous */
int mrk_batch_export_device(mrk_batch* b, uint64_t* keys_dst, uint32_t* counts_dst, uint64_t* totals_dst);

/* Shard exchange in one buffer: a row of MRK_ROW_WORDS u64 per query = MRK_MAX_K keys (as above, zero
   padded) | count | total_found.  mrk_batch_export_rows writes the last submit's results as rows into a
   caller-owned DEVICE buffer [n_queries][MRK_ROW_WORDS] (e.g. the tensor handed to one RCCL all-gather) and
   returns when the copy is done; mrk_topk_merge_rows merges rows_all[n_lists][n_queries][MRK_ROW_WORDS]
   (device) into out_rows[n_queries][MRK_ROW_WORDS]: best k keys per query, totals added up
   (CSphMatchQueue::MoveTo, sphinxsort.cpp:681-710). */
#define MRK_ROW_WORDS (MRK_MAX_K + 2)
/* flag bits in a row's total_found word; mrk_topk_merge_rows ORs them through (the counts in the low 62 bits add up):
   MRK_ROW_RERUN    a shard's candidate list overflowed and the row left before mrk_batch_wait reran the query: call
                    mrk_batch_wait + mrk_batch_export_rows on that shard's batch and exchange / merge again;
   MRK_ROW_DECLINED a shard declined the query (MRK_E_UNSUPPORTED there): the merged row is not an answer. */
#define MRK_ROW_RERUN (1ull << 63)
#define MRK_ROW_DECLINED (1ull << 62)
int mrk_batch_export_rows(mrk_batch* b, uint64_t* rows_dst);
/* standing order: every later submit also writes its rows to rows_dst (device, [max_queries][MRK_ROW_WORDS]) on the
   batch's own stream right behind the selection kernel -- valid after mrk_batch_wait, no device work at collection
   time.  NULL cancels it. */
int mrk_batch_set_rows_dst(mrk_batch* b, uint64_t* rows_dst);
/* (A query whose candidate list overflowed is rerun by mrk_batch_wait; a row that left through the standing export
   before that carries no keys and MRK_ROW_RERUN: ask that shard again, see above.) */
/* record a caller-owned hipEvent_t on the batch's stream, i.e. behind everything the last submit queued there
   (selection, standing rows export, result copies): lets another stream (RCCL's) wait for the rows without the host */
int mrk_batch_record_event(mrk_batch* b, void* hip_event);
int mrk_topk_merge_rows(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                        uint64_t* out_rows);

/* Stream-ordered form for pipelined shard merges (nothing blocks the host): the merge is queued on the context's
   merge stream behind `wait_event` (a hipEvent_t recorded by the producer of rows_all, e.g. on the RCCL stream;
   NULL = none) and completion is marked in `slot` (0..MRK_MERGE_SLOTS-1); mrk_merge_wait blocks until that slot's
   work is done.  out_rows may be device memory or PINNED HOST memory (hipHostMalloc / a pinned torch tensor): the
   merge kernel then writes the merged rows straight into host memory (multi-MiB device-to-host hipMemcpyAsync calls
   were seen to block the calling thread for 0.1-0.5 ms; a kernel writing over PCIe does not). */
#define MRK_MERGE_SLOTS 8
int mrk_topk_merge_rows_async(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                              uint64_t* out_rows, void* wait_event, uint32_t slot);
int mrk_merge_wait(mrk_ctx* ctx, uint32_t slot);

/* ------------------------------------------------------------------------------------
 * The shard exchange inside the library (one process per GPU, segments = rowid ranges): RCCL over xGMI, loaded at run
 * time; a C / C++ host needs no Python for it.  Stands in for SearchHandler_c::SetupLocalDF (searchd.cpp:5869-5990) and
 * the merge of per-chunk sorters (sphinxsort.cpp:681-710, sphinxrt.cpp:5945-5950) across the local indexes of one node.
 *
 *   rank 0: mrk_comm_unique_id(id), ship the MRK_COMM_ID_BYTES to the other ranks by any means (file, socket, MPI ...);
 *   every rank: mrk_comm_init(ctx, id, n_ranks, rank)            -- collective, like ncclCommInitRank;
 *   once per index:  mrk_comm_allreduce_i64(ctx, docs, n)       -- per-keyword document counts and the document total,
 *                    summed over the shards: the values of mrk_query.local_docs / total_docs_override (local_df);
 *   per batch:       mrk_batch_set_rows_dst(batch, rows) once, then after each mrk_batch_submit
 *                    mrk_shard_exchange(ctx, batch, rows, n_queries, k, out_rows, slot)
 *                    queues event -> all-gather of the rows -> merge kernel -> out_rows (device or pinned host memory)
 *                    as one stream-ordered chain and returns at once; mrk_merge_wait(ctx, slot) blocks until out_rows is
 *                    written.  Rows carry MRK_ROW_RERUN / MRK_ROW_DECLINED through the merge (see above).
 * ---------------------------------------------------------------------------------- */
#define MRK_COMM_ID_BYTES 128
int mrk_comm_unique_id(uint8_t id_out[MRK_COMM_ID_BYTES]);
int mrk_comm_init(mrk_ctx* ctx, const uint8_t id[MRK_COMM_ID_BYTES], int n_ranks, int rank);
void mrk_comm_destroy(mrk_ctx* ctx); /* also done by mrk_ctx_destroy */
int mrk_comm_allreduce_i64(mrk_ctx* ctx, int64_t* values, uint64_t n);
int mrk_shard_exchange(mrk_ctx* ctx, mrk_batch* batch, const uint64_t* rows, uint32_t n_queries, uint32_t k, uint64_t* out_rows,
                       uint32_t slot);
/* The exchange is PARTITIONED BY QUERY (round 3; ctx key "exchange_part", default 1; needs ncclSend / ncclRecv in the loaded
   RCCL, <= 8 ranks -- else the all-gather form, where every rank receives and merges everything): rank r owns the queries
   mrk_shard_slice(n_queries, n_ranks, r) = [first, first + count), per = ceil(n_queries / n_ranks) each.  Every rank sends each
   owner its rows of the owner's queries (one grouped all-to-all of row slices), merges its own slice and writes
   out_rows[first .. first + count) -- the other rows of out_rows are not touched.  Bytes into a GPU and merge work per GPU are
   1 / n_ranks of the all-gather form's.  "Some shard flagged a row" (MRK_ROW_RERUN / MRK_ROW_DECLINED) is no longer visible in
   every rank's rows: two dwords -- any row flagged RERUN / DECLINED, on any rank -- travel through one small all-reduce behind
   the merge; mrk_shard_flags reads them after mrk_merge_wait (every rank sees the same values and takes the same repair path).
   mrk_shard_partitioned: 1 if mrk_shard_exchange on this context partitions. */
int mrk_shard_slice(uint32_t n_queries, int n_ranks, int rank, uint32_t* first, uint32_t* count);
int mrk_shard_flags(mrk_ctx* ctx, uint32_t slot, uint32_t* rerun_any, uint32_t* declined_any);
int mrk_shard_partitioned(mrk_ctx* ctx);
/* what a rank of the partitioned exchange does with its receive buffer, callable on its own (tests; a host with its own
   transport): rows_recv[n_lists][list_stride][MRK_ROW_WORDS] (device), the rank's `count` queries -> out_rows[first + q].
   n_lists <= 8.  Synchronous. */
int mrk_topk_merge_rows_part(mrk_ctx* ctx, const uint64_t* rows_recv, uint32_t n_lists, uint32_t list_stride, uint32_t first, uint32_t count,
                             uint32_t k, uint64_t* out_rows);

/* merge n_lists sorted partial top-K lists per query (device pointers):
   in_keys[(l*n_queries + q)*MRK_MAX_K + i], in_counts[l*n_queries + q] -> out_keys[q*MRK_MAX_K + i],
   out_counts[q].  Order: weight desc, global docid asc. Synchronous on the ctx stream. */
int mrk_topk_merge(mrk_ctx* ctx, const uint64_t* in_keys, const uint32_t* in_counts, uint32_t n_lists,
                   uint32_t n_queries, uint32_t k, uint64_t* out_keys, uint32_t* out_counts);

/* IDF exactly as sphCreateRanker computes it (host libm logf) */
float mrk_idf(int64_t term_docs, int64_t total_docs, int plain_idf, int normalized, int n_qwords, float boost);

/* ------------------------------------------------------------------------------------
 * Host-side index construction (format writer + synthetic corpus); no GPU involved.
 * Writer follows CSphHitBuilder (sphinx.cpp:8378-8719) byte for byte.
 * ---------------------------------------------------------------------------------- */
typedef struct mrk_host_index mrk_host_index;

/* hits sorted by (wordid, rowid, hitpos); wordid = term id + 1 */
int mrk_index_from_hits(const uint64_t* wordid, const uint32_t* rowid, const uint32_t* hitpos, uint64_t n,
                        uint32_t n_terms, uint32_t skiplist_block_size, uint32_t hit_format, mrk_host_index** out);

typedef struct {
  uint64_t seed;
  uint64_t n_docs;          /* docs in this segment (rowids 0..n_docs-1) */
  uint64_t rowid_base;      /* global rowid of the segment's rowid 0: postings are a function of (seed, term, global rowid),
                               so shards [base, base + n_docs) are slices of the one corpus */
  const double* term_prob;  /* document probability of each generated term */
  uint32_t n_terms;
  uint32_t n_fields;        /* hits fall into field 0 with prob title_frac, else uniformly in 1..n_fields-1 */
  double title_frac;
  uint32_t max_pos;         /* positions uniform in [1, max_pos] */
  uint32_t skiplist_block_size;
  uint32_t hit_format;
  uint32_t end_markers;     /* field-end bits: 0 none; 1 on each word's own last hit per field (two words at one position may differ in it);
                               2 on the hit at the field's last POSITION in that doc, whatever the word -- what the reference's indexer writes */
  uint32_t n_threads;       /* 0 = hardware concurrency */
} mrk_synth_params;

int mrk_synth_generate(const mrk_synth_params* p, mrk_host_index** out);
void mrk_host_index_free(mrk_host_index* h);
/* buffers have 64 zero bytes of slack after *len */
const uint8_t* mrk_host_index_spd(const mrk_host_index* h, uint64_t* len);
const uint8_t* mrk_host_index_spp(const mrk_host_index* h, uint64_t* len);
const uint8_t* mrk_host_index_spe(const mrk_host_index* h, uint64_t* len);
const mrk_dict_entry* mrk_host_index_dict(const mrk_host_index* h, uint32_t* n_terms);

/* Accounting aid for the roofline figures; not on the query path.  For each keyword pair
   (pairs[2i], pairs[2i+1]): common docs, each keyword's doc count, and the distinct 128-byte lines of each keyword's packed
   tf / field words (64 words per 128-doc block, in doclist order) that the common docs touch -- what the two-bitmap AND
   kernel's gathers have to fetch, instead of "every tf / field word of both doclists" -- and the distinct 128-doc
   blocks touched (= 128-byte lines of the one-byte-per-doc "attr_nibbles" plane); lines128s_*: the same count for the
   slot-ordered two-byte plane ("attr_seq": 64 consecutive docs per 128-byte line), which is what the kernel gathers from by
   default. */
typedef struct {
  uint64_t matches, docs_a, docs_b, lines128_a, lines128_b, blocks_a, blocks_b, lines128s_a, lines128s_b;
} mrk_pair_stats;
int mrk_host_index_pair_stats(const mrk_host_index* h, uint32_t hit_format, const uint32_t* pairs, uint32_t n_pairs,
                              uint32_t n_threads, mrk_pair_stats* out);

/* ------------------------------------------------------------------------------------
 * Real index ingestion: the files a Manticore 3.x indexer / RT disk chunk wrote,
 * format versions 54..62, read from disk into the same host-side object.  Replaces CSphIndex_VLN::LoadHeader
 * (sphinx.cpp:13252-13388), CWordlist::Preread + the dictionary block readers (indexformat.cpp:279-344, 425-470,
 * 641-691) and DeadRowMap_Disk_c (killlist.cpp:171-184) for what the match -> rank -> top-K path needs.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  uint32_t version;             /* .sph format version (54..62) */
  uint32_t n_fields, n_attrs;   /* schema */
  uint32_t skiplist_block_size; /* 128 before v56, else the stored setting */
  uint32_t hit_format;          /* MRK_HITFMT_* (ESphHitFormat) */
  uint32_t hitless;             /* ESphHitless; only 0 (none) can be searched here */
  uint32_t word_dict;           /* 1 = dict=keywords, 0 = dict=crc */
  uint32_t min_prefix_len, min_infix_len;
  uint32_t index_sp, index_field_lens;
  uint32_t n_checkpoints;
  uint64_t total_docs, total_bytes; /* m_tStats */
  uint64_t n_dead;                  /* bits set in the .spm dead-row map */
} mrk_index_info;

/* path_prefix + ".sph" / ".spi" / ".spd" / ".spp" / ".spe" (/ ".spm", optional).  Term ids of the result are the
   dictionary's entries in file order (sorted by keyword for dict=keywords, by word id for dict=crc). */
int mrk_index_open(const char* path_prefix, mrk_host_index** out);
int mrk_host_index_info(const mrk_host_index* h, mrk_index_info* out); /* MRK_E_INVAL unless opened from files */
const char* mrk_host_index_field_name(const mrk_host_index* h, uint32_t field);
/* dict=keywords: term id of a keyword (sphDictCmpStrictly order, sphinxint.h), -1 = not in the dictionary */
int32_t mrk_host_index_find_word(const mrk_host_index* h, const char* word, int32_t len);
const char* mrk_host_index_word(const mrk_host_index* h, uint32_t term_id, uint32_t* len);
/* dict=crc: term id of a word id, -1 = absent */
int32_t mrk_host_index_find_wordid(const mrk_host_index* h, uint64_t wordid);
/* Schema attributes as the header lists them (CSphColumnInfo: name, ESphAttr type, locator) and the row-wise storage of
   the .spa file: the first n_rows (= m_iDocinfo) rows of stride dwords -- what mrk_segment_set_attrs takes; the min-max
   index that follows them in the file is not used.  NULL / 0 when the index has no .spa (no attributes but the id). */
typedef struct {
  const char* name;
  uint32_t type;       /* ESphAttr (sphinx.h): 1 = uint, 2 = timestamp, 4 = bool, 5 = float, 6 = bigint, ... */
  int32_t bit_offset;  /* CSphAttrLocator::m_iBitOffset; < 0 for blob-stored attributes */
  int32_t bit_count;
} mrk_attr_info;
int mrk_host_index_attr(const mrk_host_index* h, uint32_t i, mrk_attr_info* out); /* i < mrk_index_info.n_attrs */
const uint32_t* mrk_host_index_attr_rows(const mrk_host_index* h, uint32_t* stride_dwords, uint64_t* n_rows);
/* .spm bitmap (bit rowid & 31 of word rowid >> 5), as mrk_segment_set_dead_rows takes it; NULL = no map */
const uint32_t* mrk_host_index_dead_rows(const mrk_host_index* h, uint64_t* n_rows);
/* the blob pool (.spb file as it is / an RT segment's m_dBlobs), NULL = none; n_blob_attrs = the schema's blob-stored attributes
   (strings, MVAs, JSON: those with bit_count 0, in schema order = their blob attribute ids) */
const uint8_t* mrk_host_index_blobs(const mrk_host_index* h, uint64_t* len, uint32_t* n_blob_attrs);

/* ---- RT index RAM chunk (host only) ---------------------------------------------------------------------------------------
   path_prefix + ".meta" / ".ram" as RtIndex_c::SaveMeta / SaveRamChunk write them (sphinxrt.cpp:3560-3640, 4034-4103; meta
   v.14-18).  Each RAM segment (RtSegment_t: its own dictionary, doclists and hitlists in the RT codecs, sphinxrt.cpp:390-640,
   rows, dead-row map) comes back as a mrk_host_index in the disk format -- decoded and re-emitted, so that an RT segment is
   one more mrk_segment for mrk_topk_merge; rowids are segment-local as in the reference.  Stepped over: stored fields (the docstore); not read: the
   blob pool; declined: hitless words.  mrk_rt_ram_take hands segment i to the caller (free it with mrk_host_index_free). */
typedef struct mrk_rt_ram mrk_rt_ram;
int mrk_rt_ram_open(const char* path_prefix, mrk_rt_ram** out);
uint32_t mrk_rt_ram_segments(const mrk_rt_ram* rt);
int mrk_rt_ram_take(mrk_rt_ram* rt, uint32_t i, mrk_host_index** out);
void mrk_rt_ram_free(mrk_rt_ram* rt);
/* A LIVE RAM segment handed over in memory (round 3): the byte vectors of an RtSegment_t (sphinxrt.h:140-149) -- m_dWords (the
   dictionary: keywords front-coded / word-id deltas, restarted every words_checkpoint entries), m_dDocs, m_dHits in the RT codecs --
   decoded and re-emitted in the disk format exactly as mrk_rt_ram_open does for the segments of a .ram file (same code).  The
   result is an ordinary mrk_host_index (mrk_host_index_spd / _dict / _find_word ...): mrk_segment_create it once when the segment
   appears (RAM segments are immutable once committed), hand its dead-row map and attribute rows over with
   mrk_segment_set_dead_rows / _set_attrs, and let the ranker binding pick it when RtIndex_c rebinds the ranker to that segment
   (ISphRanker::Reset, sphinxrt.cpp:6313-6314).  rows = RtSegment_t::m_uRows; words_checkpoint = RtIndex_c::m_iWordsCheckpoint. */
typedef struct {
  const uint8_t* words;
  uint64_t words_len;
  const uint8_t* docs;
  uint64_t docs_len;
  const uint8_t* hits;
  uint64_t hits_len;
  uint32_t rows;
  uint32_t word_dict;        /* 1 = dict=keywords, 0 = dict=crc */
  uint32_t words_checkpoint;
  uint32_t skiplist_block_size; /* of the re-encoded doclists; 0 = 128 */
  uint32_t hit_format;       /* MRK_HITFMT_* of the re-encoded doclists */
  uint32_t n_fields;
} mrk_rt_segment_desc;
int mrk_rt_segment_open(const mrk_rt_segment_desc* desc, mrk_host_index** out);

/* ---- query text -> tree (host only) ------------------------------------------------------------------------------------
   The caller side of the path: the extended query syntax (sphParseExtendedQuery: sphinxquery.y:57-125 grammar; the lexer
   XQParser_t::GetToken, sphinxquery.cpp:1201-1553; AddKeyword / AddOp, :1600-1678; FixupNots, :499-562) restated for the
   operators this library evaluates:  a b   a | b   a MAYBE b   -a  !a   ( )   "a b"   "a b"~N   "a b c"/N   "a b c"/0.5
   a << b   a NEAR/N b   a NOTNEAR/N b   a SENTENCE b   a PARAGRAPH b   @field  @(f1,f2)  @!field  @!(f1,f2)  @*  @field[N]  @@relaxed   ^a  a$  =a  a^1.5
   and '*' inside a phrase.  The result is the flat mrk_node[] + children[] form mrk_query takes (post-order, root last):
   query positions (atom_pos) in textual order, field limits as masks, NOT folded into ANDNOT, one-word phrases folded to
   their word, fractional quorum thresholds resolved against the word count.  'a SENTENCE b' / 'a PARAGRAPH b' come back as
   MRK_OP_SENTENCE / _PARAGRAPH nodes whose keyword text (mrk_parsed_keyword) is the boundary word "\3sentence" /
   "\3paragraph" for the dictionary lookup.  ZONE / ZONESPAN are not recognised (their capitals read as keywords).
   Tokenizing is ASCII [A-Za-z0-9_] + bytes >= 0x80, lower-casing ASCII only -- a host with a charset_table, morphology,
   stopwords or wordforms runs its own tokenizer / dictionary over the keywords' text; what this entry point fixes is the
   grammar.  Words shorter than min_word_len (code points) are dropped and keep their position (overshort_step = 1).
   field_names[i] is the name of full-text field i (bit i of a field mask).  Keywords come back as text
   (mrk_parsed_keyword); mrk_parsed_resolve fills term_id from a dict=keywords dictionary.  A syntax error returns
   MRK_E_INVAL with the reason in mrk_last_error() (the reference's "query error: ..."). */
typedef struct mrk_parsed_query mrk_parsed_query;
int mrk_query_parse(const char* text, const char* const* field_names, uint32_t n_fields, uint32_t min_word_len,
                    mrk_parsed_query** out);
/* sphTransformExtendedQuery (sphinx.cpp:15345-15359), the part every query goes through between the parser and the ranker:
   a quorum with threshold 1 -> the OR of its words (TransformQuorum); AND groups among a NEAR node's operands are replaced by their
   children, '(a b c) NEAR/N d' -> 'a NEAR/N b NEAR/N c NEAR/N d' (TransformNear).  In place.  Not restated: TransformBigrams (needs
   a bigram_index index) and the boolean simplifier (sphOptimizeBoolean: OPTION boolean_simplify, off by default). */
int mrk_parsed_transform(mrk_parsed_query* q);
void mrk_parsed_free(mrk_parsed_query* q);
int32_t mrk_parsed_n_nodes(const mrk_parsed_query* q);
int32_t mrk_parsed_root(const mrk_parsed_query* q); /* -1: the query holds no keyword (matches nothing) */
const mrk_node* mrk_parsed_nodes(const mrk_parsed_query* q);
const int32_t* mrk_parsed_children(const mrk_parsed_query* q, int32_t* n);
const char* mrk_parsed_keyword(const mrk_parsed_query* q, int32_t node); /* "" for operator nodes */
int mrk_parsed_resolve(mrk_parsed_query* q, const mrk_host_index* h);

#ifdef __cplusplus
}
#endif
#endif /* MRK_H */
