/*
 * oracle/cpu_ref.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of Manticore Search 3.6.1's full-text match -> rank -> top-K
 * hot path (reference: /root/reference/src, citations as file:line). It is the
 * checker the HIP path is compared against and the "port" CPU baseline timed by
 * bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it.  Nothing under manticoresearch_amd/ links or calls it.
 *
 * Parity pin: checked against the reference's own golden vectors
 * (tests/test_oracle_golden.py: doc/internals-index-format.txt VLB bytes,
 * gtests_rtstuff.cpp WeightBoundary, test_037 / test_019 / test_322 model.bin
 * weights) and, in the build container only, against a partial link of the
 * reference's own searchnode.cpp (oracle/ref_build, outputs in oracle/_ref/).
 */
#ifndef ORACLE_CPU_REF_H
#define ORACLE_CPU_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INVALID_ROWID 0xFFFFFFFFu /* src/sphinx.h:59-62 */
#define ORC_EMPTY_HIT 0u              /* src/sphinx.h:757-760 */
#define ORC_MAX_FIELDS 256            /* src/sphinx.h:108 */

/* Hitpos_t = field:8 | end:1 | pos:23  (src/sphinx.h:767-827) */
#define ORC_HIT_FIELD(h) ((uint32_t)(h) >> 24)
#define ORC_HIT_POS(h) ((uint32_t)(h) & 0x7FFFFFu)
#define ORC_HIT_ISEND(h) (((uint32_t)(h) >> 23) & 1u)
#define ORC_HIT_POSWITHFIELD(h) ((uint32_t)(h) & ~(1u << 23))
#define ORC_HIT_MAKE(field, pos, end) (((uint32_t)(field) << 24) | ((uint32_t)((end) ? 1 : 0) << 23) | ((uint32_t)(pos) & 0x7FFFFFu))

/* ranker ids follow ESphRankMode order (src/sphinx.h, SPH_RANK_*) */
enum {
  ORC_RANK_PROXIMITY_BM25 = 0,
  ORC_RANK_BM25 = 1,
  ORC_RANK_NONE = 2,
  ORC_RANK_WORDCOUNT = 3,
  ORC_RANK_PROXIMITY = 4,
  ORC_RANK_MATCHANY = 5,
  ORC_RANK_FIELDMASK = 6,
  ORC_RANK_SPH04 = 7
};

/* query operators (subset of XQOperator_e, src/sphinxquery.h) */
enum {
  ORC_OP_TERM = 0, /* leaf keyword */
  ORC_OP_AND = 1,
  ORC_OP_OR = 2,
  ORC_OP_MAYBE = 3,
  ORC_OP_ANDNOT = 4,
  ORC_OP_PHRASE = 5,
  ORC_OP_PROXIMITY = 6,
  ORC_OP_QUORUM = 7,
  ORC_OP_BEFORE = 8, /* 'a << b << c' (ExtOrder_c) */
  ORC_OP_NEAR = 9,    /* 'a NEAR/N b NEAR/N c' (ExtNWay_T<FSMmultinear_c>): opt = N */
  ORC_OP_NOTNEAR = 10, /* 'a NOTNEAR/N b' (ExtNotNear_c): opt = N */
  ORC_OP_SENTENCE = 11, /* 'a SENTENCE b' (ExtUnit_c over the index_sp boundary keyword): the node's term_id = that keyword's slot (< 0: none indexed) */
  ORC_OP_PARAGRAPH = 12
};

/* ---- VLB codec (src/sphinxstd.h:5545-5567, src/fileio.cpp:31-45) ---- */
int orc_zip_u64(uint8_t* out, uint64_t v); /* returns bytes written (1..10) */
uint32_t orc_unzip_u32(const uint8_t** pp);
uint64_t orc_unzip_u64(const uint8_t** pp);

/* ---- dictionary entry: what CSphDictEntry carries (src/sphinx.h:542-552) ---- */
typedef struct {
  uint64_t wordid;
  uint64_t doclist_off; /* m_iDoclistOffset */
  uint64_t doclist_len; /* m_iDoclistLength (bytes, terminator included) */
  uint64_t skiplist_off; /* m_iSkiplistOffset (valid iff docs > skip_block) */
  uint32_t docs;        /* m_iDocs */
  uint32_t hits;        /* m_iHits */
} orc_dict_entry;

/* ---- index writer: CSphHitBuilder restatement (src/sphinx.cpp:8378-8719) ---- */
typedef struct orc_writer orc_writer;
orc_writer* orc_writer_new(int skiplist_block_size, int inline_hits);
void orc_writer_free(orc_writer* w);
/* hits must arrive sorted by (wordid, rowid, hitpos) like cidxHit expects */
void orc_writer_hit(orc_writer* w, uint64_t wordid, uint32_t rowid, uint32_t hitpos);
void orc_writer_hits(orc_writer* w, const uint64_t* wordid, const uint32_t* rowid, const uint32_t* hitpos, size_t n);
void orc_writer_finish(orc_writer* w);
const uint8_t* orc_writer_spd(const orc_writer* w, size_t* len);
const uint8_t* orc_writer_spp(const orc_writer* w, size_t* len);
const uint8_t* orc_writer_spe(const orc_writer* w, size_t* len);
const orc_dict_entry* orc_writer_dict(const orc_writer* w, size_t* n);

/* ---- an index segment as the searcher sees it ---- */
typedef struct {
  const uint8_t* spd;
  size_t spd_len;
  const uint8_t* spp;
  size_t spp_len;
  const uint8_t* spe;
  size_t spe_len;
  const orc_dict_entry* dict; /* indexed by term id (flat table replaces .spi) */
  uint32_t n_terms;
  int64_t total_docs; /* m_iTotalDocuments */
  int skiplist_block_size;
  int inline_hits; /* hit_format=inline (1) or plain (0) */
  int n_fields;
  const uint32_t* dead_rows; /* DeadRowMap_c bitmap (killlist.h:22-46) or NULL */
  const uint32_t* attrs;     /* row-wise attribute storage (.spa): CSphRowitem rows[total_docs][attr_stride], or NULL */
  int attr_stride;
  const uint8_t* blobs;      /* blob pool (.spb / m_dBlobs) for MVA filters, or NULL */
} orc_index;

/* CSphFilterSettings over an integer attribute (sphinx.h:2461-2496), resolved to the attribute's locator */
enum { ORC_FILTER_VALUES = 0, ORC_FILTER_RANGE = 1, ORC_FILTER_FLOATRANGE = 2 };
typedef struct {
  int kind;
  int bit_offset, bit_count; /* CSphAttrLocator */
  int exclude;
  int has_equal_min, has_equal_max, open_left, open_right;
  int64_t min_value, max_value;
  const int64_t* values; /* ascending */
  int n_values;
  float fmin, fmax; /* FLOATRANGE */
  int mva_bits, mva_all, blob_attr_id, n_blob_attrs; /* an MVA in the blob pool: 32 / 64 bits wide, ANY / ALL form */
} orc_filter;

/* ---- query tree ---- */
typedef struct {
  int op;              /* ORC_OP_* */
  int n_children;      /* for operators */
  int first_child;     /* index into children[] of the query */
  /* leaf */
  int32_t term_id;     /* index into dict; <0 = word not in dictionary */
  int atom_pos;        /* XQKeyword_t::m_iAtomPos (1-based, src/sphinxquery.cpp:1266) */
  uint32_t field_mask; /* XQLimitSpec_t::m_dFieldMask low dword (all ones = any field) */
  float boost;         /* m_fBoost */
  int opt;             /* proximity distance / quorum threshold */
  int not_weighted;
  int term_pos;        /* ORC_TERMPOS_*: XQKeyword_t::m_bFieldStart / m_bFieldEnd, XQLimitSpec_t::m_iFieldMaxPos */
  int field_max_pos;   /* m_iFieldMaxPos for ORC_TERMPOS_LIMIT */
  uint32_t field_mask_hi[7]; /* m_dFieldMask dwords 1..7: fields 32..255 (FieldMask_t, sphinx.h:830-900); read only for indexes with
                                more than 32 fields -- "any field" there = all ones in every dword */
} orc_node;

/* TermPosFilter_e (searchnode.cpp:875-878, 1145-1146) */
enum { ORC_TERMPOS_NONE = 0, ORC_TERMPOS_START = 1, ORC_TERMPOS_END = 2, ORC_TERMPOS_STARTEND = 3, ORC_TERMPOS_LIMIT = 4 };

typedef struct {
  const orc_node* nodes;
  int n_nodes;
  const int* children; /* child node indices */
  int root;
  int ranker;
  int max_matches;     /* K */
  const int32_t* field_weights; /* may be NULL => all 1 */
  int n_weights;       /* number of fields with weights */
  int index_weight;    /* default 1 (src/sphinx.cpp:12220) */
  int plain_idf;       /* CSphQuery::m_bPlainIDF */
  int normalized_tfidf;/* CSphQuery::m_bNormalizedTFIDF (default 1) */
  int64_t total_docs_override; /* local_df: m_iTotalDocs (<=0: use index) */
  const int64_t* local_docs;   /* local_df per NODE index (docs override, <0 none) or NULL */
  int cutoff;          /* 0 = none */
  const orc_filter* filters; /* all must pass (Filter_And) */
  int n_filters;
  const orc_filter* weight_filters; /* m_pWeightFilter: VALUES / RANGE over the match weight */
  int n_weight_filters;
} orc_query;

typedef struct {
  int n;               /* matches returned (<= K) */
  int64_t total_found; /* ISphMatchSorter::GetTotalCount() */
  uint32_t* rowid;     /* caller-provided, K entries, best first */
  int32_t* weight;
  /* CSphQueryStats (src/sphinx.h:2697-2705) */
  int64_t fetched_docs, fetched_hits, skips;
} orc_result;

/* returns 0 on success, <0 on error (orc_last_error) */
int orc_search(const orc_index* idx, const orc_query* q, orc_result* res);
const char* orc_last_error(void);

/* bench.py cpu_baseline: run every query `repeat` times over n_threads threads (one independent
   query per thread); returns wall seconds, < 0 on error */
double orc_search_many(const orc_index* idx, const orc_query* const* queries, int n_queries, int repeat, int n_threads);

/* IDF per sphCreateRanker (src/sphinxsearch.cpp:4317-4361) */
float orc_idf(int64_t term_docs, int64_t total_docs, int plain_idf, int normalized, int n_qwords, float boost);

/* ---- decoded views used by tests ---- */
/* decode a term's whole doclist; arrays caller-provided with dict.docs entries.
   hitpos64: m_iHitlistPos as the reader reports it (bit63 = inlined hit). */
int orc_decode_doclist(const orc_index* idx, uint32_t term_id, uint32_t* rowid, uint32_t* fields,
                       uint32_t* hits, uint64_t* hitpos64);
/* decode the hits of one doclist entry; returns count (out may be NULL) */
int orc_decode_hits(const orc_index* idx, uint64_t hitpos64, uint32_t* out, int max_out);
/* decode skiplist like DiskIndexQwordSetup_c::Setup (src/sphinx.cpp:13056-13073);
   returns entry count (0 if no skiplist) */
int orc_decode_skiplist(const orc_index* idx, uint32_t term_id, uint32_t* base_plus1, uint64_t* off,
                        uint64_t* hitbase, int max_entries);

#ifdef __cplusplus
}
#endif
#endif
