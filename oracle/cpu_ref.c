/*
 * oracle/cpu_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see cpu_ref.h).
 *
 * Plain-C restatement of the reference's match -> rank -> top-K path with the
 * reference's algorithmic shape: byte-at-a-time VLB decode behind an indirect
 * call, per-term skiplist decode at query setup, leap-frog N-way AND by
 * ascending doc count, fp32 tf/(tf+1.2)*idf, int weights, binary-heap top-K.
 * Every function cites the reference lines it follows (paths under
 * /root/reference/src unless stated).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off (no -march, no fast-math: the
 * reference is built the same way, CMakeLists.txt:297-306).
 */
#include "cpu_ref.h"

#include <limits.h>
#include <pthread.h>
#include <time.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int fail(const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return -1;
}

/* ------------------------------------------------------------------------ */
/* VLB codec                                                                */
/* ------------------------------------------------------------------------ */

/* sphZipValue / sphCalcZippedLen, sphinxstd.h:5545-5567: 7-bit groups, most
   significant first, continuation bit on all but the last byte */
int orc_zip_u64(uint8_t* out, uint64_t v) {
  int n = 1;
  uint64_t t = v >> 7;
  while (t) {
    t >>= 7;
    ++n;
  }
  for (int i = n - 1; i >= 0; --i) *out++ = (uint8_t)((0x7f & (v >> (7 * i))) | (i ? 0x80 : 0));
  return n;
}

/* SPH_VARINT_DECODE, fileio.cpp:31-45 */
uint32_t orc_unzip_u32(const uint8_t** pp) {
  const uint8_t* p = *pp;
  uint32_t b = *p++;
  uint32_t res = 0;
  while (b & 0x80) {
    res = (res << 7) + (b & 0x7f);
    b = *p++;
  }
  res = (res << 7) + b;
  *pp = p;
  return res;
}
uint64_t orc_unzip_u64(const uint8_t** pp) {
  const uint8_t* p = *pp;
  uint32_t b = *p++;
  uint64_t res = 0;
  while (b & 0x80) {
    res = (res << 7) + (b & 0x7f);
    b = *p++;
  }
  res = (res << 7) + b;
  *pp = p;
  return res;
}

/* ------------------------------------------------------------------------ */
/* growable buffers                                                         */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint8_t* p;
  size_t n, cap;
} bbuf;

static void bb_reserve(bbuf* b, size_t extra) {
  if (b->n + extra <= b->cap) return;
  size_t nc = b->cap ? b->cap * 2 : 4096;
  while (nc < b->n + extra) nc *= 2;
  b->p = (uint8_t*)realloc(b->p, nc);
  b->cap = nc;
}
static void bb_put(bbuf* b, uint8_t v) {
  bb_reserve(b, 1);
  b->p[b->n++] = v;
}
static void bb_zip(bbuf* b, uint64_t v) {
  bb_reserve(b, 10);
  b->n += (size_t)orc_zip_u64(b->p + b->n, v);
}

/* ------------------------------------------------------------------------ */
/* writer: CSphHitBuilder (sphinx.cpp:8378-8719)                            */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint32_t base_plus1;
  uint64_t off;
  uint64_t hitbase;
} skip_entry;

struct orc_writer {
  int block, inline_hits;
  bbuf spd, spp, spe;
  orc_dict_entry* dict;
  size_t n_dict, cap_dict;
  /* m_tLastHit, m_iPrevHitPos, m_bGotFieldEnd (sphinx.cpp:8414-8422) */
  uint64_t last_wordid;
  uint32_t last_rowid;
  uint32_t last_wordpos;
  uint32_t prev_hitpos;
  int got_field_end;
  /* m_tWord */
  uint32_t w_docs, w_hits;
  uint64_t w_doclist_off, w_skiplist_off;
  /* hitlist trackers */
  uint64_t last_hitlist_pos, last_hitlist_delta;
  uint32_t last_doc_hits;
  uint32_t last_doc_fields32;
  skip_entry* skips;
  size_t n_skips, cap_skips;
  int finished;
};

orc_writer* orc_writer_new(int skiplist_block_size, int inline_hits) {
  orc_writer* w = (orc_writer*)calloc(1, sizeof *w);
  w->block = skiplist_block_size;
  w->inline_hits = inline_hits;
  /* dummy first byte so that offset 0 is never valid (sphinx.cpp:8404-8409) */
  bb_put(&w->spd, 1);
  bb_put(&w->spp, 1);
  bb_put(&w->spe, 1);
  /* HitReset (sphinx.cpp:8414-8422) */
  w->last_rowid = ORC_INVALID_ROWID;
  w->last_wordid = 0;
  w->last_wordpos = ORC_EMPTY_HIT;
  w->prev_hitpos = 0;
  w->got_field_end = 0;
  return w;
}

void orc_writer_free(orc_writer* w) {
  if (!w) return;
  free(w->spd.p);
  free(w->spp.p);
  free(w->spe.p);
  free(w->dict);
  free(w->skips);
  free(w);
}

/* DoclistBeginEntry, sphinx.cpp:8441-8457 */
static void wr_doclist_begin(orc_writer* w, uint32_t rowid) {
  if ((w->w_docs & (uint32_t)(w->block - 1)) == 0) {
    if (w->n_skips == w->cap_skips) {
      w->cap_skips = w->cap_skips ? w->cap_skips * 2 : 64;
      w->skips = (skip_entry*)realloc(w->skips, w->cap_skips * sizeof(skip_entry));
    }
    skip_entry* s = &w->skips[w->n_skips++];
    s->base_plus1 = w->last_rowid + 1u;
    s->off = w->spd.n;
    s->hitbase = w->last_hitlist_pos;
  }
  bb_zip(&w->spd, (uint32_t)(rowid - w->last_rowid));
}

/* DoclistEndEntry, sphinx.cpp:8460-8497 (hitless modes not restated) */
static void wr_doclist_end(orc_writer* w, uint32_t last_pos) {
  if (w->inline_hits) {
    bb_zip(&w->spd, w->last_doc_hits);
    if (w->last_doc_hits == 1) {
      w->spp.n = (size_t)w->last_hitlist_pos; /* m_wrHitlist.SeekTo: the lone hit leaves .spp */
      bb_zip(&w->spd, last_pos & 0x7FFFFFu);
      bb_zip(&w->spd, last_pos >> 23);
      w->last_hitlist_pos -= w->last_hitlist_delta;
    } else {
      bb_zip(&w->spd, w->last_doc_fields32);
      bb_zip(&w->spd, w->last_hitlist_delta);
    }
  } else {
    bb_zip(&w->spd, w->last_hitlist_delta);
    bb_zip(&w->spd, w->last_doc_fields32);
    bb_zip(&w->spd, w->last_doc_hits);
  }
  w->last_doc_fields32 = 0;
  w->last_doc_hits = 0;
  w->w_docs++;
}

/* DoclistEndList, sphinx.cpp:8500-8542 */
static void wr_doclist_endlist(orc_writer* w) {
  bb_zip(&w->spd, 0);
  w->w_skiplist_off = 0;
  if (w->w_docs > (uint32_t)w->block) {
    w->w_skiplist_off = w->spe.n;
    skip_entry last = w->skips[0];
    for (size_t i = 1; i < w->n_skips; i++) {
      const skip_entry* t = &w->skips[i];
      bb_zip(&w->spe, (uint32_t)(t->base_plus1 - last.base_plus1 - (uint32_t)w->block));
      bb_zip(&w->spe, t->off - last.off - 4u * (uint64_t)w->block);
      bb_zip(&w->spe, t->hitbase - last.hitbase);
      last = *t;
    }
  }
  w->n_skips = 0;
}

/* cidxHit, sphinx.cpp:8554-8719 (aggregate/hitless branch not restated) */
static void wr_cidx_hit(orc_writer* w, uint64_t wordid, uint32_t rowid, uint32_t wordpos) {
  const int next_word = (w->last_wordid != wordid);
  const int next_doc = next_word || (w->last_rowid != rowid);

  if (w->got_field_end && (next_word || next_doc)) {
    w->last_wordpos |= (1u << 23); /* HITMAN::SetEndMarker */
    bb_zip(&w->spp, (uint32_t)(w->last_wordpos - w->prev_hitpos));
    w->got_field_end = 0;
  }

  if (next_doc) {
    uint32_t last_pos = w->last_wordpos;
    if (w->last_wordpos != ORC_EMPTY_HIT) {
      bb_zip(&w->spp, 0);
      w->last_wordpos = ORC_EMPTY_HIT;
      w->prev_hitpos = ORC_EMPTY_HIT;
    }
    if (w->last_rowid != ORC_INVALID_ROWID) wr_doclist_end(w, last_pos);
  }

  if (next_word) {
    if (w->last_rowid != ORC_INVALID_ROWID) {
      wr_doclist_endlist(w);
      if (w->n_dict == w->cap_dict) {
        w->cap_dict = w->cap_dict ? w->cap_dict * 2 : 64;
        w->dict = (orc_dict_entry*)realloc(w->dict, w->cap_dict * sizeof(orc_dict_entry));
      }
      orc_dict_entry* e = &w->dict[w->n_dict++];
      e->wordid = w->last_wordid;
      e->doclist_off = w->w_doclist_off;
      e->doclist_len = w->spd.n - w->w_doclist_off;
      e->skiplist_off = w->w_skiplist_off;
      e->docs = w->w_docs;
      e->hits = w->w_hits;
      w->w_docs = 0;
      w->w_hits = 0;
      w->last_rowid = ORC_INVALID_ROWID;
      w->last_hitlist_pos = 0;
    }
    if (wordpos == ORC_EMPTY_HIT) return; /* flush-hit */
    w->w_doclist_off = w->spd.n;
    w->last_wordid = wordid;
  }

  if (next_doc) {
    wr_doclist_begin(w, rowid);
    w->last_hitlist_delta = w->spp.n - w->last_hitlist_pos;
    w->last_rowid = rowid;
    w->last_hitlist_pos = w->spp.n;
  }

  /* the hit */
  uint32_t pure = ORC_HIT_POSWITHFIELD(wordpos);
  if (pure == w->last_wordpos) return; /* duplicate position, keep the first */

  if (w->got_field_end) {
    if (ORC_HIT_FIELD(wordpos) != ORC_HIT_FIELD(w->last_wordpos)) w->last_wordpos |= (1u << 23);
    bb_zip(&w->spp, (uint32_t)(w->last_wordpos - w->prev_hitpos));
    w->got_field_end = 0;
  }

  if (pure == wordpos) {
    bb_zip(&w->spp, (uint32_t)(wordpos - w->last_wordpos));
    w->last_wordpos = wordpos;
  } else {
    w->got_field_end = 1;
    w->prev_hitpos = w->last_wordpos;
    w->last_wordpos = pure;
  }

  uint32_t f = ORC_HIT_FIELD(wordpos);
  if (f < 32) w->last_doc_fields32 |= (1u << f); /* GetMask32: low dword only */
  w->last_doc_hits++;
  w->w_hits++;
}

void orc_writer_hit(orc_writer* w, uint64_t wordid, uint32_t rowid, uint32_t hitpos) {
  wr_cidx_hit(w, wordid, rowid, hitpos);
}

void orc_writer_hits(orc_writer* w, const uint64_t* wordid, const uint32_t* rowid, const uint32_t* hitpos, size_t n) {
  for (size_t i = 0; i < n; i++) wr_cidx_hit(w, wordid[i], rowid[i], hitpos[i]);
}

void orc_writer_finish(orc_writer* w) {
  if (w->finished) return;
  wr_cidx_hit(w, 0, ORC_INVALID_ROWID, ORC_EMPTY_HIT);
  w->finished = 1;
}

const uint8_t* orc_writer_spd(const orc_writer* w, size_t* len) {
  *len = w->spd.n;
  return w->spd.p;
}
const uint8_t* orc_writer_spp(const orc_writer* w, size_t* len) {
  *len = w->spp.n;
  return w->spp.p;
}
const uint8_t* orc_writer_spe(const orc_writer* w, size_t* len) {
  *len = w->spe.n;
  return w->spe.p;
}
const orc_dict_entry* orc_writer_dict(const orc_writer* w, size_t* n) {
  *n = w->n_dict;
  return w->dict;
}

/* ------------------------------------------------------------------------ */
/* reader: ThinMMapReader_c (datareader.cpp:46-156) behind an indirect call  */
/* (FileBlockReader_i, datareader.h:29-31)                                   */
/* ------------------------------------------------------------------------ */
typedef struct reader reader;
typedef struct {
  uint32_t (*unzip_int)(reader*);
  uint64_t (*unzip_off)(reader*);
} reader_vt;
struct reader {
  const reader_vt* vt;
  const uint8_t* base;
  const uint8_t* ptr;
  int64_t size;
};

static inline uint8_t rd_getbyte(reader* r) {
  int64_t pos = r->ptr - r->base;
  if (pos >= 0 && pos < r->size) return *r->ptr++;
  return 0; /* datareader.cpp:93-104: warn and return 0 */
}
static __attribute__((noinline)) uint32_t rd_unzip_int(reader* r) {
  uint32_t b = rd_getbyte(r);
  uint32_t res = 0;
  while (b & 0x80) {
    res = (res << 7) + (b & 0x7f);
    b = rd_getbyte(r);
  }
  return (res << 7) + b;
}
static __attribute__((noinline)) uint64_t rd_unzip_off(reader* r) {
  uint32_t b = rd_getbyte(r);
  uint64_t res = 0;
  while (b & 0x80) {
    res = (res << 7) + (b & 0x7f);
    b = rd_getbyte(r);
  }
  return (res << 7) + b;
}
static const reader_vt g_reader_vt = {rd_unzip_int, rd_unzip_off};

static void rd_init(reader* r, const uint8_t* base, size_t size) {
  r->vt = &g_reader_vt;
  r->base = r->ptr = base;
  r->size = (int64_t)size;
}
static inline int64_t rd_pos(const reader* r) { return r->ptr - r->base; }
static inline void rd_seek(reader* r, int64_t pos) { r->ptr = r->base + pos; }

/* ------------------------------------------------------------------------ */
/* query word: DiskIndexQword_c (sphinx.cpp:357-550) + Setup (:12953-13080)  */
/* ------------------------------------------------------------------------ */
typedef struct {
  /* setup */
  int docs, hits;
  int has_hitlist;
  int inline_hits;
  skip_entry* skips;
  int n_skips;
  reader rd_doc, rd_hit;
  /* iterator state */
  uint32_t rowid;       /* m_tDoc.m_tRowID */
  uint32_t fields32;    /* m_dQwordFields.GetMask32() */
  uint32_t match_hits;  /* m_uMatchHits */
  uint64_t hitlist_pos; /* m_iHitlistPos */
  uint64_t hit_position;/* m_uHitPosition */
  int skip_block;       /* m_iSkipListBlock */
  /* hit state */
  int hit_state;
  uint32_t inlined_hit;
  uint32_t hit_pos;
  int64_t* p_skips; /* stats */
} qword;

/* FindSpan, sphinxint.h:416-466, over skiplist base_plus1 (operators sphinxsearch.h:42-44) */
static int find_span(const skip_entry* v, int n, uint32_t ref) {
  if (!n) return -1;
  if (v[n - 1].base_plus1 <= ref) return n - 1;
  if (n <= 8) {
    for (int i = 0; i < n - 1; i++)
      if (v[i].base_plus1 <= ref && ref < v[i + 1].base_plus1) return i;
    return -1;
  }
  const skip_entry* start = v;
  const skip_entry* end = v + n - 1;
  if (start[0].base_plus1 <= ref && ref < start[1].base_plus1) return 0;
  if (end[-1].base_plus1 <= ref && ref < end[0].base_plus1) return (int)(end - v - 1);
  while (end - start > 1) {
    if (ref < start->base_plus1 || end->base_plus1 < ref) break;
    const skip_entry* mid = start + (end - start) / 2;
    if (mid[0].base_plus1 <= ref && ref < mid[1].base_plus1) return (int)(mid - v);
    if (ref < mid[0].base_plus1)
      end = mid;
    else
      start = mid;
  }
  return -1;
}

static int qw_setup(qword* q, const orc_index* idx, int32_t term_id, int64_t* p_skips) {
  memset(q, 0, sizeof *q);
  q->rowid = ORC_INVALID_ROWID; /* sphinx.cpp:12947 */
  q->skip_block = -1;
  q->inline_hits = idx->inline_hits;
  q->has_hitlist = 1; /* SPH_HITLESS_NONE */
  q->p_skips = p_skips;
  rd_init(&q->rd_doc, idx->spd, idx->spd_len);
  rd_init(&q->rd_hit, idx->spp, idx->spp_len);
  if (term_id < 0 || (uint32_t)term_id >= idx->n_terms) return 0; /* word not in dict */
  const orc_dict_entry* e = &idx->dict[term_id];
  if (!e->docs) return 0;
  q->docs = (int)e->docs;
  q->hits = (int)e->hits;
  /* read in skiplist, sphinx.cpp:13056-13073 */
  if (idx->spe && e->docs > (uint32_t)idx->skiplist_block_size) {
    int n = q->docs / idx->skiplist_block_size;
    q->skips = (skip_entry*)malloc((size_t)(n > 0 ? n : 1) * sizeof(skip_entry));
    const uint8_t* p = idx->spe + e->skiplist_off;
    q->skips[0].base_plus1 = 0;
    q->skips[0].off = e->doclist_off;
    q->skips[0].hitbase = 0;
    q->n_skips = 1;
    for (int i = 1; i < n; i++) {
      skip_entry* t = &q->skips[i];
      const skip_entry* pr = &q->skips[i - 1];
      t->base_plus1 = pr->base_plus1 + (uint32_t)idx->skiplist_block_size + orc_unzip_u32(&p);
      t->off = pr->off + 4u * (uint64_t)idx->skiplist_block_size + orc_unzip_u64(&p);
      t->hitbase = pr->hitbase + orc_unzip_u64(&p);
      q->n_skips++;
    }
  }
  rd_seek(&q->rd_doc, (int64_t)e->doclist_off);
  return 1;
}

static void qw_free(qword* q) {
  free(q->skips);
  q->skips = NULL;
}

/* ReadNext, sphinx.cpp:511-549 */
static inline void qw_read_next(qword* q) {
  uint32_t delta = q->rd_doc.vt->unzip_int(&q->rd_doc);
  if (delta) {
    q->rowid += delta;
    if (q->inline_hits) {
      q->match_hits = q->rd_doc.vt->unzip_int(&q->rd_doc);
      const uint32_t first = q->rd_doc.vt->unzip_int(&q->rd_doc);
      if (q->match_hits == 1 && q->has_hitlist) {
        uint32_t field = q->rd_doc.vt->unzip_int(&q->rd_doc);
        q->hitlist_pos = (uint64_t)first | ((uint64_t)field << 23) | (1ULL << 63);
        uint32_t f = (field >> 1) & (ORC_MAX_FIELDS - 1);
        q->fields32 = f < 32 ? (1u << f) : 0;
      } else {
        q->fields32 = first;
        q->hit_position += q->rd_doc.vt->unzip_off(&q->rd_doc);
        q->hitlist_pos = q->hit_position;
      }
    } else {
      uint64_t dpos = q->rd_doc.vt->unzip_off(&q->rd_doc);
      q->hitlist_pos += dpos;
      q->fields32 = q->rd_doc.vt->unzip_int(&q->rd_doc);
      q->match_hits = q->rd_doc.vt->unzip_int(&q->rd_doc);
    }
  } else
    q->rowid = ORC_INVALID_ROWID;
}

/* HintRowID, sphinx.cpp:407-451 */
static int qw_hint(qword* q, uint32_t rowid) {
  if (q->skip_block == -1) {
    q->skip_block = find_span(q->skips, q->n_skips, rowid);
    if (q->skip_block < 0) return 0;
  } else {
    if (q->skip_block < q->n_skips - 1) {
      int next = q->skip_block + 1;
      if (rowid >= q->skips[next].base_plus1) {
        int b = find_span(q->skips + next, q->n_skips - next, rowid);
        q->skip_block = b;
        if (b < 0) return 0;
        q->skip_block += next;
      }
    } else
      return 0;
  }
  const skip_entry* t = &q->skips[q->skip_block];
  if ((int64_t)t->off <= rd_pos(&q->rd_doc)) return 0;
  rd_seek(&q->rd_doc, (int64_t)t->off);
  q->rowid = t->base_plus1 - 1u;
  q->hit_position = q->hitlist_pos = t->hitbase;
  if (q->p_skips) (*q->p_skips)++;
  return 1;
}

/* AdvanceTo, sphinx.cpp:391-404 */
static uint32_t qw_advance_to(qword* q, uint32_t rowid) {
  if (q->rowid != ORC_INVALID_ROWID && rowid <= q->rowid) return q->rowid;
  int rewound = qw_hint(q, rowid);
  if (rewound || q->rowid == ORC_INVALID_ROWID) qw_read_next(q);
  while (q->rowid < rowid) qw_read_next(q);
  return q->rowid;
}

/* SeekHitlist, sphinx.cpp:459-477 */
static void qw_seek_hitlist(qword* q, uint64_t off) {
  if (off >> 63) {
    q->hit_state = 1;
    q->inlined_hit = (uint32_t)off;
  } else {
    q->hit_state = 0;
    q->hit_pos = ORC_EMPTY_HIT;
    rd_seek(&q->rd_hit, (int64_t)off);
  }
}

/* GetNextHit / GetHitlistEntry, sphinx.cpp:374-388, 479-501 */
static uint32_t qw_next_hit(qword* q) {
  switch (q->hit_state) {
    case 0: {
      uint32_t d = q->rd_hit.vt->unzip_int(&q->rd_hit);
      if (d)
        q->hit_pos += d;
      else
        q->hit_pos = ORC_EMPTY_HIT;
      return q->hit_pos;
    }
    case 1:
      q->hit_state = 2;
      return q->inlined_hit;
    default:
      q->hit_state = 0;
      return ORC_EMPTY_HIT;
  }
}

/* ------------------------------------------------------------------------ */
/* decoded views for tests                                                  */
/* ------------------------------------------------------------------------ */
int orc_decode_doclist(const orc_index* idx, uint32_t term_id, uint32_t* rowid, uint32_t* fields,
                       uint32_t* hits, uint64_t* hitpos64) {
  qword q;
  if (!qw_setup(&q, idx, (int32_t)term_id, NULL)) return 0;
  int n = 0;
  for (;;) {
    qw_read_next(&q);
    if (q.rowid == ORC_INVALID_ROWID) break;
    if (rowid) rowid[n] = q.rowid;
    if (fields) fields[n] = q.fields32;
    if (hits) hits[n] = q.match_hits;
    if (hitpos64) hitpos64[n] = q.hitlist_pos;
    n++;
  }
  qw_free(&q);
  return n;
}

int orc_decode_hits(const orc_index* idx, uint64_t hitpos64, uint32_t* out, int max_out) {
  qword q;
  memset(&q, 0, sizeof q);
  rd_init(&q.rd_hit, idx->spp, idx->spp_len);
  qw_seek_hitlist(&q, hitpos64);
  int n = 0;
  for (;;) {
    uint32_t h = qw_next_hit(&q);
    if (h == ORC_EMPTY_HIT) break;
    if (out && n < max_out) out[n] = h;
    n++;
  }
  return n;
}

int orc_decode_skiplist(const orc_index* idx, uint32_t term_id, uint32_t* base_plus1, uint64_t* off,
                        uint64_t* hitbase, int max_entries) {
  qword q;
  if (!qw_setup(&q, idx, (int32_t)term_id, NULL)) return 0;
  int n = q.n_skips;
  for (int i = 0; i < n && i < max_entries; i++) {
    if (base_plus1) base_plus1[i] = q.skips[i].base_plus1;
    if (off) off[i] = q.skips[i].off;
    if (hitbase) hitbase[i] = q.skips[i].hitbase;
  }
  qw_free(&q);
  return n;
}

/* ------------------------------------------------------------------------ */
/* eval tree: ExtNode_* (searchnode.cpp)                                    */
/* ------------------------------------------------------------------------ */
#define SPH_BM25_K1 1.2f    /* searchnode.cpp:45 */
#define SPH_BM25_SCALE 1000 /* sphinxsearch.cpp:31 */

/* ExtHit_t, sphinxint.h:725-743 */
typedef struct {
  uint32_t rowid, hitpos;
  uint16_t qpos, nodepos, spanlen, matchlen;
  uint32_t weight, qposmask;
} hit_t;

typedef struct {
  hit_t* p;
  int n, cap;
} hitvec;

static void hv_push(hitvec* v, const hit_t* h) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->p = (hit_t*)realloc(v->p, (size_t)v->cap * sizeof(hit_t));
  }
  v->p[v->n++] = *h;
}

enum { EN_TERM, EN_MULTIAND, EN_AND, EN_OR, EN_MAYBE, EN_ANDNOT, EN_PHRASE, EN_QUORUM, EN_ORDER, EN_NOTNEAR, EN_UNIT };

typedef struct {
  qword qw;
  uint32_t queried32;
  uint32_t queried_hi[7]; /* m_dQueriedFields dwords 1..7 (indexes with more than 32 fields) */
  int wide_index;         /* ISphQwordSetup::m_bHasWideFields: the index has more than 32 fields */
  int has_wide;           /* m_bHasWideFields of the node: wide index AND some queried field >= 32 (searchnode.cpp:1837-1841, 2717-2725) */
  float idf;
  int atom_pos;
  int nodepos;
  int word_key; /* term id (same key = same word) */
  float boost;
  uint32_t rowid; /* NodeInfo_t::m_tRowID */
  uint64_t stored_hitpos;
} mnode;

typedef struct {
  int tag_qword;
  uint32_t exp_hitpos;
} fsm_state;

typedef struct enode enode;
struct enode {
  int kind;
  /* current doc */
  uint32_t rowid, fields;
  float tfidf;
  int started;
  /* TERM */
  mnode t;
  int tp_kind, tp_max; /* ExtTermPos_T (ExtConditional_T over ExtTerm_T): keeps the docs that hold an acceptable hit */
  /* MULTIAND */
  mnode* m;
  int n_m;
  int test_fields;
  /* twofers */
  enode *l, *r;
  int l_ok, r_ok, adv_l, adv_r, from_l, from_r;
  int qpos_reverse;
  uint16_t nodepos_l, nodepos_r;
  int right_empty, passthrough;
  /* PHRASE (ExtNWay_T<FSMphrase_c>) */
  enode* inner;
  int* atom_pos;
  int n_atoms;
  int* qpos_delta;
  int n_qpos_delta;
  fsm_state* states;
  int n_states, cap_states;
  hitvec myhits;
  /* QUORUM proper (ExtQuorum_c, no duplicate keywords): kids in query-position order; q_list = m_dChildren (indices into
     kids, reordered by RemoveFast as the keywords' doclists run dry) */
  enode** kids;
  int n_kids, q_thresh;
  int q_list[32], q_n;
  int* kid_ok;
  /* ORDER (ExtOrder_c): kids in query order; per-kid hits of the current candidate doc */
  hitvec* kid_hits;
  int ord_done;
  /* PROXIMITY (ExtNWay_T<FSMproximity_c>): same node, other state machine */
  int is_proximity;
  int max_distance;          /* m_iMaxDistance = XQNode_t::m_iOpArg */
  uint32_t px_min_qpos, px_qlen, px_exp_pos, px_words;
  int px_min_qindex;
  uint32_t* px_prox;         /* [px_qlen + 1] last position of the word with that query offset, UINT_MAX = none */
  int* px_deltas;
  hitvec tmp, tmp2;
  /* NEAR (ExtNWay_T<FSMmultinear_c>): the same node again with the multinear state machine (searchnode.cpp:680-716) */
  int is_near;
  int near_dist;                  /* m_iNear */
  uint32_t nr_prelast_p, nr_prelast_ml, nr_prelast_sl, nr_prelast_w, nr_last_p, nr_last_ml, nr_last_sl, nr_last_w;
  uint32_t nr_weight, nr_first_hit;
  uint16_t nr_first_npos, nr_first_qpos;
  uint16_t nr_npos[32];           /* m_dNpos (sorted) */
  int nr_n_npos;
  hit_t nr_ring[32];              /* m_dRing */
  int nr_iring;
  /* NOTNEAR (ExtNotNear_c): l = must, r = not */
  int nn_dist;
  /* UNIT (ExtUnit_c, SENTENCE / PARAGRAPH): l, r = the arguments, dot = the boundary keyword's term (NULL: none in the index) */
  enode* dot;
  int dot_ok;
  hitvec dot_hits;
  /* common */
  int atom; /* ExtNode_i::GetAtomPos */
  int64_t* p_fetched_docs;
  int64_t* p_fetched_hits;
};

static int en_next(enode* e);
static void en_hits(enode* e, hitvec* out);
static void en_hint(enode* e, uint32_t rowid);

static int en_docs_count(const enode* e) { /* GetDocsCount, searchnode.cpp:194, searchnode.h:83 (ExtConditional_T has no override) */
  return e->kind == EN_TERM && !e->tp_kind ? e->t.qw.docs : INT_MAX;
}

/* ISphQword::CollectHitMask (sphinxsearch.cpp:50-58): the doclist entry carries the low dword of the doc's field mask only
   (Assign32, sphinx.cpp:516-535); the other dwords come from the doc's hits */
static void collect_hit_mask(qword* q, uint32_t mask[8]) {
  memset(mask, 0, 8 * sizeof(uint32_t));
  mask[0] = q->fields32;
  qw_seek_hitlist(q, q->hitlist_pos);
  for (;;) {
    const uint32_t h = qw_next_hit(q);
    if (h == ORC_EMPTY_HIT) break;
    const uint32_t f = ORC_HIT_FIELD(h);
    mask[f >> 5] |= 1u << (f & 31u);
  }
}

/* FitsFields, searchnode.cpp:2727-2747 / 1925-1939: fields 0-31 from the doclist mask; a node that queries a field >= 32 of a wide
   index collects the doc's whole mask from its hits first */
static inline int fits_fields(mnode* n) {
  if (!n->has_wide) return (n->qw.fields32 & n->queried32) != 0;
  uint32_t m[8];
  collect_hit_mask(&n->qw, m);
  if (m[0] & n->queried32) return 1;
  for (int i = 0; i < 7; i++)
    if (m[i + 1] & n->queried_hi[i]) return 1;
  return 0;
}

/* m_dQueriedFields.Test ( HITMAN::GetField ( hit ) ) */
static inline int queried_test(const mnode* n, uint32_t field) {
  if (field < 32) return (n->queried32 >> field) & 1u;
  if (n->wide_index) return (n->queried_hi[(field >> 5) - 1] >> (field & 31u)) & 1u;
  return n->queried32 == 0xFFFFFFFFu; /* (a hit in a field the schema does not have: only the unrestricted mask lets it through) */
}

static inline float term_tfidf(uint32_t hits, float idf) {
  /* float(hits) / float(hits+SPH_BM25_K1) * idf  (searchnode.cpp:1946, 2828) */
  float fh = (float)hits;
  float den = fh + SPH_BM25_K1;
  float q = fh / den;
  return q * idf;
}

/* ---- TERM: ExtTerm_T (searchnode.cpp:1876-2020) ---- */
static void term_raw_hits(enode* e, hitvec* out) {
  mnode* n = &e->t;
  qw_seek_hitlist(&n->qw, n->stored_hitpos);
  for (;;) {
    uint32_t h = qw_next_hit(&n->qw);
    if (h == ORC_EMPTY_HIT) break;
    if (!queried_test(n, ORC_HIT_FIELD(h))) continue;
    hit_t t;
    t.rowid = e->rowid;
    t.hitpos = h;
    t.qpos = (uint16_t)n->atom_pos;
    t.nodepos = 0;
    t.weight = 1;
    t.matchlen = t.spanlen = 1;
    t.qposmask = 0;
    hv_push(out, &t);
    if (e->p_fetched_hits) (*e->p_fetched_hits)++;
  }
}

/* TermAcceptor_T<>::IsAcceptableHit (searchnode.cpp:2264-2285) */
static inline int termpos_accepts(int kind, int max_pos, uint32_t hitpos) {
  const int pos = (int)ORC_HIT_POS(hitpos), end = (hitpos >> 23) & 1u;
  switch (kind) {
    case ORC_TERMPOS_START: return pos == 1;
    case ORC_TERMPOS_END: return end;
    case ORC_TERMPOS_STARTEND: return pos == 1 && end;
    case ORC_TERMPOS_LIMIT: return pos <= max_pos;
    default: return 1;
  }
}

static int term_next(enode* e) {
  mnode* n = &e->t;
  if (!n->qw.docs) return 0;
  for (;;) {
    qw_read_next(&n->qw);
    if (n->qw.rowid == ORC_INVALID_ROWID) {
      n->qw.docs = 0;
      return 0;
    }
    if (!fits_fields(n)) continue;
    e->rowid = n->qw.rowid;
    e->fields = n->qw.fields32 & n->queried32;
    e->tfidf = term_tfidf(n->qw.match_hits, n->idf);
    n->stored_hitpos = n->qw.hitlist_pos;
    if (e->p_fetched_docs) (*e->p_fetched_docs)++;
    if (e->tp_kind) {
      /* ExtConditional_T::GetDocsChunk (searchnode.cpp:2332-2405): the doc passes as the term emitted it (fields, tfidf of
         ALL its hits) once one hit is acceptable; only the acceptable hits travel on */
      e->tmp.n = 0;
      term_raw_hits(e, &e->tmp);
      e->myhits.n = 0;
      for (int i = 0; i < e->tmp.n; i++)
        if (termpos_accepts(e->tp_kind, e->tp_max, e->tmp.p[i].hitpos)) hv_push(&e->myhits, &e->tmp.p[i]);
      if (!e->myhits.n) continue;
    }
    return 1;
  }
}

static void term_hits(enode* e, hitvec* out) {
  if (e->tp_kind) {
    for (int i = 0; i < e->myhits.n; i++) hv_push(out, &e->myhits.p[i]);
    return;
  }
  term_raw_hits(e, out);
}

/* ---- MULTIAND: ExtMultiAnd_T (searchnode.cpp:2716-3223) ---- */
static uint32_t mand_advance0(enode* e, int i) { /* Advance(int), :2836-2846 */
  mnode* n = &e->m[i];
  do {
    qw_read_next(&n->qw);
    n->rowid = n->qw.rowid;
  } while (n->rowid != ORC_INVALID_ROWID && !fits_fields(n));
  return n->rowid;
}

static uint32_t mand_advance_to(enode* e, int i, uint32_t rowid) { /* Advance(int,RowID_t), :2850-2861 */
  mnode* n = &e->m[i];
  if (rowid == n->rowid) return rowid;
  n->rowid = qw_advance_to(&n->qw, rowid);
  while (n->rowid != ORC_INVALID_ROWID && !fits_fields(n)) {
    qw_read_next(&n->qw);
    n->rowid = n->qw.rowid;
  }
  return n->rowid;
}

static int mand_advance_qwords(enode* e) { /* AdvanceQwords, :2865-2889 */
  uint32_t max_rowid = e->m[0].rowid;
  for (int i = 1; i < e->n_m; i++) {
    mnode* cur = &e->m[i];
    if (cur->rowid == max_rowid) continue;
    mand_advance_to(e, i, max_rowid);
    if (cur->rowid == ORC_INVALID_ROWID)
      return 0;
    else if (cur->rowid > max_rowid) {
      if (mand_advance_to(e, 0, cur->rowid) == ORC_INVALID_ROWID) return 0;
      max_rowid = e->m[0].rowid;
      i = 0;
    }
  }
  return 1;
}

static int mand_next(enode* e) { /* GetDocsChunk, :2893-2981, one doc per call */
  if (!e->started) {
    e->started = 1;
    if (!e->m[0].qw.docs) {
      e->m[0].rowid = ORC_INVALID_ROWID;
      return 0;
    }
    mand_advance0(e, 0);
  } else {
    if (e->m[0].rowid == ORC_INVALID_ROWID) return 0;
    mand_advance0(e, 0); /* "we assume that the 1st node returns the least docs" :2969 */
  }
  if (e->m[0].rowid == ORC_INVALID_ROWID) return 0;
  if (!mand_advance_qwords(e)) {
    e->m[0].rowid = ORC_INVALID_ROWID;
    return 0;
  }
  e->rowid = e->m[0].rowid;
  uint32_t mask = 0; /* GetDocFieldsMask :2810-2817 */
  float tfidf = 0.0f; /* GetTFIDF :2821-2832, summed in sorted-node order */
  for (int i = 0; i < e->n_m; i++) {
    mnode* n = &e->m[i];
    mask |= n->qw.fields32 & n->queried32;
    tfidf += term_tfidf(n->qw.match_hits, n->idf);
    n->stored_hitpos = n->qw.hitlist_pos;
  }
  e->fields = mask;
  e->tfidf = tfidf;
  if (e->p_fetched_docs) (*e->p_fetched_docs)++;
  return 1;
}

/* MergeHits2/3/N, :3047-3181: k-way merge by (hitpos, qpos); with distinct
   atom positions the order is total, so one generic merge restates all three.
   One literal quirk is kept: the 3-stream merge finishes with the 2-stream DoHitMerge
   (:3072-3077), whose AddHit calls test the hit's field against nodes 0 and 1
   (:3052-3054) whichever two streams are left; only CopyHits (:3082-3096) uses the
   right node again. */
static void mand_hits(enode* e, hitvec* out) {
  uint32_t cur[32];
  int k = e->n_m;
  for (int i = 0; i < k; i++) {
    qw_seek_hitlist(&e->m[i].qw, e->m[i].stored_hitpos);
    cur[i] = qw_next_hit(&e->m[i].qw);
  }
  int phase = (k == 3 && e->test_fields) ? 0 : 2, tl = 0, tr = 1;
  for (;;) {
    if (phase == 0 && !(cur[0] != ORC_EMPTY_HIT && cur[1] != ORC_EMPTY_HIT && cur[2] != ORC_EMPTY_HIT)) {
      if (cur[0] == ORC_EMPTY_HIT)
        tl = 1, tr = 2;
      else if (cur[1] == ORC_EMPTY_HIT)
        tl = 0, tr = 2;
      else
        tl = 0, tr = 1;
      phase = 1;
    }
    if (phase == 1 && !(cur[tl] != ORC_EMPTY_HIT && cur[tr] != ORC_EMPTY_HIT)) phase = 2;
    int best = -1;
    for (int i = 0; i < k; i++) {
      if (cur[i] == ORC_EMPTY_HIT) continue;
      if (best < 0 || cur[i] < cur[best] ||
          (cur[i] == cur[best] && (uint16_t)e->m[i].atom_pos < (uint16_t)e->m[best].atom_pos))
        best = i;
    }
    if (best < 0) break;
    mnode* n = &e->m[best];
    const mnode* fnode = n;
    if (phase == 1) fnode = &e->m[best == tl ? 0 : 1];
    if (!e->test_fields || queried_test(fnode, ORC_HIT_FIELD(cur[best]))) {
      hit_t t;
      t.rowid = e->rowid;
      t.hitpos = cur[best];
      t.qpos = (uint16_t)n->atom_pos;
      t.nodepos = (uint16_t)n->nodepos;
      t.weight = 1;
      t.matchlen = t.spanlen = 1;
      t.qposmask = 0;
      hv_push(out, &t);
      if (e->p_fetched_hits) (*e->p_fetched_hits)++;
    }
    cur[best] = qw_next_hit(&n->qw);
  }
}

/* ---- ExtAnd_c (searchnode.cpp:2570-2706) ---- */
static inline int hit_less(const hit_t* a, const hit_t* b) { /* IsHitLess :2611-2615 */
  return a->hitpos < b->hitpos || (a->hitpos == b->hitpos && a->qpos <= b->qpos);
}

static int and_next(enode* e) {
  if (!e->started) {
    e->started = 1;
    e->l_ok = en_next(e->l);
    if (!e->l_ok) return 0;
    en_hint(e->r, e->l->rowid); /* WarmupDocs w/ hint :127-142 */
    e->r_ok = en_next(e->r);
  } else {
    if (!e->l_ok || !e->r_ok) return 0;
    e->l_ok = en_next(e->l);
    e->r_ok = e->l_ok ? en_next(e->r) : 0;
  }
  while (e->l_ok && e->r_ok) {
    if (e->l->rowid == e->r->rowid) {
      e->rowid = e->l->rowid;
      e->fields = e->l->fields | e->r->fields;
      e->tfidf = e->l->tfidf + e->r->tfidf; /* :2592 */
      return 1;
    } else if (e->l->rowid < e->r->rowid) {
      en_hint(e->l, e->r->rowid);
      e->l_ok = en_next(e->l);
    } else {
      en_hint(e->r, e->l->rowid);
      e->r_ok = en_next(e->r);
    }
  }
  return 0;
}

static int cmp_hit_reverse(const void* pa, const void* pb) { /* CmpAndHitReverse_fn :2618-2624 */
  const hit_t* a = (const hit_t*)pa;
  const hit_t* b = (const hit_t*)pb;
  if (a->rowid != b->rowid) return a->rowid < b->rowid ? -1 : 1;
  if (a->hitpos != b->hitpos) return a->hitpos < b->hitpos ? -1 : 1;
  if (a->qpos != b->qpos) return a->qpos > b->qpos ? -1 : 1;
  return 0;
}

static void merge_hits(hitvec* out, const hitvec* L, const hitvec* R, uint16_t npl, uint16_t npr) {
  int i = 0, j = 0;
  while (i < L->n && j < R->n) {
    if (hit_less(&L->p[i], &R->p[j])) {
      hv_push(out, &L->p[i++]);
      if (npl) out->p[out->n - 1].nodepos = npl;
    } else {
      hv_push(out, &R->p[j++]);
      if (npr) out->p[out->n - 1].nodepos = npr;
    }
  }
  while (i < L->n) {
    hv_push(out, &L->p[i++]);
    if (npl) out->p[out->n - 1].nodepos = npl;
  }
  while (j < R->n) {
    hv_push(out, &R->p[j++]);
    if (npr) out->p[out->n - 1].nodepos = npr;
  }
}

static void and_hits(enode* e, hitvec* out) {
  e->tmp.n = 0;
  e->tmp2.n = 0;
  en_hits(e->l, &e->tmp);
  en_hits(e->r, &e->tmp2);
  /* ExtAnd_c::CollectHits only emits a doc's hits once both sides have one (:2639-2702) */
  if (!e->tmp.n || !e->tmp2.n) return;
  int base = out->n;
  merge_hits(out, &e->tmp, &e->tmp2, e->nodepos_l, e->nodepos_r);
  if (e->qpos_reverse) qsort(out->p + base, (size_t)(out->n - base), sizeof(hit_t), cmp_hit_reverse);
}

/* ---- ExtOr_c (searchnode.cpp:3465-3545) ---- */
static int or_next(enode* e) {
  if (!e->started) {
    e->started = 1;
    e->l_ok = en_next(e->l);
    e->r_ok = en_next(e->r);
  } else {
    if (e->adv_l) e->l_ok = en_next(e->l);
    if (e->adv_r) e->r_ok = en_next(e->r);
  }
  e->adv_l = e->adv_r = 0;
  e->from_l = e->from_r = 0;
  if (!e->l_ok && !e->r_ok) return 0;
  if (e->l_ok && e->r_ok) {
    if (e->l->rowid == e->r->rowid) {
      e->rowid = e->l->rowid;
      e->fields = e->l->fields | e->r->fields;
      e->tfidf = e->l->tfidf + e->r->tfidf;
      e->adv_l = e->adv_r = 1;
      e->from_l = e->from_r = 1;
    } else if (e->l->rowid < e->r->rowid) {
      e->rowid = e->l->rowid, e->fields = e->l->fields, e->tfidf = e->l->tfidf;
      e->adv_l = e->from_l = 1;
    } else {
      e->rowid = e->r->rowid, e->fields = e->r->fields, e->tfidf = e->r->tfidf;
      e->adv_r = e->from_r = 1;
    }
  } else if (e->l_ok) {
    e->rowid = e->l->rowid, e->fields = e->l->fields, e->tfidf = e->l->tfidf;
    e->adv_l = e->from_l = 1;
  } else {
    e->rowid = e->r->rowid, e->fields = e->r->fields, e->tfidf = e->r->tfidf;
    e->adv_r = e->from_r = 1;
  }
  return 1;
}

static void or_hits(enode* e, hitvec* out) {
  e->tmp.n = 0;
  e->tmp2.n = 0;
  if (e->from_l) en_hits(e->l, &e->tmp);
  if (e->from_r) en_hits(e->r, &e->tmp2);
  merge_hits(out, &e->tmp, &e->tmp2, 0, 0);
}

/* ---- ExtMaybe_c (searchnode.cpp:3565-3604); hits via ExtOr_c::CollectHits ---- */
static int maybe_next(enode* e) {
  if (!e->started) {
    e->started = 1;
    e->r_ok = en_next(e->r);
    if (!e->r_ok) e->right_empty = 1;
  } else if (e->adv_r && !e->right_empty) {
    e->r_ok = en_next(e->r);
    if (!e->r_ok) e->right_empty = 1;
  }
  e->adv_r = 0;
  e->from_l = 1;
  e->from_r = 0;
  e->l_ok = en_next(e->l);
  if (!e->l_ok) return 0;
  while (!e->right_empty && e->r->rowid < e->l->rowid) {
    e->r_ok = en_next(e->r);
    if (!e->r_ok) e->right_empty = 1;
  }
  e->rowid = e->l->rowid;
  e->fields = e->l->fields;
  e->tfidf = e->l->tfidf;
  if (!e->right_empty && e->r->rowid == e->l->rowid) {
    e->fields = e->l->fields | e->r->fields;
    e->tfidf = e->l->tfidf + e->r->tfidf;
    e->from_r = 1;
    e->adv_r = 1;
  }
  return 1;
}

/* ---- ExtAndNot_c (searchnode.cpp:3618-3694) ---- */
static int andnot_next(enode* e) {
  if (!e->started) {
    e->started = 1;
    e->r_ok = en_next(e->r);
  }
  for (;;) {
    e->l_ok = en_next(e->l);
    if (!e->l_ok) return 0;
    while (e->r_ok && e->r->rowid < e->l->rowid) e->r_ok = en_next(e->r);
    if (e->r_ok && e->r->rowid == e->l->rowid) continue; /* rejected */
    e->rowid = e->l->rowid;
    e->fields = e->l->fields;
    e->tfidf = e->l->tfidf;
    return 1;
  }
}

/* ---- ExtNWay_T<FSMphrase_c> (searchnode.cpp:3792-3953) ---- */
static void fsm_reset(enode* e) { e->n_states = 0; }

static int fsm_hit(enode* e, const hit_t* h) { /* FSMphrase_c::HitFSM :3901-3947 */
  uint32_t hpf = ORC_HIT_POSWITHFIELD(h->hitpos);
  if (h->qpos == e->atom_pos[0]) {
    if (e->n_states == e->cap_states) {
      e->cap_states = e->cap_states ? e->cap_states * 2 : 16;
      e->states = (fsm_state*)realloc(e->states, (size_t)e->cap_states * sizeof(fsm_state));
    }
    e->states[e->n_states].tag_qword = 0;
    e->states[e->n_states].exp_hitpos = hpf + (uint32_t)e->qpos_delta[0];
    e->n_states++;
  }
  for (int i = e->n_states - 1; i >= 0; i--) {
    if (e->states[i].exp_hitpos < hpf) {
      e->states[i] = e->states[--e->n_states]; /* RemoveFast */
      continue;
    }
    if (e->states[i].exp_hitpos == hpf && e->atom_pos[e->states[i].tag_qword + 1] == h->qpos) {
      e->states[i].tag_qword++;
      e->states[i].exp_hitpos = hpf + (uint32_t)e->qpos_delta[h->qpos - e->atom_pos[0]];
    }
    if (e->states[i].tag_qword == e->n_atoms - 1) {
      uint32_t spanlen = (uint32_t)(e->atom_pos[e->n_atoms - 1] - e->atom_pos[0]);
      hit_t t;
      t.rowid = h->rowid;
      t.hitpos = hpf - spanlen;
      t.qpos = (uint16_t)e->atom_pos[0];
      t.nodepos = 0;
      t.matchlen = t.spanlen = (uint16_t)(spanlen + 1);
      t.weight = (uint32_t)e->n_atoms;
      t.qposmask = 0;
      hv_push(&e->myhits, &t);
      fsm_reset(e);
      return 1;
    }
  }
  return 0;
}

/* FSMproximity_c (searchnode.cpp:3958-4075) */
static void px_reset(enode* e) { /* ResetFSM :4068-4075 */
  e->px_exp_pos = 0;
  e->px_words = 0;
  e->px_min_qindex = -1;
  for (uint32_t i = 0; i <= e->px_qlen; i++) e->px_prox[i] = UINT32_MAX;
}

static int cmp_int(const void* a, const void* b) {
  int x = *(const int*)a, y = *(const int*)b;
  return x < y ? -1 : x > y;
}

static int px_hit(enode* e, const hit_t* h) { /* HitFSM :3973-4065 */
  int qindex = (int)h->qpos - (int)e->px_min_qpos;
  uint32_t hpf = ORC_HIT_POSWITHFIELD(h->hitpos);
  const int n = (int)e->px_qlen + 1;
  if (e->px_prox[qindex] == UINT32_MAX) e->px_words++;
  e->px_prox[qindex] = hpf;
  if (hpf >= e->px_exp_pos || qindex == e->px_min_qindex) {
    e->px_min_qindex = qindex;
    int min_pos = (int)(hpf - e->px_qlen - (uint32_t)e->max_distance);
    for (int i = 0; i < n; i++)
      if (e->px_prox[i] != UINT32_MAX) {
        if ((int)e->px_prox[i] <= min_pos) {
          e->px_prox[i] = UINT32_MAX;
          e->px_words--;
          continue;
        }
        if (e->px_prox[i] < hpf) {
          e->px_min_qindex = i;
          hpf = e->px_prox[i];
        }
      }
    e->px_exp_pos = e->px_prox[e->px_min_qindex] + e->px_qlen + (uint32_t)e->max_distance;
  }
  if (e->px_words != (uint32_t)e->n_atoms) return 0;
  /* phrase weight from the words' deltas */
  uint32_t umax = 0;
  for (int i = 0; i < n; i++)
    if (e->px_prox[i] != UINT32_MAX) {
      e->px_deltas[i] = (int)(e->px_prox[i] - (uint32_t)i);
      if (e->px_prox[i] > umax) umax = e->px_prox[i];
    } else
      e->px_deltas[i] = INT_MAX;
  qsort(e->px_deltas, (size_t)n, sizeof(int), cmp_int);
  uint32_t cur_weight = 0, weight = 0;
  int last = -INT_MAX;
  for (int i = 0; i < n && e->px_deltas[i] != INT_MAX; i++) {
    if (e->px_deltas[i] == last)
      cur_weight++;
    else {
      weight += cur_weight ? (1 + cur_weight) : 0;
      cur_weight = 0;
    }
    last = e->px_deltas[i];
  }
  weight += cur_weight ? (1 + cur_weight) : 0;
  if (!weight) weight = 1;
  hit_t t;
  t.rowid = h->rowid;
  t.hitpos = e->px_prox[e->px_min_qindex];
  t.qpos = (uint16_t)e->px_min_qpos;
  t.nodepos = 0;
  t.matchlen = t.spanlen = (uint16_t)(umax - e->px_prox[e->px_min_qindex] + 1);
  t.weight = weight;
  t.qposmask = 0;
  hv_push(&e->myhits, &t);
  /* remove the current min, force a recompute */
  e->px_prox[e->px_min_qindex] = UINT32_MAX;
  e->px_min_qindex = -1;
  e->px_words--;
  e->px_exp_pos = 0;
  return 1;
}

/* FSMmultinear_c (searchnode.cpp:4080-4318) */
static void near_reset(enode* e) { e->nr_iring = 0, e->nr_last_p = 0, e->nr_prelast_p = 0; } /* ResetFSM :4291-4294 */
static int near_ring_tail(const enode* e) { return (e->nr_iring + e->nr_n_npos - 1) % e->n_atoms; }
static void near_add2ring(enode* e, const hit_t* h) {
  if (e->n_atoms != 2) e->nr_ring[near_ring_tail(e)] = *h;
}
static void near_shift_ring(enode* e) {
  if (++e->nr_iring == e->n_atoms) e->nr_iring = 0;
}
static int near_npos_find(const enode* e, uint16_t v) { /* BinarySearch over the sorted m_dNpos */
  for (int i = 0; i < e->nr_n_npos; i++)
    if (e->nr_npos[i] == v) return i;
  return -1;
}
static void near_npos_sort(enode* e) {
  for (int i = 1; i < e->nr_n_npos; i++)
    for (int j = i; j > 0 && e->nr_npos[j - 1] > e->nr_npos[j]; j--) {
      uint16_t t = e->nr_npos[j];
      e->nr_npos[j] = e->nr_npos[j - 1];
      e->nr_npos[j - 1] = t;
    }
}
static void near_npos_insert(enode* e, int at, uint16_t v) {
  for (int i = e->nr_n_npos; i > at; i--) e->nr_npos[i] = e->nr_npos[i - 1];
  e->nr_npos[at] = v;
  e->nr_n_npos++;
}

static int near_hit(enode* e, const hit_t* h) { /* HitFSM :4096-4288 */
  const int twofer = e->n_atoms == 2;
  const uint32_t hpf = ORC_HIT_POSWITHFIELD(h->hitpos);
  const uint16_t npos = h->nodepos, qpos = h->qpos;
  /* skip dupe hit (may be emitted by OR node, for example) */
  if (e->nr_last_p == hpf) {
    if (twofer && npos < e->nr_first_npos) { /* leftmost (in the query) of all dupes: 'a NEAR/2 a' */
      e->nr_first_qpos = qpos;
      e->nr_first_npos = npos;
      return 0;
    } else if (!twofer && npos < e->nr_ring[near_ring_tail(e)].nodepos) { /* 'a NEAR/2 a NEAR/2 a' */
      if (near_npos_find(e, npos) < 0) {
        int at = near_npos_find(e, e->nr_ring[near_ring_tail(e)].nodepos);
        if (at >= 0) e->nr_npos[at] = npos; /* (the reference dereferences the search result unchecked) */
        near_npos_sort(e);
        e->nr_ring[near_ring_tail(e)].nodepos = npos;
        e->nr_ring[near_ring_tail(e)].qpos = qpos;
      }
      return 0;
    } else if (e->nr_prelast_p && e->nr_last_ml < h->matchlen) { /* the hit is a subset of another one: roll back */
      e->nr_last_ml = e->nr_prelast_ml;
      e->nr_last_sl = e->nr_prelast_sl;
      e->nr_first_hit = e->nr_last_p = e->nr_prelast_p;
      e->nr_weight = e->nr_weight - e->nr_last_w + e->nr_prelast_w;
    } else
      return 0;
  }
  /* probably new chain */
  if (e->nr_last_p == 0 || (e->nr_last_p + e->nr_last_ml + (uint32_t)e->near_dist) <= hpf) {
    e->nr_first_hit = e->nr_last_p = hpf;
    e->nr_last_ml = h->matchlen;
    e->nr_last_sl = h->spanlen;
    e->nr_weight = e->nr_last_w = h->weight;
    if (twofer) {
      e->nr_first_qpos = qpos;
      e->nr_first_npos = npos;
    } else {
      e->nr_n_npos = 1;
      e->nr_npos[0] = npos;
      near_add2ring(e, h);
    }
    return 0;
  }
  if (twofer) {
    /* special case for twofer: hold the overlapping */
    if ((e->nr_first_hit + e->nr_last_ml) > hpf && (e->nr_first_hit + e->nr_last_ml) < (hpf + h->matchlen) && e->nr_last_ml != h->matchlen) {
      e->nr_first_hit = e->nr_last_p = hpf;
      e->nr_last_ml = h->matchlen;
      e->nr_last_sl = h->spanlen;
      e->nr_weight = e->nr_last_w = h->weight;
      e->nr_first_qpos = qpos;
      e->nr_first_npos = npos;
      return 0;
    }
    if (npos == e->nr_first_npos) {
      if (e->nr_last_p < hpf) {
        e->nr_prelast_ml = e->nr_last_ml;
        e->nr_prelast_sl = e->nr_last_sl;
        e->nr_prelast_p = e->nr_last_p;
        e->nr_prelast_w = h->weight;
        e->nr_first_hit = e->nr_last_p = hpf;
        e->nr_last_ml = h->matchlen;
        e->nr_last_sl = h->spanlen;
        e->nr_weight = e->nr_last_w = e->nr_prelast_w;
        e->nr_first_qpos = qpos;
        e->nr_first_npos = npos;
      }
      return 0;
    }
  } else {
    if (npos < e->nr_npos[0]) {
      if (qpos < e->nr_first_qpos) e->nr_first_qpos = qpos;
      near_npos_insert(e, 0, npos);
    } else if (npos > e->nr_npos[e->nr_n_npos - 1]) {
      if (qpos < e->nr_first_qpos) e->nr_first_qpos = qpos;
      near_npos_insert(e, e->nr_n_npos, npos);
    } else if (npos != e->nr_npos[0] && npos != e->nr_npos[e->nr_n_npos - 1]) {
      int end = e->nr_n_npos, start = 0, mid = -1;
      while (end - start > 1) {
        mid = (start + end) / 2;
        if (npos == e->nr_npos[mid]) {
          const hit_t* rh = &e->nr_ring[e->nr_iring];
          if (npos == rh->nodepos) { /* last addition same as the first: shift */
            e->nr_weight -= rh->weight;
            e->nr_first_hit = ORC_HIT_POSWITHFIELD(rh->hitpos);
            near_shift_ring(e);
          } else if (npos == e->nr_ring[near_ring_tail(e)].nodepos)
            e->nr_weight -= e->nr_ring[near_ring_tail(e)].weight;
          else
            return 0;
        }
        if (npos < e->nr_npos[mid])
          end = mid;
        else
          start = mid;
      }
      near_npos_insert(e, end, npos);
      if (qpos < e->nr_first_qpos) e->nr_first_qpos = qpos;
    } else if (npos == e->nr_ring[e->nr_iring].nodepos) { /* same as the head: shift */
      e->nr_weight -= e->nr_ring[e->nr_iring].weight;
      e->nr_first_hit = ORC_HIT_POSWITHFIELD(e->nr_ring[e->nr_iring].hitpos);
      near_shift_ring(e);
    } else if (npos == e->nr_ring[near_ring_tail(e)].nodepos) /* same as the tail: move the tail onto it */
      e->nr_weight -= e->nr_ring[near_ring_tail(e)].weight;
    else
      return 0;
  }
  e->nr_weight += h->weight;
  e->nr_last_ml = h->matchlen;
  e->nr_last_sl = h->spanlen;
  near_add2ring(e, h);
  /* finally got the whole chain - emit it (no overlapping in generic chains) */
  if (twofer || e->n_atoms == e->nr_n_npos) {
    hit_t t;
    t.rowid = h->rowid;
    t.hitpos = e->nr_first_hit;
    t.matchlen = (uint16_t)(hpf - e->nr_first_hit + e->nr_last_ml);
    t.weight = e->nr_weight;
    t.nodepos = 0;
    t.qposmask = 0;
    e->nr_prelast_p = 0;
    t.qpos = e->nr_first_qpos < h->qpos ? e->nr_first_qpos : h->qpos;
    if (twofer) { /* for exactly 2 words allow overlapping: shift the chain, not reset it */
      t.spanlen = 2;
      e->nr_first_hit = e->nr_last_p = hpf;
      e->nr_weight = h->weight;
      e->nr_first_qpos = h->qpos;
    } else {
      t.spanlen = (uint16_t)e->nr_n_npos;
      e->nr_last_p = 0;
    }
    hv_push(&e->myhits, &t);
    return 1;
  }
  e->nr_last_p = hpf;
  return 0;
}

/* ---- ExtNotNear_c (searchnode.cpp:5325-5478): the MUST side's docs; where the NOT side holds the doc too, only the MUST
   hits that no later NOT hit comes within the distance of survive, and the doc stays iff one does ---- */
static int notnear_next(enode* e) {
  for (;;) {
    if (!en_next(e->l)) return 0;
    const uint32_t rowid = e->l->rowid;
    if (!e->right_empty && (!e->r_ok || e->r->rowid < rowid)) {
      en_hint(e->r, rowid);
      do {
        e->r_ok = en_next(e->r);
      } while (e->r_ok && e->r->rowid < rowid);
      if (!e->r_ok) e->right_empty = 1;
    }
    e->myhits.n = 0;
    e->tmp.n = 0;
    en_hits(e->l, &e->tmp);
    if (e->right_empty || e->r->rowid != rowid) { /* copy none matched from MUST */
      for (int i = 0; i < e->tmp.n; i++) hv_push(&e->myhits, &e->tmp.p[i]);
    } else { /* FilterHits :5352-5380 */
      e->tmp2.n = 0;
      en_hits(e->r, &e->tmp2);
      int j = 0;
      for (int i = 0; i < e->tmp.n; i++) {
        const hit_t* must = &e->tmp.p[i];
        const uint32_t pm = ORC_HIT_POSWITHFIELD(must->hitpos);
        while (j < e->tmp2.n && ORC_HIT_POSWITHFIELD(e->tmp2.p[j].hitpos) < pm) j++; /* the NOT hit next after this MUST hit */
        /* (no NOT hit left: bRightEmpty / the tail copy -- the MUST hit stays either way) */
        if (j >= e->tmp2.n || pm + must->matchlen - 1 + (uint32_t)e->nn_dist < ORC_HIT_POSWITHFIELD(e->tmp2.p[j].hitpos)) hv_push(&e->myhits, must);
      }
      if (!e->myhits.n) continue;
    }
    e->rowid = rowid;
    e->fields = e->l->fields;
    e->tfidf = e->l->tfidf;
    return 1;
  }
}


/* ---- ExtUnit_c, the SENTENCE / PARAGRAPH operators (searchnode.cpp:4983-5310): AND of the two arguments; where the doc
   also holds boundary hits ("dots"), only the hit pairs no dot separates match, and the hits of every matching unit are
   copied (FilterHits :5082-5166).  Positions compare raw (field, end flag, position), as the reference's Hitpos_t does. ---- */
static int unit_filter(enode* e) {
  const hitvec *L = &e->tmp, *R = &e->tmp2, *D = &e->dot_hits;
  int i1 = 0, i2 = 0, id = 0, registered = 0;
  uint32_t end = D->n ? 0 : UINT_MAX; /* no dots in the doc: copy all hits */
  for (;;) {
    if (end) { /* in a matched unit: copy hits until the next dot */
      const int v1 = i1 < L->n && L->p[i1].hitpos < end, v2 = i2 < R->n && R->p[i2].hitpos < end;
      if (!v1 && !v2) {
        end = 0;
        if (i1 < L->n && i2 < R->n) continue; /* perhaps more units in this doc */
        break;
      }
      registered = 1;
      if (v1 && (!v2 || hit_less(&L->p[i1], &R->p[i2])))
        hv_push(&e->myhits, &L->p[i1++]);
      else
        hv_push(&e->myhits, &R->p[i2++]);
    } else { /* the next hit pair */
      if (i1 >= L->n || i2 >= R->n) break; /* (the reference asserts both; an argument without hits matches nothing) */
      const uint32_t a = L->p[i1].hitpos, b = R->p[i2].hitpos, umin = a < b ? a : b, umax = a < b ? b : a;
      while (id < D->n && D->p[id].hitpos <= umin) id++; /* SkipHitsLtePos ( pDotHit, uMin ) */
      if (id >= D->n) {
        end = UINT_MAX; /* no more dots past the pair's start: match, copy to the doc's end */
        continue;
      }
      if (D->p[id].hitpos < umax) { /* "A dot B": rewind both sides past this dot */
        const uint32_t dp = D->p[id].hitpos;
        while (i1 < L->n && L->p[i1].hitpos <= dp) i1++;
        if (i1 >= L->n) break;
        while (i2 < R->n && R->p[i2].hitpos <= dp) i2++;
        if (i2 >= R->n) break;
        continue;
      }
      while (id < D->n && D->p[id].hitpos <= umax) id++;
      end = id >= D->n ? UINT_MAX : D->p[id].hitpos;
    }
  }
  return registered;
}

static int unit_next(enode* e) {
  if (!e->started) {
    e->started = 1;
    e->l_ok = en_next(e->l);
    if (!e->l_ok) return 0;
    en_hint(e->r, e->l->rowid);
    e->r_ok = en_next(e->r);
    e->dot_ok = e->dot ? en_next(e->dot) : 0;
  } else {
    if (!e->l_ok || !e->r_ok) return 0;
    e->l_ok = en_next(e->l);
    e->r_ok = e->l_ok ? en_next(e->r) : 0;
  }
  while (e->l_ok && e->r_ok) {
    if (e->l->rowid < e->r->rowid) {
      en_hint(e->l, e->r->rowid);
      e->l_ok = en_next(e->l);
      continue;
    }
    if (e->l->rowid > e->r->rowid) {
      en_hint(e->r, e->l->rowid);
      e->r_ok = en_next(e->r);
      continue;
    }
    const uint32_t rowid = e->l->rowid;
    while (e->dot_ok && e->dot->rowid < rowid) {
      en_hint(e->dot, rowid);
      e->dot_ok = en_next(e->dot);
    }
    e->tmp.n = e->tmp2.n = e->dot_hits.n = e->myhits.n = 0;
    en_hits(e->l, &e->tmp);
    en_hits(e->r, &e->tmp2);
    if (e->dot_ok && e->dot->rowid == rowid) en_hits(e->dot, &e->dot_hits);
    if (unit_filter(e)) {
      e->rowid = rowid;
      e->fields = e->l->fields | e->r->fields;
      e->tfidf = e->l->tfidf + e->r->tfidf;
      return 1;
    }
    e->l_ok = en_next(e->l);
    e->r_ok = e->l_ok ? en_next(e->r) : 0;
  }
  return 0;
}

static int phrase_next(enode* e) { /* ExtNWay_T::GetDocsChunk :3806-3848 */
  for (;;) {
    if (!en_next(e->inner)) return 0;
    e->tmp.n = 0;
    en_hits(e->inner, &e->tmp);
    e->myhits.n = 0;
    if (e->is_near)
      near_reset(e);
    else if (e->is_proximity)
      px_reset(e);
    else
      fsm_reset(e);
    int matched = 0;
    for (int i = 0; i < e->tmp.n; i++) {
      const hit_t* h = &e->tmp.p[i];
      if ((e->is_near ? near_hit(e, h) : e->is_proximity ? px_hit(e, h) : fsm_hit(e, h)) && !matched) {
        matched = 1;
        e->rowid = h->rowid;
        e->fields = 1u << (ORC_HIT_FIELD(h->hitpos) & 31);
        e->tfidf = e->inner->tfidf;
      }
    }
    if (matched) return 1;
  }
}

static void phrase_hits(enode* e, hitvec* out) {
  for (int i = 0; i < e->myhits.n; i++) hv_push(out, &e->myhits.p[i]);
}

/* ---- ExtQuorum_c (searchnode.cpp:4342-4403, 4466-4545, 4567-4572, 4602-4617), m_bHasDupes == false ---- */
static int cmp_quorum_hit(const void* pa, const void* pb) { /* QuorumCmpHitPos_fn :4548-4564: positions WITHOUT the end flag */
  const hit_t* a = (const hit_t*)pa;
  const hit_t* b = (const hit_t*)pb;
  uint32_t x = ORC_HIT_POSWITHFIELD(a->hitpos), y = ORC_HIT_POSWITHFIELD(b->hitpos);
  if (x != y) return x < y ? -1 : 1;
  if (a->qpos != b->qpos) return a->qpos < b->qpos ? -1 : 1;
  return 0;
}

static void quorum_remove_fast(enode* e, int i) { e->q_list[i] = e->q_list[--e->q_n]; }

static int quorum_next(enode* e) {
  if (!e->started) { /* warmup :4468-4483 */
    e->started = 1;
    for (int i = 0; i < e->q_n; i++) {
      int c = e->q_list[i];
      e->kid_ok[c] = en_next(e->kids[c]);
      if (!e->kid_ok[c]) {
        quorum_remove_fast(e, i);
        i--;
      }
    }
  }
  for (;;) {
    if (e->q_n < e->q_thresh) return 0; /* iQuorumLeft >= m_iThresh */
    /* find the min rowid, count occurrences; tfidf adds up in m_dChildren order */
    uint32_t cand = ORC_INVALID_ROWID, fields = 0;
    float tfidf = 0.0f;
    int quorum = 0;
    for (int i = 0; i < e->q_n; i++) {
      enode* k = e->kids[e->q_list[i]];
      if (k->rowid < cand) {
        cand = k->rowid, fields = k->fields, tfidf = k->tfidf;
        quorum = 1;
      } else if (k->rowid == cand) {
        fields |= k->fields;
        tfidf += k->tfidf;
        quorum++;
      }
    }
    int matched = quorum >= e->q_thresh;
    if (matched) { /* CollectMatchingHits :4604-4617 + CollectHits' sort */
      e->myhits.n = 0;
      for (int i = 0; i < e->q_n; i++) {
        enode* k = e->kids[e->q_list[i]];
        if (k->rowid == cand) en_hits(k, &e->myhits);
      }
      qsort(e->myhits.p, (size_t)e->myhits.n, sizeof(hit_t), cmp_quorum_hit);
      e->rowid = cand, e->fields = fields, e->tfidf = tfidf;
    }
    /* advance the children that sit on the candidate :4517-4537 */
    for (int i = 0; i < e->q_n; i++) {
      int c = e->q_list[i];
      if (e->kids[c]->rowid != cand) continue;
      e->kid_ok[c] = en_next(e->kids[c]);
      if (!e->kid_ok[c]) {
        quorum_remove_fast(e, i);
        i--;
      }
    }
    if (matched) return 1;
  }
}

/* ---- dispatch ---- */

/* ---- ExtOrder_c, the BEFORE operator (searchnode.cpp:4657-4936) ---- */
/* GetMatchingHits (:4734-4829): the hits of all children in ascending position (ties: the lowest child), two trackers --
   the longest in-order subsequence so far and the most recently started one; a full subsequence is flushed to the output */
static int order_matching_hits(enode* e) {
  const int n = e->n_kids;
  hit_t acc_l[32], acc_r[32];
  int len_l = 0, len_r = 0, pos_l = 0, pos_r = 0, field = -1;
  int cur[32];
  for (int i = 0; i < n; i++) cur[i] = 0;
  const int old = e->myhits.n;
  for (;;) {
    uint32_t best_pos = UINT_MAX; /* GetChildIdWithNextHit :4706-4731 */
    int c = -1;
    for (int i = 0; i < n; i++)
      if (cur[i] < e->kid_hits[i].n && ORC_HIT_POSWITHFIELD(e->kid_hits[i].p[cur[i]].hitpos) < best_pos) {
        best_pos = ORC_HIT_POSWITHFIELD(e->kid_hits[i].p[cur[i]].hitpos);
        c = i;
      }
    if (c < 0) break;
    const hit_t* h = &e->kid_hits[c].p[cur[c]];
    const int hfield = (int)ORC_HIT_FIELD(h->hitpos), hpos = (int)ORC_HIT_POS(h->hitpos);
    if (hfield != field) { /* new field: both trackers start over */
      len_l = len_r = 0;
      if (c == 0) {
        acc_l[len_l++] = *h;
        pos_l = hpos + h->spanlen;
        field = hfield;
      }
    } else if (c == len_l && hpos >= pos_l) {
      acc_l[len_l++] = *h;
      pos_l = hpos + h->spanlen;
      if (len_l == n) {
        for (int i = 0; i < len_l; i++) hv_push(&e->myhits, &acc_l[i]);
        len_l = len_r = 0;
        pos_r = pos_l;
      }
    } else if (c == 0) {
      len_r = 0;
      acc_r[len_r++] = *h;
      pos_r = hpos + h->spanlen;
      if (!len_l) {
        acc_l[len_l++] = *h;
        pos_l = hpos + h->spanlen;
      }
    } else if (c == len_r && hpos >= pos_r) {
      acc_r[len_r++] = *h;
      pos_r = hpos + h->spanlen;
      if (len_r == len_l) {
        for (int i = 0; i < len_r; i++) acc_l[i] = acc_r[i];
        len_r = 0;
        pos_l = pos_r;
      }
    }
    cur[c]++;
  }
  return old != e->myhits.n;
}

static int order_next(enode* e) { /* GetDocsChunk :4832-4929 */
  if (e->ord_done) return 0;
  if (!e->started) {
    e->started = 1;
    for (int i = 0; i < e->n_kids; i++)
      if (!en_next(e->kids[i])) {
        e->ord_done = 1;
        return 0;
      }
  }
  for (;;) {
    uint32_t rowid = e->kids[0]->rowid;
    int i = 1;
    while (i < e->n_kids) {
      while (e->kids[i]->rowid < rowid)
        if (!en_next(e->kids[i])) {
          e->ord_done = 1;
          return 0;
        }
      if (e->kids[i]->rowid > rowid) {
        rowid = e->kids[i]->rowid;
        i = 0;
        continue;
      }
      i++;
    }
    for (int k = 0; k < e->n_kids; k++) {
      e->kid_hits[k].n = 0;
      en_hits(e->kids[k], &e->kid_hits[k]);
    }
    e->myhits.n = 0;
    const int matched = order_matching_hits(e);
    if (matched) { /* m_dDocs[iDoc++] = *m_dChildDoc[0]: the first child's doc, as it is */
      e->rowid = rowid;
      e->fields = e->kids[0]->fields;
      e->tfidf = e->kids[0]->tfidf;
    }
    if (!en_next(e->kids[0])) e->ord_done = 1;
    if (matched) return 1;
    if (e->ord_done) return 0;
  }
}

static int en_next(enode* e) {
  int ok;
  switch (e->kind) {
    case EN_TERM: ok = term_next(e); break;
    case EN_MULTIAND: ok = mand_next(e); break;
    case EN_AND: ok = and_next(e); break;
    case EN_OR: ok = or_next(e); break;
    case EN_MAYBE: ok = maybe_next(e); break;
    case EN_ANDNOT: ok = andnot_next(e); break;
    case EN_QUORUM: ok = quorum_next(e); break;
    case EN_ORDER: ok = order_next(e); break;
    case EN_NOTNEAR: ok = notnear_next(e); break;
    case EN_UNIT: ok = unit_next(e); break;
    default: ok = phrase_next(e); break;
  }
  if (!ok) e->rowid = ORC_INVALID_ROWID;
  return ok;
}

static void en_hits(enode* e, hitvec* out) {
  switch (e->kind) {
    case EN_TERM: term_hits(e, out); break;
    case EN_MULTIAND: mand_hits(e, out); break;
    case EN_AND: and_hits(e, out); break;
    case EN_OR:
    case EN_MAYBE: or_hits(e, out); break;
    case EN_ANDNOT: en_hits(e->l, out); break; /* :3686-3694 */
    case EN_QUORUM:
    case EN_ORDER:
    case EN_NOTNEAR:
    case EN_UNIT:
      for (int i = 0; i < e->myhits.n; i++) hv_push(out, &e->myhits.p[i]);
      break;
    default: phrase_hits(e, out); break;
  }
}

static void en_hint(enode* e, uint32_t rowid) {
  switch (e->kind) {
    case EN_TERM: /* ExtTerm_T::HintRowID :2089-2098 */
      if (e->t.qw.docs) qw_hint(&e->t.qw, rowid);
      break;
    case EN_MULTIAND: /* :3333-3346; hinting is a speed-up only, results are unchanged without it */
      break;
    case EN_AND:
    case EN_OR:
    case EN_MAYBE:
    case EN_ANDNOT: /* ExtTwofer_c::HintRowID :2543-2547 */
      if (!e->started) {
        en_hint(e->l, rowid);
        en_hint(e->r, rowid);
      }
      break;
    default: break;
  }
}

static void en_free(enode* e) {
  if (!e) return;
  if (e->kind == EN_TERM) qw_free(&e->t.qw);
  for (int i = 0; i < e->n_m; i++) qw_free(&e->m[i].qw);
  free(e->m);
  en_free(e->l);
  en_free(e->r);
  en_free(e->inner);
  en_free(e->dot);
  free(e->dot_hits.p);
  free(e->atom_pos);
  free(e->qpos_delta);
  free(e->states);
  free(e->px_prox);
  free(e->px_deltas);
  for (int i = 0; i < e->n_kids; i++) en_free(e->kids[i]);
  if (e->kid_hits)
    for (int i = 0; i < e->n_kids; i++) free(e->kid_hits[i].p);
  free(e->kid_hits);
  free(e->kids);
  free(e->kid_ok);
  free(e->myhits.p);
  free(e->tmp.p);
  free(e->tmp2.p);
  free(e);
}

/* ------------------------------------------------------------------------ */
/* tree construction: ExtNode_i::Create (searchnode.cpp:1599-1811)          */
/* ------------------------------------------------------------------------ */
typedef struct {
  const orc_index* idx;
  const orc_query* q;
  int use_bm25;
  int64_t fetched_docs, fetched_hits, skips;
  int error;
} build_ctx;

static enode* en_new(build_ctx* bc, int kind) {
  enode* e = (enode*)calloc(1, sizeof *e);
  e->kind = kind;
  e->rowid = ORC_INVALID_ROWID;
  e->p_fetched_docs = &bc->fetched_docs;
  e->p_fetched_hits = &bc->fetched_hits;
  return e;
}

static void mnode_init(build_ctx* bc, mnode* n, const orc_node* qn, int nodepos) {
  qw_setup(&n->qw, bc->idx, qn->term_id, &bc->skips);
  n->queried32 = qn->field_mask;
  n->wide_index = bc->idx->n_fields > 32;
  n->has_wide = 0;
  for (int i = 0; i < 7; i++) {
    n->queried_hi[i] = n->wide_index ? qn->field_mask_hi[i] : 0u;
    if (n->queried_hi[i]) n->has_wide = 1;
  }
  n->idf = 0.0f;
  n->atom_pos = qn->atom_pos;
  n->nodepos = nodepos;
  /* same key = same word; words missing from the dictionary are different words (the reference keys by the word itself) */
  n->word_key = qn->term_id >= 0 ? qn->term_id : -(1 + qn->atom_pos); /* (query positions are unique per keyword occurrence) */
  n->boost = qn->boost;
  n->rowid = ORC_INVALID_ROWID;
}

static enode* build_term(build_ctx* bc, const orc_node* qn) {
  enode* e = en_new(bc, EN_TERM);
  mnode_init(bc, &e->t, qn, 0);
  e->atom = qn->atom_pos;
  e->tp_kind = qn->term_pos; /* ExtNode_i::Create :1141-1166 */
  e->tp_max = qn->field_max_pos;
  return e;
}

/* sphSort's small-array path (sphinxstd.h:853-869): insertion sort that moves an
   element left past every element that is NOT less than it -- equal keys end up
   in REVERSE arrival order. All query-sized arrays (<=33) take this path. */
static void sph_isort_idx(int* idx, int n, const int* key) {
  for (int i = 1; i < n; i++)
    for (int j = i; j > 0; j--) {
      if (key[idx[j - 1]] < key[idx[j]]) break;
      int t = idx[j];
      idx[j] = idx[j - 1];
      idx[j - 1] = t;
    }
}

static enode* build_node(build_ctx* bc, int ni);

static enode* build_twofer(build_ctx* bc, int kind, enode* l, enode* r) {
  enode* e = en_new(bc, kind);
  e->l = l;
  e->r = r;
  e->atom = l->atom;
  return e;
}

/* CreateMultiNode<ExtPhrase_c> plain-node path (searchnode.cpp:984-1041) +
   ExtNWay_T ctor / ConstructNode (:3767-3802) + FSMphrase_c ctor (:3884-3899) */
static enode* build_phrase(build_ctx* bc, const orc_node* qn) {
  int k = qn->n_children;
  if (k < 2 || k > 32) {
    bc->error = 1;
    fail("phrase needs 2..32 keywords");
    return NULL;
  }
  enode* terms[32];
  int key[32], pos[32];
  for (int i = 0; i < k; i++) {
    const orc_node* c = &bc->q->nodes[bc->q->children[qn->first_child + i]];
    orc_node w = *c;
    w.field_mask = qn->field_mask & c->field_mask; /* words inherit the phrase node's field spec */
    for (int d = 0; d < 7; d++) w.field_mask_hi[d] = qn->field_mask_hi[d] & c->field_mask_hi[d];
    if (w.term_pos) {
      bc->error = 1;
      fail("position modifiers on the words of a phrase are not restated in the oracle");
      for (int j = 0; j < i; j++) en_free(terms[j]);
      return NULL;
    }
    terms[i] = build_term(bc, &w);
    key[i] = en_docs_count(terms[i]);
    pos[i] = i;
  }
  sph_isort_idx(pos, k, key); /* dPositions.Sort ( ExtNodeTFExt_fn ) */
  enode* e = en_new(bc, EN_PHRASE);
  e->n_atoms = k;
  e->atom_pos = (int*)malloc((size_t)k * sizeof(int));
  for (int i = 0; i < k; i++) e->atom_pos[i] = terms[i]->atom;
  e->n_qpos_delta = e->atom_pos[k - 1] - e->atom_pos[0] + 1;
  if (e->n_qpos_delta <= 0) {
    bc->error = 1;
    fail("phrase atom positions must ascend");
    for (int i = 0; i < k; i++) en_free(terms[i]);
    en_free(e);
    return NULL;
  }
  e->qpos_delta = (int*)malloc((size_t)e->n_qpos_delta * sizeof(int));
  for (int i = 0; i < e->n_qpos_delta; i++) e->qpos_delta[i] = -INT_MAX;
  for (int i = 1; i < k; i++)
    e->qpos_delta[terms[i - 1]->atom - terms[0]->atom] = terms[i]->atom - terms[i - 1]->atom;
  e->atom = terms[0]->atom; /* ExtNWay_c ctor :3718 */
  /* ConstructNode: left-deep ExtAnd chain in ascending doc-count order */
  uint16_t lpos = (uint16_t)pos[0];
  enode* cur = terms[lpos++];
  enode* cur_ex = NULL;
  for (int i = 1; i < k; i++) {
    uint16_t rpos = (uint16_t)pos[i];
    cur = cur_ex = build_twofer(bc, EN_AND, cur, terms[rpos++]);
    cur_ex->nodepos_l = lpos;
    cur_ex->nodepos_r = rpos;
    lpos = 0;
  }
  if (cur_ex) cur_ex->qpos_reverse = 1;
  e->inner = cur;
  if (qn->op == ORC_OP_PROXIMITY) { /* FSMproximity_c ctor :3958-3970 */
    if (qn->opt <= 0) {
      bc->error = 1;
      fail("proximity needs a positive distance");
      en_free(e);
      return NULL;
    }
    e->is_proximity = 1;
    e->max_distance = qn->opt;
    e->px_min_qpos = (uint32_t)e->atom_pos[0];
    e->px_qlen = (uint32_t)(e->atom_pos[k - 1] - e->atom_pos[0]);
    e->px_prox = (uint32_t*)malloc((size_t)(e->px_qlen + 1) * sizeof(uint32_t));
    e->px_deltas = (int*)malloc((size_t)(e->px_qlen + 1) * sizeof(int));
  }
  return e;
}

static enode* build_node(build_ctx* bc, int ni) {
  const orc_query* q = bc->q;
  const orc_node* qn = &q->nodes[ni];
  switch (qn->op) {
    case ORC_OP_TERM: return build_term(bc, qn);
    case ORC_OP_PHRASE:
    case ORC_OP_PROXIMITY: return build_phrase(bc, qn);
    case ORC_OP_QUORUM: {
      /* ExtNode_i::Create, SPH_QUERY_QUORUM (searchnode.cpp:1638-1686): a threshold of 1 becomes an ExtOr_c chain,
         a threshold >= the word count an ExtAnd_c chain, both over the words sorted by ascending doc count
         (ExtNodeTF_fn); only what lies between is a real ExtQuorum_c, which is not restated here */
      int k = qn->n_children;
      if (k < 2 || k > 32) {
        bc->error = 1;
        fail("quorum needs 2..32 keywords");
        return NULL;
      }
      const int thr = qn->opt;
      if (thr < 1) {
        bc->error = 1;
        fail("quorum threshold must be >= 1");
        return NULL;
      }
      enode* terms[32];
      int key[32], pos[32];
      for (int i = 0; i < k; i++) {
        const orc_node* c = &q->nodes[q->children[qn->first_child + i]];
        if (c->op != ORC_OP_TERM || c->term_pos) {
          bc->error = 1;
          fail("quorum over plain keywords only");
          for (int j = 0; j < i; j++) en_free(terms[j]);
          return NULL;
        }
        orc_node w = *c;
        w.field_mask = qn->field_mask & c->field_mask; /* Create ( word, pNode, .. ): the quorum node's field spec */
        for (int d = 0; d < 7; d++) w.field_mask_hi[d] = qn->field_mask_hi[d] & c->field_mask_hi[d];
        terms[i] = build_term(bc, &w);
        key[i] = en_docs_count(terms[i]);
        pos[i] = i;
      }
      if (thr != 1 && thr < k) { /* a real ExtQuorum_c: children stay in query-position order (QuorumNodeAtomPos_fn) */
        for (int i = 0; i < k; i++)
          for (int j = i + 1; j < k; j++)
            if (q->nodes[q->children[qn->first_child + i]].term_id >= 0 &&
                q->nodes[q->children[qn->first_child + i]].term_id == q->nodes[q->children[qn->first_child + j]].term_id) {
              bc->error = 1;
              fail("ExtQuorum_c with duplicate keywords (m_bHasDupes) not restated in the oracle");
              for (int t = 0; t < k; t++) en_free(terms[t]);
              return NULL;
            }
        enode* e = en_new(bc, EN_QUORUM);
        e->kids = (enode**)malloc((size_t)k * sizeof(enode*));
        e->kid_ok = (int*)calloc((size_t)k, sizeof(int));
        for (int i = 1; i < k; i++) /* insertion sort by atom position (stable; positions are distinct) */
          for (int j = i; j > 0 && terms[j - 1]->atom > terms[j]->atom; j--) {
            enode* t = terms[j];
            terms[j] = terms[j - 1];
            terms[j - 1] = t;
          }
        for (int i = 0; i < k; i++) e->kids[i] = terms[i], e->q_list[i] = i;
        e->n_kids = e->q_n = k;
        e->q_thresh = thr;
        e->atom = terms[0]->atom;
        return e;
      }
      sph_isort_idx(pos, k, key); /* dTerms.Sort ( ExtNodeTF_fn() ) */
      enode* cur = terms[pos[0]];
      for (int i = 1; i < k; i++) cur = build_twofer(bc, thr == 1 ? EN_OR : EN_AND, cur, terms[pos[i]]);
      return cur;
    }
    case ORC_OP_BEFORE: { /* CreateOrderNode (searchnode.cpp:1044-1073): children as they come, in query order */
      int k = qn->n_children;
      if (k < 2 || k > 32) {
        bc->error = 1;
        fail("order node requires 2..32 children");
        return NULL;
      }
      enode* e = en_new(bc, EN_ORDER);
      e->kids = (enode**)calloc((size_t)k, sizeof(enode*));
      e->kid_hits = (hitvec*)calloc((size_t)k, sizeof(hitvec));
      for (int i = 0; i < k; i++) {
        e->kids[i] = build_node(bc, q->children[qn->first_child + i]);
        e->n_kids = i + 1;
        if (!e->kids[i]) {
          if (!bc->error) {
            bc->error = 1;
            fail("order node: a child could not be created");
          }
          e->n_kids = i;
          en_free(e);
          return NULL;
        }
      }
      e->atom = e->kids[0]->atom;
      return e;
    }
    case ORC_OP_AND: {
      int k = qn->n_children;
      if (k < 1 || k > 32) {
        bc->error = 1;
        fail("AND needs 1..32 children");
        return NULL;
      }
      int all_terms = 1;
      for (int i = 0; i < k; i++)
        if (q->nodes[q->children[qn->first_child + i]].op != ORC_OP_TERM) all_terms = 0;
      int any_termpos = 0;
      for (int i = 0; i < k; i++)
        if (q->nodes[q->children[qn->first_child + i]].term_pos) any_termpos = 1;
      if (all_terms && k > 1 && any_termpos) {
        /* a keyword with a position modifier rules the multi-and node out (:1724-1735): terms sorted by ExtNodeTF_fn -- the
           modified ones count INT_MAX docs -- and chained with ExtAnd_c (:1743-1762) */
        enode* terms[32];
        int key[32], pos[32];
        for (int i = 0; i < k; i++) {
          terms[i] = build_term(bc, &q->nodes[q->children[qn->first_child + i]]);
          key[i] = en_docs_count(terms[i]);
          pos[i] = i;
        }
        sph_isort_idx(pos, k, key);
        enode* cur = terms[pos[0]];
        for (int i = 1; i < k; i++) cur = build_twofer(bc, EN_AND, cur, terms[pos[i]]);
        return cur;
      }
      if (all_terms && k > 1) {
        /* CreateMultiAndNode (:1118-1138, 1774) + ExtMultiAnd_T ctor (:2772-2798) */
        enode* e = en_new(bc, EN_MULTIAND);
        mnode* tmp = (mnode*)calloc((size_t)k, sizeof(mnode));
        int key[32], pos[32];
        int test_fields = 0;
        for (int i = 0; i < k; i++) {
          const orc_node* c = &q->nodes[q->children[qn->first_child + i]];
          mnode_init(bc, &tmp[i], c, i);
          key[i] = tmp[i].qw.docs;
          pos[i] = i;
          if (c->field_mask != 0xFFFFFFFFu) test_fields = 1; /* TEST_FIELDS = !m_dFieldMask.TestAll ( true ) */
          if (bc->idx->n_fields > 32)
            for (int d = 0; d < 7; d++)
              if (c->field_mask_hi[d] != 0xFFFFFFFFu) test_fields = 1;
        }
        sph_isort_idx(pos, k, key); /* m_dNodes.Sort ( SelectivitySorter_t ) :2791 */
        e->m = (mnode*)calloc((size_t)k, sizeof(mnode));
        for (int i = 0; i < k; i++) e->m[i] = tmp[pos[i]];
        free(tmp);
        e->n_m = k;
        e->test_fields = test_fields;
        e->atom = e->m[0].atom_pos;
        return e;
      }
      /* non-multi path :1743-1769 / generic create :1785-1806 */
      enode* kids[32];
      int nk = 0;
      for (int i = 0; i < k; i++) {
        enode* c = build_node(bc, q->children[qn->first_child + i]);
        if (c) kids[nk++] = c;
      }
      if (!nk) return NULL;
      if (all_terms) { /* only reachable with k==1 */
        return kids[0];
      }
      enode* cur = kids[0];
      for (int i = 1; i < nk; i++) cur = build_twofer(bc, EN_AND, cur, kids[i]);
      return cur;
    }
    case ORC_OP_OR:
    case ORC_OP_MAYBE:
    case ORC_OP_ANDNOT: {
      int kind = qn->op == ORC_OP_OR ? EN_OR : qn->op == ORC_OP_MAYBE ? EN_MAYBE : EN_ANDNOT;
      enode* cur = NULL;
      for (int i = 0; i < qn->n_children; i++) {
        enode* nx = build_node(bc, q->children[qn->first_child + i]);
        if (!nx) continue;
        cur = cur ? build_twofer(bc, kind, cur, nx) : nx;
      }
      return cur;
    }
    case ORC_OP_NEAR: {
      /* CreateMultiNode<ExtMultinear_c>, non-plain path (searchnode.cpp:932-972): the operands are nodes of any kind;
         ExtNWay_T ctor + ConstructNode (:3767-3802) chain them left-deep in ascending doc-count order and number them by
         their place in the query (SetNodePos); FSMmultinear_c ctor :4080-4094 */
      int k = qn->n_children;
      if (k < 2 || k > 32 || qn->opt <= 0) {
        bc->error = 1;
        fail("NEAR needs 2..32 operands and a positive distance");
        return NULL;
      }
      enode* kids[32];
      int key[32], pos[32];
      for (int i = 0; i < k; i++) { /* keywords, phrases and nested NEARs only: see the note below */
        const int cop = q->nodes[q->children[qn->first_child + i]].op;
        if (cop != ORC_OP_TERM && cop != ORC_OP_PHRASE && cop != ORC_OP_PROXIMITY && cop != ORC_OP_NEAR) {
          /* test/test_115 ('(a b c) NEAR/3 d', 'burden NEAR/2 (financial share)') shows the reference answering AND / OR
             group operands like a NEAR over all their keywords; reading ExtNode_i::Create does not explain that (a twofer
             over the group's merged hits gives other docs and weights), so such operands are declined, not guessed at */
          bc->error = 1;
          fail("NEAR over AND / OR groups is not restated in the oracle");
          return NULL; /* (no operand has been built yet) */
        }
      }
      for (int i = 0; i < k; i++) {
        kids[i] = build_node(bc, q->children[qn->first_child + i]);
        if (!kids[i]) {
          if (!bc->error) {
            bc->error = 1;
            fail("NEAR operand without a node");
          }
          for (int j = 0; j < i; j++) en_free(kids[j]);
          return NULL;
        }
        key[i] = en_docs_count(kids[i]);
        pos[i] = i;
      }
      sph_isort_idx(pos, k, key); /* dPositions.Sort ( ExtNodeTFExt_fn ) */
      enode* e = en_new(bc, EN_PHRASE);
      e->is_near = 1;
      e->near_dist = qn->opt;
      e->n_atoms = k; /* m_uWordsExpected */
      e->nr_first_qpos = 65535;
      e->atom = kids[0]->atom; /* ExtNWay_c ctor :3718 */
      uint16_t lpos = (uint16_t)pos[0];
      enode* cur = kids[lpos++];
      enode* cur_ex = NULL;
      for (int i = 1; i < k; i++) {
        uint16_t rpos = (uint16_t)pos[i];
        cur = cur_ex = build_twofer(bc, EN_AND, cur, kids[rpos++]);
        cur_ex->nodepos_l = lpos;
        cur_ex->nodepos_r = rpos;
        lpos = 0;
      }
      if (cur_ex) cur_ex->qpos_reverse = 1;
      e->inner = cur;
      return e;
    }
    case ORC_OP_SENTENCE:
    case ORC_OP_PARAGRAPH: { /* generic create :1785-1803: pCur = new ExtUnit_c ( pCur, pNext, fields, setup, MAGIC_WORD_... ); its boundary term is
                                created with the node's field mask and bNotWeighted (:4987-4989): not a query word, no IDF */
      enode* cur = NULL;
      for (int i = 0; i < qn->n_children; i++) {
        enode* nx = build_node(bc, q->children[qn->first_child + i]);
        if (!nx) continue;
        if (!cur) {
          cur = nx;
          continue;
        }
        cur = build_twofer(bc, EN_UNIT, cur, nx);
        if (qn->term_id >= 0) {
          orc_node w;
          memset(&w, 0, sizeof w);
          w.op = ORC_OP_TERM, w.term_id = qn->term_id, w.field_mask = qn->field_mask, w.boost = 1.0f;
          memcpy(w.field_mask_hi, qn->field_mask_hi, sizeof w.field_mask_hi);
          cur->dot = build_term(bc, &w);
        }
      }
      return cur;
    }
    case ORC_OP_NOTNEAR: { /* generic create :1785-1803: pCur = new ExtNotNear_c ( pCur, pNext, .., m_iOpArg ) */
      if (qn->opt <= 0) {
        bc->error = 1;
        fail("NOTNEAR needs a positive distance");
        return NULL;
      }
      enode* cur = NULL;
      for (int i = 0; i < qn->n_children; i++) {
        enode* nx = build_node(bc, q->children[qn->first_child + i]);
        if (!nx) continue;
        if (!cur)
          cur = nx;
        else {
          cur = build_twofer(bc, EN_NOTNEAR, cur, nx);
          cur->nn_dist = qn->opt;
        }
      }
      return cur;
    }
    default:
      bc->error = 1;
      fail("operator not restated in the oracle");
      return NULL;
  }
}

/* ------------------------------------------------------------------------ */
/* IDF: sphCreateRanker (sphinxsearch.cpp:4294-4378)                         */
/* ------------------------------------------------------------------------ */
float orc_idf(int64_t term_docs, int64_t total_docs, int plain_idf, int normalized, int n_qwords, float boost) {
  float idf = 0.0f;
  if (term_docs) {
    const int64_t total_clamped = total_docs > term_docs ? total_docs : term_docs;
    if (!plain_idf) {
      float log_total = logf((float)(1 + total_clamped));
      idf = logf((float)(total_clamped - term_docs + 1) / (float)term_docs) / (2 * log_total);
    } else {
      float log_total = logf((float)(1 + total_clamped));
      idf = logf((float)total_clamped / (float)term_docs) / (2 * log_total);
    }
  }
  if (normalized) idf /= n_qwords;
  return idf * boost;
}

typedef struct {
  int key;
  int docs;
  float boost;
  float idf;
  int64_t local_docs;
} qw_info;

typedef struct {
  qw_info w[64];
  int n;
} qw_hash;

/* GetQwords traversal (ExtTerm_T::GetQwords :2029-2055, ExtMultiAnd_T::GetQword
   :3248-3272): the FIRST node seen for a word is marked (idf=-1) and later
   receives the word's IDF; later duplicates keep idf = 0 */
static void collect_mnode(mnode* n, qw_hash* h, int* dupes) {
  n->idf = 0.0f;
  for (int i = 0; i < h->n; i++)
    if (h->w[i].key == n->word_key) {
      *dupes = 1;
      return;
    }
  n->idf = -1.0f;
  if (h->n < 64) {
    h->w[h->n].key = n->word_key;
    h->w[h->n].docs = n->qw.docs;
    h->w[h->n].boost = n->boost;
    h->w[h->n].local_docs = -1;
    h->n++;
  }
}
static void collect_qwords(enode* e, qw_hash* h, int* dupes) {
  if (!e) return;
  switch (e->kind) {
    case EN_TERM: collect_mnode(&e->t, h, dupes); break;
    case EN_MULTIAND:
      for (int i = 0; i < e->n_m; i++) collect_mnode(&e->m[i], h, dupes);
      break;
    case EN_PHRASE: collect_qwords(e->inner, h, dupes); break;
    case EN_QUORUM:
    case EN_ORDER:
      for (int i = 0; i < e->n_kids; i++) collect_qwords(e->kids[i], h, dupes);
      break;
    default:
      collect_qwords(e->l, h, dupes);
      collect_qwords(e->r, h, dupes);
      break;
  }
}
static void set_mnode_idf(mnode* n, const qw_hash* h) {
  if (n->idf < 0.0f)
    for (int i = 0; i < h->n; i++)
      if (h->w[i].key == n->word_key) n->idf = h->w[i].idf;
}
static void set_idf(enode* e, const qw_hash* h) {
  if (!e) return;
  switch (e->kind) {
    case EN_TERM: set_mnode_idf(&e->t, h); break;
    case EN_MULTIAND:
      for (int i = 0; i < e->n_m; i++) set_mnode_idf(&e->m[i], h);
      break;
    case EN_PHRASE: set_idf(e->inner, h); break;
    case EN_QUORUM:
    case EN_ORDER:
      for (int i = 0; i < e->n_kids; i++) set_idf(e->kids[i], h);
      break;
    default:
      set_idf(e->l, h);
      set_idf(e->r, h);
      break;
  }
}

/* ------------------------------------------------------------------------ */
/* top-K: CSphMatchQueue<MatchRelevanceLt_fn> (sphinxsort.cpp:583-812, 4534) */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint32_t rowid;
  int32_t weight;
} match_t;

static inline int match_less(const match_t* a, const match_t* b) { /* :4541-4547 */
  if (a->weight != b->weight) return a->weight < b->weight;
  return a->rowid > b->rowid;
}

typedef struct {
  match_t* data;
  int used, size;
  int64_t total;
} mqueue;

/* heap order: InvCompareIndex_fn::IsLess(a,b) = COMP::IsLess(b,a) (:560-574); root = worst */
static inline int inv_less(const match_t* a, const match_t* b) { return match_less(b, a); }

static void mq_pop(mqueue* q) { /* PopAndProcess_T :765-811 */
  q->used--;
  if (q->used) q->data[0] = q->data[q->used];
  int entry = 0;
  for (;;) {
    int child = entry * 2 + 1;
    if (child >= q->used) break;
    if (child + 1 < q->used && inv_less(&q->data[child], &q->data[child + 1])) ++child;
    if (inv_less(&q->data[entry], &q->data[child])) {
      match_t t = q->data[child];
      q->data[child] = q->data[entry];
      q->data[entry] = t;
      entry = child;
      continue;
    }
    break;
  }
}

static int mq_push(mqueue* q, const match_t* m) { /* PushT :722-761 */
  ++q->total;
  if (q->used == q->size) {
    if (match_less(m, &q->data[0])) return 1;
    mq_pop(q);
  }
  q->data[q->used++] = *m;
  int entry = q->used - 1;
  while (entry) {
    int parent = (entry - 1) / 2;
    if (!inv_less(&q->data[parent], &q->data[entry])) break;
    match_t t = q->data[entry];
    q->data[entry] = q->data[parent];
    q->data[parent] = t;
    entry = parent;
  }
  return 1;
}

/* ------------------------------------------------------------------------ */
/* rankers                                                                  */
/* ------------------------------------------------------------------------ */
/* RankerState_Proximity_fn<USE_BM25,false> (sphinxsearch.cpp:1320-1438) */
typedef struct {
  uint8_t lcs[ORC_MAX_FIELDS];
  uint8_t cur_lcs;
  int exp_delta;
  int last_hitpos_with_field;
  /* HANDLE_DUPES */
  uint32_t lcs_tail_pos, lcs_tail_qpos_mask, cur_qpos_mask, cur_pos;
} prox_state;

static void prox_init(prox_state* s) {
  memset(s->lcs, 0, sizeof s->lcs);
  s->cur_lcs = 0;
  s->exp_delta = -INT_MAX;
  s->last_hitpos_with_field = -INT_MAX;
  s->lcs_tail_pos = s->lcs_tail_qpos_mask = s->cur_qpos_mask = s->cur_pos = 0;
}

/* RankerState_Proximity_fn<USE_BM25,true>::Update (sphinxsearch.cpp:1370-1412): query keywords repeat, several query
   positions may share a hit position */
static inline void prox_update_dupes(prox_state* s, const hit_t* h) {
  uint32_t pos = ORC_HIT_POSWITHFIELD(h->hitpos);
  uint32_t field = ORC_HIT_FIELD(h->hitpos);
  if (ORC_HIT_FIELD(s->cur_pos) != field) s->cur_qpos_mask = 0; /* reset accumulated data from the previous field */
  if (pos != s->cur_pos) {
    if (s->cur_lcs < 2) {
      s->lcs_tail_pos = s->cur_pos;
      s->lcs_tail_qpos_mask = s->cur_qpos_mask;
      s->cur_lcs = 1;
    }
    s->cur_qpos_mask = 0;
    s->cur_pos = pos;
    if (s->lcs[field] < h->weight) s->lcs[field] = (uint8_t)h->weight;
  }
  /* the reference shifts 1UL (64 bits on LP64) and stores into a DWORD: query positions 32..63 add no bit */
  s->cur_qpos_mask |= (uint32_t)(1ull << (h->qpos & 63));
  int delta = (int)(s->cur_pos - s->lcs_tail_pos);
  /* (a negative delta -- positions running backwards -- is a negative shift count in the reference, undefined in C++;
     x86 masks the count to 5 bits, and so do we) */
  if (delta && delta < 32 && ((s->cur_qpos_mask >> (delta & 31)) & s->lcs_tail_qpos_mask)) {
    s->lcs_tail_qpos_mask = (uint32_t)(1ull << (h->qpos & 63));
    s->lcs_tail_pos = s->cur_pos;
    s->cur_lcs = (uint8_t)(s->cur_lcs + h->weight);
    s->cur_qpos_mask = 0;
    if (s->cur_lcs > s->lcs[field]) s->lcs[field] = s->cur_lcs;
  }
}

static inline void prox_update(prox_state* s, const hit_t* h) {
  const int pos_with_field = (int)ORC_HIT_POSWITHFIELD(h->hitpos);
  int delta = pos_with_field - h->qpos;
  if (pos_with_field > s->last_hitpos_with_field)
    s->cur_lcs = (uint8_t)(((delta == s->exp_delta) ? s->cur_lcs : 0) + (uint8_t)h->weight);
  uint32_t field = ORC_HIT_FIELD(h->hitpos);
  if (s->cur_lcs > s->lcs[field]) s->lcs[field] = s->cur_lcs;
  s->last_hitpos_with_field = pos_with_field;
  s->exp_delta = delta + h->spanlen - 1;
}

static inline int prox_finalize(prox_state* s, int n_fields, const int32_t* weights, int use_bm25, int bm25) {
  s->cur_lcs = 0;
  s->exp_delta = -1;
  s->last_hitpos_with_field = -1;
  s->lcs_tail_pos = s->lcs_tail_qpos_mask = s->cur_qpos_mask = s->cur_pos = 0; /* if_const ( HANDLE_DUPES ) */
  int rank = 0;
  for (int i = 0; i < n_fields; i++) {
    rank += (int)(s->lcs[i]) * weights[i];
    s->lcs[i] = 0;
  }
  return use_bm25 ? (int)((uint32_t)bm25 + (uint32_t)rank * SPH_BM25_SCALE) : rank;
}

/* RankerState_ProximityBM25Exact_fn = SPH04 (sphinxsearch.cpp:1443-1530).  min_exp_pos is NOT reset by
   Finalize in the reference; it is kept here the same way (it cannot change a doc's first hit: that one
   either fails the delta test or takes the same values from both branches). */
typedef struct {
  uint8_t lcs[ORC_MAX_FIELDS];
  uint8_t cur_lcs;
  int exp_delta;
  int last_hitpos;
  uint32_t min_exp_pos;
  uint32_t head_hit, exact_hit;
  int max_qpos;
} sph04_state;

/* TagExcluded (sphinx.cpp:15107-15129): ex[node] = 1 for keywords the query excludes */
static void tag_excluded(const orc_query* q, int ni, int neg, unsigned char* ex, int depth) {
  if (ni < 0 || ni >= q->n_nodes || depth > 32) return;
  const orc_node* n = &q->nodes[ni];
  if (n->op == ORC_OP_TERM) {
    ex[ni] = (unsigned char)neg;
    return;
  }
  for (int i = 0; i < n->n_children; i++)
    tag_excluded(q, q->children[n->first_child + i], (n->op == ORC_OP_ANDNOT && i == 1) ? !neg : neg, ex, depth + 1);
}

static void sph04_init(sph04_state* s, int max_qpos) {
  memset(s, 0, sizeof *s);
  s->exp_delta = -INT_MAX;
  s->last_hitpos = -1;
  s->max_qpos = max_qpos;
}

static inline void sph04_update(sph04_state* s, const hit_t* h) {
  uint32_t field = ORC_HIT_FIELD(h->hitpos);
  int pos_with_field = (int)ORC_HIT_POSWITHFIELD(h->hitpos);
  int delta = pos_with_field - h->qpos;
  const int is_end = (h->hitpos >> 23) & 1;
  const int pos = (int)(h->hitpos & 0x7FFFFFu);
  if (delta == s->exp_delta && ORC_HIT_POSWITHFIELD(h->hitpos) >= s->min_exp_pos) {
    if (pos_with_field > s->last_hitpos) s->cur_lcs = (uint8_t)(s->cur_lcs + h->weight);
    if (is_end && (int)h->qpos == s->max_qpos && pos == s->max_qpos) s->exact_hit |= 1u << field;
  } else {
    if (pos_with_field > s->last_hitpos) s->cur_lcs = (uint8_t)h->weight;
    if (pos == 1) {
      s->head_hit |= 1u << field;
      if (is_end && s->max_qpos == 1) s->exact_hit |= 1u << field;
    }
  }
  if (s->cur_lcs > s->lcs[field]) s->lcs[field] = s->cur_lcs;
  s->exp_delta = delta + h->spanlen - 1;
  s->last_hitpos = pos_with_field;
  s->min_exp_pos = ORC_HIT_POSWITHFIELD(h->hitpos) + 1;
}

static inline int sph04_finalize(sph04_state* s, int n_fields, const int32_t* weights, int bm25) {
  s->cur_lcs = 0;
  s->exp_delta = -1;
  s->last_hitpos = -1;
  int rank = 0;
  for (int i = 0; i < n_fields; i++) {
    rank += (int)(4 * s->lcs[i] + 2 * ((s->head_hit >> i) & 1) + ((s->exact_hit >> i) & 1)) * weights[i];
    s->lcs[i] = 0;
  }
  s->head_hit = 0;
  s->exact_hit = 0;
  return (int)((uint32_t)bm25 + (uint32_t)rank * SPH_BM25_SCALE);
}

/* RankerState_MatchAny_fn (sphinxsearch.cpp:1577-1616): proximity state + per-field mask of matched query positions */
typedef struct {
  prox_state p;
  int phrase_k;
  uint8_t match_mask[ORC_MAX_FIELDS];
} matchany_state;

static void matchany_init(matchany_state* s, int n_fields, const int32_t* weights, int n_qwords) {
  prox_init(&s->p);
  s->phrase_k = 0;
  for (int i = 0; i < n_fields; i++) s->phrase_k += weights[i] * n_qwords;
  memset(s->match_mask, 0, sizeof s->match_mask);
}

static inline void matchany_update(matchany_state* s, const hit_t* h) {
  prox_update(&s->p, h);
  s->match_mask[ORC_HIT_FIELD(h->hitpos)] |= (uint8_t)(1 << (h->qpos - 1)); /* BYTE: query positions past 8 fall off */
}

static inline int matchany_finalize(matchany_state* s, int n_fields, const int32_t* weights) {
  s->p.cur_lcs = 0;
  s->p.exp_delta = -1;
  s->p.last_hitpos_with_field = -1;
  int rank = 0;
  for (int i = 0; i < n_fields; i++) {
    if (s->match_mask[i]) rank += (int)(__builtin_popcount(s->match_mask[i]) + (s->p.lcs[i] - 1) * s->phrase_k) * weights[i];
    s->match_mask[i] = 0;
    s->p.lcs[i] = 0;
  }
  return rank;
}

/* ------------------------------------------------------------------------ */
/* search = sphCreateRanker + MatchExtended + sorter                         */
/* ------------------------------------------------------------------------ */
/* ISphFilter::Eval: sphGetRowAttr (sphinx.h:993-1014), IFilter_Values::EvalValues (sphinxfilter.cpp:69-91), EvalRange
   (sphinxfilter.h:130-143), FilterNot for m_bExclude, Filter_And over the list */
static int filters_pass(const orc_index* idx, const orc_query* q, uint32_t rowid) {
  const uint32_t* row = idx->attrs + (size_t)rowid * (size_t)idx->attr_stride;
  for (int i = 0; i < q->n_filters; i++) {
    const orc_filter* f = &q->filters[i];
    if (f->mva_bits) { /* Filter_MVAValues_Any_c / _All_c / Filter_MVARange_Any_c / _All_c (sphinxfilter.cpp:340-383) */
      int pass = 0;
      if (idx->blobs) {
        const uint8_t* br = idx->blobs + ((uint64_t)row[2] | ((uint64_t)row[3] << 32)); /* sphGetBlobRowOffset */
        const int sz = br[0] == 0 ? 1 : br[0] == 1 ? 2 : 4;                              /* GetBlobAttr, attribute.cpp:495-513 */
        uint64_t l1 = 0, l0 = 0;
        memcpy(&l1, br + 1 + f->blob_attr_id * sz, (size_t)sz);
        if (f->blob_attr_id) memcpy(&l0, br + 1 + (f->blob_attr_id - 1) * sz, (size_t)sz);
        const uint8_t* data = br + 1 + f->n_blob_attrs * sz + l0;
        const int w = f->mva_bits / 8, nv = (int)((l1 - l0) / (uint64_t)w);
#define MVA_AT(i_) (w == 4 ? (int64_t)({ uint32_t t_; memcpy(&t_, data + 4 * (size_t)(i_), 4); t_; }) : ({ int64_t t_; memcpy(&t_, data + 8 * (size_t)(i_), 8); t_; }))
        if (nv > 0) {
          if (f->kind == ORC_FILTER_VALUES && !f->mva_all) { /* MvaEval_Any (sphinxfilter.h:160-184) */
            for (int a = 0; a < nv && !pass; a++)
              for (int k = 0; k < f->n_values; k++)
                if (MVA_AT(a) == f->values[k]) pass = 1;
          } else if (f->kind == ORC_FILTER_VALUES) { /* MvaEval_All (:187-201) */
            pass = 1;
            for (int a = 0; a < nv && pass; a++) {
              int in = 0;
              for (int k = 0; k < f->n_values; k++)
                if (MVA_AT(a) == f->values[k]) in = 1;
              pass = in;
            }
          } else if (!f->mva_all) { /* MvaEval_RangeAny (:203-233) */
            int L = 0, R = nv - 1, decided = 0;
            while (L <= R) {
              const int mid = L + (R - L) / 2;
              const int64_t x = MVA_AT(mid);
              if (f->min_value > x)
                L = mid + 1;
              else if (f->min_value < x)
                R = mid - 1;
              else {
                pass = f->has_equal_min || mid + 1 < nv;
                decided = 1;
                break;
              }
            }
            if (!decided && L != nv) {
              const int64_t x = MVA_AT(L);
              pass = f->has_equal_max ? x <= f->max_value : x < f->max_value;
            }
          } else { /* MvaEval_RangeAll (:244-253): ( *L, *R, (T)m_iMinValue, (T)m_iMaxValue ) */
            const int64_t a = MVA_AT(0), b = MVA_AT(nv - 1);
            const int64_t lo = w == 4 ? (int64_t)(uint32_t)f->min_value : f->min_value, hi = w == 4 ? (int64_t)(uint32_t)f->max_value : f->max_value;
            pass = (f->has_equal_min ? a >= lo : a > lo) && (f->has_equal_max ? b <= hi : b < hi);
          }
        }
#undef MVA_AT
      }
      if (f->exclude) pass = !pass;
      if (!pass) return 0;
      continue;
    }
    const int item = f->bit_offset >> 5;
    int64_t v;
    if (f->bit_count == 32)
      v = (int64_t)row[item];
    else if (f->bit_count == 64)
      v = (int64_t)((uint64_t)row[item] | ((uint64_t)row[item + 1] << 32));
    else
      v = (int64_t)((row[item] >> (f->bit_offset & 31)) & ((1ul << f->bit_count) - 1));
    int pass;
    if (f->kind == ORC_FILTER_VALUES) {
      pass = 0;
      int lo = 0, hi = f->n_values - 1;
      while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (f->values[mid] == v) {
          pass = 1;
          break;
        }
        if (f->values[mid] < v)
          lo = mid + 1;
        else
          hi = mid - 1;
      }
    } else if (f->kind == ORC_FILTER_FLOATRANGE) { /* Filter_FloatRange::Eval (sphinxfilter.cpp:286-289): sphDW2F of the dword, both bounds */
      float fv;
      const uint32_t bits = (uint32_t)v;
      memcpy(&fv, &bits, 4);
      pass = (f->has_equal_min ? fv >= f->fmin : fv > f->fmin) && (f->has_equal_max ? fv <= f->fmax : fv < f->fmax);
    } else {
      const int min_ok = f->has_equal_min ? v >= f->min_value : v > f->min_value;
      const int max_ok = f->has_equal_max ? v <= f->max_value : v < f->max_value;
      pass = f->open_left ? max_ok : f->open_right ? min_ok : (min_ok && max_ok);
    }
    if (f->exclude) pass = !pass;
    if (!pass) return 0;
  }
  return 1;
}

int orc_search(const orc_index* idx, const orc_query* q, orc_result* res) {
  g_err[0] = 0;
  res->n = 0;
  res->total_found = 0;
  res->fetched_docs = res->fetched_hits = res->skips = 0;
  if (q->max_matches <= 0) return fail("max_matches must be > 0");
  if (idx->n_fields > ORC_MAX_FIELDS) return fail("more fields than SPH_MAX_FIELDS");

  int ranker = q->ranker;
  const orc_node* rootq = &q->nodes[q->root];
  const int single_word = (rootq->op == ORC_OP_TERM); /* XQQuery_t::m_bSingleWord */
  int use_bm25, state_ranker;
  switch (ranker) {
    case ORC_RANK_PROXIMITY_BM25: use_bm25 = 1; state_ranker = !single_word; break; /* :4192-4201 */
    case ORC_RANK_BM25: use_bm25 = 1; state_ranker = 0; break;
    case ORC_RANK_NONE: use_bm25 = 0; state_ranker = 0; break;
    case ORC_RANK_PROXIMITY: use_bm25 = 0; state_ranker = !single_word; break;
    /* always ExtRanker_State_T, single keyword or not (:4214-4236) */
    case ORC_RANK_WORDCOUNT:
    case ORC_RANK_MATCHANY:
    case ORC_RANK_FIELDMASK: use_bm25 = 0; state_ranker = 1; break;
    case ORC_RANK_SPH04: use_bm25 = 1; state_ranker = 1; break;
    default: return fail("ranker not restated in the oracle");
  }

  build_ctx bc;
  memset(&bc, 0, sizeof bc);
  bc.idx = idx;
  bc.q = q;
  bc.use_bm25 = use_bm25;
  enode* root = build_node(&bc, q->root);
  if (bc.error) {
    en_free(root);
    return -1;
  }

  /* IDFs */
  qw_hash h;
  h.n = 0;
  int dupes = 0;
  collect_qwords(root, &h, &dupes);
  const int handle_dupes = dupes != 0; /* HasQwordDupes -> RankerState_Proximity_fn<.., true> (:4178, 4197, 4218) */
  int64_t total_docs = q->total_docs_override > 0 ? q->total_docs_override : idx->total_docs;
  for (int i = 0; i < h.n; i++) {
    int64_t term_docs = h.w[i].docs;
    if (q->local_docs) { /* local_df override, :4310-4315; keyed by node */
      for (int k = 0; k < q->n_nodes; k++)
        if (q->nodes[k].op == ORC_OP_TERM && q->nodes[k].term_id == h.w[i].key && q->local_docs[k] >= 0) {
          term_docs = q->local_docs[k];
          break;
        }
    }
    h.w[i].idf = orc_idf(term_docs, total_docs, q->plain_idf, q->normalized_tfidf, h.n, h.w[i].boost);
  }
  set_idf(root, &h);

  /* field weights: BindWeights (sphinx.cpp:13903-13943): default 1 per field */
  int32_t weights[ORC_MAX_FIELDS];
  int n_weights = idx->n_fields;
  for (int i = 0; i < n_weights; i++) weights[i] = (q->field_weights && i < q->n_weights) ? q->field_weights[i] : 1;
  const int ws_weights = n_weights < 32 ? n_weights : 32;

  mqueue mq;
  mq.size = q->max_matches;
  mq.used = 0;
  mq.total = 0;
  mq.data = (match_t*)malloc((size_t)mq.size * sizeof(match_t));

  hitvec hv = {0, 0, 0};
  prox_state ps;
  prox_init(&ps);
  /* m_iMaxQpos = GetQwords() (max query position over all keywords), m_iQwords = distinct words (:4294-4296, 730-731) */
  /* ... over the keywords that are not EXCLUDED: TagExcluded (sphinx.cpp:15107-15129) marks the words on the right of an ANDNOT
     (toggling with every nesting) and an excluded word's GetQwords() answers -1 (searchnode.cpp:2039, 2053) */
  int max_qpos = 0;
  {
    unsigned char* ex = (unsigned char*)calloc((size_t)(q->n_nodes > 0 ? q->n_nodes : 1), 1);
    tag_excluded(q, q->root, 0, ex, 0);
    for (int k = 0; k < q->n_nodes; k++)
      if (q->nodes[k].op == ORC_OP_TERM && !ex[k] && q->nodes[k].atom_pos > max_qpos) max_qpos = q->nodes[k].atom_pos;
    free(ex);
  }
  sph04_state s4;
  sph04_init(&s4, max_qpos);
  matchany_state ms;
  matchany_init(&ms, n_weights, weights, h.n);
  int cutoff = q->cutoff > 0 ? q->cutoff : -1;
  const int index_weight = q->index_weight ? q->index_weight : 1;

  while (root && en_next(root)) {
    int weight;
    /* GetFilteredDocs (sphinxsearch.cpp:1034-1091): no filters => every doc passes EarlyReject */
    int bm25 = 0;
    if (use_bm25) bm25 = (int)((root->tfidf + 0.5f) * SPH_BM25_SCALE);
    if (state_ranker) {
      /* ExtRanker_State_T::GetMatches (:1198-1315): a doc without hits is never flushed */
      hv.n = 0;
      en_hits(root, &hv);
      if (!hv.n) continue;
      switch (ranker) {
        case ORC_RANK_WORDCOUNT: /* RankerState_Wordcount_fn :1620-1643 */
          weight = 0;
          for (int i = 0; i < hv.n; i++) weight += weights[ORC_HIT_FIELD(hv.p[i].hitpos)];
          break;
        case ORC_RANK_FIELDMASK: { /* RankerState_Fieldmask_fn :1647-1668 */
          uint32_t mask = 0;
          for (int i = 0; i < hv.n; i++) mask |= 1u << ORC_HIT_FIELD(hv.p[i].hitpos);
          weight = (int)mask;
          break;
        }
        case ORC_RANK_MATCHANY:
          for (int i = 0; i < hv.n; i++) matchany_update(&ms, &hv.p[i]);
          weight = matchany_finalize(&ms, n_weights, weights);
          break;
        case ORC_RANK_SPH04:
          for (int i = 0; i < hv.n; i++) sph04_update(&s4, &hv.p[i]);
          weight = sph04_finalize(&s4, n_weights, weights, bm25);
          break;
        default:
          if (handle_dupes)
            for (int i = 0; i < hv.n; i++) prox_update_dupes(&ps, &hv.p[i]);
          else
            for (int i = 0; i < hv.n; i++) prox_update(&ps, &hv.p[i]);
          weight = prox_finalize(&ps, n_weights, weights, use_bm25, bm25);
      }
    } else if (ranker == ORC_RANK_NONE) {
      weight = 1; /* ExtRanker_None_c :1145-1169 */
    } else {
      /* ExtRanker_WeightSum_c :1097-1141 */
      uint32_t rank = 0;
      uint32_t mask = root->fields;
      if (!mask)
        rank = 1;
      else
        for (int i = 0; i < ws_weights; i++)
          if (mask & (1u << i)) rank += (uint32_t)weights[i];
      weight = use_bm25 ? (int)((uint32_t)bm25 + rank * SPH_BM25_SCALE) : (int)rank;
    }
    /* EarlyReject (ExtRanker_c::GetMatches, sphinxsearch.cpp:1055-1064; CSphIndex_VLN::EarlyReject, sphinx.cpp:11903-11917):
       the reference drops filtered rows before it ranks them; dropping them here gives the same matches */
    if (q->n_filters > 0 && !filters_pass(idx, q, root->rowid)) continue;
    /* MatchExtended (sphinx.cpp:12211-12263): dead rows never reach the sorter (:12213-12217) */
    if (idx->dead_rows && (idx->dead_rows[root->rowid >> 5] >> (root->rowid & 31u)) & 1u) continue;
    weight = (int)((uint32_t)weight * (uint32_t)index_weight);
    { /* m_pWeightFilter (sphinx.cpp:12223-12227): Filter_WeightValues / Filter_WeightRange (sphinxfilter.cpp:304-320) */
      int ok = 1;
      for (int i = 0; i < q->n_weight_filters && ok; i++) {
        const orc_filter* f = &q->weight_filters[i];
        const int64_t v = (int64_t)weight;
        int pass = 0;
        if (f->kind == ORC_FILTER_VALUES) {
          for (int k = 0; k < f->n_values; k++) pass = pass || f->values[k] == v;
        } else
          pass = (f->has_equal_min ? v >= f->min_value : v > f->min_value) && (f->has_equal_max ? v <= f->max_value : v < f->max_value);
        if (f->exclude) pass = !pass;
        ok = pass;
      }
      if (!ok) continue;
    }
    match_t m = {root->rowid, weight};
    mq_push(&mq, &m);
    if (--cutoff == 0) break;
  }

  /* Flatten (:627-641): best first */
  res->total_found = mq.total;
  int n = mq.used;
  res->n = n;
  for (int i = n - 1; i >= 0; i--) {
    res->rowid[i] = mq.data[0].rowid;
    res->weight[i] = mq.data[0].weight;
    mq_pop(&mq);
  }
  res->fetched_docs = bc.fetched_docs;
  res->fetched_hits = bc.fetched_hits;
  res->skips = bc.skips;
  free(mq.data);
  free(hv.p);
  en_free(root);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* throughput harness for bench.py's cpu_baseline leg: n_threads workers pull  */
/* queries off a shared counter, one independent query per thread at a time   */
/* (the reference's model: one coroutine per query x chunk, searchd.cpp:5654). */
/* ------------------------------------------------------------------------ */
typedef struct {
  const orc_index* idx;
  const orc_query* const* queries;
  int n_queries, repeat;
  volatile int next;
  int max_k;
  int errors;
} many_ctx;

static void* many_worker(void* p) {
  many_ctx* c = (many_ctx*)p;
  uint32_t* rowid = (uint32_t*)malloc((size_t)c->max_k * sizeof(uint32_t));
  int32_t* weight = (int32_t*)malloc((size_t)c->max_k * sizeof(int32_t));
  for (;;) {
    int i = __sync_fetch_and_add(&c->next, 1);
    if (i >= c->n_queries * c->repeat) break;
    orc_result r;
    r.rowid = rowid;
    r.weight = weight;
    if (orc_search(c->idx, c->queries[i % c->n_queries], &r) != 0) __sync_fetch_and_add(&c->errors, 1);
  }
  free(rowid);
  free(weight);
  return NULL;
}

/* runs every query `repeat` times over n_threads threads; returns wall seconds (<0 on error) */
double orc_search_many(const orc_index* idx, const orc_query* const* queries, int n_queries, int repeat, int n_threads) {
  many_ctx c;
  c.idx = idx;
  c.queries = queries;
  c.n_queries = n_queries;
  c.repeat = repeat;
  c.next = 0;
  c.errors = 0;
  c.max_k = 1;
  for (int i = 0; i < n_queries; i++)
    if (queries[i]->max_matches > c.max_k) c.max_k = queries[i]->max_matches;
  if (n_threads < 1) n_threads = 1;
  pthread_t* th = (pthread_t*)malloc((size_t)n_threads * sizeof(pthread_t));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, many_worker, &c);
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th);
  if (c.errors) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
