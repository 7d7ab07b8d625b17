// mrk_writer.cpp -- host-side index construction: a writer for the reference's v62
// .spd/.spp/.spe byte format and a deterministic synthetic posting generator.
//
// Format (CSphHitBuilder, sphinx.cpp:8378-8719; read back by DiskIndexQword_c, :357-550):
//   * every file starts with one dummy byte (0x01) so that offset 0 is never valid
//   * VLB ints: 7-bit groups, most significant first, bit 7 = "more" (sphinxstd.h:5545-5567)
//   * doclist entry (inline format): d(rowid), hits, then  hits==1 ? [pos23, field<<1|end]
//                                                                   : [fieldmask32, d(hitlist offset)]
//     plain format: d(rowid), d(hitlist offset), fieldmask32, hits;  list ends with a 0
//   * hitlist per (word, doc): d(hitpos)..., 0; a lone hit of an inline-format doc is not stored
//   * skiplist: one snapshot per skiplist_block_size docs taken BEFORE the doc is written:
//     {last rowid + 1, .spd position, hitlist base}; snapshot 0 is implicit; the rest is
//     delta-coded with -block and -4*block biases; written only when docs > block
//
// Unlike the reference writer this one builds each word's three byte runs independently
// (words are encoded in parallel) and concatenates them; the only cross-word dependency is
// the absolute hitlist offset in each word's first multi-hit entry, fixed by sizing the
// hitlists first.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <new>
#include <thread>
#include <vector>

#include "../../include/mrk.h"
#include "mrk_hostindex.h"

int mrk_fail(int code, const char* fmt, ...);

namespace {

// append-only byte run with amortised growth and unchecked writes after ensure()
struct Bytes {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  Bytes() = default;
  Bytes(const Bytes&) = delete;
  Bytes& operator=(const Bytes&) = delete;
  ~Bytes() { free(p); }
  void ensure(size_t extra) {
    if (n + extra <= cap) return;
    size_t nc = cap ? cap * 2 : 4096;
    while (nc < n + extra) nc *= 2;
    p = (uint8_t*)realloc(p, nc);
    cap = nc;
  }
  size_t size() const { return n; }
  const uint8_t* data() const { return p; }
  void clear() { n = 0; }
  void release() {
    free(p);
    p = nullptr;
    n = cap = 0;
  }
};

inline void put_vlb(Bytes& out, uint64_t v) {
  out.ensure(10);
  if (v < 128) {
    out.p[out.n++] = (uint8_t)v;
    return;
  }
  int n = 1;
  for (uint64_t t = v >> 7; t; t >>= 7) ++n;
  for (int i = n - 1; i >= 0; --i) out.p[out.n++] = (uint8_t)(((v >> (7 * i)) & 0x7f) | (i ? 0x80 : 0));
}

// postings of one word: docs in rowid order, each with its (deduplicated, sorted) hit positions
struct WordPostings {
  std::vector<uint32_t> rowid;
  std::vector<uint32_t> hit_begin; // index into hits, size docs+1
  std::vector<uint32_t> hits;      // raw Hitpos_t values incl. end-marker bit
};

struct WordBytes {
  Bytes spd, spp, spe;
  std::vector<uint32_t> spp_doc_off; // local .spp offset of each multi-hit doc's hitlist
  uint32_t docs = 0, n_hits = 0;
};

// hitlists first: their sizes fix every word's absolute .spp base
void encode_hitlists(const WordPostings& w, bool inline_fmt, WordBytes& o) {
  const size_t nd = w.rowid.size();
  o.spp_doc_off.assign(nd, 0);
  o.docs = (uint32_t)nd;
  o.n_hits = (uint32_t)w.hits.size();
  o.spp.ensure(w.hits.size() * 2 + nd + 16);
  for (size_t d = 0; d < nd; ++d) {
    const uint32_t hb = w.hit_begin[d], he = w.hit_begin[d + 1];
    o.spp_doc_off[d] = (uint32_t)o.spp.size();
    if (inline_fmt && he - hb == 1) continue; // lone hit lives in the doclist
    uint32_t prev = 0;
    for (uint32_t h = hb; h < he; ++h) {
      put_vlb(o.spp, w.hits[h] - prev);
      prev = w.hits[h];
    }
    put_vlb(o.spp, 0);
  }
}

void encode_doclist(const WordPostings& w, bool inline_fmt, uint32_t block, uint64_t spp_base, WordBytes& o) {
  const size_t nd = w.rowid.size();
  uint32_t last_rowid = 0xFFFFFFFFu;
  uint64_t last_hit_pos = 0; // m_iLastHitlistPos: resets to 0 per word (sphinx.cpp:8612)
  uint32_t snap_base = 0;
  uint64_t snap_off = 0, snap_hit = 0;
  o.spd.ensure(nd * 6 + 16);
  for (size_t d = 0; d < nd; ++d) {
    if ((d & (block - 1)) == 0 && d) { // snapshot 0 is implicit
      const uint32_t bp1 = last_rowid + 1u;
      put_vlb(o.spe, bp1 - snap_base - block);
      put_vlb(o.spe, (uint64_t)o.spd.size() - snap_off - 4ull * block);
      put_vlb(o.spe, last_hit_pos - snap_hit);
      snap_base = bp1;
      snap_off = o.spd.size();
      snap_hit = last_hit_pos;
    }
    put_vlb(o.spd, (uint32_t)(w.rowid[d] - last_rowid));
    last_rowid = w.rowid[d];
    const uint32_t hb = w.hit_begin[d], he = w.hit_begin[d + 1], nh = he - hb;
    uint32_t mask = 0;
    for (uint32_t h = hb; h < he; ++h) {
      const uint32_t f = w.hits[h] >> 24;
      if (f < 32) mask |= 1u << f;
    }
    const uint64_t my_hit_pos = spp_base + o.spp_doc_off[d];
    if (inline_fmt) {
      put_vlb(o.spd, nh);
      if (nh == 1) {
        put_vlb(o.spd, w.hits[hb] & 0x7FFFFFu);
        put_vlb(o.spd, w.hits[hb] >> 23);
      } else {
        put_vlb(o.spd, mask);
        put_vlb(o.spd, my_hit_pos - last_hit_pos);
        last_hit_pos = my_hit_pos;
      }
    } else {
      put_vlb(o.spd, my_hit_pos - last_hit_pos);
      put_vlb(o.spd, mask);
      put_vlb(o.spd, nh);
      last_hit_pos = my_hit_pos;
    }
  }
  put_vlb(o.spd, 0);
  if (nd <= block) o.spe.clear(); // "docs > block" gate (sphinx.cpp:8510)
}

} // namespace


namespace {

template <typename F>
void parallel_for(size_t n, uint32_t n_threads, F fn);

// concatenates per-word runs; spd runs must already carry absolute .spp offsets
int assemble(std::vector<WordBytes>& wb, const std::vector<uint64_t>& spp_base, uint32_t block, uint32_t n_threads,
             mrk_host_index** out) {
  mrk_host_index* h = new (std::nothrow) mrk_host_index();
  if (!h) return mrk_fail(MRK_E_NOMEM, "out of memory");
  const size_t nt = wb.size();
  std::vector<uint64_t> od(nt), op(nt), oe(nt);
  uint64_t pd = 1, pp = 1, pe = 1;
  h->dict.assign(nt, mrk_dict_entry{});
  for (size_t t = 0; t < nt; ++t) {
    const WordBytes& w = wb[t];
    mrk_dict_entry& e = h->dict[t];
    e.wordid = t + 1;
    od[t] = pd, op[t] = pp, oe[t] = pe;
    if (!w.docs) continue;
    if (pp != spp_base[t]) {
      delete h;
      return mrk_fail(MRK_E_FORMAT, "internal: hitlist base mismatch for word %zu", t);
    }
    e.doclist_off = pd;
    e.doclist_len = w.spd.size();
    e.skiplist_off = w.docs > block ? pe : 0;
    e.docs = w.docs;
    e.hits = w.n_hits;
    pd += w.spd.size();
    pp += w.spp.size();
    pe += w.spe.size();
  }
  h->spd_len = pd, h->spp_len = pp, h->spe_len = pe;
  h->spd = (uint8_t*)malloc(pd + 64);
  h->spp = (uint8_t*)malloc(pp + 64);
  h->spe = (uint8_t*)malloc(pe + 64);
  if (!h->spd || !h->spp || !h->spe) {
    delete h;
    return mrk_fail(MRK_E_NOMEM, "out of memory assembling the index");
  }
  h->spd[0] = h->spp[0] = h->spe[0] = 1; // dummy first byte (sphinx.cpp:8404-8409)
  memset(h->spd + pd, 0, 64);
  memset(h->spp + pp, 0, 64);
  memset(h->spe + pe, 0, 64);
  parallel_for(nt, n_threads, [&](size_t t) { // parallel copy = parallel first touch of the big buffers
    WordBytes& w = wb[t];
    if (!w.docs) return;
    if (w.spd.size()) memcpy(h->spd + od[t], w.spd.data(), w.spd.size());
    if (w.spp.size()) memcpy(h->spp + op[t], w.spp.data(), w.spp.size()); // (an all-inline term has no hitlist bytes: data() may be null)
    if (w.spe.size()) memcpy(h->spe + oe[t], w.spe.data(), w.spe.size());
    w.spd.release();
    w.spp.release();
    w.spe.release();
  });
  *out = h;
  return MRK_OK;
}

template <typename F>
void parallel_for(size_t n, uint32_t n_threads, F fn) {
  if (!n_threads) n_threads = std::max(1u, std::thread::hardware_concurrency());
  n_threads = (uint32_t)std::min<size_t>(n_threads, n ? n : 1);
  if (n_threads <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < n_threads; ++t)
    th.emplace_back([&] {
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n) break;
        fn(i);
      }
    });
  for (auto& x : th) x.join();
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool timing_on() { return getenv("MRK_TIMING") != nullptr; }

int build(std::vector<WordPostings>& words, uint32_t block, uint32_t hit_format, uint32_t n_threads, mrk_host_index** out) {
  if (!out) return mrk_fail(MRK_E_INVAL, "out is NULL");
  const double t0 = now_s();
  if (block == 0 || (block & (block - 1))) return mrk_fail(MRK_E_INVAL, "skiplist_block_size %u is not a power of two", block);
  const bool inline_fmt = hit_format == MRK_HITFMT_INLINE;
  const size_t nt = words.size();
  std::vector<WordBytes> wb(nt);
  parallel_for(nt, n_threads, [&](size_t t) { encode_hitlists(words[t], inline_fmt, wb[t]); });
  const double t1 = now_s();
  std::vector<uint64_t> spp_base(nt);
  uint64_t pp = 1;
  for (size_t t = 0; t < nt; ++t) {
    spp_base[t] = pp;
    if (wb[t].docs) pp += wb[t].spp.size();
  }
  parallel_for(nt, n_threads, [&](size_t t) {
    if (wb[t].docs) encode_doclist(words[t], inline_fmt, block, spp_base[t], wb[t]);
    std::vector<uint32_t>().swap(words[t].rowid);
    std::vector<uint32_t>().swap(words[t].hits);
    std::vector<uint32_t>().swap(words[t].hit_begin);
  });
  const double t2 = now_s();
  int rc = assemble(wb, spp_base, block, n_threads, out);
  if (timing_on()) fprintf(stderr, "[mrk] encode hitlists %.2fs, doclists %.2fs, assemble %.2fs\n", t1 - t0, t2 - t1, now_s() - t2);
  return rc;
}

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

} // namespace

extern "C" int mrk_index_from_hits(const uint64_t* wordid, const uint32_t* rowid, const uint32_t* hitpos, uint64_t n,
                                   uint32_t n_terms, uint32_t skiplist_block_size, uint32_t hit_format,
                                   mrk_host_index** out) {
  if (!out || ((!wordid || !rowid || !hitpos) && n)) return mrk_fail(MRK_E_INVAL, "mrk_index_from_hits: NULL argument");
  if (skiplist_block_size == 0 || skiplist_block_size > 65536 || hit_format > 1)
    return mrk_fail(MRK_E_INVAL, "mrk_index_from_hits: skiplist_block_size %u / hit_format %u", skiplist_block_size, hit_format);
  try {
  std::vector<WordPostings> words(n_terms);
  uint64_t prev_w = 0;
  uint32_t prev_r = 0, prev_h = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t w = wordid[i];
    if (w == 0 || w > n_terms) return mrk_fail(MRK_E_INVAL, "hit %llu: wordid %llu outside 1..%u", (unsigned long long)i, (unsigned long long)w, n_terms);
    const uint32_t pure = hitpos[i] & ~(1u << 23);
    if (i && (w < prev_w || (w == prev_w && (rowid[i] < prev_r || (rowid[i] == prev_r && pure < prev_h)))))
      return mrk_fail(MRK_E_INVAL, "hit %llu: hits must be sorted by (wordid, rowid, hitpos)", (unsigned long long)i);
    WordPostings& wp = words[w - 1];
    const bool new_doc = wp.rowid.empty() || w != prev_w || rowid[i] != prev_r;
    if (new_doc) {
      wp.rowid.push_back(rowid[i]);
      wp.hit_begin.push_back((uint32_t)wp.hits.size());
      wp.hits.push_back(hitpos[i]);
    } else if (pure == (wp.hits.back() & ~(1u << 23))) {
      // same position again: the first stays (sphinx.cpp:8672-8675); an end marker arriving with the
      // duplicate is dropped with it
    } else
      wp.hits.push_back(hitpos[i]);
    prev_w = w;
    prev_r = rowid[i];
    prev_h = pure;
  }
  for (auto& wp : words) wp.hit_begin.push_back((uint32_t)wp.hits.size());
  // the reference only sets the end marker when the NEXT hit is in another field or doc
  // (sphinx.cpp:8567-8574, 8678-8684): a marker followed by a hit in the same field is dropped
  for (auto& wp : words)
    for (size_t d = 0; d + 1 < wp.hit_begin.size(); ++d)
      for (uint32_t h = wp.hit_begin[d]; h + 1 < wp.hit_begin[d + 1]; ++h)
        if ((wp.hits[h] & (1u << 23)) && (wp.hits[h] >> 24) == (wp.hits[h + 1] >> 24)) wp.hits[h] &= ~(1u << 23);
  return build(words, skiplist_block_size, hit_format, 1, out);
  } catch (const std::bad_alloc&) {
    return mrk_fail(MRK_E_NOMEM, "mrk_index_from_hits: out of memory (%u terms, %llu hits)", n_terms, (unsigned long long)n);
  }
}

// Synthetic postings, deterministic in (seed, term index, GLOBAL rowid): the global rowid space is cut into chunks of
// SYNTH_CHUNK rowids and every (term, chunk) draws from its own splitmix64 stream (keyed on seed, term
// and rowid), so a shard [rowid_base, rowid_base + n_docs) holds exactly the postings the unsharded corpus holds for
// those rows -- "sharded result == unsharded result" is testable.  Inside a chunk term t occurs in a doc with
// probability term_prob[t] (geometric gaps; restarting them at a chunk border leaves the per-doc Bernoulli law
// untouched), tf = 1 + min(254, Geometric(1/2)), every hit falls into field 0 with probability title_frac (else one
// of the other fields) at a uniform position in [1, max_pos] (end_markers = 2: in [1, the field's length in that doc]); duplicate
// positions collapse like the reference writer collapses them.  A chunk the shard covers only partly is drawn whole and the rows outside the shard are dropped.
constexpr uint64_t SYNTH_CHUNK = 65536;

extern "C" int mrk_synth_generate(const mrk_synth_params* p, mrk_host_index** out) {
  if (!p || !out) return mrk_fail(MRK_E_INVAL, "mrk_synth_generate: NULL argument");
  if (!p->n_terms || !p->term_prob) return mrk_fail(MRK_E_INVAL, "mrk_synth_generate: no terms");
  if (p->n_docs == 0 || p->n_docs > 0xFFFFFFFEull) return mrk_fail(MRK_E_INVAL, "n_docs %llu outside 1..2^32-2", (unsigned long long)p->n_docs);
  if (p->rowid_base > (1ull << 40)) return mrk_fail(MRK_E_INVAL, "rowid_base %llu too large", (unsigned long long)p->rowid_base);
  if (p->n_fields == 0 || p->n_fields > 32 || p->max_pos == 0 || p->max_pos >= (1u << 23))
    return mrk_fail(MRK_E_INVAL, "mrk_synth_generate: bad n_fields/max_pos");
  for (uint32_t t = 0; t < p->n_terms; ++t)
    if (!(p->term_prob[t] > 0.0 && p->term_prob[t] <= 1.0)) return mrk_fail(MRK_E_INVAL, "term_prob[%u] outside (0,1]", t);
  const double tg0 = now_s();
  std::vector<WordPostings> words(p->n_terms);
  const uint64_t title_thr = (uint64_t)(p->title_frac * 18446744073709551615.0);
  const uint64_t g0 = p->rowid_base, g1 = p->rowid_base + p->n_docs; // the shard's global rowid range
  parallel_for(p->n_terms, p->n_threads, [&](size_t t) {
    WordPostings& w = words[t];
    const double pr = p->term_prob[t];
    const double inv_log1mp = pr < 1.0 ? 1.0 / log1p(-pr) : 0.0;
    w.rowid.reserve((size_t)(pr * (double)p->n_docs * 1.05) + 16);
    w.hit_begin.reserve(w.rowid.capacity() + 1);
    uint32_t tmp[256];
    for (uint64_t chunk = g0 / SYNTH_CHUNK; chunk * SYNTH_CHUNK < g1; ++chunk) {
      uint64_t s = p->seed ^ (0xD1B54A32D192ED03ull * (t + 1)) ^ (0x9FB21C651E98DF25ull * (chunk + 1));
      splitmix64(s);
      const uint64_t cbase = chunk * SYNTH_CHUNK;
      uint64_t row = 0; // next candidate rowid inside the chunk
      for (;;) {
        uint64_t gap = 0;
        if (pr < 1.0) {
          const double u = ((splitmix64(s) >> 11) + 1) * (1.0 / 9007199254740992.0); // (0,1]
          const double g = floor(log(u) * inv_log1mp);
          gap = g > 4.0e18 ? (uint64_t)4e18 : (uint64_t)g;
        }
        row += gap;
        if (row >= SYNTH_CHUNK || cbase + row >= g1) break;
        uint64_t r = splitmix64(s);
        uint32_t tf = 1 + (uint32_t)std::min(254, r ? __builtin_ctzll(r) : 64);
        uint32_t n = 0;
        for (uint32_t i = 0; i < tf; ++i) {
          const uint64_t a = splitmix64(s);
          uint32_t f = 0;
          if (p->n_fields > 1 && a >= title_thr) f = 1 + (uint32_t)((a >> 20) % (p->n_fields - 1));
          uint32_t len = p->max_pos;
          if (p->end_markers == 2) { // the field's length in this doc: a function of (seed, rowid, field) only -- every word sees the same one
            uint64_t hs = p->seed ^ (0xA24BAED4963EE407ull * (cbase + row + 1)) ^ (0x9E6C63D0676A9A99ull * (f + 1));
            len = 1 + (uint32_t)(splitmix64(hs) % p->max_pos);
          }
          const uint32_t pos = 1 + (uint32_t)((a & 0xFFFFF) * (uint64_t)len >> 20);
          tmp[n++] = (f << 24) | pos | ((p->end_markers == 2 && pos == len) ? 1u << 23 : 0u);
        }
        const uint64_t grow = cbase + row;
        ++row;
        if (grow < g0) continue; // a row of the chunk's part that belongs to the previous shard
        if (n > 1) {
          if (n == 2) {
            if (tmp[0] > tmp[1]) std::swap(tmp[0], tmp[1]);
          } else
            std::sort(tmp, tmp + n);
          n = (uint32_t)(std::unique(tmp, tmp + n) - tmp);
        }
        // end_markers = 1: each field's last hit of THIS WORD in this doc carries the flag (a stress for the hit merges: two words at one
        // position may differ in it); 2: the hit at the field's last POSITION does, whatever the word -- what the reference's indexer
        // writes (sphinx.cpp:22424-22430), set above with the position
        if (p->end_markers == 1)
          for (uint32_t i = 0; i < n; ++i)
            if (i + 1 == n || (tmp[i + 1] >> 24) != (tmp[i] >> 24)) tmp[i] |= 1u << 23;
        w.rowid.push_back((uint32_t)(grow - g0));
        w.hit_begin.push_back((uint32_t)w.hits.size());
        w.hits.insert(w.hits.end(), tmp, tmp + n);
      }
    }
    w.hit_begin.push_back((uint32_t)w.hits.size());
  });
  if (timing_on()) fprintf(stderr, "[mrk] sample postings %.2fs\n", now_s() - tg0);
  return build(words, p->skiplist_block_size, p->hit_format, p->n_threads, out);
}

extern "C" void mrk_host_index_free(mrk_host_index* h) { delete h; }
extern "C" const uint8_t* mrk_host_index_spd(const mrk_host_index* h, uint64_t* len) {
  if (len) *len = h->spd_len;
  return h->spd;
}
extern "C" const uint8_t* mrk_host_index_spp(const mrk_host_index* h, uint64_t* len) {
  if (len) *len = h->spp_len;
  return h->spp;
}
extern "C" const uint8_t* mrk_host_index_spe(const mrk_host_index* h, uint64_t* len) {
  if (len) *len = h->spe_len;
  return h->spe;
}
extern "C" const mrk_dict_entry* mrk_host_index_dict(const mrk_host_index* h, uint32_t* n_terms) {
  if (n_terms) *n_terms = (uint32_t)h->dict.size();
  return h->dict.data();
}
