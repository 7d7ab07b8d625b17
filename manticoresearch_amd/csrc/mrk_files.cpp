// mrk_files.cpp -- real index ingestion: reads the files a Manticore 3.x indexer (or an RT disk chunk save) wrote and
// hands them to the device path as a mrk_host_index.  Host only; nothing here touches the GPU.
//
//   .sph  header, format versions 54..62 (CSphIndex_VLN::LoadHeader, sphinx.cpp:13252-13388; ReadSchema :8722-8781;
//         LoadIndexSettings :13207-13249; CSphTokenizerSettings::Load / CSphDictSettings::Load /
//         CSphFieldFilterSettings::Load, indexsettings.cpp:303-334, 405-458, 506-515; CSphSavedFile::Read,
//         fileutils.cpp:114-119).  Everything is parsed -- the fields sit behind variable-length strings -- but only
//         what the match -> rank -> top-K path consumes is kept (mrk_index_info).
//   .spi  dictionary (doc/internals-index-format.txt:97-170).  dict=keywords: checkpoints {dword len, keyword, qword
//         offset} (CWordlist::Preread, indexformat.cpp:331-344), blocks of front-coded keywords
//         (KeywordsBlockReader_c::UnpackWord, :641-691).  dict=crc: checkpoints {qword wordid, qword offset}, blocks
//         of delta-coded {wordid, doclist offset, docs, hits, [skiplist offset]} (CWordlist::GetWord, :425-470).
//         The whole dictionary is expanded into the flat table the C-ABI takes: the lookup the reference does per
//         query (checkpoint search + block scan, DiskIndexQwordSetup_c::Setup, sphinx.cpp:12953-13060) becomes a
//         binary search over that table.
//   .spd / .spp / .spe   taken as they are.
//   .spm  dead-row map: one bit per row, DWORD words (DeadRowMap_c::IsSet, killlist.h:39-46).
#include <ctype.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "mrk_hostindex.h"

int mrk_fail(int code, const char* fmt, ...);

namespace {

constexpr uint32_t SPH_MAGIC = 0x58485053u; // "SPHX"
constexpr uint32_t MIN_VERSION = 54, MAX_VERSION = 62;
constexpr uint32_t DOCLIST_HINT_THRESH = 256; // indexformat.h:20
constexpr uint32_t HITLESS_FLAG = 0x80000000u;

bool read_file(const std::string& path, std::vector<uint8_t>& out, bool optional = false) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    if (!optional) mrk_fail(MRK_E_FORMAT, "cannot open %s", path.c_str());
    return false;
  }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
  fclose(f);
  if (got != out.size()) {
    mrk_fail(MRK_E_FORMAT, "short read on %s", path.c_str());
    return false;
  }
  return true;
}

// posting files go straight into the malloc'd, slack-padded buffers the host index owns
bool read_postings(const std::string& path, uint8_t*& buf, uint64_t& len) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    mrk_fail(MRK_E_FORMAT, "cannot open %s", path.c_str());
    return false;
  }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  len = n > 0 ? (uint64_t)n : 0;
  buf = (uint8_t*)calloc(len + 64, 1);
  if (!buf) {
    fclose(f);
    mrk_fail(MRK_E_NOMEM, "out of memory reading %s", path.c_str());
    return false;
  }
  const size_t got = len ? fread(buf, 1, len, f) : 0;
  fclose(f);
  if (got != len) {
    mrk_fail(MRK_E_FORMAT, "short read on %s", path.c_str());
    return false;
  }
  return true;
}

// CSphReader's getters over a byte buffer (fileio.cpp): little-endian fixed ints, dword-length strings, VLB
struct Reader {
  const uint8_t* p;
  size_t n, at = 0;
  bool bad = false;
  Reader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  bool need(size_t k) {
    if (bad || n - at < k) bad = true;
    return !bad;
  }
  uint8_t byte() { return need(1) ? p[at++] : 0; }
  uint32_t dword() {
    if (!need(4)) return 0;
    uint32_t v;
    memcpy(&v, p + at, 4);
    at += 4;
    return v;
  }
  uint64_t offset() {
    if (!need(8)) return 0;
    uint64_t v;
    memcpy(&v, p + at, 8);
    at += 8;
    return v;
  }
  std::string str() {
    const uint32_t len = dword();
    if (!need(len)) return std::string();
    std::string s((const char*)p + at, len);
    at += len;
    return s;
  }
  uint64_t zint() { // sphUnzipInt / sphUnzipOffset / sphUnzipWordid: 7-bit groups, most significant first
    uint64_t v = 0;
    for (;;) {
      const uint8_t b = byte();
      if (bad) return 0;
      v = (v << 7) | (b & 0x7fu);
      if (!(b & 0x80u)) return v;
    }
  }
  void saved_file() { // CSphSavedFile::Read: size, ctime, mtime, crc32
    offset(), offset(), offset(), dword();
  }
};

bool schema_column(Reader& r, uint32_t version, mrk_host_index::Attr* out = nullptr) { // ReadSchemaColumn; -> payload flag
  std::string name = r.str();
  const uint32_t type = r.dword();
  r.dword(); // rowitem (ignored by the reference too)
  const int32_t bit_offset = (int32_t)r.dword(), bit_count = (int32_t)r.dword();
  const bool payload = r.byte() != 0;
  if (version >= 61) r.dword(); // attr flags
  if (out) {
    for (char& c : name) c = (char)tolower((unsigned char)c);
    *out = mrk_host_index::Attr{name.empty() ? "@emptyname" : name, type, bit_offset, bit_count};
  }
  return payload;
}

// ReadSchema (sphinx.cpp:8722-8781): full-text fields, then attributes
int read_schema(Reader& r, const char* name, mrk_host_index* h) {
  mrk_index_info& I = h->info;
  // schema: full-text fields, then attributes
  I.n_fields = r.dword();
  if (I.n_fields > 256) return mrk_fail(MRK_E_FORMAT, "%s: %u fields", name, I.n_fields);
  for (uint32_t i = 0; i < I.n_fields && !r.bad; ++i) {
    if (I.version >= 57) { // ReadSchemaField
      h->fields.push_back(r.str());
      if (r.dword() & 1u) h->stored_fields = true; // field flags: FIELD_STORED
      if (r.byte()) h->payload_fields |= i < 64 ? 1ull << i : 1ull << 63; // payload
    } else {
      const size_t at = r.at;
      std::string nm = r.str();
      r.at = at;
      if (schema_column(r, I.version)) h->payload_fields |= i < 64 ? 1ull << i : 1ull << 63;
      h->fields.push_back(nm);
    }
  }
  I.n_attrs = r.dword();
  if (I.n_attrs > 65536) return mrk_fail(MRK_E_FORMAT, "%s: %u attributes", name, I.n_attrs);
  for (uint32_t i = 0; i < I.n_attrs && !r.bad; ++i) {
    mrk_host_index::Attr a;
    schema_column(r, I.version, &a);
    h->attrs.push_back(a);
  }
  return MRK_OK;
}

// LoadIndexSettings + CSphTokenizerSettings::Load + CSphDictSettings::Load, in the order both the .sph header and an RT
// index's .meta hold them
void read_settings(Reader& r, mrk_host_index* h) {
  mrk_index_info& I = h->info;
  // LoadIndexSettings
  I.min_prefix_len = r.dword();
  I.min_infix_len = r.dword();
  r.dword(); // max substring len
  r.byte();  // html strip
  r.str(), r.str();
  r.byte(); // index exact words
  I.hitless = r.dword();
  I.hit_format = r.dword();
  I.index_sp = r.byte();
  r.str();                                    // zones
  r.dword(), r.dword(), r.dword(), r.dword(); // boundary / stopword / overshort steps, embedded limit
  r.byte();                                   // bigram index
  r.str();                                    // bigram words
  I.index_field_lens = r.byte();
  r.byte();   // preprocessor
  r.str();    // was: RLP context
  r.str();    // index token filter
  r.offset(); // blob update space
  I.skiplist_block_size = I.version < 56 ? 128u : r.dword();
  if (I.version >= 60) r.str(); // hitless files
  // tokenizer settings
  r.byte(); // type
  r.str();  // case folding
  r.dword();
  if (r.byte()) { // embedded synonyms
    const uint32_t ns = r.dword();
    for (uint32_t i = 0; i < ns && !r.bad; ++i) r.str();
  }
  r.str();
  r.saved_file();
  r.str(), r.str();
  r.dword();
  r.str(), r.str(), r.str();
  // dictionary settings
  r.str(), r.str();
  const bool emb_stop = r.byte() != 0;
  if (emb_stop) {
    const uint32_t ns = r.dword();
    for (uint32_t i = 0; i < ns && !r.bad; ++i) r.zint();
  }
  r.str();
  const uint32_t n_stop_files = r.dword();
  for (uint32_t i = 0; i < n_stop_files && !r.bad; ++i) {
    r.str();
    r.saved_file();
  }
  if (r.byte()) { // embedded wordforms
    const uint32_t nw = r.dword();
    for (uint32_t i = 0; i < nw && !r.bad; ++i) r.str();
  }
  const uint32_t n_wf = r.dword();
  for (uint32_t i = 0; i < n_wf && !r.bad; ++i) {
    r.str();
    r.saved_file();
  }
  r.dword(); // min stemming len
  I.word_dict = r.byte() != 0;
  r.byte(); // stopwords unstemmed
  r.str();  // morphology fingerprint
}

int parse_header(const std::vector<uint8_t>& sph, const char* name, mrk_host_index* h, uint64_t& cp_offset) {
  Reader r(sph.data(), sph.size());
  mrk_index_info& I = h->info;
  if (r.dword() != SPH_MAGIC) return mrk_fail(MRK_E_FORMAT, "%s: not an index header (magic)", name);
  I.version = r.dword();
  if (I.version < MIN_VERSION || I.version > MAX_VERSION)
    return mrk_fail(MRK_E_FORMAT, "%s is v.%u, this reader covers v.%u..%u", name, I.version, MIN_VERSION, MAX_VERSION);
  if (int rc = read_schema(r, name, h)) return rc;
  // dictionary header
  cp_offset = r.offset();
  I.n_checkpoints = r.dword();
  r.byte();  // infix codepoint bytes
  r.dword(); // infix blocks offset
  r.dword(); // infix blocks words size
  // index stats
  I.total_docs = r.dword();
  I.total_bytes = r.offset();
  read_settings(r, h);
  h->docinfo_rows = r.offset(); // m_iDocinfo: rows in .spa
  r.offset();                   // m_iDocinfoIndex
  r.offset();                   // m_iMinMaxIndex: where the min-max index starts in .spa, in dwords
  const uint32_t n_re = r.dword();
  for (uint32_t i = 0; i < n_re && !r.bad; ++i) r.str();
  if (I.index_field_lens)
    for (uint32_t i = 0; i < I.n_fields; ++i) r.offset();
  if (r.bad) return mrk_fail(MRK_E_FORMAT, "%s: failed to parse header (unexpected eof)", name);
  if (I.skiplist_block_size == 0 || I.skiplist_block_size > (1u << 20))
    return mrk_fail(MRK_E_FORMAT, "%s: skiplist block size %u", name, I.skiplist_block_size);
  if (I.hit_format > 1) return mrk_fail(MRK_E_FORMAT, "%s: hit format %u", name, I.hit_format);
  return MRK_OK;
}

int parse_dict(const std::vector<uint8_t>& spi, const char* name, uint64_t cp_offset, mrk_host_index* h, std::vector<uint64_t>& wordids) {
  const mrk_index_info& I = h->info;
  if (I.n_checkpoints == 0) return MRK_OK; // empty index
  if (cp_offset == 0 || cp_offset >= spi.size()) return mrk_fail(MRK_E_FORMAT, "%s: checkpoints offset past the file", name);
  // a checkpoint takes >= 13 bytes (dword length + >= 1 keyword byte + qword offset) / exactly 16 (dict=crc): bound the
  // header's count by the bytes that are there BEFORE allocating from it
  if ((uint64_t)I.n_checkpoints > (spi.size() - cp_offset) / (I.word_dict ? 13u : 16u))
    return mrk_fail(MRK_E_FORMAT, "%s: %u checkpoints cannot fit the %zu bytes behind offset %llu", name, I.n_checkpoints,
                    spi.size() - (size_t)cp_offset, (unsigned long long)cp_offset);
  Reader cp(spi.data(), spi.size());
  cp.at = (size_t)cp_offset;
  std::vector<uint64_t> blocks(I.n_checkpoints);
  for (uint32_t i = 0; i < I.n_checkpoints; ++i) {
    if (I.word_dict)
      cp.str(); // dword length + the block's first keyword
    else
      cp.offset(); // the block's first word id
    blocks[i] = cp.offset();
  }
  if (cp.bad) return mrk_fail(MRK_E_FORMAT, "%s: truncated checkpoints", name);
  char word[3 * 42 + 4 + 128] = {0}; // MAX_KEYWORD_BYTES-sized scratch, generously
  for (uint32_t b = 0; b < I.n_checkpoints; ++b) {
    if (blocks[b] == 0 || blocks[b] >= cp_offset) return mrk_fail(MRK_E_FORMAT, "%s: checkpoint %u points outside the word blocks", name, b);
    Reader r(spi.data(), (size_t)cp_offset);
    r.at = (size_t)blocks[b];
    uint64_t last_id = 0, last_off = 0;
    uint32_t wlen = 0;
    for (;;) {
      mrk_dict_entry e{};
      if (I.word_dict) {
        const uint8_t pack = r.byte();
        if (r.bad || !pack) break; // block end
        uint32_t match, delta;
        if (pack & 0x80u)
          delta = ((pack >> 4) & 7u) + 1u, match = pack & 15u;
        else
          delta = pack & 127u, match = r.byte();
        if (match > wlen || match + delta >= sizeof(word) - 1 || !r.need(delta))
          return mrk_fail(MRK_E_FORMAT, "%s: broken keyword entry in block %u", name, b);
        memcpy(word + match, r.p + r.at, delta);
        r.at += delta;
        wlen = match + delta;
        word[wlen] = 0;
        e.doclist_off = r.zint();
        e.docs = (uint32_t)r.zint();
        e.hits = (uint32_t)r.zint();
        if (e.docs >= DOCLIST_HINT_THRESH) r.byte(); // doclist size hint
        e.skiplist_off = (e.docs & ~HITLESS_FLAG) > I.skiplist_block_size ? r.zint() : 0;
        h->word_off.push_back((uint32_t)h->words.size());
        h->words.insert(h->words.end(), word, word + wlen + 1);
      } else {
        const uint64_t dw = r.zint();
        if (r.bad || !dw) break; // block end (followed by the last doclist's length)
        last_id += dw;
        last_off += r.zint();
        e.wordid = last_id;
        e.doclist_off = last_off;
        e.docs = (uint32_t)r.zint();
        e.hits = (uint32_t)r.zint();
        e.skiplist_off = (e.docs & ~HITLESS_FLAG) > I.skiplist_block_size ? r.zint() : 0;
        wordids.push_back(last_id);
      }
      if (r.bad) return mrk_fail(MRK_E_FORMAT, "%s: truncated dictionary block %u", name, b);
      if (e.docs & HITLESS_FLAG) return mrk_fail(MRK_E_UNSUPPORTED, "%s: hitless keywords are not on the device path", name);
      if (!e.docs || !e.hits || !e.doclist_off) return mrk_fail(MRK_E_FORMAT, "%s: dictionary entry without postings in block %u", name, b);
      h->dict.push_back(e);
    }
    if (r.bad) return mrk_fail(MRK_E_FORMAT, "%s: dictionary block %u runs past the checkpoints", name, b);
  }
  return MRK_OK;
}

} // namespace

static int index_open(const char* path_prefix, mrk_host_index* h);

extern "C" int mrk_index_open(const char* path_prefix, mrk_host_index** out) {
  if (!path_prefix || !out) return mrk_fail(MRK_E_INVAL, "mrk_index_open: null argument");
  *out = nullptr;
  mrk_host_index* h = new (std::nothrow) mrk_host_index();
  if (!h) return mrk_fail(MRK_E_NOMEM, "out of memory");
  int rc;
  try { // the C-ABI never throws: sizes come from untrusted files
    rc = index_open(path_prefix, h);
  } catch (const std::bad_alloc&) {
    rc = mrk_fail(MRK_E_NOMEM, "%s: out of memory reading the index", path_prefix);
  } catch (const std::exception& e) {
    rc = mrk_fail(MRK_E_FORMAT, "%s: %s", path_prefix, e.what());
  }
  if (rc != MRK_OK) {
    delete h;
    return rc;
  }
  *out = h;
  return MRK_OK;
}

static int index_open(const char* path_prefix, mrk_host_index* h) {
  const std::string base(path_prefix);
  h->from_files = true;
  std::vector<uint8_t> sph, spi, spm;
  std::vector<uint64_t> wordids;
  uint64_t cp_offset = 0;
  int rc = MRK_OK;
  if (!read_file(base + ".sph", sph) || !read_file(base + ".spi", spi)) rc = MRK_E_FORMAT;
  if (rc == MRK_OK) rc = parse_header(sph, (base + ".sph").c_str(), h, cp_offset);
  if (rc == MRK_OK && h->info.hitless != 0) rc = mrk_fail(MRK_E_UNSUPPORTED, "%s: hitless index (hitless_words) is not on the device path", path_prefix);
  // a payload field switches SPH_RANK_PROXIMITY_BM25 to RankerState_ProximityPayload_fn (sphinxsearch.cpp:4175-4199), which
  // the device path does not restate: such an index is declined rather than ranked differently
  if (rc == MRK_OK && h->payload_fields) rc = mrk_fail(MRK_E_UNSUPPORTED, "%s: payload fields are not on the device path", path_prefix);
  if (rc == MRK_OK) rc = parse_dict(spi, (base + ".spi").c_str(), cp_offset, h, wordids);
  if (rc == MRK_OK && (!read_postings(base + ".spd", h->spd, h->spd_len) || !read_postings(base + ".spp", h->spp, h->spp_len) ||
                       !read_postings(base + ".spe", h->spe, h->spe_len)))
    rc = MRK_E_FORMAT;
  if (rc == MRK_OK) {
    // m_iDoclistLength: doclists lie back to back in .spd (in word-id order, not in dictionary order), each ends with its 0
    std::vector<uint32_t> by_off(h->dict.size());
    for (uint32_t i = 0; i < by_off.size(); ++i) by_off[i] = i;
    std::sort(by_off.begin(), by_off.end(), [&](uint32_t a, uint32_t b) { return h->dict[a].doclist_off < h->dict[b].doclist_off; });
    for (size_t i = 0; i < by_off.size() && rc == MRK_OK; ++i) {
      mrk_dict_entry& e = h->dict[by_off[i]];
      const uint64_t end = i + 1 < by_off.size() ? h->dict[by_off[i + 1]].doclist_off : h->spd_len;
      if (end <= e.doclist_off || end > h->spd_len || h->spd[end - 1] != 0)
        rc = mrk_fail(MRK_E_FORMAT, "%s.spd: doclist at %llu does not end where the next one starts", path_prefix, (unsigned long long)e.doclist_off);
      e.doclist_len = end - e.doclist_off;
      if (e.skiplist_off >= h->spe_len && e.docs > h->info.skiplist_block_size)
        rc = mrk_fail(MRK_E_FORMAT, "%s.spe: skiplist offset %llu past the file", path_prefix, (unsigned long long)e.skiplist_off);
    }
  }
  if (rc == MRK_OK && !h->info.word_dict)
    for (size_t i = 0; i < h->dict.size(); ++i) h->dict[i].wordid = wordids[i];
  if (rc == MRK_OK && read_file(base + ".spm", spm, true)) {
    h->dead.assign((spm.size() + 3) / 4, 0u);
    if (!spm.empty()) memcpy(h->dead.data(), spm.data(), spm.size());
    const uint64_t rows = h->info.total_docs;
    for (uint64_t r = 0; r < rows && (r >> 5) < h->dead.size(); ++r) h->info.n_dead += (h->dead[r >> 5] >> (r & 31u)) & 1u;
    if (h->dead.size() * 32ull < rows) rc = mrk_fail(MRK_E_FORMAT, "%s.spm: %zu bytes for %llu rows", path_prefix, spm.size(), (unsigned long long)rows);
  }
  if (rc == MRK_OK) {
    // row-wise attributes: row size = the widest locator, in dwords (CSphSchema::GetRowSize); rows lead the .spa file
    uint32_t bits = 0;
    for (const mrk_host_index::Attr& a : h->attrs)
      if (a.bit_offset >= 0 && a.bit_count > 0) bits = std::max<uint32_t>(bits, (uint32_t)(a.bit_offset + a.bit_count));
    h->attr_stride = (bits + 31) / 32;
    std::vector<uint8_t> spa;
    if (h->attr_stride && h->docinfo_rows && read_file(base + ".spa", spa, true)) {
      // by division: docinfo_rows is a raw header qword, the product may wrap
      if (h->docinfo_rows > spa.size() / ((uint64_t)h->attr_stride * 4ull))
        rc = mrk_fail(MRK_E_FORMAT, "%s.spa: %zu bytes for %llu rows of %u dwords", path_prefix, spa.size(), (unsigned long long)h->docinfo_rows, h->attr_stride);
      else if (h->docinfo_rows < h->info.total_docs)
        rc = mrk_fail(MRK_E_FORMAT, "%s.sph: %llu attribute rows for %llu documents", path_prefix, (unsigned long long)h->docinfo_rows,
                      (unsigned long long)h->info.total_docs);
      else {
        const uint64_t need = h->docinfo_rows * h->attr_stride * 4ull;
        h->attr_rows.resize((size_t)(need / 4));
        memcpy(h->attr_rows.data(), spa.data(), (size_t)need);
      }
    }
  }
  if (rc == MRK_OK) read_file(base + ".spb", h->blobs, true); // blob pool (strings, MVAs, JSON); absent when the schema has none
  return rc;
}

extern "C" const uint8_t* mrk_host_index_blobs(const mrk_host_index* h, uint64_t* len, uint32_t* n_blob_attrs) {
  if (n_blob_attrs) {
    *n_blob_attrs = 0;
    if (h)
      for (const mrk_host_index::Attr& a : h->attrs) *n_blob_attrs += a.bit_count == 0 ? 1u : 0u; // blob-stored: no bits in the row
  }
  if (len) *len = h ? h->blobs.size() : 0;
  return h && !h->blobs.empty() ? h->blobs.data() : nullptr;
}

extern "C" int mrk_host_index_info(const mrk_host_index* h, mrk_index_info* out) {
  if (!h || !out || !h->from_files) return mrk_fail(MRK_E_INVAL, "mrk_host_index_info: not an index opened from files");
  *out = h->info;
  return MRK_OK;
}

extern "C" const char* mrk_host_index_field_name(const mrk_host_index* h, uint32_t field) {
  return (h && field < h->fields.size()) ? h->fields[field].c_str() : nullptr;
}

extern "C" const char* mrk_host_index_word(const mrk_host_index* h, uint32_t term_id, uint32_t* len) {
  if (!h || term_id >= h->word_off.size()) return nullptr;
  const char* w = h->words.data() + h->word_off[term_id];
  if (len) *len = (uint32_t)strlen(w);
  return w;
}

// sphDictCmpStrictly: bytes first, then length
extern "C" int32_t mrk_host_index_find_word(const mrk_host_index* h, const char* word, int32_t len) {
  if (!h || !word || len <= 0 || h->word_off.empty()) return -1;
  size_t lo = 0, hi = h->word_off.size();
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    const char* w = h->words.data() + h->word_off[mid];
    const int32_t wl = (int32_t)strlen(w);
    int c = memcmp(word, w, (size_t)std::min(len, wl));
    if (c == 0) c = len - wl;
    if (c == 0) return (int32_t)mid;
    if (c < 0)
      hi = mid;
    else
      lo = mid + 1;
  }
  return -1;
}

extern "C" int32_t mrk_host_index_find_wordid(const mrk_host_index* h, uint64_t wordid) {
  if (!h || !h->from_files || h->info.word_dict) return -1;
  size_t lo = 0, hi = h->dict.size();
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (h->dict[mid].wordid == wordid) return (int32_t)mid;
    if (h->dict[mid].wordid < wordid)
      lo = mid + 1;
    else
      hi = mid;
  }
  return -1;
}

extern "C" int mrk_host_index_attr(const mrk_host_index* h, uint32_t i, mrk_attr_info* out) {
  if (!h || !out || i >= h->attrs.size()) return mrk_fail(MRK_E_INVAL, "mrk_host_index_attr: no attribute %u", i);
  const mrk_host_index::Attr& a = h->attrs[i];
  *out = mrk_attr_info{a.name.c_str(), a.type, a.bit_offset, a.bit_count};
  return MRK_OK;
}

extern "C" const uint32_t* mrk_host_index_attr_rows(const mrk_host_index* h, uint32_t* stride_dwords, uint64_t* n_rows) {
  if (!h || h->attr_rows.empty()) return nullptr;
  if (stride_dwords) *stride_dwords = h->attr_stride;
  if (n_rows) *n_rows = h->docinfo_rows;
  return h->attr_rows.data();
}

extern "C" const uint32_t* mrk_host_index_dead_rows(const mrk_host_index* h, uint64_t* n_rows) {
  if (!h || h->dead.empty()) return nullptr;
  if (n_rows) *n_rows = h->info.total_docs;
  return h->dead.data();
}

// ---------------------------------------------------------------------------------------------------------------------
// RT index RAM chunk: <prefix>.meta (RtIndex_c::SaveMeta / LoadMeta, sphinxrt.cpp:3560-3640, 3732-3868) + <prefix>.ram
// (SaveRamChunk / LoadRamChunk, :4034-4225).  Every RAM segment holds its own small dictionary, doclists and hitlists in
// the RT codecs (RtWordReader_t / RtDocReader_t / RtHitReader_t, :390-640: little-endian-first varints, keywords front-
// coded against the previous one, deltas restarted every m_iWordsCheckpoint words); each segment is decoded into hits and
// re-emitted in the disk format the device path takes -- an RT segment becomes one more mrk_host_index / mrk_segment.
struct mrk_rt_ram {
  std::vector<mrk_host_index*> seg;
  ~mrk_rt_ram() {
    for (mrk_host_index* h : seg) delete h;
  }
};

namespace {

struct RtBytes { // UnzipT (sphinxrt.cpp:141-157) over a bounded byte range
  const uint8_t* p;
  size_t n, at = 0;
  bool bad = false;
  uint64_t zip() {
    uint64_t v = 0;
    for (int off = 0;; off += 7) {
      if (at >= n || off > 63) {
        bad = true;
        return 0;
      }
      const uint8_t b = p[at++];
      v += (uint64_t)(b & 0x7Fu) << off;
      if (!(b & 0x80u)) return v;
    }
  }
};

int rt_vector(Reader& r, size_t elem, const char* what, const char* name, std::vector<uint8_t>& out) { // LoadVector (:3997-4010)
  const uint32_t cnt = r.dword();
  if (r.bad || (uint64_t)cnt * elem > r.n - r.at) return mrk_fail(MRK_E_FORMAT, "%s: %s vector of %u entries past the file", name, what, cnt);
  out.assign(r.p + r.at, r.p + r.at + (size_t)cnt * elem);
  r.at += (size_t)cnt * elem;
  return MRK_OK;
}

// One RAM segment's dictionary, doclists and hitlists (the three byte vectors of RtSegment_t, sphinxrt.h:140-149, in the RT codecs:
// LSB-first varints, sphinxrt.cpp:98-157) -> the postings as (word, rowid, hit) triples; shared by the .ram file reader and by
// mrk_rt_segment_open (a live segment's vectors handed over in memory)
static int rt_decode_segment(const uint8_t* words, size_t words_len, const uint8_t* docs, size_t docs_len, const uint8_t* hits, size_t hits_len, uint32_t rows,
                             bool word_dict, uint32_t words_checkpoint, const char* name, uint32_t si, std::vector<uint64_t>& W, std::vector<uint32_t>& Rw,
                             std::vector<uint32_t>& H, std::vector<char>& wtext, std::vector<uint32_t>& woff, std::vector<uint64_t>& wordids, uint32_t& term) {
    // ---- walk the segment's dictionary: RtWordReader_t::UnzipWord (:528-575)
    RtBytes wr{words, words_len};
    uint8_t packed[260];
    packed[0] = 0;
    uint64_t wordid = 0, doc_off = 0;
    uint32_t n_in_cp = 0;
    term = 0;
    while (wr.at < wr.n) {
      if (++n_in_cp == words_checkpoint) doc_off = 0, n_in_cp = 1, wordid = word_dict ? wordid : 0;
      if (word_dict) {
        uint32_t match, delta;
        const uint8_t pk = wr.p[wr.at++];
        if (pk & 0x80u)
          delta = ((pk >> 4) & 7u) + 1u, match = pk & 15u;
        else {
          delta = pk & 127u;
          if (wr.at >= wr.n) return mrk_fail(MRK_E_FORMAT, "%s: segment %u: truncated keyword", name, si);
          match = wr.p[wr.at++];
        }
        if (match > packed[0] || match + delta > 255u || wr.n - wr.at < delta) return mrk_fail(MRK_E_FORMAT, "%s: segment %u: bad front-coded keyword", name, si);
        packed[0] = (uint8_t)(match + delta);
        memcpy(packed + 1 + match, wr.p + wr.at, delta);
        wr.at += delta;
        woff.push_back((uint32_t)wtext.size());
        wtext.insert(wtext.end(), packed + 1, packed + 1 + packed[0]);
        wtext.push_back(0);
      } else {
        wordid += wr.zip();
        wordids.push_back(wordid);
      }
      const uint64_t n_docs = wr.zip(), n_hits = wr.zip();
      doc_off += wr.zip();
      if (wr.bad || doc_off > docs_len || n_docs > docs_len || n_hits > hits_len + n_docs)
        return mrk_fail(MRK_E_FORMAT, "%s: segment %u: dictionary entry %u points past the doclists", name, si, term);
      // ---- its docs: RtDocReader_t::UnzipDoc (:397-421); rowid deltas start from INVALID_ROWID (~0: the first delta wraps)
      RtBytes dr{docs, docs_len, (size_t)doc_off};
      uint32_t rowid = 0xFFFFFFFFu;
      for (uint64_t d = 0; d < n_docs; ++d) {
        rowid += (uint32_t)dr.zip();
        dr.zip(); // field mask (recomputed from the hits by the writer below)
        const uint64_t dh = dr.zip();
        uint32_t hit1 = 0;
        uint64_t hoff = 0;
        if (dh == 1) {
          const uint64_t a = dr.zip(), b = dr.zip();
          hit1 = (uint32_t)(a + (b << 24));
        } else
          hoff = dr.zip();
        if (dr.bad || rowid >= rows || dh == 0 || (dh != 1 && hoff > hits_len))
          return mrk_fail(MRK_E_FORMAT, "%s: segment %u: word %u: bad doclist entry", name, si, term);
        if (dh == 1) {
          W.push_back(term + 1), Rw.push_back(rowid), H.push_back(hit1);
          continue;
        }
        RtBytes hr{hits, hits_len, (size_t)hoff}; // RtHitReader_t::UnzipHit (:612-622)
        uint32_t last = 0;
        for (uint64_t k = 0; k < dh; ++k) {
          last += (uint32_t)hr.zip();
          if (hr.bad) return mrk_fail(MRK_E_FORMAT, "%s: segment %u: word %u: hitlist runs past the segment", name, si, term);
          W.push_back(term + 1), Rw.push_back(rowid), H.push_back(last);
        }
      }
      ++term;
    }
    return MRK_OK;
}

int rt_open(const char* path_prefix, mrk_rt_ram* rt) {
  const std::string base(path_prefix), mname = base + ".meta", rname = base + ".ram";
  std::vector<uint8_t> meta, ram;
  if (!read_file(mname, meta) || !read_file(rname, ram)) return MRK_E_FORMAT;
  mrk_host_index proto; // what the .meta says, copied into every segment
  proto.from_files = true;
  Reader m(meta.data(), meta.size());
  if (m.dword() != 0x54525053u) return mrk_fail(MRK_E_FORMAT, "%s: not an RT meta file (magic)", mname.c_str()); // 'SPRT'
  const uint32_t meta_ver = m.dword();
  if (meta_ver < 14 || meta_ver > 18) return mrk_fail(MRK_E_FORMAT, "%s is meta v.%u, this reader covers v.14..18", mname.c_str(), meta_ver);
  m.dword();  // total documents
  m.offset(); // total bytes
  m.offset(); // TID
  proto.info.version = m.dword(); // the settings' version = the disk format version they were saved with
  if (proto.info.version < MIN_VERSION || proto.info.version > MAX_VERSION)
    return mrk_fail(MRK_E_FORMAT, "%s: settings of index format v.%u, this reader covers v.%u..%u", mname.c_str(), proto.info.version, MIN_VERSION, MAX_VERSION);
  if (int rc = read_schema(m, mname.c_str(), &proto)) return rc;
  read_settings(m, &proto);
  const uint32_t words_checkpoint = m.dword();
  if (m.bad) return mrk_fail(MRK_E_FORMAT, "%s: failed to parse (unexpected eof)", mname.c_str());
  if (words_checkpoint < 2 || words_checkpoint > (1u << 20)) return mrk_fail(MRK_E_FORMAT, "%s: words checkpoint %u", mname.c_str(), words_checkpoint);
  if (proto.info.hitless != 0) return mrk_fail(MRK_E_UNSUPPORTED, "%s: hitless index (hitless_words) is not on the device path", path_prefix);
  if (proto.payload_fields) return mrk_fail(MRK_E_UNSUPPORTED, "%s: payload fields are not on the device path", path_prefix);
  if (proto.info.n_fields > 32) return mrk_fail(MRK_E_UNSUPPORTED, "%s: %u fields (RT doclists carry a 32-bit field mask)", path_prefix, proto.info.n_fields);
  const bool word_dict = proto.info.word_dict != 0;

  Reader r(ram.data(), ram.size());
  r.dword();
  const uint32_t n_seg = r.dword();
  if (r.bad || n_seg > (1u << 16)) return mrk_fail(MRK_E_FORMAT, "%s: %u segments", rname.c_str(), n_seg);
  for (uint32_t si = 0; si < n_seg; ++si) {
    const uint32_t rows = r.dword();
    r.dword(); // alive rows
    r.dword();
    std::vector<uint8_t> words, kwcp, docs, hits, rowdata, blobs, infix;
    int rc;
    if ((rc = rt_vector(r, 1, "ram-words", rname.c_str(), words))) return rc;
    if (word_dict && (rc = rt_vector(r, 1, "ram-checkpoints", rname.c_str(), kwcp))) return rc;
    const uint32_t n_cp = r.dword();
    if (r.bad || (uint64_t)n_cp * 16 > r.n - r.at) return mrk_fail(MRK_E_FORMAT, "%s: %u word checkpoints past the file", rname.c_str(), n_cp);
    r.at += (size_t)n_cp * 16; // (offset, keyword offset | word id) pairs: a lookup aid, the words are walked in full here
    if ((rc = rt_vector(r, 1, "ram-doclist", rname.c_str(), docs)) || (rc = rt_vector(r, 1, "ram-hitlist", rname.c_str(), hits)) ||
        (rc = rt_vector(r, 4, "ram-attributes", rname.c_str(), rowdata)))
      return rc;
    // DeadRowMap_Ram_c::Load (killlist.cpp:120-129).  `rows` is an untrusted dword: the word count in 64 bits ((rows + 31) / 32 wraps
    // to 0 for rows >= 0xFFFFFFE1 -- an empty map, then an out-of-bounds read in the popcount below), checked against the bytes
    // that are left BEFORE anything is allocated from it
    const uint64_t dead_words = ((uint64_t)rows + 31) / 32;
    if (r.bad || dead_words * 4 > (uint64_t)(r.n - r.at)) return mrk_fail(MRK_E_FORMAT, "%s: dead-row map of %u rows past the file", rname.c_str(), rows);
    std::vector<uint32_t> dead((size_t)dead_words, 0u);
    if (!r.need(dead.size() * 4)) return mrk_fail(MRK_E_FORMAT, "%s: dead-row map of %u rows past the file", rname.c_str(), rows);
    if (!dead.empty()) memcpy(dead.data(), r.p + r.at, dead.size() * 4);
    r.at += dead.size() * 4;
    if ((rc = rt_vector(r, 1, "ram-blobs", rname.c_str(), blobs))) return rc;
    if (meta_ver >= 15 && proto.stored_fields) { // DocstoreRT_c::Load (docstore.cpp:1685-1699): the stored fields' text, stepped over
      const uint64_t nd = r.zint();
      for (uint64_t i = 0; i < nd && !r.bad; ++i) {
        const uint64_t len = r.zint();
        if (!r.need((size_t)len)) break;
        r.at += (size_t)len;
      }
      if (r.bad) return mrk_fail(MRK_E_FORMAT, "%s: segment %u: docstore runs past the file", rname.c_str(), si);
    }
    if ((rc = rt_vector(r, 8, "ram-infixes", rname.c_str(), infix))) return rc;

    std::vector<uint64_t> W;
    std::vector<uint32_t> Rw, H;
    std::vector<char> wtext;
    std::vector<uint32_t> woff;
    std::vector<uint64_t> wordids;
    uint32_t term = 0;
    if ((rc = rt_decode_segment(words.data(), words.size(), docs.data(), docs.size(), hits.data(), hits.size(), rows, word_dict, words_checkpoint, rname.c_str(), si, W, Rw,
                                H, wtext, woff, wordids, term)))
      return rc;
    // ---- the same postings in the disk format
    mrk_host_index* h = nullptr;
    rc = mrk_index_from_hits(W.data(), Rw.data(), H.data(), W.size(), term, proto.info.skiplist_block_size ? proto.info.skiplist_block_size : 128u,
                             proto.info.hit_format, &h);
    if (rc != MRK_OK) return rc;
    rt->seg.push_back(h);
    h->from_files = true;
    h->info = proto.info;
    h->info.total_docs = rows;
    h->info.n_checkpoints = 0;
    h->fields = proto.fields;
    h->attrs = proto.attrs;
    h->words = wtext;
    h->word_off = woff;
    if (!word_dict)
      for (size_t i = 0; i < h->dict.size() && i < wordids.size(); ++i) h->dict[i].wordid = wordids[i];
    h->blobs = blobs;
    h->dead = dead;
    for (size_t wi = 0; wi < dead.size(); ++wi) { // popcount over the words that are there, the tail beyond `rows` masked
      uint32_t wv = dead[wi];
      if (wi + 1 == dead.size() && (rows & 31u)) wv &= (1u << (rows & 31u)) - 1u;
      h->info.n_dead += (uint64_t)__builtin_popcount(wv);
    }
    if (rows && rowdata.size() % ((size_t)rows * 4) == 0) { // CSphRowitem rows, the schema's stride
      h->attr_stride = (uint32_t)(rowdata.size() / 4 / rows);
      h->docinfo_rows = rows;
      h->attr_rows.resize(rowdata.size() / 4);
      memcpy(h->attr_rows.data(), rowdata.data(), rowdata.size());
    }
  }
  if (r.bad) return mrk_fail(MRK_E_FORMAT, "%s: failed to parse (unexpected eof)", rname.c_str());
  return MRK_OK;
}

} // namespace

extern "C" int mrk_rt_ram_open(const char* path_prefix, mrk_rt_ram** out) {
  if (!path_prefix || !out) return mrk_fail(MRK_E_INVAL, "mrk_rt_ram_open: null argument");
  *out = nullptr;
  mrk_rt_ram* rt = new (std::nothrow) mrk_rt_ram();
  if (!rt) return mrk_fail(MRK_E_NOMEM, "out of memory");
  int rc;
  try { // sizes come from untrusted files
    rc = rt_open(path_prefix, rt);
  } catch (const std::bad_alloc&) {
    rc = mrk_fail(MRK_E_NOMEM, "%s: out of memory reading the RAM chunk", path_prefix);
  } catch (const std::exception& e) {
    rc = mrk_fail(MRK_E_FORMAT, "%s: %s", path_prefix, e.what());
  }
  if (rc != MRK_OK) {
    delete rt;
    return rc;
  }
  *out = rt;
  return MRK_OK;
}
// A LIVE RAM segment, no file in between: the three byte vectors of an RtSegment_t (sphinxrt.h:140-149: m_dWords, m_dDocs, m_dHits) as
// the running index holds them.  PerformFullTextSearch (sphinxrt.cpp:6302-6384) rebinds the one ranker to every RAM segment
// (ISphRanker::Reset, :6313-6314); a device binding keeps one mrk_segment per RtSegment_t instead -- segments are immutable once
// committed, so decode + re-encode happens once per segment, when it appears -- and the adapter's Reset() picks that segment's.
extern "C" int mrk_rt_segment_open(const mrk_rt_segment_desc* d, mrk_host_index** out) {
  if (!d || !out) return mrk_fail(MRK_E_INVAL, "mrk_rt_segment_open: null argument");
  *out = nullptr;
  if ((!d->words && d->words_len) || (!d->docs && d->docs_len) || (!d->hits && d->hits_len)) return mrk_fail(MRK_E_INVAL, "mrk_rt_segment_open: null vector");
  if (d->words_checkpoint < 2 || d->words_checkpoint > (1u << 20)) return mrk_fail(MRK_E_INVAL, "mrk_rt_segment_open: words checkpoint %u", d->words_checkpoint);
  if (d->n_fields < 1 || d->n_fields > 32) return mrk_fail(MRK_E_UNSUPPORTED, "mrk_rt_segment_open: %u fields (RT doclists carry a 32-bit field mask)", d->n_fields);
  if (d->hit_format != MRK_HITFMT_INLINE && d->hit_format != MRK_HITFMT_PLAIN) return mrk_fail(MRK_E_INVAL, "mrk_rt_segment_open: bad hit_format %u", d->hit_format);
  try {
    std::vector<uint64_t> W;
    std::vector<uint32_t> Rw, H;
    std::vector<char> wtext;
    std::vector<uint32_t> woff;
    std::vector<uint64_t> wordids;
    uint32_t term = 0;
    int rc = rt_decode_segment(d->words, (size_t)d->words_len, d->docs, (size_t)d->docs_len, d->hits, (size_t)d->hits_len, d->rows, d->word_dict != 0, d->words_checkpoint,
                               "live RAM segment", 0, W, Rw, H, wtext, woff, wordids, term);
    if (rc != MRK_OK) return rc;
    mrk_host_index* h = nullptr;
    rc = mrk_index_from_hits(W.data(), Rw.data(), H.data(), W.size(), term, d->skiplist_block_size ? d->skiplist_block_size : 128u, d->hit_format, &h);
    if (rc != MRK_OK) return rc;
    h->from_files = true;
    h->info.total_docs = d->rows;
    h->info.n_fields = d->n_fields;
    h->info.word_dict = d->word_dict ? 1u : 0u;
    h->info.skiplist_block_size = d->skiplist_block_size ? d->skiplist_block_size : 128u;
    h->info.hit_format = d->hit_format;
    for (uint32_t f = 0; f < d->n_fields; ++f) h->fields.push_back("field" + std::to_string(f)); // (the schema stays with the caller)
    h->words = wtext;
    h->word_off = woff;
    if (!d->word_dict)
      for (size_t i = 0; i < h->dict.size() && i < wordids.size(); ++i) h->dict[i].wordid = wordids[i];
    *out = h;
    return MRK_OK;
  } catch (const std::bad_alloc&) {
    return mrk_fail(MRK_E_NOMEM, "mrk_rt_segment_open: out of memory");
  } catch (const std::exception& e) {
    return mrk_fail(MRK_E_FORMAT, "mrk_rt_segment_open: %s", e.what());
  }
}

extern "C" uint32_t mrk_rt_ram_segments(const mrk_rt_ram* rt) { return rt ? (uint32_t)rt->seg.size() : 0u; }
extern "C" int mrk_rt_ram_take(mrk_rt_ram* rt, uint32_t i, mrk_host_index** out) {
  if (!rt || !out || i >= rt->seg.size() || !rt->seg[i]) return mrk_fail(MRK_E_INVAL, "mrk_rt_ram_take: no segment %u (or taken already)", i);
  *out = rt->seg[i];
  rt->seg[i] = nullptr;
  return MRK_OK;
}
extern "C" void mrk_rt_ram_free(mrk_rt_ram* rt) { delete rt; }
