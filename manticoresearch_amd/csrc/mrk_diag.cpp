// mrk_diag.cpp -- host-side accounting for the roofline figures; nothing here is on the query path.
//
// mrk_host_index_pair_stats: for keyword pairs (a, b) the number of common docs and the distinct 128-byte lines of each
// keyword's packed tf / field words (DevSegment::pk_attr: 64 words per 128-doc block, slot r of the keyword's doc list
// lives in word (r & 63) of block r >> 7) that those docs' slots touch -- the bytes scan_bm_kernel's gathers must bring
// in from HBM, as opposed to every attr word of both doclists.
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "mrk_hostindex.h"

int mrk_fail(int code, const char* fmt, ...);

namespace {

// rowids of one doclist (DiskIndexQword_c::GetNextDoc, sphinx.cpp:511-549); false on malformed bytes
bool decode_rowids(const mrk_host_index* h, const mrk_dict_entry& e, bool inline_hits, std::vector<uint32_t>& out) {
  out.clear();
  out.reserve(e.docs);
  if (!e.docs) return true;
  if (e.doclist_off == 0 || e.doclist_off > h->spd_len || e.doclist_len > h->spd_len - e.doclist_off) return false;
  const uint8_t* p = h->spd + e.doclist_off;
  const uint8_t* end = p + e.doclist_len;
  auto vlb = [&](bool& ok) -> uint64_t {
    uint64_t v = 0;
    for (;;) {
      if (p >= end) {
        ok = false;
        return 0;
      }
      const uint32_t b = *p++;
      v = (v << 7) + (b & 0x7f);
      if (!(b & 0x80)) return v;
    }
  };
  bool ok = true;
  uint32_t rowid = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < e.docs && ok; ++i) {
    const uint32_t d = (uint32_t)vlb(ok);
    if (!d) return false;
    rowid += d;
    if (inline_hits) {
      const uint32_t hits = (uint32_t)vlb(ok);
      vlb(ok), vlb(ok);
      (void)hits;
    } else
      vlb(ok), vlb(ok), vlb(ok);
    out.push_back(rowid);
  }
  return ok;
}

} // namespace

extern "C" int mrk_host_index_pair_stats(const mrk_host_index* h, uint32_t hit_format, const uint32_t* pairs, uint32_t n_pairs,
                                         uint32_t n_threads, mrk_pair_stats* out) {
  if (!h || (!pairs && n_pairs) || (!out && n_pairs)) return mrk_fail(MRK_E_INVAL, "mrk_host_index_pair_stats: NULL argument");
  for (uint32_t i = 0; i < 2 * n_pairs; ++i)
    if (pairs[i] >= h->dict.size()) return mrk_fail(MRK_E_INVAL, "mrk_host_index_pair_stats: term %u of %zu", pairs[i], h->dict.size());
  const bool inl = hit_format == MRK_HITFMT_INLINE;
  std::atomic<uint32_t> next{0};
  std::atomic<bool> bad{false};
  unsigned nth = n_threads ? n_threads : std::max(1u, std::thread::hardware_concurrency());
  nth = std::min<unsigned>(nth, std::max<uint32_t>(n_pairs, 1u));
  auto work = [&] {
    std::vector<uint32_t> ra, rb;
    std::vector<uint8_t> ta, tb;
    for (;;) {
      const uint32_t i = next.fetch_add(1);
      if (i >= n_pairs) break;
      const mrk_dict_entry &ea = h->dict[pairs[2 * i]], &eb = h->dict[pairs[2 * i + 1]];
      if (!decode_rowids(h, ea, inl, ra) || !decode_rowids(h, eb, inl, rb)) {
        bad = true;
        continue;
      }
      ta.assign((ra.size() + 127) / 128, 0); // per block: bits 0-1 = lines of the interleaved words, bits 2-3 = lines of the slot-ordered plane
      tb.assign((rb.size() + 127) / 128, 0);
      uint64_t m = 0;
      size_t x = 0, y = 0;
      while (x < ra.size() && y < rb.size()) {
        if (ra[x] < rb[y])
          ++x;
        else if (rb[y] < ra[x])
          ++y;
        else {
          ++m;
          ta[x >> 7] |= (uint8_t)((1u << ((x & 63) >> 5)) | (4u << ((x & 127) >> 6)));
          tb[y >> 7] |= (uint8_t)((1u << ((y & 63) >> 5)) | (4u << ((y & 127) >> 6)));
          ++x, ++y;
        }
      }
      mrk_pair_stats& o = out[i];
      o.matches = m;
      o.docs_a = ra.size(), o.docs_b = rb.size();
      o.lines128_a = o.lines128_b = o.blocks_a = o.blocks_b = o.lines128s_a = o.lines128s_b = 0;
      for (uint8_t v : ta) o.lines128_a += (v & 1u) + ((v >> 1) & 1u), o.lines128s_a += ((v >> 2) & 1u) + ((v >> 3) & 1u), o.blocks_a += v != 0;
      for (uint8_t v : tb) o.lines128_b += (v & 1u) + ((v >> 1) & 1u), o.lines128s_b += ((v >> 2) & 1u) + ((v >> 3) & 1u), o.blocks_b += v != 0;
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nth; ++t) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  if (bad) return mrk_fail(MRK_E_FORMAT, "mrk_host_index_pair_stats: malformed doclist");
  return MRK_OK;
}
