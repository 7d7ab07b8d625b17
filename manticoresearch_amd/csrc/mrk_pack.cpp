// mrk_pack.cpp -- lossless load-time transcode of a reference-format doclist (.spd, VLB
// varints: DiskIndexQword_c::ReadNext, sphinx.cpp:511-549) into the device's packed blocks.
//
// Per block of 128 docs (the last one may be short):
//   base   first possible rowid = rowid of the previous block's last doc + 1 (0 for block 0);
//          same meaning as SkiplistEntry_t::m_tBaseRowIDPlus1
//   offs   o[i] = rowid[i] - base, packed with w = bits(o[n-1]) bits each in lane order: value
//          slot 2*l + r holds doc l + 64*r, so lane l finds both of its docs in one 2w-bit field
//          and no prefix sum is needed to get a rowid; w > 16 => two planes of 32-bit offsets
//          (offsets cost ~4 more bits per doc than deltas; the scan kernel is issue-bound, not
//          HBM-bound, and two wave-wide prefix sums per block cost more than those bits)
//   attrs  one word per lane: tf[l] | tf[l+64] << 8 | fields[l] << 16 | fields[l+64] << 24
//          (tf >= 255 is stored as 255 and listed in the term's exception array; needs <= 8 fields)
// rowid[i] = base + o[i].  Everything the BM25 / NONE rankers read survives
// exactly: rowid, hit count, low field-mask bits.
#include "mrk_pack.h"

#include <string.h>

namespace mrk {

namespace {
struct Rd {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint64_t vlb() {
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
  }
};
} // namespace

// The walk DiskIndexQword_c::GetNextDoc would do over the whole doclist (sphinx.cpp:511-549), checking what a kernel
// later relies on; independent of whether the doclist can be packed.
bool validate_term(const uint8_t* spd, uint64_t spd_len, const mrk_dict_entry& e, bool inline_hits, uint64_t total_rows,
                   uint64_t spp_len, std::string& err) {
  if (!e.docs) return true;
  if (e.doclist_off == 0 || e.doclist_off > spd_len || e.doclist_len > spd_len - e.doclist_off) {
    err = "corrupt: doclist outside .spd";
    return false;
  }
  if ((uint64_t)e.docs * 3 > e.doclist_len || (total_rows && e.docs > total_rows)) {
    err = "corrupt: the dictionary's doc count does not fit the doclist";
    return false;
  }
  Rd rd{spd + e.doclist_off, spd + e.doclist_off + e.doclist_len};
  uint32_t rowid = 0xFFFFFFFFu;
  uint64_t hit_position = 0;
  for (uint32_t i = 0; i < e.docs; ++i) {
    const uint32_t delta = (uint32_t)rd.vlb();
    if (!rd.ok || delta == 0) {
      err = "corrupt: doclist shorter than the dictionary's doc count";
      return false;
    }
    const uint32_t prev = rowid;
    rowid += delta;
    if (prev != 0xFFFFFFFFu && rowid <= prev) {
      err = "corrupt: rowids do not ascend";
      return false;
    }
    if (total_rows && rowid >= total_rows) {
      err = "corrupt: rowid beyond the segment's row count";
      return false;
    }
    bool has_list = true;
    if (inline_hits) {
      const uint32_t hits = (uint32_t)rd.vlb();
      rd.vlb();
      if (hits == 1) {
        rd.vlb();
        has_list = false;
      } else
        hit_position += rd.vlb();
    } else {
      hit_position += rd.vlb();
      rd.vlb(), rd.vlb();
    }
    if (!rd.ok) {
      err = "corrupt: truncated doclist entry";
      return false;
    }
    if (has_list && spp_len && hit_position >= spp_len) {
      err = "corrupt: hitlist offset past .spp";
      return false;
    }
  }
  if (rd.vlb() != 0 || !rd.ok) {
    err = "corrupt: doclist longer than the dictionary's doc count";
    return false;
  }
  return true;
}

bool pack_term(const uint8_t* spd, uint64_t spd_len, const mrk_dict_entry& e, bool inline_hits, uint64_t bitmap_rows,
               PackedTerm& out, std::string& err, uint64_t total_rows, uint64_t spp_len) {
  out = PackedTerm();
  if (!e.docs) return true;
  if (e.doclist_off == 0 || e.doclist_off > spd_len || e.doclist_len > spd_len - e.doclist_off) {
    err = "corrupt: doclist outside .spd";
    return false;
  }
  // (before anything is sized by the doc count: an entry takes at least 3 bytes in either format, and rowids are distinct)
  if ((uint64_t)e.docs * 3 > e.doclist_len || (total_rows && e.docs > total_rows)) {
    err = "corrupt: the dictionary's doc count does not fit the doclist";
    return false;
  }
  Rd rd{spd + e.doclist_off, spd + e.doclist_off + e.doclist_len};
  const uint32_t nd_total = e.docs;
  const uint32_t nblk = (nd_total + 127) / 128;
  out.base.reserve(nblk);
  out.doff.reserve(nblk);
  out.w.reserve(nblk);
  out.attr.assign((size_t)nblk * 64, 0);
  out.hit.assign((size_t)nblk * 128, 0);
  out.attr1.assign((size_t)nblk * 128, 0);
  if (bitmap_rows) out.attr2.assign((size_t)nblk * 128, 0);
  out.hbase.reserve(nblk);
  uint64_t hit_position = 0; // m_uHitPosition / m_iHitlistPos (sphinx.cpp:534, 542)
  out.delta.reserve((size_t)nblk * 32 + 8);

  bool want_bm = bitmap_rows != 0;
  if (want_bm) {
    const uint64_t windows = (bitmap_rows + BM_WINDOW - 1) / BM_WINDOW;
    out.bm.assign((size_t)windows * (BM_WINDOW / 32), 0u);
  }
  uint32_t rowid = 0xFFFFFFFFu; // decoder starts at INVALID_ROWID (sphinx.cpp:12947)
  uint32_t d[128], tf[128], fl[128], rows[128];
  for (uint32_t b = 0; b < nblk; ++b) {
    const uint32_t n = nd_total - b * 128 < 128 ? nd_total - b * 128 : 128;
    const uint32_t base = rowid + 1u;
    uint32_t dmax = 0;
    // hit references are relative to the block's first hitlist (a term's first delta is an absolute .spp
    // position, sphinx.cpp:534/542, so the running position at the block start may be far below it)
    uint64_t hb = hit_position;
    bool hb_set = false;
    out.hbase.push_back(0);
    uint32_t* hitref = out.hit.data() + (size_t)b * 128;
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t delta = (uint32_t)rd.vlb();
      if (!rd.ok || delta == 0) {
        err = "corrupt: doclist shorter than the dictionary's doc count";
        return false;
      }
      const uint32_t prev = rowid;
      rowid += delta;
      if (prev != 0xFFFFFFFFu && rowid <= prev) {
        err = "corrupt: rowids do not ascend";
        return false;
      }
      if (total_rows && rowid >= total_rows) { // the dead-row map, the attribute rows and the bitmaps are sized by the row count
        err = "corrupt: rowid beyond the segment's row count";
        return false;
      }
      uint32_t hits, fields;
      if (inline_hits) {
        hits = (uint32_t)rd.vlb();
        const uint32_t first = (uint32_t)rd.vlb();
        if (hits == 1) {
          const uint32_t fe = (uint32_t)rd.vlb();
          const uint32_t f = (fe >> 1) & 255u;
          fields = f < 32 ? (1u << f) : 0u;
          hitref[i] = first | (fe << 23); // the inlined hit, as SeekHitlist keeps it (sphinx.cpp:526, 464)
        } else {
          fields = first;
          hit_position += rd.vlb(); // hitlist offset delta
          if (!hb_set) hb = hit_position, hb_set = true;
          if (hit_position - hb > 0xFFFFFFFFull) {
            err = "hitlists of one block span more than 4 GiB";
            return false;
          }
          hitref[i] = (uint32_t)(hit_position - hb);
          if (spp_len && hit_position >= spp_len) {
            err = "corrupt: hitlist offset past .spp";
            return false;
          }
        }
      } else {
        hit_position += rd.vlb();
        if (!hb_set) hb = hit_position, hb_set = true;
        if (hit_position - hb > 0xFFFFFFFFull) {
          err = "hitlists of one block span more than 4 GiB";
          return false;
        }
        hitref[i] = (uint32_t)(hit_position - hb);
        if (spp_len && hit_position >= spp_len) {
          err = "corrupt: hitlist offset past .spp";
          return false;
        }
        fields = (uint32_t)rd.vlb();
        hits = (uint32_t)rd.vlb();
      }
      if (!rd.ok) {
        err = "corrupt: truncated doclist entry";
        return false;
      }
      if (fields > 0xFFu) {
        err = "field mask wider than 8 bits";
        return false;
      }
      rows[i] = rowid;
      if (want_bm) {
        if (rowid < bitmap_rows)
          out.bm[rowid >> 5] |= 1u << (rowid & 31u);
        else
          want_bm = false; // a rowid beyond the segment's row count: no bitmap for this term
      }
      d[i] = rowid - base;
      if (d[i] > dmax) dmax = d[i];
      tf[i] = hits;
      fl[i] = fields;
      if (hits >= 255) out.exc.push_back(((uint64_t)rowid << 32) | hits);
      if (fields > 15u) out.attr1_ok = false;
      out.attr1[(size_t)b * 128 + i] = (uint8_t)((hits < 15u ? hits : 15u) | ((fields & 15u) << 4));
      if (bitmap_rows) out.attr2[(size_t)b * 128 + i] = (uint16_t)((hits < 255u ? hits : 255u) | ((fields & 0xffu) << 8));
    }
    out.hbase[b] = hb;
    for (uint32_t i = n; i < 128; ++i) d[i] = 0, tf[i] = 0, fl[i] = 0;
    uint32_t w = 0;
    while (w < 32 && (dmax >> w)) ++w;
    out.base.push_back(base);
    out.doff.push_back((uint32_t)out.delta.size());
    uint32_t* attr = out.attr.data() + (size_t)b * 64;
    for (uint32_t l = 0; l < 64; ++l) {
      const uint32_t t0 = tf[l] < 255 ? tf[l] : 255, t1 = tf[l + 64] < 255 ? tf[l + 64] : 255;
      attr[l] = t0 | (t1 << 8) | (fl[l] << 16) | (fl[l + 64] << 24);
    }
    if (w <= 16) {
      out.w.push_back((uint8_t)w);
      const uint32_t words = 4 * w + 1; // 128*w bits + one spare word for the 2-word window read
      const size_t at = out.delta.size();
      out.delta.resize(at + words, 0);
      uint32_t* dst = out.delta.data() + at;
      for (uint32_t l = 0; l < 64; ++l)
        for (uint32_t r = 0; r < 2; ++r) {
          const uint64_t bit = (uint64_t)(2 * l + r) * w;
          const uint64_t v = (uint64_t)d[l + 64 * r] << (bit & 31);
          dst[bit >> 5] |= (uint32_t)v;
          if (v >> 32) dst[(bit >> 5) + 1] |= (uint32_t)(v >> 32);
        }
    } else {
      out.w.push_back((uint8_t)0xFF);
      const size_t at = out.delta.size();
      out.delta.resize(at + 128, 0);
      memcpy(out.delta.data() + at, d, 128 * 4);
    }
  }
  if (rd.vlb() != 0 || !rd.ok) {
    err = "corrupt: doclist longer than the dictionary's doc count";
    return false;
  }
  out.packed_bytes = out.delta.size() * 4 + out.attr.size() * 4 + (uint64_t)nblk * 9;
  out.last_rowid = rowid;
  if (!want_bm) {
    out.bm.clear();
    out.attr2.clear();
  } else {
    const size_t groups = out.bm.size() / (BM_GROUP / 32);
    out.bm_dir.resize(groups + 1);
    uint32_t run = 0;
    for (size_t g = 0; g < groups; ++g) {
      out.bm_dir[g] = run;
      for (uint32_t i = 0; i < BM_GROUP / 32; ++i) run += (uint32_t)__builtin_popcount(out.bm[g * (BM_GROUP / 32) + i]);
    }
    out.bm_dir[groups] = run;
  }
  return true;
}

} // namespace mrk
