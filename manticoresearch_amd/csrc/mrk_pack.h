// mrk_pack.h -- load-time transcode of one term's VLB doclist into packed 128-doc blocks.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mrk.h"

namespace mrk {

struct PackedTerm {
  std::vector<uint32_t> base;  // per block
  std::vector<uint32_t> doff;  // per block, word offset relative to this term's delta run
  std::vector<uint8_t> w;      // per block
  std::vector<uint32_t> delta; // this term's delta run
  std::vector<uint32_t> attr;  // 64 words per block
  std::vector<uint64_t> exc;   // rowid<<32 | tf for tf >= 255
  // one byte per doc slot for the bitmap kernel's gathers (half the bytes of the attr words): tf in the low nibble
  // (15 = look at the attr word), fields in the high one; attr1_ok = false when a doc has a field bit >= 4
  std::vector<uint8_t> attr1;
  // dense terms only (bitmap_rows != 0): tf | fields << 8 per doc slot IN SLOT ORDER (the attr words above interleave slots l and l + 64
  // for the block decoder; the bitmap kernel gathers by rank, and a line whose docs are scored rounds apart is fetched twice)
  std::vector<uint16_t> attr2;
  bool attr1_ok = true;
  uint32_t last_rowid = 0; // rowid of the term's last doc
  std::vector<uint32_t> hit;   // 128 per block: the inlined Hitpos_t (inline format, tf == 1) or the doc's
                               // hitlist offset in .spp relative to hbase[block]
  std::vector<uint64_t> hbase; // per block: .spp position of the block's first hitlist
  uint64_t packed_bytes = 0;   // bytes a scan of the whole term reads (deltas + attrs + block index)
  // dense terms only (bitmap_rows != 0): the doc set as a bitmap over [0, bitmap_rows), one 2048-rowid
  // window = 64 words, plus a rank directory (docs before each group of 8 words = 256 rowids).
  // A doc's rank is its slot in the packed arrays above (block = rank >> 7, slot = rank & 127).
  std::vector<uint32_t> bm;
  std::vector<uint32_t> bm_dir;
};

constexpr uint32_t BM_WINDOW = 2048; // rowids per bitmap window (64 lanes x 32 bits)
constexpr uint32_t BM_GROUP = 256;   // rowids per rank-directory entry

// returns false and sets err on input it cannot pack; err starts with "corrupt:" when the bytes are malformed (truncated
// entries, descending rowids, rowids >= total_rows, hitlist offsets >= spp_len; the last two only when the limit is
// given) -- as opposed to well-formed input beyond the packed format (field masks wider than 8 bits)
// bitmap_rows: 0 = no bitmap; else the segment's row count (every rowid of the term must be below it)
bool pack_term(const uint8_t* spd, uint64_t spd_len, const mrk_dict_entry& e, bool inline_hits, uint64_t bitmap_rows,
               PackedTerm& out, std::string& err, uint64_t total_rows = 0, uint64_t spp_len = 0);

// Validate-only walk of one term's doclist: every entry decodes, rowids ascend and stay below total_rows, hitlist
// offsets stay below spp_len, the terminator sits where the dictionary's doc count says.  false + err ("corrupt: ...")
// otherwise.  mrk_segment_create runs it on every doclist pack_term did not walk to the end (segments with > 8 fields,
// pack = 0, a packing decline half way): nothing unvalidated reaches a kernel.
bool validate_term(const uint8_t* spd, uint64_t spd_len, const mrk_dict_entry& e, bool inline_hits, uint64_t total_rows,
                   uint64_t spp_len, std::string& err);

} // namespace mrk
