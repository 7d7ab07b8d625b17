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
  std::vector<uint32_t> hit;   // 128 per block: the inlined Hitpos_t (inline format, tf == 1) or the doc's
                               // hitlist offset in .spp relative to hbase[block]
  std::vector<uint64_t> hbase; // per block: SkiplistEntry_t::m_iBaseHitlistPos
  uint64_t packed_bytes = 0;   // bytes a scan of the whole term reads (deltas + attrs + block index)
};

// returns false and sets err on malformed input
bool pack_term(const uint8_t* spd, uint64_t spd_len, const mrk_dict_entry& e, bool inline_hits, PackedTerm& out,
               std::string& err);

} // namespace mrk
