// mrk_hostindex.h -- the host-side index object behind mrk_host_index*: the three posting files as byte buffers,
// the dictionary as a flat table, and (for indexes read from disk, mrk_files.cpp) what the .sph header said.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/mrk.h"

struct mrk_host_index {
  uint8_t *spd = nullptr, *spp = nullptr, *spe = nullptr; // malloc'd, 64 zero bytes of slack each
  uint64_t spd_len = 0, spp_len = 0, spe_len = 0;
  std::vector<mrk_dict_entry> dict;
  // indexes opened from files only
  bool from_files = false;
  mrk_index_info info = {};
  std::vector<std::string> fields;  // schema full-text field names
  std::vector<char> words;          // dict=keywords: the keywords back to back, NUL terminated
  std::vector<uint32_t> word_off;   // per term: offset into words
  std::vector<uint32_t> dead;       // .spm: one bit per row
  struct Attr {
    std::string name;
    uint32_t type;
    int32_t bit_offset, bit_count;
  };
  std::vector<Attr> attrs;          // schema attributes, header order
  std::vector<uint32_t> attr_rows;  // .spa: docinfo_rows x attr_stride dwords
  uint32_t attr_stride = 0;
  uint64_t docinfo_rows = 0;
  std::vector<uint8_t> blobs;       // blob pool: the .spb file as it is / an RT segment's m_dBlobs
  bool stored_fields = false;       // some field is FIELD_STORED (a docstore rides inside RT RAM segments)
  uint64_t payload_fields = 0;      // bit f: schema field f is a payload field (bit 63: some field >= 63)
  ~mrk_host_index() {
    free(spd);
    free(spp);
    free(spe);
  }
};
