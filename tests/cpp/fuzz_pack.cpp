// fuzz_pack.cpp -- the load-time walk of untrusted postings (csrc/mrk_pack.cpp: pack_term / validate_term, what mrk_segment_create
// runs over every doclist before a byte of it reaches a kernel) under AddressSanitizer + UBSan on the CPU.  A small valid index
// comes from the format writer; every iteration damages its .spd bytes (flips, cuts, insertions, extreme varints) and / or the
// dictionary entry (offsets, lengths, doc counts), copies them into an allocation of the exact size, and walks every term: each
// call must return true or false-with-a-message, and anything it packs must index inside its own arrays.  Built and run by
// tests/test_pack_fuzz.py; no GPU, no libmrk.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../manticoresearch_amd/csrc/mrk_pack.h"

int mrk_fail(int code, const char* fmt, ...) { return code; }

static uint64_t g_s = 1;
static uint64_t rnd() {
  g_s += 0x9E3779B97F4A7C15ull;
  uint64_t z = g_s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint32_t below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0u; }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  if (argc > 2) g_s = strtoull(argv[2], nullptr, 0);
  int n_true = 0, n_false = 0;
  for (int fmt = 0; fmt < 2; ++fmt) { // both hit formats (MRK_HITFMT_*)
    const double probs[6] = {0.5, 0.2, 0.05, 0.01, 0.002, 0.3};
    mrk_synth_params sp;
    memset(&sp, 0, sizeof sp);
    sp.seed = 7 + fmt, sp.n_docs = 3000, sp.term_prob = probs, sp.n_terms = 6, sp.n_fields = fmt ? 3 : 11, sp.title_frac = 0.3, sp.max_pos = 40;
    sp.skiplist_block_size = fmt ? 128 : 32, sp.hit_format = (uint32_t)fmt, sp.end_markers = 1, sp.n_threads = 1;
    mrk_host_index* h = nullptr;
    if (mrk_synth_generate(&sp, &h) != MRK_OK) return 2;
    uint64_t spd_len = 0, spp_len = 0;
    const uint8_t* spd = mrk_host_index_spd(h, &spd_len);
    mrk_host_index_spp(h, &spp_len);
    uint32_t nt = 0;
    const mrk_dict_entry* dict = mrk_host_index_dict(h, &nt);
    const bool inl = fmt == MRK_HITFMT_INLINE;
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> bytes(spd, spd + spd_len);
      const uint32_t n_dmg = below(4);
      for (uint32_t d = 0; d < n_dmg && !bytes.empty(); ++d) {
        const size_t k = below((uint32_t)bytes.size());
        switch (below(5)) {
          case 0: bytes[k] = (uint8_t)rnd(); break;
          case 1: bytes[k] = (uint8_t[]){0, 0x80, 0xFF, 0x7F, 1}[below(5)]; break;
          case 2: bytes.resize(k); break;
          case 3: bytes.insert(bytes.begin() + (long)k, (size_t)(1 + below(6)), (uint8_t)(0x80 | rnd())); break; // a run of continuation bytes
          default: for (size_t j = k; j < bytes.size() && j < k + 5; ++j) bytes[j] = 0xFF; break;                // a varint of all ones
        }
      }
      uint8_t* exact = (uint8_t*)malloc(bytes.size() ? bytes.size() : 1);
      if (!bytes.empty()) memcpy(exact, bytes.data(), bytes.size());
      for (uint32_t t = 0; t < nt; ++t) {
        mrk_dict_entry e = dict[t];
        if (below(4) == 0) {
          switch (below(5)) {
            case 0: e.doclist_off = rnd() >> below(64); break;
            case 1: e.doclist_len = rnd() >> below(64); break;
            case 2: e.docs = (uint32_t)(rnd() >> below(64)); break;
            case 3: e.doclist_off = bytes.size() - below(3); break;
            default: e.hits = (uint32_t)rnd(); break;
          }
        }
        const uint64_t total_rows = below(8) ? sp.n_docs : below(4000), spp_lim = below(8) ? spp_len : below(1000);
        std::string err;
        mrk::PackedTerm pt;
        const bool dense = below(2) != 0;
        const bool ok = mrk::pack_term(exact, bytes.size(), e, inl, dense ? total_rows : 0, pt, err, total_rows, spp_lim);
        if (ok) {
          ++n_true;
          // what the segment loader copies to the device must be self-consistent
          const size_t nb = pt.base.size();
          if (pt.doff.size() != nb || pt.w.size() != nb || pt.attr.size() != nb * 64 || (pt.hit.size() != nb * 128 && !pt.hit.empty()) || (!pt.hbase.empty() && pt.hbase.size() != nb)) {
            fprintf(stderr, "iteration %d term %u: block arrays disagree\n", it, t);
            return 3;
          }
          for (size_t b = 0; b < nb; ++b)
            if (pt.w[b] > 32 || (uint64_t)pt.doff[b] + ((uint64_t)pt.w[b] * 128 + 31) / 32 > pt.delta.size() + 4) {
              fprintf(stderr, "iteration %d term %u block %zu: delta run out of bounds (w %u doff %u of %zu)\n", it, t, b, pt.w[b], pt.doff[b], pt.delta.size());
              return 4;
            }
          if (e.docs && total_rows && pt.last_rowid >= total_rows) return 5;
          if (!pt.bm.empty() && pt.bm.size() < ((size_t)total_rows + 31) / 32) return 7;
        } else {
          ++n_false;
          if (err.empty()) {
            fprintf(stderr, "iteration %d term %u: declined without a message\n", it, t);
            return 6;
          }
        }
        std::string err2;
        (void)mrk::validate_term(exact, bytes.size(), e, inl, total_rows, spp_lim, err2);
      }
      free(exact);
    }
    mrk_host_index_free(h);
  }
  printf("packed %d declined %d\n", n_true, n_false);
  return 0;
}
