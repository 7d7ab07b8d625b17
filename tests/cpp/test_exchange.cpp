// The shard exchange from a plain C++ host -- no Python, no torch: mrk_comm_unique_id / mrk_comm_init (one rank: the only
// world a one-GPU box can form; the calls are the N-rank ones), mrk_comm_allreduce_i64 for the document frequencies,
// then per batch mrk_batch_set_rows_dst + mrk_shard_exchange + mrk_merge_wait.  The merged rows must say what the batch
// itself reports.  Usage: test_exchange <n_docs>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/mrk.h"

#define CK(x)                                                                  \
  do {                                                                         \
    if ((x) != MRK_OK) return fprintf(stderr, "%s: %s\n", #x, mrk_last_error()), 2; \
  } while (0)

int main(int argc, char** argv) {
  const uint64_t n_docs = argc > 1 ? strtoull(argv[1], 0, 10) : 300000;
  const double probs[3] = {0.25, 0.1, 0.02};
  mrk_ctx* ctx = nullptr;
  CK(mrk_ctx_create(0, &ctx));
  uint8_t id[MRK_COMM_ID_BYTES];
  CK(mrk_comm_unique_id(id));
  CK(mrk_comm_init(ctx, id, 1, 0));
  mrk_synth_params p{};
  p.seed = 5, p.n_docs = n_docs, p.rowid_base = 7000, p.term_prob = probs, p.n_terms = 3, p.n_fields = 2, p.title_frac = 0.1, p.max_pos = 100;
  p.skiplist_block_size = 128, p.hit_format = MRK_HITFMT_INLINE, p.n_threads = 4;
  mrk_host_index* hi = nullptr;
  CK(mrk_synth_generate(&p, &hi));
  mrk_segment_desc d{};
  d.spd = mrk_host_index_spd(hi, &d.spd_len), d.spp = mrk_host_index_spp(hi, &d.spp_len), d.spe = mrk_host_index_spe(hi, &d.spe_len);
  d.dict = mrk_host_index_dict(hi, &d.n_terms);
  d.total_docs = n_docs, d.skiplist_block_size = 128, d.hit_format = MRK_HITFMT_INLINE, d.n_fields = 2, d.rowid_base = 7000;
  mrk_segment* seg = nullptr;
  CK(mrk_segment_create(ctx, &d, &seg));
  // local_df: per-keyword document counts and N summed over the shards (one shard here: the sums are its own numbers)
  int64_t df[4] = {d.dict[0].docs, d.dict[1].docs, d.dict[2].docs, (int64_t)n_docs};
  CK(mrk_comm_allreduce_i64(ctx, df, 4));
  if (df[0] != d.dict[0].docs || df[3] != (int64_t)n_docs) return fprintf(stderr, "allreduce changed a one-rank sum\n"), 1;

  const int NQ = 3;
  mrk_node nodes[NQ][3];
  int32_t children[2] = {1, 2};
  int64_t local_docs[NQ][3];
  mrk_query q[NQ];
  const int pairs[NQ][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int i = 0; i < NQ; ++i) {
    memset(nodes[i], 0, sizeof nodes[i]);
    nodes[i][0].op = MRK_OP_AND, nodes[i][0].n_children = 2, nodes[i][0].term_id = -1, nodes[i][0].field_mask = MRK_ALL_FIELDS, nodes[i][0].boost = 1.f;
    for (int k = 0; k < 2; ++k) {
      nodes[i][1 + k].op = MRK_OP_TERM, nodes[i][1 + k].term_id = pairs[i][k], nodes[i][1 + k].atom_pos = k + 1;
      nodes[i][1 + k].field_mask = MRK_ALL_FIELDS, nodes[i][1 + k].boost = 1.f;
      local_docs[i][1 + k] = df[pairs[i][k]];
    }
    local_docs[i][0] = -1;
    memset(&q[i], 0, sizeof q[i]);
    q[i].nodes = nodes[i], q[i].n_nodes = 3, q[i].children = children, q[i].root = 0, q[i].ranker = MRK_RANK_BM25, q[i].max_matches = 1000;
    q[i].normalized_tfidf = 1, q[i].total_docs_override = df[3], q[i].local_docs = local_docs[i];
  }
  mrk_batch* b = nullptr;
  CK(mrk_batch_create(ctx, NQ, &b));
  uint64_t *rows = nullptr, *merged = nullptr;
  if (hipMalloc((void**)&rows, (size_t)NQ * MRK_ROW_WORDS * 8) != hipSuccess || hipHostMalloc((void**)&merged, (size_t)NQ * MRK_ROW_WORDS * 8, 0) != hipSuccess)
    return fprintf(stderr, "hipMalloc failed\n"), 2;
  CK(mrk_batch_set_rows_dst(b, rows));
  for (int round = 0; round < 3; ++round) {
    CK(mrk_batch_submit(b, seg, q, NQ));
    CK(mrk_shard_exchange(ctx, b, rows, NQ, 1000, merged, 0)); // returns at once
    CK(mrk_merge_wait(ctx, 0));
    CK(mrk_batch_wait(b));
    for (int i = 0; i < NQ; ++i) {
      mrk_result r;
      CK(mrk_batch_result(b, (uint32_t)i, &r));
      const uint64_t* row = merged + (size_t)i * MRK_ROW_WORDS;
      if ((int64_t)row[MRK_MAX_K + 1] != r.total_found || (int)row[MRK_MAX_K] != r.n) return fprintf(stderr, "query %d: counts differ\n", i), 1;
      for (int j = 0; j < r.n; ++j) {
        const uint32_t docid = ~(uint32_t)row[j];
        const int32_t weight = (int32_t)((uint32_t)(row[j] >> 32) ^ 0x80000000u);
        if (docid != r.rowid[j] + 7000 || weight != r.weight[j]) return fprintf(stderr, "query %d match %d differs\n", i, j), 1;
      }
    }
  }
  (void)hipFree(rows);
  (void)hipHostFree(merged);
  mrk_batch_destroy(b);
  mrk_segment_destroy(seg);
  mrk_host_index_free(hi);
  mrk_ctx_destroy(ctx); // tears the communicator down too
  printf("exchange ok: %d queries x 3 rounds through RCCL (1 rank) + device merge\n", NQ);
  return 0;
}
