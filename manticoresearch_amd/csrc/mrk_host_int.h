// mrk_host_int.h -- host-side objects shared by mrk_host.cpp (segments, batches, C-ABI) and mrk_plan.cpp (query planner).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <vector>

#include "mrk_dev.h"

int mrk_fail(int code, const char* fmt, ...);

using mrk::DevItem;
using mrk::DevQuery;
using mrk::DevSegment;
using mrk::DevTerm;

struct mrk_worker; // the context's submission thread (mrk_host.cpp)
struct mrk_comm;   // the context's RCCL communicator for the shard exchange (mrk_comm.cpp)

struct mrk_ctx {
  std::atomic<int> n_segments{0}, n_batches{0}; // alive through the C-ABI: mrk_ctx_destroy refuses while any is
  mrk_worker* worker = nullptr;
  mrk_comm* comm = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t merge_stream = nullptr; // mrk_topk_merge: never queued behind the scans of a following batch
  hipEvent_t merge_done[MRK_MERGE_SLOTS] = {};
  bool merge_used[MRK_MERGE_SLOTS] = {};
  int64_t item_bytes = 128 << 10; // target doclist bytes per work item
  int path = 0;                   // 0 = packed doclists when the segment has them, 1 = VLB (.spd) direct, 2 = packed only
  int pack = 1;                   // build packed doclists at segment load
  int bitmap_inv = 64;            // terms in >= 1/bitmap_inv of the docs also get a bitmap (0 = never)
  int attr_seq = 1;               // keywords with a bitmap also get their tf / field bytes in slot order (what the bitmap kernel gathers from)
  int attr_nibbles = 0;           // also build the one-byte tf/field plane the bitmap kernel can gather from (<= 4 fields)
  int bm_target_items = 1 << 20;  // two-bitmap AND kernel: cap of the work items per launch ...
  int bm_min_windows = 128;       // ... and the least windows per work item (a wave's fixed costs show on short runs)
  int pk_min_items = 2048;        // block-scan kernel: a batch with fewer work items has its block ranges cut finer (>= one block per wave)
  int exchange_part = 1;          // mrk_shard_exchange partitions the merge by query (all-to-all of row slices); 0 = all-gather, every rank merges everything
  int prox_prune = 1;             // proximity rankers: matches whose weight upper bound cannot reach the top K skip the hit pass (counted, not ranked)
  int exchange_self_rccl = 0;     // one rank: still send the rows to itself through RCCL (rehearsal of the collective path)
  int item_order = 7;             // work items of different queries interleaved (piece-major): 1 = block scan, 2 = bitmap AND, 4 = bitmap trees; 0 = query-major
  int bt_target_items = 6144;     // ... and the tree kernel over bitmap words
  int prox_bound_keywords = 0;    // 1: equal positions of different keywords sort by query position in this context's indexes (see mrk.h) -- the tighter weight bound is sound
  int bt_phrase = 1;              // root PHRASE / PROXIMITY of common words: the AND of the words runs on bitmap words too (0 = block walk)
  int bt_cover_inv = 1024;        // trees whose candidate cover holds >= 1/bt_cover_inv of the docs run on bitmap words (0 = never; 32 until the step grew to 8192 rowids: config 3 5.4 -> 4.8 ms)
  int gen_lane_hits = 256;        // generic evaluator: hits (16 B) of per-lane list memory, GEN_GRID * 256 lanes
  int gen_spill_mb = 1024;        // ... and the shared area for lists beyond a lane's slice (exhausted: the query fails loudly)
  int mq_max_chunks = 1 << 22;    // cap of a batch's match queue, in 64-entry chunks of 1792 B (a fuller queue flags its queries: rerun alone)
};

struct HostTerm {
  uint64_t doclist_off = 0, doclist_len = 0;
  uint64_t packed_bytes = 0;
  uint32_t blk_first = 0, nblocks = 0, docs = 0, hits = 0;
  uint32_t exc_first = 0, exc_n = 0;
  uint32_t last_rowid = 0; // rowid of the term's last doc (packed segments)
  uint64_t bm_off = ~0ull, dir_off = ~0ull; // word offsets of the term's bitmap / rank directory, ~0 = none
};

struct mrk_segment {
  mrk_ctx* ctx = nullptr;
  DevSegment dev{};
  std::vector<HostTerm> terms;
  uint64_t total_docs = 0;
  uint32_t n_fields = 0;
  uint64_t device_bytes = 0;
  void* d_spd = nullptr;
  void* d_spp = nullptr;
  void* d_blk_base = nullptr;
  void* d_blk_off = nullptr;
  void* d_blk_hit = nullptr;
  bool has_packed = false;
  void* d_pk_base = nullptr;
  void* d_pk_doff = nullptr;
  void* d_pk_w = nullptr;
  void* d_pk_delta = nullptr;
  void* d_pk_attr = nullptr;
  void* d_pk_exc = nullptr;
  void* d_pk_hit = nullptr;
  void* d_pk_hbase = nullptr;
  void* d_pk_attr1 = nullptr;
  void* d_pk_attr2 = nullptr;
  void* d_dead = nullptr;
  void* d_attrs = nullptr; // .spa rows (mrk_segment_set_attrs)
  void* d_blobs = nullptr; // blob pool (mrk_segment_set_blobs)
  uint32_t n_blob_attrs = 0;
  uint64_t attr_rows = 0;
  void* d_bm = nullptr;
  void* d_bm_dir = nullptr;
};

// mrk_comm.cpp (run on the submission thread)
int mrk_comm_unique_id_impl(uint8_t* id_out);
int mrk_comm_init_impl(mrk_ctx* ctx, const uint8_t* id_bytes, int n_ranks, int rank);
void mrk_comm_destroy_impl(mrk_ctx* ctx);
int mrk_comm_allreduce_i64_impl(mrk_ctx* ctx, int64_t* values, uint64_t n);
int mrk_comm_exchange_impl(mrk_ctx* ctx, const uint64_t* rows, uint32_t n_queries, hipEvent_t after, uint32_t slot, const uint64_t** rows_all_out,
                           hipEvent_t* gathered_event_out);
int mrk_comm_ranks(const mrk_ctx* ctx);
int mrk_comm_rank(const mrk_ctx* ctx);
void mrk_shard_slice_impl(uint32_t n_queries, int n_ranks, int rank, uint32_t* per_out, uint32_t* first_out, uint32_t* count_out);
bool mrk_comm_can_partition(mrk_ctx* ctx);
int mrk_comm_exchange_part_impl(mrk_ctx* ctx, const uint64_t* rows, uint32_t n_queries, hipEvent_t after, uint32_t slot, const uint64_t** recv_out,
                                hipEvent_t* gathered_event_out, uint32_t* per_out, uint32_t* first_out, uint32_t* count_out);
int mrk_comm_flags_begin(mrk_ctx* ctx, uint32_t slot, uint32_t** flags_dev_out);
int mrk_comm_flags_finish(mrk_ctx* ctx, uint32_t slot);
int mrk_comm_flags_read(mrk_ctx* ctx, uint32_t slot, uint32_t* rerun_any, uint32_t* declined_any);
hipEvent_t mrk_comm_rows_ready_event(mrk_ctx* ctx);

namespace mrk {

// Plans one query of a batch: validates it, builds the reference-shaped evaluation tree, computes IDFs, pruning
// histogram geometry and candidate capacity, and emits the query's passes and work items.  Returns MRK_OK, or
// MRK_E_UNSUPPORTED / MRK_E_INVAL with the message set.  dq = the query's head pass (index qi); further passes go to
// `extra` and get pass indices n_queries + position; bitmap-kernel work goes to items_bm as one whole-range entry.
int plan_query(const mrk_segment* seg, const mrk_query& q, int64_t item_bytes, bool use_packed, DevQuery& dq,
               std::vector<DevQuery>& extra, uint32_t n_queries, std::vector<DevItem>& items, std::vector<DevItem>& items_bm,
               uint32_t qi, uint64_t& algo_bytes, uint64_t& dev_bytes, uint64_t& cand_total, bool& prox_out, bool& tree_out,
               std::vector<mrk::GenProg>& gen_progs, uint32_t rowid_max = 0xFFFFFFFFu); // gen_progs: programs of the generic evaluator (DevQuery::gen_prog indexes it); their
                                                      // work items go to items_bm with kind 2, already cut

} // namespace mrk
