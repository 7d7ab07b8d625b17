// mrk_kmq.h -- the HBM match queue between the scan kernels and rank_kernel: chunk reservation and chunk writes.
//
// Returning atomics on ONE address run at about 70 ns each on MI355X (they execute at the memory side; measured: 707 K
// chunk reservations over 8 counters cost 6 ms of a 9 ms launch), so the allocator is sharded 64 ways (workgroup b
// uses shard b % 64) and every wave reserves MQ_BATCH chunks at a time: a launch that queues 45 M docs makes 177 K
// atomics over 64 addresses instead of 707 K over 8.
#pragma once
#include "mrk_kcommon.h"

namespace mrk {

struct MqWriter {
  uint32_t next = 0, left = 0; // the wave's reservation: chunks [next, next + left)
};

// the chunk the wave's next (up to) 64 entries go to, or ~0 when the queue is full (the caller flags QF_OVERFLOW)
__device__ __forceinline__ uint32_t mq_take(const MatchQueue& MQ, MqWriter& W) {
  if (!W.left) {
    // own shard first, then whichever still has room: together the shards hold every chunk the batch can produce, but a
    // single busy workgroup may need more than its own shard's share.  A shard that is known to be full is passed over with a
    // plain load: once the whole queue is full every further reservation used to cost 64 returning atomics (a config-5 launch
    // of 1024 queries that outgrew its queue took 0.9 s in the scan alone).
    for (uint32_t k = 0; k < (uint32_t)MQ_SHARDS && !W.left; ++k) {
      const uint32_t shard = (blockIdx.x + k) & (MQ_SHARDS - 1);
      if (__hip_atomic_load(MQ.count + shard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= MQ.cap) continue;
      uint32_t got = 0;
      if (lane_id() == 0) got = atomicAdd(MQ.count + shard, (uint32_t)MQ_BATCH);
      got = rdlane(got, 0);
      if (got < MQ.cap) { // (a batch that straddles the shard's end keeps its chunks below the end)
        W.next = shard * MQ.cap + got;
        W.left = MQ.cap - got < (uint32_t)MQ_BATCH ? MQ.cap - got : (uint32_t)MQ_BATCH;
      }
    }
    if (!W.left) return 0xFFFFFFFFu;
  }
  --W.left;
  return W.next++;
}

// reserved but never filled: hand the chunks back as empty ones (every chunk below a shard's count carries a header)
__device__ __forceinline__ void mq_close(const MatchQueue& MQ, MqWriter& W, uint32_t pass) {
  if (lane_id() < W.left) MQ.hdr[W.next + lane_id()] = pass; // 0 entries
  W.left = 0;
}

} // namespace mrk
