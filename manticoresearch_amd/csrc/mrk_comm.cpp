// mrk_comm.cpp -- the one exchange step of the sharded path behind the C-ABI: RCCL all-gather of the shards' result rows
// over xGMI + the device merge, and the all-reduce of document frequencies (local_df), so that a C++ host needs no Python.
// Replaces, for one node, what SearchHandler_c::SetupLocalDF (searchd.cpp:5869-5990) and the per-query merge of chunk
// sorters (CSphMatchQueue::MoveTo, sphinxsort.cpp:681-710; sphinxrt.cpp:5945-5950) do across local indexes.
//
// librccl is loaded at run time (dlopen) the first time a communicator is asked for: libmrk.so itself carries no RCCL
// dependency, single-GPU users never touch it, and a host that already holds an RCCL (PyTorch's) shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <new>
#include <vector>

#include "mrk_host_int.h"

using namespace mrk;

namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (api.lib) {
      *(void**)&api.GetUniqueId = dlsym(api.lib, "ncclGetUniqueId");
      *(void**)&api.CommInitRank = dlsym(api.lib, "ncclCommInitRank");
      *(void**)&api.CommDestroy = dlsym(api.lib, "ncclCommDestroy");
      *(void**)&api.AllGather = dlsym(api.lib, "ncclAllGather");
      *(void**)&api.AllReduce = dlsym(api.lib, "ncclAllReduce");
      *(void**)&api.GetErrorString = dlsym(api.lib, "ncclGetErrorString");
      *(void**)&api.Send = dlsym(api.lib, "ncclSend"); // (the partitioned exchange; without them the all-gather form remains)
      *(void**)&api.Recv = dlsym(api.lib, "ncclRecv");
      *(void**)&api.GroupStart = dlsym(api.lib, "ncclGroupStart");
      *(void**)&api.GroupEnd = dlsym(api.lib, "ncclGroupEnd");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.GetErrorString) api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}

} // namespace

struct mrk_comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 0, rank = 0;
  hipStream_t stream = nullptr; // the collectives' own stream: never queued behind the scans of a following batch
  hipEvent_t rows_ready = nullptr, gathered[MRK_MERGE_SLOTS] = {};
  void* rows_all[MRK_MERGE_SLOTS] = {}; // [n_ranks][n_queries][MRK_ROW_WORDS] per slot
  size_t rows_all_bytes[MRK_MERGE_SLOTS] = {};
  void* scratch = nullptr; // all-reduce staging
  size_t scratch_bytes = 0;
  // the partitioned exchange: per slot two dwords (any merged row flagged RERUN / DECLINED, maximum over the ranks), device + pinned host
  uint32_t* flags_dev = nullptr;
  uint32_t* flags_host = nullptr;
  hipEvent_t merged[MRK_MERGE_SLOTS] = {};
};

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return mrk_fail(MRK_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define RCCL_TRY(api, expr)                                                                         \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return mrk_fail(MRK_E_HIP, "%s: %s", #expr, (api)->GetErrorString(r_)); \
  } while (0)

int mrk_comm_unique_id_impl(uint8_t* id_out) {
  RcclApi* api = rccl();
  if (!api) return mrk_fail(MRK_E_UNSUPPORTED, "librccl could not be loaded: %s", dlerror() ? dlerror() : "symbols missing");
  static_assert(MRK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the C-ABI's id size is RCCL's");
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MRK_OK;
}

int mrk_comm_init_impl(mrk_ctx* ctx, const uint8_t* id_bytes, int n_ranks, int rank) {
  if (ctx->comm) return mrk_fail(MRK_E_INVAL, "mrk_comm_init: the context already has a communicator");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return mrk_fail(MRK_E_INVAL, "mrk_comm_init: rank %d of %d", rank, n_ranks);
  RcclApi* api = rccl();
  if (!api) return mrk_fail(MRK_E_UNSUPPORTED, "librccl could not be loaded");
  HIP_TRY(hipSetDevice(ctx->device));
  mrk_comm* c = new (std::nothrow) mrk_comm();
  if (!c) return mrk_fail(MRK_E_NOMEM, "out of memory");
  ncclUniqueId id;
  memcpy(id.internal, id_bytes, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = api->CommInitRank(&c->comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return mrk_fail(MRK_E_HIP, "ncclCommInitRank(%d of %d): %s", rank, n_ranks, api->GetErrorString(r));
  }
  c->n_ranks = n_ranks, c->rank = rank;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->rows_ready, hipEventDisableTiming);
  for (int i = 0; i < MRK_MERGE_SLOTS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->gathered[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    (void)api->CommDestroy(c->comm);
    delete c;
    return mrk_fail(MRK_E_HIP, "mrk_comm_init: %s", hipGetErrorString(e));
  }
  ctx->comm = c;
  return MRK_OK;
}

void mrk_comm_destroy_impl(mrk_ctx* ctx) {
  mrk_comm* c = ctx->comm;
  if (!c) return;
  (void)hipSetDevice(ctx->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (RcclApi* api = rccl()) (void)api->CommDestroy(c->comm);
  for (int i = 0; i < MRK_MERGE_SLOTS; ++i) {
    if (c->rows_all[i]) (void)hipFree(c->rows_all[i]);
    if (c->gathered[i]) (void)hipEventDestroy(c->gathered[i]);
  }
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->flags_dev) (void)hipFree(c->flags_dev);
  if (c->flags_host) (void)hipHostFree(c->flags_host);
  for (int i = 0; i < MRK_MERGE_SLOTS; ++i)
    if (c->merged[i]) (void)hipEventDestroy(c->merged[i]);
  if (c->rows_ready) (void)hipEventDestroy(c->rows_ready);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  ctx->comm = nullptr;
}

// sum of int64 values over the ranks, in place in host memory (per-keyword document counts + the document total: the
// reference's local_df, sphinxrt.cpp:6501-6521, sphinxsearch.cpp:4308-4315)
int mrk_comm_allreduce_i64_impl(mrk_ctx* ctx, int64_t* values, uint64_t n) {
  mrk_comm* c = ctx->comm;
  if (!c) return mrk_fail(MRK_E_INVAL, "mrk_comm_allreduce_i64: no communicator (mrk_comm_init)");
  if (!n) return MRK_OK;
  RcclApi* api = rccl();
  HIP_TRY(hipSetDevice(ctx->device));
  if (c->scratch_bytes < n * 8) {
    if (c->scratch) (void)hipFree(c->scratch);
    c->scratch = nullptr, c->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&c->scratch, n * 8));
    c->scratch_bytes = n * 8;
  }
  HIP_TRY(hipMemcpyAsync(c->scratch, values, n * 8, hipMemcpyHostToDevice, c->stream));
  RCCL_TRY(api, api->AllReduce(c->scratch, c->scratch, n, ncclInt64, ncclSum, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(values, c->scratch, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return MRK_OK;
}

// rows (this shard's [n_queries][MRK_ROW_WORDS], device) --all-gather--> rows_all[slot] --merge kernel--> out_rows;
// ordered behind `after` (a hipEvent_t recorded behind the rows' producer; NULL = the rows are ready) without a host wait;
// completion is the merge slot's (mrk_merge_wait).  *gathered_event_out = the event recorded behind the collective.
int mrk_comm_exchange_impl(mrk_ctx* ctx, const uint64_t* rows, uint32_t n_queries, hipEvent_t after, uint32_t slot, const uint64_t** rows_all_out,
                           hipEvent_t* gathered_event_out) {
  mrk_comm* c = ctx->comm;
  if (!c) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: no communicator (mrk_comm_init)");
  RcclApi* api = rccl();
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t row_bytes = (size_t)n_queries * MRK_ROW_WORDS * 8, need = row_bytes * (size_t)c->n_ranks;
  if (c->rows_all_bytes[slot] < need) {
    if (c->rows_all[slot]) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      (void)hipFree(c->rows_all[slot]);
    }
    c->rows_all[slot] = nullptr, c->rows_all_bytes[slot] = 0;
    HIP_TRY(hipMalloc(&c->rows_all[slot], need));
    c->rows_all_bytes[slot] = need;
  }
  // (a slot that is reused before mrk_merge_wait: its previous merge still reads rows_all[slot])
  if (ctx->merge_used[slot] && ctx->merge_done[slot]) HIP_TRY(hipStreamWaitEvent(c->stream, ctx->merge_done[slot], 0));
  if (after) HIP_TRY(hipStreamWaitEvent(c->stream, after, 0));
  RCCL_TRY(api, api->AllGather(rows, c->rows_all[slot], (size_t)n_queries * MRK_ROW_WORDS, ncclUint64, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->gathered[slot], c->stream));
  *rows_all_out = (const uint64_t*)c->rows_all[slot];
  *gathered_event_out = c->gathered[slot];
  return MRK_OK;
}

// ---- the exchange partitioned by QUERY -------------------------------------------------------------------------------------
// The all-gather form above hands every rank every shard's rows of every query and every rank merges all of them: N times
// the bytes into each GPU, N times the merge work, for N identical copies of the answer.  Here rank r owns the queries
// [r * per, r * per + count), per = ceil(Q / N): every rank sends each owner its rows of the owner's queries (grouped ncclSend /
// ncclRecv: an all-to-all of row slices), merges its own Q / N queries and writes that slice of the merged rows.  Bytes into a
// GPU and merge work per GPU drop by N.  What used to be visible in every rank's merged rows -- "some shard flagged a row: all
// ranks must rerun / report" -- travels as two dwords through one small all-reduce (max) behind the merge.
void mrk_shard_slice_impl(uint32_t n_queries, int n_ranks, int rank, uint32_t* per_out, uint32_t* first_out, uint32_t* count_out) {
  const uint32_t n = n_ranks > 0 ? (uint32_t)n_ranks : 1u, per = (n_queries + n - 1) / n;
  const uint64_t first = (uint64_t)per * (uint32_t)rank;
  const uint32_t f = first < n_queries ? (uint32_t)first : n_queries;
  const uint32_t cnt = n_queries - f < per ? n_queries - f : per;
  if (per_out) *per_out = per;
  if (first_out) *first_out = f;
  if (count_out) *count_out = cnt;
}

bool mrk_comm_can_partition(mrk_ctx* ctx) {
  RcclApi* api = rccl();
  return ctx->comm && api && api->Send && api->Recv && api->GroupStart && api->GroupEnd;
}

// rows -> (all-to-all of row slices) -> recv [n_ranks][per][MRK_ROW_WORDS] of this rank's queries; ordered behind `after`
int mrk_comm_exchange_part_impl(mrk_ctx* ctx, const uint64_t* rows, uint32_t n_queries, hipEvent_t after, uint32_t slot, const uint64_t** recv_out,
                                hipEvent_t* gathered_event_out, uint32_t* per_out, uint32_t* first_out, uint32_t* count_out) {
  mrk_comm* c = ctx->comm;
  if (!c) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: no communicator (mrk_comm_init)");
  RcclApi* api = rccl();
  HIP_TRY(hipSetDevice(ctx->device));
  uint32_t per, first, count;
  mrk_shard_slice_impl(n_queries, c->n_ranks, c->rank, &per, &first, &count);
  const size_t need = (size_t)c->n_ranks * per * MRK_ROW_WORDS * 8;
  if (c->rows_all_bytes[slot] < need) {
    if (c->rows_all[slot]) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipStreamSynchronize(ctx->merge_stream));
      (void)hipFree(c->rows_all[slot]);
    }
    c->rows_all[slot] = nullptr, c->rows_all_bytes[slot] = 0;
    HIP_TRY(hipMalloc(&c->rows_all[slot], need ? need : 8));
    c->rows_all_bytes[slot] = need;
  }
  // a slot that is reused: its previous merge still reads the receive buffer
  if (ctx->merge_used[slot] && ctx->merge_done[slot]) HIP_TRY(hipStreamWaitEvent(c->stream, ctx->merge_done[slot], 0));
  if (after) HIP_TRY(hipStreamWaitEvent(c->stream, after, 0));
  uint64_t* recv = (uint64_t*)c->rows_all[slot];
  if (c->n_ranks == 1 && !ctx->exchange_self_rccl) {
    // one rank: nothing to exchange -- the shard's own rows ARE the receive buffer (RCCL's send-to-self of 6 MB was seen to take
    // 0.2 ms on MI355X; ctx key "exchange_self_rccl" = 1 keeps it, to rehearse the collective's code path with one rank)
    HIP_TRY(hipEventRecord(c->gathered[slot], c->stream));
    *recv_out = rows;
    *gathered_event_out = c->gathered[slot];
    *per_out = per, *first_out = first, *count_out = count;
    return MRK_OK;
  }
  RCCL_TRY(api, api->GroupStart());
  for (int p = 0; p < c->n_ranks; ++p) {
    uint32_t pf, pc;
    mrk_shard_slice_impl(n_queries, c->n_ranks, p, nullptr, &pf, &pc);
    if (pc) RCCL_TRY(api, api->Send(rows + (size_t)pf * MRK_ROW_WORDS, (size_t)pc * MRK_ROW_WORDS, ncclUint64, p, c->comm, c->stream));
    if (count) RCCL_TRY(api, api->Recv(recv + (size_t)p * per * MRK_ROW_WORDS, (size_t)count * MRK_ROW_WORDS, ncclUint64, p, c->comm, c->stream));
  }
  RCCL_TRY(api, api->GroupEnd());
  HIP_TRY(hipEventRecord(c->gathered[slot], c->stream));
  *recv_out = recv;
  *gathered_event_out = c->gathered[slot];
  *per_out = per, *first_out = first, *count_out = count;
  return MRK_OK;
}

// the two flag dwords of `slot`: cleared on the merge stream in front of the merge (flags_dev_out), and -- mrk_comm_flags_finish,
// behind the merge -- maximum over the ranks, copied to pinned host memory, the slot's completion event recorded behind it all
int mrk_comm_flags_begin(mrk_ctx* ctx, uint32_t slot, uint32_t** flags_dev_out) {
  mrk_comm* c = ctx->comm;
  if (!c->flags_dev) {
    HIP_TRY(hipMalloc((void**)&c->flags_dev, MRK_MERGE_SLOTS * 2 * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void**)&c->flags_host, MRK_MERGE_SLOTS * 2 * sizeof(uint32_t), hipHostMallocDefault));
    memset(c->flags_host, 0, MRK_MERGE_SLOTS * 2 * sizeof(uint32_t));
    for (int i = 0; i < MRK_MERGE_SLOTS; ++i) HIP_TRY(hipEventCreateWithFlags(&c->merged[i], hipEventDisableTiming));
  }
  HIP_TRY(hipMemsetAsync(c->flags_dev + 2 * slot, 0, 2 * sizeof(uint32_t), ctx->merge_stream));
  *flags_dev_out = c->flags_dev + 2 * slot;
  return MRK_OK;
}

int mrk_comm_flags_finish(mrk_ctx* ctx, uint32_t slot) {
  mrk_comm* c = ctx->comm;
  RcclApi* api = rccl();
  HIP_TRY(hipEventRecord(c->merged[slot], ctx->merge_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->merged[slot], 0));
  if (c->n_ranks > 1) RCCL_TRY(api, api->AllReduce(c->flags_dev + 2 * slot, c->flags_dev + 2 * slot, 2, ncclUint32, ncclMax, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->flags_host + 2 * slot, c->flags_dev + 2 * slot, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (!ctx->merge_done[slot]) HIP_TRY(hipEventCreateWithFlags(&ctx->merge_done[slot], hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ctx->merge_done[slot], c->stream)); // behind the merge (the stream waited for it), the all-reduce and the copy
  ctx->merge_used[slot] = true;
  return MRK_OK;
}

int mrk_comm_flags_read(mrk_ctx* ctx, uint32_t slot, uint32_t* rerun_any, uint32_t* declined_any) {
  mrk_comm* c = ctx->comm;
  if (!c || !c->flags_host) return mrk_fail(MRK_E_INVAL, "mrk_shard_flags: no partitioned exchange has run on this context");
  if (rerun_any) *rerun_any = c->flags_host[2 * slot];
  if (declined_any) *declined_any = c->flags_host[2 * slot + 1];
  return MRK_OK;
}

int mrk_comm_ranks(const mrk_ctx* ctx) { return ctx->comm ? ctx->comm->n_ranks : 0; }
int mrk_comm_rank(const mrk_ctx* ctx) { return ctx->comm ? ctx->comm->rank : 0; }
hipEvent_t mrk_comm_rows_ready_event(mrk_ctx* ctx) { return ctx->comm ? ctx->comm->rows_ready : nullptr; }
