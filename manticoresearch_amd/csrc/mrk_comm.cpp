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
  if (after) HIP_TRY(hipStreamWaitEvent(c->stream, after, 0));
  RCCL_TRY(api, api->AllGather(rows, c->rows_all[slot], (size_t)n_queries * MRK_ROW_WORDS, ncclUint64, c->comm, c->stream));
  HIP_TRY(hipEventRecord(c->gathered[slot], c->stream));
  *rows_all_out = (const uint64_t*)c->rows_all[slot];
  *gathered_event_out = c->gathered[slot];
  return MRK_OK;
}

int mrk_comm_ranks(const mrk_ctx* ctx) { return ctx->comm ? ctx->comm->n_ranks : 0; }
hipEvent_t mrk_comm_rows_ready_event(mrk_ctx* ctx) { return ctx->comm ? ctx->comm->rows_ready : nullptr; }
