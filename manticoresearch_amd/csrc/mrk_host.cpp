// mrk_host.cpp -- host side of the C-ABI in include/mrk.h: device context, segment ingest
// (reference-format bytes -> HBM: device block index, packed doclists, bitmaps), batch
// submission / collection and the merge entry points.  The query planner is mrk_plan.cpp;
// the kernels live in mrk_scan_bm.hip, mrk_scan_pk.hip and mrk_kernels.hip.  Compiled with hipcc.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mrk_dev.h"
#include "mrk_pack.h"

using namespace mrk;

// ----------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------
static thread_local char g_err[512];

extern "C" const char* mrk_last_error(void) { return g_err; }

int mrk_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return mrk_fail(MRK_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ----------------------------------------------------------------------------------------
// objects
// ----------------------------------------------------------------------------------------
#include "mrk_host_int.h"

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return MRK_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 64;
    HIP_TRY(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return MRK_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

template <typename T>
struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  int reserve(size_t n) {
    if (n <= cap) return MRK_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 4 + 64;
    HIP_TRY(hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault));
    cap = want;
    return MRK_OK;
  }
  // grow but keep the first `keep` elements
  int reserve_keep(size_t n, size_t keep) {
    if (n <= cap) return MRK_OK;
    T* np = nullptr;
    size_t want = n + n / 4 + 64;
    HIP_TRY(hipHostMalloc((void**)&np, want * sizeof(T), hipHostMallocDefault));
    if (p && keep) memcpy(np, p, keep * sizeof(T));
    if (p) (void)hipHostFree(p);
    p = np;
    cap = want;
    return MRK_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

// per-query scan state in front of the histograms: total u64 | tau u64 | flags u32 | cand_n, tau_bin: QSTRIDE dwords each
constexpr size_t STATE_BYTES = 8 + 8 + 4 + 2 * 4 * mrk::QSTRIDE;

struct mrk_batch {
  mrk_ctx* ctx = nullptr;
  uint64_t* rows_dst = nullptr; // mrk_batch_set_rows_dst
  bool host_copied = true;      // per-query keys / counts / totals of the last submit are in pinned host memory
  hipStream_t stream = nullptr; // every batch runs on its own stream: batches of one context overlap on the device
  uint32_t max_queries = 0;
  uint32_t n_queries = 0; // of the last submit
  bool in_flight = false;
  uint32_t rowid_base = 0; // of the segment of the last submit
  // host (pinned) staging
  PinBuf<DevQuery> h_queries;
  PinBuf<DevItem> h_items;
  PinBuf<uint32_t> h_list_first, h_list_n, h_kq;
  PinBuf<uint64_t> h_keys;   // [max_queries][KCAP]
  PinBuf<uint32_t> h_cnt;    // [max_queries]
  PinBuf<uint64_t> h_total;  // [max_queries]
  // device
  DevBuf<DevQuery> d_queries;
  DevBuf<DevItem> d_items;
  DevBuf<uint64_t> d_item_cand;
  DevBuf<uint32_t> d_item_cnt;
  // per-query scan state lives in ONE allocation (q_total | q_tau | q_cand_n | q_flags | q_tau_bin | q_hist) so
  // that a submit clears it with a single memset
  DevBuf<uint8_t> d_state;
  template <typename T>
  struct View {
    T* p = nullptr;
  };
  View<uint64_t> d_q_total, d_q_tau;
  DevBuf<uint32_t> d_list_first, d_list_n, d_kq;
  DevBuf<uint64_t> d_out_keys;
  DevBuf<uint32_t> d_out_cnt;
  DevBuf<uint32_t> d_lb;  // hit-ranked queries: histograms of the matches' lower weight bounds [n][NBINS], their second levels [n][NBINS], then the threshold words [n * QSTRIDE]
  DevBuf<uint32_t> d_sel; // selection scratch: threshold bin | slices in use per query, then the survivors per slice
  // packed path: pruning histograms, candidate lists
  View<uint32_t> d_q_hist, d_q_cand_n, d_q_flags, d_q_tau_bin;
  PinBuf<uint32_t> h_cand_n;
  DevBuf<uint64_t> d_cand;
  PinBuf<uint32_t> h_flags;
  // per query: != 0 when the planner declined it on the last submit's segment; travels in the exchange rows
  PinBuf<uint32_t> h_decl;
  DevBuf<uint32_t> d_decl;
  bool decl_dirty = false; // d_decl holds non-zero words of an earlier submit
  bool any_declined = false;
  // HBM match queues of the hit-ranked queries (scan kernels -> rank_kernel): [0] plain trees, [1] PHRASE & co
  DevBuf<uint32_t> d_mq_data[3], d_mq_hdr[3];
  DevBuf<uint32_t> d_mq_count; // [3][MQ_SHARDS]
  // the generic evaluator (mrk_keval.h): programs of the batch's TF_GEN passes, per-lane + shared hit-list memory
  DevBuf<mrk::GenProg> d_gen_progs;
  DevBuf<mrk::GenHit> d_gen_lane, d_gen_spill;
  DevBuf<unsigned long long> d_gen_used;
  DevBuf<uint32_t> d_gen_near; // GenArgs::near_tab
  std::vector<mrk::GenProg> gen_progs;
  bool last_fat = false;
  // decoded results
  std::vector<uint32_t> rowid;
  std::vector<int32_t> weight;
  std::vector<int32_t> status;
  bool decoded = false;
  bool packed_run = false;
  // what a rerun of an overflowed query needs (mrk_batch_wait)
  mrk_segment* last_seg = nullptr;
  uint32_t n_pass = 0, last_max_terms = 1;
  bool last_prox = false, last_tree = false, last_ext = false;
  mrk_batch* retry = nullptr;
  mrk_batch* probe = nullptr; // cutoff_probe's launches
  hipEvent_t ev_scan0 = nullptr, ev_scan1 = nullptr, ev_merge1 = nullptr;
  mrk_batch_stats stats{};
};

// ----------------------------------------------------------------------------------------
// ctx
// ----------------------------------------------------------------------------------------
static int mrk_ctx_create_impl(int device, mrk_ctx** out) {
  if (!out) return mrk_fail(MRK_E_INVAL, "mrk_ctx_create: out is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return mrk_fail(MRK_E_INVAL, "mrk_ctx_create: device %d of %d", device, n);
  HIP_TRY(hipSetDevice(device));
  mrk_ctx* c = new (std::nothrow) mrk_ctx();
  if (!c) return mrk_fail(MRK_E_NOMEM, "out of memory");
  c->device = device;
  // (stream priorities were tried -- scans low, selection / merges high -- and cost 10-35 % of the throughput on
  // MI355X: the scan launches stalled around every high-priority kernel; all streams stay at the default priority)
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->merge_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return mrk_fail(MRK_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  *out = c;
  return MRK_OK;
}

static void mrk_ctx_destroy_impl(mrk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  mrk_comm_destroy_impl(c);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->merge_stream) (void)hipStreamDestroy(c->merge_stream);
  for (hipEvent_t e : c->merge_done)
    if (e) (void)hipEventDestroy(e);
  delete c;
}

extern "C" int mrk_ctx_set(mrk_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return mrk_fail(MRK_E_INVAL, "mrk_ctx_set: NULL argument");
  if (!strcmp(key, "item_bytes")) {
    if (value < 4096) return mrk_fail(MRK_E_INVAL, "item_bytes must be >= 4096");
    c->item_bytes = value;
    return MRK_OK;
  }
  if (!strcmp(key, "path")) {
    if (value < 0 || value > 2) return mrk_fail(MRK_E_INVAL, "path must be 0 (auto), 1 (vlb) or 2 (packed)");
    c->path = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "pack")) {
    c->pack = value != 0;
    return MRK_OK;
  }
  if (!strcmp(key, "attr_seq")) {
    c->attr_seq = value != 0;
    return MRK_OK;
  }
  if (!strcmp(key, "attr_nibbles")) {
    c->attr_nibbles = value != 0;
    return MRK_OK;
  }
  if (!strcmp(key, "bt_target_items")) {
    if (value < 1 || value > (1 << 20)) return mrk_fail(MRK_E_INVAL, "bt_target_items must be 1 .. 2^20");
    c->bt_target_items = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "bm_target_items")) {
    if (value < 1 || value > (1 << 20)) return mrk_fail(MRK_E_INVAL, "bm_target_items must be 1 .. 2^20");
    c->bm_target_items = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "prox_bound_keywords")) {
    if (value < 0 || value > 1) return mrk_fail(MRK_E_INVAL, "prox_bound_keywords must be 0 or 1");
    c->prox_bound_keywords = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "bt_phrase")) {
    if (value < 0 || value > 1) return mrk_fail(MRK_E_INVAL, "bt_phrase must be 0 or 1");
    c->bt_phrase = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "bt_cover_inv")) {
    if (value < 0 || value > (1 << 20)) return mrk_fail(MRK_E_INVAL, "bt_cover_inv must be 0 (off) .. 2^20");
    c->bt_cover_inv = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "gen_lane_hits")) {
    if (value < 16 || value > (1 << 20)) return mrk_fail(MRK_E_INVAL, "gen_lane_hits must be 16 .. 2^20");
    c->gen_lane_hits = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "gen_spill_mb")) {
    if (value < 1 || value > (64 << 10)) return mrk_fail(MRK_E_INVAL, "gen_spill_mb must be 1 .. 65536");
    c->gen_spill_mb = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "mq_max_chunks")) {
    if (value < 1 || value > (1 << 24)) return mrk_fail(MRK_E_INVAL, "mq_max_chunks must be 1 .. 2^24");
    c->mq_max_chunks = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "prox_prune")) {
    c->prox_prune = value != 0;
    return MRK_OK;
  }
  if (!strcmp(key, "exchange_self_rccl")) {
    c->exchange_self_rccl = value != 0;
    return MRK_OK;
  }
  if (!strcmp(key, "exchange_part")) {
    c->exchange_part = value != 0;
    return MRK_OK;
  }
  if (!strcmp(key, "item_order")) {
    if (value < 0 || value > 15) return mrk_fail(MRK_E_INVAL, "item_order is a mask 0 .. 15");
    c->item_order = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "pk_min_items")) {
    if (value < 1 || value > (1 << 20)) return mrk_fail(MRK_E_INVAL, "pk_min_items must be 1 .. 2^20");
    c->pk_min_items = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "bm_min_windows")) {
    if (value < 16 || value > 4096) return mrk_fail(MRK_E_INVAL, "bm_min_windows must be 16 .. 4096");
    c->bm_min_windows = (int)value;
    return MRK_OK;
  }
  if (!strcmp(key, "bitmap_inv")) {
    if (value < 0 || value > 4096) return mrk_fail(MRK_E_INVAL, "bitmap_inv must be 0 (off) .. 4096");
    c->bitmap_inv = (int)value;
    return MRK_OK;
  }
  return mrk_fail(MRK_E_INVAL, "mrk_ctx_set: unknown key '%s'", key);
}

// ----------------------------------------------------------------------------------------
// segment ingest
// ----------------------------------------------------------------------------------------
static inline uint64_t unzip64(const uint8_t*& p, const uint8_t* end, bool& ok) {
  // SPH_VARINT_DECODE (fileio.cpp:31-45), bounds-checked
  uint64_t res = 0;
  for (;;) {
    if (p >= end) {
      ok = false;
      return 0;
    }
    uint32_t b = *p++;
    res = (res << 7) + (b & 0x7f);
    if (!(b & 0x80)) break;
  }
  return res;
}

static void mrk_segment_destroy_impl(mrk_segment* s) {
  if (!s) return;
  if (s->ctx) (void)hipSetDevice(s->ctx->device);
  if (s->d_spd) (void)hipFree(s->d_spd);
  if (s->d_spp) (void)hipFree(s->d_spp);
  if (s->d_blk_base) (void)hipFree(s->d_blk_base);
  if (s->d_blk_off) (void)hipFree(s->d_blk_off);
  if (s->d_blk_hit) (void)hipFree(s->d_blk_hit);
  if (s->d_pk_base) (void)hipFree(s->d_pk_base);
  if (s->d_pk_doff) (void)hipFree(s->d_pk_doff);
  if (s->d_pk_w) (void)hipFree(s->d_pk_w);
  if (s->d_pk_delta) (void)hipFree(s->d_pk_delta);
  if (s->d_pk_attr) (void)hipFree(s->d_pk_attr);
  if (s->d_pk_exc) (void)hipFree(s->d_pk_exc);
  if (s->d_pk_hit) (void)hipFree(s->d_pk_hit);
  if (s->d_pk_hbase) (void)hipFree(s->d_pk_hbase);
  if (s->d_pk_attr1) (void)hipFree(s->d_pk_attr1);
  if (s->d_pk_attr2) (void)hipFree(s->d_pk_attr2);
  if (s->d_dead) (void)hipFree(s->d_dead);
  if (s->d_attrs) (void)hipFree(s->d_attrs);
  if (s->d_blobs) (void)hipFree(s->d_blobs);
  if (s->d_bm) (void)hipFree(s->d_bm);
  if (s->d_bm_dir) (void)hipFree(s->d_bm_dir);
  delete s;
}

static int mrk_segment_set_dead_rows_impl(mrk_segment* s, const uint32_t* bitmap, uint64_t n_rows) {
  if (!s || !s->ctx) return mrk_fail(MRK_E_INVAL, "mrk_segment_set_dead_rows: NULL segment");
  if (bitmap && n_rows < s->total_docs)
    return mrk_fail(MRK_E_INVAL, "mrk_segment_set_dead_rows: map covers %llu rows, segment has %llu", (unsigned long long)n_rows,
                    (unsigned long long)s->total_docs);
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipDeviceSynchronize()); // batches of this context run on their own streams
  const size_t words = bitmap ? (size_t)((n_rows + 31) / 32) : 0;
  void* fresh = nullptr;
  if (words) {
    // rowids the doclists may name: [0, total_docs); zero-padded to whole 2048-rowid windows (the bitmap
    // kernel reads it window by window) plus a spare word
    const size_t alloc = std::max<size_t>((words + 63) / 64, (size_t)s->dev.n_windows) * 64 + 1;
    HIP_TRY(hipMalloc(&fresh, alloc * 4));
    HIP_TRY(hipMemset(fresh, 0, alloc * 4));
    HIP_TRY(hipMemcpy(fresh, bitmap, words * 4, hipMemcpyHostToDevice));
  }
  if (s->d_dead) (void)hipFree(s->d_dead);
  s->d_dead = fresh;
  s->dev.dead = (const uint32_t*)fresh;
  return MRK_OK;
}

static int mrk_segment_set_attrs_impl(mrk_segment* s, const uint32_t* rows, uint32_t stride, uint64_t n_rows) {
  if (!s || !s->ctx) return mrk_fail(MRK_E_INVAL, "mrk_segment_set_attrs: NULL segment");
  if (rows && (stride == 0 || stride > 4096)) return mrk_fail(MRK_E_INVAL, "mrk_segment_set_attrs: row stride %u dwords", stride);
  if (rows && n_rows < s->total_docs)
    return mrk_fail(MRK_E_INVAL, "mrk_segment_set_attrs: %llu rows, segment has %llu", (unsigned long long)n_rows, (unsigned long long)s->total_docs);
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipDeviceSynchronize()); // batches of this context run on their own streams
  void* fresh = nullptr;
  if (rows) {
    const size_t bytes = (size_t)n_rows * stride * 4;
    HIP_TRY(hipMalloc(&fresh, bytes + 8)); // + a spare dword pair: a 64-bit attribute read never runs off the end
    HIP_TRY(hipMemset((char*)fresh + bytes, 0, 8));
    HIP_TRY(hipMemcpy(fresh, rows, bytes, hipMemcpyHostToDevice));
  }
  if (s->d_attrs) (void)hipFree(s->d_attrs);
  s->d_attrs = fresh;
  s->attr_rows = rows ? n_rows : 0;
  s->dev.attrs = (const uint32_t*)fresh;
  s->dev.attr_stride = rows ? stride : 0;
  return MRK_OK;
}

// The blob pool for MVA filters.  Every row's blob row is walked on the host first (attribute.cpp:495-513: length-size byte,
// n cumulative lengths, data): what the device later dereferences without a check was inside the pool at load.
static int mrk_segment_set_blobs_impl(mrk_segment* s, const uint8_t* pool, uint64_t len, uint32_t n_blob, const uint32_t* rows, uint32_t stride, uint64_t n_rows) {
  if (!s || !s->ctx) return mrk_fail(MRK_E_INVAL, "mrk_segment_set_blobs: NULL segment");
  if (pool) {
    if (!rows || stride < 4 || n_rows < s->total_docs) return mrk_fail(MRK_E_INVAL, "mrk_segment_set_blobs: needs the segment's attribute rows (stride >= 4 dwords: id + blob locator)");
    if (n_blob < 1 || n_blob > 255) return mrk_fail(MRK_E_INVAL, "mrk_segment_set_blobs: %u blob attributes", n_blob);
    for (uint64_t r = 0; r < s->total_docs; ++r) {
      const uint32_t* row = rows + r * stride;
      const uint64_t off = (uint64_t)row[2] | ((uint64_t)row[3] << 32);
      if (off >= len) return mrk_fail(MRK_E_FORMAT, "row %llu: blob offset %llu past the pool (%llu bytes)", (unsigned long long)r, (unsigned long long)off, (unsigned long long)len);
      const uint8_t* br = pool + off;
      if (br[0] > 2) return mrk_fail(MRK_E_FORMAT, "row %llu: blob row type %u", (unsigned long long)r, br[0]);
      const uint32_t sz = br[0] == 0 ? 1u : br[0] == 1 ? 2u : 4u;
      const uint64_t head = 1ull + (uint64_t)n_blob * sz;
      if (len - off < head) return mrk_fail(MRK_E_FORMAT, "row %llu: blob row header past the pool", (unsigned long long)r);
      uint64_t prev = 0;
      for (uint32_t a = 0; a < n_blob; ++a) {
        uint64_t l = 0;
        for (uint32_t i = 0; i < sz; ++i) l |= (uint64_t)br[1 + a * sz + i] << (8 * i);
        if (l < prev || l > len - off - head) return mrk_fail(MRK_E_FORMAT, "row %llu: blob attribute %u runs past the pool", (unsigned long long)r, a);
        prev = l;
      }
    }
  }
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  void* fresh = nullptr;
  if (pool) {
    HIP_TRY(hipMalloc(&fresh, (size_t)len + 16));
    HIP_TRY(hipMemset((char*)fresh + len, 0, 16));
    HIP_TRY(hipMemcpy(fresh, pool, (size_t)len, hipMemcpyHostToDevice));
  }
  if (s->d_blobs) (void)hipFree(s->d_blobs);
  s->d_blobs = fresh;
  s->dev.blobs = (const uint8_t*)fresh;
  s->n_blob_attrs = pool ? n_blob : 0;
  return MRK_OK;
}

static int upload(void** dptr, const void* src, size_t bytes, size_t pad, hipStream_t st) {
  HIP_TRY(hipMalloc(dptr, bytes + pad));
  HIP_TRY(hipMemsetAsync((char*)*dptr + bytes, 0, pad, st));
  if (bytes) HIP_TRY(hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, st));
  return MRK_OK;
}

// argument checks of a segment descriptor that need no device
static int check_desc(const mrk_segment_desc* d) {
  if (!d->spd || !d->spd_len) return mrk_fail(MRK_E_INVAL, "mrk_segment_create: empty .spd");
  const uint32_t sb = d->skiplist_block_size;
  if (sb == 0 || (sb & (sb - 1)) || sb > (uint32_t)DEVBLK)
    return mrk_fail(MRK_E_UNSUPPORTED, "skiplist_block_size %u: device path needs a power of two <= %d", sb, DEVBLK);
  if (d->n_fields > 32) return mrk_fail(MRK_E_UNSUPPORTED, "%u fields: device path covers <= 32", d->n_fields);
  if (d->hit_format != MRK_HITFMT_INLINE && d->hit_format != MRK_HITFMT_PLAIN)
    return mrk_fail(MRK_E_INVAL, "bad hit_format %u", d->hit_format);
  if (d->n_terms && !d->dict) return mrk_fail(MRK_E_INVAL, "mrk_segment_create: %u terms but no dictionary table", d->n_terms);
  return MRK_OK;
}

// device block index: every (DEVBLK / skiplist_block_size)-th SkiplistEntry_t, rebuilt the way
// DiskIndexQwordSetup_c::Setup does per query (sphinx.cpp:13056-13073) -- once, at load time.
// The writer emits ceil(docs/block) snapshots (sphinx.cpp:8447-8453), all of which are used.
static int build_block_index(const mrk_segment_desc* d, std::vector<HostTerm>& terms, std::vector<uint32_t>& blk_base,
                             std::vector<uint64_t>& blk_off, std::vector<uint64_t>& blk_hit) {
  const uint32_t sb = d->skiplist_block_size;
  const uint32_t step = (uint32_t)DEVBLK / sb;
  terms.resize(d->n_terms);
  for (uint32_t t = 0; t < d->n_terms; ++t) {
    const mrk_dict_entry& e = d->dict[t];
    HostTerm& h = terms[t];
    h.docs = e.docs;
    h.hits = e.hits;
    h.doclist_off = e.doclist_off;
    h.doclist_len = e.doclist_len;
    h.blk_first = (uint32_t)blk_base.size();
    if (!e.docs) continue;
    if (e.doclist_off == 0 || e.doclist_off > d->spd_len || e.doclist_len > d->spd_len - e.doclist_off)
      return mrk_fail(MRK_E_FORMAT, "term %u: doclist [%llu,+%llu) outside .spd (%llu bytes)", t,
                      (unsigned long long)e.doclist_off, (unsigned long long)e.doclist_len, (unsigned long long)d->spd_len);
    h.nblocks = (e.docs + DEVBLK - 1) / DEVBLK;
    uint32_t base = 0;
    uint64_t off = e.doclist_off, hit = 0;
    blk_base.push_back(base);
    blk_off.push_back(off);
    blk_hit.push_back(hit);
    if (e.docs > sb) {
      const uint32_t n_snap = (e.docs + sb - 1) / sb; // entry 0 is implicit
      if (!d->spe || e.skiplist_off == 0 || e.skiplist_off >= d->spe_len)
        return mrk_fail(MRK_E_FORMAT, "term %u: skiplist offset %llu outside .spe", t, (unsigned long long)e.skiplist_off);
      const uint8_t* p = d->spe + e.skiplist_off;
      const uint8_t* end = d->spe + d->spe_len;
      bool ok = true;
      for (uint32_t i = 1; i < n_snap; ++i) {
        base += sb + (uint32_t)unzip64(p, end, ok);
        off += 4ull * sb + unzip64(p, end, ok);
        hit += unzip64(p, end, ok);
        if (!ok) return mrk_fail(MRK_E_FORMAT, "term %u: truncated skiplist", t);
        if (i % step == 0) {
          if (off >= e.doclist_off + e.doclist_len) return mrk_fail(MRK_E_FORMAT, "term %u: skiplist entry %u points outside the doclist", t, i);
          blk_base.push_back(base);
          blk_off.push_back(off);
          blk_hit.push_back(hit);
        }
      }
    }
    if (blk_base.size() - h.blk_first != h.nblocks)
      return mrk_fail(MRK_E_FORMAT, "term %u: %u docs need %u blocks, skiplist gave %zu", t, e.docs, h.nblocks, blk_base.size() - h.blk_first);
  }
  return MRK_OK;
}

// validate-only walk (mrk_pack.cpp) of the doclists in `which` (empty = all), in parallel; MRK_E_FORMAT on the first bad one
static int validate_doclists(const mrk_segment_desc* d, const std::vector<uint32_t>& which) {
  const size_t n = which.empty() ? d->n_terms : which.size();
  if (!n) return MRK_OK;
  std::atomic<size_t> next{0};
  std::atomic<int64_t> bad_term{-1};
  std::vector<std::string> errs(n);
  unsigned nth = (unsigned)std::min<size_t>(std::max(1u, std::min(64u, std::thread::hardware_concurrency())), n);
  auto work = [&] {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n || bad_term.load() >= 0) break;
      const uint32_t t = which.empty() ? (uint32_t)i : which[i];
      if (!validate_term(d->spd, d->spd_len, d->dict[t], d->hit_format == MRK_HITFMT_INLINE, d->total_docs, d->spp ? d->spp_len : 0, errs[i])) {
        int64_t none = -1;
        bad_term.compare_exchange_strong(none, (int64_t)i);
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned i = 1; i < nth; ++i) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
  const int64_t bi = bad_term.load();
  if (bi >= 0) {
    const std::string& e = errs[(size_t)bi];
    return mrk_fail(MRK_E_FORMAT, "mrk_segment_create: term %u: %s", which.empty() ? (uint32_t)bi : which[(size_t)bi],
                    e.compare(0, 9, "corrupt: ") == 0 ? e.c_str() + 9 : e.c_str());
  }
  return MRK_OK;
}

extern "C" int mrk_segment_validate(const mrk_segment_desc* d) {
  if (!d) return mrk_fail(MRK_E_INVAL, "mrk_segment_validate: NULL argument");
  int rc = check_desc(d);
  if (rc != MRK_OK) return rc;
  try {
    std::vector<HostTerm> terms;
    std::vector<uint32_t> blk_base;
    std::vector<uint64_t> blk_off, blk_hit;
    if ((rc = build_block_index(d, terms, blk_base, blk_off, blk_hit)) != MRK_OK) return rc;
    return validate_doclists(d, {});
  } catch (const std::bad_alloc&) {
    return mrk_fail(MRK_E_NOMEM, "mrk_segment_validate: out of memory");
  }
}

static int mrk_segment_create_impl(mrk_ctx* ctx, const mrk_segment_desc* d, mrk_segment** out) {
  if (!ctx || !d || !out) return mrk_fail(MRK_E_INVAL, "mrk_segment_create: NULL argument");
  {
    const int rc0 = check_desc(d);
    if (rc0 != MRK_OK) return rc0;
  }
  HIP_TRY(hipSetDevice(ctx->device));

  mrk_segment* s = new (std::nothrow) mrk_segment();
  if (!s) return mrk_fail(MRK_E_NOMEM, "out of memory");
  s->ctx = ctx;
  s->total_docs = d->total_docs;
  s->n_fields = d->n_fields;

  std::vector<uint32_t> blk_base;
  std::vector<uint64_t> blk_off, blk_hit;
  {
    const int rc0 = build_block_index(d, s->terms, blk_base, blk_off, blk_hit);
    if (rc0 != MRK_OK) {
      delete s;
      return rc0;
    }
  }

  // ---- packed doclists: lossless transcode of every term's .spd run (mrk_pack.cpp)
  std::vector<uint32_t> pk_base, pk_doff, pk_delta, pk_attr, pk_hit;
  std::vector<uint64_t> pk_hbase;
  std::vector<uint8_t> pk_w;
  std::vector<uint64_t> pk_exc;
  std::vector<uint32_t> bm_words, bm_dir;
  std::vector<uint8_t> pk_attr1;
  std::vector<uint16_t> pk_attr2; // (indexed like pk_hit; holds zeros for keywords without a bitmap, cut after the last keyword with one)
  size_t attr2_end = 0;
  bool attr2_ok = ctx->attr_seq != 0;
  bool attr1_ok = ctx->attr_nibbles && d->n_fields <= 4;
  bool packed = ctx->pack && d->n_fields <= 8;
  if (!packed) { // no transcode, no walk by pack_term: validate every doclist before any of it reaches the VLB kernel
    const int rcv = validate_doclists(d, {});
    if (rcv != MRK_OK) {
      mrk_segment_destroy_impl(s);
      return rcv;
    }
  }
  if (packed) {
    std::vector<PackedTerm> pt(d->n_terms);
    std::vector<std::string> errs(d->n_terms);
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    const bool inl = d->hit_format == MRK_HITFMT_INLINE;
    unsigned nth = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    nth = (unsigned)std::min<size_t>(nth, d->n_terms ? d->n_terms : 1);
    auto work = [&] {
      for (;;) {
        const size_t t = next.fetch_add(1);
        if (t >= d->n_terms) break;
        // dense terms also get a bitmap of their doc set (the two-bitmap AND kernel, mrk_scan_bm.hip)
        const bool dense = ctx->bitmap_inv > 0 && d->total_docs > 0 && d->total_docs < (1ull << 32) &&
                           (uint64_t)d->dict[t].docs * (uint64_t)ctx->bitmap_inv >= d->total_docs;
        if (!pack_term(d->spd, d->spd_len, d->dict[t], inl, dense ? d->total_docs : 0, pt[t], errs[t], d->total_docs, d->spp ? d->spp_len : 0)) bad = true;
      }
    };
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nth; ++i) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    if (bad) {
      for (uint32_t t = 0; t < d->n_terms; ++t)
        if (errs[t].compare(0, 8, "corrupt:") == 0) { // malformed bytes: nothing of this segment may reach a kernel
          const int rc = mrk_fail(MRK_E_FORMAT, "mrk_segment_create: term %u: %s", t, errs[t].c_str() + 9);
          mrk_segment_destroy_impl(s);
          return rc;
        }
      // e.g. field masks wider than 8 bits: this segment is served by the VLB path only.  pack_term stopped at the
      // decline: the rest of such a doclist is still unchecked -- walk it (a bad rowid behind a wide mask must not
      // reach scan_kernel's dead-row / attribute reads)
      packed = false;
      std::vector<uint32_t> rest;
      for (uint32_t t = 0; t < d->n_terms; ++t)
        if (!errs[t].empty()) rest.push_back(t);
      const int rcv = validate_doclists(d, rest);
      if (rcv != MRK_OK) {
        mrk_segment_destroy_impl(s);
        return rcv;
      }
      for (uint32_t t = 0; t < d->n_terms; ++t)
        if (!errs[t].empty()) {
          fprintf(stderr, "mrk: segment stays on the VLB path: term %u: %s\n", t, errs[t].c_str());
          break;
        }
    } else {
      size_t nd = 0, ne = 0, nblk = 0;
      for (auto& x : pt) nd += x.delta.size(), ne += x.exc.size(), nblk += x.base.size();
      if (nblk != blk_base.size() || nd + 64 > 0xFFFFFFFFull) {
        packed = false;
        fprintf(stderr, "mrk: segment stays on the VLB path: %zu blocks vs %zu in the skiplists, %zu delta words\n", nblk,
                blk_base.size(), nd);
      } else {
        pk_base.reserve(nblk), pk_doff.reserve(nblk), pk_w.reserve(nblk);
        pk_delta.reserve(nd + 64), pk_attr.reserve(nblk * 64), pk_exc.reserve(ne + 1);
        pk_hit.reserve(nblk * 128), pk_hbase.reserve(nblk);
        for (uint32_t t = 0; t < d->n_terms; ++t) {
          PackedTerm& x = pt[t];
          HostTerm& h = s->terms[t];
          h.packed_bytes = x.packed_bytes;
          h.last_rowid = x.last_rowid;
          h.exc_first = (uint32_t)pk_exc.size();
          h.exc_n = (uint32_t)x.exc.size();
          const uint32_t dbase = (uint32_t)pk_delta.size();
          pk_base.insert(pk_base.end(), x.base.begin(), x.base.end());
          for (uint32_t o : x.doff) pk_doff.push_back(dbase + o);
          pk_w.insert(pk_w.end(), x.w.begin(), x.w.end());
          pk_delta.insert(pk_delta.end(), x.delta.begin(), x.delta.end());
          pk_attr.insert(pk_attr.end(), x.attr.begin(), x.attr.end());
          pk_exc.insert(pk_exc.end(), x.exc.begin(), x.exc.end());
          pk_hit.insert(pk_hit.end(), x.hit.begin(), x.hit.end());
          attr1_ok = attr1_ok && x.attr1_ok;
          if (attr1_ok) pk_attr1.insert(pk_attr1.end(), x.attr1.begin(), x.attr1.end());
          pk_hbase.insert(pk_hbase.end(), x.hbase.begin(), x.hbase.end());
          if (attr2_ok) {
            if (!x.bm.empty() && x.attr2.size() == x.hit.size()) {
              pk_attr2.resize(pk_hit.size() - x.hit.size(), 0); // (zeros for the keywords since the last dense one)
              pk_attr2.insert(pk_attr2.end(), x.attr2.begin(), x.attr2.end());
              attr2_end = pk_attr2.size();
            } else if (!x.bm.empty())
              attr2_ok = false;
          }
          if (!x.bm.empty()) {
            h.bm_off = bm_words.size();
            h.dir_off = bm_dir.size();
            bm_words.insert(bm_words.end(), x.bm.begin(), x.bm.end());
            bm_dir.insert(bm_dir.end(), x.bm_dir.begin(), x.bm_dir.end());
          }
          x = PackedTerm();
        }
      }
    }
  }

  int rc;
  const size_t nb = blk_base.size();
  if (packed) {
    if ((rc = upload(&s->d_pk_base, pk_base.data(), pk_base.size() * 4, 64, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_doff, pk_doff.data(), pk_doff.size() * 4, 64, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_w, pk_w.data(), pk_w.size(), 64, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_delta, pk_delta.data(), pk_delta.size() * 4, 1024, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_attr, pk_attr.data(), pk_attr.size() * 4, 64, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_exc, pk_exc.data(), pk_exc.size() * 8, 64, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_hit, pk_hit.data(), pk_hit.size() * 4, 64, ctx->stream)) != MRK_OK ||
        (rc = upload(&s->d_pk_hbase, pk_hbase.data(), pk_hbase.size() * 8, 64, ctx->stream)) != MRK_OK) {
      mrk_segment_destroy_impl(s);
      return rc;
    }
    if (attr1_ok && !bm_words.empty() && pk_attr1.size() == pk_hit.size()) { // only the bitmap kernel reads it
      if ((rc = upload(&s->d_pk_attr1, pk_attr1.data(), pk_attr1.size(), 64, ctx->stream)) != MRK_OK) {
        mrk_segment_destroy_impl(s);
        return rc;
      }
      s->device_bytes += pk_attr1.size();
    }
    if (attr2_ok && !bm_words.empty() && attr2_end) { // only the bitmap kernel reads it
      // (the plane only saves the bitmap kernel a second fetch of some lines: a segment that does not fit with it loads without it)
      if (upload(&s->d_pk_attr2, pk_attr2.data(), attr2_end * 2, 256, ctx->stream) != MRK_OK) {
        if (s->d_pk_attr2) (void)hipFree(s->d_pk_attr2);
        s->d_pk_attr2 = nullptr;
        (void)hipGetLastError();
      } else
        s->device_bytes += attr2_end * 2;
    }
    if (!bm_words.empty()) {
      if ((rc = upload(&s->d_bm, bm_words.data(), bm_words.size() * 4, 1024, ctx->stream)) != MRK_OK ||
          (rc = upload(&s->d_bm_dir, bm_dir.data(), bm_dir.size() * 4, 64, ctx->stream)) != MRK_OK) {
        mrk_segment_destroy_impl(s);
        return rc;
      }
      s->device_bytes += bm_words.size() * 4 + bm_dir.size() * 4;
    }
    s->has_packed = true;
    s->device_bytes += pk_base.size() * 17 + pk_delta.size() * 4 + pk_attr.size() * 4 + pk_exc.size() * 8 + pk_hit.size() * 4;
  }
  if ((rc = upload(&s->d_spd, d->spd, d->spd_len, 64, ctx->stream)) != MRK_OK ||
      (rc = upload(&s->d_spp, d->spp, d->spp ? d->spp_len : 0, 64, ctx->stream)) != MRK_OK ||
      (rc = upload(&s->d_blk_base, blk_base.data(), nb * 4, 64, ctx->stream)) != MRK_OK ||
      (rc = upload(&s->d_blk_off, blk_off.data(), nb * 8, 64, ctx->stream)) != MRK_OK ||
      (rc = upload(&s->d_blk_hit, blk_hit.data(), nb * 8, 64, ctx->stream)) != MRK_OK) {
    mrk_segment_destroy_impl(s);
    return rc;
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    mrk_segment_destroy_impl(s);
    return mrk_fail(MRK_E_HIP, "segment upload: %s", hipGetErrorString(e));
  }
  s->device_bytes += d->spd_len + (d->spp ? d->spp_len : 0) + nb * 20 + 5 * 64;
  s->dev.pk_base = (const uint32_t*)s->d_pk_base;
  s->dev.pk_doff = (const uint32_t*)s->d_pk_doff;
  s->dev.pk_w = (const uint8_t*)s->d_pk_w;
  s->dev.pk_delta = (const uint32_t*)s->d_pk_delta;
  s->dev.pk_attr = (const uint32_t*)s->d_pk_attr;
  s->dev.pk_exc = (const uint64_t*)s->d_pk_exc;
  s->dev.pk_hit = (const uint32_t*)s->d_pk_hit;
  s->dev.pk_hbase = (const uint64_t*)s->d_pk_hbase;
  s->dev.pk_attr1 = (const uint8_t*)s->d_pk_attr1;
  s->dev.pk_attr2 = (const uint16_t*)s->d_pk_attr2;
  s->dev.bm = (const uint32_t*)s->d_bm;
  s->dev.bm_dir = (const uint32_t*)s->d_bm_dir;
  s->dev.n_windows = (uint32_t)((d->total_docs + 2047) / 2048);
  s->dev.spd = (const uint8_t*)s->d_spd;
  s->dev.spp = (const uint8_t*)s->d_spp;
  s->dev.blk_base = (const uint32_t*)s->d_blk_base;
  s->dev.blk_off = (const uint64_t*)s->d_blk_off;
  s->dev.blk_hit = (const uint64_t*)s->d_blk_hit;
  s->dev.spd_len = d->spd_len;
  s->dev.spp_len = d->spp ? d->spp_len : 0;
  s->dev.rowid_base = d->rowid_base;
  s->dev.inline_hits = d->hit_format == MRK_HITFMT_INLINE ? 1u : 0u;
  *out = s;
  return MRK_OK;
}

extern "C" uint64_t mrk_segment_device_bytes(const mrk_segment* s) { return s ? s->device_bytes : 0; }

// ----------------------------------------------------------------------------------------
// IDF  (sphCreateRanker, sphinxsearch.cpp:4317-4361) -- host libm logf like the reference
// ----------------------------------------------------------------------------------------
extern "C" float mrk_idf(int64_t term_docs, int64_t total_docs, int plain_idf, int normalized, int n_qwords, float boost) {
  float idf = 0.0f;
  if (term_docs) {
    const int64_t total_clamped = std::max(total_docs, term_docs);
    const float log_total = logf(float(1 + total_clamped));
    if (!plain_idf)
      idf = logf(float(total_clamped - term_docs + 1) / float(term_docs)) / (2 * log_total);
    else
      idf = logf(float(total_clamped) / float(term_docs)) / (2 * log_total);
  }
  if (normalized) idf /= n_qwords;
  return idf * boost;
}

// ----------------------------------------------------------------------------------------
// batch
// ----------------------------------------------------------------------------------------
static void mrk_batch_destroy_impl(mrk_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  if (b->in_flight && b->stream) (void)hipStreamSynchronize(b->stream);
  b->h_queries.release();
  b->h_items.release();
  b->h_list_first.release();
  b->h_list_n.release();
  b->h_kq.release();
  b->h_keys.release();
  b->h_cnt.release();
  b->h_total.release();
  b->d_queries.release();
  b->d_items.release();
  b->d_item_cand.release();
  b->d_item_cnt.release();
  b->d_state.release();
  b->d_list_first.release();
  b->d_list_n.release();
  b->d_kq.release();
  b->d_out_keys.release();
  b->d_out_cnt.release();
  b->d_sel.release();
  b->d_lb.release();

  b->h_cand_n.release();
  b->d_cand.release();
  b->h_flags.release();
  b->h_decl.release();
  b->d_decl.release();
  for (int i = 0; i < 3; ++i) b->d_mq_data[i].release(), b->d_mq_hdr[i].release();
  b->d_mq_count.release();
  b->d_gen_progs.release(), b->d_gen_lane.release(), b->d_gen_spill.release(), b->d_gen_used.release(), b->d_gen_near.release();
  if (b->retry) mrk_batch_destroy_impl(b->retry);
  if (b->probe) mrk_batch_destroy_impl(b->probe);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  if (b->ev_scan0) (void)hipEventDestroy(b->ev_scan0);
  if (b->ev_scan1) (void)hipEventDestroy(b->ev_scan1);
  if (b->ev_merge1) (void)hipEventDestroy(b->ev_merge1);
  delete b;
}

static int mrk_batch_create_impl(mrk_ctx* ctx, uint32_t max_queries, mrk_batch** out) {
  if (!ctx || !out || !max_queries) return mrk_fail(MRK_E_INVAL, "mrk_batch_create: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  mrk_batch* b = new (std::nothrow) mrk_batch();
  if (!b) return mrk_fail(MRK_E_NOMEM, "out of memory");
  b->ctx = ctx;
  b->max_queries = max_queries;
  int rc = MRK_OK;
  const size_t nq = max_queries;
  if ((rc = b->h_queries.reserve(nq)) || (rc = b->h_list_first.reserve(nq)) || (rc = b->h_list_n.reserve(nq)) ||
      (rc = b->h_kq.reserve(nq)) || (rc = b->h_keys.reserve(nq * KCAP)) || (rc = b->h_cnt.reserve(nq)) ||
      (rc = b->h_total.reserve(nq)) || (rc = b->d_queries.reserve(nq)) ||
      (rc = b->d_state.reserve(nq * (STATE_BYTES + (size_t)NBINS * 4))) || (rc = b->d_list_first.reserve(nq)) || (rc = b->d_list_n.reserve(nq)) ||
      (rc = b->d_kq.reserve(nq)) || (rc = b->d_out_keys.reserve(nq * KCAP)) || (rc = b->d_out_cnt.reserve(nq)) ||
      (rc = b->h_flags.reserve(nq)) || (rc = b->h_cand_n.reserve(nq)) || (rc = b->h_decl.reserve(nq)) || (rc = b->d_decl.reserve(nq)) || (rc = b->d_mq_count.reserve(3 * mrk::MQ_SHARDS))) {
    mrk_batch_destroy_impl(b);
    return rc;
  }
  {
    uint8_t* base = b->d_state.p;
    b->d_q_total.p = (uint64_t*)base;
    b->d_q_tau.p = (uint64_t*)(base + nq * 8);
    b->d_q_flags.p = (uint32_t*)(base + nq * 16);
    b->d_q_cand_n.p = (uint32_t*)(base + nq * 20);                        // QSTRIDE dwords per query
    b->d_q_tau_bin.p = (uint32_t*)(base + nq * (20 + 4 * mrk::QSTRIDE));  // QSTRIDE dwords per query
    b->d_q_hist.p = (uint32_t*)(base + nq * STATE_BYTES);
  }
  hipError_t e1 = hipEventCreate(&b->ev_scan0), e2 = hipEventCreate(&b->ev_scan1), e3 = hipEventCreate(&b->ev_merge1);
  if (e1 == hipSuccess) e1 = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    mrk_batch_destroy_impl(b);
    return mrk_fail(MRK_E_HIP, "hipEventCreate failed");
  }
  b->rowid.resize(nq * KCAP);
  b->weight.resize(nq * KCAP);
  b->status.assign(nq, MRK_OK);
  *out = b;
  return MRK_OK;
}

// mirrors scan_pk_kernel: matches of these passes leave through the match queue (state rankers over more than one
// keyword, whole-query PHRASE); `fat` = the queue whose consumer carries the word state machines
static bool pass_queues_matches(const DevQuery& P, bool& fat) {
  const uint32_t rk = P.ranker;
  if (P.tree_flags & mrk::TF_GEN) return fat = false, true; // (queue 2: see queue_of)
  const bool prox_ranker = (rk == MRK_RANK_PROXIMITY_BM25 || rk == MRK_RANK_PROXIMITY)
                               ? P.n_terms > 1
                               : (rk == MRK_RANK_WORDCOUNT || rk == MRK_RANK_MATCHANY || rk == MRK_RANK_FIELDMASK || rk == MRK_RANK_SPH04);
  fat = (P.tree_flags & mrk::TF_FAT) != 0;
  return prox_ranker || (P.tree_flags & mrk::TF_PHRASE) != 0;
}

static int queue_of(const DevQuery& P, bool fat) { return (P.tree_flags & mrk::TF_GEN) ? 2 : fat ? 1 : 0; }

// upper bound of the docs a pass can match: its driver's docs; the common docs for the two-bitmap AND; any keyword's docs for
// a tree evaluated on bitmap words
static uint64_t pass_max_matches(const DevQuery& P) {
  if (P.tree_flags & mrk::TF_BITMAP) return std::min<uint64_t>(P.t[0].docs, P.t[1].docs);
  if (P.tree_flags & mrk::TF_BTREE) {
    uint64_t d = 0, least = ~0ull;
    for (uint32_t k = 0; k < P.n_terms && k < (uint32_t)MRK_MAX_AND_TERMS; ++k) d += P.t[k].docs, least = std::min<uint64_t>(least, P.t[k].docs);
    return (P.tree_flags & mrk::TF_MULTIAND) && P.n_terms ? least : d; // (an AND of keywords holds no more docs than its rarest one)
  }
  return P.t[0].docs;
}

// size the batch's match queues for `chunks[q]` chunks (0 = queue unused) and point the scan arguments at them
static int bind_match_queues(mrk_batch* b, const uint64_t chunks[3], mrk::ScanArgs& sa) {
  for (int i = 0; i < 3; ++i) {
    const int planes = i == 2 ? mrk::MQ_GEN_PLANES : mrk::MQ_PLANES;
    sa.mq[i] = mrk::MatchQueue{};
    sa.mq[i].count = b->d_mq_count.p + mrk::MQ_SHARDS * i;
    if (!chunks[i]) continue;
    // per shard: its share of the chunks + slack for the spread between shards (workgroups are dealt round-robin)
    const uint64_t per = chunks[i] / mrk::MQ_SHARDS + chunks[i] / (8 * mrk::MQ_SHARDS) + 64;
    int rc;
    if ((rc = b->d_mq_data[i].reserve((size_t)per * mrk::MQ_SHARDS * planes * 64)) || (rc = b->d_mq_hdr[i].reserve((size_t)per * mrk::MQ_SHARDS))) return rc;
    sa.mq[i].data = b->d_mq_data[i].p;
    sa.mq[i].hdr = b->d_mq_hdr[i].p;
    sa.mq[i].cap = (uint32_t)per;
  }
  return MRK_OK;
}

// the generic evaluator's memory: allocated the first time a batch holds such a query, kept with the batch
static int bind_gen(mrk_batch* b, mrk_batch* owner_of_progs, mrk::ScanArgs& sa, hipStream_t st, uint32_t n_queries, bool nearn) {
  const uint32_t n_lanes = (uint32_t)mrk::GEN_GRID * mrk::WG, lane_hits = (uint32_t)b->ctx->gen_lane_hits;
  const size_t spill = (size_t)b->ctx->gen_spill_mb * (1u << 20) / sizeof(mrk::GenHit);
  int rc;
  if ((rc = b->d_gen_lane.reserve((size_t)n_lanes * lane_hits)) || (rc = b->d_gen_spill.reserve(spill)) || (rc = b->d_gen_used.reserve(1))) return rc;
  HIP_TRY(hipMemsetAsync(b->d_gen_used.p, 0, sizeof(unsigned long long), st));
  sa.gen.progs = owner_of_progs->d_gen_progs.p;
  sa.gen.lane_arena = b->d_gen_lane.p;
  sa.gen.lane_hits = lane_hits, sa.gen.n_lanes = n_lanes;
  sa.gen.spill = b->d_gen_spill.p;
  sa.gen.spill_cap = spill;
  sa.gen.spill_used = b->d_gen_used.p;
  sa.gen.near_tab = nullptr, sa.gen.phase = 0;
  if (nearn) {
    if ((rc = b->d_gen_near.reserve((size_t)n_queries * 64))) return rc;
    HIP_TRY(hipMemsetAsync(b->d_gen_near.p, 0xFF, (size_t)n_queries * 64 * 4, st));
    sa.gen.near_tab = b->d_gen_near.p;
  }
  return MRK_OK;
}

// the generic evaluator over queue 2; queries with a NEAR over 3+ operands at the root need the probe launch first
static int launch_gen_rank(mrk::ScanArgs& sa, bool nearn, hipStream_t st) {
  if (nearn) {
    sa.gen.phase = 1;
    launch_rank(sa, 2, st);
    HIP_TRY(hipMemsetAsync(sa.gen.spill_used, 0, sizeof(unsigned long long), st)); // the probe's lists are gone
    sa.gen.phase = 0;
  }
  launch_rank(sa, 2, st);
  return MRK_OK;
}

static int mrk_batch_submit_impl(mrk_batch* b, mrk_segment* seg, const mrk_query* queries, uint32_t n);
static int mrk_batch_wait_impl(mrk_batch* b);
static int mrk_batch_result_impl(mrk_batch* b, uint32_t q, mrk_result* out);

// CSphQuery::m_iCutoff (MatchExtended, sphinx.cpp:12197-12199, 12261-12267): the reference walks the matches in rowid order and
// stops after the first `cutoff` of them that reached the sorter (past filters, dead rows; CSphMatchQueue::PushT returns true
// whether or not the heap kept the match, sphinxsort.cpp:722-759).  Here the matches of a launch come in no order, so the
// queries with a cutoff first run as a probe -- same tree, same filters, ranker NONE, K = cutoff: weights are all 1 and the
// top-K order falls back to the rowid, so the K-th row is where the reference stops -- and the launch proper keeps the rows
// up to it (DevQuery::rowid_max).  Costs a launch where the reference saves work; the results are the reference's.
static int cutoff_probe(mrk_batch* b, mrk_segment* seg, const mrk_query* queries, uint32_t n, std::vector<uint32_t>& rowid_max) {
  std::vector<mrk_query> pq;
  std::vector<uint32_t> who;
  for (uint32_t i = 0; i < n; ++i)
    if (queries[i].cutoff > 0 && queries[i].cutoff <= MRK_MAX_K && queries[i].n_weight_filters == 0) { // (the others: plan_query says why not)
      mrk_query q = queries[i];
      q.ranker = MRK_RANK_NONE;
      q.max_matches = q.cutoff;
      q.cutoff = 0;
      pq.push_back(q);
      who.push_back(i);
    }
  if (pq.empty()) return MRK_OK;
  int rc;
  if (b->probe && b->probe->max_queries < pq.size()) {
    mrk_batch_destroy_impl(b->probe);
    b->probe = nullptr;
  }
  if (!b->probe && (rc = mrk_batch_create_impl(b->ctx, (uint32_t)std::max<size_t>(pq.size(), 16), &b->probe))) return rc;
  if ((rc = mrk_batch_submit_impl(b->probe, seg, pq.data(), (uint32_t)pq.size())) || (rc = mrk_batch_wait_impl(b->probe))) return rc;
  for (size_t j = 0; j < pq.size(); ++j) {
    mrk_result r{};
    if ((rc = mrk_batch_result_impl(b->probe, (uint32_t)j, &r))) return rc;
    // (a probe that was declined: the launch proper is declined for the same reason, with its own message)
    if (r.status == MRK_OK && r.total_found > (int64_t)pq[j].max_matches && r.n == pq[j].max_matches) rowid_max[who[j]] = r.rowid[r.n - 1];
  }
  return MRK_OK;
}

// the selection's sub-bin geometry: the segment's largest global rowid and the bits its rowid range needs
static void sel_rowid_range(const mrk_segment* seg, mrk::SelectArgs& se) {
  const uint64_t docs = std::max<uint64_t>(seg->total_docs, 1);
  uint32_t bits = 1;
  while (bits < 32 && ((docs - 1) >> bits)) ++bits;
  se.rowid_bits = bits;
  se.rowid_hi = (uint32_t)(seg->dev.rowid_base + docs - 1);
}

static int mrk_batch_submit_impl(mrk_batch* b, mrk_segment* seg, const mrk_query* queries, uint32_t n) {
  if (!b || !seg || (!queries && n)) return mrk_fail(MRK_E_INVAL, "mrk_batch_submit: NULL argument");
  if (n > b->max_queries) return mrk_fail(MRK_E_INVAL, "mrk_batch_submit: %u queries > batch capacity %u", n, b->max_queries);
  if (seg->ctx != b->ctx) return mrk_fail(MRK_E_INVAL, "mrk_batch_submit: segment and batch belong to different contexts");
  HIP_TRY(hipSetDevice(b->ctx->device));
  // descriptors + scan kernels go down the context's stream (scans of different batches run back to back, each
  // with the whole chip); the top-K selection and the result copies follow on the batch's own stream, so they
  // overlap the next batch's scan
  hipStream_t st = b->ctx->stream, st2 = b->stream;
  if (b->in_flight) HIP_TRY(hipStreamSynchronize(st2)); // pinned staging is about to be rewritten
  b->in_flight = false;
  b->decoded = false;
  b->n_queries = n;
  b->rowid_base = seg->dev.rowid_base;
  b->stats = mrk_batch_stats{};
  if (!n) return MRK_OK;
  std::vector<uint32_t> rowid_max;
  for (uint32_t i = 0; i < n; ++i)
    if (queries[i].cutoff > 0) {
      rowid_max.assign(n, 0xFFFFFFFFu);
      const int rc = cutoff_probe(b, seg, queries, n, rowid_max);
      if (rc != MRK_OK) return rc;
      break;
    }

  // ---- plan
  const auto t_submit0 = std::chrono::steady_clock::now();
  std::vector<DevItem> items, items_bm;
  items.reserve(n * 4);
  uint64_t algo_bytes = 0, dev_bytes = 0, cand_total = 0;
  uint32_t max_terms = 1;
  bool any_prox = false, any_tree = false, any_ext = false; // ext: position modifiers, BEFORE, attribute filters
  if (b->ctx->path == 2 && !seg->has_packed) return mrk_fail(MRK_E_UNSUPPORTED, "path=packed but the segment has no packed doclists");
  const bool use_packed = seg->has_packed && b->ctx->path != 1;
  std::vector<DevQuery> extra; // passes beyond the first of tree queries; pass index = n + position
  b->gen_progs.clear();
  for (uint32_t i = 0; i < n; ++i) {
    const size_t extra0 = extra.size(), items0 = items.size(), items_bm0 = items_bm.size(), gen0 = b->gen_progs.size();
    int rc = plan_query(seg, queries[i], b->ctx->item_bytes, use_packed, b->h_queries.p[i], extra, n, items, items_bm, i, algo_bytes,
                        dev_bytes, cand_total, any_prox, any_tree, b->gen_progs, rowid_max.empty() ? 0xFFFFFFFFu : rowid_max[i]);
    b->status[i] = rc;
    if (rc == MRK_E_INVAL) return rc;
    if (rc != MRK_OK) { // unsupported: reported per query, runs no device work
      items.resize(items0);
      items_bm.resize(items_bm0);
      extra.resize(extra0);
      b->gen_progs.resize(gen0);
      b->h_queries.p[i].n_items = 0;
      b->h_queries.p[i].n_terms = 0;
    }
    max_terms = std::max(max_terms, b->h_queries.p[i].n_terms);
    any_ext = any_ext || (b->h_queries.p[i].tree_flags & (mrk::TF_TERMPOS | mrk::TF_ORDER | mrk::TF_PHRASE_LEAF | mrk::TF_NOTNEAR)) != 0 || b->h_queries.p[i].n_filters != 0 || b->h_queries.p[i].n_wfilters != 0 ||
              b->h_queries.p[i].rowid_max != 0xFFFFFFFFu;
    b->h_list_first.p[i] = b->h_queries.p[i].item_first;
    b->h_list_n.p[i] = b->h_queries.p[i].n_items;
    b->h_kq.p[i] = b->h_queries.p[i].k ? b->h_queries.p[i].k : 1;
  }
  // A small batch (one-eighth shards, selective keywords, a lone query): the planner cuts a driver doclist into ~item_bytes
  // pieces whatever the batch holds, and 280 workgroups that each walk 70 blocks per wave one after the other leave the chip
  // idle for 0.1 ms.  Cut the block ranges finer until the launch has pk_min_items work items (never under one block per wave).
  if (use_packed && !items.empty() && items.size() < (size_t)b->ctx->pk_min_items) {
    uint64_t total_blocks = 0;
    for (const DevItem& it : items) total_blocks += it.blk_end - it.blk_begin;
    uint64_t per = (total_blocks + (uint64_t)b->ctx->pk_min_items - 1) / (uint64_t)b->ctx->pk_min_items;
    per = std::max<uint64_t>(T0_BLOCKS, (per + T0_BLOCKS - 1) / T0_BLOCKS * T0_BLOCKS);
    std::vector<DevItem> cut;
    cut.reserve(items.size() + (size_t)(total_blocks / per) + 1);
    std::vector<uint32_t> per_pass((size_t)n + extra.size(), 0);
    for (const DevItem& whole : items)
      for (uint64_t x = whole.blk_begin; x < whole.blk_end; x += per) {
        DevItem it = whole;
        it.blk_begin = (uint32_t)x;
        it.blk_end = (uint32_t)std::min<uint64_t>(whole.blk_end, x + per);
        cut.push_back(it);
        if (it.query < per_pass.size()) ++per_pass[it.query];
      }
    items.swap(cut);
    // (the counts feed the match-queue sizing below; the VLB path's per-query list ranges are not built from a packed plan)
    for (uint32_t i = 0; i < n; ++i)
      if (per_pass[i]) b->h_queries.p[i].n_items = per_pass[i];
    for (size_t e = 0; e < extra.size(); ++e)
      if (per_pass[n + e]) extra[e].n_items = per_pass[n + e];
  }
  // ... and in piece-major order, for the reason given at the window-range items below: concurrent workgroups should belong
  // to different queries (the planner emits a query's items back to back; only the VLB path needs them that way)
  // (not for batches whose matches travel through the match queue to the hit pass: config 3 measured 6.6 ms query-major, 7.1 ms
  // interleaved -- the rank kernel likes a query's chunks in rowid order)
  if (use_packed && items.size() > 1 && (b->ctx->item_order & 1) && (!any_prox || (b->ctx->item_order & 8))) {
    std::vector<DevItem> rr;
    rr.reserve(items.size());
    std::vector<size_t> run_begin, run_end; // runs of items of one pass
    for (size_t i = 0; i < items.size();) {
      size_t j = i + 1;
      while (j < items.size() && items[j].query == items[i].query) ++j;
      run_begin.push_back(i), run_end.push_back(j);
      i = j;
    }
    if (run_begin.size() > 1) {
      for (size_t k = 0; rr.size() < items.size(); ++k)
        for (size_t r = 0; r < run_begin.size(); ++r)
          if (run_begin[r] + k < run_end[r]) rr.push_back(items[run_begin[r] + k]);
      items.swap(rr);
    }
  }
  const float plan_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_submit0).count();
  const size_t n_items_pk = items.size();
  // window-range work items (two-bitmap AND kernel, then the window-driven tree kernel) ride behind the block work
  // items; each kind's whole-range entries are cut once the batch's total is known (a wave's fixed costs -- tables, final
  // publish, atomics on the query's counters -- want long runs of windows)
  size_t n_items_kind[3] = {0, 0, 0};
  for (uint32_t kind = 0; kind < 2; ++kind) {
    uint64_t total_win = 0;
    for (const DevItem& it : items_bm)
      if (it.kind == kind) total_win += it.blk_end - it.blk_begin;
    if (!total_win) continue;
    const uint64_t unit = 4 * WAVES; // one burst per wave
    uint64_t wpi = (total_win / (uint64_t)(kind == 0 ? b->ctx->bm_target_items : b->ctx->bt_target_items) / unit) * unit;
    wpi = std::min<uint64_t>(std::max<uint64_t>(wpi, kind == 0 ? (uint64_t)b->ctx->bm_min_windows / unit * unit : 4 * unit), 4096); // (short runs: a wave's fixed costs show -- 12.5 M docs, 8192 items: 0.55 vs 0.47 ms)
    // Piece-major order: the k-th piece of every query, then the (k+1)-th ...  Workgroups that run at the same time then
    // belong to DIFFERENT queries.  Query-major order put a query's 20-50 workgroups on the chip together, all of them adding
    // to the one candidate counter, the same few histogram bins and the one threshold word of that query: device-scope
    // atomics on one address serialize at the memory side (~70 ns each), about 0.1 ms per query whatever the shard size --
    // hidden behind 100 M docs, the whole launch at 12.5 M (12288 work items: 0.85 ms; 4096: 0.39 ms, same bytes).
    const size_t before = items.size();
    if (!(b->ctx->item_order & (kind == 0 ? 2 : 4)) || (kind == 1 && any_prox && !(b->ctx->item_order & 8))) { // query-major (experiments; trees that feed the match queue)
      for (const DevItem& whole : items_bm)
        if (whole.kind == kind)
          for (uint64_t w = whole.blk_begin; w < whole.blk_end; w += wpi) {
            DevItem it = whole;
            it.blk_begin = (uint32_t)w;
            it.blk_end = (uint32_t)std::min<uint64_t>(whole.blk_end, w + wpi);
            items.push_back(it);
          }
    } else
    for (uint64_t piece = 0;; ++piece) {
      bool any = false;
      for (const DevItem& whole : items_bm) {
        if (whole.kind != kind) continue;
        const uint64_t w = whole.blk_begin + piece * wpi;
        if (w >= whole.blk_end) continue;
        any = true;
        DevItem it = whole;
        it.blk_begin = (uint32_t)w;
        it.blk_end = (uint32_t)std::min<uint64_t>(whole.blk_end, w + wpi);
        items.push_back(it);
      }
      if (!any) break;
    }
    n_items_kind[kind] = items.size() - before;
  }
  for (const DevItem& it : items_bm) // the generic evaluator's candidates: block ranges, cut by the planner
    if (it.kind == 2) items.push_back(it), ++n_items_kind[2];
  // match queues: a pass hands over at most one entry per doc it can match, plus one partial chunk per wave of its items
  uint64_t mq_chunks[3] = {0, 0, 0};
  if (use_packed && any_prox) {
    bool bt_feeds[3] = {false, false, false};
    auto account = [&](const DevQuery& P) {
      bool fat = false;
      if (!P.n_items || !pass_queues_matches(P, fat)) return;
      const bool bt = (P.tree_flags & mrk::TF_BTREE) != 0;
      mq_chunks[queue_of(P, fat)] += pass_max_matches(P) / 64 + (bt ? 0 : 4ull * (mrk::MQ_BATCH + 1) * P.n_items) + 1;
      bt_feeds[queue_of(P, fat)] = bt_feeds[queue_of(P, fat)] || bt;
    };
    for (uint32_t i = 0; i < n; ++i) account(b->h_queries.p[i]);
    for (const DevQuery& P : extra) account(P);
    for (int i = 0; i < 2; ++i) // per wave one partial chunk + the unused rest of a reservation (its work items were only cut just now)
      if (bt_feeds[i]) mq_chunks[i] += 4ull * (mrk::MQ_BATCH + 1) * n_items_kind[1];
    for (int i = 0; i < 3; ++i) mq_chunks[i] = std::min<uint64_t>(mq_chunks[i], (uint64_t)b->ctx->mq_max_chunks);
  }
  const size_t n_items_bm = items.size() - n_items_pk;
  const size_t n_items = items.size();
  b->stats.algo_bytes = algo_bytes;
  b->stats.dev_bytes = dev_bytes;
  b->stats.packed = use_packed ? 1 : 0;
  b->stats.n_items = n_items;
  b->stats.n_items_bm = n_items_bm;
  int rc;
  if ((rc = b->h_items.reserve(n_items + 1)) || (rc = b->d_items.reserve(n_items + 1)) ||
      (rc = b->d_item_cand.reserve((n_items + 1) * KCAP)) || (rc = b->d_item_cnt.reserve(n_items + 1)))
    return rc;
  if (n_items) memcpy(b->h_items.p, items.data(), n_items * sizeof(DevItem));
  const size_t n_pass = (size_t)n + extra.size();
  if ((rc = b->h_queries.reserve_keep(n_pass, n)) || (rc = b->d_queries.reserve(n_pass))) return rc;
  if (!extra.empty()) memcpy(b->h_queries.p + n, extra.data(), extra.size() * sizeof(DevQuery));
  if (use_packed && ((rc = b->d_cand.reserve(cand_total + 64)) || (rc = b->d_sel.reserve(2 * (size_t)n + mrk::sel_slice_slots(cand_total, n))))) return rc;

  static const bool phase_timing = getenv("MRK_SUBMIT_TIMING") != nullptr;
  auto lap = [&](const char* what) {
    if (phase_timing)
      fprintf(stderr, "mrk submit %p n=%u %s @%.3f ms (abs %.3f)\n", (void*)b, n, what,
              std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_submit0).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
  };
  lap("planned+staged");
  // ---- descriptors up + scan state cleared: one kernel on the scan stream (prep_kernel reads the pinned staging itself)
  {
    static_assert(sizeof(DevQuery) % 4 == 0 && sizeof(DevItem) % 4 == 0, "descriptors are copied dword-wise");
    mrk::PrepArgs pa{};
    pa.dst[0] = (uint32_t*)b->d_queries.p, pa.src[0] = (const uint32_t*)b->h_queries.p, pa.n4[0] = (uint32_t)(n_pass * sizeof(DevQuery) / 4);
    pa.dst[1] = (uint32_t*)b->d_items.p, pa.src[1] = (const uint32_t*)b->h_items.p, pa.n4[1] = (uint32_t)(n_items * sizeof(DevItem) / 4);
    // totals, thresholds, counters (+ the pruning histograms of the queries in use)
    pa.zero = (uint32_t*)b->d_state.p;
    pa.zero_n4 = (uint32_t)(((size_t)b->max_queries * STATE_BYTES + (use_packed ? (size_t)n * NBINS * 4 : 0)) / 4);
    launch_prep(pa, st);
  }
  if (!use_packed) { // the VLB path's merge kernel reads per-query list ranges
    HIP_TRY(hipMemcpyAsync(b->d_list_first.p, b->h_list_first.p, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_list_n.p, b->h_list_n.p, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_kq.p, b->h_kq.p, n * 4, hipMemcpyHostToDevice, st));
  }

  ScanArgs sa{};
  sa.seg = seg->dev;
  sa.queries = b->d_queries.p;
  sa.items = b->d_items.p;
  sa.item_cand = b->d_item_cand.p;
  sa.item_cnt = b->d_item_cnt.p;
  sa.q_total = b->d_q_total.p;
  sa.q_tau = b->d_q_tau.p;
  sa.n_items = (uint32_t)(use_packed ? n_items_pk : n_items);
  sa.q_hist = b->d_q_hist.p;
  sa.q_cand_n = b->d_q_cand_n.p;
  sa.q_flags = b->d_q_flags.p;
  sa.q_tau_bin = b->d_q_tau_bin.p;
  sa.cand = b->d_cand.p;
  bool any_nearn = false;
  if (use_packed && any_prox && b->ctx->prox_prune) { // pruning in front of the hit pass (mrk_kprune.h, prox_bounds)
    const size_t words = (size_t)n * (2 * NBINS + mrk::QSTRIDE);
    if ((rc = b->d_lb.reserve(words))) return rc;
    HIP_TRY(hipMemsetAsync(b->d_lb.p, 0, words * 4, st));
    sa.q_hist_lb = b->d_lb.p;
    sa.q_hist_lb2 = b->d_lb.p + (size_t)n * NBINS;
    sa.q_tau_lb = b->d_lb.p + (size_t)n * 2 * NBINS;
  }
  if ((rc = bind_match_queues(b, mq_chunks, sa))) return rc;
  if (mq_chunks[0] || mq_chunks[1] || mq_chunks[2]) HIP_TRY(hipMemsetAsync(b->d_mq_count.p, 0, 3 * mrk::MQ_SHARDS * 4, st));
  if (!b->gen_progs.empty()) {
    if ((rc = b->d_gen_progs.reserve(b->gen_progs.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(b->d_gen_progs.p, b->gen_progs.data(), b->gen_progs.size() * sizeof(mrk::GenProg), hipMemcpyHostToDevice, st)); // (pageable: the vector lives until the next submit)
    for (uint32_t i = 0; i < n; ++i) any_nearn = any_nearn || (b->h_queries.p[i].tree_flags & mrk::TF_GEN_NEARN) != 0;
    if ((rc = bind_gen(b, b, sa, st, n, any_nearn))) return rc;
  }
  lap("h2d+memset");
  HIP_TRY(hipEventRecord(b->ev_scan0, st));
  if (use_packed) {
    launch_scan_pk(sa, max_terms, any_prox, any_tree, any_ext, st);
    if (n_items_kind[1]) { // before the rank kernels: it feeds the match queue too
      ScanArgs sb = sa;
      sb.items = b->d_items.p + n_items_pk + n_items_kind[0];
      sb.n_items = (uint32_t)n_items_kind[1];
      launch_scan_bt(sb, st);
    }
    if (n_items_kind[2]) { // candidates of the generic evaluator
      ScanArgs sg = sa;
      sg.items = b->d_items.p + n_items_pk + n_items_kind[0] + n_items_kind[1];
      sg.n_items = (uint32_t)n_items_kind[2];
      launch_scan_pk(sg, max_terms, true, true, true, st, true);
    }
    // the queued matches of hit-ranked queries: hit pass + state rankers (mrk_rank.hip), behind the scans on the same stream
    if (mq_chunks[0]) launch_rank(sa, 0, st);
    if (mq_chunks[1]) launch_rank(sa, 1, st);
    if (mq_chunks[2] && (rc = launch_gen_rank(sa, any_nearn, st))) return rc;
    if (n_items_kind[0]) {
      ScanArgs sb = sa;
      sb.items = b->d_items.p + n_items_pk;
      sb.n_items = (uint32_t)n_items_kind[0];
      launch_scan_bm(sb, st);
    }
  } else
    launch_scan(sa, st);
  HIP_TRY(hipEventRecord(b->ev_scan1, st));
  lap("scan launched");
  HIP_TRY(hipStreamWaitEvent(st2, b->ev_scan1, 0));

  MergeArgs ma{};
  ma.in_keys = b->d_item_cand.p;
  ma.in_cnt = b->d_item_cnt.p;
  ma.list_first = b->d_list_first.p;
  ma.list_n = b->d_list_n.p;
  ma.n_lists = 0;
  ma.n_queries = n;
  ma.k_per_query = b->d_kq.p;
  ma.k = KCAP;
  ma.out_keys = b->d_out_keys.p;
  ma.out_cnt = b->d_out_cnt.p;
  if (use_packed) {
    SelectArgs se{};
    se.queries = b->d_queries.p;
    se.q_hist = b->d_q_hist.p;
    se.q_cand_n = b->d_q_cand_n.p;
    se.cand = b->d_cand.p;
    se.n_queries = n;
    se.rowid_base = seg->dev.rowid_base;
    se.out_keys = b->d_out_keys.p;
    se.out_cnt = b->d_out_cnt.p;
    se.sel_tau = b->d_sel.p;
    se.sel_nslice = b->d_sel.p + n;
    se.slice_cnt = b->d_sel.p + 2 * (size_t)n;
    se.max_slices = (uint32_t)std::min<uint64_t>(mrk::sel_slice_slots(cand_total, n), 1u << 30);
    sel_rowid_range(seg, se);
    // results straight into the batch's pinned host memory (a batch with a standing rows destination feeds a shard merge:
    // its own lists stay on the device unless mrk_batch_result asks for them)
    se.q_total = b->d_q_total.p;
    se.q_flags = b->d_q_flags.p;
    se.h_flags = b->h_flags.p;
    se.h_cand_n = b->h_cand_n.p;
    if (!b->rows_dst) se.h_keys = b->h_keys.p, se.h_cnt = b->h_cnt.p, se.h_total = b->h_total.p;
    // a standing rows destination (shard exchange): the sort pass writes the exchange rows itself
    se.rows_dst = b->rows_dst;
    se.declined = nullptr;
    launch_select(se, st2);
  } else
    launch_merge(ma, st2);
  HIP_TRY(hipEventRecord(b->ev_merge1, st2));
  {
    bool any = false;
    for (uint32_t i = 0; i < n; ++i) any = any || b->status[i] != MRK_OK;
    b->any_declined = any;
    if (any || b->decl_dirty) {
      for (uint32_t i = 0; i < n; ++i) b->h_decl.p[i] = b->status[i] != MRK_OK ? 1u : 0u;
      HIP_TRY(hipMemcpyAsync(b->d_decl.p, b->h_decl.p, n * 4, hipMemcpyHostToDevice, st2));
    }
    b->decl_dirty = any;
  }
  if (b->rows_dst && (!use_packed || b->any_declined)) { // standing export for the shard exchange (the packed path's sort pass wrote the rows already, unless a query was declined: those rows are marked here)
    PackRowsArgs pa{};
    pa.keys = b->d_out_keys.p;
    pa.cnt = b->d_out_cnt.p;
    pa.total = b->d_q_total.p;
    pa.rows = b->rows_dst;
    pa.flags = use_packed ? b->d_q_flags.p : nullptr;
    pa.declined = b->any_declined ? b->d_decl.p : nullptr;
    pa.n = n;
    launch_pack_rows(pa, st2);
  }
  HIP_TRY(hipGetLastError());
  b->packed_run = use_packed;
  b->last_seg = seg;
  b->n_pass = (uint32_t)n_pass;
  b->last_max_terms = max_terms;
  b->last_prox = any_prox;
  b->last_tree = any_tree;
  b->last_ext = any_ext;
  lap("select launched");
  // ---- results to pinned host memory: the packed path's selection wrote them itself; the VLB path copies
  b->host_copied = b->rows_dst == nullptr;
  if (b->host_copied && !use_packed) {
    HIP_TRY(hipMemcpyAsync(b->h_cnt.p, b->d_out_cnt.p, n * 4, hipMemcpyDeviceToHost, st2));
    HIP_TRY(hipMemcpyAsync(b->h_total.p, b->d_q_total.p, n * 8, hipMemcpyDeviceToHost, st2));
    HIP_TRY(hipMemcpyAsync(b->h_keys.p, b->d_out_keys.p, (size_t)n * KCAP * 8, hipMemcpyDeviceToHost, st2));
  }
  lap("d2h queued");
  b->in_flight = true;
  b->stats.plan_ms = plan_ms;
  b->stats.submit_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_submit0).count();
  return MRK_OK;
}

// One query of the last submit again, with a candidate list sized for every doc of its driver keywords.  The passes
// planned at submit are reused (they hold everything but the work items); results replace the query's host rows.
static int rerun_overflowed(mrk_batch* b, uint32_t qi) {
  mrk_segment* seg = b->last_seg;
  if (!seg) return mrk_fail(MRK_E_INVAL, "query %u: candidate list overflowed and the segment is gone", qi);
  if (!b->retry) {
    int rc = mrk_batch_create_impl(b->ctx, 1 + MAX_PASSES, &b->retry);
    if (rc != MRK_OK) return rc;
  }
  mrk_batch* r = b->retry;
  hipStream_t st = r->stream;
  std::vector<DevQuery> passes;
  passes.push_back(b->h_queries.p[qi]);
  for (uint32_t e = b->n_queries; e < b->n_pass; ++e)
    if (b->h_queries.p[e].out_q == qi) passes.push_back(b->h_queries.p[e]);
  uint64_t cap = 0;
  for (const DevQuery& p : passes) cap += pass_max_matches(p);
  cap = std::max<uint64_t>(cap, 1);
  if (cap > 0xFFFFFFF0ull) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: candidate list overflowed", qi);
  std::vector<DevItem> items_pk, items_bm, items_bt, items_gen;
  for (size_t p = 0; p < passes.size(); ++p) {
    DevQuery& P = passes[p];
    P.out_q = 0;
    P.cand_off = 0;
    P.cand_cap = (uint32_t)cap;
    const bool bm = (P.tree_flags & TF_BITMAP) != 0, bt = (P.tree_flags & TF_BTREE) != 0;
    const uint32_t n = (bm || bt) ? seg->dev.n_windows : P.t[0].nblocks, step = (bm || bt) ? 1024u : 256u;
    for (uint32_t x = 0; x < n; x += step) {
      DevItem it{};
      it.query = (uint32_t)p;
      it.blk_begin = x;
      it.blk_end = std::min(n, x + step);
      it.kind = bt ? 1u : (P.tree_flags & TF_GEN) ? 2u : 0u;
      (bm ? items_bm : bt ? items_bt : (P.tree_flags & TF_GEN) ? items_gen : items_pk).push_back(it);
    }
  }
  const size_t n_pk = items_pk.size(), n_bm = items_bm.size(), n_bt = items_bt.size(), n_gen = items_gen.size();
  items_pk.insert(items_pk.end(), items_bm.begin(), items_bm.end());
  items_pk.insert(items_pk.end(), items_bt.begin(), items_bt.end());
  items_pk.insert(items_pk.end(), items_gen.begin(), items_gen.end());
  const size_t n_items = items_pk.size();
  int rc;
  if ((rc = r->h_queries.reserve(passes.size())) || (rc = r->d_queries.reserve(passes.size())) || (rc = r->h_items.reserve(n_items + 1)) ||
      (rc = r->d_items.reserve(n_items + 1)) || (rc = r->d_cand.reserve(cap + 64)) || (rc = r->d_sel.reserve(2 + mrk::sel_slice_slots(cap, 1))))
    return rc;
  memcpy(r->h_queries.p, passes.data(), passes.size() * sizeof(DevQuery));
  if (n_items) memcpy(r->h_items.p, items_pk.data(), n_items * sizeof(DevItem));
  HIP_TRY(hipMemcpyAsync(r->d_queries.p, r->h_queries.p, passes.size() * sizeof(DevQuery), hipMemcpyHostToDevice, st));
  if (n_items) HIP_TRY(hipMemcpyAsync(r->d_items.p, r->h_items.p, n_items * sizeof(DevItem), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(r->d_state.p, 0, (size_t)r->max_queries * STATE_BYTES + (size_t)NBINS * 4, st));
  ScanArgs sa{};
  sa.seg = seg->dev;
  sa.queries = r->d_queries.p;
  sa.items = r->d_items.p;
  sa.q_total = r->d_q_total.p;
  sa.q_tau = r->d_q_tau.p;
  sa.n_items = (uint32_t)n_pk;
  sa.q_hist = r->d_q_hist.p;
  sa.q_cand_n = r->d_q_cand_n.p;
  sa.q_flags = r->d_q_flags.p;
  sa.q_tau_bin = r->d_q_tau_bin.p;
  sa.cand = r->d_cand.p;
  {
    uint64_t chunks[3] = {0, 0, 0};
    for (size_t p = 0; p < passes.size(); ++p) {
      bool fat = false;
      if (pass_queues_matches(passes[p], fat)) chunks[queue_of(passes[p], fat)] += pass_max_matches(passes[p]) / 64 + 4ull * (mrk::MQ_BATCH + 1) * n_items + 1;
    }
    for (int i = 0; i < 3; ++i)
      if (chunks[i] > (1ull << 25)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: match queue for the rerun too large", qi);
    if ((rc = bind_match_queues(r, chunks, sa))) return rc;
    if (chunks[0] || chunks[1] || chunks[2]) HIP_TRY(hipMemsetAsync(r->d_mq_count.p, 0, 3 * mrk::MQ_SHARDS * 4, st));
    const bool nearn = (passes[0].tree_flags & TF_GEN_NEARN) != 0;
    if (n_gen && (rc = bind_gen(r, b, sa, st, 1, nearn))) return rc; // (the programs are the submit's, still on the device)
    launch_scan_pk(sa, b->last_max_terms, b->last_prox, b->last_tree, b->last_ext, st);
    if (n_gen) {
      ScanArgs sg = sa;
      sg.items = r->d_items.p + n_pk + n_bm + n_bt;
      sg.n_items = (uint32_t)n_gen;
      launch_scan_pk(sg, b->last_max_terms, true, true, true, st, true);
    }
    if (n_bt) {
      ScanArgs sb = sa;
      sb.items = r->d_items.p + n_pk + n_bm;
      sb.n_items = (uint32_t)n_bt;
      launch_scan_bt(sb, st);
    }
    if (chunks[0]) launch_rank(sa, 0, st);
    if (chunks[1]) launch_rank(sa, 1, st);
    if (chunks[2] && (rc = launch_gen_rank(sa, nearn, st))) return rc;
  }
  if (n_bm) {
    ScanArgs sb = sa;
    sb.items = r->d_items.p + n_pk;
    sb.n_items = (uint32_t)n_bm;
    launch_scan_bm(sb, st);
  }
  SelectArgs se{};
  se.queries = r->d_queries.p;
  se.q_hist = r->d_q_hist.p;
  se.q_cand_n = r->d_q_cand_n.p;
  se.cand = r->d_cand.p;
  se.n_queries = 1;
  se.rowid_base = seg->dev.rowid_base;
  se.out_keys = r->d_out_keys.p;
  se.out_cnt = r->d_out_cnt.p;
  se.sel_tau = r->d_sel.p;
  se.sel_nslice = r->d_sel.p + 1;
  se.slice_cnt = r->d_sel.p + 2;
  se.max_slices = (uint32_t)std::min<uint64_t>(mrk::sel_slice_slots(cap, 1), 1u << 30);
  sel_rowid_range(seg, se);
  launch_select(se, st);
  HIP_TRY(hipGetLastError());
  uint32_t flags = 0;
  HIP_TRY(hipMemcpyAsync(&flags, r->d_q_flags.p, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_cnt.p + qi, r->d_out_cnt.p, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_total.p + qi, r->d_q_total.p, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(b->h_keys.p + (size_t)qi * KCAP, r->d_out_keys.p, (size_t)KCAP * 8, hipMemcpyDeviceToHost, st));
  // the batch's device-side results (mrk_batch_device_results / export) get the repaired row too
  HIP_TRY(hipMemcpyAsync(b->d_out_cnt.p + qi, r->d_out_cnt.p, 4, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(b->d_q_total.p + qi, r->d_q_total.p, 8, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(b->d_out_keys.p + (size_t)qi * KCAP, r->d_out_keys.p, (size_t)KCAP * 8, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemsetAsync(b->d_q_flags.p + qi, 0, 4, st)); // the device-side row is good again
  HIP_TRY(hipStreamSynchronize(st));
  if (flags & QF_ARENA) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: the generic evaluator ran out of hit-list memory (ctx tunable gen_spill_mb)", qi);
  if (flags & (QF_OVERFLOW | QF_FSM)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: candidate list overflowed again on the rerun", qi);
  b->decoded = false;
  return MRK_OK;
}

static int mrk_batch_wait_impl(mrk_batch* b) {
  if (!b) return mrk_fail(MRK_E_INVAL, "mrk_batch_wait: NULL batch");
  if (!b->in_flight) return MRK_OK;
  HIP_TRY(hipSetDevice(b->ctx->device));
  static const bool wait_timing = getenv("MRK_SUBMIT_TIMING") != nullptr;
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (wait_timing) fprintf(stderr, "mrk wait   %p end   (abs %.3f)\n", (void*)b, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
  b->in_flight = false;
  if (wait_timing && b->d_mq_count.p) { // how full the hit pass's queues got, and how many queries will run again
    uint32_t cnt[3 * mrk::MQ_SHARDS];
    if (hipMemcpy(cnt, b->d_mq_count.p, sizeof cnt, hipMemcpyDeviceToHost) == hipSuccess) {
      uint64_t sum[3] = {0, 0, 0};
      for (int i = 0; i < 3 * mrk::MQ_SHARDS; ++i) sum[i / mrk::MQ_SHARDS] += cnt[i];
      uint32_t over = 0;
      if (b->packed_run)
        for (uint32_t i = 0; i < b->n_queries; ++i) over += (b->h_flags.p[i] & QF_OVERFLOW) != 0;
      fprintf(stderr, "mrk wait   %p match-queue chunks reserved %llu / %llu / %llu, queries to rerun %u\n", (void*)b, (unsigned long long)sum[0],
              (unsigned long long)sum[1], (unsigned long long)sum[2], over);
    }
  }
  float ms = 0;
  if (hipEventElapsedTime(&ms, b->ev_scan0, b->ev_scan1) == hipSuccess) b->stats.scan_ms = ms;
  if (hipEventElapsedTime(&ms, b->ev_scan1, b->ev_merge1) == hipSuccess) b->stats.merge_ms = ms;
  b->stats.n_cands = 0;
  if (b->packed_run)
    for (uint32_t i = 0; i < b->n_queries; ++i) b->stats.n_cands += b->h_cand_n.p[i];
  if (b->packed_run)
    for (uint32_t i = 0; i < b->n_queries; ++i) {
      if (b->status[i] != MRK_OK) continue;
      // never hand back a silently truncated result
      if (b->h_flags.p[i] & QF_FSM)
        b->status[i] = mrk_fail(MRK_E_UNSUPPORTED, "query %u: a doc held more live phrase states than the device path keeps", i);
      else if (b->h_flags.p[i] & QF_ARENA)
        b->status[i] = mrk_fail(MRK_E_UNSUPPORTED, "query %u: the generic evaluator ran out of hit-list memory (ctx tunable gen_spill_mb)", i);
      else if (b->h_flags.p[i] & QF_OVERFLOW) {
        // more matches tied at the pruning threshold than the candidate list holds (e.g. millions of docs with the
        // very same weight): run the query again, alone, with room for every doc its drivers can deliver
        const int rc = rerun_overflowed(b, i);
        if (rc != MRK_OK) b->status[i] = rc;
      }
    }
  return MRK_OK;
}

static int mrk_batch_result_impl(mrk_batch* b, uint32_t q, mrk_result* out) {
  if (!b || !out) return mrk_fail(MRK_E_INVAL, "mrk_batch_result: NULL argument");
  if (b->in_flight) return mrk_fail(MRK_E_INVAL, "mrk_batch_result: call mrk_batch_wait first");
  if (q >= b->n_queries) return mrk_fail(MRK_E_INVAL, "mrk_batch_result: query %u of %u", q, b->n_queries);
  if (!b->host_copied) {
    const size_t n = b->n_queries;
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipMemcpyAsync(b->h_cnt.p, b->d_out_cnt.p, n * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->h_total.p, b->d_q_total.p, n * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(b->h_keys.p, b->d_out_keys.p, n * KCAP * 8, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->host_copied = true;
  }
  if (!b->decoded) {
    for (uint32_t i = 0; i < b->n_queries; ++i) {
      const uint32_t cnt = b->status[i] == MRK_OK ? std::min<uint32_t>(b->h_cnt.p[i], KCAP) : 0;
      const uint64_t* keys = b->h_keys.p + (size_t)i * KCAP;
      const uint32_t rb = b->rowid_base; // report segment-local rowids, as the reference's sorter holds them
      for (uint32_t j = 0; j < cnt; ++j) {
        b->rowid[(size_t)i * KCAP + j] = key_rowid(keys[j]) - rb;
        b->weight[(size_t)i * KCAP + j] = key_weight(keys[j]);
      }
    }
    b->decoded = true;
  }
  out->status = b->status[q];
  out->n = b->status[q] == MRK_OK ? (int32_t)std::min<uint32_t>(b->h_cnt.p[q], KCAP) : 0;
  out->total_found = b->status[q] == MRK_OK ? (int64_t)b->h_total.p[q] : 0;
  out->rowid = b->rowid.data() + (size_t)q * KCAP;
  out->weight = b->weight.data() + (size_t)q * KCAP;
  return MRK_OK;
}

extern "C" int mrk_batch_stats_get(mrk_batch* b, mrk_batch_stats* out) {
  if (!b || !out) return mrk_fail(MRK_E_INVAL, "mrk_batch_stats_get: NULL argument");
  *out = b->stats;
  return MRK_OK;
}

extern "C" int mrk_batch_device_results(mrk_batch* b, const uint64_t** keys, const uint32_t** counts, const uint64_t** totals) {
  if (!b) return mrk_fail(MRK_E_INVAL, "mrk_batch_device_results: NULL batch");
  if (b->in_flight) return mrk_fail(MRK_E_INVAL, "mrk_batch_device_results: call mrk_batch_wait first");
  if (keys) *keys = b->d_out_keys.p;
  if (counts) *counts = b->d_out_cnt.p;
  if (totals) *totals = b->d_q_total.p;
  return MRK_OK;
}

static int mrk_batch_export_device_impl(mrk_batch* b, uint64_t* keys_dst, uint32_t* counts_dst, uint64_t* totals_dst) {
  if (!b) return mrk_fail(MRK_E_INVAL, "mrk_batch_export_device: NULL batch");
  HIP_TRY(hipSetDevice(b->ctx->device));
  hipStream_t st = b->stream;
  const size_t n = b->n_queries;
  if (keys_dst) HIP_TRY(hipMemcpyAsync(keys_dst, b->d_out_keys.p, n * KCAP * 8, hipMemcpyDeviceToDevice, st));
  if (counts_dst) HIP_TRY(hipMemcpyAsync(counts_dst, b->d_out_cnt.p, n * 4, hipMemcpyDeviceToDevice, st));
  if (totals_dst) HIP_TRY(hipMemcpyAsync(totals_dst, b->d_q_total.p, n * 8, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  b->in_flight = false;
  return MRK_OK;
}

extern "C" int mrk_batch_set_rows_dst(mrk_batch* b, uint64_t* rows_dst) {
  if (!b) return mrk_fail(MRK_E_INVAL, "mrk_batch_set_rows_dst: NULL batch");
  b->rows_dst = rows_dst;
  return MRK_OK;
}

static int mrk_batch_record_event_impl(mrk_batch* b, void* hip_event) {
  if (!b || !hip_event) return mrk_fail(MRK_E_INVAL, "mrk_batch_record_event: NULL argument");
  HIP_TRY(hipSetDevice(b->ctx->device));
  HIP_TRY(hipEventRecord((hipEvent_t)hip_event, b->stream));
  return MRK_OK;
}

static int mrk_batch_export_rows_impl(mrk_batch* b, uint64_t* rows_dst) {
  if (!b || !rows_dst) return mrk_fail(MRK_E_INVAL, "mrk_batch_export_rows: NULL argument");
  HIP_TRY(hipSetDevice(b->ctx->device));
  PackRowsArgs pa{};
  pa.keys = b->d_out_keys.p;
  pa.cnt = b->d_out_cnt.p;
  pa.total = b->d_q_total.p;
  pa.rows = rows_dst;
  pa.flags = b->packed_run ? b->d_q_flags.p : nullptr;
  pa.declined = b->any_declined ? b->d_decl.p : nullptr;
  pa.n = b->n_queries;
  launch_pack_rows(pa, b->stream); // behind the batch's selection kernel
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->in_flight = false;
  return MRK_OK;
}

// rows of n_lists shards -> merged rows; lists [l][list_stride][ROW_WORDS], queries [0, n_queries) of each, out rows at out_first + q
static void launch_rows_merge(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t list_stride, uint32_t n_queries, uint32_t k,
                              uint64_t* out_rows, uint32_t out_first, uint32_t* flags_any) {
  if (n_lists <= 8) {
    mrk::MergeRowsArgs mr{};
    mr.in_rows = rows_all, mr.n_lists = n_lists, mr.list_stride = list_stride, mr.n_queries = n_queries, mr.k = k;
    mr.out_rows = out_rows, mr.out_first = out_first, mr.flags_any = flags_any;
    launch_merge_rows(mr, ctx->merge_stream);
    return;
  }
  MergeArgs ma{}; // many lists (one GPU serving many segments): the general kernel; same layout only when the stride is the query count
  ma.in_rows = rows_all;
  ma.out_rows = out_rows + (size_t)out_first * ROW_WORDS;
  ma.n_lists = n_lists;
  ma.n_queries = n_queries;
  ma.k = k;
  launch_merge(ma, ctx->merge_stream);
}

static int mrk_topk_merge_rows_impl(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                                   uint64_t* out_rows) {
  if (!ctx || !rows_all || !out_rows) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows: NULL argument");
  if (k == 0 || k > MRK_MAX_K) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows: k %u outside 1..%d", k, MRK_MAX_K);
  HIP_TRY(hipSetDevice(ctx->device));
  launch_rows_merge(ctx, rows_all, n_lists, n_queries, n_queries, k, out_rows, 0, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->merge_stream));
  return MRK_OK;
}

// what a rank of the query-partitioned exchange does with its receive buffer: lists [n_lists][list_stride] rows, the rank's
// `count` queries, merged rows written at out_rows[first + q]
static int mrk_topk_merge_rows_part_impl(mrk_ctx* ctx, const uint64_t* rows_recv, uint32_t n_lists, uint32_t list_stride, uint32_t first,
                                        uint32_t count, uint32_t k, uint64_t* out_rows) {
  if (!ctx || !rows_recv || !out_rows) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows_part: NULL argument");
  if (k == 0 || k > MRK_MAX_K) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows_part: k %u outside 1..%d", k, MRK_MAX_K);
  if (n_lists == 0 || n_lists > 8 || count > list_stride) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows_part: %u lists (1..8), %u queries of stride %u", n_lists, count, list_stride);
  HIP_TRY(hipSetDevice(ctx->device));
  launch_rows_merge(ctx, rows_recv, n_lists, list_stride, count, k, out_rows, first, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->merge_stream));
  return MRK_OK;
}

static int mrk_topk_merge_rows_async_impl(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                                         uint64_t* out_rows, void* wait_event, uint32_t slot) {
  if (!ctx || !rows_all || !out_rows) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows_async: NULL argument");
  if (k == 0 || k > MRK_MAX_K) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows_async: k %u outside 1..%d", k, MRK_MAX_K);
  if (slot >= MRK_MERGE_SLOTS) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge_rows_async: slot %u of %d", slot, MRK_MERGE_SLOTS);
  HIP_TRY(hipSetDevice(ctx->device));
  if (!ctx->merge_done[slot]) HIP_TRY(hipEventCreateWithFlags(&ctx->merge_done[slot], hipEventDisableTiming));
  if (wait_event) HIP_TRY(hipStreamWaitEvent(ctx->merge_stream, (hipEvent_t)wait_event, 0));
  launch_rows_merge(ctx, rows_all, n_lists, n_queries, n_queries, k, out_rows, 0, nullptr);
  HIP_TRY(hipGetLastError());

  HIP_TRY(hipEventRecord(ctx->merge_done[slot], ctx->merge_stream));
  ctx->merge_used[slot] = true;
  return MRK_OK;
}

static int mrk_merge_wait_impl(mrk_ctx* ctx, uint32_t slot) {
  if (!ctx || slot >= MRK_MERGE_SLOTS) return mrk_fail(MRK_E_INVAL, "mrk_merge_wait: bad argument");
  if (!ctx->merge_used[slot]) return MRK_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventSynchronize(ctx->merge_done[slot]));
  return MRK_OK;
}

static int mrk_topk_merge_impl(mrk_ctx* ctx, const uint64_t* in_keys, const uint32_t* in_counts, uint32_t n_lists,
                              uint32_t n_queries, uint32_t k, uint64_t* out_keys, uint32_t* out_counts) {
  if (!ctx || !in_keys || !in_counts || !out_keys || !out_counts) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge: NULL argument");
  if (k == 0 || k > MRK_MAX_K) return mrk_fail(MRK_E_INVAL, "mrk_topk_merge: k %u outside 1..%d", k, MRK_MAX_K);
  HIP_TRY(hipSetDevice(ctx->device));
  MergeArgs ma{};
  ma.in_keys = in_keys;
  ma.in_cnt = in_counts;
  ma.list_first = nullptr;
  ma.list_n = nullptr;
  ma.n_lists = n_lists;
  ma.n_queries = n_queries;
  ma.k_per_query = nullptr;
  ma.k = k;
  ma.out_keys = out_keys;
  ma.out_cnt = out_counts;
  launch_merge(ma, ctx->merge_stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->merge_stream));
  return MRK_OK;
}

// ----------------------------------------------------------------------------------------
// The C-ABI proper: every entry point that reaches the HIP runtime runs its body on the context's SUBMISSION THREAD.
//
// The reference runs rankers on 128 KB coroutine stacks (coroutine.cpp:47) and many of them at once on a thread pool
// (searchd.cpp:5654); HIP runtime calls need a real thread stack, and hipSetDevice is per thread.  So a context owns one
// worker thread: callers post a closure and sleep until it has run (plain mutex + condition variable on the caller's
// side: nothing of HIP on the caller's stack), the worker runs closures in arrival order.  That also serializes all
// host-side state of a context (its streams, merge slots, the segment tables): batches driven from different threads
// are safe by construction.  A wait never parks the worker inside hipStreamSynchronize while other work is queued: it
// polls the batch's stream and steps aside (see mrk_batch_wait below), so one thread's wait does not hold up another
// thread's submit.  MRK_INLINE_HIP=1 (environment, read at mrk_ctx_create) runs everything on the calling thread
// instead -- for debuggers and profilers.
// ----------------------------------------------------------------------------------------
struct mrk_worker {
  std::thread th;
  std::thread::id tid;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  bool stop = false;

  void loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return; // stop requested and nothing left to run
        f = std::move(q.front());
        q.pop_front();
      }
      f();
    }
  }
  size_t pending() {
    std::lock_guard<std::mutex> lk(mu);
    return q.size();
  }
};

// run f (-> int status) on the context's worker and hand back its status; the worker's error text travels with it
template <typename F>
static int on_worker(mrk_ctx* c, F&& f) {
  mrk_worker* w = c ? c->worker : nullptr;
  if (!w || std::this_thread::get_id() == w->tid) return f();
  int rc = MRK_OK;
  char err[sizeof g_err];
  err[0] = 0;
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->q.emplace_back([&] {
      rc = f();
      if (rc != MRK_OK) memcpy(err, g_err, sizeof err);
      std::lock_guard<std::mutex> lk2(m);
      done = true;
      cv.notify_one();
    });
  }
  w->cv.notify_one();
  {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done; });
  }
  if (rc != MRK_OK) memcpy(g_err, err, sizeof g_err);
  return rc;
}

extern "C" int mrk_ctx_create(int device, mrk_ctx** out) {
  if (!out) return mrk_fail(MRK_E_INVAL, "mrk_ctx_create: out is NULL");
  static const bool inline_hip = getenv("MRK_INLINE_HIP") != nullptr;
  if (inline_hip) return mrk_ctx_create_impl(device, out);
  mrk_worker* w = new (std::nothrow) mrk_worker();
  if (!w) return mrk_fail(MRK_E_NOMEM, "out of memory");
  w->th = std::thread([w] { w->loop(); });
  w->tid = w->th.get_id();
  // the context is created ON the worker (hipSetDevice is per thread) through a stand-in that only carries the worker
  mrk_ctx boot;
  boot.worker = w;
  const int rc = on_worker(&boot, [&] { return mrk_ctx_create_impl(device, out); });
  if (rc != MRK_OK) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
    }
    w->cv.notify_one();
    w->th.join();
    delete w;
    return rc;
  }
  (*out)->worker = w;
  return MRK_OK;
}

extern "C" int mrk_ctx_destroy(mrk_ctx* c) {
  if (!c) return MRK_OK;
  // children alive: their destructors would post to a submission thread that is gone and wait forever (the hang of round 2,
  // gpurun_out/hang_lib.txt: Context.close <- __del__ with a Batch still alive).  Refuse, leave everything as it is.
  const int ns = c->n_segments.load(), nb = c->n_batches.load();
  if (ns || nb)
    return mrk_fail(MRK_E_INVAL, "mrk_ctx_destroy: %d segment(s) and %d batch(es) of this context are still alive; destroy them first", ns, nb);
  mrk_worker* w = c->worker;
  if (!w) {
    mrk_ctx_destroy_impl(c);
    return MRK_OK;
  }
  mrk_ctx boot;
  boot.worker = w;
  (void)on_worker(&boot, [&] {
    mrk_ctx_destroy_impl(c);
    return MRK_OK;
  });
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->stop = true;
  }
  w->cv.notify_one();
  w->th.join();
  delete w;
  return MRK_OK;
}

extern "C" int mrk_segment_create(mrk_ctx* ctx, const mrk_segment_desc* d, mrk_segment** out) {
  const int rc = on_worker(ctx, [&] { return mrk_segment_create_impl(ctx, d, out); });
  if (rc == MRK_OK) ++ctx->n_segments;
  return rc;
}
extern "C" void mrk_segment_destroy(mrk_segment* s) {
  if (!s) return;
  mrk_ctx* ctx = s->ctx;
  (void)on_worker(ctx, [&] {
    mrk_segment_destroy_impl(s);
    return MRK_OK;
  });
  if (ctx) --ctx->n_segments;
}
extern "C" int mrk_segment_set_dead_rows(mrk_segment* s, const uint32_t* bitmap, uint64_t n_rows) {
  return on_worker(s ? s->ctx : nullptr, [&] { return mrk_segment_set_dead_rows_impl(s, bitmap, n_rows); });
}
extern "C" int mrk_segment_set_attrs(mrk_segment* s, const uint32_t* rows, uint32_t stride, uint64_t n_rows) {
  return on_worker(s ? s->ctx : nullptr, [&] { return mrk_segment_set_attrs_impl(s, rows, stride, n_rows); });
}
extern "C" int mrk_segment_set_blobs(mrk_segment* s, const uint8_t* pool, uint64_t len, uint32_t n_blob, const uint32_t* rows, uint32_t stride, uint64_t n_rows) {
  return on_worker(s ? s->ctx : nullptr, [&] { return mrk_segment_set_blobs_impl(s, pool, len, n_blob, rows, stride, n_rows); });
}
extern "C" int mrk_batch_create(mrk_ctx* ctx, uint32_t max_queries, mrk_batch** out) {
  const int rc = on_worker(ctx, [&] { return mrk_batch_create_impl(ctx, max_queries, out); });
  if (rc == MRK_OK) ++ctx->n_batches;
  return rc;
}
extern "C" void mrk_batch_destroy(mrk_batch* b) {
  if (!b) return;
  mrk_ctx* ctx = b->ctx;
  (void)on_worker(ctx, [&] {
    mrk_batch_destroy_impl(b);
    return MRK_OK;
  });
  --ctx->n_batches;
}
extern "C" int mrk_batch_submit(mrk_batch* b, mrk_segment* seg, const mrk_query* queries, uint32_t n) {
  return on_worker(b ? b->ctx : nullptr, [&] { return mrk_batch_submit_impl(b, seg, queries, n); });
}

// The wait polls instead of parking the worker in hipStreamSynchronize: while the batch's stream is still busy AND
// other closures are queued (another thread's submit), the attempt steps aside and the caller tries again; with
// nothing else queued the worker keeps polling itself, so a lone caller sees the result as soon as the stream drains.
extern "C" int mrk_batch_wait(mrk_batch* b) {
  if (!b) return mrk_fail(MRK_E_INVAL, "mrk_batch_wait: NULL batch");
  mrk_worker* w = b->ctx ? b->ctx->worker : nullptr;
  static const bool wait_timing = getenv("MRK_SUBMIT_TIMING") != nullptr;
  if (wait_timing) fprintf(stderr, "mrk wait   %p begin (abs %.3f)\n", (void*)b, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
  if (!w) return mrk_batch_wait_impl(b);
  constexpr int NOT_READY = 1;
  for (;;) {
    const int rc = on_worker(b->ctx, [&]() -> int {
      if (!b->in_flight) return MRK_OK;
      for (;;) {
        const hipError_t e = hipStreamQuery(b->stream);
        if (e == hipSuccess) return mrk_batch_wait_impl(b); // (drained: the synchronize inside returns at once)
        if (e != hipErrorNotReady) return mrk_fail(MRK_E_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
        if (w->pending()) return NOT_READY; // let the queued closures run
        std::this_thread::yield();
      }
    });
    if (rc != NOT_READY) return rc;
    std::this_thread::yield();
  }
}

extern "C" int mrk_batch_test(mrk_batch* b) {
  if (!b) return mrk_fail(MRK_E_INVAL, "mrk_batch_test: NULL batch");
  if (!b->in_flight) return MRK_OK;
  static thread_local int device_set = -1;
  if (device_set != b->ctx->device) {
    HIP_TRY(hipSetDevice(b->ctx->device));
    device_set = b->ctx->device;
  }
  const hipError_t e = hipStreamQuery(b->stream);
  if (e == hipSuccess) return MRK_OK;
  if (e == hipErrorNotReady) return 1;
  return mrk_fail(MRK_E_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
}

extern "C" int mrk_batch_result(mrk_batch* b, uint32_t q, mrk_result* out) {
  // results already in host memory: plain reads, no hop to the submission thread (16 callers' rows used to cost 16 hops)
  if (b && b->host_copied && !b->in_flight) return mrk_batch_result_impl(b, q, out);
  return on_worker(b ? b->ctx : nullptr, [&] { return mrk_batch_result_impl(b, q, out); });
}
extern "C" int mrk_batch_export_device(mrk_batch* b, uint64_t* keys_dst, uint32_t* counts_dst, uint64_t* totals_dst) {
  return on_worker(b ? b->ctx : nullptr, [&] { return mrk_batch_export_device_impl(b, keys_dst, counts_dst, totals_dst); });
}
extern "C" int mrk_batch_record_event(mrk_batch* b, void* hip_event) {
  return on_worker(b ? b->ctx : nullptr, [&] { return mrk_batch_record_event_impl(b, hip_event); });
}
extern "C" int mrk_batch_export_rows(mrk_batch* b, uint64_t* rows_dst) {
  return on_worker(b ? b->ctx : nullptr, [&] { return mrk_batch_export_rows_impl(b, rows_dst); });
}
extern "C" int mrk_topk_merge_rows(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t n_queries, uint32_t k, uint64_t* out_rows) {
  return on_worker(ctx, [&] { return mrk_topk_merge_rows_impl(ctx, rows_all, n_lists, n_queries, k, out_rows); });
}
extern "C" int mrk_topk_merge_rows_async(mrk_ctx* ctx, const uint64_t* rows_all, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                                         uint64_t* out_rows, void* wait_event, uint32_t slot) {
  return on_worker(ctx, [&] { return mrk_topk_merge_rows_async_impl(ctx, rows_all, n_lists, n_queries, k, out_rows, wait_event, slot); });
}
extern "C" int mrk_topk_merge_rows_part(mrk_ctx* ctx, const uint64_t* rows_recv, uint32_t n_lists, uint32_t list_stride, uint32_t first, uint32_t count,
                                        uint32_t k, uint64_t* out_rows) {
  return on_worker(ctx, [&] { return mrk_topk_merge_rows_part_impl(ctx, rows_recv, n_lists, list_stride, first, count, k, out_rows); });
}
extern "C" int mrk_merge_wait(mrk_ctx* ctx, uint32_t slot) {
  return on_worker(ctx, [&] { return mrk_merge_wait_impl(ctx, slot); });
}
extern "C" int mrk_topk_merge(mrk_ctx* ctx, const uint64_t* in_keys, const uint32_t* in_counts, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                              uint64_t* out_keys, uint32_t* out_counts) {
  return on_worker(ctx, [&] { return mrk_topk_merge_impl(ctx, in_keys, in_counts, n_lists, n_queries, k, out_keys, out_counts); });
}

// ---- the shard exchange (mrk_comm.cpp) ----
extern "C" int mrk_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return mrk_fail(MRK_E_INVAL, "mrk_comm_unique_id: NULL argument");
  return mrk_comm_unique_id_impl(id_out); // (no device work: ncclGetUniqueId only opens a socket)
}
extern "C" int mrk_comm_init(mrk_ctx* ctx, const uint8_t* id, int n_ranks, int rank) {
  if (!ctx || !id) return mrk_fail(MRK_E_INVAL, "mrk_comm_init: NULL argument");
  return on_worker(ctx, [&] { return mrk_comm_init_impl(ctx, id, n_ranks, rank); });
}
extern "C" void mrk_comm_destroy(mrk_ctx* ctx) {
  if (ctx) (void)on_worker(ctx, [&] {
    mrk_comm_destroy_impl(ctx);
    return MRK_OK;
  });
}
extern "C" int mrk_comm_allreduce_i64(mrk_ctx* ctx, int64_t* values, uint64_t n) {
  if (!ctx || (!values && n)) return mrk_fail(MRK_E_INVAL, "mrk_comm_allreduce_i64: NULL argument");
  return on_worker(ctx, [&] { return mrk_comm_allreduce_i64_impl(ctx, values, n); });
}
extern "C" int mrk_shard_exchange(mrk_ctx* ctx, mrk_batch* batch, const uint64_t* rows, uint32_t n_queries, uint32_t k, uint64_t* out_rows,
                                  uint32_t slot) {
  if (!ctx || !rows || !out_rows) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: NULL argument");
  if (slot >= MRK_MERGE_SLOTS) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: slot %u of %d", slot, MRK_MERGE_SLOTS);
  if (batch && batch->ctx != ctx) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: batch and context do not belong together");
  return on_worker(ctx, [&]() -> int {
    // behind everything the batch's last submit queued on its stream (selection, the standing rows export)
    hipEvent_t after = nullptr;
    if (batch) {
      after = mrk_comm_rows_ready_event(ctx);
      if (!after) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: no communicator (mrk_comm_init)");
      int rc = mrk_batch_record_event_impl(batch, (void*)after);
      if (rc != MRK_OK) return rc;
    }
    const uint64_t* rows_all = nullptr;
    hipEvent_t gathered = nullptr;
    if (ctx->exchange_part && mrk_comm_can_partition(ctx) && mrk_comm_ranks(ctx) <= 8) {
      // partitioned by query: this rank receives and merges its slice only (mrk_comm.cpp)
      uint32_t per = 0, first = 0, count = 0;
      int rc = mrk_comm_exchange_part_impl(ctx, rows, n_queries, after, slot, &rows_all, &gathered, &per, &first, &count);
      if (rc != MRK_OK) return rc;
      if (k == 0 || k > MRK_MAX_K) return mrk_fail(MRK_E_INVAL, "mrk_shard_exchange: k %u outside 1..%d", k, MRK_MAX_K);
      uint32_t* flags_dev = nullptr;
      HIP_TRY(hipStreamWaitEvent(ctx->merge_stream, gathered, 0));
      if ((rc = mrk_comm_flags_begin(ctx, slot, &flags_dev))) return rc;
      if (count) launch_rows_merge(ctx, rows_all, (uint32_t)mrk_comm_ranks(ctx), per, count, k, out_rows, first, flags_dev);
      HIP_TRY(hipGetLastError());
      return mrk_comm_flags_finish(ctx, slot);
    }
    int rc = mrk_comm_exchange_impl(ctx, rows, n_queries, after, slot, &rows_all, &gathered);
    if (rc != MRK_OK) return rc;
    return mrk_topk_merge_rows_async_impl(ctx, rows_all, (uint32_t)mrk_comm_ranks(ctx), n_queries, k, out_rows, (void*)gathered, slot);
  });
}
extern "C" int mrk_shard_slice(uint32_t n_queries, int n_ranks, int rank, uint32_t* first, uint32_t* count) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return mrk_fail(MRK_E_INVAL, "mrk_shard_slice: rank %d of %d", rank, n_ranks);
  mrk_shard_slice_impl(n_queries, n_ranks, rank, nullptr, first, count);
  return MRK_OK;
}
extern "C" int mrk_shard_flags(mrk_ctx* ctx, uint32_t slot, uint32_t* rerun_any, uint32_t* declined_any) {
  if (!ctx || slot >= MRK_MERGE_SLOTS) return mrk_fail(MRK_E_INVAL, "mrk_shard_flags: bad argument");
  return mrk_comm_flags_read(ctx, slot, rerun_any, declined_any);
}
extern "C" int mrk_shard_partitioned(mrk_ctx* ctx) { return ctx && ctx->exchange_part && mrk_comm_can_partition(ctx) && mrk_comm_ranks(ctx) <= 8 ? 1 : 0; }
