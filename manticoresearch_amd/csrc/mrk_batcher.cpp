// mrk_batcher.cpp -- the batching front of the binding: concurrent callers hand in ONE query each, the batcher sends whatever
// has arrived down as one mrk_batch_submit and hands every caller its own rows.
//
// The reference runs one ranker per (query x index) on a pool of worker threads (SearchHandler_c::RunLocalSearches,
// searchd.cpp:5596-5797; CoExecuteN :5654) -- a ranker that submits a batch of one pays a whole launch chain per query and
// sees the single-query latency, not the batch throughput.  Here the workers' queries meet: while the device works on the
// launch before, new arrivals queue up and leave together as the next one (no timer needed under load; `max_wait_us` only
// delays a launch when the device is idle and the batch is not full).  Up to SLOTS launches are in flight, so planning and
// submitting the next overlaps the kernels of the previous.  Host code only: everything device-side goes through the C-ABI.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/mrk.h"

int mrk_fail(int code, const char* fmt, ...);

namespace {

struct Req {
  mrk_segment* seg;
  const mrk_query* q;
  uint32_t* rowid;
  int32_t* weight;
  int32_t cap;
  mrk_result* res;
  int rc = MRK_OK;
  char err[256] = {0};
  bool done = false;
  std::condition_variable cv;
  std::chrono::steady_clock::time_point t_in;
};

constexpr int SLOTS = 3;

struct Slot {
  mrk_batch* batch = nullptr;
  std::vector<Req*> reqs;
  std::chrono::steady_clock::time_point t_submitted;
};

} // namespace

struct mrk_batcher {
  mrk_ctx* ctx = nullptr;
  uint32_t max_batch = 0, max_wait_us = 0;
  std::mutex mu;
  std::condition_variable cv; // driver: work arrived / stop
  std::deque<Req*> pending;
  Slot slots[SLOTS];
  std::deque<int> in_flight; // slot indices, oldest first
  bool stop = false;
  std::thread driver;
  mrk_batcher_stats stats{};

  void finish(Req* r, int rc, const char* err) {
    std::lock_guard<std::mutex> lk(mu);
    r->rc = rc;
    if (err) snprintf(r->err, sizeof r->err, "%s", err);
    r->done = true;
    r->cv.notify_one();
  }

  // one request through a slot's batch on its own (the fallback when a common launch was refused)
  void run_alone(Slot& s, Req* r) {
    int rc = mrk_batch_submit(s.batch, r->seg, r->q, 1);
    if (rc == MRK_OK) rc = mrk_batch_wait(s.batch);
    if (rc != MRK_OK) return finish(r, rc, mrk_last_error());
    deliver(s, r, 0);
  }

  void deliver(Slot& s, Req* r, uint32_t qi) {
    mrk_result res{};
    const int rc = mrk_batch_result(s.batch, qi, &res);
    if (rc != MRK_OK) return finish(r, rc, mrk_last_error());
    const int32_t n = res.n < r->cap ? res.n : r->cap;
    if (n > 0) {
      memcpy(r->rowid, res.rowid, (size_t)n * sizeof(uint32_t));
      memcpy(r->weight, res.weight, (size_t)n * sizeof(int32_t));
    }
    r->res->n = n;
    r->res->total_found = res.total_found;
    r->res->rowid = r->rowid;
    r->res->weight = r->weight;
    r->res->status = res.status;
    finish(r, MRK_OK, res.status != MRK_OK ? "the device path declined this query (MRK_E_UNSUPPORTED): keep the CPU ranker" : nullptr);
  }

  // The driver is an event loop: (1) collect the oldest launch if the device has finished it (mrk_batch_test: a stream query,
  // no blocking), (2) send the next launch down when a batch is due -- full, or its oldest query has waited max_wait_us, and a
  // slot is free --, (3) otherwise sleep until the next deadline / arrival, polling in short steps while launches are in flight.
  // Never blocks on the device while callers could be leaving: a completion wakes its callers at once, they come back with
  // their next queries inside the window, and the next launch carries all of them.
  void loop() {
    std::vector<mrk_query> flat;
    for (;;) {
      int oldest = -1;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (stop && pending.empty() && in_flight.empty()) return;
        if (!in_flight.empty()) oldest = in_flight.front();
      }
      if (oldest >= 0 && mrk_batch_test(slots[oldest].batch) == MRK_OK) {
        Slot& s = slots[oldest];
        const auto t_c0 = std::chrono::steady_clock::now();
        const int rc = mrk_batch_wait(s.batch); // (drained: returns at once; reruns an overflowed query)
        if (rc != MRK_OK) {
          const char* e = mrk_last_error();
          for (Req* r : s.reqs) finish(r, rc, e);
        } else
          for (size_t i = 0; i < s.reqs.size(); ++i) deliver(s, s.reqs[i], (uint32_t)i);
        s.reqs.clear();
        std::lock_guard<std::mutex> lk(mu);
        in_flight.pop_front();
        stats.collect_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c0).count();
        stats.flight_ms += std::chrono::duration<double, std::milli>(t_c0 - s.t_submitted).count();
        continue;
      }
      int free_slot = -1;
      std::vector<Req*> take;
      {
        std::unique_lock<std::mutex> lk(mu);
        const auto now = std::chrono::steady_clock::now();
        const bool due = !pending.empty() && (stop || !max_wait_us || pending.size() >= max_batch ||
                                              now >= pending.front()->t_in + std::chrono::microseconds(max_wait_us));
        if (due && in_flight.size() < (size_t)SLOTS) {
          for (int i = 0; i < SLOTS && free_slot < 0; ++i) {
            bool busy = false;
            for (int f : in_flight) busy = busy || f == i;
            if (!busy) free_slot = i;
          }
          // one launch = one segment: the oldest request's, and everybody else who asked for the same
          mrk_segment* seg = pending.front()->seg;
          for (auto it = pending.begin(); it != pending.end() && take.size() < max_batch;) {
            if ((*it)->seg == seg) {
              take.push_back(*it);
              it = pending.erase(it);
            } else
              ++it;
          }
        } else if (stop && pending.empty() && in_flight.empty())
          return;
        else if (pending.empty() && in_flight.empty())
          cv.wait(lk, [&] { return stop || !pending.empty(); });
        else if (in_flight.empty() && (!max_wait_us || pending.empty())) {
          const size_t seen = pending.size();
          cv.wait(lk, [&] { return stop || pending.size() != seen; });
        } else {
          // a launch in flight or a window running: spin (a timed sleep wakes 50+ us late -- the kernel's timer slack --, which
          // is the whole run time of a small launch); the driver only ever spins while the device or a caller is waiting on it
          lk.unlock();
          std::this_thread::yield();
        }
      }
      if (free_slot < 0) continue;
      Slot& s = slots[free_slot];
      s.reqs = take;
      flat.clear();
      for (Req* r : take) flat.push_back(*r->q); // (shallow: the callers' trees outlive the launch, they are asleep)
      const auto t_s0 = std::chrono::steady_clock::now();
      const int rc = mrk_batch_submit(s.batch, take[0]->seg, flat.data(), (uint32_t)flat.size());
      s.t_submitted = std::chrono::steady_clock::now();
      if (rc != MRK_OK) {
        for (Req* r : take) run_alone(s, r); // MRK_E_INVAL names one query: the others must not pay for it
        s.reqs.clear();
        std::lock_guard<std::mutex> lk(mu);
        stats.launches += take.size(), stats.queries += take.size();
        continue;
      }
      std::lock_guard<std::mutex> lk(mu);
      in_flight.push_back(free_slot);
      stats.launches += 1, stats.queries += take.size();
      stats.submit_ms += std::chrono::duration<double, std::milli>(s.t_submitted - t_s0).count();
      if (take.size() > stats.max_batch) stats.max_batch = (uint32_t)take.size();
    }
  }
};

extern "C" int mrk_batcher_create(mrk_ctx* ctx, uint32_t max_batch, uint32_t max_wait_us, mrk_batcher** out) {
  if (!ctx || !out || !max_batch) return mrk_fail(MRK_E_INVAL, "mrk_batcher_create: bad argument");
  mrk_batcher* b = new (std::nothrow) mrk_batcher();
  if (!b) return mrk_fail(MRK_E_NOMEM, "out of memory");
  b->ctx = ctx;
  b->max_batch = max_batch;
  b->max_wait_us = max_wait_us;
  for (int i = 0; i < SLOTS; ++i) {
    const int rc = mrk_batch_create(ctx, max_batch, &b->slots[i].batch);
    if (rc != MRK_OK) {
      for (int j = 0; j < i; ++j) mrk_batch_destroy(b->slots[j].batch);
      delete b;
      return rc;
    }
  }
  b->driver = std::thread([b] { b->loop(); });
  *out = b;
  return MRK_OK;
}

extern "C" void mrk_batcher_destroy(mrk_batcher* b) {
  if (!b) return;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    b->stop = true; // (what is queued or in flight is still answered: the driver drains before it leaves)
  }
  b->cv.notify_all();
  if (b->driver.joinable()) b->driver.join();
  for (int i = 0; i < SLOTS; ++i) mrk_batch_destroy(b->slots[i].batch);
  delete b;
}

extern "C" int mrk_batcher_search(mrk_batcher* b, mrk_segment* seg, const mrk_query* q, uint32_t* rowid_out, int32_t* weight_out, int32_t cap,
                                  mrk_result* res) {
  if (!b || !seg || !q || !res || cap < 0 || (cap && (!rowid_out || !weight_out))) return mrk_fail(MRK_E_INVAL, "mrk_batcher_search: bad argument");
  Req r;
  r.seg = seg, r.q = q, r.rowid = rowid_out, r.weight = weight_out, r.cap = cap, r.res = res;
  r.t_in = std::chrono::steady_clock::now();
  memset(res, 0, sizeof *res);
  {
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) return mrk_fail(MRK_E_INVAL, "mrk_batcher_search: the batcher is shutting down");
    b->pending.push_back(&r);
    b->cv.notify_all();
    r.cv.wait(lk, [&] { return r.done; });
  }
  if (r.rc != MRK_OK) return mrk_fail(r.rc, "%s", r.err);
  if (res->status != MRK_OK) (void)mrk_fail(res->status, "%s", r.err); // per-query decline: the text for mrk_last_error on THIS thread
  return MRK_OK;
}

extern "C" int mrk_batcher_stats_get(mrk_batcher* b, mrk_batcher_stats* out) {
  if (!b || !out) return mrk_fail(MRK_E_INVAL, "mrk_batcher_stats_get: NULL argument");
  std::lock_guard<std::mutex> lk(b->mu);
  *out = b->stats;
  return MRK_OK;
}
