// The batching front (mrk_batcher, include/mrk.h): T host threads, ONE query per call, against the 256-query batch.
//   1. every thread's results equal the ones a plain batch returned for the same query (bit for bit);
//   2. a malformed query (MRK_E_INVAL) fails alone, a declined one (MRK_E_UNSUPPORTED) comes back as its own status;
//   3. throughput: T threads x 1 query through the batcher vs the same queries as batches of 256 -- printed, the caller asserts;
//   4. the destroy-order guard: mrk_ctx_destroy with a segment / batch / batcher alive returns MRK_E_INVAL and destroys nothing.
// Usage: test_batcher <n_docs> <threads> <queries per thread> [max_wait_us]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mrk.h"

struct Q {
  mrk_node nodes[3];
  int32_t children[2];
  mrk_query q;
  void point() { q.nodes = nodes, q.children = children; }
};

static void make(Q& x, int a, int b, int ranker, int k) {
  memset(&x, 0, sizeof x);
  x.nodes[0].op = MRK_OP_AND, x.nodes[0].n_children = 2, x.nodes[0].term_id = -1, x.nodes[0].field_mask = MRK_ALL_FIELDS, x.nodes[0].boost = 1.f;
  for (int i = 0; i < 2; ++i) {
    mrk_node& n = x.nodes[1 + i];
    n.op = MRK_OP_TERM, n.term_id = i ? b : a, n.atom_pos = 1 + i, n.field_mask = MRK_ALL_FIELDS, n.boost = 1.f;
    x.children[i] = 1 + i;
  }
  x.q.n_nodes = 3, x.q.root = 0, x.q.ranker = ranker, x.q.max_matches = k, x.q.normalized_tfidf = 1;
  x.point();
}

struct Res {
  int64_t total = 0;
  std::vector<uint32_t> rowid;
  std::vector<int32_t> weight;
};

int main(int argc, char** argv) {
  const uint64_t n_docs = argc > 1 ? strtoull(argv[1], 0, 10) : 2000000;
  const int T = argc > 2 ? atoi(argv[2]) : 16;
  const int per = argc > 3 ? atoi(argv[3]) : 64;
  const uint32_t wait_us = argc > 4 ? (uint32_t)atoi(argv[4]) : 30;
  const int NT = 12;
  double probs[NT];
  for (int i = 0; i < NT; ++i) probs[i] = 0.30 / (1.0 + 0.35 * i); // 0.30 .. 0.06: dense keywords, the headline kernel
  mrk_ctx* ctx = nullptr;
  if (mrk_ctx_create(0, &ctx) != MRK_OK) return fprintf(stderr, "ctx: %s\n", mrk_last_error()), 2;
  mrk_synth_params p{};
  p.seed = 21, p.n_docs = n_docs, p.term_prob = probs, p.n_terms = NT, p.n_fields = 2, p.title_frac = 0.1, p.max_pos = 64;
  p.skiplist_block_size = 128, p.hit_format = MRK_HITFMT_INLINE, p.n_threads = 8;
  mrk_host_index* hi = nullptr;
  if (mrk_synth_generate(&p, &hi) != MRK_OK) return fprintf(stderr, "synth: %s\n", mrk_last_error()), 2;
  mrk_segment_desc d{};
  d.spd = mrk_host_index_spd(hi, &d.spd_len), d.spp = mrk_host_index_spp(hi, &d.spp_len), d.spe = mrk_host_index_spe(hi, &d.spe_len);
  d.dict = mrk_host_index_dict(hi, &d.n_terms);
  d.total_docs = n_docs, d.skiplist_block_size = 128, d.hit_format = MRK_HITFMT_INLINE, d.n_fields = 2;
  mrk_segment* seg = nullptr;
  if (mrk_segment_create(ctx, &d, &seg) != MRK_OK) return fprintf(stderr, "segment: %s\n", mrk_last_error()), 2;

  // ---- 4a. the guard, with a segment alive
  if (mrk_ctx_destroy(ctx) != MRK_E_INVAL || !strstr(mrk_last_error(), "still alive")) return fprintf(stderr, "ctx_destroy with a live segment: no error\n"), 1;

  const int NQ = T * per;
  std::vector<Q> qs(NQ);
  for (int i = 0; i < NQ; ++i) {
    const int a = i % NT, b = (a + 1 + (i / NT) % (NT - 1)) % NT;
    make(qs[i], a, b, MRK_RANK_BM25, 1000);
  }
  for (Q& x : qs) x.point();

  // ---- reference answers + batch throughput: batches of 256 on two alternating mrk_batch objects
  std::vector<Res> want(NQ);
  mrk_batch* bb[2] = {nullptr, nullptr};
  for (int i = 0; i < 2; ++i)
    if (mrk_batch_create(ctx, 256, &bb[i]) != MRK_OK) return fprintf(stderr, "batch: %s\n", mrk_last_error()), 2;
  auto collect = [&](mrk_batch* b, int first, int n) {
    if (mrk_batch_wait(b) != MRK_OK) return false;
    for (int i = 0; i < n; ++i) {
      mrk_result r;
      if (mrk_batch_result(b, (uint32_t)i, &r) != MRK_OK || r.status != MRK_OK) return false;
      want[first + i].total = r.total_found;
      want[first + i].rowid.assign(r.rowid, r.rowid + r.n);
      want[first + i].weight.assign(r.weight, r.weight + r.n);
    }
    return true;
  };
  double batch_qps = 0;
  for (int pass = 0; pass < 2; ++pass) { // pass 0 warms up
    std::vector<mrk_query> flat(NQ);
    for (int i = 0; i < NQ; ++i) flat[i] = qs[i].q;
    const auto t0 = std::chrono::steady_clock::now();
    int pend_first[2] = {-1, -1}, pend_n[2] = {0, 0};
    int k = 0;
    for (int first = 0; first < NQ; first += 256, ++k) {
      const int n = NQ - first < 256 ? NQ - first : 256, s = k & 1;
      if (pend_first[s] >= 0 && !collect(bb[s], pend_first[s], pend_n[s])) return fprintf(stderr, "batch run: %s\n", mrk_last_error()), 2;
      if (mrk_batch_submit(bb[s], seg, flat.data() + first, (uint32_t)n) != MRK_OK) return fprintf(stderr, "submit: %s\n", mrk_last_error()), 2;
      pend_first[s] = first, pend_n[s] = n;
    }
    for (int s = 0; s < 2; ++s)
      if (pend_first[s] >= 0 && !collect(bb[s], pend_first[s], pend_n[s])) return fprintf(stderr, "batch run: %s\n", mrk_last_error()), 2;
    batch_qps = NQ / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  // ---- 4b. the guard, with batches alive too; then the right order works
  if (mrk_ctx_destroy(ctx) != MRK_E_INVAL) return fprintf(stderr, "ctx_destroy with live batches: no error\n"), 1;
  mrk_batch_destroy(bb[0]), mrk_batch_destroy(bb[1]);

  // ---- the batcher: T threads, one query per call
  mrk_batcher* bt = nullptr;
  if (mrk_batcher_create(ctx, 256, wait_us, &bt) != MRK_OK) return fprintf(stderr, "batcher: %s\n", mrk_last_error()), 2;
  if (mrk_ctx_destroy(ctx) != MRK_E_INVAL) return fprintf(stderr, "ctx_destroy with a live batcher: no error\n"), 1;
  std::atomic<int> bad{0};
  std::vector<std::string> errs(T);
  double batcher_qps = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        std::vector<uint32_t> rowid(1000);
        std::vector<int32_t> weight(1000);
        for (int j = 0; j < per && !bad; ++j) {
          const int i = j * T + t; // (threads interleave over the query list)
          mrk_result r;
          if (mrk_batcher_search(bt, seg, &qs[i].q, rowid.data(), weight.data(), 1000, &r) != MRK_OK || r.status != MRK_OK) {
            errs[t] = std::string("search: ") + mrk_last_error();
            ++bad;
            return;
          }
          if (r.total_found != want[i].total || (size_t)r.n != want[i].rowid.size() || memcmp(r.rowid, want[i].rowid.data(), r.n * 4) ||
              memcmp(r.weight, want[i].weight.data(), r.n * 4)) {
            errs[t] = "query " + std::to_string(i) + ": the batcher's rows differ from the batch's";
            ++bad;
            return;
          }
        }
      });
    for (auto& x : th) x.join();
    batcher_qps = NQ / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (bad) {
    for (auto& e : errs)
      if (!e.empty()) fprintf(stderr, "FAILED: %s\n", e.c_str());
    return 1;
  }
  // ---- 2. a malformed query next to good ones: it alone fails; a declined one reports its status
  {
    Q badq = qs[0], declq = qs[1], good = qs[2];
    badq.point(), declq.point(), good.point();
    badq.q.root = 7;                        // outside the tree: MRK_E_INVAL in the planner
    declq.q.max_matches = 1000, declq.q.cutoff = 5000; // cutoff beyond the device top-K: MRK_E_UNSUPPORTED for this query
    int rc_bad = 0, rc_decl = 0, rc_good = 0;
    mrk_result r_bad, r_decl, r_good;
    std::vector<uint32_t> ra(1000), rb(1000), rc(1000);
    std::vector<int32_t> wa(1000), wb(1000), wc(1000);
    std::thread a([&] { rc_bad = mrk_batcher_search(bt, seg, &badq.q, ra.data(), wa.data(), 1000, &r_bad); });
    std::thread b([&] { rc_decl = mrk_batcher_search(bt, seg, &declq.q, rb.data(), wb.data(), 1000, &r_decl); });
    std::thread c([&] { rc_good = mrk_batcher_search(bt, seg, &good.q, rc.data(), wc.data(), 1000, &r_good); });
    a.join(), b.join(), c.join();
    if (rc_bad != MRK_E_INVAL) return fprintf(stderr, "malformed query: rc %d\n", rc_bad), 1;
    if (rc_decl != MRK_OK || r_decl.status != MRK_E_UNSUPPORTED) return fprintf(stderr, "declined query: rc %d status %d\n", rc_decl, r_decl.status), 1;
    if (rc_good != MRK_OK || r_good.status != MRK_OK || r_good.total_found != want[2].total || memcmp(r_good.rowid, want[2].rowid.data(), r_good.n * 4))
      return fprintf(stderr, "good query next to a malformed one: rc %d status %d\n", rc_good, r_good.status), 1;
  }
  mrk_batcher_stats st{};
  mrk_batcher_stats_get(bt, &st);
  mrk_batcher_destroy(bt);
  mrk_segment_destroy(seg);
  mrk_host_index_free(hi);
  // ---- 4c. nothing alive: the context goes
  if (mrk_ctx_destroy(ctx) != MRK_OK) return fprintf(stderr, "ctx_destroy: %s\n", mrk_last_error()), 1;
  printf("batcher ok: %d threads x %d queries, batcher %.0f q/s, batches of 256 %.0f q/s, ratio %.3f, %llu launches, largest %u; per launch: submit %.3f ms, in flight %.3f ms, collect %.3f ms\n", T, per, batcher_qps,
         batch_qps, batcher_qps / batch_qps, (unsigned long long)st.launches, st.max_batch, st.submit_ms / st.launches, st.flight_ms / st.launches, st.collect_ms / st.launches);
  return 0;
}
