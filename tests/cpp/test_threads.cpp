// Two host threads, one context: each thread drives its own mrk_batch against the same segment, over and over, with
// different query shapes (a bitmap-kernel AND, a tree, a proximity-ranked AND that goes through the match queue and the
// rank kernel).  Every round's results must equal the ones the main thread computed alone beforehand -- the C-ABI's claim
// "different batches may be driven from different threads" (include/mrk.h), which holds because every HIP call runs on the
// context's submission thread.  Errors are per thread (mrk_last_error).  Usage: test_threads <n_docs> <rounds>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mrk.h"

struct Q {
  mrk_node nodes[6];
  int32_t children[6];
  mrk_query q;
};

static void kw(mrk_node& n, int term, int pos) {
  memset(&n, 0, sizeof n);
  n.op = MRK_OP_TERM, n.term_id = term, n.atom_pos = pos, n.field_mask = MRK_ALL_FIELDS, n.boost = 1.f;
}
static void opn(mrk_node& n, int op, int nkids, int first) {
  memset(&n, 0, sizeof n);
  n.op = op, n.n_children = nkids, n.first_child = first, n.term_id = -1, n.field_mask = MRK_ALL_FIELDS, n.boost = 1.f;
}

// shape 0: a b (BM25)   1: (a | b) c (BM25)   2: a b c (PROXIMITY_BM25)   3: a b (NONE)
static void make(Q& x, int shape, int a, int b, int c) {
  memset(&x, 0, sizeof x);
  if (shape == 1) {
    opn(x.nodes[0], MRK_OP_AND, 2, 0);
    opn(x.nodes[1], MRK_OP_OR, 2, 2);
    kw(x.nodes[2], a, 1), kw(x.nodes[3], b, 2), kw(x.nodes[4], c, 3);
    x.children[0] = 1, x.children[1] = 4, x.children[2] = 2, x.children[3] = 3;
    x.q.n_nodes = 5;
  } else if (shape == 2) {
    opn(x.nodes[0], MRK_OP_AND, 3, 0);
    kw(x.nodes[1], a, 1), kw(x.nodes[2], b, 2), kw(x.nodes[3], c, 3);
    x.children[0] = 1, x.children[1] = 2, x.children[2] = 3;
    x.q.n_nodes = 4;
  } else {
    opn(x.nodes[0], MRK_OP_AND, 2, 0);
    kw(x.nodes[1], a, 1), kw(x.nodes[2], b, 2);
    x.children[0] = 1, x.children[1] = 2;
    x.q.n_nodes = 3;
  }
  x.q.nodes = x.nodes, x.q.children = x.children, x.q.root = 0;
  x.q.ranker = shape == 2 ? MRK_RANK_PROXIMITY_BM25 : shape == 3 ? MRK_RANK_NONE : MRK_RANK_BM25;
  x.q.max_matches = 100 + 300 * shape, x.q.normalized_tfidf = 1;
}

struct Res {
  int64_t total;
  std::vector<uint32_t> rowid;
  std::vector<int32_t> weight;
};

static bool run(mrk_batch* b, mrk_segment* seg, const std::vector<Q>& qs, std::vector<Res>& out) {
  std::vector<mrk_query> flat;
  for (const Q& x : qs) flat.push_back(x.q);
  if (mrk_batch_submit(b, seg, flat.data(), (uint32_t)flat.size()) != MRK_OK || mrk_batch_wait(b) != MRK_OK) return false;
  out.resize(qs.size());
  for (size_t i = 0; i < qs.size(); ++i) {
    mrk_result r;
    if (mrk_batch_result(b, (uint32_t)i, &r) != MRK_OK || r.status != MRK_OK) return false;
    out[i].total = r.total_found;
    out[i].rowid.assign(r.rowid, r.rowid + r.n);
    out[i].weight.assign(r.weight, r.weight + r.n);
  }
  return true;
}

int main(int argc, char** argv) {
  const uint64_t n_docs = argc > 1 ? strtoull(argv[1], 0, 10) : 400000;
  const int rounds = argc > 2 ? atoi(argv[2]) : 60;
  const double probs[5] = {0.3, 0.12, 0.05, 0.01, 0.2};
  mrk_ctx* ctx = nullptr;
  if (mrk_ctx_create(0, &ctx) != MRK_OK) return fprintf(stderr, "ctx: %s\n", mrk_last_error()), 2;
  mrk_synth_params p{};
  p.seed = 11, p.n_docs = n_docs, p.term_prob = probs, p.n_terms = 5, p.n_fields = 2, p.title_frac = 0.1, p.max_pos = 64;
  p.skiplist_block_size = 128, p.hit_format = MRK_HITFMT_INLINE, p.n_threads = 4;
  mrk_host_index* hi = nullptr;
  if (mrk_synth_generate(&p, &hi) != MRK_OK) return fprintf(stderr, "synth: %s\n", mrk_last_error()), 2;
  mrk_segment_desc d{};
  d.spd = mrk_host_index_spd(hi, &d.spd_len), d.spp = mrk_host_index_spp(hi, &d.spp_len), d.spe = mrk_host_index_spe(hi, &d.spe_len);
  d.dict = mrk_host_index_dict(hi, &d.n_terms);
  d.total_docs = n_docs, d.skiplist_block_size = 128, d.hit_format = MRK_HITFMT_INLINE, d.n_fields = 2;
  mrk_segment* seg = nullptr;
  if (mrk_segment_create(ctx, &d, &seg) != MRK_OK) return fprintf(stderr, "segment: %s\n", mrk_last_error()), 2;

  // the two threads' query sets, and what they must return (computed here, alone)
  std::vector<Q> sets[2];
  for (int t = 0; t < 2; ++t) {
    sets[t].resize(8);
    for (int i = 0; i < 8; ++i) make(sets[t][i], (i + t) % 4, (i + t) % 5, (i + 2 * t + 1) % 5 == (i + t) % 5 ? (i + t + 2) % 5 : (i + 2 * t + 1) % 5, (i + 3) % 5);
    for (Q& x : sets[t]) x.q.nodes = x.nodes, x.q.children = x.children; // (vector moves: re-point)
  }
  std::vector<Res> want[2];
  mrk_batch* b0 = nullptr;
  if (mrk_batch_create(ctx, 8, &b0) != MRK_OK) return fprintf(stderr, "batch: %s\n", mrk_last_error()), 2;
  for (int t = 0; t < 2; ++t)
    if (!run(b0, seg, sets[t], want[t])) return fprintf(stderr, "reference run: %s\n", mrk_last_error()), 2;
  mrk_batch_destroy(b0);

  std::atomic<int> bad{0};
  std::string errs[2];
  auto worker = [&](int t) {
    mrk_batch* b = nullptr;
    if (mrk_batch_create(ctx, 8, &b) != MRK_OK) {
      errs[t] = mrk_last_error();
      ++bad;
      return;
    }
    std::vector<Res> got;
    for (int r = 0; r < rounds && !bad; ++r) {
      if (!run(b, seg, sets[t], got)) {
        errs[t] = mrk_last_error();
        ++bad;
        break;
      }
      for (size_t i = 0; i < got.size(); ++i)
        if (got[i].total != want[t][i].total || got[i].rowid != want[t][i].rowid || got[i].weight != want[t][i].weight) {
          errs[t] = "thread " + std::to_string(t) + " round " + std::to_string(r) + " query " + std::to_string(i) + ": results differ";
          ++bad;
        }
    }
    // an error on this thread must not show up on the other one: submit more queries than the batch holds
    std::vector<mrk_query> many(9, sets[t][0].q);
    if (mrk_batch_submit(b, seg, many.data(), 9) != MRK_E_INVAL || !strstr(mrk_last_error(), "capacity")) {
      errs[t] = "expected a capacity error on this thread";
      ++bad;
    }
    mrk_batch_destroy(b);
  };
  std::thread t0(worker, 0), t1(worker, 1);
  t0.join(), t1.join();
  mrk_segment_destroy(seg);
  mrk_host_index_free(hi);
  mrk_ctx_destroy(ctx);
  if (bad) return fprintf(stderr, "FAILED: %s | %s\n", errs[0].c_str(), errs[1].c_str()), 1;
  long long m = 0;
  for (int t = 0; t < 2; ++t)
    for (const Res& r : want[t]) m += r.total;
  printf("two threads ok: %d rounds x 2 x 8 queries, %lld matches per round\n", rounds, m);
  return 0;
}
