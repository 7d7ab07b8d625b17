// integration/mrk_replay.h EXECUTED over stub match / sorter types (the adapter instantiates the same templates over CSphMatch /
// ISphMatchSorter; it can only be syntax-checked here).  What is pinned:
//   * MatchExtended's loop (sphinx.cpp:12190-12269) restated over the stubs: GetMatches() frames, "weight *= index weight",
//     Push, the cutoff countdown that leaves the loop WITHOUT another GetMatches() when it runs out on a frame's last row;
//   * the sorter's total ends up at total_found in every one of those cases (end of stream, cutoff == n, cutoff < n is not
//     something the device hands back, index weight 3 with the division before the multiplication);
//   * keyword statistics: one entry per distinct query word in first-position order, not-weighted occurrences skipped.
// No GPU, no libmrk.so.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../integration/mrk_replay.h"

struct StubMatch {
  uint32_t m_tRowID = 0;
  int m_iWeight = 0;
};
struct StubSorter {
  int64_t m_iTotal = 0;
  std::vector<StubMatch> rows;
  bool Push(const StubMatch& m) { // CSphMatchQueue::PushT: counts every push, returns true whether or not the heap kept it
    ++m_iTotal;
    rows.push_back(m);
    return true;
  }
};
struct StubRanker {
  static const int FRAME = 4;
  MrkFrameReplay_T<StubMatch> replay;
  StubSorter* sorter;
  StubMatch buf[FRAME];
  int GetMatches() {
    const int n = replay.Next(buf, FRAME, [](StubMatch&) { return false; });
    if (!n) replay.Finish(sorter);
    return n;
  }
  void FinalizeCache() {
    if (replay.AtEnd()) replay.Finish(sorter);
  }
};

// CSphIndex_VLN::MatchExtended, sphinx.cpp:12196-12268, over the stubs
static void match_extended(StubRanker& rk, StubSorter& so, int cutoff, int index_weight) {
  int iCutoff = cutoff <= 0 ? -1 : cutoff;
  for (;;) {
    const int n = rk.GetMatches();
    if (n <= 0) break;
    for (int i = 0; i < n; ++i) {
      StubMatch& m = rk.buf[i];
      m.m_iWeight *= index_weight;
      const bool bNew = so.Push(m);
      if (bNew)
        if (--iCutoff == 0) break;
    }
    if (iCutoff == 0) break;
  }
  rk.FinalizeCache(); // ParsedMultiQuery, sphinx.cpp:15919
}

static int check(const char* what, int n, int64_t total_found, int cutoff, int index_weight) {
  std::vector<uint32_t> rowid(n);
  std::vector<int32_t> weight(n);
  for (int i = 0; i < n; ++i) rowid[i] = 10 + i, weight[i] = (1000 - i) * index_weight; // (the device multiplied already)
  mrk_result r{};
  r.n = n, r.total_found = total_found, r.rowid = rowid.data(), r.weight = weight.data(), r.status = MRK_OK;
  StubSorter so;
  StubRanker rk;
  rk.sorter = &so;
  rk.replay.Start(r, index_weight);
  match_extended(rk, so, cutoff, index_weight);
  if (so.m_iTotal != total_found) return fprintf(stderr, "%s: sorter total %lld, want %lld\n", what, (long long)so.m_iTotal, (long long)total_found), 1;
  if ((int)so.rows.size() != n) return fprintf(stderr, "%s: %zu rows pushed, want %d\n", what, so.rows.size(), n), 1;
  for (int i = 0; i < n; ++i)
    if (so.rows[i].m_tRowID != rowid[i] || so.rows[i].m_iWeight != (1000 - i) * index_weight)
      return fprintf(stderr, "%s: row %d = (%u, %d)\n", what, i, so.rows[i].m_tRowID, so.rows[i].m_iWeight), 1;
  return 0;
}

int main() {
  int bad = 0;
  bad += check("end of stream, K rows of many", 10, 12345, 0, 1);
  bad += check("fewer matches than K", 3, 3, 0, 1);
  bad += check("no match", 0, 0, 0, 1);
  bad += check("cutoff beyond the rows (K < cutoff)", 8, 500, 500, 1);      // the device: best 8 of the first 500
  bad += check("cutoff == n on a frame boundary", 8, 8, 8, 1);              // the loop breaks on row 8 = last row of frame 2: no GetMatches() after it
  bad += check("cutoff == n inside a frame", 6, 6, 6, 1);
  bad += check("index weight 3 (a weight filter made the device multiply)", 10, 77, 0, 3);
  // keyword statistics: 'b a b -c a' with c not weighted -> b (pos 1), a (pos 2); c not reported
  {
    const char* words[5] = {"b", "a", "b", "c", "a"};
    const int pos[5] = {1, 2, 3, 4, 5};
    const bool nw[5] = {false, false, false, true, false};
    MrkWordStat_t out[5];
    const int n = MrkDistinctWords(5, pos, nw, [&](int x, int y) { return !strcmp(words[x], words[y]); }, out);
    if (n != 2 || out[0].m_iNode != 0 || out[1].m_iNode != 1 || out[0].m_iQueryPos != 1 || out[1].m_iQueryPos != 2)
      bad += fprintf(stderr, "distinct words: n %d\n", n), 1;
    // a word whose first occurrence comes later in the tree than another's but earlier in the query: order by position
    const int pos2[3] = {5, 2, 1};
    const char* w2[3] = {"x", "y", "x"};
    const bool nw2[3] = {false, false, false};
    const int n2 = MrkDistinctWords(3, pos2, nw2, [&](int x, int y) { return !strcmp(w2[x], w2[y]); }, out);
    if (n2 != 2 || out[0].m_iNode != 0 || out[0].m_iQueryPos != 1 || out[1].m_iNode != 1) bad += fprintf(stderr, "distinct words 2: n %d\n", n2), 1;
  }
  if (bad) return 1;
  printf("replay ok\n");
  return 0;
}
