// mrk_ranker.h -- host-side C++ mirror of the reference's plugin surface for this path, over the
// C-ABI of include/mrk.h.  Header-only; no HIP types.
//
//   mrk::GpuRanker  has the shape of ISphRanker   (sphinxsearch.h:132-141, ISphExtra sphinxint.h:129-142):
//                   GetMatchesBuffer / GetMatches / Reset / IsCache / FinalizeCache / ExtraData.
//                   Like QcacheRanker_c (sphinxqcache.cpp:601-661) it replays precomputed
//                   (rowid, weight) frames; unlike ExtRanker_c it hands back only the K best matches,
//                   already in MatchRelevanceLt_fn order, plus the total match count.
//   mrk::GpuTopK    has the shape of ISphMatchSorter (sphinxsort.h:39-133) for the plain relevance
//                   queue CSphMatchQueue<MatchRelevanceLt_fn> (sphinxsort.cpp:583-812): Push / GetLength /
//                   GetTotalCount / Flatten / MoveTo / GetWorst, so that MatchExtended's loop
//                   (sphinx.cpp:12201-12268) runs unchanged and chunk results merge the same way
//                   (sphinxrt.cpp:5945-5950).
//
// Error convention follows the reference: factories return nullptr and fill sError
// (sphinxsearch.cpp:4377-4378); nothing throws.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mrk.h"

namespace mrk {

// the CSphMatch members this path produces (sphinx.h:1104-1116); attribute rows stay with the host
struct Match {
  uint32_t m_tRowID = MRK_INVALID_ROWID;
  int m_iWeight = 0;
  int m_iTag = 0;
};

class GpuRanker {
 public:
  static const int FRAME = 1024; // frame length is the ranker's choice (MatchExtended only uses the count)

  // sphCreateRanker (sphinxsearch.cpp:4167): nullptr + sError when the query cannot run on the device
  static GpuRanker* Create(mrk_batch* pBatch, mrk_segment* pSegment, const mrk_query& tQuery, std::string& sError) {
    if (!pBatch || !pSegment) {
      sError = "GpuRanker: no batch / segment";
      return nullptr;
    }
    GpuRanker* p = new GpuRanker();
    p->m_pBatch = pBatch;
    p->m_tQuery = tQuery;
    if (!p->Run(pSegment, sError)) {
      delete p;
      return nullptr;
    }
    return p;
  }

  Match* GetMatchesBuffer() { return m_dMatches; }

  // fills the buffer with the next frame; 0 = end of stream (ISphRanker::GetMatches contract)
  int GetMatches() {
    int n = 0;
    while (n < FRAME && m_iNext < m_tResult.n) {
      m_dMatches[n].m_tRowID = m_tResult.rowid[m_iNext];
      m_dMatches[n].m_iWeight = m_tResult.weight[m_iNext];
      ++n;
      ++m_iNext;
    }
    return n;
  }

  // rebinding to the next segment (RT calls ISphRanker::Reset per RAM segment, sphinxrt.cpp:6313-6314)
  bool Reset(mrk_segment* pSegment, std::string& sError) { return Run(pSegment, sError); }

  bool IsCache() const { return false; }
  void FinalizeCache() {}
  bool ExtraData(int, void**) { return false; }

  // all matches the segment holds for the query, not only the K returned (sphinxsort.cpp:724)
  int64_t GetTotalFound() const { return m_tResult.total_found; }

 private:
  bool Run(mrk_segment* pSegment, std::string& sError) {
    m_iNext = 0;
    memset(&m_tResult, 0, sizeof m_tResult);
    if (mrk_batch_submit(m_pBatch, pSegment, &m_tQuery, 1) != MRK_OK || mrk_batch_wait(m_pBatch) != MRK_OK ||
        mrk_batch_result(m_pBatch, 0, &m_tResult) != MRK_OK) {
      sError = mrk_last_error();
      return false;
    }
    if (m_tResult.status != MRK_OK) {
      sError = mrk_last_error();
      return false;
    }
    return true;
  }

  mrk_batch* m_pBatch = nullptr;
  mrk_query m_tQuery{};
  mrk_result m_tResult{};
  int m_iNext = 0;
  Match m_dMatches[FRAME];
};

// Relevance top-K with the reference's comparator; Push() keeps it usable as a drop-in sorter for
// matches that come from elsewhere (e.g. a CPU-ranked RAM segment), SetTotal() lets a fused device
// result carry its exact total.
class GpuTopK {
 public:
  explicit GpuTopK(int iSize) : m_iSize(iSize) { m_dData.reserve(iSize + 1); }

  static bool IsLess(const Match& a, const Match& b) { // MatchRelevanceLt_fn, sphinxsort.cpp:4541-4547
    if (a.m_iWeight != b.m_iWeight) return a.m_iWeight < b.m_iWeight;
    return a.m_tRowID > b.m_tRowID;
  }

  bool Push(const Match& tEntry) { // CSphMatchQueue::PushT, sphinxsort.cpp:722-761
    ++m_iTotal;
    auto worse_first = [](const Match& x, const Match& y) { return IsLess(y, x); }; // root = worst
    if ((int)m_dData.size() == m_iSize) {
      if (IsLess(tEntry, m_dData.front())) return true;
      std::pop_heap(m_dData.begin(), m_dData.end(), worse_first);
      m_dData.pop_back();
    }
    m_dData.push_back(tEntry);
    std::push_heap(m_dData.begin(), m_dData.end(), worse_first);
    return true;
  }

  int GetLength() const { return (int)m_dData.size(); }
  int64_t GetTotalCount() const { return m_iTotal; }
  void SetTotal(int64_t iTotal) { m_iTotal = iTotal; }
  const Match* GetWorst() const { return m_dData.empty() ? nullptr : &m_dData.front(); }

  int Flatten(Match* pTo) { // best first, sphinxsort.cpp:627-641
    std::vector<Match> d = m_dData;
    std::sort(d.begin(), d.end(), [](const Match& x, const Match& y) { return IsLess(y, x); });
    for (size_t i = 0; i < d.size(); ++i) pTo[i] = d[i];
    const int n = (int)d.size();
    m_dData.clear();
    m_iTotal = 0;
    return n;
  }

  void MoveTo(GpuTopK* pRhs) { // sphinxsort.cpp:681-710: totals add up, matches are re-pushed
    const int64_t iTotal = pRhs->m_iTotal + m_iTotal;
    for (const Match& m : m_dData) pRhs->Push(m);
    pRhs->m_iTotal = iTotal;
    m_dData.clear();
    m_iTotal = 0;
  }

 private:
  int m_iSize;
  int64_t m_iTotal = 0;
  std::vector<Match> m_dData;
};

} // namespace mrk
