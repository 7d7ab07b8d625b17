// mrk_replay.h -- the frame replay of the reference-side binding, free of the reference's headers so that it can be EXECUTED
// in this repository's tests (tests/cpp/test_replay.cpp instantiates it over stub match / sorter types; mrk_adapter.h
// instantiates it over CSphMatch / ISphMatchSorter).
//
// The device hands back the K best (rowid, weight) rows of a query and the number of docs it matched.  An ISphRanker has to
// present them to CSphIndex_VLN::MatchExtended (sphinx.cpp:12190-12269) as frames of matches, the way QcacheRanker_c replays
// cached frames (sphinxqcache.cpp:601-661), and the sorter has to end up reporting total_found matches although only the K
// rows were pushed (CSphMatchQueueTraits::m_iTotal counts pushes, sphinxsort.cpp:724).
//
//   Next()    fills the ranker's match buffer with the next rows; rowid and weight are set, the caller-supplied reject
//             functor stands for CSphIndex::EarlyReject (sets m_pStatic; the device already applied every filter: none
//             rejects).  Weights were multiplied by the index weight on the device when a weight filter needed the final
//             value there (sphinx.cpp:12220-12227); MatchExtended multiplies again, so the replay divides first (exact).
//   Finish()  adds total_found - pushed to the sorter's total, once.  Called at end of stream AND from FinalizeCache
//             (sphinx.cpp:15919, sphinxrt.cpp:6441: after the match loop whatever ended it) -- MatchExtended leaves its loop
//             without another GetMatches() when the cutoff runs out on the last row of a frame (sphinx.cpp:12261-12267).
#pragma once

#include <stdint.h>

#include "mrk.h"

template <typename MATCH>
class MrkFrameReplay_T
{
public:
	void Start ( const mrk_result & tResult, int iIndexWeight )
	{
		m_tResult = tResult;
		m_iNext = 0;
		m_iHanded = 0;
		m_bDone = false;
		m_iIndexWeight = iIndexWeight>0 ? iIndexWeight : 1;
	}

	/// next frame of at most iFrame rows into pBuf; 0 = end of stream.  fnReject ( MATCH & ) -> true drops the row.
	template <typename REJECT>
	int Next ( MATCH * pBuf, int iFrame, REJECT && fnReject )
	{
		int iRes = 0;
		while ( !iRes && m_iNext<m_tResult.n )
			while ( iRes<iFrame && m_iNext<m_tResult.n )
			{
				MATCH & tMatch = pBuf[iRes];
				tMatch.m_tRowID = m_tResult.rowid[m_iNext];
				tMatch.m_iWeight = m_tResult.weight[m_iNext] / m_iIndexWeight;
				++m_iNext;
				if ( !fnReject ( tMatch ) )
					++iRes;
			}
		m_iHanded += iRes;
		return iRes;
	}

	bool AtEnd () const { return m_iNext>=m_tResult.n; }

	/// the sorter counted the rows it was pushed; the query matched total_found docs
	template <typename SORTER>
	void Finish ( SORTER * pSorter )
	{
		if ( m_bDone )
			return;
		m_bDone = true;
		if ( pSorter && m_tResult.total_found>m_iHanded )
			pSorter->m_iTotal += m_tResult.total_found - m_iHanded;
	}

	/// a ranker that is rebound to another segment has nothing more to hand out for this one
	void Exhaust () { m_iNext = m_tResult.n; }

	const mrk_result & Result () const { return m_tResult; }

private:
	mrk_result	m_tResult {};
	int			m_iNext = 0;
	int64_t		m_iHanded = 0;	///< rows handed to the caller so far (each becomes one Push)
	int			m_iIndexWeight = 1;
	bool		m_bDone = false;
};


/// CSphQueryResultMeta::AddStat accumulates (sphinx.cpp:27907-27914) and sphCreateRanker reports every DISTINCT query word
/// once, in the order of its first (smallest) query position (hQwords is keyed by the query word, ExtTerm_T::GetQwords keeps
/// the minimum position, ExtQwordOrderbyQueryPos_t sorts; sphinxsearch.cpp:4365-4371, searchnode.cpp:2030-2055).  Not-weighted
/// occurrences (XQNode_t::m_bNotWeighted) never enter hQwords (ExtTerm_T::GetQwords returns before the insert): a word that
/// only occurs that way is not reported at all.  Expanded keywords are skipped by the caller.
struct MrkWordStat_t
{
	int		m_iNode;		///< index of the keyword's first node
	int		m_iQueryPos;
};

/// dNodes: (word id for equality, atom position, not-weighted) per keyword node in tree order; fills pOut with one entry per
/// distinct word, sorted by query position; returns their number.  nNodes <= 64 on the device path; O(n^2) is fine.
template <typename SAME>
inline int MrkDistinctWords ( int nNodes, const int * pAtomPos, const bool * pNotWeighted, SAME && fnSame, MrkWordStat_t * pOut )
{
	int nOut = 0;
	for ( int i=0; i<nNodes; ++i )
	{
		if ( pNotWeighted[i] )
			continue;
		int iFound = -1;
		for ( int j=0; j<nOut && iFound<0; ++j )
			if ( fnSame ( pOut[j].m_iNode, i ) )
				iFound = j;
		if ( iFound<0 )
		{
			pOut[nOut].m_iNode = i;
			pOut[nOut].m_iQueryPos = pAtomPos[i];
			++nOut;
		} else if ( pAtomPos[i]<pOut[iFound].m_iQueryPos )
			pOut[iFound].m_iQueryPos = pAtomPos[i];
	}
	for ( int i=1; i<nOut; ++i )	// insertion sort by query position (distinct words never share one)
	{
		const MrkWordStat_t t = pOut[i];
		int j = i;
		for ( ; j>0 && pOut[j-1].m_iQueryPos>t.m_iQueryPos; --j )
			pOut[j] = pOut[j-1];
		pOut[j] = t;
	}
	return nOut;
}
