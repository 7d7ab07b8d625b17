// mrk_adapter.h -- the reference-side binding of libmrk.so: what a Manticore 3.6.1 maintainer adds next to
// src/sphinxsearch.cpp so that sphCreateRanker can hand the match -> rank -> top-K path to the device.
//
// This file is written against the REFERENCE's own headers (sphinx.h, sphinxint.h, sphinxsearch.h, sphinxquery.h,
// sphinxsort.h); it compiles inside the reference tree and nowhere else.  integration/check_adapter.sh syntax-checks it
// (-fsyntax-only) against /root/reference/src in the build container.  Nothing of the reference is copied: the adapter
// only USES its public interfaces.
//
//   MrkEligible          may this (query, sorters) pair take the device ranker at all?
//   FlattenXQ            XQQuery_t / CSphQuery / CSphQueryContext  ->  mrk_query   (include/mrk.h)
//   MrkRankerAdapter_c   ISphRanker (sphinxsearch.h:132-141) over one mrk_batch: replays the device's K best
//                        (rowid, weight) frames the way QcacheRanker_c::GetMatches replays cached ones
//                        (sphinxqcache.cpp:601-661): EarlyReject per row (sets m_pStatic, sphinx.cpp:11903-11917), and
//                        at end of stream restores the sorter's total (ISphMatchSorter::m_iTotal is a public member:
//                        nothing inside the reference's sorter or MatchExtended changes).
//   MrkCreateRanker      the three calls above in the order sphCreateRanker needs them; nullptr = not for the device,
//                        keep the ExtRanker_* path.
//
// Why the sorter matters.  The device returns ONLY the K best matches by (weight desc, rowid asc) -- MatchRelevanceLt_fn,
// sphinxsort.cpp:4541-4547 -- not the whole match stream.  Feeding those K rows to any queue whose order is not exactly
// that (ORDER BY attribute, expressions, group-by, several sorters, random) would silently drop rows the queue wanted, and
// so would a filter the device did not evaluate (EarlyReject would thin the K rows out).  MrkEligible therefore admits:
// one sorter, not group-by, not random, relevance order; no packed factors; max_matches and cutoff <= MRK_MAX_K; filters
// (attribute filters: integer VALUES / RANGE, FLOATRANGE, MVA any / all; weight filters) only if FlattenXQ could hand every
// one of them to the device; a weight filter next to a cutoff stays on the CPU (the cutoff counts matches in rowid order
// AFTER the weight filter, sphinx.cpp:12223-12267; the device's cutoff probe runs without weights).
#pragma once

#include "sphinx.h"
#include "sphinxint.h"
#include "sphinxsearch.h"
#include "sphinxquery.h"
#include "sphinxsort.h"

#include "mrk.h" // include/mrk.h of the mrk repository
#include "mrk_replay.h" // the frame replay + keyword-statistics order, free of reference types (executed by tests/cpp/test_replay.cpp)

/// what the host keeps per index / disk chunk next to its CSphIndex
struct MrkIndexBinding_t
{
	mrk_segment *			m_pSegment = nullptr;	///< mrk_segment_create over the chunk's .spd/.spp/.spe (+ set_attrs / set_dead_rows)
	const mrk_host_index *	m_pFiles = nullptr;		///< mrk_index_open: the flat dictionary (term id = row)
	bool					m_bWordDict = true;		///< dict=keywords (lookup by m_sDictWord) or dict=crc (lookup by word id)
};

/// RT RAM segments (round 3): PerformFullTextSearch rebinds the ONE ranker to every RAM segment (ISphRanker::Reset, sphinxrt.cpp:6313-6314).
/// The host keeps a device segment per RtSegment_t (mrk_rt_segment_open over the segment's m_dWords / m_dDocs / m_dHits once, when the
/// segment is committed -- RAM segments never change afterwards --, then mrk_segment_create; dead rows and attribute rows handed over
/// with mrk_segment_set_dead_rows / _set_attrs) and tells the adapter how to find it from the setup object Reset() receives.
/// nullptr = that segment has no device copy (the caller must then not have offered the query to the device at all).
typedef const MrkIndexBinding_t * (*MrkSegmentLookup_fn) ( const ISphQwordSetup & tSetup, void * pUser );

/// a flattened query: the arrays mrk_query points into
struct MrkFlatQuery_t
{
	CSphVector<mrk_node>	m_dNodes;
	CSphVector<int32_t>		m_dChildren;
	CSphVector<int32_t>		m_dWeights;
	CSphVector<int64_t>		m_dLocalDocs;
	CSphVector<int64_t>		m_dFilterValues [ MRK_MAX_FILTERS ];
	mrk_filter				m_dFilters [ MRK_MAX_FILTERS ];
	CSphVector<int64_t>		m_dWeightFilterValues [ MRK_MAX_FILTERS ];
	mrk_filter				m_dWeightFilters [ MRK_MAX_FILTERS ];	///< filters on @weight / a weight column (CSphQueryContext::m_pWeightFilter)
	mrk_query				m_tQuery;
	CSphVector<CSphString>	m_dDictWords;			///< per node (empty for operators), for tMeta.AddStat
	CSphVector<CSphString>	m_dWords;				///< per node: the query word hQwords is keyed by (XQKeyword_t::m_sWord)
	int						m_iIndexWeight = 1;		///< what the device multiplied the weights by (only when a weight filter needs the final value there)
};


/// ISphMatchSorter order == MatchRelevanceLt_fn?  SPH_SORT_RELEVANCE builds CSphMatchQueue<MatchRelevanceLt_fn> (sphinxsort.cpp:6597,
/// 5813); "ORDER BY weight() DESC" builds MatchGeneric1_fn over one WEIGHT keypart, which compares the same two things in
/// the same order (weight, then rowid ascending)
inline bool MrkSorterIsRelevance ( const CSphQuery & tQuery, const ISphMatchSorter * pSorter )
{
	if ( !pSorter || pSorter->IsGroupby() || pSorter->m_bRandomize )
		return false;
	if ( tQuery.m_eSort==SPH_SORT_RELEVANCE )
		return true;
	const CSphMatchComparatorState & tState = pSorter->GetState();
	return tQuery.m_eSort==SPH_SORT_EXTENDED
		&& tState.m_eKeypart[0]==SPH_KEYPART_WEIGHT && ( tState.m_uAttrDesc & 1 )!=0
		&& tState.m_eKeypart[1]==SPH_KEYPART_ROWID && ( tState.m_uAttrDesc & 2 )==0;
}


inline bool MrkEligible ( const CSphQuery & tQuery, const CSphQueryContext & tCtx, const VecTraits_T<ISphMatchSorter *> & dSorters,
	DWORD uPackedFactorFlags, CSphString & sWhy )
{
	if ( dSorters.GetLength()!=1 )					{ sWhy = "several sorters"; return false; }
	if ( !MrkSorterIsRelevance ( tQuery, dSorters[0] ) )	{ sWhy = "sorter order is not (weight desc, rowid asc)"; return false; }
	if ( tCtx.m_pWeightFilter && tQuery.m_iCutoff>0 )	{ sWhy = "weight filter next to a cutoff"; return false; }	// sphinx.cpp:12223-12267
	if ( tQuery.m_iCutoff>MRK_MAX_K )				{ sWhy = "cutoff beyond the device top-K"; return false; }	// sphinx.cpp:12261-12267; smaller ones: mrk_query::cutoff
	if ( uPackedFactorFlags!=SPH_FACTOR_DISABLE )	{ sWhy = "packed factors"; return false; }
	if ( tQuery.m_iMaxMatches<1 || tQuery.m_iMaxMatches>MRK_MAX_K )	{ sWhy = "max_matches beyond the device top-K"; return false; }
	if ( tQuery.m_dFilterTree.GetLength() )			{ sWhy = "filter tree"; return false; }
	if ( tQuery.m_eRanker==SPH_RANK_EXPR || tQuery.m_eRanker==SPH_RANK_EXPORT || tQuery.m_eRanker==SPH_RANK_PLUGIN )
													{ sWhy = "expression / plugin ranker"; return false; }
	return true;
}


/// XQOperator_e -> MRK_OP_*; -1 = not on the device path
inline int MrkOpOf ( XQOperator_e eOp )
{
	switch ( eOp )
	{
		case SPH_QUERY_AND:			return MRK_OP_AND;
		case SPH_QUERY_OR:			return MRK_OP_OR;
		case SPH_QUERY_MAYBE:		return MRK_OP_MAYBE;
		case SPH_QUERY_ANDNOT:		return MRK_OP_ANDNOT;
		case SPH_QUERY_BEFORE:		return MRK_OP_BEFORE;
		case SPH_QUERY_PHRASE:		return MRK_OP_PHRASE;
		case SPH_QUERY_PROXIMITY:	return MRK_OP_PROXIMITY;
		case SPH_QUERY_QUORUM:		return MRK_OP_QUORUM;
		case SPH_QUERY_NEAR:		return MRK_OP_NEAR;
		case SPH_QUERY_NOTNEAR:		return MRK_OP_NOTNEAR;
		case SPH_QUERY_SENTENCE:	return MRK_OP_SENTENCE;
		case SPH_QUERY_PARAGRAPH:	return MRK_OP_PARAGRAPH;
		default:					return -1;	// NOT, NULL, SCAN
	}
}


/// the dictionary row of a keyword, normalized exactly as CreateQueryWord does it (searchnode.cpp:858-873)
inline int MrkLookupTerm ( const MrkIndexBinding_t & tIndex, const ISphQwordSetup & tSetup, const XQKeyword_t & tWord, CSphString & sDictWord )
{
	BYTE sTmp [ 3*SPH_MAX_WORD_LEN + 16 ];
	strncpy ( (char*)sTmp, tWord.m_sWord.cstr(), sizeof(sTmp) );
	sTmp[sizeof(sTmp)-1] = '\0';
	CSphDict * pDict = tSetup.Dict();
	const SphWordID_t uWordID = tWord.m_bMorphed ? pDict->GetWordIDNonStemmed ( sTmp ) : pDict->GetWordID ( sTmp );
	sDictWord = (const char*)sTmp;
	if ( !uWordID )
		return -1;	// a stop word / unknown to the dictionary: no postings
	return tIndex.m_bWordDict
		? mrk_host_index_find_word ( tIndex.m_pFiles, (const char*)sTmp, (int32_t) strlen ( (const char*)sTmp ) )
		: mrk_host_index_find_wordid ( tIndex.m_pFiles, (uint64_t)uWordID );
}


class MrkFlattener_c
{
public:
	MrkFlattener_c ( const MrkIndexBinding_t & tIndex, const ISphQwordSetup & tSetup, const CSphQueryContext & tCtx, MrkFlatQuery_t & tOut, CSphString & sWhy )
		: m_tIndex ( tIndex ), m_tSetup ( tSetup ), m_tCtx ( tCtx ), m_tOut ( tOut ), m_sWhy ( sWhy )
	{}

	/// -> index of the node in m_dNodes, -1 = decline
	int Node ( const XQNode_t * pNode )
	{
		if ( !pNode )												return Fail ( "empty node" );
		if ( pNode->m_dSpec.m_dZones.GetLength() )					return Fail ( "zone limits" );
		if ( m_tSetup.m_bHasWideFields )							return Fail ( "more than 32 fields" );
		const DWORD uMask = pNode->m_dSpec.m_dFieldMask.GetMask32();
		const int iMaxPos = pNode->m_dSpec.m_iFieldMaxPos;
		const int iWords = pNode->m_dWords.GetLength();

		if ( iWords==1 && ( pNode->GetOp()==SPH_QUERY_AND || pNode->GetOp()==SPH_QUERY_OR ) )
			return Keyword ( pNode->m_dWords[0], uMask, iMaxPos, pNode->m_bNotWeighted );

		const int iOp = MrkOpOf ( pNode->GetOp() );
		if ( iOp<0 )												return Fail ( "operator not on the device path" );
		if ( pNode->m_bPercentOp )									return Fail ( "percent quorum" );	// (resolve first: ExtQuorum_c::GetThreshold)

		CSphVector<int> dKids;
		if ( iWords )	// a plain multi-keyword node: PHRASE / PROXIMITY / QUORUM, or AND / OR over bare words
			for ( const XQKeyword_t & tWord : pNode->m_dWords )
			{
				const int iKid = Keyword ( tWord, uMask, iMaxPos, false );
				if ( iKid<0 ) return -1;
				dKids.Add ( iKid );
			}
		else
			for ( const XQNode_t * pKid : pNode->m_dChildren )
			{
				const int iKid = Node ( pKid );
				if ( iKid<0 ) return -1;
				dKids.Add ( iKid );
			}
		if ( !dKids.GetLength() )									return Fail ( "operator without operands" );

		mrk_node & tNode = m_tOut.m_dNodes.Add();
		memset ( &tNode, 0, sizeof(tNode) );
		tNode.op = iOp;
		tNode.n_children = dKids.GetLength();
		tNode.first_child = m_tOut.m_dChildren.GetLength();
		tNode.term_id = -1;
		if ( iOp==MRK_OP_SENTENCE || iOp==MRK_OP_PARAGRAPH )
		{
			// the boundary keyword ExtUnit_c's ctor looks up (searchnode.cpp:4987-4989): MAGIC_WORD_SENTENCE / MAGIC_WORD_PARAGRAPH
			XQKeyword_t tDot;
			tDot.m_sWord = iOp==MRK_OP_SENTENCE ? MAGIC_WORD_SENTENCE : MAGIC_WORD_PARAGRAPH;
			CSphString sDot;
			tNode.term_id = MrkLookupTerm ( m_tIndex, m_tSetup, tDot, sDot );
		}
		tNode.field_mask = uMask;
		tNode.boost = 1.0f;
		tNode.opt = pNode->m_iOpArg;
		for ( int iKid : dKids )
			m_tOut.m_dChildren.Add ( iKid );
		m_tOut.m_dLocalDocs.Add ( -1 );
		m_tOut.m_dDictWords.Add ( CSphString() );
		m_tOut.m_dWords.Add ( CSphString() );
		return m_tOut.m_dNodes.GetLength()-1;
	}

private:
	int Fail ( const char * sWhy )
	{
		m_sWhy = sWhy;
		return -1;
	}

	int Keyword ( const XQKeyword_t & tWord, DWORD uMask, int iMaxPos, bool bNotWeighted )
	{
		if ( tWord.m_bExpanded )									return Fail ( "expanded keyword" );	// (prefix / infix expansion stays on the CPU)
		if ( tWord.m_pPayload )										return Fail ( "keyword payload" );
		CSphString sDictWord;
		mrk_node & tNode = m_tOut.m_dNodes.Add();
		memset ( &tNode, 0, sizeof(tNode) );
		tNode.op = MRK_OP_TERM;
		tNode.term_id = MrkLookupTerm ( m_tIndex, m_tSetup, tWord, sDictWord );
		tNode.atom_pos = tWord.m_iAtomPos;
		tNode.field_mask = uMask;
		tNode.boost = tWord.m_fBoost;
		tNode.not_weighted = bNotWeighted ? 1 : 0;
		// TermPosFilter_e as ExtNode_i::Create derives it (searchnode.cpp:875-878, 1145-1146): a field position limit wins
		tNode.term_pos = iMaxPos ? MRK_TERMPOS_LIMIT
			: ( tWord.m_bFieldStart && tWord.m_bFieldEnd ) ? MRK_TERMPOS_STARTEND
			: tWord.m_bFieldStart ? MRK_TERMPOS_START
			: tWord.m_bFieldEnd ? MRK_TERMPOS_END : MRK_TERMPOS_NONE;
		tNode.field_max_pos = iMaxPos;
		// local_df: per-keyword document counts summed over the local indexes (sphinxsearch.cpp:4308-4315)
		int64_t iLocal = -1;
		if ( m_tCtx.m_pLocalDocs )
		{
			const int64_t * pDocs = (*m_tCtx.m_pLocalDocs)( sDictWord );
			if ( pDocs )
				iLocal = *pDocs;
		}
		m_tOut.m_dLocalDocs.Add ( iLocal );
		m_tOut.m_dDictWords.Add ( sDictWord );
		m_tOut.m_dWords.Add ( tWord.m_sWord );
		return m_tOut.m_dNodes.GetLength()-1;
	}

	const MrkIndexBinding_t &	m_tIndex;
	const ISphQwordSetup &		m_tSetup;
	const CSphQueryContext &	m_tCtx;
	MrkFlatQuery_t &			m_tOut;
	CSphString &				m_sWhy;
};


/// IsWeightColumn (sphinxfilter.cpp:1895-1902): such a filter becomes part of CSphQueryContext::m_pWeightFilter
inline bool MrkIsWeightFilter ( const CSphFilterSettings & tFilter, const ISphSchema & tSchema )
{
	if ( tFilter.m_sAttrName=="@weight" )
		return true;
	const CSphColumnInfo * pCol = tSchema.GetAttr ( tFilter.m_sAttrName.cstr() );
	return pCol && pCol->m_bWeight;
}


/// the VALUES of a filter, ascending (IFilter_Values::SetValues expects them so), into the flat query's storage
inline bool MrkFilterValues ( const CSphFilterSettings & tFilter, mrk_filter & tOut, CSphVector<int64_t> & dValues )
{
	if ( tFilter.GetNumValues()<1 || tFilter.GetNumValues()>MRK_MAX_FILTER_VALUES )
		return false;
	dValues.Resize ( 0 );
	for ( int i=0; i<tFilter.GetNumValues(); ++i )
		dValues.Add ( tFilter.GetValue(i) );
	dValues.Sort();
	tOut.values = dValues.Begin();
	tOut.n_values = dValues.GetLength();
	return true;
}


/// a filter on the match weight (Filter_WeightValues / Filter_WeightRange, sphinxfilter.cpp:304-320, created by
/// CreateSpecialFilter :944-951: VALUES or RANGE, exclusion wraps it in FilterNot) -> mrk_query::weight_filters
inline bool MrkFlattenWeightFilter ( const CSphFilterSettings & tFilter, mrk_filter & tOut, CSphVector<int64_t> & dValues )
{
	if ( tFilter.m_eType!=SPH_FILTER_VALUES && tFilter.m_eType!=SPH_FILTER_RANGE )
		return false;
	memset ( &tOut, 0, sizeof(tOut) );
	tOut.kind = tFilter.m_eType==SPH_FILTER_VALUES ? MRK_FILTER_VALUES : MRK_FILTER_RANGE;
	tOut.exclude = tFilter.m_bExclude;
	tOut.has_equal_min = tFilter.m_bHasEqualMin;
	tOut.has_equal_max = tFilter.m_bHasEqualMax;
	tOut.min_value = tFilter.m_iMinValue;
	tOut.max_value = tFilter.m_iMaxValue;
	return tFilter.m_eType!=SPH_FILTER_VALUES || MrkFilterValues ( tFilter, tOut, dValues );
}


/// CSphFilterSettings over a row-stored attribute -> mrk_filter, following CreateFilter's dispatch on the attribute type
/// (sphinxfilter.cpp:954-1060): integer VALUES / RANGE; float columns (FLOATRANGE, and RANGE / one-value VALUES after
/// FixupFilterSettings :1565-1583 turned them into a float range); MVA columns VALUES / RANGE with ANY() / ALL()
/// (Filter_MVAValues_Any_c ... Filter_MVARange_All_c, :340-383).  false = this filter cannot travel: the query stays on the CPU
inline bool MrkFlattenFilter ( const CSphFilterSettings & tFilter, const ISphSchema & tSchema, mrk_filter & tOut, CSphVector<int64_t> & dValues )
{
	if ( tFilter.m_bIsNull )
		return false;
	const int iAttr = tSchema.GetAttrIndex ( tFilter.m_sAttrName.cstr() );
	if ( iAttr<0 )
		return false;
	const CSphColumnInfo & tCol = tSchema.GetAttr ( iAttr );
	if ( tCol.m_pExpr || tCol.m_tLocator.m_bDynamic )
		return false;	// computed / dynamic: not in the .spa row the device holds
	memset ( &tOut, 0, sizeof(tOut) );
	tOut.exclude = tFilter.m_bExclude;
	tOut.has_equal_min = tFilter.m_bHasEqualMin;
	tOut.has_equal_max = tFilter.m_bHasEqualMax;
	switch ( tCol.m_eAttrType )
	{
		case SPH_ATTR_INTEGER: case SPH_ATTR_TIMESTAMP: case SPH_ATTR_BOOL: case SPH_ATTR_BIGINT:
			if ( tFilter.m_eType!=SPH_FILTER_VALUES && tFilter.m_eType!=SPH_FILTER_RANGE )
				return false;
			if ( tFilter.m_eMvaFunc!=SPH_MVAFUNC_NONE || tCol.m_tLocator.IsBlobAttr() )
				return false;
			tOut.kind = tFilter.m_eType==SPH_FILTER_VALUES ? MRK_FILTER_VALUES : MRK_FILTER_RANGE;
			tOut.bit_offset = tCol.m_tLocator.m_iBitOffset;
			tOut.bit_count = tCol.m_tLocator.m_iBitCount;
			tOut.open_left = tFilter.m_bOpenLeft;
			tOut.open_right = tFilter.m_bOpenRight;
			tOut.min_value = tFilter.m_iMinValue;
			tOut.max_value = tFilter.m_iMaxValue;
			return tFilter.m_eType!=SPH_FILTER_VALUES || MrkFilterValues ( tFilter, tOut, dValues );

		case SPH_ATTR_FLOAT:
			if ( tCol.m_tLocator.IsBlobAttr() )
				return false;
			tOut.kind = MRK_FILTER_FLOATRANGE;
			tOut.bit_offset = tCol.m_tLocator.m_iBitOffset;
			tOut.bit_count = tCol.m_tLocator.m_iBitCount;
			if ( tFilter.m_eType==SPH_FILTER_FLOATRANGE )
			{
				tOut.fmin = tFilter.m_fMinValue;
				tOut.fmax = tFilter.m_fMaxValue;
			} else if ( tFilter.m_eType==SPH_FILTER_RANGE )				// "fltcol BETWEEN 1 AND 3"
			{
				tOut.fmin = (float)tFilter.m_iMinValue;
				tOut.fmax = (float)tFilter.m_iMaxValue;
			} else if ( tFilter.m_eType==SPH_FILTER_VALUES && tFilter.GetNumValues()==1 )	// "fltcol=intval"
				tOut.fmin = tOut.fmax = (float)tFilter.GetValue(0);
			else
				return false;
			return true;

		case SPH_ATTR_UINT32SET: case SPH_ATTR_INT64SET:
			if ( tFilter.m_eType!=SPH_FILTER_VALUES && tFilter.m_eType!=SPH_FILTER_RANGE )
				return false;
			if ( !tCol.m_tLocator.IsBlobAttr() )
				return false;
			tOut.kind = tFilter.m_eType==SPH_FILTER_VALUES ? MRK_FILTER_VALUES : MRK_FILTER_RANGE;
			tOut.mva_bits = tCol.m_eAttrType==SPH_ATTR_INT64SET ? 64 : 32;
			tOut.mva_all = tFilter.m_eMvaFunc==SPH_MVAFUNC_ALL;		// (no explicit ANY() / ALL(): the reference warns and takes ANY)
			tOut.blob_attr_id = tCol.m_tLocator.m_iBlobAttrId;
			tOut.n_blob_attrs = tCol.m_tLocator.m_nBlobAttrs;
			tOut.min_value = tFilter.m_iMinValue;
			tOut.max_value = tFilter.m_iMaxValue;
			return tFilter.m_eType!=SPH_FILTER_VALUES || MrkFilterValues ( tFilter, tOut, dValues );

		default:
			return false;	// string, JSON, pointer-typed attributes of the result set ...
	}
}


/// XQQuery_t + CSphQuery + CSphQueryContext -> mrk_query.  false + sWhy = keep the CPU ranker.
inline bool FlattenXQ ( const XQQuery_t & tXQ, const CSphQuery & tQuery, const CSphQueryContext & tCtx, const ISphQwordSetup & tSetup,
	const MrkIndexBinding_t & tIndex, const ISphSchema & tIndexSchema, int iIndexWeight, MrkFlatQuery_t & tOut, CSphString & sWhy )
{
	if ( !tXQ.m_pRoot || tXQ.m_bEmpty )				{ sWhy = "empty query"; return false; }
	if ( tXQ.m_dZones.GetLength() )					{ sWhy = "zones"; return false; }
	MrkFlattener_c tFlat ( tIndex, tSetup, tCtx, tOut, sWhy );
	const int iRoot = tFlat.Node ( tXQ.m_pRoot );
	if ( iRoot<0 )
		return false;

	mrk_query & q = tOut.m_tQuery;
	memset ( &q, 0, sizeof(q) );
	q.nodes = tOut.m_dNodes.Begin();
	q.n_nodes = tOut.m_dNodes.GetLength();
	q.children = tOut.m_dChildren.Begin();
	q.root = iRoot;
	q.ranker = (int)tQuery.m_eRanker;				// MRK_RANK_* keeps ESphRankMode's values
	q.max_matches = tQuery.m_iMaxMatches;
	for ( int i=0; i<tCtx.m_iWeights && i<32; ++i )	// CSphQueryContext::BindWeights already resolved names and defaults
		tOut.m_dWeights.Add ( tCtx.m_dWeights[i] );
	q.field_weights = tOut.m_dWeights.Begin();
	q.n_weights = tOut.m_dWeights.GetLength();
	q.index_weight = 1;								// MatchExtended multiplies by iIndexWeight itself (sphinx.cpp:12220); positive, so the order holds (weight filters: below)
	q.plain_idf = tQuery.m_bPlainIDF;
	q.normalized_tfidf = tQuery.m_bNormalizedTFIDF;
	q.total_docs_override = tCtx.m_iTotalDocs;
	q.local_docs = tCtx.m_pLocalDocs ? tOut.m_dLocalDocs.Begin() : nullptr;
	q.cutoff = tQuery.m_iCutoff>0 ? tQuery.m_iCutoff : 0;	// the device hands back the best of the first m_iCutoff matches; MatchExtended's own count then runs out on the last of them

	// filters: every one of them on the device, or the query stays on the CPU (EarlyReject would thin the K rows out).
	// sphCreateFilters' split (sphinxfilter.cpp:2040-2105): filters on @weight / a weight column make up m_pWeightFilter,
	// the rest m_pFilter; nameless entries are skipped
	int nFilters = 0, nWeightFilters = 0;
	ARRAY_FOREACH ( i, tQuery.m_dFilters )
	{
		const CSphFilterSettings & tFilter = tQuery.m_dFilters[i];
		if ( tFilter.m_sAttrName.IsEmpty() )
			continue;
		const bool bWeight = MrkIsWeightFilter ( tFilter, tIndexSchema );
		int & nKind = bWeight ? nWeightFilters : nFilters;
		if ( nKind>=MRK_MAX_FILTERS )					{ sWhy = "more filters than the device evaluates"; return false; }
		const bool bOk = bWeight
			? MrkFlattenWeightFilter ( tFilter, tOut.m_dWeightFilters[nKind], tOut.m_dWeightFilterValues[nKind] )
			: MrkFlattenFilter ( tFilter, tIndexSchema, tOut.m_dFilters[nKind], tOut.m_dFilterValues[nKind] );
		if ( !bOk )
		{
			sWhy.SetSprintf ( "filter on '%s' is not one the device evaluates (integer VALUES / RANGE, FLOATRANGE, MVA, weight)", tFilter.m_sAttrName.cstr() );
			return false;
		}
		++nKind;
	}
	q.filters = nFilters ? tOut.m_dFilters : nullptr;
	q.n_filters = nFilters;
	q.weight_filters = nWeightFilters ? tOut.m_dWeightFilters : nullptr;
	q.n_weight_filters = nWeightFilters;
	// the weight filter sees the weight AFTER "m_iWeight *= iIndexWeight" (sphinx.cpp:12220-12223): only then does the device
	// multiply too, and the replay divides again before MatchExtended multiplies (MrkFrameReplay_T::Next)
	tOut.m_iIndexWeight = nWeightFilters ? Max ( iIndexWeight, 1 ) : 1;
	q.index_weight = tOut.m_iIndexWeight;
	return true;
}


/// ISphRanker over one device batch
class MrkRankerAdapter_c final : public ISphRanker
{
public:
	static const int FRAME = 256;	///< the buffer length is the ranker's choice: MatchExtended only uses the returned count

	MrkRankerAdapter_c ( mrk_batch * pBatch, mrk_batcher * pBatcher, const MrkIndexBinding_t & tIndex, ISphMatchSorter * pSorter, MrkFlatQuery_t * pFlat, const ISphQwordSetup & tSetup )
		: m_pBatch ( pBatch ), m_pBatcher ( pBatcher ), m_tIndex ( tIndex ), m_pSorter ( pSorter ), m_pFlat ( pFlat )
	{
		Bind ( tSetup );
	}

	/// what Reset() needs to run the same query against the next RAM segment: the parsed query and its context (they outlive the ranker:
	/// DoFullTextSearch holds them, sphinxrt.cpp:6387-6441) and the host's segment lookup
	void EnableRebind ( const XQQuery_t & tXQ, const CSphQuery & tQuery, const CSphQueryContext & tCtx, const ISphSchema & tIndexSchema, int iIndexWeight,
		MrkSegmentLookup_fn fnLookup, void * pUser )
	{
		m_pXQ = &tXQ; m_pQuery = &tQuery; m_pQueryCtx = &tCtx; m_pIndexSchema = &tIndexSchema; m_iIndexWeight = iIndexWeight;
		m_fnLookup = fnLookup; m_pLookupUser = pUser;
	}

	/// run the query on the device; false = the device declined (MRK_E_UNSUPPORTED) or failed: the caller falls back to ExtRanker_*.
	/// With a batcher the query joins whatever the other workers have queued: one mrk_batch_submit for
	/// all of them, the calling coroutine's thread sleeps on a condition variable meanwhile (no HIP on its 128 KB stack); the rows
	/// land in this ranker's own buffers.  Without one: a batch of one on the worker's own mrk_batch.
	bool Run ( CSphString & sError )
	{
		mrk_result tResult;
		memset ( &tResult, 0, sizeof(tResult) );
		if ( m_pBatcher )
		{
			const int iCap = m_pFlat->m_tQuery.max_matches;
			m_dRowIDs.Resize ( iCap );
			m_dWeights.Resize ( iCap );
			if ( mrk_batcher_search ( m_pBatcher, m_tIndex.m_pSegment, &m_pFlat->m_tQuery, m_dRowIDs.Begin(), m_dWeights.Begin(), iCap, &tResult )!=MRK_OK
				|| tResult.status!=MRK_OK )
			{
				sError = mrk_last_error();
				return false;
			}
		} else if ( mrk_batch_submit ( m_pBatch, m_tIndex.m_pSegment, &m_pFlat->m_tQuery, 1 )!=MRK_OK || mrk_batch_wait ( m_pBatch )!=MRK_OK
			|| mrk_batch_result ( m_pBatch, 0, &tResult )!=MRK_OK || tResult.status!=MRK_OK )
		{
			sError = mrk_last_error();
			return false;
		}
		m_tReplay.Start ( tResult, m_pFlat->m_iIndexWeight );
		return true;
	}

	CSphMatch * GetMatchesBuffer() final { return m_dMatches; }

	/// next frame of the device's K best; 0 = end of stream
	int GetMatches() final
	{
		// EarlyReject sets m_pStatic and runs the query's filters (the device already applied them all: none rejects)
		const int iRes = m_tReplay.Next ( m_dMatches, FRAME, [this] ( CSphMatch & tMatch ) { return m_pIndex->EarlyReject ( m_pCtx, tMatch ); } );
		// End of stream: every row above has been Push()ed by now (MatchExtended pushes a frame before it asks for the next
		// one).  The queue counted K pushes; the query matched total_found docs (CSphMatchQueueTraits::m_iTotal,
		// sphinxsort.cpp:724) -- m_iTotal is a public member of ISphMatchSorter, no setter needed.
		if ( !iRes )
			m_tReplay.Finish ( m_pSorter );
		return iRes;
	}

	/// called once behind the match loop (sphinx.cpp:15919, sphinxrt.cpp:6441), also when a cutoff ended the loop on the last
	/// row of a frame and GetMatches() never saw the end of the stream
	void FinalizeCache ( const ISphSchema & ) final
	{
		if ( m_tReplay.AtEnd() )
			m_tReplay.Finish ( m_pSorter );
	}

	/// RT rebinding to the next RAM segment (PerformFullTextSearch, sphinxrt.cpp:6313-6314): the sorter keeps collecting, the ranker moves
	/// on.  What the previous segment still owed the sorter's total is settled first; then the query is flattened against the NEXT
	/// segment's dictionary (keyword ids are per dictionary) and run on that segment's device copy.  A segment without a device copy, or
	/// a failure, ends this segment's stream empty and leaves a warning: the host only offers the device RT queries whose segments all
	/// have one (MrkCreateRanker's contract).
	void Reset ( const ISphQwordSetup & tSetup ) final
	{
		m_tReplay.Finish ( m_pSorter );
		Bind ( tSetup );
		m_tReplay.Exhaust();
		if ( !m_fnLookup )
			return;
		const MrkIndexBinding_t * pNext = m_fnLookup ( tSetup, m_pLookupUser );
		CSphString sWhy;
		if ( pNext && pNext->m_pSegment && pNext->m_pFiles )
		{
			m_tIndex = *pNext;
			CSphScopedPtr<MrkFlatQuery_t> pFlat ( new MrkFlatQuery_t );
			if ( FlattenXQ ( *m_pXQ, *m_pQuery, *m_pQueryCtx, tSetup, m_tIndex, *m_pIndexSchema, m_iIndexWeight, *pFlat.Ptr(), sWhy ) )
			{
				m_pFlat = pFlat.LeakPtr();
				if ( Run ( sWhy ) )
					return;
			}
		} else
			sWhy = "RAM segment without a device copy";
		mrk_result tNone;
		memset ( &tNone, 0, sizeof(tNone) );
		m_tReplay.Start ( tNone, 1 );
		if ( tSetup.m_pWarning )
			tSetup.m_pWarning->SetSprintf ( "device ranker: %s", sWhy.cstr() );
	}

	bool IsCache() const final { return false; }

	int64_t GetTotalFound() const { return m_tReplay.Result().total_found; }
	const MrkFlatQuery_t & GetFlatQuery() const { return *m_pFlat; }

private:
	bool ExtraDataImpl ( ExtraData_e, void ** ) final { return false; }

	void Bind ( const ISphQwordSetup & tSetup )
	{
		m_pIndex = tSetup.m_pIndex;
		m_pCtx = tSetup.m_pCtx;
		for ( CSphMatch & tMatch : m_dMatches )
			tMatch.Reset ( tSetup.m_iDynamicRowitems );	// the ranker owns the buffer and each match's dynamic row (sphinxsearch.cpp:545-550)
	}

	mrk_batch *					m_pBatch;
	mrk_batcher *				m_pBatcher;
	MrkIndexBinding_t			m_tIndex;
	ISphMatchSorter *			m_pSorter;
	CSphScopedPtr<MrkFlatQuery_t> m_pFlat;
	const CSphIndex *			m_pIndex = nullptr;
	CSphQueryContext *			m_pCtx = nullptr;
	MrkFrameReplay_T<CSphMatch>	m_tReplay;
	const XQQuery_t *			m_pXQ = nullptr;			///< EnableRebind
	const CSphQuery *			m_pQuery = nullptr;
	const CSphQueryContext *	m_pQueryCtx = nullptr;
	const ISphSchema *			m_pIndexSchema = nullptr;
	int							m_iIndexWeight = 1;
	MrkSegmentLookup_fn			m_fnLookup = nullptr;
	void *						m_pLookupUser = nullptr;
	CSphVector<uint32_t>		m_dRowIDs;		///< the batcher copies the query's rows here (a shared batch is resubmitted at once)
	CSphVector<int32_t>			m_dWeights;
	CSphMatch					m_dMatches[FRAME];
};


/// What sphCreateRanker calls before its "switch ( tQuery.m_eRanker )" (INTEGRATION.md section 2).  nullptr (sWhy says why)
/// = this query keeps the ExtRanker_* path; never an error.  pBatcher: the index's shared MrkBatcher (all workers' queries
/// leave in common launches); or pBatch: the calling worker's own mrk_batch (one per thread, one query per launch).
/// iIndexWeight: MatchExtended's argument (CSphMultiQueryArgs::m_iIndexWeight, visible where sphCreateRanker is called,
/// sphinx.cpp:15770): only a weight filter needs it on the device.
inline ISphRanker * MrkCreateRanker ( const XQQuery_t & tXQ, const CSphQuery & tQuery, CSphQueryResultMeta & tMeta, const ISphQwordSetup & tSetup,
	const CSphQueryContext & tCtx, const ISphSchema & tIndexSchema, const VecTraits_T<ISphMatchSorter *> & dSorters, DWORD uPackedFactorFlags,
	const MrkIndexBinding_t & tIndex, mrk_batch * pBatch, mrk_batcher * pBatcher, int iIndexWeight, CSphString & sWhy )
{
	if ( !tIndex.m_pSegment || !tIndex.m_pFiles || ( !pBatch && !pBatcher ) )	{ sWhy = "index has no device segment"; return nullptr; }
	if ( !MrkEligible ( tQuery, tCtx, dSorters, uPackedFactorFlags, sWhy ) )
		return nullptr;
	CSphScopedPtr<MrkFlatQuery_t> pFlat ( new MrkFlatQuery_t );
	if ( !FlattenXQ ( tXQ, tQuery, tCtx, tSetup, tIndex, tIndexSchema, iIndexWeight, *pFlat.Ptr(), sWhy ) )
		return nullptr;
	MrkFlatQuery_t * pRawFlat = pFlat.LeakPtr();
	CSphScopedPtr<MrkRankerAdapter_c> pRanker ( new MrkRankerAdapter_c ( pBatch, pBatcher, tIndex, dSorters[0], pRawFlat, tSetup ) );
	if ( !pRanker->Run ( sWhy ) )
		return nullptr;	// MRK_E_UNSUPPORTED (DESIGN.md section 1 lists what the device declines): not an error
	// keyword statistics: once per DISTINCT query word, in query-position order, as sphCreateRanker reports them
	// (sphinxsearch.cpp:4365-4371: hQwords is keyed by the query word; CSphQueryResultMeta::AddStat accumulates,
	// sphinx.cpp:27907-27914, so a repeated keyword must not be reported twice)
	CSphVector<int> dTermNodes, dAtomPos;
	CSphVector<bool> dNotWeighted;
	ARRAY_FOREACH ( i, pRawFlat->m_dNodes )
		if ( pRawFlat->m_dNodes[i].op==MRK_OP_TERM )
		{
			dTermNodes.Add ( i );
			dAtomPos.Add ( pRawFlat->m_dNodes[i].atom_pos );
			dNotWeighted.Add ( pRawFlat->m_dNodes[i].not_weighted!=0 );
		}
	CSphVector<MrkWordStat_t> dStats ( dTermNodes.GetLength() );
	const int nStats = MrkDistinctWords ( dTermNodes.GetLength(), dAtomPos.Begin(), dNotWeighted.Begin(),
		[&] ( int a, int b ) { return pRawFlat->m_dWords[dTermNodes[a]]==pRawFlat->m_dWords[dTermNodes[b]]; }, dStats.Begin() );
	uint32_t nTerms = 0;
	const mrk_dict_entry * pDict = mrk_host_index_dict ( tIndex.m_pFiles, &nTerms );
	for ( int i=0; i<nStats; ++i )
	{
		const int iNode = dTermNodes[dStats[i].m_iNode];
		const mrk_node & tNode = pRawFlat->m_dNodes[iNode];
		const bool bKnown = tNode.term_id>=0 && (uint32_t)tNode.term_id<nTerms;
		tMeta.AddStat ( pRawFlat->m_dDictWords[iNode], bKnown ? pDict[tNode.term_id].docs : 0, bKnown ? pDict[tNode.term_id].hits : 0 );
	}
	return pRanker.LeakPtr();
}
