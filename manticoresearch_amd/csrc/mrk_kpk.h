// mrk_kpk.h -- packed-doclist block access shared by the block-scan kernel and the window-driven tree kernel: a
// 64-entry register chunk of a keyword's block index, a block's words straight into registers, and the decode of the
// two docs a lane owns (mrk_pack.cpp layout).  gfx950 / wave64.
#pragma once
#include "mrk_kcommon.h"

namespace mrk {

struct PkChunk {
  uint32_t first; // block index (within the term) held by lane 0
  uint32_t bp1;   // first possible rowid of block first+lane (INF past the end)
  uint32_t doff;  // word offset of its deltas
  uint32_t w;     // bits per delta / PK_WIDE
};

__device__ __forceinline__ void load_pk_chunk(PkChunk& c, const DevSegment& seg, const DevTerm& T, uint32_t first) {
  const uint32_t i = first + lane_id();
  c.first = first;
  if (i < T.nblocks) {
    const uint32_t g = T.blk_first + i;
    c.bp1 = seg.pk_base[g];
    c.doff = seg.pk_doff[g];
    c.w = seg.pk_w[g];
  } else {
    c.bp1 = INF_ROWID;
    c.doff = 0;
    c.w = 0;
  }
}

struct PkRaw {
  uint32_t lo, hi, attr;
};

// a block's words for this lane, straight into registers (issued early, used late)
__device__ __forceinline__ PkRaw issue_pk(const DevSegment& seg, const DevTerm& T, const PkChunk& c, uint32_t ci) {
  const uint32_t lane = lane_id();
  const uint32_t w = rdlane(c.w, ci);
  const uint32_t* __restrict__ dp = seg.pk_delta + rdlane(c.doff, ci);
  PkRaw r;
  if (w == PK_WIDE) {
    r.lo = dp[lane];
    r.hi = dp[64 + lane];
  } else {
    const uint32_t wi = (lane * 2 * w) >> 5;
    r.lo = dp[wi];
    r.hi = dp[wi + 1];
  }
  r.attr = seg.pk_attr[(uint64_t)(T.blk_first + c.first + ci) * 64 + lane];
  return r;
}

// rowids of the block's docs lane and lane+64 (o0/o1: their offsets from the block base)
__device__ __forceinline__ void decode_pk(const PkRaw& raw, uint32_t w, uint32_t bp1, uint32_t nd, uint32_t& r0, uint32_t& r1,
                                          uint32_t& o0, uint32_t& o1, bool& ok0, bool& ok1) {
  const uint32_t lane = lane_id();
  if (w == PK_WIDE) {
    o0 = raw.lo;
    o1 = raw.hi;
  } else {
    const uint32_t f = __builtin_amdgcn_alignbit(raw.hi, raw.lo, (lane * 2 * w) & 31u);
    const uint32_t mask = (1u << w) - 1u;
    o0 = f & mask;
    o1 = (f >> w) & mask;
  }
  ok0 = lane < nd;
  ok1 = lane + 64 < nd;
  r0 = bp1 + o0;
  r1 = bp1 + o1;
}

} // namespace mrk
