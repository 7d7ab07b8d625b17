"""CPU: the synthetic corpus is a function of (seed, term, GLOBAL rowid) -- SURVEY 8(d) -- so rowid-range shards are
slices of the one corpus.  Checked posting by posting through the oracle's decoder: rowids, field masks, hit counts
and every hit position of uneven shards laid end to end equal the unsharded segment's."""
import numpy as np

import manticoresearch_amd as m

PROBS = [0.4, 0.07, 0.003]


def _postings(orc, hi, term):
    oi = orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), hi.total_docs, hi.skiplist_block_size, hi.hit_format, hi.n_fields)
    rowid, fields, hits, hp = oi.decode_doclist(term)
    # multi-hit docs: the hit positions behind the doc's hitlist pointer (bit 63 = the one hit was inlined)
    def hits_of(i):
        return oi.decode_hits(int(hp[i])) if not (int(hp[i]) >> 63) else [int(hp[i]) & 0xFFFFFFFF]

    return rowid, fields, hits, hits_of


def test_shards_are_slices_of_the_one_corpus(orc):
    n = 300_000 + 12_345  # not a multiple of the generator's chunk size
    whole = m.synth_index(n, PROBS, seed=77, skiplist_block_size=32, n_threads=2)
    cuts = [0, 65_536, 150_001, n]  # a cut on a chunk border and one inside a chunk
    shards = [m.synth_index(cuts[i + 1] - cuts[i], PROBS, seed=77, skiplist_block_size=32, n_threads=2, rowid_base=cuts[i]) for i in range(3)]
    for t in range(len(PROBS)):
        w_row, w_fld, w_hits, w_hits_of = _postings(orc, whole, t)
        rows, flds, hits, at = [], [], [], 0
        for s, sh in enumerate(shards):
            r, f, h, hits_of = _postings(orc, sh, t)
            rows.append(r.astype(np.int64) + cuts[s])
            flds.append(f)
            hits.append(h)
            for i in range(0, len(r), max(1, len(r) // 300)):  # every hit position of a sample of the shard's docs
                assert int(w_row[at + i]) == int(r[i]) + cuts[s]
                assert w_hits_of(at + i) == hits_of(i)
            at += len(r)
        assert np.array_equal(np.concatenate(rows), w_row.astype(np.int64))
        assert np.array_equal(np.concatenate(flds), w_fld) and np.array_equal(np.concatenate(hits), w_hits)
        assert int(whole.dict[t]["docs"]) == sum(int(sh.dict[t]["docs"]) for sh in shards)


def test_shard_shorthand_matches_rowid_base():
    a = m.synth_index(70_000, PROBS, seed=5, shard=1, n_threads=1)
    b = m.synth_index(70_000, PROBS, seed=5, rowid_base=70_000, n_threads=1)
    assert np.array_equal(a.spd, b.spd) and np.array_equal(a.spp, b.spp) and np.array_equal(a.spe, b.spe)
    c = m.synth_index(70_000, PROBS, seed=5, shard=0, n_threads=1)
    assert not np.array_equal(a.spd[: min(a.spd.size, c.spd.size)], c.spd[: min(a.spd.size, c.spd.size)])
