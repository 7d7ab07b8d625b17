"""CPU: untrusted posting bytes are validated on the host whatever path the segment ends up on -- mrk_segment_validate is
the host-only half of mrk_segment_create (the create call runs the same walk on every doclist the load-time transcode
did not walk to its end).  The cases the packed-format transcode used to let through: a doclist whose first entry makes
the transcode decline (field mask wider than 8 bits) with a rowid beyond the row count behind it, and a segment with
more than 8 fields (never transcoded) holding such a rowid.  The same descriptors fail mrk_segment_create on the GPU
(tests/test_gpu_parity.py::test_corrupt_postings_are_rejected_at_load)."""
import numpy as np
import pytest

import manticoresearch_amd as m


def crafted():
    """-> {name: HostIndex} of descriptors that must be rejected, plus "good" ones that must pass."""
    hp = lambda f, pos: (f << 24) | pos
    out = {}
    # (a) term 0: doc 0 has its single hit in field 9 (mask 0x200: "wider than 8 bits" for the transcode), doc 1 is rowid 50
    W = np.array([1, 1], np.uint64)
    R = np.array([0, 50], np.uint32)
    H = np.array([hp(9, 1), hp(0, 3)], np.uint32)
    hi = m.index_from_hits(W, R, H, n_terms=1, total_docs=51, n_fields=2)
    out["good_wide_mask"] = hi
    out["wide_mask_then_bad_rowid"] = m.HostIndex(hi.spd, hi.spp, hi.spe, hi.dict.copy(), 10, hi.skiplist_block_size, hi.hit_format, 2)
    # (b) a 9-field segment (never transcoded), rowid 50 of 10 rows
    H9 = np.array([hp(8, 1), hp(8, 2), hp(0, 3)], np.uint32)
    hi9 = m.index_from_hits(np.array([1, 1, 1], np.uint64), np.array([0, 0, 50], np.uint32), H9, n_terms=1, total_docs=51, n_fields=9)
    out["good_9_fields"] = hi9
    out["nine_fields_bad_rowid"] = m.HostIndex(hi9.spd, hi9.spp, hi9.spe, hi9.dict.copy(), 10, hi9.skiplist_block_size, hi9.hit_format, 9)
    # a hitlist offset past .spp behind a wide mask; descending rowids; a doclist running into its neighbour
    hi2 = m.index_from_hits(np.array([1, 1, 1, 1, 1], np.uint64), np.array([0, 7, 7, 9, 9], np.uint32),
                            np.array([hp(9, 1), hp(0, 1), hp(0, 5), hp(1, 2), hp(1, 3)], np.uint32), n_terms=1, total_docs=10, n_fields=2)
    out["good_hitlists"] = hi2
    out["wide_mask_then_hitlist_past_spp"] = m.HostIndex(hi2.spd, hi2.spp[:4], hi2.spe, hi2.dict.copy(), 10, hi2.skiplist_block_size, hi2.hit_format, 2)
    d = hi2.dict.copy()
    d[0]["docs"] += 1
    out["more_docs_than_entries"] = m.HostIndex(hi2.spd, hi2.spp, hi2.spe, d, 10, hi2.skiplist_block_size, hi2.hit_format, 2)
    d = hi2.dict.copy()
    d[0]["doclist_off"] = 2**63
    out["doclist_offset_wraps"] = m.HostIndex(hi2.spd, hi2.spp, hi2.spe, d, 10, hi2.skiplist_block_size, hi2.hit_format, 2)
    return out


@pytest.mark.parametrize("name", ["wide_mask_then_bad_rowid", "nine_fields_bad_rowid", "wide_mask_then_hitlist_past_spp",
                                  "more_docs_than_entries", "doclist_offset_wraps"])
def test_bad_postings_fail_validation(name):
    with pytest.raises(m.MrkError) as e:
        m.validate_index(crafted()[name])
    assert e.value.code == -5  # MRK_E_FORMAT


@pytest.mark.parametrize("name", ["good_wide_mask", "good_9_fields", "good_hitlists"])
def test_wellformed_postings_pass(name):
    m.validate_index(crafted()[name])


def test_synthetic_segments_pass():
    for fmt in (0, 1):
        for block in (32, 128):
            m.validate_index(m.synth_index(20000, [0.3, 0.01], seed=5, skiplist_block_size=block, hit_format=fmt, n_threads=1))
