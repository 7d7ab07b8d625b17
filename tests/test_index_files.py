"""Real index ingestion: mrk_index_open over index files the reference's own tests hold, plus a
round trip through files this test writes in the documented layout (doc/internals-index-format.txt:97-170 and the
readers cited in csrc/mrk_files.cpp).

Fixtures (tests/golden/indexes/, data files copied from the reference tree -- bytes its indexer wrote):
  t250_plain2   test/test_250/data/plain2.*     v54, rows (3 'third'), (4 'fourth')            (test_250/test.xml)
  t233_test     test/test_233/data/test.*       v54, row  (3 'RELOAD INDEX')                   (test_233/model.bin)
  t233_reload   test/test_233/data/reload.*     v54, row  (2 'RELOAD INDEX FROM')              (test_233/model.bin)
  t406_index0   test/test_406/data/index.0.*    v57 RT disk chunk, row (1 'doc one')           (test_406/model.bin)
No GPU: the library's host side only; the oracle decodes the posting bytes."""
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
IDX = os.path.join(HERE, "golden", "indexes")

EXPECT = {
    # name: (version, skiplist block, field, total_docs, total_bytes, [(keyword, [positions in the one doc that holds it], rowid)])
    "t250_plain2": (54, 128, "text", 2, 11, [("fourth", [1], 1), ("third", [1], 0)]),
    "t233_test": (54, 128, "title", 1, 12, [("index", [2], 0), ("reload", [1], 0)]),
    "t233_reload": (54, 128, "title", 1, 17, [("from", [3], 0), ("index", [2], 0), ("reload", [1], 0)]),
    "t406_index0": (57, 32, "title", 1, 7, [("doc", [1], 0), ("one", [2], 0)]),
}


ATTRS = {  # name -> (ESphAttr, bit offset, bit count); rows of the .spa file
    "t250_plain2": ({"id": (6, 0, 64), "mode": (1, 64, 32)}, [[3, 0, 2], [4, 0, 2]]),
    "t233_test": ({"id": (6, 0, 64), "$_blob_locator": (6, 64, 64), "gid": (1, 128, 32), "title": (7, 0, 0)}, [[3, 0, 8, 0, 123]]),
    "t233_reload": ({"id": (6, 0, 64), "$_blob_locator": (6, 64, 64), "gid": (1, 128, 32), "title": (7, 0, 0)}, [[2, 0, 8, 0, 123]]),
    "t406_index0": ({"id": (6, 0, 64)}, [[1, 0]]),
}


def orc_index(orc, hi):
    return orc.Index(hi.spd, hi.spp, hi.spe, hi.dict.view(orc.DICT_DTYPE), hi.total_docs, hi.skiplist_block_size,
                     hi.hit_format, hi.n_fields)


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_reference_index_files(orc, name):
    import manticoresearch_amd as m

    version, block, fld, total_docs, total_bytes, words = EXPECT[name]
    hi = m.open_index(os.path.join(IDX, name))
    assert hi.info["version"] == version and hi.info["word_dict"] == 1 and hi.info["hitless"] == 0
    assert hi.skiplist_block_size == block and hi.hit_format == m.SPH_HIT_FORMAT_INLINE
    assert hi.fields == [fld] and hi.n_fields == 1
    assert hi.total_docs == total_docs and hi.info["total_bytes"] == total_bytes
    assert hi.words == [w for w, _, _ in words]  # dictionary order
    assert hi.dead_rows is None and hi.info["n_dead"] == 0
    # schema attributes + the .spa rows: ids (and gid / mode) as that test's expected result rows list them
    want_attrs, want_rows = ATTRS[name]
    assert hi.attrs == want_attrs and hi.attr_rows.tolist() == want_rows
    oi = orc_index(orc, hi)
    for t, (w, pos, rowid) in enumerate(words):
        assert hi.find_word(w) == t
        e = hi.dict[t]
        assert int(e["docs"]) == 1 and int(e["hits"]) == len(pos) and int(e["doclist_len"]) == 5
        r, f, nh, _ = oi.decode_doclist(t)
        assert list(r) == [rowid] and list(f) == [1] and list(nh) == [len(pos)]
    assert hi.find_word("zz") == -1 and hi.find_word(words[0][0][:-1]) == -1 and hi.find_word(words[0][0] + "x") == -1
    # doclists tile .spd: dummy byte + 5 bytes per keyword
    assert len(hi.spd) == 1 + 5 * len(words) and len(hi.spp) == 1 and len(hi.spe) == 1
    assert sorted(int(e["doclist_off"]) for e in hi.dict) == [1 + 5 * i for i in range(len(words))]


def test_reference_index_positions_through_the_rankers(orc):
    """'RELOAD INDEX FROM': the words sit at positions 1, 2, 3 of one field, 'from' closes the field."""
    import manticoresearch_amd as m

    hi = m.open_index(os.path.join(IDX, "t233_reload"))
    oi = orc_index(orc, hi)
    t = {w: hi.find_word(w) for w in ("reload", "index", "from")}
    kw = lambda w, p: orc.term(t[w], p)
    # N = n = 1: IDF 0, so PROXIMITY_BM25 = 1000 * LCS + 500
    r = orc.search(oi, orc.op(orc.OP_AND, kw("reload", 1), kw("index", 2), kw("from", 3)), ranker=orc.RANK_PROXIMITY_BM25)
    assert list(r.rowid) == [0] and list(r.weight) == [3500]
    r = orc.search(oi, orc.op(orc.OP_AND, kw("index", 1), kw("reload", 2)), ranker=orc.RANK_PROXIMITY_BM25)
    assert list(r.weight) == [1500]
    assert orc.search(oi, orc.op(orc.OP_PHRASE, kw("reload", 1), kw("index", 2)), ranker=orc.RANK_PROXIMITY_BM25).total_found == 1
    assert orc.search(oi, orc.op(orc.OP_PHRASE, kw("index", 1), kw("reload", 2)), ranker=orc.RANK_PROXIMITY_BM25).total_found == 0
    assert orc.search(oi, orc.op(orc.OP_PHRASE, kw("reload", 1), kw("from", 2)), ranker=orc.RANK_PROXIMITY_BM25).total_found == 0
    # filters over the ingested .spa rows, by attribute name: gid = 123, id = 2
    oi.attrs = hi.attr_rows
    _, off, cnt = hi.attrs["gid"]
    assert orc.search(oi, kw("index", 1), ranker=orc.RANK_BM25, filters=[dict(bit_offset=off, bit_count=cnt, values=[123])]).total_found == 1
    assert orc.search(oi, kw("index", 1), ranker=orc.RANK_BM25, filters=[dict(bit_offset=off, bit_count=cnt, values=[122, 124])]).total_found == 0
    _, off, cnt = hi.attrs["id"]
    assert orc.search(oi, kw("index", 1), ranker=orc.RANK_BM25, filters=[dict(bit_offset=off, bit_count=cnt, min=2, max=2)]).total_found == 1
    assert orc.search(oi, kw("index", 1), ranker=orc.RANK_BM25, filters=[dict(bit_offset=off, bit_count=cnt, min=2, max=2, exclude=True)]).total_found == 0
    # SPH04: 'reload index from' is the whole field => exact-match bonus: 4 * lcs + 2 (head) + 1 (exact)
    r = orc.search(oi, orc.op(orc.OP_AND, kw("reload", 1), kw("index", 2), kw("from", 3)), ranker=orc.RANK_SPH04)
    assert list(r.weight) == [(4 * 3 + 2 + 1) * 1000 + 500]


# ------------------------------------------------------------------ files written here, in the documented layout
def zint(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    return bytes(reversed(out))


def sph_str(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def write_index(prefix, hi, words, version=62, word_dict=True, wordids=None, dead_rows=(), checkpoint_every=64):
    """.sph/.spi/.spd/.spp/.spe/.spm for the postings of `hi` (term t <-> words[t] / wordids[t], already sorted)."""
    n_fields = hi.n_fields
    spi = bytearray(b"\x01")
    cps = []
    block = hi.skiplist_block_size
    for t in range(len(hi.dict)):
        e = hi.dict[t]
        docs, hits, off, skip = int(e["docs"]), int(e["hits"]), int(e["doclist_off"]), int(e["skiplist_off"])
        if t % checkpoint_every == 0:
            if t:
                spi += b"\x00" if word_dict else b"\x00" + zint(int(hi.dict[t - 1]["doclist_len"]))
            cps.append((words[t] if word_dict else wordids[t], len(spi)))
            prev_w, prev_id, prev_off = b"", 0, 0
        if word_dict:
            w = words[t].encode()
            match = 0
            while match < min(len(w), len(prev_w)) and w[match] == prev_w[match]:
                match += 1
            delta = len(w) - match
            if delta <= 8 and match <= 15:
                spi.append(0x80 | ((delta - 1) << 4) | match)
            else:
                spi.append(delta)
                spi.append(match)
            spi += w[match:] + zint(off) + zint(docs) + zint(hits)
            if docs >= 256:
                spi.append(0x42)  # doclist size hint
            if docs > block:
                spi += zint(skip)
            prev_w = w
        else:
            spi += zint(wordids[t] - prev_id) + zint(off - prev_off) + zint(docs) + zint(hits)
            if docs > block:
                spi += zint(skip)
            prev_id, prev_off = wordids[t], off
    spi += b"\x00" if word_dict else b"\x00" + zint(int(hi.dict[-1]["doclist_len"]))
    cp_off = len(spi)
    for key, off in cps:
        spi += (sph_str(key) if word_dict else struct.pack("<Q", key)) + struct.pack("<Q", off)
    h = bytearray(struct.pack("<II", 0x58485053, version))
    h += struct.pack("<I", n_fields)
    for f in range(n_fields):
        if version >= 57:
            h += sph_str("f%d" % f) + struct.pack("<IB", 1, 0)
        else:
            h += sph_str("f%d" % f) + struct.pack("<IIIIB", 0, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0) + (b"" if version < 61 else struct.pack("<I", 0))
    h += struct.pack("<I", 1) + sph_str("id") + struct.pack("<IIIIB", 6, 0, 0, 64, 0) + (struct.pack("<I", 0) if version >= 61 else b"")
    h += struct.pack("<QIBII", cp_off, len(cps), 0, 0, 0)
    h += struct.pack("<IQ", hi.total_docs, 12345)
    # index settings
    h += struct.pack("<III", 0, 0, 0) + b"\x00" + sph_str("") + sph_str("") + b"\x00" + struct.pack("<II", 0, hi.hit_format) + b"\x00"
    h += sph_str("") + struct.pack("<IIII", 15, 1, 1, 16384) + b"\x00" + sph_str("") + b"\x00" + b"\x00" + sph_str("") + sph_str("")
    h += struct.pack("<Q", 0)
    if version >= 56:
        h += struct.pack("<I", block)
    if version >= 60:
        h += sph_str("")
    # tokenizer: embedded synonyms present to exercise that branch
    h += b"\x01" + sph_str("non_cjk") + struct.pack("<I", 1) + b"\x01" + struct.pack("<I", 2) + sph_str("a => b") + sph_str("c => d")
    h += sph_str("exc.txt") + struct.pack("<QQQI", 14, 1, 2, 3) + sph_str("") + sph_str("") + struct.pack("<I", 0) + sph_str("") + sph_str("") + sph_str("")
    # dictionary: embedded stopwords (zipped ids), one stopword file, embedded wordforms + one wordform file
    h += sph_str("stem_en") + sph_str("") + b"\x01" + struct.pack("<I", 2) + zint(1 << 40) + zint(77)
    h += sph_str("stop.txt") + struct.pack("<I", 1) + sph_str("stop.txt") + struct.pack("<QQQI", 1, 2, 3, 4)
    h += b"\x01" + struct.pack("<I", 1) + sph_str("walks > walk") + struct.pack("<I", 1) + sph_str("wf.txt") + struct.pack("<QQQI", 5, 6, 7, 8)
    h += struct.pack("<I", 1) + (b"\x01" if word_dict else b"\x00") + b"\x00" + sph_str("")
    h += struct.pack("<QQQ", 0, 0, 0) + struct.pack("<I", 0)
    for ext, data in (("sph", bytes(h)), ("spi", bytes(spi)), ("spd", bytes(hi.spd)), ("spp", bytes(hi.spp)), ("spe", bytes(hi.spe))):
        with open(prefix + "." + ext, "wb") as f:
            f.write(data)
    bm = np.zeros((hi.total_docs + 31) // 32, np.uint32)
    for r in dead_rows:
        bm[r >> 5] |= np.uint32(1 << (r & 31))
    with open(prefix + ".spm", "wb") as f:
        f.write(bm.tobytes())


def synth(m, n_terms=150, n_docs=3000, block=32, fmt=1):
    probs = [min(0.5, 0.9 / (1 + t) ** 0.7) for t in range(n_terms)]
    return m.synth_index(n_docs, probs, seed=77, n_fields=3, max_pos=40, skiplist_block_size=block, hit_format=fmt, end_markers=True, n_threads=2)


@pytest.mark.parametrize("version,word_dict,block,fmt", [(62, True, 32, 1), (54, True, 128, 0), (58, False, 32, 1), (62, False, 128, 1)])
def test_written_index_round_trip(tmp_path, orc, version, word_dict, block, fmt):
    import manticoresearch_amd as m

    hi = synth(m, block=block, fmt=fmt)
    n = len(hi.dict)
    assert int(hi.dict["docs"].max()) >= 256 and int((hi.dict["docs"] > block).sum()) > 10  # hint bytes and skiplists occur
    words = sorted("w%05d%s" % (t * 7919 % 100003, "x" * (t % 23)) for t in range(n))  # shared prefixes, some > 15 bytes
    wordids = [1000 + 17 * t + (t * t) % 13 for t in range(n)]
    dead = [0, 31, 32, 2999]
    prefix = str(tmp_path / "idx")
    write_index(prefix, hi, words, version=version, word_dict=word_dict, wordids=wordids, dead_rows=dead)
    got = m.open_index(prefix)
    assert got.info["version"] == version and got.info["word_dict"] == int(word_dict) and got.info["n_checkpoints"] == 3
    assert got.skiplist_block_size == (128 if version < 56 else block) == block and got.hit_format == fmt and got.n_fields == 3
    assert got.total_docs == hi.total_docs and got.fields == ["f0", "f1", "f2"]
    for col in ("doclist_off", "doclist_len", "docs", "hits"):
        assert (got.dict[col] == hi.dict[col]).all(), col
    big = hi.dict["docs"] > block
    assert (got.dict["skiplist_off"][big] == hi.dict["skiplist_off"][big]).all() and (got.dict["skiplist_off"][~big] == 0).all()
    assert bytes(got.spd) == bytes(hi.spd) and bytes(got.spp) == bytes(hi.spp) and bytes(got.spe) == bytes(hi.spe)
    if word_dict:
        assert got.words == words
        assert [got.find_word(w) for w in words[::7]] == list(range(0, n, 7))
        assert got.find_word("w") == -1 and got.find_word(words[3] + "y") == -1
    else:
        assert [int(x) for x in got.dict["wordid"]] == wordids
        assert [got.find_wordid(w) for w in wordids[::5]] == list(range(0, n, 5)) and got.find_wordid(7) == -1
    assert list(got.dead_rows) == dead and got.info["n_dead"] == 4
    assert [int(r) for r in np.flatnonzero(np.unpackbits(got.dead_bitmap.view(np.uint8), bitorder="little"))] == dead
    # the same answers as from the index the files were written from
    a, b = orc_index(orc, hi), orc_index(orc, got)
    root = orc.op(orc.OP_AND, orc.term(3, 1), orc.term(9, 2))
    ra, rb = orc.search(a, root, ranker=orc.RANK_PROXIMITY_BM25), orc.search(b, root, ranker=orc.RANK_PROXIMITY_BM25)
    assert ra.total_found == rb.total_found > 0 and (ra.rowid == rb.rowid).all() and (ra.weight == rb.weight).all()


def test_open_errors(tmp_path):
    import manticoresearch_amd as m

    src = os.path.join(IDX, "t233_reload")
    def variant(name, **patch):
        p = str(tmp_path / name)
        for ext in ("sph", "spi", "spd", "spp", "spe", "spm"):
            data = open(src + "." + ext, "rb").read()
            if ext in patch:
                data = patch[ext](data)
            if data is not None:
                open(p + "." + ext, "wb").write(data)
        return p

    with pytest.raises(m.MrkError, match="cannot open"):
        m.open_index(str(tmp_path / "none"))
    with pytest.raises(m.MrkError, match="magic"):
        m.open_index(variant("magic", sph=lambda d: b"XXXX" + d[4:]))
    with pytest.raises(m.MrkError, match="v.63"):
        m.open_index(variant("new", sph=lambda d: d[:4] + struct.pack("<I", 63) + d[8:]))
    with pytest.raises(m.MrkError, match="v.53"):
        m.open_index(variant("old", sph=lambda d: d[:4] + struct.pack("<I", 53) + d[8:]))
    with pytest.raises(m.MrkError, match="unexpected eof"):
        m.open_index(variant("short", sph=lambda d: d[:200]))
    with pytest.raises(m.MrkError, match="cannot open"):
        m.open_index(variant("nospd", spd=lambda d: None))
    with pytest.raises(m.MrkError, match="does not end"):
        m.open_index(variant("cutspd", spd=lambda d: d[:-2]))
    with pytest.raises(m.MrkError, match="checkpoint"):
        m.open_index(variant("cutspi", spi=lambda d: d[:30]))
    hi = m.open_index(variant("nospm", spm=lambda d: None))  # the dead-row map is optional
    assert hi.dead_rows is None and hi.words == ["from", "index", "reload"]


def test_header_counts_are_bounded_before_they_are_trusted(tmp_path):
    """Every dword / qword of a real header overwritten with an extreme count (n_checkpoints = 0x40000000, m_iDocinfo =
    2^62 + 1, ...): the open ends in MRK_OK or an error code -- never an abort (std::bad_alloc out of an extern "C"
    function), never a multi-gigabyte allocation, never more attribute rows than the .spa file holds."""
    import resource
    import time

    import manticoresearch_amd as m

    for name in ("t233_test", "t406_index0"):
        src = os.path.join(IDX, name)
        files = {ext: open(src + "." + ext, "rb").read() for ext in ("sph", "spi", "spd", "spp", "spe", "spm", "spa") if os.path.exists(src + "." + ext)}
        p = str(tmp_path / name)
        for ext, data in files.items():
            open(p + "." + ext, "wb").write(data)
        sph = files["sph"]
        t0 = time.time()
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        opened = 0
        for k in range(8, len(sph) - 3):
            for v in (struct.pack("<I", 0x40000000), struct.pack("<I", 0x08000000), struct.pack("<I", 0xFFFFFFFF), struct.pack("<Q", (1 << 62) + 1)):
                open(p + ".sph", "wb").write(sph[:k] + v + sph[k + len(v):])
                try:
                    hi = m.open_index(p)
                except m.MrkError as e:
                    assert e.code in (-2, -4, -5), (k, e)
                    continue
                opened += 1
                if hi.attr_rows is not None:
                    assert hi.attr_rows.nbytes <= len(files.get("spa", b"")) and hi.attr_rows.shape[0] >= hi.total_docs
        assert time.time() - t0 < 60
        assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0 < 600_000  # KiB: no count was allocated from unchecked
        assert opened > 0


# ---------------------------------------------------------------------------------------------------------------------
# RT RAM chunks: <prefix>.meta + <prefix>.ram, the files RtIndex_c::SaveMeta / SaveRamChunk wrote
#   t406_index    test/test_406/data/index.{meta,ram}          meta v17 (stored field: a docstore rides in the segment), RAM row (2 'doc two');
#                                                              with its disk chunk t406_index0 the table test_406 imports: rows 2, 1 (model.bin)
#   t406_idx320   test/test_406/data/rel320/idx320.{meta,ram}  written by release 3.2.0; 'select .. where match(@keywords kw1)' -> id 10 (model.bin)
#   ql_rt         test/ql/data/rt.{meta,ram}                   meta v7: older than the reference itself still loads (LoadMeta: "prior to v.14")
def test_rt_ram_chunks(orc):
    import manticoresearch_amd as m

    segs = m.open_rt_ram(os.path.join(IDX, "t406_index"))
    assert len(segs) == 1
    hi = segs[0]
    assert (hi.total_docs, hi.fields, hi.words, hi.info["word_dict"], hi.info["n_dead"]) == (1, ["title"], ["doc", "two"], 1, 0)
    assert hi.attrs == {"id": (6, 0, 64)} and hi.attr_rows.tolist() == [[2, 0]]
    oi = orc_index(orc, hi)
    for w, pos in (("doc", 1), ("two", 2)):
        t = hi.find_word(w)
        rows, fields, nhits, _ = oi.decode_doclist(t)
        assert rows.tolist() == [0] and fields.tolist() == [1] and nhits.tolist() == [1]
    # the RAM segment answers like a disk chunk: rank both parts of the table and merge (ids from the attribute rows)
    dk = m.open_index(os.path.join(IDX, "t406_index0"))
    got = []
    for part in (hi, dk):
        r = orc.FlatQuery(orc.term(part.find_word("doc"), 1), ranker=orc.RANK_BM25).run(orc_index(orc, part))
        got += [int(part.attr_rows[x][0]) for x in r.rowid]
    assert sorted(got) == [1, 2]  # test_406/model.bin: "select * from test1" -> ids 2, 1

    (old,) = m.open_rt_ram(os.path.join(IDX, "t406_idx320"))
    assert (old.total_docs, old.fields, old.words, old.info["version"]) == (1, ["title", "keywords"], ["data", "kw1", "of"], 58)
    assert int(old.attr_rows[0][0]) == 10 and old.attrs["idd"] == (1, 128, 32)
    q = orc.FlatQuery(orc.term(old.find_word("kw1"), 1, field_mask=1 << old.fields.index("keywords")), ranker=orc.RANK_PROXIMITY_BM25)
    r = q.run(orc_index(orc, old))
    assert [int(old.attr_rows[x][0]) for x in r.rowid] == [10]  # model.bin: match('@keywords kw1') -> id 10
    assert orc.FlatQuery(orc.term(old.find_word("kw1"), 1, field_mask=1 << old.fields.index("title")), ranker=orc.RANK_BM25).run(orc_index(orc, old)).total_found == 0

    with pytest.raises(m.MrkError, match="meta v.7"):
        m.open_rt_ram(os.path.join(IDX, "ql_rt"))
    with pytest.raises(m.MrkError):
        m.open_rt_ram(os.path.join(IDX, "no_such_index"))


def test_rt_ram_rejects_damage(tmp_path):
    """Truncations and bit flips of a RAM chunk end in an error code or in a segment that passes validation, never in a crash."""
    import manticoresearch_amd as m

    meta = open(os.path.join(IDX, "t406_idx320.meta"), "rb").read()
    ram = open(os.path.join(IDX, "t406_idx320.ram"), "rb").read()
    rng = np.random.default_rng(5)
    n_err = 0
    for trial in range(400):
        mb, rb = bytearray(meta), bytearray(ram)
        target = rb if trial % 4 else mb
        k = int(rng.integers(0, 3))
        if k == 0:
            del target[int(rng.integers(0, len(target))):]
        elif k == 1:
            target[int(rng.integers(0, len(target)))] ^= 1 << int(rng.integers(0, 8))
        else:
            i = int(rng.integers(0, len(target) - 4))
            target[i:i + 4] = struct.pack("<I", int(rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x40000000, 65536])))
        p = str(tmp_path / "x")
        open(p + ".meta", "wb").write(mb)
        open(p + ".ram", "wb").write(rb)
        try:
            for hi in m.open_rt_ram(p):
                m.validate_index(hi)
        except m.MrkError:
            n_err += 1
    assert n_err > 100


def _ram_segments(ram: bytes, word_dict: bool):
    """The vectors of a .ram file's segments, cut out as data (RtIndex_c::SaveRamChunk's layout, sphinxrt.cpp:4034-4100: per segment
    rows, alive rows, a dword, then length-prefixed vectors words [checkpoints] docs hits attributes, the dead-row map, blobs ...)."""
    at = 8
    n_seg = struct.unpack_from("<I", ram, 4)[0]
    out = []

    def vec(elem):
        nonlocal at
        n = struct.unpack_from("<I", ram, at)[0]
        at += 4
        v = ram[at:at + n * elem]
        at += n * elem
        return v

    for _ in range(n_seg):
        rows = struct.unpack_from("<I", ram, at)[0]
        at += 12
        words = vec(1)
        if word_dict:
            vec(1)
        n_cp = struct.unpack_from("<I", ram, at)[0]
        at += 4 + 16 * n_cp
        docs, hits = vec(1), vec(1)
        vec(4)  # attribute rows
        at += ((rows + 31) // 32) * 4
        vec(1)  # blobs
        out.append((rows, words, docs, hits))
        break  # (what follows -- docstore, infixes -- depends on the meta version: the first segment is enough here)
    return out


def test_live_rt_segment_in_memory_equals_the_file_reader():
    """mrk_rt_segment_open: a RAM segment's three byte vectors handed over in memory (a live RtSegment_t: no .ram file in between)
    give the same disk-format postings, dictionary and keywords as mrk_rt_ram_open reading them from the reference's .ram file."""
    import manticoresearch_amd as m

    for name in ("t406_index", "t406_idx320"):
        from_file = m.open_rt_ram(os.path.join(IDX, name))[0]
        ram = open(os.path.join(IDX, name + ".ram"), "rb").read()
        rows, words, docs, hits = _ram_segments(ram, bool(from_file.info["word_dict"]))[0]
        live = m.open_rt_segment(words, docs, hits, rows, word_dict=bool(from_file.info["word_dict"]), words_checkpoint=64,
                                 skiplist_block_size=from_file.skiplist_block_size, hit_format=from_file.hit_format, n_fields=from_file.n_fields)
        assert live.total_docs == from_file.total_docs == rows
        assert bytes(live.spd) == bytes(from_file.spd) and bytes(live.spp) == bytes(from_file.spp) and bytes(live.spe) == bytes(from_file.spe)
        assert np.array_equal(live.dict, from_file.dict) and live.words == from_file.words
        m.validate_index(live)
    with pytest.raises(m.MrkError):  # damaged vectors end in an error code
        m.open_rt_segment(words[:-1] + b"\xff", docs[: len(docs) // 2], hits, rows, n_fields=from_file.n_fields)


def test_rt_ram_row_count_that_wraps_the_dead_map_size(tmp_path):
    """A RAM segment's row count is an untrusted dword: (rows + 31) / 32 wraps to 0 words for rows >= 0xFFFFFFE1 -- the dead-row map
    came out empty and the count of dead rows read past it (advisor, round 2: SIGSEGV with rows = 0xFFFFFFF0 and the 4-byte map
    removed).  Such a file must end in an error code, with the map in place and with any 4 bytes behind the header removed."""
    import manticoresearch_amd as m

    meta = open(os.path.join(IDX, "t406_index.meta"), "rb").read()
    ram = open(os.path.join(IDX, "t406_index.ram"), "rb").read()
    p = str(tmp_path / "x")
    open(p + ".meta", "wb").write(meta)
    n = 0
    for rows in (0xFFFFFFF0, 0xFFFFFFFF, 0xFFFFFFE1, 0x80000000):
        cuts = [None] + list(range(12, len(ram) - 4, 4))
        for cut in cuts:
            rb = bytearray(ram)
            rb[8:12] = struct.pack("<I", rows)
            if cut is not None:
                del rb[cut:cut + 4]
            open(p + ".ram", "wb").write(rb)
            with pytest.raises(m.MrkError):
                m.open_rt_ram(p)
            n += 1
    assert n > 20


def test_blob_pool_of_reference_files():
    """The blob pool travels with the index: the string attribute of test_233's index (the used part of its .spb: the first qword
    is the used size) and of the release-3.2.0 RT segment decode to the values the tests' model.bin shows."""
    import manticoresearch_amd as m

    def blob_attr(pool, off, attr_id, n_attrs):  # GetBlobAttr, attribute.cpp:495-513
        sz = (1, 2, 4)[pool[off]]
        lens = [int.from_bytes(bytes(pool[off + 1 + i * sz: off + 1 + (i + 1) * sz]), "little") for i in range(n_attrs)]
        l0 = lens[attr_id - 1] if attr_id else 0
        data = off + 1 + n_attrs * sz
        return bytes(pool[data + l0: data + lens[attr_id]])

    hi = m.open_index(os.path.join(IDX, "t233_test"))
    assert hi.n_blob_attrs == 1 and int(hi.attr_rows[0][2]) == 8  # the row's blob locator: right behind the size qword
    assert blob_attr(hi.blobs, 8, 0, 1) == b"RELOAD INDEX"
    (rt,) = m.open_rt_ram(os.path.join(IDX, "t406_idx320"))
    assert rt.n_blob_attrs == 1 and blob_attr(rt.blobs, int(rt.attr_rows[0][2]), 0, 1) == b"kw1 at doc10"  # test_406/model.bin: postingtitle
