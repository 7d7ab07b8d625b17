"""Test helpers: a tiny text -> hits pipeline so that the reference's own test corpora
(test/test_019, test_037, test_322, gtests_rtstuff.cpp) can be turned into postings.

Only what those corpora need: lower-casing, [a-z0-9_] + Cyrillic words, single CJK
characters as 1-grams, min_word_len with position-preserving overshort skips
(overshort_step=1), 1-based in-field positions, field-end marker on a field's last hit.
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

_TOKEN = re.compile(r"[a-z0-9_а-я]+|[㐀-龿]")


def tokenize(text: str, min_word_len: int = 1) -> List[Tuple[str, int]]:
    """-> [(token, position)], positions 1-based; overshort ASCII words keep their slot."""
    out = []
    pos = 0
    for m in _TOKEN.finditer(text.lower()):
        t = m.group(0)
        pos += 1
        if len(t) < min_word_len and t.isascii():
            continue
        out.append((t, pos))
    return out


MAGIC_SENTENCE, MAGIC_PARAGRAPH = "\x03sentence", "\x03paragraph"  # MAGIC_WORD_SENTENCE / _PARAGRAPH (sphinx.cpp:163-164)


def tokenize_sp(text: str, min_word_len: int = 1) -> List[Tuple[str, int]]:
    """index_sp = 1 (+ html_strip for '<p>'): the tokenizer of `tokenize` for ASCII text, plus the boundary tokens.  A sentence
    boundary ('?', '!', or a '.' that CSphTokenizerBase::CodepointArbitrationI, sphinx.cpp:4575-4655, does not take for an
    in-word dot, an in-phrase dot or a middle name / salutation) takes the next position and is indexed as the keyword
    MAGIC_SENTENCE there; a paragraph boundary ('<p>') as both MAGIC_SENTENCE and MAGIC_PARAGRAPH at one position
    (CSphSource_Document::BuildZoneHits, sphinx.cpp:22231-22245)."""
    out: List[Tuple[str, int]] = []
    pos, i, n = 0, 0, len(text)
    accum = ""  # the word right before the current character (m_sAccum), "" after a separator

    def word(c):
        return c.isascii() and (c.isalnum() or c == "_")

    while i < n:
        c = text[i]
        if text.startswith("<p>", i):
            pos += 1
            out += [(MAGIC_SENTENCE, pos), (MAGIC_PARAGRAPH, pos)]
            i, accum = i + 3, ""
            continue
        if word(c):
            j = i
            while j < n and word(text[j]):
                j += 1
            accum = text[i:j]
            pos += 1
            if len(accum) >= min_word_len:
                out.append((accum.lower(), pos))
            i = j
            continue
        boundary = c in "?!"
        if c == ".":
            nx, nx2, nx3 = (text[i + 1:i + 2] or "\0"), (text[i + 2:i + 3] or "\0"), (text[i + 3:i + 4] or "\0")
            inword = word(nx) or nx == "-" or ord(nx) >= 0x80 or nx == ","
            inphrase = nx in " \t\r\n" and (("a" <= nx2 <= "z") or (nx2 == "(" and "a" <= nx3 <= "z"))
            middle = False
            if len(accum) == 1:
                middle = accum.isupper()
            elif len(accum) == 2 and accum[0].isupper():
                middle = (not accum[1].isupper()) or accum in ("MR", "MS", "DR")
            elif len(accum) == 3:
                middle = accum.lower() in ("mrs", "drs")
            boundary = not (inword or inphrase or middle)
        if boundary:
            pos += 1
            out.append((MAGIC_SENTENCE, pos))
        accum = ""
        i += 1
    return out


def make_hits(docs: Sequence[Sequence[str]], min_word_len: int = 1, index_sp: bool = False):
    """docs[rowid] = [field0 text, field1 text, ...] -> (wordid, rowid, hitpos) arrays sorted
    by (wordid, rowid, hitpos), plus the vocabulary {token: term_id} (wordid = term_id + 1).
    The field-end marker goes on every hit at the field's last position (sphinx.cpp:22533-22548)."""
    vocab: Dict[str, int] = {}
    raw = []
    for rowid, fields in enumerate(docs):
        for f, text in enumerate(fields):
            toks = tokenize_sp(text, min_word_len) if index_sp else tokenize(text, min_word_len)
            last = toks[-1][1] if toks else 0
            for i, (t, pos) in enumerate(toks):
                raw.append((t, rowid, f, pos, pos == last))
    for t in sorted({r[0] for r in raw}):
        vocab[t] = len(vocab)
    hits = sorted((vocab[t] + 1, rowid, (f << 24) | pos, end) for t, rowid, f, pos, end in raw)
    wordid = np.array([h[0] for h in hits], np.uint64)
    rowid = np.array([h[1] for h in hits], np.uint32)
    hitpos = np.array([h[2] | ((1 << 23) if h[3] else 0) for h in hits], np.uint32)
    return wordid, rowid, hitpos, vocab


def mini_index(orc, docs, min_word_len: int = 1, skiplist_block_size: int = 128, inline_hits: int = 1, index_sp: bool = False):
    wordid, rowid, hitpos, vocab = make_hits(docs, min_word_len, index_sp)
    n_fields = max(len(d) for d in docs)
    idx = orc.build_index(wordid, rowid, hitpos, total_docs=len(docs), skiplist_block_size=skiplist_block_size,
                          inline_hits=inline_hits, n_fields=n_fields, n_terms=len(vocab))
    return idx, vocab


def synth_postings(rng: np.random.Generator, n_docs: int, probs: Sequence[float], n_fields: int = 2,
                   max_pos: int = 1024, title_frac: float = 0.1, end_markers: bool = False):
    """Small random corpus for parity tests: term t appears in a doc with probability probs[t];
    tf = 1 + min(254, geometric(0.5)); hits spread over fields (field 0 w.p. title_frac)."""
    W, R, H = [], [], []
    for t, p in enumerate(probs):
        mask = rng.random(n_docs) < p
        rows = np.nonzero(mask)[0].astype(np.uint32)
        for r in rows:
            tf = 1 + min(254, int(rng.geometric(0.5)) - 1)
            f = (rng.random(tf) >= title_frac).astype(np.uint32) if n_fields > 1 else np.zeros(tf, np.uint32)
            if n_fields > 2:
                f = rng.integers(0, n_fields, tf).astype(np.uint32)
            pos = rng.integers(1, max_pos + 1, tf).astype(np.uint32)
            hp = np.unique((f << 24) | pos)
            if end_markers and rng.random() < 0.2:
                hp[-1] |= 1 << 23
            W.append(np.full(hp.size, t + 1, np.uint64))
            R.append(np.full(hp.size, r, np.uint32))
            H.append(hp.astype(np.uint32))
    if not W:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32)
    return np.concatenate(W), np.concatenate(R), np.concatenate(H)


def build_blob_pool(rows_values, header: int = 8):
    """The attribute blob pool as BlobRowBuilder writes it (attribute.cpp:22-46, 495-513): per row a blob row = one byte for the
    width of the length fields (0 / 1 / 2 = 1 / 2 / 4 bytes, by the row's total data size), the blob attributes' CUMULATIVE
    lengths, then their bytes back to back.  rows_values[r] = [bytes of blob attribute 0, bytes of attribute 1, ...]
    (an MVA = its sorted values as little-endian 32- / 64-bit integers).  -> (pool uint8 array, per-row offsets)."""
    pool = bytearray(header)  # a .spb file starts with the used size
    offs = []
    for vals in rows_values:
        total = sum(len(v) for v in vals)
        kind = 0 if total < 256 else 1 if total < 65536 else 2
        sz = (1, 2, 4)[kind]
        offs.append(len(pool))
        pool.append(kind)
        acc = 0
        for v in vals:
            acc += len(v)
            pool += acc.to_bytes(sz, "little")
        for v in vals:
            pool += bytes(v)
    pool[:8] = len(pool).to_bytes(8, "little")[:header] if header else b""
    return np.frombuffer(bytes(pool), np.uint8).copy(), offs
