"""The query-text front end (mrk_query_parse, csrc/mrk_query.cpp) against what the reference's parser builds.

Two pins, no GPU:
  * the reference's own parser test (src/gtests/gtests_tokenizer.cpp:703-751, QueryParser.test_many; :753-759
    NEAR_with_NOT; :761-779 soft_whitespace1/2): query text -> sphReconstructNode() string, with the fixture's
    settings (fields title + body, min_word_len 2).  The strings are data kept in tests/golden/query_parser_vectors.json;
    the reconstruction below follows sphReconstructNode (sphinxquery.cpp:1907-1987) over our tree.
  * every golden ranking case whose name carries its query text (tests/golden/reference_vectors.json): the tree the
    parser builds from the text must equal the hand-built tree the ranking goldens were pinned with -- operators,
    operator arguments, keyword text, query positions, field masks and position modifiers.
"""
import json
import os
import re

import pytest

from manticoresearch_amd import api
from manticoresearch_amd._lib import MrkError

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_vectors.json")))
PARSER = json.load(open(os.path.join(HERE, "golden", "query_parser_vectors.json")))
ALL = 0xFFFFFFFF
OPNAME = {api.SPH_QUERY_AND: "and", api.SPH_QUERY_OR: "or", api.SPH_QUERY_MAYBE: "maybe", api.SPH_QUERY_ANDNOT: "andnot",
          api.SPH_QUERY_PHRASE: "phrase", api.SPH_QUERY_PROXIMITY: "proximity", api.SPH_QUERY_QUORUM: "quorum",
          api.SPH_QUERY_BEFORE: "before", api.SPH_QUERY_NEAR: "near", api.SPH_QUERY_NOTNEAR: "notnear", api.SPH_QUERY_SENTENCE: "sentence",
          api.SPH_QUERY_PARAGRAPH: "paragraph"}


# ---------------------------------------------------------------- sphReconstructNode over an XQNode tree
def reconstruct(n, fields):
    if n is None:
        return ""
    words_ops = {api.SPH_QUERY_PHRASE: '"%s"', api.SPH_QUERY_PROXIMITY: '"%s"~' + str(n.opt), api.SPH_QUERY_QUORUM: '"%s"/' + str(n.opt)}
    if n.word is not None or n.op in words_ops:  # a node that holds words
        ws = [n.word.text] if n.word is not None else [k.word.text for k in n.children]
        s = " ".join(ws)
        if n.word is None:
            s = words_ops[n.op] % s
        mask = n.field_mask if n.word is not None else n.children[0].field_mask
        if mask != ALL:
            s = "( @%s: %s )" % (",".join(f for i, f in enumerate(fields) if mask >> i & 1), s)
        return s
    sop = {api.SPH_QUERY_AND: " ", api.SPH_QUERY_OR: "|", api.SPH_QUERY_MAYBE: "MAYBE", api.SPH_QUERY_ANDNOT: "AND NOT",
           api.SPH_QUERY_BEFORE: "BEFORE", api.SPH_QUERY_NEAR: "NEAR"}[n.op]
    s = reconstruct(n.children[0], fields)
    for k in n.children[1:]:
        s = "%s %s %s" % (s, sop, reconstruct(k, fields))
    return "( %s )" % s if len(n.children) > 1 else s


@pytest.mark.parametrize("case", PARSER["reconstruct"], ids=lambda c: c["query"])
def test_reference_parser_vectors(case):
    try:
        tree = api.parse_query(case["query"], PARSER["fields"], PARSER["min_word_len"])
    except MrkError:
        tree = None  # the reference test reconstructs the NULL root of a failed parse as ""
    assert reconstruct(tree, PARSER["fields"]) == case["reconstruct"]


def test_not_inside_near_is_rejected():
    with pytest.raises(MrkError, match="non-computable"):
        api.parse_query("me -test NEAR/2 off", PARSER["fields"], PARSER["min_word_len"])
    for q in ("-one", "-one -two"):
        with pytest.raises(MrkError, match="non-computable"):
            api.parse_query(q, PARSER["fields"], PARSER["min_word_len"])


@pytest.mark.parametrize("q", PARSER["soft_whitespace"])
def test_soft_whitespace_keeps_positions(q):
    tree = api.parse_query(q, PARSER["fields"], PARSER["min_word_len"])
    assert [(k.word.text, k.word.atom_pos) for k in tree.children] == [("me", 1), ("off", 2)]


# ---------------------------------------------------------------- the ranking goldens' hand-built trees
FIELDS = {"test_019": ["title", "body"], "test_019_fld": ["title", "body"], "test_037": ["title", "body"], "test_080": ["first", "second"]}
NOT_TEXT = re.compile(r"^(weight_boundary|037 phrase |205 )")
SUFFIX = re.compile(r"(, field_weights [-0-9,]+| wordcount| \(bm25\)| sph04| fieldmask| bm25)$")


def query_text(name):
    if NOT_TEXT.match(name):
        return None
    t = re.sub(r"^\d+(/test2)? (fld |sphinxql )?", "", name)
    while SUFFIX.search(t):
        t = SUFFIX.sub("", t)
    return t


def as_golden(n):
    """XQNode -> the dict shape of reference_vectors.json"""
    if n.word is not None:
        d = {"word": n.word.text, "pos": n.word.atom_pos, "mask": n.field_mask}
        tp = n.term_pos()
        if tp:
            d["tp"] = {1: "start", 2: "end", 3: "startend", 4: "limit"}[tp]
            d["max_pos"] = n.field_max_pos
        return d
    return {"op": OPNAME[n.op], "kids": [as_golden(k) for k in n.children], "opt": n.opt}


def strip_op_masks(q):
    if "word" in q:
        return q
    return {"op": q["op"], "kids": [strip_op_masks(k) for k in q["kids"]], "opt": q["opt"]}


TEXT_CASES = [(c, query_text(c["name"])) for c in GOLD["cases"] if query_text(c["name"]) is not None]


def test_most_goldens_carry_their_text():
    assert len(TEXT_CASES) >= 135


@pytest.mark.parametrize("case,text", TEXT_CASES, ids=[c["name"] for c, _ in TEXT_CASES])
def test_parsed_text_equals_the_golden_tree(case, text):
    corpus = GOLD["corpora"][case["corpus"]]
    # ("transformed": the stored tree is the one the ranker sees, behind sphTransformExtendedQuery's always-on rewrites)
    tree = api.parse_query(text, FIELDS.get(case["corpus"], []), corpus["min_word_len"], transform=bool(case.get("transformed")))
    assert as_golden(tree) == strip_op_masks(case["query"])


def test_transform_near_groups_and_quorum_one():
    """mrk_parsed_transform = the always-on part of sphTransformExtendedQuery (sphinx.cpp:15345-15359): TransformNear flattens AND
    groups among a NEAR node's operands in place and in order (nested groups too; OR groups and phrases stay operands), TransformQuorum
    turns a quorum of threshold 1 into the OR of its words; everything else is left alone."""
    def shape(n):
        if n.word is not None:
            return n.word.text
        return (OPNAME[n.op], n.opt, [shape(k) for k in n.children])

    F = ["title", "body"]
    assert shape(api.parse_query("(a b c) NEAR/3 d", F, transform=True)) == ("near", 3, ["a", "b", "c", "d"])
    assert shape(api.parse_query("x NEAR/2 (a (b c)) NEAR/2 (d | e)", F, transform=True)) == ("near", 2, ["x", "a", "b", "c", ("or", 0, ["d", "e"])])
    assert shape(api.parse_query('(a b) NEAR/4 "c d"', F, transform=True)) == ("near", 4, ["a", "b", ("phrase", 0, ["c", "d"])])
    assert shape(api.parse_query("(a b c) NEAR/3 d", F)) == ("near", 3, [("and", 0, ["a", "b", "c"]), "d"])  # the parser's own output keeps the group
    assert shape(api.parse_query('"a b c"/1', F, transform=True)) == ("or", 0, ["a", "b", "c"])
    assert shape(api.parse_query('"a b c"/2', F, transform=True)) == ("quorum", 2, ["a", "b", "c"])
    assert shape(api.parse_query("a (b c) | d", F, transform=True)) == shape(api.parse_query("a (b c) | d", F))


# ---------------------------------------------------------------- syntax the goldens do not reach
def test_field_limits_modifiers_and_errors():
    F = ["title", "body", "tags"]
    t = api.parse_query("@(title,tags) hello @!body world @* again", F)
    assert [(k.word.text, k.field_mask) for k in t.children] == [("hello", 0b101), ("world", ~0b010 & ALL), ("again", ALL)]
    t = api.parse_query("^hello$ =world^2.5 tail$^0.5", F)
    assert [(k.word.text, k.term_pos(), k.word.boost) for k in t.children] == [("hello", 3, 1.0), ("=world", 0, 2.5), ("tail", 2, 0.5)]
    t = api.parse_query("@body[5] hello", F)
    assert (t.field_mask, t.term_pos(), t.field_max_pos) == (0b010, 4, 5)
    t = api.parse_query("a MAYBE b MAYBE c", F)
    assert (t.op, len(t.children)) == (api.SPH_QUERY_MAYBE, 3)
    t = api.parse_query("a NEAR/2 b NEAR/3 c", F)  # a different distance starts a new node (AddOp compares the argument)
    assert (t.op, t.opt, t.children[0].op, t.children[0].opt) == (api.SPH_QUERY_NEAR, 3, api.SPH_QUERY_NEAR, 2)
    t = api.parse_query("a NOTNEAR/1 b NOTNEAR/2 c", F)  # %left
    assert (t.opt, t.children[0].opt, t.children[1].word.text) == (2, 1, "c")
    t = api.parse_query('"a * * b * c"', F)  # PhraseShiftQpos
    assert [k.word.atom_pos for k in t.children] == [1, 4, 6]
    t = api.parse_query('"one"~3 two', F)  # FixupDegenerates
    assert [k.word.text for k in t.children] == ["one", "two"]
    assert api.parse_query("", F) is None and api.parse_query('""', F) is None
    for bad, why in (("@nosuch a", "no field"), ("(a b", "parenthesis"), ('"a b', "unterminated"), ('"a b"/0', "quorum threshold"),
                     ('"a b"/1.5', "out of bounds"), ('"a b"~0', "proximity threshold"), ("@(title,body a", "field")):
        with pytest.raises(MrkError, match=why):
            api.parse_query(bad, F)
    # a callable resolves keywords; unknown words stay < 0
    t = api.parse_query("alpha beta", F, lookup=lambda w: {"alpha": 7}.get(w, -1))
    assert [k.word.term_id for k in t.children] == [7, -1]


def test_parser_survives_junk():
    """Arbitrary text either parses or fails with MRK_E_INVAL and a message -- never a crash, never a hang."""
    import random
    rnd = random.Random(1234)
    alphabet = ["a", "bb", "ccc", " ", " ", "(", ")", "|", "-", "!", '"', "~", "/", "1", "0.5", "<<", "NEAR/2", "NOTNEAR/3", "NEAR/", "MAYBE", "SENTENCE", "PARAGRAPH",
                "@title", "@(title,body)", "@!", "@*", "@body[3]", "@nosuch", "@@relaxed", "^", "$", "=", "*", "^1.5", "\\", "\xc3\xa9", "\x03", "["]
    n_ok = 0
    for _ in range(3000):
        q = "".join(rnd.choice(alphabet) for _ in range(rnd.randint(1, 24)))
        try:
            api.parse_query(q, ["title", "body"], rnd.choice([1, 2]))
            n_ok += 1
        except MrkError as e:
            assert str(e)
    assert n_ok > 100
    with pytest.raises(MrkError, match="too deep"):
        api.parse_query("(" * 200 + "a" + ")" * 200, [])
    assert api.parse_query("(" * 60 + "a" + ")" * 60, []).word.text == "a"


def test_parser_under_sanitizers(tmp_path):
    """The parser alone (it is plain host C++), built with -fsanitize=address,undefined, over 20 000 junk queries."""
    import random
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "fuzz_parser")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(root, "manticoresearch_amd", "csrc"), os.path.join(HERE, "cpp", "fuzz_parser.cpp"),
                           os.path.join(root, "manticoresearch_amd", "csrc", "mrk_query.cpp"), "-o", exe])
    rnd = random.Random(99)
    alphabet = ["a", "bb", "ccc", " ", " ", "(", ")", "|", "-", "!", '"', "~", "/", "1", "0.5", "<<", "NEAR/2", "NOTNEAR/3", "NEAR/", "MAYBE", "SENTENCE",
                "PARAGRAPH", "@title", "@(title,body)", "@!", "@*", "@body[3]", "@nosuch", "@@relaxed", "^", "$", "=", "*", "^1.5", "\\", "\xc3\xa9", "\x03",
                "[", "99999999999999999999", "~99999999999", "/0", "@(", "@(title,", "@title[", "NEAR", "NOTNEAR/", "SENTENC", "@", "@@", "@@relaxe"]
    # (short queries too: what a query ENDS with matters -- the harness hands the parser an allocation of the text's exact size, so a
    # lexer that steps over the terminator, as '... ^' once made it do, lands in ASan's red zone)
    text = "\n".join("".join(rnd.choice(alphabet) for _ in range(rnd.randint(1, 40 if i & 1 else 6))) for i in range(20000)) + "\n"
    out = subprocess.run([exe], input=text.encode("utf-8"), capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-2000:]
    ok, bad = (int(x) for x in out.stdout.split()[1::2])
    assert ok + bad == 20000 and ok > 200
