"""Writes tests/golden/reference_vectors.json.

The values below were copied by hand, as data, from the reference's own test directories and docs while
reading them (nothing of the reference is imported or executed): corpora = the <db_insert> rows of
test/test_0NN/test.xml, expected docid:weight lists = the rows of the tests' model.bin (PHP-serialised text),
plus the byte examples of doc/internals-index-format.txt and the weight asserted by gtest RTN.WeightBoundary.
Run:  python tests/golden/make_vectors.py
"""
import json
import os

ALL = 0xFFFFFFFF


def T(word, pos, mask=ALL, tp=None, max_pos=0):
    """keyword leaf; tp = position modifier: "start" ('^word'), "end" ('word$'), "startend", "limit" ('@field[N] word')"""
    d = {"word": word, "pos": pos, "mask": mask}
    if tp:
        d["tp"] = tp
        d["max_pos"] = max_pos
    return d


def OP(op, *kids, mask=ALL, opt=0):
    return {"op": op, "kids": list(kids), "mask": mask, "opt": opt}


G = {
    "_about": "Golden vectors of the reference for the match -> rank -> top-K path; data only, see README.md",
    "vlb_bytes": {"source": "doc/internals-index-format.txt:44-60",
                  "cases": [[0x12345, [0x84, 0xC6, 0x45]], [0, [0]], [127, [0x7F]], [128, [0x81, 0]],
                            [0xFFFFFFFF, [0x8F, 0xFF, 0xFF, 0xFF, 0x7F]]]},
    "hitlist_bytes": {"source": "doc/internals-index-format.txt (hitlist example)", "hits": [2, 16777224, 16777229],
                      "spp": [1, 2, 0x88, 0x80, 0x80, 6, 5, 0]},
    "corpora": {
        "weight_boundary": {"source": "src/gtests/gtests_rtstuff.cpp:257-335 (RTN.WeightBoundary)", "min_word_len": 1,
                            "ids": [1], "docs": [["If I were a cat...", "We are the greatest cat"]]},
        "test_037": {"source": "test/test_037/test.xml + model.bin", "min_word_len": 1, "ids": list(range(1, 11)),
                     "docs": [["зимние шины диски чего то тут зимние шины", ""],
                              ["test doc two", "second stupid test document with random content"]] + [["filler", "filler"]] * 8},
        "test_019": {"source": "test/test_019/test.xml + model.bin", "min_word_len": 2,
                     "ids": [111, 222, 333, 444, 555, 666, 777, 888, 999, 901, 902, 903, 910],
                     "docs": [["", "basic query"], ["", "phrase query on steroids"],
                              ["sample program", 'this is a test program that prints out "hello world" to the console'],
                              ["", "china 吐我"], ["sample program two", "something written in basic | canon ef 16-35 lens"],
                              ["sample program three", "something written in perl"], ["", "77 lies multiplied by 77"],
                              ["", "agent 0077"], ["", "1234567812345678"], ["aaa", "aaa"], ["aaa", ""], ["", "aaa"],
                              ["", "wordbefore\u0000\u0000wordafter"]]},
        "test_037_test2": {"source": "test/test_037/test.xml (index test2) + model.bin", "min_word_len": 1,
                           "ids": [11, 12, 13, 14, 15, 16],
                           "docs": [["market street", ""], ["market street west", ""], ["north market street", ""],
                                    ["farmers market street north", ""], ["flower street market", ""],
                                    ["market street is so very market street", ""]]},
        # test_114 builds its rows in a loop: 510 identical rows, then one long row (520 repetitions)
        "test_114": {"source": "test/test_114/test.xml (custom_insert) + model.bin", "min_word_len": 1, "ids": list(range(1, 512)),
                     "docs_spec": [{"count": 510, "fields": ["aaaa bbbb cccc dddd"]},
                                   {"count": 1, "fields": ["aaaa bbbb x aaaa bbbb " + " x cccc dddd" * 520]}]},
        # test_116 (bound cases of the proximity node) also builds its rows in loops
        "test_116": {"source": "test/test_116/test.xml (custom_insert) + model.bin", "min_word_len": 1, "ids": list(range(1, 523)),
                     "docs_spec": [{"count": 1, "fields": ["a " + "x " * i + " b"]} for i in range(10)] +
                                  [{"count": 510, "fields": ["e x f"]},
                                   {"count": 1, "fields": ["e x f x x x e x f x x e x f x x"]},
                                   {"count": 1, "fields": [" y y i x j" * 532]}]},
        "test_052": {"source": "test/test_052/test.xml (index test: rows 1..5) + model.bin", "min_word_len": 1, "ids": [1, 2, 3, 4, 5],
                     "docs": [["aaa bbb", "ccc ddd"], ["xxx", "ccc ddd eee fff ggg"], ["yyy", "one one one two three"],
                              ["zzz", "one two three one three one two four one two three four"], ["", "a b c d e f g"]]},
        "test_054": {"source": "test/test_054/test.xml (index test: rows 1, 2) + model.bin", "min_word_len": 1, "ids": [1, 2],
                     "docs": [["hello world"], ["one two three four five"]]},
        # test_205 (idf=plain, local_df): four plain indexes over one table, searched one by one / as "l1,l2" with global statistics
        "test_205_i1": {"source": "test/test_205/test.xml (gid=1) + model.bin", "min_word_len": 1, "ids": [1, 2, 3],
                        "docs": [["da one"], ["da two"], ["da three"]]},
        "test_205_i2": {"source": "test/test_205/test.xml (gid=2) + model.bin", "min_word_len": 1, "ids": [11, 12, 13, 14, 15],
                        "docs": [["da blow"], ["da pills"], ["da yak"], ["da herb"], ["da blow"]]},
        "test_205_l1": {"source": "test/test_205/test.xml (gid=3) + model.bin", "min_word_len": 1, "ids": [100, 101, 102, 103, 104],
                        "docs": [["da blow"], ["da win"], ["da yak"], ["da herb"], ["da blow"]]},
        "test_205_l2": {"source": "test/test_205/test.xml (gid=4) + model.bin", "min_word_len": 1, "ids": [200, 201, 202, 203, 204],
                        "docs": [["da white"], ["da win"], ["da win"], ["da win"], ["da blow"]]},
        "test_019_fld": {"source": "test/test_019/test.xml (index fld: rows 1000..1010) + model.bin", "min_word_len": 1,
                         "ids": list(range(1000, 1011)),
                         "docs": [["spec1", "dummy1"], ["spec1 dummy1", ""], ["", "spec1 dummy1"], ["spec2 dummy2 text2", ""], ["spec2", "dummy2 text2"],
                                  ["spec3", "dummy3 text3"], ["spec3 dummy3 text3", "spec3"], ["spec4 dummy4", "text4"], ["spec4", "dummy4 text4"],
                                  ["spec5 of my", "dummy5"], ["spec5", "of my text5"]]},
        "test_157": {"source": "test/test_157/test.xml (RT inserts) + model.bin", "min_word_len": 1, "ids": [1, 2, 3],
                     "docs": [["this is cool place"], ["cool place is like no other"], ["place is cool becouse there is no things like this"]]},
        # test_055 (position anchors): rows 10..19 are doubled six times by INSERT .. SELECT document_id+N
        "test_055": {"source": "test/test_055/test.xml + model.bin", "min_word_len": 1,
                     "ids": [1, 2, 3, 4, 9] + list(range(10, 650)) + [2000, 1000, 1001],
                     "docs_spec": [{"count": 1, "fields": ["", t]} for t in
                                   ("one", "one and two", "one but not the other one", "two and one", "other three")] +
                                  [{"count": 640, "fields": ["", "three"]},
                                   {"count": 1, "fields": ["badger " * 600, "badger badger mushroom"]},
                                   {"count": 1, "fields": ["", "other"]}, {"count": 1, "fields": ["", "other three blind mice"]}]},
        # test_080: index "main" after "indexer --merge main delta"; the delta row is built in a loop:
        # 299992 x "C", then "B A A A" (a field given as {"runs": [[text, count], ...]} is the texts repeated and joined)
        "test_080": {"source": "test/test_080/test.xml (custom_insert) + model.bin", "min_word_len": 1, "ids": [2, 1],
                     "docs_spec": [{"count": 1, "fields": ["X", "Y"]},
                                   {"count": 1, "fields": ["", {"runs": [["C ", 299992], ["B A A A", 1]]}]}]},
        # test_115 (NEAR syntax), index idx: rows 1..17, 20..22 of one text field; row 21 is built by CONCAT / REPEAT.
        # blend_chars = '-' there: 'aleph-bet-gimel' is indexed as its parts at positions 1, 2, 3 (plus the blended token, which
        # the queries below do not ask for) -- the same positions this fixture's tokenizer gives the parts
        "test_115": {"source": "test/test_115/test.xml (index idx) + model.bin", "min_word_len": 1,
                     "ids": list(range(1, 18)) + [20, 21, 22],
                     "docs_spec": [{"count": 1, "fields": [t]} for t in
                                   ("a b c d", "a x b c d", "a x x b c d", "a b x c d", "a b x x c d", "a b x x x c d", "a b x x x x c d",
                                    "a x b x x x x c x d", "a x x b x x x c x x d", "c d x x x x a b", "c d x x x a b", "c d x x a b",
                                    "c d x a b", "c d a b", '... is the clearinghouse associated with such exchange. In general, clearinghouses are backed by the corporate members of the clearinghouse who are required to share any financial burden resulting from the non-performance by one of their members and, as such, should significantly reduce this credit risk. In cases where the clearinghouse... is the clearinghouse associated with such exchange. In general, clearinghouses are backed by the corporate members of the clearinghouse who are required to share any financial burden resulting from the non-performance by one of their members and, as such, should significantly reduce this credit risk. In cases where the clearinghouse... be able to meet its obligations to a Trading Company. The counterparty for futures contracts traded in the United States and on most foreign exchanges is the clearinghouse associated with such exchange. In general, clearinghouses are backed by the corporate members of the clearinghouse who are required to share any financial...',
                                    "one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteena",
                                    "aleph-bet-gimel dalet he wav zajin het", "ein oy vey")] +
                                  [{"count": 1, "fields": [{"runs": [["zwei ", 1], ["oy vey ho ho ho ", 1024]]}]},
                                   {"count": 1, "fields": ["oy vey drei"]}]},
        "test_349": {"source": "test/test_349/test.xml (index idx: the rows of test_115 with another row 21) + model.bin", "min_word_len": 1,
                     "ids": list(range(1, 18)) + [20, 21, 22], "like": "test_115",
                     "row_21": {"runs": [["zwei ", 1], ["oy vey ho ho ho ", 1023], ["oy vey ho h", 1]]}},
        "test_322": {"source": "test/test_322/test.xml + model.bin", "min_word_len": 1, "ids": [1, 2, 3, 100],
                     "docs": [["|sample program", "|program flow direct", "|sample program flow"],
                              ["|one sample program", "|program rev flow", "|one rev flow"],
                              ["|sample two program", "|sub program flow", "|two sub program"], ["unsigned", "", ""]]},
    },
    "cases": [
        {"name": "weight_boundary", "corpus": "weight_boundary", "query": T("cat", 1, 0b01), "ranker": "proximity_bm25",
         "expect": [[1, 1500]]},
        {"name": "037 phrase proximity_bm25", "corpus": "test_037", "query": OP("phrase", T("зимние", 1), T("шины", 2)),
         "ranker": "proximity_bm25", "expect": [[1, 2800]], "total_found": 1},
        {"name": "037 phrase bm25", "corpus": "test_037", "query": OP("phrase", T("зимние", 1), T("шины", 2)), "ranker": "bm25",
         "expect": [[1, 1800]], "total_found": 1},
        {"name": "037 phrase none", "corpus": "test_037", "query": OP("phrase", T("зимние", 1), T("шины", 2)), "ranker": "none",
         "expect": [[1, 1]], "total_found": 1},
        {"name": "037 @title test bm25", "corpus": "test_037", "query": T("test", 1, 0b01), "ranker": "bm25", "expect": [[2, 1800]]},
        {"name": "019 basic query", "corpus": "test_019", "query": OP("and", T("basic", 1), T("query", 2)),
         "ranker": "proximity_bm25", "expect": [[111, 2654]]},
        {"name": "019 \"phrase query\"", "corpus": "test_019", "query": OP("phrase", T("phrase", 1), T("query", 2)),
         "ranker": "proximity_bm25", "expect": [[222, 2687]]},
        {"name": "019 @title sample @body world", "corpus": "test_019",
         "query": OP("and", T("sample", 1, 0b01), T("world", 2, 0b10)), "ranker": "proximity_bm25", "expect": [[333, 2666]]},
        {"name": "019 basic | china", "corpus": "test_019", "query": OP("or", T("basic", 1), T("china", 2)),
         "ranker": "proximity_bm25", "expect": [[444, 1610], [111, 1577], [555, 1577]]},
        {"name": "019 \"test program\" | basic", "corpus": "test_019",
         "query": OP("or", OP("phrase", T("test", 1), T("program", 2)), T("basic", 3)), "ranker": "proximity_bm25",
         "expect": [[333, 2624], [111, 1551], [555, 1551]]},
        {"name": "019 @title sample @body -basic", "corpus": "test_019",
         "query": OP("andnot", T("sample", 1, 0b01), T("basic", 2, 0b10)), "ranker": "proximity_bm25",
         "expect": [[333, 1555], [666, 1555]]},
        {"name": "019 77", "corpus": "test_019", "query": T("77", 1), "ranker": "proximity_bm25", "expect": [[777, 1803]]},
    ],
}
for spam, exp in [(1, [[1, 7415], [3, 6426], [2, 4421]]), (10, [[1, 25415], [3, 15426], [2, 13421]]),
                  (0, [[3, 5426], [1, 5415], [2, 3421]]), (-2, [[3, 3426], [2, 1421], [1, 1415]]),
                  (-10, [[3, -4574], [2, -6579], [1, -14585]])]:
    G["cases"].append({"name": f"322 program flow, field_weights 1,2,{spam}", "corpus": "test_322",
                       "query": OP("and", T("program", 1), T("flow", 2)), "ranker": "proximity_bm25",
                       "field_weights": [1, 2, spam], "expect": exp})

G["cases"] += [
    {"name": "019 \"quorum query test\"/1", "corpus": "test_019",
     "query": OP("quorum", T("quorum", 1), T("query", 2), T("test", 3), opt=1), "ranker": "proximity_bm25",
     "expect": [[333, 1573], [111, 1551], [222, 1551]]},
    {"name": "019 \"hello program\"~4", "corpus": "test_019", "query": OP("proximity", T("hello", 1), T("program", 2), opt=4),
     "ranker": "proximity_bm25", "expect": [[333, 1687]]},
    {"name": "037 phrase wordcount", "corpus": "test_037", "query": OP("phrase", T("зимние", 1), T("шины", 2)),
     "ranker": "wordcount", "expect": [[1, 2]], "total_found": 1},
    {"name": "037/test2 market street sph04", "corpus": "test_037_test2", "query": OP("and", T("market", 1), T("street", 2)),
     "ranker": "sph04", "expect": [[11, 11290], [12, 10290], [16, 10212], [13, 8290], [14, 8290], [15, 4290]]},
    {"name": "114 \"aaaa bbbb\" wordcount", "corpus": "test_114", "query": OP("phrase", T("aaaa", 1), T("bbbb", 2)),
     "ranker": "wordcount", "expect": [[511, 2]] + [[i, 1] for i in range(1, 20)], "limit": 20, "total_found": 511},
    {"name": "114 \"cccc dddd\" wordcount", "corpus": "test_114", "query": OP("phrase", T("cccc", 1), T("dddd", 2)),
     "ranker": "wordcount", "expect": [[511, 520]] + [[i, 1] for i in range(1, 20)], "limit": 20, "total_found": 511},
]
G["cases"] += [
    {"name": "116 \"a b\"~3 wordcount", "corpus": "test_116", "query": OP("proximity", T("a", 1), T("b", 2), opt=3),
     "ranker": "wordcount", "expect": [[1, 1], [2, 1], [3, 1]], "total_found": 3},
    {"name": "116 \"e f\"~2 wordcount", "corpus": "test_116", "query": OP("proximity", T("e", 1), T("f", 2), opt=2),
     "ranker": "wordcount", "expect": [[521, 3]] + [[i, 1] for i in range(11, 30)], "limit": 20, "total_found": 511},
    {"name": "116 \"i j\"~2 wordcount", "corpus": "test_116", "query": OP("proximity", T("i", 1), T("j", 2), opt=2),
     "ranker": "wordcount", "expect": [[522, 532]], "total_found": 1},
]
G["cases"] += [  # test_052: the BEFORE operator '<<' (ExtOrder_c, searchnode.cpp:4657-4936); keywords count 1, 2, 3.. in query text order
    {"name": "052 " + n, "corpus": "test_052", "query": q, "ranker": "proximity_bm25", "expect": e, "total_found": len(e)}
    for n, q, e in [
        ("aaa << ccc", OP("before", T("aaa", 1), T("ccc", 2)), []),
        ("aaa << bbb << ccc", OP("before", T("aaa", 1), T("bbb", 2), T("ccc", 3)), []),
        ("aaa << ccc << ddd", OP("before", T("aaa", 1), T("ccc", 2), T("ddd", 3)), []),
        ("ccc << ddd", OP("before", T("ccc", 1), T("ddd", 2)), [[1, 2543], [2, 2543]]),
        ("ccc << eee << fff", OP("before", T("ccc", 1), T("eee", 2), T("fff", 3)), [[2, 2529]]),
        ("ccc << ddd << ggg", OP("before", T("ccc", 1), T("ddd", 2), T("ggg", 3)), [[2, 2529]]),
        ("ccc << ddd << xxx", OP("before", T("ccc", 1), T("ddd", 2), T("xxx", 3)), []),
        ("eee << ddd << ggg", OP("before", T("eee", 1), T("ddd", 2), T("ggg", 3)), []),
        ("one << two << three", OP("before", T("one", 1), T("two", 2), T("three", 3)), [[4, 3549], [3, 2546]]),
        ("one << three", OP("before", T("one", 1), T("three", 2)), [[4, 2574], [3, 1569]]),
        ("one << one << three", OP("before", T("one", 1), T("one", 2), T("three", 3)), [[4, 2574], [3, 2569]]),
        ("one << one << one << three", OP("before", T("one", 1), T("one", 2), T("one", 3), T("three", 4)), [[3, 3569], [4, 1574]]),
        ("one << one << one << one << three", OP("before", T("one", 1), T("one", 2), T("one", 3), T("one", 4), T("three", 5)), [[4, 1574]]),
        ("one << two << three << four", OP("before", T("one", 1), T("two", 2), T("three", 3), T("four", 4)), [[4, 4537]]),
        ("\"a b c\" << b << c << d", OP("before", OP("phrase", T("a", 1), T("b", 2), T("c", 3)), T("b", 4), T("c", 5), T("d", 6)), []),
        ("\"a b c\" << c << d << e", OP("before", OP("phrase", T("a", 1), T("b", 2), T("c", 3)), T("c", 4), T("d", 5), T("e", 6)), []),
        ("\"a b c\" << e << f << g", OP("before", OP("phrase", T("a", 1), T("b", 2), T("c", 3)), T("e", 4), T("f", 5), T("g", 6)), [[5, 3602]]),
        ("a << \"b c d\" << e", OP("before", T("a", 1), OP("phrase", T("b", 2), T("c", 3), T("d", 4)), T("e", 5)), [[5, 4540]]),
        ("\"a b c d\" << \"d e f\"", OP("before", OP("phrase", T("a", 1), T("b", 2), T("c", 3), T("d", 4)),
                                          OP("phrase", T("d", 5), T("e", 6), T("f", 7))), []),
        ("\"a b c d\" << \"e f g\"", OP("before", OP("phrase", T("a", 1), T("b", 2), T("c", 3), T("d", 4)),
                                          OP("phrase", T("e", 5), T("f", 6), T("g", 7))), [[5, 4616]]),
        ("(ccc | \"ddd eee\") << (ddd | ggg)", OP("before", OP("or", T("ccc", 1), OP("phrase", T("ddd", 2), T("eee", 3))),
                                                  OP("or", T("ddd", 4), T("ggg", 5))), [[2, 1594], [1, 1521]]),
        ("ccc << ddd$", OP("before", T("ccc", 1), T("ddd", 2, tp="end")), [[1, 2543]]),
        ("^one << two << three$", OP("before", T("one", 1, tp="start"), T("two", 2), T("three", 3, tp="end")), [[3, 2546]]),
        ("^one << \"one one\" << two << three$", OP("before", T("one", 1, tp="start"), OP("phrase", T("one", 2), T("one", 3)), T("two", 4),
                                                    T("three", 5, tp="end")), [[3, 5546]]),
        # the threshold "1" is itself a keyword-sized token here (min_word_len = 1) and takes query position 3
        # (XQParser_t::ParseNumeric, sphinxquery.cpp:1160-1170): the next keyword stands at 4
        ("\"zzz aaa\"/1 << bbb", OP("before", OP("quorum", T("zzz", 1), T("aaa", 2), opt=1), T("bbb", 4)), [[1, 1568]]),
        ("\"zzz aaa\"/1 << ddd", OP("before", OP("quorum", T("zzz", 1), T("aaa", 2), opt=1), T("ddd", 4)), []),
    ]
]
TI, BO = 0b01, 0b10  # field masks of index fld: title, body
G["cases"] += [  # test_019, index fld: the FIELDMASK ranker (weights = matched-fields masks) and field limits inside brackets
    # (model.bin lists the rows only for those: "expect_ids")
    {"name": "019 fld spec1 | dummy1 fieldmask", "corpus": "test_019_fld", "query": OP("or", T("spec1", 1), T("dummy1", 2)), "ranker": "fieldmask",
     "expect_ids": [1000, 1001, 1002], "expect_weights": {"1000": 3, "1001": 1, "1002": 2}, "total_found": 3},
    {"name": "019 fld @title spec1 @body dummy1", "corpus": "test_019_fld", "query": OP("and", T("spec1", 1, TI), T("dummy1", 2, BO)),
     "ranker": "proximity_bm25", "expect_ids": [1000], "total_found": 1},
    {"name": "019 fld @title ( ( spec2 ) (dummy2) | text2 )", "corpus": "test_019_fld",
     "query": OP("and", T("spec2", 1, TI), OP("or", T("dummy2", 2, TI), T("text2", 3, TI))), "ranker": "proximity_bm25", "expect_ids": [1003], "total_found": 1},
    {"name": "019 fld @title ( ( spec3 ) @body (dummy3) | text3 )", "corpus": "test_019_fld",
     "query": OP("and", T("spec3", 1, TI), OP("or", T("dummy3", 2, BO), T("text3", 3, BO))), "ranker": "proximity_bm25", "expect_ids": [1005], "total_found": 1},
    {"name": "019 fld @title ( ( spec4 ) (dummy4) | @body text4 )", "corpus": "test_019_fld",
     "query": OP("and", T("spec4", 1, TI), OP("or", T("dummy4", 2, TI), T("text4", 3, BO))), "ranker": "proximity_bm25", "expect_ids": [1007, 1008],
     "total_found": 2},
    {"name": "019 fld @title ( ( spec4 ) (dummy4) @body text4 )", "corpus": "test_019_fld",
     "query": OP("and", T("spec4", 1, TI), T("dummy4", 2, TI), T("text4", 3, BO)), "ranker": "proximity_bm25", "expect_ids": [1007], "total_found": 1},
]
DA_WIN = OP("and", T("da", 1), T("win", 2))
LDF = {"total_docs": 10, "local_docs": {"da": 10, "win": 4, "blow": 3}}  # local_df=1 over l1 + l2 (SetupLocalDF, searchd.cpp:5869-5990)
G["cases"] += [  # test_205: each index ranks with its own statistics unless local_df hands it the global ones; idf=plain
    {"name": "205 i1 da", "corpus": "test_205_i1", "query": T("da", 1), "ranker": "proximity_bm25", "expect": [[1, 1319], [2, 1319], [3, 1319]]},
    {"name": "205 i2 da", "corpus": "test_205_i2", "query": T("da", 1), "ranker": "proximity_bm25", "expect": [[i, 1295] for i in range(11, 16)]},
    {"name": "205 i1 da idf=plain", "corpus": "test_205_i1", "query": T("da", 1), "ranker": "proximity_bm25", "plain_idf": True,
     "expect": [[1, 1500], [2, 1500], [3, 1500]]},
    {"name": "205 i2 da idf=plain", "corpus": "test_205_i2", "query": T("da", 1), "ranker": "proximity_bm25", "plain_idf": True,
     "expect": [[i, 1500] for i in range(11, 16)]},
    {"name": "205 l1 da win", "corpus": "test_205_l1", "query": DA_WIN, "ranker": "proximity_bm25", "expect": [[101, 2500]]},
    {"name": "205 l2 da win", "corpus": "test_205_l2", "query": DA_WIN, "ranker": "proximity_bm25", "expect": [[201, 2397], [202, 2397], [203, 2397]]},
    dict({"name": "205 l1 da win local_df", "corpus": "test_205_l1", "query": DA_WIN, "ranker": "proximity_bm25", "expect": [[101, 2417]]}, **LDF),
    dict({"name": "205 l2 da win local_df", "corpus": "test_205_l2", "query": DA_WIN, "ranker": "proximity_bm25",
          "expect": [[201, 2417], [202, 2417], [203, 2417]]}, **LDF),
    {"name": "205 l1 blow", "corpus": "test_205_l1", "query": T("blow", 1), "ranker": "proximity_bm25", "expect": [[100, 1587], [104, 1587]]},
    {"name": "205 l2 blow", "corpus": "test_205_l2", "query": T("blow", 1), "ranker": "proximity_bm25", "expect": [[204, 1704]]},
    dict({"name": "205 l1 blow local_df", "corpus": "test_205_l1", "query": T("blow", 1), "ranker": "proximity_bm25",
          "expect": [[100, 1592], [104, 1592]]}, **LDF),
    dict({"name": "205 l2 blow local_df", "corpus": "test_205_l2", "query": T("blow", 1), "ranker": "proximity_bm25", "expect": [[204, 1592]]}, **LDF),
]
SIX = [T(w, i + 1) for i, w in enumerate(("five", "tree", "oak", "one", "two", "hive"))]
G["cases"] += [  # test_054 (quorum): the queries without repeated or wildcard words.  Percent thresholds are resolved by the
    # parser: '/0.4' of 6 words = floor(0.4 * 6 + 0.5) = 2; '/0.59' is stored as 58 % (float to int) = floor(3.48 + 0.5) = 3;
    # '/0.60' = floor(3.6 + 0.5) = 4 (ExtQuorum_c::GetThreshold, searchnode.cpp:4598-4601)
    {"name": "054 \"hello heaven\"/1", "corpus": "test_054", "query": OP("quorum", T("hello", 1), T("heaven", 2), opt=1),
     "ranker": "proximity_bm25", "expect": [[1, 1571]], "total_found": 1},
    {"name": "054 \"hello from above\"/2", "corpus": "test_054", "query": OP("quorum", T("hello", 1), T("from", 2), T("above", 3), opt=2),
     "ranker": "proximity_bm25", "expect": [], "total_found": 0},
    {"name": "054 \"one two foo bar\"/3", "corpus": "test_054",
     "query": OP("quorum", T("one", 1), T("two", 2), T("foo", 3), T("bar", 4), opt=3), "ranker": "proximity_bm25", "expect": [], "total_found": 0},
    {"name": "054 \"five tree oak one two hive\"/0.4", "corpus": "test_054", "query": OP("quorum", *SIX, opt=2),
     "ranker": "proximity_bm25", "expect": [[2, 2571]], "total_found": 1},
    # the same node under BM25, for the device (its hit rankers take <= 4 keywords): 1000 x the field weight + the 571 of
    # the reference's 2571 -- the tfidf sum over six query words, three of them not in the dictionary
    {"name": "054 \"five tree oak one two hive\"/0.4 (bm25)", "corpus": "test_054", "query": OP("quorum", *SIX, opt=2),
     "ranker": "bm25", "expect": [[2, 1571]], "total_found": 1},
    {"name": "054 \"five tree oak one two hive\"/0.59", "corpus": "test_054", "query": OP("quorum", *SIX, opt=3),
     "ranker": "proximity_bm25", "expect": [[2, 2571]], "total_found": 1},
    {"name": "054 \"five tree oak one two hive\"/0.60", "corpus": "test_054", "query": OP("quorum", *SIX, opt=4),
     "ranker": "proximity_bm25", "expect": [], "total_found": 0},
]
G["cases"] += [  # test_157: a real ExtQuorum_c (5 words, threshold 3).  model.bin lists the matching rows only ("expect_ids");
    # the second spelling of each case asks for BM25 -- the matching set does not depend on the ranker -- so that the device,
    # whose hit rankers take <= 4 keywords, runs the node too
    {"name": "157 \"is cool place\"/3", "corpus": "test_157", "query": OP("quorum", T("is", 1), T("cool", 2), T("place", 3), opt=3),
     "ranker": "proximity_bm25", "expect_ids": [1, 2, 3], "total_found": 3},
    {"name": "157 \"there things is cool place\"/3", "corpus": "test_157",
     "query": OP("quorum", T("there", 1), T("things", 2), T("is", 3), T("cool", 4), T("place", 5), opt=3),
     "ranker": "proximity_bm25", "expect_ids": [1, 2, 3], "total_found": 3},
    {"name": "157 \"there things is cool place\"/3 (bm25)", "corpus": "test_157",
     "query": OP("quorum", T("there", 1), T("things", 2), T("is", 3), T("cool", 4), T("place", 5), opt=3),
     "ranker": "bm25", "expect_ids": [1, 2, 3], "total_found": 3},
]
G["cases"] += [  # test_055: '^' and '$' (ExtTermPos_T, searchnode.cpp:2259-2405)
    {"name": "055 ^one two", "corpus": "test_055", "query": OP("and", T("one", 1, tp="start"), T("two", 2)),
     "ranker": "proximity_bm25", "expect": [[2, 1690]], "total_found": 1},
    {"name": "055 ^other three", "corpus": "test_055", "query": OP("and", T("other", 1, tp="start"), T("three", 2)),
     "ranker": "proximity_bm25", "expect": [[9, 2509], [1001, 2509]], "total_found": 2},
    {"name": "055 three$", "corpus": "test_055", "query": T("three", 1, tp="end"), "ranker": "proximity_bm25",
     "expect": [[i, 1341] for i in range(9, 29)], "limit": 20, "total_found": 641},
    {"name": "055 ^badger", "corpus": "test_055", "query": T("badger", 1, tp="start"), "ranker": "proximity_bm25",
     "expect": [[2000, 2998]], "total_found": 1},
]
G["cases"] += [  # test_080: '@field[N]' position limits at the far end of a 299996-word field, '$' on a repeated keyword
    {"name": "080 C", "corpus": "test_080", "query": T("c", 1), "ranker": "proximity_bm25", "expect": [[1, 1815]], "total_found": 1},
    {"name": "080 @second[299992] B", "corpus": "test_080", "query": T("b", 1, 0b10, tp="limit", max_pos=299992),
     "ranker": "proximity_bm25", "expect": [], "total_found": 0},
    {"name": "080 @second[299993] B", "corpus": "test_080", "query": T("b", 1, 0b10, tp="limit", max_pos=299993),
     "ranker": "proximity_bm25", "expect": [[1, 1643]], "total_found": 1},
    {"name": "080 @second[299994] B", "corpus": "test_080", "query": T("b", 1, 0b10, tp="limit", max_pos=299994),
     "ranker": "proximity_bm25", "expect": [[1, 1643]], "total_found": 1},
    {"name": "080 \"C B A A A\"", "corpus": "test_080", "query": OP("phrase", T("c", 1), T("b", 2), T("a", 3), T("a", 4), T("a", 5)),
     "ranker": "proximity_bm25", "expect": [[1, 5728]], "total_found": 1},
    {"name": "080 A", "corpus": "test_080", "query": T("a", 1), "ranker": "proximity_bm25", "expect": [[1, 1725]], "total_found": 1},
    {"name": "080 A$", "corpus": "test_080", "query": T("a", 1, tp="end"), "ranker": "proximity_bm25", "expect": [[1, 1725]], "total_found": 1},
    {"name": "080 X", "corpus": "test_080", "query": T("x", 1), "ranker": "proximity_bm25", "expect": [[2, 1643]], "total_found": 1},
    {"name": "080 Y", "corpus": "test_080", "query": T("y", 1), "ranker": "proximity_bm25", "expect": [[2, 1643]], "total_found": 1},
]
G["cases"] += [  # the rest of test_019's query list, as far as its trees can be written down without the query parser
    {"name": "019 \"test that\"~3 | basic", "corpus": "test_019",
     # the keyword after '"..."~N' stands one position further on (XQNode_t::FixupAtomPos, sphinxquery.cpp:923-938)
     "query": OP("or", OP("proximity", T("test", 1), T("that", 2), opt=3), T("basic", 4)), "ranker": "proximity_bm25",
     "expect": [[333, 1647], [111, 1551], [555, 1551]]},
    {"name": "019 \"hello program\"~3", "corpus": "test_019", "query": OP("proximity", T("hello", 1), T("program", 2), opt=3),
     "ranker": "proximity_bm25", "expect": []},
    {"name": "019 \"quorum query test\"/4", "corpus": "test_019",
     "query": OP("quorum", T("quorum", 1), T("query", 2), T("test", 3), opt=4), "ranker": "proximity_bm25", "expect": []},
    {"name": "019 0077", "corpus": "test_019", "query": T("0077", 1), "ranker": "proximity_bm25", "expect": [[888, 1720]]},
    {"name": "019 @title test", "corpus": "test_019", "query": T("test", 1, 0b01), "ranker": "proximity_bm25", "expect": []},
    {"name": "019 @!title aaa", "corpus": "test_019", "query": T("aaa", 1, 0xFFFFFFFE), "ranker": "proximity_bm25",
     "expect": [[901, 1653], [903, 1611]]},
    {"name": "019 @@relaxed @!nonexistent test", "corpus": "test_019", "query": T("test", 1), "ranker": "proximity_bm25",
     "expect": [[333, 1720]]},
    {"name": "019 \"phrase (!query)/ ~on @steroids\"", "corpus": "test_019",
     "query": OP("phrase", T("phrase", 1), T("query", 2), T("on", 3), T("steroids", 4)), "ranker": "proximity_bm25",
     "expect": [[222, 4704]]},
    {"name": "019 1234567812345678", "corpus": "test_019", "query": T("1234567812345678", 1), "ranker": "proximity_bm25",
     "expect": [[999, 1720]]},
    {"name": "019 canon 16 35", "corpus": "test_019", "query": OP("and", T("canon", 1), T("16", 2), T("35", 3)),
     "ranker": "proximity_bm25", "expect": [[555, 2720]]},
]
for spam, exp in [(10, [[1, 25], [2, 15], [3, 15]]), (0, [[1, 5], [2, 5], [3, 5]]), (-10, [[2, -5], [3, -5], [1, -15]])]:
    G["cases"].append({"name": f"322 program flow wordcount, field_weights 1,2,{spam}", "corpus": "test_322",
                       "query": OP("and", T("program", 1), T("flow", 2)), "ranker": "wordcount",
                       "field_weights": [1, 2, spam], "expect": exp})


# test_115: NEAR over keywords and phrases (ExtNWay_T<FSMmultinear_c>); keywords keep the positions the
# query parser numbers them with, left to right; 'a NEAR/2 b NEAR/5 c NEAR/2 d' nests (different distances do not merge)
def NEAR(n, *kids):
    return OP("near", *kids, opt=n)


A_, B_, C_, D_ = T("a", 1), T("b", 2), T("c", 3), T("d", 4)
W4444 = lambda ids: [[i, 4444] for i in ids]
for name, query, expect in [
    ('"a b" NEAR/2 "c d"', NEAR(2, OP("phrase", A_, B_), OP("phrase", C_, D_)), W4444([1, 4, 13, 14])),
    ('"c d" NEAR/2 "a b"', NEAR(2, OP("phrase", T("c", 1), T("d", 2)), OP("phrase", T("a", 3), T("b", 4))), W4444([1, 4, 13, 14])),
    ("a b NEAR/2 c d", OP("and", A_, NEAR(2, B_, C_), D_), [[4, 4444], [1, 3444], [2, 2444], [3, 2444]]),
    ("a NEAR/2 b NEAR/5 c NEAR/2 d", NEAR(2, NEAR(5, NEAR(2, A_, B_), C_), D_), W4444([1, 2, 4, 5, 6, 7, 8, 11, 12, 13, 14])),
    ("a NEAR/3 b NEAR/3 c NEAR/3 d", NEAR(3, A_, B_, C_, D_), W4444([1, 2, 3, 4, 5, 12, 13, 14])),
    ("a NEAR/3 d NEAR/3 b NEAR/3 c", NEAR(3, T("a", 1), T("d", 2), T("b", 3), T("c", 4)), W4444([1, 2, 3, 4, 5, 12, 13, 14])),
    ('"a b" NEAR/2 "c d" NEAR/2 "f g"', NEAR(2, OP("phrase", A_, B_), OP("phrase", C_, D_), OP("phrase", T("f", 5), T("g", 6))), []),
    ("five NEAR/3 one", NEAR(3, T("five", 1), T("one", 2)), []),
    ("six NEAR/3 one", NEAR(3, T("six", 1), T("one", 2)), []),
    ("aleph NEAR/2 gimel", NEAR(2, T("aleph", 1), T("gimel", 2)), [[17, 2723]]),
    ("bet NEAR/3 he", NEAR(3, T("bet", 1), T("he", 2)), [[17, 2723]]),
]:
    G["cases"].append({"name": "115 " + name, "corpus": "test_115", "query": query, "ranker": "proximity_bm25", "expect": expect,
                       "total_found": len(expect)})
# ... and the test's cases with an AND GROUP as a NEAR operand (round 3).  sphTransformExtendedQuery (sphinx.cpp:15345-15359) runs
# TransformNear (:15049-15105) over every query before the ranker is built: '(a b c) NEAR/3 d' IS 'a NEAR/3 b NEAR/3 c NEAR/3 d' by then
# (which is why the reference "answers them like a NEAR over all the group's keywords").  The tree stored here is the one the ranker
# sees ("transformed": the parser's own output has the group; mrk_parsed_transform flattens it), keywords numbered left to right.
for name, query, expect in [
    ("(a b c) NEAR/3 d", NEAR(3, A_, B_, C_, D_), W4444([1, 2, 3, 4, 5, 12, 13, 14])),
    ("burden NEAR/2 (financial share)", NEAR(2, T("burden", 1), T("financial", 2), T("share", 3)), [[15, 3836]]),
    ("burden NEAR/2 (share financial)", NEAR(2, T("burden", 1), T("share", 2), T("financial", 3)), [[15, 3836]]),
    ("(share financial) NEAR/2 burden", NEAR(2, T("share", 1), T("financial", 2), T("burden", 3)), [[15, 3836]]),
    ("(financial share) NEAR/2 burden", NEAR(2, T("financial", 1), T("share", 2), T("burden", 3)), [[15, 3836]]),
]:
    G["cases"].append({"name": "115 " + name, "corpus": "test_115", "query": query, "ranker": "proximity_bm25", "expect": expect,
                       "total_found": len(expect), "transformed": True})
# the same test's SphinxQL section lists matching rows only (the statements carry an id filter, applied here by expect_in)
for name, query, row, hit in [("bet NEAR/2 he", NEAR(2, T("bet", 1), T("he", 2)), 17, False),
                              ("oy NEAR/1 vey", NEAR(1, T("oy", 1), T("vey", 2)), 22, True),
                              ("x NEAR/2 x", NEAR(2, T("x", 1), T("x", 2)), 3, True),
                              ("x NEAR/2 x NEAR/2 x", NEAR(2, T("x", 1), T("x", 2), T("x", 3)), 9, True)]:
    G["cases"].append({"name": "115 sphinxql " + name, "corpus": "test_115", "query": query, "ranker": "proximity_bm25",
                       "expect_row": [row, hit]})

# test_349: NOTNEAR (ExtNotNear_c) over keywords, phrases, OR groups, proximity and nested NOTNEAR; the test lists matching ids
def NOTNEAR(n, a, b):
    return OP("notnear", a, b, opt=n)


PH = lambda *w: OP("phrase", *[T(x, i + 1) for i, x in enumerate(w)])
for name, query, ids in [
    ("a NOTNEAR/1 c", NOTNEAR(1, T("a", 1), T("c", 2)), list(range(1, 16))),
    ("a NOTNEAR/2 c", NOTNEAR(2, T("a", 1), T("c", 2)), list(range(2, 16))),
    ("a NOTNEAR/3 c", NOTNEAR(3, T("a", 1), T("c", 2)), [3] + list(range(5, 16))),
    ("a NOTNEAR/5 c", NOTNEAR(5, T("a", 1), T("c", 2)), list(range(7, 16))),
    ("a NOTNEAR/6 c", NOTNEAR(6, T("a", 1), T("c", 2)), list(range(8, 16))),
    ("a NOTNEAR/7 c", NOTNEAR(7, T("a", 1), T("c", 2)), list(range(10, 16))),
    ("a NOTNEAR/15 c", NOTNEAR(15, T("a", 1), T("c", 2)), list(range(10, 16))),
    ("b NOTNEAR/1 c", NOTNEAR(1, T("b", 1), T("c", 2)), list(range(4, 15))),
    ("b NOTNEAR/2 c", NOTNEAR(2, T("b", 1), T("c", 2)), list(range(5, 15))),
    ('"a b" NOTNEAR/2 c', NOTNEAR(2, PH("a", "b"), T("c", 3)), [5, 6, 7, 10, 11, 12, 13, 14]),
    ('"a b" NOTNEAR/3 c', NOTNEAR(3, PH("a", "b"), T("c", 3)), [6, 7, 10, 11, 12, 13, 14]),
    ("a NOTNEAR/3 (c | d)", NOTNEAR(3, T("a", 1), OP("or", T("c", 2), T("d", 3))), [3] + list(range(5, 16))),
    ('a NOTNEAR/3 "c x d"', NOTNEAR(3, T("a", 1), OP("phrase", T("c", 2), T("x", 3), T("d", 4))), list(range(1, 16))),
    ('a NOTNEAR/9 "c x d"', NOTNEAR(9, T("a", 1), OP("phrase", T("c", 2), T("x", 3), T("d", 4))), [1, 2, 3, 4, 5, 6, 7] + list(range(9, 16))),
    ('a NOTNEAR/11 "c x x d"', NOTNEAR(11, T("a", 1), OP("phrase", T("c", 2), T("x", 3), T("x", 4), T("d", 5))), list(range(1, 9)) + list(range(10, 16))),
    ("oy NOTNEAR/1 ho", NOTNEAR(1, T("oy", 1), T("ho", 2)), [20, 21, 22]),
    ("oy NOTNEAR/2 ho", NOTNEAR(2, T("oy", 1), T("ho", 2)), [20, 22]),
    ("zwei NOTNEAR/2 ho", NOTNEAR(2, T("zwei", 1), T("ho", 2)), [21]),
    ("zwei NOTNEAR/4 ho", NOTNEAR(4, T("zwei", 1), T("ho", 2)), []),
    ("zwei NOTNEAR/1 vey", NOTNEAR(1, T("zwei", 1), T("vey", 2)), [21]),
    ("zwei NOTNEAR/2 vey", NOTNEAR(2, T("zwei", 1), T("vey", 2)), []),
    ("vey NOTNEAR/1 ho", NOTNEAR(1, T("vey", 1), T("ho", 2)), [20, 22]),
    ("vey NOTNEAR/1 oy", NOTNEAR(1, T("vey", 1), T("oy", 2)), [20, 21, 22]),
    ("d NOTNEAR/1 a", NOTNEAR(1, T("d", 1), T("a", 2)), list(range(1, 14))),
    ("d NOTNEAR/3 a", NOTNEAR(3, T("d", 1), T("a", 2)), list(range(1, 12))),
    ("c NOTNEAR/1 x", NOTNEAR(1, T("c", 1), T("x", 2)), [1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14]),
    ("c NOTNEAR/2 x", NOTNEAR(2, T("c", 1), T("x", 2)), [1, 2, 3, 4, 5, 6, 7, 14]),
    ("c NOTNEAR/3 x", NOTNEAR(3, T("c", 1), T("x", 2)), [1, 2, 3, 4, 5, 6, 7, 14]),
    ("x NOTNEAR/2 c", NOTNEAR(2, T("x", 1), T("c", 2)), [3, 6, 7, 8, 9, 10, 11, 12, 13]),
    ('("a b" NOTNEAR/3 (d |e)) NOTNEAR/2 c', NOTNEAR(2, NOTNEAR(3, PH("a", "b"), OP("or", T("d", 3), T("e", 4))), T("c", 5)), [5, 6, 7, 10, 11, 12, 13, 14]),
    ('"a b"~4 NOTNEAR/1 c', NOTNEAR(1, OP("proximity", T("a", 1), T("b", 2), opt=4), T("c", 4)), list(range(4, 15))),
    ('( "a b" | "a x b" ) NOTNEAR/1 c', NOTNEAR(1, OP("or", PH("a", "b"), OP("phrase", T("a", 3), T("x", 4), T("b", 5))), T("c", 6)), [4, 5, 6, 7, 8, 10, 11, 12, 13, 14]),
]:
    G["cases"].append({"name": "349 " + name, "corpus": "test_349", "query": query, "ranker": "proximity_bm25", "expect_ids": ids})


# test_133: SENTENCE / PARAGRAPH (ExtUnit_c) over an index_sp = 1, html_strip = 1 index.  Rows 1-6, 100, 101 as the test inserts them
# (the SQL's '&lt;p&gt;' is '<p>' by the time the stripper sees it); the index holds 17 docs in all (model.bin's BM25 parts fit
# N = 17: 598 for two keywords of 4 docs each) -- the other nine (zone / entity rows) share no keyword with these queries and stand
# here as filler text.  The boundary keywords are indexed like words ("unit": their dictionary text).
SENT, PARA = "\x03sentence", "\x03paragraph"
G["corpora"]["test_133"] = {
    "source": "test/test_133/test.xml (index test: rows 1-6, 100, 101 + nine rows that hold none of the queried words) + model.bin",
    "min_word_len": 1, "index_sp": True,
    "ids": [1, 2, 3, 4, 5, 6, 100, 101, 200, 201, 211, 202, 300, 301, 310, 311, 400],
    "docs_spec": [{"count": 1, "fields": [t]} for t in [
        "One and one and one. And two. And three, all separate.",
        "And then we'll have him, one two three! Kidnap the Sandy Clawz...",
        "One two something. But not three.",
        "Two says hello to one more three.",
        {"runs": [["A ram zam zam. ", 171], ["Zam ram!", 1]]},
        "A ram zam zam, a ram zam zam, guli guli guli guli ram zam zam.",
        "Quick brown fox<p>jumps over a lazy dog.",
        "In paragraph, yes. Not in sentence, no."]] +
    [{"count": 9, "fields": ["the walrus said of shoes and ships and sealing wax"]}]}


def UNIT(kind, *kids):
    n = OP(kind, *kids)
    n["unit"] = SENT if kind == "sentence" else PARA
    return n


for name, query, exp in [
    ("one SENTENCE two", UNIT("sentence", T("one", 1), T("two", 2)), [[2, 2598], [3, 2598], [4, 1598]]),
    ("one SENTENCE two three", OP("and", UNIT("sentence", T("one", 1), T("two", 2)), T("three", 3)), [[2, 3598], [3, 2598], [4, 2598]]),
    ("one SENTENCE two SENTENCE three", UNIT("sentence", T("one", 1), T("two", 2), T("three", 3)), [[2, 3598], [4, 2598]]),
    ('"one two" SENTENCE three', UNIT("sentence", OP("phrase", T("one", 1), T("two", 2)), T("three", 3)), [[2, 2598]]),
    ("zam SENTENCE ram", UNIT("sentence", T("zam", 1), T("ram", 2)), [[5, 2857], [6, 1778]]),
    ("fox PARAGRAPH dog", UNIT("paragraph", T("fox", 1), T("dog", 2)), []),
    ("sentence SENTENCE paragraph", UNIT("sentence", T("sentence", 1), T("paragraph", 2)), []),
    ("sentence PARAGRAPH paragraph", UNIT("paragraph", T("sentence", 1), T("paragraph", 2)), [[101, 1722]]),
    ("one SENTENCE three", UNIT("sentence", T("one", 1), T("three", 2)), [[2, 1598], [4, 1598]]),
]:
    G["cases"].append({"name": "133 " + name, "corpus": "test_133", "query": query, "ranker": "proximity_bm25", "expect": exp, "total_found": len(exp)})

for _name, _c in G["corpora"].items():  # a corpus given as "like another one, with row 21 replaced"
    if "like" in _c:
        spec = [dict(r) for r in G["corpora"][_c.pop("like")]["docs_spec"]]
        spec[-2] = {"count": 1, "fields": [_c.pop("row_21")]}
        _c["docs_spec"] = spec

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump(G, f, ensure_ascii=True, indent=1)
    print(f"{out}: {len(G['cases'])} cases")
