"""Writes tests/golden/wide_fields_vectors.json: the reference's own pin for indexes with more than 32 fields.

Copied by hand, as data, from test/test_183 ("support for 256 fields in disk/RT/percolate indexes"): the rows of its <db_insert> /
its RT inserts, its queries, and the row ids its model.bin lists for them (nothing of the reference is imported or executed).
Fields are given as {field index: text}; `n_fields` is the schema's field count.  Run:  python tests/golden/make_wide_vectors.py
"""
import json
import os

T21, T141, T241 = 20, 140, 240  # columns t21 / t141 / t241 of test_table = fields 20 / 140 / 240 of index `test` (t1 .. t256)

G = {
    "_about": "Reference vectors for field masks beyond 32 fields (ISphQword::CollectHitMask, sphinxsearch.cpp:50-58; searchnode.cpp:1925-1939, 2727-2747); data only",
    "corpora": {
        "test_183_test": {"source": "test/test_183/test.xml: index test (256 fields t1..t256), <db_insert> rows 1-3", "n_fields": 256, "ids": [1, 2, 3],
                          "docs": [{str(T21): "field_one", str(T141): "field_one field_one", str(T241): "field_one field_two"},
                                   {str(T21): "field_three", str(T141): "field_two", str(T241): "field_three"},
                                   {str(T21): "field_one", str(T141): "field_one", str(T241): "field_one"}]},
        "test_183_rt40": {"source": "test/test_183/test.xml: index rt40 (40 fields field1..field40), the inserts of ids 123, 125, 126", "n_fields": 40,
                          "ids": [123, 125, 126],
                          "docs": [{str(i): "kw%d" % (i + 1) for i in range(40)}, {"36": "kw37 copy2"}, {"36": "kw37 copy3"}]},
    },
    "cases": [],
}


def term(word, pos, fields=None):
    return {"word": word, "pos": pos, "fields": fields}  # fields: list of field indexes the keyword is limited to (None = any)


def case(name, corpus, query, ids):
    G["cases"].append({"name": name, "corpus": corpus, "query": query, "expect_ids": ids})


# model.bin of test_183, index `test` (the 3-field index `tests` over the same columns returns the same rows)
case("183 test field_one", "test_183_test", [term("field_one", 1)], [1, 3])
case("183 test field_two", "test_183_test", [term("field_two", 1)], [1, 2])
case("183 test @t21 field_one", "test_183_test", [term("field_one", 1, [T21])], [1, 3])
case("183 test @t21 field_two", "test_183_test", [term("field_two", 1, [T21])], [])
case("183 test @t141 field_three", "test_183_test", [term("field_three", 1, [T141])], [])
case("183 test @t141 field_two", "test_183_test", [term("field_two", 1, [T141])], [2])
case("183 test @t241 field_two", "test_183_test", [term("field_two", 1, [T241])], [1])
case("183 test field_one field_two", "test_183_test", [term("field_one", 1), term("field_two", 2)], [1])
case("183 test field_one @t141 field_two", "test_183_test", [term("field_one", 1), term("field_two", 2, [T141])], [])
case("183 test field_one @t21 field_two", "test_183_test", [term("field_one", 1), term("field_two", 2, [T21])], [])
# rt40 after the inserts of 123, 125, 126 (model.bin: 'kw37' and '@field37 kw37' -> 123 125 126, in the order of the reference's
# default sort before the merge-forcing dummies; '@field5 kw37' -> none while only 123 was there)
case("183 rt40 kw37", "test_183_rt40", [term("kw37", 1)], [123, 125, 126])
case("183 rt40 @field5 kw37", "test_183_rt40", [term("kw37", 1, [4])], [])
case("183 rt40 @field37 kw37", "test_183_rt40", [term("kw37", 1, [36])], [123, 125, 126])

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wide_fields_vectors.json")
json.dump(G, open(out, "w", encoding="utf-8"), indent=1, sort_keys=True)
print("wrote", out, len(G["cases"]), "cases")
