// mrk_plan.cpp -- the query planner: mrk_query (flattened XQNode_t tree + ranker knobs) -> device passes and work
// items.  Mirrors what sphCreateRanker / ExtNode_i::Create decide on the host in the reference
// (sphinxsearch.cpp:4167-4378, searchnode.cpp:1599-1811).
#include "mrk_host_int.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

using namespace mrk;

// Fixed-capacity vector for the planner's scratch lists: planning runs per query on the submit path (a few hundred
// nanoseconds each), heap allocations would dominate it.  Pushing past the capacity sets `overflow` (checked once per
// query: such a query is declined) instead of growing.
template <typename T, int N>
struct SmallVec {
  T v[N];
  int n = 0;
  bool overflow = false;
  SmallVec() {}
  explicit SmallVec(int count) : n(count <= N ? count : N), overflow(count > N) {
    for (int i = 0; i < n; ++i) v[i] = T();
  }
  void push_back(const T& x) {
    if (n < N)
      v[n++] = x;
    else
      overflow = true;
  }
  size_t size() const { return (size_t)n; }
  bool empty() const { return n == 0; }
  T& operator[](size_t i) { return v[i]; }
  const T& operator[](size_t i) const { return v[i]; }
  T* begin() { return v; }
  T* end() { return v + n; }
  const T* begin() const { return v; }
  const T* end() const { return v + n; }
  T& back() { return v[n - 1]; }
  const T& back() const { return v[n - 1]; }
  T& front() { return v[0]; }
  const T& front() const { return v[0]; }
  void clear() { n = 0; }
  void append(const T* b, const T* e) {
    for (; b != e; ++b) push_back(*b);
  }
};
constexpr int PLAN_CAP = 40; // keywords / nodes / children a plan may hold before the size checks decline the query
typedef SmallVec<int, PLAN_CAP> IntVec;

// planner: mrk_query -> DevQuery + work items
// ----------------------------------------------------------------------------------------
struct PlanKw { // one keyword occurrence of the query tree
  int32_t term_id;
  int32_t node;
  int docs;
  float boost, idf;
  uint32_t queried32;
  int atom_pos;
  int tp_kind, tp_max; // MRK_TERMPOS_*
  bool weighted_first; // first node of its word in GetQwords order gets the IDF, later dupes get 0
  bool hidden = false; // the boundary keyword of a SENTENCE / PARAGRAPH node: read for its hits, not a query word (bNotWeighted, no IDF, no query position)
};

struct PlanNode { // binary eval-tree node, post-order
  uint32_t op;
  int l = -1, r = -1, kw = -1;
};

struct PlanTree {
  SmallVec<PlanKw, PLAN_CAP> kws;     // in GetQwords traversal order
  SmallVec<PlanNode, PLAN_CAP> nodes; // post-order; root = last
  bool multiand3_inner = false; // a 3-keyword ExtMultiAnd_T below the root (MergeHits3 quirk not restated there)
  bool phrase = false;          // root is a PHRASE (ExtNWay_T<FSMphrase_c>)
  bool ph_leaf = false;         // one PHRASE below other operators; its words are kws[ph_kw0 .. ph_kw0 + ph_n)
  int ph_kw0 = 0, ph_n = 0;
  int px_dist = 0;              // > 0: the phrase node is a PROXIMITY operator ('"a b"~N')
  bool force_tree = false;      // the query must run as a tree program even if it only holds TERM / AND nodes
  // one real ExtQuorum_c ('"a b c"/N', 1 < N < words): its keywords are kws[q_kw0 .. q_kw0 + q_n) in query-position order
  bool quorum = false, quorum_root = false;
  bool order = false;           // the ph_leaf node is a BEFORE operator (ExtOrder_c), not a PHRASE
  bool termpos = false;         // some keyword carries a position modifier ('^word', 'word$', '@field[N] word')
  bool notnear = false;         // one NOTNEAR node over two plain keywords: kws[nn_a] NOTNEAR/nn_dist kws[nn_b]
  int nn_a = 0, nn_b = 0, nn_dist = 0;
  int q_kw0 = 0, q_n = 0, q_thr = 0;
  IntVec atoms;                 // its words' query positions, phrase order
  bool gen_nearn = false;       // ... with a NEAR over three and more operands at its root
  bool gen = false;             // planned for the generic per-doc evaluator: nodes = the doc-level superset tree, the real tree is a GenProg
};

// Mirrors ExtNode_i::Create (searchnode.cpp:1599-1811) for the operators the device path knows:
// AND over plain keywords -> ExtMultiAnd_T (nodes sorted by ascending docs with sphSort's small-array
// insertion sort, sphinxstd.h:853-869: equal keys end in reverse arrival order); other AND / OR / MAYBE /
// ANDNOT -> left-deep chains in child order (searchnode.cpp:1785-1806).  An N-way MultiAnd is emitted as
// a left-deep AND chain: ((0+t0)+t1)+t2 and (t0+t1)+t2 are the same fp32 value.
static int build_tree(const mrk_segment* seg, const mrk_query& q, int32_t ni, PlanTree& T, uint32_t qi, int depth, bool is_root, int& err) {
  if (ni < 0 || ni >= q.n_nodes || depth > 16) return err = mrk_fail(MRK_E_INVAL, "query %u: bad tree", qi), -1;
  const mrk_node& n = q.nodes[ni];
  auto leaf = [&](int32_t li) -> int {
    const mrk_node& t = q.nodes[li];
    PlanKw k{};
    k.term_id = t.term_id;
    k.node = li;
    k.docs = (t.term_id >= 0 && (uint32_t)t.term_id < seg->terms.size()) ? (int)seg->terms[t.term_id].docs : 0;
    k.boost = t.boost;
    k.queried32 = t.field_mask;
    k.atom_pos = t.atom_pos;
    k.tp_kind = t.term_pos;
    k.tp_max = t.field_max_pos;
    if (t.term_pos) T.termpos = T.force_tree = true; // ExtTermPos_T never sits inside an ExtMultiAnd_T (searchnode.cpp:1724-1735)
    T.kws.push_back(k);
    PlanNode pn;
    pn.op = PN_TERM;
    pn.kw = (int)T.kws.size() - 1;
    T.nodes.push_back(pn);
    return (int)T.nodes.size() - 1;
  };
  auto join = [&](uint32_t op, int l, int r) -> int {
    PlanNode pn;
    pn.op = op;
    pn.l = l;
    pn.r = r;
    T.nodes.push_back(pn);
    return (int)T.nodes.size() - 1;
  };
  if (n.op == MRK_OP_TERM) return leaf(ni);
  const bool near = n.op == MRK_OP_NEAR; // ExtNWay_T<FSMmultinear_c>: the same node with a third state machine
  const bool nway = n.op == MRK_OP_PHRASE || n.op == MRK_OP_PROXIMITY || near; // ExtNWay_T<FSMphrase_c / FSMproximity_c / FSMmultinear_c>
  if (nway && (T.phrase || T.ph_leaf))
    return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: more than one PHRASE (device path: one per query)", qi), -1;
  if (n.op == MRK_OP_QUORUM) {
    // ExtNode_i::Create, SPH_QUERY_QUORUM (searchnode.cpp:1638-1686): threshold 1 = an ExtOr_c chain, threshold >= word
    // count = an ExtAnd_c chain, both over the words sorted by ascending doc count; a real ExtQuorum_c in between
    if (n.n_children < 2 || n.n_children > 16 || n.first_child < 0) return err = mrk_fail(MRK_E_INVAL, "query %u: bad child list", qi), -1;
    if (n.opt < 1) return err = mrk_fail(MRK_E_INVAL, "query %u: quorum threshold %d", qi, n.opt), -1;
    const bool real_quorum = n.opt != 1 && n.opt < n.n_children;
    if (real_quorum && (T.quorum || n.n_children > QUORUM_EVENTS))
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: quorum nodes on the device path: one per query, <= %d keywords", qi, QUORUM_EVENTS), -1;
    IntVec kids(n.n_children), ord(n.n_children), docs(n.n_children);
    for (int i = 0; i < n.n_children; ++i) {
      kids[i] = q.children[n.first_child + i];
      if (kids[i] < 0 || kids[i] >= q.n_nodes || q.nodes[kids[i]].op != MRK_OP_TERM || q.nodes[kids[i]].term_pos)
        return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: quorum over plain keywords only", qi), -1;
      const mrk_node& t = q.nodes[kids[i]];
      docs[i] = (t.term_id >= 0 && (uint32_t)t.term_id < seg->terms.size()) ? (int)seg->terms[t.term_id].docs : 0;
      ord[i] = i;
    }
    for (int i = 1; i < n.n_children; ++i)
      for (int j = i; j > 0; --j) {
        if (docs[ord[j - 1]] < docs[ord[j]]) break;
        std::swap(ord[j], ord[j - 1]);
      }
    const size_t kw0 = T.kws.size();
    if (real_quorum) {
      // ExtQuorum_c (searchnode.cpp:4342-4403): children in query-position order, no TERM nodes of their own in the
      // program -- the node reads its keywords' presence bits itself
      for (int i = 0; i < n.n_children; ++i) ord[i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return q.nodes[kids[a]].atom_pos < q.nodes[kids[b]].atom_pos; });
      const size_t nodes0 = T.nodes.size();
      for (int i = 0; i < n.n_children; ++i) {
        for (int j = 0; j < i; ++j)
          if (q.nodes[kids[ord[i]]].term_id >= 0 && q.nodes[kids[ord[j]]].term_id == q.nodes[kids[ord[i]]].term_id)
            return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: quorum with a repeated keyword (m_bHasDupes) is not on the device path", qi), -1;
        (void)leaf(kids[ord[i]]);
      }
      T.nodes.n = (int)nodes0; // drop the TERM nodes leaf() appended; the keywords stay in T.kws
      for (size_t k = kw0; k < T.kws.size(); ++k) {
        T.kws[k].queried32 &= n.field_mask;
        // where a keyword's stream ends is only known for unrestricted keywords (its last doc); see qr_row
        const uint32_t all = seg->n_fields >= 32 ? 0xFFFFFFFFu : (1u << seg->n_fields) - 1u;
        if ((T.kws[k].queried32 & all) != all)
          return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: field-limited keywords in a quorum are not on the device path", qi), -1;
      }
      T.quorum = true;
      T.quorum_root = is_root;
      T.q_kw0 = (int)kw0;
      T.q_n = n.n_children;
      T.q_thr = n.opt;
      PlanNode pn;
      pn.op = PN_QUORUM;
      T.nodes.push_back(pn);
      return (int)T.nodes.size() - 1;
    }
    int cur = leaf(kids[ord[0]]);
    for (int i = 1; i < n.n_children; ++i) {
      const int r = leaf(kids[ord[i]]);
      cur = join(n.opt == 1 ? PN_OR : PN_AND, cur, r);
    }
    for (size_t k = kw0; k < T.kws.size(); ++k) T.kws[k].queried32 &= n.field_mask; // Create ( word, pNode, .. )
    T.force_tree = true; // an ExtAnd_c chain is not an ExtMultiAnd_T (no MergeHits3 quirk): always the tree program
    return cur;
  }
  if (n.op == MRK_OP_NOTNEAR) {
    // ExtNotNear_c (generic create, searchnode.cpp:1785-1803): the must side's docs; where the not side holds the doc too, only
    // the must hits no later not-hit comes within the distance of survive, and the doc stays iff one does
    if (T.notnear || T.phrase || T.ph_leaf || T.order)
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: NOTNEAR next to another NOTNEAR / PHRASE / BEFORE node (device path: one such node per query)", qi), -1;
    if (n.n_children != 2 || n.first_child < 0) return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: NOTNEAR over %d operands (device path: two plain keywords)", qi, n.n_children), -1;
    if (n.opt <= 0 || n.opt > (1 << 20)) return err = mrk_fail(MRK_E_INVAL, "query %u: NOTNEAR distance %d", qi, n.opt), -1;
    int kid[2];
    for (int i = 0; i < 2; ++i) {
      kid[i] = q.children[n.first_child + i];
      if (kid[i] < 0 || kid[i] >= q.n_nodes) return err = mrk_fail(MRK_E_INVAL, "query %u: child index out of range", qi), -1;
      if (q.nodes[kid[i]].op != MRK_OP_TERM || q.nodes[kid[i]].term_pos)
        return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: NOTNEAR over plain keywords only on the device path", qi), -1;
    }
    const int l = leaf(kid[0]);
    T.nn_a = (int)T.kws.size() - 1;
    const int r = leaf(kid[1]);
    T.nn_b = (int)T.kws.size() - 1;
    T.nn_dist = n.opt;
    T.notnear = T.force_tree = true;
    return join(PN_NOTNEAR, l, r);
  }
  if (n.op == MRK_OP_BEFORE) {
    // ExtOrder_c (CreateOrderNode, searchnode.cpp:1044-1073): children in query order, no sorting; the doc is the FIRST
    // child's doc (fields, tfidf) once all children hold it and their hits line up in order inside one field
    if (T.phrase || T.ph_leaf)
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: more than one PHRASE / BEFORE node (device path: one per query)", qi), -1;
    if (n.n_children < 2 || n.n_children > MAX_PROX_TERMS || n.first_child < 0)
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: BEFORE over %d nodes (device path: 2..%d plain keywords)", qi, n.n_children, MAX_PROX_TERMS), -1;
    const int kw0 = (int)T.kws.size();
    int cur = -1;
    for (int i = 0; i < n.n_children; ++i) {
      const int32_t ki = q.children[n.first_child + i];
      if (ki < 0 || ki >= q.n_nodes) return err = mrk_fail(MRK_E_INVAL, "query %u: child index out of range", qi), -1;
      if (q.nodes[ki].op != MRK_OP_TERM)
        return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: BEFORE over plain keywords only on the device path", qi), -1;
      T.atoms.push_back(q.nodes[ki].atom_pos);
      if (i && T.atoms[i] <= T.atoms[i - 1]) return err = mrk_fail(MRK_E_INVAL, "query %u: BEFORE operands' query positions must ascend", qi), -1;
      const int l = leaf(ki);
      cur = cur < 0 ? l : join(PN_AND, cur, l);
    }
    T.ph_leaf = T.order = T.force_tree = true;
    T.ph_kw0 = kw0;
    T.ph_n = n.n_children;
    PlanNode pn;
    pn.op = PN_ORDERFIX;
    pn.l = cur;
    pn.kw = kw0; // the first child: its doc is the node's doc
    T.nodes.push_back(pn);
    return (int)T.nodes.size() - 1;
  }
  if (!nway && n.op != MRK_OP_AND && n.op != MRK_OP_OR && n.op != MRK_OP_MAYBE && n.op != MRK_OP_ANDNOT)
    return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: operator %d not on the device path yet", qi, n.op), -1;
  if (n.n_children < 1 || n.n_children > 16 || n.first_child < 0) return err = mrk_fail(MRK_E_INVAL, "query %u: bad child list", qi), -1;
  IntVec kids(n.n_children);
  bool all_terms = true;
  for (int i = 0; i < n.n_children; ++i) {
    kids[i] = q.children[n.first_child + i];
    if (kids[i] < 0 || kids[i] >= q.n_nodes) return err = mrk_fail(MRK_E_INVAL, "query %u: child index out of range", qi), -1;
    all_terms &= q.nodes[kids[i]].op == MRK_OP_TERM;
  }
  if (nway) {
    if (n.op == MRK_OP_PROXIMITY || near) {
      if (n.opt <= 0 || n.opt > (1 << 20)) return err = mrk_fail(MRK_E_INVAL, "query %u: proximity / NEAR distance %d", qi, n.opt), -1;
      T.px_dist = n.opt | (near ? (int)0x80000000u : 0);
    }
    // NEAR: two plain keywords on the device.  With three or more operands the reference's folded hit carries a query position
    // that depends on the docs evaluated before (FSMmultinear_c::m_uFirstQpos is never reset for the ring form), and operands
    // that are phrases / groups need their own hit streams: both are declined
    if (near && (!all_terms || n.n_children != 2))
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: NEAR over %d operands (device path: two plain keywords)", qi, n.n_children), -1;
    // CreateMultiNode<ExtPhrase_c / ExtProximity_c> (searchnode.cpp:984-1041): plain keywords only; ExtNWay_T::ConstructNode
    // (:3767-3787) chains them left-deep in ascending doc-count order, so docs / tfidf come out as for a MultiAnd
    if (!all_terms || n.n_children < 2 || n.n_children > MAX_PROX_TERMS)
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: PHRASE of %d nodes (device path: 2..%d plain keywords)", qi, n.n_children,
                            MAX_PROX_TERMS), -1;
    (is_root ? T.phrase : T.ph_leaf) = true;
    for (int i = 0; i < n.n_children; ++i) {
      T.atoms.push_back(q.nodes[kids[i]].atom_pos);
      if (i && T.atoms[i] <= T.atoms[i - 1]) return err = mrk_fail(MRK_E_INVAL, "query %u: phrase atom positions must ascend", qi), -1;
    }
    if (!near && T.atoms.back() - T.atoms.front() >= PHRASE_STATES)
      return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: phrase spans %d positions (device path: < %d)", qi,
                            T.atoms.back() - T.atoms.front(), PHRASE_STATES), -1;
  }
  if ((n.op == MRK_OP_AND || nway) && all_terms && n.n_children > 1) {
    IntVec ord(n.n_children), docs(n.n_children);
    bool any_tp = false;
    for (int i = 0; i < n.n_children; ++i) {
      const mrk_node& t = q.nodes[kids[i]];
      docs[i] = (t.term_id >= 0 && (uint32_t)t.term_id < seg->terms.size()) ? (int)seg->terms[t.term_id].docs : 0;
      ord[i] = i;
      if (t.term_pos) { // ExtConditional_T keeps ExtNode_i's GetDocsCount() = INT_MAX (searchnode.h:83): sorted last
        if (nway) return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: position modifiers on the words of a phrase", qi), -1;
        docs[i] = INT_MAX;
        any_tp = true;
      }
    }
    for (int i = 1; i < n.n_children; ++i)
      for (int j = i; j > 0; --j) {
        if (docs[ord[j - 1]] < docs[ord[j]]) break;
        std::swap(ord[j], ord[j - 1]);
      }
    if (n.op == MRK_OP_AND && n.n_children == 3 && !is_root && !any_tp) T.multiand3_inner = true;
    const int kw0 = (int)T.kws.size();
    int cur = leaf(kids[ord[0]]);
    for (int i = 1; i < n.n_children; ++i) {
      const int r = leaf(kids[ord[i]]);
      cur = join(PN_AND, cur, r);
    }
    if (nway) { // the words are created with the phrase node's field limit (searchnode.cpp:1020-1024)
      for (size_t k = (size_t)kw0; k < T.kws.size(); ++k) T.kws[k].queried32 &= n.field_mask;
      if (!is_root) { // ExtNWay_T on top of the words' AND chain: keeps the docs where the words line up
        T.ph_kw0 = kw0;
        T.ph_n = n.n_children;
        PlanNode pn;
        pn.op = PN_PHRASEFIX;
        pn.l = cur;
        T.nodes.push_back(pn);
        cur = (int)T.nodes.size() - 1;
      }
    }
    return cur;
  }
  const uint32_t op = n.op == MRK_OP_AND ? PN_AND : n.op == MRK_OP_OR ? PN_OR : n.op == MRK_OP_MAYBE ? PN_MAYBE : PN_ANDNOT;
  int cur = -1;
  for (int i = 0; i < n.n_children; ++i) {
    const int c = build_tree(seg, q, kids[i], T, qi, depth + 1, false, err);
    if (c < 0) return -1;
    cur = cur < 0 ? c : join(op, cur, c);
  }
  return cur;
}


// ---- the generic path: the reference's evaluation tree as a GenProg (mrk_keval.h evaluates it per candidate doc), plus
// the doc-level superset tree the scan kernel runs to find the candidates: every PHRASE / PROXIMITY / NEAR / BEFORE node as
// the AND of its operands, NOTNEAR / ANDNOT as their left side (MAYBE: the right side's keywords are still located).
// Follows ExtNode_i::Create (searchnode.cpp:1599-1811) node for node; keywords enter T.kws in GetQwords traversal order.
struct GenBuild {
  mrk::GenProg prog;
  int plan[mrk::GEN_MAX_NODES]; // the superset tree's node for each program node
  int docs[mrk::GEN_MAX_NODES]; // GetDocsCount(): a plain keyword's docs, INT_MAX for everything else (searchnode.h:83)
  int atom[mrk::GEN_MAX_NODES]; // GetAtomPos()
  bool overflow = false;
  GenBuild() { memset(&prog, 0, sizeof prog); }
  int add(const mrk::GenNode& n, int plan_node, int ndocs, int natom) {
    if (prog.n_nodes >= (uint32_t)mrk::GEN_MAX_NODES) {
      overflow = true;
      return 0;
    }
    const int i = (int)prog.n_nodes++;
    prog.nodes[i] = n;
    plan[i] = plan_node, docs[i] = ndocs, atom[i] = natom;
    return i;
  }
};

static int node_docs_key(const mrk_segment* seg, const mrk_node& t) { // what ExtNodeTF(Ext)_fn sorts operands by
  if (t.op != MRK_OP_TERM || t.term_pos) return INT_MAX;
  return (t.term_id >= 0 && (uint32_t)t.term_id < seg->terms.size()) ? (int)seg->terms[t.term_id].docs : 0;
}

static void sph_isort(IntVec& ord, const IntVec& key) { // sphSort's small-array path: equal keys end in reverse arrival order
  for (int i = 1; i < (int)ord.size(); ++i)
    for (int j = i; j > 0; --j) {
      if (key[ord[j - 1]] < key[ord[j]]) break;
      std::swap(ord[j], ord[j - 1]);
    }
}

static int build_gen(const mrk_segment* seg, const mrk_query& q, int32_t ni, PlanTree& T, GenBuild& G, uint32_t qi, int depth, int& err) {
  using namespace mrk;
  if (ni < 0 || ni >= q.n_nodes || depth > 16) return err = mrk_fail(MRK_E_INVAL, "query %u: bad tree", qi), -1;
  const mrk_node& n = q.nodes[ni];
  auto pjoin = [&](uint32_t op, int l, int r) -> int {
    PlanNode pn;
    pn.op = op, pn.l = l, pn.r = r;
    T.nodes.push_back(pn);
    return (int)T.nodes.size() - 1;
  };
  // a keyword: its slot in T.kws + its PN_TERM node in the superset tree
  auto keyword = [&](int32_t li, uint32_t mask_and, int& plan_node) -> int {
    const mrk_node& t = q.nodes[li];
    PlanKw k{};
    k.term_id = t.term_id;
    k.node = li;
    k.docs = (t.term_id >= 0 && (uint32_t)t.term_id < seg->terms.size()) ? (int)seg->terms[t.term_id].docs : 0;
    k.boost = t.boost;
    k.queried32 = t.field_mask & mask_and;
    k.atom_pos = t.atom_pos;
    k.tp_kind = t.term_pos;
    k.tp_max = t.field_max_pos;
    T.kws.push_back(k);
    PlanNode pn;
    pn.op = PN_TERM;
    pn.kw = (int)T.kws.size() - 1;
    T.nodes.push_back(pn);
    plan_node = (int)T.nodes.size() - 1;
    return pn.kw;
  };
  auto term_node = [&](int32_t li, uint32_t mask_and) -> int {
    int pl;
    const int slot = keyword(li, mask_and, pl);
    GenNode g{};
    g.kind = GN_TERM;
    g.kid[0] = (uint8_t)slot;
    return G.add(g, pl, node_docs_key(seg, q.nodes[li]), q.nodes[li].atom_pos);
  };
  auto twofer = [&](uint32_t kind, uint32_t pop, int l, int r, int opt = 0) -> int {
    GenNode g{};
    g.kind = (uint8_t)kind, g.n_kids = 2, g.kid[0] = (uint8_t)l, g.kid[1] = (uint8_t)r, g.opt = opt;
    return G.add(g, pjoin(pop, G.plan[l], G.plan[r]), INT_MAX, G.atom[l]);
  };
  // ExtNWay_T::ConstructNode (:3767-3802): the operands chained left-deep in ascending doc-count order, every hit relabelled
  // with its operand's place in the query (1-based), the last ExtAnd_c emitting in reverse query-position order.  One
  // operand at a time (operand, AND, operand, AND ...: the superset tree's program stays two values deep)
  auto nway_step = [&](int& cur, int c, const IntVec& ord, int i) {
    if (i == 0) {
      cur = c;
      return;
    }
    const int a = twofer(GN_AND, PN_AND, cur, c);
    G.prog.nodes[a].npl = (uint16_t)(i == 1 ? ord[0] + 1 : 0), G.prog.nodes[a].npr = (uint16_t)(ord[i] + 1);
    if (i + 1 == (int)ord.size()) G.prog.nodes[a].flags |= 1u;
    cur = a;
  };
  if (n.op == MRK_OP_TERM) {
    if (n.term_pos < 0 || n.term_pos > MRK_TERMPOS_LIMIT || (n.term_pos == MRK_TERMPOS_LIMIT && n.field_max_pos <= 0))
      return err = mrk_fail(MRK_E_INVAL, "query %u: bad position modifier", qi), -1;
    return term_node(ni, 0xFFFFFFFFu);
  }
  if (n.n_children < 1 || n.n_children > 16 || n.first_child < 0) return err = mrk_fail(MRK_E_INVAL, "query %u: bad child list", qi), -1;
  IntVec kids(n.n_children);
  bool all_terms = true, any_tp = false;
  for (int i = 0; i < n.n_children; ++i) {
    kids[i] = q.children[n.first_child + i];
    if (kids[i] < 0 || kids[i] >= q.n_nodes) return err = mrk_fail(MRK_E_INVAL, "query %u: child index out of range", qi), -1;
    all_terms &= q.nodes[kids[i]].op == MRK_OP_TERM;
    any_tp |= q.nodes[kids[i]].op == MRK_OP_TERM && q.nodes[kids[i]].term_pos != 0;
  }
  IntVec ord(n.n_children), key(n.n_children);
  for (int i = 0; i < n.n_children; ++i) ord[i] = i, key[i] = node_docs_key(seg, q.nodes[kids[i]]);
  switch (n.op) {
    case MRK_OP_PHRASE:
    case MRK_OP_PROXIMITY: { // CreateMultiNode<ExtPhrase_c / ExtProximity_c> (:984-1041): plain keywords under the node's field limit
      if (!all_terms || any_tp || n.n_children < 2 || n.n_children > MRK_MAX_AND_TERMS)
        return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: PHRASE of %d nodes (generic path: 2..%d plain keywords)", qi, n.n_children, MRK_MAX_AND_TERMS), -1;
      if (n.op == MRK_OP_PROXIMITY && (n.opt <= 0 || n.opt > (1 << 20))) return err = mrk_fail(MRK_E_INVAL, "query %u: proximity distance %d", qi, n.opt), -1;
      for (int i = 1; i < n.n_children; ++i)
        if (q.nodes[kids[i]].atom_pos <= q.nodes[kids[i - 1]].atom_pos) return err = mrk_fail(MRK_E_INVAL, "query %u: phrase atom positions must ascend", qi), -1;
      sph_isort(ord, key);
      IntVec nodes_q(n.n_children);
      int inner = -1;
      for (int i = 0; i < n.n_children; ++i) { // (keywords enter in chain order)
        nodes_q[ord[i]] = term_node(kids[ord[i]], n.field_mask);
        nway_step(inner, nodes_q[ord[i]], ord, i);
      }
      GenNode g{};
      g.kind = n.op == MRK_OP_PHRASE ? GN_PHRASE : GN_PROX;
      g.n_kids = 1, g.kid[0] = (uint8_t)inner, g.n_words = (uint8_t)n.n_children, g.opt = n.opt;
      for (int i = 0; i < n.n_children; ++i) g.aux[i] = G.prog.nodes[nodes_q[i]].kid[0];
      return G.add(g, G.plan[inner], INT_MAX, G.atom[nodes_q[0]]);
    }
    case MRK_OP_NEAR: { // CreateMultiNode<ExtMultinear_c>: operands of any kind; two of them on the device (see mrk_keval.h)
      // three and more operands: the folded hit's query position depends on the docs the node evaluated before (m_uFirstQpos
      // is never reset), which a probe launch reconstructs -- for a node every doc of which reaches the evaluator: the root
      if (n.n_children > 2 && (depth != 0 || n.n_children > MRK_MAX_AND_TERMS))
        return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: NEAR over %d operands below another operator (device path: two there, up to %d at the root)", qi,
                              n.n_children, MRK_MAX_AND_TERMS), -1;
      if (n.n_children > 2) T.gen_nearn = true;
      if (n.opt <= 0 || n.opt > (1 << 20)) return err = mrk_fail(MRK_E_INVAL, "query %u: NEAR distance %d", qi, n.opt), -1;
      for (int i = 0; i < n.n_children; ++i) {
        const int cop = q.nodes[kids[i]].op;
        if (cop != MRK_OP_TERM && cop != MRK_OP_PHRASE && cop != MRK_OP_PROXIMITY && cop != MRK_OP_NEAR)
          return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: NEAR over AND / OR groups (the reference's answer for them is not understood: parity unpinned)", qi), -1;
      }
      sph_isort(ord, key);
      IntVec nodes_q(n.n_children);
      int inner = -1;
      for (int i = 0; i < n.n_children; ++i) {
        const int c = build_gen(seg, q, kids[ord[i]], T, G, qi, depth + 1, err);
        if (c < 0) return -1;
        nodes_q[ord[i]] = c;
        nway_step(inner, c, ord, i);
      }
      GenNode g{};
      g.kind = GN_NEAR, g.n_kids = 1, g.kid[0] = (uint8_t)inner, g.opt = n.opt, g.n_words = (uint8_t)n.n_children;
      return G.add(g, G.plan[inner], INT_MAX, G.atom[nodes_q[0]]);
    }
    case MRK_OP_QUORUM: { // :1638-1686
      if (!all_terms || any_tp || n.n_children < 2) return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: quorum over plain keywords only", qi), -1;
      if (n.opt < 1) return err = mrk_fail(MRK_E_INVAL, "query %u: quorum threshold %d", qi, n.opt), -1;
      if (n.opt != 1 && n.opt < n.n_children) { // a real ExtQuorum_c: keywords in query-position order
        if (T.quorum || n.n_children > QUORUM_EVENTS || n.n_children > MRK_MAX_AND_TERMS)
          return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: quorum nodes on the device path: one per query, <= %d keywords", qi, QUORUM_EVENTS), -1;
        for (int i = 0; i < n.n_children; ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return q.nodes[kids[a]].atom_pos < q.nodes[kids[b]].atom_pos; });
        for (int i = 0; i < n.n_children; ++i)
          for (int j = 0; j < i; ++j)
            if (q.nodes[kids[i]].term_id >= 0 && q.nodes[kids[j]].term_id == q.nodes[kids[i]].term_id)
              return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: quorum with a repeated keyword (m_bHasDupes) is not on the device path", qi), -1;
        const uint32_t all = seg->n_fields >= 32 ? 0xFFFFFFFFu : (1u << seg->n_fields) - 1u;
        GenNode g{};
        g.kind = GN_QUORUM, g.n_kids = (uint8_t)n.n_children, g.opt = n.opt;
        const size_t nodes0 = T.nodes.size();
        T.q_kw0 = (int)T.kws.size();
        for (int i = 0; i < n.n_children; ++i) {
          int pl;
          g.kid[i] = (uint8_t)keyword(kids[ord[i]], n.field_mask, pl);
          if ((T.kws.back().queried32 & all) != all) // where a keyword's stream ends is only known for unrestricted keywords
            return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: field-limited keywords in a quorum are not on the device path", qi), -1;
        }
        T.nodes.n = (int)nodes0; // the quorum node reads its keywords' presence bits itself
        T.quorum = true;
        T.q_n = n.n_children, T.q_thr = n.opt;
        PlanNode pn;
        pn.op = PN_QUORUM;
        T.nodes.push_back(pn);
        return G.add(g, (int)T.nodes.size() - 1, INT_MAX, q.nodes[kids[ord[0]]].atom_pos);
      }
      sph_isort(ord, key); // threshold 1: an ExtOr_c chain; threshold >= words: an ExtAnd_c chain; both over the words by doc count
      int cur = term_node(kids[ord[0]], n.field_mask);
      for (int i = 1; i < n.n_children; ++i) {
        const int r = term_node(kids[ord[i]], n.field_mask);
        cur = n.opt == 1 ? twofer(GN_OR, PN_OR, cur, r) : twofer(GN_AND, PN_AND, cur, r);
      }
      return cur;
    }
    case MRK_OP_BEFORE: { // CreateOrderNode (:1044-1073): children as they come
      if (n.n_children < 2 || n.n_children > MRK_MAX_AND_TERMS) return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: BEFORE over %d nodes", qi, n.n_children), -1;
      GenNode g{};
      g.kind = GN_ORDER, g.n_kids = (uint8_t)n.n_children;
      int pl = -1;
      for (int i = 0; i < n.n_children; ++i) {
        const int c = build_gen(seg, q, kids[i], T, G, qi, depth + 1, err);
        if (c < 0) return -1;
        g.kid[i] = (uint8_t)c;
        pl = pl < 0 ? G.plan[c] : pjoin(PN_AND, pl, G.plan[c]);
      }
      return G.add(g, pl, INT_MAX, G.atom[g.kid[0]]);
    }
    case MRK_OP_AND: {
      if (all_terms && n.n_children > 1 && any_tp) { // a position modifier rules the multi-and node out (:1724-1762): an ExtAnd_c chain by doc count
        sph_isort(ord, key);
        int cur = term_node(kids[ord[0]], 0xFFFFFFFFu);
        for (int i = 1; i < n.n_children; ++i) cur = twofer(GN_AND, PN_AND, cur, term_node(kids[ord[i]], 0xFFFFFFFFu));
        return cur;
      }
      if (all_terms && n.n_children > 1) { // CreateMultiAndNode (:1118-1138) + ExtMultiAnd_T ctor (:2772-2798)
        if (n.n_children > MRK_MAX_AND_TERMS) return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d keywords (device path: <= %d)", qi, n.n_children, MRK_MAX_AND_TERMS), -1;
        sph_isort(ord, key);
        GenNode g{};
        g.kind = GN_MULTIAND, g.n_kids = (uint8_t)n.n_children;
        int pl = -1;
        for (int i = 0; i < n.n_children; ++i) {
          int p1;
          g.kid[i] = (uint8_t)keyword(kids[ord[i]], 0xFFFFFFFFu, p1);
          g.aux[i] = (uint8_t)ord[i];
          if (q.nodes[kids[ord[i]]].field_mask != 0xFFFFFFFFu) g.flags |= 2u;
          pl = pl < 0 ? p1 : pjoin(PN_AND, pl, p1);
        }
        return G.add(g, pl, INT_MAX, T.kws[g.kid[0]].atom_pos);
      }
      int cur = -1;
      for (int i = 0; i < n.n_children; ++i) {
        const int c = build_gen(seg, q, kids[i], T, G, qi, depth + 1, err);
        if (c < 0) return -1;
        cur = cur < 0 ? c : twofer(GN_AND, PN_AND, cur, c);
      }
      return cur;
    }
    case MRK_OP_SENTENCE:
    case MRK_OP_PARAGRAPH: { // generic create (:1785-1803): pCur = new ExtUnit_c ( pCur, pNext, fields, setup, MAGIC_WORD_... )
      int cur = -1;
      for (int i = 0; i < n.n_children; ++i) {
        const int c = build_gen(seg, q, kids[i], T, G, qi, depth + 1, err);
        if (c < 0) return -1;
        cur = cur < 0 ? c : twofer(GN_UNIT, PN_AND, cur, c);
        if (G.prog.nodes[cur].kind == GN_UNIT && cur != c) G.prog.nodes[cur].aux[0] = 0xFF;
      }
      if (n.term_id >= 0 && (uint32_t)n.term_id < seg->terms.size() && seg->terms[n.term_id].docs) {
        // the boundary keyword: one slot, located for every candidate (MAYBE: a doc without boundaries is a plain AND)
        PlanKw k{};
        k.term_id = n.term_id, k.node = ni, k.docs = (int)seg->terms[n.term_id].docs, k.boost = 1.0f, k.queried32 = n.field_mask, k.hidden = true;
        T.kws.push_back(k);
        PlanNode pn;
        pn.op = PN_TERM;
        pn.kw = (int)T.kws.size() - 1;
        T.nodes.push_back(pn);
        const int dot_plan = (int)T.nodes.size() - 1;
        for (uint32_t g = 0; g < G.prog.n_nodes; ++g)
          if (G.prog.nodes[g].kind == GN_UNIT && G.prog.nodes[g].aux[0] == 0xFF && G.prog.nodes[g].aux[1] == 0) G.prog.nodes[g].aux[0] = (uint8_t)pn.kw, G.prog.nodes[g].aux[1] = 1;
        G.plan[cur] = pjoin(PN_MAYBE, G.plan[cur], dot_plan);
      } else
        for (uint32_t g = 0; g < G.prog.n_nodes; ++g)
          if (G.prog.nodes[g].kind == GN_UNIT && G.prog.nodes[g].aux[0] == 0xFF) G.prog.nodes[g].aux[1] = 1; // (settled: no boundary keyword)
      return cur;
    }
    case MRK_OP_OR:
    case MRK_OP_MAYBE:
    case MRK_OP_ANDNOT:
    case MRK_OP_NOTNEAR: {
      if (n.op == MRK_OP_NOTNEAR && (n.opt <= 0 || n.opt > (1 << 20))) return err = mrk_fail(MRK_E_INVAL, "query %u: NOTNEAR distance %d", qi, n.opt), -1;
      int cur = -1;
      for (int i = 0; i < n.n_children; ++i) {
        const int c = build_gen(seg, q, kids[i], T, G, qi, depth + 1, err);
        if (c < 0) return -1;
        if (cur < 0)
          cur = c;
        else if (n.op == MRK_OP_OR)
          cur = twofer(GN_OR, PN_OR, cur, c);
        else if (n.op == MRK_OP_MAYBE)
          cur = twofer(GN_MAYBE, PN_MAYBE, cur, c);
        else // whether the right side holds the doc is only known after its hits were read: the left side carries the candidates
          cur = twofer(n.op == MRK_OP_ANDNOT ? GN_ANDNOT : GN_NOTNEAR, PN_MAYBE, cur, c, n.opt);
      }
      return cur;
    }
    default: return err = mrk_fail(MRK_E_UNSUPPORTED, "query %u: operator %d not on the device path yet", qi, n.op), -1;
  }
}

// keywords whose doc streams together contain every possible match of the subtree
static void cover_of(const PlanTree& T, int ni, IntVec& out) {
  const PlanNode& n = T.nodes[ni];
  if (n.op == PN_QUORUM) { // a match holds >= thr of the n keywords, hence at least one of ANY n - thr + 1: the cheapest ones
    IntVec k;
    for (int i = 0; i < T.q_n; ++i) k.push_back(T.q_kw0 + i);
    std::stable_sort(k.begin(), k.end(), [&](int a, int b) { return T.kws[a].docs < T.kws[b].docs; });
    for (int i = 0; i < T.q_n - T.q_thr + 1; ++i)
      if (T.kws[k[i]].docs > 0) out.push_back(k[i]); // a keyword without postings drives nothing
    return;
  }
  if (n.op == PN_TERM) {
    out.push_back(n.kw);
    return;
  }
  if (n.op == PN_OR) {
    cover_of(T, n.l, out);
    cover_of(T, n.r, out);
    return;
  }
  if (n.op == PN_AND) {
    IntVec a, b;
    cover_of(T, n.l, a);
    cover_of(T, n.r, b);
    uint64_t ca = 0, cb = 0;
    for (int k : a) ca += (uint64_t)T.kws[k].docs;
    for (int k : b) cb += (uint64_t)T.kws[k].docs;
    const IntVec& w = (cb < ca || (cb == ca && b.size() < a.size())) ? b : a;
    out.append(w.begin(), w.end());
    return;
  }
  cover_of(T, n.l, out); // MAYBE, ANDNOT, PHRASEFIX: the left side carries the docs
}

// keywords that must be present for the subtree to match
static uint32_t required_of(const PlanTree& T, int ni) {
  const PlanNode& n = T.nodes[ni];
  if (n.op == PN_QUORUM) return 0; // no single keyword is needed
  if (n.op == PN_TERM) return 1u << n.kw;
  if (n.op == PN_AND) return required_of(T, n.l) | required_of(T, n.r);
  if (n.op == PN_OR) return required_of(T, n.l) & required_of(T, n.r);
  return required_of(T, n.l);
}

static void fill_term(const mrk_segment* seg, const PlanKw& k, DevTerm& dt) {
  memset(&dt, 0, sizeof dt);
  dt.queried32 = k.queried32;
  dt.idf = k.weighted_first ? k.idf : 0.0f;
  dt.qpos = (uint32_t)k.atom_pos;
  dt.tp_kind = (uint32_t)k.tp_kind;
  dt.tp_max = (uint32_t)k.tp_max;
  if (!k.docs) return; // keyword without postings: nblocks = 0, never present
  const HostTerm& h = seg->terms[k.term_id];
  dt.blk_first = h.blk_first;
  dt.nblocks = h.nblocks;
  dt.docs = h.docs;
  dt.spd_end = h.doclist_off + h.doclist_len;
  dt.exc_first = h.exc_first;
  dt.exc_n = h.exc_n;
  dt.bm_off = h.bm_off;
  dt.dir_off = h.dir_off;
}

// the weight-range estimates below only shape the pruning bins: with absurd field weights they saturate instead of overflowing
static int64_t sat_mul(int64_t a, int64_t b) {
  const __int128 v = (__int128)a * b, lim = (__int128)1 << 61;
  return (int64_t)(v > lim ? lim : v < -lim ? -lim : v);
}
static int64_t sat_add(int64_t a, int64_t b) {
  const __int128 v = (__int128)a + b, lim = (__int128)1 << 61;
  return (int64_t)(v > lim ? lim : v < -lim ? -lim : v);
}

// returns MRK_OK, or MRK_E_UNSUPPORTED / MRK_E_INVAL with the message set.  dq = the query's head pass
// (index qi); further passes go to `extra` and get pass indices n_queries + position.
int mrk::plan_query(const mrk_segment* seg, const mrk_query& q, int64_t item_bytes, bool use_packed, DevQuery& dq,
                      std::vector<DevQuery>& extra, uint32_t n_queries, std::vector<DevItem>& items,
                      std::vector<DevItem>& items_bm, uint32_t qi,
                      uint64_t& algo_bytes, uint64_t& dev_bytes, uint64_t& cand_total, bool& prox_out, bool& tree_out, std::vector<mrk::GenProg>& gen_progs,
                      uint32_t rowid_max) {
  memset(&dq, 0, sizeof dq);
  dq.item_first = (uint32_t)items.size();
  dq.out_q = qi;
  if (!q.nodes || q.n_nodes <= 0 || q.root < 0 || q.root >= q.n_nodes) return mrk_fail(MRK_E_INVAL, "query %u: bad tree", qi);
  for (int i = 0; i < q.n_nodes; ++i) // (ExtHit_t::m_uQuerypos is a WORD, sphinxint.h:733; the arithmetic on positions below assumes as much)
    if (q.nodes[i].op == MRK_OP_TERM && (q.nodes[i].atom_pos < 0 || q.nodes[i].atom_pos > 0xFFFF))
      return mrk_fail(MRK_E_INVAL, "query %u: query position %d of node %d", qi, q.nodes[i].atom_pos, i);
  if (q.max_matches <= 0 || q.max_matches > MRK_MAX_K)
    return mrk_fail(MRK_E_UNSUPPORTED, "query %u: max_matches %d outside 1..%d", qi, q.max_matches, MRK_MAX_K);
  // cutoff (MatchExtended, sphinx.cpp:12197-12199, 12261-12267): the sorter's Push() never says no (sphinxsort.cpp:722-759), so the
  // scan stops after the first `cutoff` rows that got as far as the sorter -- the caller found the last of them with a probe
  // launch (cutoff_probe, mrk_host.cpp) and hands it down as rowid_max: rows past it never reach the ranker
  if (q.cutoff > 0 && !use_packed) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: cutoff runs on the packed path only", qi);
  if (q.cutoff > MRK_MAX_K) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: cutoff %d (device path: <= %d)", qi, q.cutoff, MRK_MAX_K);
  if (q.cutoff > 0 && q.n_weight_filters > 0)
    return mrk_fail(MRK_E_UNSUPPORTED, "query %u: cutoff next to a weight filter (which rows count depends on their weights)", qi);
  const bool filtered = q.n_filters > 0 || rowid_max != 0xFFFFFFFFu;

  // The specialised paths first; a shape they decline goes to the generic per-doc evaluator (mrk_keval.h) when the segment
  // has what it reads (packed doclists + hit references), else the decline stands.
  PlanTree T;
  GenBuild* Gp = nullptr; // (built only for the shapes that go to the generic evaluator: 0.8 KB to clear)
  struct GenHolder {
    alignas(GenBuild) unsigned char raw[sizeof(GenBuild)];
    bool live = false;
    ~GenHolder() {
      if (live) reinterpret_cast<GenBuild*>(raw)->~GenBuild();
    }
  } gen_holder;
  int root = -1;
  bool single_word = false, pure_and = false, prox = false;
  uint32_t ranker = 0;
  int n = 0;
  bool T_used = false;
  auto shape = [&](bool gen) -> int {
    if (T_used) T = PlanTree(); // (the first call finds it fresh)
    T_used = true;
    int tree_err = MRK_OK;
    if (gen) {
      if (gen_holder.live) reinterpret_cast<GenBuild*>(gen_holder.raw)->~GenBuild();
      Gp = new (gen_holder.raw) GenBuild();
      gen_holder.live = true;
      GenBuild& G = *Gp;
      root = build_gen(seg, q, q.root, T, G, qi, 0, tree_err);
      if (root >= 0 && G.overflow) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: tree too large for the device path", qi);
      if (root >= 0) {
        T.gen = T.force_tree = true;
        root = G.plan[root];
      }
    } else
      root = build_tree(seg, q, q.root, T, qi, 0, true, tree_err);
    if (root < 0) return tree_err;
    if (T.kws.overflow || T.nodes.overflow || T.atoms.overflow)
      return mrk_fail(MRK_E_UNSUPPORTED, "query %u: tree too large for the device path", qi);
    n = (int)T.kws.size();
    if (n > MRK_MAX_AND_TERMS) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d keywords (device path: <= %d)", qi, n, MRK_MAX_AND_TERMS);
    if (T.nodes.size() > 16) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: tree too large for the device path", qi);
    single_word = T.nodes.size() == 1 && T.nodes[0].op == PN_TERM; // XQQuery_t::m_bSingleWord
    pure_and = !T.force_tree; // single keyword or one ExtMultiAnd_T: the kernel's N-way AND loop, no program
    for (const PlanNode& pn : T.nodes) pure_and &= pn.op == PN_TERM || pn.op == PN_AND;
    if (pure_and && (q.nodes[q.root].op == MRK_OP_AND || q.nodes[q.root].op == MRK_OP_PHRASE || q.nodes[q.root].op == MRK_OP_PROXIMITY || q.nodes[q.root].op == MRK_OP_NEAR))
      for (int i = 0; i < q.nodes[q.root].n_children; ++i) pure_and &= q.nodes[q.children[q.nodes[q.root].first_child + i]].op == MRK_OP_TERM;
    if (!pure_and && !use_packed) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: boolean trees run on the packed path only", qi);
    if (!pure_and) { // the device evaluates the program on a TREE_STACK-deep register stack
      int sp = 0, deep = 0;
      for (const PlanNode& pn : T.nodes) {
        sp += (pn.op == PN_TERM || pn.op == PN_QUORUM) ? 1 : (pn.op == PN_PHRASEFIX || pn.op == PN_ORDERFIX) ? 0 : -1;
        deep = std::max(deep, sp);
      }
      if (deep > TREE_STACK) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: tree nests deeper than the device path evaluates", qi);
    }

    prox = false; // a state ranker reads the hit streams
    if (gen && T.gen_nearn) {
      int mq = 0;
      for (const PlanKw& k : T.kws) mq = std::max(mq, k.atom_pos);
      if (filtered || q.cutoff > 0 || seg->dev.dead || mq >= 64)
        return mrk_fail(MRK_E_UNSUPPORTED, "query %u: NEAR over 3+ operands next to filters / cutoff / dead rows / query positions past 63 (every doc of the node must reach the evaluator)", qi);
    }
    if (gen && single_word) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: a single keyword is not a case for the generic evaluator", qi);
    if (gen) {
      if (!use_packed || !seg->dev.pk_hit || seg->total_docs >= (1ull << 31)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: the generic evaluator runs on the packed path, < 2^31 docs per segment", qi);
    } else if (T.quorum && !T.quorum_root && q.ranker != MRK_RANK_NONE && q.ranker != MRK_RANK_BM25)
      return mrk_fail(MRK_E_UNSUPPORTED, "query %u: a quorum below another operator with a hit ranker is not on the device path", qi);
    if (T.ph_leaf && n > MAX_PROX_TERMS)
      return mrk_fail(MRK_E_UNSUPPORTED, "query %u: PHRASE in a tree of %d keywords (device path: <= %d)", qi, n, MAX_PROX_TERMS);
    if (T.notnear) { // decided over the two keywords' hits: the hit-reading kernel, <= 4 hit streams
      if (!use_packed || !seg->dev.pk_hit) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: NOTNEAR runs on the packed path only", qi);
      if (n > MAX_PROX_TERMS) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: NOTNEAR in a query of %d keywords (device path: <= %d)", qi, n, MAX_PROX_TERMS);
      if (T.termpos || T.quorum) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: NOTNEAR next to position modifiers / a quorum node", qi);
      if (seg->total_docs >= (1ull << 31)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: hit path needs < 2^31 docs per segment", qi);
    }
    if (T.termpos) { // whether a keyword holds a doc is decided over its hits: the hit-reading kernel, <= 4 hit streams
      if (!use_packed || !seg->dev.pk_hit) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: position modifiers run on the packed path only", qi);
      if (n > MAX_PROX_TERMS) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: position modifiers in a query of %d keywords (device path: <= %d)", qi, n, MAX_PROX_TERMS);
      if (seg->total_docs >= (1ull << 31)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: hit path needs < 2^31 docs per segment", qi);
      for (const PlanKw& k : T.kws)
        if (k.tp_kind < 0 || k.tp_kind > MRK_TERMPOS_LIMIT || (k.tp_kind == MRK_TERMPOS_LIMIT && k.tp_max <= 0))
          return mrk_fail(MRK_E_INVAL, "query %u: bad position modifier", qi);
    }
    if (T.phrase || T.ph_leaf) {
      if (!use_packed || !seg->dev.pk_hit) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: PHRASE runs on the packed path only", qi);
      if (seg->total_docs >= (1ull << 31)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: hit path needs < 2^31 docs per segment", qi);
    }
    switch (q.ranker) {
      case MRK_RANK_NONE: ranker = MRK_RANK_NONE; break;
      case MRK_RANK_BM25: ranker = MRK_RANK_BM25; break;
      case MRK_RANK_PROXIMITY_BM25:
      case MRK_RANK_PROXIMITY:
        // a single keyword is ranked by ExtRanker_WeightSum_c (sphinxsearch.cpp:4195-4196, 4216-4217)
        if (single_word) {
          ranker = q.ranker == MRK_RANK_PROXIMITY_BM25 ? MRK_RANK_BM25 : MRK_RANK_PROXIMITY;
          if (ranker == MRK_RANK_PROXIMITY && !use_packed)
            return mrk_fail(MRK_E_UNSUPPORTED, "query %u: ranker=proximity runs on the packed path only", qi);
        } else {
          if (!use_packed || !seg->dev.pk_hit)
            return mrk_fail(MRK_E_UNSUPPORTED, "query %u: proximity rankers run on the packed path only", qi);
          if (n > MAX_PROX_TERMS && !gen)
            return mrk_fail(MRK_E_UNSUPPORTED, "query %u: proximity over %d keywords (device path: <= %d)", qi, n, MAX_PROX_TERMS);
          if (seg->total_docs >= (1ull << 31)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: proximity path needs < 2^31 docs per segment", qi);
          if (T.multiand3_inner && !gen) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: 3-keyword AND below another operator with a hit ranker", qi);
          ranker = (uint32_t)q.ranker;
          prox = true;
        }
        break;
      case MRK_RANK_WORDCOUNT:
      case MRK_RANK_MATCHANY:
      case MRK_RANK_FIELDMASK:
      case MRK_RANK_SPH04:
        // always ExtRanker_State_T over the hit stream, single keyword or not (sphinxsearch.cpp:4214-4236)
        if (!use_packed || !seg->dev.pk_hit)
          return mrk_fail(MRK_E_UNSUPPORTED, "query %u: hit rankers run on the packed path only", qi);
        if (n > MAX_PROX_TERMS && !gen)
          return mrk_fail(MRK_E_UNSUPPORTED, "query %u: hit ranker over %d keywords (device path: <= %d)", qi, n, MAX_PROX_TERMS);
        if (seg->total_docs >= (1ull << 31)) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: hit path needs < 2^31 docs per segment", qi);
        if (T.multiand3_inner && !gen) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: 3-keyword AND below another operator with a hit ranker", qi);
        ranker = (uint32_t)q.ranker;
        prox = true;
        break;
      default: return mrk_fail(MRK_E_UNSUPPORTED, "query %u: ranker %d not on the device path", qi, q.ranker);
    }

    return MRK_OK;
  };
  {
    int rc = shape(false);
    if (rc == MRK_E_UNSUPPORTED && use_packed && seg->dev.pk_hit) {
      char fast_msg[256];
      snprintf(fast_msg, sizeof fast_msg, "%s", mrk_last_error());
      rc = shape(true);
      if (rc == MRK_E_UNSUPPORTED) { // both declined: say why, the specialised path's reason first
        char gen_msg[256];
        snprintf(gen_msg, sizeof gen_msg, "%s", mrk_last_error());
        return mrk_fail(MRK_E_UNSUPPORTED, "%s; generic evaluator: %s", fast_msg, gen_msg);
      }
    }
    if (rc != MRK_OK) return rc;
  }

  // attribute filters (EarlyReject): resolved locators over the segment's .spa rows
  dq.n_filters = 0;
  dq.rowid_max = rowid_max;
  if (q.n_filters < 0 || (q.n_filters > 0 && !q.filters)) return mrk_fail(MRK_E_INVAL, "query %u: bad filter list", qi);
  if (q.n_filters > 0) {
    if (!seg->dev.attrs) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: filters need the segment's attribute rows (mrk_segment_set_attrs)", qi);
    if (!use_packed) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: filters run on the packed path only", qi);
    if (q.n_filters > MRK_MAX_FILTERS) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d filters (device path: <= %d)", qi, q.n_filters, MRK_MAX_FILTERS);
    for (int i = 0; i < q.n_filters; ++i) {
      const mrk_filter& f = q.filters[i];
      DevFilter& d = dq.filters[i];
      memset(&d, 0, sizeof d);
      if (f.kind != MRK_FILTER_VALUES && f.kind != MRK_FILTER_RANGE && f.kind != MRK_FILTER_FLOATRANGE)
        return mrk_fail(MRK_E_UNSUPPORTED, "query %u: filter kind %d not on the device path", qi, f.kind);
      if (f.kind == MRK_FILTER_FLOATRANGE && f.bit_count != 32) return mrk_fail(MRK_E_INVAL, "query %u: a float filter needs a 32-bit attribute", qi);
      if (f.mva_bits) { // a multi-value attribute in the blob pool
        if (f.mva_bits != 32 && f.mva_bits != 64) return mrk_fail(MRK_E_INVAL, "query %u: MVA width %d", qi, f.mva_bits);
        if (f.kind == MRK_FILTER_FLOATRANGE) return mrk_fail(MRK_E_INVAL, "query %u: a float range over an MVA", qi);
        if (!seg->dev.blobs) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: MVA filters need the segment's blob pool (mrk_segment_set_blobs)", qi);
        if (f.n_blob_attrs < 1 || f.n_blob_attrs > 255 || f.blob_attr_id < 0 || f.blob_attr_id >= f.n_blob_attrs || (uint32_t)f.n_blob_attrs != seg->n_blob_attrs)
          return mrk_fail(MRK_E_INVAL, "query %u: blob attribute %d of %d (the segment's rows hold %u)", qi, f.blob_attr_id, f.n_blob_attrs, seg->n_blob_attrs);
        if (seg->dev.attr_stride < 4) return mrk_fail(MRK_E_INVAL, "query %u: rows of %u dwords hold no blob locator", qi, seg->dev.attr_stride);
        d.kind = (uint32_t)f.kind | (f.exclude ? 1u << 8 : 0) | (f.has_equal_min ? 1u << 9 : 0) | (f.has_equal_max ? 1u << 10 : 0);
        d.mva = (uint32_t)f.mva_bits | (f.mva_all ? 1u << 8 : 0) | ((uint32_t)f.blob_attr_id << 16) | ((uint32_t)f.n_blob_attrs << 24);
        d.lo = f.min_value, d.hi = f.max_value;
        if (f.kind == MRK_FILTER_VALUES) {
          if (f.n_values < 1 || !f.values) return mrk_fail(MRK_E_INVAL, "query %u: values filter without values", qi);
          if (f.n_values > MRK_MAX_FILTER_VALUES) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d filter values (device path: <= %d)", qi, f.n_values, MRK_MAX_FILTER_VALUES);
          d.n_values = (uint32_t)f.n_values;
          for (int k = 0; k < f.n_values; ++k) d.values[k] = f.values[k];
        }
        continue;
      }
      const bool wide = f.bit_count == 64;
      if (f.bit_offset < 0 || f.bit_count < 1 || (!wide && (f.bit_count > 32 || (f.bit_offset & 31) + f.bit_count > 32)) || (wide && (f.bit_offset & 31)) ||
          (uint64_t)(f.bit_offset + f.bit_count) > (uint64_t)seg->dev.attr_stride * 32)
        return mrk_fail(MRK_E_INVAL, "query %u: filter locator %d/%d outside the %u-dword row", qi, f.bit_offset, f.bit_count, seg->dev.attr_stride);
      d.kind = (uint32_t)f.kind | (f.exclude ? 1u << 8 : 0) | (f.has_equal_min ? 1u << 9 : 0) | (f.has_equal_max ? 1u << 10 : 0) |
               (f.open_left ? 1u << 11 : 0) | (f.open_right ? 1u << 12 : 0);
      d.item = (uint32_t)f.bit_offset >> 5;
      d.shift = (uint32_t)f.bit_offset & 31u;
      d.bits = (uint32_t)f.bit_count;
      d.lo = f.min_value, d.hi = f.max_value;
      if (f.kind == MRK_FILTER_FLOATRANGE) { // the bounds travel as their bit patterns
        uint32_t lo_bits, hi_bits;
        memcpy(&lo_bits, &f.fmin, 4), memcpy(&hi_bits, &f.fmax, 4);
        d.lo = lo_bits, d.hi = hi_bits;
      }
      if (f.kind == MRK_FILTER_VALUES) {
        if (f.n_values < 1 || !f.values) return mrk_fail(MRK_E_INVAL, "query %u: values filter without values", qi);
        if (f.n_values > MRK_MAX_FILTER_VALUES) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d filter values (device path: <= %d)", qi, f.n_values, MRK_MAX_FILTER_VALUES);
        d.n_values = (uint32_t)f.n_values;
        for (int k = 0; k < f.n_values; ++k) d.values[k] = f.values[k];
      }
    }
    dq.n_filters = (uint32_t)q.n_filters;
  }
  // filters on the match weight (m_pWeightFilter): evaluated where a match's weight is final
  dq.n_wfilters = 0;
  if (q.n_weight_filters < 0 || (q.n_weight_filters > 0 && !q.weight_filters)) return mrk_fail(MRK_E_INVAL, "query %u: bad weight filter list", qi);
  if (q.n_weight_filters > 0) {
    if (!use_packed) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: weight filters run on the packed path only", qi);
    if (q.n_weight_filters > MRK_MAX_FILTERS) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d weight filters (device path: <= %d)", qi, q.n_weight_filters, MRK_MAX_FILTERS);
    for (int i = 0; i < q.n_weight_filters; ++i) {
      const mrk_filter& f = q.weight_filters[i];
      DevFilter& d = dq.wfilters[i];
      memset(&d, 0, sizeof d);
      if (f.kind != MRK_FILTER_VALUES && f.kind != MRK_FILTER_RANGE) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: weight filter kind %d", qi, f.kind);
      d.kind = (uint32_t)f.kind | (f.exclude ? 1u << 8 : 0) | (f.has_equal_min ? 1u << 9 : 0) | (f.has_equal_max ? 1u << 10 : 0);
      d.lo = f.min_value, d.hi = f.max_value;
      if (f.kind == MRK_FILTER_VALUES) {
        if (f.n_values < 1 || !f.values) return mrk_fail(MRK_E_INVAL, "query %u: values filter without values", qi);
        if (f.n_values > MRK_MAX_FILTER_VALUES) return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %d filter values (device path: <= %d)", qi, f.n_values, MRK_MAX_FILTER_VALUES);
        d.n_values = (uint32_t)f.n_values;
        for (int k = 0; k < f.n_values; ++k) d.values[k] = f.values[k];
      }
    }
    dq.n_wfilters = (uint32_t)q.n_weight_filters;
  }

  // IDFs: distinct words in GetQwords traversal order (searchnode.cpp:2029-2055, 3276-3286)
  IntVec words;
  for (int i = 0; i < n; ++i) {
    if (T.kws[i].hidden) { // read for its hits only
      T.kws[i].weighted_first = false;
      continue;
    }
    bool seen = false;
    for (int w : words) seen |= T.kws[i].term_id >= 0 && T.kws[w].term_id == T.kws[i].term_id; // words missing from the dictionary are distinct words
    T.kws[i].weighted_first = !seen;
    if (!seen) words.push_back(i);
  }
  int n_visible = 0;
  for (const PlanKw& k : T.kws) n_visible += k.hidden ? 0 : 1;
  const bool got_dupes = (int)words.size() != n_visible; // HasQwordDupes: proximity rankers switch to their HANDLE_DUPES update
  const int64_t total_docs = q.total_docs_override > 0 ? q.total_docs_override : (int64_t)seg->total_docs;
  for (int w : words) {
    PlanKw& t = T.kws[w];
    int64_t term_docs = t.docs;
    if (q.local_docs && q.local_docs[t.node] >= 0) term_docs = q.local_docs[t.node];
    t.idf = mrk_idf(term_docs, total_docs, q.plain_idf, q.normalized_tfidf, (int)words.size(), t.boost);
  }

  dq.ranker = ranker;
  dq.n_qwords = (uint32_t)words.size(); // ExtRanker_c::m_iQwords (sphinxsearch.cpp:730-731)
  dq.max_qpos = 0;                      // ... m_iMaxQpos = GetQwords() (:4294-4296, 4372)
  {
    // ... over the keywords the query does not EXCLUDE: TagExcluded (sphinx.cpp:15107-15129) marks the words on the right of an
    // ANDNOT, toggling with every nesting, and an excluded word's GetQwords() answers -1 (searchnode.cpp:2039, 2053).  (The tree
    // was walked by build_tree / build_gen above: it is a tree of at most PLAN_CAP nodes.)
    std::vector<uint8_t> ex((size_t)q.n_nodes, 0);
    struct Walk {
      const mrk_query& q;
      std::vector<uint8_t>& ex;
      void go(int32_t ni, bool neg, int depth) {
        if (ni < 0 || ni >= q.n_nodes || depth > 16) return;
        const mrk_node& nd = q.nodes[ni];
        if (nd.op == MRK_OP_TERM) {
          ex[(size_t)ni] = neg ? 1 : 0;
          return;
        }
        if (nd.n_children < 0 || nd.first_child < 0) return;
        for (int i = 0; i < nd.n_children; ++i) go(q.children[nd.first_child + i], (nd.op == MRK_OP_ANDNOT && i == 1) ? !neg : neg, depth + 1);
      }
    } walk{q, ex};
    walk.go(q.root, false, 0);
    for (const PlanKw& k : T.kws)
      if (!k.hidden && !(k.node >= 0 && k.node < q.n_nodes && ex[(size_t)k.node])) dq.max_qpos = std::max<uint32_t>(dq.max_qpos, (uint32_t)std::max(k.atom_pos, 0));
  }
  dq.k = (uint32_t)q.max_matches;
  dq.n_weights = seg->n_fields;
  dq.index_weight = (uint32_t)(q.index_weight ? q.index_weight : 1);
  for (uint32_t f = 0; f < 32; ++f)
    dq.weights[f] = (q.field_weights && (int)f < q.n_weights) ? q.field_weights[f] : 1; // BindWeights default

  // ---- passes: one per driver keyword of the tree's candidate cover
  IntVec cover;
  if (pure_and)
    cover.push_back(0); // kws are already in ExtMultiAnd_T node order: the rarest keyword drives
  else
    cover_of(T, root, cover);
  {
    IntVec uniq;
    for (int k : cover)
      if (std::find(uniq.begin(), uniq.end(), k) == uniq.end()) uniq.push_back(k);
    cover = uniq;
  }
  if ((int)cover.size() > MAX_PASSES)
    return mrk_fail(MRK_E_UNSUPPORTED, "query %u: %zu driver keywords (device path: <= %d)", qi, cover.size(), MAX_PASSES);
  const uint32_t req = pure_and ? (n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u) : required_of(T, root);
  bool empty = false;
  for (int k = 0; k < n; ++k)
    if ((req >> k & 1u) && !T.kws[k].docs) empty = true; // a required keyword without postings (searchnode.cpp:2922)

  uint64_t bytes = 0, pbytes = 0;
  for (int k = 0; k < n; ++k)
    if (T.kws[k].docs) {
      bytes += seg->terms[T.kws[k].term_id].doclist_len;
      pbytes += seg->terms[T.kws[k].term_id].packed_bytes;
    }

  // pruning histogram geometry (packed path): bins must be monotone in the sorter's order
  dq.bin_mode = BIN_WEIGHT;
  dq.bin_lo = INT32_MIN;
  dq.bin_shift = 31; // fallback: (almost) no pruning, always correct
  if (ranker == MRK_RANK_NONE) {
    // all weights equal: order is rowid ascending => bin on the (global) rowid
    dq.bin_mode = BIN_ROWID;
    const uint64_t max_row = (uint64_t)seg->dev.rowid_base + (seg->total_docs ? seg->total_docs : 1);
    uint32_t sh = 0;
    while (sh < 31 && (max_row >> sh) >= (uint64_t)NBINS) ++sh;
    if (max_row > 0xFFFFFFFFull) sh = 22;
    dq.bin_shift = sh;
    dq.bin_lo = 0;
  } else {
    // weight = ((int)((sum tfidf + 0.5f) * 1000) + rank * 1000) * index_weight; any keyword may be absent
    double lo = 0.0, hi = 0.0;
    for (int i = 0; i < n; ++i) {
      const double idf = T.kws[i].weighted_first ? T.kws[i].idf : 0.0;
      const double a0 = idf * (1.0 / 2.2), a1 = idf;
      lo += std::min(0.0, std::min(a0, a1));
      hi += std::max(0.0, std::max(a0, a1));
      if (pure_and) { // every keyword contributes
        lo += std::min(a0, a1) - std::min(0.0, std::min(a0, a1));
        hi += std::max(a0, a1) - std::max(0.0, std::max(a0, a1));
      }
    }
    // (a boost of inf / NaN makes the weights meaningless in the reference too; the estimate just must stay defined)
    lo = std::isfinite(lo) ? std::max(lo, -1e12) : -1e12;
    hi = std::isfinite(hi) ? std::min(hi, 1e12) : 1e12;
    const int64_t bm_lo = (int64_t)floor((lo + 0.5) * 1000.0) - 2, bm_hi = (int64_t)ceil((hi + 0.5) * 1000.0) + 2;
    int64_t rmin = INT64_MAX, rmax = INT64_MIN;
    const uint32_t nwf = std::min<uint32_t>(dq.n_weights, 8u);
    if (prox) {
      // sum_f LCS[f] * w[f] with 0 <= LCS[f] <= number of keywords (hit weight 1, unique keywords)
      // (a phrase occurrence weighs its word count; back-to-back occurrences can add up -- beyond 2n the bins clamp)
      // The other state rankers: SPH04 4*LCS + 2 + 1 per field; WORDCOUNT one field weight per hit (unbounded:
      // 32 hits per keyword span the bins, more clamp); MATCHANY bits + (LCS-1) * K, K = sum(w) * words;
      // FIELDMASK the mask itself.  Bounds only shape the pruning bins -- values outside clamp to the edge bins.
      rmin = rmax = 0;
      int64_t top = (T.phrase || T.ph_leaf || T.gen) ? 2 * n : n;
      if (ranker == MRK_RANK_SPH04) top = 4 * top + 3;
      if (ranker == MRK_RANK_WORDCOUNT) top = 32 * n;
      if (ranker == MRK_RANK_MATCHANY) {
        int64_t k = 0;
        for (uint32_t f = 0; f < nwf; ++f) k += dq.weights[f];
        top = sat_add(n, sat_mul(top, std::llabs(sat_mul(k, (int64_t)words.size()))));
      }
      for (uint32_t f = 0; f < nwf; ++f) {
        const int64_t v = sat_mul(top, dq.weights[f]);
        rmin = sat_add(rmin, ranker == MRK_RANK_MATCHANY ? -std::llabs(v) : std::min<int64_t>(0, v));
        rmax = sat_add(rmax, ranker == MRK_RANK_MATCHANY ? std::llabs(v) : std::max<int64_t>(0, v));
      }
      if (ranker == MRK_RANK_FIELDMASK) rmin = 0, rmax = (1ll << nwf) - 1;
    } else
      for (uint32_t m = 0; m < (1u << nwf); ++m) { // (every mask of the segment's fields; bits beyond them change nothing)
        int64_t r = 0;
        if (!m)
          r = 1;
        else
          for (uint32_t f = 0; f < nwf; ++f)
            if (m & (1u << f)) r += dq.weights[f];
        rmin = std::min(rmin, r);
        rmax = std::max(rmax, r);
      }
    const bool with_bm = ranker == MRK_RANK_BM25 || ranker == MRK_RANK_PROXIMITY_BM25 || ranker == MRK_RANK_SPH04;
    const int64_t iw = (int32_t)dq.index_weight;
    const int64_t sc = with_bm ? 1000 : 1, b0 = with_bm ? bm_lo : 0, b1 = with_bm ? bm_hi : 0;
    // (128-bit: absurd field weights x an absurd index weight must not overflow the estimate itself)
    const __int128 c[4] = {((__int128)b0 + (__int128)rmin * sc) * iw, ((__int128)b0 + (__int128)rmax * sc) * iw, ((__int128)b1 + (__int128)rmin * sc) * iw,
                           ((__int128)b1 + (__int128)rmax * sc) * iw};
    const __int128 wlo = *std::min_element(c, c + 4), whi = *std::max_element(c, c + 4);
    if (wlo > INT32_MIN && whi < INT32_MAX && std::llabs(rmin) < INT32_MAX / 1000 && std::llabs(rmax) < INT32_MAX / 1000) {
      const uint64_t span = (uint64_t)(whi - wlo) + 1;
      uint32_t sh = 0;
      while (sh < 31 && ((span - 1) >> sh) >= (uint64_t)NBINS) ++sh;
      dq.bin_lo = (int32_t)wlo;
      dq.bin_shift = sh;
    }
  }
  {
    uint64_t cap = 0;
    for (int k : cover) cap += (uint64_t)T.kws[k].docs;
    cap = std::min<uint64_t>(std::max<uint64_t>(cap, 1), (uint64_t)1 << 20);
    dq.cand_cap = (uint32_t)cap;
    dq.cand_off = cand_total;
    cand_total += cap;
  }
  if (empty) {
    dq.n_terms = (uint32_t)n;
    dq.n_items = 0;
    return MRK_OK;
  }
  algo_bytes += bytes;
  dev_bytes += use_packed ? pbytes : bytes;
  prox_out = prox_out || prox || T.phrase || T.ph_leaf || T.termpos || T.notnear || T.gen; // (every generic-path candidate goes through the queue)
  tree_out = tree_out || !pure_and;

  // two dense keywords: the bitmap kernel (mrk_scan_bm.hip) walks 2048-rowid windows instead of blocks
  if (use_packed && pure_and && !T.phrase && n == 2 && !filtered && q.n_weight_filters == 0 && (ranker == MRK_RANK_NONE || ranker == MRK_RANK_BM25) && seg->dev.bm &&
      seg->ctx->bitmap_inv > 0 && seg->terms[T.kws[0].term_id].bm_off != ~0ull && seg->terms[T.kws[1].term_id].bm_off != ~0ull) {
    dq.n_terms = 2;
    for (int i = 0; i < 2; ++i) fill_term(seg, T.kws[i], dq.t[i]);
    dq.tree_flags = TF_MULTIAND | TF_BITMAP;
    dq.item_first = (uint32_t)items_bm.size();
    const uint64_t nwin = seg->dev.n_windows;
    // bitmaps + tf / field bytes of the docs (one byte each where the segment has the nibble plane, else the attr words)
    const uint64_t bm_bytes = 2 * nwin * 256 + ((uint64_t)dq.t[0].nblocks + dq.t[1].nblocks) * (seg->dev.pk_attr1 ? 128 : 256);
    dev_bytes += bm_bytes - pbytes; // (pbytes was added above)
    // one entry for the whole window range; mrk_batch_submit cuts it once the batch's total is known (a wave's
    // fixed costs -- tables, final publish, atomics on the query's counters -- want long runs of windows)
    DevItem it{};
    it.query = qi;
    it.blk_begin = 0;
    it.blk_end = (uint32_t)nwin;
    items_bm.push_back(it);
    dq.n_items = 1;
    return MRK_OK;
  }

  // A tree whose candidate cover is a common keyword: evaluate it on bitmap words, 2048 rowids per step, instead of
  // walking the cover's docs block by block (mrk_scan_bt.hip).  Needs every keyword unrestricted in fields (a bitmap bit
  // is then "the keyword holds the doc"); sparse keywords are fine, their window words are assembled from a block cursor.
  {
    const uint32_t all_fields = seg->n_fields >= 32 ? 0xFFFFFFFFu : (1u << seg->n_fields) - 1u;
    bool ok = use_packed && seg->dev.bm && seg->ctx->bitmap_inv > 0 && seg->ctx->bt_cover_inv > 0 && !T.gen && !T.phrase && !T.ph_leaf && !T.quorum && !T.order &&
              !T.termpos && !T.notnear && !filtered && q.n_weight_filters == 0 && n <= MAX_PROX_TERMS && seg->total_docs < (1ull << 32) && T.nodes.size() <= 16;
    uint64_t cover_docs = 0;
    for (int k : cover) cover_docs += (uint64_t)T.kws[k].docs;
    // (a pure AND keeps the old bar of 1/32: below it the block walk behind a selective driver -- skiplist seeks, a probe per doc --
    // beats streaming every keyword's bitmap; with 1/1024 the headline's selective x common stratum went from 0.36 to 0.54 ms)
    ok = ok && cover_docs * (uint64_t)(pure_and ? std::min(seg->ctx->bt_cover_inv, 32) : seg->ctx->bt_cover_inv) >= seg->total_docs;
    int n_dense = 0;
    for (int k = 0; ok && k < n; ++k) {
      ok = (T.kws[k].queried32 & all_fields) == all_fields;
      if (T.kws[k].docs && seg->terms[T.kws[k].term_id].bm_off != ~0ull) ++n_dense;
    }
    if (ok && pure_and) { // (the tree program of a pure AND is its left-deep chain: at most two values on the stack)
      int sp = 0, deep = 0;
      for (const PlanNode& pn : T.nodes) sp += pn.op == PN_TERM ? 1 : -1, deep = std::max(deep, sp);
      ok = deep <= TREE_STACK;
    }
    if (ok && n_dense > 0) {
      dq.n_terms = (uint32_t)n;
      for (int i = 0; i < n; ++i) fill_term(seg, T.kws[i], dq.t[i]);
      dq.tree_flags = (pure_and ? TF_MULTIAND : 0) | (got_dupes ? TF_DUPES : 0) | TF_BTREE | (seg->ctx->prox_bound_keywords ? TF_LCS_BY_KEYWORDS : 0);
      dq.n_nodes = (uint32_t)T.nodes.size();
      for (size_t i = 0; i < T.nodes.size(); ++i) {
        const PlanNode& pn = T.nodes[i];
        dq.prog[i] = pn.op | ((uint32_t)(pn.l < 0 ? 0 : pn.l) << 8) | ((uint32_t)(pn.r < 0 ? 0 : pn.r) << 16) | ((uint32_t)(pn.kw < 0 ? 0 : pn.kw) << 24);
      }
      dq.item_first = (uint32_t)items_bm.size();
      const uint64_t nwin = seg->dev.n_windows;
      uint64_t bt_bytes = 0; // bitmaps of the dense keywords + packed blocks of the sparse ones + every keyword's tf / field words
      for (int k = 0; k < n; ++k)
        if (T.kws[k].docs) {
          const HostTerm& h = seg->terms[T.kws[k].term_id];
          bt_bytes += h.bm_off != ~0ull ? nwin * 256 + (uint64_t)h.nblocks * 256 : h.packed_bytes;
        }
      dev_bytes += bt_bytes - pbytes; // (pbytes was added above)
      DevItem it{};
      it.query = qi;
      it.blk_begin = 0;
      it.blk_end = (uint32_t)nwin;
      it.kind = 1;
      items_bm.push_back(it);
      dq.n_items = 1;
      return MRK_OK;
    }
  }

  DevQuery base_copy; // (1.3 KB: only copied when the query runs as several passes)
  if (cover.size() > 1) base_copy = dq;
  const DevQuery& base = base_copy;
  for (size_t p = 0; p < cover.size(); ++p) {
    DevQuery* P = &dq;
    uint32_t pass_index = qi;
    if (p > 0) {
      extra.push_back(base);
      P = &extra.back();
      pass_index = n_queries + (uint32_t)extra.size() - 1;
      P->item_first = (uint32_t)items.size();
    }
    if (T.gen) P->item_first = (uint32_t)items_bm.size(), P->n_items = 0;
    // keyword order of this pass: driver, then required keywords by ascending docs, then the rest
    IntVec order;
    const int drv = cover[p];
    order.push_back(drv);
    if (pure_and)
      for (int k = 1; k < n; ++k) order.push_back(k);
    else {
      IntVec rq, rest;
      for (int k = 0; k < n; ++k)
        if (k != drv) ((req >> k & 1u) ? rq : rest).push_back(k);
      auto by_docs = [&](int a, int b) { return T.kws[a].docs < T.kws[b].docs; };
      std::stable_sort(rq.begin(), rq.end(), by_docs);
      std::stable_sort(rest.begin(), rest.end(), by_docs);
      order.append(rq.begin(), rq.end());
      order.append(rest.begin(), rest.end());
    }
    IntVec slot(n);
    for (int i = 0; i < n; ++i) slot[order[i]] = i;
    P->n_terms = (uint32_t)n;
    for (int i = 0; i < n; ++i) fill_term(seg, T.kws[order[i]], P->t[i]);
    P->req_mask = P->excl_mask = 0;
    P->tree_flags = (T.phrase ? TF_PHRASE : pure_and ? TF_MULTIAND : T.ph_leaf ? TF_PHRASE_LEAF : 0) | (got_dupes ? TF_DUPES : 0) | (T.termpos ? TF_TERMPOS : 0) | (T.order ? TF_ORDER : 0) | (T.notnear ? TF_NOTNEAR : 0) | (T.gen ? TF_GEN : 0) | (T.gen_nearn ? TF_GEN_NEARN : 0);
    if (T.gen) { // the evaluator's program, keyword slots as this pass orders them
      GenProg gp = Gp->prog;
      for (uint32_t i = 0; i < gp.n_nodes; ++i) {
        GenNode& g = gp.nodes[i];
        if (g.kind == GN_TERM) g.kid[0] = (uint8_t)slot[g.kid[0]];
        if (g.kind == GN_MULTIAND || g.kind == GN_QUORUM)
          for (int k = 0; k < g.n_kids; ++k) g.kid[k] = (uint8_t)slot[g.kid[k]];
        if (g.kind == GN_PHRASE || g.kind == GN_PROX)
          for (int k = 0; k < g.n_words; ++k) g.aux[k] = (uint8_t)slot[g.aux[k]];
        if (g.kind == GN_UNIT && g.aux[0] != 0xFF) g.aux[0] = (uint8_t)slot[g.aux[0]];
      }
      P->gen_prog = (uint32_t)gen_progs.size();
      gen_progs.push_back(gp);
    }
    P->nn_a = T.notnear ? (uint32_t)slot[T.nn_a] : 0u, P->nn_b = T.notnear ? (uint32_t)slot[T.nn_b] : 0u, P->nn_dist = (uint32_t)T.nn_dist;
    P->px_dist = (uint32_t)T.px_dist;
    P->qr_mask = P->qr_thr = P->qr_n = 0;
    if (T.quorum) {
      // m_dChildren over time: query-position order; a keyword leaves (RemoveFast: the last one takes its place) once the
      // doc it sits on was its last -- keywords without docs right at the warmup (searchnode.cpp:4468-4483, 4517-4537)
      P->qr_thr = (uint32_t)T.q_thr;
      if (T.quorum_root && prox) P->tree_flags |= TF_QUORUM_HITS;
      int list[QUORUM_EVENTS], ln = T.q_n;
      for (int i = 0; i < ln; ++i) list[i] = T.q_kw0 + i, P->qr_mask |= 1u << slot[T.q_kw0 + i];
      auto pack_order = [&]() {
        uint32_t o = 0xFFFFFFFFu;
        for (int i = ln - 1; i >= 0; --i) o = (o << 4) | (uint32_t)slot[list[i]];
        return o;
      };
      auto last_of = [&](int k) -> int64_t { return T.kws[k].docs ? (int64_t)seg->terms[T.kws[k].term_id].last_rowid : -1; };
      for (int i = 0; i < ln; ++i) // warmup: keywords that hold no doc at all
        if (last_of(list[i]) < 0) {
          list[i] = list[--ln];
          --i;
        }
      P->qr_ord[0] = pack_order();
      while (ln > 0 && P->qr_n < (uint32_t)QUORUM_EVENTS) {
        int64_t r = INT64_MAX;
        for (int i = 0; i < ln; ++i) r = std::min(r, last_of(list[i]));
        for (int i = 0; i < ln; ++i)
          if (last_of(list[i]) == r) {
            list[i] = list[--ln];
            --i;
          }
        P->qr_row[P->qr_n] = (uint32_t)r;
        P->qr_ord[++P->qr_n] = pack_order();
      }
    }
    P->ph_mask = 0;
    for (int k = 0; k < T.ph_n; ++k) P->ph_mask |= 1u << slot[T.ph_kw0 + k];
    for (size_t i = 0; i < T.atoms.size(); ++i) P->ph_atoms[i] = (uint32_t)T.atoms[i];
    {
      for (int k = 0; k < n; ++k)
        if (req >> k & 1u) P->req_mask |= 1u << slot[k];
      for (size_t e = 0; e < p; ++e) P->excl_mask |= 1u << slot[cover[e]];
      P->n_nodes = (uint32_t)T.nodes.size();
      for (size_t i = 0; i < T.nodes.size(); ++i) {
        const PlanNode& pn = T.nodes[i];
        P->prog[i] = pn.op | ((uint32_t)(pn.l < 0 ? 0 : pn.l) << 8) | ((uint32_t)(pn.r < 0 ? 0 : pn.r) << 16) |
                     ((uint32_t)(pn.kw < 0 ? 0 : slot[pn.kw]) << 24);
      }
    }
    // A root PHRASE / PROXIMITY whose words are common: the AND of its words -- the candidates the word state machine has to look
    // at -- comes off the doc-set bitmaps, 8192 rowids per step (scan_bt_kernel), instead of the rarest word's blocks one by one
    // with a probe per doc and word; the candidates travel through the same queue to the same hit pass (rank_kernel<1>).
    // Config 5's phrase fifth spent 20 of its 36 ms per launch in the block walk.
    if (T.phrase && p == 0 && cover.size() == 1 && use_packed && seg->dev.bm && seg->ctx->bitmap_inv > 0 && seg->ctx->bt_cover_inv > 0 && seg->ctx->bt_phrase &&
        pure_and && !got_dupes && !T.gen && !T.termpos && !T.notnear && !T.order && !T.quorum && !filtered && q.n_weight_filters == 0 && n >= 2 && n <= MAX_PROX_TERMS &&
        seg->total_docs < (1ull << 32) && T.nodes.size() <= 16 && (uint64_t)T.kws[cover[0]].docs * (uint64_t)seg->ctx->bt_cover_inv >= seg->total_docs) {
      const uint32_t all_fields = seg->n_fields >= 32 ? 0xFFFFFFFFu : (1u << seg->n_fields) - 1u;
      bool ok = true;
      int n_dense = 0, sp = 0, deep = 0;
      for (int k = 0; k < n; ++k) {
        ok = ok && T.kws[k].docs && (T.kws[k].queried32 & all_fields) == all_fields;
        if (T.kws[k].docs && seg->terms[T.kws[k].term_id].bm_off != ~0ull) ++n_dense;
      }
      for (const PlanNode& pn : T.nodes) sp += pn.op == PN_TERM ? 1 : -1, deep = std::max(deep, sp);
      if (ok && n_dense > 0 && deep <= TREE_STACK) {
        P->tree_flags |= TF_BTREE | TF_MULTIAND;
        P->item_first = (uint32_t)items_bm.size();
        const uint64_t nwin = seg->dev.n_windows;
        uint64_t bt_bytes = 0;
        for (int k = 0; k < n; ++k) {
          const HostTerm& h = seg->terms[T.kws[k].term_id];
          bt_bytes += h.bm_off != ~0ull ? nwin * 256 + (uint64_t)h.nblocks * 256 : h.packed_bytes;
        }
        dev_bytes += bt_bytes - pbytes; // (pbytes was added above)
        DevItem it{};
        it.query = pass_index;
        it.blk_begin = 0;
        it.blk_end = (uint32_t)nwin;
        it.kind = 1;
        items_bm.push_back(it);
        P->n_items = 1;
        continue;
      }
    }
    // work items: contiguous ranges of driver-term blocks, ~item_bytes of doclist each
    const uint32_t nb0 = P->t[0].nblocks;
    if (nb0) {
      const double per_block = (double)(use_packed ? pbytes : bytes) / (double)nb0 / (double)cover.size();
      uint64_t bpi = (uint64_t)((double)item_bytes / std::max(per_block, 1.0));
      bpi = std::max<uint64_t>(T0_BLOCKS, (bpi / T0_BLOCKS) * T0_BLOCKS);
      for (uint64_t b = 0; b < nb0; b += bpi) {
        DevItem it{};
        it.query = pass_index;
        it.blk_begin = (uint32_t)b;
        it.blk_end = (uint32_t)std::min<uint64_t>(nb0, b + bpi);
        if (T.gen) { // its own launch (the scan instance that hands over a reference per keyword), behind the block items
          it.kind = 2;
          items_bm.push_back(it);
          ++P->n_items;
        } else
          items.push_back(it);
      }
    }
    if (!T.gen) P->n_items = (uint32_t)items.size() - P->item_first;
  }
  return MRK_OK;
}

// ----------------------------------------------------------------------------------------
