// mrk_query.cpp -- the extended query syntax, text -> mrk_query tree.  Host only.
//
// A hand-written restatement of what the reference's bison grammar + lexer build (sphinxquery.y:57-125; XQParser_t::GetToken,
// AddKeyword, AddOp, FixupNots: sphinxquery.cpp:1180-1560, 1600-1678, 499-562) for the operators the match -> rank -> top-K
// path evaluates:
//
//   keywords (implicit AND)   a | b   a MAYBE b   -a / !a   ( ... )   "a b c"   "a b"~N   "a b c"/N   a << b   a NEAR/N b
//   a SENTENCE b   a PARAGRAPH b   a NOTNEAR/N b   @field  @(f1,f2)  @!field  @!(f1,f2)  @*  @field[N]   ^a   a$   =a   a^1.5   * inside a phrase
//
// Precedence as in the grammar: NOTNEAR binds atoms; '|' and MAYBE bind atoms into an or-list; '-' / '!' negate an or-list;
// '<<' and NEAR/N chain or-lists left to right (equal operator + argument extend the node: AddOp); juxtaposition is AND.
// Atom positions: every keyword token takes the next position in textual order, whatever operator it sits under
// (m_iAtomPos += 1 per token, sphinxquery.cpp:1266-1269); overshort words (shorter than min_word_len) are dropped but keep
// their slot (overshort_step = 1); '*' inside a phrase is a position without a word.  A field limit applies from where it is
// written to the end of the enclosing parentheses.  NOT nodes fold into ANDNOT as FixupNots does.
//
// The tokenizer is the fixture's: lower-cased [a-z0-9_] runs and UTF-8 bytes >= 0x80 are word characters (a real host passes
// text through its own tokenizer / dictionary first; what this file pins is the grammar, the operator tree and the position
// numbering).  Keywords come back as text; mrk_parsed_resolve() looks them up in a dictionary.
#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "mrk_hostindex.h"

int mrk_fail(int code, const char* fmt, ...);

struct mrk_parsed_query {
  std::vector<mrk_node> nodes;
  std::vector<int32_t> children;
  std::vector<std::string> words; // per node: the keyword's text ("" for operators and position-only placeholders)
  int32_t root = -1;
};

namespace {

enum Tok { T_END, T_WORD, T_OR, T_NOT, T_MAYBE, T_LP, T_RP, T_QUOTE, T_BEFORE, T_NEAR, T_NOTNEAR, T_FIELD, T_TILDE, T_SLASH, T_STAR, T_SENTENCE, T_PARAGRAPH };

struct Token {
  Tok t = T_END;
  std::string word;
  int ival = 0;
  double fval = 0;
  bool is_float = false;
  bool start = false, end = false, exact = false; // ^word, word$, =word
  float boost = 1.0f;
  uint32_t mask = 0xFFFFFFFFu; // T_FIELD
  int max_pos = 0;
};

struct PNode { // parse tree
  int op = MRK_OP_TERM; // MRK_OP_*; -1 = NOT (folded away before output)
  std::vector<int> kids;
  std::string word;
  bool placeholder = false; // a position without a word ('*' in a phrase, an overshort word)
  int atom_pos = 0, opt = 0;
  uint32_t mask = 0xFFFFFFFFu;
  int max_pos = 0;
  bool fs = false, fe = false;
  float boost = 1.0f;
};

struct Parser {
  const char* p;
  const char* const* fields;
  uint32_t n_fields, min_word_len;
  std::vector<PNode> N;
  std::string err;
  Token cur;
  int atom = 0;
  bool in_phrase = false, relaxed = false;
  int depth = 0; // open parentheses (bounded: the descent is recursive)
  uint32_t spec_mask = 0xFFFFFFFFu; // the field limit in force
  int spec_max_pos = 0;

  static bool wordch(unsigned char c) { return isalnum(c) || c == '_' || c >= 0x80; }

  bool fail(const std::string& m) {
    if (err.empty()) err = m;
    return false;
  }

  int field_index(const std::string& name) {
    for (uint32_t i = 0; i < n_fields; ++i)
      if (fields && fields[i] && !strcasecmp(fields[i], name.c_str())) return (int)i;
    return -1;
  }

  // '@field', '@(f1,f2)', '@!field', '@!(f1,f2)', '@*', each optionally followed by '[N]' (ParseFields, sphinxquery.cpp:96-330)
  static bool fieldch(unsigned char c) { return isalnum(c) || c == '_' || c == '-'; } // sphIsAlpha

  // returns 1 = a field limit, 0 = not one (the '@' means nothing here and is skipped), -1 = error
  int lex_field(Token& t) {
    ++p; // '@'
    if (!strncmp(p, "@relaxed", 8) && !fieldch((unsigned char)p[8])) { // '@@relaxed': unknown fields are skipped, not errors
      p += 8;
      relaxed = true;
      return 0;
    }
    bool neg = false, block = false;
    uint32_t mask = 0;
    if (*p == '*') {
      ++p;
      t.t = T_FIELD, t.mask = 0xFFFFFFFFu, t.max_pos = 0;
      return 1;
    }
    if (*p == '!') neg = true, ++p;
    if (*p == '(') block = true, ++p;
    if (!fieldch((unsigned char)*p)) return 0;
    for (;;) {
      const char* b = p;
      while (fieldch((unsigned char)*p)) ++p;
      if (p == b) return fail("error parsing field list: invalid field block operator syntax"), -1;
      const int f = field_index(std::string(b, p));
      if (f < 0 && !relaxed) return fail("no field '" + std::string(b, p) + "' found in schema"), -1;
      if (f >= 0 && f < 32) mask |= 1u << f;
      if (!block) break;
      if (*p == ',') {
        ++p;
        continue;
      }
      if (*p == ')') {
        ++p;
        break;
      }
      return fail(*p ? "error parsing field list: invalid character in field block operator" : "error parsing field list: missing closing ')' in field block operator"), -1;
    }
    if (neg) mask = ~mask;
    t.t = T_FIELD;
    t.mask = mask;
    t.max_pos = 0;
    if (*p == '[' && isdigit((unsigned char)p[1])) {
      char* e = nullptr;
      const unsigned long v = strtoul(p + 1, &e, 10);
      if (e && *e == ']') {
        t.max_pos = (int)v;
        p = e + 1;
      }
    }
    return 1;
  }

  bool next() {
    Token t;
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') ++p;
    const unsigned char c = (unsigned char)*p;
    if (!c) {
      cur = t;
      return true;
    }
    if (in_phrase && c != '"' && c != '*' && !wordch(c) && c != '^' && c != '=' && c != '$') { // inside quotes only words, '*' and the closing quote mean anything
      ++p;
      return next();
    }
    if (!in_phrase) {
      if (!strncmp(p, "NOTNEAR/", 8) && isdigit((unsigned char)p[8])) {
        t.t = T_NOTNEAR;
        t.ival = (int)strtol(p + 8, (char**)&p, 10);
        cur = t;
        return true;
      }
      if (!strncmp(p, "NEAR/", 5) && isdigit((unsigned char)p[5])) {
        t.t = T_NEAR;
        t.ival = (int)strtol(p + 5, (char**)&p, 10);
        cur = t;
        return true;
      }
      if ((!strncmp(p, "SENTENCE", 8) && !wordch((unsigned char)p[8])) || (!strncmp(p, "PARAGRAPH", 9) && !wordch((unsigned char)p[9]))) {
        t.t = *p == 'S' ? T_SENTENCE : T_PARAGRAPH; // (capitals only; not a query position: sphinxquery.cpp:1284-1300)
        p += *p == 'S' ? 8 : 9;
        cur = t;
        return true;
      }
      if (!strncmp(p, "MAYBE", 5) && !wordch((unsigned char)p[5])) {
        p += 5;
        t.t = T_MAYBE;
        cur = t;
        return true;
      }
      if (c == '<' && p[1] == '<') {
        p += 2;
        t.t = T_BEFORE;
        cur = t;
        return true;
      }
      if (c == '@') {
        const int r = lex_field(t);
        if (r < 0) return false;
        if (!r) return next();
        cur = t;
        return true;
      }
      if (c == '|' || c == '-' || c == '!' || c == '(' || c == ')' || c == '~' || c == '/') {
        ++p;
        t.t = c == '|' ? T_OR : (c == '-' || c == '!') ? T_NOT : c == '(' ? T_LP : c == ')' ? T_RP : c == '~' ? T_TILDE : T_SLASH;
        if (t.t == T_TILDE || t.t == T_SLASH) { // the argument of "..."~N / "..."/N / "..."/0.5
          char* e = nullptr;
          t.fval = strtod(p, &e);
          if (e == p) return fail("a number is expected after ~ or /");
          t.is_float = memchr(p, '.', (size_t)(e - p)) != nullptr;
          t.ival = (int)t.fval;
          // the number is offered to the tokenizer as well; when it comes back as a keyword-sized token it takes the next
          // query position although it is not a keyword (GetNumber, sphinxquery.cpp:1160-1170)
          size_t digits = 0;
          for (const char* q = p; q < e; ++q) {
            if (isdigit((unsigned char)*q)) {
              ++digits;
              continue;
            }
            if (digits >= min_word_len) break;
            digits = 0;
          }
          if (digits >= min_word_len) ++atom;
          p = e;
        }
        cur = t;
        return true;
      }
    }
    if (c == '"') {
      ++p;
      t.t = T_QUOTE;
      cur = t;
      return true;
    }
    if (c == '*' && in_phrase) {
      ++p;
      t.t = T_STAR;
      cur = t;
      return true;
    }
    // a keyword with its modifiers: ^word  =word  word$  word^1.5
    if (c == '^') t.start = true, ++p;
    if (*p == '=') t.exact = true, ++p;
    if (*p == '^' && !t.start) t.start = true, ++p;
    const char* b = p;
    while (wordch((unsigned char)*p)) ++p;
    if (p == b) { // a character that means nothing here: skip it like the tokenizer would ('^' / '=' at the very end: nothing to skip)
      if (*p) ++p;
      return next();
    }
    t.t = T_WORD;
    t.word.assign(b, p);
    for (char& ch : t.word) ch = (char)tolower((unsigned char)ch);
    if (*p == '$') t.end = true, ++p;
    if (*p == '^' && (isdigit((unsigned char)p[1]) || p[1] == '.')) {
      char* e = nullptr;
      t.boost = strtof(p + 1, &e);
      p = e;
    }
    cur = t;
    return true;
  }

  int add(const PNode& n) {
    N.push_back(n);
    return (int)N.size() - 1;
  }

  // AddKeyword (sphinxquery.cpp:1600-1632): the next atom position; an overshort word keeps its slot but yields no node
  int keyword(const Token& t) {
    ++atom;
    size_t len = 0;
    for (unsigned char ch : t.word) len += (ch & 0xC0) != 0x80; // code points
    PNode n;
    n.op = MRK_OP_TERM;
    n.word = t.exact ? "=" + t.word : t.word; // '=' keyword: the word carries the '=' (sphinxquery.y:115)
    n.atom_pos = atom;
    n.mask = spec_mask;
    n.max_pos = spec_max_pos;
    n.fs = t.start, n.fe = t.end;
    n.boost = t.boost;
    if (len < min_word_len) {
      if (!in_phrase) return -1; // dropped; the position stays used
      n.placeholder = true;
      n.word.clear();
    }
    return add(n);
  }

  // AddOp (sphinxquery.cpp:1634-1678): equal operator + argument extend the left node
  int add_op(int op, int l, int r, int opt = 0) {
    if (l < 0 || r < 0) return l < 0 ? r : l;
    if (N[l].op == op && !N[l].kids.empty() && N[l].opt == opt && N[l].word.empty()) {
      N[l].kids.push_back(r);
      return l;
    }
    PNode n;
    n.op = op;
    n.opt = opt;
    n.mask = N[r].mask; // "it's right (!) spec which is chosen for the resulting node"
    n.kids = {l, r};
    return add(n);
  }

  // '"' phrase '"' [ '~' N | '/' N | '/' F ]
  bool phrase(int& out) {
    in_phrase = true;
    if (!next()) return false;
    PNode n;
    n.op = MRK_OP_PHRASE;
    n.mask = spec_mask;
    std::vector<int> star_at; // '*' = a position without a word; counted when the operator turns out to be a plain phrase
    while (cur.t == T_WORD || cur.t == T_STAR) {
      if (cur.t == T_STAR)
        star_at.push_back((int)n.kids.size());
      else {
        const int k = keyword(cur);
        if (k >= 0 && !N[k].placeholder) n.kids.push_back(k);
      }
      if (!next()) return false;
    }
    if (cur.t != T_QUOTE) return fail("unterminated phrase");
    in_phrase = false;
    if (!next()) return false;
    if (cur.t == T_TILDE) {
      if (cur.is_float || cur.ival < 1) return fail("proximity threshold too low");
      n.op = MRK_OP_PROXIMITY, n.opt = cur.ival;
      if (!n.kids.empty()) atom = N[n.kids.back()].atom_pos + 1; // XQNode_t::FixupAtomPos (sphinxquery.cpp:923-938)
      if (!next()) return false;
    } else if (cur.t == T_SLASH) {
      n.op = MRK_OP_QUORUM;
      if (cur.is_float) {
        // a share of the words: kept as whole percents, resolved as ExtQuorum_c::GetThreshold does (searchnode.cpp:4598-4601)
        const int pct = (int)((float)cur.fval * 100);
        if (pct <= 0 || pct > 100) return fail("quorum threshold out of bounds 0.0 and 1.0f");
        n.opt = (int)floorf(1.0f / 100.0f * (float)pct * (float)n.kids.size() + 0.5f);
      } else {
        if (cur.ival <= 0) return fail("quorum threshold too low");
        n.opt = cur.ival;
      }
      if (!next()) return false;
    } else { // PhraseShiftQpos (sphinxquery.cpp:1701-1740): every '*' moves the words after it one position on
      size_t s = 0;
      int shift = 0;
      for (size_t i = 0; i < n.kids.size(); ++i) {
        while (s < star_at.size() && star_at[s] <= (int)i) ++s, ++shift;
        N[n.kids[i]].atom_pos += shift;
      }
    }
    if (n.kids.empty()) {
      out = -1;
      return true;
    }
    if (n.kids.size() == 1) { // FixupDegenerates (sphinxquery.cpp:311-326): a one-word phrase / proximity / quorum is that word
      out = n.kids[0];
      return true;
    }
    out = add(n);
    return true;
  }

  bool atom_(int& out) {
    if (!primary(out)) return false;
    // sentence: sp_item SENTENCE sp_item | sentence SENTENCE sp_item (sphinxquery.y:117-124); sp_item = a keyword or a plain phrase
    while (cur.t == T_SENTENCE || cur.t == T_PARAGRAPH) {
      const int op = cur.t == T_SENTENCE ? MRK_OP_SENTENCE : MRK_OP_PARAGRAPH;
      auto sp_item = [&](int x) { return x >= 0 && (N[x].op == MRK_OP_TERM || N[x].op == MRK_OP_PHRASE) && N[x].opt == 0; };
      if (out >= 0 && !sp_item(out) && N[out].op != op) return fail("SENTENCE / PARAGRAPH take keywords and phrases");
      if (!next()) return false;
      int r;
      if (!primary(r)) return false;
      if (r >= 0 && !sp_item(r)) return fail("SENTENCE / PARAGRAPH take keywords and phrases");
      out = add_op(op, out, r);
    }
    while (cur.t == T_NOTNEAR) { // atom TOK_NOTNEAR atom, left-associative
      const int dist = cur.ival;
      if (!next()) return false;
      int r;
      if (!primary(r)) return false;
      out = add_op(MRK_OP_NOTNEAR, out, r, dist);
    }
    return true;
  }

  bool primary(int& out) {
    out = -1;
    if (cur.t == T_FIELD) {
      spec_mask = cur.mask, spec_max_pos = cur.max_pos;
      if (!next()) return false;
      if (cur.t == T_END || cur.t == T_RP) return true;
      return primary(out);
    }
    if (cur.t == T_WORD) {
      out = keyword(cur);
      if (!next()) return false;
    } else if (cur.t == T_QUOTE) {
      if (!phrase(out)) return false;
    } else if (cur.t == T_LP) {
      const uint32_t m0 = spec_mask;
      const int p0 = spec_max_pos;
      if (++depth > 64) return fail("query nests too deep");
      if (!next()) return false;
      if (!expr(out)) return false;
      --depth;
      if (cur.t != T_RP) return fail("missing closing parenthesis");
      spec_mask = m0, spec_max_pos = p0; // a field limit ends with its parentheses
      if (!next()) return false;
    } else
      return fail("a keyword, a phrase or '(' is expected");
    return true;
  }

  bool orlist(int& out) {
    if (!atom_(out)) return false;
    while (cur.t == T_OR || cur.t == T_MAYBE) {
      const int op = cur.t == T_OR ? MRK_OP_OR : MRK_OP_MAYBE;
      if (!next()) return false;
      int r;
      if (!atom_(r)) return false;
      out = add_op(op, out, r);
    }
    return true;
  }

  bool orlistf(int& out) {
    if (cur.t == T_FIELD) { // tok_limiter '-' orlist
      spec_mask = cur.mask, spec_max_pos = cur.max_pos;
      if (!next()) return false;
    }
    if (cur.t == T_NOT) {
      if (!next()) return false;
      int x;
      if (!orlist(x)) return false;
      PNode n;
      n.op = -1;
      n.kids = {x};
      out = x < 0 ? -1 : add(n);
      return true;
    }
    return orlist(out);
  }

  bool beforelist(int& out) {
    if (!orlistf(out)) return false;
    while (cur.t == T_BEFORE || cur.t == T_NEAR) {
      const int op = cur.t == T_BEFORE ? MRK_OP_BEFORE : MRK_OP_NEAR, opt = cur.t == T_NEAR ? cur.ival : 0;
      if (!next()) return false;
      int r;
      if (!orlistf(r)) return false;
      out = add_op(op, out, r, opt);
    }
    return true;
  }

  bool expr(int& out) {
    out = -1;
    while (cur.t != T_END && cur.t != T_RP) {
      int r;
      if (!beforelist(r)) return false;
      out = add_op(MRK_OP_AND, out, r);
    }
    return true;
  }

  // FixupNots (sphinxquery.cpp:499-562): AND ( x.., NOT y.. ) -> ANDNOT ( AND ( x.. ), y | OR ( y.. ) )
  bool fixup_nots(int ni) {
    if (ni < 0) return true;
    for (size_t i = 0; i < N[ni].kids.size(); ++i)
      if (!fixup_nots(N[ni].kids[i])) return false;
    if (N[ni].op == -1) return true;
    std::vector<int> nots, rest;
    for (int k : N[ni].kids) (N[k].op == -1 ? nots : rest).push_back(k);
    if (nots.empty()) return true;
    if (rest.empty()) return fail("query is non-computable (node consists of NOT operators only)");
    if (N[ni].op != MRK_OP_AND) return fail("query is non-computable (NOT is not allowed within this operator)");
    int land;
    if (rest.size() == 1)
      land = rest[0];
    else {
      PNode a;
      a.op = MRK_OP_AND;
      a.kids = rest;
      a.mask = N[ni].mask;
      land = add(a);
    }
    int lnot;
    if (nots.size() == 1)
      lnot = N[nots[0]].kids[0];
    else {
      PNode o;
      o.op = MRK_OP_OR;
      for (int k : nots) o.kids.push_back(N[k].kids[0]);
      lnot = add(o);
    }
    N[ni].op = MRK_OP_ANDNOT;
    N[ni].kids = {land, lnot};
    return true;
  }
};

// parse tree -> flat mrk_node[] (post-order), keywords as text
int emit(const Parser& P, int ni, mrk_parsed_query& out) {
  const PNode& n = P.N[ni];
  std::vector<int> kids;
  for (int k : n.kids) kids.push_back(emit(P, k, out));
  mrk_node m;
  memset(&m, 0, sizeof m);
  m.op = n.op;
  m.term_id = -1;
  m.field_mask = n.mask;
  m.boost = n.boost;
  m.opt = n.opt;
  if (n.op == MRK_OP_TERM) {
    m.atom_pos = n.atom_pos;
    m.term_pos = n.max_pos ? MRK_TERMPOS_LIMIT : (n.fs && n.fe) ? MRK_TERMPOS_STARTEND : n.fs ? MRK_TERMPOS_START : n.fe ? MRK_TERMPOS_END : MRK_TERMPOS_NONE;
    m.field_max_pos = n.max_pos;
  } else {
    m.n_children = (int32_t)kids.size();
    m.first_child = (int32_t)out.children.size();
    for (int k : kids) out.children.push_back(k);
  }
  out.nodes.push_back(m);
  out.words.push_back(n.op == MRK_OP_TERM ? n.word : n.op == MRK_OP_SENTENCE ? std::string("\3sentence") : n.op == MRK_OP_PARAGRAPH ? std::string("\3paragraph") : std::string());
  return (int)out.nodes.size() - 1;
}

} // namespace

extern "C" int mrk_query_parse(const char* text, const char* const* field_names, uint32_t n_fields, uint32_t min_word_len, mrk_parsed_query** out) {
  if (!text || !out || (n_fields && !field_names)) return mrk_fail(MRK_E_INVAL, "mrk_query_parse: NULL argument");
  for (uint32_t i = 0; i < n_fields; ++i)
    if (!field_names[i]) return mrk_fail(MRK_E_INVAL, "mrk_query_parse: field name %u is NULL", i);
  *out = nullptr;
  Parser P;
  P.p = text, P.fields = field_names, P.n_fields = n_fields, P.min_word_len = min_word_len ? min_word_len : 1;
  int root = -1;
  try {
    if (!P.next() || !P.expr(root) || (P.cur.t != T_END && !P.fail("unexpected ')'")) || !P.fixup_nots(root))
      return mrk_fail(MRK_E_INVAL, "query parse error: %s", P.err.c_str());
    if (root >= 0 && P.N[root].op == -1) return mrk_fail(MRK_E_INVAL, "query parse error: query is non-computable (single NOT operator)");
    mrk_parsed_query* q = new mrk_parsed_query();
    if (root >= 0) q->root = emit(P, root, *q);
    *out = q;
    return MRK_OK;
  } catch (const std::bad_alloc&) {
    return mrk_fail(MRK_E_NOMEM, "mrk_query_parse: out of memory");
  }
}

// sphTransformExtendedQuery (sphinx.cpp:15345-15359) as every query goes through it before the ranker is built -- the part that is not an
// option: TransformQuorum (:14700-14727: a quorum with threshold 1 becomes the OR of its words), TransformNear (:15049-15105:
// '(a b c) NEAR/N d' becomes 'a NEAR/N b NEAR/N c NEAR/N d' -- every AND group among a NEAR node's operands is replaced by its
// children, in place and in order, again and again until none is left; OR groups and phrases stay operands).  TagExcluded
// (:15107-15129) marks the keywords on the right of an ANDNOT: the planner derives that from the tree itself.  TransformBigrams
// needs an index built with bigram_index (declined at open) and sphOptimizeBoolean is an option (OPTION boolean_simplify, off by
// default): neither is restated.
static void transform_nodes(mrk_parsed_query* q) {
  // the child lists as vectors, rewritten, then laid out again
  const size_t n = q->nodes.size();
  std::vector<std::vector<int32_t>> kids(n);
  for (size_t i = 0; i < n; ++i) {
    const mrk_node& nd = q->nodes[i];
    if (nd.op != MRK_OP_TERM && nd.n_children > 0) kids[i].assign(q->children.begin() + nd.first_child, q->children.begin() + nd.first_child + nd.n_children);
  }
  for (size_t i = 0; i < n; ++i) {
    mrk_node& nd = q->nodes[i];
    if (nd.op == MRK_OP_QUORUM && nd.opt == 1) {
      bool plain = true; // (the reference's quorum node holds words, never nodes: only a quorum over plain keywords is one)
      for (int32_t c : kids[i]) plain = plain && c >= 0 && (size_t)c < n && q->nodes[c].op == MRK_OP_TERM;
      if (plain) nd.op = MRK_OP_OR, nd.opt = 0;
    }
    if (nd.op == MRK_OP_NEAR) {
      for (bool again = true; again;) {
        again = false;
        std::vector<int32_t> flat;
        for (int32_t c : kids[i]) {
          if (c >= 0 && (size_t)c < n && q->nodes[c].op == MRK_OP_AND && !kids[c].empty()) {
            flat.insert(flat.end(), kids[c].begin(), kids[c].end());
            again = true;
          } else
            flat.push_back(c);
        }
        kids[i].swap(flat);
      }
    }
  }
  q->children.clear();
  for (size_t i = 0; i < n; ++i) {
    mrk_node& nd = q->nodes[i];
    if (nd.op == MRK_OP_TERM) continue;
    nd.first_child = (int32_t)q->children.size();
    nd.n_children = (int32_t)kids[i].size();
    q->children.insert(q->children.end(), kids[i].begin(), kids[i].end());
  }
}

extern "C" int mrk_parsed_transform(mrk_parsed_query* q) {
  if (!q) return mrk_fail(MRK_E_INVAL, "mrk_parsed_transform: NULL argument");
  try {
    transform_nodes(q);
  } catch (const std::bad_alloc&) {
    return mrk_fail(MRK_E_NOMEM, "mrk_parsed_transform: out of memory");
  }
  return MRK_OK;
}

extern "C" void mrk_parsed_free(mrk_parsed_query* q) { delete q; }
extern "C" int32_t mrk_parsed_n_nodes(const mrk_parsed_query* q) { return q ? (int32_t)q->nodes.size() : 0; }
extern "C" int32_t mrk_parsed_root(const mrk_parsed_query* q) { return q ? q->root : -1; }
extern "C" const mrk_node* mrk_parsed_nodes(const mrk_parsed_query* q) { return q && !q->nodes.empty() ? q->nodes.data() : nullptr; }
extern "C" const int32_t* mrk_parsed_children(const mrk_parsed_query* q, int32_t* n) {
  if (n) *n = q ? (int32_t)q->children.size() : 0;
  return q && !q->children.empty() ? q->children.data() : nullptr;
}
extern "C" const char* mrk_parsed_keyword(const mrk_parsed_query* q, int32_t node) {
  return (q && node >= 0 && (size_t)node < q->words.size()) ? q->words[(size_t)node].c_str() : nullptr;
}
// dictionary lookup of every keyword (dict=keywords indexes opened from files); unknown words keep term_id -1
extern "C" int mrk_parsed_resolve(mrk_parsed_query* q, const mrk_host_index* h) {
  if (!q || !h) return mrk_fail(MRK_E_INVAL, "mrk_parsed_resolve: NULL argument");
  for (size_t i = 0; i < q->nodes.size(); ++i)
    if (q->nodes[i].op == MRK_OP_TERM || q->nodes[i].op == MRK_OP_SENTENCE || q->nodes[i].op == MRK_OP_PARAGRAPH) q->nodes[i].term_id = mrk_host_index_find_word(h, q->words[i].c_str(), (int32_t)q->words[i].size());
  return MRK_OK;
}
