// fuzz_parser.cpp -- the query parser (csrc/mrk_query.cpp) under AddressSanitizer + UBSan on the CPU: one query per stdin line, every line either
// parses or is rejected; built and fed by tests/test_query_parser.py::test_parser_under_sanitizers (no GPU, no libmrk.so: the two
// library symbols the parser needs are stubbed here).
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "../../include/mrk.h"
int mrk_fail(int code, const char* fmt, ...) { return code; }
extern "C" int32_t mrk_host_index_find_word(const mrk_host_index*, const char*, int32_t) { return -1; }
int main() {
  char buf[4096];
  const char* fields[2] = {"title", "body"};
  int ok = 0, bad = 0;
  while (fgets(buf, sizeof buf, stdin)) {
    size_t n = strlen(buf);
    if (n && buf[n - 1] == '\n') buf[n - 1] = 0;
    mrk_parsed_query* pq = nullptr;
    // (an allocation of the query's exact size: a lexer that looks past the terminator runs into ASan's red zone)
    const std::string exact(buf);
    char* text = (char*)malloc(exact.size() + 1);
    memcpy(text, exact.c_str(), exact.size() + 1);
    int rc = mrk_query_parse(text, fields, 2, 1 + (n & 1), &pq);
    free(text);
    if (rc == 0) { ++ok; for (int i = 0; i < mrk_parsed_n_nodes(pq); ++i) (void)mrk_parsed_keyword(pq, i); mrk_parsed_free(pq); } else ++bad;
  }
  printf("ok %d bad %d\n", ok, bad);
}
