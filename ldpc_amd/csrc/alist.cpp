// alist.cpp -- MacKay alist loader for libldpc_hip (host, C++17).
//
// Replaces the reference's initdec() family (test_code_min_sum_B.c:210-286,
// test_code_B.c:235-314, GFq_LDPC.c:469-535, GFq_LDPC_NTT.c:547-639).  Those
// are fscanf token streams that disagree with each other on whether the
// column lists are padded to cmax; this loader is line-based first (n column
// lines then m row lines, any padding) and falls back to a token stream when
// the file is not laid out one list per line.  Like the reference only the row
// lists define the graph; the header weights are checked against them.
#include "graph.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/ldpc_hip.h"

namespace {

struct Tokens {
  std::vector<long> val;
  std::vector<int> line;   // line number of each token (only non-empty lines are counted)
  int nlines = 0;
};

bool tokenize(const std::string &text, Tokens &t, std::string &err) {
  const char *p = text.c_str();
  const char *end = p + text.size();
  int line = 0;
  bool line_has_token = false;
  while (p < end) {
    char ch = *p;
    if (ch == '\n') {
      if (line_has_token) line++;
      line_has_token = false;
      p++;
    } else if (ch == ' ' || ch == '\t' || ch == '\r' || ch == ',' || ch == '\0') {
      p++;
    } else {
      char *q = nullptr;
      errno = 0;
      long v = strtol(p, &q, 10);
      if (q == p || errno) {
        err = "non-numeric token in alist";
        return false;
      }
      t.val.push_back(v);
      t.line.push_back(line);
      line_has_token = true;
      p = q;
    }
  }
  t.nlines = line + (line_has_token ? 1 : 0);
  return true;
}

}  // namespace

int ldpc_finish_code(ldpc_code &c, std::string &err) {
  const int E = c.E();
  if (c.n <= 0 || c.m <= 0 || (int)c.row_ptr.size() != c.m + 1 || c.row_ptr[c.m] != E) {
    err = "inconsistent CSR arrays";
    return LDPC_ERR_FORMAT;
  }
  if (c.row_ptr[0] != 0) { err = "row_ptr[0] is not 0"; return LDPC_ERR_FORMAT; }
  if ((int)c.col_idx.size() != E || (!c.hval.empty() && (int)c.hval.size() != E)) {
    err = "inconsistent CSR arrays";
    return LDPC_ERR_FORMAT;
  }
  if (c.hval.empty()) c.hval.assign(E, 1);
  std::vector<int32_t> deg(c.n, 0);
  c.rmax = 0;
  c.min_row_weight = E;
  for (int j = 0; j < c.m; j++) {
    int w = c.row_ptr[j + 1] - c.row_ptr[j];
    if (w < 0) { err = "row_ptr not monotone"; return LDPC_ERR_FORMAT; }
    if (w > c.rmax) c.rmax = w;
    if (w < c.min_row_weight) c.min_row_weight = w;
  }
  for (int e = 0; e < E; e++) {
    int v = c.col_idx[e];
    if (v < 0 || v >= c.n) { err = "column index out of range"; return LDPC_ERR_FORMAT; }
    if (c.q > 2 && (c.hval[e] <= 0 || c.hval[e] >= c.q)) {
      err = "GF coefficient out of range";
      return LDPC_ERR_FORMAT;
    }
    deg[v]++;
  }
  // a variable twice in one row is not a parity-check matrix entry (over GF(2) the two would
  // cancel); the kernels' slot tables match edges by position and would decode another graph
  {
    std::vector<int32_t> last(c.n, -1);
    for (int j = 0; j < c.m; j++)
      for (int e = c.row_ptr[j]; e < c.row_ptr[j + 1]; e++) {
        if (last[c.col_idx[e]] == j) {
          err = "row " + std::to_string(j) + " holds variable " + std::to_string(c.col_idx[e]) + " twice";
          return LDPC_ERR_FORMAT;
        }
        last[c.col_idx[e]] = j;
      }
  }
  c.col_ptr.assign(c.n + 1, 0);
  c.cmax = 0;
  for (int v = 0; v < c.n; v++) {
    c.col_ptr[v + 1] = c.col_ptr[v] + deg[v];
    if (deg[v] > c.cmax) c.cmax = deg[v];
  }
  // slot order = ascending edge id = order of appearance scanning rows upward
  // (test_code_min_sum_B.c:258-262: count[v]++)
  std::vector<int32_t> fill(c.col_ptr.begin(), c.col_ptr.end() - 1);
  c.edge_of.assign(E, 0);
  for (int e = 0; e < E; e++) c.edge_of[fill[c.col_idx[e]]++] = e;
  return LDPC_OK;
}

int ldpc_parse_alist(const char *path, ldpc_code &c, std::string &err) {
  FILE *fp = fopen(path, "rb");
  if (!fp) {
    err = std::string("cannot open ") + path;   // reference: exit(-2)
    return LDPC_ERR_IO;
  }
  std::string text;
  char buf[1 << 16];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
  fclose(fp);

  Tokens t;
  if (!tokenize(text, t, err)) return LDPC_ERR_FORMAT;
  const size_t T = t.val.size();
  if (T < 4) { err = "alist too short"; return LDPC_ERR_FORMAT; }

  // header line: "n m" (binary) or "n m q" (GFq_LDPC_NTT.c:557-563)
  size_t pos = 0;
  int hdr = 0;
  while (hdr < (int)T && t.line[hdr] == 0) hdr++;
  bool binary;
  long cmax_h, rmax_h;
  if (hdr == 2 || hdr == 3) {
    binary = hdr == 2;
    c.n = (int)t.val[0];
    c.m = (int)t.val[1];
    c.q = binary ? 2 : (int)t.val[2];
    pos = hdr;
  } else if (hdr >= 4) {          // everything on one line: assume binary
    binary = true;
    c.n = (int)t.val[0];
    c.m = (int)t.val[1];
    c.q = 2;
    pos = 2;
  } else {
    err = "bad alist header";
    return LDPC_ERR_FORMAT;
  }
  if (c.n <= 0 || c.m <= 0 || c.q < 2 || c.q > 256) { err = "bad alist dimensions"; return LDPC_ERR_FORMAT; }
  if (pos + 2 + (size_t)c.n + c.m > T) { err = "alist truncated"; return LDPC_ERR_FORMAT; }
  cmax_h = t.val[pos++];
  rmax_h = t.val[pos++];
  std::vector<long> cw(c.n), rw(c.m);
  long sum_cw = 0, sum_rw = 0;
  for (int i = 0; i < c.n; i++) { cw[i] = t.val[pos++]; sum_cw += cw[i]; if (cw[i] < 0 || cw[i] > cmax_h) { err = "bad column weight"; return LDPC_ERR_FORMAT; } }
  for (int j = 0; j < c.m; j++) { rw[j] = t.val[pos++]; sum_rw += rw[j]; if (rw[j] < 0 || rw[j] > rmax_h) { err = "bad row weight"; return LDPC_ERR_FORMAT; } }

  const int w = binary ? 1 : 2;
  const size_t R = T - pos;
  // where does each row list start, and how many tokens does it own?
  std::vector<size_t> row_start(c.m), row_len(c.m);
  bool ok = false;
  int lines_left = pos < T ? t.nlines - t.line[pos] : 0;
  bool starts_line = pos < T && (pos == 0 || t.line[pos] != t.line[pos - 1]);
  if (starts_line && lines_left == c.n + c.m) {
    // one list per line; column lines skipped
    size_t p = pos;
    int line0 = t.line[pos];
    for (int l = 0; l < c.n + c.m; l++) {
      size_t s = p;
      while (p < T && t.line[p] == line0 + l) p++;
      if (l >= c.n) { row_start[l - c.n] = s; row_len[l - c.n] = p - s; }
    }
    ok = true;
  } else {
    // token stream: columns padded (cmax) or not, rows padded (rmax) or not
    for (int cp = 1; cp >= 0 && !ok; cp--)
      for (int rp = 1; rp >= 0 && !ok; rp--) {
        size_t need = (size_t)w * ((cp ? (size_t)c.n * cmax_h : (size_t)sum_cw) +
                                   (rp ? (size_t)c.m * rmax_h : (size_t)sum_rw));
        if (need != R) continue;
        size_t p = pos + (size_t)w * (cp ? (size_t)c.n * cmax_h : (size_t)sum_cw);
        for (int j = 0; j < c.m; j++) {
          row_start[j] = p;
          row_len[j] = (size_t)w * (rp ? (size_t)rmax_h : (size_t)rw[j]);
          p += row_len[j];
        }
        ok = true;
      }
  }
  if (!ok) { err = "alist body does not match its header (neither one list per line nor a padded/unpadded token stream)"; return LDPC_ERR_FORMAT; }

  c.row_ptr.assign(c.m + 1, 0);
  c.col_idx.clear();
  c.hval.clear();
  c.col_idx.reserve(sum_rw);
  c.hval.reserve(sum_rw);
  std::vector<long> count(c.n, 0);
  for (int j = 0; j < c.m; j++) {
    if (row_len[j] < (size_t)w * rw[j]) { err = "row list shorter than its weight"; return LDPC_ERR_FORMAT; }
    size_t p = row_start[j];
    for (long i = 0; i < rw[j]; i++) {
      long v = t.val[p++];
      long h = 1;
      if (!binary) h = t.val[p++];
      if (v < 1 || v > c.n) { err = "row list entry out of range"; return LDPC_ERR_FORMAT; }
      c.col_idx.push_back((int32_t)(v - 1));
      c.hval.push_back((int32_t)h);
      count[v - 1]++;
    }
    if (!binary)                                  // fillers must be "0 0" (GFq_LDPC.c:517-524)
      for (size_t e = p; e < row_start[j] + row_len[j]; e++)
        if (t.val[e] != 0) { err = "non-zero filler in a non-binary row list"; return LDPC_ERR_FORMAT; }
    c.row_ptr[j + 1] = (int32_t)c.col_idx.size();
  }
  for (int i = 0; i < c.n; i++)
    if (count[i] != cw[i]) {
      // the reference loops j < col_weight[i] over storage filled from the rows;
      // a mismatch makes it read unset pointers, so refuse the file instead
      err = "column weights in the header do not match the row lists";
      return LDPC_ERR_FORMAT;
    }
  c.source = path;
  return ldpc_finish_code(c, err);
}
