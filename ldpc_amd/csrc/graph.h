// graph.h -- host-side Tanner graph of one code (CSR by check + CSC view).
//
// Edge e = (check j, row slot k) has id row_ptr[j] + k: "check-major" order,
// which is also the order the device message array uses.  The CSC view lists,
// for every variable, its edges in the reference's column-slot order: slot k of
// variable v is the k-th row (ascending scan) containing v
// (test_code_min_sum_B.c:253-264), i.e. ascending edge id.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct ldpc_code {
  int n = 0, m = 0, q = 2;
  int cmax = 0, rmax = 0;            // recomputed from the row lists
  std::vector<int32_t> row_ptr;      // [m+1]
  std::vector<int32_t> col_idx;      // [E] variable of edge
  std::vector<int32_t> hval;         // [E] GF coefficient (1 for binary)
  std::vector<int32_t> col_ptr;      // [n+1]
  std::vector<int32_t> edge_of;      // [E] edge id of (variable, slot)
  int min_row_weight = 0;
  std::string source;                // file name, for messages

  int E() const { return (int)col_idx.size(); }
};

// Parses a MacKay alist file.  Returns LDPC_OK or a negative ldpc error code,
// with a message in `err`.
int ldpc_parse_alist(const char *path, ldpc_code &out, std::string &err);
// Finishes a code whose n, m, q, row_ptr, col_idx, hval are set.
int ldpc_finish_code(ldpc_code &c, std::string &err);
