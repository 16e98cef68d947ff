/*
 * ldpc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, one-frame-at-a-time CPU restatement of the
 * reference decoders' algorithm (mohammedmidhat/LDPC).  It exists so that the
 * HIP path (libldpc_hip) can be checked against something that follows the
 * reference line by line.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product never does.
 *
 * Every function cites the reference file:line it restates.  How each
 * restatement is pinned (golden vectors / oracle/_ref builds of the reference)
 * is listed in oracle/README.md.
 */
#ifndef LDPC_ORACLE_H
#define LDPC_ORACLE_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Graph exactly as the reference loaders leave it in their globals
 * (test_code_min_sum_B.c:16-19 + the locals row_N/col_row/col_N :212-213). */
typedef struct ora_code {
  int n, m, cmax, rmax, q;     /* q = 2 for binary files                     */
  int *col_weight;             /* [n]                                        */
  int *row_weight;             /* [m]                                        */
  int *row_col;                /* [m][rmax]  variable index of (row, slot)   */
  int *row_N;                  /* [m][rmax]  column slot of that edge        */
  int *col_row;                /* [n][cmax]  check index of (col, slot)      */
  int *col_N;                  /* [n][cmax]  row slot of that edge           */
  int *H;                      /* [m][rmax]  GF coefficient (1 for binary)   */
} ora_code;

/* dialect of the column-list skip:
 *   0: skip cmax tokens per column   (test_code_min_sum_B.c:231-236)
 *   1: skip col_weight[i] per column (test_code_B.c:256-261)              */
int  ora_load_alist(const char *path, int dialect, ora_code **out);
void ora_free_code(ora_code *c);

/* enc(): s_j = XOR of y over row j (test_code_min_sum_B.c:69-79).  */
void ora_enc(const ora_code *c, const int *y, int *s);

/* float_to_fix(), INT=8 FRAC=8 (test_code_min_sum_B_fixed.c:43-53). */
double ora_float_to_fix(double v);

/* All decoders: return 0 = target syndrome reached, 1 = not reached.
 * bits[n] = tmp_bit, synd[m] = tmp_s at return, *iters = iterations executed
 * (loop+1 at the successful exit, loop_max on failure, 0 for the fixed
 * variant's "already a codeword" pre-check).  post[n] (may be NULL) receives
 * the posterior `sum` of the last executed iteration.                       */

/* dec(), double min-sum (test_code_min_sum_B.c:137-208). */
int ora_dec_minsum(const ora_code *c, const double *q0, const int *s,
                   int loop_max, int *bits, int *synd, int *iters, double *post);

/* Same arithmetic carried out in IEEE binary32 with the same operation order:
 * the bit-exact oracle for the GPU's MINSUM_F32 kernel.  Not in the reference;
 * derived from ora_dec_minsum by replacing double with float.               */
int ora_dec_minsum_f32(const ora_code *c, const float *q0, const int *s,
                       int loop_max, int *bits, int *synd, int *iters, float *post);

/* dec(), Q8.8 min-sum (test_code_min_sum_B_fixed.c:192-316), including the
 * Verilog trace dump (:245-260, :288-296) when trace != NULL.  circ_size is
 * the reference's global of that name (:40; 128 in the reference).  When
 * trace == NULL the dump loops are skipped altogether.                      */
int ora_dec_minsum_fixed(const ora_code *c, const double *q0, const int *s,
                         int loop_max, int *bits, int *synd, int *iters,
                         double *post, FILE *trace, int circ_size);

/* The same arithmetic on a layered schedule (layers of `layer` consecutive checks; not in the
 * reference -- see the function's header in ldpc_oracle.c).  layer >= m: flooding, identical
 * to ora_dec_minsum_fixed.                                                    */
int ora_dec_minsum_fixed_layered(const ora_code *c, const double *q0, const int *s, int loop_max,
                                 int layer, int *bits, int *synd, int *iters, double *post);

/* dec(), double sum-product (test_code_B_float.c:134-199). */
int ora_dec_spa(const ora_code *c, const double *q0, const int *s,
                int loop_max, int *bits, int *synd, int *iters, double *post);
/* binary32 version of the same (oracle for SPA_F32's structure; libm floats) */
int ora_dec_spa_f32(const ora_code *c, const float *q0, const int *s,
                    int loop_max, int *bits, int *synd, int *iters, float *post);

/* inittab() (test_code_B.c:52-74): fills two int[2^22] tables; the pointers
 * stay owned by the oracle.  float2fix (test_code_B.c:36-43).               */
void ora_inittab(const int **flogtanh, const int **fgallag);
int  ora_float2fix14(double x);
/* dec(), table sum-product on Q8.14 ints (test_code_B.c:171-233).  Returns -2
 * where the reference would have hit one of its assert()s (:78, :86).       */
int ora_dec_spa_table(const ora_code *c, const int *q0, const int *s,
                      int loop_max, int *bits, int *synd, int *iters, int *post);

#ifdef __cplusplus
}
#endif
#endif
