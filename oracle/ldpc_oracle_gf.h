/* ldpc_oracle_gf.h -- TEST INFRASTRUCTURE ONLY (see ldpc_oracle.h). */
#ifndef LDPC_ORACLE_GF_H
#define LDPC_ORACLE_GF_H

#include "ldpc_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* discrete-log tables of GF(2^p), q = 2^p, 2 <= p <= 8 (GFq_LDPC_NTT.c:25-92) */
void ora_gf_tables(int q, int *logq /*[q]*/, int *expq /*[q-1]*/);
int  ora_gf_mul(int q, int a, int b);
/* enc(): s_j = sum_k x[row_col[j][k]] * H[j][k] over GF(q) (GFq_LDPC.c:299-310) */
void ora_gf_enc(const ora_code *c, int q, const int *x, int *s);

/* initlogexptab2() (GFq_LDPC_NTT.c:196-219): flog int[2^20]; fexp is the
 * CENTRE pointer of unsigned[2^21] (valid indices -2^20 .. 2^20-1). */
void ora_wht_tables(const int **flog, const unsigned int **fexp_centre);
int  ora_wht_float2fix(double x);           /* float2fix, DECI=14 (:177-184) */

/* dec() of GFq_LDPC_NTT.c:437-544.  logfna [n][Q] ints (log2 domain, Q.14).
 * Returns 0 converged, -1 not converged / aborted by the non-convergence
 * heuristic (:533-539), -2 where the reference would index fexp[] out of
 * range or trip its assert.  *iters = loop+1 at exit. */
int ora_dec_gf2p_wht(const ora_code *c, int log2q, const int *logfna, const int *z,
                     int loop_max, int *xhat, int *zhat, int *iters);

/* dec() of GFq_LDPC.c:356-467.  logfna [Q][n] doubles (natural log).
 * Returns 0 converged, 1 not.  post (may be NULL) [Q][n] posterior logprod. */
int ora_dec_gfq_direct(const ora_code *c, int Q, const double *logfna, const int *z,
                       int loop_max, int *xhat, int *zhat, int *iters, double *post);

/* GFq_LDPC():548-558 column normalisation; make_p_sent_given_rec_T :177-197 +
 * assign_llr :199-211. */
void ora_gf_normalise_confusion(int Q, int num_reads, double *p_rec_given_sent);
void ora_gf_assign_llr(int Q, int num_reads, const double *p_rec_given_sent,
                       const int *y, int n, double *logfna /*[Q][n]*/);

#ifdef __cplusplus
}
#endif
/* channel() :213-231: the read drawn for each symbol from the caller's uniform draws */
void ora_gf_channel_draw(int Q, int num_reads, const double *p_rec_given_sent, const int *x,
                         const float *rand_select_in, int n, int *y);

#endif
