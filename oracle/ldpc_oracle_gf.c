/*
 * ldpc_oracle_gf.c -- TEST INFRASTRUCTURE ONLY (see ldpc_oracle.h).
 *
 * CPU restatement of the two non-binary decoders of mohammedmidhat/LDPC:
 *   - GF(2^p) Walsh-Hadamard-domain integer sum-product (GFq_LDPC_NTT.c)
 *   - GF(q) log-domain forward/backward sum-product     (GFq_LDPC.c)
 * The field order is a run-time argument here (a compile-time macro there).
 */
#include "ldpc_oracle_gf.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ----------------------------------------------------------- GF arithmetic */

/* The reference's logq/expq literals (GFq_LDPC_NTT.c:25-92) are the
 * discrete-log tables of these primitive polynomials (checked by
 * tests/test_gf_tables.py against the literals when the reference tree is
 * present). */
static const int gf_poly[9] = {0, 0, 0x7, 0xB, 0x13, 0x25, 0x43, 0x89, 0x11D};

void ora_gf_tables(int q, int *logq, int *expq)
{
  int p = 0, i, x = 1;
  while ((1 << p) < q) p++;
  for (i = 0; i < q - 1; i++) {
    expq[i] = x;
    logq[x] = i;
    x <<= 1;
    if (x & q) x ^= gf_poly[p];
  }
  logq[0] = 0;
}

typedef struct { int q, pow2, logq[256], expq[256]; } gf_t;

static void gf_init(gf_t *g, int q)
{
  g->q = q;
  g->pow2 = (q & (q - 1)) == 0;
  if (g->pow2 && q > 2) ora_gf_tables(q, g->logq, g->expq);
}

/* GF_mul/GF_add/GF_sub: prime field GFq_LDPC.c:22-24, GF(2^p) :104-113 */
static int gf_mul(const gf_t *g, int a, int b)
{
  if (!g->pow2) return (a * b) % g->q;
  if (g->q == 2) return a & b;
  if (a == 0 || b == 0) return 0;
  if (a == 1) return b;
  if (b == 1) return a;
  return g->expq[(g->logq[a] + g->logq[b]) % (g->q - 1)];
}
static int gf_add(const gf_t *g, int a, int b) { return g->pow2 ? (a ^ b) : (a + b) % g->q; }
static int gf_sub(const gf_t *g, int a, int b) { return g->pow2 ? (a ^ b) : (a - b + g->q * 2) % g->q; }

int ora_gf_mul(int q, int a, int b) { gf_t g; gf_init(&g, q); return gf_mul(&g, a, b); }

/* enc() (GFq_LDPC.c:299-310, GFq_LDPC_NTT.c:367-378) */
void ora_gf_enc(const ora_code *c, int q, const int *x, int *s)
{
  gf_t g;
  int i, j;
  gf_init(&g, q);
  for (j = 0; j < c->m; j++) {
    int sum = 0;
    for (i = 0; i < c->row_weight[j]; i++)
      sum = gf_add(&g, sum, gf_mul(&g, x[c->row_col[j * c->rmax + i]], c->H[j * c->rmax + i]));
    s[j] = sum;
  }
}

static int differs(const int *a, const int *b, int len)
{
  int i, d = 0;
  for (i = 0; i < len; i++) if (a[i] != b[i]) d++;
  return d;
}
/* HamDist() counting differing bits of symbols (GFq_LDPC_NTT.c:253-262) */
static int hamdist_bits(const int *x, const int *y, int len, int q)
{
  int i, j, sum = 0;
  for (i = 0; i < len; i++) {
    int xy = x[i] ^ y[i];
    for (j = 1; j < q; j <<= 1) if (xy & j) sum++;
  }
  return sum;
}

/* ------------------------------------------- GF(2^p) WHT integer decoder */

#define W_INT   6
#define W_DECI  14
#define W_FMUL  (1 << W_DECI)
#define W_PREC  (1.0 / W_FMUL)
#define W_LEVELS (1 << (W_INT + W_DECI))
static int *w_flog;
static unsigned int *w_fexp_base, *w_fexp;

static int w_float2fix(double x)
{
  if (x >= 0) return (int)(x * W_FMUL + 0.5);
  return -(int)((-x) * W_FMUL + 0.5);
}
int ora_wht_float2fix(double x) { return w_float2fix(x); }
static unsigned int w_float2fixu(double x) { return (unsigned int)(x * W_FMUL + 0.5); }
#define w_fix2float(x) ((x) * W_PREC)
#define EXP2(x) pow(2, x)                 /* GFq_LDPC_NTT.c:193 */
#define LOG2(x) (log(x) / log(2.0))       /* GFq_LDPC_NTT.c:194 */

/* initlogexptab2() (GFq_LDPC_NTT.c:196-219) */
void ora_wht_tables(const int **flog, const unsigned int **fexp_centre)
{
  if (!w_flog) {
    int i = 1;
    double right;
    w_flog = (int *)malloc(sizeof(int) * W_LEVELS);
    w_fexp_base = (unsigned int *)malloc(sizeof(unsigned int) * W_LEVELS * 2);
    w_fexp = w_fexp_base + W_LEVELS;
    right = LOG2(w_fix2float(i) - 0.5 * W_PREC);
    w_flog[0] = -W_FMUL * 14;
    for (; i < W_LEVELS; i++) {
      double d = w_fix2float(i);
      double left = LOG2(d + 0.5 * W_PREC);
      w_flog[i] = w_float2fix((4 * LOG2(d) + right + left) / 6.0);
      right = left;
    }
    i = -W_LEVELS;
    right = EXP2(w_fix2float(i) - 0.5 * W_PREC);
    for (; i < W_LEVELS; i++) {
      double d = w_fix2float(i);
      double expd = EXP2(d);
      double left = EXP2(d + 0.5 * W_PREC);
      if (expd > (1UL << (31 - W_DECI))) w_fexp[i] = 1UL << 31;
      else w_fexp[i] = w_float2fixu((4 * expd + right + left) / 6.0);
      right = left;
    }
  }
  if (flog) *flog = w_flog;
  if (fexp_centre) *fexp_centre = w_fexp;
}

/* Flog() (GFq_LDPC_NTT.c:240-247) */
static int w_Flog(int x)
{
  int i;
  if (x <= 0) return -W_FMUL * 14;
  for (i = 0; x >= W_LEVELS; i++) x >>= 1;
  return w_flog[x] + (i << W_DECI);
}

/* ntt() (GFq_LDPC_NTT.c:118-135) */
static void w_ntt(int *p, int log2q)
{
  int b, factor = 1, rest, Q = 1 << log2q;
  for (b = 0; b < log2q; b++) {
    for (rest = 0; rest < Q / 2; rest++) {
      int restH = rest >> b;
      int restL = rest & (factor - 1);
      int rest0 = (restH << (b + 1)) + restL;
      int rest1 = rest0 + factor;
      int prest0 = p[rest0];
      p[rest0] += p[rest1];
      p[rest1] = prest0 - p[rest1];
    }
    factor += factor;
  }
}

/* Fexp(x) = fexp[x] unchecked (GFq_LDPC_NTT.c:236).  An index outside the
 * table is undefined behaviour there; here it aborts the decode with -2. */
#define FEXP_OR_FAIL(dst, x)                                   \
  do {                                                         \
    int x_ = (x);                                              \
    if (x_ < -W_LEVELS || x_ >= W_LEVELS) { rc = -2; goto out; } \
    (dst) = (int)w_fexp[x_];                                   \
  } while (0)

/* dec() (GFq_LDPC_NTT.c:437-544).  logfna is [n][Q]. */
int ora_dec_gf2p_wht(const ora_code *c, int log2q, const int *logfna, const int *z,
                     int loop_max, int *xhat, int *zhat, int *iters)
{
  const int Q = 1 << log2q, n = c->n, m = c->m, rmax = c->rmax;
  int a, i, j, k, loop, rc = -1;
  int iir = 0, prev = 999999, nodecr = 0;
  gf_t g;
  int *logra = (int *)calloc((size_t)m * rmax * Q, sizeof(int));
  int *logqa = (int *)calloc((size_t)m * rmax * Q, sizeof(int));
  int *fQa = (int *)calloc((size_t)rmax * Q, sizeof(int));
  int *isneg = (int *)calloc((size_t)rmax * Q, sizeof(int));
  int *logprodqa = (int *)calloc(Q, sizeof(int)), *sgnsum = (int *)calloc(Q, sizeof(int));
  int *fRa = (int *)calloc(Q, sizeof(int));
#define LRA(j, k) (logra + ((size_t)(j) * rmax + (k)) * Q)
#define LQA(j, k) (logqa + ((size_t)(j) * rmax + (k)) * Q)
  gf_init(&g, Q);
  ora_wht_tables(NULL, NULL);
  *iters = 0;

  for (i = 0; i < n; i++)                                   /* :442-446 */
    for (k = 0; k < c->col_weight[i]; k++)
      memcpy(LQA(c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k]),
             logfna + (size_t)i * Q, sizeof(int) * Q);

  for (loop = 0; loop < loop_max; loop++) {
    for (j = 0; j < m; j++) {                               /* :449-484 */
      int rw = c->row_weight[j];
      memset(sgnsum, 0, sizeof(int) * Q);
      memset(logprodqa, 0, sizeof(int) * Q);
      memset(isneg, 0, sizeof(int) * rmax * Q);
      for (k = 0; k < rw; k++) {
        int h = c->H[j * rmax + k];
        for (a = 0; a < Q; a++) FEXP_OR_FAIL(fQa[k * Q + gf_mul(&g, a, h)], LQA(j, k)[a]);
        w_ntt(fQa + k * Q, log2q);
        for (a = 0; a < Q; a++) {
          int *f = &fQa[k * Q + a];
          if (*f < 0) { *f = -*f; isneg[k * Q + a] = 1; sgnsum[a] ^= 1; }
          *f = (*f + (1 << (log2q / 2 - 1))) >> (log2q / 2);
          logprodqa[a] += w_Flog(*f);
          if (!(logprodqa[a] > -(1 << 29))) { rc = -2; goto out; }   /* assert :469 */
        }
      }
      for (k = 0; k < rw; k++) {
        int h = c->H[j * rmax + k];
        for (a = 0; a < Q; a++) {
          FEXP_OR_FAIL(fRa[a], logprodqa[a] - w_Flog(fQa[k * Q + a]) + log2q * W_FMUL);
          if (isneg[k * Q + a] != sgnsum[a]) fRa[a] = -fRa[a];
        }
        w_ntt(fRa, log2q);
        for (a = 0; a < Q; a++)
          LRA(j, k)[a] = w_Flog(fRa[gf_sub(&g, z[j], gf_mul(&g, a, h))]);
      }
    }
    for (i = 0; i < n; i++) {                               /* :486-504 */
      int col = c->col_weight[i], argmaxa = -1, maxp = -(1 << 30);
      for (a = 0; a < Q; a++) {
        int logprod = logfna[(size_t)i * Q + a];
        for (k = 0; k < col; k++)
          logprod += LRA(c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k])[a];
        for (k = 0; k < col; k++)
          LQA(c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k])[a] =
              logprod - LRA(c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k])[a];
        if (logprod > maxp) { maxp = logprod; argmaxa = a; }
      }
      xhat[i] = argmaxa;
    }
    for (j = 0; j < m; j++)                                 /* :506-523 */
      for (k = 0; k < c->row_weight[j]; k++) {
        int *qajk = LQA(j, k), sum = 0, t;
        for (a = 0; a < Q; a++) { FEXP_OR_FAIL(t, qajk[a]); sum += t; }
        if (sum <= 0) sum = Q;
        sum = w_Flog(sum);
        for (a = 0; a < Q; a++) qajk[a] -= sum;
      }
    ora_gf_enc(c, Q, xhat, zhat);                           /* :526-540 */
    *iters = loop + 1;
    {
      int dist = hamdist_bits(z, zhat, m, Q);
      if (dist == 0) { rc = 0; goto out; }
      if (loop == 0) iir = dist;
      else iir = (int)(iir * 0.85 + dist * 0.15 + 0.5);
      if (prev <= dist) nodecr++;
      else nodecr = 0;
      if (dist > iir * 1.1 || nodecr > 10) break;
      prev = dist;
    }
  }
out:
  free(logra); free(logqa); free(fQa); free(isneg); free(logprodqa); free(sgnsum); free(fRa);
  return rc;
#undef LRA
#undef LQA
}

/* ------------------------------------------------ GF(q) direct decoder */

/* dec() (GFq_LDPC.c:356-467).  logfna is [Q][n] as in the reference. */
int ora_dec_gfq_direct(const ora_code *c, int Q, const double *logfna, const int *z,
                       int loop_max, int *xhat, int *zhat, int *iters, double *post)
{
  const int n = c->n, m = c->m, rmax = c->rmax;
  int a, i, j, k, s, t, loop, rc = 1;
  double sum;
  gf_t g;
  size_t plane = (size_t)m * rmax;
  double *logra = (double *)calloc(plane * Q, sizeof(double));
  double *logqa = (double *)calloc(plane * Q, sizeof(double));
  double *logsigma = (double *)calloc(plane * Q, sizeof(double));
  double *logrho = (double *)calloc(plane * Q, sizeof(double));
#define A3(arr, a, j, k) arr[(size_t)(a) * plane + (size_t)(j) * rmax + (k)]
  gf_init(&g, Q);
  *iters = 0;

  for (a = 0; a < Q; a++)                                   /* :361-366 */
    for (i = 0; i < n; i++)
      for (k = 0; k < c->col_weight[i]; k++)
        A3(logqa, a, c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k]) = logfna[(size_t)a * n + i];

  for (loop = 1; loop <= loop_max; loop++) {
    for (j = 0; j < m; j++) {
      int rw = c->row_weight[j];
      for (k = 0; k < rw; k++) {                            /* sigma :371-385 */
        int h = c->H[j * rmax + k];
        for (a = 0; a < Q; a++) {
          if (k == 0) {
            A3(logsigma, gf_mul(&g, a, h), j, k) = A3(logqa, a, j, k);
          } else {
            sum = 0;
            for (t = 0; t < Q; t++) {
              s = gf_sub(&g, a, gf_mul(&g, t, h));
              sum += exp(A3(logsigma, s, j, k - 1) + A3(logqa, t, j, k));
            }
            A3(logsigma, a, j, k) = log(sum);
          }
        }
      }
      for (k = rw - 1; k >= 0; k--) {                       /* rho :387-403 */
        int h = c->H[j * rmax + k];
        if (k == rw - 1) {
          for (a = 0; a < Q; a++) A3(logrho, gf_mul(&g, a, h), j, k) = A3(logqa, a, j, k);
        } else {
          for (a = 0; a < Q; a++) {
            sum = 0;
            for (t = 0; t < Q; t++) {
              s = gf_sub(&g, a, gf_mul(&g, t, h));
              sum += exp(A3(logrho, s, j, k + 1) + A3(logqa, t, j, k));
            }
            A3(logrho, a, j, k) = log(sum);
          }
        }
      }
    }
    for (j = 0; j < m; j++) {                               /* ra :406-424 */
      int rw = c->row_weight[j];
      for (k = 0; k < rw; k++) {
        int h = c->H[j * rmax + k];
        for (a = 0; a < Q; a++) {
          int z_aH = gf_sub(&g, z[j], gf_mul(&g, a, h));
          if (k == 0) {
            A3(logra, a, j, k) = (rw > 1) ? A3(logrho, z_aH, j, k + 1) : 0.0;
          } else if (k == rw - 1) {
            A3(logra, a, j, k) = A3(logsigma, z_aH, j, k - 1);
          } else {
            sum = 0;
            for (t = 0; t < Q; t++) {
              s = gf_sub(&g, z_aH, t);
              sum += exp(A3(logsigma, s, j, k - 1) + A3(logrho, t, j, k + 1));
            }
            A3(logra, a, j, k) = log(sum);
          }
        }
      }
    }
    for (i = 0; i < n; i++) {                               /* VN + argmax :426-444 */
      int col = c->col_weight[i], argmaxa = -1;
      double maxp = -9e99;
      for (a = 0; a < Q; a++) {
        double logprod = logfna[(size_t)a * n + i];
        for (k = 0; k < col; k++)
          logprod += A3(logra, a, c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k]);
        for (k = 0; k < col; k++)
          A3(logqa, a, c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k]) =
              logprod - A3(logra, a, c->col_row[i * c->cmax + k], c->col_N[i * c->cmax + k]);
        if (post) post[(size_t)a * n + i] = logprod;
        if (logprod > maxp) { maxp = logprod; argmaxa = a; }
      }
      xhat[i] = argmaxa;
    }
    for (j = 0; j < m; j++)                                 /* normalise :446-457 */
      for (k = 0; k < c->row_weight[j]; k++) {
        sum = 0;
        for (a = 0; a < Q; a++) sum += exp(A3(logqa, a, j, k));
        sum = log(sum);
        for (a = 0; a < Q; a++) A3(logqa, a, j, k) -= sum;
      }
    ora_gf_enc(c, Q, xhat, zhat);                           /* :459-463 */
    *iters = loop;
    if (differs(z, zhat, m) == 0) { rc = 0; break; }
  }
  free(logra); free(logqa); free(logsigma); free(logrho);
  return rc;
#undef A3
}

/* make_p_sent_given_rec_T() (GFq_LDPC.c:177-197) followed by assign_llr()
 * (:199-211): logfna[j][i] = log P(j sent | y_i received), floor -100.
 * p_rec_given_sent is [Q*num_reads][Q] row-major, already column-normalised
 * as GFq_LDPC() does at :548-558 (ora_gf_normalise_confusion). */
void ora_gf_normalise_confusion(int Q, int num_reads, double *p_rec_given_sent)
{
  int i, j;
  for (i = 0; i < Q; i++) {
    double norm = 0;
    for (j = 0; j < Q * num_reads; j++) norm += p_rec_given_sent[j * Q + i];
    for (j = 0; j < Q * num_reads; j++) p_rec_given_sent[j * Q + i] /= norm;
  }
}

void ora_gf_assign_llr(int Q, int num_reads, const double *p_rec_given_sent,
                       const int *y, int n, double *logfna)
{
  int i, j, r;
  double P_x = 1 / (1.0 * Q);
  double *T = (double *)calloc((size_t)Q * num_reads * Q, sizeof(double));
  for (r = 0; r < Q * num_reads; r++) {
    double P_y = 0;
    for (j = 0; j < Q; j++) P_y += p_rec_given_sent[r * Q + j] * P_x;
    if (P_y) for (j = 0; j < Q; j++) T[r * Q + j] = P_x * p_rec_given_sent[r * Q + j] / P_y;
    else for (j = 0; j < Q; j++) T[r * Q + j] = 0;
  }
  for (i = 0; i < n; i++)
    for (j = 0; j < Q; j++) {
      double p = T[y[i] * Q + j];
      logfna[(size_t)j * n + i] = (p == 0) ? -100 : log(p);
    }
  free(T);
}

/* channel() (GFq_LDPC.c:213-231), the choice of the read y[i]: rand_select is the
 * reference's `((float)rand())*100/((float)RAND_MAX)` handed in by the caller (its
 * rand() stream is seeded from the clock); p_rec_given_sent as for ora_gf_assign_llr.
 * The reference's first loop (:224-226) has no end test; neither has this one beyond
 * the table (a column of a normalised matrix always holds a non-zero entry). */
void ora_gf_channel_draw(int Q, int num_reads, const double *p_rec_given_sent, const int *x,
                         const float *rand_select_in, int n, int *y)
{
  int i, rec_ind;
  double temp;
  float rand_select;
  for (i = 0; i < n; i++) {
    temp = 0;
    rec_ind = 0;
    rand_select = rand_select_in[i];
    if (!rand_select) {
      while (rec_ind < Q * num_reads && !p_rec_given_sent[(rec_ind++) * Q + x[i]]);
    }
    while (temp < rand_select && rec_ind != Q * num_reads) {
      temp += 100 * p_rec_given_sent[(rec_ind++) * Q + x[i]];
    }
    y[i] = --rec_ind;
  }
}
