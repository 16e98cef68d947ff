/*
 * ldpc_oracle.c -- TEST INFRASTRUCTURE ONLY (see ldpc_oracle.h).
 *
 * CPU restatement of the binary decoders of mohammedmidhat/LDPC.  The message
 * stores keep the reference's shapes: c2v messages live per variable
 * (qin[n][cmax]), v2c magnitude/sign live per check ([m][rmax]); where the
 * reference wires pointer tables (qin_row, LogTanhtin_row, Sgntin_row) this
 * file uses the (row_N, col_row, col_N) index tables those pointers were
 * built from (test_code_min_sum_B.c:253-277).
 */
#include "ldpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ loader */

static int *ialloc(size_t k) { return (int *)calloc(k ? k : 1, sizeof(int)); }

/* initdec(): token-stream alist parse (test_code_min_sum_B.c:210-286;
 * dialect 1 = test_code_B.c:235-314; NB files = GFq_LDPC_NTT.c:547-639). */
int ora_load_alist(const char *path, int dialect, ora_code **out)
{
  FILE *fp = fopen(path, "rt");
  char buf[8192];
  int i, j, q = 0, isbinary = 0, *count, tmp;
  ora_code *c;
  if (!fp) return -2;
  c = (ora_code *)calloc(1, sizeof(*c));
  /* header: "n m" or "n m q" (GFq_LDPC_NTT.c:557-563) */
  if (!fgets(buf, sizeof buf, fp)) { fclose(fp); free(c); return -1; }
  if (sscanf(buf, "%d%d%d", &c->n, &c->m, &q) == 2) { isbinary = 1; q = 2; }
  c->q = q;
  if (fscanf(fp, "%d%d", &c->cmax, &c->rmax) != 2) { fclose(fp); free(c); return -1; }
  c->col_weight = ialloc(c->n);
  c->row_weight = ialloc(c->m);
  for (i = 0; i < c->n; i++) if (fscanf(fp, "%d", &c->col_weight[i]) != 1) goto bad;
  for (j = 0; j < c->m; j++) if (fscanf(fp, "%d", &c->row_weight[j]) != 1) goto bad;
  /* skip the column lists */
  for (i = 0; i < c->n; i++) {
    int lim = dialect ? c->col_weight[i] : c->cmax;
    for (j = 0; j < lim; j++) {
      if (fscanf(fp, "%d", &tmp) != 1) goto bad;
      if (!isbinary && fscanf(fp, "%d", &tmp) != 1) goto bad;
    }
  }
  count = ialloc(c->n);
  c->row_col = ialloc((size_t)c->m * c->rmax);
  c->row_N   = ialloc((size_t)c->m * c->rmax);
  c->H       = ialloc((size_t)c->m * c->rmax);
  c->col_row = ialloc((size_t)c->n * c->cmax);
  c->col_N   = ialloc((size_t)c->n * c->cmax);
  for (j = 0; j < c->m; j++) {
    for (i = 0; i < c->row_weight[j]; i++) {
      int v, h = 1;
      if (fscanf(fp, "%d", &v) != 1) { free(count); goto bad; }
      if (!isbinary && fscanf(fp, "%d", &h) != 1) { free(count); goto bad; }
      v--;
      if (v < 0 || v >= c->n || count[v] >= c->cmax) { free(count); goto bad; }
      c->row_col[j * c->rmax + i] = v;
      c->row_N[j * c->rmax + i] = count[v];
      c->H[j * c->rmax + i] = h;
      c->col_row[v * c->cmax + count[v]] = j;
      c->col_N[v * c->cmax + count[v]] = i;
      count[v]++;
    }
    for (; i < c->rmax; i++) {        /* fillers (:266-268); EOF tolerated like fscanf */
      if (fscanf(fp, "%d", &tmp) != 1) break;
      if (!isbinary && fscanf(fp, "%d", &tmp) != 1) break;
    }
  }
  free(count);
  fclose(fp);
  *out = c;
  return 0;
bad:
  fclose(fp);
  ora_free_code(c);
  return -1;
}

void ora_free_code(ora_code *c)
{
  if (!c) return;
  free(c->col_weight); free(c->row_weight); free(c->row_col); free(c->row_N);
  free(c->col_row); free(c->col_N); free(c->H); free(c);
}

/* enc() (test_code_min_sum_B.c:69-79) */
void ora_enc(const ora_code *c, const int *y, int *s)
{
  int i, j;
  for (j = 0; j < c->m; j++) {
    int k = 0;
    for (i = 0; i < c->row_weight[j]; i++) k ^= y[c->row_col[j * c->rmax + i]];
    s[j] = k;
  }
}

/* HamDist() (test_code_min_sum_B.c:32-39) */
static int hamdist(const int *x, const int *y, int len)
{
  int i, sum = 0;
  for (i = 0; i < len; i++) if (x[i] != y[i]) sum++;
  return sum;
}

/* float_to_fix() with INT_LEVELS=128, FRAC_LEVELS=256
 * (test_code_min_sum_B_fixed.c:43-53, :397-398) */
double ora_float_to_fix(double val)
{
  double intermed = round(val * 256);
  double result = intermed / 256;
  if (result > 128 - 1) return 128 - 1;
  else if (result < -128) return -128;
  return result;
}

/* ------------------------------------------------- double / Q8.8 min-sum */

#define QIN(i, j)  qin[(size_t)(i) * c->cmax + (j)]
#define MAG(j, k)  mag[(size_t)(j) * c->rmax + (k)]
#define SGN(j, k)  sgn[(size_t)(j) * c->rmax + (k)]

/* One template for test_code_min_sum_B.c:137-208 (fixed=0) and
 * test_code_min_sum_B_fixed.c:192-316 (fixed=1). */
static int dec_minsum_any(const ora_code *c, const double *q0, const int *s,
                          int loop_max, int *bits, int *synd, int *iters,
                          double *post, int fixed, FILE *trace, int circ_size)
{
  int i, j, k, l, loop, n = c->n, m = c->m, rc = 1;
  double *qin = (double *)calloc((size_t)n * c->cmax + 1, sizeof(double)); /* memset 0, :141 */
  double *mag = (double *)calloc((size_t)m * c->rmax + 1, sizeof(double));
  int *sgn = ialloc((size_t)m * c->rmax);
  int *shown = ialloc(n);
  *iters = 0;
  if (post) for (i = 0; i < n; i++) post[i] = q0[i];

  if (fixed) {              /* pre-check, _fixed.c:198-207 */
    for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
    ora_enc(c, bits, synd);
    if (hamdist(s, synd, m) == 0) {
      if (trace) fprintf(trace, "already cw\n");
      rc = 0;
      goto out;
    }
  }

  for (loop = 0; loop < loop_max; loop++) {
    /* variable-node pass, :144-160 / _fixed.c:213-229 */
    for (i = 0; i < n; i++) {
      double sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      for (j = 0; j < c->col_weight[i]; j++) {
        double qout = sum - QIN(i, j);
        int r = c->col_row[i * c->cmax + j], kk = c->col_N[i * c->cmax + j];
        if (qout < 0) {
          MAG(r, kk) = fixed ? ora_float_to_fix(-qout) : -qout;
          SGN(r, kk) = 1;
        } else {
          MAG(r, kk) = fixed ? ora_float_to_fix(qout) : qout;
          SGN(r, kk) = 0;
        }
      }
    }

    /* Verilog v2c dump, _fixed.c:245-260 */
    if (fixed && trace) {
      memset(shown, 0, sizeof(int) * n);
      for (j = 0; j < circ_size; j++)
        for (i = 0; i < m / circ_size; i++)
          for (k = 0; k < n / circ_size; k++) {
            int v = circ_size * k + j, hit = 0;
            for (l = 0; l < c->col_weight[v]; l++) {   /* array_element_in_range :60-70 */
              int r = c->col_row[v * c->cmax + l];
              if (r >= circ_size * i && r <= circ_size * (i + 1) - 1) { hit = 1; break; }
            }
            if (hit) {
              double sum = q0[v];
              for (l = 0; l < c->col_weight[v]; l++) sum += QIN(v, l);
              fprintf(trace, "%08x\n", (int)((sum - QIN(v, shown[v]++)) * 256));
            }
          }
    }

    /* check-node pass, :162-187 */
    for (j = 0; j < m; j++) {
      int sgnprod = s[j];
      for (k = 0; k < c->row_weight[j]; k++) sgnprod ^= SGN(j, k);
      for (k = 0; k < c->row_weight[j]; k++) {
        double min_msg = 200;
        int v = c->row_col[j * c->rmax + k], slot = c->row_N[j * c->rmax + k];
        for (l = 0; l < c->row_weight[j]; l++)
          if (l != k)
            if (fabs(MAG(j, l)) < min_msg) min_msg = fabs(MAG(j, l));
        if (sgnprod != SGN(j, k)) QIN(v, slot) = -min_msg;
        else QIN(v, slot) = min_msg;
      }
    }

    /* Verilog c2v dump, _fixed.c:288-296 */
    if (fixed && trace) {
      for (j = 0; j < circ_size; j++)
        for (k = 0; k < m / circ_size; k++)
          for (i = 0; i < c->row_weight[circ_size * k]; i++) {
            int r = circ_size * k + j;
            int v = c->row_col[r * c->rmax + i], slot = c->row_N[r * c->rmax + i];
            fprintf(trace, "%08x\n", (int)(QIN(v, slot) * 256));
          }
    }

    /* posterior + hard decision, :189-197 / _fixed.c:298-306 */
    for (i = 0; i < n; i++) {
      double sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      if (fixed) sum = ora_float_to_fix(sum);
      if (post) post[i] = sum;
      bits[i] = (sum < 0) ? 1 : 0;
    }
    ora_enc(c, bits, synd);
    *iters = loop + 1;
    if (hamdist(s, synd, m) == 0) { rc = 0; goto out; }
  }
  if (loop_max <= 0 && !fixed) {     /* tmp_bit/tmp_s undefined in the reference; define them */
    for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
    ora_enc(c, bits, synd);
  }
out:
  free(qin); free(mag); free(sgn); free(shown);
  return rc;
}

int ora_dec_minsum(const ora_code *c, const double *q0, const int *s, int loop_max,
                   int *bits, int *synd, int *iters, double *post)
{
  return dec_minsum_any(c, q0, s, loop_max, bits, synd, iters, post, 0, NULL, 1);
}

int ora_dec_minsum_fixed(const ora_code *c, const double *q0, const int *s, int loop_max,
                         int *bits, int *synd, int *iters, double *post,
                         FILE *trace, int circ_size)
{
  return dec_minsum_any(c, q0, s, loop_max, bits, synd, iters, post, 1, trace, circ_size);
}

/* Layered (row-block serial) schedule of the Q8.8 min-sum decoder.  NOT in the reference (SURVEY 8 f4
 * "alternative schedules"): the arithmetic of every message is dec()'s (test_code_min_sum_B_fixed.c:
 * 213-229 v2c = float_to_fix(|sum - c2v|) with its sign, :264-283 min / sign product with the 200 cap),
 * the pre-check (:198-207) and the per-iteration exit test (:298-313) are dec()'s; only the ORDER of
 * the updates differs: an iteration walks the checks in layers of `layer` consecutive rows; the checks
 * of one layer all read the sums as the previous layer left them, then their new messages replace
 * their old ones in the sums (sum += new - old) before the next layer reads.  layer >= m is the
 * reference's flooding schedule, and then this function returns exactly what ora_dec_minsum_fixed
 * does (tests/test_oracle_golden.py pins it that way). */
int ora_dec_minsum_fixed_layered(const ora_code *c, const double *q0, const int *s, int loop_max,
                                 int layer, int *bits, int *synd, int *iters, double *post)
{
  int i, j, k, l, loop, n = c->n, m = c->m, rc = 1;
  double *sum = (double *)calloc((size_t)n + 1, sizeof(double));
  double *c2v = (double *)calloc((size_t)m * c->rmax + 1, sizeof(double));   /* all 0, _fixed.c:196 */
  double *nw = (double *)calloc((size_t)m * c->rmax + 1, sizeof(double));
  *iters = 0;
  if (layer < 1) layer = m;
  for (i = 0; i < n; i++) { sum[i] = q0[i]; if (post) post[i] = q0[i]; }
  for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
  ora_enc(c, bits, synd);
  if (hamdist(s, synd, m) == 0) { rc = 0; goto out; }

  for (loop = 0; loop < loop_max; loop++) {
    int j0;
    for (j0 = 0; j0 < m; j0 += layer) {
      int j1 = j0 + layer < m ? j0 + layer : m;
      for (j = j0; j < j1; j++) {                 /* every check of the layer reads the same sums */
        int w = c->row_weight[j], sgnprod = s[j];
        double mag[256];
        int sg[256];
        for (k = 0; k < w; k++) {
          double qout = sum[c->row_col[j * c->rmax + k]] - c2v[(size_t)j * c->rmax + k];
          sg[k] = qout < 0;
          mag[k] = ora_float_to_fix(qout < 0 ? -qout : qout);
          sgnprod ^= sg[k];
        }
        for (k = 0; k < w; k++) {
          double min_msg = 200;
          for (l = 0; l < w; l++)
            if (l != k && mag[l] < min_msg) min_msg = mag[l];
          nw[(size_t)j * c->rmax + k] = (sgnprod != sg[k]) ? -min_msg : min_msg;
        }
      }
      for (j = j0; j < j1; j++)                   /* then their messages are exchanged in the sums */
        for (k = 0; k < c->row_weight[j]; k++) {
          size_t e = (size_t)j * c->rmax + k;
          sum[c->row_col[j * c->rmax + k]] += nw[e] - c2v[e];
          c2v[e] = nw[e];
        }
    }
    for (i = 0; i < n; i++) {
      double sat = ora_float_to_fix(sum[i]);      /* _fixed.c:303 */
      if (post) post[i] = sat;
      bits[i] = (sat < 0) ? 1 : 0;
    }
    ora_enc(c, bits, synd);
    *iters = loop + 1;
    if (hamdist(s, synd, m) == 0) { rc = 0; goto out; }
  }
out:
  free(sum); free(c2v); free(nw);
  return rc;
}

/* binary32 twin of ora_dec_minsum (same statements, float arithmetic). */
int ora_dec_minsum_f32(const ora_code *c, const float *q0, const int *s, int loop_max,
                       int *bits, int *synd, int *iters, float *post)
{
  int i, j, k, l, loop, n = c->n, m = c->m, rc = 1;
  float *qin = (float *)calloc((size_t)n * c->cmax + 1, sizeof(float));
  float *mag = (float *)calloc((size_t)m * c->rmax + 1, sizeof(float));
  int *sgn = ialloc((size_t)m * c->rmax);
  *iters = 0;
  if (post) for (i = 0; i < n; i++) post[i] = q0[i];
  for (loop = 0; loop < loop_max; loop++) {
    for (i = 0; i < n; i++) {
      volatile float sum = q0[i];       /* volatile: forbid wider intermediates */
      for (j = 0; j < c->col_weight[i]; j++) sum = sum + QIN(i, j);
      for (j = 0; j < c->col_weight[i]; j++) {
        volatile float qout = sum - QIN(i, j);
        int r = c->col_row[i * c->cmax + j], kk = c->col_N[i * c->cmax + j];
        if (qout < 0) { MAG(r, kk) = -qout; SGN(r, kk) = 1; }
        else          { MAG(r, kk) = qout;  SGN(r, kk) = 0; }
      }
    }
    for (j = 0; j < m; j++) {
      int sgnprod = s[j];
      for (k = 0; k < c->row_weight[j]; k++) sgnprod ^= SGN(j, k);
      for (k = 0; k < c->row_weight[j]; k++) {
        float min_msg = 200;
        int v = c->row_col[j * c->rmax + k], slot = c->row_N[j * c->rmax + k];
        for (l = 0; l < c->row_weight[j]; l++)
          if (l != k)
            if (fabsf(MAG(j, l)) < min_msg) min_msg = fabsf(MAG(j, l));
        if (sgnprod != SGN(j, k)) QIN(v, slot) = -min_msg;
        else QIN(v, slot) = min_msg;
      }
    }
    for (i = 0; i < n; i++) {
      volatile float sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum = sum + QIN(i, j);
      if (post) post[i] = sum;
      bits[i] = (sum < 0) ? 1 : 0;
    }
    ora_enc(c, bits, synd);
    *iters = loop + 1;
    if (hamdist(s, synd, m) == 0) { rc = 0; break; }
  }
  if (loop_max <= 0) {
    for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
    ora_enc(c, bits, synd);
  }
  free(qin); free(mag); free(sgn);
  return rc;
}

/* ------------------------------------------------------ float sum-product */

/* atanh2 / logtanh2 (test_code_B_float.c:20-27) */
static double atanh2(double x) { return log((1.0 + x) / (1.0 - x)); }
static double logtanh2(double x) { return log(tanh(fabs(x * 0.5))); }

/* dec() (test_code_B_float.c:134-199) */
int ora_dec_spa(const ora_code *c, const double *q0, const int *s, int loop_max,
                int *bits, int *synd, int *iters, double *post)
{
  int i, j, k, loop, n = c->n, m = c->m, rc = 1;
  double *qin = (double *)calloc((size_t)n * c->cmax + 1, sizeof(double));
  double *mag = (double *)calloc((size_t)m * c->rmax + 1, sizeof(double));
  int *sgn = ialloc((size_t)m * c->rmax);
  *iters = 0;
  if (post) for (i = 0; i < n; i++) post[i] = q0[i];
  for (loop = 0; loop < loop_max; loop++) {
    for (i = 0; i < n; i++) {
      double sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      for (j = 0; j < c->col_weight[i]; j++) {
        double qout = sum - QIN(i, j);
        int r = c->col_row[i * c->cmax + j], kk = c->col_N[i * c->cmax + j];
        if (qout < 0) { MAG(r, kk) = logtanh2(-qout); SGN(r, kk) = 1; }
        else          { MAG(r, kk) = logtanh2(qout);  SGN(r, kk) = 0; }
      }
    }
    for (j = 0; j < m; j++) {
      int sgnprod = s[j];
      double logprod = 0;
      for (k = 0; k < c->row_weight[j]; k++) { logprod += MAG(j, k); sgnprod ^= SGN(j, k); }
      for (k = 0; k < c->row_weight[j]; k++) {
        double tout = atanh2(exp(logprod - MAG(j, k)));
        int v = c->row_col[j * c->rmax + k], slot = c->row_N[j * c->rmax + k];
        if (sgnprod != SGN(j, k)) QIN(v, slot) = -tout;
        else QIN(v, slot) = tout;
      }
    }
    for (i = 0; i < n; i++) {
      double sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      if (post) post[i] = sum;
      bits[i] = (sum < 0) ? 1 : 0;
    }
    ora_enc(c, bits, synd);
    *iters = loop + 1;
    if (hamdist(s, synd, m) == 0) { rc = 0; break; }
  }
  if (loop_max <= 0) {
    for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
    ora_enc(c, bits, synd);
  }
  free(qin); free(mag); free(sgn);
  return rc;
}

/* binary32 twin of logtanh2 / atanh2(exp(.)) (test_code_B_float.c:20-27).  Not in the reference
 * (it computes in double).  The literal binary32 statement saturates at |LLR| ~ 17 (tanhf -> 1,
 * then atanh2(1) = inf and inf - inf = NaN), where the reference's double still decodes; the
 * twin therefore evaluates the same two functions without that cancellation (series of
 * -2 atanh(e^-a) above 3, of -expm1(x) above -0.1 and of 2 atanh(e^x) below -3), as the
 * binary32 kernels do (ldpc_amd/csrc/binary_kernels.hip.h).  -inf, NaN behave as before; a zero
 * magnitude is floored at 2^-100, which takes away the reference's log tanh(0) = -inf, -inf - (-inf)
 * = NaN hazard (:149,:168): binary32 rounding cancels variable sums to exactly 0 on frames where
 * the reference's double does not. */
static float logtanh2f(float x) {
  float a = fabsf(x), t, num;
  if (a < 0x1p-100f) a = 0x1p-100f;
  t = expf(-a);
  if (a > 3.0f) { float t2 = t * t; return -2.0f * t * (1.0f + t2 * (1.0f / 3.0f + t2 * 0.2f)); }
  num = 1.0f - t;
  if (a < 0.25f)
    num = a * (1.0f - a * (0.5f - a * (1.0f / 6.0f - a * (1.0f / 24.0f - a * (1.0f / 120.0f - a * (1.0f / 720.0f))))));
  return logf(num / (1.0f + t));
}
static float atanh2expf(float x) {
  float e = expf(x);
  if (x < -3.0f) { const float e2 = e * e; return 2.0f * e * (1.0f + e2 * (1.0f / 3.0f + e2 * 0.2f)); }
  if (x > -0.1f) return logf((1.0f + e) / (fabsf(x) * (1.0f + x * (0.5f + x * (1.0f / 6.0f + x * (1.0f / 24.0f + x * (1.0f / 120.0f)))))));   /* x = 0: log(2 / +0) = +inf like the reference */
  return logf((1.0f + e) / (1.0f - e));
}

int ora_dec_spa_f32(const ora_code *c, const float *q0, const int *s, int loop_max,
                    int *bits, int *synd, int *iters, float *post)
{
  int i, j, k, loop, n = c->n, m = c->m, rc = 1;
  float *qin = (float *)calloc((size_t)n * c->cmax + 1, sizeof(float));
  float *mag = (float *)calloc((size_t)m * c->rmax + 1, sizeof(float));
  int *sgn = ialloc((size_t)m * c->rmax);
  *iters = 0;
  if (post) for (i = 0; i < n; i++) post[i] = q0[i];
  for (loop = 0; loop < loop_max; loop++) {
    for (i = 0; i < n; i++) {
      float sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      for (j = 0; j < c->col_weight[i]; j++) {
        float qout = sum - QIN(i, j);
        int r = c->col_row[i * c->cmax + j], kk = c->col_N[i * c->cmax + j];
        if (qout < 0) { MAG(r, kk) = logtanh2f(-qout); SGN(r, kk) = 1; }
        else          { MAG(r, kk) = logtanh2f(qout);  SGN(r, kk) = 0; }
      }
    }
    for (j = 0; j < m; j++) {
      int sgnprod = s[j];
      double logprod = 0;   /* binary64 sum, like the reference and the binary32 kernels */
      for (k = 0; k < c->row_weight[j]; k++) { logprod += MAG(j, k); sgnprod ^= SGN(j, k); }
      for (k = 0; k < c->row_weight[j]; k++) {
        float tout = atanh2expf((float)(logprod - (double)MAG(j, k)));
        int v = c->row_col[j * c->rmax + k], slot = c->row_N[j * c->rmax + k];
        if (sgnprod != SGN(j, k)) QIN(v, slot) = -tout;
        else QIN(v, slot) = tout;
      }
    }
    for (i = 0; i < n; i++) {
      float sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      if (post) post[i] = sum;
      bits[i] = (sum < 0) ? 1 : 0;
    }
    ora_enc(c, bits, synd);
    *iters = loop + 1;
    if (hamdist(s, synd, m) == 0) { rc = 0; break; }
  }
  if (loop_max <= 0) {
    for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
    ora_enc(c, bits, synd);
  }
  free(qin); free(mag); free(sgn);
  return rc;
}

/* ----------------------------------------------------- table sum-product */

#define T_INT   8
#define T_DECI  14
#define T_FMUL  (1 << T_DECI)
#define T_PREC  (1.0 / T_FMUL)
#define T_LEVELS (1 << (T_INT + T_DECI))
static int *t_flogtanh, *t_fgallag;

/* float2fix() (test_code_B.c:36-43) */
int ora_float2fix14(double x)
{
  if (x >= 0) return (int)(x * T_FMUL + 0.5);
  return -(int)((-x) * T_FMUL + 0.5);
}
#define fix2float(x) ((x) * T_PREC)

/* inittab() (test_code_B.c:52-74) */
void ora_inittab(const int **flt, const int **fg)
{
  if (!t_flogtanh) {
    int i = 1;
    double right;
    t_flogtanh = (int *)malloc(sizeof(int) * T_LEVELS);
    t_fgallag = (int *)malloc(sizeof(int) * T_LEVELS);
    right = logtanh2(fix2float(i) - 0.5 * T_PREC);
    t_flogtanh[0] = -T_FMUL * 14;
    for (; i < T_LEVELS; i++) {
      double d = fix2float(i);
      double left = logtanh2(d + 0.5 * T_PREC);
      t_flogtanh[i] = ora_float2fix14((4 * logtanh2(d) + right + left) / 6.0);
      right = left;
    }
    i = 1;
    t_fgallag[0] = T_FMUL * 14;
    right = atanh2(exp(fix2float(-i) - 0.5 * T_PREC));
    for (; i < T_LEVELS; i++) {
      double d = fix2float(-i);
      double expd = atanh2(exp(d));
      double left = atanh2(exp(d + 0.5 * T_PREC));
      t_fgallag[i] = ora_float2fix14((4 * expd + right + left) / 6.0);
      right = left;
    }
  }
  if (flt) *flt = t_flogtanh;
  if (fg) *fg = t_fgallag;
}

/* dec() (test_code_B.c:171-233) with Flogtanh/Fgallag (:76-90) inlined */
int ora_dec_spa_table(const ora_code *c, const int *q0, const int *s, int loop_max,
                      int *bits, int *synd, int *iters, int *post)
{
  int i, j, k, loop, n = c->n, m = c->m, rc = 1;
  int *qin = ialloc((size_t)n * c->cmax);
  int *mag = ialloc((size_t)m * c->rmax);
  int *sgn = ialloc((size_t)m * c->rmax);
  ora_inittab(NULL, NULL);
  *iters = 0;
  if (post) for (i = 0; i < n; i++) post[i] = q0[i];
  for (loop = 0; loop < loop_max; loop++) {
    for (i = 0; i < n; i++) {
      int sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      for (j = 0; j < c->col_weight[i]; j++) {
        int qout = sum - QIN(i, j);
        int r = c->col_row[i * c->cmax + j], kk = c->col_N[i * c->cmax + j];
        int a = qout < 0 ? -qout : qout;
        if (a < 0) { rc = -2; goto out; }                 /* assert(x>=0), :78 (INT_MIN) */
        MAG(r, kk) = (a >= T_LEVELS) ? 0 : t_flogtanh[a];
        SGN(r, kk) = qout < 0 ? 1 : 0;
      }
    }
    for (j = 0; j < m; j++) {
      int sgnprod = s[j];
      int logprod = 0;
      for (k = 0; k < c->row_weight[j]; k++) { logprod += MAG(j, k); sgnprod ^= SGN(j, k); }
      for (k = 0; k < c->row_weight[j]; k++) {
        int x = logprod - MAG(j, k), tout;
        int v = c->row_col[j * c->rmax + k], slot = c->row_N[j * c->rmax + k];
        if (x > 0) { rc = -2; goto out; }                 /* assert(x<=0), :86 */
        tout = (x <= -T_LEVELS) ? 0 : t_fgallag[-x];
        if (sgnprod != SGN(j, k)) QIN(v, slot) = -tout;
        else QIN(v, slot) = tout;
      }
    }
    for (i = 0; i < n; i++) {
      int sum = q0[i];
      for (j = 0; j < c->col_weight[i]; j++) sum += QIN(i, j);
      if (post) post[i] = sum;
      bits[i] = (sum < 0) ? 1 : 0;
    }
    ora_enc(c, bits, synd);
    *iters = loop + 1;
    if (hamdist(s, synd, m) == 0) { rc = 0; break; }
  }
  if (loop_max <= 0) {
    for (i = 0; i < n; i++) bits[i] = (q0[i] < 0) ? 1 : 0;
    ora_enc(c, bits, synd);
  }
out:
  free(qin); free(mag); free(sgn);
  return rc;
}
