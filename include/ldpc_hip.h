/*
 * ldpc_hip.h -- C ABI of libldpc_hip, the MI355X (gfx950) batched LDPC decoder.
 *
 * Drop-in for ONE path of mohammedmidhat/LDPC: the flooding-schedule dec()
 * message-passing decoders and their alist loader.  Every entry point names
 * the reference interface it replaces (file:line, relative to the reference
 * tree).  Plain C types only; no torch / HIP types in the signatures (a
 * hipStream_t crosses as void*).
 *
 * Conventions that differ from the reference on purpose:
 *   - errors are returned as negative codes (the reference calls exit());
 *   - there is no global state: a code and a decoder are handles, one decoder
 *     per GPU, a decoder is not re-entrant but different decoders are
 *     independent and host-thread-safe;
 *   - the iteration count and the computed syndrome are outputs (the
 *     reference keeps them in globals or drops them).
 * Conventions kept: LLR > 0 <=> bit 0, hard decision is (sum < 0);
 * status 0 = target syndrome reached, 1 = not reached; alist indices 1-based
 * on disk, 0-based in memory; all decoders are syndrome decoders (zero
 * syndrome = ordinary codeword decoding).
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_OK            0
#define LDPC_ERR_FORMAT   -1   /* malformed alist (reference: exit(-1), GFq_LDPC.c:520-523)   */
#define LDPC_ERR_IO       -2   /* cannot open file (reference: exit(-2), test_code_min_sum_B.c:216-219) */
#define LDPC_ERR_NOMEM    -3   /* host or device allocation failed (reference: exit(-1), :86)  */
#define LDPC_ERR_ARG      -4   /* bad argument / size beyond the decoder's limits              */
#define LDPC_ERR_HIP      -5   /* HIP runtime error; text via ldpc_last_error()                */
#define LDPC_ERR_UNSUPPORTED -6 /* algorithm/code combination not supported                    */

const char *ldpc_strerror(int code);
/* Text of the last error raised on the calling thread ("" if none). */
const char *ldpc_last_error(void);
/* Library version / build target string, e.g. "libldpc_hip 0.1 gfx950". */
const char *ldpc_version(void);

/* ------------------------------------------------------------------ codes */

typedef struct ldpc_code ldpc_code;

/* Replaces initdec() (test_code_min_sum_B.c:210-286, test_code_B.c:235-314,
 * test_code_no_mex_B.c:244-319) and, for "n m q" headers,
 * InitSumProductDecoderSyndrome() (GFq_LDPC.c:469-535) / the NTT initdec()
 * (GFq_LDPC_NTT.c:547-639).  MacKay alist, padded or unpadded, tabs or
 * spaces; like the reference, only the ROW lists define the graph (column
 * lists are skipped) and edges keep the reference numbering: column slot k of
 * variable v is the k-th row, scanning rows in ascending order, that holds v. */
int ldpc_code_load_alist(const char *path, ldpc_code **out);

/* Graph handed over as arrays, the shape of the WD-farm boundary
 * (WD_farm/GFq_LDPC.c:477): row_ptr[m+1], col_idx[E] 0-based, hval[E] or NULL
 * (binary).  q = 2 for binary codes.  LDPC_ERR_FORMAT unless row_ptr starts at 0 and is
 * monotone, every column is in range and no variable appears twice in a row. */
int ldpc_code_from_csr(int n, int m, int q, const int32_t *row_ptr,
                       const int32_t *col_idx, const int32_t *hval, ldpc_code **out);

void ldpc_code_free(ldpc_code *c);

int ldpc_code_n(const ldpc_code *c);     /* variables                         */
int ldpc_code_m(const ldpc_code *c);     /* checks                            */
int ldpc_code_q(const ldpc_code *c);     /* field order, 2 for binary         */
int ldpc_code_edges(const ldpc_code *c); /* E                                 */
int ldpc_code_cmax(const ldpc_code *c);  /* max column weight (from the rows) */
int ldpc_code_rmax(const ldpc_code *c);  /* max row weight                    */
/* Copies the CSR form out: row_ptr[m+1], col_idx[E], hval[E] (any may be NULL). */
int ldpc_code_get_csr(const ldpc_code *c, int32_t *row_ptr, int32_t *col_idx, int32_t *hval);
/* Copies the CSC view out: col_ptr[n+1], edge_of[E] = CSR edge id of (v, slot). */
int ldpc_code_get_csc(const ldpc_code *c, int32_t *col_ptr, int32_t *edge_of);

/* enc() (test_code_min_sum_B.c:69-79; GF: GFq_LDPC.c:299-310) on the host:
 * s = H x over GF(q).  x[n], s[m] one symbol per int32. */
int ldpc_code_syndrome(const ldpc_code *c, const int32_t *x, int32_t *s);

/* --------------------------------------------------------------- decoders */

typedef enum ldpc_alg {
  LDPC_ALG_MINSUM_F32    = 0, /* dec(), test_code_min_sum_B.c:137-208, binary32 messages       */
  LDPC_ALG_MINSUM_F64    = 1, /* same in binary64: the reference's own arithmetic, bit-exact   */
  LDPC_ALG_MINSUM_Q88    = 2, /* dec(), test_code_min_sum_B_fixed.c:192-316, int16 Q8.8        */
  LDPC_ALG_SPA_F32       = 3, /* dec(), test_code_B_float.c:134-199, binary32 messages: logtanh2 /
                                 atanh2 evaluated without cancellation, logprod summed in binary64,
                                 so that it follows the reference's double past |LLR| ~ 17; a zero
                                 message is floored at 2^-100 (no log tanh(0) = -inf -> NaN)      */
  LDPC_ALG_SPA_F64       = 4, /* same in binary64                                              */
  LDPC_ALG_SPA_TABLE_Q14 = 5, /* dec(), test_code_B.c:171-233 / test_code_no_mex_B.c:170-242   */
  LDPC_ALG_GF2P_WHT      = 6, /* dec(), GFq_LDPC_NTT.c:437-544, GF(2^p) integer WHT domain     */
  LDPC_ALG_GFQ_DIRECT    = 7  /* dec(), GFq_LDPC.c:356-467, GF(q) log-domain forward/backward  */
} ldpc_alg;

typedef enum ldpc_dtype {
  LDPC_DT_F64 = 0, LDPC_DT_F32 = 1, LDPC_DT_I16 = 2, LDPC_DT_I32 = 3
} ldpc_dtype;

typedef struct ldpc_opts {
  int32_t alg;          /* ldpc_alg                                                     */
  int32_t device;       /* HIP device ordinal                                           */
  int32_t max_frames;   /* frames resident per pass (0 = default 4096); larger batches
                           are processed in passes of this size                         */
  int32_t early_exit;   /* 1: stop a pass once every frame in it has reached its
                           syndrome (host polls a device counter, one iteration behind);
                           0: always run max_iter iterations (no host sync inside)      */
  int32_t want_llr;     /* 1: keep posterior LLRs so llr_out can be filled              */
  int32_t cn_variant;   /* 0 = default; kernel selection knob for benchmarking          */
  int32_t layer_checks; /* 0: the reference's flooding schedule.  L > 0 (MINSUM_Q88, frames
                           that fit LDS; L a multiple of 64, m/8 <= L <= 1024): layered
                           schedule -- an iteration walks the checks in layers of L
                           consecutive rows; the checks of a layer read the sums as the
                           previous layer left them and their new messages replace the old
                           ones in the sums before the next layer reads.  Every message is
                           computed as dec() computes it (test_code_min_sum_B_fixed.c:213-283);
                           pre-check and exit test are dec()'s (:198-207, :298-313).  An
                           alternative schedule with NO counterpart in the
                           reference: results differ from dec()'s by design (it converges
                           in fewer iterations); checked bit for bit against
                           oracle/ldpc_oracle.c ora_dec_minsum_fixed_layered().          */
  int32_t reserved[9];
} ldpc_opts;

typedef struct ldpc_decoder ldpc_decoder;

int  ldpc_decoder_create(const ldpc_code *c, const ldpc_opts *opts, ldpc_decoder **out);
void ldpc_decoder_free(ldpc_decoder *d);
int  ldpc_decoder_max_frames(const ldpc_decoder *d);

/*
 * Batched form of  int dec(q0[], s[], loop_max, x[])  (all binary variants)
 * and of dec(logfna, z, loop_max) (GFq_LDPC.c:356, GFq_LDPC_NTT.c:437).
 *
 *  llr       [nframes][n]      binary; llr_dtype one of F64/F32/I16/I32:
 *                                MINSUM/SPA float: F64 or F32 values;
 *                                MINSUM_Q88: I16 raw Q8.8 (value*256), or F64/F32
 *                                  values which are put on the Q8.8 grid with the
 *                                  reference's float_to_fix() (_fixed.c:43-53) as
 *                                  its bsc() does (:117-120);
 *                                SPA_TABLE_Q14: I32 raw Q8.14 (float2fix,
 *                                  test_code_B.c:36-43) or F64 values (converted
 *                                  with float2fix).
 *            [nframes][n][q]   non-binary: log-probabilities per symbol value;
 *                                GF2P_WHT: I32 raw log2-domain Q.14
 *                                  (GFq_LDPC_NTT.c:312) or F64 log2 values;
 *                                GFQ_DIRECT: F64/F32 natural-log values.
 *  syndrome  [nframes][m] one symbol per byte, or NULL = all-zero syndrome
 *  bits_out  [nframes][n] one symbol per byte   (reference global tmp_bit / tmp_x)
 *  iters_out [nframes]  iterations executed: loop+1 at the successful exit,
 *                       0 for MINSUM_Q88's "already a codeword" pre-check
 *                       (_fixed.c:198-207), max_iter when not converged
 *  synd_out  [nframes][m] H * bits_out          (reference global tmp_s / tmp_z)
 *  status_out[nframes]  0 converged, 1 not (dec()'s return value)
 *  llr_out   [nframes][n] float posterior sums of the iteration that produced
 *                       bits_out (binary algorithms; needs opts.want_llr)
 * Any output pointer may be NULL.  Host pointers.  Returns LDPC_OK or <0.
 */
int ldpc_decode_batch(ldpc_decoder *d, int nframes, const void *llr, int llr_dtype,
                      const uint8_t *syndrome, int max_iter,
                      uint8_t *bits_out, int32_t *iters_out, uint8_t *synd_out,
                      int32_t *status_out, float *llr_out);

/* Same with DEVICE pointers (buffers resident in HBM) on `stream`
 * (a hipStream_t, NULL = default stream).  nframes <= ldpc_decoder_max_frames().
 * Asynchronous when opts.early_exit == 0; otherwise returns after the pass. */
int ldpc_decode_batch_device(ldpc_decoder *d, int nframes, const void *llr, int llr_dtype,
                             const uint8_t *syndrome, int max_iter,
                             uint8_t *bits_out, int32_t *iters_out, uint8_t *synd_out,
                             int32_t *status_out, float *llr_out, void *stream);

/* ldpc_decode_batch_device() for callers that cannot allocate device memory -- code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has NO
 * counterpart in the reference (whose dec() works in file-scope host arrays).  Same arguments,
 * same results; but every byte of device scratch the call needs comes from one file-scope
 * `static __device__` array of 32 MiB that each GPU the library runs on holds until the
 * process ends (its per-device address is resolved once, in ldpc_decoder_create()).  The call
 * makes no hipMalloc / hipFree / allocating library call.  Binary algorithms only.
 * What comes out of the array, per decoder:
 *   frame-resident Q8.8 / float min-sum:  the frame-queue head (4 bytes); the binary64 decoder on
 *     codes whose channel LLRs do not fit LDS beside the sums (PEG-16000) also keeps one placed
 *     copy of q0 per workgroup there, n * 8 bytes each: min(nframes, workgroups) of them must fit;
 *   staircase Q8.8 (DVB-S2 long frames):  the queue head plus 10 bytes per check and workgroup
 *     (channel values of the staircase variables, two sets of minima rows: 320 KiB per workgroup
 *     for dvb_H_0.5) -- the grid is CAPPED at the workgroups whose rows fit (102 of 256 for that
 *     code), so throughput is below ldpc_decode_batch_device()'s there; results are the same;
 *   staircase binary32 min-sum (MINSUM_F32 on DVB-S2 long frames):  the queue head plus 28 bytes per check
 *     (padded to whole rounds of 1 024) and workgroup -- channel values of the staircase variables and two
 *     sets of 12-byte check records: 896 KiB per workgroup for dvb_H_0.5 -- the grid is capped the same
 *     way (36 of 256 workgroups);
 *   two-pass decoders:  edge-message store, hard decisions and per-frame state,
 *     64 * ceil(nframes / SF) * (16 n + 16 E + n + 2 m + ...) bytes (SF = 512 frames for int16
 *     messages, 256 for binary32 / int32, 128 for binary64), which must fit the 32 MiB; their
 *     max_iter is at most 254 and they never take the second stage of the two-stage early exit.
 * A size beyond these limits is refused with LDPC_ERR_ARG and a message naming the figure.
 * Per-kernel timing (ldpc_decoder_profile, LDPC_HIP_FUSED_TIMING) is off inside the call.
 * All calls on one GPU share that GPU's array; the call clears what must start at zero.  Calls on
 * one GPU must therefore not overlap, and the library refuses (LDPC_ERR_ARG) the overlaps it can
 * see: a call that arrives while another is inside the library (two host threads, two decoders),
 * and a call on a stream other than the previous call's while that call's kernels are still
 * running (checked with an event recorded at the end of every call outside graph capture).
 * Successive calls on ONE stream are ordered by the stream and always served.  Work replayed from
 * a captured graph is invisible to these checks: ordering graph launches is the caller's. */
int ldpc_decode_batch_device_noalloc(ldpc_decoder *d, int nframes, const void *llr, int llr_dtype,
                                     const uint8_t *syndrome, int max_iter,
                                     uint8_t *bits_out, int32_t *iters_out, uint8_t *synd_out,
                                     int32_t *status_out, float *llr_out, void *stream);

/* Single-frame MINSUM_Q88 decode that also writes the Verilog message trace
 * of test_code_min_sum_B_fixed.c (:205 "already cw", :245-260 the v2c dump,
 * :288-296 the c2v dump; one "%08x" line per message per iteration) to
 * trace_path.  circ_size is the reference's global of that name (:40, 128
 * there).  The messages are read back from the GPU after each pass; the host
 * only orders and formats them.  Returns LDPC_ERR_UNSUPPORTED for a code on
 * which the reference's dump loops index past a row (its QC assumption).
 * Host pointers; synchronous. */
int ldpc_decode_trace_q88(ldpc_decoder *d, const void *llr, int llr_dtype, const uint8_t *syndrome,
                          int max_iter, uint8_t *bits_out, int32_t *iters_out, uint8_t *synd_out,
                          int32_t *status_out, const char *trace_path, int circ_size);

/* Counters of the last ldpc_decode_batch*_ call (for benchmarks):
 * iterations actually executed per pass, kernel launches. */
int ldpc_decoder_last_iterations(const ldpc_decoder *d);

/* Diagnostics of the fused Q8.8 decoder's LDS placement (no counterpart in the
 * reference): builds the schedule for workgroups of `block` threads and returns
 * stats[5] = {extra LDS cycles from bank conflicts per pass over the graph,
 * half-wave accesses per pass, LDS words used, slots per check, self-check
 * errors (must be 0)}.  optimise = 0 gives the natural layout for comparison.
 * Host only; needs no GPU. */
int ldpc_code_lds_schedule_stats(const ldpc_code *c, int block, int optimise, int64_t *stats);

/* Diagnostics of the variable-pass placement of the fused floating-point min-sum
 * decoder (no counterpart in the reference): builds the LDS schedule for workgroups
 * of `block` threads, re-ranks the variables within their banks and returns
 * stats[4] = {modelled extra LDS cycles of the variable pass per iteration before,
 * after, 16-lane group reads per iteration, self-check errors of the rewritten
 * schedule (must be 0: every variable still in its bank, on its own word, and every
 * check still visiting exactly its variables)}.  Host only; needs no GPU. */
int ldpc_code_vpass_place_stats(const ldpc_code *c, int block, int64_t *stats);

/* Diagnostics of the GF(2^p) decoder's compact tables (no counterpart in the reference): the
 * kernels evaluate Fexp / Flog (GFq_LDPC_NTT.c:196-247) from an 80 KiB LDS image instead of the
 * reference's 12 MiB of tables; this replays their arithmetic on the host against every entry of
 * both full tables.  stats[4] = {Fexp entries that differ (must be 0), Flog entries that differ
 * without having been handed back to the full table (must be 0), Flog entries handed back,
 * Flog entries kept verbatim}.  Returns the number of differing entries.  Host only. */
int64_t ldpc_wht_tables_selfcheck(int64_t *stats);

/* The same for the binary table sum-product (no counterpart in the reference): its check pass reads
 * Flogtanh / Fgallag (inittab(), test_code_B.c:52-90) from a 153 KiB LDS image built from the two
 * 16 MiB tables; this replays the kernels' arithmetic on the host over all 2^22 arguments of both.
 * stats[4] = {1 if the tables had the structure the image needs (else the decoder gathers from the
 * full tables as before), Flogtanh entries that differ, Fgallag entries that differ, image bytes}.
 * Returns the number of differing entries (must be 0), -1 if the image was refused.  Host only. */
int64_t ldpc_spa_tables_selfcheck(int64_t *stats);

/* Per-kernel timing for benchmarks (no counterpart in the reference, whose only
 * instrumentation is clock() around the trial loop, test_code_B.c:334,354-356).
 * With profiling enabled every launch of the decoding kernels is bracketed by a
 * hipEvent pair on the caller's stream.  ldpc_decoder_profile_read() waits for
 * them and returns, since the previous read, ms[4] / launches[4] summed per
 * kernel class: 0 variable-node pass, 1 check-node pass, 2 fused
 * whole-decode kernel, 3 other. */
int ldpc_decoder_profile(ldpc_decoder *d, int enable);
int ldpc_decoder_profile_read(ldpc_decoder *d, double *ms, int32_t *launches);
/* Companion of ldpc_decoder_profile_read() (no counterpart in the reference): per kernel class, the sum
 * over the launches that read covered of the frames each launch worked on (the second stage of a
 * two-stage early exit launches on fewer frames than the first), so that a benchmark can price a
 * class's launches by the bytes they really had to move.  frame_launches[4]. */
int ldpc_decoder_profile_frames(ldpc_decoder *d, int64_t *frame_launches);

/* ------------------------------------------------ Monte-Carlo loop on the device
 * The trial loop of the reference's *_cw_noise_gen harnesses
 * (test_code_min_sum_B.c:307-323, test_code_min_sum_B_fixed.c:415-434,
 * test_code_B_float.c:299-316, test_code_no_mex_B.c:345-372) with every step on
 * the GPU: source word (x[j] = rand() & 1), s = enc(x) (:69-79), channel and LLRs
 * (bsc(), :41-67), dec(), and the classification (dec()'s return value, HamDist
 * :81-90).  Nothing but eight counters crosses PCIe.
 *
 * Random numbers are counter-based: (seed, global frame index) fixes a frame's
 * source word and noise whatever the batch size, the number of GPUs or the order
 * of the passes -- shard a run by giving each rank its own first_frame. */
typedef enum ldpc_channel {
  LDPC_CH_BSC_EXACT = 0, /* the reference's bsc(): exactly round(n p) distinct flips, LLR
                            magnitude log((1-p')/p') with p' = flips / n                   */
  LDPC_CH_BSC_IID   = 1, /* independent flips with probability p                           */
  LDPC_CH_AWGN      = 2, /* BPSK 1-2x plus N(0, sigma^2); LLR = 2 y / sigma^2              */
  LDPC_CH_QLC       = 3  /* channel() of test_code_B_MSDP.c:411-440: 4 bits per flash cell,
                            Gray mapped, read drawn from a confusion matrix, bit LLRs from
                            assign_llr_one_sym() (:339-395); tables: ldpc_mc_set_qlc_channel */
} ldpc_channel;
typedef enum ldpc_source {
  LDPC_SRC_RANDOM = 0,   /* a fresh random word per frame (the mex harnesses)              */
  LDPC_SRC_FIXED  = 1,   /* one random word for the whole run (test_code_no_mex_B.c:339-343) */
  LDPC_SRC_ZERO   = 2    /* the all-zero word                                              */
} ldpc_source;
typedef struct ldpc_mc_opts {
  int32_t  channel;      /* ldpc_channel                                                   */
  int32_t  source;       /* ldpc_source                                                    */
  int32_t  llr_dtype;    /* what the decoder is fed: F64 / F32 values, I16 = Q8.8 raw through
                            float_to_fix() (_fixed.c:117-120), I32 = Q8.14 raw through
                            float2fix() (test_code_B.c:122-125)                            */
  int32_t  max_iter;     /* dec()'s loop_max                                               */
  double   param;        /* BSC: p;  AWGN: sigma                                           */
  uint64_t seed;
  uint64_t first_frame;  /* global index of the run's first frame                          */
  int32_t  reserved[8];
} ldpc_mc_opts;
typedef struct ldpc_mc ldpc_mc;

/* Buffers for passes of ldpc_decoder_max_frames(d) frames on `device` (the
 * decoder's device).  The code and the decoder must outlive the context. */
int  ldpc_mc_create(const ldpc_code *c, ldpc_decoder *d, int device, ldpc_mc **out);
void ldpc_mc_free(ldpc_mc *mc);
/* Tables of LDPC_CH_QLC from p_rec_given_sent[rows][16] (P(i received | j sent), columns
 * already normalised as test_code_B_MSDP.c:455-465 does; rows = 16 for num_reads 1, 46
 * for 3): make_p_sent_given_rec_T() (:297-323) and the per-page bit LLRs (:339-395). */
int  ldpc_mc_set_qlc_channel(ldpc_mc *mc, const double *p_rec_given_sent, int num_reads);
/* Runs nframes trials; counters[8] = {frames, not converged (errors[0]),
 * converged to a word != x (errors[1]), bit errors over all frames, iterations
 * executed, frames with any bit error, bit errors of converged frames, 0}. */
int  ldpc_mc_run(ldpc_mc *mc, const ldpc_mc_opts *o, int64_t nframes, uint64_t *counters);
/* Generation only, for inspection: frames first_frame .. +nframes (at most one
 * pass) copied to host x[nframes][n], llr[nframes][n] of llr_dtype, synd[nframes][m];
 * any pointer may be NULL. */
int  ldpc_mc_generate(ldpc_mc *mc, const ldpc_mc_opts *o, int nframes, uint8_t *x, void *llr, uint8_t *synd);
/* The last pass of ldpc_mc_run copied to the host (its first nframes frames). */
int  ldpc_mc_last_pass(ldpc_mc *mc, int nframes, uint8_t *x, uint8_t *bits, int32_t *status, int32_t *iters);

/* ------------------------------------------- the GF(q) trial loop on the device
 * The trial loop of GFq_LDPC() (GFq_LDPC.c:620-650; WD_farm/GFq_LDPC.c:600-630) with every step on the
 * GPU, as ldpc_mc_* is for the binary harnesses: x[i] = rand() % Q (:626), s = enc(x) (:632, :299-310),
 * channel() (:213-231: the read y walks the column of the sent symbol in the confusion matrix with a
 * percent-granular uniform draw) and assign_llr() (:199-211: logfna[.][i] = log P(sent | y), floor -100),
 * dec() (GFQ_DIRECT or GF2P_WHT decoder `d`), classification (:641-649).  Eight counters cross PCIe.
 * Draws are counter-based, keyed by (seed, global frame index): frames do not depend on the pass size or
 * on the number of ranks (shard a run through first_frame). */
typedef struct ldpc_gfmc ldpc_gfmc;
int  ldpc_gfmc_create(const ldpc_code *c, ldpc_decoder *d, int device, ldpc_gfmc **out);
void ldpc_gfmc_free(ldpc_gfmc *mc);
/* p_rec_given_sent[q * num_reads][q], P(read r | sent a), columns already normalised as GFq_LDPC() does
 * (:548-558): make_p_sent_given_rec_T() (:177-197) and the log / -100 floor of assign_llr(). */
int  ldpc_gfmc_set_channel(ldpc_gfmc *mc, const double *p_rec_given_sent, int num_reads);
/* counters[8] = {frames, not converged (errors[0]), converged to a word != x (errors[1]), symbol errors,
 * iterations executed, frames with any symbol error, symbol errors of converged frames, 0}. */
int  ldpc_gfmc_run(ldpc_gfmc *mc, int max_iter, int64_t nframes, uint64_t seed, uint64_t first_frame,
                   uint64_t *counters);
/* Generation only, for inspection (at most one pass): x[nframes][n] one symbol per byte, the reads
 * y[nframes][n], the uniform draws sel[nframes][n] behind them, logfna[nframes][n][q], synd[nframes][m];
 * any pointer may be NULL.  GFq_LDPC_experiment() restates the same frames on the host. */
int  ldpc_gfmc_generate(ldpc_gfmc *mc, uint64_t seed, uint64_t first_frame, int nframes, uint8_t *x,
                        int32_t *y, float *sel, double *logfna, uint8_t *synd);
/* The last pass of ldpc_gfmc_run copied to the host (its first nframes frames). */
int  ldpc_gfmc_last_pass(ldpc_gfmc *mc, int nframes, uint8_t *x, uint8_t *xhat, int32_t *status, int32_t *iters);

/* ------------------------------------------- reference-shaped entry points
 * Same names, argument order and meaning as the reference's C harness
 * functions; `errors` is the caller's double[2] = {not converged, converged to
 * a word != x}.  They return LDPC_OK or a negative code instead of void.
 * The graph/decoder for `code_filename` is cached per process and device 0 is
 * used (set LDPC_HIP_DEVICE to override).  The cached decoders are shared, so
 * these entry points are serialised per process by one lock (the reference's
 * functions work in file-scope globals and are not re-entrant either); for
 * concurrent decoding use one ldpc_decoder per host thread.
 * LDPC_HIP_DEVICES = "all" or a list ("0,1,2,3"): the Monte-Carlo harnesses (*_cw_noise_gen,
 * test_code_B) split the trials of ONE call over the listed GPUs -- a worker thread, decoder and
 * generator per entry, counters added up; the result does not depend on the list.            */

/* test_code_min_sum_B(), test_code_min_sum_B.c:331-359 (single frame, caller's x,s,y,q0) */
int test_code_min_sum_B(const char *code_filename, int iteration, int *x, int *s, int *y,
                        double *q0, double *errors);
/* test_code_min_sum_B_cw_noise_gen(), test_code_min_sum_B.c:288-329 (Monte Carlo on a BSC) */
int test_code_min_sum_B_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                     double p_bsc, double *errors);
/* test_code_min_sum_B() of test_code_min_sum_B_fixed.c:442-473.  The reference
 * reuses the float function's name with one more argument; in one C library
 * the symbol needs its own name.  q0 must hold Q8.8 values (k/256 in [-128, 127], what the
 * reference's bsc() :117-120 produces): the reference passes q0 to dec() unquantised, the
 * int16 decoder is its arithmetic exactly on such values only, and any other q0 is refused
 * with LDPC_ERR_UNSUPPORTED rather than rounded.  Verilog_sim_filename may be NULL (no trace);
 * otherwise the message trace of _fixed.c:245-260,:288-296 is written (the
 * trace is produced by a host replay).                  */
int test_code_min_sum_B_fixed(const char *code_filename, int iteration, int *x, int *s, int *y,
                              double *q0, double *errors, const char *Verilog_sim_filename);
/* test_code_min_sum_B_cw_noise_gen() of test_code_min_sum_B_fixed.c:396-440 */
int test_code_min_sum_B_fixed_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                           double p_bsc, double *errors);
/* test_code_B_float() / _cw_noise_gen(), test_code_B_float.c:280-351 */
int test_code_B_float(const char *code_filename, int iteration, int *x, int *s, int *y,
                      double *q0, double *errors);
int test_code_B_float_cw_noise_gen(const char *code_filename, int iteration, int trials,
                                   double p_bsc, double *errors);
/* test_code_B(), test_code_no_mex_B.c:321-381 (table sum-product, Monte Carlo;
 * x is drawn once, as that driver does at :339-343).  The mex variant
 * (test_code_B.c:316-360) is the same call with H_file = "peg_16000_3_0.9.txt". */
int test_code_B(int iteration, int trials, double p_bsc, const char *H_file, double *errors);
/* test_code_B_MSDP(), test_code_B_MSDP.c:442-543, decode_mode 0 (random data through
 * the QLC confusion-matrix channel, float sum-product dec()); the whole loop runs on the
 * device.  The reference hard-codes "peg_16000_3_0.9.txt" (:467); here the code file is
 * an argument.  p_rec_given_sent is [16 or 46][16] row-major and is column-normalised in
 * place like the reference (:455-465).  decode_mode 1 replays snowbird_sym.bin, which
 * the reference does not ship: LDPC_ERR_UNSUPPORTED. */
int test_code_B_MSDP(const char *code_filename, int iteration, int num_trials, int num_reads,
                     int decode_mode, double *p_rec_given_sent, double *errors);
/* GFq_LDPC(), GFq_LDPC.c:540-655, decode_mode 0 (synthetic symbols through the
 * confusion-matrix channel).  The reference hard-codes "547_NB.txt"; here the
 * code file is an argument.  p_rec_given_sent is [q*num_reads][q] row-major
 * and is column-normalised in place like the reference (:548-558).         */
int GFq_LDPC(const char *code_filename, int iteration, int num_trials, int num_strs_seek,
             int num_reads, int decode_mode, double *p_rec_given_sent, double *errors);

/* The inputs of GFq_LDPC()'s trials for a given generator state, for inspection: per frame
 * x[n] (x[i] = rand() % Q, :626), s[m] = enc(x) (:632), the reads y[n] drawn by channel()
 * (:213-231), logfna[n][q] = log P(a sent | y_i) with the -100 floor (assign_llr, :199-211,
 * tables from make_p_sent_given_rec_T, :177-197) and the percent-granular uniform draws
 * rand_select[n] behind y (the reference takes them from rand(), seeded from the clock, so
 * the stream itself cannot be matched; the channel law can, draw by draw).  Frame f of this call
 * is frame f of ldpc_gfmc_generate / ldpc_gfmc_run for the same seed (counter-based draws): the
 * host restatement of what the device front-end decodes.  p_rec_given_sent
 * is column-normalised in place.  s, y, rand_select may be NULL.  Host only; needs no GPU. */
int GFq_LDPC_experiment(const char *code_filename, int num_reads, double *p_rec_given_sent,
                        uint64_t seed, int nframes, int32_t *x, int32_t *s, int32_t *y,
                        double *logfna, float *rand_select);

/* GFq_LDPC() of WD_farm/GFq_LDPC.c:477 (mex gateway :635-709): same harness, but the graph is
 * passed in as the flat arrays WD_farm/InitSumProductDecoderSyndrome.m builds, row-major and
 * 0-based as that function indexes them (:310, :370): row_list_col[nchk][rmax], H[nchk][rmax]
 * (coefficients 1..q-1), col_list_row / col_list_N[nvar][cmax] (row and position-in-row of each
 * variable's k-th edge), col_weight[nvar], row_weight[nchk].  The column-side arrays are
 * redundant with the rows; they are checked against them (NULL = not checked).  q replaces the
 * reference's compile-time `#define Q`.  A distinct symbol: the reference reuses the name
 * GFq_LDPC for both arities. */
int GFq_LDPC_graph(int q, int iteration, int num_trials, int num_strs_seek, int num_reads,
                   int decode_mode, double *p_rec_given_sent, int nvar, int nchk, int cmax_in,
                   int rmax_in, const int *col_weight_in, const int *row_weight_in,
                   const int *row_list_col_in, const int *col_list_row_in, const int *col_list_N_in,
                   const int *H_in, double *errors);

/* main() of GFq_LDPC_NTT.c:641-703, the GF(2^p) Walsh-Hadamard program
 *   GFq_LDPC_NTT [-iter num] <lap|bsc> NoiseLevel ParityMatrix        (:656-663)
 * as a function: three experiments, SRand(j) for j = 1..3 (:681-695), each
 * x[i] = Rand() % Q, s = enc(x), channel(x, y, level, logfna), dec(logfna, s,
 * iteration, x); the three frames are decoded as one batch on the GPU.
 * verdict[3] = dec()'s return value per experiment, 0 ("converged.") or -1
 * ("failed."), which is all the program reports (:691-694); iters[3] (may be
 * NULL) = iterations executed.  The reference fixes Q at compile time (-DLog2Q);
 * here it is the field order in the matrix file's header.  Its -iter default
 * is 1000 (:647).  lap()'s statistics printout (:357-364) is not produced. */
int GFq_LDPC_NTT(int iteration, const char *channel, double level, const char *matrix_file,
                 int32_t *verdict, int32_t *iters);
/* The inputs of one such experiment (:683-687) for SRand(seed): x[n], s[m], the
 * side information y[n] and logfna[n][q] (log2 domain, Q.14, float2fix :177-184).
 * Rand() is a 31-bit LCG (:163-167), so these are the reference's values exactly.
 * Host only; needs no GPU. */
int GFq_LDPC_NTT_experiment(const char *matrix_file, int seed, const char *channel, double level,
                            int32_t *x, int32_t *s, int32_t *y, int32_t *logfna);

/* Seed of the library's Monte-Carlo generator (the reference uses
 * srand(time(NULL)); 0 = seed from the clock like the reference). */
void ldpc_set_seed(uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
