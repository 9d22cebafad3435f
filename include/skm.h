/*
 * skm.h -- C ABI of libskm.so: the MI355X (gfx950) implementation of SIDEKIT's GMM-UBM /
 * i-vector sufficient-statistics hot path.
 *
 * The reference (Anwarvic/Speaker-Recognition = SIDEKIT 1.3.2) has no FFI for this path: it
 * is NumPy called from Python methods.  Each entry point below names the
 * reference call site(s) whose arithmetic it replaces; the Python classes in
 * speaker-recognition_amd/ (Mixture, StatServer, FactorAnalyser) keep the reference's method
 * surface and call these through ctypes.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative code otherwise; skm_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - "dev" pointers are device (HBM) addresses on the ctx's GPU, "host" pointers are ordinary
 *     host memory; the library never frees or keeps caller buffers;
 *   - all work is enqueued on the hipStream_t passed as `stream` (NULL = default stream) and is
 *     asynchronous unless stated; workspace sizes are queried, the launch functions allocate
 *     nothing;
 *   - float32 arithmetic on the MFMA units, float64 for every cross-tile reduction and output.
 */
#ifndef SKM_H
#define SKM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKM_OK            0
#define SKM_ERR_ARG      -1   /* bad argument (shape, alignment, NULL)            */
#define SKM_ERR_HIP      -2   /* a HIP runtime call failed                         */
#define SKM_ERR_WORKSPACE -3  /* workspace too small                               */
#define SKM_ERR_UNSUPPORTED -4/* shape outside what the kernels were built for     */

int         skm_version(void);
const char *skm_last_error(void);

/* One context per GPU: device id, CU count, launch geometry.  No device memory is owned. */
int skm_ctx_create(int device, void **ctx);
int skm_ctx_destroy(void *ctx);
int skm_ctx_cu_count(void *ctx);

/* Per-kernel timing for bench.py's roofline: when enabled, every kernel launch of the library is
 * bracketed by HIP events recorded on the launch stream.  skm_prof_read waits for them, writes
 * one text line "kernel_name launches total_ms" per kernel into buf and clears the records;
 * returns the number of bytes written.  No counterpart in the reference (it has no profiling). */
int  skm_prof_enable(void *ctx, int on);
long skm_prof_read(void *ctx, char *buf, size_t cap);

/* ------------------------------------------------------------------ diagonal GMM ------ */

/* Bytes of the packed, MFMA-fragment-ordered GMM image for C components of dimension D. */
size_t skm_gmm_packed_bytes(int C, int D);

/* Packs host float64 parameters (w[C], mu[C*D], invcov[C*D], row-major) into the device image
 * used by every kernel below: [-1/2 invcov ; mu*invcov] in MFMA B-fragment order plus the per
 * component constant -1/2 A_c, all pre-scaled by log2(e).
 * Replaces Mixture._compute_all (sidekit/mixture.py:382-396) and the operand set-up of
 * Mixture.compute_log_posterior_probabilities (sidekit/mixture.py:473-483).
 * Synchronous with respect to `stream` on return (the host staging buffer is freed). */
int skm_gmm_pack(void *ctx, const double *w_host, const double *mu_host, const double *invcov_host,
                 int C, int D, void *packed_dev, void *stream);

/* The same image RELATIVE TO A REFERENCE GAUSSIAN R = N(0, diag(1 / ref_invvar)) (ref_invvar_host: D doubles, the
 * inverse of the per-dimension variance of all the data or of the whole model; an entry <= 0 leaves its dimension
 * alone; NULL = skm_gmm_pack): square rows -1/2 (invcov - ref_invvar), kc without log R's constant, so the logit
 * the kernels accumulate is log2(w_c N_c(x) / R(x)).  R(x) is common to all components of a frame: posteriors, N, F,
 * S and the Baum-Welch statistics are unchanged, while the float32 partial sums lose their largest part (x^2 / v
 * summed over the dimensions) and with it most of their rounding error -- the reference forms the same logits in
 * float64 (sidekit/mixture.py:482-486).  What is NOT unchanged: skm_gmm_estep's llk is then sum_t log(p(x_t) / R(x_t))
 * and skm_gmm_llk_ref adds sum_t log R(x_t) back from the statistics themselves (T frames; in place, before any
 * skm_gmm_stats_unshift); skm_gmm_loglik / skm_gmm_logprob give per-frame values and want a plain image. */
int skm_gmm_pack_ref(void *ctx, const double *w_host, const double *mu_host, const double *invcov_host,
                     const double *ref_invvar_host, int C, int D, void *packed_dev, void *stream);
int skm_gmm_llk_ref(void *ctx, double *stats_dev, int C, int D, long T, const double *ref_invvar_dev, void *stream);

/* M-step and re-pack on the device, so that an EM iteration needs no host round trip:
 *   stats_dev        [ N | F | S | llk ] as skm_gmm_estep leaves it (after the all-reduce in a multi-GPU run)
 *   cov_var_ctl_dev  C*D float64, the variance-control reference of the model (sidekit/mixture.py:636-640)
 *   params_dev       out: [ w (C) | mu (C*D) | invcov (C*D) | sum N ] float64
 *   packed_dev       out: the image skm_gmm_pack would build from those parameters
 *   llk_mean_dev     out (optional): llk / sum N, the value EM_split appends per iteration (mixture.py:706)
 *   ref_invvar_dev   D doubles on the device or NULL: re-pack relative to that reference Gaussian (skm_gmm_pack_ref)
 * w = N / sum N, mu = F / N, cov = S / N - mu^2 clamped to [floor_cov, ceil_cov] * cov_var_ctl with the reference's
 * inclusive comparisons, invcov = 1 / cov.  Replaces Mixture._maximization + variance_control + _compute_all
 * (sidekit/mixture.py:590-611, 489-506, 382-396) inside the EM loop; float64 throughout. */
int skm_gmm_mstep_pack(void *ctx, const double *stats_dev, const double *cov_var_ctl_dev, int C, int D,
                       double ceil_cov, double floor_cov, double *params_dev, void *packed_dev,
                       double *llk_mean_dev, const double *ref_invvar_dev, void *stream);

/* Workspace bytes for skm_gmm_estep / skm_gmm_loglik / skm_bw_stats over T frames. */
size_t skm_gmm_workspace_bytes(void *ctx, long T, int C, int D);

/* E-step of the EM over T frames X[T*D] (float32, row-major, dev):
 *   out_dev = [ N[C] | F[C*D] | S[C*D] | llk ]  float64, (2*C*D + C + 1) values,
 *   N = sum_t gamma, F = gamma^T X, S = gamma^T X^2, llk = sum_t log sum_c w_c N(x_t).
 * accumulate != 0 adds to out_dev instead of overwriting it.
 * Replaces Mixture._expectation (sidekit/mixture.py:537-568) = compute_log_posterior_probabilities
 * (:461-487) + sum_log_probabilities (:52-64) + the three accumulations (:557-562), and the
 * per-file loop of Mixture._expectation_list (:584-588). */
int skm_gmm_estep(void *ctx, const float *X_dev, long T, int C, int D, const void *packed_dev,
                  double *out_dev, int accumulate, void *ws_dev, size_t ws_bytes, void *stream);

/* skm_gmm_estep for callers that cannot allocate or size a workspace: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Same arguments and results
 * as skm_gmm_estep minus the workspace: the kernels' scratch (per-wave float64 slabs, log-likelihood partials,
 * team hand-off rows) is a file-scope device array of skm_gmm_estep_noalloc_max_bytes() = 144 MiB that every GPU
 * the library runs on holds until the process ends.  The call allocates nothing (no hipMalloc / hipFree; keep
 * skm_prof_enable off, its events are created per launch).
 *   - shapes with a single-sweep kernel (C <= 128 with D <= 63; 129..512 components with D <= 64; C in {1024, 2048,
 *     4096} with D % 4 == 0 and D <= 64; rows of 64..159 features with D % 4 == 0 and C <= 128 or C/128 in
 *     {2, 4, 8, 16}): any T;
 *   - other shapes take two sweeps whose per-frame partials live in the scratch too: T is bounded, and a call
 *     that does not fit returns SKM_ERR_WORKSPACE with the sizes in skm_last_error();
 *   - all calls on one GPU share that GPU's array: what must start at zero is cleared on `stream` by every
 *     call, and calls that overlap in time on one GPU (two streams, two threads) race. */
size_t skm_gmm_estep_noalloc_max_bytes(void);
int skm_gmm_estep_noalloc(void *ctx, const float *X_dev, long T, int C, int D, const void *packed_dev,
                          double *out_dev, int accumulate, void *stream);

/* Per-frame log-likelihood log sum_c w_c N(x_t; mu_c, Sigma_c) (natural log, float32, dev).
 * Replaces compute_log_posterior_probabilities + the log_lk of sum_log_probabilities
 * (sidekit/statserver.py:83-95; used by gmm_scoring, sidekit/gmm_scoring.py:102-111). */
int skm_gmm_loglik(void *ctx, const float *X_dev, long T, int C, int D, const void *packed_dev,
                   float *loglik_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* lp_dev[T*C] (float64, row-major, dev) = ln(w_c N(x_t; mu_c, Sigma_c)), the matrix returned by
 * Mixture.compute_log_posterior_probabilities (sidekit/mixture.py:461-487; its `mu=` override is a
 * second skm_gmm_pack with the replacement means).  loglik_dev (T float32) is optional (NULL) and
 * receives the row log-sum-exp, i.e. the log_lk of sum_log_probabilities (sidekit/mixture.py:52-64).
 * Arithmetic is float32 (MFMA), widened on store.  Only callers that need the T x C matrix itself use
 * this; the statistics entry points never materialise it. */
int skm_gmm_logprob(void *ctx, const float *X_dev, long T, int C, int D, const void *packed_dev,
                    double *lp_dev, float *loglik_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* sum_log_probabilities on a matrix the caller already holds (sidekit/mixture.py:52-64, sidekit/statserver.py:83-95):
 * lp_dev[T*C] float64 -> pp_dev[T*C] = exp(lp - log_lk) (optional, NULL) and loglk_dev[T] = row log-sum-exp (optional);
 * a row whose maximum is not finite gets log_lk = that maximum, NaN propagates as numpy.max does.  float64. */
int skm_row_logsumexp(void *ctx, const double *lp_dev, long T, int C, double *pp_dev, double *loglk_dev, void *stream);

/* Baum-Welch statistics per session.  Session s owns frames [seg_off[s], seg_off[s+1]) of X
 * (seg_off: S+1 int64 on the device, non-decreasing, seg_off[S] <= T).
 *   stat0_dev[S*C]      float64   zero order  (sum_t gamma)
 *   stat1_dev[S*C*D]    float64   first order, component-major (index c*D + d)
 * Rows of empty sessions are set to zero.
 * Replaces the body of StatServer.accumulate_stat (sidekit/statserver.py:659-667). */
int skm_bw_stats(void *ctx, const float *X_dev, long T, const int64_t *seg_off_dev, int S, int C, int D,
                 const void *packed_dev, double *stat0_dev, double *stat1_dev,
                 void *ws_dev, size_t ws_bytes, void *stream);

/* Number of teams skm_bw_stats deals whole sessions to for a (C, D) model on this device, 0 when the shape takes the
 * two-sweep kernels (which split sessions freely).  A host that holds FEW LONG sessions (fewer than a few per team)
 * cuts them into 32-frame-aligned pieces itself, calls skm_bw_stats on the pieces and adds the rows of a session's
 * pieces (speaker-recognition_amd/engine.py: _bw_split_plan).  No counterpart in the reference: its loop over
 * sessions (sidekit/statserver.py:647-667) is serial. */
int skm_bw_team_count(void *ctx, int C, int D);

/* Frame centring for un-normalised features (raw cepstra with c0 / log-energy offsets, no cmvn).  The kernels above
 * accumulate the logits in float32, whose absolute error grows with sum_d mu_d^2 / sigma_d^2; the reference does the
 * same two products in float64 (sidekit/mixture.py:482-486).  The whole path is translation invariant, so a caller
 * whose data sit far from the origin runs it on x' = x - m with a model packed from mu' = mu - m (skm_gmm_pack on the
 * shifted means) and maps the statistics back exactly:
 *   skm_frames_center      out_dev[t*D + d] = X_dev[t*D + d] - (float)shift_dev[d]      (in place when out_dev == X_dev)
 *   skm_gmm_stats_unshift  [N | F' | S' | llk] of skm_gmm_estep, in place:  F = F' + N m,  S = S' + 2 m F' + N m^2
 *                          (mixture.py:557-562 on the un-shifted frames); N and llk do not change
 *   skm_bw_unshift         stat1[s][c*D + d] += stat0[s][c] * m[d]                      (statserver.py:664-666)
 * with m[d] = (double)(float)shift_dev[d] in all three (shift_dev: D doubles on the device), so that the frames, the
 * model and the correction use the same number.  speaker-recognition_amd/mixture.py decides when to centre (mean of the
 * model or of the data further than half a standard deviation from the origin in some dimension). */
int skm_frames_center(void *ctx, const float *X_dev, long T, int D, const double *shift_dev, float *out_dev,
                      void *stream);
int skm_gmm_stats_unshift(void *ctx, double *stats_dev, int C, int D, const double *shift_dev, void *stream);
int skm_bw_unshift(void *ctx, const double *stat0_dev, double *stat1_dev, long S, int C, int D,
                   const double *shift_dev, void *stream);

/* Mean of v over each segment [seg_off[s], seg_off[s+1]) (seg_off: S+1 int64, dev), float64 sums:
 * out_dev[s] = mean(v[seg_off[s] : seg_off[s+1]]); an empty segment gives NaN (numpy's mean of nothing).
 * With v = skm_gmm_loglik's output this is the per-segment mean frame log-likelihood `log_lk.mean()` of
 * gmm_scoring_singleThread (sidekit/gmm_scoring.py:105-107 and :114-116). */
int skm_segment_mean(void *ctx, const float *v_dev, const int64_t *seg_off_dev, int S, double *out_dev,
                     void *stream);

/* Column sums of X and X^2 over T frames: out_dev = [ sum x (D) | sum x^2 (D) ] float64.
 * Replaces the two reductions of Mixture._init (sidekit/mixture.py:631-632).
 * ws_dev needs skm_moments_workspace_bytes(ctx, D) bytes. */
size_t skm_moments_workspace_bytes(void *ctx, int D);
int skm_moments(void *ctx, const float *X_dev, long T, int D, double *out_dev,
                void *ws_dev, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ total variability ---- */

/* GEMM kinds of skm_gemm: input/compute type -> storage type of C. */
#define SKM_GEMM_F32_F32    0  /* float MFMA (v_mfma_f32_32x32x2_f32), C float                         */
#define SKM_GEMM_F32_F64ACC 1  /* float MFMA, accumulators folded into float64 every 512 k, C double   */
#define SKM_GEMM_F64_F64    2  /* double MFMA (v_mfma_f64_16x16x4_f64), C double                        */
#define SKM_GEMM_F64_F32OUT 3  /* double MFMA, C float                                                  */
#define SKM_GEMM_F32_F64OUT 4  /* float MFMA, plain float accumulation (short K), C double               */
#define SKM_GEMM_F32S_F32   5  /* float MFMA on 16x16x4 tiles (64 x 64 workgroup tiles: batches of small matrices), C float */
#define SKM_GEMM_F32S_F64OUT 6 /* the same 64 x 64-tile float kernel, plain float accumulation, C double (ragged rank-sized products) */

/* Strided-batched row-major GEMM  C[z] = alpha op(A[z]) op(B[z]) + beta C[z]  (all pointers dev; trans
 * flags mean "stored transposed").  Replaces the numpy.dot call sites of the TV path that are plain
 * products, e.g. the minimum-divergence update F.dot(ch) (sidekit/factor_analyser.py:665). */
int skm_gemm(void *ctx, int kind, const void *A, const void *B, void *C, int M, int N, int K, long lda, long ldb,
             long ldc, int transA, int transB, double alpha, double beta, int batch, long sA, long sB, long sC,
             void *stream);

/* C = A B (+ C when accumulate) for row-major float32 A[M*K], B[K*N] and C[M*N] float (c_f64 = 0) or double, computed
 * on the bf16 matrix cores from operands split into three bf16 terms each (x = x1 + x2 + x3 exactly; six of the nine
 * partial products are kept, the dropped ones are below 2^-26 of the product; float32 accumulation): float32-grade
 * products at 6/16 of the f32 MFMA time.  This is how skm_tv_estep forms L_s = I + sum_c N_sc G_c
 * (sidekit/factor_analyser.py:87) and A_c += sum_s N_sc e_hh (sidekit/factor_analyser.py:124) unless SKM_TV_SPLIT=0;
 * the entry point itself serves tests and timing.  ws_dev: skm_gemm_split_workspace_bytes(M, N, K) bytes. */
size_t skm_gemm_split_workspace_bytes(int M, int N, int K);
int skm_gemm_split(void *ctx, const float *A_dev, const float *B_dev, void *C_dev, int M, int N, int K, int c_f64,
                   int accumulate, void *ws_dev, size_t ws_bytes, void *stream);

/* Workspace bytes of skm_tv_estep (S sessions, its `chain` argument) / skm_tv_mstep (pass S = 0, chain = 0) for a
 * given batch. */
size_t skm_tv_workspace_bytes(void *ctx, int C, int D, int R, long S, int batch, int chain);

/* Row stride (floats) of the augmented zero-order matrix naug = [stat0 | 1 | zero pad]: C + 1 rounded up
 * to a multiple of 4 so that rows stay 16-byte aligned for the GEMM's vector loads. */
int skm_tv_naug_stride(int C);

/* Centre and whiten first-order statistics and build the augmented zero-order matrix:
 *   ftilde[S*C*D] float32 = (stat1 - stat0[:,c] * mu) * sqrt(invcov)   with the reference's rounding (in
 *   place in the INPUT dtype: float32 statistics are rounded to float32 after each of the two updates),
 *   naug[S*skm_tv_naug_stride(C)] float32 = [stat0 | 1 | 0-pad].
 * in_f64 = 0: stat0/stat1 are float32 (statistics read from a StatServer file), 1: float64.
 * Replaces sidekit/factor_analyser.py:76-79 (and StatServer.whiten_stat1, statserver.py:743-745). */
int skm_tv_whiten(void *ctx, const void *stat0_dev, const void *stat1_dev, int in_f64, long S, int C, int D,
                  const double *mu_dev, const double *sqrt_invcov_dev, float *ftilde_dev, float *naug_dev,
                  void *stream);

/* Once per EM iteration: T32 = float(T) and the Gram matrices G_c = T_c' T_c (G_dev[C*R*R] float32, only
 * the blocks on/above the 32-block diagonal are written).  T_dev is the CD x R loading matrix (float64).
 * Replaces the per-session product (F.T * stat0[idx, index_map]).dot(F) of factor_analyser.py:87 by
 * L_s = I + sum_c N_sc G_c. */
int skm_tv_prepare(void *ctx, const double *T_dev, int C, int D, int R, float *T32_dev, float *G_dev, void *stream);

/* TV E-step over S sessions (batches of `batch`):
 *   e_h_dev[S*R]        float64  posterior means (= the i-vectors)
 *   iv_sigma_dev[S*R]   float64  diag(L^-1) + e_h^2, or NULL        (factor_analyser.py:142)
 *   A_aug_dev[(C+1)*R*R] float64 += N' e_hh per component; row C accumulates sum_s e_hh (= _R); full
 *                       [R][R] storage, only blocks on/above the 32-block diagonal are touched; or NULL
 *   Cacc_dev[R*C*D]     float64 += e_h' ftilde; or NULL (A_aug and Cacc go together: training)
 *   chain               storage / arithmetic type of the per-session matrices L_s, their Cholesky factors and
 *                       inverses (the reference: float64, scipy.linalg.inv): SKM_TV_CHAIN_F32 costs cond(L_s) * 6e-8
 *                       on e_h and is 1.5 times faster on that part, SKM_TV_CHAIN_F64 is for long sessions / large
 *                       loadings; SKM_TV_PRECISE (T64_dev = the CD x R float64 loading matrix, NULL otherwise) also
 *                       forms G_c, L_s and f~ T in float64 -- with float32 products e_h is off by ~cond(L_s) * 1e-8
 *                       whatever the chain (measured: cond 3e5 -> 3e-3 normwise, 3e6 -> cosine 0.9993) -- at about
 *                       half the speed; FactorAnalyser chooses per call from a bound on cond(L_s)
 * Replaces e_on_batch (factor_analyser.py:52-92), e_gather (:109-128), iv_extract_on_batch (:131-142). */
#define SKM_TV_CHAIN_DEFAULT 0   /* float32, or what the SKM_TV_CHAIN environment variable says */
#define SKM_TV_CHAIN_F32     1
#define SKM_TV_CHAIN_F64     2
#define SKM_TV_PRECISE       3   /* float64 Gram matrices, L build, aux product and chain (needs T64_dev) */
int skm_tv_estep(void *ctx, const float *naug_dev, const float *ftilde_dev, long S, int C, int D, int R,
                 const float *T32_dev, const float *G_dev, int batch, double *A_aug_dev, double *Cacc_dev,
                 double *e_h_dev, double *iv_sigma_dev, int chain, const double *T64_dev, void *ws_dev,
                 size_t ws_bytes, void *stream);

/* The same E-step with the two products that read f~ -- aux = f~ T (sidekit/factor_analyser.py:88) and
 * C += e_h' f~ (:125) -- on the bf16 matrix cores as well (three-term split operands, see skm_gemm_split): f~ is a
 * constant of the whole training run, so its planes are built ONCE by skm_tv_ftilde_planes (12 bytes per element:
 * one image with k = supervector index for the aux product, one with k = session for the C accumulation;
 * skm_tv_ftilde_planes_bytes(S, C, D) bytes) and handed to every iteration.  ft_planes_dev = NULL, SKM_TV_SPLIT=0 or
 * SKM_TV_PRECISE: exactly skm_tv_estep.  The C accumulation takes the planes only when batch is a multiple of 16. */
size_t skm_tv_ftilde_planes_bytes(long S, int C, int D);
/* extract_ivectors reads f~ once, for the aux product only: skm_tv_whiten_planes is skm_tv_whiten (same values, same
 * naug) writing the rows image of the planes (skm_tv_ftilde_rows_bytes(S, C, D) bytes, the first part of what
 * skm_tv_ftilde_planes builds) instead of the float32 matrix; skm_tv_estep_planes then takes ftilde_dev = NULL with
 * these planes for extraction (A_aug_dev = NULL, not SKM_TV_PRECISE).  sidekit/factor_analyser.py:76-79, 88. */
size_t skm_tv_ftilde_rows_bytes(long S, int C, int D);
int skm_tv_whiten_planes(void *ctx, const void *stat0_dev, const void *stat1_dev, int in_f64, long S, int C, int D,
                         const double *mu_dev, const double *sqrt_invcov_dev, void *rows_planes_dev, float *naug_dev,
                         void *stream);
int skm_tv_ftilde_planes(void *ctx, const float *ftilde_dev, long S, int C, int D, void *planes_dev, void *stream);
int skm_tv_estep_planes(void *ctx, const float *naug_dev, const float *ftilde_dev, const void *ft_planes_dev, long S,
                        int C, int D, int R, const float *T32_dev, const float *G_dev, int batch, double *A_aug_dev,
                        double *Cacc_dev, double *e_h_dev, double *iv_sigma_dev, int chain, const double *T64_dev,
                        void *ws_dev, size_t ws_bytes, void *stream);

/* TV M-step: F_out_dev[C*D*R] = per component (A_c^-1 C_c)'; A_aug rows 0..C-1 are destroyed.
 * Replaces the solve loop of factor_analyser.py:653-658 (the reference's LU solve of the SPD system is
 * done by a blocked Cholesky inverse).  Minimum divergence (:661-665) = skm_gemm with chol(R). */
int skm_tv_mstep(void *ctx, double *A_aug_dev, const double *Cacc_dev, int C, int D, int R, int batch,
                 double *F_out_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ i-vector back-end (next row) ---- */

/* Length normalisation in place of the rows of X_dev[S*R] (float64): x / max(||x||, 1e-8).
 * Replaces StatServer.norm_stat1 (sidekit/statserver.py:677-680) as used by cosine_scoring
 * (sidekit/iv_scoring.py:96-98); the score matrix itself (iv_scoring.py:99) is skm_gemm with transB. */
int skm_row_normalize(void *ctx, double *X_dev, long S, int R, void *stream);

/* Class sums / means of rows: out_dev[k*R + j] = alpha * (mean ? 1/n_k : 1) * sum_{i in [cls_off[k], cls_off[k+1])}
 * X_dev[src[i]*ldx + j]; src_dev (S int64: the sessions sorted by class) may be NULL = identity; an empty class with
 * mean = 1 gives NaN.  Replaces the per-speaker `numpy.mean(self.get_model_stat1(id), axis=0)` / `.sum(axis=0)` loops
 * of sidekit/statserver.py:818-824, 843-846, 868-872, 913-917, 1222-1224, 1242-1244 and `get_mean_stat1` (:669-675). */
int skm_class_sums(void *ctx, const double *X_dev, long ldx, const int64_t *src_dev, const int64_t *cls_off_dev, int K,
                   int R, int mean, double alpha, double *out_dev, void *stream);

/* out_dev[i*R + j] = (X_dev[src[i]*ldx + j] - Mu_dev[cls[i]*R + j]) * scale[i]; src NULL = i, cls NULL = row 0 of Mu,
 * Mu NULL = nothing subtracted, scale NULL = 1.  The centring `get_model_stat1(id) - mean` of the same loops (with
 * scale = 1/sqrt(n_k) the scatter GEMM of the centred rows is the `/ spk_ctr_vec.shape[0]`-weighted sum of
 * statserver.py:871,917) and `stat1 - mean` of :807 / center_stat1 on i-vectors (:690-697). */
int skm_rows_center(void *ctx, const double *X_dev, long ldx, const int64_t *src_dev, const int64_t *cls_dev,
                    const double *Mu_dev, const double *scale_dev, long S, int R, double *out_dev, void *stream);

/* out_dev[i] = sum_j A_dev[i*R + j] * B_dev[i*R + j]: with B = X.M the quadratic forms x'Mx of
 * sidekit/iv_scoring.py:142,192-193,197. */
int skm_row_dot(void *ctx, const double *A_dev, const double *B_dev, long S, int R, double *out_dev, void *stream);

/* Sc_dev[i*N + j] = alpha * Sc_dev[i*N + j] + ca * a_dev[i] + cb * b_dev[j]: the score matrix of
 * mahalanobis_scoring (iv_scoring.py:139-142) and two_covariance_scoring (:195-198) from the cross-term GEMM and the
 * per-row quadratic forms. */
int skm_pair_combine(void *ctx, double *Sc_dev, long M, long N, double alpha, const double *a_dev, double ca,
                     const double *b_dev, double cb, void *stream);

/* ------------------------------------------------------------------ feature post-processing (next row) ---- */

/* FeaturesServer.post_processing (sidekit/features_server.py:202-246) for a batch of S sessions whose raw
 * frames are in HBM; session s owns rows [seg_off[s], seg_off[s+1]) of X (seg_off: S+1 int64, dev).
 *   X_dev        raw frames, float32 (in_f64 = 0) or float64, row stride ldx elements
 *   cols_dev     D0 int32 column indices = the server's `mask` (features_server.py:248-258), or NULL (0..D0-1)
 *   label_dev    one byte per input row (VAD label), or NULL = every frame is speech
 *   rasta        1: RASTA filter (frontend/normfeat.py:45-74); the first two labels of a session become its
 *                third (features_server.py:322-323)
 *   ndelta       0 / 1 / 2: append delta / delta + double delta (frontend/features.py:136-169: zero-padded
 *                'same' convolution with the 7-tap filter, each rounded to float32 like the reference's
 *                PARAM_TYPE array); delta_filter_host: 7 doubles on the host, NULL = features_server.py:129
 *   norm         0 none, 1 cms, 2 cmvn (frontend/normfeat.py:77-126): per-session mean / population standard
 *                deviation over the labelled frames, applied to every frame; gmean_dev / ginvstd_dev
 *                (D0*(ndelta+1) doubles, ginvstd = 1/global_std) replace them when given
 *   keep_all     1: every frame is written (row of frame t of session s = out_off[s] + t);
 *                0: only labelled frames, compacted (features_server.py:241-244)
 *   out_dev      [rows][D0*(ndelta+1)] float32 (out_f64 = 0) or float64; rows <= seg_off[S] - seg_off[0]
 *   label_out_dev one byte per output row, or NULL;  out_off_dev: S+1 int64, out_off[S] = rows written
 *   T            rows of the batch, >= seg_off[S] - seg_off[0] (sizes the chunk tables: sessions are cut into
 *                256-frame chunks, one lane per chunk and column)
 * Arithmetic is float64.  ws_dev needs skm_feat_workspace_bytes(T, S, D0, ndelta) bytes.
 * 'stg' is broken in the reference (float slice indices, normfeat.py:166) and the sliding-window
 * normalisations are not covered by this entry point. */
size_t skm_feat_workspace_bytes(long T, int S, int D0, int ndelta);
int skm_feat_postproc(void *ctx, const void *X_dev, int in_f64, long ldx, const int *cols_dev, int D0,
                      const uint8_t *label_dev, const int64_t *seg_off_dev, int S, long T, int rasta, int ndelta,
                      const double *delta_filter_host, int norm, const double *gmean_dev, const double *ginvstd_dev,
                      int keep_all, void *out_dev, int out_f64, uint8_t *label_out_dev, int64_t *out_off_dev,
                      void *ws_dev, size_t ws_bytes, void *stream);

/* Sliding-window normalisation 'cms_sliding' / 'cmvn_sliding' (sidekit/features_server.py:282-287 ->
 * cep_sliding_norm, sidekit/frontend/normfeat.py:194-229) of a batch of sessions, after skm_feat_postproc with
 * norm = 0 and keep_all = 1:
 *   Y_dev      [rows][D] float64 (in_f64 = 1) or float32, the un-normalised frames (Yn_dev: same type); the window
 *              sums are float64 either way; a float32 input needs a float32 output
 *   compact=0: out_dev is another [rows][D] float64 buffer (out_f64 = 1) that starts as a copy of Y
 *   compact=1: out_dev is [labelled frames][D], float32 or float64: frame selection (features_server.py:241-244)
 *              fused in, row of labelled frame p of session s = coff[s] + p; chunk entries of a short session then
 *              count labelled positions too
 *   idx_dev    row of every labelled frame (ascending), coff_dev[S+1] = labelled frames before each session
 *   a session with more than `win` labelled frames: labelled frame p gets (y - mean) [/ std] over the labelled
 *   frames [c - win/2, c + win/2], c = p clipped to [win/2, n - win/2 - 1] (pandas rolling, centred, sample std;
 *   edges replicated, normfeat.py:215-224); unlabelled frames keep their value
 *   a session with at most `win` labelled frames: all its rows are copied from Yn_dev = the whole-session cms /
 *   cmvn result of skm_feat_postproc (normfeat.py:206-210); Yn_dev may be NULL when no such session exists
 *   chunk_sess_dev / chunk_p0_dev [nchunks] int32: work list, chunk_len positions per entry -- labelled positions of a
 *   long session, frames of a short one.  win odd (301 in the reference), reduce = 1 for 'cmvn_sliding'. */
int skm_feat_sliding_norm(void *ctx, const void *Y_dev, const void *Yn_dev, int in_f64, void *out_dev, int out_f64,
                          int compact, int D,
                          const int64_t *idx_dev, const int64_t *coff_dev, const int64_t *seg_off_dev,
                          const int *chunk_sess_dev, const int *chunk_p0_dev, int nchunks, int chunk_len, int win,
                          int reduce, void *stream);

/* The same normalisation for the sessions with MORE than `win` labelled frames, window held in LDS: a workgroup loads
 * the rows its windows touch once (gathered through idx_dev, 16 columns at a time) instead of three row reads per
 * output.  Work list entries are chunks of skm_feat_sliding_lds_chunk(in_f64) labelled positions of one session
 * (chunk_p0 = first position); win <= 601.  Same arguments otherwise; sessions with at most `win` labelled frames go
 * through skm_feat_sliding_norm. */
int skm_feat_sliding_lds_chunk(int in_f64);
int skm_feat_sliding_norm_lds(void *ctx, const void *Y_dev, int in_f64, void *out_dev, int out_f64, int compact, int D,
                              const int64_t *idx_dev, const int64_t *coff_dev, const int *chunk_sess_dev,
                              const int *chunk_p0_dev, int nchunks, int win, int reduce, void *stream);

/* Index of the labelled frames of a batch (what the sliding normalisation walks; replaces a prefix sum and a
 * compaction done with tensor-library calls): idx_dev[k] = row of the k-th labelled frame among rows 0..T-1 (room for T
 * entries), coff_dev[s] = labelled frames in front of row seg_off_dev[s], s = 0..S (S + 1 entries).  label_dev: one
 * byte per row, non-zero = labelled (the VAD labels of sidekit/features_server.py:241-244).  No atomics. */
size_t skm_label_index_workspace_bytes(long T);
int skm_label_index(void *ctx, const uint8_t *label_dev, long T, const int64_t *seg_off_dev, int S, int64_t *idx_dev,
                    int64_t *coff_dev, void *ws_dev, size_t ws_bytes, void *stream);

/* ---- One process, several GPUs -------------------------------------------------------------------------------------
 * The reference fans its hot loops out from ONE Python process (num_thread / multiprocessing: sidekit/mixture.py:570-588,
 * sidekit/sidekit_wrappers.py:141-214) or over MPI ranks (sidekit/sidekit_mpi.py: Reduce + bcast of the EM accumulators
 * :455-481, of the TV accumulators :186-214, Gatherv of the i-vectors :313-314).  For a caller that stays one plain
 * process: one SkmCtx and one host thread per GPU, and these collectives between the GPUs -- RCCL communicators from
 * ncclCommInitAll (librccl is opened at run time: SKM_ERR_UNSUPPORTED when the machine has none), every call issues the
 * operation for ALL devices inside one group, on streams[i] of device i, asynchronously.  Pointer lists have
 * skm_comm_size() entries in the order of `devices`.  Not thread-safe per communicator: one caller at a time. */
int skm_comm_init_all(int ndev, const int *devices, void **comm_out);
int skm_comm_size(void *comm);
int skm_comm_destroy(void *comm);
/* in place over `count` float64 values; op 0 = sum, 1 = maximum */
int skm_allreduce_f64(void *comm, void *const *bufs_dev, size_t count, int op, void *const *streams);
/* sum of [ndev * recv_count] float64 values, device i keeps slice i in recv_dev[i] */
int skm_reduce_scatter_f64(void *comm, void *const *send_dev, void *const *recv_dev, size_t recv_count,
                           void *const *streams);
/* recv_dev[i] (ndev * send_bytes bytes) = the send buffers of all devices in device order */
int skm_allgather_bytes(void *comm, void *const *send_dev, void *const *recv_dev, size_t send_bytes,
                        void *const *streams);

#ifdef __cplusplus
}
#endif
#endif /* SKM_H */
