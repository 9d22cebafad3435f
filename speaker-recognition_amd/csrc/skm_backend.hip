// skm_backend.hip -- i-vector back-end (next row): the row / class reductions around the float64
// GEMMs of the covariance, LDA, WCCN and Mahalanobis / two-covariance scoring code of sidekit/statserver.py:800-925
// and sidekit/iv_scoring.py:109-205.  All float64, HBM-bound (every matrix is read once), fixed summation order:
// bit-identical from run to run.  gfx950 only.
#include "skm_common.h"

// out[k][j] = alpha * (mean ? 1/n_k : 1) * sum_{i in [off[k], off[k+1])} X[src[i]][j]        (src NULL: identity)
// One thread per column (adjacent lanes read adjacent columns of a row), rows of a class walked in order.
__global__ __launch_bounds__(256) void k_class_sums(const double *__restrict__ X, long ldx, const int64_t *__restrict__ src,
                                                    const int64_t *__restrict__ off, int R, int mean, double alpha,
                                                    double *__restrict__ out)
{
    const int k = blockIdx.x;
    const int j = blockIdx.y * 256 + threadIdx.x;
    if (j >= R) return;
    const long a = off[k], b = off[k + 1];
    double s0 = 0.0, s1 = 0.0;
    long i = a;
    for (; i + 1 < b; i += 2) {     // two independent loads in flight; the order of the additions is fixed
        const long r0 = src ? src[i] : i, r1 = src ? src[i + 1] : i + 1;
        s0 += X[r0 * ldx + j];
        s1 += X[r1 * ldx + j];
    }
    if (i < b) s0 += X[(src ? src[i] : i) * ldx + j];
    double s = s0 + s1;
    if (mean) s = s / (double)(b - a);      // an empty class gives NaN like numpy's mean of nothing
    out[(long)k * R + j] = alpha * s;
}

// out[i][j] = (X[src[i]][j] - Mu[cls[i]][j]) * scale[i]      (src NULL: i; cls NULL: row 0 of Mu; Mu NULL: 0; scale NULL: 1)
__global__ __launch_bounds__(256) void k_rows_center(const double *__restrict__ X, long ldx, const int64_t *__restrict__ src,
                                                     const int64_t *__restrict__ cls, const double *__restrict__ Mu,
                                                     const double *__restrict__ scale, long S, int R,
                                                     double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S) return;
    const double *x = X + (src ? src[row] : row) * ldx;
    const double *m = Mu ? Mu + (cls ? cls[row] : 0) * (long)R : nullptr;
    const double sc = scale ? scale[row] : 1.0;
    double *o = out + row * (long)R;
    for (int j = lane; j < R; j += 64) o[j] = (x[j] - (m ? m[j] : 0.0)) * sc;
}

// out[i] = sum_j A[i][j] * B[i][j]    (one wave per row; the quadratic forms x'Mx with B = X.M)
__global__ __launch_bounds__(256) void k_row_dot(const double *__restrict__ A, const double *__restrict__ B, long S, int R,
                                                 double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S) return;
    const double *a = A + row * (long)R, *b = B + row * (long)R;
    double s = 0.0;
    for (int j = lane; j < R; j += 64) s += a[j] * b[j];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) out[row] = s;
}

// Sc[i][j] = alpha * Sc[i][j] + ca * a[i] + cb * b[j]
__global__ __launch_bounds__(256) void k_pair_combine(double *__restrict__ Sc, long M, long N, double alpha,
                                                      const double *__restrict__ a, double ca,
                                                      const double *__restrict__ b, double cb)
{
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const double bj = cb * b[j];
    for (long i = blockIdx.y; i < M; i += gridDim.y) Sc[i * N + j] = alpha * Sc[i * N + j] + ca * a[i] + bj;
}

extern "C" int skm_class_sums(void *ctx_, const double *X, long ldx, const int64_t *src, const int64_t *cls_off, int K,
                              int R, int mean, double alpha, double *out, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && K >= 0 && R > 0 && ldx >= R && (K == 0 || (X && cls_off && out)), "skm_class_sums: bad argument");
    if (K == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_class_sums", (hipStream_t)stream);
    hipLaunchKernelGGL(k_class_sums, dim3((unsigned)K, (unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                       ldx, src, cls_off, R, mean, alpha, out);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" int skm_rows_center(void *ctx_, const double *X, long ldx, const int64_t *src, const int64_t *cls,
                               const double *Mu, const double *scale, long S, int R, double *out, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && S >= 0 && R > 0 && ldx >= R && (S == 0 || (X && out)) && (!cls || Mu),
                "skm_rows_center: bad argument");
    if (S == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_rows_center", (hipStream_t)stream);
    hipLaunchKernelGGL(k_rows_center, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X, ldx, src, cls,
                       Mu, scale, S, R, out);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" int skm_row_dot(void *ctx_, const double *A, const double *B, long S, int R, double *out, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && S >= 0 && R > 0 && (S == 0 || (A && B && out)), "skm_row_dot: bad argument");
    if (S == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_row_dot", (hipStream_t)stream);
    hipLaunchKernelGGL(k_row_dot, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, A, B, S, R, out);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" int skm_pair_combine(void *ctx_, double *Sc, long M, long N, double alpha, const double *a, double ca,
                                const double *b, double cb, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && M >= 0 && N >= 0 && (M == 0 || N == 0 || (Sc && a && b)), "skm_pair_combine: bad argument");
    if (M == 0 || N == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_pair_combine", (hipStream_t)stream);
    const unsigned gy = (unsigned)(M < 1024 ? M : 1024);
    hipLaunchKernelGGL(k_pair_combine, dim3((unsigned)((N + 255) / 256), gy), dim3(256), 0, (hipStream_t)stream, Sc, M, N,
                       alpha, a, ca, b, cb);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}
