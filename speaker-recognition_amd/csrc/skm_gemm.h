// skm_gemm.h -- generic strided-batched MFMA GEMM for gfx950 used by the TV path.
//
//   C[z] = alpha * op(A[z]) * op(B[z]) + beta * C[z]        (row-major, leading dimensions in elements)
//
// TI = float  -> v_mfma_f32_32x32x2_f32, block tile 128 x 128, BK 32 (16 with ACC64)
// TI = double -> v_mfma_f64_16x16x4_f64, block tile  64 x  64, BK 16
// TO is the storage type of C (float or double).  With ACC64 (TI = float only) the f32 MFMA
// accumulators are folded into float64 registers every 256 k-steps; the TV hot path no longer uses it
// (long reductions are cut into K = 2048 products accumulated in a float64 C with beta = 1).
// Kernel (skm_gemm.hip, device helpers in skm_gemm_dev.h): 8 waves per workgroup, waves 0-3 read LDS and issue
// MFMAs only, waves 4-7 stage tiles (buffer loads with hardware range checks -> registers -> ds_write_b128);
// per-operand LDS layout chosen by the operand's contiguous dimension (AKF / BKF template flags).
//
// Extras used by the blocked Cholesky / TV E-step:
//   tile_mask  0: all of C, 1: only the 32 x 32 blocks of C on or above the block diagonal (column block >=
//              row block) are written; everything else of C is left untouched
//   colmap     flattened view of "the part of every row of an R x R matrix at/right of its diagonal block":
//              one dense GEMM over sum_z (R - col0(z)) logical columns instead of R small ones (tri_n mode)
//   tri_n > 0  "triangular rows" batch mode: batch z is row z of a [tri_n x tri_n] matrix stored behind
//              B and C (row stride tri_ldz); only columns >= col0(z) = (z / tri_blk) * tri_blk are produced:
//              B += z*tri_ldz + col0, C += z*tri_ldz + col0, N = tri_n - col0.  (A is shared.)
#pragma once
#include "skm_common.h"

struct GemmArgs {
    const void *A;
    const void *B;
    void *C;
    int M, N, K;
    long lda, ldb, ldc;
    long sA, sB, sC;
    int transA, transB;  // transX: the matrix is stored transposed (A as K x M, B as N x K)
    double alpha, beta;
    int tile_mask;
    int mask_blk;  // block size of tile_mask (32)
    int tri_n, tri_blk;
    long tri_ldz;
    int vec;  // set by skm_gemm_launch: bit 0 / bit 1 = 16-byte vector loads legal for A / B
    // optional column map (device, N ints): logical column n of B and C lives at element offset colmap[n]
    // of a stored row (B not transposed).  colmap_vec: 4 logical columns starting at a multiple of 4 are
    // contiguous and 16-byte aligned.
    const int *colmap;
    int colmap_vec;
    // k-range trimming for triangular operands whose zero part is physically stored as zeros (the result is
    // the same with ktrim = 0, only slower):
    //   1: B[k][n] = 0 for k > n (B upper triangular, not transposed)    -> k < n0 + BN
    //   2: op(A)[m][k] = 0 for k < m and op(B)[k][n] = 0 for k < n        -> k >= max(m0, n0)   (X X' with X upper)
    //   4: op(B)[k][n] = 0 for k < n                                        -> k >= n0
    int ktrim;
    // XCD-aware tile order: 0 = plain 2-D grid; 1 = the M tiles of one N tile column run back to back on one XCD
    // (they share the B tile in that XCD's L2); 2 = the N tiles of one M tile row do (they share the A tile).
    int swz;
    int swz_tiles_m, swz_tiles_n, swz_batch;  // set by skm_gemm_launch (swz = 3: batched launch, one matrix per XCD at a time)
    // optional symmetric rank-one term of the epilogue (square C): C[z][m][n] += r1[z * r1_stride + m] * r1[z * r1_stride + n]
    // after alpha / beta (the e_h e_h' of e_hh = L^-1 + e_h e_h', sidekit/factor_analyser.py:90, fused into X X')
    const double *r1;
    long r1_stride;
};

static inline GemmArgs skm_gemm_args(const void *A, const void *B, void *C, int M, int N, int K, long lda,
                                     long ldb, long ldc)
{
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A;
    g.B = B;
    g.C = C;
    g.M = M;
    g.N = N;
    g.K = K;
    g.lda = lda;
    g.ldb = ldb;
    g.ldc = ldc;
    g.alpha = 1.0;
    g.beta = 0.0;
    g.mask_blk = 32;
    return g;
}

// GEMM kinds: SKM_GEMM_* in include/skm.h

// kind: input/compute type and output type
//   SKM_GEMM_F32_F32     A,B float,  C float
//   SKM_GEMM_F32_F64ACC  A,B float,  C double (f64 folding of the accumulators)
//   SKM_GEMM_F32_F64OUT  A,B float,  C double (plain f32 accumulation, for short K)
//   SKM_GEMM_F64_F64     A,B double, C double
//   SKM_GEMM_F64_F32OUT  A,B double, C float
//   SKM_GEMM_F32S_F32    A,B float,  C float, 64 x 64 tiles of 16 x 16 x 4 MFMAs (batches of small matrices)
//   SKM_GEMM_F32S_F64OUT the same kernel, C double (rank-sized products whose rank is far from a multiple of 128)
int skm_gemm_launch(SkmCtx *ctx, int kind, const GemmArgs &g, int batch, hipStream_t st, const char *tag);


// ---- float32 GEMM on the bf16 matrix cores by operand splitting (skm_gemm_b3.hip) ----
// Bytes of the three bf16 planes [3][Kp/8][rows_p][8] of a rows x K operand (K padded to 16, rows to 128).
size_t skm_b3_plane_bytes(long rows, long K);
// Split a float32 operand into its planes.  kfast: element (row, k) = src[row * ld + k]; otherwise element (row, k) =
// src[k * ld + (rowmap ? rowmap[row] : row)].
int skm_b3_split(SkmCtx *ctx, const float *src, long rows, long K, long ld, int kfast, const int *rowmap, void *dst,
                 hipStream_t st, const char *tag);
// C[m * ldc + (colmap ? colmap[n] : n)] (+)= sum_k A(m, k) B(n, k) from split planes; C float or double.
// plane_a / plane_b: distance between the three planes of A / B in 16-byte units when the product runs over a k range
// of larger planes (Ap / Bp then point at the first k-group of the range in plane 0); 0 = planes of exactly K.
// nz > 1: nz independent products in one launch (split-K slabs): product z reads Ap + 16 zA z, Bp + 16 zB z bytes and
// writes C + zC z elements.
int skm_b3_gemm(SkmCtx *ctx, const void *Ap, long M, const void *Bp, long N, long K, void *C, int c_f64, long ldc,
                const int *colmap, int accumulate, hipStream_t st, const char *tag, long plane_a = 0, long plane_b = 0,
                int nz = 1, long zA = 0, long zB = 0, long zC = 0);
// padded row count / k-groups of a planes image (layout arithmetic for callers that address k ranges)
long skm_b3_rows_padded(long rows);
long skm_b3_kgroups(long K);
