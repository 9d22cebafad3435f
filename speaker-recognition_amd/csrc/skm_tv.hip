// skm_tv.hip -- total-variability (i-vector) E-step / M-step for gfx950 (MI355X).
//
// Replaces (reference = SIDEKIT 1.3.2 fork):
//   sidekit/factor_analyser.py:52-92    e_on_batch      (whitening, per-session inv(I + T' diag(N) T), e_h, e_hh)
//   sidekit/factor_analyser.py:109-128  e_gather        (_A += N' e_hh, _C += e_h' f~, _R += sum e_hh)
//   sidekit/factor_analyser.py:131-164  iv_extract_on_batch / iv_collect
//   sidekit/factor_analyser.py:653-658  M-step, per component solve(A_c, C_c).T
//
// Formulation.  T is the CD x R loading matrix in the whitened space.
//   prepare : G_c = T_c' T_c for every component (R x R, f64 MFMA -> f32), T32 = float(T)
//   E-step  : aux = f~ T                       one GEMM over all sessions   (f32 MFMA, f64 folding)
//             per batch of sessions:
//               L_s = I + sum_c N_sc G_c       GEMM  [B x C] . [C x R*R]     (f32 MFMA, f64 folding), upper rows only
//               L_s^-1                         blocked upper Cholesky + triangular inverse + X X' (f64 MFMA GEMMs
//                                              over the batch, 32x32 diagonal blocks factored in LDS)
//               e_h = L^-1 aux, e_hh = L^-1 + e_h e_h'   (per-session kernel)
//               A[c] += sum_s N_sc e_hh(s)     GEMM  [C+1 x B] . [B x R*R]; the extra row of ones yields _R
//             C += e_h' f~                     one GEMM over all sessions
//   M-step  : A_c^-1 through the same blocked inverse (batch = components), T_c = C_c' A_c^-1
// Symmetric R x R matrices are kept in full [R][R] row-major storage; only the part of each row at or right
// of its 32-aligned diagonal block is produced/consumed ("triangular rows" mode of the GEMM), the rest is
// never read except where the buffer was zero-filled.
#include "skm_gemm.h"

#define TV_BLK 4

// row stride of naug = [N | 1 | 0-pad]: C+1 rounded up to 4 floats so that rows stay 16-byte aligned
static inline int tv_ldn(int C) { return (C + 1 + 3) / 4 * 4; }

// ------------------------------------------------------------------------------------------------
// whitening (factor_analyser.py:76-79): in place in the input dtype in the reference; with float32
// statistics every update is computed in float64 and rounded to float32, which is reproduced here.
// ------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void k_tv_whiten(const TS *__restrict__ stat0, const TS *__restrict__ stat1,
                                                   long S, int C, int D, const double *__restrict__ mu,
                                                   const double *__restrict__ sqrt_ic, float *__restrict__ ft,
                                                   float *__restrict__ naug, int ldn)
{
    const long CD = (long)C * D;
    const long total = S * CD;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long s = i / CD;
        const int idx = (int)(i - s * CD);
        const int c = idx / D;
        const double n = (double)stat0[s * C + c];
        if (sizeof(TS) == 4) {
            const float t1 = (float)((double)stat1[i] - n * mu[idx]);
            ft[i] = (float)((double)t1 * sqrt_ic[idx]);
        } else {
            ft[i] = (float)(((double)stat1[i] - n * mu[idx]) * sqrt_ic[idx]);
        }
    }
    const long totn = S * ldn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < totn; i += (long)gridDim.x * blockDim.x) {
        const long s = i / ldn;
        const int c = (int)(i - s * ldn);
        naug[i] = (c < C) ? (float)stat0[s * C + c] : (c == C ? 1.0f : 0.0f);
    }
}

// The same whitening written straight into the rows image of the bf16 x 3 planes of f~ (skm_gemm_b3.hip:
// planes[3][KG][Sp][8 bf16], k = supervector index): what extract_ivectors needs of f~ is the aux product only, which
// the split GEMM takes from these planes; the float32 f~ is never stored.  Values are exactly those of k_tv_whiten
// (same float64 arithmetic, same float32 roundings) and the split of a float32 value is exact.  Thread walk as in
// k_b3_split for a k-fast source: four threads per 128 bytes of a session's statistics, rows four threads apart.
typedef int tv_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned tv_bf16_rn(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
template <typename TS>
__global__ __launch_bounds__(256) void k_tv_whiten_planes(const TS *__restrict__ stat0, const TS *__restrict__ stat1,
                                                          long S, int C, int D, const double *__restrict__ mu,
                                                          const double *__restrict__ sqrt_ic, tv_i32x4 *__restrict__ dst,
                                                          long Sp, long KG, float *__restrict__ naug, int ldn)
{
    const long CD = (long)C * D, total = Sp * KG, KQ = (KG + 3) / 4, walk = Sp * KQ * 4;
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < walk; w += (long)gridDim.x * blockDim.x) {
        const long q = w / (Sp * 4), rem = w - q * (Sp * 4);
        const long s = rem >> 2, kg = q * 4 + (rem & 3);
        if (kg >= KG) continue;
        unsigned t1[8], t2[8], t3[8];
        // one division per thread: the component of the first of the 8 columns, then a running remainder
        const long k0 = kg * 8;
        int c = (int)(k0 / D), r = (int)(k0 - (long)c * D);
        const bool whole = s < S && k0 + 8 <= CD;
        TS raw[8];
        double m8[8], q8[8];
        if (whole) {
            const TS *p1 = stat1 + s * CD + k0;
            if (((uintptr_t)p1 & 15) == 0) {
                typedef TS vec_t __attribute__((ext_vector_type(16 / sizeof(TS))));
                constexpr int VW = 16 / sizeof(TS);
#pragma unroll
                for (int v = 0; v < 8 / VW; ++v) {
                    const vec_t t = *reinterpret_cast<const vec_t *>(p1 + v * VW);
#pragma unroll
                    for (int e = 0; e < VW; ++e) raw[v * VW + e] = t[e];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) raw[j] = p1[j];
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {               // k0 is a multiple of 8: 64-byte aligned doubles
                const double2 a = *reinterpret_cast<const double2 *>(mu + k0 + 2 * v);
                const double2 b = *reinterpret_cast<const double2 *>(sqrt_ic + k0 + 2 * v);
                m8[2 * v] = a.x; m8[2 * v + 1] = a.y;
                q8[2 * v] = b.x; q8[2 * v + 1] = b.y;
            }
        }
        double n = (s < S && k0 < CD) ? (double)stat0[s * C + c] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long idx = k0 + j;
            if (r == D) {
                r = 0;
                ++c;
                n = (s < S && idx < CD) ? (double)stat0[s * C + c] : 0.0;
            }
            ++r;
            float x = 0.f;
            if (s < S && idx < CD) {
                const double v1 = whole ? (double)raw[j] : (double)stat1[s * CD + idx];
                const double mj = whole ? m8[j] : mu[idx], qj = whole ? q8[j] : sqrt_ic[idx];
                if (sizeof(TS) == 4) {
                    const float a = (float)(v1 - n * mj);
                    x = (float)((double)a * qj);
                } else {
                    x = (float)((v1 - n * mj) * qj);
                }
            }
            t1[j] = tv_bf16_rn(x);
            const float r1 = x - __uint_as_float(t1[j] << 16);
            t2[j] = tv_bf16_rn(r1);
            const float r2 = r1 - __uint_as_float(t2[j] << 16);
            t3[j] = tv_bf16_rn(r2);
        }
        tv_i32x4 v1, v2, v3;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            v1[h] = (int)(t1[2 * h] | (t1[2 * h + 1] << 16));
            v2[h] = (int)(t2[2 * h] | (t2[2 * h + 1] << 16));
            v3[h] = (int)(t3[2 * h] | (t3[2 * h + 1] << 16));
        }
        const long o = kg * Sp + s;
        dst[o] = v1;
        dst[total + o] = v2;
        dst[2 * total + o] = v3;
    }
    const long totn = S * ldn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < totn; i += (long)gridDim.x * blockDim.x) {
        const long s = i / ldn;
        const int c = (int)(i - s * ldn);
        naug[i] = (c < C) ? (float)stat0[s * C + c] : (c == C ? 1.0f : 0.0f);
    }
}

// Column map of the flattened "upper part" view of an R x R matrix: row z keeps the columns from the start of its
// 4-wide block (TV_BLK: 16-byte vectors stay aligned; 80 800 of 160 000 entries at R = 400, the exact triangle has
// 80 200): for row z (block b = z / TV_BLK) the logical columns start(z) .. start(z) + R - TV_BLK b - 1 map to element
// offsets z*R + TV_BLK b + j.  Nothing downstream reads below the diagonal: the diagonal-block kernel loads r <= c
// only, the panel products read rows above their block, e_hh / A are consumed as upper triangles.
#define TV_CS_SPLIT 16   // session ranges of the e_hh column sums
#define TV_AUX_CHUNK 2048 // k per float32 accumulation run of the aux product (float64 across runs)
// The aux product has N = R columns; R = 400 is 3 x 128 + 16 and the 128-wide kernel spends a whole (guarded) tile
// column on the 16: 1.07 ms per 50 000 x 400 x 2048 chunk against 0.73 ms for the 64 x 64-tile kernel (7 columns of 64,
// tools/time_aux_shapes.py).  The C accumulation (M = R) is the other way round (3.7 against 3.9 ms) and stays.
#define TV_AUX_KIND SKM_GEMM_F32S_F64OUT
static inline long tv_tri_cols(int R)
{
    long n = 0;
    for (int z = 0; z < R; ++z) n += R - (z / TV_BLK) * TV_BLK;
    return n;
}

__global__ __launch_bounds__(256) void k_tri_map(int *__restrict__ map, int R)
{
    const int z = blockIdx.x;
    const int b = z / TV_BLK, col0 = b * TV_BLK;
    // start(z) = sum_{b' < b} BLK (R - BLK b') + (z - BLK b)(R - BLK b)
    long start = (long)TV_BLK * ((long)b * R - (long)TV_BLK * b * (b - 1) / 2) + (long)(z - col0) * (R - col0);
    for (int j = threadIdx.x; j < R - col0; j += blockDim.x) map[start + j] = z * R + col0 + j;
}

// Sum nz slabs of n doubles (fixed order).
__global__ __launch_bounds__(256) void k_sum_slabs(double *__restrict__ dst, const double *__restrict__ src, long n, int nz)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int z = 0; z < nz; ++z) s += src[(long)z * n + i];
        dst[i] = s;
    }
}

__global__ __launch_bounds__(256) void k_cvt_f64_f32(const double *__restrict__ src, float *__restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

__global__ __launch_bounds__(256) void k_cvt_f32_f64(const float *__restrict__ src, double *__restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = (double)src[i];
}

// ------------------------------------------------------------------------------------------------
// diagonal block of the blocked Cholesky: U = chol_upper(M_kk) (n <= 64), then W = U^-1 written back in
// place (upper triangular, strictly lower part zeroed).
//
// One WAVE per matrix, no LDS and no barrier: lane c keeps column c of the block in 64 float64 registers
// (col[r] = a[r][c]), every loop is fully unrolled so the register indices are constants, and the one value a
// step needs from another lane (u[j][r] from lane r, the pivot from lane j) is a v_readlane broadcast that
// lands in scalar registers and feeds the FMA directly.  Right-looking factorisation, then the inverse by
// rows bottom-up IN PLACE: at row i the registers col[l], l > i, already hold W[l][c], col[i] still holds
// U[i][c] until every lane has read it.  Entries below the diagonal of a lane's column are scratch during the
// factorisation (never read) and are zeroed before the inverse.  The workgroup version this replaces (block in
// LDS, two barriers per column) took 4.3 ms per training batch set against 1.x ms for this one.
// ------------------------------------------------------------------------------------------------
#define CH_BLK 64
__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
__device__ __forceinline__ double tv_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float tv_sqrt(float v) { return __fsqrt_rn(v); }

// T = double: the M-step (one matrix per component) and, with SKM_TV_CHAIN=f64, the per-session chain;
// T = float: the per-session chain (default, see tv_chain_f64()).
template <typename T>
__global__ __launch_bounds__(256) void k_chol_diag(T *__restrict__ Mb, long stride, int ld, int n, int add_identity,
                                                   int nb)
{
    const int lane = threadIdx.x & 63;
    const long mi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (mi >= nb) return;                       // wave-uniform
    T *M = Mb + mi * stride;
    T col[CH_BLK];
#pragma unroll
    for (int r = 0; r < CH_BLK; ++r) {
        T v = 0;
        if (r < n && lane < n && r <= lane) v = M[(long)r * ld + lane];
        if (add_identity && r == lane && r < n) v += (T)1;
        col[r] = v;
    }
    T dinv_own = 0;                             // 1 / U[c][c] of this lane's column
#pragma unroll
    for (int j = 0; j < CH_BLK; ++j) {
        if (j < n) {                            // uniform
            const T d = tv_sqrt(lane_bcast(col[j], j));
            const T u = col[j] / d;
            dinv_own = (lane == j) ? (T)1 / d : dinv_own;
            col[j] = (lane > j) ? u : col[j];   // row j scaled right of the pivot; the pivot itself stays
#pragma unroll
            for (int r = j + 1; r < CH_BLK; ++r) {
                // a[r][c] -= u[j][r] * u[j][c]; lanes c < r write scratch that nobody reads
                col[r] -= lane_bcast(col[j], r) * col[j];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CH_BLK; ++r) col[r] = (r > lane) ? (T)0 : col[r];
    // W = U^-1 by rows, bottom-up: W[i][c] = -(sum_{l=i+1..c} U[i][l] W[l][c]) / U[i][i], W[c][c] = 1 / U[c][c]
#pragma unroll
    for (int i = CH_BLK - 1; i >= 0; --i) {
        if (i < n) {                            // uniform
            T acc = 0;
#pragma unroll
            for (int l = i + 1; l < CH_BLK; ++l) acc += lane_bcast(col[i], l) * col[l];   // W[l][c] = 0 for l > c
            const T w = -acc * lane_bcast(dinv_own, i);
            col[i] = (lane > i) ? w : ((lane == i) ? dinv_own : (T)0);
        }
    }
    if (lane < n) {
#pragma unroll
        for (int r = 0; r < CH_BLK; ++r)
            if (r < n) M[(long)r * ld + lane] = col[r];
    }
}

// ------------------------------------------------------------------------------------------------
// per session: e_h = L^-1 aux and diag(L^-1) from X = U^-1; e_hh = L^-1 + e_h e_h' is the X X' product with a
// rank-one epilogue (tv_batch_chain)
// ------------------------------------------------------------------------------------------------
// e_h = X (X' aux) and diag(L^-1) = row norms^2 of X, with X = U^-1 (upper triangular) left by the blocked inverse
// (extraction with uncertainty stops here: the R^3 product X X' is only formed when the A accumulation needs e_hh).
template <typename T>
__global__ __launch_bounds__(256) void k_tv_finish_tri(const T *__restrict__ X, int R, const double *__restrict__ aux,
                                                       double *__restrict__ e_h, float *__restrict__ e_h32,
                                                       double *__restrict__ iv_sigma)
{
    extern __shared__ double sh[];  // aux[R], y[R]
    double *ax = sh, *y = sh + R;
    const long b = blockIdx.x;
    const T *Xi = X + b * (long)R * R;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < R; i += 256) ax[i] = aux[b * R + i];
    __syncthreads();
    // y = X' aux: y[j] = sum_{i<=j} X[i][j] aux[i]   (column sweep, coalesced along j)
    for (int j = t; j < R; j += 256) {
        double s = 0.0;
#pragma unroll 8
        for (int i = 0; i <= j; ++i) s += (double)Xi[(long)i * R + j] * ax[i];      // 8 loads in flight
        y[j] = s;
    }
    __syncthreads();
    // e_h[i] = sum_{j>=i} X[i][j] y[j], diag[i] = sum_{j>=i} X[i][j]^2   (row reduce, one wave per row)
    for (int i = wave; i < R; i += 4) {
        double s = 0.0, d = 0.0;
#pragma unroll 7
        for (int j = i + lane; j < R; j += 64) {
            const double x = (double)Xi[(long)i * R + j];
            s += x * y[j];
            d += x * x;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            s += __shfl_xor(s, m);
            d += __shfl_xor(d, m);
        }
        if (lane == 0) {
            e_h[b * R + i] = s;
            if (e_h32) e_h32[b * R + i] = (float)s;
            if (iv_sigma) iv_sigma[b * R + i] = d + s * s;
        }
    }
}

// Extraction without uncertainty: e_h = L^-1 aux from the blocked Cholesky factor alone (two triangular solves,
// R^2 work per session) -- neither U^-1 nor L^-1 is formed.  After the factorisation the block row k of the buffer
// holds W_k = U_kk^-1 on the diagonal block (upper, strictly lower part zero) and U[k, j > k] to its right.
//   forward   U' y = aux :  y_k = W_k' (aux_k - sum_{i < k} U[i, k]' y_i)
//   backward  U  x = y   :  x_k = W_k  (y_k   - sum_{j > k} U[k, j]  x_j)
// One workgroup (4 waves) per session, y / x in LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_tv_solve(const T *__restrict__ Ub, int R, const double *__restrict__ aux,
                                                  double *__restrict__ e_h, float *__restrict__ e_h32)
{
    extern __shared__ double sh[];          // y[R] (overwritten by x), t[CH_BLK], part[4][CH_BLK]
    double *y = sh, *tv = sh + R, *part = sh + R + CH_BLK;
    const long b = blockIdx.x;
    const T *U = Ub + b * (long)R * R;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int Kb = (R + CH_BLK - 1) / CH_BLK;
    for (int i = t; i < R; i += 256) y[i] = aux[b * R + i];
    __syncthreads();
    // ---- forward
    for (int k = 0; k < Kb; ++k) {
        const int r0 = k * CH_BLK, bs = (R - r0 < CH_BLK) ? (R - r0) : CH_BLK;
        // column sums over the rows above the block: lane = column, wave = row phase (coalesced along the row)
        double s = 0.0;
        if (lane < bs) {
#pragma unroll 8
            for (int i = wave; i < r0; i += 4) s += (double)U[(long)i * R + r0 + lane] * y[i];      // 8 loads in flight
        }
        part[wave * CH_BLK + lane] = s;
        __syncthreads();
        if (t < bs) tv[t] = y[r0 + t] - ((part[t] + part[CH_BLK + t]) + (part[2 * CH_BLK + t] + part[3 * CH_BLK + t]));
        __syncthreads();
        // y_k = W_k' t : y[c] = sum_{i <= c} W[i][c] t[i]
        s = 0.0;
        if (lane < bs)
            for (int i = wave; i <= lane; i += 4) s += (double)U[(long)(r0 + i) * R + r0 + lane] * tv[i];
        part[wave * CH_BLK + lane] = s;
        __syncthreads();
        if (t < bs) y[r0 + t] = (part[t] + part[CH_BLK + t]) + (part[2 * CH_BLK + t] + part[3 * CH_BLK + t]);
        __syncthreads();
    }
    // ---- backward: rows of the block are dealt to the waves, lanes stride along the row (coalesced), shuffle reduce
    for (int k = Kb - 1; k >= 0; --k) {
        const int r0 = k * CH_BLK, bs = (R - r0 < CH_BLK) ? (R - r0) : CH_BLK, c0 = r0 + bs;
        for (int r = wave; r < bs; r += 4) {
            double s = 0.0;
#pragma unroll 6
            for (int j = c0 + lane; j < R; j += 64) s += (double)U[(long)(r0 + r) * R + j] * y[j];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
            if (lane == 0) tv[r] = y[r0 + r] - s;
        }
        __syncthreads();
        // x_k = W_k s : x[r] = sum_{c >= r} W[r][c] s[c]
        for (int r = wave; r < bs; r += 4) {
            double s = (lane >= r && lane < bs) ? (double)U[(long)(r0 + r) * R + r0 + lane] * tv[lane] : 0.0;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
            if (lane == 0) y[r0 + r] = s;
        }
        __syncthreads();
    }
    for (int i = t; i < R; i += 256) {
        const double v = y[i];
        e_h[b * R + i] = v;
        if (e_h32) e_h32[b * R + i] = (float)v;
    }
}

// _R accumulator: column sums of E[nb][R*R] (float32, block-upper entries only) over the sessions of a batch.
// Two steps without atomics: TV_CS_SPLIT session ranges -> part[split][NT] (float64), then row C of A_aug += their sum.
__global__ __launch_bounds__(256) void k_ehh_colsum(const float *__restrict__ E, long RR, int nb, const int *__restrict__ cmap,
                                                    int NT, double *__restrict__ part)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= NT) return;
    const int per = (nb + TV_CS_SPLIT - 1) / TV_CS_SPLIT;
    const int s0 = blockIdx.y * per, s1 = (s0 + per < nb) ? s0 + per : nb;
    const float *p = E + cmap[n];
    double acc = 0.0;
    for (int s = s0; s < s1; ++s) acc += (double)p[(long)s * RR];
    part[(long)blockIdx.y * NT + n] = acc;
}

__global__ __launch_bounds__(256) void k_ehh_colsum_add(const double *__restrict__ part, const int *__restrict__ cmap, int NT,
                                                        double *__restrict__ row)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= NT) return;
    double acc = 0.0;
#pragma unroll
    for (int z = 0; z < TV_CS_SPLIT; ++z) acc += part[(long)z * NT + n];
    row[cmap[n]] += acc;
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
// In-place inverse of `nb` SPD matrices (upper part valid, everything else of the buffer ZERO):
//   Mb  [nb][R][R] f64, destroyed (ends as X = U^-1)
//   out [nb][R][R] f64 = M^-1 (upper 64-tiles only when upper_only)
//   Tb  [nb][CH_BLK][R] f64 scratch
template <typename T>
static int blocked_inverse(SkmCtx *ctx, T *Mb, T *out, T *Tb, int nb, int R, int add_identity,
                           int upper_only, hipStream_t st, bool factor_only = false)
{
    constexpr int KIND = sizeof(T) == 8 ? SKM_GEMM_F64_F64 : SKM_GEMM_F32S_F32;
    const long RR = (long)R * R;
    const int Kb = (R + CH_BLK - 1) / CH_BLK;
    int rc;
    // left-looking by block rows: row k first receives the contributions of all the rows above it in ONE product
    // with K = 64 k (each block row is read-modify-written once, not once per earlier panel), then its diagonal
    // block is factored and the row is scaled
    for (int k = 0; k < Kb; ++k) {
        const int r0 = k * CH_BLK, bs = (R - r0 < CH_BLK) ? (R - r0) : CH_BLK, rem = R - r0 - bs;
        if (k > 0) {
            // M[k, j >= k] -= U[0:r0, k]' U[0:r0, j >= k]   (rows above sit right of their own diagonal blocks)
            GemmArgs u = skm_gemm_args(Mb + r0, Mb + r0, Mb + (long)r0 * R + r0, bs, R - r0, r0, R, R, R);
            u.transA = 1;
            u.alpha = -1.0;
            u.beta = 1.0;
            u.tile_mask = 1;
            u.sA = u.sB = u.sC = RR;
            if ((rc = skm_gemm_launch(ctx, KIND, u, nb, st, "k_gemm_chol_update"))) return rc;
        }
        {
            SkmProfScope ps(ctx, "k_chol_diag", st);
            hipLaunchKernelGGL(k_chol_diag<T>, dim3((nb + 3) / 4), dim3(256), 0, st, Mb + (long)r0 * R + r0, RR, R, bs,
                               add_identity, nb);
        }
        SKM_HIP(hipGetLastError());
        if (rem <= 0) continue;
        T *panel = Mb + (long)r0 * R + r0 + bs;
        // U[k, j>k] = W_k' M[k, j>k]   (in place: every tile reads its whole K range before it writes)
        GemmArgs g = skm_gemm_args(Mb + (long)r0 * R + r0, panel, panel, bs, rem, bs, R, R, R);
        g.transA = 1;
        g.sA = g.sB = g.sC = RR;
        if ((rc = skm_gemm_launch(ctx, KIND, g, nb, st, "k_gemm_chol_panel"))) return rc;
    }
    if (factor_only) return SKM_OK;   // block rows hold [W_k | U[k, j > k]]: enough for triangular solves
    // X = U^-1 in place, block rows bottom-up (diagonal blocks already hold W_i = U_ii^-1)
    for (int i = Kb - 2; i >= 0; --i) {
        const int r0 = i * CH_BLK, bs = CH_BLK, c0 = r0 + bs, rem = R - c0;
        GemmArgs t = skm_gemm_args(Mb + (long)r0 * R + c0, Mb + (long)c0 * R + c0, Tb, bs, rem, rem, R, R, R);
        t.sA = t.sB = RR;
        t.sC = (long)CH_BLK * R;
        t.ktrim = 1;        // X[c0:, c0:] is upper triangular
        if ((rc = skm_gemm_launch(ctx, KIND, t, nb, st, "k_gemm_trinv_a"))) return rc;
        GemmArgs x = skm_gemm_args(Mb + (long)r0 * R + r0, Tb, Mb + (long)r0 * R + c0, bs, rem, bs, R, R, R);
        x.alpha = -1.0;
        x.sA = x.sC = RR;
        x.sB = (long)CH_BLK * R;
        if ((rc = skm_gemm_launch(ctx, KIND, x, nb, st, "k_gemm_trinv_b"))) return rc;
    }
    if (!out) return SKM_OK;   // the caller works with X = U^-1 itself
    // M^-1 = X X'
    GemmArgs p = skm_gemm_args(Mb, Mb, out, R, R, R, R, R, R);
    p.transB = 1;
    p.tile_mask = upper_only ? 1 : 0;
    p.ktrim = 2;            // X upper triangular: (X X')[i][j] sums k >= max(i, j)
    p.sA = p.sB = p.sC = RR;
    return skm_gemm_launch(ctx, KIND, p, nb, st, "k_gemm_xxt");
}

struct TvWs {
    size_t off_mb, off_linv, off_e, off_tb, off_aux, off_eh32, off_map, off_cs, off_auxs, off_y, total;
    size_t off_gs, off_ns, off_nt, off_es;      // bf16 x 3 planes of G, of the batch's N (both orientations) and of e_hh
    size_t off_tp, off_ep;                      // planes of the loading matrix (aux product) and of a batch's e_h' (C accumulation)
    size_t off_g64, off_f64, off_n64;           // SKM_TV_PRECISE: float64 Gram matrices, f~ rows and N of a chunk
    int ksplit;
    int nzb3;       // split-K slabs of the aux product on the split GEMM (tv_nz_b3)
};

// The L build and the A accumulation run on the bf16 matrix cores from operands split into three bf16 terms
// (skm_gemm_b3.hip) unless SKM_TV_SPLIT=0 (the float32 MFMA products of rounds 1-2).
static bool tv_split()
{
    static const bool on = [] {
        const char *e = getenv("SKM_TV_SPLIT");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

// split-K factor of the aux = f~ T GEMM.  Every K chunk is a launch of its own (float32 sums over TV_AUX_CHUNK terms,
// float64 across chunks), and a launch whose 128 x 128 output tiles make only a few rounds over the 2 x CU resident
// workgroups idles in its last round: with 128 x 128 tiles 50 000 sessions x R = 400 were 1564 tiles = 3.05 rounds, a
// quarter of the fourth lost sixty times.  Splitting K over slabs puts several chunks into one launch until it has at least 8 rounds.
// R = 400 is 3.125 row tiles of 128 but 6.25 of 64: the 64-wide tile is the default (48.0 -> 43.7 ms per 50 000 sessions), SKM_TV_CACC=big restores the 128-wide one
static int tv_cacc_kind()
{
    static const int kind = [] {
        const char *e = getenv("SKM_TV_CACC");
        return (e && !strcmp(e, "big")) ? SKM_GEMM_F32_F64OUT : SKM_GEMM_F32S_F64OUT;
    }();
    return kind;
}

// The same question for the aux product on the split GEMM (256 x 256 tiles, f~ from its planes): K is cut into slabs
// that run side by side in one launch until the launch makes at least 8 rounds over the CUs -- 4000 sessions are 32
// tiles, and the 392 tiles of 50 000 sessions are one full round and one half-empty one (28.2 -> 22 ms with 8 slabs).
static int tv_nz_b3(int cu, long S, int R, long CD)
{
    const long tiles = ((S + 255) / 256) * ((R + 255) / 256);
    int nz = 1;
    while (nz < 16 && tiles * nz < 8L * cu && CD % (nz * 2) == 0 && (CD / (nz * 2)) % 16 == 0) nz *= 2;
    return nz;
}

static int tv_ksplit(int cu, long S, int R, int C)
{
    const long tiles = ((S + 63) / 64) * ((R + 63) / 64);      // 64 x 64 tiles (TV_AUX_KIND)
    int ks = 1;
    while (ks < 16 && tiles * ks < 16L * cu && C % (ks * 2) == 0) ks *= 2;
    return ks;
}

#define TV_PRECISE_ROWS 1024    // sessions per float64 conversion chunk of the aux product (SKM_TV_PRECISE)
static TvWs tv_ws(int R, long S, int batch, int cu = 256, int C = 0, int D = 0, bool precise = false)
{
    TvWs w;
    const size_t RR = (size_t)R * R;
    size_t off = 0;
    w.off_mb = off;
    off += skm_align_up((size_t)batch * RR * sizeof(double), 256);
    w.off_linv = off;
    off += skm_align_up((size_t)batch * RR * sizeof(double), 256);
    w.off_e = off;
    off += skm_align_up((size_t)batch * RR * sizeof(float), 256);
    w.off_tb = off;
    off += skm_align_up((size_t)batch * CH_BLK * R * sizeof(double), 256);
    w.off_aux = off;
    off += skm_align_up((size_t)S * R * sizeof(double), 256);
    w.off_eh32 = off;
    off += skm_align_up((size_t)S * R * sizeof(float), 256);
    w.off_map = off;
    off += skm_align_up((size_t)tv_tri_cols(R) * sizeof(int), 256);
    w.off_cs = off;
    off += skm_align_up((size_t)TV_CS_SPLIT * tv_tri_cols(R) * sizeof(double), 256);
    w.ksplit = (S > 0 && C > 0) ? tv_ksplit(cu, S, R, C) : 1;
    w.nzb3 = (S > 0 && C > 0 && tv_split()) ? tv_nz_b3(cu, S, R, (long)C * D) : 1;
    w.off_auxs = off;
    {
        const int slabs = w.ksplit > w.nzb3 ? w.ksplit : w.nzb3;
        if (slabs > 1) off += skm_align_up((size_t)slabs * S * R * sizeof(double), 256);
    }
    // M-step product Y[batch][D][R]: fits the L^-1 region when D <= R, else a region of its own
    w.off_y = w.off_linv;
    if (D > R) {
        w.off_y = off;
        off += skm_align_up((size_t)batch * D * R * sizeof(double), 256);
    }
    w.off_g64 = w.off_f64 = w.off_n64 = 0;
    if (precise && S > 0 && C > 0) {
        w.off_g64 = off;
        off += skm_align_up((size_t)C * RR * sizeof(double), 256);
        w.off_f64 = off;
        off += skm_align_up((size_t)(batch < TV_PRECISE_ROWS ? batch : TV_PRECISE_ROWS) * C * D * sizeof(double), 256);
        w.off_n64 = off;
        off += skm_align_up((size_t)batch * tv_ldn(C) * sizeof(double), 256);
    }
    w.off_gs = w.off_ns = w.off_nt = w.off_es = w.off_tp = w.off_ep = 0;
    if (S > 0 && C > 0 && tv_split()) {
        const long NT = tv_tri_cols(R);
        w.off_gs = off;
        off += skm_align_up(skm_b3_plane_bytes(NT, C), 256);
        w.off_ns = off;
        off += skm_align_up(skm_b3_plane_bytes(batch, C), 256);
        w.off_nt = off;
        off += skm_align_up(skm_b3_plane_bytes(C, batch), 256);
        w.off_es = off;
        off += skm_align_up(skm_b3_plane_bytes(NT, batch), 256);
        w.off_tp = off;
        off += skm_align_up(skm_b3_plane_bytes(R, (long)C * D), 256);
        w.off_ep = off;
        off += skm_align_up(skm_b3_plane_bytes(R, batch), 256);
    }
    w.total = off;
    return w;
}

// Precision of the per-session chain L -> Cholesky -> (U^-1 -> L^-1).  Default float32 ("fp32 compute, fp64 final
// reductions"): the batched 64-wide panel products of the chain are bound by HBM bytes (4.3 TB/s of minimal traffic in
// float64 whatever the panel, tools/time_chain_gemm.py), so float32 storage halves their time; L has eigenvalues >= 1
// and e_h costs cond(L) * 6e-8.  SKM_TV_CHAIN=f64 keeps the float64 chain (the M-step is float64 either way).
static bool tv_chain_f64()
{
    const char *e = getenv("SKM_TV_CHAIN");
    return e && (strcmp(e, "f64") == 0 || strcmp(e, "F64") == 0);
}

// Sessions per launch of the per-session chain: the whole batch.  SKM_TV_SUB=<n> cuts it into sub-batches of n sessions
// -- the experiment that asked whether chains of launches on a sub-batch that fits the 256 MB Infinity Cache beat
// streaming the batch from HBM.  They do not (round 3, one box, 50 000 sessions: chain 98 ms whole, 112 ms at 1024,
// 136 at 512, 183 at 256, 279 at 128): the one-wave-per-matrix diagonal kernel and the 64-wide products run under-filled
// long before the working set is small enough.
static int tv_chain_sub(int nb)
{
    static const int env = [] {
        const char *e = getenv("SKM_TV_SUB");
        return e ? atoi(e) : 0;
    }();
    return (env > 0 && env < nb) ? env : nb;
}

static bool tv_poison()
{
    static const bool on = [] {
        const char *e = getenv("SKM_TV_POISON");
        return e && atoi(e) != 0;
    }();
    return on;
}

struct TvBatch {
    const double *G64;      // SKM_TV_PRECISE: float64 Gram matrices (the L build is a float64 product), or NULL
    double *n64;            // scratch for the batch's N in float64
    const void *Gs;         // bf16 x 3 planes of G over the mapped columns, or NULL (float32 MFMA L build)
    void *Ns;               // scratch for the planes of this batch's N
    const float *naug, *G;
    const int *cmap;
    const double *aux;
    double *e_h, *iv_sigma;
    float *eh32, *E;
    int nb, NT, C, R;
    long ldn;
    bool train;
};

// One batch of sessions: L = sum_c N_sc G_c (identity added by the diagonal-block kernel), blocked factorisation /
// inverse, e_h (+ iv_sigma, + e_hh for training).  T is the type the matrices are stored and factored in.
template <typename T>
static int tv_batch_chain(SkmCtx *ctx, const TvBatch &b, char *mb_, char *linv_, char *tb_, hipStream_t st)
{
    T *Mb = reinterpret_cast<T *>(mb_), *Linv = reinterpret_cast<T *>(linv_), *Tb = reinterpret_cast<T *>(tb_);
    const int R = b.R, nb = b.nb;
    const long RR = (long)R * R;
    int rc;
    // No clearing of the batch's matrices: the L build writes every row from the start of its 4-wide diagonal block,
    // the diagonal-block kernel reads r <= c only and writes its whole 64 x 64 block (zeros below the diagonal), and
    // every product of the chain starts its tiles at multiples of 64 and is trimmed to the block-upper part of its
    // triangular operands -- the blocks below the block diagonal are never read (they cost 32 GB of memset per 50 000
    // sessions).  SKM_TV_POISON=1 fills them with NaN instead: tests/test_gpu_tv.py proves the claim with it.
    if (tv_poison()) SKM_HIP(hipMemsetAsync(Mb, 0xFF, (size_t)nb * RR * sizeof(T), st));
    GemmArgs l = skm_gemm_args(b.naug, b.G, Mb, nb, b.NT, b.C, b.ldn, RR, RR);
    l.colmap = b.cmap;
    l.colmap_vec = (R % 4 == 0);
    l.swz = 1;          // the session tiles of one G column tile share it in one XCD's L2
    // K = C float32 MFMA terms per entry, accumulated in float32 like the <= `batch` terms of the A accumulation
    // (the operands are float32 already; ~sqrt(C) * 6e-8 on top of their rounding)
    if (b.G64) {
        // float64 all the way: N (float32 values, exact) widened, G from the float64 loading matrix
        {
            SkmProfScope ps(ctx, "k_cvt_f32_f64", st);
            hipLaunchKernelGGL(k_cvt_f32_f64, dim3(ctx->cu_count * 4), dim3(256), 0, st, b.naug, b.n64, (long)nb * b.ldn);
        }
        SKM_HIP(hipGetLastError());
        SKM_REQUIRE(sizeof(T) == 8, "precise TV E-step needs the float64 chain");
        GemmArgs l64 = skm_gemm_args(b.n64, b.G64, Mb, nb, b.NT, b.C, b.ldn, RR, RR);
        l64.colmap = b.cmap;
        l64.colmap_vec = (R % 2 == 0);
        l64.swz = 1;
        if ((rc = skm_gemm_launch(ctx, SKM_GEMM_F64_F64, l64, 1, st, "k_gemm_lbuild"))) return rc;
    } else if (b.Gs) {
        // the same product on the bf16 matrix cores: N split per batch (33 MB), G split once per call
        if ((rc = skm_b3_split(ctx, b.naug, nb, b.C, b.ldn, 1, nullptr, b.Ns, st, "k_b3_split"))) return rc;
        if ((rc = skm_b3_gemm(ctx, b.Ns, nb, b.Gs, b.NT, b.C, Mb, sizeof(T) == 8, RR, b.cmap, 0, st, "k_gemm_lbuild")))
            return rc;
    } else if ((rc = skm_gemm_launch(ctx, sizeof(T) == 8 ? SKM_GEMM_F32_F64OUT : SKM_GEMM_F32_F32, l, 1, st,
                                     "k_gemm_lbuild")))
        return rc;
    const bool solve_only = !b.train && !b.iv_sigma;
    // The ~25 panel products of the factorisation / inverse each touch a few hundred KB per session; over the whole
    // batch (4096 sessions: 0.9 GB per launch) every one of them streams from HBM (PMC: 3.2-3.6 TB/s, profiles/
    // r3_pmc_tv_start.json).  See tv_chain_sub for the sub-batch experiment.
    const int sub = tv_chain_sub(nb);
    for (int s0 = 0; s0 < nb; s0 += sub) {
        const int ns = (nb - s0) < sub ? (nb - s0) : sub;
        if ((rc = blocked_inverse<T>(ctx, Mb + (long)s0 * RR, (T *)nullptr, Tb, ns, R, 1, 1, st, solve_only))) return rc;
    }
    (void)Linv;
    if (solve_only) {
        SkmProfScope ps(ctx, "k_tv_solve", st);
        hipLaunchKernelGGL(k_tv_solve<T>, dim3(nb), dim3(256), (R + 5 * CH_BLK) * sizeof(double), st, Mb, R, b.aux, b.e_h,
                           b.eh32);
    } else {
        // e_h = X (X' aux) and diag(L^-1) from X = U^-1 alone (two triangular products, R^2 work, X read once)
        {
            SkmProfScope ps(ctx, "k_tv_finish", st);
            hipLaunchKernelGGL(k_tv_finish_tri<T>, dim3(nb), dim3(256), 2 * R * sizeof(double), st, Mb, R, b.aux, b.e_h,
                               b.eh32, b.iv_sigma);
        }
        SKM_HIP(hipGetLastError());
        if (b.train) {
            // e_hh = L^-1 + e_h e_h' = X X' + e_h e_h' written ONCE, as the float32 operand of the A accumulation:
            // the rank-one term rides in the epilogue of the X X' product (L^-1 itself is never stored, and no
            // kernel reads it back to add e_h e_h')
            GemmArgs p = skm_gemm_args(Mb, Mb, b.E, R, R, R, R, R, R);
            p.transB = 1;
            p.tile_mask = 1;
            p.ktrim = 2;            // X upper triangular: (X X')[i][j] sums k >= max(i, j)
            p.sA = p.sB = p.sC = RR;
            p.r1 = b.e_h;
            p.r1_stride = R;
            if ((rc = skm_gemm_launch(ctx, sizeof(T) == 8 ? SKM_GEMM_F64_F32OUT : SKM_GEMM_F32S_F32, p, nb, st,
                                      "k_gemm_xxt")))
                return rc;
        }
    }
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" int skm_tv_naug_stride(int C) { return tv_ldn(C); }

extern "C" size_t skm_tv_workspace_bytes(void *ctx, int C, int D, int R, long S, int batch, int chain)
{
    if (R <= 0 || S < 0 || batch <= 0) return 0;
    const int cu = ctx ? static_cast<SkmCtx *>(ctx)->cu_count : 256;
    return tv_ws(R, S, batch, cu, C, D, chain == SKM_TV_PRECISE).total;
}


extern "C" int skm_tv_whiten(void *ctx_, const void *stat0, const void *stat1, int in_f64, long S, int C, int D,
                             const double *mu, const double *sqrt_ic, float *ftilde, float *naug, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && stat0 && stat1 && mu && sqrt_ic && ftilde && naug, "skm_tv_whiten: NULL argument");
    SKM_REQUIRE(S > 0 && C > 0 && D > 0, "skm_tv_whiten: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = ctx->cu_count * 8;
    SkmProfScope ps(ctx, "k_tv_whiten", st);
    if (in_f64)
        hipLaunchKernelGGL(k_tv_whiten<double>, dim3(blocks), dim3(256), 0, st, static_cast<const double *>(stat0),
                           static_cast<const double *>(stat1), S, C, D, mu, sqrt_ic, ftilde, naug, tv_ldn(C));
    else
        hipLaunchKernelGGL(k_tv_whiten<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float *>(stat0),
                           static_cast<const float *>(stat1), S, C, D, mu, sqrt_ic, ftilde, naug, tv_ldn(C));
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" int skm_tv_prepare(void *ctx_, const double *T, int C, int D, int R, float *T32, float *G, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && T && T32 && G, "skm_tv_prepare: NULL argument");
    SKM_REQUIRE(C > 0 && D > 0 && R > 0, "skm_tv_prepare: bad shape");
    hipStream_t st = (hipStream_t)stream;
    {
        SkmProfScope ps(ctx, "k_cvt_f64_f32", st);
        hipLaunchKernelGGL(k_cvt_f64_f32, dim3(ctx->cu_count * 8), dim3(256), 0, st, T, T32, (long)C * D * R);
    }
    SKM_HIP(hipGetLastError());
    // G_c = T_c' T_c, T_c = rows [cD, (c+1)D) of T
    GemmArgs g = skm_gemm_args(T, T, G, R, R, D, R, R, R);
    g.transA = 1;
    g.tile_mask = 1;
    g.sA = g.sB = (long)D * R;
    g.sC = (long)R * R;
    return skm_gemm_launch(ctx, SKM_GEMM_F64_F32OUT, g, C, st, "k_gemm_gram");
}

// Planes of f~ for skm_tv_estep_planes: [ rows image: sessions x (k = supervector) | columns image: supervector x (k = sessions) ]
static size_t tv_ft_rows_bytes(long S, int C, int D) { return skm_align_up(skm_b3_plane_bytes(S, (long)C * D), 256); }

extern "C" size_t skm_tv_ftilde_rows_bytes(long S, int C, int D)
{
    if (S <= 0 || C <= 0 || D <= 0) return 0;
    return tv_ft_rows_bytes(S, C, D);
}

extern "C" int skm_tv_whiten_planes(void *ctx_, const void *stat0, const void *stat1, int in_f64, long S, int C, int D,
                                    const double *mu, const double *sqrt_ic, void *rows_planes, float *naug, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && stat0 && stat1 && mu && sqrt_ic && rows_planes && naug, "skm_tv_whiten_planes: NULL argument");
    SKM_REQUIRE(S > 0 && C > 0 && D > 0, "skm_tv_whiten_planes: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long Sp = skm_b3_rows_padded(S), KG = skm_b3_kgroups((long)C * D);
    const int blocks = ctx->cu_count * 32;
    SkmProfScope ps(ctx, "k_tv_whiten_planes", st);
    if (in_f64)
        hipLaunchKernelGGL(k_tv_whiten_planes<double>, dim3(blocks), dim3(256), 0, st, static_cast<const double *>(stat0),
                           static_cast<const double *>(stat1), S, C, D, mu, sqrt_ic, static_cast<tv_i32x4 *>(rows_planes),
                           Sp, KG, naug, tv_ldn(C));
    else
        hipLaunchKernelGGL(k_tv_whiten_planes<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float *>(stat0),
                           static_cast<const float *>(stat1), S, C, D, mu, sqrt_ic, static_cast<tv_i32x4 *>(rows_planes),
                           Sp, KG, naug, tv_ldn(C));
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" size_t skm_tv_ftilde_planes_bytes(long S, int C, int D)
{
    if (S <= 0 || C <= 0 || D <= 0) return 0;
    return tv_ft_rows_bytes(S, C, D) + skm_align_up(skm_b3_plane_bytes((long)C * D, S), 256);
}

extern "C" int skm_tv_ftilde_planes(void *ctx_, const float *ftilde, long S, int C, int D, void *planes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && ftilde && planes, "skm_tv_ftilde_planes: NULL argument");
    SKM_REQUIRE(S > 0 && C > 0 && D > 0, "skm_tv_ftilde_planes: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const long CD = (long)C * D;
    char *p = static_cast<char *>(planes);
    int rc;
    if ((rc = skm_b3_split(ctx, ftilde, S, CD, CD, 1, nullptr, p, st, "k_b3_split_f"))) return rc;
    return skm_b3_split(ctx, ftilde, CD, S, CD, 0, nullptr, p + tv_ft_rows_bytes(S, C, D), st, "k_b3_split_f");
}

// k per float32 accumulation run of the aux product on the split GEMM (float64 across runs, in the output)
#define TV_AUX_CHUNK_B3 4096

static int tv_estep_impl(void *ctx_, const float *naug, const float *ftilde, long S, int C, int D, int R,
                         const float *T32, const float *G, int batch, double *A_aug, double *Cacc, double *e_h,
                         double *iv_sigma, int chain, const double *T64, void *ws_, size_t ws_bytes, void *stream,
                         const void *ft_planes)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && naug && T32 && G && e_h && ws_, "skm_tv_estep: NULL argument");
    // extraction may come with the rows image of the planes alone (skm_tv_whiten_planes): no float32 f~ then
    SKM_REQUIRE(ftilde || (ft_planes && !A_aug && chain != SKM_TV_PRECISE && tv_split()),
                "skm_tv_estep: f~ may only be NULL for extraction from its planes (not training, not SKM_TV_PRECISE, "
                "not with SKM_TV_SPLIT=0)");
    SKM_REQUIRE(S > 0 && C > 0 && D > 0 && R > 0 && batch > 0, "skm_tv_estep: bad shape");
    SKM_REQUIRE((A_aug == nullptr) == (Cacc == nullptr), "skm_tv_estep: A_aug and Cacc go together");
    const bool precise = chain == SKM_TV_PRECISE;
    SKM_REQUIRE(!precise || T64 != nullptr, "skm_tv_estep: SKM_TV_PRECISE needs the float64 loading matrix");
    const TvWs w = tv_ws(R, S, batch, ctx->cu_count, C, D, precise);
    if (ws_bytes < w.total) {
        skm_set_error("skm_tv_estep: workspace too small (%zu < %zu)", ws_bytes, w.total);
        return SKM_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = static_cast<char *>(ws_);
    float *E = reinterpret_cast<float *>(ws + w.off_e);
    double *aux = reinterpret_cast<double *>(ws + w.off_aux);
    float *eh32 = reinterpret_cast<float *>(ws + w.off_eh32);
    const long RR = (long)R * R, CD = (long)C * D, ldn = tv_ldn(C);
    const bool train = A_aug != nullptr;
    int rc;

    int *cmap = reinterpret_cast<int *>(ws + w.off_map);
    const int NT = (int)tv_tri_cols(R);
    {
        SkmProfScope ps(ctx, "k_tri_map", st);
        hipLaunchKernelGGL(k_tri_map, dim3(R), dim3(256), 0, st, cmap, R);
    }
    SKM_HIP(hipGetLastError());
    // aux = f~ T for every session (factor_analyser.py:88): K = C*D is long (122 880), so the float32 MFMA sums run
    // over chunks of TV_AUX_CHUNK k (plain k loop, no float64 registers: the fast instantiation) and the chunks are
    // added in float64 -- in the output itself (beta = 1) when there are enough output tiles to fill the chip, in
    // split-K slabs when S is small.
    double *G64 = nullptr;
    if (precise) {
        // Everything e_h depends on in float64, as the reference has it (factor_analyser.py:86-90): G_c = T_c'T_c from
        // the float64 T, L_s = I + sum_c N_sc G_c and aux = f~ T as float64 products (f~ and N are float32 VALUES in the
        // reference too -- read from the float32 statistics file -- so widening them is exact).  For long sessions /
        // large loadings: the float32 products perturb L by 6e-8 |L|, which e_h sees multiplied by cond(L_s).
        G64 = reinterpret_cast<double *>(ws + w.off_g64);
        GemmArgs gg = skm_gemm_args(T64, T64, G64, R, R, D, R, R, R);
        gg.transA = 1;
        gg.tile_mask = 1;
        gg.sA = gg.sB = (long)D * R;
        gg.sC = RR;
        if ((rc = skm_gemm_launch(ctx, SKM_GEMM_F64_F64, gg, C, st, "k_gemm_gram"))) return rc;
        double *f64 = reinterpret_cast<double *>(ws + w.off_f64);
        const long rows = batch < TV_PRECISE_ROWS ? batch : TV_PRECISE_ROWS;
        for (long r0 = 0; r0 < S; r0 += rows) {
            const long nr = (S - r0) < rows ? (S - r0) : rows;
            {
                SkmProfScope ps(ctx, "k_cvt_f32_f64", st);
                hipLaunchKernelGGL(k_cvt_f32_f64, dim3(ctx->cu_count * 8), dim3(256), 0, st, ftilde + r0 * CD, f64, nr * CD);
            }
            SKM_HIP(hipGetLastError());
            GemmArgs g = skm_gemm_args(f64, T64, aux + r0 * R, (int)nr, R, (int)CD, CD, R, R);
            if ((rc = skm_gemm_launch(ctx, SKM_GEMM_F64_F64, g, 1, st, "k_gemm_aux"))) return rc;
        }
    } else if (ft_planes && w.off_tp) {
        // f~ was split once for the whole training run (skm_tv_ftilde_planes): the aux product on the bf16 matrix
        // cores, float32 sums over runs of TV_AUX_CHUNK_B3 k, float64 across the runs in the output
        void *Tp = ws + w.off_tp;
        if ((rc = skm_b3_split(ctx, T32, R, CD, R, 0, nullptr, Tp, st, "k_b3_split"))) return rc;
        const long Sp = skm_b3_rows_padded(S), Rp = skm_b3_rows_padded(R), KGt = skm_b3_kgroups(CD);
        const char *fa = static_cast<const char *>(ft_planes);
        // few sessions = few 256 x 256 output tiles (4000 sessions: 32): K is then cut into nz slabs that run side by
        // side in one launch (float64 slabs, summed afterwards), like the split-K of the float32 route
        const int nz = w.nzb3;
        const long Kz = CD / nz;
        double *out = nz > 1 ? reinterpret_cast<double *>(ws + w.off_auxs) : aux;
        for (long k0 = 0; k0 < Kz; k0 += TV_AUX_CHUNK_B3) {
            const long kn = (Kz - k0) < TV_AUX_CHUNK_B3 ? (Kz - k0) : TV_AUX_CHUNK_B3;
            if ((rc = skm_b3_gemm(ctx, fa + (k0 / 8) * Sp * 16, S, static_cast<const char *>(Tp) + (k0 / 8) * Rp * 16, R, kn,
                                  out, 1, R, nullptr, k0 ? 1 : 0, st, "k_gemm_aux", KGt * Sp, KGt * Rp, nz, (Kz / 8) * Sp,
                                  (Kz / 8) * Rp, S * R)))
                return rc;
        }
        if (nz > 1) {
            SkmProfScope ps(ctx, "k_sum_slabs", st);
            hipLaunchKernelGGL(k_sum_slabs, dim3(ctx->cu_count * 2), dim3(256), 0, st, aux, out, S * R, nz);
            SKM_HIP(hipGetLastError());
        }
    } else if (w.ksplit > 1) {
        double *slabs = reinterpret_cast<double *>(ws + w.off_auxs);
        const long kc = CD / w.ksplit;
        for (long k0 = 0; k0 < kc; k0 += TV_AUX_CHUNK) {
            const int kn = (int)((kc - k0) < TV_AUX_CHUNK ? (kc - k0) : TV_AUX_CHUNK);
            GemmArgs g = skm_gemm_args(ftilde + k0, T32 + k0 * R, slabs, (int)S, R, kn, CD, R, R);
            g.sA = kc;
            g.sB = kc * R;
            g.sC = S * R;
            g.beta = k0 ? 1.0 : 0.0;
            g.swz = 2;          // the R/128 column tiles share the f~ row tile
            if ((rc = skm_gemm_launch(ctx, TV_AUX_KIND, g, w.ksplit, st, "k_gemm_aux"))) return rc;
        }
        SkmProfScope ps(ctx, "k_sum_slabs", st);
        hipLaunchKernelGGL(k_sum_slabs, dim3(ctx->cu_count * 2), dim3(256), 0, st, aux, slabs, S * R, w.ksplit);
        SKM_HIP(hipGetLastError());
    } else {
        for (long k0 = 0; k0 < CD; k0 += TV_AUX_CHUNK) {
            const int kn = (int)((CD - k0) < TV_AUX_CHUNK ? (CD - k0) : TV_AUX_CHUNK);
            GemmArgs g = skm_gemm_args(ftilde + k0, T32 + k0 * R, aux, (int)S, R, kn, CD, R, R);
            g.beta = k0 ? 1.0 : 0.0;
            g.swz = 2;
            if ((rc = skm_gemm_launch(ctx, TV_AUX_KIND, g, 1, st, "k_gemm_aux"))) return rc;
        }
    }
    const bool split = !precise && (w.off_gs != 0 || w.off_ns != 0);
    void *Gs = nullptr;
    if (split) {
        Gs = ws + w.off_gs;
        if ((rc = skm_b3_split(ctx, G, NT, C, RR, 0, cmap, Gs, st, "k_b3_split"))) return rc;
    }
    SKM_REQUIRE(chain >= SKM_TV_CHAIN_DEFAULT && chain <= SKM_TV_PRECISE,
                "skm_tv_estep: chain must be SKM_TV_CHAIN_DEFAULT / _F32 / _F64 or SKM_TV_PRECISE");
    const bool chain64 = chain == SKM_TV_CHAIN_DEFAULT ? tv_chain_f64() : chain >= SKM_TV_CHAIN_F64;
    for (long b0 = 0; b0 < S; b0 += batch) {
        const int nb = (int)((S - b0) < batch ? (S - b0) : batch);
        TvBatch tb;
        tb.naug = naug + b0 * ldn;
        tb.G = G;
        tb.Gs = Gs;
        tb.G64 = G64;
        tb.n64 = precise ? reinterpret_cast<double *>(ws + w.off_n64) : nullptr;
        tb.Ns = split ? ws + w.off_ns : nullptr;
        tb.cmap = cmap;
        tb.aux = aux + b0 * R;
        tb.e_h = e_h + b0 * R;
        tb.eh32 = eh32 + b0 * R;
        tb.iv_sigma = iv_sigma ? iv_sigma + b0 * R : nullptr;
        tb.E = train ? E : nullptr;
        tb.nb = nb;
        tb.NT = NT;
        tb.C = C;
        tb.R = R;
        tb.ldn = ldn;
        tb.train = train;
        rc = chain64 ? tv_batch_chain<double>(ctx, tb, ws + w.off_mb, ws + w.off_linv, ws + w.off_tb, st)
                     : tv_batch_chain<float>(ctx, tb, ws + w.off_mb, ws + w.off_linv, ws + w.off_tb, st);
        if (rc) return rc;
        if (train) {
            // A[c] += sum_s N_sc e_hh(s).  Row C of A_aug (= _R, the plain sum of e_hh over the sessions) is NOT taken
            // through the ones column of naug: C + 1 = 2049 rows would open a 17th row of 128-wide tiles for one
            // row; a column-sum kernel reads E once instead (float64 sums, fixed order).
            GemmArgs a = skm_gemm_args(naug + b0 * ldn, E, A_aug, C, NT, nb, ldn, RR, RR);
            a.transA = 1;
            a.beta = 1.0;
            a.colmap = cmap;
            a.colmap_vec = (R % 4 == 0);
            a.swz = 1;      // the component tiles of one e_hh column tile share it
            if (split) {
                // N' (k = session) and e_hh (k = session, mapped columns) split for this batch, then the product
                if ((rc = skm_b3_split(ctx, naug + b0 * ldn, C, nb, ldn, 0, nullptr, ws + w.off_nt, st, "k_b3_split")))
                    return rc;
                if ((rc = skm_b3_split(ctx, E, NT, nb, RR, 0, cmap, ws + w.off_es, st, "k_b3_split_e"))) return rc;
                if ((rc = skm_b3_gemm(ctx, ws + w.off_nt, C, ws + w.off_es, NT, nb, A_aug, 1, RR, cmap, 1, st,
                                      "k_gemm_aacc")))
                    return rc;
            } else if ((rc = skm_gemm_launch(ctx, SKM_GEMM_F32_F64OUT, a, 1, st, "k_gemm_aacc"))) return rc;
            SkmProfScope ps(ctx, "k_ehh_colsum", st);
            double *part = reinterpret_cast<double *>(ws + w.off_cs);      // TV_CS_SPLIT x NT doubles
            hipLaunchKernelGGL(k_ehh_colsum, dim3((NT + 255) / 256, TV_CS_SPLIT), dim3(256), 0, st, E, RR, nb, cmap, NT, part);
            hipLaunchKernelGGL(k_ehh_colsum_add, dim3((NT + 255) / 256), dim3(256), 0, st, part, cmap, NT,
                               A_aug + (long)C * RR);
            SKM_HIP(hipGetLastError());
        }
    }
    if (train) {
        // C += e_h' f~ (factor_analyser.py:125), one product per batch like the A accumulation: float32 MFMA sums over
        // the <= `batch` sessions of a batch, float64 across batches (the plain k-loop with BK = 32 is 10 % faster here
        // than one float64-folding product over all sessions)
        const bool cacc_planes = ft_planes && w.off_ep && !precise && batch % 16 == 0;
        for (long b0 = 0; b0 < S; b0 += batch) {
            const int nb = (int)((S - b0) < batch ? (S - b0) : batch);
            if (cacc_planes) {
                // e_h' of the batch split (k = session), f~ from the columns image at the batch's k range
                void *Ep = ws + w.off_ep;
                if ((rc = skm_b3_split(ctx, eh32 + b0 * R, R, nb, R, 0, nullptr, Ep, st, "k_b3_split"))) return rc;
                const long CDp = skm_b3_rows_padded(CD), KGs = skm_b3_kgroups(S);
                const char *fc = static_cast<const char *>(ft_planes) + tv_ft_rows_bytes(S, C, D);
                if ((rc = skm_b3_gemm(ctx, Ep, R, fc + (b0 / 8) * CDp * 16, CD, nb, Cacc, 1, CD, nullptr, 1, st, "k_gemm_cacc",
                                      0, KGs * CDp)))
                    return rc;
                continue;
            }
            GemmArgs c = skm_gemm_args(eh32 + b0 * R, ftilde + b0 * CD, Cacc, R, (int)CD, nb, R, CD, CD);
            c.transA = 1;
            c.beta = 1.0;
            c.swz = 1;      // the R/128 row tiles share the f~ column tile
            if ((rc = skm_gemm_launch(ctx, tv_cacc_kind(), c, 1, st, "k_gemm_cacc"))) return rc;
        }
    }
    return SKM_OK;
}

extern "C" int skm_tv_estep(void *ctx, const float *naug, const float *ftilde, long S, int C, int D, int R,
                            const float *T32, const float *G, int batch, double *A_aug, double *Cacc, double *e_h,
                            double *iv_sigma, int chain, const double *T64, void *ws, size_t ws_bytes, void *stream)
{
    return tv_estep_impl(ctx, naug, ftilde, S, C, D, R, T32, G, batch, A_aug, Cacc, e_h, iv_sigma, chain, T64, ws, ws_bytes,
                         stream, nullptr);
}

extern "C" int skm_tv_estep_planes(void *ctx, const float *naug, const float *ftilde, const void *ft_planes, long S, int C,
                                   int D, int R, const float *T32, const float *G, int batch, double *A_aug, double *Cacc,
                                   double *e_h, double *iv_sigma, int chain, const double *T64, void *ws, size_t ws_bytes,
                                   void *stream)
{
    return tv_estep_impl(ctx, naug, ftilde, S, C, D, R, T32, G, batch, A_aug, Cacc, e_h, iv_sigma, chain, T64, ws, ws_bytes,
                         stream, ft_planes);
}

// M-step: T_c = (A_c^-1 C_c)' for every component; A_aug rows 0..C-1 are destroyed.
extern "C" int skm_tv_mstep(void *ctx_, double *A_aug, const double *Cacc, int C, int D, int R, int batch,
                            double *F_out, void *ws_, size_t ws_bytes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && A_aug && Cacc && F_out && ws_, "skm_tv_mstep: NULL argument");
    SKM_REQUIRE(C > 0 && D > 0 && R > 0 && batch > 0, "skm_tv_mstep: bad shape");
    const TvWs w = tv_ws(R, 0, batch, 256, 0, D);
    if (ws_bytes < w.total) {
        skm_set_error("skm_tv_mstep: workspace too small (%zu < %zu)", ws_bytes, w.total);
        return SKM_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = static_cast<char *>(ws_);
    double *Ybuf = reinterpret_cast<double *>(ws + w.off_y);
    double *Tb = reinterpret_cast<double *>(ws + w.off_tb);
    const long RR = (long)R * R, CD = (long)C * D;
    int rc;
    for (int c0 = 0; c0 < C; c0 += batch) {
        const int nb = (C - c0) < batch ? (C - c0) : batch;
        // A_c^-1 = X X' with X = U^-1 (upper triangular, left in place of A_c): T_c = (C_c' X) X' is two D x R x R
        // products per component instead of the R^3 product X X' plus one
        if ((rc = blocked_inverse<double>(ctx, A_aug + c0 * RR, (double *)nullptr, Tb, nb, R, 0, 0, st))) return rc;
        double *Y = Ybuf;   // [nb][D][R]
        GemmArgs g = skm_gemm_args(Cacc + (long)c0 * D, A_aug + c0 * RR, Y, D, R, R, CD, R, R);
        g.transA = 1;       // Y[d][r] = sum_k C[k][cD+d] X[k][r]
        g.ktrim = 1;        // X[k][r] = 0 for k > r
        g.sA = D;
        g.sB = RR;
        g.sC = (long)D * R;
        if ((rc = skm_gemm_launch(ctx, SKM_GEMM_F64_F64, g, nb, st, "k_gemm_mstep"))) return rc;
        GemmArgs t = skm_gemm_args(Y, A_aug + c0 * RR, F_out + (long)c0 * D * R, D, R, R, R, R, R);
        t.transB = 1;       // T_c[d][r] = sum_j Y[d][j] X[r][j]
        t.ktrim = 4;        // X[r][j] = 0 for j < r
        t.sA = (long)D * R;
        t.sB = RR;
        t.sC = (long)D * R;
        if ((rc = skm_gemm_launch(ctx, SKM_GEMM_F64_F64, t, nb, st, "k_gemm_mstep"))) return rc;
    }
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// i-vector back-end (next row): length normalisation of the rows of X[S][R] in place, norm clipped below at
// 1e-8 (StatServer.norm_stat1, sidekit/statserver.py:677-680).  One wave per row, float64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_normalize(double *__restrict__ X, long S, int R)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S) return;
    double *x = X + row * R;
    double s = 0.0;
    for (int j = lane; j < R; j += 64) s += x[j] * x[j];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    double nrm = sqrt(s);
    if (nrm < 1e-8) nrm = 1e-8;
    for (int j = lane; j < R; j += 64) x[j] = x[j] / nrm;
}

extern "C" int skm_row_normalize(void *ctx_, double *X, long S, int R, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && S >= 0 && R > 0 && (S == 0 || X), "skm_row_normalize: bad argument");
    if (S == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_row_normalize", (hipStream_t)stream);
    hipLaunchKernelGGL(k_row_normalize, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, (hipStream_t)stream, X, S, R);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}
