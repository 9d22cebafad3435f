// skm_gemm.hip -- generic strided-batched MFMA GEMM (see skm_gemm.h).
#include "skm_gemm.h"

template <typename T>
struct Mf;

template <>
struct Mf<float> {
    static constexpr int TM = 32, KS = 2, NR = 16;
    typedef f32x16 acc_t;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return SKM_MFMA32(a, b, c); }
    static __device__ __forceinline__ int row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};

template <>
struct Mf<double> {
    static constexpr int TM = 16, KS = 4, NR = 4;
    typedef f64x4 acc_t;
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 MFMA does NOT use the f32 C/D map: row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int row(int r, int lane) { return (lane >> 4) + 4 * r; }
};

template <typename TI, typename TO, bool ACC64>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g)
{
    typedef Mf<TI> M;
    constexpr int TM = M::TM, KS = M::KS, NR = M::NR;
    constexpr int BM = 4 * TM, BN = 4 * TM, BK = 16, PAD = 1;
    constexpr int LDT = BM + PAD;
    constexpr int EPT = BM * BK / 256;  // elements per thread per operand tile
    constexpr int FOLD = 32;            // k-tiles between f64 folds (ACC64)

    __shared__ TI As[BK * LDT];
    __shared__ TI Bs[BK * LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    const TI *A = static_cast<const TI *>(g.A) + (long)z * g.sA;
    const TI *B = static_cast<const TI *>(g.B) + (long)z * g.sB;
    TO *C = static_cast<TO *>(g.C) + (long)z * g.sC;
    int N = g.N;
    if (g.tri_n > 0) {
        const int col0 = (z / g.tri_blk) * g.tri_blk;
        B += (long)z * g.tri_ldz + col0;
        C += (long)z * g.tri_ldz + col0;
        N = g.tri_n - col0;
    }
    if (n0 >= N || m0 >= g.M) return;
    if (g.tile_mask == 1 && (n0 + BN - 1) / g.mask_blk < m0 / g.mask_blk) return;
    const int Mr = g.M, K = g.K;

    TI ra[EPT], rb[EPT];
    auto load_tiles = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int m, k;
            if (g.transA) {
                m = e % BM;
                k = e / BM;
            } else {
                k = e % BK;
                m = e / BK;
            }
            const int gm = m0 + m, gk = k0 + k;
            TI v = 0;
            if (gm < Mr && gk < K) v = g.transA ? A[(long)gk * g.lda + gm] : A[(long)gm * g.lda + gk];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int n, k;
            if (g.transB) {
                k = e % BK;
                n = e / BK;
            } else {
                n = e % BN;
                k = e / BN;
            }
            const int gn = n0 + n, gk = k0 + k;
            TI v = 0;
            if (gn < N && gk < K) v = g.transB ? B[(long)gn * g.ldb + gk] : B[(long)gk * g.ldb + gn];
            rb[i] = v;
        }
    };
    auto store_tiles = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int m, k;
            if (g.transA) {
                m = e % BM;
                k = e / BM;
            } else {
                k = e % BK;
                m = e / BK;
            }
            As[k * LDT + m] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int n, k;
            if (g.transB) {
                k = e % BK;
                n = e / BK;
            } else {
                n = e % BN;
                k = e / BN;
            }
            Bs[k * LDT + n] = rb[i];
        }
    };

    typename M::acc_t acc[2][2];
    double acc64[ACC64 ? 2 : 1][ACC64 ? 2 : 1][ACC64 ? NR : 1];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                acc[mi][ni][r] = 0;
                if (ACC64) acc64[mi][ni][r] = 0.0;
            }

    const int nk = (K + BK - 1) / BK;
    const int lm = lane % TM, lk = lane / TM;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    int since = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / KS; ++kk) {
            TI a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = As[(kk * KS + lk) * LDT + wm * 2 * TM + mi * TM + lm];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = Bs[(kk * KS + lk) * LDT + wn * 2 * TM + ni * TM + lm];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = M::mfma(a[mi], b[ni], acc[mi][ni]);
        }
        if (ACC64 && ++since == FOLD) {
            since = 0;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        acc64[mi][ni][r] += (double)acc[mi][ni][r];
                        acc[mi][ni][r] = 0;
                    }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tiles();
            __syncthreads();
        }
    }

    // epilogue
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int gm = m0 + wm * 2 * TM + mi * TM + M::row(r, lane);
                const int gn = n0 + wn * 2 * TM + ni * TM + lm;
                if (gm < Mr && gn < N && !(g.tile_mask == 1 && gn / g.mask_blk < gm / g.mask_blk)) {
                    double v = (double)acc[mi][ni][r];
                    if (ACC64) v += acc64[mi][ni][r];
                    v *= g.alpha;
                    TO *p = C + (long)gm * g.ldc + gn;
                    if (g.beta != 0.0) v += g.beta * (double)(*p);
                    *p = (TO)v;
                }
            }
}

int skm_gemm_launch(SkmCtx *ctx, int kind, const GemmArgs &g, int batch, hipStream_t st, const char *tag)
{
    SKM_REQUIRE(g.A && g.B && g.C, "gemm(%s): NULL operand", tag);
    SKM_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0 && batch > 0, "gemm(%s): bad shape M=%d N=%d K=%d batch=%d", tag,
                g.M, g.N, g.K, batch);
    const int tile = (kind == SKM_GEMM_F32_F32 || kind == SKM_GEMM_F32_F64ACC) ? 128 : 64;
    const int ncols = g.tri_n > 0 ? g.tri_n : g.N;
    const dim3 grid((ncols + tile - 1) / tile, (g.M + tile - 1) / tile, batch);
    SKM_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm(%s): grid too large", tag);
    SkmProfScope ps(ctx, tag, st);
    switch (kind) {
    case SKM_GEMM_F32_F32: hipLaunchKernelGGL((k_gemm<float, float, false>), grid, dim3(256), 0, st, g); break;
    case SKM_GEMM_F32_F64ACC: hipLaunchKernelGGL((k_gemm<float, double, true>), grid, dim3(256), 0, st, g); break;
    case SKM_GEMM_F64_F64: hipLaunchKernelGGL((k_gemm<double, double, false>), grid, dim3(256), 0, st, g); break;
    case SKM_GEMM_F64_F32OUT: hipLaunchKernelGGL((k_gemm<double, float, false>), grid, dim3(256), 0, st, g); break;
    default: skm_set_error("gemm(%s): unknown kind %d", tag, kind); return SKM_ERR_ARG;
    }
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// Plain GEMM entry point (tests, minimum-divergence product).
extern "C" int skm_gemm(void *ctx_, int kind, const void *A, const void *B, void *C, int M, int N, int K, long lda,
                        long ldb, long ldc, int transA, int transB, double alpha, double beta, int batch, long sA,
                        long sB, long sC, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_gemm: NULL ctx");
    GemmArgs g = skm_gemm_args(A, B, C, M, N, K, lda, ldb, ldc);
    g.transA = transA;
    g.transB = transB;
    g.alpha = alpha;
    g.beta = beta;
    g.sA = sA;
    g.sB = sB;
    g.sC = sC;
    return skm_gemm_launch(ctx, kind, g, batch, (hipStream_t)stream, "k_gemm");
}
