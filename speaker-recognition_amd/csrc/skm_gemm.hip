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

// Operand tile loader: BROWS x BK logical tile (row = m or n, col = k), 256 threads, 16-byte vectors along
// the contiguous dimension of the stored matrix when `vec` (alignment checked on the host).
//   kfast: the stored matrix is contiguous along k (A not transposed / B transposed), else along the row.
template <typename TI, int BROWS, int BK>
struct TileIO {
    static constexpr int VW = 16 / sizeof(TI);
    static constexpr int NV = BROWS * BK / VW / 256;  // vectors per thread
    typedef TI vec_t __attribute__((ext_vector_type(16 / sizeof(TI))));

    // (row, k) of element 0 of vector i of thread tid
    static __device__ __forceinline__ void coord(int tid, int i, bool kfast, int &row, int &k)
    {
        const int e = tid + i * 256;
        if (kfast) {
            k = (e % (BK / VW)) * VW;
            row = e / (BK / VW);
        } else {
            row = (e % (BROWS / VW)) * VW;
            k = e / (BROWS / VW);
        }
    }

    // map != nullptr (row-fast layouts only): logical row gr lives at element offset map[gr] of a stored row
    // (flattened "rows right of the diagonal block" view of an R x R matrix; 4 consecutive logical rows that
    // start at a multiple of 4 are contiguous when vec is set)
    static __device__ __forceinline__ void load(TI (&r)[NV * VW], const TI *__restrict__ P, long ld, bool kfast, bool vec,
                                               int row0, int k0, int rows, int K, int tid,
                                               const int *__restrict__ map = nullptr)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int row, k;
            coord(tid, i, kfast, row, k);
            const int gr = row0 + row, gk = k0 + k;
            if (map && !kfast) {
                const TI *pk = P + (long)gk * ld;
                const bool fullm = gk < K && gr + VW <= rows;
                if (vec && fullm) {
                    const vec_t v = *reinterpret_cast<const vec_t *>(pk + map[gr]);
#pragma unroll
                    for (int j = 0; j < VW; ++j) r[i * VW + j] = v[j];
                } else {
#pragma unroll
                    for (int j = 0; j < VW; ++j) r[i * VW + j] = (gk < K && gr + j < rows) ? pk[map[gr + j]] : (TI)0;
                }
                continue;
            }
            const TI *p = kfast ? (P + (long)gr * ld + gk) : (P + (long)gk * ld + gr);
            const bool full = kfast ? (gr < rows && gk + VW <= K) : (gk < K && gr + VW <= rows);
            if (vec && full) {
                const vec_t v = *reinterpret_cast<const vec_t *>(p);
#pragma unroll
                for (int j = 0; j < VW; ++j) r[i * VW + j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const bool ok = kfast ? (gr < rows && gk + j < K) : (gk < K && gr + j < rows);
                    r[i * VW + j] = ok ? p[j] : (TI)0;
                }
            }
        }
    }

    // LDS image: T[k][row], row stride LDT
    static __device__ __forceinline__ void store(const TI (&r)[NV * VW], TI *T, int LDT, bool kfast, int tid)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int row, k;
            coord(tid, i, kfast, row, k);
            if (kfast) {
#pragma unroll
                for (int j = 0; j < VW; ++j) T[(k + j) * LDT + row] = r[i * VW + j];
            } else {
                vec_t v;
#pragma unroll
                for (int j = 0; j < VW; ++j) v[j] = r[i * VW + j];
                *reinterpret_cast<vec_t *>(T + k * LDT + row) = v;
            }
        }
    }
};

// GUARD: waves test a 4-bit mask before each of their 2 x 2 MFMA blocks and skip the blocks that lie outside the
// matrix or under the block mask.  Chosen per launch (skm_gemm_launch): it pays when edge or diagonal tiles are a
// sizeable share of the grid (R = 400 leaves 16 valid columns in the last of its four 128-wide tiles: the aux
// product went from 7.1 to 4.7 ms) and costs ~6 % on grids of full tiles, which keep the unguarded loop.
template <typename TI, typename TO, bool ACC64, bool GUARD>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g)
{
    typedef Mf<TI> M;
    constexpr int TM = M::TM, KS = M::KS, NR = M::NR;
    constexpr int BM = 4 * TM, BN = 4 * TM;
    constexpr int BK = (sizeof(TI) == 4 && !ACC64) ? 32 : 16;
    constexpr int LDT = BM + 16 / sizeof(TI);  // rows stay 16-byte aligned
    constexpr int FOLD = 512 / BK;             // k-tiles between f64 folds (ACC64): every 512 k
    typedef TileIO<TI, BM, BK> IO;

    __shared__ __align__(16) TI As[2][BK * LDT];
    __shared__ __align__(16) TI Bs[2][BK * LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.swz) {
        // XCD-aware tile order (1-D launch): workgroups w and w + 8 share an XCD and its L2.  Groups of 8 "slow"
        // tile indices (one per XCD) x all "fast" tile indices: the workgroups that an XCD runs back to back
        // walk the fast dimension and re-use the slow dimension's operand tile from that XCD's L2.
        const int fast = g.swz == 1 ? g.swz_tiles_m : g.swz_tiles_n;
        const int grp = bx / (8 * fast), r = bx % (8 * fast);
        const int slow_idx = grp * 8 + (r & 7), fast_idx = r >> 3;
        by = g.swz == 1 ? fast_idx : slow_idx;
        bx = g.swz == 1 ? slow_idx : fast_idx;
    }
    const int m0 = by * BM, n0 = bx * BN;

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
    const bool akfast = !g.transA, bkfast = g.transB != 0;
    const bool avec = (g.vec & 1) != 0, bvec = (g.vec & 2) != 0;

    TI ra[IO::NV * IO::VW], rb[IO::NV * IO::VW];

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

    // k range of this tile: all of K unless a triangular operand makes part of it an exact zero (ktrim)
    int k_begin = 0, k_end = K;
    if (g.ktrim & 1) k_end = (n0 + BN < K) ? (n0 + BN) : K;
    if (g.ktrim & 2) k_begin = (m0 > n0) ? m0 : n0;
    if (g.ktrim & 4) k_begin = n0;
    if (k_begin > k_end) k_begin = k_end;
    const int kt0 = k_begin / BK, nk = (k_end + BK - 1) / BK;
    const int lm = lane % TM, lk = lane / TM;

    // Interior tiles (all rows and columns inside the matrix, 16-byte loads legal) take a loader without any
    // bounds test: one 32-bit element offset per vector, fixed for the whole k loop, on top of a wave-uniform
    // base pointer that advances by one k-tile per iteration.  Edge tiles and a ragged last k-tile take the
    // general loader.
    const bool interior = avec && bvec && m0 + BM <= Mr && n0 + BN <= N;
    unsigned offa[IO::NV], offb[IO::NV];
#pragma unroll
    for (int i = 0; i < IO::NV; ++i) {
        int row, k;
        IO::coord(tid, i, akfast, row, k);
        offa[i] = akfast ? (unsigned)((long)row * g.lda + k) : (unsigned)((long)k * g.lda + row);
        IO::coord(tid, i, bkfast, row, k);
        if (bkfast) offb[i] = (unsigned)((long)row * g.ldb + k);
        else offb[i] = (unsigned)((long)k * g.ldb) + (unsigned)((g.colmap && interior) ? g.colmap[n0 + row] - g.colmap[n0] : row);
    }
    const TI *At = akfast ? (A + (long)m0 * g.lda) : (A + m0);                                  // tile origin at k = 0
    const TI *Bt = bkfast ? (B + (long)n0 * g.ldb) : (B + ((g.colmap && interior) ? g.colmap[n0] : n0));
    const long inca = akfast ? 1 : g.lda, incb = bkfast ? 1 : g.ldb;                          // elements per unit of k
    auto load_tile = [&](int kt) {
        if (interior && (kt + 1) * BK <= K) {
            const TI *ap = At + (long)kt * BK * inca, *bp = Bt + (long)kt * BK * incb;
#pragma unroll
            for (int i = 0; i < IO::NV; ++i) {
                const typename IO::vec_t va = *reinterpret_cast<const typename IO::vec_t *>(ap + offa[i]);
                const typename IO::vec_t vb = *reinterpret_cast<const typename IO::vec_t *>(bp + offb[i]);
#pragma unroll
                for (int j = 0; j < IO::VW; ++j) {
                    ra[i * IO::VW + j] = va[j];
                    rb[i * IO::VW + j] = vb[j];
                }
            }
        } else {
            IO::load(ra, A, g.lda, akfast, avec, m0, kt * BK, Mr, K, tid);
            IO::load(rb, B, g.ldb, bkfast, bvec, n0, kt * BK, N, K, tid, g.colmap);
        }
    };
    // which of this wave's 2 x 2 MFMA blocks produce anything that the epilogue stores (wave-uniform)
    int need_bits = 15;
    if (GUARD) {
    need_bits = 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int bm0 = m0 + wm * 2 * TM + mi * TM, bn0 = n0 + wn * 2 * TM + ni * TM;
            bool nd = bm0 < Mr && bn0 < N;
            if (g.tile_mask == 1 && (bn0 + TM - 1) / g.mask_blk < bm0 / g.mask_blk) nd = false;
            need_bits |= nd ? (1 << (mi * 2 + ni)) : 0;
        }
    need_bits = __builtin_amdgcn_readfirstlane(need_bits);
    }
    load_tile(kt0);
    IO::store(ra, As[0], LDT, akfast, tid);
    IO::store(rb, Bs[0], LDT, bkfast, tid);
    __syncthreads();
    int since = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const TI *Ac = As[cur] + wm * 2 * TM + lm;
        const TI *Bc = Bs[cur] + wn * 2 * TM + lm;
        // a wave skips its TM x TM blocks that lie outside the matrix or under the block mask (wave-uniform bits:
        // R = 400 leaves 16 valid columns in its last 128-wide tile)
#pragma unroll
        for (int kk = 0; kk < BK / KS; ++kk) {
            TI a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = Ac[(kk * KS + lk) * LDT + mi * TM];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = Bc[(kk * KS + lk) * LDT + ni * TM];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    if (!GUARD || (need_bits & (1 << (mi * 2 + ni)))) acc[mi][ni] = M::mfma(a[mi], b[ni], acc[mi][ni]);
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
        if (kt + 1 < nk) {
            // the other buffer was last read before the previous barrier: one barrier per k-tile
            IO::store(ra, As[cur ^ 1], LDT, akfast, tid);
            IO::store(rb, Bs[cur ^ 1], LDT, bkfast, tid);
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
                    TO *p = C + (long)gm * g.ldc + (g.colmap ? g.colmap[gn] : gn);
                    if (g.beta != 0.0) v += g.beta * (double)(*p);
                    *p = (TO)v;
                }
            }
}

// 16-byte vector loads are legal for an operand when base, leading dimension, batch stride and the
// triangular-rows offsets all keep 16-byte alignment.
static int vec_ok(const void *p, long ld, long stride, long tri_ldz, int tri_on, size_t esz)
{
    const long vw = (long)(16 / esz);
    if (((uintptr_t)p & 15) || (ld % vw) || (stride % vw)) return 0;
    if (tri_on && (tri_ldz % vw)) return 0;
    return 1;
}

int skm_gemm_launch(SkmCtx *ctx, int kind, const GemmArgs &g, int batch, hipStream_t st, const char *tag)
{
    SKM_REQUIRE(g.A && g.B && g.C, "gemm(%s): NULL operand", tag);
    SKM_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0 && batch > 0, "gemm(%s): bad shape M=%d N=%d K=%d batch=%d", tag,
                g.M, g.N, g.K, batch);
    const int tile = (kind == SKM_GEMM_F32_F32 || kind == SKM_GEMM_F32_F64ACC || kind == SKM_GEMM_F32_F64OUT) ? 128 : 64;
    const int ncols = g.tri_n > 0 ? g.tri_n : g.N;
    dim3 grid((ncols + tile - 1) / tile, (g.M + tile - 1) / tile, batch);
    SKM_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm(%s): grid too large", tag);
    int swz_tm = 0, swz_tn = 0;
    if (g.swz) {
        SKM_REQUIRE(g.tri_n == 0 && (g.swz == 1 || g.swz == 2), "gemm(%s): bad swizzle", tag);
        swz_tm = (int)grid.y;
        swz_tn = (int)grid.x;
        const int fast = g.swz == 1 ? swz_tm : swz_tn, slow = g.swz == 1 ? swz_tn : swz_tm;
        grid = dim3((unsigned)((slow + 7) / 8 * 8 * fast), 1, batch);
    }
    SKM_REQUIRE(!g.colmap || (!g.transB && g.tile_mask == 0 && g.tri_n == 0), "gemm(%s): colmap needs plain B", tag);
    const size_t esz = (kind == SKM_GEMM_F64_F64 || kind == SKM_GEMM_F64_F32OUT) ? 8 : 4;
    GemmArgs gv = g;
    gv.swz_tiles_m = swz_tm;
    gv.swz_tiles_n = swz_tn;
    gv.vec = (vec_ok(g.A, g.lda, g.sA, 0, 0, esz) ? 1 : 0) |
             (vec_ok(g.B, g.ldb, g.sB, g.tri_ldz, g.tri_n > 0 && g.tri_blk % (16 / (int)esz) == 0, esz) ? 2 : 0);
    if (g.tri_n > 0 && g.tri_blk % (16 / (int)esz) != 0) gv.vec &= 1;
    if (g.colmap && !g.colmap_vec) gv.vec &= 1;
    // guard the MFMA blocks when edge tiles (at least one TM-wide block of the last tile row / column is outside the
    // matrix) or masked diagonal tiles are a sizeable share of the grid
    const int tmb = tile / 4;
    const int tiles_m = (g.M + tile - 1) / tile, tiles_n = (ncols + tile - 1) / tile;
    const bool edge_m = (tiles_m * tile - g.M) >= tmb, edge_n = (tiles_n * tile - ncols) >= tmb;
    const double share = (edge_m ? 1.0 / tiles_m : 0.0) + (edge_n ? 1.0 / tiles_n : 0.0) +
                         (g.tile_mask == 1 ? 1.0 / (tiles_m < tiles_n ? tiles_m : tiles_n) : 0.0);
    const bool guard = share >= 0.10 || g.tri_n > 0;
    SkmProfScope ps(ctx, tag, st);
#define SKM_GEMM_GO(TI_, TO_, ACC_)                                                                   \
    do {                                                                                               \
        if (guard) hipLaunchKernelGGL((k_gemm<TI_, TO_, ACC_, true>), grid, dim3(256), 0, st, gv);     \
        else hipLaunchKernelGGL((k_gemm<TI_, TO_, ACC_, false>), grid, dim3(256), 0, st, gv);          \
    } while (0)
    switch (kind) {
    case SKM_GEMM_F32_F32: SKM_GEMM_GO(float, float, false); break;
    case SKM_GEMM_F32_F64ACC: SKM_GEMM_GO(float, double, true); break;
    case SKM_GEMM_F32_F64OUT: SKM_GEMM_GO(float, double, false); break;
    case SKM_GEMM_F64_F64: SKM_GEMM_GO(double, double, false); break;
    case SKM_GEMM_F64_F32OUT: SKM_GEMM_GO(double, float, false); break;
    default: skm_set_error("gemm(%s): unknown kind %d", tag, kind); return SKM_ERR_ARG;
    }
#undef SKM_GEMM_GO
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
