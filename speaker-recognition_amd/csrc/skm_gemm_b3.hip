// skm_gemm_b3.hip -- float32 GEMM on the bf16 matrix cores by operand splitting ("bf16 x 3").
//
// The two products that dominate the TV E-step, L_s = sum_c N_sc G_c (factor_analyser.py:87) and
// A_c += sum_s N_sc e_hh(s) (factor_analyser.py:124), are plain float32 GEMMs that sat at 0.80 of the f32 MFMA peak
// (157 TF: v_mfma_f32_32x32x2_f32 IS the FP32 vector datapath).  gfx950's bf16 MFMA runs 16 times faster, and a float32
// number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand bits):
//     x = x1 + x2 + x3,   x1 = bf16(x),  x2 = bf16(x - x1),  x3 = bf16(x - x1 - x2)
// so  a b = a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1) + [a2 b3 + a3 b2 + a3 b3]: the bracket is below 2^-26 |a b|
// and is dropped; the six kept products are six v_mfma_f32_32x32x16_bf16 into the same float32 accumulator -- the
// accumulation is float32 as before, the products are as good as float32 products, at 6/16 of the f32 MFMA time.
//
// Operands are split ONCE, outside the GEMM (skm_b3_split: N and G are constants of an EM iteration, e_hh is written
// once per batch), into the layout the MFMA wants: planes[3][K/8][rows][8 bf16] -- the 8 consecutive k values a lane
// feeds to one MFMA are one 16-byte vector, rows are contiguous, K and rows are zero-padded to the tile so that the
// kernel has no bounds test on its loads.  The column map of the TV path (block-upper part of an R x R matrix as the
// dense N dimension) is applied by the split of B and by the epilogue's store.
//
// Kernel: workgroup = 8 waves = 256 x 256 tile (wave: 128 x 64 = 4 x 2 MFMA blocks, 128 accumulator registers), BK = 16
// (one MFMA k-step) per stage, two LDS stages of 48 KB, global -> registers -> LDS one stage ahead, one barrier per
// stage; one workgroup per CU, two waves per SIMD.  Per stage and wave: 18 ds_read_b128, 48 MFMAs.  The tile is this
// large because the operands cost 6 bytes per element and the loads come from L2: the first version (128 x 128 tiles)
// pulled 8 TB/s out of the L2s and stopped there, at 0.43 of the bf16 MFMA rate.
#include "skm_gemm.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define B3_TILE 256
#define B3_BK 16

__device__ __forceinline__ unsigned b3_bf16_rn(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;          // round to nearest even (finite inputs)
}

// three bf16 terms of x, each as the low 16 bits
__device__ __forceinline__ void b3_split(float x, unsigned &b1, unsigned &b2, unsigned &b3)
{
    b1 = b3_bf16_rn(x);
    const float r1 = x - __uint_as_float(b1 << 16);            // exact
    b2 = b3_bf16_rn(r1);
    const float r2 = r1 - __uint_as_float(b2 << 16);           // exact
    b3 = b3_bf16_rn(r2);
}

// dst[p][kg][row][j] = term p of element (row, k = 8 kg + j); zero beyond rows / K.
//   kfast : element (row, k) = src[row * ld + k]
//   else  : element (row, k) = src[k * ld + (rowmap ? rowmap[row] : row)]
__global__ __launch_bounds__(256) void k_b3_split(const float *__restrict__ src, long rows, long K, long ld, int kfast,
                                                  const int *__restrict__ rowmap, i32x4 *__restrict__ dst, long Rp, long KG)
{
    const long total = Rp * KG;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long kg = i / Rp, row = i - kg * Rp;
        unsigned t1[8], t2[8], t3[8];
        long off = 0;
        if (row < rows) off = kfast ? row * ld : (rowmap ? (long)rowmap[row] : row);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long k = kg * 8 + j;
            float x = 0.f;
            if (row < rows && k < K) x = kfast ? src[off + k] : src[k * ld + off];
            b3_split(x, t1[j], t2[j], t3[j]);
        }
        i32x4 v1, v2, v3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v1[q] = (int)(t1[2 * q] | (t1[2 * q + 1] << 16));
            v2[q] = (int)(t2[2 * q] | (t2[2 * q + 1] << 16));
            v3[q] = (int)(t3[2 * q] | (t3[2 * q + 1] << 16));
        }
        dst[i] = v1;
        dst[total + i] = v2;
        dst[2 * total + i] = v3;
    }
}

static inline long b3_pad(long v, long a) { return (v + a - 1) / a * a; }

size_t skm_b3_plane_bytes(long rows, long K)
{
    return (size_t)3 * (size_t)(b3_pad(K, B3_BK) / 8) * (size_t)b3_pad(rows, B3_TILE) * 16;
}

int skm_b3_split(SkmCtx *ctx, const float *src, long rows, long K, long ld, int kfast, const int *rowmap, void *dst,
                 hipStream_t st, const char *tag)
{
    SKM_REQUIRE(ctx && src && dst && rows > 0 && K > 0, "b3_split(%s): bad argument", tag);
    const long Rp = b3_pad(rows, B3_TILE), KG = b3_pad(K, B3_BK) / 8;
    long blocks = (Rp * KG + 255) / 256;
    if (blocks > (long)ctx->cu_count * 32) blocks = (long)ctx->cu_count * 32;
    SkmProfScope ps(ctx, tag, st);
    hipLaunchKernelGGL(k_b3_split, dim3((unsigned)blocks), dim3(256), 0, st, src, rows, K, ld, kfast, rowmap,
                       static_cast<i32x4 *>(dst), Rp, KG);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

struct B3Args {
    const i32x4 *A, *B;       // planes[3][KG][Mp or Np] of 16-byte k-groups
    long Mp, Np, KG;          // padded rows / columns, k-groups of 8 (KG * 8 = padded K, a multiple of B3_BK)
    long M, N;
    void *C;
    long ldc;
    const int *colmap;
    int accumulate;
    int tiles_m, tiles_n;
};

template <typename TO>
__global__ __launch_bounds__(512, 1) void k_gemm_b3(B3Args g)
{
    // [stage][operand A/B][plane][k-group of the stage][row]
    __shared__ i32x4 sh[2][2][3][B3_BK / 8][B3_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;      // wave tile: rows wm*128.., cols wn*64..
    // XCD-aware order: workgroups w, w + 8, ... share an XCD; one XCD walks the M tiles of ONE N tile column back to
    // back, so the (large) B tile of the column is fetched into that XCD's L2 once
    const int bx = blockIdx.x, fast = g.tiles_m;
    const int grp = bx / (8 * fast), rr = bx % (8 * fast);
    const int tn = grp * 8 + (rr & 7), tm = rr >> 3;
    if (tn >= g.tiles_n) return;
    const long m0 = (long)tm * B3_TILE, n0 = (long)tn * B3_TILE;

    // staging: plane p = i, k-group of the stage = tid >> 8, row = tid & 255
    const int skg = tid >> 8, srow = tid & 255;
    const long plA = g.KG * g.Mp, plB = g.KG * g.Np;
    const i32x4 *ga = g.A + (long)skg * g.Mp + m0 + srow;
    const i32x4 *gb = g.B + (long)skg * g.Np + n0 + srow;
    const long stepA = (long)(B3_BK / 8) * g.Mp, stepB = (long)(B3_BK / 8) * g.Np;
    const int nk = (int)(g.KG / (B3_BK / 8));

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

    i32x4 ra[3], rb[3];
#define B3_FETCH(S)                                                          \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                           \
        ra[p] = ga[p * plA + (long)(S) * stepA];                             \
        rb[p] = gb[p * plB + (long)(S) * stepB];                             \
    }
#define B3_PUT(ST)                                                           \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                           \
        sh[ST][0][p][skg][srow] = ra[p];                                     \
        sh[ST][1][p][skg][srow] = rb[p];                                     \
    }
#define B3_MM(PA, PB)                                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)         \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][mi], fb[PB][ni], acc[mi][ni], 0, 0, 0);
    B3_FETCH(0)
    B3_PUT(0)
    if (nk > 1) { B3_FETCH(1) }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        bf16x8 fa[3][4], fb[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int b = 0; b < 4; ++b) fa[p][b] = __builtin_bit_cast(bf16x8, sh[cur][0][p][h][wm * 128 + b * 32 + r]);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[p][b] = __builtin_bit_cast(bf16x8, sh[cur][1][p][h][wn * 64 + b * 32 + r]);
        }
        // six products per block, smallest terms first; consecutive MFMAs go to different accumulators
        B3_MM(2, 0) B3_MM(0, 2) B3_MM(1, 1) B3_MM(1, 0) B3_MM(0, 1) B3_MM(0, 0)
        // stage kt + 1 (requested one stage ago) goes to the other LDS buffer behind this stage's MFMAs
        if (kt + 1 < nk) { B3_PUT(cur ^ 1) }
        if (kt + 2 < nk) { B3_FETCH(kt + 2) }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#undef B3_MM
#undef B3_PUT
#undef B3_FETCH

    TO *C = static_cast<TO *>(g.C);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const long gn = n0 + wn * 64 + ni * 32 + r;
            if (gn >= g.N) continue;
            const long col = g.colmap ? (long)g.colmap[gn] : gn;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const long gm = m0 + wm * 128 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (gm < g.M) {
                    TO *p = C + gm * g.ldc + col;
                    const TO v = (TO)acc[mi][ni][q];
                    *p = g.accumulate ? (TO)(*p + v) : v;
                }
            }
        }
}

int skm_b3_gemm(SkmCtx *ctx, const void *Ap, long M, const void *Bp, long N, long K, void *C, int c_f64, long ldc,
                const int *colmap, int accumulate, hipStream_t st, const char *tag)
{
    SKM_REQUIRE(ctx && Ap && Bp && C && M > 0 && N > 0 && K > 0, "b3_gemm(%s): bad argument", tag);
    B3Args g;
    g.A = static_cast<const i32x4 *>(Ap);
    g.B = static_cast<const i32x4 *>(Bp);
    g.Mp = b3_pad(M, B3_TILE);
    g.Np = b3_pad(N, B3_TILE);
    g.KG = b3_pad(K, B3_BK) / 8;
    g.M = M;
    g.N = N;
    g.C = C;
    g.ldc = ldc;
    g.colmap = colmap;
    g.accumulate = accumulate;
    g.tiles_m = (int)(g.Mp / B3_TILE);
    g.tiles_n = (int)(g.Np / B3_TILE);
    const long grid = (long)((g.tiles_n + 7) / 8 * 8) * g.tiles_m;
    SKM_REQUIRE(grid < 0x7FFFFFFFL, "b3_gemm(%s): grid too large", tag);
    SkmProfScope ps(ctx, tag, st);
    if (c_f64)
        hipLaunchKernelGGL(k_gemm_b3<double>, dim3((unsigned)grid), dim3(512), 0, st, g);
    else
        hipLaunchKernelGGL(k_gemm_b3<float>, dim3((unsigned)grid), dim3(512), 0, st, g);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// C ABI (include/skm.h): the split product of two row-major float32 matrices, C = A B (+ C), for tests and timing.
extern "C" size_t skm_gemm_split_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return skm_align_up(skm_b3_plane_bytes(M, K), 256) + skm_align_up(skm_b3_plane_bytes(N, K), 256);
}

extern "C" int skm_gemm_split(void *ctx_, const float *A, const float *B, void *C, int M, int N, int K, int c_f64,
                              int accumulate, void *ws, size_t ws_bytes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && A && B && C && ws, "skm_gemm_split: NULL argument");
    SKM_REQUIRE(M > 0 && N > 0 && K > 0, "skm_gemm_split: bad shape");
    if (ws_bytes < skm_gemm_split_workspace_bytes(M, N, K)) {
        skm_set_error("skm_gemm_split: workspace too small");
        return SKM_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *w = static_cast<char *>(ws);
    void *Ap = w, *Bp = w + skm_align_up(skm_b3_plane_bytes(M, K), 256);
    int rc;
    if ((rc = skm_b3_split(ctx, A, M, K, K, 1, nullptr, Ap, st, "k_b3_split"))) return rc;
    if ((rc = skm_b3_split(ctx, B, N, K, N, 0, nullptr, Bp, st, "k_b3_split"))) return rc;
    return skm_b3_gemm(ctx, Ap, M, Bp, N, K, C, c_f64, N, nullptr, accumulate, st, "k_gemm_b3");
}
