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
// (one MFMA k-step) per stage, a ring of three LDS stages of 48 KB filled by LDS-direct loads two stages ahead of the
// one being multiplied (round 3: the first version staged through registers ONE stage ahead and waited for its loads
// at the end of every stage); one workgroup per CU, two waves per SIMD, and the two waves of a SIMD run one barrier
// apart -- one reads its fragments while the other multiplies (two barriers per stage, see the loop).  Per stage and
// wave: 6 buffer_load ... lds, 18 ds_read_b128, 48 MFMAs.  The tile is this large because the operands cost 6 bytes
// per element and the loads come from L2: the first version (128 x 128 tiles) pulled 8 TB/s out of the L2s and stopped
// there, at 0.43 of the bf16 MFMA rate.  Where it stands: the matrix pipe busy 0.83 of the cycles at the 1.6 GHz the
// power management leaves under this load (tools/ubench/mfma_bf16_peak.hip).  A launch can hold several
// independent products (grid y: split-K slabs) and a product can run over a k range of larger planes (plane strides).
#include "skm_gemm.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define B3_TILE 256
#define B3_BK 16
#define B3_STAGES 3
// the wait that retires a stage's LDS-direct loads (tests/test_isa_lint.py builds this file without it: lint rule iii)
#ifdef SKM_LINT_DEMO_B3_NO_DMA_WAIT
#define B3_WAIT_LOADS ""
#else
#define B3_WAIT_LOADS "s_waitcnt vmcnt(6)\n\t"
#endif

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
    // thread -> (row, k-group).  A k-fast source is walked so that four neighbouring threads read the 128 contiguous bytes
    // of one row (k-groups 4 q .. 4 q + 3) and threads four apart take consecutive rows: whole cache lines on the way
    // in, 256-byte runs per k-group on the way out (one thread per row and k-group with the rows fastest used 32 bytes
    // of every line it fetched: 4 x the reads).  A row-fast source has the rows fastest on both sides.
    const long total = Rp * KG, KQ = (KG + 3) / 4, walk = kfast ? Rp * KQ * 4 : total;
    const bool vec = kfast && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0;       // 16-byte loads along k
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < walk; w += (long)gridDim.x * blockDim.x) {
        long kg, row;
        if (kfast) {
            const long q = w / (Rp * 4), rem = w - q * (Rp * 4);
            row = rem >> 2;
            kg = q * 4 + (rem & 3);
            if (kg >= KG) continue;
        } else {
            kg = w / Rp;
            row = w - kg * Rp;
        }
        const long i = kg * Rp + row;
        unsigned t1[8], t2[8], t3[8];
        long off = 0;
        if (row < rows) off = kfast ? row * ld : (rowmap ? (long)rowmap[row] : row);
        float x[8];
        if (vec && row < rows && kg * 8 + 8 <= K) {
            const float4 lo = *reinterpret_cast<const float4 *>(src + off + kg * 8);
            const float4 hi = *reinterpret_cast<const float4 *>(src + off + kg * 8 + 4);
            x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w;
            x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long k = kg * 8 + j;
                x[j] = 0.f;
                if (row < rows && k < K) x[j] = kfast ? src[off + k] : src[k * ld + off];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) b3_split(x[j], t1[j], t2[j], t3[j]);
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

long skm_b3_rows_padded(long rows) { return b3_pad(rows, B3_TILE); }
long skm_b3_kgroups(long K) { return b3_pad(K, B3_BK) / 8; }

size_t skm_b3_plane_bytes(long rows, long K)
{
    return (size_t)3 * (size_t)(b3_pad(K, B3_BK) / 8) * (size_t)b3_pad(rows, B3_TILE) * 16;
}

int skm_b3_split(SkmCtx *ctx, const float *src, long rows, long K, long ld, int kfast, const int *rowmap, void *dst,
                 hipStream_t st, const char *tag)
{
    SKM_REQUIRE(ctx && src && dst && rows > 0 && K > 0, "b3_split(%s): bad argument", tag);
    const long Rp = b3_pad(rows, B3_TILE), KG = b3_pad(K, B3_BK) / 8;
    long blocks = (Rp * ((KG + 3) / 4 * 4) + 255) / 256;
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
    long plA, plB;            // distance between the planes of an operand (16-byte units): KG * Mp / KG * Np unless the
                              // product runs over a k range of larger planes
    long M, N;
    void *C;
    long ldc;
    const int *colmap;
    int accumulate;
    int tiles_m, tiles_n;
    int swap;                 // XCD-aware tile order: 0 = an XCD walks the M tiles of one N tile, 1 = the N tiles of one M tile
    long zA, zB, zC;          // blockIdx.y = z: independent products of one launch (split-K slabs): A, B advance by
                              // zA, zB 16-byte units (k ranges of the same planes), C by zC elements
};

template <typename TO>
__global__ __launch_bounds__(512, 1) void k_gemm_b3(B3Args g)
{
    // [stage][operand A/B][plane][k-group of the stage][row]: a ring of three 48 KB stages
    __shared__ i32x4 sh[B3_STAGES][2][3][B3_BK / 8][B3_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;      // wave tile: rows wm*128.., cols wn*64..
    // XCD-aware order: workgroups w, w + 8, ... share an XCD; one XCD walks the M tiles of ONE N tile column back to
    // back, so the (large) B tile of the column is fetched into that XCD's L2 once
    // (with fewer N tiles than M tiles -- the aux product of the TV path has two -- the roles are swapped: an XCD walks
    // the N tiles of one M tile)
    const int bx = blockIdx.x, fast = g.swap ? g.tiles_n : g.tiles_m;
    const int grp = bx / (8 * fast), rr = bx % (8 * fast);
    const int slow_i = grp * 8 + (rr & 7), fast_i = rr >> 3;
    const int tn = g.swap ? fast_i : slow_i, tm = g.swap ? slow_i : fast_i;
    if (tn >= g.tiles_n || tm >= g.tiles_m) return;
    const long m0 = (long)tm * B3_TILE, n0 = (long)tn * B3_TILE;
    const long z = blockIdx.y;
    g.A += z * g.zA;
    g.B += z * g.zB;

    // Staging: the stages go from L2 straight into LDS (buffer_load ... lds, 16 bytes per lane: one instruction of a
    // wave fills 64 consecutive rows of one k-group of one plane = 1 KB of the stage), two stages ahead of the one
    // being multiplied: no staging registers, no LDS stores, and a load has two stages' worth of MFMAs (~2.5 us) to
    // arrive.  Wave w stages k-group w >> 2, rows 64 (w & 3) .. + 63 of every plane of both operands: 6 loads per stage.
    // One descriptor per plane, based at this tile's first row; the stage advances in the scalar offset.
    const int skg = wave >> 2, srow0 = (wave & 3) * 64;
    const long plA = g.plA, plB = g.plB;
    const long extA = g.KG * g.Mp, extB = g.KG * g.Np;        // what this product reads of a plane
    __amdgpu_buffer_rsrc_t rsa[3], rsb[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        rsa[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<i32x4 *>(g.A + p * plA + m0), 0,
                                                   (int)(unsigned)((extA - m0) * 16), 0x00027000);
        rsb[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<i32x4 *>(g.B + p * plB + n0), 0,
                                                   (int)(unsigned)((extB - n0) * 16), 0x00027000);
    }
    const int voa = (int)(((long)skg * g.Mp + srow0 + lane) * 16), vob = (int)(((long)skg * g.Np + srow0 + lane) * 16);
    const int stepA = (int)((long)(B3_BK / 8) * g.Mp * 16), stepB = (int)((long)(B3_BK / 8) * g.Np * 16);   // bytes per stage
    const int nk = (int)(g.KG / (B3_BK / 8));
    typedef __attribute__((address_space(3))) void *lds_ptr;
#define B3_ISSUE(S, ST)                                                                                              \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa[p], (lds_ptr)&sh[ST][0][p][skg][srow0], 16, voa, (S) * stepA, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb[p], (lds_ptr)&sh[ST][1][p][skg][srow0], 16, vob, (S) * stepB, 0, 0); \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

#define B3_MM(PA, PB)                                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)         \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][mi], fb[PB][ni], acc[mi][ni], 0, 0, 0);
    // Two wave groups (waves 0-3 and 4-7: one wave of each per SIMD) run the same program ONE BARRIER APART.  A stage
    // is two phases, each closed by a barrier: R = issue the loads of stage kt + 2, read the 18 fragments of stage kt;
    // M = the 48 MFMAs.  While one group multiplies, the other reads: the LDS reads (and their latency behind the
    // barrier) of one wave sit under the MFMAs of the other wave of the SIMD, instead of both waves reading, then both
    // multiplying.  Global phase 2 kt: group 0 in R(kt), group 1 in M(kt - 1); phase 2 kt + 1: group 0 in M(kt),
    // group 1 in R(kt).
    //   visibility of stage kt + 1 (first read: group 0's R(kt + 1), global phase 2 kt + 2): every wave waits for its own
    //   pieces in front of the barrier that closes global phase 2 kt + 1 -- group 0 at the end of M(kt), group 1 at the
    //   end of R(kt); the six loads of stage kt + 2, issued in R(kt), stay in flight across it.
    //   reuse of a slot: stage kt + 2 goes where stage kt - 1 was; its last readers (group 1, R(kt - 1), global phase
    //   2 kt - 1) retire their reads (lgkmcnt(0)) in front of the barrier that closes that phase, and the earliest
    //   issue of the loads is group 0's R(kt), global phase 2 kt.
    // Every iteration issues exactly six loads (behind the last stage the last stage again, into the free slot), so the
    // wait is the same vmcnt(6) everywhere; the loop is written once per group, without a branch on the group inside.
    const int wg = wave >> 2;
    B3_ISSUE(0, 0)
    B3_ISSUE(nk > 1 ? 1 : 0, 1)
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
#define B3_READ_FRAGMENTS()                                                                                          \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                   \
        _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                                 \
            fa[p][b] = __builtin_bit_cast(bf16x8, sh[cur][0][p][h][wm * 128 + b * 32 + r]);                           \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
            fb[p][b] = __builtin_bit_cast(bf16x8, sh[cur][1][p][h][wn * 64 + b * 32 + r]);                            \
    }
    // six products per block, smallest terms first; consecutive MFMAs go to different accumulators
#define B3_PRODUCTS() B3_MM(2, 0) B3_MM(0, 2) B3_MM(1, 1) B3_MM(1, 0) B3_MM(0, 1) B3_MM(0, 0)
#define B3_ADVANCE()                                                                                                 \
    cur = (cur == B3_STAGES - 1) ? 0 : cur + 1;                                                                      \
    nxt = (nxt == B3_STAGES - 1) ? 0 : nxt + 1;
    int cur = 0, nxt = 2;       // ring slots of stage kt and of stage kt + 2
    bf16x8 fa[3][4], fb[3][2];
    if (wg == 0) {
        for (int kt = 0; kt < nk; ++kt) {
            B3_ISSUE(kt + 2 < nk ? kt + 2 : nk - 1, nxt)                                    // phase R
            B3_READ_FRAGMENTS()
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            B3_PRODUCTS()                                                                   // phase M
            __builtin_amdgcn_sched_barrier(0);
            asm volatile(B3_WAIT_LOADS "s_barrier" ::: "memory");                           // stage kt + 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            B3_ADVANCE()
        }
    } else {
        asm volatile("s_barrier" ::: "memory");                                             // one phase behind
        for (int kt = 0; kt < nk; ++kt) {
            B3_ISSUE(kt + 2 < nk ? kt + 2 : nk - 1, nxt)                                    // phase R
            B3_READ_FRAGMENTS()
            __builtin_amdgcn_sched_barrier(0);
            asm volatile(B3_WAIT_LOADS "s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage kt + 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            B3_PRODUCTS()                                                                   // phase M
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) asm volatile("s_barrier" ::: "memory");                        // the last phase has no partner
            __builtin_amdgcn_sched_barrier(0);
            B3_ADVANCE()
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued last stage: nothing lands in LDS behind the exit
#undef B3_ADVANCE
#undef B3_PRODUCTS
#undef B3_READ_FRAGMENTS
#undef B3_MM
#undef B3_ISSUE

    // Epilogue.  Accumulating onto C reads the 16 old values of a 32 x 32 block with 16 loads in flight (rows and
    // columns beyond the matrix are clamped to its last ones for the load and masked for the store): with the load
    // inside the bounds test hipcc waited for every single one (128 round trips to HBM per wave, a quarter of the A
    // accumulation's time).
    TO *C = static_cast<TO *>(g.C) + z * g.zC;
    long colv[2];
    bool okn[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const long gn = n0 + wn * 64 + ni * 32 + r;
        okn[ni] = gn < g.N;
        const long gc = okn[ni] ? gn : g.N - 1;
        colv[ni] = g.colmap ? (long)g.colmap[gc] : gc;
    }
    if (g.accumulate) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                TO old[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long gm = m0 + wm * 128 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    old[q] = C[(gm < g.M ? gm : g.M - 1) * g.ldc + colv[ni]];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long gm = m0 + wm * 128 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (okn[ni] && gm < g.M) C[gm * g.ldc + colv[ni]] = (TO)(old[q] + (TO)acc[mi][ni][q]);
                }
            }
    } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long gm = m0 + wm * 128 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (okn[ni] && gm < g.M) C[gm * g.ldc + colv[ni]] = (TO)acc[mi][ni][q];
                }
    }
}

int skm_b3_gemm(SkmCtx *ctx, const void *Ap, long M, const void *Bp, long N, long K, void *C, int c_f64, long ldc,
                const int *colmap, int accumulate, hipStream_t st, const char *tag, long plane_a, long plane_b, int nz,
                long zA, long zB, long zC)
{
    SKM_REQUIRE(ctx && Ap && Bp && C && M > 0 && N > 0 && K > 0, "b3_gemm(%s): bad argument", tag);
    B3Args g;
    g.A = static_cast<const i32x4 *>(Ap);
    g.B = static_cast<const i32x4 *>(Bp);
    g.Mp = b3_pad(M, B3_TILE);
    g.Np = b3_pad(N, B3_TILE);
    g.KG = b3_pad(K, B3_BK) / 8;
    g.plA = plane_a ? plane_a : g.KG * g.Mp;
    g.plB = plane_b ? plane_b : g.KG * g.Np;
    g.M = M;
    g.N = N;
    g.C = C;
    g.ldc = ldc;
    g.colmap = colmap;
    g.accumulate = accumulate;
    g.tiles_m = (int)(g.Mp / B3_TILE);
    g.tiles_n = (int)(g.Np / B3_TILE);
    SKM_REQUIRE(nz >= 1 && nz <= 65535, "b3_gemm(%s): bad slab count %d", tag, nz);
    g.zA = zA;
    g.zB = zB;
    g.zC = zC;
    g.swap = g.tiles_n < g.tiles_m;
    const long grid = g.swap ? (long)((g.tiles_m + 7) / 8 * 8) * g.tiles_n : (long)((g.tiles_n + 7) / 8 * 8) * g.tiles_m;
    SKM_REQUIRE(grid < 0x7FFFFFFFL, "b3_gemm(%s): grid too large", tag);
    // one buffer descriptor per plane: 32-bit extents and offsets
    SKM_REQUIRE(g.KG * g.Mp * 16 < 0xFFFFFFFFL && g.KG * g.Np * 16 < 0xFFFFFFFFL,
                "b3_gemm(%s): a plane of an operand exceeds 4 GB (M=%ld N=%ld K=%ld)", tag, M, N, K);
    SkmProfScope ps(ctx, tag, st);
    if (c_f64)
        hipLaunchKernelGGL(k_gemm_b3<double>, dim3((unsigned)grid, (unsigned)nz), dim3(512), 0, st, g);
    else
        hipLaunchKernelGGL(k_gemm_b3<float>, dim3((unsigned)grid, (unsigned)nz), dim3(512), 0, st, g);
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
