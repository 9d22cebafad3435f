// skm_gemm.hip -- generic strided-batched MFMA GEMM (see skm_gemm.h).
#include "skm_gemm.h"
#include "skm_gemm_dev.h"

// GUARD: waves test a 4-bit mask before each of their 2 x 2 MFMA blocks and skip the blocks that lie outside the
// matrix or under the block mask.  Chosen per launch (skm_gemm_launch): it pays when edge or diagonal tiles are a
// sizeable share of the grid (R = 400 leaves 16 valid columns in the last of its four 128-wide tiles: the aux
// product went from 7.1 to 4.7 ms) and costs ~6 % on grids of full tiles, which keep the unguarded loop.
// AKF / BKF: the stored A / B is contiguous along k (A not transposed / B transposed).
//
// Wave specialisation: a workgroup is 8 waves, two per SIMD.  Waves 0-3 ("compute") only read fragments from LDS and
// issue MFMAs; waves 4-7 ("staging") only move the next k-tiles global -> registers -> LDS, so the loads, the LDS
// stores and the waits for both never sit in an MFMA wave's instruction stream.  (What a staging wave may NOT do is
// vector arithmetic: the f32 MFMA shares the VALU datapath, and every vector-ALU instruction on a SIMD takes its
// cycles from that SIMD's MFMA stream whichever wave issues it -- see the staging loop.)  One raw s_barrier per k-tile
// joins the two roles; the staging waves wait only for their LDS stores in front of it (never vmcnt(0): the loads of
// later tiles stay in flight across it).
template <typename TI, typename TO, bool ACC64, bool GUARD, bool AKF, bool BKF, bool SM = false>
__global__ __launch_bounds__(512, (sizeof(TI) == 4 && ACC64) ? 1 : 2) void k_gemm(GemmArgs g)
{
    typedef typename MfSel<TI, SM>::type M;     // SM: float32 on 16 x 16 x 4 MFMAs, 64 x 64 tiles
    constexpr int TM = M::TM, KS = M::KS, NR = M::NR;
    constexpr int BM = 4 * TM, BN = 4 * TM;
    constexpr int BK = (sizeof(TI) == 4 && !ACC64) ? 32 : 16;
    constexpr int FOLD = 512 / BK;             // k-tiles between f64 folds (ACC64): every 512 k
    typedef TileIO<TI, BM, BK, AKF> IOA;
    typedef TileIO<TI, BN, BK, BKF> IOB;
    constexpr int VW = IOA::VW, NLK = 64 / TM, NSTEP = BK / KS, NGRP = NSTEP / VW;
    static_assert(KS == NLK && NSTEP % VW == 0, "k map");
    typedef typename IOA::vec_t vec_t;

    __shared__ __align__(16) TI As[2][IOA::LDS_ELEMS];
    __shared__ __align__(16) TI Bs[2][IOB::LDS_ELEMS];

    const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool staging = wave8 >= 4;
    const int wave = wave8 & 3, tid = threadIdx.x & 255;     // index inside the role
    const int wm = wave >> 1, wn = wave & 1;
    int z = blockIdx.z;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.swz == 3) {
        // batched launch, 1-D: workgroups w, w + 8, w + 16, ... run on one XCD, so they walk the tiles of ONE matrix
        // of the batch before the next (XCD x owns matrices x, x + 8, ...): the operand panels that the tiles of a
        // small matrix share are fetched into one L2 once instead of into all eight
        const int tiles = g.swz_tiles_m * g.swz_tiles_n, slot = bx >> 3;
        z = (slot / tiles) * 8 + (bx & 7);
        if (z >= g.swz_batch) return;
        const int t = slot % tiles;
        by = t / g.swz_tiles_n;
        bx = t % g.swz_tiles_n;
    } else if (g.swz) {
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
    const bool avec = (g.vec & 1) != 0, bvec = (g.vec & 2) != 0;

    // k range of this tile: all of K unless a triangular operand makes part of it an exact zero (ktrim)
    int k_begin = 0, k_end = K;
    if (g.ktrim & 1) k_end = (n0 + BN < K) ? (n0 + BN) : K;
    if (g.ktrim & 2) k_begin = (m0 > n0) ? m0 : n0;
    if (g.ktrim & 4) k_begin = n0;
    if (k_begin > k_end) k_begin = k_end;
    const int kt0 = k_begin / BK, nk = (k_end + BK - 1) / BK;
    const bool empty = kt0 >= nk;        // nothing to contract: C = beta C, no tile is touched
    const int lm = lane % TM, lk = lane / TM;

    if (staging) {
        if (empty) return;
        // Two tiles are in flight in registers (sets S0 / S1) while tile kt is computed.  Step kt: write tile kt + 1
        // to the other LDS buffer (whose readers passed the barrier that ended step kt - 1), re-issue that register
        // set with tile kt + 3, barrier.
        //
        // What this loop may cost is set by two measurements.  (1) EVERY vector-ALU instruction issued on a SIMD,
        // by whichever wave, takes its cycles from the f32 MFMA stream of that SIMD (same datapath): with per-load
        // 64-bit address arithmetic and per-element selects in here the kernel ran at 0.70 of the MFMA peak, with
        // the loads removed at 0.91.  So the loop has NO vector arithmetic: buffer loads with a per-lane byte offset
        // that is fixed for the whole k loop, a buffer descriptor rebuilt per tile by the scalar unit (base advanced,
        // extent shrunk), and the hardware's range check as the bounds test -- what lies outside the matrix reads as
        // zero (rows beyond the edge get an offset beyond any extent; k beyond K falls behind the shrinking extent of a
        // row-fast operand).  (2) hipcc copies the results of loads that sit on two sides of a branch at the join and
        // waits vmcnt(0) there, right behind their issue: no control flow between a load and its use.
        // The general loader (vectors that straddle an edge, ragged K of a k-fast operand, extents beyond 32 bits) has
        // a loop of its own below.
        const size_t esz = sizeof(TI);
        const size_t ext_a = AKF ? ((size_t)((Mr - m0 < BM ? Mr - m0 : BM) - 1) * g.lda + K) * esz
                                 : ((size_t)(K - 1) * g.lda + (Mr - m0)) * esz;
        const size_t ext_b = BKF ? ((size_t)((N - n0 < BN ? N - n0 : BN) - 1) * g.ldb + K) * esz
                                 : (g.colmap ? (size_t)K * g.ldb * esz : ((size_t)(K - 1) * g.ldb + (N - n0)) * esz);
        const bool bpath = avec && bvec && ext_a < 0xE0000000ull && ext_b < 0xE0000000ull &&
                           (AKF ? (K % BK == 0) : ((Mr - m0) % VW == 0 || m0 + BM <= Mr)) &&
                           (BKF ? (K % BK == 0) : ((N - n0) % VW == 0 || n0 + BN <= N));
        if (bpath) {
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            vec_t ra[2][IOA::NV], rb[2][IOB::NV];
            int va[IOA::NV], vb[IOB::NV];            // byte offset of the vector from the tile origin at k-tile 0
            const TI *A0 = AKF ? (A + (long)m0 * g.lda) : (A + m0);
            const TI *B0 = BKF ? (B + (long)n0 * g.ldb) : (g.colmap ? B : B + n0);
#pragma unroll
            for (int i = 0; i < IOA::NV; ++i) {
                int row, k;
                IOA::coord(tid, i, row, k);
                const long off = AKF ? ((long)row * g.lda + k) : ((long)k * g.lda + row);
                va[i] = (m0 + row < Mr) ? (int)(unsigned)(off * (long)esz) : (int)0xFFFFFFF0u;
            }
#pragma unroll
            for (int i = 0; i < IOB::NV; ++i) {
                int row, k;
                IOB::coord(tid, i, row, k);
                const bool ok = n0 + row < N;
                const long off = BKF ? ((long)row * g.ldb + k)
                                     : ((long)k * g.ldb + (g.colmap ? (long)g.colmap[ok ? n0 + row : 0] : (long)row));
                vb[i] = ok ? (int)(unsigned)(off * (long)esz) : (int)0xFFFFFFF0u;
            }
            const long stepa = (long)BK * (AKF ? 1 : g.lda), stepb = (long)BK * (BKF ? 1 : g.ldb);   // elements per k-tile
            auto fetch = [&](int kt, vec_t (&qa)[IOA::NV], vec_t (&qb)[IOB::NV]) {
                const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TI *>(A0 + kt * stepa), 0, (int)(unsigned)(ext_a - (size_t)(kt * stepa) * esz), 0x00027000);
                const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<TI *>(B0 + kt * stepb), 0, (int)(unsigned)(ext_b - (size_t)(kt * stepb) * esz), 0x00027000);
#pragma unroll
                for (int i = 0; i < IOA::NV; ++i)
                    qa[i] = __builtin_bit_cast(vec_t, (i32x4)__builtin_amdgcn_raw_buffer_load_b128(rsa, va[i], 0, 0));
#pragma unroll
                for (int i = 0; i < IOB::NV; ++i)
                    qb[i] = __builtin_bit_cast(vec_t, (i32x4)__builtin_amdgcn_raw_buffer_load_b128(rsb, vb[i], 0, 0));
            };
            auto put = [&](const vec_t (&qa)[IOA::NV], const vec_t (&qb)[IOB::NV], TI *Ad, TI *Bd) {
#pragma unroll
                for (int i = 0; i < IOA::NV; ++i) {
                    int row, k;
                    IOA::coord(tid, i, row, k);
                    *reinterpret_cast<vec_t *>(AKF ? (Ad + row * IOA::LDK + k) : (Ad + k * IOA::LDT + row)) = qa[i];
                }
#pragma unroll
                for (int i = 0; i < IOB::NV; ++i) {
                    int row, k;
                    IOB::coord(tid, i, row, k);
                    *reinterpret_cast<vec_t *>(BKF ? (Bd + row * IOB::LDK + k) : (Bd + k * IOB::LDT + row)) = qb[i];
                }
            };
            fetch(kt0, ra[0], rb[0]);
            put(ra[0], rb[0], As[0], Bs[0]);
            if (kt0 + 1 < nk) fetch(kt0 + 1, ra[0], rb[0]);
            if (kt0 + 2 < nk) fetch(kt0 + 2, ra[1], rb[1]);
            gemm_barrier();
            for (int kt = kt0; kt + 1 < nk; kt += 2) {
                put(ra[0], rb[0], As[1], Bs[1]);             // tile kt + 1
                if (kt + 3 < nk) fetch(kt + 3, ra[0], rb[0]);
                gemm_barrier();
                if (kt + 2 >= nk) break;
                put(ra[1], rb[1], As[0], Bs[0]);             // tile kt + 2
                if (kt + 4 < nk) fetch(kt + 4, ra[1], rb[1]);
                gemm_barrier();
            }
            return;
        }
        // general loader: one tile in flight, bounds tests per element
        TI ra[IOA::NV * VW], rb[IOB::NV * VW];
        IOA::load(ra, A, g.lda, avec, m0, kt0 * BK, Mr, K, tid);
        IOB::load(rb, B, g.ldb, bvec, n0, kt0 * BK, N, K, tid, g.colmap);
        IOA::store(ra, As[0], tid);
        IOB::store(rb, Bs[0], tid);
        gemm_barrier();
        for (int kt = kt0; kt + 1 < nk; ++kt) {
            const int cur = (kt - kt0) & 1;
            IOA::load(ra, A, g.lda, avec, m0, (kt + 1) * BK, Mr, K, tid);
            IOB::load(rb, B, g.ldb, bvec, n0, (kt + 1) * BK, N, K, tid, g.colmap);
            IOA::store(ra, As[cur ^ 1], tid);
            IOB::store(rb, Bs[cur ^ 1], tid);
            gemm_barrier();
        }
        return;
    }

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

    // Fragments of one group of VW MFMA steps: fa[mi][j], fb[ni][j] feed step q*VW + j.  A k-fast operand gets them
    // with one 16-byte LDS read per block, a row-fast one with VW scalar reads.
    auto read_group = [&](const TI *Ab, const TI *Bb, int q, TI (&fa)[2][VW], TI (&fb)[2][VW]) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (AKF) {
                const vec_t v = *reinterpret_cast<const vec_t *>(Ab + (wm * 2 * TM + mi * TM + lm) * IOA::LDK + q * (VW * NLK) + lk * VW);
#pragma unroll
                for (int j = 0; j < VW; ++j) fa[mi][j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < VW; ++j) fa[mi][j] = Ab[(q * (VW * NLK) + lk * VW + j) * IOA::LDT + wm * 2 * TM + mi * TM + lm];
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if (BKF) {
                const vec_t v = *reinterpret_cast<const vec_t *>(Bb + (wn * 2 * TM + ni * TM + lm) * IOB::LDK + q * (VW * NLK) + lk * VW);
#pragma unroll
                for (int j = 0; j < VW; ++j) fb[ni][j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < VW; ++j) fb[ni][j] = Bb[(q * (VW * NLK) + lk * VW + j) * IOB::LDT + wn * 2 * TM + ni * TM + lm];
            }
        }
    };

    // compute waves: inside a k-tile the LDS reads of fragment group q + 1 are issued in front of the MFMAs of group q
    if (!empty) gemm_barrier();
    int since = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        TI fa[2][2][VW], fb[2][2][VW];
        read_group(As[cur], Bs[cur], 0, fa[0], fb[0]);
#pragma unroll
        for (int q = 0; q < NGRP; ++q) {
            if (q + 1 < NGRP) read_group(As[cur], Bs[cur], q + 1, fa[(q + 1) & 1], fb[(q + 1) & 1]);
#pragma unroll
            for (int j = 0; j < VW; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (!GUARD || (need_bits & (1 << (mi * 2 + ni))))
                            acc[mi][ni] = M::mfma(fa[q & 1][mi][j], fb[q & 1][ni][j], acc[mi][ni]);
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
        if (kt + 1 < nk) gemm_barrier();   // the reads of this buffer end here; the staging waves overwrite it next
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
                    if (g.r1) v += g.r1[(long)z * g.r1_stride + gm] * g.r1[(long)z * g.r1_stride + gn];
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
    SKM_REQUIRE((kind != SKM_GEMM_F32S_F32 && kind != SKM_GEMM_F32S_F64OUT) || !g.colmap,
                "gemm(%s): the small-tile float32 kernel has no column map", tag);
    const int ncols = g.tri_n > 0 ? g.tri_n : g.N;
    dim3 grid((ncols + tile - 1) / tile, (g.M + tile - 1) / tile, batch);
    SKM_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm(%s): grid too large", tag);
    int swz_tm = 0, swz_tn = 0, swz_mode = g.swz;
    if (!g.swz && batch >= 16 && g.tri_n == 0) {
        swz_mode = 3;
        swz_tm = (int)grid.y;
        swz_tn = (int)grid.x;
        grid = dim3((unsigned)(swz_tm * swz_tn) * (unsigned)((batch + 7) / 8 * 8), 1, 1);
    } else if (g.swz) {
        SKM_REQUIRE(g.tri_n == 0 && (g.swz == 1 || g.swz == 2), "gemm(%s): bad swizzle", tag);
        swz_tm = (int)grid.y;
        swz_tn = (int)grid.x;
        const int fast = g.swz == 1 ? swz_tm : swz_tn, slow = g.swz == 1 ? swz_tn : swz_tm;
        grid = dim3((unsigned)((slow + 7) / 8 * 8 * fast), 1, batch);
    }
    SKM_REQUIRE(!g.colmap || (!g.transB && g.tile_mask == 0 && g.tri_n == 0), "gemm(%s): colmap needs plain B", tag);
    const size_t esz = (kind == SKM_GEMM_F64_F64 || kind == SKM_GEMM_F64_F32OUT) ? 8 : 4;
    GemmArgs gv = g;
    gv.swz = swz_mode;
    gv.swz_batch = batch;
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
#define SKM_GEMM_L(TI_, TO_, ACC_, G_)                                                                          \
    do {                                                                                                             \
        if (!g.transA && !g.transB) hipLaunchKernelGGL((k_gemm<TI_, TO_, ACC_, G_, true, false>), grid, dim3(512), 0, st, gv);      \
        else if (g.transA && !g.transB) hipLaunchKernelGGL((k_gemm<TI_, TO_, ACC_, G_, false, false>), grid, dim3(512), 0, st, gv); \
        else if (!g.transA && g.transB) hipLaunchKernelGGL((k_gemm<TI_, TO_, ACC_, G_, true, true>), grid, dim3(512), 0, st, gv);   \
        else hipLaunchKernelGGL((k_gemm<TI_, TO_, ACC_, G_, false, true>), grid, dim3(512), 0, st, gv);              \
    } while (0)
#define SKM_GEMM_LS(TO_, G_)                                                                                        \
    do {                                                                                                             \
        if (!g.transA && !g.transB) hipLaunchKernelGGL((k_gemm<float, TO_, false, G_, true, false, true>), grid, dim3(512), 0, st, gv);      \
        else if (g.transA && !g.transB) hipLaunchKernelGGL((k_gemm<float, TO_, false, G_, false, false, true>), grid, dim3(512), 0, st, gv); \
        else if (!g.transA && g.transB) hipLaunchKernelGGL((k_gemm<float, TO_, false, G_, true, true, true>), grid, dim3(512), 0, st, gv);   \
        else hipLaunchKernelGGL((k_gemm<float, TO_, false, G_, false, true, true>), grid, dim3(512), 0, st, gv);     \
    } while (0)
#define SKM_GEMM_GO(TI_, TO_, ACC_)                    \
    do {                                               \
        if (guard) SKM_GEMM_L(TI_, TO_, ACC_, true);   \
        else SKM_GEMM_L(TI_, TO_, ACC_, false);        \
    } while (0)
    switch (kind) {
    case SKM_GEMM_F32_F32: SKM_GEMM_GO(float, float, false); break;
    case SKM_GEMM_F32_F64ACC: SKM_GEMM_GO(float, double, true); break;
    case SKM_GEMM_F32_F64OUT: SKM_GEMM_GO(float, double, false); break;
    case SKM_GEMM_F64_F64: SKM_GEMM_GO(double, double, false); break;
    case SKM_GEMM_F64_F32OUT: SKM_GEMM_GO(double, float, false); break;
    case SKM_GEMM_F32S_F32:
        if (guard) SKM_GEMM_LS(float, true);
        else SKM_GEMM_LS(float, false);
        break;
    case SKM_GEMM_F32S_F64OUT:
        if (guard) SKM_GEMM_LS(double, true);
        else SKM_GEMM_LS(double, false);
        break;
    default: skm_set_error("gemm(%s): unknown kind %d", tag, kind); return SKM_ERR_ARG;
    }
#undef SKM_GEMM_GO
#undef SKM_GEMM_L
#undef SKM_GEMM_LS
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
