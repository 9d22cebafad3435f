// Device code of the GMM kernels (E-step families, Baum-Welch segment kernel) -- included by skm_gmm.hip (host entry
// points, plans, launches) and by the translation units that only hold explicit instantiations of the team / segment
// kernels (skm_gmm_team.hip, skm_gmm_seg.hip, skm_gmm_seg2.hip), so that the ~100 instantiations of those two kernels
// compile in parallel instead of inside one 8-minute translation unit.  The instance lists at the end are the single
// source of both the `extern template` declarations of the main unit and the definitions of the others: a launch of an
// instance that is in neither list fails at link time, not at run time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skm_common.h"

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#define SKM_LDS_BUDGET (160 * 1024)
#define SKM_FLUSH_TILES 256  // f32 MFMA accumulators are folded into f64 every 8192 frames

struct TileGeom {
    int D, XS, NG, NP, tail;
    int vec4;  // 1: rows are 16-byte aligned (D % 4 == 0 and X 16-byte aligned)
};

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Columns D..XS-1 of the wave's staged tile never change: column D is the ones column that
// makes the statistics GEMM emit N, the rest is zero.
__device__ __forceinline__ void init_tile_padding(float *xs, const TileGeom &g, int lane)
{
    const int padw = g.XS - g.D;
    for (int i = lane; i < 32 * padw; i += 64) {
        const int row = i / padw, col = g.D + (i - row * padw);
        xs[row * g.XS + col] = (col == g.D) ? 1.0f : 0.0f;
    }
}

// Stage frames [t0, t0+32) (rows >= nvalid zero-filled) into xs[32][XS].
// A 32-frame tile is one contiguous span of 32*D floats of X: fully coalesced loads.
__device__ __forceinline__ void stage_tile(float *xs, const float *__restrict__ X, long t0, int nvalid,
                                           const TileGeom &g, int lane)
{
    if (g.vec4) {
        const int n4 = g.D >> 2;
        const f32x4 *X4 = reinterpret_cast<const f32x4 *>(X + t0 * (long)g.D);
        int row = lane / n4, c4 = lane - row * n4;
        const int rstep = 64 / n4, cstep = 64 - rstep * n4;
        const int total = 32 * n4;
        for (int e = lane; e < total; e += 64) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < nvalid) v = X4[e];
            *reinterpret_cast<f32x4 *>(xs + row * g.XS + 4 * c4) = v;
            row += rstep;
            c4 += cstep;
            if (c4 >= n4) {
                c4 -= n4;
                row++;
            }
        }
    } else {
        const int n = g.D;
        const float *Xs = X + t0 * (long)g.D;
        int row = lane / n, col = lane - row * n;
        const int rstep = 64 / n, cstep = 64 - rstep * n;
        const int total = 32 * n;
        for (int e = lane; e < total; e += 64) {
            float v = 0.f;
            if (row < nvalid) v = Xs[e];
            xs[row * g.XS + col] = v;
            row += rstep;
            col += cstep;
            if (col >= n) {
                col -= n;
                row++;
            }
        }
    }
}

// Register-staged variant for 16-byte-aligned rows with D <= 64: the tile's 32*D/4 float4 elements are
// requested SKM_PF_REGS per lane ahead of time (tile_prefetch) and written to LDS one tile later
// (tile_commit), so the HBM latency of tile i+1 is covered by the MFMAs of tile i.
#define SKM_PF_REGS 8
__device__ __forceinline__ bool tile_prefetchable(const TileGeom &g) { return g.vec4 && g.D <= 64; }

__device__ __forceinline__ void tile_prefetch(f32x4 (&pf)[SKM_PF_REGS], const float *__restrict__ X, long t0,
                                              int nvalid, const TileGeom &g, int lane)
{
    const int n4 = g.D >> 2, total = 32 * n4;
    const f32x4 *X4 = reinterpret_cast<const f32x4 *>(X + t0 * (long)g.D);
    const int lim = nvalid * n4;   // elements of the valid rows (rows are contiguous)
#pragma unroll
    for (int i = 0; i < SKM_PF_REGS; ++i) {
        const int e = lane + 64 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e < total && e < lim) v = X4[e];
        pf[i] = v;
    }
}

__device__ __forceinline__ void tile_commit(float *xs, const f32x4 (&pf)[SKM_PF_REGS], const TileGeom &g, int lane)
{
    const int n4 = g.D >> 2, total = 32 * n4;
    int row = lane / n4, c4 = lane - row * n4;
    const int rstep = 64 / n4, cstep = 64 - rstep * n4;
#pragma unroll
    for (int i = 0; i < SKM_PF_REGS; ++i) {
        if (lane + 64 * i < total) *reinterpret_cast<f32x4 *>(xs + row * g.XS + 4 * c4) = pf[i];
        row += rstep;
        c4 += cstep;
        if (c4 >= n4) {
            c4 -= n4;
            row++;
        }
    }
}

// Order LDS traffic of ONE wave: DS operations of a wave execute in issue order, so only the
// compiler has to be kept from moving reads above the writes of other lanes.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one settle for all accumulators: only the last MFMA can still be in flight (each earlier one issued >= 64
// cycles before it); 20 wait states cover the 18 a 16-pass MFMA needs before a VALU may read its result
template <int NCT>
__device__ __forceinline__ void vgpr_settle_all(f32x16 (&lg)[NCT])
{
    static_assert(NCT == 1 || NCT == 2 || NCT == 4, "operand lists are written for 1, 2 or 4 accumulators");
    if constexpr (NCT == 4)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(lg[0]), "+v"(lg[1]), "+v"(lg[2]), "+v"(lg[3]));
    else if constexpr (NCT == 2)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(lg[0]), "+v"(lg[1]));
    else
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(lg[0]));
}

// GEMM 1: lg[ct] += x~ tile (32 frames x 2D) . W block.  xs_row = xs + c*XS, Wl4 = W block + lane.
template <int NCT>
__device__ __forceinline__ void gemm_logits(f32x16 (&lg)[NCT], const float *xs_row, int h, const f32x4 *Wl4,
                                            const TileGeom &g)
{
    const float *xr = xs_row + 4 * h;
    const int NG = g.NG;
    for (int q = 0; q < g.NP; ++q) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(xr + 8 * q);
        const f32x4 s4 = a4 * a4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 wl = Wl4[(ct * NG + 2 * q) * 64];
            const f32x4 wq = Wl4[(ct * NG + 2 * q + 1) * 64];
            lg[ct] = SKM_MFMA32(a4[0], wl[0], lg[ct]);
            lg[ct] = SKM_MFMA32(a4[1], wl[1], lg[ct]);
            lg[ct] = SKM_MFMA32(a4[2], wl[2], lg[ct]);
            lg[ct] = SKM_MFMA32(a4[3], wl[3], lg[ct]);
            lg[ct] = SKM_MFMA32(s4[0], wq[0], lg[ct]);
            lg[ct] = SKM_MFMA32(s4[1], wq[1], lg[ct]);
            lg[ct] = SKM_MFMA32(s4[2], wq[2], lg[ct]);
            lg[ct] = SKM_MFMA32(s4[3], wq[3], lg[ct]);
        }
    }
    if (g.tail) {
        f32x4 a4 = *reinterpret_cast<const f32x4 *>(xs_row + 8 * g.NP);
        if (h) a4 = a4 * a4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 wm = Wl4[(ct * NG + NG - 1) * 64];
            lg[ct] = SKM_MFMA32(a4[0], wm[0], lg[ct]);
            lg[ct] = SKM_MFMA32(a4[1], wm[1], lg[ct]);
            lg[ct] = SKM_MFMA32(a4[2], wm[2], lg[ct]);
            lg[ct] = SKM_MFMA32(a4[3], wm[3], lg[ct]);
        }
    }
    // these are builtin MFMAs and the compiler pads its own readers, but the first readers of lg are inline asm
    // (vmax2 / vmax3 of tile_row_max), which it does not: k_estep_fused<2> took its row maxima straight behind the last
    // MFMA (tools/isa_lint.py rule i; harmless there, the maximum only centres the exponentials, but not by design)
#ifndef SKM_LINT_DEMO_LOGITS_NO_SETTLE
    vgpr_settle_all<NCT>(lg);
#endif
}


// ------------------------------------------------------------------------------------------------
// Row logsumexp of a logits tile: frames sit in the 16 accumulator registers (x2 lane halves), components
// on the 32 lanes of a half.  "Transposed" butterfly on (max, sum-of-exp2) pairs: each exchange step
// halves the number of live registers, so 16 registers x 32 lanes cost 24 cross-lane moves instead of
// 160.  On return lane (c, h) holds the pair of frame acc_row((c >> 1) & 15, h), reduced over the 32
// components of the half (and the NCT tiles); lanes c and c^1 hold the same pair.
// ------------------------------------------------------------------------------------------------
template <int MASK>
__device__ __forceinline__ float xor_lane(float v)
{
    // ds_swizzle bit mode: lane' = ((lane & 0x1f) | 0) ^ MASK inside each group of 32 lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (MASK << 10) | 0x1F));
}

template <int N, int MASK>
__device__ __forceinline__ void lse_halve(float (&m)[16], float (&s)[16], bool upper)
{
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        const float km = upper ? m[j + N / 2] : m[j], ks = upper ? s[j + N / 2] : s[j];
        const float sm = upper ? m[j] : m[j + N / 2], ss = upper ? s[j] : s[j + N / 2];
        const float rm = xor_lane<MASK>(sm), rs = xor_lane<MASK>(ss);
        const float nm = fmaxf(km, rm);
        m[j] = nm;
        s[j] = ks * __builtin_amdgcn_exp2f(km - nm) + rs * __builtin_amdgcn_exp2f(rm - nm);
    }
}

template <int NCT>
__device__ __forceinline__ void wave_lse_reduce(const f32x16 (&lg)[NCT], int c, float &m_out, float &s_out)
{
    float m[16], s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float mm = lg[0][r];
#pragma unroll
        for (int ct = 1; ct < NCT; ++ct) mm = fmaxf(mm, lg[ct][r]);
        float ss = 0.f;
        if (NCT == 1) {
            ss = 1.f;
        } else {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) ss += __builtin_amdgcn_exp2f(lg[ct][r] - mm);
        }
        m[r] = mm;
        s[r] = ss;
    }
    lse_halve<16, 16>(m, s, (c & 16) != 0);
    lse_halve<8, 8>(m, s, (c & 8) != 0);
    lse_halve<4, 4>(m, s, (c & 4) != 0);
    lse_halve<2, 2>(m, s, (c & 2) != 0);
    const float rm = xor_lane<1>(m[0]), rs = xor_lane<1>(s[0]);
    const float nm = fmaxf(m[0], rm);
    m_out = nm;
    s_out = s[0] * __builtin_amdgcn_exp2f(m[0] - nm) + rs * __builtin_amdgcn_exp2f(rm - nm);
}

// Transposed butterfly without LDS traffic: reduces v[r] (r = frame register) over the 32 component lanes
// of each wave half with `op`, halving the live registers at every level.  Level 16 is one
// v_permlane16_swap per register pair, the levels inside a 16-lane row are DPP operands.  On return
// v[0] of lane (c, h) holds the result for frame acc_row((c >> 1) & 15, h); lanes c and c^1 agree.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

struct OpMax {
    __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fmaxf(a, b); }
};
struct OpAdd {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};

template <typename Op>
__device__ __forceinline__ void lane_reduce16(float (&v)[16], int c, Op op)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // (hipcc 7.2 returns the first result twice from __builtin_amdgcn_permlane16_swap, hence the asm;
        // the s_nop pairs cover the VALU-write -> permlane and permlane -> VALU-read wait states, which the
        // hazard recogniser does not see inside an asm block)
        float lo = v[j], hi = v[j + 8];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
        v[j] = op(lo, hi);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // lane ^ 8: row_ror:8
        const float lo = op(v[j], dpp_f32<0x128>(v[j])), hi = op(v[j + 4], dpp_f32<0x128>(v[j + 4]));
        v[j] = (c & 8) ? hi : lo;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // lane -> 7 - lane inside groups of 8: row_half_mirror
        const float lo = op(v[j], dpp_f32<0x141>(v[j])), hi = op(v[j + 2], dpp_f32<0x141>(v[j + 2]));
        v[j] = (c & 4) ? hi : lo;
    }
    {                               // lane ^ 2: quad_perm [2,3,0,1]
        const float lo = op(v[0], dpp_f32<0x4E>(v[0])), hi = op(v[1], dpp_f32<0x4E>(v[1]));
        v[0] = (c & 2) ? hi : lo;
    }
    v[0] = op(v[0], dpp_f32<0xB1>(v[0]));   // lane ^ 1: quad_perm [1,0,3,2]
}

// VALU diet of the single-sweep kernels (every vector instruction takes issue cycles from the SIMD's f32 MFMA stream).
// fmaxf() on operands the compiler cannot prove quiet (asm MFMA results) costs a canonicalising v_max_f32 v, v, v per
// operand; the hardware maximum needs none, and v_max3 takes three operands.
__device__ __forceinline__ float vmax2(float a, float b)
{
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
struct OpMaxHw {
    __device__ __forceinline__ float operator()(float a, float b) const { return vmax2(a, b); }
};

// m[r] = max over the wave's NCT component tiles of lg[ct][r]
template <int NCT>
__device__ __forceinline__ void tile_row_max(const f32x16 (&lg)[NCT], float (&m)[16])
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if constexpr (NCT == 1)
            m[r] = lg[0][r];
        else if constexpr (NCT == 2)
            m[r] = vmax2(lg[0][r], lg[1][r]);
        else if constexpr (NCT == 4)
            m[r] = vmax2(vmax3(lg[0][r], lg[1][r], lg[2][r]), lg[3][r]);
        else {
            m[r] = lg[0][r];
#pragma unroll
            for (int ct = 1; ct < NCT; ++ct) m[r] = vmax2(m[r], lg[ct][r]);
        }
    }
}

// lg <- 2^(lg - max of the row), sm[r] = sum over the tiles; rmax = ls + 4 h holds the row maxima of this half-wave's
// 16 rows in groups of four (stride 8).  Two rows at a time: the subtraction and the sum are packed (v_pk_add_f32).
template <int NCT>
__device__ __forceinline__ void tile_exp_sum(f32x16 (&lg)[NCT], const float *rmax, float (&sm)[16])
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 m4 = *reinterpret_cast<const f32x4 *>(rmax + 8 * q4);
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const f32x2 mm = {m4[i], m4[i + 1]};
            f32x2 acc = {0.f, 0.f};
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x2 l = {lg[ct][4 * q4 + i], lg[ct][4 * q4 + i + 1]};
                const f32x2 d = l - mm;
                const f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                lg[ct][4 * q4 + i] = e[0];
                lg[ct][4 * q4 + i + 1] = e[1];
                acc += e;
            }
            sm[4 * q4 + i] = acc[0];
            sm[4 * q4 + i + 1] = acc[1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sweep 1: per (component block, frame) partial max / sum of exp2
// ------------------------------------------------------------------------------------------------
#ifdef SKM_TRACE
#ifdef SKM_GMM_MAIN_TU
static long long *g_trace = nullptr;
extern "C" void skm_trace_buffer(void *p) { g_trace = static_cast<long long *>(p); }
#endif
#define SKM_TRACE_DECL bool trace_now = false
#define SKM_TRACE_TILE trace_now = a.trace && blockIdx.x == 1 && lane == 0 && tile - ta == 40
#define SKM_STAMP(k) \
    if (trace_now) a.trace[wave * 16 + (k)] = __builtin_readcyclecounter()
#else
#define SKM_TRACE_DECL
#define SKM_TRACE_TILE
#define SKM_STAMP(k)
#endif

struct LseArgs {
    const float *X;
    long T;
    TileGeom g;
    const float *wp;  // packed W  [CTpad][NG][64][4]
    const float *kc;  // [CTpad*32]
    float *pm;        // [n_cb][T]
    float *ps;        // [n_cb][T]
    int n_cb, n_fs;
    double *lp;       // optional [T][C] natural-log weighted component densities (NULL: not written)
    int C;
#ifdef SKM_TRACE
    long long *trace;
#endif
};

template <int NCT>
__global__ __launch_bounds__(512, 2) void k_lse_partial(LseArgs a)
{
    extern __shared__ __align__(16) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int cb = blockIdx.x % a.n_cb, fs = blockIdx.x / a.n_cb;
    const TileGeom g = a.g;

    float *Wl = smem;
    const int wfloats = NCT * g.NG * 256;
    float *xs = smem + wfloats + wave * (32 * g.XS + 32);

    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wp) + (long)cb * (wfloats / 4);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
        for (int i = threadIdx.x; i < wfloats / 4; i += blockDim.x) dst[i] = src[i];
    }
    init_tile_padding(xs, g, lane);
    __syncthreads();

    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[(cb * NCT + ct) * 32 + c];

    const long tiles = (a.T + 31) >> 5;
    const long slot = (long)fs * NW + wave, n_slots = (long)a.n_fs * NW;
    const long ta = tiles * slot / n_slots, tb = tiles * (slot + 1) / n_slots;
    const f32x4 *Wl4 = reinterpret_cast<const f32x4 *>(Wl) + lane;
    float *pm = a.pm + (long)cb * a.T;
    float *ps = a.ps + (long)cb * a.T;

    const bool pfok = tile_prefetchable(g);
    f32x4 pf[SKM_PF_REGS];
    if (pfok && ta < tb) tile_prefetch(pf, a.X, ta << 5, (int)((a.T - (ta << 5)) < 32 ? (a.T - (ta << 5)) : 32), g, lane);
    SKM_TRACE_DECL;
    for (long tile = ta; tile < tb; ++tile) {
        const long t0 = tile << 5;
        const int nvalid = (int)((a.T - t0) < 32 ? (a.T - t0) : 32);
        SKM_TRACE_TILE;
        SKM_STAMP(0);
        if (pfok) {
            tile_commit(xs, pf, g, lane);
            if (tile + 1 < tb) {
                const long t1 = t0 + 32;
                tile_prefetch(pf, a.X, t1, (int)((a.T - t1) < 32 ? (a.T - t1) : 32), g, lane);
            }
        } else {
            stage_tile(xs, a.X, t0, nvalid, g, lane);
        }
        wave_lds_fence();

        f32x16 lg[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[ct][r] = kcv[ct];

        SKM_STAMP(1);
        gemm_logits<NCT>(lg, xs + c * g.XS, h, Wl4, g);
        SKM_STAMP(2);

        if (a.lp) {
            // lp[t][c] = ln(w_c N(x_t; c)): a row of 32 consecutive components per register
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int comp = (cb * NCT + ct) * 32 + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int frame = acc_row(r, h);
                    if (frame < nvalid && comp < a.C)
                        a.lp[(t0 + frame) * (long)a.C + comp] = (double)lg[ct][r] * 0.6931471805599453094;
                }
            }
        }
        // Row (= frame) maximum and sum of 2^(l - max) over this block's NCT * 32 components: exact maximum
        // first (DPP butterfly), broadcast through the wave's LDS row, one exponential per logit.
        float m[16];
        tile_row_max<NCT>(lg, m);
        lane_reduce16(m, c, OpMaxHw());
        const int frame = acc_row((c >> 1) & 15, h);
        float *mrow = xs + 32 * g.XS;   // 32 floats behind the wave's tile
        if ((c & 1) == 0) mrow[frame] = m[0];
        wave_lds_fence();
        float sm[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 m4 = *reinterpret_cast<const f32x4 *>(mrow + 8 * q4 + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float acc1 = 0.f;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc1 += __builtin_amdgcn_exp2f(lg[ct][4 * q4 + i] - m4[i]);
                sm[4 * q4 + i] = acc1;
            }
        }
        lane_reduce16(sm, c, OpAdd());
        if ((c & 1) == 0 && frame < nvalid) {
            pm[t0 + frame] = m[0];
            ps[t0 + frame] = sm[0];
        }
        wave_lds_fence();  // all reads of xs done before the next tile is staged
        SKM_STAMP(3);
    }
}

// lse[t] = log2 sum_c 2^l[t,c] from the per-block partials; block partial sums of lse*ln2 in f64.
static __global__ __launch_bounds__(256) void k_lse_combine(const float *__restrict__ pm, const float *__restrict__ ps,
                                                     int n_cb, long T, float *__restrict__ lse2,
                                                     float *__restrict__ loglik, double *__restrict__ llk_part)
{
    __shared__ double red[256];
    double local = 0.0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (long)gridDim.x * blockDim.x) {
        float m = pm[t];
        for (int b = 1; b < n_cb; ++b) m = fmaxf(m, pm[(long)b * T + t]);
        float s = 0.f;
        for (int b = 0; b < n_cb; ++b) s += ps[(long)b * T + t] * __builtin_amdgcn_exp2f(pm[(long)b * T + t] - m);
        const float l2 = m + __builtin_amdgcn_logf(s);  // v_log_f32 is log2
        if (lse2) lse2[t] = l2;
        if (loglik) loglik[t] = l2 * 0.6931471805599453f;
        local += (double)l2;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && llk_part) llk_part[blockIdx.x] = red[0] * 0.6931471805599453094;
}

// ------------------------------------------------------------------------------------------------
// sweep 2: posteriors and statistics
// ------------------------------------------------------------------------------------------------
struct StatsArgs {
    const float *X;
    long T;
    TileGeom g;
    const float *wp;
    const float *kc;
    const float *lse2;  // [T] base-2 logsumexp
    int n_cb, n_fs;
    int C;
    // EM mode: one f64 slab [NCT*32][NF*32] per wave slot
    double *slabs;
    // Baum-Welch mode
    const int64_t *seg_off;
    int S;
    double *stat0;
    double *stat1;
};

// GEMM 2: acc[ct][jf] += gamma' [x | x^2].  Register r of the logits tile holds gamma[frame acc_row(r,h)]
// [comp c] = the A operand of the k-pair {acc_row(r,0), acc_row(r,1)}; B = the staged frame row (32
// feature columns per tile), x^2 squared on the fly.  The B operands of step r+1 are fetched while the
// MFMAs of step r issue; the scheduling barrier keeps the compiler from hoisting all 16 steps' loads
// (which costs more registers than the kernel has).
template <int NJT, int NCT, bool WITH_S>
__device__ __forceinline__ void gemm_stats(f32x16 (&acc)[NCT][NJT * (WITH_S ? 2 : 1)], const f32x16 (&lg)[NCT],
                                           const float *xcol, int XS)
{
    float bn[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) bn[jt] = xcol[jt * 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float b[NJT];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) b[jt] = bn[jt];
        if (r < 15) {
            const float *xrow = xcol + acc_row(r + 1, 0) * XS;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) bn[jt] = xrow[jt * 32];
        }
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct][jt] = SKM_MFMA32(lg[ct][r], b[jt], acc[ct][jt]);
            if (WITH_S) {
                const float b2 = b[jt] * b[jt];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct][NJT * (WITH_S ? 1 : 0) + jt] = SKM_MFMA32(lg[ct][r], b2, acc[ct][NJT * (WITH_S ? 1 : 0) + jt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One 32-frame tile: stage, logits - lse, gamma, statistics GEMM into acc.
template <int NJT, int NCT, bool WITH_S>
__device__ __forceinline__ void stats_tile(f32x16 (&acc)[NCT][NJT * (WITH_S ? 2 : 1)], float *xs, float *ls,
                                           const float *__restrict__ X, const float *__restrict__ lse2, long t0,
                                           int nvalid, const TileGeom &g, const f32x4 *Wl4,
                                           const float (&kcv)[NCT], int lane)
{
    const int h = lane >> 5, c = lane & 31;
    stage_tile(xs, X, t0, nvalid, g, lane);
    if (lane < 32) ls[lane] = (lane < nvalid) ? lse2[t0 + lane] : 3.0e38f;
    wave_lds_fence();

    // The logits are accumulated exactly as in sweep 1 (same start value, same order: the same bits), and the
    // row's log-sum-exp is taken off afterwards.  Starting the accumulator at kc - lse instead saves 16 * NCT
    // subtractions but rounds every step of the chain differently from the sweep that made lse: the posteriors
    // of a frame then miss sum 1 by the accumulation error of a logit (4e-5 at 156 columns, 6e-6 at 60).
    f32x16 lg[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[ct][r] = kcv[ct];
    gemm_logits<NCT>(lg, xs + c * g.XS, h, Wl4, g);

#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4 *>(ls + 8 * q4 + 4 * h);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[ct][4 * q4 + i] = __builtin_amdgcn_exp2f(lg[ct][4 * q4 + i] - l4[i]);
        }
    }

    gemm_stats<NJT, NCT, WITH_S>(acc, lg, xs + 4 * h * g.XS + c, g.XS);
    wave_lds_fence();  // every read of xs retired (in order) before the next tile is staged
}

// Fold the f32 accumulators into this wave's private f64 slab (zeroed by the host before the
// launch; plain read-modify-write, one accumulator tile = 16 loads in flight at a time) and clear them.
template <int NCT, int NF>
__device__ __forceinline__ void flush_slab(f32x16 (&acc)[NCT][NF], double *slab_lane)
{
    constexpr int NFW = NF * 32;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf) {
            double *p = slab_lane + ct * 32 * NFW + jf * 32;
            double old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = p[acc_row(r, 0) * NFW];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[acc_row(r, 0) * NFW] = old[r] + (double)acc[ct][jf][r];
                acc[ct][jf][r] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// Baum-Welch flush (one store per session): lane (c,h) owns column jf*32+c of components
// acc_row(r,h).  Column < D goes to stat1 (component-major, index comp*D + d), column D (the ones
// column) to stat0.
template <int NCT, int NF>
__device__ __forceinline__ void flush_session(f32x16 (&acc)[NCT][NF], double *s0, double *s1, int cb, int C,
                                              int D, int lane, bool add = false)
{
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf) {
            const int col = jf * 32 + c;
            const int comp0 = (cb * NCT + ct) * 32 + 4 * h;
            double *p = (col < D) ? (s1 + (long)comp0 * D + col) : (s0 + comp0);
            const long stride = (col < D) ? D : 1;
            const int nlive = (col <= D) ? (C - comp0) : 0;  // rows acc_row(r,0) < nlive are stored
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (acc_row(r, 0) < nlive) {
                    double *q = p + acc_row(r, 0) * stride;
                    *q = add ? (*q + (double)acc[ct][jf][r]) : (double)acc[ct][jf][r];
                }
                acc[ct][jf][r] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <int NJT, int NCT, bool WITH_S, bool BW>
__global__ __launch_bounds__(512, 2) void k_stats(StatsArgs a)
{
    constexpr int NF = NJT * (WITH_S ? 2 : 1);
    constexpr int NFW = NF * 32;
    extern __shared__ __align__(16) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int cb = blockIdx.x % a.n_cb, fs = blockIdx.x / a.n_cb;
    const TileGeom g = a.g;

    float *Wl = smem;
    const int wfloats = NCT * g.NG * 256;
    float *xs = smem + wfloats + wave * (32 * g.XS + 32);
    float *ls = xs + 32 * g.XS;

    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.wp) + (long)cb * (wfloats / 4);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wl);
        for (int i = threadIdx.x; i < wfloats / 4; i += blockDim.x) dst[i] = src[i];
    }
    init_tile_padding(xs, g, lane);
    __syncthreads();

    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[(cb * NCT + ct) * 32 + c];

    const f32x4 *Wl4 = reinterpret_cast<const f32x4 *>(Wl) + lane;
    const long slot = (long)fs * NW + wave, n_slots = (long)a.n_fs * NW;

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    if (!BW) {
        // EM: this wave owns tiles [ta, tb); accumulators are folded into f64 every SKM_FLUSH_TILES.
        double *slab_lane =
            a.slabs + ((long)blockIdx.x * NW + wave) * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
        const long tiles = (a.T + 31) >> 5;
        const long ta = tiles * slot / n_slots, tb = tiles * (slot + 1) / n_slots;
        for (long base = ta; base < tb; base += SKM_FLUSH_TILES) {
            const long end = (base + SKM_FLUSH_TILES < tb) ? base + SKM_FLUSH_TILES : tb;
#pragma unroll 1
            for (long tile = base; tile < end; ++tile) {
                const long t0 = tile << 5;
                const int nvalid = (int)((a.T - t0) < 32 ? (a.T - t0) : 32);
                stats_tile<NJT, NCT, WITH_S>(acc, xs, ls, a.X, a.lse2, t0, nvalid, g, Wl4, kcv, lane);
            }
            flush_slab<NCT, NF>(acc, slab_lane);
        }
    } else {
        // Baum-Welch: sessions are dealt round-robin to the wave slots of this component block;
        // the f32 accumulators of a session are folded into its float64 output rows every SKM_FLUSH_TILES
        // tiles (first fold stores, later ones add: one owner per entry), as the EM slabs are.
        const int C = a.C, D = g.D;
#pragma unroll 1
        for (long s = slot; s < a.S; s += n_slots) {
            const long b = a.seg_off[s], e = a.seg_off[s + 1];
            long pb = b;
            do {
                const long pe = (pb + 32L * SKM_FLUSH_TILES < e) ? pb + 32L * SKM_FLUSH_TILES : e;
#pragma unroll 1
                for (long t0 = pb; t0 < pe; t0 += 32) {
                    const int nvalid = (int)((pe - t0) < 32 ? (pe - t0) : 32);
                    stats_tile<NJT, NCT, WITH_S>(acc, xs, ls, a.X, a.lse2, t0, nvalid, g, Wl4, kcv, lane);
                }
                flush_session<NCT, NF>(acc, a.stat0 + s * (long)C, a.stat1 + s * (long)C * D, cb, C, D, lane, pb > b);
                pb = pe;
            } while (pb < e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// single-sweep E-step for C <= 512 (7..16 component tiles): one workgroup covers ALL components, so the
// row logsumexp is finished inside the workgroup and the logits are never recomputed (124 instead of 184
// MFMAs per 32x32 logits tile).  Wave w owns component tiles 2w, 2w+1 (W streamed from L2 in fragment
// order, 1 KB per wave-instruction); all waves share one staged frame tile (three LDS buffers, so one
// workgroup barrier per tile orders both the staging and the exchange of the per-wave partials).
// ------------------------------------------------------------------------------------------------
struct FusedArgs {
    const float *X;
    long T;
    TileGeom g;
    const float *wp;
    const float *kc;
    double *slabs;     // [n_blocks][NWV] slabs of [64][NF*32] f64 (zeroed by the host)
    double *llk_part;  // [n_blocks]
    int stagger;       // 0 lock-step, 1 second half of the waves runs half a tile late, 2 and at priority 1
    int flush_tiles;   // frame tiles accumulated in f32 between two folds into the f64 slab
#ifdef SKM_TRACE
    long long *trace;  // [NWV][16] cycle stamps of one tile of block 1 (tools/trace_estep.py)
#endif
};


template <int NJT>
__global__ __launch_bounds__(512, 2) void k_estep_fused(FusedArgs a)
{
    constexpr int NCT = 2, NF = 2 * NJT, NFW = NF * 32, PRE = 8;  // PRE floats staged per thread (>= 256 threads, D <= 63)
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, NWV = nthr >> 6;
    const int h = lane >> 5, c = lane & 31;
    const TileGeom g = a.g;
    const int tile_floats = 32 * g.XS;

    float *xs = smem;                                // [3][32][XS]
    float *partm = xs + 3 * tile_floats;             // [2][NWV][32]
    float *parts = partm + 2 * NWV * 32;             // [2][NWV][32]
    float *ls = parts + 2 * NWV * 32 + wave * 32;    // wave-private lse[32]
    f32x4 *stash = reinterpret_cast<f32x4 *>(parts + 2 * NWV * 32 + NWV * 32) + wave * (NCT * 4 * 64) + lane;  // wave-private

    // constant padding columns (ones column at D) of the three buffers
    {
        const int padw = g.XS - g.D;
        for (int i = tid; i < 3 * 32 * padw; i += nthr) {
            const int row = i / padw, col = g.D + (i - row * padw);
            xs[row * g.XS + col] = (col == g.D) ? 1.0f : 0.0f;
        }
    }
    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[(wave * NCT + ct) * 32 + c];
    const f32x4 *Wg4 = reinterpret_cast<const f32x4 *>(a.wp) + (long)(wave * NCT) * g.NG * 64 + lane;

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    const long tiles = (a.T + 31) >> 5;
    const long ta = tiles * blockIdx.x / gridDim.x, tb = tiles * (blockIdx.x + 1) / gridDim.x;
    double *slab_lane = a.slabs + ((long)blockIdx.x * NWV + wave) * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
    double llk_lane = 0.0;

    // cooperative staging: element e of the tile's contiguous span -> (row, col); up to PRE floats per thread
    const int n_el = g.vec4 ? 32 * (g.D >> 2) : 32 * g.D;   // float4 or float elements
    float pre[PRE];
    auto load_pre = [&](long t0, int nvalid) __attribute__((always_inline)) {
        if (g.vec4) {
            const int n4 = g.D >> 2;
            const f32x4 *X4 = reinterpret_cast<const f32x4 *>(a.X + t0 * (long)g.D);
#pragma unroll
            for (int i = 0; i < PRE / 4; ++i) {
                const int e = tid + i * nthr;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (e < n_el && e / n4 < nvalid) v = X4[e];
                pre[4 * i + 0] = v[0];
                pre[4 * i + 1] = v[1];
                pre[4 * i + 2] = v[2];
                pre[4 * i + 3] = v[3];
            }
        } else {
            const float *Xs = a.X + t0 * (long)g.D;
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int e = tid + i * nthr;
                pre[i] = (e < n_el && e / g.D < nvalid) ? Xs[e] : 0.f;
            }
        }
    };
    auto store_pre = [&](float *dst) __attribute__((always_inline)) {
        if (g.vec4) {
            const int n4 = g.D >> 2;
#pragma unroll
            for (int i = 0; i < PRE / 4; ++i) {
                const int e = tid + i * nthr;
                if (e < n_el) {
                    const int row = e / n4, c4 = e - row * n4;
                    f32x4 v = {pre[4 * i], pre[4 * i + 1], pre[4 * i + 2], pre[4 * i + 3]};
                    *reinterpret_cast<f32x4 *>(dst + row * g.XS + 4 * c4) = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int e = tid + i * nthr;
                if (e < n_el) {
                    const int row = e / g.D;
                    dst[row * g.XS + (e - row * g.D)] = pre[i];
                }
            }
        }
    };

    if (ta < tb) {
        const long t0 = ta << 5;
        load_pre(t0, (int)((a.T - t0) < 32 ? (a.T - t0) : 32));
        store_pre(xs);
    }
    __syncthreads();

    SKM_TRACE_DECL;
    // logits of one tile and this wave's (max, sum) partials of its 64 components -> LDS slot
    auto logits_phase = [&](f32x16 (&lg)[NCT], const float *xt, int pslot) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[ct][r] = kcv[ct];
        gemm_logits<NCT>(lg, xt + c * g.XS, h, Wg4, g);
        SKM_STAMP(1);
        // exact per-frame maximum over this wave's components ...
        float m[16];
        tile_row_max<NCT>(lg, m);
        lane_reduce16(m, c, OpMaxHw());
        const int frame = acc_row((c >> 1) & 15, h);
        if ((c & 1) == 0) ls[frame] = m[0];
        wave_lds_fence();
        // ... e = 2^(l - max) replaces the logits (each exponential is taken once; gamma = e * factor later),
        // and the wave's per-frame sum of e
        float sm[16];
        tile_exp_sum<NCT>(lg, ls + 4 * h, sm);
        lane_reduce16(sm, c, OpAdd());
        if ((c & 1) == 0) {
            partm[(pslot * NWV + wave) * 32 + frame] = m[0];
            parts[(pslot * NWV + wave) * 32 + frame] = sm[0];
        }
        wave_lds_fence();   // ls is rewritten by stats_phase
    };
    // combine the NWV partial pairs of every frame (each wave for itself): gamma = e * 2^(m_wave - m) / sum
    auto stats_phase = [&](f32x16 (&lg)[NCT], const float *xt, int pslot, int nvalid) __attribute__((always_inline)) {
        {
            const float *pmv = partm + pslot * NWV * 32 + c;
            const float *psv = parts + pslot * NWV * 32 + c;
            float m = pmv[0];
            for (int w = 1; w < NWV; ++w) m = __builtin_fmaxf(m, pmv[w * 32]);
            float sum = 0.f;
            for (int w = 0; w < NWV; ++w) sum += psv[w * 32] * __builtin_amdgcn_exp2f(pmv[w * 32] - m);
            const bool valid = c < nvalid;
            const float f = __builtin_amdgcn_exp2f(pmv[wave * 32] - m) * __builtin_amdgcn_rcpf(sum);
            if (h == 0) ls[c] = valid ? f : 0.f;
            if (wave == 0 && h == 0 && valid) llk_lane += (double)(m + __builtin_amdgcn_logf(sum));
        }
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 f4 = *reinterpret_cast<const f32x4 *>(ls + 8 * q4 + 4 * h);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                lg[ct][4 * q4 + 0] *= f4[0];
                lg[ct][4 * q4 + 1] *= f4[1];
                lg[ct][4 * q4 + 2] *= f4[2];
                lg[ct][4 * q4 + 3] *= f4[3];
            }
        }
        SKM_STAMP(5);
        gemm_stats<NJT, NCT, true>(acc, lg, xt + 4 * h * g.XS + c, g.XS);
        SKM_STAMP(6);
        wave_lds_fence();
    };

    // Between two workgroup barriers every wave owes: statistics of tile i-1 and logits of tile i.  The two
    // waves of a SIMD (w, w + NWV/2) run them in opposite order, so one is in its VALU stretch (logsumexp,
    // exp2) while the other streams MFMAs, instead of both reaching each phase together.
    const bool late = (a.stagger != 0) && wave >= (NWV >> 1);
    if (late && a.stagger > 1) __builtin_amdgcn_s_setprio(1);
    int cur = 0, since = 0;
    if (!late) {
        for (long tile = ta; tile < tb; ++tile) {
            const long t0 = tile << 5;
            const int nvalid = (int)((a.T - t0) < 32 ? (a.T - t0) : 32);
            const int nxt = (cur == 2) ? 0 : cur + 1;
            const int pslot = (int)((tile - ta) & 1);
            const bool has_next = tile + 1 < tb;
            if (has_next) {
                const long t1 = (tile + 1) << 5;
                load_pre(t1, (int)((a.T - t1) < 32 ? (a.T - t1) : 32));
            }
            const float *xt = xs + cur * tile_floats;
            f32x16 lg[NCT];
            SKM_TRACE_TILE;
            SKM_STAMP(0);
            logits_phase(lg, xt, pslot);
            SKM_STAMP(2);
            // next tile goes to the buffer last read two tiles ago (every wave is past that tile's barrier)
            if (has_next) store_pre(xs + nxt * tile_floats);
            SKM_STAMP(3);
            __syncthreads();
            SKM_STAMP(4);
            stats_phase(lg, xt, pslot, nvalid);
            SKM_STAMP(7);
            if (++since == SKM_FLUSH_TILES) {
                flush_slab<NCT, NF>(acc, slab_lane);
                since = 0;
            }
            cur = nxt;
        }
    } else {
        // the logits of a tile wait in a wave-private LDS stash while the next tile's logits are computed
        auto swap_stash = [&](f32x16 (&lg)[NCT], bool rd, bool wr) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (rd) o = stash[(ct * 4 + q) * 64];
                    if (wr) {
                        const f32x4 n = {lg[ct][4 * q], lg[ct][4 * q + 1], lg[ct][4 * q + 2], lg[ct][4 * q + 3]};
                        stash[(ct * 4 + q) * 64] = n;
                    }
                    lg[ct][4 * q] = o[0];
                    lg[ct][4 * q + 1] = o[1];
                    lg[ct][4 * q + 2] = o[2];
                    lg[ct][4 * q + 3] = o[3];
                }
        };
        int prv = 0;
        for (long tile = ta; tile < tb; ++tile) {
            const int nxt = (cur == 2) ? 0 : cur + 1;
            const int pslot = (int)((tile - ta) & 1);
            const bool has_next = tile + 1 < tb;
            if (has_next) {
                const long t1 = (tile + 1) << 5;
                load_pre(t1, (int)((a.T - t1) < 32 ? (a.T - t1) : 32));
            }
            f32x16 lg[NCT];
            SKM_TRACE_TILE;
            SKM_STAMP(0);
            logits_phase(lg, xs + cur * tile_floats, pslot);
            SKM_STAMP(2);
            if (has_next) store_pre(xs + nxt * tile_floats);
            SKM_STAMP(3);
            if (tile > ta) {
                swap_stash(lg, true, true);
                SKM_STAMP(4);
                stats_phase(lg, xs + prv * tile_floats, pslot ^ 1, 32);   // only the last tile can be ragged
                SKM_STAMP(7);
                if (++since == SKM_FLUSH_TILES) {
                    flush_slab<NCT, NF>(acc, slab_lane);
                    since = 0;
                }
            } else {
                swap_stash(lg, false, true);
            }
            __syncthreads();
            SKM_STAMP(8);
            prv = cur;
            cur = nxt;
        }
        if (ta < tb) {
            const long t0 = (tb - 1) << 5;
            f32x16 lg[NCT];
            swap_stash(lg, true, false);
            stats_phase(lg, xs + prv * tile_floats, (int)((tb - 1 - ta) & 1), (int)((a.T - t0) < 32 ? (a.T - t0) : 32));
            ++since;
        }
    }
    if (since) flush_slab<NCT, NF>(acc, slab_lane);
    if (wave == 0) {
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) llk_lane += __shfl_xor(llk_lane, msk);
        if (lane == 0) a.llk_part[blockIdx.x] = llk_lane * 0.6931471805599453094;
    }
}

// ------------------------------------------------------------------------------------------------
// Single-sweep E-step, one wave per SIMD.  Measured on gfx950 (tools/ubench/mfma_valu_mix.hip): a VALU
// instruction issued between two f32 MFMAs of the SAME wave costs ~2.5-3 matrix-pipe cycles (the f32 MFMA
// shares the FP32 datapath), and a second wave on the SIMD only gets one VALU issue per ~12 cycles while its
// partner streams MFMAs.  So instead of two 256-register waves per SIMD that fight over the issue port,
// this kernel runs ONE 512-register wave per SIMD that owns NCT = 4 component tiles: the per-tile
// logsumexp work (two lane butterflies, the cross-wave combine) is paid once per 128 components instead
// of once per 64 and runs at the full VALU rate, and the register file has room to request the W
// fragments of the next pair (32 registers) while the 32 MFMAs of the current pair issue.
// ------------------------------------------------------------------------------------------------
// The statistics accumulators of the one-wave kernel fill the 256 AGPRs exactly.  hipcc 7.2 left to itself
// parks them in VGPRs and shuttles them through AGPRs and scratch every tile, so these MFMAs are written
// with an explicit "a" (AGPR) constraint.  Hazards the compiler cannot see inside an asm block are covered
// by hand: `s_nop 1` where a B operand was just produced by a VALU instruction, and agpr_settle() before the
// accumulators are read back.
__device__ __forceinline__ void mfma_agpr(f32x16 &acc, float a, float b)
{
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_agpr_after_valu(f32x16 &acc, float a, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// logits accumulators stay in VGPRs (a builtin MFMA in this kernel makes the allocator claim AGPRs for its
// result and evict the statistics accumulators every tile)
__device__ __forceinline__ void mfma_vgpr(f32x16 &acc, float a, float b)
{
#ifdef SKM_FAT_BUILTIN
    acc = SKM_MFMA32(a, b, acc);
    return;
#endif
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr_after_valu(f32x16 &acc, float a, float b)
{
#ifdef SKM_FAT_BUILTIN
    acc = SKM_MFMA32(a, b, acc);
    return;
#endif
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// VALU write -> MFMA read of the same VGPRs needs 2 wait states the compiler cannot place for an asm MFMA
__device__ __forceinline__ void valu_to_mfma(f32x16 &v)
{
    asm volatile("s_nop 3" : "+v"(v));
}
__device__ __forceinline__ void vgpr_settle(f32x16 &acc)
{
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
}
__device__ __forceinline__ void agpr_settle(f32x16 &acc)
{
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc));
}

template <int NCT>
struct WSet {
    f32x4 l[NCT], q[NCT];
};

// W fragments are fetched with buffer loads: one per-lane VGPR offset (lane * 16) for every load, the group
// address is a scalar offset, so the stream costs no vector instructions for addressing and no address
// registers (a flat-address version kept 120 64-bit pointers alive and spilled them).
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;    // lane * 16
    int sbase;   // byte offset of this wave's first component tile
};

template <int NCT>
__device__ __forceinline__ void wset_load(WSet<NCT> &w, const WStream &ws, int NG, int NP, int k)
{
    // pair k: linear group 2k, square group 2k+1; k == NP names the mixed tail group
    const int gl = (k < NP) ? 2 * k : NG - 1, gq = (k < NP) ? 2 * k + 1 : NG - 1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        w.l[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff,
                                                                                  ws.sbase + (ct * NG + gl) * 1024, 0));
        w.q[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff,
                                                                                  ws.sbase + (ct * NG + gq) * 1024, 0));
    }
}

__device__ __forceinline__ f32x4 wfrag_load(const WStream &ws, int soff)
{
#ifdef SKM_ABL_W0
    soff = 0;
#endif
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, ws.sbase + soff, 0));
}

// 8 * NCT MFMAs of pair `cur` on the tile row a4, with the 2 * NCT fragment loads of set `knext` spread
// between them (one load behind every 4 MFMAs: a single wave issues in order, so a cluster of loads and
// their scalar address arithmetic between two MFMA groups leaves the matrix pipe idle for ~300 cycles).
// All 8 MFMAs of one accumulator run back to back.
template <int NCT>
__device__ __forceinline__ void wset_mma_pf(f32x16 (&lg)[NCT], const WSet<NCT> &cur, const f32x4 a4, WSet<NCT> &nxt,
                                            const WStream &ws, int NG, int NP, int knext)
{
    f32x4 s4 = a4 * a4;
    asm volatile("s_nop 1" : "+v"(s4));
    const int gl = (knext < NP) ? 2 * knext : NG - 1, gq = (knext < NP) ? 2 * knext + 1 : NG - 1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mfma_vgpr(lg[ct], a4[i], cur.l[ct][i]);
        nxt.l[ct] = wfrag_load(ws, (ct * NG + gl) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) mfma_vgpr(lg[ct], s4[i], cur.q[ct][i]);
        nxt.q[ct] = wfrag_load(ws, (ct * NG + gq) * 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// mixed tail group (held in wl) with the loads of pair 0 for the next tile spread between the MFMAs.
// Branch-free, and the results are settled before the function returns: wherever control flow merges
// behind asm MFMAs the compiler places register copies that would read the accumulators too early.
template <int NCT, bool TAIL>
__device__ __forceinline__ void wset_tail_pf(f32x16 (&lg)[NCT], const f32x4 (&wl)[NCT], f32x4 xt, WSet<NCT> &nxt,
                                             const WStream &ws, int NG)
{
    asm volatile("s_nop 1" : "+v"(xt));
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (TAIL) {
            // where this block follows a branch the compiler may have just copied lg[ct] into the tuple it names here
            // (seen at NCT = 2: v_mov_b64 one wait state ahead of the MFMA that reads it as SrcC; tools/isa_lint.py rule ii)
#ifndef SKM_LINT_DEMO_TAIL_NO_PAD
            mfma_vgpr_after_valu(lg[ct], xt[0], wl[ct][0]);
#else
            mfma_vgpr(lg[ct], xt[0], wl[ct][0]);
#endif
#pragma unroll
            for (int i = 1; i < 4; ++i) mfma_vgpr(lg[ct], xt[i], wl[ct][i]);
        }
        nxt.l[ct] = wfrag_load(ws, (ct * NG) * 1024);
        nxt.q[ct] = wfrag_load(ws, (ct * NG + 1) * 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
    vgpr_settle_all<NCT>(lg);
}

// lg += [x, x^2] . W with W streamed from L2; `wa` must already hold pair 0 on entry (requested during the
// previous tile) and holds pair 0 again on return.
template <int NCT, int GEO = -1>
__device__ __forceinline__ void gemm_logits_fat(f32x16 (&lg)[NCT], WSet<NCT> &wa, const float *xs_row, int h,
                                                const WStream &Wg4, const TileGeom &g, long long *tr = nullptr)
{
    const float *xr = xs_row + 4 * h;
    const int NG = g.NG, NP = g.NP;
    WSet<NCT> wb;
    f32x4 xa = *reinterpret_cast<const f32x4 *>(xr), xb;
    f32x4 xt = *reinterpret_cast<const f32x4 *>(xs_row + 8 * NP);
    if (h) xt = xt * xt;
    int q = 0;
#ifdef SKM_TRACE
    if (tr) tr[8] = __builtin_readcyclecounter();
#endif
    for (; q + 2 <= NP; q += 2) {
#ifdef SKM_TRACE
        if (tr && q == 2) tr[9] = __builtin_readcyclecounter();
        if (tr && q == 4) tr[10] = __builtin_readcyclecounter();
#endif
        xb = *reinterpret_cast<const f32x4 *>(xr + 8 * (q + 1));
        wset_mma_pf<NCT>(lg, wa, xa, wb, Wg4, NG, NP, q + 1);
        xa = *reinterpret_cast<const f32x4 *>(xr + 8 * ((q + 2 < NP) ? q + 2 : 0));
        wset_mma_pf<NCT>(lg, wb, xb, wa, Wg4, NG, NP, q + 2);   // q + 2 == NP: the tail group
    }
    vgpr_settle_all<NCT>(lg);   // the blocks below may start with copies of lg
#ifdef SKM_TRACE
    if (tr) tr[11] = __builtin_readcyclecounter();
#endif
    // GEO >= 0: pair-count parity (bit 1) and tail group (bit 0) are compile-time, so this ending is straight-line
    // code (the runtime form costs ~150 register copies per tile where its four branches merge)
    const bool odd = (GEO >= 0) ? ((GEO & 2) != 0) : (q < NP);
    const bool tail = (GEO >= 0) ? ((GEO & 1) != 0) : (g.tail != 0);
    if (odd) {      // pair q sits in wa, the tail group goes to wb
        wset_mma_pf<NCT>(lg, wa, xa, wb, Wg4, NG, NP, q + 1);
        // runtime geometry: the branch on `tail` follows, and the compiler opens one of its arms with copies of lg
        // three wait states behind the last MFMA (tools/isa_lint.py rule i found them in every GEO = -1 instance)
#ifndef SKM_LINT_DEMO_ODD_NO_SETTLE
        if (GEO < 0) vgpr_settle_all<NCT>(lg);
#endif
        if (tail)
            wset_tail_pf<NCT, true>(lg, wb.l, xt, wa, Wg4, NG);
        else
            wset_tail_pf<NCT, false>(lg, wb.l, xt, wa, Wg4, NG);
    } else {        // wa holds the tail group
        WSet<NCT> w0;
        if (tail)
            wset_tail_pf<NCT, true>(lg, wa.l, xt, w0, Wg4, NG);
        else
            wset_tail_pf<NCT, false>(lg, wa.l, xt, w0, Wg4, NG);
        wa = w0;
    }
}

// VACC: the accumulators live in VGPRs (a kernel that uses no AGPR at all gets the whole 256-register budget of a
// two-waves-per-SIMD launch as VGPRs; with any AGPR use hipcc splits that budget 128 / 128)
template <int NJT, int NCT, bool WITH_S = true, bool VACC = false>
__device__ __forceinline__ void gemm_stats_agpr(f32x16 (&acc)[NCT][NJT * (WITH_S ? 2 : 1)], const f32x16 (&lg)[NCT],
                                                const float *xcol, int XS)
{
    float bn[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) bn[jt] = xcol[jt * 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float b[NJT], b2[NJT];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            b[jt] = bn[jt];
            b2[jt] = b[jt] * b[jt];
        }
        if (r < 15) {
            const float *xrow = xcol + acc_row(r + 1, 0) * XS;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) bn[jt] = xrow[jt * 32];
        }
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (VACC)
                    mfma_vgpr(acc[ct][jt], lg[ct][r], b[jt]);
                else
                    mfma_agpr(acc[ct][jt], lg[ct][r], b[jt]);
            }
        if (WITH_S) {
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    if (VACC) {
                        if (ct == 0)
                            mfma_vgpr_after_valu(acc[ct][NJT * (WITH_S ? 1 : 0) + jt], lg[ct][r], b2[jt]);
                        else
                            mfma_vgpr(acc[ct][NJT * (WITH_S ? 1 : 0) + jt], lg[ct][r], b2[jt]);
                    } else if (ct == 0)
                        mfma_agpr_after_valu(acc[ct][NJT * (WITH_S ? 1 : 0) + jt], lg[ct][r], b2[jt]);
                    else
                        mfma_agpr(acc[ct][NJT * (WITH_S ? 1 : 0) + jt], lg[ct][r], b2[jt]);
                }
        }
    }
}

// STORE: the first flush of a wave writes its slab instead of adding to it (no memset, no read)
template <int NCT, int NF, bool STORE = false, bool VACC = false>
__device__ __forceinline__ void flush_slab_agpr(f32x16 (&acc)[NCT][NF], double *slab_lane)
{
    constexpr int NFW = NF * 32;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf) {
            if (VACC)
                vgpr_settle(acc[ct][jf]);
            else
                agpr_settle(acc[ct][jf]);
            double *p = slab_lane + ct * 32 * NFW + jf * 32;
            const f32x16 v = acc[ct][jf];
            double old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = STORE ? 0.0 : p[acc_row(r, 0) * NFW];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[acc_row(r, 0) * NFW] = old[r] + (double)v[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <int NCT, int NJT, int GEO, int PRE>
__global__ __launch_bounds__(256, 1) void k_estep_fat(FusedArgs a)
{
    constexpr int NF = 2 * NJT, NFW = NF * 32;   // PRE = floats staged per thread: 8 (4 waves) or 16 (2 waves)
    // the wide-row instantiations (one component tile per wave) are only launched with 16-byte rows: without the
    // scalar staging path their tile loop keeps its 20 / 40 staging registers out of scratch
    constexpr bool VEC4 = (NCT == 1);
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, NWV = nthr >> 6;
    const int h = lane >> 5, c = lane & 31;
    const TileGeom g = a.g;
    const int tile_floats = 32 * g.XS;

    float *xs = smem;                                // [3][32][XS]
    float *partm = xs + 3 * tile_floats;             // [2][NWV][32]
    float *parts = partm + 2 * NWV * 32;             // [2][NWV][32]
    float *ls = parts + 2 * NWV * 32 + wave * 32;    // wave-private row of 32

    {
        const int padw = g.XS - g.D;
        for (int i = tid; i < 3 * 32 * padw; i += nthr) {
            const int row = i / padw, col = g.D + (i - row * padw);
            xs[row * g.XS + col] = (col == g.D) ? 1.0f : 0.0f;
        }
    }
    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[(wave * NCT + ct) * 32 + c];
    WStream Wg4;
    Wg4.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, NWV * NCT * g.NG * 1024, 0x00027000);
    Wg4.voff = lane * 16;
    Wg4.sbase = __builtin_amdgcn_readfirstlane(wave) * (NCT * g.NG * 1024);

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    const long tiles = (a.T + 31) >> 5;
    const long ta = tiles * blockIdx.x / gridDim.x, tb = tiles * (blockIdx.x + 1) / gridDim.x;
    double *slab_lane = a.slabs + ((long)blockIdx.x * NWV + wave) * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
    double llk_lane = 0.0;

    // cooperative staging of the contiguous 32*D span of a tile: PRE floats per thread (the host only selects
    // this kernel when 32*D <= PRE*nthr); source / destination offsets are fixed per thread
    constexpr int NST = PRE / 4;
    float pre[PRE];
    int st_src[NST], st_dst[NST], st_row[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        // vec4: element = float4 index; scalar geometry: elements are handled 4 consecutive floats per slot
        const int e = tid + i * nthr;
        if (VEC4 || g.vec4) {
            const int n4 = g.D >> 2;
            const int row = e / n4;
            st_src[i] = e;
            st_row[i] = (e < 32 * n4) ? row : 1 << 20;
            st_dst[i] = row * g.XS + 4 * (e - row * n4);
        } else {
            st_src[i] = e;
            st_row[i] = 0;
            st_dst[i] = 0;
        }
    }
    auto load_pre = [&](long t0, int nvalid) __attribute__((always_inline)) {
        if (VEC4 || g.vec4) {
            const f32x4 *X4 = reinterpret_cast<const f32x4 *>(a.X + t0 * (long)g.D);
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (st_row[i] < nvalid) v = X4[st_src[i]];
                pre[4 * i + 0] = v[0];
                pre[4 * i + 1] = v[1];
                pre[4 * i + 2] = v[2];
                pre[4 * i + 3] = v[3];
            }
        } else {
            const float *Xs = a.X + t0 * (long)g.D;
            const int n_el = 32 * g.D;
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int e = tid + i * nthr;
                pre[i] = (e < n_el && e / g.D < nvalid) ? Xs[e] : 0.f;
            }
        }
    };
    auto store_pre = [&](float *dst) __attribute__((always_inline)) {
        if (VEC4 || g.vec4) {
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                if (st_row[i] < 32) {
                    f32x4 v = {pre[4 * i], pre[4 * i + 1], pre[4 * i + 2], pre[4 * i + 3]};
                    *reinterpret_cast<f32x4 *>(dst + st_dst[i]) = v;
                }
            }
        } else {
            const int n_el = 32 * g.D;
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int e = tid + i * nthr;
                if (e < n_el) {
                    const int row = e / g.D;
                    dst[row * g.XS + (e - row * g.D)] = pre[i];
                }
            }
        }
    };

    if (ta < tb) {
        const long t0 = ta << 5;
        load_pre(t0, (int)((a.T - t0) < 32 ? (a.T - t0) : 32));
        store_pre(xs);
        if (ta + 1 < tb) load_pre(t0 + 32, (int)((a.T - t0 - 32) < 32 ? (a.T - t0 - 32) : 32));
    }
    WSet<NCT> wa;
    wset_load<NCT>(wa, Wg4, g.NG, g.NP, 0);
    __syncthreads();

    SKM_TRACE_DECL;
    int cur = 0;
    // The logits of a tile start from the constants kc.  The splat for the NEXT tile is issued at the end of a tile,
    // behind the statistics GEMM, where it costs what it cost at the top (it runs in the shadow of the last MFMA
    // either way) and puts 16 NCT + 4 wait states between that MFMA and the loop's exit edge: the compiler moves
    // accumulators between registers there, ahead of flush_slab_agpr's own settles (v_accvgpr_read of the last-written
    // tuple 10 wait states behind its MFMA; tools/isa_lint.py rule i), and a wait placed behind the loop, or in an
    // `if (last tile)`, lands after those moves or costs the allocator its AGPR assignment.
    f32x16 lg[NCT];
    auto logits_start = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            // opaque per tile: otherwise the 64-register splat is hoisted out of the loop, spilled, and
            // reloaded from scratch in front of every tile's first MFMA
            float k = kcv[ct];
            asm volatile("" : "+v"(k));
#ifdef SKM_ABL_NOSPLAT
            asm volatile("" : "=v"(lg[ct]));
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[ct][r] = k;
#endif
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
    };
    for (long base = ta; base < tb; base += a.flush_tiles) {
    const long end = (base + a.flush_tiles < tb) ? base + a.flush_tiles : tb;
#ifndef SKM_LINT_DEMO_EXIT_NO_SETTLE
    logits_start();
#endif
#pragma unroll 1
    for (long tile = base; tile < end; ++tile) {
        const long t0 = tile << 5;
        const int nvalid = (int)((a.T - t0) < 32 ? (a.T - t0) : 32);
        const int nxt = (cur == 2) ? 0 : cur + 1;
        const int pslot = (int)((tile - ta) & 1);
        const bool has_next = tile + 1 < tb;
        SKM_TRACE_TILE;
        SKM_STAMP(0);
        const float *xt = xs + cur * tile_floats;

#ifdef SKM_LINT_DEMO_EXIT_NO_SETTLE
        logits_start();
#endif
#ifdef SKM_TRACE
        gemm_logits_fat<NCT, GEO>(lg, wa, xt + c * g.XS, h, Wg4, g, trace_now ? a.trace + wave * 16 : nullptr);
#elif !defined(SKM_ABL_NOLOGITS)
        gemm_logits_fat<NCT, GEO>(lg, wa, xt + c * g.XS, h, Wg4, g);
#endif
        SKM_STAMP(1);
#ifndef SKM_ABL_NOLSE

        // exact per-frame maximum over this wave's NCT*32 components, e = 2^(l - max), per-frame sum of e
        float m[16];
        tile_row_max<NCT>(lg, m);
        lane_reduce16(m, c, OpMaxHw());
        const int frame = acc_row((c >> 1) & 15, h);
        if ((c & 1) == 0) ls[frame] = m[0];
        wave_lds_fence();
        float sm[16];
        tile_exp_sum<NCT>(lg, ls + 4 * h, sm);
        lane_reduce16(sm, c, OpAdd());
        if ((c & 1) == 0) {
            partm[(pslot * NWV + wave) * 32 + frame] = m[0];
            parts[(pslot * NWV + wave) * 32 + frame] = sm[0];
        }
#endif
        SKM_STAMP(2);
        // next tile goes to the buffer last read two tiles ago (every wave is past that tile's barrier)
        if (has_next) store_pre(xs + nxt * tile_floats);
        SKM_STAMP(3);
        __syncthreads();
        SKM_STAMP(4);
#ifndef SKM_ABL_NOLSE
        {
            const float *pmv = partm + pslot * NWV * 32 + c;
            const float *psv = parts + pslot * NWV * 32 + c;
            float mx = pmv[0];
            for (int w = 1; w < NWV; ++w) mx = __builtin_fmaxf(mx, pmv[w * 32]);
            float sum = 0.f;
            for (int w = 0; w < NWV; ++w) sum += psv[w * 32] * __builtin_amdgcn_exp2f(pmv[w * 32] - mx);
            const bool valid = c < nvalid;
            const float f = __builtin_amdgcn_exp2f(pmv[wave * 32] - mx) * __builtin_amdgcn_rcpf(sum);
            if (h == 0) ls[c] = valid ? f : 0.f;
            if (wave == 0 && h == 0 && valid) llk_lane += (double)(mx + __builtin_amdgcn_logf(sum));
        }
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 f4 = *reinterpret_cast<const f32x4 *>(ls + 8 * q4 + 4 * h);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                lg[ct][4 * q4 + 0] *= f4[0];
                lg[ct][4 * q4 + 1] *= f4[1];
                lg[ct][4 * q4 + 2] *= f4[2];
                lg[ct][4 * q4 + 3] *= f4[3];
            }
        }
#endif
        // frames of tile+2 are requested here: the statistics GEMM that follows has no other vector-memory
        // wait, so the HBM latency never sits in front of an in-order vmcnt wait of the W stream
        if (tile + 2 < tb) {
            const long t2 = (tile + 2) << 5;
            load_pre(t2, (int)((a.T - t2) < 32 ? (a.T - t2) : 32));
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
        SKM_STAMP(5);
#ifndef SKM_ABL_NOSTATS
        gemm_stats_agpr<NJT, NCT>(acc, lg, xt + 4 * h * g.XS + c, g.XS);
#endif
        SKM_STAMP(6);
#ifndef SKM_LINT_DEMO_EXIT_NO_SETTLE
        logits_start();     // the next tile's, and the wait states in front of the exit edge (see above the loop)
#endif
        wave_lds_fence();
        cur = nxt;
    }
    // the accumulators are touched by nothing but the asm MFMAs inside the tile loop, so they stay in AGPRs
    if (base == ta)
        flush_slab_agpr<NCT, NF, true>(acc, slab_lane);
    else
        flush_slab_agpr<NCT, NF, false>(acc, slab_lane);
    }
    if (ta >= tb) flush_slab_agpr<NCT, NF, true>(acc, slab_lane);   // empty range: the slab must still read as zero
    if (wave == 0) {
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) llk_lane += __shfl_xor(llk_lane, msk);
        if (lane == 0) a.llk_part[blockIdx.x] = llk_lane * 0.6931471805599453094;
    }
}

// ------------------------------------------------------------------------------------------------
// Single-sweep E-step for C <= 128 (the first seven stages of EM_split): ONE wave holds every component
// (NCT = 1, 2 or 4 tiles of 32), so each wave is a workgroup of its own that walks its own range of frame
// tiles -- no cross-wave exchange, no workgroup barrier, and the posterior normalisation is exact inside
// the wave.  Same building blocks as k_estep_fat (W streamed through buffer loads, statistics in AGPRs,
// f64 slab per wave).  The frame ring has two slots: a single in-order wave has finished reading slot
// cur^1 (statistics GEMM of the previous tile) before it stages the next tile there.  NCT = 1 leaves
// room for two waves per SIMD, which hides the LDS / settle / HBM waits of one wave behind the MFMAs of
// the other.
// ------------------------------------------------------------------------------------------------
template <int NCT, int NJT, int GEO>
__global__ __launch_bounds__(64, NCT == 1 ? 2 : 1) void k_estep_solo(FusedArgs a)
{
    constexpr int NF = 2 * NJT, NFW = NF * 32;
    constexpr bool VACC = NCT == 1;     // two waves per SIMD: 256 VGPRs, no AGPR
    extern __shared__ __align__(16) float smem[];
    const int lane = threadIdx.x;
    const int h = lane >> 5, c = lane & 31;
    const TileGeom g = a.g;
    const int tile_floats = 32 * g.XS;

    float *xs = smem;                       // [2][32][XS]
    float *ls = xs + 2 * tile_floats;       // [2][32]: per-frame maximum, per-frame 1/sum

    {
        const int padw = g.XS - g.D;
        for (int i = lane; i < 2 * 32 * padw; i += 64) {
            const int row = i / padw, col = g.D + (i - row * padw);
            xs[row * g.XS + col] = (col == g.D) ? 1.0f : 0.0f;
        }
    }
    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[ct * 32 + c];
    WStream Wg4;
    Wg4.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, NCT * g.NG * 1024, 0x00027000);
    Wg4.voff = lane * 16;
    Wg4.sbase = 0;

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    const long tiles = (a.T + 31) >> 5;
    const long ta = tiles * blockIdx.x / gridDim.x, tb = tiles * (blockIdx.x + 1) / gridDim.x;
    double *slab_lane = a.slabs + (long)blockIdx.x * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
    double llk_lane = 0.0;

    // Frame tiles go from HBM straight into LDS (buffer_load ... lds, one row of D floats per instruction, lanes
    // < D): no staging registers, no vector ALU.  The row's descriptor is rebuilt by the scalar unit; rows past
    // the end of X get an empty extent, so their lanes write zeros.  hipcc does not order these loads against
    // later LDS reads, hence the explicit vmcnt(0) at the top of every tile.
    const int row_bytes = g.D * 4;
    auto dma_tile = [&](long tile, float *slot) __attribute__((always_inline)) {
        const long t0 = tile << 5;
        const long left = a.T - t0;
        const char *base = reinterpret_cast<const char *>(a.X + t0 * (long)g.D);
        if (lane < g.D) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char *>(base + (long)r * row_bytes), 0, r < left ? row_bytes : 0, 0x00027000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + r * g.XS),
                                                         4, lane * 4, 0, 0, 0);
            }
        }
    };
    auto valid_in = [&](long t0) __attribute__((always_inline)) { return (int)((a.T - t0) < 32 ? (a.T - t0) : 32); };

    wave_lds_fence();
    if (ta < tb) dma_tile(ta, xs);
    WSet<NCT> wa;
    wset_load<NCT>(wa, Wg4, g.NG, g.NP, 0);

    const int frame = acc_row((c >> 1) & 15, h);
    bool first = true;
    int cur = 0;
    // as in k_estep_fat: the splat of the next tile's logits closes every tile, in front of the loop's exit edge
    f32x16 lg[NCT];
    auto logits_start = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float k = kcv[ct];
            asm volatile("" : "+v"(k));     // see k_estep_fat: keeps the splat inside the loop
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[ct][r] = k;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
    };
    for (long base = ta; base < tb; base += a.flush_tiles) {
    const long end = (base + a.flush_tiles < tb) ? base + a.flush_tiles : tb;
    logits_start();
#pragma unroll 1
    for (long tile = base; tile < end; ++tile) {
        const int nvalid = valid_in(tile << 5);
        const float *xt = xs + cur * tile_floats;
#ifndef SKM_LINT_DEMO_NO_DMA_WAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this tile's frames are in LDS
#endif

        gemm_logits_fat<NCT, GEO>(lg, wa, xt + c * g.XS, h, Wg4, g);

        // exact per-frame maximum over all components, e = 2^(l - max), per-frame sum of e
        float m[16];
        tile_row_max<NCT>(lg, m);
        lane_reduce16(m, c, OpMaxHw());
        if ((c & 1) == 0) ls[frame] = m[0];
        wave_lds_fence();
        float sm[16];
        tile_exp_sum<NCT>(lg, ls + 4 * h, sm);
        lane_reduce16(sm, c, OpAdd());
        if ((c & 1) == 0) {
            const bool valid = frame < nvalid;
            ls[32 + frame] = valid ? __builtin_amdgcn_rcpf(sm[0]) : 0.f;
            if (valid) llk_lane += (double)(m[0] + __builtin_amdgcn_logf(sm[0]));
        }
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 f4 = *reinterpret_cast<const f32x4 *>(ls + 32 + 8 * q4 + 4 * h);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                lg[ct][4 * q4 + 0] *= f4[0];
                lg[ct][4 * q4 + 1] *= f4[1];
                lg[ct][4 * q4 + 2] *= f4[2];
                lg[ct][4 * q4 + 3] *= f4[3];
            }
        }
        // the next tile is requested in front of the statistics GEMM, which has no vector-memory wait; it goes to the
        // other slot, whose last reader was the previous tile's statistics GEMM
        if (tile + 1 < tb) dma_tile(tile + 1, xs + (cur ^ 1) * tile_floats);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
        gemm_stats_agpr<NJT, NCT, true, VACC>(acc, lg, xt + 4 * h * g.XS + c, g.XS);
        logits_start();
        wave_lds_fence();
        cur ^= 1;
    }
    if (first)
        flush_slab_agpr<NCT, NF, true, VACC>(acc, slab_lane);
    else
        flush_slab_agpr<NCT, NF, false, VACC>(acc, slab_lane);
    first = false;
    }
    if (first) flush_slab_agpr<NCT, NF, true, VACC>(acc, slab_lane);   // empty range: the slab must still read as zero
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) llk_lane += __shfl_xor(llk_lane, msk);
    if (lane == 0) a.llk_part[blockIdx.x] = llk_lane * 0.6931471805599453094;
}

// ------------------------------------------------------------------------------------------------
// Single-sweep E-step for C > 512: B = C/512 workgroups (a "team", one per CU, same XCD) cover the
// components of the SAME frame tiles, each exactly like k_estep_fat (4 waves x 4 component tiles), and
// exchange their per-frame (max, sum) partials through a small buffer in global memory instead of
// recomputing the logits in a second sweep (124 instead of 184 MFMAs per 32x32 logits tile).
//
// Pipeline per workgroup, one tile late on the consumer side so the exchange latency never shows:
//   iteration i:  logits(i) -> e = 2^(l - m_wave), publish (m_wave, s_wave) of tile i, stash e(i) in LDS
//                 barrier (frame-tile ring only)
//                 collect the 4*B partials of tile i-1 (published one whole logits phase ago),
//                 gamma(i-1) = e(i-1) * 2^(m_wave - m) / s, statistics GEMM of tile i-1
// Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility): every partial is stored and loaded
// with agent-scope (sc1) accesses; the producer drains them (release fence) before it stores the flag
// = tile sequence number; consumers poll the flags with agent-scope loads.  Four slots: a producer may be
// two tiles ahead of the slowest consumer of its team.  The kernel is launched cooperatively (all
// workgroups co-resident or the launch fails and the host falls back to two sweeps); every poll loop is
// bounded and raises *err instead of spinning forever.
// ------------------------------------------------------------------------------------------------
// Baum-Welch flush of AGPR accumulators (one store per session, see flush_session).  Rows at or beyond C are
// padding components (kc = -1e30, gamma = 0) of a model that does not fill its last component block.
template <int NCT, int NF>
__device__ __forceinline__ void flush_session_agpr(f32x16 (&acc)[NCT][NF], double *s0, double *s1, int cb, int D,
                                                   int lane, bool add, int C)
{
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf) {
            agpr_settle(acc[ct][jf]);
            const f32x16 v = acc[ct][jf];
            const int col = jf * 32 + c;
            const int comp0 = (cb * NCT + ct) * 32 + 4 * h;
            double *p = (col < D) ? (s1 + (long)comp0 * D + col) : (s0 + comp0);
            const long stride = (col < D) ? D : 1;
            if (col <= D && comp0 + 27 < C) {      // every row of this lane is a live component
                if (!add) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[acc_row(r, 0) * stride] = (double)v[r];
                } else {   // later fold of a long session: this lane is the only writer of these entries
                    double old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = p[acc_row(r, 0) * stride];
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[acc_row(r, 0) * stride] = old[r] + (double)v[r];
                }
            } else if (col <= D && comp0 < C) {    // the model ends inside this component tile
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (comp0 + acc_row(r, 0) < C) {
                        double *q = p + acc_row(r, 0) * stride;
                        *q = (add ? *q : 0.0) + (double)v[r];
                    }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;
            // the next asm MFMA on this tile may follow at once (a session that ends inside a tile: flush, then the rest of
            // the tile): a vector write of the accumulator needs wait states before an MFMA reads it as its C operand, and
            // the compiler places none in front of asm (it also may sink the sixteen writes right in front of that asm)
            asm volatile("s_nop 3" : "+a"(acc[ct][jf]));
            __builtin_amdgcn_sched_barrier(0);
        }
}

// Cooperative staging of the contiguous 32 * D span of a frame tile (PRE floats per thread, 32 * D <= PRE * threads).
// Rows of a multiple of 4 floats move as 16-byte pieces (a piece never straddles two rows).  GENERAL adds the path
// for any D: single floats, row / column of every element fixed at init (the instantiations with a run-time logits
// geometry carry it; the others are only launched with 16-byte rows).
template <int PRE, bool GENERAL = false>
struct TileStager {
    static constexpr int NST = PRE / 4, NIX = GENERAL ? PRE : NST;
    float pre[PRE];
    int dst[NIX], row[NIX];
    bool vec4;
    __device__ __forceinline__ void init(const TileGeom &g, int tid, int nthr)
    {
        vec4 = !GENERAL || g.vec4;
        if (vec4) {
            const int n4 = g.D >> 2;
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = tid + i * nthr, r = e / n4;
                row[i] = (e < 32 * n4) ? r : 1 << 20;
                dst[i] = r * g.XS + 4 * (e - r * n4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NIX; ++i) {
                const int e = tid + i * nthr, r = e / g.D;
                row[i] = (e < 32 * g.D) ? r : 1 << 20;
                dst[i] = r * g.XS + (e - r * g.D);
            }
        }
    }
    __device__ __forceinline__ void load(const float *__restrict__ X, long t0, int nvalid, int D)
    {
        const int tid = threadIdx.x, nthr = blockDim.x;
        if (vec4) {
            const f32x4 *X4 = reinterpret_cast<const f32x4 *>(X + t0 * (long)D);
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row[i] < nvalid) v = X4[tid + i * nthr];
                pre[4 * i + 0] = v[0];
                pre[4 * i + 1] = v[1];
                pre[4 * i + 2] = v[2];
                pre[4 * i + 3] = v[3];
            }
        } else {
            const float *Xs = X + t0 * (long)D;
#pragma unroll
            for (int i = 0; i < NIX; ++i) pre[i] = (row[i] < nvalid) ? Xs[tid + i * nthr] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float *xs_tile) const
    {
        if (vec4) {
#pragma unroll
            for (int i = 0; i < NST; ++i)
                if (row[i] < 32) {
                    const f32x4 v = {pre[4 * i], pre[4 * i + 1], pre[4 * i + 2], pre[4 * i + 3]};
                    *reinterpret_cast<f32x4 *>(xs_tile + dst[i]) = v;
                }
        } else {
#pragma unroll
            for (int i = 0; i < NIX; ++i)
                if (row[i] < 32) xs_tile[dst[i]] = pre[i];
        }
    }
};

struct TeamArgs {
    const float *X;
    long T;
    TileGeom g;
    const float *wp;
    const float *kc;
    double *slabs;        // [team][member][wave] slabs of [128][NF*32] f64 (zeroed by the host)
    double *llk_part;     // [n_teams]
    float *xch;           // [team][4 slots][4*B contributions][32 frames][2] (max, sum)
    unsigned *flags;      // [team][4 slots][4*B]  tile sequence number + 1 (zeroed by the host)
    int *err;             // set to 1 when a bounded wait ran out
    int B, teams_per_xcd; // members per team; teams whose members share blockIdx % 8
    // Baum-Welch mode: sessions [seg_off[s], seg_off[s+1]) -> stat0[S][C], stat1[S][C*D]
    const int64_t *seg_off;
    int S, C;
    double *stat0, *stat1;
    int w_bytes;          // size of the packed W image
    int debug_drop;       // tests only (SKM_DEBUG_TEAM_DROP=1, Baum-Welch mode): contribution 1 never raises its flag
#ifdef SKM_TRACE
    long long *trace;
#endif
};

#define SKM_TEAM_SPIN_LIMIT (1 << 20)

// EM statistics: the team's tiles in one run, slabs folded every SKM_FLUSH_TILES tiles.  (k_seg_team below is
// the same pipeline cut into segments with a flush per segment -- one session each -- for Baum-Welch; its
// EM instantiation allocates registers worse than this dedicated kernel, 50 vs 36 ms at C=2048, so both stay.)
template <int NCT, int NJT, int B, int GEO, bool GEN = (GEO < 0 && NJT <= 2)>   // GEN: rows of any length (TileStager)
__global__ __launch_bounds__(256, 1) void k_estep_team(TeamArgs a)
{
    constexpr int NF = 2 * NJT, NFW = NF * 32, PRE = (NJT >= 3) ? 20 : 8, RING = 4;   // PRE: see k_seg_team
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), NWV = 4;
    const int h = lane >> 5, c = lane & 31;
    const TileGeom g = a.g;
    const int tile_floats = 32 * g.XS;
    constexpr int NC = 4 * B;

    // members of a team share blockIdx % 8 (blocks are dealt round-robin over the 8 XCDs: same L2)
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int member = kk % B, team = xcd * a.teams_per_xcd + kk / B, n_teams = 8 * a.teams_per_xcd;
    const int me = member * 4 + wave;   // this wave's contribution index = its 128-component block

    float *xs = smem;                                            // [RING][32][XS]
    float *ls = xs + RING * tile_floats + wave * 32;             // wave-private row of 32
    f32x4 *stash = reinterpret_cast<f32x4 *>(xs + RING * tile_floats + NWV * 32) + wave * (NCT * 4 * 64) + lane;

    {
        const int padw = g.XS - g.D;
        for (int i = tid; i < RING * 32 * padw; i += nthr) {
            const int row = i / padw, col = g.D + (i - row * padw);
            xs[row * g.XS + col] = (col == g.D) ? 1.0f : 0.0f;
        }
    }
    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[(me * NCT + ct) * 32 + c];
    WStream Wg4;
    Wg4.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, a.w_bytes, 0x00027000);
    Wg4.voff = lane * 16;
    Wg4.sbase = me * (NCT * g.NG * 1024);

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    const long tiles = (a.T + 31) >> 5;
    const long ta = tiles * team / n_teams, tb = tiles * (team + 1) / n_teams;
    double *slab_lane = a.slabs + ((long)(team * B + member) * NWV + wave) * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
    double llk_lane = 0.0;
    float *xch = a.xch + (long)team * (4 * NC * 64);
    unsigned *flags = a.flags + (long)team * (4 * NC);
    bool dead = false;
    SKM_TRACE_DECL;

    TileStager<PRE, GEN> stg;
    stg.init(g, tid, nthr);
    if (ta < tb) {
        const long t0 = ta << 5;
        stg.load(a.X, t0, (int)((a.T - t0) < 32 ? (a.T - t0) : 32), g.D);
        stg.store(xs);
        if (ta + 1 < tb) stg.load(a.X, t0 + 32, (int)((a.T - t0 - 32) < 32 ? (a.T - t0 - 32) : 32), g.D);
    }
    WSet<NCT> wa;
    wset_load<NCT>(wa, Wg4, g.NG, g.NP, 0);
    __syncthreads();

    // The partial rows of a tile are fetched in two steps that are issued early and consumed late:
    // poll_issue (one agent-scope load of the team's flags, behind the logits GEMM), collect (the 8-byte
    // (max, sum) pairs, requested before the LDS stash swap and the barrier), finish_tile (uses them).
    constexpr int KH = NC / 2;   // contributions per lane half
    unsigned long long part[KH];
    unsigned flag_seen = 0;
    auto poll_issue = [&](long tile) __attribute__((always_inline)) {
        const int slot = (int)((tile - ta) & 3);
        flag_seen = 0;
        if (lane < NC) flag_seen = __hip_atomic_load(flags + slot * NC + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto collect = [&](long tile) __attribute__((always_inline)) {
        const int slot = (int)((tile - ta) & 3);
        const unsigned expect = (unsigned)(tile - ta) + 1u;
        const unsigned *fl = flags + slot * NC;
        int spins = 0;
        unsigned v = (lane < NC) ? flag_seen : expect;
        while (!dead && __builtin_amdgcn_ballot_w64(v != expect) != 0) {
            if (++spins > SKM_TEAM_SPIN_LIMIT) {
                dead = true;
                if (lane == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_sleep(8);
            if (lane < NC) v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" ::: "memory");
#ifdef SKM_TRACE
        if (trace_now) a.trace[wave * 16 + 12] = 1000000 + spins;
#endif
        // lane (c, h) takes the contributions k = h, h+2, ... of frame c: one base address, constant strides
        const unsigned long long *row =
            reinterpret_cast<const unsigned long long *>(xch) + (long)slot * NC * 32 + h * 32 + c;
#pragma unroll
        for (int i = 0; i < KH; ++i) part[i] = __hip_atomic_load(row + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // gamma and statistics of tile `tile` (its e values are in lg, its partial rows in part[])
    // Flag of a published row (the sc1 stores of the row are complete at the coherence point once acknowledged, so
    // draining them is all the release this hand-off needs; an agent-scope release fence would write back the whole
    // L2 every tile).  flag_val = 0: nothing to publish.
    auto publish_flag = [&](int flag_slot, unsigned flag_val) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(flags + flag_slot * NC + me, flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto finish_tile = [&](f32x16 (&lg)[NCT], long tile, int flag_slot, unsigned flag_val) __attribute__((always_inline)) {
        const long t0 = tile << 5;
        const int nvalid = (int)((a.T - t0) < 32 ? (a.T - t0) : 32);
        float mx = -3.0e38f, sum = 0.f, m_me = 0.f;
        {
#pragma unroll
            for (int i = 0; i < KH; ++i) {
                const float pm = __builtin_bit_cast(float, (unsigned)(part[i] & 0xffffffffull));
                mx = __builtin_fmaxf(mx, pm);
                if (2 * i == (me & ~1)) m_me = pm;   // the half h == (me & 1) holds this wave's own row
            }
#pragma unroll
            for (int i = 0; i < KH; ++i) {           // independent exponentials, no serial rescaling chain
                const float pm = __builtin_bit_cast(float, (unsigned)(part[i] & 0xffffffffull));
                const float ps = __builtin_bit_cast(float, (unsigned)(part[i] >> 32));
                sum += ps * __builtin_amdgcn_exp2f(pm - mx);
            }
            const float omx = __shfl_xor(mx, 32), osum = __shfl_xor(sum, 32), ome = __shfl_xor(m_me, 32);
            const float nm = __builtin_fmaxf(mx, omx);
            sum = sum * __builtin_amdgcn_exp2f(mx - nm) + osum * __builtin_amdgcn_exp2f(omx - nm);
            mx = nm;
            if ((me & 1) != h) m_me = ome;
        }
        {
            const bool valid = c < nvalid;
            const float f = __builtin_amdgcn_exp2f(m_me - mx) * __builtin_amdgcn_rcpf(sum);
            if (h == 0) ls[c] = valid ? f : 0.f;
            if (me == 0 && h == 0 && valid) llk_lane += (double)(mx + __builtin_amdgcn_logf(sum));
        }
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 f4 = *reinterpret_cast<const f32x4 *>(ls + 8 * q4 + 4 * h);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                lg[ct][4 * q4 + 0] *= f4[0];
                lg[ct][4 * q4 + 1] *= f4[1];
                lg[ct][4 * q4 + 2] *= f4[2];
                lg[ct][4 * q4 + 3] *= f4[3];
            }
        }
        // The flag of the row published in THIS iteration goes out here: its stores are a barrier and the gamma
        // arithmetic old, so the drain no longer stalls on their acknowledgement (~1 k cycles per tile when it sat
        // right behind them), no vector load of this iteration is pending yet, and no control flow follows the
        // statistics MFMAs.  The consumers look at the flag a logits GEMM later.
        if (flag_val) publish_flag(flag_slot, flag_val);
        // frames of tile+3 are requested here (the statistics GEMM has no other vector-memory wait)
        if (tile + 3 < tb) {
            const long t3 = (tile + 3) << 5;
            stg.load(a.X, t3, (int)((a.T - t3) < 32 ? (a.T - t3) : 32), g.D);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
        const float *xt = xs + (int)((tile - ta) & (RING - 1)) * tile_floats;
        SKM_STAMP(5);
        gemm_stats_agpr<NJT, NCT>(acc, lg, xt + 4 * h * g.XS + c, g.XS);
        wave_lds_fence();
    };

    for (long base = ta; base < tb; base += SKM_FLUSH_TILES) {
        const long end = (base + SKM_FLUSH_TILES < tb) ? base + SKM_FLUSH_TILES : tb;
#pragma unroll 1
        for (long tile = base; tile < end; ++tile) {
            const int cur = (int)((tile - ta) & (RING - 1)), nxt = (cur + 1) & (RING - 1);
            const int slot = (int)((tile - ta) & 3);
            const float *xt = xs + cur * tile_floats;
            SKM_TRACE_TILE;
            SKM_STAMP(0);

            f32x16 lg[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float k = kcv[ct];
                asm volatile("" : "+v"(k));
#pragma unroll
                for (int r = 0; r < 16; ++r) lg[ct][r] = k;
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
            gemm_logits_fat<NCT, GEO>(lg, wa, xt + c * g.XS, h, Wg4, g);
            SKM_STAMP(1);
            if (tile > ta) poll_issue(tile - 1);

            // exact per-frame maximum over this wave's 128 components, e = 2^(l - max), per-frame sum of e
            float m[16];
            tile_row_max<NCT>(lg, m);
            lane_reduce16(m, c, OpMaxHw());
            const int frame = acc_row((c >> 1) & 15, h);
            if ((c & 1) == 0) ls[frame] = m[0];
            wave_lds_fence();
            float sm[16];
            tile_exp_sum<NCT>(lg, ls + 4 * h, sm);
            lane_reduce16(sm, c, OpAdd());
            // publish this wave's row of partials, then its flag
            if ((c & 1) == 0) {
                unsigned long long *row = reinterpret_cast<unsigned long long *>(xch) + ((long)slot * NC + me) * 32 + frame;
                const unsigned long long pr = (unsigned long long)__builtin_bit_cast(unsigned, m[0]) |
                                              ((unsigned long long)__builtin_bit_cast(unsigned, sm[0]) << 32);
                __hip_atomic_store(row, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_lds_fence();   // ls is rewritten by finish_tile
            SKM_STAMP(2);

            // e(tile) waits in the wave's LDS stash while e(tile-1) comes back out
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (tile > ta) o = stash[(ct * 4 + q) * 64];
                    const f32x4 n = {lg[ct][4 * q], lg[ct][4 * q + 1], lg[ct][4 * q + 2], lg[ct][4 * q + 3]};
                    stash[(ct * 4 + q) * 64] = n;
                    lg[ct][4 * q] = o[0];
                    lg[ct][4 * q + 1] = o[1];
                    lg[ct][4 * q + 2] = o[2];
                    lg[ct][4 * q + 3] = o[3];
                }
            if (tile + 1 < tb) stg.store(xs + nxt * tile_floats);
            // tile+2 must be in registers before finish_tile requests tile+3 ... on the first tile only
            // tile+1 has been requested so far
            if (tile == ta && tile + 2 < tb) {
                const long t2 = (tile + 2) << 5;
                stg.load(a.X, t2, (int)((a.T - t2) < 32 ? (a.T - t2) : 32), g.D);
            }
            // the first tile has no finish_tile to carry its flag, the last one's consumers read it right after the loop
            const bool early_flag = tile == ta || tile + 1 == tb;
            if (early_flag) publish_flag(slot, (unsigned)(tile - ta) + 1u);
            SKM_STAMP(3);
            if (tile > ta) collect(tile - 1);   // the rows of tile-1 travel while the workgroup meets at the barrier
            __syncthreads();
            SKM_STAMP(4);
            if (tile > ta) finish_tile(lg, tile - 1, slot, early_flag ? 0u : (unsigned)(tile - ta) + 1u);
            SKM_STAMP(6);
        }
        // the accumulators are touched by nothing but the asm MFMAs inside the tile loop: they stay in AGPRs
        if (end == tb && ta < tb) {
            f32x16 lg[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 o = stash[(ct * 4 + q) * 64];
                    lg[ct][4 * q] = o[0];
                    lg[ct][4 * q + 1] = o[1];
                    lg[ct][4 * q + 2] = o[2];
                    lg[ct][4 * q + 3] = o[3];
                }
            poll_issue(tb - 1);
            collect(tb - 1);
            finish_tile(lg, tb - 1, 0, 0u);
        }
        flush_slab_agpr<NCT, NF>(acc, slab_lane);
    }
    if (me == 0) {
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) llk_lane += __shfl_xor(llk_lane, msk);
        if (lane == 0) a.llk_part[team] = llk_lane * 0.6931471805599453094;
    }
}

// Baum-Welch statistics of the frames [lo, hi) of a tile only (a tile that holds the end of one session and the start
// of the next): the B operand -- lane (c, h) holds x[acc_row(r, h)][c] at step r -- is zeroed outside the range, gamma
// stays as it is.  No S accumulators.
template <int NJT, int NCT>
__device__ __forceinline__ void gemm_stats_agpr_masked(f32x16 (&acc)[NCT][NJT], const f32x16 (&lg)[NCT], const float *xcol,
                                                       int XS, int lo, int hi, int h)
{
    float bn[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) bn[jt] = xcol[jt * 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int fr = acc_row(r, 0) + 4 * h;
        const bool in = fr >= lo && fr < hi;
        float b[NJT];
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) b[jt] = in ? bn[jt] : 0.f;
        if (r < 15) {
            const float *xrow = xcol + acc_row(r + 1, 0) * XS;
#pragma unroll
            for (int jt = 0; jt < NJT; ++jt) bn[jt] = xrow[jt * 32];
        }
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (ct == 0)
                    mfma_agpr_after_valu(acc[ct][jt], lg[ct][r], b[jt]);
                else
                    mfma_agpr(acc[ct][jt], lg[ct][r], b[jt]);
            }
    }
}

// NWV = 2 (round 3): a workgroup of TWO 512-register waves for models of one or two component tiles (the reference's own
// UBM: 64 components) -- two such workgroups share a CU, one wave per SIMD, where the four-wave workgroup left two
// waves on padding tiles.  Its frame ring has three tiles (a team of one workgroup has no partner that runs ahead), so
// that two workgroups' rings fit the LDS.
template <int NCT, int NJT, int B, bool BW, int GEO, bool GEN = (GEO < 0 && NJT <= 2), int NWV = 4>
__global__ __launch_bounds__(64 * NWV, 1) void k_seg_team(TeamArgs a)
{
    // PRE floats of a frame tile staged per thread: 32 * D <= PRE * threads (rows of up to 160 floats at NJT >= 3)
    constexpr int NF = BW ? NJT : 2 * NJT, NFW = NF * 32, PRE = ((NJT >= 3) ? 20 : 8) * (4 / NWV), RING = (NWV == 2) ? 3 : 4,
                  NC = NWV * B;
    static_assert(NWV == 4 || (NWV == 2 && B == 1), "two-wave workgroups are teams of one");
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, c = lane & 31;
    const TileGeom g = a.g;
    const int tile_floats = 32 * g.XS;

    // members of a team share blockIdx % 8 (blocks are dealt round-robin over the 8 XCDs: same L2)
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int member = kk % B, team = xcd * a.teams_per_xcd + kk / B, n_teams = 8 * a.teams_per_xcd;
    const int me = member * NWV + wave;   // this wave's contribution index = its block of NCT component tiles

    float *xs = smem;                                            // [RING][32][XS]
    float *ls = xs + RING * tile_floats + wave * 32;             // wave-private row of 32
    f32x4 *stash = reinterpret_cast<f32x4 *>(xs + RING * tile_floats + NWV * 32) + wave * (NCT * 4 * 64) + lane;

    {
        const int padw = g.XS - g.D;
        for (int i = tid; i < RING * 32 * padw; i += nthr) {
            const int row = i / padw, col = g.D + (i - row * padw);
            xs[row * g.XS + col] = (col == g.D) ? 1.0f : 0.0f;
        }
    }
    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[(me * NCT + ct) * 32 + c];
    WStream Wg4;
    Wg4.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, a.w_bytes, 0x00027000);
    Wg4.voff = lane * 16;
    Wg4.sbase = me * (NCT * g.NG * 1024);

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    double llk_lane = 0.0;
    float *xch = a.xch + (long)team * (4 * NC * 64);
    unsigned *flags = a.flags + (long)team * (4 * NC);
    bool dead = false;
    unsigned n = 0;   // tiles this team has started: slot = n & 3, flag value = n + 1 (same on every member)
    SKM_TRACE_DECL;

    TileStager<PRE, GEN> stg;
    stg.init(g, tid, nthr);
    WSet<NCT> wa;
    wset_load<NCT>(wa, Wg4, g.NG, g.NP, 0);

    // The partial rows of a tile are fetched in two steps that are issued early and consumed late:
    // poll_issue (one agent-scope load of the team's flags, behind the logits GEMM), collect (the 8-byte
    // (max, sum) pairs, requested right before the barrier), finish_tile (uses them).
    constexpr int KH = NC / 2;   // contributions per lane half
    unsigned long long part[KH];
    unsigned flag_seen = 0;
    auto poll_issue = [&](unsigned ni) __attribute__((always_inline)) {
        flag_seen = 0;
        if (lane < NC)
            flag_seen = __hip_atomic_load(flags + (ni & 3) * NC + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto collect = [&](unsigned ni) __attribute__((always_inline)) {
        const int slot = (int)(ni & 3);
        const unsigned expect = ni + 1u;
        const unsigned *fl = flags + slot * NC;
        int spins = 0;
        unsigned v = (lane < NC) ? flag_seen : expect;
        while (!dead && __builtin_amdgcn_ballot_w64(v != expect) != 0) {
            if (++spins > SKM_TEAM_SPIN_LIMIT) {
                dead = true;
                if (lane == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_sleep(8);
            if (lane < NC) v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" ::: "memory");
        // lane (c, h) takes the contributions k = h, h+2, ... of frame c: one base address, constant strides
        const unsigned long long *row =
            reinterpret_cast<const unsigned long long *>(xch) + (long)slot * NC * 32 + h * 32 + c;
#pragma unroll
        for (int i = 0; i < KH; ++i) part[i] = __hip_atomic_load(row + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // gamma and statistics of tile ni (its e values are in lg, its partial rows in part[]); t_pf >= 0 names
    // the frames to request for a later tile (the statistics GEMM has no other vector-memory wait)
    // flag of a published row, as in k_estep_team (flag_val = 0: nothing to publish)
    const bool muted = BW && a.debug_drop && me == 1;   // wave-uniform, tests only
    auto publish_flag = [&](int flag_slot, unsigned flag_val) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && !muted)
            __hip_atomic_store(flags + flag_slot * NC + me, flag_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // The tile loop is left and re-entered at session ends, and the instantiations with few free registers spill: whatever
    // copy or spill of an accumulator the compiler places behind a statistics GEMM must not read it while the last asm
    // MFMA is in flight (the hazard recogniser cannot see inside asm)
    auto settle_acc = [&]() __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0][0]));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int jf = 0; jf < NF; ++jf) asm volatile("" : "+a"(acc[ct][jf]));
    };
    auto finish_tile = [&](f32x16 (&lg)[NCT], unsigned ni, int nvalid, long t_pf, int nv_pf, int flag_slot,
                           unsigned flag_val, int hi_first = 32) __attribute__((always_inline)) {
        float mx = -3.0e38f, sum = 0.f, m_me = 0.f;
        {
#pragma unroll
            for (int i = 0; i < KH; ++i) {
                const float pm = __builtin_bit_cast(float, (unsigned)(part[i] & 0xffffffffull));
                mx = __builtin_fmaxf(mx, pm);
                if (2 * i == (me & ~1)) m_me = pm;   // the half h == (me & 1) holds this wave's own row
            }
#pragma unroll
            for (int i = 0; i < KH; ++i) {           // independent exponentials, no serial rescaling chain
                const float pm = __builtin_bit_cast(float, (unsigned)(part[i] & 0xffffffffull));
                const float ps = __builtin_bit_cast(float, (unsigned)(part[i] >> 32));
                sum += ps * __builtin_amdgcn_exp2f(pm - mx);
            }
            const float omx = __shfl_xor(mx, 32), osum = __shfl_xor(sum, 32), ome = __shfl_xor(m_me, 32);
            const float nm = __builtin_fmaxf(mx, omx);
            sum = sum * __builtin_amdgcn_exp2f(mx - nm) + osum * __builtin_amdgcn_exp2f(omx - nm);
            mx = nm;
            if ((me & 1) != h) m_me = ome;
        }
        {
            const bool valid = c < nvalid;
            const float f = __builtin_amdgcn_exp2f(m_me - mx) * __builtin_amdgcn_rcpf(sum);
            if (h == 0) ls[c] = valid ? f : 0.f;
            if (!BW && me == 0 && h == 0 && valid) llk_lane += (double)(mx + __builtin_amdgcn_logf(sum));
        }
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 f4 = *reinterpret_cast<const f32x4 *>(ls + 8 * q4 + 4 * h);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                lg[ct][4 * q4 + 0] *= f4[0];
                lg[ct][4 * q4 + 1] *= f4[1];
                lg[ct][4 * q4 + 2] *= f4[2];
                lg[ct][4 * q4 + 3] *= f4[3];
            }
        }
        if (flag_val) publish_flag(flag_slot, flag_val);   // see k_estep_team: the drain is free here
        if (t_pf >= 0) stg.load(a.X, t_pf, nv_pf, g.D);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
        const float *xt = xs + (int)(ni % RING) * tile_floats;
        SKM_STAMP(5);
        if constexpr (BW) {     // the frames [0, hi_first) of the tile: the whole tile unless a session ends inside it
            gemm_stats_agpr_masked<NJT, NCT>(acc, lg, xt + 4 * h * g.XS + c, g.XS, 0, hi_first, h);
            settle_acc();
        } else {
            gemm_stats_agpr<NJT, NCT, true>(acc, lg, xt + 4 * h * g.XS + c, g.XS);
        }
        wave_lds_fence();
    };

    // One segment = frames [b, e) whose statistics go to one flush (EM: 256 tiles, BW: one session).  The
    // pipeline is filled and drained per segment; the accumulators are touched by nothing but the asm MFMAs
    // inside the tile loop, so they stay in AGPRs between the flushes.
    auto run_segment = [&](long b, long e) __attribute__((always_inline)) {
        const int ntile = (int)((e - b + 31) >> 5);
        if (ntile == 0) return;
        auto nv = [&](int i) __attribute__((always_inline)) {
            const long left = e - (b + 32L * i);
            return (int)(left < 32 ? left : 32);
        };
        stg.load(a.X, b, nv(0), g.D);
        stg.store(xs + (int)(n % RING) * tile_floats);
        if (ntile > 1) stg.load(a.X, b + 32, nv(1), g.D);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < ntile; ++i) {
            const unsigned ni = n + (unsigned)i;
            const int cur = (int)(ni % RING), nxt = (cur + 1) % RING, slot = (int)(ni & 3);
            const float *xt = xs + cur * tile_floats;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int jf = 0; jf < NF; ++jf) asm volatile("" : "+a"(acc[ct][jf]));
#ifdef SKM_TRACE
            trace_now = a.trace && blockIdx.x == 8 && lane == 0 && ni == 40;
#endif
            SKM_STAMP(0);

            f32x16 lg[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float k = kcv[ct];
                asm volatile("" : "+v"(k));
#pragma unroll
                for (int r = 0; r < 16; ++r) lg[ct][r] = k;
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
            gemm_logits_fat<NCT, GEO>(lg, wa, xt + c * g.XS, h, Wg4, g);
            SKM_STAMP(1);
            if (i > 0) poll_issue(ni - 1);

            // exact per-frame maximum over this wave's 128 components, e = 2^(l - max), per-frame sum of e
            float m[16];
            tile_row_max<NCT>(lg, m);
            lane_reduce16(m, c, OpMaxHw());
            const int frame = acc_row((c >> 1) & 15, h);
            if ((c & 1) == 0) ls[frame] = m[0];
            wave_lds_fence();
            float sm[16];
            tile_exp_sum<NCT>(lg, ls + 4 * h, sm);
            lane_reduce16(sm, c, OpAdd());
            // publish this wave's row of partials (the flag follows below, once the stores have drained)
            if ((c & 1) == 0) {
                unsigned long long *row = reinterpret_cast<unsigned long long *>(xch) + ((long)slot * NC + me) * 32 + frame;
                const unsigned long long pr = (unsigned long long)__builtin_bit_cast(unsigned, m[0]) |
                                              ((unsigned long long)__builtin_bit_cast(unsigned, sm[0]) << 32);
                __hip_atomic_store(row, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_lds_fence();   // ls is rewritten by finish_tile
            SKM_STAMP(2);

            // e(i) waits in the wave's LDS stash while e(i-1) comes back out
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (i > 0) o = stash[(ct * 4 + q) * 64];
                    const f32x4 nw = {lg[ct][4 * q], lg[ct][4 * q + 1], lg[ct][4 * q + 2], lg[ct][4 * q + 3]};
                    stash[(ct * 4 + q) * 64] = nw;
                    lg[ct][4 * q] = o[0];
                    lg[ct][4 * q + 1] = o[1];
                    lg[ct][4 * q + 2] = o[2];
                    lg[ct][4 * q + 3] = o[3];
                }
            // three-tile ring: the slot of tile i + 1 is the one the statistics GEMM of tile i - 2 read at the end of
            // the previous iteration -- every wave must have left it
            if constexpr (RING == 3) __syncthreads();
            if (i + 1 < ntile) stg.store(xs + nxt * tile_floats);
            // from here on finish_tile(i-1) requests tile i+2; the first iteration has no finish_tile
            if (i == 0 && ntile > 2) stg.load(a.X, b + 64, nv(2), g.D);
            // flag of this tile's row: carried by finish_tile(i-1) below unless this is the first or last tile
            const bool early_flag = i == 0 || i + 1 == ntile;
            if (early_flag) publish_flag(slot, ni + 1u);
            SKM_STAMP(3);
            if (i > 0) collect(ni - 1);   // the rows of tile i-1 travel while the workgroup meets at the barrier
            __syncthreads();
            SKM_STAMP(4);
            if (i > 0)
                finish_tile(lg, ni - 1, nv(i - 1), (i + 2 < ntile) ? b + 32L * (i + 2) : -1L, (i + 2 < ntile) ? nv(i + 2) : 0,
                            slot, early_flag ? 0u : ni + 1u);
            SKM_STAMP(6);
        }
        {   // drain: the last tile of the segment
            f32x16 lg[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 o = stash[(ct * 4 + q) * 64];
                    lg[ct][4 * q] = o[0];
                    lg[ct][4 * q + 1] = o[1];
                    lg[ct][4 * q + 2] = o[2];
                    lg[ct][4 * q + 3] = o[3];
                }
            const unsigned nl = n + (unsigned)ntile - 1u;
            poll_issue(nl);
            collect(nl);
            finish_tile(lg, nl, nv(ntile - 1), -1L, 0, 0, 0u);
        }
        n += (unsigned)ntile;
    };

    // Baum-Welch: the team walks the frames of its sessions as ONE stream of 32-frame tiles (no pipeline fill / drain and
    // no partial tile per session).  The tile loop is the EM loop; it is LEFT when the tile it has just finished holds the
    // end of a session (finish_tile took only that session's frames of the tile): the session's float64 rows are written
    // out here, outside the loop -- with the flush inside the loop hipcc shuffled the 128 accumulator registers at every
    // merge --, the rest of the tile goes to the next session(s) through the same masked GEMM, and the loop resumes.
    int cur_s = 0, cur_end_s = 0, acc_tiles = 0;
    long cur_e = 0;
    bool cur_add = false;
    auto close_rows = [&](bool add) __attribute__((always_inline)) {
        flush_session_agpr<NCT, NF>(acc, a.stat0 + (long)cur_s * a.C, a.stat1 + (long)cur_s * a.C * g.D, me, g.D, lane, add,
                                    a.C);
    };
    // after finish_tile(…, hi_first) of a tile [t0, t0 + nvalid) in which the current session ends at hi_first
    auto close_boundary = [&](f32x16 (&lg)[NCT], unsigned ni, long t0, int nvalid, int hi_first) __attribute__((always_inline)) {
        const float *xt = xs + (int)(ni % RING) * tile_floats;
        int lo = hi_first;
        close_rows(cur_add);
        cur_add = false;
        acc_tiles = 0;
        ++cur_s;
        if (cur_s < cur_end_s) cur_e = a.seg_off[cur_s + 1];
#pragma unroll 1
        while (lo < nvalid || (cur_s < cur_end_s && cur_e - t0 <= (long)nvalid)) {
            const bool ends = cur_s < cur_end_s && cur_e - t0 <= (long)nvalid;
            const int hi = ends ? (int)(cur_e - t0) : 32;
            if constexpr (BW) {
                if (hi > lo) {
                    gemm_stats_agpr_masked<NJT, NCT>(acc, lg, xt + 4 * h * g.XS + c, g.XS, lo, hi, h);
                    settle_acc();
                }
            }
            if (!ends) {
                acc_tiles = 1;
                break;
            }
            close_rows(false);
            lo = hi;
            ++cur_s;
            if (cur_s < cur_end_s) cur_e = a.seg_off[cur_s + 1];
        }
    };
    auto run_stream = [&](long b, long e) __attribute__((always_inline)) {
        const int ntile = (int)((e - b + 31) >> 5);
        if (ntile == 0) return;
        auto nv = [&](int i) __attribute__((always_inline)) {
            const long left = e - (b + 32L * i);
            return (int)(left < 32 ? left : 32);
        };
        // frames of the current session inside tile j (32: the session goes on beyond the tile)
        auto first_piece = [&](int j) __attribute__((always_inline)) {
            const long left = cur_e - (b + 32L * j);
            return (int)(left <= (long)nv(j) ? left : 32);
        };
        stg.load(a.X, b, nv(0), g.D);
        stg.store(xs + (int)(n % RING) * tile_floats);
        if (ntile > 1) stg.load(a.X, b + 32, nv(1), g.D);
        __syncthreads();
        f32x16 lg[NCT];
        int i = 0;
#pragma unroll 1
        while (i < ntile) {
            int stop = 0, hi_first = 32;      // 1: a session ended in tile i - 1, 2: fold of a long session
#pragma unroll 1
            for (; i < ntile && !stop; ++i) {
                const unsigned ni = n + (unsigned)i;
                const int cur = (int)(ni % RING), nxt = (cur + 1) % RING, slot = (int)(ni & 3);
                const float *xt = xs + cur * tile_floats;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int jf = 0; jf < NF; ++jf) asm volatile("" : "+a"(acc[ct][jf]));
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    float k = kcv[ct];
                    asm volatile("" : "+v"(k));
#pragma unroll
                    for (int r = 0; r < 16; ++r) lg[ct][r] = k;
                }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) valu_to_mfma(lg[ct]);
                gemm_logits_fat<NCT, GEO>(lg, wa, xt + c * g.XS, h, Wg4, g);
                if (i > 0) poll_issue(ni - 1);
                float m[16];
                tile_row_max<NCT>(lg, m);
                lane_reduce16(m, c, OpMaxHw());
                const int frame = acc_row((c >> 1) & 15, h);
                if ((c & 1) == 0) ls[frame] = m[0];
                wave_lds_fence();
                float sm[16];
                tile_exp_sum<NCT>(lg, ls + 4 * h, sm);
                lane_reduce16(sm, c, OpAdd());
                if ((c & 1) == 0) {
                    unsigned long long *row = reinterpret_cast<unsigned long long *>(xch) + ((long)slot * NC + me) * 32 + frame;
                    const unsigned long long pr = (unsigned long long)__builtin_bit_cast(unsigned, m[0]) |
                                                  ((unsigned long long)__builtin_bit_cast(unsigned, sm[0]) << 32);
                    __hip_atomic_store(row, pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wave_lds_fence();
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 o = {0.f, 0.f, 0.f, 0.f};
                        if (i > 0) o = stash[(ct * 4 + q) * 64];
                        const f32x4 nw = {lg[ct][4 * q], lg[ct][4 * q + 1], lg[ct][4 * q + 2], lg[ct][4 * q + 3]};
                        stash[(ct * 4 + q) * 64] = nw;
                        lg[ct][4 * q] = o[0];
                        lg[ct][4 * q + 1] = o[1];
                        lg[ct][4 * q + 2] = o[2];
                        lg[ct][4 * q + 3] = o[3];
                    }
                if constexpr (RING == 3) __syncthreads();     // see run_segment (also covers close_boundary's reads)
                if (i + 1 < ntile) stg.store(xs + nxt * tile_floats);
                if (i == 0 && ntile > 2) stg.load(a.X, b + 64, nv(2), g.D);
                const bool early_flag = i == 0 || i + 1 == ntile;
                if (early_flag) publish_flag(slot, ni + 1u);
                if (i > 0) collect(ni - 1);
                __syncthreads();
                if (i > 0) {
                    hi_first = first_piece(i - 1);
                    finish_tile(lg, ni - 1, nv(i - 1), (i + 2 < ntile) ? b + 32L * (i + 2) : -1L,
                                (i + 2 < ntile) ? nv(i + 2) : 0, slot, early_flag ? 0u : ni + 1u, hi_first);
                    if (hi_first < 32 || (long)hi_first == cur_e - (b + 32L * (i - 1)))
                        stop = 1;
                    else if (++acc_tiles >= SKM_FLUSH_TILES)
                        stop = 2;
                }
            }
            if (stop == 1) {
                close_boundary(lg, n + (unsigned)i - 2u, b + 32L * (i - 2), nv(i - 2), hi_first);
            } else if (stop == 2) {         // long session: fold into its float64 rows like the EM slabs
                close_rows(cur_add);
                cur_add = true;
                acc_tiles = 0;
            }
        }
        {   // drain: the last tile of the stream (the team's last session ends in it)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 o = stash[(ct * 4 + q) * 64];
                    lg[ct][4 * q] = o[0];
                    lg[ct][4 * q + 1] = o[1];
                    lg[ct][4 * q + 2] = o[2];
                    lg[ct][4 * q + 3] = o[3];
                }
            const unsigned nl = n + (unsigned)ntile - 1u;
            poll_issue(nl);
            collect(nl);
            const int hi_last = first_piece(ntile - 1);
            finish_tile(lg, nl, nv(ntile - 1), -1L, 0, 0, 0u, hi_last);
            close_boundary(lg, nl, b + 32L * (ntile - 1), nv(ntile - 1), hi_last);
        }
        n += (unsigned)ntile;
    };

    if (!BW) {
        const long tiles = (a.T + 31) >> 5;
        const long fa = (tiles * team / n_teams) << 5;
        long fb = (tiles * (team + 1) / n_teams) << 5;
        if (fb > a.T) fb = a.T;
        double *slab_lane =
            a.slabs + ((long)(team * B + member) * NWV + wave) * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
        for (long b = fa; b < fb; b += 32L * SKM_FLUSH_TILES) {
            const long e = (b + 32L * SKM_FLUSH_TILES < fb) ? b + 32L * SKM_FLUSH_TILES : fb;
            run_segment(b, e);
            flush_slab_agpr<NCT, NF>(acc, slab_lane);
        }
        if (me == 0) {
#pragma unroll
            for (int msk = 1; msk < 32; msk <<= 1) llk_lane += __shfl_xor(llk_lane, msk);
            if (lane == 0) a.llk_part[team] = llk_lane * 0.6931471805599453094;
        }
    } else {
        // sessions are cut into contiguous runs of about equal frame counts (lower bounds on seg_off)
        const int64_t *so = a.seg_off;
        const long total = so[a.S];
        auto lower = [&](long target) __attribute__((always_inline)) {
            int lo = 0, hi = a.S;   // first s in [0, S] with so[s] >= target
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (so[mid] < target) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        const int sa = (team == 0) ? 0 : lower(total * team / n_teams);
        const int sb = (team == n_teams - 1) ? a.S : lower(total * (team + 1) / n_teams);
        cur_s = sa;
        cur_end_s = sb;
        cur_e = (sa < sb) ? so[sa + 1] : 0;
        if constexpr (BW) {
            if (sa < sb) run_stream(so[sa], so[sb]);
        }
#pragma unroll 1
        for (; cur_s < sb; ++cur_s) close_rows(false);      // no frame at all in the team's range: empty sessions read zero
    }
}


// ------------------------------------------------------------------------------------------------
// instance lists of the team / segment kernels (X-macros)
// ------------------------------------------------------------------------------------------------
// k_estep_team<NCT, NJT, B, GEO[, GEN]>
#define SKM_ESTEP_TEAM_INSTANCES(X)                                                                              \
    X(4, 2, 2, 0) X(4, 2, 2, 1) X(4, 2, 2, 2) X(4, 2, 2, 3)                                                     \
    X(4, 2, 4, 0) X(4, 2, 4, 1) X(4, 2, 4, 2) X(4, 2, 4, 3)                                                     \
    X(4, 2, 8, 0) X(4, 2, 8, 1) X(4, 2, 8, 2) X(4, 2, 8, 3)                                                     \
    X(4, 1, 2, -1) X(4, 1, 4, -1) X(4, 1, 8, -1)                                                                \
    X(4, 2, 2, 0, true) X(4, 2, 2, 1, true) X(4, 2, 2, 2, true) X(4, 2, 2, 3, true)                             \
    X(4, 2, 4, 0, true) X(4, 2, 4, 1, true) X(4, 2, 4, 2, true) X(4, 2, 4, 3, true)                             \
    X(4, 2, 8, 0, true) X(4, 2, 8, 1, true) X(4, 2, 8, 2, true) X(4, 2, 8, 3, true)                             \
    X(1, 3, 2, -1) X(1, 3, 4, -1) X(1, 3, 8, -1) X(1, 3, 16, -1)                                                \
    X(1, 4, 2, -1) X(1, 4, 4, -1) X(1, 4, 8, -1) X(1, 4, 16, -1)                                                \
    X(1, 5, 2, -1) X(1, 5, 4, -1) X(1, 5, 8, -1) X(1, 5, 16, -1)
// k_seg_team<NCT, NJT, B, true, GEO[, GEN]>: four component tiles per wave (C in blocks of 512)
#define SKM_SEG_TEAM_INSTANCES_A(X)                                                                              \
    X(4, 2, 1, true, 0) X(4, 2, 1, true, 1) X(4, 2, 1, true, 2) X(4, 2, 1, true, 3)                             \
    X(4, 2, 2, true, 0) X(4, 2, 2, true, 1) X(4, 2, 2, true, 2) X(4, 2, 2, true, 3)                             \
    X(4, 2, 4, true, 0) X(4, 2, 4, true, 1) X(4, 2, 4, true, 2) X(4, 2, 4, true, 3)                             \
    X(4, 2, 8, true, 0) X(4, 2, 8, true, 1) X(4, 2, 8, true, 2) X(4, 2, 8, true, 3)                             \
    X(4, 1, 1, true, -1) X(4, 1, 2, true, -1) X(4, 1, 4, true, -1) X(4, 1, 8, true, -1)
// ... rows of any length (float-by-float stager), and one or two component tiles per wave (small models, wide rows)
#define SKM_SEG_TEAM_INSTANCES_B(X)                                                                              \
    X(4, 2, 1, true, 0, true) X(4, 2, 1, true, 1, true) X(4, 2, 1, true, 2, true) X(4, 2, 1, true, 3, true)     \
    X(4, 2, 2, true, 0, true) X(4, 2, 2, true, 1, true) X(4, 2, 2, true, 2, true) X(4, 2, 2, true, 3, true)     \
    X(4, 2, 4, true, 0, true) X(4, 2, 4, true, 1, true) X(4, 2, 4, true, 2, true) X(4, 2, 4, true, 3, true)     \
    X(4, 2, 8, true, 0, true) X(4, 2, 8, true, 1, true) X(4, 2, 8, true, 2, true) X(4, 2, 8, true, 3, true)
#define SKM_SEG_TEAM_INSTANCES_C(X)                                                                              \
    X(1, 1, 1, true, -1) X(1, 2, 1, true, -1) X(1, 3, 1, true, -1) X(1, 4, 1, true, -1) X(1, 5, 1, true, -1)    \
    X(1, 1, 1, true, -1, true, 2) X(1, 2, 1, true, -1, true, 2) X(1, 3, 1, true, -1, false, 2)                   \
    X(1, 4, 1, true, -1, false, 2) X(1, 5, 1, true, -1, false, 2)                                               \
    X(2, 1, 1, true, -1) X(2, 2, 1, true, -1)                                                                   \
    X(2, 3, 1, true, -1) X(2, 3, 2, true, -1) X(2, 3, 4, true, -1) X(2, 3, 8, true, -1)                         \
    X(2, 4, 1, true, -1) X(2, 4, 2, true, -1) X(2, 4, 4, true, -1) X(2, 4, 8, true, -1)                         \
    X(2, 5, 1, true, -1) X(2, 5, 2, true, -1) X(2, 5, 4, true, -1) X(2, 5, 8, true, -1)
#define SKM_DECLARE_ESTEP_TEAM(...) extern template __global__ void k_estep_team<__VA_ARGS__>(TeamArgs);
#define SKM_DEFINE_ESTEP_TEAM(...) template __global__ void k_estep_team<__VA_ARGS__>(TeamArgs);
#define SKM_DECLARE_SEG_TEAM(...) extern template __global__ void k_seg_team<__VA_ARGS__>(TeamArgs);
#define SKM_DEFINE_SEG_TEAM(...) template __global__ void k_seg_team<__VA_ARGS__>(TeamArgs);
