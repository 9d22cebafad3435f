// Split-precision E-step (opt-in: SKM_SPLIT=bf16x3): the two GEMMs of k_estep_fat on the bf16 matrix cores.
//
// A float32 value is the exact sum of three bf16 values (8 + 8 + 8 mantissa bits).  Of the nine partial products of two
// such sums the six largest carry everything above 2^-24 relative; v_mfma_f32_32x32x16_bf16 does sixteen k-steps in half
// the time v_mfma_f32_32x32x2_f32 needs for two, so a float32-grade product costs 6/16 of the f32 MFMA time (the same
// arithmetic as csrc/skm_gemm_b3.hip, which runs the two large TV products).  Everything else is k_estep_fat: one
// 512-register wave per SIMD owns four component tiles, statistics accumulators in the 256 AGPRs behind asm MFMAs,
// exact (max, sum) exchange between the four waves, f64 slabs.
//
//   logits   l[t, c] = kc[c] + sum_d x[t, d] wl[d, c] + sum_d x^2[t, d] wq[d, c]      A = frame tile, B = model
//            K = 64 (x, zero beyond D) + 64 (x^2) = 8 k-steps of 16, 6 MFMAs each: 48 per component tile
//   stats    acc[c, d] += sum_t gamma[t, c] [x | 1 | x^2][t, d]                        A = posteriors, B = frame tile
//            K = the 32 frames of the tile = 2 k-steps, 4 column tiles (x 0..31, x 32..63, x^2 0..31, x^2 32..63)
//
// Operand fragments (lane l: r = l & 31, h = l >> 5): A holds A[row r][k = 8 h + j], B holds B[k = 8 h + j][col r],
// j = 0..7 as eight bf16 in four registers.  The result tile is laid out as for the f32 instruction (column = lane & 31,
// row = acc_row(register, h)), so the logits come out with frames in registers and components on lanes -- and the
// posteriors ARE an A operand of the statistics product if k-slot (s, h, j) of that product stands for frame
// 4 h + (j & 3) + 8 (2 s + (j >> 2)), the frame register 8 s + j of lane (c, h) holds: no cross-lane movement, the
// frame tile is staged a second time in that k order.
//
// LDS per frame tile (48 KB, ring of three): LA[ks 8][plane 3][lane 64][16 B] for the logits, SB[jt 4][s 2][plane 3]
// [lane 64][16 B] for the statistics.  The 256 threads stage a tile cooperatively, each splitting 16 + 16 values:
// thread (frame r, h, ks) two float4 of a row for LA, thread (column d, s, h) eight frames of a column for SB.
// The model's planes Wb[tile][ks 8][plane 3][lane 64][16 B] are made from the float32 image at the start of the call
// (k_w_b3: the split of the SAME float32 values the f32 kernels multiply).
#include "skm_gmm_kernels.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define B3E_TILE_BYTES 49152
#define B3E_SB_OFF 24576
#define B3E_NB 6

struct B3EArgs {
    const float *X;
    long T;
    int D;
    const i32x4 *wb;
    const float *kc;
    double *slabs;
    double *llk_part;
    int flush_tiles;
    // models of n_cb blocks of 512 components (MODE 1 / 2): per-(block, frame) partials out, the frames' log-sum-exp in
    int n_cb;
    float *pm, *ps;       // [n_cb][T] base-2 maximum and sum of 2^(l - maximum) of a block's logits
    const float *lse;     // [T] base-2 log-sum-exp over all components (k_lse_combine)
    long long *trace;     // -DB3E_TRACE builds: cycle stamps of one tile of workgroup 1, wave 0 (tools/time_b3.py)
};

#ifdef B3E_TRACE
static long long *g_b3_trace = nullptr;
extern "C" void skm_b3_trace_buffer(void *p) { g_b3_trace = static_cast<long long *>(p); }
#define B3E_STAMP(k) \
    if (trace_now) a.trace[k] = __builtin_readcyclecounter()
#else
#define B3E_STAMP(k)
#endif

__device__ __forceinline__ unsigned b3e_pk(float a, float b)
{
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));       // v_cvt_pk_bf16_f32 (RNE)
}

// three planes of a pair of values, each plane's pair packed in one register (first value in the low half)
__device__ __forceinline__ void b3e_split2(float a, float b, unsigned &p0, unsigned &p1, unsigned &p2)
{
    p0 = b3e_pk(a, b);
    a -= __uint_as_float(p0 << 16);             // exact
    b -= __uint_as_float(p0 & 0xffff0000u);
    p1 = b3e_pk(a, b);
    a -= __uint_as_float(p1 << 16);
    b -= __uint_as_float(p1 & 0xffff0000u);
    p2 = b3e_pk(a, b);
}

__device__ __forceinline__ void b3e_split8(const float (&v)[8], i32x4 (&pl)[3])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b, c;
        b3e_split2(v[2 * q], v[2 * q + 1], a, b, c);
        pl[0][q] = (int)a;
        pl[1][q] = (int)b;
        pl[2][q] = (int)c;
    }
}

// the six partial products, smallest first; plane 0 is the leading term.  `s_nop 1` opens each group: an operand may
// have been written (or copied by the compiler) by a vector instruction right in front of it.
__device__ __forceinline__ void b3e_mm6_v(f32x16 &acc, const i32x4 (&a)[3], const i32x4 (&b)[3])
{
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %6, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %2, %5, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %4, %0"
                 : "+v"(acc)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]));
}
__device__ __forceinline__ void b3e_mm6_a(f32x16 &acc, const i32x4 (&a)[3], const i32x4 (&b)[3])
{
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %6, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %2, %5, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %5, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %4, %0"
                 : "+a"(acc)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(b[0]), "v"(b[1]), "v"(b[2]));
}

// The same six products as three statements of two, so that vector work can be placed between them: a wave issues in
// order, an MFMA of this kind occupies the matrix pipe for 32 cycles, and the few vector instructions put behind each
// pair run in its shadow.  `first` carries the s_nop 1 of the group.
#define B3E_MM2(CONSTRAINT, ACC, A0, B0, A1, B1, FIRST)                                                         \
    do {                                                                                                       \
        if (FIRST)                                                                                             \
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" \
                         : CONSTRAINT(ACC) : "v"(A0), "v"(B0), "v"(A1), "v"(B1));                              \
        else                                                                                                   \
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %3, %4, %0" \
                         : CONSTRAINT(ACC) : "v"(A0), "v"(B0), "v"(A1), "v"(B1));                              \
    } while (0)

// One MFMA per statement, with the vector work of FILL(gap) behind each: a wave issues in order, an MFMA of this kind
// holds the matrix pipe for 32 cycles and the vector issue port for 8 of them, and single-issue instructions whose issue
// costs sum to about 24 cycles hide in that gap (MI355X_MICROARCH.md, "single-issue instructions HIDDEN per gap").
// Two or three 4-cycle instructions and one memory instruction per gap; packed-f32 instructions are kept out (they cost
// ~13 cycles beside MFMAs: b3e_subf / b3e_mulf are asm so that the SLP vectoriser cannot pair them).
#define B3E_MF(CONSTRAINT, ACC, A_, B_)                                                                          \
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : CONSTRAINT(ACC) : "v"(A_), "v"(B_));               \
    __builtin_amdgcn_sched_barrier(0)
#define B3E_GROUP(CONSTRAINT, ACC, A, B, FILL)                                                                   \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        B3E_MF(CONSTRAINT, ACC, A[2], B[0]); FILL(0); __builtin_amdgcn_sched_barrier(0);                        \
        B3E_MF(CONSTRAINT, ACC, A[0], B[2]); FILL(1); __builtin_amdgcn_sched_barrier(0);                        \
        B3E_MF(CONSTRAINT, ACC, A[1], B[1]); FILL(2); __builtin_amdgcn_sched_barrier(0);                        \
        B3E_MF(CONSTRAINT, ACC, A[1], B[0]); FILL(3); __builtin_amdgcn_sched_barrier(0);                        \
        B3E_MF(CONSTRAINT, ACC, A[0], B[1]); FILL(4); __builtin_amdgcn_sched_barrier(0);                        \
        B3E_MF(CONSTRAINT, ACC, A[0], B[0]); FILL(5); __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)

__device__ __forceinline__ float b3e_subf(float a, float b)
{
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float b3e_mulf(float a, float b)
{
    float d;
    asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// one pair of values on its way to three packed bf16 pairs, one gap of a group at a time (2, 2, 2, 1, 2, 3 instructions)
struct B3EPairG {
    float a, b, t0, t1;
    unsigned p0, p1, p2;
};
__device__ __forceinline__ void b3e_pair_gap(B3EPairG &q, int gap)
{
    if (gap == 0) {
        q.p0 = b3e_pk(q.a, q.b);
    } else if (gap == 1) {
        q.t0 = __uint_as_float(q.p0 << 16);
        q.t1 = __uint_as_float(q.p0 & 0xffff0000u);
    } else if (gap == 2) {
        q.a = b3e_subf(q.a, q.t0);
        q.b = b3e_subf(q.b, q.t1);
    } else if (gap == 3) {
        q.p1 = b3e_pk(q.a, q.b);
    } else if (gap == 4) {
        q.t0 = __uint_as_float(q.p1 << 16);
        q.t1 = __uint_as_float(q.p1 & 0xffff0000u);
    } else {
        q.a = b3e_subf(q.a, q.t0);
        q.b = b3e_subf(q.b, q.t1);
        q.p2 = b3e_pk(q.a, q.b);
    }
}

// one pair of values on its way to three packed bf16 pairs, in three steps of 4, 4 and 1 instructions
struct B3EPair {
    float a, b;
    unsigned p0, p1, p2;
};
__device__ __forceinline__ void b3e_pair_step(B3EPair &q, int step)
{
    if (step == 0) {
        q.p0 = b3e_pk(q.a, q.b);
        q.a -= __uint_as_float(q.p0 << 16);
        q.b -= __uint_as_float(q.p0 & 0xffff0000u);
    } else if (step == 1) {
        q.p1 = b3e_pk(q.a, q.b);
        q.a -= __uint_as_float(q.p1 << 16);
        q.b -= __uint_as_float(q.p1 & 0xffff0000u);
    } else {
        q.p2 = b3e_pk(q.a, q.b);
    }
}

// 12 wait states behind the last bf16 MFMA (8 passes + 4) before a vector instruction may read a logits accumulator;
// only the last one issued can still be in flight
__device__ __forceinline__ void b3e_settle4(f32x16 (&lg)[4])
{
    asm volatile("s_nop 13" : "+v"(lg[0]), "+v"(lg[1]), "+v"(lg[2]), "+v"(lg[3]));
}

// float32 image (skm_gmm_pack / k_pack_w) -> the three bf16 planes in logits-B fragment order
__global__ __launch_bounds__(256) void k_w_b3(const float *__restrict__ wp, int CTpad, int NCH, int NP, int D,
                                               i32x4 *__restrict__ wb)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (tile * 8 + ks) * 64 + lane
    if (e >= (long)CTpad * 8 * 64) return;
    const int lane = (int)(e & 63), ks = (int)((e >> 6) & 7), tile = (int)(e >> 9);
    const int comp = lane & 31, h = lane >> 5, kind = ks >> 2;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 16 * (ks & 3) + 8 * h + j;
        float x = 0.f;
        if (d < D) {
            int grp, ln, i;
            if (d < 8 * NP) {
                grp = 2 * (d >> 3) + kind;
                ln = comp + 32 * ((d >> 2) & 1);
                i = d & 3;
            } else {
                grp = NCH - 1;
                ln = comp + 32 * kind;
                i = d - 8 * NP;
            }
            x = wp[(((long)tile * NCH + grp) * 64 + ln) * 4 + i];
        }
        v[j] = x;
    }
    i32x4 pl[3];
    b3e_split8(v, pl);
    i32x4 *dst = wb + ((long)tile * 8 + ks) * 3 * 64 + lane;
    dst[0] = pl[0];
    dst[64] = pl[1];
    dst[128] = pl[2];
}

// MODE 0: the workgroup holds every component (C <= 512): one sweep.
// MODE 1 / 2: a model of n_cb blocks of 512 components in two sweeps over the frames (the accumulators of 2048 components
//   do not fit one CU, and the team kernels' hand-off pipeline needs LDS this kernel spends on its bf16 planes).  Workgroup
//   = (component block, frame range); the blocks of one frame range take workgroup ids 8 apart (one XCD: the frames come
//   from that L2 once).  MODE 1: logits, (max, sum) of the block per frame -> pm / ps; k_lse_combine makes the frames'
//   log-sum-exp (and the log-likelihood); MODE 2: the SAME logits again bit for bit, posteriors 2^(l - lse), statistics.
//   2 x logits + statistics = 1.5 x the MFMA work of one sweep, at 6 / 16 of the f32 MFMA time each.
template <int MODE>
__global__ __launch_bounds__(256, MODE == 1 ? 2 : 1) void k_estep_b3(B3EArgs a)
{
    constexpr int NCT = 4, NF = 4, NFW = NF * 32, NWV = 4;
    // the first sweep stages no statistics planes and keeps no accumulators: half the LDS, under 256 registers -- two
    // workgroups per CU, so one wave's log-sum-exp stretch runs beside the other's MFMAs
    constexpr int TILE_BYTES = (MODE == 1) ? B3E_SB_OFF : B3E_TILE_BYTES;
    const int n_cb = MODE ? a.n_cb : 1;
    const int cbk = MODE ? (int)((blockIdx.x >> 3) % n_cb) : 0;
    const int fsi = MODE ? (int)((blockIdx.x & 7) + 8 * (blockIdx.x / (8 * n_cb))) : (int)blockIdx.x;
    const int n_fs = (int)gridDim.x / n_cb;
    extern __shared__ __align__(16) char smem_b3[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int D = a.D;

    char *ring = smem_b3;                                                        // [3][48 KB]
    float *partm = reinterpret_cast<float *>(smem_b3 + 3 * TILE_BYTES);      // [2][NWV][32]
    float *parts = partm + 2 * NWV * 32;                                         // [2][NWV][32]
    float *ls = parts + 2 * NWV * 32 + wave * 32;                                // wave-private row of 32

    float kcv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) kcv[ct] = a.kc[((cbk * NWV + wave) * NCT + ct) * 32 + c];
    // this wave's model planes [ct][ks][plane][lane] behind a buffer descriptor: one per-lane offset register for every
    // load and an immediate / scalar offset per fragment (flat addresses cost 96 address pairs, which hipcc spilled)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<i32x4 *>(a.wb + (long)(__builtin_amdgcn_readfirstlane(cbk * NWV + wave) * NCT) * 8 * 3 * 64), 0,
        NCT * 8 * 3 * 1024,
        0x00027000);
    const int wvoff = lane * 16;

    f32x16 acc[NCT][NF];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int jf = 0; jf < NF; ++jf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][jf][r] = 0.f;

    const long tiles = (a.T + 31) >> 5;
    const long ta = tiles * fsi / n_fs, tb = tiles * (fsi + 1) / n_fs;
    // slot (frame range, block, wave) as k_reduce_slabs walks it: ((fs * (n_cb * NWV)) + cbk * NWV + wave)
    double *slab_lane = a.slabs + ((long)fsi * (n_cb * NWV) + cbk * NWV + wave) * (long)(NCT * 32 * NFW) + (4 * h) * NFW + c;
    double llk_lane = 0.0;

    // ---- staging roles of this thread
    // LA: frame ra, features 16 * wave + 8 * ha + (0..7)
    const int ra = tid & 31, ha = (tid >> 5) & 1, da = 16 * wave + 8 * ha;
    // SB: column db (0..63), frames 4 * hb + (j & 3) + 8 * (2 * sb + (j >> 2))
    const int db = tid & 63, sb = (tid >> 6) & 1, hb = tid >> 7;
    const int la_off = ((wave * 3) * 64 + ha * 32 + ra) * 16;                    // plane stride 1024, x^2: + 4 * 3 * 1024
    const int sb_off = B3E_SB_OFF + ((((db >> 5) * 2 + sb) * 3) * 64 + hb * 32 + (db & 31)) * 16;   // x^2: + 2 * 2 * 3 * 1024
    float pre_a[8], pre_b[8];
    auto load_pre = [&](long t0, int nvalid) __attribute__((always_inline)) {
        const float *row = a.X + (t0 + ra) * (long)D + da;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (ra < nvalid && da + 4 <= D) v0 = *reinterpret_cast<const f32x4 *>(row);
        if (ra < nvalid && da + 8 <= D) v1 = *reinterpret_cast<const f32x4 *>(row + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pre_a[i] = v0[i];
            pre_a[4 + i] = v1[i];
        }
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int fr = 4 * hb + (j & 3) + 8 * (2 * sb + (j >> 2));
                float x = (db == D) ? 1.0f : 0.f;                                    // the ones column makes N
                if (db < D && fr < nvalid) x = a.X[(t0 + fr) * (long)D + db];
                pre_b[j] = x;
            }
        }
    };
    auto store_pre = [&](char *slot) __attribute__((always_inline)) {
        i32x4 pl[3];
        float sq[8];
        b3e_split8(pre_a, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<i32x4 *>(slot + la_off + p * 1024) = pl[p];
#pragma unroll
        for (int i = 0; i < 8; ++i) sq[i] = pre_a[i] * pre_a[i];
        b3e_split8(sq, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<i32x4 *>(slot + la_off + 4 * 3 * 1024 + p * 1024) = pl[p];
        if (MODE == 1) return;
        b3e_split8(pre_b, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<i32x4 *>(slot + sb_off + p * 1024) = pl[p];
#pragma unroll
        for (int i = 0; i < 8; ++i) sq[i] = pre_b[i] * pre_b[i];
        b3e_split8(sq, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<i32x4 *>(slot + sb_off + 2 * 2 * 3 * 1024 + p * 1024) = pl[p];
    };

    if (ta < tb) {
        const long t0 = ta << 5;
        load_pre(t0, (int)((a.T - t0) < 32 ? (a.T - t0) : 32));
        store_pre(ring);
        if (ta + 1 < tb) load_pre(t0 + 32, (int)((a.T - t0 - 32) < 32 ? (a.T - t0 - 32) : 32));
    }
    __syncthreads();

    int cur = 0;
    f32x16 lg[NCT];
    i32x4 wf[B3E_NB][3];
    auto wload = [&](int item, i32x4 (&dst)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dst[p] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   wrs, wvoff, (((item & 3) * 8 + (item >> 2)) * 3 + p) * 1024, 0));
    };
#pragma unroll
    for (int i = 0; i < B3E_NB - 1; ++i) wload(i, wf[i]);
    auto logits_start = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float k = kcv[ct];
            asm volatile("" : "+v"(k));             // keeps the splat inside the loop (see k_estep_fat)
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[ct][r] = k;
        }
        // a vector write needs two wait states before an MFMA reads the register as its C operand; the statement also
        // keeps the splat here, where it closes the tile, instead of in front of each accumulator's first MFMA
        asm volatile("s_nop 1" : "+v"(lg[0]), "+v"(lg[1]), "+v"(lg[2]), "+v"(lg[3]));
    };
    for (long base = ta; base < tb; base += a.flush_tiles) {
    const long end = (base + a.flush_tiles < tb) ? base + a.flush_tiles : tb;
    logits_start();
#pragma unroll 1
    for (long tile = base; tile < end; ++tile) {
        const long t0 = tile << 5;
        const int nvalid = (int)((a.T - t0) < 32 ? (a.T - t0) : 32);
        const int nxt = (cur == 2) ? 0 : cur + 1;
        const int pslot = (int)((tile - ta) & 1);
        const char *slot = ring + cur * TILE_BYTES;
#ifdef B3E_TRACE
        const bool trace_now = a.trace && blockIdx.x == 1 && tid == 0 && tile - ta == 40;
#endif
        B3E_STAMP(0);
        f32x4 l4[4];
        if (MODE == 2) {
            // the log-sum-exp of this lane's 16 frames (register 4 q + k holds frame 8 q + 4 h + k); frames beyond T get
            // +inf, i.e. a posterior of exactly zero (the reads themselves stay inside the workspace)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                l4[q4] = *reinterpret_cast<const f32x4 *>(a.lse + t0 + 8 * q4 + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (8 * q4 + 4 * h + k >= nvalid) l4[q4][k] = __builtin_inff();
            }
        }

        // ---- logits: 32 items (k-step, component tile) of 6 MFMAs; the model planes stream from L2 through a ring of
        // B3E_NB items, B3E_NB - 1 items (about a thousand cycles of MFMAs) ahead of their use; the first ones of a tile
        // were requested in front of the previous tile's statistics product.  The gaps behind the MFMAs carry: the three
        // plane loads of item + B3E_NB - 1 (gaps 1, 3, 5); the frame planes of the next k-step (gaps 0, 2, 4 of the
        // k-step's last item); and, in the first 16 items, the staging of the NEXT frame tile -- one pair of values
        // split per item (16 pairs = the thread's 8 + 8 values and their squares), its three 16-byte LDS stores in gaps 0,
        // 2, 4 of the item behind every fourth.  The staging goes to the slot last read two tiles ago (every wave is past
        // that tile's barrier) and is harmless when there is no next tile.
        {
            const i32x4 *la = reinterpret_cast<const i32x4 *>(slot) + lane;
            char *dst = ring + nxt * TILE_BYTES;
            constexpr int NSTG = (MODE == 1) ? 8 : 16;      // pairs of values this thread stages (no SB planes in MODE 1)
            i32x4 af[2][3], pl[3], plw[3];
            B3EPairG pr;
#pragma unroll
            for (int p = 0; p < 3; ++p) af[0][p] = la[p * 64];
#pragma unroll
            for (int item = 0; item < 32; ++item) {
                const int ks = item >> 2, ct = item & 3;
                const int grp = item >> 2, q = item & 3;         // staging: pair q of value group grp (item < 16)
                if (item < NSTG) {
                    const float u = (grp < 2) ? pre_a[2 * q] : pre_b[2 * q], v = (grp < 2) ? pre_a[2 * q + 1] : pre_b[2 * q + 1];
                    pr.a = u;
                    pr.b = v;
                }
                const i32x4(&w)[3] = wf[item % B3E_NB];
                const i32x4(&fa)[3] = af[ks & 1];
                auto fill = [&](int gap) __attribute__((always_inline)) {
                    if ((gap & 1) && item + B3E_NB - 1 < 32) {
                        const int it2 = item + B3E_NB - 1, p = gap >> 1;
                        wf[it2 % B3E_NB][p] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                  wrs, wvoff, (((it2 & 3) * 8 + (it2 >> 2)) * 3 + p) * 1024, 0));
                    }
                    if (!(gap & 1) && ct == 3 && ks < 7) af[(ks + 1) & 1][gap >> 1] = la[((ks + 1) * 3 + (gap >> 1)) * 64];
                    if (!(gap & 1) && q == 0 && item >= 4 && item <= NSTG) {
                        const int g0 = grp - 1;                  // the value group finished by the previous item
                        const int off = (g0 < 2 ? la_off : sb_off) + ((g0 & 1) ? 12 * 1024 : 0);
                        *reinterpret_cast<i32x4 *>(dst + off + (gap >> 1) * 1024) = plw[gap >> 1];
                    }
                    if (item < NSTG) {
                        if (gap == 0 && (grp & 1)) {
                            pr.a = b3e_mulf(pr.a, pr.a);
                            pr.b = b3e_mulf(pr.b, pr.b);
                        }
                        b3e_pair_gap(pr, gap);
                        if (gap == 5) {
                            pl[0][q] = (int)pr.p0;
                            pl[1][q] = (int)pr.p1;
                            pl[2][q] = (int)pr.p2;
                            if (q == 3) {
#pragma unroll
                                for (int p = 0; p < 3; ++p) plw[p] = pl[p];
                            }
                        }
                    }
                };
                B3E_GROUP("+v", lg[ct], fa, w, fill);
            }
        }
        B3E_STAMP(1);
        b3e_settle4(lg);

        float fs[16];
        if (MODE == 2) {
            // ---- posteriors from the frames' log-sum-exp: one subtraction and one exponential per logit, no exchange
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) lg[ct][r] = __builtin_amdgcn_exp2f(lg[ct][r] - l4[r >> 2][r & 3]);
#pragma unroll
            for (int r = 0; r < 16; ++r) fs[r] = 1.0f;
            B3E_STAMP(2);
            __syncthreads();        // ring discipline only: the staging of tile + 1 is visible, tile - 1's slot is free
            B3E_STAMP(3);
        } else {
        // ---- exact per-frame maximum over this wave's 128 components, e = 2^(l - max), per-frame sum of e
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
        B3E_STAMP(2);
        __syncthreads();
        B3E_STAMP(3);
        {
            const float *pmv = partm + pslot * NWV * 32 + c;
            const float *psv = parts + pslot * NWV * 32 + c;
            float mx = pmv[0];
            for (int w = 1; w < NWV; ++w) mx = __builtin_fmaxf(mx, pmv[w * 32]);
            float sum = 0.f;
            for (int w = 0; w < NWV; ++w) sum += psv[w * 32] * __builtin_amdgcn_exp2f(pmv[w * 32] - mx);
            const bool valid = c < nvalid;
            if (MODE == 1) {
                // the block's (max, sum) of frame c: all k_lse_combine needs
                if (wave == 0 && h == 0 && valid) {
                    a.pm[(long)cbk * a.T + t0 + c] = mx;
                    a.ps[(long)cbk * a.T + t0 + c] = sum;
                }
            } else {
                const float f = __builtin_amdgcn_exp2f(pmv[wave * 32] - mx) * __builtin_amdgcn_rcpf(sum);
                if (h == 0) ls[c] = valid ? f : 0.f;
                if (wave == 0 && h == 0 && valid) llk_lane += (double)(mx + __builtin_amdgcn_logf(sum));
            }
        }
        if (MODE == 0) {
            wave_lds_fence();
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 f4 = *reinterpret_cast<const f32x4 *>(ls + 8 * q4 + 4 * h);
#pragma unroll
                for (int k = 0; k < 4; ++k) fs[4 * q4 + k] = f4[k];
            }
        }
        }
        if (tile + 2 < tb) {
            const long t2 = (tile + 2) << 5;
            load_pre(t2, (int)((a.T - t2) < 32 ? (a.T - t2) : 32));
        }

        if (MODE != 1) {
        // ---- statistics: per component tile 8 groups (k-step s, column tile jt) of 6 MFMAs.  The posteriors
        // gamma = e * f of tile ct + 1 are scaled and split into their three planes between the MFMAs of tile ct (frames
        // 8 s + j of a lane are k-slot j of k-step s: one pair of registers per group); tile 0 is split up front, and the
        // groups of the last tile carry the constants the next frame tile's logits start from.
        {
            i32x4 gp[2][2][3];            // [ct & 1][k-step][plane]
            B3EPair pr;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                pr.a = lg[0][2 * g] * fs[2 * g];
                pr.b = lg[0][2 * g + 1] * fs[2 * g + 1];
                b3e_pair_step(pr, 0);
                b3e_pair_step(pr, 1);
                b3e_pair_step(pr, 2);
                gp[0][g >> 2][0][g & 3] = (int)pr.p0;
                gp[0][g >> 2][1][g & 3] = (int)pr.p1;
                gp[0][g >> 2][2][g & 3] = (int)pr.p2;
            }
            // the next tile's first model planes (the same every tile, but 60 registers are better used here)
#pragma unroll
            for (int i = 0; i < B3E_NB - 1; ++i) wload(i, wf[i]);
            B3E_STAMP(4);
            const i32x4 *sbp = reinterpret_cast<const i32x4 *>(slot + B3E_SB_OFF) + lane;
            i32x4 bf[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[0][p] = sbp[p * 64];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int sk = g >> 2, jt = g & 3, it = ct * 8 + g;
                    B3EPairG pq;
                    if (ct + 1 < NCT) {
                        pq.a = lg[(ct + 1) % NCT][2 * g];
                        pq.b = lg[(ct + 1) % NCT][2 * g + 1];
                    }
                    const i32x4(&ga)[3] = gp[ct & 1][sk];
                    const i32x4(&b)[3] = bf[it & 1];
                    auto fill = [&](int gap) __attribute__((always_inline)) {
                        if ((gap & 1) && it + 1 < 32) {
                            const int g1 = (g + 1) & 7;
                            bf[(it + 1) & 1][gap >> 1] = sbp[(((g1 & 3) * 2 + (g1 >> 2)) * 3 + (gap >> 1)) * 64];
                        }
                        if (ct + 1 < NCT) {
                            if (gap == 0) {
                                pq.a = b3e_mulf(pq.a, fs[2 * g]);
                                pq.b = b3e_mulf(pq.b, fs[2 * g + 1]);
                            }
                            b3e_pair_gap(pq, gap);
                            if (gap == 5) {
                                gp[(ct + 1) & 1][sk][0][jt] = (int)pq.p0;
                                gp[(ct + 1) & 1][sk][1][jt] = (int)pq.p1;
                                gp[(ct + 1) & 1][sk][2][jt] = (int)pq.p2;
                            }
                        }
                    };
                    B3E_GROUP("+a", acc[ct][jt], ga, b, fill);
                }
            }
        }
        // the wait states a bf16 MFMA result needs before anything else may touch it, in front of the loop's exit edge
        // (the compiler moves accumulators there: see k_estep_fat); the last MFMA issued wrote acc[NCT-1][NF-1]
        asm volatile("s_nop 13" : "+a"(acc[NCT - 1][NF - 1]));
        } else {
            // first sweep: no statistics; the next tile's first model planes
#pragma unroll
            for (int i = 0; i < B3E_NB - 1; ++i) wload(i, wf[i]);
        }
        B3E_STAMP(5);
        logits_start();     // the next tile's; also the wait states in front of the loop's exit edge (see k_estep_fat)
        wave_lds_fence();
        cur = nxt;
    }
    if (MODE != 1) {
        if (base == ta)
            flush_slab_agpr<NCT, NF, true>(acc, slab_lane);
        else
            flush_slab_agpr<NCT, NF, false>(acc, slab_lane);
    }
    }
    if (MODE != 1 && ta >= tb) flush_slab_agpr<NCT, NF, true>(acc, slab_lane);   // empty range: the slab must still read as zero
    if (MODE == 0 && wave == 0) {
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) llk_lane += __shfl_xor(llk_lane, msk);
        if (lane == 0) a.llk_part[blockIdx.x] = llk_lane * 0.6931471805599453094;
    }
}

size_t skm_estep_b3_lds() { return (size_t)3 * B3E_TILE_BYTES + (size_t)(4 * 4 * 32 + 4 * 32) * sizeof(float); }
static size_t skm_estep_b3_lds_sweep1() { return (size_t)3 * B3E_SB_OFF + (size_t)(4 * 4 * 32 + 4 * 32) * sizeof(float); }

size_t skm_estep_b3_wb_bytes(int CTpad) { return (size_t)CTpad * 8 * 3 * 64 * 16; }

// wp: the float32 image; wb: room for skm_estep_b3_wb_bytes(CTpad).  D % 4 == 0, 32 <= D <= 60.
// n_cb == 1: C <= 512 (16 component tiles), one sweep, nblk workgroups (llk_part gets one partial per workgroup).
// n_cb  > 1: CTpad = 16 n_cb tiles in two sweeps; nblk must be a multiple of 8 n_cb; pm / ps: n_cb * T floats each, lse: T
//            floats (+ 32 readable beyond), llk_part: n_llk partials written by k_lse_combine.
int skm_estep_b3_launch(SkmCtx *ctx, const float *X, long T, int D, int CTpad, int NCH, int NP, const float *wp,
                        const float *kc, void *wb, double *slabs, double *llk_part, int n_llk, int nblk, int flush_tiles,
                        int n_cb, float *pm, float *ps, float *lse, hipStream_t st)
{
    {
        SkmProfScope ps_(ctx, "k_w_b3", st);
        hipLaunchKernelGGL(k_w_b3, dim3((unsigned)((CTpad * 8 * 64 + 255) / 256)), dim3(256), 0, st, wp, CTpad, NCH, NP, D,
                           static_cast<i32x4 *>(wb));
    }
    B3EArgs a;
    a.X = X;
    a.T = T;
    a.D = D;
    a.wb = static_cast<const i32x4 *>(wb);
    a.kc = kc;
    a.slabs = slabs;
    a.llk_part = llk_part;
    a.flush_tiles = flush_tiles;
    a.n_cb = n_cb;
    a.pm = pm;
    a.ps = ps;
    a.lse = lse;
#ifdef B3E_TRACE
    a.trace = g_b3_trace;
#else
    a.trace = nullptr;
#endif
    const size_t lds = skm_estep_b3_lds();
    if (n_cb <= 1) {
        SKM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_estep_b3<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        SkmProfScope ps_(ctx, "k_estep_b3", st);
        hipLaunchKernelGGL(k_estep_b3<0>, dim3(nblk), dim3(256), lds, st, a);
        SKM_HIP(hipGetLastError());
        return SKM_OK;
    }
    SKM_REQUIRE(nblk % (8 * n_cb) == 0 && pm && ps && lse, "split-precision E-step in two sweeps: bad launch geometry");
    const size_t lds1 = skm_estep_b3_lds_sweep1();
    SKM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_estep_b3<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds1));
    SKM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_estep_b3<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    {
        SkmProfScope ps_(ctx, "k_b3_sweep1", st);
        hipLaunchKernelGGL(k_estep_b3<1>, dim3(2 * nblk), dim3(256), lds1, st, a);      // two workgroups per CU
    }
    SKM_HIP(hipGetLastError());
    long nb = (T + 255) / 256;
    if (nb > n_llk) nb = n_llk;
    if (nb < 1) nb = 1;
    {
        SkmProfScope ps_(ctx, "k_lse_combine", st);
        hipLaunchKernelGGL(k_lse_combine, dim3((unsigned)nb), dim3(256), 0, st, pm, ps, n_cb, T, lse, (float *)nullptr, llk_part);
    }
    SKM_HIP(hipGetLastError());
    {
        SkmProfScope ps_(ctx, "k_b3_sweep2", st);
        hipLaunchKernelGGL(k_estep_b3<2>, dim3(nblk), dim3(256), lds, st, a);
    }
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}
