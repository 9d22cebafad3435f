// skm_gemm_dev.h -- device-side pieces shared by the GEMM kernels (skm_gemm.hip, skm_gemm_wide.hip).
#pragma once
#include "skm_common.h"

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

// float32 on 16 x 16 x 4 MFMAs: 64 x 64 workgroup tiles for batches of small matrices (the per-session Cholesky
// chain of the TV E-step, whose panels are 64 wide); same flop rate per SIMD as the 32 x 32 x 2 instruction.
struct MfF32Small {
    static constexpr int TM = 16, KS = 4, NR = 4;
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int r, int lane) { return 4 * (lane >> 4) + r; }
};

template <typename T, bool SMALL>
struct MfSel {
    typedef Mf<T> type;
};
template <>
struct MfSel<float, true> {
    typedef MfF32Small type;
};

// Operand tile loader: BROWS x BK logical tile (row = m or n, col = k), 256 threads, 16-byte vectors along
// the contiguous dimension of the stored matrix when `vec` (alignment checked on the host).
//   KFAST: the stored matrix is contiguous along k (A not transposed / B transposed), else along the row.
// LDS image, chosen so that the global vector goes to LDS as ONE 16-byte store either way:
//   row-fast operand  T[k][row]   row stride LDT = BROWS + VW   fragments: one scalar read per MFMA k-step
//   k-fast operand    T[row][k]   row stride LDK = BK + VW      fragments: one 16-byte read per VW k-steps
// (row strides of 33 / 9 / 5 sixteen-byte units: the lanes of a read or write group fall into distinct banks).
template <typename TI, int BROWS, int BK, bool KFAST>
struct TileIO {
    static constexpr int VW = 16 / sizeof(TI);
    static constexpr int NV = BROWS * BK / VW / 256;  // vectors per thread
    static constexpr int LDT = BROWS + VW, LDK = BK + VW;
    static constexpr int LDS_ELEMS = KFAST ? BROWS * LDK : BK * LDT;
    typedef TI vec_t __attribute__((ext_vector_type(16 / sizeof(TI))));

    // (row, k) of element 0 of vector i of thread tid
    static __device__ __forceinline__ void coord(int tid, int i, int &row, int &k)
    {
        const int e = tid + i * 256;
        if (KFAST) {
            k = (e % (BK / VW)) * VW;
            row = e / (BK / VW);
        } else {
            row = (e % (BROWS / VW)) * VW;
            k = e / (BROWS / VW);
        }
    }

    // General (bounds-checked) loader.  map != nullptr (row-fast layouts only): logical row gr lives at element
    // offset map[gr] of a stored row (flattened "rows right of the diagonal block" view of an R x R matrix; 4
    // consecutive logical rows that start at a multiple of 4 are contiguous when vec is set)
    static __device__ __forceinline__ void load(TI (&r)[NV * VW], const TI *__restrict__ P, long ld, bool vec,
                                               int row0, int k0, int rows, int K, int tid,
                                               const int *__restrict__ map = nullptr)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int row, k;
            coord(tid, i, row, k);
            const int gr = row0 + row, gk = k0 + k;
            if (map && !KFAST) {
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
            const TI *p = KFAST ? (P + (long)gr * ld + gk) : (P + (long)gk * ld + gr);
            const bool full = KFAST ? (gr < rows && gk + VW <= K) : (gk < K && gr + VW <= rows);
            if (vec && full) {
                const vec_t v = *reinterpret_cast<const vec_t *>(p);
#pragma unroll
                for (int j = 0; j < VW; ++j) r[i * VW + j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    const bool ok = KFAST ? (gr < rows && gk + j < K) : (gk < K && gr + j < rows);
                    r[i * VW + j] = ok ? p[j] : (TI)0;
                }
            }
        }
    }

    static __device__ __forceinline__ void store(const TI (&r)[NV * VW], TI *T, int tid)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int row, k;
            coord(tid, i, row, k);
            vec_t v;
#pragma unroll
            for (int j = 0; j < VW; ++j) v[j] = r[i * VW + j];
            *reinterpret_cast<vec_t *>(KFAST ? (T + row * LDK + k) : (T + k * LDT + row)) = v;
        }
    }
};

// The k index that lane group lk (= lane / TM) feeds to MFMA step s of a k-tile.  Steps come in groups of VW (the
// elements of one 16-byte LDS read of a k-fast operand): group q covers k in [q*VW*NLK, (q+1)*VW*NLK), lane
// group lk owns VW consecutive k of it.  Both operands use the same map, so any order is a valid contraction.
template <int VW, int NLK>
__device__ __forceinline__ constexpr int gemm_k_of(int s, int lk)
{
    return (s / VW) * (VW * NLK) + lk * VW + (s % VW);
}

// One barrier per k-tile joins the compute and the staging waves of a workgroup.  Raw s_barrier behind a wait for the
// wave's own LDS operations only -- never vmcnt(0): the staging waves' loads of later tiles stay in flight across it.
__device__ __forceinline__ void gemm_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

