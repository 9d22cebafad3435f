// Microbenchmark: can VALU work hide behind v_mfma_f32_32x32x2_f32 on one SIMD?
//  (1) one wave: K independent v_fma_f32 after every MFMA             -> cycles per (MFMA + K VALU)
//  (2) two waves per SIMD: wave A streams MFMAs (optionally paced by s_nop), wave B runs a VALU stream
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_mix.hip -o mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(256) void k_one(float *out, float a0, int iters, unsigned long long *cyc)
{
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = 1e-3f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// waves 0-3: role A (MFMA, PAD s_nop groups after each), waves 4-7: role B (VALU stream); MODE bit0 = A active, bit1 = B active
template <int PAD, bool EXP>
__global__ __launch_bounds__(512) void k_two(float *out, float a0, int iters, int mode, unsigned long long *cyc)
{
    const int wave = threadIdx.x >> 6;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = 1e-3f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 8; ++rep) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                    for (int p = 0; p < PAD; ++p) asm volatile("s_nop 7");
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                    for (int p = 0; p < PAD; ++p) asm volatile("s_nop 7");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        v[i] = EXP ? __builtin_amdgcn_exp2f(v[i]) : __builtin_fmaf(v[i], 1.0001f, 0.5f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float *out;
static unsigned long long *cyc, h[256 * 8];

template <int K>
void one()
{
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_one<K>, dim3(256), dim3(256), 0, 0, out, 1.0f, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    printf("one wave/SIMD, %2d v_fma per MFMA: %.1f cycles per MFMA\n", K, (double)h[7] / (iters * 16.0));
}

template <int PAD, bool EXP>
void two()
{
    const int iters = 1000;
    double ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL((k_two<PAD, EXP>), dim3(256), dim3(512), 0, 0, out, 1.0f, iters, mode, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
        ra[mode] = (double)h[7 * 8 + 1] / (iters * 16.0);
        rb[mode] = (double)h[7 * 8 + 5] / (iters * 128.0);
    }
    printf("two waves/SIMD, pad %d x s_nop 7, B=%s: A alone %.1f cyc/MFMA, B alone %.2f cyc/VALU | together: A %.1f cyc/MFMA, B %.2f cyc/VALU\n",
           PAD, EXP ? "v_exp" : "v_fma", ra[1], rb[2], ra[3], rb[3]);
}

int main()
{
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    one<0>();
    one<2>();
    one<4>();
    one<8>();
    one<12>();
    one<16>();
    two<0, false>();
    two<1, false>();
    two<2, false>();
    two<4, false>();
    two<6, false>();
    two<0, true>();
    two<4, true>();
    return 0;
}
