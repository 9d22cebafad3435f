// Microbenchmark: sustained whole-chip rate of v_mfma_f32_32x32x16_bf16 (register operands only, no memory traffic),
// wall clock by HIP events: what the clock under a sustained bf16 MFMA load leaves of the 2.5 PF dense peak.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip -o mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the same loop on RANDOM operands (four A and four B fragments per lane, read once from memory): the switching
// activity of real data, which is what the power management sees
template <int NACC>
__global__ __launch_bounds__(512, 1) void k_random(float *out, const bf16x8 *src, int iters)
{
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = src[(blockIdx.x * 8 + j) * 512 + threadIdx.x];
        b[j] = src[(blockIdx.x * 8 + 4 + j) * 512 + threadIdx.x];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + rep) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512, 1) void k(float *out, float a0, int iters)
{
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)i;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j);
        b[j] = (__bf16)(1e-3f * j);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out;
    hipMalloc(&out, 4096 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int threads = 256; threads <= 512; threads += 256)
        for (int iters = 2000; iters <= 32000; iters *= 4) {
            const int blocks = 256;
            hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * (threads / 64) * iters * 48.0 * 32 * 32 * 16 * 2;
            printf("%d waves/SIMD, %6d x 48 MFMAs per wave: %8.3f ms  %.0f TF (%.2f of 2500), %.1f cycles/MFMA/SIMD at 2.4 GHz\n",
                   threads / 256, iters, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0,
                   ms * 1e-3 * 2.4e9 / ((double)iters * 48 * (threads / 256)));
        }
    // random operands: bf16 values with random sign and mantissa, exponents within 2^-4 .. 2^4
    const size_t n = (size_t)256 * 8 * 512 * 8;
    unsigned short *h = (unsigned short *)malloc(n * 2), *d;
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned e = 123 + (unsigned)((x >> 20) % 9);
        h[i] = (unsigned short)(((x & 1) << 15) | (e << 7) | ((x >> 8) & 0x7F));
    }
    hipMalloc(&d, n * 2);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    for (int iters = 8000; iters <= 32000; iters *= 4) {
        const int blocks = 256, threads = 512;
        hipLaunchKernelGGL(k_random<8>, dim3(blocks), dim3(threads), 0, 0, out, (const bf16x8 *)d, iters);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_random<8>, dim3(blocks), dim3(threads), 0, 0, out, (const bf16x8 *)d, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * (threads / 64) * iters * 48.0 * 32 * 32 * 16 * 2;
        printf("random operands, 2 waves/SIMD, %6d x 48 MFMAs per wave: %8.3f ms  %.0f TF (%.2f of 2500)\n", iters, ms,
               fl / ms / 1e9, fl / ms / 1e9 / 2500.0);
    }
    return 0;
}
