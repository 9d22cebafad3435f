// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 as a function of the number of independent
// accumulators per wave and of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, float a0, float b0, int iters, unsigned long long *cyc)
{
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)i;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int waves_per_simd)
{
    float *out;
    unsigned long long *cyc, h[256];
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 2000, threads = 256 * waves_per_simd;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 1.0f, 1e-3f, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double per = (double)h[7] / ((double)iters * 16 * waves_per_simd);
    printf("NACC=%d waves/SIMD=%d : %.1f cycles per MFMA per SIMD\n", NACC, waves_per_simd, per);
    hipFree(out);
    hipFree(cyc);
}
int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<1>(w);
        run<2>(w);
        run<4>(w);
        run<8>(w);
    }
    return 0;
}
