// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 issued by ONE wave per SIMD through inline asm, with the four
// accumulators in VGPRs ("+v", as the logits GEMM of the E-step kernels keeps them) or in AGPRs ("+a", the statistics
// accumulators), optionally with one buffer_load_dwordx4 behind every 4 MFMAs (the W stream).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_acc_file.hip -o mfma_acc_file
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool AG>
__device__ __forceinline__ void mf(f32x16 &acc, float a, float b)
{
    if (AG)
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <bool AG, int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, const float *w, int iters, unsigned long long *cyc)
{
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)i;
    const int lane = threadIdx.x & 63;
    f32x4 a4 = {1.f + lane * 1e-9f, 2.f, 3.f, 4.f};
    f32x4 b4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = *reinterpret_cast<const f32x4 *>(w + (i * 64 + lane) * 4);
    float e[4] = {-1.f - lane, -2.f, -3.f, -4.f};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[3] = {{1.f, 2.f}, {1e-3f, 1e-3f}, {3.f, 4.f}};
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 1 << 20, 0x00027000);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mf<AG>(acc[ct], a4[i], b4[ct][i]);
                if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(e[0]));
                if (MODE == 3) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(e[0]), "+v"(e[1]));
                if (MODE == 4) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(p[1]));
                if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4" : "+v"(p[0]), "+v"(p[2]) : "v"(p[1]), "v"(e[0]), "v"(e[1]));
                if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(e[0]) : "v"(e[1]));
            }
            if (MODE >= 2) {    // vector ALU work of the same wave between the MFMAs: MODE 2/3 = 1/2 v_exp_f32 per 4 MFMAs ... 
            }
            if (MODE == 1) {    // a load into the registers of the set used 3 groups later
                b4[(ct + 3) & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((it * 4 + ct) & 63) * 1024, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (AG)
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[i]));
        else
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[i]));
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    s += e[0] + e[1] + e[2] + e[3] + p[0][0] + p[2][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool AG, int MODE>
void run(const char *what)
{
    float *out, *w;
    unsigned long long *cyc, h[256];
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&w, 1 << 20);
    hipMemset(w, 0, 1 << 20);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<AG, MODE>), dim3(256), dim3(256), 0, 0, out, w, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    printf("%-44s : %.2f cycles per MFMA\n", what, (double)h[7] / ((double)iters * 16));
    hipFree(out);
    hipFree(w);
    hipFree(cyc);
}
int main()
{
    run<false, 0>("accumulators in VGPRs");
    run<true, 0>("accumulators in AGPRs");
    run<false, 1>("VGPR accumulators + W stream loads");
    run<true, 1>("AGPR accumulators + W stream loads");
    run<false, 2>("VGPR acc + 1 v_exp_f32 per MFMA");
    run<false, 3>("VGPR acc + 2 v_exp_f32 per MFMA");
    run<false, 4>("VGPR acc + 4 v_exp_f32 per MFMA");
    run<false, 5>("VGPR acc + 1 v_pk_add_f32 per MFMA");
    run<false, 6>("VGPR acc + 2 v_pk_add + 2 v_exp per MFMA");
    run<false, 7>("VGPR acc + 1 v_add_f32 per MFMA");
    run<true, 3>("AGPR acc + 2 v_exp_f32 per MFMA");
    return 0;
}
