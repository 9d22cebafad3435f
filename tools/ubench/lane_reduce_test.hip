// Checks the cross-lane primitives used by lane_reduce16 (skm_gmm.hip) on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ void k(float *o)
{
    const int lane = threadIdx.x;
    float a = (float)lane, b = 100.f + lane;
    float lo = a, hi = b;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
    o[0 * 64 + lane] = lo;
    o[1 * 64 + lane] = hi;
    o[2 * 64 + lane] = dpp_f32<0x128>(a);
    o[3 * 64 + lane] = dpp_f32<0x141>(a);
    o[4 * 64 + lane] = dpp_f32<0x4E>(a);
    o[5 * 64 + lane] = dpp_f32<0xB1>(a);
}
int main()
{
    float *d, h[6 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[6] = {"swap r0", "swap r1", "ror8", "halfmirror", "qp 2301", "qp 1032"};
    for (int i = 0; i < 6; ++i) {
        printf("%-10s", nm[i]);
        for (int l = 0; l < 64; ++l) printf(" %g", h[i * 64 + l]);
        printf("\n");
    }
    return 0;
}
