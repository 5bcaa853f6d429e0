// Micro-benchmark (tuning aid): issue cost of v_permlane32_swap / v_permlane16_swap against DPP moves and FMAs.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/swap_bench tools/swap_bench.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    double   f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {          // 8 DPP moves
                a0 = __builtin_amdgcn_mov_dpp(a0, 0xB1, 0xF, 0xF, true);
                a1 = __builtin_amdgcn_mov_dpp(a1, 0xB1, 0xF, 0xF, true);
                a2 = __builtin_amdgcn_mov_dpp(a2, 0x4E, 0xF, 0xF, true);
                a3 = __builtin_amdgcn_mov_dpp(a3, 0x4E, 0xF, 0xF, true);
                a4 = __builtin_amdgcn_mov_dpp(a4, 0x141, 0xF, 0xF, true);
                a5 = __builtin_amdgcn_mov_dpp(a5, 0x141, 0xF, 0xF, true);
                a6 = __builtin_amdgcn_mov_dpp(a6, 0xB1, 0xF, 0xF, true);
                a7 = __builtin_amdgcn_mov_dpp(a7, 0xB1, 0xF, 0xF, true);
            } else if (MODE == 1) {   // 4 permlane32 swaps (8 registers touched)
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
            } else if (MODE == 2) {   // 4 permlane16 swaps
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
            } else {                  // 4 independent f64 FMAs
                f0 = fma(f0, 1.0000001, 0.5);
                f1 = fma(f1, 1.0000001, 0.5);
                f2 = fma(f2, 1.0000001, 0.5);
                f3 = fma(f3, 1.0000001, 0.5);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (double)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) + f0 + f1 + f2 + f3;
}

template <int MODE>
float run(double* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double* d;
    hipMalloc(&d, 2048 * 256 * 8);
    const int iters = 20000;
    // 2048 blocks x 4 waves over 256 CUs x 4 SIMDs = 8 waves per SIMD; instructions per wave = iters * 8 * n
    const char* names[4] = {"8 dpp movs", "4 permlane32_swap", "4 permlane16_swap", "4 f64 fma"};
    float ms[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
    int n[4] = {8, 4, 4, 4};
    for (int m = 0; m < 4; m++) {
        double inst = (double)iters * 8 * n[m] * 8;                    // per SIMD
        printf("%-20s %8.2f ms   %.2f ns per instruction per SIMD\n", names[m], ms[m], ms[m] * 1e6 / inst);
    }
    return 0;
}
