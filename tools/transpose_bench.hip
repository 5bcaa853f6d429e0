// Micro-benchmark (tuning aid): the 6-stage Kronecker transition of the sweep with its three lane-bit stages done
//   MODE 0: as DPP exchanges (what fb_fast_kernel does: 48 v_mov_dpp + 48 FMAs per transition), or
//   MODE 1: after a transpose of the wave's 8 x 8 (lane, register) blocks through LDS, as register stages
//           (48 FMAs + 8 ds_write_b64 + 8 ds_read_b64; the layout alternates from marker to marker), or
//   MODE 3: a LOWER BOUND of doing them on the matrix cores: a chain's 8 x 8 (lane, register) block times the 8 x 8
//           Kronecker matrix of the three lane-held state bits is a dense product; over the wave it is (8 x 8) . (8 x 64),
//           which v_mfma_f64_16x16x4 can only tile as 4 output tiles (M = 16 holds the 8 rows twice: half of every tile is
//           waste) x 2 steps of K = 4 -- 8 MFMAs per transition.  The operands would also have to be brought into the
//           MFMA lane layout (lane n + 16 k holds B[k][n]; here a chain's l sits in lane bits 0-2), i.e. exactly the
//           cross-lane movement the DPP form spends its time on; this mode issues the 8 MFMAs on whatever the registers
//           hold (results are NOT the transition) and so measures the matrix-core time alone,
// at the sweep's occupancy (2 waves per SIMD), as one dependent chain per wave with an emission-like multiply between
// transitions.  build: hipcc -O3 --offload-arch=gfx950 -o tools/transpose_bench tools/transpose_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double dpp_mov(double v, int)
{
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_all(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void reg_stages(double (&a)[8], double t0, double t1)
{
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const double x = a[j], y = a[j + 1];
        a[j]     = fma(t1, y, x);
        a[j + 1] = fma(t1, x, y);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j & 2) continue;
        const double x = a[j], y = a[j + 2];
        a[j]     = fma(t0, y, x);
        a[j + 2] = fma(t0, x, y);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double x = a[j], y = a[j + 4];
        a[j]     = fma(t0, y, x);
        a[j + 4] = fma(t0, x, y);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters, double t0, double t1)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    constexpr int RS = (MODE == 2) ? 10 : 9;               // row stride (doubles): spreads the chains over the banks
    double* T = lds + wib * (64 * 10 + 8);
    const int s = lane >> 3, l = lane & 7;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = 1.0 / 64 + 1e-3 * (lane + j);
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = dpp_all<0xB1>(a[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = fma(t1, q[j], a[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = dpp_all<0x4E>(a[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = fma(t0, q[j], a[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = dpp_all<0x141>(a[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = fma(t0, q[j], a[j]);
            reg_stages(a, t0, t1);
        } else if (MODE == 3) {
            typedef double d4 __attribute__((ext_vector_type(4)));
            const double kc = (lane & 1) ? t0 : 1.0;           // a Kronecker coefficient: what the A operand would hold
#pragma unroll
            for (int tile = 0; tile < 4; tile++) {
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kc, a[2 * tile], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kc, a[2 * tile + 1], acc, 0, 0, 0);
                a[2 * tile]     = acc.x + acc.z;            // (half of every tile is the duplicate)
                a[2 * tile + 1] = acc.y + acc.w;
            }
            reg_stages(a, t0, t1);
        } else {
            reg_stages(a, t0, t1);
            // transpose: (lane l, register j) -> (lane j, register l) inside every chain
#pragma unroll
            for (int j = 0; j < 8; j++) T[(s * 8 + j) * RS + l] = a[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const double2 v = *(const double2*)(T + (s * 8 + l) * RS + j);
                    a[j] = v.x;
                    a[j + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = T[(s * 8 + l) * RS + j];
            }
            reg_stages(a, t0, t1);
        }
        // emission-like multiply and a cheap renormalisation so that the chain stays finite
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] *= 0.124;
    }
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) acc += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
float run(double* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t shm = 68 * 1024;                           // 2 blocks per CU, as the sweep
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), shm, 0, d, 10, 0.01, 0.012);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), shm, 0, d, iters, 0.01, 0.012);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms;
}

int main()
{
    double* d;
    hipMalloc(&d, 512 * 256 * 8);
    const int iters = 200000;
    const float a = run<0>(d, iters), b = run<1>(d, iters), c = run<2>(d, iters), m = run<3>(d, iters);
    // 512 blocks x 4 waves on 256 CUs x 4 SIMDs = 2 waves per SIMD
    printf("DPP lane stages        %8.2f ms   %.1f ns per transition per wave (2 waves / SIMD)\n", a, a * 1e6 / iters);
    printf("LDS transpose          %8.2f ms   %.1f ns per transition per wave (2 waves / SIMD)\n", b, b * 1e6 / iters);
    printf("LDS transpose b128 rd  %8.2f ms   %.1f ns per transition per wave (2 waves / SIMD)\n", c, c * 1e6 / iters);
    printf("MFMA lower bound       %8.2f ms   %.1f ns per transition per wave (8 x v_mfma_f64_16x16x4, no operand re-layout)\n", m, m * 1e6 / iters);
    return 0;
}
