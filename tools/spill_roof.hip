// Measurement aid (not part of the product): HBM rate of the sweep's spill access pattern alone --
// every wave writes len rows of 4160 B front to back, then reads them back to front -- to know the
// ceiling that the forward-backward kernel's traffic can reach on this GPU.
//   hipcc -O3 --offload-arch=gfx950 tools/spill_roof.hip -o /tmp/spill_roof && /tmp/spill_roof
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void spill_pattern(double* spill, size_t stride, int len, int reps, double* sink)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double*   base = spill + (size_t)wave * stride;
    double    acc  = 0.0;
    for (int r = 0; r < reps; r++) {
        for (int m = 0; m < len; m++) {
            double* sp = base + (size_t)m * 520 + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) sp[j * 64] = (double)(m + j) + acc;
            if (lane < 8) sp[512 - lane + lane] = 1.0;
        }
        for (int m = len - 1; m >= 0; m--) {
            const double* sp = base + (size_t)m * 520 + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) acc += sp[j * 64];
            acc += sp[512 - lane + (lane >> 3)];
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

// same traffic with 16-byte accesses: lane holds register pairs (j, j+1) side by side
__global__ __launch_bounds__(256) void spill_pattern16(double* spill, size_t stride, int len, int reps, double* sink)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double*   base = spill + (size_t)wave * stride;
    double    acc  = 0.0;
    for (int r = 0; r < reps; r++) {
        for (int m = 0; m < len; m++) {
            double2* sp = (double2*)(base + (size_t)m * 520) + lane;
#pragma unroll
            for (int j = 0; j < 4; j++) sp[j * 64] = make_double2((double)(m + j) + acc, acc);
            if (lane < 8) base[(size_t)m * 520 + 512 + lane] = 1.0;
        }
        for (int m = len - 1; m >= 0; m--) {
            const double2* sp = (const double2*)(base + (size_t)m * 520) + lane;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double2 v = sp[j * 64];
                acc += v.x + v.y;
            }
            acc += base[(size_t)m * 520 + 512 + (lane >> 3)];
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

// marker-major layout: rows of all resident waves for the same marker are adjacent, so that the
// waves (which advance at the same pace) sweep one contiguous window through memory
__global__ __launch_bounds__(256) void spill_pattern_mm(double* spill, size_t nwave, int len, int reps, double* sink)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double    acc  = 0.0;
    for (int r = 0; r < reps; r++) {
        for (int m = 0; m < len; m++) {
            double* sp = spill + ((size_t)m * nwave + wave) * 520 + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) sp[j * 64] = (double)(m + j) + acc;
            if (lane < 8) sp[512] = 1.0;
        }
        for (int m = len - 1; m >= 0; m--) {
            const double* sp = spill + ((size_t)m * nwave + wave) * 520 + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) acc += sp[j * 64];
            acc += sp[512 - lane + (lane >> 3)];
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

// ring: every wave rewrites and re-reads the same K rows (recompute-per-tile scheme): does the
// traffic stay in L2 / Infinity Cache?
__global__ __launch_bounds__(256) void spill_ring(double* spill, int K, int len, int reps, double* sink)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    double*   base = spill + (size_t)wave * K * 520;
    double    acc  = 0.0;
    for (int r = 0; r < reps; r++) {
        for (int t = 0; t < len / K; t++) {
            for (int m = 0; m < K; m++) {
                double* sp = base + (size_t)m * 520 + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) sp[j * 64] = (double)(m + j) + acc;
            }
            for (int m = K - 1; m >= 0; m--) {
                const double* sp = base + (size_t)m * 520 + lane;
#pragma unroll
                for (int j = 0; j < 8; j++) acc += sp[j * 64];
            }
        }
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main()
{
    const int len = 2501, reps = 20;
    for (int blocks : {256, 512, 768, 1024}) {
        const size_t stride = (size_t)len * 520;
        double *spill, *sink;
        hipMalloc(&spill, (size_t)blocks * 4 * stride * 8);
        hipMalloc(&sink, 8);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        spill_pattern<<<blocks, 256>>>(spill, stride, len, 1, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spill_pattern<<<blocks, 256>>>(spill, stride, len, reps, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double bytes = (double)blocks * 4 * len * 4160.0 * 2 * reps;
        printf("blocks %4d  waves/CU %2d  8B: %.1f ms  %.2f TB/s", blocks, blocks * 4 / 256, ms, bytes / ms / 1e9);
        spill_pattern16<<<blocks, 256>>>(spill, stride, len, 1, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spill_pattern16<<<blocks, 256>>>(spill, stride, len, reps, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("   16B: %.1f ms  %.2f TB/s", ms, bytes / ms / 1e9);
        spill_pattern_mm<<<blocks, 256>>>(spill, (size_t)blocks * 4, len, 1, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spill_pattern_mm<<<blocks, 256>>>(spill, (size_t)blocks * 4, len, reps, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("   marker-major: %.1f ms  %.2f TB/s (write+read)\n", ms, bytes / ms / 1e9);
        for (int K : {4, 8, 16, 32}) {
            spill_ring<<<blocks, 256>>>(spill, K, len, 1, sink);
            hipDeviceSynchronize();
            hipEventRecord(a);
            spill_ring<<<blocks, 256>>>(spill, K, len, reps, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            double by = (double)blocks * 4 * (len / K * K) * 4096.0 * 2 * reps;
            printf("      ring K=%2d (%5.1f MB total): %.1f ms  %.2f TB/s\n", K, blocks * 4.0 * K * 4160 / 1e6, ms, by / ms / 1e9);
        }
        hipFree(spill);
        hipFree(sink);
    }
    return 0;
}
