// Probe of the gfx950 LDS-direct load (global_load_lds_dwordx4): where does lane L's 16 bytes land?
// build: hipcc -O2 --offload-arch=gfx950 -o tools/lds_dma_test tools/lds_dma_test.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const double* src, double* dst)
{
    __shared__ __attribute__((aligned(16))) double buf[2][256];
    const int lane = threadIdx.x & 63;
    const int w    = threadIdx.x >> 6;
    for (int i = lane; i < 256; i += 64) buf[w][i] = -1.0;
    __syncthreads();
    // lane L fetches the pair (2*(63-L), 2*(63-L)+1): a permuted source shows which lane wrote where
    const double* g = src + w * 128 + 2 * (63 - lane);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)&buf[w][8], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0) lgkmcnt(0)
    __syncthreads();
    for (int i = lane; i < 256; i += 64) dst[w * 256 + i] = buf[w][i];
}

int main()
{
    std::vector<double> h(256);
    for (int i = 0; i < 256; i++) h[i] = i;
    double *src, *dst;
    hipMalloc(&src, 256 * 8);
    hipMalloc(&dst, 512 * 8);
    hipMemcpy(src, h.data(), 256 * 8, hipMemcpyHostToDevice);
    probe<<<1, 128>>>(src, dst);
    std::vector<double> o(512);
    hipMemcpy(o.data(), dst, 512 * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; w++) {
        printf("wave %d:", w);
        for (int i = 0; i < 144; i++) printf(" %g", o[w * 256 + i]);
        printf("\n");
    }
    return 0;
}
