// Accuracy probe (GPU box): the device logit() of cnf2_update.h against the host's long-double logarithm on a sweep of
// x over [1e-7, 1 - 1e-7].  build: hipcc -O3 --offload-arch=gfx950 -I cnf2freq_amd/csrc -o /tmp/logit_check tools/logit_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "cnf2_update.h"

__global__ void k(double* o, const double* a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = cnf2::logit(a[i]);
}

int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), y(n);
    for (int i = 0; i < n; i++) {
        const double u = (i + 0.5) / n;                              // dense near both ends and in the middle
        x[i] = (i % 3 == 0) ? 1e-7 + u * 1e-3 : ((i % 3 == 1) ? 1.0 - 1e-7 - u * 1e-3 : 1e-7 + u * (1 - 2e-7));
    }
    double *dx, *dy;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dy, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dy, dx, n);
    hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
    double worst_rel = 0, worst_abs = 0;
    for (int i = 0; i < n; i++) {
        const long double ref = logl((long double)x[i] / (1.0L - (long double)x[i]));
        const double      d = fabs((double)(y[i] - ref));
        worst_abs = fmax(worst_abs, d);
        if (fabsl(ref) > 1e-3L) worst_rel = fmax(worst_rel, d / (double)fabsl(ref));
    }
    printf("device logit vs long double: worst absolute error %.3g, worst relative error (|logit| > 1e-3) %.3g\n", worst_abs, worst_rel);
    return (worst_abs < 1e-14 && worst_rel < 1e-12) ? 0 : 1;   // relative: the rounding of x / (1 - x) near x = 1/2, as in the library form
}
