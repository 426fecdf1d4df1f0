// Diagnostic: issue rate and dependent latency of v_fma_f64 on one wavefront / four wavefronts of a workgroup,
// in s_memtime ticks and in wall_clock64 (100 MHz) ticks.  hipcc --offload-arch=gfx950 -O3 -o dp_rate dp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, long long *t, int iters, int mode) {
    double a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-3 + i;
    const double m = 1.0000001, c = 1e-9;
    __syncthreads();
    long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {            // 16 independent chains
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_fma(a[i], m, c);
    } else if (mode == 1) {     // one dependent chain
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) a[0] = __builtin_fma(a[0], m, c);
    } else {                    // dependent chain of divisions
        for (int it = 0; it < iters; it++) a[0] = 1.0 / (a[0] + 1.5);
    }
    long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    double s = 0; for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    double *o; long long *t; hipMalloc(&o, 1 << 20); hipMallocManaged(&t, 64);
    for (int threads : {64, 256, 512}) for (int mode = 0; mode < 3; mode++) {
        const int iters = 10000;
        k<<<1, threads>>>(o, t, iters, mode); hipDeviceSynchronize();
        k<<<1, threads>>>(o, t, iters, mode); hipDeviceSynchronize();
        const double n = mode == 2 ? iters : 16.0 * iters;
        printf("threads %d mode %s: %.2f memtime ticks per op, %.3f ns per op (wall_clock64 at 100 MHz)\n", threads, mode == 0 ? "fma x16 independent" : mode == 1 ? "fma dependent" : "1/(x+1.5) dependent", t[0] / n, t[1] * 10.0 / n);
    }
    return 0;
}
