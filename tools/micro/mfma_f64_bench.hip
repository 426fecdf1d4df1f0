// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (the guide has no fp64
// rows). One wave per SIMD (256 threads per CU-sized block), independent accumulators, s_memtime cycles.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_bench.hip -o gpurun_out/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_kernel(double *out, long long *cyc, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void fma_kernel(double *out, long long *cyc, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(acc[i], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 1024 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 512}) {
        {
            hipLaunchKernelGGL((mfma_kernel<8>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e0);
            hipLaunchKernelGGL((mfma_kernel<8>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double flops = (double)blocks * 4 * iters * 8 * 2048.0;
            printf("mfma_f64_16x16x4 blocks=%d: %.1f 'cycles'(100MHz ticks?) per MFMA per wave by counter=%lld; %.3f ms -> %.2f TFLOP/s; ns per MFMA per wave = %.2f\n",
                   blocks, (double)c / (iters * 8.0), c, ms, flops / ms / 1e9, ms * 1e6 / (iters * 8.0));
        }
        {
            hipLaunchKernelGGL((fma_kernel<16>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e0);
            hipLaunchKernelGGL((fma_kernel<16>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 256 * iters * 16 * 2.0;
            printf("v_fma_f64 blocks=%d: %.3f ms -> %.2f TFLOP/s; ns per wave-FMA = %.3f\n", blocks, ms, flops / ms / 1e9, ms * 1e6 / (iters * 16.0));
        }
    }
    return 0;
}
