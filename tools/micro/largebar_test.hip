// Can the host write device memory directly (large BAR)? hipDeviceAttributeIsLargeBar + a guarded try.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
__global__ void sum_kernel(const double *p, int n, double *out) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}
int main() {
    int large = 0;
    hipError_t e = hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, 0);
    printf("IsLargeBar attr: err=%d value=%d\n", (int)e, large);
    if (!large) return 0;
    double *d = nullptr, *out = nullptr;
    if (hipExtMallocWithFlags((void **)&d, 1 << 20, hipDeviceMallocFinegrained) != hipSuccess) { printf("finegrained malloc failed\n"); return 0; }
    hipMalloc(&out, 8);
    hipMemset(out, 0, 8);
    hipPointerAttribute_t at;
    hipPointerGetAttributes(&at, d);
    printf("ptr type=%d device=%d hostPointer=%p devicePointer=%p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 1024; i++) d[i] = 1.0;   // host store into device memory
    auto t1 = std::chrono::steady_clock::now();
    printf("host wrote 8 KB into device memory in %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count());
    {
        static double src[2048];
        for (int i = 0; i < 2048; i++) src[i] = 1.0;
        for (int rep = 0; rep < 3; rep++) {
            auto a = std::chrono::steady_clock::now();
            memcpy(d, src, 12288);
            auto b = std::chrono::steady_clock::now();
            printf("memcpy 12 KB host -> device memory: %.2f us\n", std::chrono::duration<double, std::micro>(b - a).count());
        }
        for (int i = 1024; i < 1536; i++) d[i] = 0.0;
    }
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, d, 1024, out);
    double h = 0;
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("kernel saw sum = %.1f (expect 1024)\n", h);
    return 0;
}
