// Round-trip latency of the host <-> persistent-kernel signalling used by the worker (csrc/spg_kernels.hip):
// the host stores a sequence number into a doorbell, a spinning wavefront echoes it into a word of pinned host memory.
// Variants: doorbell in fine-grained DEVICE memory written through the PCIe BAR vs in pinned HOST memory (the device
// polls over PCIe); device poll = system-scope relaxed load, or the same followed by an acquire (cache invalidation);
// echo = plain store + system fence vs system-scope (write-through) store. Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

__global__ void echo_kernel(const unsigned long long *bell, unsigned long long *echo, int mode, int iters, int busy_wgs_poll) {
    // workgroup 0 echoes; the others (if any) just poll the bell like idle workers do
    unsigned long long last = 0;
    const bool echoer = blockIdx.x == 0;
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    for (;;) {
        unsigned long long v = (mode & 1) ? __hip_atomic_load(bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
                                          : __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == ~0ULL) return;
        if (v != last) {
            last = v;
            if (echoer) {
                if (mode & 2) { *echo = v; __threadfence_system(); }
                else __hip_atomic_store(echo, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            t0 = wall_clock64();
        } else if (wall_clock64() - t0 > 500000000LL) return;   // 5 s without a change: leave
        if (!echoer) __builtin_amdgcn_s_sleep(8);
    }
}

static double run(const char *name, volatile unsigned long long *bell_host, unsigned long long *bell_dev, int mode, int wgs) {
    unsigned long long *echo_h = nullptr, *echo_d = nullptr;
    hipHostMalloc((void **)&echo_h, 64, hipHostMallocMapped);
    hipHostGetDevicePointer((void **)&echo_d, echo_h, 0);
    *echo_h = 0;
    *bell_host = 0;
#if defined(__x86_64__)
    _mm_sfence();
#endif
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(echo_kernel, dim3(wgs), dim3(64), 0, s, bell_dev, echo_d, mode, 0, 0);
    std::vector<double> us;
    volatile unsigned long long *eh = echo_h;
    for (unsigned long long i = 1; i <= 2000; i++) {
        auto a = std::chrono::steady_clock::now();
        *bell_host = i;
#if defined(__x86_64__)
        _mm_sfence();
#endif
        while (*eh != i) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 2.0) { printf("%s: timeout at %llu\n", name, i); goto out; }
        }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
    }
out:
    *bell_host = ~0ULL;
#if defined(__x86_64__)
    _mm_sfence();
#endif
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    hipHostFree(echo_h);
    if (us.size() < 100) return -1;
    std::sort(us.begin() + 100, us.end());
    size_t n = us.size() - 100;
    printf("%-72s median %6.2f us  p10 %6.2f  p90 %6.2f  (%d workgroups polling)\n", name, us[100 + n / 2], us[100 + n / 10], us[100 + 9 * n / 10], wgs);
    return us[100 + n / 2];
}

int main() {
    unsigned long long *bar = nullptr;
    int large = 0;
    hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, 0);
    if (large && hipExtMallocWithFlags((void **)&bar, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
        for (int wgs : {1, 64, 256}) {
            run("bell in fine-grained device memory (BAR store), relaxed poll, sys store echo", bar, bar, 0, wgs);
            run("bell in fine-grained device memory (BAR store), acquire poll, sys store echo", bar, bar, 1, wgs);
        }
        run("bell in fine-grained device memory (BAR store), relaxed poll, plain store + fence echo", bar, bar, 2, 1);
    } else printf("no large BAR\n");
    unsigned long long *hb = nullptr, *hbd = nullptr;
    hipHostMalloc((void **)&hb, 4096, hipHostMallocMapped);
    hipHostGetDevicePointer((void **)&hbd, hb, 0);
    for (int wgs : {1, 64, 256}) run("bell in pinned host memory (device polls over PCIe), relaxed poll, sys store echo", hb, hbd, 0, wgs);
    run("bell in pinned host memory, acquire poll, sys store echo", hb, hbd, 1, 1);
    return 0;
}
