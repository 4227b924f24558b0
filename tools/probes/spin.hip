// tools/contention.py: a kernel that holds `blocks` workgroup slots for a fixed wall-clock time, the way RCCL's channel kernels hold CUs
// while a data-parallel backward runs.  Built by tools/probes/build_spin.sh into tools/probes/libspin.so.
#include <hip/hip_runtime.h>
__global__ void spin_kernel(long long ticks, unsigned long long* sink) {
    __shared__ float hold[8192];                           // 32 KB of LDS per workgroup, like a collective's staging buffers
    hold[threadIdx.x] = (float)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    unsigned long long n = 0;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) { ++n; __builtin_amdgcn_s_sleep(8); }
    if (sink != nullptr && threadIdx.x == 0 && hold[1] < 0.f) sink[blockIdx.x] = n;
}
extern "C" int spin_launch(int blocks, int threads, double milliseconds, void* stream) {
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), (long long)(milliseconds * 1e5), (unsigned long long*)nullptr);
    return (int)hipGetLastError();
}
