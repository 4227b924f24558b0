// tools/mfma_peak.py: what the fp32 matrix pipe of this chip sustains, and whether vector instructions run beside it.
// One workgroup = 4 waves (one per SIMD); every wave issues `iters` x 4 v_mfma_f32_32x32x2_f32 into 4 independent accumulators, with V independent
// v_fma_f32 behind every MFMA (V = 0, 4, 8, 16; 16 vector instructions are 64 issue cycles - exactly one MFMA's 16 passes).  No memory traffic.
// Built by tools/probes/build_spin.sh into tools/probes/libmfma_peak.so.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    float f[16];
    for (int j = 0; j < 16; ++j) f[j] = (float)j;
    const float c = 0.999f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v]) : "v"(c));
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int j = 0; j < 16; ++j) s += f[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// returns the milliseconds of one launch (after a warm-up launch), or a negative HIP error
extern "C" float mfma_peak_run(int valu_per_mfma, int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = -1.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        switch (valu_per_mfma) {
            case 0: hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            case 2: hipLaunchKernelGGL(mfma_peak_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            case 4: hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            case 8: hipLaunchKernelGGL(mfma_peak_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            case 12: hipLaunchKernelGGL(mfma_peak_kernel<12>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
            default: hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (hipGetLastError() != hipSuccess) return -2.f;
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}
