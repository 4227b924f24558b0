// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for out-of-range lanes (or skip them)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + (threadIdx.x / 64) * 256), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 v = *reinterpret_cast<f32x4*>(smem + threadIdx.x * 4);
    *reinterpret_cast<f32x4*>(y + threadIdx.x * 4) = v;
}
int main() {
    float *x, *y; float h[1024];
    hipMalloc(&x, 4096); hipMalloc(&y, 4096);
    for (int i = 0; i < 1024; ++i) h[i] = 100.f + i;
    hipMemcpy(x, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, x, y, 1024);
    hipMemcpy(h, y, 4096, hipMemcpyDeviceToHost);
    printf("lane0: %g %g %g %g | lane1 (OOB): %g %g %g %g | lane2: %g | lane65(OOB): %g lane66: %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[65*4], h[66*4]);
    return 0;
}
