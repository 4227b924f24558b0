// Shared helpers for the gfx950 kernels of libyolo2_hip.so (no torch, no host allocation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "yolo2_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define Y2_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return -(1000 + (int)_e);     \
    } while (0)

static inline hipStream_t y2_s(y2_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int y2_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline bool y2_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// library-internal cross-file entry points (hidden: not part of the C ABI)
__attribute__((visibility("hidden"))) int y2_internal_conv_grouped(const y2_conv_params* p, int groups, long long gx, long long gw, long long gy,
                                                                    y2_stream_t stream, size_t* ws_need);
__attribute__((visibility("hidden"))) int y2_internal_wgrad_grouped(const float* x, const float* dz, float* dw, long long M, int Cin, int Cout, int groups,
                                                                     long long gx, long long gz, long long gw, y2_stream_t stream);
__attribute__((visibility("hidden"))) int y2_internal_wgrad_needs_zero(long long M, int Cin, int Cout, int groups);
__attribute__((visibility("hidden"))) int y2_internal_wgrad_grouped_dz(const float* v, const float* dz_raw, const int32_t* tile_pix, float* dw, long long M, int Cin, int Cout,
                                                                         int ldz, int zW, unsigned long long dz_bytes, long long gx, long long gw, y2_stream_t stream);
__attribute__((visibility("hidden"))) int y2_internal_wino_conv(const y2_conv_params* p, y2_stream_t stream, size_t* ws_need);
__attribute__((visibility("hidden"))) int y2_internal_wino6_conv(const y2_conv_params* p, y2_stream_t stream, size_t* ws_need);
// gemm_split.hip: fp32-accurate GEMMs on the bf16 matrix pipe (three bf16 planes per operand, six plane products)
__attribute__((visibility("hidden"))) int y2_internal_gemm_split(const void* A, long long planeA, const void* B, long long planeB, float* C, long long M, int N, int K, int ldc, int groups, int planes,
                                                                  float out_scale, y2_stream_t stream);
__attribute__((visibility("hidden"))) int y2_internal_wino_input_split(const float* x, void* v, int B, int H, int W, int Cin, int ldx, int planes, float scale, y2_stream_t stream);
// fixed power-of-two operand scales of the fp16 split variant (Y2_ALGO_WINOGRAD_SPLIT_F16): V * 2^-4 (|V| up to 10^6 stays finite), U * 2^8
// (y2_split_f16x2's caller), product rescaled by 2^-4 in the GEMM epilogue
constexpr float Y2_F16_V_SCALE = 0.0625f, Y2_F16_U_SCALE = 256.f, Y2_F16_OUT_SCALE = 0.0625f;

// ---- per-device host-side caches (a process may touch several GPUs; symbol addresses and function attributes are per device)
constexpr int Y2_MAX_DEVICES = 64;
static inline int y2_current_device() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= Y2_MAX_DEVICES) return -1;
    return d;
}
// "this kernel may use the whole 160 KB LDS of a gfx950 CU", set once per (kernel instantiation, device)
struct Y2LdsAttr {
    bool done[Y2_MAX_DEVICES] = {};
    int ensure(const void* kern) {
        const int d = y2_current_device();
        if (d < 0) return Y2_EINVAL;
        if (!done[d]) {
            const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return -(1000 + (int)e);
            done[d] = true;
        }
        return 0;
    }
};

// ---- deterministic mode (det.hip; include/yolo2_hip.h: y2_set_deterministic): reductions write partials into `ws` and add them
// in a fixed tree instead of using atomics.  Process-global opt-in state; the scratch area is reused by consecutive calls on one stream.
struct Y2Det { int on; float* ws; size_t bytes; };
__attribute__((visibility("hidden"))) extern Y2Det y2_det;
__attribute__((visibility("hidden"))) int y2_det_reduce_f32(const float* part, int R, long long N, long long stride, double* out_d, float* out_f, hipStream_t s);

// ---- measurement hooks (prof.hip; include/yolo2_hip.h: y2_prof_*): when recording is on, every kernel launch of the library is
// bracketed by a HIP event pair on ITS launch stream and tagged with the multiply-add work it executes.
__attribute__((visibility("hidden"))) extern int y2_prof_on;
__attribute__((visibility("hidden"))) void y2_prof_begin(const char* name, hipStream_t s, double flops);
__attribute__((visibility("hidden"))) void y2_prof_end(hipStream_t s);
struct Y2ProfScope {
    hipStream_t s; bool on;
    Y2ProfScope(const char* name, hipStream_t s_, double flops) : s(s_), on(y2_prof_on != 0) { if (on) y2_prof_begin(name, s, flops); }
    ~Y2ProfScope() { if (on) y2_prof_end(s); }
};
#define Y2_LAUNCH(NAME, FLOPS, KERN, GRID, BLOCK, LDS, STREAM, ...)              \
    do {                                                                         \
        Y2ProfScope y2_prof_scope_(NAME, STREAM, FLOPS);                         \
        hipLaunchKernelGGL(KERN, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);         \
    } while (0)

constexpr int Y2_NUM_CU = 256;   // MI355X: 8 XCD x 32 CU
constexpr int Y2_NUM_XCD = 8;

// Bijective XCD-aware remap of a 1-D grid: hardware places block b on XCD b % 8, so give every XCD a
// contiguous chunk of the logical tile list (neighbouring tiles share operand panels -> L2 hits).
__device__ __forceinline__ int y2_xcd_remap(int bid, int nwg) {
    const int q = nwg / Y2_NUM_XCD, r = nwg % Y2_NUM_XCD;
    const int xcd = bid % Y2_NUM_XCD, i = bid / Y2_NUM_XCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// Exact unsigned division by a run-time constant without the ~30-instruction v_rcp sequence (Granlund-Montgomery,
// "branch-free" form): q = (t + ((n - t) >> 1)) >> (l - 1),  t = umulhi(n, mul),  l = ceil(log2 d),
// mul = floor(2^32 * (2^l - d) / d) + 1.  Valid for every 32-bit n; d == 1 is the identity.
struct y2_fastdiv {
    uint32_t mul, sh, d;
};
static inline y2_fastdiv y2_make_fastdiv(uint32_t d) {
    y2_fastdiv f; f.d = d; f.mul = 0; f.sh = 0;
    if (d <= 1) return f;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh = l - 1;
    return f;
}
__device__ __forceinline__ uint32_t y2_div(uint32_t n, const y2_fastdiv& f) {
    if (f.d <= 1) return n;
    const uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> 1)) >> f.sh;
}

#include <stdlib.h>
// Tile grid of the 4x4-tile forms.  Per image, a map of H rows takes ceil(H / 4) tile rows: 13 rows pay for 16 (the 13x13 layers at 416x416: 34 % of the
// multiply-adds of their 36 GEMMs are spent on rows and columns that do not exist).  "Mosaic" tiling lays the batch's images out as ONE image - gy rows of gx
// images with a single zero row / zero column between neighbours; that line IS the bottom (right) padding of one image and the top (left) padding of the next,
// so every 3x3 neighbourhood of the mosaic is the neighbourhood of its own image - and cuts THAT into 4x4 tiles: 64 images of 13x13 as 8 x 8 -> 111 x 111
// pixels -> 28 x 28 = 784 tiles instead of 64 x 16 = 1024 (-23 %; 26x26: 3136 -> 2916).  Only the tile -> pixel maps of the three transform kernels change
// (mosaic row r -> image row r / (H + 1), pixel row r % (H + 1); a row H does not exist: reads give zero, writes are dropped; columns alike); the GEMMs see
// fewer rows.  gx is chosen to minimise the tile count; the per-image grid stays when nothing is smaller (Y2_WINO6_TALL=0: always).
struct Wino6Grid { int th, tw, tall, wide, gx; long long T; };
inline Wino6Grid wino6_grid(int B, int H, int W) {
    static const bool allow = getenv("Y2_WINO6_TALL") == nullptr || atoi(getenv("Y2_WINO6_TALL")) != 0;
    Wino6Grid g;
    g.tw = (W + 3) / 4;
    g.th = (H + 3) / 4;
    g.tall = g.wide = 0;
    g.gx = 1;
    g.T = (long long)B * g.th * g.tw;
    if (!allow) return g;
    for (int gx = 1; gx <= B && gx <= 4096; ++gx) {
        const long long gy = (B + gx - 1) / gx;
        const long long rows = gy * (H + 1) - 1, cols = (long long)gx * (W + 1) - 1;
        const long long th = (rows + 3) / 4, tw = (cols + 3) / 4;
        if (rows >= 0x7fffffff || cols >= 0x7fffffff) continue;
        if (th * tw < g.T) { g.T = th * tw; g.th = (int)th; g.tw = (int)tw; g.tall = H + 1; g.wide = gx > 1 ? W + 1 : 0; g.gx = gx; }
    }
    return g;
}
