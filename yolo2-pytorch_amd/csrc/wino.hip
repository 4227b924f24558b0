// wino.hip — Winograd F(2x2,3x3) path for the 3x3 / stride-1 / same-padding convolutions of the Darknet stages (gfx950).
//
// Replaces the same nn.Conv2d + BatchNorm2d + LeakyReLU (+ MaxPool2d) blocks as conv_fwd.hip (model/yolo2.py:50-68,
// 76-113); selected per layer by y2_conv_params.algo = Y2_ALGO_WINOGRAD when it measures faster (deep layers: Cin >= 128).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 2x2 output tile, 4x4 input patch d, 3x3 filter g  (Lavin & Gray 2015)
//
//   B^T = | 1  0 -1  0 |     G = | 1    0    0  |     A^T = | 1  1  1  0 |
//         | 0  1  1  0 |         | .5   .5   .5 |           | 0  1 -1 -1 |
//         | 0 -1  1  0 |         | .5  -.5   .5 |
//         | 0  1  0 -1 |         | 0    0    1  |
//
// Three stages, all operands fp32:
//   1. wino_input_kernel:   V[p][t][ci] = (B^T d B)[p]      p = 4*xi + nu (16 positions), t = (b, ty, tx) tile index
//   2. 16 independent GEMMs M[p] = V[p] (T x Cin) * U[p]^T (Cin x Cout) in ONE launch of conv_fwd_dma_kernel (its grouped
//      mode): the same LDS-DMA / MFMA 32x32x2 fp32 pipeline, 16*T*Cin*Cout MACs instead of 9*H*W*Cin*Cout (2.25x fewer
//      for even H, W)
//   3. wino_output_kernel:  y = act(scale * (A^T M A) + shift), 2x2 pixels per tile, optional in-thread 2x2 max-pool and
//      per-channel sum / sum of squares of the raw output (training-mode BatchNorm statistics)
// Stages 1 and 3 are streaming kernels (HBM/Infinity-Cache bound: V is 4x the input, M is 4x the output).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

struct WinoInArgs {
    const float* x;
    float* v;
    int32_t* tile_pix;     // optional [T]: output pixel index (b*H + 2*ty)*W + 2*tx of tile t | (2*ty+1 < H) << 30 | (2*tx+1 < W) << 31
    int32_t* sched;        // optional [8]: per-XCD tile counters of the fused kernel that follows, set to sched_init here
    int sched_init;
    int B, H, W, Cin, ldx, th, tw, T, c4n;
    y2_fastdiv d_c4, d_tt, d_tw;
};

__global__ __launch_bounds__(256) void wino_input_kernel(const WinoInArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (a.sched != nullptr && idx < (uint32_t)Y2_NUM_XCD) a.sched[idx] = a.sched_init;
    const uint32_t t = y2_div(idx, a.d_c4);
    if (t >= (uint32_t)a.T) return;
    const int c4 = (int)(idx - t * (uint32_t)a.c4n);
    const int b = (int)y2_div(t, a.d_tt);
    const int r = (int)t - b * a.th * a.tw;
    const int ty = (int)y2_div((uint32_t)r, a.d_tw);
    const int tx = r - ty * a.tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    if (a.tile_pix != nullptr && c4 == 0)     // decode table for the fused GEMM + output-transform kernel (one entry per tile)
        a.tile_pix[t] = (int32_t)(((uint32_t)((b * a.H + 2 * ty) * a.W + 2 * tx)) | (2 * ty + 1 < a.H ? 0x40000000u : 0u) | (2 * tx + 1 < a.W ? 0x80000000u : 0u));
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = y0 + i;
        const bool yok = (unsigned)yy < (unsigned)a.H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = x0 + j;
            const bool ok = yok && (unsigned)xx < (unsigned)a.W;
            d[i][j] = ok ? *reinterpret_cast<const f32x4*>(a.x + ((size_t)(b * a.H + yy) * a.W + xx) * a.ldx + 4 * c4) : zero;
        }
    }
    f32x4 s[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = d[0][j] - d[2][j];
        s[1][j] = d[1][j] + d[2][j];
        s[2][j] = d[2][j] - d[1][j];
        s[3][j] = d[1][j] - d[3][j];
    }
    float* dst = a.v + (size_t)t * a.Cin + 4 * c4;
    const size_t plane = (size_t)a.T * a.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(dst + (4 * i + 0) * plane) = s[i][0] - s[i][2];
        *reinterpret_cast<f32x4*>(dst + (4 * i + 1) * plane) = s[i][1] + s[i][2];
        *reinterpret_cast<f32x4*>(dst + (4 * i + 2) * plane) = s[i][2] - s[i][1];
        *reinterpret_cast<f32x4*>(dst + (4 * i + 3) * plane) = s[i][1] - s[i][3];
    }
}

// The decode table alone (Y2_ALGO_WINOGRAD_IMPLICIT has no input-transform kernel to write it).
__global__ __launch_bounds__(256) void wino_tile_table_kernel(int32_t* tile_pix, int T, int H, int W, int th, int tw, y2_fastdiv d_tt, y2_fastdiv d_tw, int32_t* sched, int sched_init) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (sched != nullptr && t < (uint32_t)Y2_NUM_XCD) sched[t] = sched_init;
    if (t >= (uint32_t)T) return;
    const int b = (int)y2_div(t, d_tt);
    const int r = (int)t - b * th * tw;
    const int ty = (int)y2_div((uint32_t)r, d_tw);
    const int tx = r - ty * tw;
    tile_pix[t] = (int32_t)(((uint32_t)((b * H + 2 * ty) * W + 2 * tx)) | (2 * ty + 1 < H ? 0x40000000u : 0u) | (2 * tx + 1 < W ? 0x80000000u : 0u));
}

struct WinoOutArgs {
    const float* m;
    const float* scale;
    const float* shift;
    float* y;
    float* y_pool;
    double* stats;
    int B, H, W, Cout, ldy, coff, ldp, poff, th, tw, T, n4n, loop;
    float slope;
    y2_fastdiv d_tt, d_tw;
};

// Block = blockDim.x channel quads x blockDim.y tiles (256 threads); thread (x, y) handles channel quad
// blockIdx.y*blockDim.x + x of the tiles (blockIdx.x*loop + i)*blockDim.y + y, i < loop.  STATS: per-channel sum / sum of
// squares of the raw convolution output z (training-mode BatchNorm, model/yolo2.py:59) over the VALID pixels, reduced over
// the block's tiles in registers + LDS and added with one fp64 atomic pair per channel per block into replicated
// accumulators (Y2_STATS_REPL, as conv_fwd.hip's epilogue).
template <bool STATS>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoOutArgs a) {
    __shared__ float red[STATS ? 256 * 8 : 1];
    const int n4 = blockIdx.y * blockDim.x + threadIdx.x;
    const bool nok = n4 < a.n4n;
    const int n = 4 * n4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (nok && a.scale != nullptr) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
    if (nok && a.shift != nullptr) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const size_t plane = (size_t)a.T * a.Cout;
    for (int it = 0; it < a.loop; ++it) {
        const int t = (blockIdx.x * a.loop + it) * blockDim.y + threadIdx.y;
        if (t >= a.T || !nok) continue;
        const int b = (int)y2_div((uint32_t)t, a.d_tt);
        const int r = t - b * a.th * a.tw;
        const int ty = (int)y2_div((uint32_t)r, a.d_tw);
        const int tx = r - ty * a.tw;
        const float* src = a.m + (size_t)t * a.Cout + n;
        f32x4 m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = *reinterpret_cast<const f32x4*>(src + (4 * i + j) * plane);
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j] - m[3][j];
        }
        f32x4 o[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            o[i][0] = s[i][0] + s[i][1] + s[i][2];
            o[i][1] = s[i][1] - s[i][2] - s[i][3];
        }
        const bool y1 = 2 * ty + 1 < a.H, x1 = 2 * tx + 1 < a.W;
        if (STATS) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if ((i == 1 && !y1) || (j == 1 && !x1)) continue;
                    s1 += o[i][j];
                    s2 += o[i][j] * o[i][j];
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = o[i][j][e] * sc[e] + sh[e];
                    o[i][j][e] = u > 0.f ? u : u * a.slope;
                }
        if (a.y != nullptr) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 1 && !y1) break;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && !x1) break;
                    *reinterpret_cast<f32x4*>(a.y + ((size_t)(b * a.H + 2 * ty + i) * a.W + 2 * tx + j) * a.ldy + a.coff + n) = o[i][j];
                }
            }
        }
        if (a.y_pool != nullptr) {   // H, W even (host check): the tile IS one pooling window
            f32x4 p;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = fmaxf(fmaxf(o[0][0][e], o[0][1][e]), fmaxf(o[1][0][e], o[1][1][e]));
            *reinterpret_cast<f32x4*>(a.y_pool + ((size_t)(b * a.th + ty) * a.tw + tx) * a.ldp + a.poff + n) = p;
        }
    }
    if (STATS) {
        float* mine = red + (threadIdx.y * blockDim.x + threadIdx.x) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { mine[e] = s1[e]; mine[4 + e] = s2[e]; }
        __syncthreads();
        if (threadIdx.y == 0 && nok) {
            double* st = a.stats + (size_t)(blockIdx.x % Y2_STATS_REPL) * 2 * a.Cout;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t1 = 0.f, t2 = 0.f;
                for (int yy = 0; yy < (int)blockDim.y; ++yy) {
                    t1 += red[(yy * blockDim.x + threadIdx.x) * 8 + e];
                    t2 += red[(yy * blockDim.x + threadIdx.x) * 8 + 4 + e];
                }
                atomicAdd(st + n + e, (double)t1);
                atomicAdd(st + a.Cout + n + e, (double)t2);
            }
        }
    }
}

__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int ci = (int)(idx % Cin);
    const int co = (int)(idx / Cin);
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((size_t)co * 9 + t) * Cin + ci];
    float s[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s[0][j] = g[0][j];
        s[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        s[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        s[3][j] = g[2][j];
    }
    const size_t plane = (size_t)Cout * Cin;
    float* dst = u + (size_t)co * Cin + ci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[(4 * i + 0) * plane] = s[i][0];
        dst[(4 * i + 1) * plane] = 0.5f * (s[i][0] + s[i][1] + s[i][2]);
        dst[(4 * i + 2) * plane] = 0.5f * (s[i][0] - s[i][1] + s[i][2]);
        dst[(4 * i + 3) * plane] = s[i][2];
    }
}

// ---- stage 2 + 3 fused (Y2_ALGO_WINOGRAD_FUSED): one workgroup owns 64 tiles x 64 output channels for ALL 16 positions, so the
// output transform A^T M A is in-lane arithmetic on the MFMA accumulators and M never goes to memory.
//   * 4 waves (2 x 2), wave tile 32 tiles x 32 channels = one 32x32 MFMA block per position: 16 accumulators = 256 AGPRs, one
//     wave per SIMD (measured: the LDS-DMA pipeline keeps ~95 % of its 2-workgroup rate with one workgroup per CU).
//   * stage = (K-slab of 32 channels, position p): A = V[p][64 tiles][32] (8 KB) + B = U[p][64 couts][32] (8 KB), fetched by
//     LDS-DMA (same 128-B rows + XOR chunk swizzle as conv_fwd_dma_kernel) into a WF_STAGES-deep ring: WF_STAGES-1 stages in
//     flight, counted vmcnt + one raw s_barrier per stage; 8 ds_read_b128 + 16 MFMA 32x32x2 per wave per stage.
//   * epilogue: accumulator register r of the 16 positions belongs to the same (tile, channel): 24 adds give the 2x2 output
//     pixels, then affine + LeakyReLU, optional 2x2 max-pool (a tile is a pooling window) and BN statistics (valid pixels only).
// Packed fp32 add / subtract (2 lanes of a register pair per instruction).  The compiler selects v_pk_add_f32 for vector adds only
// when it feels like it and never for subtracts; inside the MFMA stream of the fused kernel every VALU instruction costs MFMA issue
// slots (measured: 64 scalar adds per stage = +15 % kernel time), so the in-loader input transform spells them out.  a + (-b) is
// the same IEEE operation as a - b: bit-identical to the scalar form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 y2_pk_add2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 y2_pk_sub2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 y2_pk_mul2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x4 y2_pk_add(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 y2_pk_sub(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

#ifdef Y2_STAMPS
constexpr size_t WF_DUMP_BYTES = 1024 + 8192;
#else
constexpr size_t WF_DUMP_BYTES = 1024;      // where the lanes of pixels / tiles / channels that do not exist store (branch-free epilogue)
#endif
constexpr int WF_POS_FLOATS = (64 + 64) * 32;      // one position's A + B slab (16 KB)
constexpr int WF2_DEFAULT_VARIANT = 3;             // which fused kernel y2_conv_fwd launches by default: -1 = first generation, else the feature mask of wino_fused2_kernel

struct WinoFusedArgs {
    const float* v;       // [16][T][Cin]
    const float* u;       // [16][Cout][Cin]
    const int32_t* tile_pix;   // [T] (wino_input_kernel): output pixel index of each tile + edge flags
    float* dump;               // >= 256 floats of scratch: where the lanes of pixels / tiles / channels that do not exist store (branch-free epilogue)
    const float* scale; const float* shift;
    float* y; float* y_pool; double* stats;
    int H, W, Cin, Cout, ldy, coff, ldp, poff, th, tw, T, tiles_m, tiles_n;
    unsigned v_bytes, u_bytes;
    float slope;
    y2_fastdiv d_tt, d_tw;
    const float* x;       // VAR bit 2 of wino_fused2_kernel: the chunk's NHWC input itself (v unused)
    int ldx;
    unsigned x_bytes;
    unsigned y_bytes, yp_bytes;      // wino_fused2_kernel stores through buffer descriptors (< 2^31 bytes each, host check)
    int32_t* sched;                  // wino_fused2_kernel: per-XCD counters of the next unclaimed tile (set by the kernel launched in front of it)
    int sched_static;                // != 0: static stride instead of claiming (Y2_WF_STATIC=1: the A/B of tools/contention.py)
};

// PG = positions per pipeline stage (one barrier per stage: 16*PG MFMAs per wave between barriers), WF_STAGES = ring depth.
template <int PG, int WF_STAGES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fused_kernel(const WinoFusedArgs a) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PER = 4 * PG;                              // DMA instructions per thread per stage (2 A + 2 B per position)
    constexpr int D = WF_STAGES - 1;                         // stages in flight ahead of the one being consumed
    constexpr int SPK = 16 / PG;                             // stages per K-slab
    constexpr int WF_STAGE_FLOATS = PG * WF_POS_FLOATS;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    // Persistent workgroups (one per CU): XCD x owns the contiguous chunk [x*per_xcd, (x+1)*per_xcd) of the tile list (tile_n
    // fastest, so the workgroups of one XCD share V rows / the filter panel in that L2) and its workgroups stride through it.
    const int ntiles = a.tiles_m * a.tiles_n;
    const int xcd = blockIdx.x % Y2_NUM_XCD, wg_in_xcd = blockIdx.x / Y2_NUM_XCD;
    const int wgs_per_xcd = gridDim.x / Y2_NUM_XCD;                    // host: gridDim.x is a multiple of 8
    const int per_xcd = (ntiles + Y2_NUM_XCD - 1) / Y2_NUM_XCD;
    const int xcd_end = min(ntiles, (xcd + 1) * per_xcd);
    int tile = xcd * per_xcd + wg_in_xcd;
    if (tile >= xcd_end) return;

    // staging: lane -> physical 16-B slot (lane & 7) of row (t >> 3) + 32*i, fetching logical chunk slot ^ swz(row)
    const int srow = t >> 3;
    const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);
    unsigned a_off[2], b_off[2];
    int m0 = 0, n0 = 0;
    auto place = [&](int tl) {          // operand row offsets of tile tl
        m0 = (tl / a.tiles_n) * 64;
        n0 = (tl % a.tiles_n) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + srow + 32 * i, n = n0 + srow + 32 * i;
            a_off[i] = m < a.T ? (unsigned)(((size_t)m * a.Cin + 4 * lchunk) * 4) : OOB;
            b_off[i] = n < a.Cout ? (unsigned)(((size_t)n * a.Cin + 4 * lchunk) * 4) : OOB;
        }
    };
    place(tile);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v), 0, a.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    const unsigned v_plane = (unsigned)((size_t)a.T * a.Cin * 4), u_plane = (unsigned)((size_t)a.Cout * a.Cin * 4);

    auto issue = [&](int kslab, int g, int slot) {
#pragma unroll
        for (int pp = 0; pp < PG; ++pp) {
            const int p = g * PG + pp;
            float* sa = smem + slot * WF_STAGE_FLOATS + pp * WF_POS_FLOATS + wave * (8 * 32);
            float* sb = sa + 64 * 32;
            const unsigned sv = (unsigned)p * v_plane + (unsigned)kslab * 128u;      // uniform part of the address
            const unsigned su = (unsigned)p * u_plane + (unsigned)kslab * 128u;
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(sa + i * 32 * 32), 16, (int)a_off[i], (int)sv, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(sb + i * 32 * 32), 16, (int)b_off[i], (int)su, 0, 0);
        }
    };

    f32x16 acc[16];
    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = l31 * 32 + (((2 * q + half) ^ sw) << 2);
    const int fa = wm * 32 * 32, fb = 64 * 32 + wn * 32 * 32;

    // one stage: PG independent accumulator chains, interleaved so consecutive MFMAs never depend on each other
    auto compute = [&](int slot, f32x16* c) {
        const float* sbuf = smem + slot * WF_STAGE_FLOATS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[PG], bv[PG];
#pragma unroll
            for (int pp = 0; pp < PG; ++pp) {
                av[pp] = *reinterpret_cast<const f32x4*>(sbuf + pp * WF_POS_FLOATS + fa + foff[q]);
                bv[pp] = *reinterpret_cast<const f32x4*>(sbuf + pp * WF_POS_FLOATS + fb + foff[q]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int pp = 0; pp < PG; ++pp) c[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pp][e], bv[pp][e], c[pp], 0, 0, 0);
        }
    };

    const int nks = a.Cin / 32;
    const int NS = nks * SPK;                                // stages, position group fastest
    int ik = 0, ip = 0, islot = 0;                           // next stage to issue
    auto issue_next = [&]() {
        issue(ik, ip, islot);
        islot = islot == WF_STAGES - 1 ? 0 : islot + 1;
        if (++ip == SPK) { ip = 0; ++ik; }
    };
    for (int s = 0; s < (NS < D ? NS : D); ++s) issue_next();      // prologue of the first tile
    int cur = 0;
    for (;;) {
    const int em0 = m0, en0 = n0;                                  // this tile's origin (place() moves m0 / n0 to the next tile)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    // steady K-slabs: every stage has D-1 younger stages in flight behind it
    static_assert(D - 1 <= SPK - 1 && D <= 5 && PER * (D > 1 ? D - 1 : 1) < 64, "ring deeper than a K-slab / vmcnt range");
    static_assert(D == 1, "persistent loop: with D > 1 the epilogue stores sit between DMA stages in the vmcnt order - re-derive the counts first");
    for (int ks = 0; ks < nks - 1; ++ks) {
#pragma unroll
        for (int g = 0; g < SPK; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (D - 1)) : "memory");
            __builtin_amdgcn_s_barrier();            // stage landed for everyone; everyone finished reading the previous stage
            issue_next();                            // into the slot of the previous stage (a stage D ahead always exists here)
            compute(cur, &acc[g * PG]);
            cur = cur == WF_STAGES - 1 ? 0 : cur + 1;
        }
    }
    // last K-slab: the number of younger stages shrinks to 0 -> exact counts
#pragma unroll
    for (int g = 0; g < SPK; ++g) {
        const int younger = (SPK - 1 - g) < (D - 1) ? (SPK - 1 - g) : (D - 1);
        if (younger == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D > 4 ? PER * 4 : 0) : "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D > 3 ? PER * 3 : 0) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D > 2 ? PER * 2 : 0) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D > 1 ? PER * 1 : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g + D < SPK) issue_next();
        compute(cur, &acc[g * PG]);
        cur = cur == WF_STAGES - 1 ? 0 : cur + 1;
    }
    // the next tile's first D stages go out BEFORE the epilogue: their DMA latency hides behind the output transform.  They
    // land in the slots of stages NS-2 ... NS-D-1 (never the one just consumed), which every wave left before the last barrier.
    // (The epilogue's stores also count in vmcnt; the ring in use is one stage deep, so the next wait is vmcnt(0) anyway.)
    tile += wgs_per_xcd;
    const bool more = tile < xcd_end;
    if (more) {
        place(tile);
        ik = 0; ip = 0;
        for (int s = 0; s < (NS < D ? NS : D); ++s) issue_next();
    }

    // ---- epilogue: output transform in registers.  C layout: lane -> channel n (l31), register r -> tile row (r&3) + 8*(r>>2) + 4*half
    const int n = en0 + wn * 32 + l31;
    const bool nok = n < a.Cout;
    const float sc = (a.scale != nullptr && nok) ? a.scale[n] : 1.f;
    const float sh = (a.shift != nullptr && nok) ? a.shift[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tt = em0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float sm[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sm[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
            sm[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
        float o[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            o[i][0] = sm[i][0] + sm[i][1] + sm[i][2];
            o[i][1] = sm[i][1] - sm[i][2] - sm[i][3];
        }
        if (tt >= a.T || !nok) continue;
        const int b = (int)y2_div((uint32_t)tt, a.d_tt);
        const int rr = tt - b * a.th * a.tw;
        const int ty = (int)y2_div((uint32_t)rr, a.d_tw);
        const int tx = rr - ty * a.tw;
        const bool y1 = 2 * ty + 1 < a.H, x1 = 2 * tx + 1 < a.W;
        if (a.stats != nullptr) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if ((i == 1 && !y1) || (j == 1 && !x1)) continue;
                    s1 += o[i][j];
                    s2 += o[i][j] * o[i][j];
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float uu = o[i][j] * sc + sh;
                o[i][j] = uu > 0.f ? uu : uu * a.slope;
            }
        if (a.y != nullptr) {
            float* dst = a.y + ((size_t)(b * a.H + 2 * ty) * a.W + 2 * tx) * a.ldy + a.coff + n;
            dst[0] = o[0][0];
            if (x1) dst[a.ldy] = o[0][1];
            if (y1) {
                dst[(size_t)a.W * a.ldy] = o[1][0];
                if (x1) dst[(size_t)(a.W + 1) * a.ldy] = o[1][1];
            }
        }
        if (a.y_pool != nullptr)
            a.y_pool[((size_t)(b * a.th + ty) * a.tw + tx) * a.ldp + a.poff + n] = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1]));
    }
    if (a.stats != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (half == 0 && nok) {
            double* st = a.stats + (size_t)((em0 >> 6) % Y2_STATS_REPL) * 2 * a.Cout;
            atomicAdd(st + n, (double)s1);
            atomicAdd(st + a.Cout + n, (double)s2);
        }
    }
    if (!more) break;
    }   // persistent tile loop
}

// ---- fused kernel, second generation.  Same tile (64 tiles x 64 channels x 16 positions per workgroup, 4 waves, 256 accumulator
// registers, one wave per SIMD, 2 x 64 KB LDS ring) and the same arithmetic (bit-identical results) as wino_fused_kernel; what
// changes is the instruction stream of the single wave a SIMD runs, because with one wave nothing else hides a stall:
//   * DMA spread (VAR bit 0): the 16 LDS-DMA instructions that fetch stage s+1 are issued two at a time behind the first 8 groups
//     of four MFMAs of stage s (an LDS-DMA issue costs ~60 cycles, a 32x32x2 fp32 MFMA occupies the pipe for 64) instead of all 16
//     in front of the stage's first MFMAs (~0.7k of 4.1k cycles per stage with an idle matrix pipe); the second half of the stage is
//     left for them to land before the next vmcnt(0) (one piece per slot over the whole stage measured 4-8 % slower).  The fetch
//     stream runs ACROSS tiles: the last stage of a tile fetches the first stage of the workgroup's next tile.
//   * fragment double buffering (VAR bit 1): the 8 ds_read_b128 of k-group q+1 are issued before the 16 MFMAs of group q.
//   * branch-free epilogue with compile-time output set (OUT): stores go through buffer descriptors and the lanes of pixels /
//     tiles / channels that do not exist get the out-of-range offset (dropped by the hardware) instead of being predicated off,
//     the tile row -> pixel decode comes from a table (one entry per tile), y / pooled output / statistics are template flags.
//     No control flow.  (Spelling the transform out as packed adds on two tile rows at a time made the epilogue LONGER: the asm
//     operands cost more register moves than the packing saves.)
//   * the accumulators of a tile start from the zero C operand of its first MFMAs instead of 256 v_accvgpr_write.
//   * tiles beyond a workgroup's first are claimed from a per-XCD counter (see the tile loop).
//   * DMA plane offsets are formed at the instruction (one s_mul) - hoisted, their 32 SGPRs spilled to VGPR lanes.
//   Every MFMA "slot" ends with a scheduling barrier so that the compiler keeps the hand-placed DMA interleave.
// Measured B=32 (kernel alone, first generation -> this): 104x104 Cin 64 0.319 -> 0.249 ms, 52x52 Cin 128 0.270 -> 0.230,
// 26x26 Cin 512 0.443 -> 0.397, 13x13 Cin 512 0.305 -> 0.271.
// Negative results kept out of the code: (a) deferring the output rows into the next tile's first stages (interleaved row by row,
// or phase by phase behind single MFMAs) was 8-25 % SLOWER than this stand-alone epilogue once it was branch-free - the extra 64
// live VGPRs spill and the VALU/stores delay MFMA issue; (b) exchanging the MFMA operand roles (accumulator rows = channels, so
// a lane stores 16 bytes) lost 10-30 %: 64 lanes x 16 B to 64 different cache lines per store instruction; (c) skipping the
// filter-operand DMA altogether (a wrong-result experiment) gains only 2-6 %: the K loop is not LDS-DMA-bandwidth bound.
//
// Implicit input transform (VAR bit 2, Y2_ALGO_WINOGRAD_IMPLICIT): the input operand of a stage is not DMA'd from a transformed
// tensor V but built by the loader: a thread owns the same two (tile row, 4-channel chunk) items the DMA lane owned, loads their
// 4x4 input pixels with buffer_load_dwordx4 (zero padding = buffer out-of-range), forms row g of B^T d B for the stage's 4
// positions with 16 packed adds per item and ds_writes them where the DMA would have put V.  wino_input_kernel and the 4x-input
// tensor V (write + read) disappear; the kernel itself runs 0-9 % slower than with DMA'd V (B=32: 0.258 vs 0.249 ms on
// 104x104x64->128, 0.254 vs 0.256 on 104x104x128->64, 0.251 vs 0.230 on 52x52x128->256), the layer 5-40 % faster.  Schedule
// of a stage: pixel loads behind the first 8 MFMAs, filter DMA behind MFMAs 8..39, transform + store behind the last 16.
// What was measured on the way (B=32, kernel alone): loading rows per stage instead of keeping rows 1 / 2 in registers across
// the K slab: +3 %; scalar instead of packed adds: +-0; fragment double buffering (VAR bit 1) on top: +3-7 % (its 32 registers
// spill); pixel loads a whole stage ahead (fourth row buffer, 128 registers of pixels): +8-25 % - the tile transition spills
// ~100 registers; dword "touch" loads one stage ahead as a cache prefetch: +1-7 %; pixels by LDS-DMA into a scratch area and read
// back: -1-3 % (not worth 32 KB of LDS); every row loaded as early as its register frees: +4-24 % (tile-transition spills).
// What DID help were registers: pixel offsets formed at the load instead of 32 hoisted sums, the decode-table loads from one
// base address, the DMA plane offsets as s_mul - each 3-8 %.  Per-stage stamps (-DY2_STAMPS): a stage costs 4.45k cycles with
// DMA'd V, 4.55k with the transform's adds and stores but no pixel loads, 5.0k with 8 pixel loads.
// VAR bit 3: Cin == 32, one K slab per tile (the 208x208 layer).
template <int VAR, int OUT>     // OUT: bit 0 = full-resolution output y, bit 1 = pooled output, bit 2 = BatchNorm statistics
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fused2_kernel(const WinoFusedArgs a) {
    constexpr bool HAS_Y = (OUT & 1) != 0, HAS_POOL = (OUT & 2) != 0, HAS_STATS = (OUT & 4) != 0;
    constexpr bool SPREAD = (VAR & 1) != 0, DBUF = (VAR & 2) != 0, RAWIN = (VAR & 4) != 0;
    constexpr bool ONE = (VAR & 8) != 0;         // Cin == 32: a tile is ONE K slab (4 stages); its last-slab stages start the accumulators
    constexpr unsigned OOB = 0x80000000u, OOB_COL = 0x40000000u;
    constexpr int PG = 4;
    constexpr int STAGE_FLOATS = PG * WF_POS_FLOATS;          // 64 KB
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int ntiles = a.tiles_m * a.tiles_n;
    const int xcd = blockIdx.x % Y2_NUM_XCD, wg_in_xcd = blockIdx.x / Y2_NUM_XCD;
    const int wgs_per_xcd = gridDim.x / Y2_NUM_XCD;
    const int per_xcd = (ntiles + Y2_NUM_XCD - 1) / Y2_NUM_XCD;
    const int xcd_end = min(ntiles, (xcd + 1) * per_xcd);
    int tile = xcd * per_xcd + wg_in_xcd;
    if (tile >= xcd_end) return;

    const int srow = t >> 3;
    const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);
    unsigned a_off[2], b_off[2];                 // operand row offsets of the tile whose stages are being FETCHED
    unsigned rowoff[2][4], coloff[2][4];         // RAWIN: byte offsets of the 4 rows / 4 columns of the 4x4 input patch of the thread's two tiles
    int fm0 = 0, fn0 = 0;                        // ... and its origin
    auto place = [&](int tl) {
        fm0 = (tl / a.tiles_n) * 64;
        fn0 = (tl % a.tiles_n) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = fm0 + srow + 32 * i, n = fn0 + srow + 32 * i;
            b_off[i] = n < a.Cout ? (unsigned)(((size_t)n * a.Cin + 4 * lchunk) * 4) : OOB;
            if (!RAWIN) {
                a_off[i] = m < a.T ? (unsigned)(((size_t)m * a.Cin + 4 * lchunk) * 4) : OOB;
            } else {
                // pixel (y, x) of the patch sits at rowoff[y] + coloff[x]; a row or column outside the image gets a sentinel that
                // pushes the sum past the buffer's num_records (x_bytes < 2^30, host check), where a buffer load returns zeros:
                // the zero padding of the convolution and the tiles past T cost no branch
                const bool mok = m < a.T;
                const uint32_t mm = mok ? (uint32_t)m : 0u;
                const int b = (int)y2_div(mm, a.d_tt);
                const int rr = (int)mm - b * a.th * a.tw;
                const int ty = (int)y2_div((uint32_t)rr, a.d_tw);
                const int tx = rr - ty * a.tw;
                const unsigned pix_bytes = (unsigned)a.ldx * 4u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int yy = 2 * ty - 1 + r, xx = 2 * tx - 1 + r;
                    rowoff[i][r] = (mok && (unsigned)yy < (unsigned)a.H) ? (unsigned)((b * a.H + yy) * a.W) * pix_bytes : OOB;
                    coloff[i][r] = (unsigned)xx < (unsigned)a.W ? (unsigned)xx * pix_bytes + 16u * (unsigned)lchunk : OOB_COL;
                }
            }
        }
    };
    place(tile);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RAWIN ? a.x : a.v), 0, RAWIN ? a.x_bytes : a.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v), 0, a.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    const unsigned v_plane = (unsigned)((size_t)a.T * a.Cin * 4), u_plane = (unsigned)((size_t)a.Cout * a.Cin * 4);

    // one of the 16 DMA instructions of stage (kslab, g) into ring slot `slot`: j = 4*pp + w, w = 0,1: A rows, 2,3: B rows
    auto dma_piece = [&](int kslab, int g, int slot, int j) {
        const int pp = j >> 2, w = j & 3;
        int p = g * PG + pp;
        asm volatile("" : "+s"(p));          // opaque: p * plane is formed here (one s_mul), not hoisted into 32 SGPRs that spill to VGPR lanes
        float* sa = smem + slot * STAGE_FLOATS + pp * WF_POS_FLOATS + wave * (8 * 32);
        if (w < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(sa + w * 32 * 32), 16, (int)a_off[w], (int)((unsigned)p * v_plane + (unsigned)kslab * 128u), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(sa + 64 * 32 + (w - 2) * 32 * 32), 16, (int)b_off[w - 2], (int)((unsigned)p * u_plane + (unsigned)kslab * 128u), 0, 0);
    };

    f32x16 acc[16];
    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = l31 * 32 + (((2 * q + half) ^ sw) << 2);
    const int fa = wm * 32 * 32, fb = 64 * 32 + wn * 32 * 32;

    // RAWIN: one pixel chunk (4 channels) of input row `row` / column `col` of the 4x4 patch of the thread's tile i, K slab `kslab`
    auto raw_px = [&](int i, int row, int col, int kslab) -> f32x4 {
        // the sum is formed here, at the load (volatile: left to itself the compiler hoists all 32 row + column sums of a tile into
        // registers of their own, which is what tips the kernel into spilling inside the K loop)
        unsigned voff;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(voff) : "v"(rowoff[i][row]), "v"(coloff[i][col]));
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff, kslab * 128, 0));
    };
    // RAWIN: positions 4*g .. 4*g+3 are row g of B^T d B: s = d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3 per column (g = 0..3), then the
    // same column combination as wino_input_kernel (same operations in the same order: bit-identical V).  Patch rows 1 and 2 feed
    // three position rows each: they stay in registers across the stages of a K slab (pr1, pr2; pr0 holds row 0, later row 3), so a
    // K slab loads every input pixel of the tile once - 32 loads per thread, the same bytes the DMA of V moved.
    f32x4 pr0[2][4], pr1[2][4], pr2[2][4];
    auto col_combine = [&](const f32x4* sv, int j) -> f32x4 {
        return j == 0 ? y2_pk_sub(sv[0], sv[2]) : (j == 1 ? y2_pk_add(sv[1], sv[2]) : (j == 2 ? y2_pk_sub(sv[2], sv[1]) : y2_pk_sub(sv[1], sv[3])));
    };

    // ---- one pipeline stage: consume ring slot SLOT into the 4 accumulators c[0..3];  FETCH: also fetch stage (fk, FG) into the
    //      other slot: 16 DMA pieces, or (RAWIN) 8 filter DMA pieces + the thread's 2 x 8 input pixels, transformed in registers
    auto stage = [&](auto SLOT, f32x16* c, auto FETCH, int fk, auto FG, auto ZC) {      // ZC: the accumulators START here (first K slab of a tile)
        constexpr int slot = decltype(SLOT)::value;
        constexpr bool fetch = decltype(FETCH)::value;
        constexpr int fg = decltype(FG)::value;
        constexpr bool zc = decltype(ZC)::value;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* sbuf = smem + slot * STAGE_FLOATS;
        float* const wbuf = smem + (slot ^ 1) * STAGE_FLOATS + t * 4;      // RAWIN: where this thread's V chunks go (position 0, tile 0)
        f32x4 av[2][PG], bv[2][PG];
        f32x4 sv[4];
        auto reads = [&](int q, int buf) {
#pragma unroll
            for (int pp = 0; pp < PG; ++pp) {
                av[buf][pp] = *reinterpret_cast<const f32x4*>(sbuf + pp * WF_POS_FLOATS + fa + foff[q]);
                bv[buf][pp] = *reinterpret_cast<const f32x4*>(sbuf + pp * WF_POS_FLOATS + fb + foff[q]);
            }
        };
        if (fetch && !SPREAD && !RAWIN) {
#pragma unroll
            for (int j = 0; j < 16; ++j) dma_piece(fk, fg, slot ^ 1, j);
        }
        reads(0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = DBUF ? (q & 1) : 0;
            if (DBUF && q < 3) reads(q + 1, cb ^ 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sl = q * 4 + e;                 // MFMA slot 0..15 of the stage
#pragma unroll
                for (int pp = 0; pp < PG; ++pp) {
                    // the first MFMA of a tile's chain takes the inline constant 0 as its C operand: no 256 v_accvgpr_write per tile
                    c[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][pp][e], bv[cb][pp][e], (zc && sl == 0) ? zero16 : c[pp], 0, 0, 0);
                    if (!RAWIN) {
                        // two DMA pieces per slot in the first 8 slots, behind MFMAs 0 and 2 of the slot
                        if (SPREAD && fetch && sl < 8 && (pp == 0 || pp == 2)) dma_piece(fk, fg, slot ^ 1, 2 * sl + (pp >> 1));
                    } else if (fetch) {
                        if (sl < 2) {                     // slots 0, 1: the pixel loads of tile 0 / tile 1 (rows 0 + 2 | 1 | none | 3)
                            if (fg == 0) { pr0[sl][pp] = raw_px(sl, 0, pp, fk); pr2[sl][pp] = raw_px(sl, 2, pp, fk); }
                            else if (fg == 1) pr1[sl][pp] = raw_px(sl, 1, pp, fk);
                            else if (fg == 3) pr0[sl][pp] = raw_px(sl, 3, pp, fk);
                        } else if (sl < 10) {             // slots 2..9: the 8 filter DMA pieces
                            if (pp == 0) dma_piece(fk, fg, slot ^ 1, 4 * ((sl - 2) >> 1) + 2 + ((sl - 2) & 1));
                        } else if (sl >= 12) {            // slots 12, 13: tile 0, slots 14, 15: tile 1 (>= 12 slots = 3k cycles after the loads)
                            const int i = (sl - 12) >> 1;
                            if (((sl - 12) & 1) == 0)
                                sv[pp] = fg == 0 ? y2_pk_sub(pr0[i][pp], pr2[i][pp]) : (fg == 1 ? y2_pk_add(pr1[i][pp], pr2[i][pp]) : (fg == 2 ? y2_pk_sub(pr2[i][pp], pr1[i][pp]) : y2_pk_sub(pr1[i][pp], pr0[i][pp])));
                            else *reinterpret_cast<f32x4*>(wbuf + pp * WF_POS_FLOATS + i * 32 * 32) = col_combine(sv, pp);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!DBUF && q < 3) reads(q + 1, 0);
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using G0 = std::integral_constant<int, 0>;
    using G1 = std::integral_constant<int, 1>;
    using G2 = std::integral_constant<int, 2>;
    using G3 = std::integral_constant<int, 3>;
    using Z0 = std::false_type;
#if defined(Y2_EXP) && (Y2_EXP & 1)
    using Z1 = std::integral_constant<bool, !RAWIN>;
#else
    using Z1 = std::true_type;
#endif

    // output stores go through buffer descriptors: a lane whose pixel / tile / channel does not exist gets the out-of-range offset
    // and the hardware drops its store (no branch, no 64-bit address arithmetic); the 2x2 pixels of a tile differ in the scalar offset
    volatile int* const sched_lds = reinterpret_cast<volatile int*>(smem + 2 * STAGE_FLOATS + (WF_DUMP_BYTES - 1024) / sizeof(float));
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, HAS_Y ? a.y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryp = __builtin_amdgcn_make_buffer_rsrc(a.y_pool, 0, HAS_POOL ? a.yp_bytes : 0, 0x00020000);
    const int so_x = a.ldy * 4, so_y = a.W * a.ldy * 4;
    const int nks = ONE ? 1 : a.Cin / 32;        // (host: nks >= 2 unless ONE) every tile runs 4 * nks stages, an even number, so a
    // prologue: the first stage of the first tile     // tile always starts in ring slot 0 and stage g of a K slab sits in slot g & 1
    if (!RAWIN) {
#pragma unroll
        for (int j = 0; j < 16; ++j) dma_piece(0, 0, 0, j);
    } else {
#pragma unroll
        for (int pp = 0; pp < PG; ++pp) {
            dma_piece(0, 0, 0, 4 * pp + 2);
            dma_piece(0, 0, 0, 4 * pp + 3);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 sv[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                pr2[i][cc] = raw_px(i, 2, cc, 0);
                sv[cc] = y2_pk_sub(raw_px(i, 0, cc, 0), pr2[i][cc]);
            }
#pragma unroll
            for (int pp = 0; pp < PG; ++pp) *reinterpret_cast<f32x4*>(smem + t * 4 + pp * WF_POS_FLOATS + i * 32 * 32) = col_combine(sv, pp);
        }
    }
    // -DY2_STAMPS (tools/wf_bench.py --stamps): wave 0 of workgroup 0 records s_memtime behind every stage barrier and around the
    // epilogue (in LDS - a global store would sit in the vmcnt queue of the loads being timed - copied out behind the dump area at
    // the end; the host side reserves the room in the same build).  Caveat: the extra state changes the register allocation of the
    // tile transition; the steady-state stages are what these stamps are good for.
#ifdef Y2_STAMPS
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(smem + 2 * STAGE_FLOATS);
    int nstamp = 0;
#define Y2_STAMP() do { if (blockIdx.x == 0 && wave == 0 && nstamp < 1000) { const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[nstamp] = tm_; ++nstamp; } } while (0)
#else
#define Y2_STAMP() do {} while (0)
#endif
    for (;;) {
        const int em0 = fm0, en0 = fn0;           // this tile's origin (the fetch cursor is still on this tile)
        if (!Z1::value) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
        }
#define Y2_WF2_SYNC()                                              \
        if (RAWIN) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   /* + this wave's ds_write of V */ \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                              \
        Y2_STAMP()
        // first K slab (never the last one unless ONE): the four accumulator groups start from the MFMA's zero operand
        using ZL = std::integral_constant<bool, ONE>;
        if (!ONE) {
        Y2_WF2_SYNC();
        stage(S0{}, &acc[0], T_{}, 0, G1{}, Z1{});
        Y2_WF2_SYNC();
        stage(S1{}, &acc[4], T_{}, 0, G2{}, Z1{});
        Y2_WF2_SYNC();
        stage(S0{}, &acc[8], T_{}, 0, G3{}, Z1{});
        Y2_WF2_SYNC();
        stage(S1{}, &acc[12], T_{}, 1, G0{}, Z1{});
        }
        for (int ks = 1; ks < nks - 1; ++ks) {
            Y2_WF2_SYNC();
            stage(S0{}, &acc[0], T_{}, ks, G1{}, Z0{});
            Y2_WF2_SYNC();
            stage(S1{}, &acc[4], T_{}, ks, G2{}, Z0{});
            Y2_WF2_SYNC();
            stage(S0{}, &acc[8], T_{}, ks, G3{}, Z0{});
            Y2_WF2_SYNC();
            stage(S1{}, &acc[12], T_{}, ks + 1, G0{}, Z0{});
        }
        // ---- last K slab; its last stage fetches the first stage of the workgroup's next tile.  Which tile that is, is claimed
        // from the XCD's counter (one atomic by thread 0, handed to the other waves through LDS across the stage barriers): a
        // workgroup's FIRST tile is its static one, the rest go to whoever is running.  With a static stride a workgroup that
        // could not start with the others (RCCL's all-reduce kernels hold CUs while the data-parallel backward runs, and this
        // kernel needs a whole CU) kept its full share of tiles for a second pass; now it costs one tile.
        int claimed = 0;
        if (t == 0) claimed = a.sched_static ? (tile - xcd * per_xcd + wgs_per_xcd) : atomicAdd(a.sched + xcd, 1);      // (static stride: A/B runs, tools/contention.py)
        Y2_WF2_SYNC();
        stage(S0{}, &acc[0], T_{}, nks - 1, G1{}, ZL{});
        Y2_WF2_SYNC();
        if (t == 0) *sched_lds = claimed;
        stage(S1{}, &acc[4], T_{}, nks - 1, G2{}, ZL{});
        Y2_WF2_SYNC();
        stage(S0{}, &acc[8], T_{}, nks - 1, G3{}, ZL{});
        tile = xcd * per_xcd + __builtin_amdgcn_readfirstlane(*sched_lds);
        const bool more = tile < xcd_end;
        Y2_WF2_SYNC();
        // decode-table entries of this tile's 16 rows (two distinct addresses per wave: broadcast loads), in flight during the last
        // stage (issued in front of the barrier above, their latency - ~2k cycles per tile - was exposed)
        // ... and the channel's affine parameters: loaded here, right behind the barrier's vmcnt(0) - at the top of the epilogue the
        // compiler guarded them with a vmcnt(0) of its own, which waited for the next tile's first fetch (~1.7k cycles per tile)
        const int pn = en0 + wn * 32 + l31;
        const bool nok = pn < a.Cout;
        const float psc = (a.scale != nullptr && nok) ? a.scale[pn] : 1.f;
        const float psh = (a.shift != nullptr && nok) ? a.shift[pn] : 0.f;
        int prow[16];                              // (the table is padded to whole 64-tile blocks: one base address, immediate offsets)
        const int32_t* const pbase = a.tile_pix + (em0 + wm * 32 + 4 * half);
#pragma unroll
        for (int r = 0; r < 16; ++r) prow[r] = pbase[(r & 3) + 8 * (r >> 2)];
        if (more) {
            place(tile);
            stage(S1{}, &acc[12], T_{}, 0, G0{}, ZL{});
        } else {
            stage(S1{}, &acc[12], F_{}, 0, G0{}, ZL{});
        }
#undef Y2_WF2_SYNC
        Y2_STAMP();
        // ---- epilogue: output transform A^T M A in registers (accumulator register r of the 16 positions belongs to the same
        //      (tile row, channel)), then affine + LeakyReLU, pooling, statistics, stores - branch-free
        float s1 = 0.f, s2 = 0.f;
        const unsigned chan_off = (unsigned)(a.coff + pn) * 4u, pool_off = (unsigned)(a.poff + pn) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sm[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sm[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
                sm[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
            }
            float o[4];                                       // o[2*i + j]: raw output pixel (i, j) of the tile
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                o[2 * i + 0] = sm[i][0] + sm[i][1] + sm[i][2];
                o[2 * i + 1] = sm[i][1] - sm[i][2] - sm[i][3];
            }
            const int e = prow[r];
            const bool ok = nok && em0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < a.T;
            const bool y1 = ok && (e & 0x40000000) != 0, x1 = ok && e < 0;
            const unsigned voff = ((unsigned)e & 0x3fffffffu) * (unsigned)so_x + chan_off;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool valid = k == 0 ? ok : (k == 1 ? x1 : (k == 2 ? y1 : (y1 && x1)));
                if (HAS_STATS) { const float m = valid ? o[k] : 0.f; s1 += m; s2 += m * m; }
                const float uu = o[k] * psc + psh;
                v[k] = uu > 0.f ? uu : uu * a.slope;
                if (HAS_Y) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), ry, (int)(valid ? voff : OOB), ((k & 1) ? so_x : 0) + ((k >> 1) ? so_y : 0), 0);
            }
            if (HAS_POOL) {                                   // H, W even (host check): the pooled pixel index IS the tile index
                const int tt = em0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx), ryp, (int)(ok ? (unsigned)tt * (unsigned)(a.ldp * 4) + pool_off : OOB), 0, 0);
            }
        }
        if (HAS_STATS) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (half == 0 && nok) {
                double* st = a.stats + (size_t)((em0 >> 6) % Y2_STATS_REPL) * 2 * a.Cout;
                atomicAdd(st + pn, (double)s1);
                atomicAdd(st + a.Cout + pn, (double)s2);
            }
        }
        Y2_STAMP();
        if (!more) break;
    }
#ifdef Y2_STAMPS
    if (blockIdx.x == 0 && wave == 0)
        for (int i = lane; i < 1000; i += 64) reinterpret_cast<unsigned long long*>(a.dump + 256)[i] = i < nstamp ? stamps[i] : 0ull;
#endif
#undef Y2_STAMP
}

// ---- fused kernel, third generation: TWO workgroups per CU.  wino_fused2_kernel runs one wave per SIMD (256 accumulator
// registers), so nothing covers its stage barriers, its epilogue (MFMA pipe idle for ~6k cycles per tile) or the register-path
// loader of the implicit variant.  Here a workgroup owns 32 tiles x 64 channels x 16 positions: 4 waves (2 x 2), wave tile 16 tiles
// x 32 channels as 2 blocks of v_mfma_f32_16x16x4_f32 (same 64 FLOP / cycle / SIMD as 32x32x2) = 16 positions x 2 blocks x 4 =
// 128 accumulator registers, two waves per SIMD, and the second workgroup's MFMAs run under the first one's barrier waits,
// fragment reads, claims and epilogue.  The output transform stays in-lane (a lane holds all 16 positions of its 4 tile rows x 2
// channels).  Stage = 2 positions x 32 input channels: A = V[p][32 tiles][32] (4 KB) + B = U[p][64 channels][32] (8 KB) per position,
// 24 KB per stage, 3-deep ring (72 KB per workgroup): the pieces of stage i+2 are issued at the top of stage i, so two stages of
// LDS-DMA are in flight and a stage waits with vmcnt(6), not vmcnt(0).  Same 128-B rows and XOR chunk swizzle as the other
// kernels: a ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) sees 16 distinct 16-B slots.
// The K sum runs in another order than in the 32x32x2 kernels (4 channels per MFMA instead of 2): results agree with the other
// Winograd algorithms to fp32 rounding, not bit for bit; the DMA'd-V and the implicit variant of THIS kernel are bit-identical.
constexpr int F3_TM = 32;
constexpr int F3_POS_FLOATS = (F3_TM + 64) * 32;      // one position's A + B slab (12 KB)
constexpr int F3_STAGE_FLOATS = 2 * F3_POS_FLOATS;    // two positions per stage
constexpr int F3_NST = 3;

template <int VAR, int OUT>     // VAR bit 2: implicit input transform; OUT as in wino_fused2_kernel
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_fused3_kernel(const WinoFusedArgs a) {
    constexpr bool HAS_Y = (OUT & 1) != 0, HAS_POOL = (OUT & 2) != 0, HAS_STATS = (OUT & 4) != 0;
    constexpr bool RAWIN = (VAR & 4) != 0;
    constexpr bool ONE = (VAR & 8) != 0;         // Cin == 32: a tile is ONE K slab, at once its first, its next-to-last and its last
    constexpr bool HALF = (VAR & 16) != 0;       // Cout <= 32 (the data gradient of the 208x208 layer): channels 32..63 of every unit do not exist - their B rows
                                                 // are not fetched and the waves that would multiply zeros (wn = 1) only load, transform and keep the barriers:
                                                 // their SIMDs' matrix pipes belong to the other workgroup's working waves
    constexpr unsigned OOB = 0x80000000u, OOB_COL = 0x40000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const int ntiles = a.tiles_m * a.tiles_n;
    const int xcd = blockIdx.x % Y2_NUM_XCD, wg_in_xcd = blockIdx.x / Y2_NUM_XCD;
    const int wgs_per_xcd = gridDim.x / Y2_NUM_XCD;
    const int per_xcd = (ntiles + Y2_NUM_XCD - 1) / Y2_NUM_XCD;
    const int xcd_end = min(ntiles, (xcd + 1) * per_xcd);
    int tile = xcd * per_xcd + wg_in_xcd;
    if (tile >= xcd_end) return;

    // operand ownership: thread t moves chunk (lane & 7) of row t >> 3 of the A slab (LDS-DMA, or - implicit - built from the 4x4
    // input patch of that tile) and of rows t >> 3, 32 + (t >> 3) of the B slab
    const int rowA = t >> 3;
    const int lchunk = (lane & 7) ^ ((rowA >> 1) & 7);
    unsigned a_off = OOB, b_off[2] = {OOB, OOB};
    unsigned rowoff[4], coloff[4];               // RAWIN: byte offsets of the rows / columns of the thread's patch (see wino_fused2_kernel)
    int fm0 = 0, fn0 = 0;
    auto place_a = [&](int tl, bool ok) {        // the input side of tile tl (ok == false: nothing left - every load returns zeros)
        fm0 = (tl / a.tiles_n) * F3_TM;
        const int m = fm0 + rowA;
        const bool mok = ok && m < a.T;
        if (!RAWIN) {
            a_off = mok ? (unsigned)(((size_t)m * a.Cin + 4 * lchunk) * 4) : OOB;
        } else {
            const uint32_t mm = mok ? (uint32_t)m : 0u;
            const int b = (int)y2_div(mm, a.d_tt);
            const int rr = (int)mm - b * a.th * a.tw;
            const int ty = (int)y2_div((uint32_t)rr, a.d_tw);
            const int tx = rr - ty * a.tw;
            const unsigned pix_bytes = (unsigned)a.ldx * 4u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int yy = 2 * ty - 1 + r, xx = 2 * tx - 1 + r;
                rowoff[r] = (mok && (unsigned)yy < (unsigned)a.H) ? (unsigned)((b * a.H + yy) * a.W) * pix_bytes : OOB;
                coloff[r] = (unsigned)xx < (unsigned)a.W ? (unsigned)xx * pix_bytes + 16u * (unsigned)lchunk : OOB_COL;
            }
        }
    };
    auto place_b = [&](int tl) {
        fn0 = (tl % a.tiles_n) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = fn0 + rowA + 32 * i;
            b_off[i] = n < a.Cout ? (unsigned)(((size_t)n * a.Cin + 4 * lchunk) * 4) : OOB;
        }
    };
    place_a(tile, true);
    place_b(tile);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RAWIN ? a.x : a.v), 0, RAWIN ? a.x_bytes : a.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v), 0, a.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u_bytes, 0x00020000);
    const unsigned v_plane = (unsigned)((size_t)a.T * a.Cin * 4), u_plane = (unsigned)((size_t)a.Cout * a.Cin * 4);

    // LDS-DMA piece j of stage (kslab, g) into ring slot `slot`: j = 3 * pp + w, w = 0: the A rows (not RAWIN), 1, 2: the B rows
    auto piece = [&](int kslab, int g, int slot, int j) {
        const int pp = j / 3, w = j % 3;
        int p = 2 * g + pp;
        asm volatile("" : "+s"(p));
        float* sa = smem + slot * F3_STAGE_FLOATS + pp * F3_POS_FLOATS + wave * (8 * 32);
        if (w == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)sa, 16, (int)a_off, (int)((unsigned)p * v_plane + (unsigned)kslab * 128u), 0, 0);
        else if (w == 1 || !HALF) __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(sa + w * 32 * 32), 16, (int)b_off[w - 1], (int)((unsigned)p * u_plane + (unsigned)kslab * 128u), 0, 0);
    };
    auto fetch = [&](int kslab, int g, int slot) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (!RAWIN || (j % 3) != 0) piece(kslab, g, slot, j);
    };
    // RAWIN.  V[4g + j] = column combination j of s_g, s_g = row combination g of the 4x4 patch (the operations and their order are
    // wino_input_kernel's: bit-identical V).  A stage holds positions (g, 2h), (g, 2h + 1): the even stage of a pair forms s_g, writes
    // V[.][0..1] and keeps V[.][2..3] for the odd one.  The patch lives in two register rows: px = row 0, after s_0 row 1;
    // py = row 2, after s_2 row 3 (s_0 = px - py, s_1 = px + py, s_2 = py - px, s_3 = px - py); behind s_3 both are re-loaded for the
    // next K slab.  Every re-load is two stages ahead of its first use: 16 loads per thread per K slab.
    f32x4 px[4], py[4], vh[2];
    auto raw_one = [&](f32x4* dst, int row, int c, int kslab) {
        unsigned voff;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(voff) : "v"(rowoff[row]), "v"(coloff[c]));
        dst[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff, kslab * 128, 0));
    };
    auto col_combine = [&](const f32x4* q, int j) -> f32x4 {
        return j == 0 ? y2_pk_sub(q[0], q[2]) : (j == 1 ? y2_pk_add(q[1], q[2]) : (j == 2 ? y2_pk_sub(q[2], q[1]) : y2_pk_sub(q[1], q[3])));
    };
    // the loader's share of fetch stage (kslab, fg) into ring slot `slot` (behind the four B pieces: in the vmcnt order the re-loaded
    // patch rows come AFTER them); kslab_next: the K slab behind the fetch stage's
    auto raw_fetch = [&](auto FG_, auto DRAIN_, int kslab, int kslab_next, int slot) {
        constexpr int fg = decltype(FG_)::value;
        constexpr bool drain = decltype(DRAIN_)::value;
        constexpr int g = fg >> 1;
        float* const wbuf = smem + slot * F3_STAGE_FLOATS + t * 4;
        if ((fg & 1) == 0) {
            f32x4 sv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) sv[c] = g == 1 ? y2_pk_add(px[c], py[c]) : (g == 2 ? y2_pk_sub(py[c], px[c]) : y2_pk_sub(px[c], py[c]));
            const f32x4 v0 = col_combine(sv, 0), v1 = col_combine(sv, 1);
            if (!drain) {
                *reinterpret_cast<f32x4*>(wbuf) = v0;
                *reinterpret_cast<f32x4*>(wbuf + F3_POS_FLOATS) = v1;
            }
            vh[0] = col_combine(sv, 2);
            vh[1] = col_combine(sv, 3);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (g == 0) { raw_one(px, 1, c, kslab); }
                else if (g == 2) { raw_one(py, 3, c, kslab); }
                else if (g == 3) { raw_one(px, 0, c, kslab_next); raw_one(py, 2, c, kslab_next); }
            }
        } else if (!drain) {
            *reinterpret_cast<f32x4*>(wbuf) = vh[0];
            *reinterpret_cast<f32x4*>(wbuf + F3_POS_FLOATS) = vh[1];
        }
    };

    f32x4 acc[16][2];
    const int sw = (l15 >> 1) & 7;
    int offA[2], offB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        offA[h] = (wm * 16 + l15) * 32 + (((4 * h + kq) ^ sw) << 2);
        offB[h] = (32 + wn * 32 + l15) * 32 + (((4 * h + kq) ^ sw) << 2);
    }
    const int nks = ONE ? 1 : a.Cin / 32;         // host: >= 2 unless ONE
    int slot = 0;                                 // ring slot of the stage being consumed
    int claimed = 0;
    bool more = false;
    volatile int* const sched_lds = reinterpret_cast<volatile int*>(smem + F3_NST * F3_STAGE_FLOATS);

    // One stage = two halves of 16 input channels (6 ds_read_b128 + 16 MFMAs each).  The stage barrier sits in the MIDDLE: the first
    // half runs on fragments read during the previous stage, then the wave waits for its own pieces of stage i+1 (issued a whole
    // stage ago), the barrier publishes everybody's - and says that nobody reads ring slot i-1 any more - , the pieces of stage
    // i+2 go out into that slot, the first-half fragments of stage i+1 are read, and the second half's MFMAs cover their latency.
    f32x4 a0[2], b0[2][2];                      // first-half fragments of the stage about to run [position][block]
    // (-DY2_F3EXP=1|2|3 via Y2_EXTRA_FLAGS of build.sh compiles the LDS-DMA pieces / the fragment reads out - wrong results, timing only: what
    //  DESIGN.md 3.5 prices them with)
    auto reads = [&](const float* sb, int h, f32x4* av, f32x4 (*bv)[2]) {
#if defined(Y2_F3EXP) && (Y2_F3EXP & 2)
        if (a.T < 0)
#endif
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            av[pp] = *reinterpret_cast<const f32x4*>(sb + pp * F3_POS_FLOATS + offA[h]);
            bv[pp][0] = *reinterpret_cast<const f32x4*>(sb + pp * F3_POS_FLOATS + offB[h]);
            bv[pp][1] = *reinterpret_cast<const f32x4*>(sb + pp * F3_POS_FLOATS + offB[h] + 16 * 32);
        }
    };
    // fk: K slab of fetch stage i+2; fkn: the K slab behind it (RAWIN re-loads); DRAIN: the workgroup's last two stages fetch nothing
    // (no LDS-DMA may be in flight when it ends).  The order inside a half is the compiler's: pinning it with scheduling barriers
    // (pieces and loader steps behind individual MFMAs, as in wino_fused2_kernel) measured 3-5 % slower and spilled accumulators.
    auto stage = [&](auto G_, auto ZC_, auto DRAIN_, auto IDLE_, int fk, int fkn) {
        constexpr int G = decltype(G_)::value;
        constexpr bool zc = decltype(ZC_)::value;
        constexpr bool drain = decltype(DRAIN_)::value;
        constexpr bool idle = decltype(IDLE_)::value;      // a wave without channels (HALF): no fragments, no MFMAs
        constexpr int FG = (G + 2) & 7;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 a1[2], b1[2][2];
        if (!idle) {
            if (RAWIN) reads(smem + slot * F3_STAGE_FLOATS, 0, a0, b0);      // (the loader's registers: no fragments carried across its work)
            reads(smem + slot * F3_STAGE_FLOATS, 1, a1, b1);
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) {
                const int s = idx >> 2, pp = (idx >> 1) & 1, blk = idx & 1;
                acc[2 * G + pp][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[pp][s], b0[pp][blk][s], (zc && s == 0) ? zero4 : acc[2 * G + pp][blk], 0, 0, 0);
            }
        }
        if (!RAWIN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // behind the pieces of stage i+1 sit only the patch rows re-loaded at the previous fetch point (fetch stage G + 1): they may stay in flight
            constexpr int PFG = (G + 1) & 7;
            if (PFG == 0 || PFG == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else if (PFG == 6) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const int fslot = slot == 0 ? 2 : slot - 1;
        slot = slot == 2 ? 0 : slot + 1;
#if defined(Y2_F3EXP) && (Y2_F3EXP & 1)
        if (a.T < 0)
#endif
        if (!drain) fetch(fk, FG, fslot);
        if (RAWIN) raw_fetch(std::integral_constant<int, FG>{}, DRAIN_, fk, fkn, fslot);
        else if (!idle) reads(smem + slot * F3_STAGE_FLOATS, 0, a0, b0);
        if (!idle) {
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) {
                const int s = idx >> 2, pp = (idx >> 1) & 1, blk = idx & 1;
                acc[2 * G + pp][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[pp][s], b1[pp][blk][s], acc[2 * G + pp][blk], 0, 0, 0);
            }
        }
    };
#define Y2_G(n) std::integral_constant<int, n>{}
    // the 8 stages of K slab ks.  The slab before the last one claims the workgroup's next tile from the XCD's counter (thread 0;
    // handed over through LDS across stage barriers); the last slab's fetch stream moves on to that tile: its patch rows from fetch
    // stage 6 on, its operand rows in the last two stages.
    auto slab = [&](auto ZC_, auto IDLE_, int ks) {
        const bool pen = ONE || ks == nks - 2, last = ONE || ks == nks - 1;      // ONE: the slab reads the claim made during the previous tile and makes the next one
        if (last) {
            tile = xcd * per_xcd + __builtin_amdgcn_readfirstlane(*sched_lds);
            more = tile < xcd_end;
        }
        const int fk2 = last ? 0 : ks + 1;
        using NO = std::false_type;
        stage(Y2_G(0), ZC_, NO{}, IDLE_, ks, fk2);
        stage(Y2_G(1), ZC_, NO{}, IDLE_, ks, fk2);
        stage(Y2_G(2), ZC_, NO{}, IDLE_, ks, fk2);
        if (RAWIN && last) place_a(tile, more);      // (row 3 of this tile's last K slab has just been requested: the patch offsets move on)
        stage(Y2_G(3), ZC_, NO{}, IDLE_, ks, fk2);
        stage(Y2_G(4), ZC_, NO{}, IDLE_, ks, fk2);
        if (pen && t == 0) claimed = a.sched_static ? (tile - xcd * per_xcd + wgs_per_xcd) : atomicAdd(a.sched + xcd, 1);
        stage(Y2_G(5), ZC_, NO{}, IDLE_, ks, fk2);
        if (pen && t == 0) *sched_lds = claimed;
        if (last && !more) {
            stage(Y2_G(6), ZC_, std::true_type{}, IDLE_, fk2, fk2);
            stage(Y2_G(7), ZC_, std::true_type{}, IDLE_, fk2, fk2);
        } else {
            if (last) { if (!RAWIN) place_a(tile, true); place_b(tile); }
            stage(Y2_G(6), ZC_, NO{}, IDLE_, fk2, fk2);
            stage(Y2_G(7), ZC_, NO{}, IDLE_, fk2, fk2);
        }
    };

    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, HAS_Y ? a.y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryp = __builtin_amdgcn_make_buffer_rsrc(a.y_pool, 0, HAS_POOL ? a.yp_bytes : 0, 0x00020000);
    const int so_x = a.ldy * 4, so_y = a.W * a.ldy * 4;

    // prologue: stages 0 and 1 of the first tile, the first-half fragments of stage 0
    if (RAWIN) {
        f32x4 sv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { raw_one(px, 0, c, 0); raw_one(py, 2, c, 0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) sv[c] = y2_pk_sub(px[c], py[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) raw_one(px, 1, c, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(smem + (j >> 1) * F3_STAGE_FLOATS + (j & 1) * F3_POS_FLOATS + t * 4) = col_combine(sv, j);
    }
    fetch(0, 0, 0);
    fetch(0, 1, 1);
    if (ONE && t == 0) *sched_lds = a.sched_static ? (wg_in_xcd + wgs_per_xcd) : atomicAdd(a.sched + xcd, 1);      // the first tile's successor
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    reads(smem, 0, a0, b0);
    auto run = [&](auto IDLE_) {
    for (;;) {
        const int em0 = fm0, en0 = fn0;            // this tile's origin (the fetch cursor moves on during its last K slab)
        slab(std::true_type{}, IDLE_, 0);                 // the accumulators start from the MFMA's zero operand
        if (!ONE)
            for (int ks = 1; ks < nks; ++ks) slab(std::false_type{}, IDLE_, ks);
        if (!decltype(IDLE_)::value) {
        // ---- epilogue: A^T M A in registers, affine + LeakyReLU, pooling, statistics, stores - branch-free (see wino_fused2_kernel).
        //      Accumulator register `reg` of block `blk` of the 16 positions belongs to tile row 4 * kq + reg, channel 16 * blk + l15.
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const int trow0 = em0 + wm * 16 + 4 * kq;
        const i32x4 prow = *reinterpret_cast<const i32x4*>(a.tile_pix + trow0);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int pn = en0 + wn * 32 + 16 * blk + l15;
            const bool nok = pn < a.Cout;
            const float psc = (a.scale != nullptr && nok) ? a.scale[pn] : 1.f;
            const float psh = (a.shift != nullptr && nok) ? a.shift[pn] : 0.f;
            const unsigned chan_off = (unsigned)(a.coff + pn) * 4u, pool_off = (unsigned)(a.poff + pn) * 4u;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sm[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sm[0][j] = acc[0 + j][blk][r] + acc[4 + j][blk][r] + acc[8 + j][blk][r];
                    sm[1][j] = acc[4 + j][blk][r] - acc[8 + j][blk][r] - acc[12 + j][blk][r];
                }
                float o[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    o[2 * i + 0] = sm[i][0] + sm[i][1] + sm[i][2];
                    o[2 * i + 1] = sm[i][1] - sm[i][2] - sm[i][3];
                }
                const int e = prow[r];
                const int tt = trow0 + r;
                const bool ok = nok && tt < a.T;
                const bool y1 = ok && (e & 0x40000000) != 0, x1 = ok && e < 0;
                const unsigned voff = ((unsigned)e & 0x3fffffffu) * (unsigned)so_x + chan_off;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool valid = k == 0 ? ok : (k == 1 ? x1 : (k == 2 ? y1 : (y1 && x1)));
                    if (HAS_STATS) { const float m = valid ? o[k] : 0.f; s1 += m; s2 += m * m; }
                    const float uu = o[k] * psc + psh;
                    v[k] = uu > 0.f ? uu : uu * a.slope;
                    if (HAS_Y) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), ry, (int)(valid ? voff : OOB), ((k & 1) ? so_x : 0) + ((k >> 1) ? so_y : 0), 0);
                }
                if (HAS_POOL) {
                    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx), ryp, (int)(ok ? (unsigned)tt * (unsigned)(a.ldp * 4) + pool_off : OOB), 0, 0);
                }
            }
            if (HAS_STATS) {
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (kq == 0 && nok) {
                    double* st = a.stats + (size_t)((em0 >> 6) % Y2_STATS_REPL) * 2 * a.Cout;
                    atomicAdd(st + pn, (double)s1);
                    atomicAdd(st + a.Cout + pn, (double)s2);
                }
            }
        }
        }
        if (!more) break;
    }
    };
    if (HALF && wn == 1) run(std::true_type{});
    else run(std::false_type{});
#undef Y2_G
}

// ---- weight gradient:  dU[p][co][ci] = sum_t dM[p][t][co] * V[p][t][ci],   dM = A dz A^T (4x4 from the 2x2 gradient tile),
//      dg = G^T dU G.  16 reductions over T tiles instead of 9 shifted reductions over 4T pixels (2.25x fewer MACs).
struct WinoDzArgs {
    const float* dz;
    float* dm;
    int B, H, W, Cout, ldz, th, tw, T, c4n;
    y2_fastdiv d_c4, d_tt, d_tw;
};

__global__ __launch_bounds__(256) void wino_dz_kernel(const WinoDzArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = y2_div(idx, a.d_c4);
    if (t >= (uint32_t)a.T) return;
    const int c4 = (int)(idx - t * (uint32_t)a.c4n);
    const int b = (int)y2_div(t, a.d_tt);
    const int r = (int)t - b * a.th * a.tw;
    const int ty = (int)y2_div((uint32_t)r, a.d_tw);
    const int tx = r - ty * a.tw;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 d[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yy = 2 * ty + i, xx = 2 * tx + j;
            d[i][j] = (yy < a.H && xx < a.W) ? *reinterpret_cast<const f32x4*>(a.dz + ((size_t)(b * a.H + yy) * a.W + xx) * a.ldz + 4 * c4) : zero;
        }
    f32x4 s[4][2];                 // A d   (A = [1 0; 1 1; 1 -1; 0 -1])
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s[0][j] = d[0][j];
        s[1][j] = d[0][j] + d[1][j];
        s[2][j] = d[0][j] - d[1][j];
        s[3][j] = -d[1][j];
    }
    float* dst = a.dm + (size_t)t * a.Cout + 4 * c4;
    const size_t plane = (size_t)a.T * a.Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(dst + (4 * i + 0) * plane) = s[i][0];
        *reinterpret_cast<f32x4*>(dst + (4 * i + 1) * plane) = s[i][0] + s[i][1];
        *reinterpret_cast<f32x4*>(dst + (4 * i + 2) * plane) = s[i][0] - s[i][1];
        *reinterpret_cast<f32x4*>(dst + (4 * i + 3) * plane) = -s[i][1];
    }
}

// dw_packed[co][tap][ci] = (G^T dU G)[tap];  NATIVE: dw[co][ci][tap], the nn.Conv2d.weight layout (a thread's 9 taps are 36 contiguous bytes,
// neighbouring threads continue them: no unpack pass behind the weight gradient)
template <bool NATIVE>
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ du, float* __restrict__ dw, int Cout, int Cin) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int ci = (int)(idx % Cin);
    const int co = (int)(idx / Cin);
    const size_t plane = (size_t)Cout * Cin;
    const float* src = du + (size_t)co * Cin + ci;
    float u[4][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p / 4][p % 4] = src[p * plane];
    float s[3][4];                 // G^T u   (G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1])
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
        s[1][j] = 0.5f * (u[1][j] - u[2][j]);
        s[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float* dst = NATIVE ? dw + ((size_t)co * Cin + ci) * 9 + 3 * i : dw + ((size_t)co * 9 + 3 * i) * Cin + ci;
        const size_t step = NATIVE ? 1 : (size_t)Cin;
        dst[0] = s[i][0] + 0.5f * (s[i][1] + s[i][2]);
        dst[step] = 0.5f * (s[i][1] - s[i][2]);
        dst[2 * step] = 0.5f * (s[i][1] + s[i][2]) + s[i][3];
    }
}

// ---- weight gradient on 4x4 gradient tiles: Winograd F(3x3, 4x4) - dW (3x3) is the "output", the 4x4 gradient tile the "filter", the
// 6x6 input patch the "input" - 36 multiplications per 16 gradient pixels instead of 64 (1.78x fewer MACs than the F(3x3, 2x2) form
// above on even maps; 1.36x on 13x13, whose 4-pixel grid is more ragged).  Interpolation points 0, 1, -1, 2, -1/2, inf (Cook-Toom):
//     dW = A^T [ (G dz G^T) .* (B^T x B) ] A
//   A^T = | 1  1  1  1   1   0 |   G = |   1      0      0      0   |   B^T = | 1  3/2  -2  -3/2  1  0 |
//         | 0  1 -1  2 -1/2  0 |       | -1/3   -1/3   -1/3   -1/3  |         | 0  -1  -5/2 -1/2  1  0 |
//         | 0  1  1  4  1/4  1 |       |  1/3   -1/3    1/3   -1/3  |         | 0   1   1/2 -5/2  1  0 |
//                                      |  1/15   2/15   4/15   8/15 |         | 0 -1/2  -1    1/2  1  0 |
//                                      | -16/15  8/15  -4/15   2/15 |         | 0   2   -1   -2    1  0 |
//                                      |   0      0      0      1   |         | 0   1   3/2  -2  -3/2 1 |
// The larger transform constants cost accuracy: 1.2-1.4e-5 x rms of the gradient in fp32 (simulated and measured) against 2.7e-6 for
// the 2x2 form - weight gradients only (tests hold them to 4e-5; a training step's gradients to 2e-3): activations and data
// gradients stay on F(2x2, 3x3).  Three streaming kernels around the same grouped reduction (36 groups):
//   wino6_in_kernel<false>: V6[p][t][ci] = (B^T x B)[p], x patch rows 4ty-1 .. 4ty+4 (zero outside the image)      2.25 x |x|
//   wino6_in_kernel<true>:  M6[p][t][co] = (G dz G^T)[p], gradient tile rows 4ty .. 4ty+3 (zero outside)           2.25 x |dz|
//   wino6_dw_kernel:        dW[co][ci][3][3] = A^T dU A
// one thread per (tile, channel), channel fastest: every load / store instruction of a wave covers 256 contiguous bytes.
struct Wino6Args {
    const float* src;     // [B,H,W,ld]
    float* dst;           // [36][T][C]
    int B, H, W, C, ld, th, tw, T;
    y2_fastdiv d_c, d_tt, d_tw;
    int tall, wide, gx;   // 0, or H + 1 / W + 1 / images per mosaic row: "mosaic" tiling (see Wino6Grid)
    y2_fastdiv d_tall, d_wide;
};

// (Wino6Grid / wino6_grid(): common.h - the fused BatchNorm-backward + transform kernel of train.hip cuts the same grid)

template <bool GRAD>
__global__ __launch_bounds__(256) void wino6_in_kernel(const Wino6Args a) {
    constexpr int NI = GRAD ? 4 : 6;                 // rows / columns read
    constexpr float BT[6][6] = {{1.f, 1.5f, -2.f, -1.5f, 1.f, 0.f}, {0.f, -1.f, -2.5f, -0.5f, 1.f, 0.f}, {0.f, 1.f, 0.5f, -2.5f, 1.f, 0.f},
                                {0.f, -0.5f, -1.f, 0.5f, 1.f, 0.f}, {0.f, 2.f, -1.f, -2.f, 1.f, 0.f}, {0.f, 1.f, 1.5f, -2.f, -1.5f, 1.f}};
    constexpr float GM[6][4] = {{1.f, 0.f, 0.f, 0.f}, {-1.f / 3, -1.f / 3, -1.f / 3, -1.f / 3}, {1.f / 3, -1.f / 3, 1.f / 3, -1.f / 3},
                                {1.f / 15, 2.f / 15, 4.f / 15, 8.f / 15}, {-16.f / 15, 8.f / 15, -4.f / 15, 2.f / 15}, {0.f, 0.f, 0.f, 1.f}};
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = y2_div(idx, a.d_c);
    if (t >= (uint32_t)a.T) return;
    const int c = (int)(idx - t * (uint32_t)a.C);
    int b = 0, ty, tx;
    if (a.tall) {                      // tiles of the batch's mosaic (Wino6Grid)
        ty = (int)y2_div(t, a.d_tw);
        tx = (int)t - ty * a.tw;
    } else {
        b = (int)y2_div(t, a.d_tt);
        const int r = (int)t - b * a.th * a.tw;
        ty = (int)y2_div((uint32_t)r, a.d_tw);
        tx = r - ty * a.tw;
    }
    const int y0 = 4 * ty - (GRAD ? 0 : 1), x0 = 4 * tx - (GRAD ? 0 : 1);
    int colx[NI], colb[NI];            // pixel column and image-within-mosaic-row of the patch's columns (colx < 0: no such pixel)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        int xx = x0 + j, bx = 0;
        bool ok = (unsigned)xx < (unsigned)a.W;
        if (a.wide) {
            ok = (unsigned)xx < (unsigned)(a.gx * a.wide);
            bx = ok ? (int)y2_div((uint32_t)xx, a.d_wide) : 0;
            xx -= bx * a.wide;
            ok = ok && xx < a.W;
        }
        colx[j] = ok ? xx : -1;
        colb[j] = bx;
    }
    float d[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int yy = y0 + i, by = 0;
        bool rowok = (unsigned)yy < (unsigned)a.H;
        if (a.tall) {
            rowok = yy >= 0;
            by = rowok ? (int)y2_div((uint32_t)yy, a.d_tall) : 0;
            yy -= by * a.tall;
            rowok = rowok && yy < a.H;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int bb = a.tall ? by * a.gx + colb[j] : b;
            d[i][j] = (rowok && colx[j] >= 0 && bb < a.B) ? a.src[((size_t)(bb * a.H + yy) * a.W + colx[j]) * a.ld + c] : 0.f;
        }
    }
    float sm[6][NI];               // rows transformed
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float k = GRAD ? GM[u][i] : BT[u][i];
                if (k != 0.f) v += k * d[i][j];
            }
            sm[u][j] = v;
        }
    float* dst = a.dst + (size_t)t * a.C + c;
    const size_t plane = (size_t)a.T * a.C;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v6 = 0; v6 < 6; ++v6) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float k = GRAD ? GM[v6][j] : BT[v6][j];
                if (k != 0.f) v += k * sm[u][j];
            }
            dst[(size_t)(6 * u + v6) * plane] = v;
        }
}

template <bool NATIVE>
__global__ __launch_bounds__(256) void wino6_dw_kernel(const float* __restrict__ du, float* __restrict__ dw, int Cout, int Cin) {
    constexpr float AT[3][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -0.5f, 0.f}, {0.f, 1.f, 1.f, 4.f, 0.25f, 1.f}};
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int ci = (int)(idx % Cin);
    const int co = (int)(idx / Cin);
    const size_t plane = (size_t)Cout * Cin;
    const float* src = du + (size_t)co * Cin + ci;
    float s[3][6];                 // A^T u
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int v = 0; v < 6; ++v) s[i][v] = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const float x = src[(size_t)(6 * u + v) * plane];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (AT[i][u] != 0.f) s[i][v] += AT[i][u] * x;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (AT[j][q] != 0.f) v += AT[j][q] * s[i][q];
            if (NATIVE) dw[((size_t)co * Cin + ci) * 9 + 3 * i + j] = v;
            else dw[((size_t)co * 9 + 3 * i + j) * Cin + ci] = v;
        }
}

// ---- convolution on 4x4 output tiles: Winograd F(4x4, 3x3) with the same interpolation points (Y2_ALGO_WINOGRAD_F43): 36 multiplications
// per 16 outputs instead of 64 (2x2 tiles) or 144 (direct).  Its error is 8-9e-6 x rms per layer in an fp32 model (1.3e-6 for F(2x2,3x3)):
// offered where the caller says the result is a GRADIENT (the training step's data gradients of the 13x13 layers, whose 2x2 form runs
// as three kernels anyway); inference and the training forward stay on F(2x2,3x3).  Input transform: wino6_in_kernel<false> (the B^T of
// F(3x3,4x4) and of F(4x4,3x3) are the same matrix: it depends on the points only).
//   A^T = | 1  1  1  1   1    0 |     G = |   1      0      0   |
//         | 0  1 -1  2 -1/2   0 |         | -1/3   -1/3   -1/3  |
//         | 0  1  1  4  1/4   0 |         |  1/3   -1/3    1/3  |
//         | 0  1 -1  8 -1/8   1 |         |  1/15   2/15   4/15 |
//                                         | -16/15  8/15  -4/15 |
//                                         |   0      0      1   |
__global__ __launch_bounds__(256) void wino6_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
    constexpr float GM[6][3] = {{1.f, 0.f, 0.f}, {-1.f / 3, -1.f / 3, -1.f / 3}, {1.f / 3, -1.f / 3, 1.f / 3},
                                {1.f / 15, 2.f / 15, 4.f / 15}, {-16.f / 15, 8.f / 15, -4.f / 15}, {0.f, 0.f, 1.f}};
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)Cout * Cin;
    if (idx >= total) return;
    const int k = (int)(idx % Cin);
    const int n = (int)(idx / Cin);
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((size_t)n * 9 + t) * Cin + k];      // packed [n][tap][k]
    float sm[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (GM[a][i] != 0.f) v += GM[a][i] * g[i][j];
            sm[a][j] = v;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (GM[b][j] != 0.f) v += GM[b][j] * sm[a][j];
            u[(size_t)(6 * a + b) * total + idx] = v;
        }
}

struct Wino6OutArgs {
    const float* m;       // [36][T][C]
    const float* scale; const float* shift;
    float* y;
    int B, H, W, C, ldy, coff, th, tw, T;
    float slope;
    y2_fastdiv d_c, d_tt, d_tw;
    int tall, wide, gx;   // see Wino6Grid
    y2_fastdiv d_tall, d_wide;
};

__global__ __launch_bounds__(256) void wino6_out_kernel(const Wino6OutArgs a) {
    constexpr float AT[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -0.5f, 0.f}, {0.f, 1.f, 1.f, 4.f, 0.25f, 0.f}, {0.f, 1.f, -1.f, 8.f, -0.125f, 1.f}};
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = y2_div(idx, a.d_c);
    if (t >= (uint32_t)a.T) return;
    const int c = (int)(idx - t * (uint32_t)a.C);
    int b = 0, ty, tx;
    if (a.tall) {
        ty = (int)y2_div(t, a.d_tw);
        tx = (int)t - ty * a.tw;
    } else {
        b = (int)y2_div(t, a.d_tt);
        const int r = (int)t - b * a.th * a.tw;
        ty = (int)y2_div((uint32_t)r, a.d_tw);
        tx = r - ty * a.tw;
    }
    const float* src = a.m + (size_t)t * a.C + c;
    const size_t plane = (size_t)a.T * a.C;
    float s[4][6];                 // A^T M
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 6; ++v) s[i][v] = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const float x = src[(size_t)(6 * u + v) * plane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (AT[i][u] != 0.f) s[i][v] += AT[i][u] * x;
        }
    const float sc = a.scale != nullptr ? a.scale[c] : 1.f, sh = a.shift != nullptr ? a.shift[c] : 0.f;
    int colx[4], colb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int xx = 4 * tx + j, bx = 0;
        bool ok = xx < a.W;
        if (a.wide) {
            ok = xx < a.gx * a.wide;
            bx = ok ? (int)y2_div((uint32_t)xx, a.d_wide) : 0;
            xx -= bx * a.wide;
            ok = ok && xx < a.W;
        }
        colx[j] = ok ? xx : -1;
        colb[j] = bx;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int yy = 4 * ty + i, by = 0;
        bool rowok = yy < a.H;
        if (a.tall) {
            by = (int)y2_div((uint32_t)yy, a.d_tall);
            yy -= by * a.tall;
            rowok = yy < a.H;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (AT[j][q] != 0.f) v += AT[j][q] * s[i][q];
            const int bb = a.tall ? by * a.gx + colb[j] : b;
            if (rowok && colx[j] >= 0 && bb < a.B) {
                const float uu = v * sc + sh;
                a.y[((size_t)(bb * a.H + yy) * a.W + colx[j]) * a.ldy + a.coff + c] = uu > 0.f ? uu : uu * a.slope;
            }
        }
    }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// V + M bytes per batch chunk (Y2_WINO_CHUNK_MB).  Chunks small enough to keep V and M in the 256 MB Infinity Cache were
// measured (B=32: 48 / 96 / 192 MB budgets -> 6.26 / 5.76 / 5.58 ms for the 22-layer chain against 5.50 ms unchunked): the
// smaller GEMMs lose more than the transforms gain, so the default only bounds the workspace of very large batches.
inline size_t wino_chunk_bytes() {
    const char* e = getenv("Y2_WINO_CHUNK_MB");          // read per call: tests shrink it to force several chunks
    const long long mb = (e != nullptr && atoll(e) > 0) ? atoll(e) : 4096;
    return (size_t)mb << 20;
}

}  // namespace

extern "C" int y2_wino6_weight(const float* w_packed, float* u6, int32_t Cout, int32_t Cin, y2_stream_t stream) {
    if (w_packed == nullptr || u6 == nullptr || Cout <= 0 || Cin <= 0) return Y2_EINVAL;
    const long long n = (long long)Cout * Cin;
    Y2_LAUNCH("wino6_weight_kernel", 0.0, wino6_weight_kernel, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, y2_s(stream), w_packed, u6, Cout, Cin);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// y2_conv_fwd with algo == Y2_ALGO_WINOGRAD_F43: input transform, 36 grouped GEMMs, output transform (+ affine, LeakyReLU).  y only.
int y2_internal_wino6_conv(const y2_conv_params* p, y2_stream_t stream, size_t* ws_need) {
    if (ws_need != nullptr) *ws_need = 0;
    if (p == nullptr || p->x == nullptr || p->w == nullptr) return Y2_EINVAL;
    if (p->y == nullptr || p->y_pool != nullptr || p->stats != nullptr || p->residual != nullptr || p->out_mode != 0) return Y2_ENOSUP;
    if (p->B <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->ldx < p->Cin || p->ldy < p->coff + p->Cout) return Y2_EINVAL;
    const int stride = p->stride > 0 ? p->stride : 1;
    const int pad = p->pad_plus1 > 0 ? p->pad_plus1 - 1 : 1;
    if (p->ksize != 3 || stride != 1 || pad != 1 || p->transposed != 0) return Y2_ENOSUP;
    if ((p->Cin % 4) != 0 || (p->Cout % 4) != 0 || !y2_aligned16(p->w)) return Y2_ENOSUP;
    const Wino6Grid g6 = wino6_grid(p->B, p->H, p->W);
    const int th = g6.th, tw = g6.tw;
    const long long T = g6.T;
    if (T * p->Cin >= 0xffffffffLL || T * p->Cout >= 0xffffffffLL || T > 0x7fffffff) return Y2_ENOSUP;
    const size_t vbytes = align256((size_t)36 * T * p->Cin * 4), mbytes = align256((size_t)36 * T * p->Cout * 4);
    y2_conv_params q = {};
    q.B = 1; q.H = 1; q.W = (int)T; q.Cin = p->Cin; q.ldx = p->Cin; q.Cout = p->Cout; q.ksize = 1;
    q.ldy = p->Cout; q.slope = 1.f; q.tile = p->tile; q.algo = Y2_ALGO_DIRECT;
    if (ws_need != nullptr) {
        float* const dummy = reinterpret_cast<float*>(256);
        q.x = dummy; q.w = dummy; q.y = dummy;
        size_t inner = 0;
        const int rc = y2_internal_conv_grouped(&q, 36, T * p->Cin, (long long)p->Cout * p->Cin, T * p->Cout, stream, &inner);
        if (rc != Y2_OK) return rc;
        *ws_need = (p->algo == Y2_ALGO_WINOGRAD_F43_PRE ? 0 : vbytes) + mbytes + inner;
        return Y2_OK;
    }
    // Y2_ALGO_WINOGRAD_F43_PRE: x IS the transformed input [36][T][Cin] (y2_bn_act_bwd_wino6 wrote it): no input transform, no V in the workspace
    const bool pre = p->algo == Y2_ALGO_WINOGRAD_F43_PRE;
    if (pre && (p->ldx != p->Cin || !y2_aligned16(p->x))) return Y2_EINVAL;
    if (p->workspace == nullptr || !y2_aligned16(p->workspace) || (size_t)p->workspace_bytes < (pre ? 0 : vbytes) + mbytes) return Y2_EINVAL;
    float* V = pre ? const_cast<float*>(p->x) : p->workspace;
    float* M = pre ? p->workspace : V + vbytes / sizeof(float);
    const size_t vused = pre ? 0 : vbytes;
    hipStream_t s = y2_s(stream);
    Wino6Args ia;
    ia.B = p->B; ia.H = p->H; ia.W = p->W; ia.th = th; ia.tw = tw; ia.T = (int)T;
    ia.d_tt = y2_make_fastdiv((uint32_t)(th * tw)); ia.d_tw = y2_make_fastdiv((uint32_t)tw);
    ia.tall = g6.tall; ia.wide = g6.wide; ia.gx = g6.gx;
    ia.d_tall = y2_make_fastdiv((uint32_t)(g6.tall > 0 ? g6.tall : 1)); ia.d_wide = y2_make_fastdiv((uint32_t)(g6.wide > 0 ? g6.wide : 1));
    ia.src = p->x; ia.dst = V; ia.C = p->Cin; ia.ld = p->ldx; ia.d_c = y2_make_fastdiv((uint32_t)p->Cin);
    if (!pre) Y2_LAUNCH("wino6_in_kernel", 0.0, wino6_in_kernel<false>, dim3((unsigned)y2_cdiv(T * p->Cin, 256)), dim3(256), 0, s, ia);
    q.x = V; q.w = p->w; q.y = M;
    q.workspace = M + mbytes / sizeof(float);
    q.workspace_bytes = (long long)((size_t)p->workspace_bytes - vused - mbytes);
    const int rc = y2_internal_conv_grouped(&q, 36, T * p->Cin, (long long)p->Cout * p->Cin, T * p->Cout, stream, nullptr);
    if (rc != Y2_OK) return rc;
    Wino6OutArgs oa;
    oa.m = M; oa.scale = p->scale; oa.shift = p->shift; oa.y = p->y;
    oa.B = p->B; oa.H = p->H; oa.W = p->W; oa.C = p->Cout; oa.ldy = p->ldy; oa.coff = p->coff; oa.th = th; oa.tw = tw; oa.T = (int)T; oa.slope = p->slope;
    oa.d_c = y2_make_fastdiv((uint32_t)p->Cout); oa.d_tt = ia.d_tt; oa.d_tw = ia.d_tw; oa.tall = ia.tall; oa.wide = ia.wide; oa.gx = ia.gx; oa.d_tall = ia.d_tall; oa.d_wide = ia.d_wide;
    Y2_LAUNCH("wino6_out_kernel", 0.0, wino6_out_kernel, dim3((unsigned)y2_cdiv(T * p->Cout, 256)), dim3(256), 0, s, oa);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_wino_weight(const float* w_packed, float* u, int32_t Cout, int32_t Cin, y2_stream_t stream) {
    if (w_packed == nullptr || u == nullptr || Cout <= 0 || Cin <= 0) return Y2_EINVAL;
    const long long n = (long long)Cout * Cin;
    Y2_LAUNCH("wino_weight_kernel", 0.0, wino_weight_kernel, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, y2_s(stream), w_packed, u, Cout, Cin);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// y2_conv_fwd with algo == Y2_ALGO_WINOGRAD (dispatched from conv_fwd.hip).  ws_need != nullptr: size query only.
int y2_internal_wino_conv(const y2_conv_params* p, y2_stream_t stream, size_t* ws_need) {
    if (ws_need != nullptr) *ws_need = 0;
    if (p == nullptr || p->x == nullptr || p->w == nullptr) return Y2_EINVAL;
    if (p->y == nullptr && p->y_pool == nullptr && p->stats == nullptr) return Y2_EINVAL;
    if (p->B <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0 || p->ldx < p->Cin) return Y2_EINVAL;
    const int stride = p->stride > 0 ? p->stride : 1;
    const int pad = p->pad_plus1 > 0 ? p->pad_plus1 - 1 : 1;
    if (p->ksize != 3 || stride != 1 || pad != 1 || p->transposed != 0) return Y2_ENOSUP;
    if (p->residual != nullptr || p->out_mode != 0) return Y2_ENOSUP;
    if ((p->Cin % 4) != 0 || (p->Cout % 4) != 0 || (p->ldx % 4) != 0 || !y2_aligned16(p->x) || !y2_aligned16(p->w)) return Y2_ENOSUP;
    if (p->y != nullptr && (p->ldy < p->coff + p->Cout)) return Y2_EINVAL;
    if (p->y != nullptr && ((p->ldy % 4) != 0 || (p->coff % 4) != 0 || !y2_aligned16(p->y))) return Y2_ENOSUP;
    if (p->y_pool != nullptr && ((p->H & 1) || (p->W & 1) || p->ldp < p->poff + p->Cout)) return Y2_EINVAL;
    if (p->y_pool != nullptr && ((p->ldp % 4) != 0 || (p->poff % 4) != 0 || !y2_aligned16(p->y_pool))) return Y2_ENOSUP;
    if ((p->scale != nullptr && !y2_aligned16(p->scale)) || (p->shift != nullptr && !y2_aligned16(p->shift))) return Y2_ENOSUP;
    const bool implicit = p->algo == Y2_ALGO_WINOGRAD_IMPLICIT;      // fused, and the input transform happens in the fused kernel's loader
    const bool fused = p->algo == Y2_ALGO_WINOGRAD_FUSED || implicit;
    const bool split16 = p->algo == Y2_ALGO_WINOGRAD_SPLIT_F16;      // ... as fp16 plane PAIRS (three products, operands scaled by fixed powers of two)
    const bool split = p->algo == Y2_ALGO_WINOGRAD_SPLIT || split16; // three kernels, the 16 GEMMs on the bf16 / fp16 matrix pipe (gemm_split.hip): V and w are plane tuples
    const int np = split16 ? 2 : 3;
    if ((fused || split) && (p->Cin % 32) != 0) return Y2_ENOSUP;
    if (implicit && p->Cin < 32) return Y2_ENOSUP;
    const int th = (p->H + 1) / 2, tw = (p->W + 1) / 2;
    // Batch chunks bound the workspace (V = 4x the input, M = 4x the output of a chunk); see wino_chunk_bytes().
    const size_t img_bytes = implicit ? (size_t)th * tw * sizeof(int32_t) :
                             split ? (size_t)16 * th * tw * ((size_t)p->Cin * 2 * np + (size_t)p->Cout * 4) :
                                     (size_t)16 * th * tw * ((size_t)p->Cin + (fused ? 0 : p->Cout)) * sizeof(float);
    int cb = (int)(wino_chunk_bytes() / (img_bytes > 0 ? img_bytes : 1));
    if (cb < 1) cb = 1;
    if (cb > p->B) cb = p->B;
    // the fused kernel addresses V through one 32-bit buffer descriptor: keep a chunk's V below 2 GB
    while (fused && !implicit && cb > 1 && (size_t)16 * cb * th * tw * p->Cin * sizeof(float) >= 0x7fffffffull) cb = (cb + 1) / 2;
    // the implicit kernel addresses the chunk's input through one buffer descriptor with 2^30 / 2^31 as out-of-image sentinels
    while (implicit && cb > 1 && (size_t)cb * p->H * p->W * p->ldx * sizeof(float) >= 0x40000000ull) cb = (cb + 1) / 2;
    if (implicit && (size_t)cb * p->H * p->W * p->ldx * sizeof(float) >= 0x40000000ull) return Y2_ENOSUP;
    // the split GEMM addresses the three planes of one position's V through one buffer descriptor: 66 * T * Cin bytes below 2^31
    while (split && cb > 1 && (size_t)66 * cb * th * tw * p->Cin >= 0x7fffffffull) cb = (cb + 1) / 2;
    if (split && ((size_t)66 * cb * th * tw * p->Cin >= 0x7fffffffull || (size_t)66 * p->Cout * p->Cin >= 0x7fffffffull)) return Y2_ENOSUP;
    if (split && p->w_plane != 0 && ((size_t)(2 * p->w_plane + (long long)p->Cout * p->Cin) * 2 >= 0x7fffffffull || p->w_plane < (long long)16 * p->Cout * p->Cin)) return Y2_ENOSUP;
    const int nchunks = y2_cdiv(p->B, cb);
    cb = y2_cdiv(p->B, nchunks);                     // equal chunks
    const long long T = (long long)cb * th * tw;    // tiles of a full chunk
    if (T * (p->Cin / 4) >= 0xffffffffLL || T * (p->Cout / 4) >= 0xffffffffLL || T > 0x7fffffff) return Y2_ENOSUP;
    const size_t vbytes = implicit ? 0 : align256((size_t)16 * T * p->Cin * (split ? 2 * np : sizeof(float)));
    const size_t mbytes = fused ? align256((size_t)(T + 63) * sizeof(int32_t)) + WF_DUMP_BYTES + 256 :      // ... + the 8 tile counters
                                 align256((size_t)16 * T * p->Cout * sizeof(float));   // fused: tile decode table + 1 KB dump area
    if (fused && (vbytes >= 0x7fffffffull || (size_t)16 * p->Cout * p->Cin * 4 >= 0x7fffffffull)) return Y2_ENOSUP;

    // stage 2 as a grouped 1x1 "convolution" over an image of 1 x T pixels
    y2_conv_params q = {};
    q.B = 1; q.H = 1; q.W = (int)T; q.Cin = p->Cin; q.ldx = p->Cin; q.Cout = p->Cout; q.ksize = 1;
    q.ldy = p->Cout; q.slope = 1.f; q.tile = p->tile; q.algo = Y2_ALGO_DIRECT;
    if (ws_need != nullptr) {
        float* const dummy = reinterpret_cast<float*>(256);
        q.x = dummy; q.w = dummy; q.y = dummy;
        size_t inner = 0;
        if (!fused && !split) {
            const int rc = y2_internal_conv_grouped(&q, 16, T * p->Cin, (long long)p->Cout * p->Cin, T * p->Cout, stream, &inner);
            if (rc != Y2_OK) return rc;
        }
        *ws_need = vbytes + mbytes + inner;
        return Y2_OK;
    }
    if (p->workspace == nullptr || !y2_aligned16(p->workspace) || (size_t)p->workspace_bytes < vbytes + mbytes) return Y2_EINVAL;
    float* V = p->workspace;
    float* M = V + vbytes / sizeof(float);
    hipStream_t s = y2_s(stream);
    const y2_fastdiv d_tt = y2_make_fastdiv((uint32_t)(th * tw)), d_tw = y2_make_fastdiv((uint32_t)tw);

    for (int b0 = 0; b0 < p->B; b0 += cb) {
        const int nb = p->B - b0 < cb ? p->B - b0 : cb;
        const long long Tc = (long long)nb * th * tw;
        const size_t in_off = (size_t)b0 * p->H * p->W;               // pixels before this chunk

        WinoInArgs ia;
        ia.x = p->x + in_off * p->ldx; ia.v = V; ia.B = nb; ia.H = p->H; ia.W = p->W; ia.Cin = p->Cin; ia.ldx = p->ldx; ia.th = th; ia.tw = tw;
        ia.T = (int)Tc; ia.c4n = p->Cin / 4;
        ia.d_c4 = y2_make_fastdiv((uint32_t)ia.c4n); ia.d_tt = d_tt; ia.d_tw = d_tw;
        ia.tile_pix = fused ? reinterpret_cast<int32_t*>(M) : nullptr;      // the fused path has no product tensor: the table sits behind V
        // persistent fused kernel: one workgroup per CU; the kernel launched in front of it also sets the per-XCD tile counters to
        // "every workgroup has taken its first tile"
        const char* ve = getenv("Y2_WF_VARIANT");           // read per call (experiments / tests switch it at run time)
        const int variant = ve != nullptr ? atoi(ve) : WF2_DEFAULT_VARIANT;
        const bool small_out = (unsigned long long)p->B * p->H * p->W * (unsigned long long)(p->ldy > p->ldp ? p->ldy : p->ldp) * 4ull < 0x80000000ull;      // bytes: 2^31 is the dropped-store offset
        const bool gen3 = fused && (variant >= 32 || p->tile == 3) && small_out;      // wino_fused3_kernel (y2_conv_params.tile = 3): 32 x 64 units, two workgroups per CU
        const int unit_m = gen3 ? F3_TM : 64, slots = gen3 ? 2 * Y2_NUM_CU : Y2_NUM_CU;
        const long long fused_tiles = (long long)y2_cdiv(Tc, unit_m) * y2_cdiv(p->Cout, 64);
        const long long fused_grid = fused_tiles < slots ? ((fused_tiles + Y2_NUM_XCD - 1) / Y2_NUM_XCD) * Y2_NUM_XCD : slots;
        ia.sched = fused ? reinterpret_cast<int32_t*>(M + (mbytes - 256) / sizeof(float)) : nullptr;
        ia.sched_init = (int)(fused_grid / Y2_NUM_XCD);
        if (implicit) Y2_LAUNCH("wino_tile_table_kernel", 0.0, wino_tile_table_kernel, dim3((unsigned)y2_cdiv(Tc, 256)), dim3(256), 0, s, ia.tile_pix, ia.T, ia.H, ia.W, th, tw, d_tt, d_tw, ia.sched, ia.sched_init);
        else if (!split) Y2_LAUNCH("wino_input_kernel", 0.0, wino_input_kernel, dim3((unsigned)y2_cdiv(Tc * ia.c4n, 256)), dim3(256), 0, s, ia);

        if (fused) {
            WinoFusedArgs fa;
            fa.y_bytes = (unsigned)((size_t)nb * p->H * p->W * (p->y != nullptr ? p->ldy : 0) * sizeof(float));
            fa.yp_bytes = (unsigned)((size_t)nb * th * tw * (p->y_pool != nullptr ? p->ldp : 0) * sizeof(float));
            fa.x = ia.x; fa.ldx = p->ldx; fa.x_bytes = (unsigned)((size_t)nb * p->H * p->W * p->ldx * sizeof(float));
            fa.v = V; fa.u = p->w; fa.tile_pix = ia.tile_pix; fa.dump = M + (mbytes - 256 - WF_DUMP_BYTES) / sizeof(float); fa.sched = ia.sched; fa.scale = p->scale; fa.shift = p->shift; fa.stats = p->stats;
            fa.y = p->y != nullptr ? p->y + in_off * p->ldy : nullptr;
            fa.y_pool = p->y_pool != nullptr ? p->y_pool + (size_t)b0 * th * tw * p->ldp : nullptr;
            fa.H = p->H; fa.W = p->W; fa.Cin = p->Cin; fa.Cout = p->Cout; fa.ldy = p->ldy; fa.coff = p->coff; fa.ldp = p->ldp; fa.poff = p->poff;
            fa.th = th; fa.tw = tw; fa.T = (int)Tc; fa.tiles_m = y2_cdiv(Tc, unit_m); fa.tiles_n = y2_cdiv(p->Cout, 64);
            fa.v_bytes = implicit ? 0u : (unsigned)((size_t)16 * Tc * p->Cin * 4); fa.u_bytes = (unsigned)((size_t)16 * p->Cout * p->Cin * 4);
            fa.slope = p->slope; fa.d_tt = d_tt; fa.d_tw = d_tw;
            { const char* se = getenv("Y2_WF_STATIC"); fa.sched_static = (se != nullptr && atoi(se) != 0) ? 1 : 0; }
            const long long ntiles = (long long)fa.tiles_m * fa.tiles_n;
            if (ntiles > 0x7fffffffLL) return Y2_EINVAL;
            const long long grid = fused_grid;
#define Y2_WF_LAUNCH(PG_, NST_)                                                                                                    \
            do {                                                                                                                   \
                auto kern = wino_fused_kernel<PG_, NST_>;                                                                          \
                const size_t lds = (size_t)NST_ * PG_ * WF_POS_FLOATS * sizeof(float);                                             \
                static Y2LdsAttr attr;                                                                                             \
                if (const int rc_ = attr.ensure(reinterpret_cast<const void*>(kern))) return rc_;                                  \
                Y2_LAUNCH("wino_fused_kernel", 2.0 * 16.0 * (double)fa.T * fa.Cout * fa.Cin, kern, dim3((unsigned)grid), dim3(256), lds, s, fa);                                             \
            } while (0)
#define Y2_WF2_LAUNCH_(VAR_, OUT_)                                                                                                 \
            do {                                                                                                                   \
                auto kern = wino_fused2_kernel<VAR_, OUT_>;                                                                        \
                const size_t lds = (size_t)2 * 4 * WF_POS_FLOATS * sizeof(float) + (WF_DUMP_BYTES - 1024) + 16;                    \
                static Y2LdsAttr attr;                                                                                             \
                if (const int rc_ = attr.ensure(reinterpret_cast<const void*>(kern))) return rc_;                                  \
                Y2_LAUNCH(((VAR_) & 4) ? "wino_fused2_kernel[implicit]" : "wino_fused2_kernel", 2.0 * 16.0 * (double)fa.T * fa.Cout * fa.Cin, kern, dim3((unsigned)grid), dim3(256), lds, s, fa); \
            } while (0)
#define Y2_WF2_LAUNCH(VAR_)                                                                                                        \
            do {                                                                                                                   \
                if (out_mask == 1) Y2_WF2_LAUNCH_(VAR_, 1);                                                                        \
                else if (out_mask == 2) Y2_WF2_LAUNCH_(VAR_, 2);                                                                   \
                else if (out_mask == 3) Y2_WF2_LAUNCH_(VAR_, 3);                                                                   \
                else if (out_mask == 5) Y2_WF2_LAUNCH_(VAR_, 5);                                                                   \
                else Y2_WF2_LAUNCH_(VAR_, 7);                                                                                      \
            } while (0)
            const int out_mask = (p->y != nullptr ? 1 : 0) | (p->y_pool != nullptr ? 2 : 0) | (p->stats != nullptr ? 4 : 0);
            // second-generation instruction stream (see wino_fused2_kernel); Y2_WF_VARIANT = -1 selects the first generation, 0 / 1 / 3 a
            // feature mask (A/B runs), 32 the third generation.  Their buffer-descriptor stores need the output tensors below 2^31 bytes.
#define Y2_WF3_LAUNCH_(VAR_, OUT_)                                                                                                 \
            do {                                                                                                                   \
                auto kern = wino_fused3_kernel<VAR_, OUT_>;                                                                        \
                const size_t lds = (size_t)F3_NST * F3_STAGE_FLOATS * sizeof(float) + 64;                                          \
                static Y2LdsAttr attr;                                                                                             \
                if (const int rc_ = attr.ensure(reinterpret_cast<const void*>(kern))) return rc_;                                  \
                Y2_LAUNCH(((VAR_) & 4) ? "wino_fused3_kernel[implicit]" : "wino_fused3_kernel", 2.0 * 16.0 * (double)fa.T * fa.Cout * fa.Cin, kern, dim3((unsigned)grid), dim3(256), lds, s, fa); \
            } while (0)
#define Y2_WF3_LAUNCH(VAR_)                                                                                                        \
            do {                                                                                                                   \
                if (out_mask == 1) Y2_WF3_LAUNCH_(VAR_, 1);                                                                        \
                else if (out_mask == 2) Y2_WF3_LAUNCH_(VAR_, 2);                                                                   \
                else if (out_mask == 3) Y2_WF3_LAUNCH_(VAR_, 3);                                                                   \
                else if (out_mask == 5) Y2_WF3_LAUNCH_(VAR_, 5);                                                                   \
                else Y2_WF3_LAUNCH_(VAR_, 7);                                                                                      \
            } while (0)
            if (gen3) {
                if (p->Cin == 32) { if (implicit) Y2_WF3_LAUNCH(12); else Y2_WF3_LAUNCH(8); }      // a tile is one K slab
                else if (implicit && p->Cout <= 32) Y2_WF3_LAUNCH(20);                              // half-empty units: the waves without channels idle
                else if (implicit) Y2_WF3_LAUNCH(4);
                else Y2_WF3_LAUNCH(0);
                continue;
            }
#undef Y2_WF3_LAUNCH
#undef Y2_WF3_LAUNCH_
            if (implicit) {
                if (!small_out) return Y2_ENOSUP;
                if (p->Cin == 32) { Y2_WF2_LAUNCH(13); continue; }      // a tile is one K slab
                Y2_WF2_LAUNCH(5);           // no fragment double buffering: its 32 registers hold patch rows (with it: spills, 3-7 % slower)
                continue;
            }
            if (variant >= 0 && p->Cin == 32 && small_out) {
                Y2_WF2_LAUNCH(11);
                continue;
            }
            if (variant >= 0 && p->Cin >= 64 && small_out) {
                switch (variant) {
                    case 0: Y2_WF2_LAUNCH(0); break;       // branch-free epilogue only
                    case 1: Y2_WF2_LAUNCH(1); break;       // + DMA spread
                    default: Y2_WF2_LAUNCH(3); break;      // + fragment double buffering
                }
                continue;
            }
#undef Y2_WF2_LAUNCH
#undef Y2_WF2_LAUNCH_
            // measured on the 52x52 / 26x26 / 13x13 layers (B=32): 4 positions per stage + 2-deep ring (64 MFMAs per wave between
            // barriers) 0.341 / 0.283 / 0.330 ms; 2 positions x 4-deep 0.357 / 0.289 / 0.336; 1 position x 6-deep 0.373 / 0.315 / 0.369
            Y2_WF_LAUNCH(4, 2);
#undef Y2_WF_LAUNCH
            continue;
        }
        if (split) {
            // stage 1 again with plane output (the fp32 launch above is skipped: see the `split` test in front of it), stage 2 on the bf16 pipe
            const int rc1 = y2_internal_wino_input_split(ia.x, V, nb, p->H, p->W, p->Cin, p->ldx, np, Y2_F16_V_SCALE, stream);
            if (rc1 != Y2_OK) return rc1;
            const int rc2 = y2_internal_gemm_split(V, 0, p->w, p->w_plane, M, Tc, p->Cout, p->Cin, p->Cout, 16, np, split16 ? Y2_F16_OUT_SCALE : 1.f, stream);
            if (rc2 != Y2_OK) return rc2;
        } else {
        q.W = (int)Tc;
        q.x = V; q.w = p->w; q.y = M;
        q.workspace = M + mbytes / sizeof(float);
        q.workspace_bytes = (long long)((size_t)p->workspace_bytes - vbytes - mbytes);
        const int rc = y2_internal_conv_grouped(&q, 16, Tc * p->Cin, (long long)p->Cout * p->Cin, Tc * p->Cout, stream, nullptr);
        if (rc != Y2_OK) return rc;
        }

        WinoOutArgs oa;
        oa.m = M; oa.scale = p->scale; oa.shift = p->shift; oa.stats = p->stats;
        oa.y = p->y != nullptr ? p->y + in_off * p->ldy : nullptr;
        oa.y_pool = p->y_pool != nullptr ? p->y_pool + (size_t)b0 * th * tw * p->ldp : nullptr;
        oa.B = nb; oa.H = p->H; oa.W = p->W; oa.Cout = p->Cout; oa.ldy = p->ldy; oa.coff = p->coff; oa.ldp = p->ldp; oa.poff = p->poff;
        oa.th = th; oa.tw = tw; oa.T = (int)Tc; oa.n4n = p->Cout / 4; oa.slope = p->slope;
        oa.d_tt = d_tt; oa.d_tw = d_tw;
        int nx = 64;
        while (nx > 4 && nx / 2 >= oa.n4n) nx /= 2;
        const int ny = 256 / nx;
        oa.loop = p->stats != nullptr ? 8 : 1;
        const dim3 grid((unsigned)y2_cdiv(Tc, (long long)ny * oa.loop), (unsigned)y2_cdiv(oa.n4n, nx));
        if (p->stats != nullptr) Y2_LAUNCH("wino_output_kernel", 0.0, wino_output_kernel<true>, grid, dim3(nx, ny), 0, s, oa);
        else Y2_LAUNCH("wino_output_kernel", 0.0, wino_output_kernel<false>, grid, dim3(nx, ny), 0, s, oa);
    }
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// Winograd weight gradient of a 3x3 / stride-1 / same-padding convolution (replaces y2_conv_wgrad for the deep layers).
extern "C" long long y2_wino_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return Y2_EINVAL;
    const long long T = (long long)B * ((H + 1) / 2) * ((W + 1) / 2), T6 = (long long)B * ((H + 3) / 4) * ((W + 3) / 4);
    const long long R = 16 * T > 36 * T6 ? 16 * T : 36 * T6;      // operand rows: 16 positions x 2x2 tiles, or 36 x 4x4 tiles (more only on maps of 1-2 pixels)
    return (long long)(align256((size_t)R * Cin * 4) + align256((size_t)R * Cout * 4) + align256((size_t)36 * Cout * Cin * 4));
}

// ... of ONE form (native_layout as y2_wino_wgrad_ex takes it): the 2x2-tile form moves 16 T rows per operand, the 4x4-tile form 36 T6 (2.25 instead of 4 times the
// tensors: 2.55 instead of 4.5 GB for the 152x152 64 -> 128 layer at batch 64), and with a transformed gradient (bit 2) no gradient slot at all
extern "C" long long y2_wino_wgrad_workspace_bytes_ex(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t native_layout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return Y2_EINVAL;
    if (native_layout & 2) {
        const long long T6 = wino6_grid(B, H, W).T;
        return (long long)(align256((size_t)36 * T6 * Cin * 4) + align256((size_t)36 * T6 * Cout * 4) + align256((size_t)36 * Cout * Cin * 4));
    }
    const long long T = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
    return (long long)(align256((size_t)16 * T * Cin * 4) + align256((size_t)16 * T * Cout * 4) + align256((size_t)16 * Cout * Cin * 4));
}

extern "C" int y2_wino_wgrad(const float* x, const float* dz, float* dw_packed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t ldx,
                             int32_t Cout, int32_t ldz, const float* v_transformed, float* workspace, long long workspace_bytes, y2_stream_t stream) {
    return y2_wino_wgrad_ex(x, dz, dw_packed, B, H, W, Cin, ldx, Cout, ldz, v_transformed, workspace, workspace_bytes, 0, stream);
}

// Zero fill of the split accumulators by a KERNEL, not hipMemsetAsync: captured into a hipGraph in front of the kernels that accumulate into the buffer,
// the memset node did its job at the graph's first launch and not at the later ones on this runtime (ROCm 7.0 / 7.2: a round-4 replay probe, docs/history - every
// later replay accumulated onto whatever the scratch held; a memset node with nothing behind it replays fine, same probe, so it is the
// ordering against the following kernel nodes that is lost).  The training step is a replayed graph (model.train_graph.StepPlan).
__global__ void zero_fill_kernel(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static int zero_fill(float* p, size_t bytes, hipStream_t s) {      // bytes % 16 == 0, p 16-byte aligned (workspace sections are 256-byte aligned)
    const size_t n4 = bytes / 16;
    if (n4 == 0) return Y2_OK;
    const unsigned grid = (unsigned)(n4 / 256 + 1 < 2048 ? n4 / 256 + 1 : 2048);
    Y2_LAUNCH("zero_fill_kernel", 0.0, zero_fill_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<float4*>(p), n4);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// F(3x3, 4x4): x and dz transformed on 4x4 gradient tiles, 36 grouped reductions, dW = A^T dU A (see wino6_in_kernel)
static int wino6_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int ldx, int Cout, int ldz, float* workspace, bool native, y2_stream_t stream, bool dz_pre = false) {
    if (x == nullptr) return Y2_EINVAL;                  // (a transformed input of the 2x2 form is of no use here)
    if (y2_det.on) return Y2_ENOSUP;
    const Wino6Grid g6 = wino6_grid(B, H, W);
    const int th = g6.th, tw = g6.tw;
    const long long T = g6.T;          // (never more than the per-image grid's: y2_wino_wgrad_workspace_bytes covers it)
    if (T * Cin >= 0xffffffffLL || T * Cout >= 0xffffffffLL || T > 0x7fffffff) return Y2_ENOSUP;
    float* V = workspace;
    float* DM = V + align256((size_t)36 * T * Cin * 4) / 4;
    float* DU = DM + align256((size_t)36 * T * Cout * 4) / 4;
    hipStream_t s = y2_s(stream);
    if (y2_internal_wgrad_needs_zero(T, Cin, Cout, 36)) {
        if (const int rc = zero_fill(DU, (size_t)36 * Cout * Cin * 4, s)) return rc;
    }
    Wino6Args a;
    a.B = B; a.H = H; a.W = W; a.th = th; a.tw = tw; a.T = (int)T;
    a.d_tt = y2_make_fastdiv((uint32_t)(th * tw)); a.d_tw = y2_make_fastdiv((uint32_t)tw);
    a.tall = g6.tall; a.wide = g6.wide; a.gx = g6.gx;
    a.d_tall = y2_make_fastdiv((uint32_t)(g6.tall > 0 ? g6.tall : 1)); a.d_wide = y2_make_fastdiv((uint32_t)(g6.wide > 0 ? g6.wide : 1));
    a.src = x; a.dst = V; a.C = Cin; a.ld = ldx; a.d_c = y2_make_fastdiv((uint32_t)Cin);
    Y2_LAUNCH("wino6_in_kernel", 0.0, wino6_in_kernel<false>, dim3((unsigned)y2_cdiv(T * Cin, 256)), dim3(256), 0, s, a);
    if (!dz_pre) {          // (dz_pre: `dz` IS the transformed gradient [36][T][Cout], written by y2_bn_act_bwd_wino6)
        a.src = dz; a.dst = DM; a.C = Cout; a.ld = ldz; a.d_c = y2_make_fastdiv((uint32_t)Cout);
        Y2_LAUNCH("wino6_in_kernel[dz]", 0.0, wino6_in_kernel<true>, dim3((unsigned)y2_cdiv(T * Cout, 256)), dim3(256), 0, s, a);
    }
    const int rc = y2_internal_wgrad_grouped(V, dz_pre ? dz : DM, DU, T, Cin, Cout, 36, T * Cin, T * Cout, (long long)Cout * Cin, stream);
    if (rc != Y2_OK) return rc;
    const long long n = (long long)Cout * Cin;
    if (native) Y2_LAUNCH("wino6_dw_kernel", 0.0, wino6_dw_kernel<true>, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, s, DU, dw, Cout, Cin);
    else Y2_LAUNCH("wino6_dw_kernel", 0.0, wino6_dw_kernel<false>, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, s, DU, dw, Cout, Cin);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_wino_wgrad_ex(const float* x, const float* dz, float* dw_packed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t ldx,
                                int32_t Cout, int32_t ldz, const float* v_transformed, float* workspace, long long workspace_bytes, int32_t native_layout,
                                y2_stream_t stream) {
    if ((x == nullptr && v_transformed == nullptr) || dz == nullptr || dw_packed == nullptr || workspace == nullptr) return Y2_EINVAL;
    if (v_transformed != nullptr && !y2_aligned16(v_transformed)) return Y2_EALIGN;
    if (x == nullptr) x = v_transformed;      // only the alignment checks below look at it
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldz < Cout) return Y2_EINVAL;
    if ((Cin & 3) || (Cout & 3) || (ldx & 3) || (ldz & 3) || !y2_aligned16(x) || !y2_aligned16(dz) || !y2_aligned16(workspace)) return Y2_EALIGN;
    if (workspace_bytes < y2_wino_wgrad_workspace_bytes_ex(B, H, W, Cin, Cout, native_layout)) return Y2_EINVAL;
    if ((native_layout & 4) && !(native_layout & 2)) return Y2_EINVAL;          // a transformed gradient exists for the 4x4-tile form only
    if (native_layout & 2) return wino6_wgrad(v_transformed != nullptr && x == v_transformed ? nullptr : x, dz, dw_packed, B, H, W, Cin, ldx, Cout, ldz, workspace, (native_layout & 1) != 0, stream, (native_layout & 4) != 0);
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    if (T * (Cin / 4) >= 0xffffffffLL || T * (Cout / 4) >= 0xffffffffLL || T > 0x7fffffff) return Y2_ENOSUP;
    float* V = workspace;
    float* DM = V + align256((size_t)16 * T * Cin * 4) / 4;
    float* DU = DM + align256((size_t)16 * T * Cout * 4) / 4;
    hipStream_t s = y2_s(stream);
    const size_t du_bytes = (size_t)16 * Cout * Cin * 4;
    if (y2_internal_wgrad_needs_zero(T, Cin, Cout, 16)) {      // split partial sums are added atomically; an unsplit launch (the deep layers) stores
        if (const int rc = zero_fill(DU, du_bytes, s)) return rc;
    }

    WinoInArgs ia;
    ia.tile_pix = nullptr; ia.sched = nullptr; ia.sched_init = 0;
    ia.x = x; ia.v = V; ia.B = B; ia.H = H; ia.W = W; ia.Cin = Cin; ia.ldx = ldx; ia.th = th; ia.tw = tw; ia.T = (int)T; ia.c4n = Cin / 4;
    ia.d_c4 = y2_make_fastdiv((uint32_t)ia.c4n); ia.d_tt = y2_make_fastdiv((uint32_t)(th * tw)); ia.d_tw = y2_make_fastdiv((uint32_t)tw);
    if (v_transformed == nullptr) Y2_LAUNCH("wino_input_kernel", 0.0, wino_input_kernel, dim3((unsigned)y2_cdiv(T * ia.c4n, 256)), dim3(256), 0, s, ia);
    const float* Vsrc = v_transformed != nullptr ? v_transformed : V;

    // the gradient operand dM = A dz A^T: formed in the loader of the grouped reduction from the raw gradient (no wino_dz_kernel, no 4x tensor) -
    // Y2_WGRAD_DZRAW=0: the three-step form (A/B runs, and the fallback when the raw gradient does not fit one buffer descriptor)
    static const bool dzraw_env = getenv("Y2_WGRAD_DZRAW") == nullptr || atoi(getenv("Y2_WGRAD_DZRAW")) != 0;
    int rc = Y2_ENOSUP;
    if (dzraw_env && !y2_det.on && Cout <= 256) {      // measured (B=64): 104x104 64->128 0.99 -> 0.80 ms, 52x52 128->256 0.63 -> 0.56; 26x26 256->512 0.49 -> 0.52, 13x13 layers +15-20 %
        int32_t* table = reinterpret_cast<int32_t*>(DM);                 // (the tensor's place in the workspace)
        Y2_LAUNCH("wino_tile_table_kernel", 0.0, wino_tile_table_kernel, dim3((unsigned)y2_cdiv(T, 256)), dim3(256), 0, s, table, (int)T, H, W, th, tw, ia.d_tt, ia.d_tw, (int32_t*)nullptr, 0);
        rc = y2_internal_wgrad_grouped_dz(Vsrc, dz, table, DU, T, Cin, Cout, ldz, W, (unsigned long long)B * H * W * ldz * 4ull, T * Cin, (long long)Cout * Cin, stream);
    }
    if (rc == Y2_ENOSUP) {
        WinoDzArgs za;
        za.dz = dz; za.dm = DM; za.B = B; za.H = H; za.W = W; za.Cout = Cout; za.ldz = ldz; za.th = th; za.tw = tw; za.T = (int)T; za.c4n = Cout / 4;
        za.d_c4 = y2_make_fastdiv((uint32_t)za.c4n); za.d_tt = ia.d_tt; za.d_tw = ia.d_tw;
        Y2_LAUNCH("wino_dz_kernel", 0.0, wino_dz_kernel, dim3((unsigned)y2_cdiv(T * za.c4n, 256)), dim3(256), 0, s, za);
        rc = y2_internal_wgrad_grouped(Vsrc, DM, DU, T, Cin, Cout, 16, T * Cin, T * Cout, (long long)Cout * Cin, stream);
    }
    if (rc != Y2_OK) return rc;
    const long long n = (long long)Cout * Cin;
    if ((native_layout & 1) != 0) Y2_LAUNCH("wino_dw_kernel", 0.0, wino_dw_kernel<true>, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, s, DU, dw_packed, Cout, Cin);
    else Y2_LAUNCH("wino_dw_kernel", 0.0, wino_dw_kernel<false>, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, s, DU, dw_packed, Cout, Cin);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
