// conv_fwd.hip — fp32 MFMA implicit-GEMM convolution for gfx950 (MI355X), fused epilogue.
//
// Replaces model.yolo2.Conv2d.forward (model/yolo2.py:61-65: nn.Conv2d stride 1 "same" padding ->
// BatchNorm2d (eval, folded) or bias -> LeakyReLU(0.1)) together with the MaxPool2d(2) that follows it
// (model/yolo2.py:79,86,97), the passthrough reorg (model/yolo2.py:33-46) and the concat
// (model/yolo2.py:129), which become output addressing.
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[k][n]
//   m = output pixel (b, y, x)            M = B*H*W
//   n = output channel                    N = Cout
//   k = (tap, cin)                        K = ksize^2 * Cin       (tap-major, cin contiguous)
//   A[m][k] = x[b, y+dy, x+dx, cin] (0 outside the image)   -> gathered on the fly from NHWC
//   B[k][n] = w[n][tap][cin]                                 -> y2_pack_weight mode 0
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate — exact fp32, the parity requirement of
// BASELINE.json "conv tensors within an fp32 tolerance"; gfx950 has no xf32).  Peak 157.3 TFLOP/s.
//
// Workgroup = 256 threads = 4 waves, tile BM x BN, K-slab BK = 32 per pipeline step:
//   global -> registers (16-B loads, prefetch of slab s+1 issued before the MFMAs of slab s)
//          -> LDS (double buffered, row stride 36 floats = 144 B so that ds_read_b128 of 16 different
//             rows hits 16 different 16-B slots of the 256-B bank row: conflict-free)
//          -> each lane reads a float4 = 4 consecutive k of its row; lanes 0-31 take k-offset 0 and lanes
//             32-63 k-offset 4 inside each group of 8, so the 4 MFMAs fed by one read contract
//             k = {j, 4+j} (j = 0..3).  A and B use the same permutation, so the sum over k is complete.
//   one __syncthreads() per slab.  2 workgroups per CU stay resident (<= 73.7 KB LDS each).
//
// Row order inside a tile (POOLORD): when a 2x2 max-pool follows, pixel index m enumerates the image in
// 2x2-window-major order (m = ((p*W/2 + q)*4 + wy*2 + wx), y = 2p+wy, x = 2q+wx), so the four rows held by
// one lane in accumulator registers 4g..4g+3 (MFMA C layout: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) are
// exactly one pooling window: the pool is 3 v_max per output in-lane, and m>>2 is the pooled pixel's
// ordinary row-major index.
// y2-build-flags: -mllvm -amdgpu-mfma-vgpr-form
//   (accumulators in ordinary VGPRs: the epilogues (affine / activation / statistics / pooled stores) read every accumulator once; from AGPRs that is a v_accvgpr_read apiece, and the kernels need 4 - 40 fewer registers)
#include <stdlib.h>
#include "common.h"

#ifndef Y2_CONV_SPREAD
#define Y2_CONV_SPREAD 1      // 1: issue the LDS-DMA of slab s+1 between the MFMAs of slab s (see compute_slab_spread); 0: all in front (A/B builds)
#endif

namespace {

constexpr int BK_DEFAULT = 32;   // floats of K per pipeline step (template parameter BK of the kernel)
constexpr int NT = 256;

__device__ __attribute__((aligned(16))) float y2_zero16_storage[4] = {0.f, 0.f, 0.f, 0.f};

struct ConvArgs {
    const float* x;
    const float* w;
    const float* zeros;   // 16 B of zeros in global memory (masked loads read here; a kernarg pointer keeps them global_load)
    const float* scale;
    const float* shift;
    float* y;
    float* y_pool;
    double* stats;
    int B, H, W, Cin, ldx, Cout, taps;  // taps = 1 or 9
    int ldy, coff, ldp, poff, out_mode;
    float slope;
    int M, tiles_m, tiles_n, cchunks;   // cchunks = ceil(Cin / BK)
    unsigned x_bytes, w_bytes;          // buffer-descriptor ranges of x and w (DMA kernel)
    // split-K of the remainder tiles (see launch_dma): blocks [0, full_tiles) compute whole tiles; the following blocks
    // compute 1/ksplit of the K range of tile full_tiles + (block - full_tiles) / ksplit and park raw accumulators in `partial`
    int full_tiles, ksplit;
    float* partial;
    y2_fastdiv d_hw, d_w, d_w2;         // exact division by H*W, W, 2*W (row decode)
    // general convolution (GEN kernels: any stride / kernel size / padding, K = k*k*Cin treated as one linear axis)
    int stride, pad, KW, Ho, Wo, K;
    int tstride;                        // > 1: transposed (fractionally strided) convolution = data gradient of a strided conv
    y2_fastdiv d_ts;
    const float* res;                   // optional residual added before the activation: res[m*ldr + n]
    int ldr;
    y2_fastdiv d_cin, d_kw, d_howo, d_wo;
    // grouped (batched) GEMM: `groups` independent problems of identical shape; group g reads x + g*gx, w + g*gw and writes
    // y + g*gy (floats).  Tiles are numbered group-major.  Used by the Winograd path (16 transform positions).
    int groups;
    long long gx, gw, gy;
    int fast_epi, epi_bm, epi_bn;       // straight-line epilogue allowed (host) + the launch's tile size for the per-tile inside test
};

// pixel index (b*H + y)*W + x and (y, x) of GEMM row m
template <bool POOLORD>
__device__ __forceinline__ void decode_row(const ConvArgs& a, int m, int& pix, int& y, int& x) {
    const int hw = a.H * a.W;
    const int b = (int)y2_div((uint32_t)m, a.d_hw);
    const int idx = m - b * hw;
    if (POOLORD) {
        const int w2 = 2 * a.W;
        const int p = (int)y2_div((uint32_t)idx, a.d_w2);
        const int rem = idx - p * w2;
        y = 2 * p + ((rem >> 1) & 1);
        x = 2 * (rem >> 2) + (rem & 1);
    } else {
        y = (int)y2_div((uint32_t)idx, a.d_w);
        x = idx - y * a.W;
    }
    pix = (b * a.H + y) * a.W + x;
}

// ------------------------------------------------------------------------------------------------ epilogue
// Straight-line epilogue for the common cases (decided once per tile, uniformly): no residual, plain output
// addressing, the tile completely inside the problem.  The general epilogue below tests per element what these cases know per
// launch (measured on the second-generation fused Winograd kernel: control flow and dead arithmetic in an epilogue cost more
// than its stores); on the short-K layers (1x1 convolutions, K = 128 ... 1024) the epilogue is a visible share of a tile.
//   non-POOLORD: y[(m) * ldy + coff + n] = act(z * scale + shift)          (also the raw product tensor of the grouped Winograd GEMM)
//   POOLORD:     y_pool[(m >> 2) * ldp + poff + n] = max over the lane's 2x2 window (pooled output only)
template <int MB, int NB, int WM, int WN, bool POOLORD>
__device__ __forceinline__ void conv_epilogue_fast(const ConvArgs& a, f32x16 (&acc)[MB][NB], int m0, int n0, int wm, int wn, int l31, int half) {
    const bool affine = a.scale != nullptr || a.shift != nullptr || a.slope != 1.f;       // uniform
    const bool stats = !POOLORD && a.stats != nullptr;                                        // uniform
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        float sc = 1.f, sh = 0.f;
        if (affine) {
            if (a.scale != nullptr) sc = a.scale[n];
            if (a.shift != nullptr) sh = a.shift[n];
        }
        const int mrow = m0 + wm * WM + 4 * half;
        if (!POOLORD) {
            float* col = a.y + (size_t)mrow * a.ldy + a.coff + n;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (stats) { s1 += v; s2 += v * v; }          // training-mode BatchNorm statistics of the RAW output
                    if (affine) { const float u = v * sc + sh; v = u > 0.f ? u : u * a.slope; }
                    col[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * a.ldy] = v;
                }
            if (stats) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (half == 0) {
                    double* st = a.stats + (size_t)((m0 >> 6) % Y2_STATS_REPL) * 2 * a.Cout;
                    atomicAdd(st + n, (double)s1);
                    atomicAdd(st + a.Cout + n, (double)s2);
                }
            }
        } else {
            float* col = a.y_pool + (size_t)(mrow >> 2) * a.ldp + a.poff + n;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = acc[i][j][4 * g + e] * sc + sh;
                        v[e] = u > 0.f ? u : u * a.slope;
                    }
                    col[(size_t)(i * 8 + 2 * g) * a.ldp] = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                }
        }
    }
}

// C layout of v_mfma_f32_32x32x2_f32: lane -> column n = lane&31; register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int MB, int NB, int WM, int WN, bool POOLORD>
__device__ __forceinline__ void conv_epilogue_general(const ConvArgs& a, f32x16 (&acc)[MB][NB], int m0, int n0, int wm, int wn, int l31, int half) {
    const bool do_full = a.y != nullptr;
    const bool do_pool = a.y_pool != nullptr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        const bool nok = n < a.Cout;
        const float sc = (a.scale != nullptr && nok) ? a.scale[n] : 1.f;
        const float sh = (a.shift != nullptr && nok) ? a.shift[n] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int mq = m0 + wm * WM + i * 32 + 8 * g + 4 * half;   // first row of this register quad
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = acc[i][j][4 * g + e];
                    s1 += z;
                    s2 += z * z;
                    float u = z * sc + sh;
                    if (!POOLORD && a.res != nullptr && nok && mq + e < a.M) u += a.res[(size_t)(mq + e) * a.ldr + n];   // residual branch (model/resnet.py:59,101)
                    v[e] = u > 0.f ? u : u * a.slope;
                }
                if (!nok || mq >= a.M) continue;
                if (do_full) {
                    if (POOLORD && a.out_mode == 0) {
                        // the quad is one 2x2 window (M % 4 == 0): decode its top-left pixel once
                        int pix, yy, xx;
                        decode_row<true>(a, mq, pix, yy, xx);
                        float* dst = a.y + (size_t)pix * a.ldy + a.coff + n;
                        dst[0] = v[0];
                        dst[a.ldy] = v[1];
                        dst[(size_t)a.W * a.ldy] = v[2];
                        dst[(size_t)(a.W + 1) * a.ldy] = v[3];
                    } else if (POOLORD || a.out_mode == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (mq + e >= a.M) break;
                            int pix, yy, xx;
                            decode_row<POOLORD>(a, mq + e, pix, yy, xx);
                            if (a.out_mode == 1) {
                                const int b = (int)y2_div((uint32_t)pix, a.d_hw);
                                const int opix = (b * (a.H >> 1) + (yy >> 1)) * (a.W >> 1) + (xx >> 1);
                                a.y[(size_t)opix * a.ldy + a.coff + ((yy & 1) * 2 + (xx & 1)) * a.Cout + n] = v[e];
                            } else {
                                a.y[(size_t)pix * a.ldy + a.coff + n] = v[e];
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (mq + e < a.M) a.y[(size_t)(mq + e) * a.ldy + a.coff + n] = v[e];
                    }
                }
                if (POOLORD && do_pool) {
                    // H, W even => M % 4 == 0 and the quad is one complete 2x2 window
                    const float pm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    a.y_pool[(size_t)(mq >> 2) * a.ldp + a.poff + n] = pm;
                }
            }
        }
        if (a.stats != nullptr) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (half == 0 && nok) {
                double* st = a.stats + (size_t)((m0 >> 6) % Y2_STATS_REPL) * 2 * a.Cout;   // replicated accumulators: see Y2_STATS_REPL
                atomicAdd(st + n, (double)s1);
                atomicAdd(st + a.Cout + n, (double)s2);
            }
        }
    }
}

template <int MB, int NB, int WM, int WN, bool POOLORD>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MB][NB], int m0, int n0, int wm, int wn, int l31, int half) {
    // a.fast_epi (host): no residual / statistics, out_mode 0, exactly one output of the kind the layout serves.  Per tile: fully inside.
    if (a.fast_epi && m0 + a.epi_bm <= a.M && n0 + a.epi_bn <= a.Cout) conv_epilogue_fast<MB, NB, WM, WN, POOLORD>(a, acc, m0, n0, wm, wn, l31, half);
    else conv_epilogue_general<MB, NB, WM, WN, POOLORD>(a, acc, m0, n0, wm, wn, l31, half);
}

template <int BM, int BN, int WAVES_M, int BK, bool POOLORD, bool VEC, int ABLATE = 0>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(const ConvArgs a) {
    constexpr int LDSS = BK + 4;                 // LDS row stride (floats): (BK+4)*4 B = odd multiple of 16 B -> conflict-free b128 reads
    constexpr int SLOTS = BK / 4;                // 16-B slots per row
    constexpr int RPP = NT / SLOTS;              // rows staged per pass of the 256 threads
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MB = WM / 32, NB = WN / 32;
    constexpr int AR = BM / RPP, BR = BN / RPP;  // rows each thread stages per slab
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    constexpr int STAGE = (BM + BN) * LDSS;      // floats per LDS buffer
    static_assert(MB >= 1 && NB >= 1, "wave tile must be a multiple of 32x32");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware remap applied separately to the whole tiles and to the K-slices, so that both classes of work are spread
    // evenly over the 8 XCDs (a joint remap would hand all the heavy whole tiles to the first XCDs)
    const bool is_split = (int)blockIdx.x >= a.full_tiles;
    int tile, part = 0;
    if (!is_split) {
        tile = y2_xcd_remap(blockIdx.x, min((int)gridDim.x, a.full_tiles));
    } else {
        const int r = y2_xcd_remap(blockIdx.x - a.full_tiles, gridDim.x - a.full_tiles);
        tile = a.full_tiles + r / a.ksplit;
        part = r % a.ksplit;
    }
    const int tile_n = tile % a.tiles_n;
    const int tile_m = tile / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging assignment: thread t -> 16-B slot (t&7) of rows (t>>3) + 32*i
    const int slot = t % SLOTS;
    const int srow = t / SLOTS;
    int a_pix[AR], a_y[AR], a_x[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RPP * i;
        if (m < a.M) {
            decode_row<POOLORD>(a, m, a_pix[i], a_y[i], a_x[i]);
        } else {
            a_pix[i] = 0; a_y[i] = -(1 << 20); a_x[i] = -(1 << 20);   // never inside the image
        }
    }
    const int ktot = a.taps * a.Cin;
    size_t b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + srow + RPP * i;
        b_ok[i] = n < a.Cout;
        b_off[i] = (size_t)(b_ok[i] ? n : 0) * ktot;
    }

    f32x4 ra[AR], rb[BR];

    auto load_slab = [&](int tap, int c0) {
        const int dy = (a.taps == 9) ? tap / 3 - 1 : 0;
        const int dx = (a.taps == 9) ? tap % 3 - 1 : 0;
        const int c = c0 + 4 * slot;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const bool ok = (unsigned)(a_y[i] + dy) < (unsigned)a.H && (unsigned)(a_x[i] + dx) < (unsigned)a.W;
            // Masked elements are fetched from a 16-B block of zeros instead of being selected after the load: the
            // loaded registers then have no VALU consumer before the ds_write, so the compiler's s_waitcnt vmcnt
            // lands AFTER the MFMAs of the current slab (the select form put it right behind the loads).
            if (VEC) {
                const bool okc = ok && c < a.Cin;
                const float* src = okc ? a.x + ((size_t)(a_pix[i] + dy * a.W + dx) * a.ldx + c) : a.zeros;
                ra[i] = *reinterpret_cast<const f32x4*>(src);
            } else {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool okc = ok && c + e < a.Cin;
                    const float* src = okc ? a.x + ((size_t)(a_pix[i] + dy * a.W + dx) * a.ldx + c + e) : a.zeros;
                    v[e] = *src;
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            if (VEC) {
                const bool okc = b_ok[i] && c < a.Cin;
                const float* src = okc ? a.w + (b_off[i] + (size_t)tap * a.Cin + c) : a.zeros;
                rb[i] = *reinterpret_cast<const f32x4*>(src);
            } else {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool okc = b_ok[i] && c + e < a.Cin;
                    const float* src = okc ? a.w + (b_off[i] + (size_t)tap * a.Cin + c + e) : a.zeros;
                    v[e] = *src;
                }
                rb[i] = v;
            }
        }
    };
    auto store_slab = [&](int buf) {
        float* sa = smem + buf * STAGE;
        float* sb = sa + BM * LDSS;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4*>(sa + (srow + RPP * i) * LDSS + 4 * slot) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4*>(sb + (srow + RPP * i) * LDSS + 4 * slot) = rb[i];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.taps * a.cchunks;
    int tap = 0, c0 = 0;
    load_slab(0, 0);
    store_slab(0);
    __syncthreads();

    // fragment read bases (floats): row * LDSS + 4*half, + 8*q per k-group
    const int fa = (wm * WM + l31) * LDSS + 4 * half;
    const int fb = BM * LDSS + (wn * WN + l31) * LDSS + 4 * half;

    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) {
            c0 += BK;
            if (c0 >= a.Cin) { c0 = 0; ++tap; }
            if (ABLATE != 1) load_slab(tap, c0);               // global loads in flight under the MFMAs below
        }
        const float* sbuf = smem + buf * STAGE;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 fa4[MB], fb4[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                fa4[i] = (ABLATE == 3) ? ra[i] : *reinterpret_cast<const f32x4*>(sbuf + fa + i * 32 * LDSS + 8 * q);
#pragma unroll
            for (int j = 0; j < NB; ++j)
                fb4[j] = (ABLATE == 3) ? rb[j] : *reinterpret_cast<const f32x4*>(sbuf + fb + j * 32 * LDSS + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa4[i][e], fb4[j][e], acc[i][j], 0, 0, 0);
        }
        if (ABLATE != 2) {
            if (more) store_slab(buf ^ 1);
            __syncthreads();
        }
    }

    conv_epilogue<MB, NB, WM, WN, POOLORD>(a, acc, m0, n0, wm, wn, l31, half);
}

// ------------------------------------------------------------------------------------------------ DMA kernel
// Same GEMM, tile and fragment scheme as conv_fwd_kernel, but the K-slabs go HBM/L2 -> LDS directly with
// `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging VGPRs, no ds_write pass, almost no VALU in the loop.
//   * zero padding / ragged edges for free: masked lanes get an out-of-range voffset and the buffer unit writes
//     zeros into LDS (verified on gfx950, tools/probes/dma_oob.hip); rows n >= Cout likewise.
//   * the DMA destination is lane-linear (wave base + lane*16 B), so LDS rows are unpadded 128-B rows (BK = 32
//     floats).  ds_read_b128 of 16 different rows would be 8-way bank conflicted; instead the 16-B chunk a lane
//     FETCHES is XOR-swizzled (physical slot p of row r holds logical chunk p ^ ((r>>1)&7)) and the fragment read
//     applies the same involution: the 16 lanes of every ds_read_b128 group hit 16 distinct 16-B slots.
//   * 2-deep ring: iteration s = {vmcnt(0); barrier; issue DMA of slab s+1; 16 ds_read_b128 + 64 MFMA on slab s}.
// Requirements (host checks, else the register-staged kernel runs): Cin, ldx multiples of 4, 16-B aligned bases,
// tensors < 2^31 bytes.
//
// PERSIST (round 4; tile ids 11 / 12 / 13 / 15 = the 128x128 / 128x64 / 64x64 / 64x128 tiles): a workgroup keeps its CU slot and walks
// tiles b, b + G, b + 2G, ... (G = resident workgroups of the launch); the first K slab of the NEXT tile is requested behind the barrier of
// the current tile's last slab, so its DMA latency runs under the last slab's MFMAs and the epilogue.  Aimed at the short-K layers (1x1
// convolutions: 2 ... 32 slabs per tile), where every tile of the one-tile-per-workgroup form starts with an exposed first fetch - and
// all workgroups of a CU reach that point together, because they were launched together and their tiles cost the same.
template <int BM, int BN, int WAVES_M, bool POOLORD, bool CTAIL, int STAGES = 2, bool GEN = false, int NTH = NT, bool PERSIST = false>
__global__ __launch_bounds__(NTH) void conv_fwd_dma_kernel(const ConvArgs a) {
    constexpr int BK = 32;
    static_assert(!PERSIST || (!GEN && STAGES == 2), "the persistent form covers the standard two-stage kernel");
    constexpr int WAVES_N = (NTH / 64) / WAVES_M;
    constexpr int RPP = NTH / 8;                  // rows staged per DMA pass (8 lanes per 128-B row)
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MB = WM / 32, NB = WN / 32;
    constexpr int AR = BM / RPP, BR = BN / RPP;     // DMA instructions per thread per slab (A rows, B rows)
    constexpr int STAGE = (BM + BN) * BK;          // floats per LDS stage
    constexpr unsigned OOB = 0x80000000u;          // beyond num_records of any supported tensor -> zeros
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware remap applied separately to the whole tiles and to the K-slices, so that both classes of work are spread
    // evenly over the 8 XCDs (a joint remap would hand all the heavy whole tiles to the first XCDs)
    const bool is_split = (int)blockIdx.x >= a.full_tiles;
    int tile, part = 0;
    if (!is_split) {
        tile = y2_xcd_remap(blockIdx.x, min((int)gridDim.x, a.full_tiles));
    } else {
        const int r = y2_xcd_remap(blockIdx.x - a.full_tiles, gridDim.x - a.full_tiles);
        tile = a.full_tiles + r / a.ksplit;
        part = r % a.ksplit;
    }
    const int flat_tile = tile;                   // group-major tile number (split-K scratch is indexed by it)
    int grp = 0;
    if (a.groups > 1) {
        const int tpg = a.tiles_m * a.tiles_n;
        grp = tile / tpg;
        tile -= grp * tpg;
    }
    const float* gx_ptr = a.x + (size_t)grp * a.gx;
    const float* gw_ptr = a.w + (size_t)grp * a.gw;
    const int tile_n = tile % a.tiles_n;
    const int tile_m = tile / a.tiles_n;
    int m0 = tile_m * BM, n0 = tile_n * BN;                    // (PERSIST: moved to the next tile by setup_tile)

    // ---- staging assignment: lane -> physical 16-B slot p = lane & 7 of row (t>>3) + 32*i; it fetches logical chunk p ^ swz(row)
    const int srow = t >> 3;                                   // 0..31 (rows 8*wave .. 8*wave+7)
    const int lchunk = (lane & 7) ^ ((srow >> 1) & 7);         // row + 32*i has the same swizzle
    unsigned a_base[AR];                                        // byte offset of (pixel row, tap (0,0) = up-left neighbour, chunk) or OOB
    unsigned a_mask[AR];                                        // bit tap = 1 when that tap is inside the image (GEN: unused)
    int a_yb[AR], a_xb[AR];                                     // GEN: input coordinates of tap (0,0) for this output pixel
    const int up_left = (a.taps == 9) ? (a.W + 1) : 0;
    const int ktot = GEN ? a.K : a.taps * a.Cin;
    unsigned b_base[BR];
    // PERSIST: the row / column bookkeeping of tile number v (virtual block index: the XCD remap sees the same b % 8 for every tile of a workgroup)
    auto setup_tile = [&](int v) {
        const int tl = y2_xcd_remap(v, a.full_tiles);
        const int tn = tl % a.tiles_n, tm = tl / a.tiles_n;
        m0 = tm * BM; n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + srow + RPP * i;
            unsigned mask = 0;
            int pix = 0;
            if (m < a.M) {
                int y, x;
                decode_row<POOLORD>(a, m, pix, y, x);
                if (a.taps == 9) {
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                        if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) mask |= 1u << tp;
                    }
                } else {
                    mask = 1u;
                }
            }
            a_mask[i] = mask;
            a_base[i] = (unsigned)(((long long)(pix - up_left) * a.ldx + 4 * lchunk) * 4);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + srow + RPP * i;
            b_base[i] = n < a.Cout ? (unsigned)(((size_t)n * ktot + 4 * lchunk) * 4) : OOB;
        }
    };
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RPP * i;
        if (GEN) {
            a_mask[i] = 0;
            if (m < a.M) {
                const int b = (int)y2_div((uint32_t)m, a.d_howo);
                const int idx = m - b * (a.Ho * a.Wo);
                const int yo = (int)y2_div((uint32_t)idx, a.d_wo);
                const int xo = idx - yo * a.Wo;
                a_yb[i] = yo * a.stride - a.pad;
                a_xb[i] = xo * a.stride - a.pad;
                a_base[i] = (unsigned)((long long)((b * a.H + a_yb[i]) * a.W + a_xb[i]) * a.ldx * 4);   // may wrap: only used when the tap is valid
                a_mask[i] = (unsigned)(b * a.H * a.W);     // transposed mode: pixel index of the image's first input pixel
            } else {
                a_yb[i] = -(1 << 20); a_xb[i] = -(1 << 20); a_base[i] = 0;
            }
        } else {
            unsigned mask = 0;
            int pix = 0;
            if (m < a.M) {
                int y, x;
                decode_row<POOLORD>(a, m, pix, y, x);
                if (a.taps == 9) {
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                        if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) mask |= 1u << tp;
                    }
                } else {
                    mask = 1u;
                }
            }
            a_mask[i] = mask;
            a_yb[i] = 0; a_xb[i] = 0;
            a_base[i] = (unsigned)(((long long)(pix - up_left) * a.ldx + 4 * lchunk) * 4);   // may wrap below 0: only used when the tap is valid
        }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + srow + RPP * i;
        b_base[i] = n < a.Cout ? (unsigned)(((size_t)n * ktot + (GEN ? 0 : 4 * lchunk)) * 4) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gx_ptr), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gw_ptr), 0, a.w_bytes, 0x00020000);

    auto issue_slab = [&](int ks_abs, int tap, int c0, int buf) {
        float* sa = smem + buf * STAGE + wave * (8 * BK);
        float* sb = sa + BM * BK;
        if (GEN) {
            // K = (ky, kx, ci) linear: this lane's 16-B chunk is 4 consecutive channels of ONE tap (Cin % 4 == 0)
            const int kq = ks_abs * BK + 4 * lchunk;
            const bool kok = kq < a.K;
            const int tp = (int)y2_div((uint32_t)kq, a.d_cin);
            const int c = kq - tp * a.Cin;
            const int ky = (int)y2_div((uint32_t)tp, a.d_kw);
            const int kx = tp - ky * a.KW;
            const unsigned toff = (unsigned)(((ky * a.W + kx) * a.ldx + c) * 4);
            if (a.tstride > 1) {
                // transposed convolution (data gradient of a stride-s conv): the input (dz) is read at (yu/s, xu/s) where
                // (yu, xu) = output pixel - pad + tap lies on the stride grid; other taps contribute nothing
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const int yu = a_yb[i] + ky, xu = a_xb[i] + kx;
                    const unsigned yq = y2_div((unsigned)yu, a.d_ts), xq = y2_div((unsigned)xu, a.d_ts);
                    const bool ok = kok && yu >= 0 && xu >= 0 && (int)yq * a.tstride == yu && (int)xq * a.tstride == xu && yq < (unsigned)a.H && xq < (unsigned)a.W;
                    const unsigned voff = ok ? (unsigned)(((size_t)(a_mask[i] + yq * a.W + xq) * a.ldx + c) * 4) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * RPP * BK), 16, (int)voff, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const bool ok = kok && (unsigned)(a_yb[i] + ky) < (unsigned)a.H && (unsigned)(a_xb[i] + kx) < (unsigned)a.W;
                    const unsigned voff = ok ? a_base[i] + toff : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * RPP * BK), 16, (int)voff, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const unsigned voff = (kok && b_base[i] != OOB) ? b_base[i] + (unsigned)kq * 4u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sb + i * RPP * BK), 16, (int)voff, 0, 0, 0);
            }
            return;
        }
        // A: voffset = a_base + ((ky*W + kx)*ldx + c0)*4   (ky, kx in 0..2 relative to the up-left neighbour)
        const int ky = (a.taps == 9) ? tap / 3 : 0, kx = (a.taps == 9) ? tap % 3 : 0;
        const unsigned toff = (unsigned)(((ky * a.W + kx) * a.ldx + c0) * 4);
        const unsigned tbit = 1u << tap;
        const bool cok = !CTAIL || (c0 + 4 * lchunk) < a.Cin;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const bool ok = (a_mask[i] & tbit) != 0 && cok;
            const unsigned voff = ok ? a_base[i] + toff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * RPP * BK), 16, (int)voff, 0, 0, 0);
        }
        const unsigned woff = (unsigned)((tap * a.Cin + c0) * 4);
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const unsigned voff = (CTAIL && !cok) ? OOB : b_base[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sb + i * RPP * BK), 16, (int)voff, (int)woff, 0, 0);
        }
    };

    // one DMA instruction of a standard slab (piece j < AR: A rows, else B rows): the pieces of slab s+1 can be issued one by one
    // BETWEEN the MFMAs of slab s (Y2_CONV_SPREAD) instead of all in front of them
    auto issue_piece = [&](int tap, int c0, int buf, int j) {
        float* sa = smem + buf * STAGE + wave * (8 * BK);
        float* sb = sa + BM * BK;
        const int ky = (a.taps == 9) ? tap / 3 : 0, kx = (a.taps == 9) ? tap % 3 : 0;
        const bool cok = !CTAIL || (c0 + 4 * lchunk) < a.Cin;
        if (j < AR) {
            const unsigned toff = (unsigned)(((ky * a.W + kx) * a.ldx + c0) * 4);
            const bool ok = (a_mask[j] & (1u << tap)) != 0 && cok;
            const unsigned voff = ok ? a_base[j] + toff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + j * RPP * BK), 16, (int)voff, 0, 0, 0);
        } else {
            const int i = j - AR;
            const unsigned woff = (unsigned)((tap * a.Cin + c0) * 4);
            const unsigned voff = (CTAIL && !cok) ? OOB : b_base[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sb + i * RPP * BK), 16, (int)voff, (int)woff, 0, 0);
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (floats): physical slot of logical chunk 2q+half in row l31 (+32*block)
    const int sw = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = l31 * BK + (((2 * q + half) ^ sw) << 2);
    const int fa = wm * WM * BK;
    const int fb = BM * BK + wn * WN * BK;

    auto compute_slab = [&](int buf) {
        const float* sbuf = smem + buf * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa4[MB], fb4[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) fa4[i] = *reinterpret_cast<const f32x4*>(sbuf + fa + i * 32 * BK + foff[q]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb4[j] = *reinterpret_cast<const f32x4*>(sbuf + fb + j * 32 * BK + foff[q]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa4[i][e], fb4[j][e], acc[i][j], 0, 0, 0);
        }
    };

    // compute_slab with the DMA pieces of the NEXT slab spread over the first half of its MFMAs (one piece per MFMA group of
    // TOTAL/2/NPIECES instructions; the second half of the slab is left for the pieces to land before the next vmcnt(0))
    auto compute_slab_spread = [&](int buf, int ntap, int nc0, int nbuf) {
        const float* sbuf = smem + buf * STAGE;
        constexpr int TOTAL = 16 * MB * NB, NPIECES = AR + BR;
        constexpr int EVERY = (TOTAL / 2) / NPIECES > 0 ? (TOTAL / 2) / NPIECES : 1;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa4[MB], fb4[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) fa4[i] = *reinterpret_cast<const f32x4*>(sbuf + fa + i * 32 * BK + foff[q]);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb4[j] = *reinterpret_cast<const f32x4*>(sbuf + fb + j * 32 * BK + foff[q]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa4[i][e], fb4[j][e], acc[i][j], 0, 0, 0);
                        if (cnt % EVERY == 0 && cnt / EVERY < NPIECES) {
                            issue_piece(ntap, nc0, nbuf, cnt / EVERY);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ++cnt;
                    }
        }
        // tiles with fewer MFMAs than pieces in half a slab: the rest goes out behind the last MFMA
#pragma unroll
        for (int j = (TOTAL + EVERY - 1) / EVERY; j < NPIECES; ++j) issue_piece(ntap, nc0, nbuf, j);
    };

    if constexpr (PERSIST) {
        // ---- tiles b, b + G, ...: slab s of a tile lives in LDS buffer (pb + s) & 1; the barrier in front of a tile's LAST slab has seen every
        // wave finish slab nk-2, whose buffer is the one the next tile's first slab is fetched into
        const int nkp = a.taps * a.cchunks;
        int v = blockIdx.x, pb = 0;
        setup_tile(v);                  // (the remap must be the one of the whole tile list for EVERY tile of the walk, the first included)
        issue_slab(0, 0, 0, 0);
        for (;;) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            int ptap = 0, pc0 = 0;
            for (int ks = 0; ks < nkp - 1; ++ks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                pc0 += BK;
                if (pc0 >= a.Cin) { pc0 = 0; ++ptap; }
                compute_slab_spread((pb + ks) & 1, ptap, pc0, (pb + ks + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int cm0 = m0, cn0 = n0;
            const int vn = v + (int)gridDim.x;
            const bool more = vn < a.full_tiles;
            if (more) {
                setup_tile(vn);
                issue_slab(0, 0, 0, (pb + nkp) & 1);
            }
            compute_slab((pb + nkp - 1) & 1);
            conv_epilogue<MB, NB, WM, WN, POOLORD>(a, acc, cm0, cn0, wm, wn, l31, half);
            if (!more) return;
            v = vn;
            pb = (pb + nkp) & 1;
        }
    }
    const int nk_all = GEN ? (a.K + BK - 1) / BK : a.taps * a.cchunks;
    int ks0 = 0, ks1 = nk_all;
    if (is_split) {
        ks0 = (int)((long long)nk_all * part / a.ksplit);
        ks1 = (int)((long long)nk_all * (part + 1) / a.ksplit);
    }
    const int nk = ks1 - ks0;
    int tap = GEN ? 0 : ks0 / a.cchunks, c0 = GEN ? 0 : (ks0 % a.cchunks) * BK;
    if (nk > 0) {
        if (STAGES == 2) {
            issue_slab(ks0, tap, c0, 0);
            for (int ks = 0; ks < nk - 1; ++ks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of slab ks has landed
                __syncthreads();                                     // ... everybody's has, and nobody still reads the other buffer
                c0 += BK;
                if (c0 >= a.Cin) { c0 = 0; ++tap; }
                if (Y2_CONV_SPREAD && !GEN) {
                    compute_slab_spread(ks & 1, tap, c0, (ks + 1) & 1);
                } else {
                    issue_slab(ks0 + ks + 1, tap, c0, (ks + 1) & 1);
                    compute_slab(ks & 1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute_slab((nk - 1) & 1);
        } else {
            // 3-deep ring: the DMA of slab s+2 is issued before slab s is consumed, so one slab stays in flight ACROSS the
            // barrier (counted vmcnt + raw s_barrier: __syncthreads() would drain the DMA queue, cdna guide "glds span").
            issue_slab(ks0, tap, c0, 0);
            if (nk > 1) {
                c0 += BK;
                if (c0 >= a.Cin) { c0 = 0; ++tap; }
                issue_slab(ks0 + 1, tap, c0, 1);
            }
            int cur = 0, nxt = 2;
            for (int ks = 0; ks < nk; ++ks) {
                if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR + BR) : "memory");   // slab ks landed, ks+1 may be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();        // everyone's part of slab ks is in LDS; everyone finished reading slab ks-1
                if (ks + 2 < nk) {
                    c0 += BK;
                    if (c0 >= a.Cin) { c0 = 0; ++tap; }
                    issue_slab(ks0 + ks + 2, tap, c0, nxt);         // overwrites the buffer of slab ks-1
                }
                compute_slab(cur);
                cur = cur == 2 ? 0 : cur + 1;
                nxt = nxt == 2 ? 0 : nxt + 1;
            }
        }
    }

    if (is_split) {
        // raw accumulators -> partial[(tile - full_tiles) * ksplit + part][MB][NB][4][256 threads][4]: 16-B coalesced stores;
        // conv_splitk_fixup_kernel (same thread geometry) adds the parts and runs the ordinary epilogue
        float* dst = a.partial + ((size_t)(flat_tile - a.full_tiles) * a.ksplit + part) * (BM * BN);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dst + (((i * NB + j) * 4 + g) * NTH + t) * 4) = v;
                }
        return;
    }
    if (a.groups > 1) {
        ConvArgs e = a;
        e.y = a.y + (size_t)grp * a.gy;
        conv_epilogue<MB, NB, WM, WN, POOLORD>(e, acc, m0, n0, wm, wn, l31, half);
        return;
    }
    conv_epilogue<MB, NB, WM, WN, POOLORD>(a, acc, m0, n0, wm, wn, l31, half);
}

// Adds the K-parts of the split remainder tiles and applies the ordinary epilogue (one workgroup per split tile, same
// thread -> accumulator mapping as conv_fwd_dma_kernel).
template <int BM, int BN, int WAVES_M, bool POOLORD, int NTHR = NT>
__global__ __launch_bounds__(NTHR) void conv_splitk_fixup_kernel(const ConvArgs a) {
    constexpr int WAVES_N = (NTHR / 64) / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MB = WM / 32, NB = WN / 32;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;
    int tile = a.full_tiles + blockIdx.x;
    int grp = 0;
    if (a.groups > 1) {
        const int tpg = a.tiles_m * a.tiles_n;
        grp = tile / tpg;
        tile -= grp * tpg;
    }
    const int tile_n = tile % a.tiles_n;
    const int tile_m = tile / a.tiles_n;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                for (int p = 0; p < a.ksplit; ++p) {
                    const float* src = a.partial + ((size_t)blockIdx.x * a.ksplit + p) * (BM * BN);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((i * NB + j) * 4 + g) * NTHR + t) * 4);
                    sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
                }
                acc[i][j][4 * g] = sum[0]; acc[i][j][4 * g + 1] = sum[1]; acc[i][j][4 * g + 2] = sum[2]; acc[i][j][4 * g + 3] = sum[3];
            }
    if (a.groups > 1) {
        ConvArgs e = a;
        e.y = a.y + (size_t)grp * a.gy;
        conv_epilogue<MB, NB, WM, WN, POOLORD>(e, acc, tile_m * BM, tile_n * BN, wm, wn, l31, half);
        return;
    }
    conv_epilogue<MB, NB, WM, WN, POOLORD>(a, acc, tile_m * BM, tile_n * BN, wm, wn, l31, half);
}

template <int BM, int BN, int WAVES_M, int BK, bool POOLORD, bool VEC, int ABLATE = 0>
int launch(const ConvArgs& a0, hipStream_t stream) {
    ConvArgs a = a0;
    a.tiles_m = y2_cdiv(a.M, BM);
    a.epi_bm = BM; a.epi_bn = BN;
    a.tiles_n = y2_cdiv(a.Cout, BN);
    size_t lds = 2u * (BM + BN) * (BK + 4) * sizeof(float);
    if (const char* pad = getenv("Y2_CONV_LDS_MIN")) { const size_t m = (size_t)atol(pad); if (lds < m) lds = m; }   // occupancy experiments only
    a.cchunks = y2_cdiv(a.Cin, BK);
    auto kern = conv_fwd_kernel<BM, BN, WAVES_M, BK, POOLORD, VEC, ABLATE>;
    static Y2LdsAttr attr_set;
    if (const int rc = attr_set.ensure(reinterpret_cast<const void*>(kern))) return rc;
    const long long grid = (long long)a.tiles_m * a.tiles_n;
    if (grid <= 0 || grid > 0x7fffffffLL) return Y2_EINVAL;
    Y2_LAUNCH("conv_fwd_kernel", 2.0 * (double)a.M * a.Cout * (a.K > 0 ? a.K : a.taps * a.Cin) * (a.groups > 1 ? a.groups : 1), kern, dim3((unsigned)grid), dim3(NT), lds, stream, a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// Remainder split: T tiles on P CUs run floor(T/P) full rounds; the last T mod P tiles would occupy only part of the chip
// for a whole tile time.  They are cut into s K-slices each (s chosen to minimise ceil(rem*s/P)/s) and a tiny fixup kernel
// adds the slices.  Needs caller workspace; without it (or when nothing is gained) tiles run whole.
inline void plan_split(long long tiles, int nk, long long tile_elems, size_t ws_bytes, int& full_tiles, int& ksplit, int P = Y2_NUM_CU) {
    full_tiles = (int)tiles; ksplit = 1;
    const long long rem = tiles % P;
    if (rem == 0 || nk < 8) return;
    double best = 1.0; int bs = 1;
    // (round 5, measured and not kept: aiming the split of launches smaller than one round - single images - at 512 / 768 / 1024 slots with up
    // to 32 K-slices, so that several workgroups per CU cover each other's HBM round trips: batch-1 detect 0.621 -> 0.621 / 0.621 / 0.623 ms,
    // the same plan chosen; these launches are bound by the filter bytes they stream)
    for (int s = 2; s <= 16 && nk / s >= 4; ++s) {
        const double t = (double)y2_cdiv(rem * s, P) / s * 1.02 + 0.01;   // small penalty for the extra prologue/epilogue + fixup
        if (t < best - 1e-9) { best = t; bs = s; }
    }
    if (bs == 1) return;
    const double before = (double)y2_cdiv(tiles, P), after = (double)(tiles / P) + best;
    if (after > before * 0.97) return;                                      // < 3 % gain: not worth a second launch
    if ((size_t)rem * bs * tile_elems * sizeof(float) > ws_bytes) return;
    full_tiles = (int)(tiles - rem); ksplit = bs;
}

template <int BM, int BN, int WAVES_M, bool POOLORD, bool GEN = false, int NTH = NT, bool PERSIST = false>
int launch_dma(const ConvArgs& a0, hipStream_t stream, float* ws, size_t ws_bytes, size_t* ws_need) {
    ConvArgs a = a0;
    a.tiles_m = y2_cdiv(a.M, BM);
    a.epi_bm = BM; a.epi_bn = BN;
    a.tiles_n = y2_cdiv(a.Cout, BN);
    a.cchunks = y2_cdiv(a.Cin, 32);
    size_t lds = 2u * (BM + BN) * 32 * sizeof(float);
    if (const char* pad = getenv("Y2_CONV_LDS_MIN")) { const size_t m = (size_t)atol(pad); if (lds < m) lds = m; }   // occupancy experiments only
    const bool ctail = !GEN && (a.Cin % 32) != 0;
    const int nk_all = GEN ? y2_cdiv(a.K, 32) : a.taps * a.cchunks;
    const long long tiles = (long long)a.tiles_m * a.tiles_n * (a.groups > 1 ? a.groups : 1);
    if (tiles <= 0 || tiles > 0x7fffffffLL) return Y2_EINVAL;
    if (ws_need != nullptr) {   // workspace query: the largest split this layer could use
        int ft, ks;
        plan_split(tiles, nk_all, (long long)BM * BN, (size_t)-1, ft, ks);
        *ws_need = (size_t)(tiles - ft) * ks * BM * BN * sizeof(float);
        return Y2_OK;
    }
    if (PERSIST) {
        // persistent workgroups: as many as the chip holds at once (never split: the tile loop evens the rounds out)
        if (GEN || ctail || a.groups > 1) return Y2_ENOSUP;
        if (ws_need != nullptr) { *ws_need = 0; return Y2_OK; }
        a.full_tiles = (int)tiles; a.ksplit = 1; a.partial = nullptr;
        auto kern = conv_fwd_dma_kernel<BM, BN, WAVES_M, POOLORD, false, 2, false, NTH, PERSIST>;
        static Y2LdsAttr attr_p;
        if (const int rc_ = attr_p.ensure(reinterpret_cast<const void*>(kern))) return rc_;
        static int resident = 0;                           // workgroups one CU holds (registers, LDS), asked once per instantiation
        if (resident == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NTH, lds) != hipSuccess || nb < 1) nb = 1;
            resident = nb > 4 ? 4 : nb;
        }
        const long long slots = (long long)Y2_NUM_CU * resident;
        // (measured and not kept: ceil(tiles / rounds) workgroups, so that every workgroup walks the same number of tiles - fewer resident
        // workgroups cost more than the even split gains: -4 ... -7 % on the 1x1 layers where `slots` workgroups had gained +2 ... +16 %)
        const long long g = tiles < slots ? tiles : slots;
        Y2_LAUNCH("conv_fwd_dma_kernel[persistent]", 2.0 * (double)a.M * a.Cout * a.taps * a.Cin, kern, dim3((unsigned)g), dim3(NTH), lds, stream, a);
        Y2_LAUNCH_CHECK();
        return Y2_OK;
    }
    plan_split(tiles, nk_all, (long long)BM * BN, ws != nullptr ? ws_bytes : 0, a.full_tiles, a.ksplit);
    a.partial = ws;
    const long long grid = a.full_tiles + (tiles - a.full_tiles) * a.ksplit;
    static Y2LdsAttr attr_set[4];
    static int stages3 = -1;
    if (stages3 < 0) { const char* e = getenv("Y2_CONV_STAGES"); stages3 = (e != nullptr && atoi(e) == 3) ? 1 : 0; }
#define Y2_DMA_LAUNCH(KERN, SLOT, LDSB)                                                                                     \
    do {                                                                                                                    \
        auto kern = KERN;                                                                                                   \
        if (const int rc_ = attr_set[SLOT].ensure(reinterpret_cast<const void*>(kern))) return rc_;                        \
        Y2_LAUNCH(a.groups > 1 ? "conv_fwd_dma_kernel[grouped]" : "conv_fwd_dma_kernel", 2.0 * (double)a.M * a.Cout * (a.K > 0 ? a.K : a.taps * a.Cin) * (a.groups > 1 ? a.groups : 1), kern, dim3((unsigned)grid), dim3(NTH), LDSB, stream, a);                                         \
    } while (0)
    if (GEN) Y2_DMA_LAUNCH((conv_fwd_dma_kernel<BM, BN, WAVES_M, false, false, 2, true, NTH>), 3, lds);
    else if (stages3 && !ctail) Y2_DMA_LAUNCH((conv_fwd_dma_kernel<BM, BN, WAVES_M, POOLORD, false, 3, false, NTH>), 2, lds / 2 * 3);
    else if (ctail) Y2_DMA_LAUNCH((conv_fwd_dma_kernel<BM, BN, WAVES_M, POOLORD, true, 2, false, NTH>), 1, lds);
    else Y2_DMA_LAUNCH((conv_fwd_dma_kernel<BM, BN, WAVES_M, POOLORD, false, 2, false, NTH>), 0, lds);
#undef Y2_DMA_LAUNCH
    if (a.ksplit > 1)
        Y2_LAUNCH("conv_splitk_fixup_kernel", 0.0, (conv_splitk_fixup_kernel<BM, BN, WAVES_M, (POOLORD && !GEN), NTH>), dim3((unsigned)(tiles - a.full_tiles)), dim3(NTH), 0, stream, a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// ------------------------------------------------------------------------------------------------ wave-private kernel
// Barrier-free variant for the standard convolutions (3x3 / 1x1, stride 1, Cin % 16 == 0): ONE wave owns a 64 x 64 output tile
// and its own LDS ring, so there is no workgroup barrier and no coupling between the four SIMDs of a CU (in the workgroup
// kernel a wave that shares its SIMD with a busier neighbour delays the other three waves at every slab barrier).
//   * workgroup = 64 threads; slab = 16 floats of K (64-B rows): A 64 rows + B 64 rows = 8 KB per stage, 4 stages = 32 KB,
//     four waves per CU = one per SIMD; the ring keeps up to three slabs (6144 MFMA cycles) in flight with counted vmcnt;
//   * LDS-DMA with the XOR swizzle of the 64-B-row geometry: physical 16-B slot p of row r holds logical chunk p ^ ((r>>2)&3)
//     (4 rows share one 256-B bank row), fragment reads apply the same involution -> conflict-free ds_read_b128;
//   * per slab: 8 DMA instructions, 8 ds_read_b128, 32 MFMAs (2048 cycles) and no s_barrier at all.
template <bool POOLORD, int STAGES = 4, int BK = 16>
__global__ __launch_bounds__(64) void conv_fwd_wave_kernel(const ConvArgs a) {
    constexpr int TM = 64, TN = 64;
    constexpr int CH = BK / 4;                       // 16-B chunks per row
    constexpr int RP = 64 / CH;                      // rows per DMA instruction
    constexpr int NP = 64 / RP;                      // DMA instructions per operand per slab
    constexpr int QG = BK / 8;                       // k-groups (one ds_read_b128 per block each) per slab
    constexpr int STAGE = (TM + TN) * BK;            // floats per stage
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int lane = threadIdx.x;
    const int l31 = lane & 31, half = lane >> 5;
    const bool is_split = (int)blockIdx.x >= a.full_tiles;
    int tile, part = 0;
    if (!is_split) {
        tile = y2_xcd_remap(blockIdx.x, min((int)gridDim.x, a.full_tiles));
    } else {
        const int r = y2_xcd_remap(blockIdx.x - a.full_tiles, gridDim.x - a.full_tiles);
        tile = a.full_tiles + r / a.ksplit;
        part = r % a.ksplit;
    }
    const int tile_n = tile % a.tiles_n;
    const int tile_m = tile / a.tiles_n;
    const int m0 = tile_m * TM, n0 = tile_n * TN;

    // staging: lane -> physical slot p = lane % CH of row lane / CH + RP*i; it fetches logical chunk p ^ swz(row)
    // swz: rows sharing one 256-B bank row get different XOR masks (BK = 16: 4 rows per bank row; BK = 32: 2 rows)
    const int srow = lane / CH;
    const int lchunk = (lane % CH) ^ (BK == 32 ? ((srow >> 1) & 7) : ((srow >> 2) & 3));
    unsigned a_base[NP], a_mask[NP], b_base[NP];
    const int up_left = (a.taps == 9) ? (a.W + 1) : 0;
    const int ktot = a.taps * a.Cin;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int m = m0 + srow + RP * i;
        unsigned mask = 0;
        int pix = 0;
        if (m < a.M) {
            int y, x;
            decode_row<POOLORD>(a, m, pix, y, x);
            if (a.taps == 9) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                    if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) mask |= 1u << tp;
                }
            } else {
                mask = 1u;
            }
        }
        a_mask[i] = mask;
        a_base[i] = (unsigned)(((long long)(pix - up_left) * a.ldx + 4 * lchunk) * 4);
        const int n = n0 + srow + RP * i;
        b_base[i] = n < a.Cout ? (unsigned)(((size_t)n * ktot + 4 * lchunk) * 4) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.w_bytes, 0x00020000);

    auto issue_slab = [&](int tap, int c0, int stage) {
        const int ky = (a.taps == 9) ? tap / 3 : 0, kx = (a.taps == 9) ? tap % 3 : 0;
        const unsigned toff = (unsigned)(((ky * a.W + kx) * a.ldx + c0) * 4);
        const unsigned tbit = 1u << tap;
        float* sa = smem + stage * STAGE;
        float* sb = sa + TM * BK;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned voff = (a_mask[i] & tbit) ? a_base[i] + toff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sa + i * RP * BK), 16, (int)voff, 0, 0, 0);
        }
        const unsigned woff = (unsigned)((tap * a.Cin + c0) * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sb + i * RP * BK), 16, (int)b_base[i], (int)woff, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment offsets (floats): row l31 (+32 per block), logical chunk 2q + half -> physical chunk (same involution)
    const int sw = BK == 32 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    int foff[QG];
#pragma unroll
    for (int q = 0; q < QG; ++q) foff[q] = l31 * BK + (((2 * q + half) ^ sw) << 2);

    // software-pipelined fragment reads: while the 16 MFMAs of one k-group run, the fragments of the next group (possibly of
    // the next slab, after its counted vmcnt) are already being read from LDS
    auto load_frags = [&](int stage, int q, f32x4 (&fa4)[2], f32x4 (&fb4)[2]) {
        const float* sbuf = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa4[i] = *reinterpret_cast<const f32x4*>(sbuf + i * 32 * BK + foff[q]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb4[j] = *reinterpret_cast<const f32x4*>(sbuf + TM * BK + j * 32 * BK + foff[q]);
    };
    auto mfma_group = [&](const f32x4 (&fa4)[2], const f32x4 (&fb4)[2]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa4[i][e], fb4[j][e], acc[i][j], 0, 0, 0);
    };
    constexpr int PER = 2 * NP;                       // DMA instructions per slab
    auto wait_slab = [&](int ks, int nk) {             // slab ks landed; up to min(STAGES-2, nk-1-ks) younger slabs stay in flight
        const int ahead = min(STAGES - 2, nk - 1 - ks);
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    const int cch = a.Cin / BK;                      // chunks per tap
    const int nk_all = a.taps * cch;
    int ks0 = 0, ks1 = nk_all;
    if (is_split) {
        ks0 = (int)((long long)nk_all * part / a.ksplit);
        ks1 = (int)((long long)nk_all * (part + 1) / a.ksplit);
    }
    const int nk = ks1 - ks0;
    int tap = ks0 / cch, c0 = (ks0 % cch) * BK;      // position of the NEXT slab to issue
    auto advance = [&]() { c0 += BK; if (c0 >= a.Cin) { c0 = 0; ++tap; } };
    int issued = 0;
    for (; issued < min(STAGES - 1, nk); ++issued) { issue_slab(tap, c0, issued); advance(); }
    f32x4 xa[2], xb[2], ya[2], yb[2];
    if (nk > 0) {
        wait_slab(0, nk);
        load_frags(0, 0, xa, xb);
    }
    int cur = 0, nxt = STAGES - 1;
    int ks = 0;
    // one slab: QG k-groups alternating between the fragment sets X (even groups) and Y (odd groups)
    auto slab_body = [&](bool steady) {
        const int nc = cur == STAGES - 1 ? 0 : cur + 1;
#pragma unroll
        for (int q = 0; q < QG; q += 2) {
            load_frags(cur, q + 1, ya, yb);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(xa, xb);
            if (q + 2 < QG) {
                load_frags(cur, q + 2, xa, xb);
            } else if (steady) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (STAGES - 2)) : "memory");   // slab ks+1 landed; the younger ones stay in flight
                load_frags(nc, 0, xa, xb);
            } else if (ks + 1 < nk) {
                wait_slab(ks + 1, nk);
                load_frags(nc, 0, xa, xb);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(ya, yb);
        }
        cur = nc;
        nxt = nxt == STAGES - 1 ? 0 : nxt + 1;
    };
    // steady state (branch-free body: exact lgkmcnt / vmcnt counts): slab ks + STAGES - 1 exists
    for (; ks + (STAGES - 1) < nk; ++ks) {
        issue_slab(tap, c0, nxt); advance(); ++issued;     // into the stage of slab ks-1 (fully read)
        slab_body(true);
    }
    // drain
    for (; ks < nk; ++ks) {
        if (issued < nk) { issue_slab(tap, c0, nxt); advance(); ++issued; }
        slab_body(false);
    }

    if (is_split) {
        float* dst = a.partial + ((size_t)(tile - a.full_tiles) * a.ksplit + part) * (TM * TN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(dst + (((i * 2 + j) * 4 + g) * 64 + lane) * 4) = v;
                }
        return;
    }
    conv_epilogue<2, 2, 64, 64, POOLORD>(a, acc, m0, n0, 0, 0, l31, half);
}

template <bool POOLORD>
int launch_wave(const ConvArgs& a0, hipStream_t stream, float* ws, size_t ws_bytes, size_t* ws_need) {
    ConvArgs a = a0;
    a.tiles_m = y2_cdiv(a.M, 64);
    a.tiles_n = y2_cdiv(a.Cout, 64);
    const char* bke = getenv("Y2_WAVE_BK");
    const int nk_all = a.taps * (a.Cin / ((bke && atoi(bke) == 32) ? 32 : 16));
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return Y2_EINVAL;
    const int P = 4 * Y2_NUM_CU;     // one tile stream per SIMD
    if (ws_need != nullptr) {
        int ft, ks;
        plan_split(tiles, nk_all, 64 * 64, (size_t)-1, ft, ks, P);
        *ws_need = (size_t)(tiles - ft) * ks * 64 * 64 * sizeof(float);
        return Y2_OK;
    }
    plan_split(tiles, nk_all, 64 * 64, ws != nullptr ? ws_bytes : 0, a.full_tiles, a.ksplit, P);
    a.partial = ws;
    const long long grid = a.full_tiles + (tiles - a.full_tiles) * a.ksplit;
    static int stages = -1, bk = 16;
    if (stages < 0) { const char* e = getenv("Y2_WAVE_STAGES"); stages = e ? atoi(e) : 4; const char* b = getenv("Y2_WAVE_BK"); bk = (b && atoi(b) == 32) ? 32 : 16; }
    if (bk == 32 && (a.Cin % 32) != 0) return Y2_ENOSUP;
    const size_t lds = (size_t)stages * (64 + 64) * bk * sizeof(float);
    static bool attr = false;
#define Y2_WAVE(ST, BKV)                                                                                                          \
    do {                                                                                                                          \
        auto kern = conv_fwd_wave_kernel<POOLORD, ST, BKV>;                                                                       \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        Y2_LAUNCH("conv_fwd_wave_kernel", 2.0 * (double)a.M * a.Cout * (a.K > 0 ? a.K : a.taps * a.Cin) * (a.groups > 1 ? a.groups : 1), kern, dim3((unsigned)grid), dim3(64), lds, stream, a);                                                 \
    } while (0)
    (void)attr;
    if (bk == 32) { if (stages == 2) Y2_WAVE(2, 32); else if (stages == 3) Y2_WAVE(3, 32); else Y2_WAVE(4, 32); }
    else { if (stages == 2) Y2_WAVE(2, 16); else if (stages == 3) Y2_WAVE(3, 16); else Y2_WAVE(4, 16); }
#undef Y2_WAVE
    if (a.ksplit > 1)
        Y2_LAUNCH("conv_splitk_fixup_kernel", 0.0, (conv_splitk_fixup_kernel<64, 64, 1, POOLORD, 64>), dim3((unsigned)(tiles - a.full_tiles)), dim3(64), 0, stream, a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

template <bool POOLORD, bool GEN = false>
int dispatch_dma(const ConvArgs& a, int tile, hipStream_t s, float* ws, size_t ws_bytes, size_t* ws_need) {
    switch (tile) {
        case 1: return launch_dma<128, 128, 2, POOLORD, GEN>(a, s, ws, ws_bytes, ws_need);
        case 2: return launch_dma<128, 64, 2, POOLORD, GEN>(a, s, ws, ws_bytes, ws_need);
        case 3: return launch_dma<64, 64, 2, POOLORD, GEN>(a, s, ws, ws_bytes, ws_need);
        case 5: return launch_dma<64, 128, 2, POOLORD, GEN>(a, s, ws, ws_bytes, ws_need);
        case 6: return launch_dma<128, 32, 4, POOLORD, GEN>(a, s, ws, ws_bytes, ws_need);
        // persistent workgroups with the next tile's first slab fetched under the current tile's last one (un-pooled standard convolutions)
        case 11: if (!GEN && !POOLORD) return launch_dma<128, 128, 2, false, false, NT, true>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;
        case 12: if (!GEN && !POOLORD) return launch_dma<128, 64, 2, false, false, NT, true>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;
        case 13: if (!GEN && !POOLORD) return launch_dma<64, 64, 2, false, false, NT, true>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;
        case 15: if (!GEN && !POOLORD) return launch_dma<64, 128, 2, false, false, NT, true>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;
        case 8: if (!GEN) return launch_dma<256, 128, 4, POOLORD, false, 512>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;   // 8 waves: 2 per SIMD, 96 KB LDS, 6 B/clk operand DMA
        case 9: if (!GEN) return launch_dma<128, 256, 2, POOLORD, false, 512>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;
        case 7: if (!GEN) return launch_wave<POOLORD>(a, s, ws, ws_bytes, ws_need); return Y2_ENOSUP;   // barrier-free wave-private 64x64 tiles   // narrow outputs (Cout <= 32: dgrad into the first layers)
        default: return Y2_ENOSUP;
    }
}

template <bool POOLORD, bool VEC>
int dispatch_tile(const ConvArgs& a, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch<128, 128, 2, 32, POOLORD, VEC>(a, s);
        case 2: return launch<128, 64, 2, 32, POOLORD, VEC>(a, s);
        case 3: return launch<64, 64, 2, 32, POOLORD, VEC>(a, s);
        case 4: return launch<256, 64, 4, 32, POOLORD, VEC>(a, s);
        case 5: return launch<64, 128, 2, 32, POOLORD, VEC>(a, s);
        case 11: return launch<128, 128, 2, 16, POOLORD, VEC>(a, s);
        case 91: if (!POOLORD && VEC) return launch<128, 128, 2, 32, false, true, 1>(a, s); return Y2_ENOSUP;   // timing ablations (wrong results)
        case 92: if (!POOLORD && VEC) return launch<128, 128, 2, 32, false, true, 2>(a, s); return Y2_ENOSUP;
        case 93: if (!POOLORD && VEC) return launch<128, 128, 2, 32, false, true, 3>(a, s); return Y2_ENOSUP;
        case 13: return launch<64, 64, 2, 16, POOLORD, VEC>(a, s);
        case 21: return launch<128, 128, 2, 64, POOLORD, VEC>(a, s);
        case 23: return launch<64, 64, 2, 64, POOLORD, VEC>(a, s);
        default: return Y2_ENOSUP;
    }
}

// Rounds of whole tiles per CU, including the split-K remainder scheme of launch_dma (nk = K slabs per tile).
inline double effective_rounds(long long tiles, int nk) {
    const int P = Y2_NUM_CU;
    const long long rem = tiles % P;
    double frac = rem ? 1.0 : 0.0;
    if (rem && nk >= 8)
        for (int s = 2; s <= 16 && nk / s >= 4; ++s) {
            const double t = (double)y2_cdiv(rem * s, P) / s * 1.02 + 0.01;
            if (t < frac) frac = t;
        }
    return (double)(tiles / P) + frac;
}

// Pick the tile that minimises (tile rounds per CU) x (tile area) / (measured main-loop efficiency).
// Efficiencies measured on MI355X with the LDS-DMA kernel at B=32..128 (tools/layer_bench.py, profiles/).
int choose_tile(long long M, int Cout, int nk) {
    struct Cand { int id, bm, bn; double eff; };
    const Cand cands[] = {{5, 64, 128, 0.80}, {3, 64, 64, 0.78}, {2, 128, 64, 0.775}, {1, 128, 128, 0.74}, {6, 128, 32, 0.60}};
    int best = 3;
    double best_cost = 1e300;
    for (const Cand& c : cands) {
        const long long tiles = (long long)y2_cdiv(M, c.bm) * y2_cdiv(Cout, c.bn);
        const double cost = effective_rounds(tiles, nk) * c.bm * c.bn / c.eff;
        if (cost < best_cost) { best_cost = cost; best = c.id; }
    }
    return best;
}

}  // namespace

static int conv_fwd_impl(const y2_conv_params* p, y2_stream_t stream, size_t* ws_need, int groups = 1, long long gx = 0, long long gw = 0, long long gy = 0) {
    if (ws_need != nullptr) *ws_need = 0;
    if (p != nullptr && (p->algo == Y2_ALGO_WINOGRAD_F43 || p->algo == Y2_ALGO_WINOGRAD_F43_PRE) && groups == 1) return y2_internal_wino6_conv(p, stream, ws_need);
    if (p != nullptr && (p->algo == Y2_ALGO_WINOGRAD || p->algo == Y2_ALGO_WINOGRAD_FUSED || p->algo == Y2_ALGO_WINOGRAD_IMPLICIT || p->algo == Y2_ALGO_WINOGRAD_SPLIT || p->algo == Y2_ALGO_WINOGRAD_SPLIT_F16) && groups == 1) return y2_internal_wino_conv(p, stream, ws_need);
    if (p != nullptr && p->algo != Y2_ALGO_DIRECT && groups == 1) return Y2_EINVAL;
    if (p == nullptr || p->x == nullptr || p->w == nullptr) return Y2_EINVAL;
    if (p->y == nullptr && p->y_pool == nullptr && p->stats == nullptr) return Y2_EINVAL;
    if (p->B <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0) return Y2_EINVAL;
    if (p->ksize < 1 || p->ksize > 7) return Y2_ENOSUP;
    if (p->ldx < p->Cin) return Y2_EINVAL;
    const int stride = p->stride > 0 ? p->stride : 1;
    const int pad = p->pad_plus1 > 0 ? p->pad_plus1 - 1 : (p->ksize - 1) / 2;
    const bool transposed = p->transposed != 0;
    int Ho = (p->H + 2 * pad - p->ksize) / stride + 1, Wo = (p->W + 2 * pad - p->ksize) / stride + 1;
    if (transposed) {   // x is the gradient of a stride-s conv output; the result has the explicit size out_h x out_w of that conv's input
        Ho = p->out_h; Wo = p->out_w;
        if (Ho <= 0 || Wo <= 0 || (Ho + 2 * pad - p->ksize) / stride + 1 != p->H || (Wo + 2 * pad - p->ksize) / stride + 1 != p->W) return Y2_EINVAL;
    }
    if (Ho <= 0 || Wo <= 0) return Y2_EINVAL;
    const bool standard = !transposed && stride == 1 && pad == (p->ksize - 1) / 2 && (p->ksize == 1 || p->ksize == 3);
    if (p->y != nullptr && p->out_mode == 0 && p->ldy < p->coff + p->Cout) return Y2_EINVAL;
    if (p->y != nullptr && p->out_mode == 1 && (p->ldy < p->coff + 4 * p->Cout || (p->H & 1) || (p->W & 1))) return Y2_EINVAL;
    if (p->out_mode != 0 && p->out_mode != 1) return Y2_EINVAL;
    const bool pool = p->y_pool != nullptr;
    if (pool && ((p->H & 1) || (p->W & 1) || p->ldp < p->poff + p->Cout)) return Y2_EINVAL;
    if (p->residual != nullptr && (pool || p->out_mode != 0 || p->ldr < p->Cout)) return Y2_ENOSUP;
    if (p->stats != nullptr && y2_det.on && ws_need == nullptr) return Y2_ENOSUP;      // deterministic mode: statistics come from y2_colstats_det, not from epilogue atomics
    const long long Min = (long long)p->B * p->H * p->W;
    const long long M = (long long)p->B * Ho * Wo;
    if (M > 0x7fffffffLL / 2 || Min > 0x7fffffffLL / 2) return Y2_EINVAL;

    ConvArgs a;
    a.x = p->x; a.w = p->w; a.scale = p->scale; a.shift = p->shift;
    a.y = p->y; a.y_pool = p->y_pool; a.stats = p->stats;
    a.B = p->B; a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.ldx = p->ldx; a.Cout = p->Cout;
    a.taps = p->ksize * p->ksize;
    a.ldy = p->ldy; a.coff = p->coff; a.ldp = p->ldp; a.poff = p->poff; a.out_mode = p->out_mode;
    a.slope = p->slope;
    a.M = (int)M;
    a.cchunks = y2_cdiv(p->Cin, BK_DEFAULT);
    a.tiles_m = a.tiles_n = 0;
    a.x_bytes = a.w_bytes = 0;
    a.full_tiles = 0x7fffffff; a.ksplit = 1; a.partial = nullptr;
    a.fast_epi = (p->residual == nullptr && p->out_mode == 0 && ((pool && p->y == nullptr && p->stats == nullptr) || (!pool && p->y != nullptr))) ? 1 : 0;
    a.epi_bm = 64; a.epi_bn = 64;      // overwritten by the launcher with its tile size
    a.groups = groups; a.gx = gx; a.gw = gw; a.gy = gy;
    a.d_hw = y2_make_fastdiv((uint32_t)(p->H * p->W)); a.d_w = y2_make_fastdiv((uint32_t)p->W); a.d_w2 = y2_make_fastdiv((uint32_t)(2 * p->W));
    a.stride = stride; a.pad = pad; a.KW = p->ksize; a.Ho = Ho; a.Wo = Wo; a.K = a.taps * p->Cin;
    a.tstride = 1; a.d_ts = y2_make_fastdiv(1);
    if (transposed) {   // correlation of the s-dilated input with the (already 180-degree-rotated, y2_pack_weight mode 1) filter, padding k-1-pad
        a.tstride = stride; a.d_ts = y2_make_fastdiv((uint32_t)stride);
        a.stride = 1; a.pad = p->ksize - 1 - pad;
    }
    a.res = p->residual; a.ldr = p->ldr;
    a.d_cin = y2_make_fastdiv((uint32_t)p->Cin); a.d_kw = y2_make_fastdiv((uint32_t)p->ksize);
    a.d_howo = y2_make_fastdiv((uint32_t)(Ho * Wo)); a.d_wo = y2_make_fastdiv((uint32_t)Wo);
    // per-DEVICE cache: a __device__ symbol has one address per GPU (a process may touch more than one device)
    static const float* zeros_dev[Y2_MAX_DEVICES] = {};
    const float* zeros = nullptr;
    if (ws_need == nullptr) {
        const int dev = y2_current_device();
        if (dev < 0) return Y2_EINVAL;
        zeros = zeros_dev[dev];
        if (zeros == nullptr) {
            void* zp = nullptr;
            hipError_t e = hipGetSymbolAddress(&zp, HIP_SYMBOL(y2_zero16_storage));
            if (e != hipSuccess) return -(1000 + (int)e);
            zeros = zeros_dev[dev] = static_cast<const float*>(zp);
        }
    }
    a.zeros = zeros;

    const bool vec = (p->Cin % 4 == 0) && (p->ldx % 4 == 0) && y2_aligned16(p->x) && y2_aligned16(p->w);
    const int nk = standard ? a.taps * y2_cdiv(p->Cin, 32) : y2_cdiv(a.K, 32);
    int tile = p->tile > 0 ? p->tile : choose_tile(M * groups, p->Cout, nk);
    hipStream_t s = y2_s(stream);
    // tile ids 1,2,3,5,6: LDS-DMA kernel when the operands allow it; 101.. force the register-staged kernel (also the
    // path for channel counts / strides that are not multiples of 4, e.g. pruned checkpoints).  Strided / 7x7 / padded
    // variants and small Cin (K handled as one linear axis) use the GEN instantiation of the DMA kernel.
    const unsigned long long xb = (unsigned long long)Min * p->ldx * 4ull, wb = (unsigned long long)p->Cout * a.taps * p->Cin * 4ull;
    if (tile == 7 && (!standard || (p->Cin % 16) != 0 || groups > 1)) tile = 3;      // the wave-private kernel covers the standard convolutions only
    const bool persist = tile == 11 || tile == 12 || tile == 13 || tile == 15;
    if (persist && (!standard || groups > 1 || pool || (p->Cin % 32) != 0 || !vec)) return Y2_ENOSUP;      // (the caller offered a form this problem has none of)
    const bool dma_tile = (tile == 1 || tile == 2 || tile == 3 || tile == 5 || tile == 6 || tile == 7 || tile == 8 || tile == 9 || persist);
    const bool dma_ok = vec && xb < 0x7fffffffull && wb < 0x7fffffffull && dma_tile;
    const bool gen = !standard || (groups == 1 && dma_ok && tile != 7 && tile != 8 && tile != 9 && !persist && p->Cin < 32 && !pool && p->out_mode == 0);
    if (groups > 1 && (!dma_ok || gen || pool || p->out_mode != 0 || p->stats != nullptr || p->residual != nullptr)) return Y2_ENOSUP;
    float* ws = p->workspace;
    const size_t wsb = (ws != nullptr && y2_aligned16(ws)) ? (size_t)p->workspace_bytes : 0;
    if (gen) {
        if (!dma_ok || pool || p->out_mode != 0) return Y2_ENOSUP;
        a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
        return dispatch_dma<false, true>(a, tile, s, ws, wsb, ws_need);
    }
    if (persist && !dma_ok) return Y2_ENOSUP;
    if (dma_ok) {
        a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
        return pool ? dispatch_dma<true>(a, tile, s, ws, wsb, ws_need) : dispatch_dma<false>(a, tile, s, ws, wsb, ws_need);
    }
    if (ws_need != nullptr) return Y2_OK;
    if (p->residual != nullptr) return Y2_ENOSUP;
    if (tile > 100) tile -= 100;
    if (pool) return vec ? dispatch_tile<true, true>(a, tile, s) : dispatch_tile<true, false>(a, tile, s);
    return vec ? dispatch_tile<false, true>(a, tile, s) : dispatch_tile<false, false>(a, tile, s);
}

extern "C" int y2_conv_fwd(const y2_conv_params* p, y2_stream_t stream) { return conv_fwd_impl(p, stream, nullptr); }

// `groups` independent GEMMs of one shape in one launch (see ConvArgs::groups); library-internal (wino.hip).
int y2_internal_conv_grouped(const y2_conv_params* p, int groups, long long gx, long long gw, long long gy, y2_stream_t stream, size_t* ws_need) {
    if (groups < 1) return Y2_EINVAL;
    return conv_fwd_impl(p, stream, ws_need, groups, gx, gw, gy);
}

// Bytes of scratch y2_conv_fwd can use for this problem (0 = none needed); pass it via y2_conv_params.workspace.
extern "C" long long y2_conv_fwd_workspace_bytes(const y2_conv_params* p) {
    size_t need = 0;
    const int rc = conv_fwd_impl(p, nullptr, &need);
    return rc == Y2_OK ? (long long)need : (long long)rc;
}

// Coarse entry: a whole chain of convolutions (the Darknet stages) in one call — the host enqueues the launches
// back to back without returning to Python between layers (SURVEY.md 8e "coarse C-ABI entry").
extern "C" int y2_conv_fwd_batch(const y2_conv_params* params, int count, y2_stream_t stream) {
    if (params == nullptr || count < 0) return Y2_EINVAL;
    for (int i = 0; i < count; ++i) {
        const int rc = y2_conv_fwd(params + i, stream);
        if (rc != Y2_OK) return rc;
    }
    return Y2_OK;
}
