// conv0_fwd.hip — first Darknet layer (model/yolo2.py:78 'layers1.0': Conv2d(3, 32, 3) + BN + LeakyReLU,
// followed by MaxPool2d(2) at :79) for gfx950.
//
// Reads the plugin's NCHW fp32 image directly (the only NCHW tensor on the path) and writes NHWC, so no
// layout-conversion pass exists anywhere.  K = Cin*9 = 27 is far too small for the LDS-slab pipeline of
// conv_fwd.hip, and with 12 flop/B the layer sits next to the HBM roofline; it is still a dense contraction, so
// it runs on v_mfma_f32_32x32x2_f32 with the whole weight matrix (Cout x 27) resident in 14 VGPRs per lane:
//   workgroup = 16 x 32 output pixels of one image; the (16+2) x (32+2) x Cin input patch is staged in LDS
//   once (plane row stride 48 floats: the two image rows a wave reads land on disjoint banks);
//   each wave owns 4 row-pairs x 16 columns; lane l (l&31 = pixel in 2x2-window-major order, l>>5 = k parity)
//   gathers A[m][k] = patch[c][py+ky][px+kx] with one ds_read_b32 per MFMA;
//   epilogue: affine + LeakyReLU, 2x2 max over the lane's 4-register quad (window-major order makes the pool
//   in-lane), stores 128 B per pixel row (32 channels contiguous in NHWC).
//
// y2-build-flags: -mllvm -amdgpu-mfma-vgpr-form
//   (the accumulators live in ordinary VGPRs: the epilogue reads every one of them exactly once, and from AGPRs that is a v_accvgpr_read apiece)
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TH = 16, TW = 32;
constexpr int PW = 48;                 // LDS patch row stride (floats)
constexpr int PLANE = (TH + 2) * PW;   // floats per input channel

struct Conv0Args {
    const float* x; const float* w; const float* scale; const float* shift;
    float* y; float* y_pool; double* stats;
    int B, H, W, Cout, ldy, ldp;
    float slope;
    int tiles_y, tiles_x, ntiles;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The taps (c, ky, kx) of the K = 9 CIN contraction, two per MFMA step (k = 2s and 2s + 1 are read by lane halves 0 and 1), in four groups of steps whose
// second tap lies a fixed LDS distance from the first:
//   group 0: (c, ky, 0) + (c, ky, 1)            3 CIN steps, distance 1
//   group 1: (c, 0, 2) + (c, 1, 2)              CIN steps, distance PW
//   group 2: (2i, 2, 2) + (2i + 1, 2, 2)        CIN / 2 steps, distance PLANE
//   group 3: (CIN - 1, 2, 2) + nothing          CIN odd: one step whose second k is the zero padding
template <int CIN>
struct StepTap {
    static constexpr int G0 = 3 * CIN, G1 = G0 + CIN, G2 = G1 + CIN / 2;
    static constexpr int group(int s) { return s < G0 ? 0 : s < G1 ? 1 : s < G2 ? 2 : 3; }
    static constexpr int tap0(int s) {          // state_dict index c * 9 + ky * 3 + kx of the step's first tap
        return s < G0 ? (s / 3) * 9 + (s % 3) * 3 : s < G1 ? (s - G0) * 9 + 2 : s < G2 ? (2 * (s - G1)) * 9 + 8 : (CIN - 1) * 9 + 8;
    }
    static constexpr int tap1(int s) {          // ... of its second tap; -1: padding
        return s < G0 ? tap0(s) + 1 : s < G1 ? tap0(s) + 3 : s < G2 ? tap0(s) + 9 : -1;
    }
    static constexpr int lds(int tap) { return (tap / 9) * PLANE + ((tap % 9) / 3) * PW + (tap % 3); }          // patch offset of a tap
    static constexpr int dist(int g) { return g == 0 ? 1 : g == 1 ? PW : g == 2 ? PLANE : 0; }
};

// OUT >= 0: the output set is a compile-time constant (bit 0 = y, bit 1 = pooled output, bit 2 = statistics); FULL: every tile lies
// inside the image, Cout == 32 * NBLK and the outputs are dense (ld == Cout), so the epilogue has no bounds checks and every store is
// base + lane + constant.  RAW: no affine and no activation (the training forward: z and its statistics).  With OUT and FULL the epilogue
// is straight-line code (the runtime-flag form spends more instructions on tests than on arithmetic).  OUT = -1, FULL = false: the general kernel.
//
// What bounds this kernel (round 6, `tools/conv0_probe.py` with the -DY2_C0_DBG builds): a tile's 56 MFMAs per wave alone run at 0.75 of the matrix rate; the
// patch staging and the epilogue each ADDED their whole duration on top (0.167 + 0.053 + 0.054 ms at batch 64) - with 3 resident waves per SIMD a wave's
// serial instruction stream is what is timed, so the work is to make that stream short: row/channel arithmetic of the staging at compile time, one tile
// decode per tile, the pool taken BEFORE affine and activation (exact, see below), packed statistics.
template <int CIN, int NBLK, int OUT = -1, bool FULL = false, bool RAW = false>
__global__ __launch_bounds__(256) void conv0_kernel(const Conv0Args a) {
    const bool has_y = OUT < 0 ? a.y != nullptr : (OUT & 1) != 0;
    const bool has_pool = OUT < 0 ? a.y_pool != nullptr : (OUT & 2) != 0;
    const bool has_stats = OUT < 0 ? a.stats != nullptr : (OUT & 4) != 0;
    // pooled output only, no bounds: max over the window first, then ONE affine + activation.  Exact: the sign of the channel's scale is folded into its weights
    // (negation commutes with every rounding, so the accumulators are the reference's times +-1), and with |scale| >= 0 and 0 <= slope <= 1 (launch0 checks)
    // z -> leaky(z * |scale| + shift) is non-decreasing in every rounding step, so it commutes with max.
    constexpr bool POOL_FIRST = FULL && OUT == 2 && !RAW;
    constexpr int K = CIN * 9;
    constexpr int KS = (K + 1) / 2;
    constexpr int PR = TH + 2;                          // patch rows per channel
    constexpr int RPW = (PR + 3) / 4, RIT = CIN * RPW;   // rows of one channel per wave (wave w: rows w, w+4, ...); staging steps per wave
    __shared__ __attribute__((aligned(16))) float patch[3][CIN * PLANE];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ldy = FULL ? 32 * NBLK : a.ldy, ldp = FULL ? 32 * NBLK : a.ldp;

    float sc[NBLK], sh[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int n = j * 32 + l31;
        sc[j] = (!RAW && a.scale != nullptr && n < a.Cout) ? a.scale[n] : 1.f;
        sh[j] = (!RAW && a.shift != nullptr && n < a.Cout) ? a.shift[n] : 0.f;
    }
    // ---- weights: B[k][n] = w[n][k'], lane holds n = l31 (+32 per block) and the MFMA's k = 2s + half.  The MFMA k index is a PERMUTATION of the taps
    // k' = (c, ky, kx) (state_dict order) chosen so that the two taps of a step lie a constant LDS distance apart within a group of steps (StepTap below): a lane's
    // operand address is then one of four per-step bases + a compile-time offset instead of a select + add per read.  (A sum over k in another order: the MFMA
    // accumulates the two k of a step and the steps in sequence either way, and fp32 parity with the oracle is a tolerance, tests/test_gpu_kernels.py.)
    float wreg[NBLK][KS];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int n = j * 32 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = half ? StepTap<CIN>::tap1(s) : StepTap<CIN>::tap0(s);
            const float w = (n < a.Cout && k >= 0) ? a.w[(size_t)n * K + k] : 0.f;
            wreg[j][s] = (POOL_FIRST && sc[j] < 0.f) ? -w : w;
        }
        if (POOL_FIRST) sc[j] = fabsf(sc[j]);
    }

    // ---- staging of one tile's (TH+2) x (TW+2) x CIN input patch, one patch row per wave and step: channel and row-in-channel are compile-time + wave,
    // the row's address and bounds are wave-uniform (scalar registers), a lane adds its column.  issue() only LOADS (from an address clamped into the
    // image); commit() selects the zero padding and writes LDS, so the loads of a tile stay in flight across a whole tile of MFMAs.
    float pv[RIT];
    const int HW = a.H * a.W;
    auto origin = [&](int tile, int& b, int& y0, int& x0) {
        const unsigned u = (unsigned)tile, row = u / (unsigned)a.tiles_x, tx = u - row * (unsigned)a.tiles_x;
        const unsigned ub = row / (unsigned)a.tiles_y, ty = row - ub * (unsigned)a.tiles_y;
        b = (int)ub; y0 = (int)ty * TH; x0 = (int)tx * TW;
    };
    auto issue = [&](int b, int y0, int x0) {
        const float* xb = a.x + (size_t)b * CIN * HW;
        const int gx = min(max(x0 + lane - 1, 0), a.W - 1);
#pragma unroll
        for (int it = 0; it < RIT; ++it) {
            const int c = it / RPW, py = min(wave + 4 * (it % RPW), PR - 1);          // (a last step's surplus waves reload the last row; commit() drops it)
            const int gy = min(max(y0 + py - 1, 0), a.H - 1);
            const float* rowp = xb + (c * HW + gy * a.W);
            pv[it] = rowp[gx];
        }
    };
    auto commit = [&](int y0, int x0, float* dst) {
        const bool okx = (unsigned)(x0 + lane - 1) < (unsigned)a.W;
        float* d = dst + wave * PW + lane;
        if (lane < TW + 2) {
#pragma unroll
            for (int it = 0; it < RIT; ++it) {
                const int c = it / RPW, k4 = 4 * (it % RPW), py = wave + k4;
                const bool oky = (unsigned)(y0 + py - 1) < (unsigned)a.H;
                const float v = (okx && oky) ? pv[it] : 0.f;
                if (k4 + 3 < PR || py < PR) d[c * PLANE + k4 * PW] = v;
            }
        }
    };

    // Workgroups are PERSISTENT over tiles (round 6) and the patches are pipelined two tiles deep through THREE LDS buffers: while tile T computes out of
    // one, tile T+1 already sits in the second, and T+2 is loaded behind the first step's MFMAs and written to the third in front of the second step's stores.
    // Load and LDS write of a patch are in ONE iteration with only the first step's stores between them, so the wait in front of the LDS write is an exact
    // vmcnt(stores of one step + younger loads): the memory counter is in order, and the earlier form (load in one iteration, write in the next) drained a whole
    // tile's 64 stores per wave in front of every patch - 0.08 ms of the training-mode kernel at batch 64.
    // The statistics' per-tile fp32 sums are folded into fp64 accumulators that live across the tiles.
    double d1[NBLK], d2[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) { d1[j] = 0.0; d2[j] = 0.0; }
    const int stride = gridDim.x;
    int tile = blockIdx.x;
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): weights and affine are in registers (otherwise every block's first MFMA waits on "all but the newest 14" memory operations - the next patch's loads)
    int b, y0, x0, b1 = 0, y1 = 0, x1 = 0;        // origin of this tile and of the next
    origin(tile, b, y0, x0);
    issue(b, y0, x0);
    commit(y0, x0, patch[0]);
    if (tile + stride < a.ntiles) { origin(tile + stride, b1, y1, x1); issue(b1, y1, x1); commit(y1, x1, patch[1]); }
    __syncthreads();

    // lane's pixel inside a 2-row x 16-col block, 2x2-window-major: window wi = l31>>2, element e = l31&3
    const int wi = l31 >> 2, e = l31 & 3;
    const int lpy = e >> 1, lpx = 2 * wi + (e & 1);
    int buf = 0;
    for (; tile < a.ntiles; tile += stride, buf = buf == 2 ? 0 : buf + 1) {
    const float* cur = patch[buf];
    const bool more = tile + 2 * stride < a.ntiles;
    int b2 = 0, y2 = 0, x2 = 0;
    f32x2 s1[NBLK], s2[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int bt = b, yt = y0, xt = x0;

#pragma unroll 1          // (unrolled, the two steps measured 5-8 % slower at batch 64 although the wait in front of the LDS write then is an exact count)
    for (int blk = 0; blk < 4; blk += 2) {
        // TWO blocks per step - the two column halves of one row pair - with their MFMA chains interleaved
        const int rp = (wave * 4 + blk) >> 1;  // row pair (16 blocks per tile: 8 row pairs x 2 column halves)
        const int pbase = (2 * rp + lpy) * PW + lpx;   // patch index of tap (ky=0,kx=0), column half 0
        f32x16 acc[2][NBLK];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][j][r] = 0.f;
        // all operand reads of the step first, then the MFMAs (read -> wait -> MFMA per step left the matrix pipe idle for an LDS round trip 14 times per block)
        float av[2][KS];
        const float* gb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gb[g] = cur + pbase + (half ? StepTap<CIN>::dist(g) : 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int g = StepTap<CIN>::group(s);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                av[u][s] = gb[g][StepTap<CIN>::lds(StepTap<CIN>::tap0(s)) + 16 * u];
                if (g == 3) av[u][s] = half ? 0.f : av[u][s];          // the padding k: its weight is 0, and 0 x (whatever the patch holds there) must not be NaN
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#if defined(Y2_C0_DBG) && (Y2_C0_DBG & 1)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][j][r] = av[u][r % KS] * wreg[j][r % KS];
#else
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NBLK; ++j)
                    acc[u][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][s], wreg[j][s], acc[u][j], 0, 0, 0);
#endif
#if !(defined(Y2_C0_DBG) && (Y2_C0_DBG & 4))
        if (more) {
            if (blk == 0) { origin(tile + 2 * stride, b2, y2, x2); issue(b2, y2, x2); }
            else commit(y2, x2, patch[buf == 0 ? 2 : buf - 1]);
        }
#endif
        // ---- epilogue: register quad g -> window 2g + half of its block
#if defined(Y2_C0_DBG) && (Y2_C0_DBG & 2)
        if (acc[0][0][0] + acc[1][0][5] == 12345.678f) a.y_pool[lane] = 1.f;
        if (false)
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {
        const int Y0 = yt + 2 * rp, X0 = xt + 16 * u;                 // top-left pixel of the block (wave-uniform)
        // wave-uniform row pointers (scalar registers) + one 32-bit lane offset: every store is `saddr + lane + constant`, no vector address arithmetic
        float* yb0 = has_y ? a.y + ((size_t)(bt * a.H + Y0) * a.W + X0) * ldy : nullptr;
        float* yb1 = has_y ? yb0 + (size_t)a.W * ldy : nullptr;
        float* pb = has_pool ? a.y_pool + ((size_t)(bt * (a.H >> 1) + (Y0 >> 1)) * (a.W >> 1) + (X0 >> 1)) * ldp : nullptr;
        const unsigned ylane = (unsigned)(2 * half * ldy + l31), plane = (unsigned)(half * ldp + l31);
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int n = j * 32 + l31;
            const bool nok = FULL || n < a.Cout;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int w = 2 * g + half;
                const int Y = Y0, X = X0 + 2 * w;   // top-left pixel of the window
                if (has_stats) {
                    if (FULL) {          // two partial sums per lane, packed arithmetic
                        const f32x2 za = {acc[u][j][4 * g], acc[u][j][4 * g + 1]}, zb = {acc[u][j][4 * g + 2], acc[u][j][4 * g + 3]};
                        s1[j] += za; s2[j] = __builtin_elementwise_fma(za, za, s2[j]);
                        s1[j] += zb; s2[j] = __builtin_elementwise_fma(zb, zb, s2[j]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float z = acc[u][j][4 * g + q];
                            if ((Y + (q >> 1)) < a.H && (X + (q & 1)) < a.W) { s1[j][q & 1] += z; s2[j][q & 1] += z * z; }
                        }
                    }
                }
                if (!nok) continue;
                if (POOL_FIRST) {
                    const float m = fmaxf(fmaxf(acc[u][j][4 * g], acc[u][j][4 * g + 1]), fmaxf(acc[u][j][4 * g + 2], acc[u][j][4 * g + 3]));
                    const float uu = m * sc[j] + sh[j];
                    (pb + (2 * g * ldp + j * 32))[plane] = fmaxf(uu, uu * a.slope);
                    continue;
                }
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = acc[u][j][4 * g + q];
                    if (RAW) { v[q] = z; continue; }
                    const float uu = z * sc[j] + sh[j];
                    v[q] = uu > 0.f ? uu : uu * a.slope;
                }
                if (has_y) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int yy = Y + (q >> 1), xx = X + (q & 1);
                        if (FULL || (yy < a.H && xx < a.W)) (((q >> 1) ? yb1 : yb0) + ((4 * g + (q & 1)) * ldy + j * 32))[ylane] = v[q];
                    }
                }
                if (has_pool && (FULL || (Y < a.H && X < a.W))) {
                    const float pm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    (pb + (2 * g * ldp + j * 32))[plane] = pm;
                }
            }
        }
        }
    }
    if (has_stats) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) { d1[j] += (double)s1[j][0] + (double)s1[j][1]; d2[j] += (double)s2[j][0] + (double)s2[j][1]; }
    }
    b = b1; y0 = y1; x0 = x1; b1 = b2; y1 = y2; x1 = x2;
    __syncthreads();          // this tile's buffer has been read by every wave; the patch of the tile after the next is complete
    }
    if (has_stats) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int n = j * 32 + l31;
            const double t1 = d1[j] + __shfl_xor(d1[j], 32);
            const double t2 = d2[j] + __shfl_xor(d2[j], 32);
            if (half == 0 && n < a.Cout) {
                double* st = a.stats + (size_t)(blockIdx.x % Y2_STATS_REPL) * 2 * a.Cout;   // replicated accumulators
                atomicAdd(st + n, t1);
                atomicAdd(st + a.Cout + n, t2);
            }
        }
    }
}

template <int CIN>
int launch0(const Conv0Args& a, hipStream_t s) {
    const long long tiles = (long long)a.B * a.tiles_y * a.tiles_x;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return Y2_EINVAL;
    static const int per_cu = getenv("Y2_CONV0_WGS") != nullptr ? atoi(getenv("Y2_CONV0_WGS")) : 8;      // resident workgroups per CU the persistent grid aims at (0: one workgroup per tile)
    const long long grid = per_cu > 0 && tiles > (long long)Y2_NUM_CU * per_cu ? (long long)Y2_NUM_CU * per_cu : tiles;
    const double flops = 2.0 * (double)a.B * a.H * a.W * 9 * CIN * a.Cout;
    if (CIN == 3 && (a.H % TH) == 0 && (a.W % TW) == 0 && (a.Cout == 32 || a.Cout == 64) && (a.y == nullptr || a.ldy == a.Cout) && (a.y_pool == nullptr || a.ldp == a.Cout)) {
        // the shipped first layers (3 -> 32 Darknet / 3 -> 16 Tiny falls through) on tile-aligned images: straight-line epilogues
        const int out = (a.y != nullptr ? 1 : 0) | (a.y_pool != nullptr ? 2 : 0) | (a.stats != nullptr ? 4 : 0);
        const bool raw = a.scale == nullptr && a.shift == nullptr && a.slope == 1.f;           // the training forward: z and its statistics
        const bool mono = a.slope >= 0.f && a.slope <= 1.f;                                      // the pool-first epilogue's condition
#define Y2_C0(NB_, OUT_, RAW_) Y2_LAUNCH("conv0_kernel", flops, (conv0_kernel<3, NB_, OUT_, true, RAW_>), dim3((unsigned)grid), dim3(256), 0, s, a)
        bool done = true;
        if (a.Cout == 32) {
            if (out == 2 && mono) Y2_C0(1, 2, false); else if (out == 5 && raw) Y2_C0(1, 5, true); else if (out == 5) Y2_C0(1, 5, false);
            else if (out == 1) Y2_C0(1, 1, false); else if (out == 3) Y2_C0(1, 3, false); else done = false;
        } else {
            if (out == 2 && mono) Y2_C0(2, 2, false); else if (out == 5 && raw) Y2_C0(2, 5, true); else if (out == 1) Y2_C0(2, 1, false); else done = false;
        }
#undef Y2_C0
        if (done) { Y2_LAUNCH_CHECK(); return Y2_OK; }
    }
    if (a.Cout <= 32) Y2_LAUNCH("conv0_kernel", flops, (conv0_kernel<CIN, 1>), dim3((unsigned)grid), dim3(256), 0, s, a);
    else Y2_LAUNCH("conv0_kernel", flops, (conv0_kernel<CIN, 2>), dim3((unsigned)grid), dim3(256), 0, s, a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

}  // namespace

extern "C" int y2_conv0_fwd(const float* x_nchw, const float* w, const float* scale, const float* shift,
                            float* y, float* y_pool, double* stats,
                            int B, int H, int W, int Cin, int Cout, int ldy, int ldp, float slope, y2_stream_t stream) {
    if (x_nchw == nullptr || w == nullptr || (y == nullptr && y_pool == nullptr && stats == nullptr)) return Y2_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || Cin < 1 || Cin > 4 || Cout < 1 || Cout > 64) return Y2_ENOSUP;
    if (y != nullptr && ldy < Cout) return Y2_EINVAL;
    if (y_pool != nullptr && (ldp < Cout || (H & 1) || (W & 1))) return Y2_EINVAL;
    if (stats != nullptr && y2_det.on) return Y2_ENOSUP;      // deterministic mode: statistics come from y2_colstats_det
    Conv0Args a;
    a.x = x_nchw; a.w = w; a.scale = scale; a.shift = shift; a.y = y; a.y_pool = y_pool; a.stats = stats;
    a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.ldy = ldy; a.ldp = ldp; a.slope = slope;
    a.tiles_y = y2_cdiv(H, TH); a.tiles_x = y2_cdiv(W, TW);
    a.ntiles = B * a.tiles_y * a.tiles_x;
    hipStream_t s = y2_s(stream);
    switch (Cin) {
        case 1: return launch0<1>(a, s);
        case 2: return launch0<2>(a, s);
        case 3: return launch0<3>(a, s);
        default: return launch0<4>(a, s);
    }
}
