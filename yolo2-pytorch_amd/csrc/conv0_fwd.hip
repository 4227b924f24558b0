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
    int tiles_y, tiles_x;
};

// OUT >= 0: the output set is a compile-time constant (bit 0 = y, bit 1 = pooled output, bit 2 = statistics); FULL: every tile lies
// inside the image and Cout == 32 * NBLK, so the epilogue has no bounds checks.  With both, the epilogue is straight-line code
// (the runtime-flag form spends more instructions on tests than on arithmetic).  OUT = -1, FULL = false: the general kernel.
template <int CIN, int NBLK, int OUT = -1, bool FULL = false>
__global__ __launch_bounds__(256) void conv0_kernel(const Conv0Args a) {
    const bool has_y = OUT < 0 ? a.y != nullptr : (OUT & 1) != 0;
    const bool has_pool = OUT < 0 ? a.y_pool != nullptr : (OUT & 2) != 0;
    const bool has_stats = OUT < 0 ? a.stats != nullptr : (OUT & 4) != 0;
    constexpr int K = CIN * 9;
    constexpr int KS = (K + 1) / 2;
    __shared__ __attribute__((aligned(16))) float patch[CIN * PLANE];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int b = bid / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the input patch (zero outside the image = conv padding)
    const float* xb = a.x + (size_t)b * CIN * a.H * a.W;
    for (int i = t; i < CIN * (TH + 2) * (TW + 2); i += 256) {
        const int c = i / ((TH + 2) * (TW + 2));
        const int r = i - c * ((TH + 2) * (TW + 2));
        const int py = r / (TW + 2), px = r - py * (TW + 2);
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) v = xb[((size_t)c * a.H + gy) * a.W + gx];
        patch[c * PLANE + py * PW + px] = v;
    }

    // ---- weights: B[k][n] = w[n][k], k = (c, ky, kx) in state_dict order; lane holds n = l31 (+32 per block), k = 2s + half
    float wreg[NBLK][KS];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int n = j * 32 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 2 * s + half;
            wreg[j][s] = (n < a.Cout && k < K) ? a.w[(size_t)n * K + k] : 0.f;
        }
    }
    float sc[NBLK], sh[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int n = j * 32 + l31;
        sc[j] = (a.scale != nullptr && n < a.Cout) ? a.scale[n] : 1.f;
        sh[j] = (a.shift != nullptr && n < a.Cout) ? a.shift[n] : 0.f;
    }
    float s1[NBLK], s2[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

    __syncthreads();

    // lane's pixel inside a 2-row x 16-col block, 2x2-window-major: window wi = l31>>2, element e = l31&3
    const int wi = l31 >> 2, e = l31 & 3;
    const int lpy = e >> 1, lpx = 2 * wi + (e & 1);

#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
        const int bi = wave * 4 + blk;        // 16 blocks per tile
        const int rp = bi >> 1, ch = bi & 1;  // row pair, column half
        const int pbase = (2 * rp + lpy) * PW + 16 * ch + lpx;   // patch index of tap (ky=0,kx=0)
        f32x16 acc[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // k = 2s + half -> (c, ky, kx); both candidates are compile-time constants
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int o0 = (k0 / 9) * PLANE + ((k0 % 9) / 3) * PW + (k0 % 3);
            const int o1 = (k1 < K) ? (k1 / 9) * PLANE + ((k1 % 9) / 3) * PW + (k1 % 3) : 0;
            float av = patch[pbase + (half ? o1 : o0)];
            if (k1 >= K) av = half ? 0.f : av;
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wreg[j][s], acc[j], 0, 0, 0);
        }
        // ---- epilogue: register quad g -> window 2g + half of this block
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int n = j * 32 + l31;
            const bool nok = FULL || n < a.Cout;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int w = 2 * g + half;
                const int Y = y0 + 2 * rp, X = x0 + 16 * ch + 2 * w;   // top-left pixel of the window
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = acc[j][4 * g + q];
                    const bool in = FULL || ((Y + (q >> 1)) < a.H && (X + (q & 1)) < a.W);
                    if (has_stats && in) { s1[j] += z; s2[j] += z * z; }
                    const float u = z * sc[j] + sh[j];
                    v[q] = u > 0.f ? u : u * a.slope;
                }
                if (!nok) continue;
                if (has_y) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int yy = Y + (q >> 1), xx = X + (q & 1);
                        if (FULL || (yy < a.H && xx < a.W)) a.y[((size_t)(b * a.H + yy) * a.W + xx) * a.ldy + n] = v[q];
                    }
                }
                if (has_pool && (FULL || (Y < a.H && X < a.W))) {
                    const float pm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    a.y_pool[((size_t)(b * (a.H >> 1) + (Y >> 1)) * (a.W >> 1) + (X >> 1)) * a.ldp + n] = pm;
                }
            }
        }
    }
    if (has_stats) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int n = j * 32 + l31;
            float t1 = s1[j] + __shfl_xor(s1[j], 32);
            float t2 = s2[j] + __shfl_xor(s2[j], 32);
            if (half == 0 && n < a.Cout) {
                double* st = a.stats + (size_t)(blockIdx.x % Y2_STATS_REPL) * 2 * a.Cout;   // replicated accumulators
                atomicAdd(st + n, (double)t1);
                atomicAdd(st + a.Cout + n, (double)t2);
            }
        }
    }
}

template <int CIN>
int launch0(const Conv0Args& a, hipStream_t s) {
    const long long grid = (long long)a.B * a.tiles_y * a.tiles_x;
    if (grid <= 0 || grid > 0x7fffffffLL) return Y2_EINVAL;
    const double flops = 2.0 * (double)a.B * a.H * a.W * 9 * CIN * a.Cout;
    if (CIN == 3 && (a.H % TH) == 0 && (a.W % TW) == 0 && (a.Cout == 32 || a.Cout == 64)) {
        // the shipped first layers (3 -> 32 Darknet / 3 -> 16 Tiny falls through) on tile-aligned images: straight-line epilogues
        const int out = (a.y != nullptr ? 1 : 0) | (a.y_pool != nullptr ? 2 : 0) | (a.stats != nullptr ? 4 : 0);
#define Y2_C0(NB_, OUT_) Y2_LAUNCH("conv0_kernel", flops, (conv0_kernel<3, NB_, OUT_, true>), dim3((unsigned)grid), dim3(256), 0, s, a)
        bool done = true;
        if (a.Cout == 32) {
            if (out == 2) Y2_C0(1, 2); else if (out == 5) Y2_C0(1, 5); else if (out == 1) Y2_C0(1, 1); else if (out == 3) Y2_C0(1, 3); else done = false;
        } else {
            if (out == 2) Y2_C0(2, 2); else if (out == 5) Y2_C0(2, 5); else if (out == 1) Y2_C0(2, 1); else done = false;
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
    hipStream_t s = y2_s(stream);
    switch (Cin) {
        case 1: return launch0<1>(a, s);
        case 2: return launch0<2>(a, s);
        case 3: return launch0<3>(a, s);
        default: return launch0<4>(a, s);
    }
}
