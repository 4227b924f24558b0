// conv_wgrad.hip — weight gradient of a stride-1 "same" convolution as an fp32-MFMA GEMM over pixels (gfx950).
//
// Replaces the autograd weight-gradient of nn.Conv2d inside model.yolo2.Conv2d (model/yolo2.py:57; train.py:351
// `loss_total.backward()`):   dW[co][tap][ci] = sum_m dz[m][co] * x[m + tap][ci]      (m = pixel, zero outside image)
//
// GEMM view:  D[i][j] = sum_k A[i][k] * B[k][j]
//   i = output channel co            M' = Cout
//   j = (tap, ci)                    N' = taps * Cin      -> D is exactly the packed fprop weight layout [Cout][tap][Cin]
//   k = pixel m                      K' = B*H*W           -> split over `splits` workgroups, partial sums combined with
//                                                            fp32 atomic adds into a pre-zeroed dW
//   A[i][k] = dz[k][i]   (NHWC rows of dz: 128 contiguous channels per pixel)
//   B[k][j] = x[k + off(tap_j)][ci_j]   (NHWC rows of x shifted by the tap; 0 outside the image)
// Both operands are k-major in memory, so LDS holds them exactly as loaded: [32 pixels][128 channels] per slab, filled
// by LDS-DMA (buffer_load_dwordx4 ... lds, out-of-range voffset -> zeros), and the MFMA fragments are column reads:
// lane (i = l&31, k = 2s + (l>>5)) reads one dword at [k][i] — 32 consecutive banks per half-wave, conflict-free.
// Workgroup = 4 waves (2x2), tile 128 (co) x 128 (j), wave tile 64x64 = 2x2 MFMA 32x32x2 blocks; 64 MFMAs per slab per
// wave against 16 KB + 16 KB of DMA: the same MFMA-bound balance as the forward kernel.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int TI = 128, TJ = 128, KS = 32;   // tile rows (co), tile cols (tap,ci), pixels per slab

struct WgradArgs {
    const float* x;     // [B,H,W,Cin] pixel stride ldx
    const float* dz;    // [B,H,W,Cout] pixel stride ldz
    float* dw;          // [Cout][taps*Cin], pre-zeroed
    int B, H, W, Cin, ldx, Cout, ldz, taps;
    int M, tiles_i, tiles_j, splits, slabs_per_split;
    unsigned x_bytes, dz_bytes;
};

__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int STAGE = 2 * KS * 128;   // floats per stage: A [32][128] then B [32][128]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    int bid = blockIdx.x;
    const int split = bid % a.splits; bid /= a.splits;
    const int tj = bid % a.tiles_j;
    const int ti = bid / a.tiles_j;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int ncols = a.taps * a.Cin;

    const int slab_lo = split * a.slabs_per_split;
    const int nslab_total = (a.M + KS - 1) / KS;
    const int slab_hi = min(nslab_total, slab_lo + a.slabs_per_split);
    const int nk = slab_hi - slab_lo;
    if (nk <= 0) return;

    // ---- DMA assignment: lane -> 16-B chunk (t&31) of pixel row (t>>5) + 8*p, p = 0..3
    const int chunk = t & 31;
    const int prow = t >> 5;
    // A (dz): column i0 + 4*chunk
    const int ca = i0 + 4 * chunk;
    const bool a_ok = ca < a.Cout;
    // B (x): column j = j0 + 4*chunk -> (tap, ci), fixed for the whole kernel
    const int jb = j0 + 4 * chunk;
    const bool b_ok = jb < ncols;
    const int tap = b_ok ? jb / a.Cin : 0;
    const int ci = b_ok ? jb - tap * a.Cin : 0;
    const int dy = (a.taps == 9) ? tap / 3 - 1 : 0;
    const int dx = (a.taps == 9) ? tap % 3 - 1 : 0;
    // (y, x) of this thread's 4 staged pixels in the first slab; advanced by 32 pixels per slab
    int py[4], px[4];
    const int q32 = KS / a.W, r32 = KS % a.W;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = slab_lo * KS + prow + 8 * p;
        const int idx = m % (a.H * a.W);
        py[p] = idx / a.W;
        px[p] = idx - py[p] * a.W;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, a.dz_bytes, 0x00020000);

    auto issue_slab = [&](int slab, int buf) {
        float* sa = smem + buf * STAGE + wave * 256;          // wave covers 2 rows x 128 floats = 256 floats per pass
        float* sb = sa + KS * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = slab * KS + prow + 8 * p;
            const bool mok = m < a.M;
            const unsigned va = (mok && a_ok) ? (unsigned)(((size_t)m * a.ldz + ca) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (lds_ptr_t)(sa + p * 8 * 128), 16, (int)va, 0, 0, 0);
            const bool in = (unsigned)(py[p] + dy) < (unsigned)a.H && (unsigned)(px[p] + dx) < (unsigned)a.W;
            const unsigned vb = (mok && b_ok && in) ? (unsigned)(((size_t)(m + dy * a.W + dx) * a.ldx + ci) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sb + p * 8 * 128), 16, (int)vb, 0, 0, 0);
        }
        // advance the pixel coordinates to the next slab (+32 pixels, row-major, wraps at image boundaries)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            px[p] += r32; py[p] += q32;
            if (px[p] >= a.W) { px[p] -= a.W; py[p] += 1; }
            if (py[p] >= a.H) py[p] %= a.H;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fa = wm * 64 + l31 + half * 128;            // + (2s)*128 + 32*block
    const int fb = KS * 128 + wn * 64 + l31 + half * 128;

    auto compute_slab = [&](int buf) {
        const float* sbuf = smem + buf * STAGE;
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = sbuf[fa + s * 256 + 32 * i];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = sbuf[fb + s * 256 + 32 * j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    issue_slab(slab_lo, 0);
    for (int ks = 0; ks < nk - 1; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue_slab(slab_lo + ks + 1, (ks + 1) & 1);
        compute_slab(ks & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    compute_slab((nk - 1) & 1);

    // ---- epilogue: lane -> column j (l31), register r -> row co = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int jbk = 0; jbk < 2; ++jbk) {
        const int j = j0 + wn * 64 + jbk * 32 + l31;
        if (j >= ncols) continue;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i0 + wm * 64 + ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < a.Cout) {
                    float* dst = a.dw + (size_t)co * ncols + j;
                    if (a.splits > 1) atomicAdd(dst, acc[ib][jbk][r]);
                    else *dst = acc[ib][jbk][r];
                }
            }
        }
    }
}

}  // namespace

// dW[Cout][k*k][Cin] (packed layout, y2_unpack_weight_grad converts to the state_dict layout) += / = wgrad.
// dw must be zero-filled by the caller when the kernel decides to split (it always may): zero it unconditionally.
extern "C" int y2_conv_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int ldx, int Cout, int ldz,
                             int ksize, y2_stream_t stream) {
    if (!x || !dz || !dw || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return Y2_EINVAL;
    if (ksize != 1 && ksize != 3) return Y2_ENOSUP;
    if (ldx < Cin || ldz < Cout) return Y2_EINVAL;
    if ((Cin & 3) || (ldx & 3) || (Cout & 3) || (ldz & 3) || !y2_aligned16(x) || !y2_aligned16(dz)) return Y2_EALIGN;
    const long long M = (long long)B * H * W;
    const unsigned long long xb = (unsigned long long)M * ldx * 4ull, zb = (unsigned long long)M * ldz * 4ull;
    if (M > 0x3fffffffLL || xb >= 0x7fffffffull || zb >= 0x7fffffffull) return Y2_ENOSUP;
    WgradArgs a;
    a.x = x; a.dz = dz; a.dw = dw;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.Cout = Cout; a.ldz = ldz; a.taps = ksize * ksize;
    a.M = (int)M;
    a.tiles_i = y2_cdiv(Cout, TI);
    a.tiles_j = y2_cdiv(a.taps * Cin, TJ);
    const int tiles = a.tiles_i * a.tiles_j;
    const int slabs = y2_cdiv(M, KS);
    // enough workgroups to fill the chip several times over, but >= 8 slabs each so the atomic epilogue stays small
    int splits = y2_cdiv(4 * Y2_NUM_CU, tiles);
    const int max_splits = slabs / 8 > 0 ? slabs / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.slabs_per_split = y2_cdiv(slabs, splits);
    a.splits = y2_cdiv(slabs, a.slabs_per_split);
    a.x_bytes = (unsigned)xb; a.dz_bytes = (unsigned)zb;
    const size_t lds = 2u * 2u * KS * 128 * sizeof(float);   // 64 KB
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return -(1000 + (int)e);
        attr = true;
    }
    const long long grid = (long long)tiles * a.splits;
    if (grid > 0x7fffffffLL) return Y2_EINVAL;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)grid), dim3(NT), lds, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
