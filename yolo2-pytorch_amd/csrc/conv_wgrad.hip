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
// y2-build-flags: -mllvm -amdgpu-mfma-vgpr-form
//   (accumulators in ordinary VGPRs: same reason as conv_fwd.hip: no v_accvgpr_read in front of the partial-sum stores, up to 43 fewer registers)
#include <stdlib.h>
#include "common.h"

#ifndef Y2_WGRAD_SPREAD
#define Y2_WGRAD_SPREAD 1     // 1: LDS-DMA of slab s+1 between the MFMAs of slab s; 0: all in front (A/B builds)
#endif

namespace {

constexpr int NT = 256;
constexpr int KS = 32;             // pixels per slab; tile rows TI (co) and tile columns TJ ((tap, ci): 128 or 64) are template parameters

struct WgradArgs {
    const float* x;     // [B,H,W,Cin] pixel stride ldx
    const float* dz;    // [B,H,W,Cout] pixel stride ldz
    float* dw;          // [Cout][taps*Cin], pre-zeroed
    float* partial;     // deterministic mode: [groups][splits][Cout][taps*Cin] partial sums, plain stores (added by det_reduce_rows_kernel)
    int B, H, W, Cin, ldx, Cout, ldz, taps;   // H, W: spatial size of dz (the conv OUTPUT)
    int Hi, Wi, stride, pad, KW;              // conv input size, stride, padding, kernel width
    int M, tiles_i, tiles_j, splits, slabs_per_split;
    unsigned x_bytes, dz_bytes;
    // grouped mode (Winograd wgrad: 16 independent reductions of one shape): group g uses x + g*gx, dz + g*gz, dw + g*gw
    int groups;
    long long gx, gz, gw;
    // DZRAW (the Winograd weight gradient without a dM tensor): dz is the layer's raw output gradient [B,zH,zW,Cout] (pixel stride ldz, gz = 0)
    // and group g = position (xi, nu) builds its operand rows dM_g[t] = (A dz_t A^T)[xi][nu] in the loader; tile_pix[t] = pixel index of tile t's
    // first pixel | bit 30: its second row exists | bit 31: its second column exists (wino_tile_table_kernel)
    const int32_t* tile_pix;
    int zW;
};

// LIN: 1x1 / stride 1 / no padding (every grouped Winograd reduction and the 1x1 layers): input pixel = output pixel, so the operand
// offsets of the next slab are the current ones + 32 rows (one v_add per DMA instruction) and the buffer's num_records supplies the
// zeros past the last pixel.  The general form recomputes (image, y, x) -> address per piece: ~100 VALU instructions per slab next to
// 64 MFMAs per wave (PMC: twice the VALU activity of the forward GEMM kernel at 0.69 instead of 0.81 MFMA-busy).
// DZRAW (with LIN, 16 groups): the A operand is not DMA'd from a transformed gradient tensor; a thread owns the same (tile row, 4-channel chunk)
// items the DMA lane owned, loads the 1, 2 or 4 gradient pixels its position needs (buffer loads, a missing pixel = out-of-range offset = 0),
// combines them with the operations of wino_dz_kernel in the same order and ds_writes the chunk where the DMA would have put it.  wino_dz_kernel
// and the 4x-gradient tensor dM (write + read) disappear; the gradient pixels are read 2.25x on average (36 pixel reads for 16 positions), from L2.
template <int TI, int WAVES_I, int TJ = 128, bool LIN = false, bool DZRAW = false>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int WAVES_J = 4 / WAVES_I;
    constexpr int WI = TI / WAVES_I, WJ = TJ / WAVES_J;
    constexpr int IB = WI / 32, JB = WJ / 32;          // 32x32 MFMA blocks per wave
    constexpr int CA = TI / 4;                          // 16-B chunks per A row
    constexpr int RA = NT / CA;                         // A rows per DMA pass
    constexpr int PA = KS / RA;                         // A passes per slab
    constexpr int CB = TJ / 4, RB = NT / CB, PB = KS / RB;   // the same for B: 32 chunks x 8 rows x 4 passes (TJ = 128), 16 x 16 x 2 (TJ = 64)
    static_assert(WJ >= 32 && WI >= 32, "a wave owns at least one 32x32 block");
    constexpr int STAGE = KS * TI + KS * TJ;            // floats per stage: A [32][TI] then B [32][128]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WAVES_J, wn = wave % WAVES_J;
    const int l31 = lane & 31, half = lane >> 5;

    int bid = blockIdx.x;
    int grp = 0;
    if (a.groups > 1) {
        const int per = a.tiles_i * a.tiles_j * a.splits;
        grp = bid / per;
        bid -= grp * per;
    }
    const float* gx_ptr = a.x + (size_t)grp * a.gx;
    const float* gz_ptr = a.dz + (size_t)grp * a.gz;
    float* gw_ptr = a.dw + (size_t)grp * a.gw;
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

    // ---- DMA assignment.  B: lane -> 16-B chunk t % CB of pixel row t / CB + RB*p, p < PB.  A: chunk t % CA of row t / CA + RA*p
    const int chunk = t % CB;
    const int prow = t / CB;
    const int chunk_a = t % CA;
    const int prow_a = t / CA;
    // A (dz): column i0 + 4*chunk_a
    const int ca = i0 + 4 * chunk_a;
    const bool a_ok = ca < a.Cout;
    // B (x): column j = j0 + 4*chunk -> (tap, ci), fixed for the whole kernel
    const int jb = j0 + 4 * chunk;
    const bool b_ok = jb < ncols;
    const int tap = b_ok ? jb / a.Cin : 0;
    const int ci = b_ok ? jb - tap * a.Cin : 0;
    const int dy = tap / a.KW - a.pad;         // input row = yo*stride + dy
    const int dx = tap % a.KW - a.pad;
    // (image, y, x) of this thread's 4 staged OUTPUT pixels in the first slab; advanced by 32 pixels per slab
    int pb[PB], py[PB], px[PB];
    const int q32 = KS / a.W, r32 = KS % a.W;
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int m = slab_lo * KS + prow + RB * p;
        pb[p] = m / (a.H * a.W);
        const int idx = m - pb[p] * (a.H * a.W);
        py[p] = idx / a.W;
        px[p] = idx - py[p] * a.W;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gx_ptr), 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gz_ptr), 0, a.dz_bytes, 0x00020000);

    auto issue_slab = [&](int slab, int buf) {
        float* sa = smem + buf * STAGE + wave * 256;          // every DMA instruction of a wave fills 256 contiguous floats
        float* sb = smem + buf * STAGE + KS * TI + wave * 256;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = slab * KS + prow_a + RA * p;
            const unsigned va = (m < a.M && a_ok) ? (unsigned)(((size_t)m * a.ldz + ca) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (lds_ptr_t)(sa + p * RA * TI), 16, (int)va, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int m = slab * KS + prow + RB * p;
            const bool mok = m < a.M;
            const int yi = py[p] * a.stride + dy, xi = px[p] * a.stride + dx;
            const bool in = (unsigned)yi < (unsigned)a.Hi && (unsigned)xi < (unsigned)a.Wi;
            const unsigned vb = (mok && b_ok && in) ? (unsigned)(((size_t)((pb[p] * a.Hi + yi) * a.Wi + xi) * a.ldx + ci) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(sb + p * RB * TJ), 16, (int)vb, 0, 0, 0);
        }
        // advance the pixel coordinates to the next slab (+32 pixels, row-major, wraps at image boundaries)
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            px[p] += r32; py[p] += q32;
            if (px[p] >= a.W) { px[p] -= a.W; py[p] += 1; }
            while (py[p] >= a.H) { py[p] -= a.H; pb[p] += 1; }
        }
    };

    // ---- DZRAW loader state
    const int xi = grp >> 2, nu = grp & 3;                  // position of this group
    const bool two_i = xi == 1 || xi == 2, two_j = nu == 1 || nu == 2;       // rows / columns of the 2x2 gradient tile the position reads
    const int i_first = xi == 3 ? 1 : 0, j_first = nu == 3 ? 1 : 0;
    const unsigned pixb = (unsigned)a.ldz * 4u;
    int e_next[PA];                                          // decode-table entries of the thread's rows of the slab after the one being fetched
    f32x4 dpx[PA][2][2];
    auto dz_table = [&](int slab) {                          // (plain loads: 4 B per row per slab from an L2-resident table)
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = slab * KS + prow_a + RA * p;
            e_next[p] = m < a.M ? a.tile_pix[m] : 0x3fffffff;      // no such tile: a pixel index past every image
        }
    };
    auto dz_loads = [&]() {                                  // the pixels of the rows described by e_next (consumed: dz_table may run again afterwards)
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int e = e_next[p];
            const unsigned pix = (unsigned)e & 0x3fffffffu;
            const bool rowok = a_ok && pix != 0x3fffffffu;
            const bool y1 = (e & 0x40000000) != 0, x1 = e < 0;
            const unsigned base = pix * pixb + (unsigned)ca * 4u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if ((i == 1 && !two_i) || (j == 1 && !two_j)) continue;      // (uniform)
                    const int ii = i_first + i, jj = j_first + j;                // pixel (ii, jj) of the tile
                    const bool ok = rowok && (ii == 0 || y1) && (jj == 0 || x1);
                    dpx[p][i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)(ok ? base : OOB), (ii * a.zW + jj) * (int)pixb, 0));
                }
        }
    };
    auto dz_store = [&](int buf) {                           // dM chunk = column combination nu of the row combinations xi (wino_dz_kernel's order)
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            f32x4 sj[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && !two_j) continue;
                sj[j] = xi == 0 ? dpx[p][0][j] : (xi == 1 ? dpx[p][0][j] + dpx[p][1][j] : (xi == 2 ? dpx[p][0][j] - dpx[p][1][j] : -dpx[p][0][j]));
            }
            const f32x4 r = nu == 0 ? sj[0] : (nu == 1 ? sj[0] + sj[1] : (nu == 2 ? sj[0] - sj[1] : -sj[0]));
            *reinterpret_cast<f32x4*>(smem + buf * STAGE + p * RA * TI + t * 4) = r;
        }
    };

    f32x16 acc[IB][JB];
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fa = wm * WI + l31 + half * TI;              // + (2s)*TI + 32*block
    const int fb = KS * TI + wn * WJ + l31 + half * TJ;    // + (2s)*TJ + 32*block

    // The same slab fetch, split: voffsets of the NEXT slab first (VALU), then one DMA instruction at a time between the MFMAs of
    // the current slab (an LDS-DMA issue costs ~60 cycles, an fp32 32x32x2 MFMA keeps the pipe busy for 64): pieces go out in the
    // first half of the slab so that they have landed by the next vmcnt(0).
    unsigned nva[PA], nvb[PB];
    if (LIN) {          // offsets of slab slab_lo (the first compute_slab_spread issues slab_lo + 1: plan_slab adds one step before)
#pragma unroll
        for (int p = 0; p < PA; ++p) nva[p] = a_ok ? (unsigned)(((size_t)(slab_lo * KS + prow_a + RA * p) * a.ldz + ca) * 4) : OOB;
#pragma unroll
        for (int p = 0; p < PB; ++p) nvb[p] = b_ok ? (unsigned)(((size_t)(slab_lo * KS + prow + RB * p) * a.ldx + ci) * 4) : OOB;
    }
    const unsigned step_a = a_ok ? (unsigned)(KS * a.ldz * 4) : 0u, step_b = b_ok ? (unsigned)(KS * a.ldx * 4) : 0u;
    auto plan_slab = [&](int slab) {
        if (LIN) {
#pragma unroll
            for (int p = 0; p < PA; ++p) nva[p] += step_a;
#pragma unroll
            for (int p = 0; p < PB; ++p) nvb[p] += step_b;
            return;
        }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = slab * KS + prow_a + RA * p;
            nva[p] = (m < a.M && a_ok) ? (unsigned)(((size_t)m * a.ldz + ca) * 4) : OOB;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int m = slab * KS + prow + RB * p;
            const bool mok = m < a.M;
            const int yi = py[p] * a.stride + dy, xi = px[p] * a.stride + dx;
            const bool in = (unsigned)yi < (unsigned)a.Hi && (unsigned)xi < (unsigned)a.Wi;
            nvb[p] = (mok && b_ok && in) ? (unsigned)(((size_t)((pb[p] * a.Hi + yi) * a.Wi + xi) * a.ldx + ci) * 4) : OOB;
            px[p] += r32; py[p] += q32;
            if (px[p] >= a.W) { px[p] -= a.W; py[p] += 1; }
            while (py[p] >= a.H) { py[p] -= a.H; pb[p] += 1; }
        }
    };
    auto issue_piece = [&](int buf, int j) {
        if (j < PA) { if (!DZRAW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (lds_ptr_t)(smem + buf * STAGE + wave * 256 + j * RA * TI), 16, (int)nva[j], 0, 0, 0); }
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(smem + buf * STAGE + KS * TI + wave * 256 + (j - PA) * RB * TJ), 16, (int)nvb[j - PA], 0, 0, 0);
    };
    auto compute_slab_spread = [&](int buf, int nbuf, int next_slab = 0) {
        // volatile: the operand reads stay single ds_read_b32 with 16-bit immediate offsets.  Merged into ds_read2_b32 (8-bit offsets) every pair needs a v_add for
        // its base, and a vector instruction beside fp32 MFMAs costs its issue time (tools/mfma_peak.py): 0.5 of the kernel's 1.1 vector instructions per MFMA
        const volatile __attribute__((address_space(3))) float* sbuf = (const volatile __attribute__((address_space(3))) float*)(smem + buf * STAGE);
        constexpr int TOTAL = (KS / 2) * IB * JB, NPIECES = PA + PB;
        constexpr int EVERY = (TOTAL / 2) / NPIECES > 0 ? (TOTAL / 2) / NPIECES : 1;
        int cnt = 0;
        // operands of step s + 1 are read (in program order: the reads are volatile) in front of the MFMAs of step s: the LDS round trip runs under four MFMAs
        // instead of in front of them
        float av[2][IB], bv[2][JB];
#pragma unroll
        for (int i = 0; i < IB; ++i) av[0][i] = sbuf[fa + 32 * i];
#pragma unroll
        for (int j = 0; j < JB; ++j) bv[0][j] = sbuf[fb + 32 * j];
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) {
            if (s + 1 < KS / 2) {
#pragma unroll
                for (int i = 0; i < IB; ++i) av[(s + 1) & 1][i] = sbuf[fa + (s + 1) * 2 * TI + 32 * i];
#pragma unroll
                for (int j = 0; j < JB; ++j) bv[(s + 1) & 1][j] = sbuf[fb + (s + 1) * 2 * TJ + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i], bv[s & 1][j], acc[i][j], 0, 0, 0);
                    if (cnt % EVERY == 0 && cnt / EVERY < NPIECES) {
                        issue_piece(nbuf, cnt / EVERY);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (DZRAW) {
                        // the loader's share of the next slab: its pixels behind the first MFMA (their table entries came in during the
                        // previous slab), the entries of the slab after it behind the third, combination + store in the last quarter
                        if (cnt == 0) { dz_loads(); __builtin_amdgcn_sched_barrier(0); }
                        if (cnt == 2) { dz_table(next_slab + 1); __builtin_amdgcn_sched_barrier(0); }
                        if (cnt == (TOTAL * 3) / 4) { dz_store(nbuf); __builtin_amdgcn_sched_barrier(0); }
                    }
                    ++cnt;
                }
        }
#pragma unroll
        for (int j = (TOTAL + EVERY - 1) / EVERY; j < NPIECES; ++j) issue_piece(nbuf, j);
    };

    // MFMA blocks of this wave that lie completely outside the output (the last column tile of a 288-column problem holds 32
    // valid columns of 128): wave-uniform, so they are skipped instead of multiplied by zeros
    bool blk_ok[IB][JB];
    bool all_ok = true;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            blk_ok[i][j] = (i0 + wm * WI + i * 32) < a.Cout && (j0 + wn * WJ + j * 32) < ncols;
            all_ok = all_ok && blk_ok[i][j];
        }
    auto compute_slab_checked = [&](int buf) {
        const float* sbuf = smem + buf * STAGE;
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) {
            float av[IB], bv[JB];
#pragma unroll
            for (int i = 0; i < IB; ++i) av[i] = sbuf[fa + s * 2 * TI + 32 * i];
#pragma unroll
            for (int j = 0; j < JB; ++j) bv[j] = sbuf[fb + s * 2 * TJ + 32 * j];
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j)
                    if (blk_ok[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    auto compute_slab = [&](int buf) {
        const float* sbuf = smem + buf * STAGE;
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) {
            float av[IB], bv[JB];
#pragma unroll
            for (int i = 0; i < IB; ++i) av[i] = sbuf[fa + s * 2 * TI + 32 * i];
#pragma unroll
            for (int j = 0; j < JB; ++j) bv[j] = sbuf[fb + s * 2 * TJ + 32 * j];
#pragma unroll
            for (int i = 0; i < IB; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    // One loop per case (the case is fixed for the workgroup): with the branch INSIDE the loop the accumulators of the two paths
    // met in phi nodes and the compiler paid 64 v_accvgpr_mov per slab - one per MFMA - to bring them back to one register set
    // (PMC: VALU activity 6.3 against the forward GEMM's 3.8 even with linear offsets).
    if (DZRAW) {
        // first slab: table -> pixels -> operand chunk, in line; then the steady state above.  (Blocks outside the output multiply zeros.)
        dz_table(slab_lo);
        dz_loads();
        dz_table(slab_lo + 1);
        dz_store(0);
#pragma unroll
        for (int j = PA; j < PA + PB; ++j) issue_piece(0, j);
        for (int ks = 0; ks < nk - 1; ++ks) {
            plan_slab(slab_lo + ks + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute_slab_spread(ks & 1, (ks + 1) & 1, slab_lo + ks + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_slab((nk - 1) & 1);
    } else {
    issue_slab(slab_lo, 0);
    if (!all_ok) {                           // ragged tile: plain fetch, MFMAs of the outside blocks skipped
        for (int ks = 0; ks < nk - 1; ++ks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue_slab(slab_lo + ks + 1, (ks + 1) & 1);
            compute_slab_checked(ks & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_slab_checked((nk - 1) & 1);
    } else {
        for (int ks = 0; ks < nk - 1; ++ks) {
            if (Y2_WGRAD_SPREAD) plan_slab(slab_lo + ks + 1);      // VALU only: overlaps the tail of the previous slab's MFMAs
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (Y2_WGRAD_SPREAD) {
                compute_slab_spread(ks & 1, (ks + 1) & 1);
            } else {
                issue_slab(slab_lo + ks + 1, (ks + 1) & 1);
                compute_slab(ks & 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_slab((nk - 1) & 1);
    }
    }

    // ---- epilogue: lane -> column j (l31), register r -> row co = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int jbk = 0; jbk < JB; ++jbk) {
        const int j = j0 + wn * WJ + jbk * 32 + l31;
        if (j >= ncols) continue;
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = i0 + wm * WI + ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < a.Cout) {
                    if (a.partial != nullptr) {
                        a.partial[((size_t)(grp * a.splits + split) * a.Cout + co) * ncols + j] = acc[ib][jbk][r];
                        continue;
                    }
                    float* dst = gw_ptr + (size_t)co * ncols + j;
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
// K (pixel) splits of one weight-gradient launch: enough workgroups to fill the chip several times over, but >= 8 slabs each so that
// the atomic epilogue stays small.  1 = every output element is written by exactly one workgroup (plain stores, dw need not be zero).
// Column-tile width: 64 instead of 128 when that cuts the padded columns by >= 13 % (288 columns: 320 instead of 384; the 64-column
// GEMMs of the Winograd gradient of a Cin = 64 layer: 64 instead of 128); needs TI >= 64 (a wave owns at least one 32x32 block).
static int wgrad_tj(int ncols, int Cout) {
    if (Cout <= 32) return 128;
    const long long p128 = (long long)y2_cdiv(ncols, 128) * 128, p64 = (long long)y2_cdiv(ncols, 64) * 64;
    return (p128 * 100 >= p64 * 115) ? 64 : 128;
}

static int wgrad_splits(long long M, int ncols, int Cout, int groups, int* slabs_per_split) {
    const int TI = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    const int TJ = wgrad_tj(ncols, Cout);
    const long long tiles = (long long)y2_cdiv(Cout, TI) * y2_cdiv(ncols, TJ);
    const int slabs = y2_cdiv(M, KS);
    // workgroups per CU the split aims at = what is resident at once (2 of the 64 KB 128x128 tiles, 3 of 128x64, 4+ of the smaller
    // ones): one full wave of workgroups.  More splits only add atomic epilogues (measured, B=64: the 1x1 layers 0.130 -> 0.120 ms and
    // the 52x52 / 26x26 grouped reductions 0.43 -> 0.41 with 2 instead of 4 for the 128x128 tile; the 64x64-tile 208x208 layer
    // 1.02 -> 1.50 the other way)
    static const int fill_env = getenv("Y2_WGRAD_FILL") != nullptr ? atoi(getenv("Y2_WGRAD_FILL")) : 0;
    const int fill = fill_env > 0 ? fill_env : (TI * TJ >= 128 * 128 ? 2 : (TI * TJ >= 128 * 64 ? 3 : 4));
    int splits = y2_cdiv(fill * Y2_NUM_CU, tiles * groups);
    const int max_splits = slabs / 8 > 0 ? slabs / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int sps = y2_cdiv(slabs, splits);
    if (slabs_per_split != nullptr) *slabs_per_split = sps;
    return y2_cdiv(slabs, sps);
}

int y2_internal_wgrad_needs_zero(long long M, int Cin, int Cout, int groups) {
    return (wgrad_splits(M, Cin, Cout, groups, nullptr) > 1 && !y2_det.on) ? 1 : 0;      // 1x1 shape of the grouped Winograd reductions
}

struct WgradRawDz { const int32_t* tile_pix; int zW, ldz; unsigned long long bytes; };      // DZRAW: the raw gradient behind `dz` (see WgradArgs)

static int wgrad_impl(const float* x, const float* dz, float* dw, int B, int Hi, int Wi, int Cin, int ldx, int Cout, int ldz,
                      int ksize, int stride, int pad, y2_stream_t stream, int groups, long long gx, long long gz, long long gw, const WgradRawDz* raw = nullptr) {
    if (!x || !dz || !dw || B <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0) return Y2_EINVAL;
    if (ksize < 1 || ksize > 7 || stride < 1 || pad < 0) return Y2_ENOSUP;
    if (ldx < Cin || ldz < Cout) return Y2_EINVAL;
    if ((Cin & 3) || (ldx & 3) || (Cout & 3) || (ldz & 3) || !y2_aligned16(x) || !y2_aligned16(dz)) return Y2_EALIGN;
    const int H = (Hi + 2 * pad - ksize) / stride + 1, W = (Wi + 2 * pad - ksize) / stride + 1;   // dz spatial size
    if (H <= 0 || W <= 0) return Y2_EINVAL;
    const long long M = (long long)B * H * W;
    const unsigned long long xb = (unsigned long long)B * Hi * Wi * ldx * 4ull, zb = (unsigned long long)M * ldz * 4ull;
    if (M > 0x3fffffffLL || xb >= 0x7fffffffull || zb >= 0x7fffffffull) return Y2_ENOSUP;
    WgradArgs a;
    a.x = x; a.dz = dz; a.dw = dw;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.Cout = Cout; a.ldz = ldz; a.taps = ksize * ksize;
    a.Hi = Hi; a.Wi = Wi; a.stride = stride; a.pad = pad; a.KW = ksize;
    a.M = (int)M;
    a.groups = groups; a.gx = gx; a.gz = gz; a.gw = gw;
    const int TI = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
    a.tiles_i = y2_cdiv(Cout, TI);
    const int TJ = wgrad_tj(a.taps * Cin, Cout);
    a.tiles_j = y2_cdiv(a.taps * Cin, TJ);
    const int tiles = a.tiles_i * a.tiles_j;
    const int slabs = y2_cdiv(M, KS);
    a.splits = wgrad_splits(M, a.taps * Cin, Cout, groups, &a.slabs_per_split);
    a.x_bytes = (unsigned)xb; a.dz_bytes = (unsigned)zb;
    a.tile_pix = nullptr; a.zW = 0;
    if (raw != nullptr) {
        if (groups != 16 || raw->bytes >= 0x7fffffffull || (raw->ldz & 3) || raw->ldz < Cout) return Y2_ENOSUP;
        a.tile_pix = raw->tile_pix; a.zW = raw->zW; a.ldz = raw->ldz; a.dz_bytes = (unsigned)raw->bytes; a.gz = 0;
    }
    a.partial = nullptr;
    const long long wsize = (long long)Cout * a.taps * Cin;            // floats per group
    if (y2_det.on && a.splits > 1) {
        if ((size_t)groups * a.splits * wsize * sizeof(float) > y2_det.bytes) return Y2_EINVAL;     // deterministic scratch too small
        a.partial = y2_det.ws;
    }
    const long long grid = (long long)tiles * a.splits * groups;
    if (grid > 0x7fffffffLL) return Y2_EINVAL;
    hipStream_t s = y2_s(stream);
    const size_t lds = 2u * (size_t)(KS * TI + KS * TJ) * sizeof(float);
    const bool lin = ksize == 1 && stride == 1 && pad == 0;
#define Y2_WGRAD_LAUNCH_(TI_, WI_, TJ_, LIN_)                                                                             \
    do {                                                                                                                  \
        static Y2LdsAttr attr;                                                                                            \
        if (const int rc_ = attr.ensure(reinterpret_cast<const void*>(conv_wgrad_kernel<TI_, WI_, TJ_, LIN_>))) return rc_;     \
        Y2_LAUNCH(a.groups > 1 ? "conv_wgrad_kernel[grouped]" : "conv_wgrad_kernel", 2.0 * (double)a.M * a.Cout * a.taps * a.Cin * (a.groups > 1 ? a.groups : 1), (conv_wgrad_kernel<TI_, WI_, TJ_, LIN_>), dim3((unsigned)grid), dim3(NT), lds, s, a);                 \
    } while (0)
#define Y2_WGRAD_LAUNCH_RAW(TI_, WI_, TJ_)                                                                                \
    do {                                                                                                                  \
        static Y2LdsAttr attr;                                                                                            \
        if (const int rc_ = attr.ensure(reinterpret_cast<const void*>(conv_wgrad_kernel<TI_, WI_, TJ_, true, true>))) return rc_;     \
        Y2_LAUNCH("conv_wgrad_kernel[grouped,dz]", 2.0 * (double)a.M * a.Cout * a.taps * a.Cin * a.groups, (conv_wgrad_kernel<TI_, WI_, TJ_, true, true>), dim3((unsigned)grid), dim3(NT), lds, s, a);                 \
    } while (0)
#define Y2_WGRAD_LAUNCH(TI_, WI_, TJ_)                                                                                    \
    do {                                                                                                                  \
        if (raw != nullptr) Y2_WGRAD_LAUNCH_RAW(TI_, WI_, TJ_);                                                           \
        else if (lin) Y2_WGRAD_LAUNCH_(TI_, WI_, TJ_, true); else Y2_WGRAD_LAUNCH_(TI_, WI_, TJ_, false);                 \
    } while (0)
    if (TI == 32) Y2_WGRAD_LAUNCH(32, 1, 128);
    else if (TI == 64 && TJ == 64) Y2_WGRAD_LAUNCH(64, 2, 64);
    else if (TI == 64) Y2_WGRAD_LAUNCH(64, 1, 128);
    else if (TJ == 64) Y2_WGRAD_LAUNCH(128, 2, 64);
    else Y2_WGRAD_LAUNCH(128, 2, 128);
#undef Y2_WGRAD_LAUNCH
#undef Y2_WGRAD_LAUNCH_RAW
#undef Y2_WGRAD_LAUNCH_
    Y2_LAUNCH_CHECK();
    if (a.partial != nullptr) {       // fixed-order sum of the K-split partials; group g's result goes to dw + g*gw
        for (int g = 0; g < groups; ++g) {
            const int rc = y2_det_reduce_f32(a.partial + (size_t)g * a.splits * wsize, a.splits, wsize, wsize, nullptr, dw + (size_t)g * gw, s);
            if (rc != Y2_OK) return rc;
        }
    }
    return Y2_OK;
}

// ------------------------------------------------------------------------------------------------ first layer
// Weight gradient of 'layers1.0' (model/yolo2.py:78): x is the plugin's NCHW input (Cin <= 3), dz NHWC [B,H,W,Cout<=64].
// dW[co][c][ky][kx] = sum_pixels dz[pix][co] * x[c][y+ky-1][x+kx-1]: a 32(64) x 27 output with an 11-million-long
// reduction — one 32x32 MFMA block per 32 output channels, no LDS staging: lane (co = l&31, parity = l>>5) reads
// dz[pix][co] (128 B contiguous per pixel), lane (tap j = l&31) gathers its shifted input pixel (L1/L2 resident);
// each wave walks image rows, the 4 waves of a workgroup are reduced through LDS and added atomically to dW (which is
// in the state_dict layout already: column j = (c*3 + ky)*3 + kx).
namespace {

struct W0Args {
    const float* x; const float* dz; float* dw;
    float* partial;      // deterministic mode: [gridDim.x][Cout*K] per-workgroup sums (plain stores)
    int B, H, W, Cin, Cout, ldz, rows_total;
};

template <int IB>
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const W0Args a) {
    __shared__ float red[4][IB * 32 * 33];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int K = a.Cin * 9;
    const bool jok = l31 < K;
    const int c = jok ? l31 / 9 : 0;
    const int ky = jok ? (l31 % 9) / 3 : 0, kx = jok ? l31 % 3 : 0;
    f32x16 acc[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    for (int row = wid; row < a.rows_total; row += nw) {          // row = b*H + y
        const int b = row / a.H, y = row - b * a.H;
        const int yy = y + ky - 1;
        const bool yok = jok && (unsigned)yy < (unsigned)a.H;
        const float* xr = a.x + (((size_t)b * a.Cin + c) * a.H + (yok ? yy : 0)) * a.W;
        const float* zr = a.dz + (size_t)row * a.W * a.ldz;
        for (int x0 = 0; x0 < a.W; x0 += 16) {                   // 8 MFMA steps (16 pixels) per batch of loads
            float av[IB][8], bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int px = x0 + 2 * s + half;
                const bool pok = px < a.W;
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const int co = i * 32 + l31;
                    av[i][s] = (pok && co < a.Cout) ? zr[(size_t)px * a.ldz + co] : 0.f;
                }
                const int xx = px + kx - 1;
                bv[s] = (pok && yok && (unsigned)xx < (unsigned)a.W) ? xr[xx] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < IB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[s], acc[i], 0, 0, 0);
        }
    }
    // ---- reduce the 4 waves through LDS, then one atomic per output element per workgroup
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave][co * 33 + l31] = acc[i][r];
        }
    __syncthreads();
    for (int e = t; e < IB * 32 * 32; e += 256) {
        const int co = e >> 5, j = e & 31;
        if (co < a.Cout && j < K) {
            const float v = red[0][co * 33 + j] + red[1][co * 33 + j] + red[2][co * 33 + j] + red[3][co * 33 + j];
            if (a.partial != nullptr) a.partial[(size_t)blockIdx.x * a.Cout * K + (size_t)co * K + j] = v;
            else atomicAdd(a.dw + (size_t)co * K + j, v);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ first layer, dz on the fly
// The same reduction with the BatchNorm / LeakyReLU / 2x2 max-pool backward of y2_bn_act_bwd (pass 2) applied in the LOADER: the gradient
// of the first layer's raw convolution output (1.4 GB at batch 64, written by y2_bn_act_bwd and read exactly once, here) never exists.
// A wave walks image row PAIRS (one row of pooling windows); lane (co, half) owns column `half` of every window: it loads its two z
// values per window, gets the other column's activations from lane ^ 32 (ds_bpermute), finds the window's first maximum in nn.MaxPool2d's
// scan order, routes the pooled gradient, applies the LeakyReLU derivative and the BatchNorm backward with the pass-1 sums, and feeds the
// two rows' dz into two MFMA steps (K = the two pixels of a window row, as in conv0_wgrad_kernel).
namespace {

struct W0FArgs {
    const float* x; const float* z; const float* scale; const float* shift; const float* mean; const float* invstd; const float* gamma;
    const float* dy_pool; const double* sums; float* dw;
    float* partial;
    int B, H, W, Cin, Cout, ldz, ldp, pairs_total, has_bn;
    unsigned z_bytes, p_bytes;
    float slope;
    double n;
};

template <int IB>
__global__ __launch_bounds__(256) void conv0_wgrad_fused_kernel(const W0FArgs a) {
    __shared__ float red[4][IB * 32 * 33];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int K = a.Cin * 9;
    const bool jok = l31 < K;
    const int c = jok ? l31 / 9 : 0;
    const int ky = jok ? (l31 % 9) / 3 : 0, kx = jok ? l31 % 3 : 0;
    float sc[IB], sh[IB], mu[IB], is[IB], gs[IB], ma[IB], mb[IB];
    bool cok[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int co = i * 32 + l31;
        cok[i] = co < a.Cout;
        const int cc = cok[i] ? co : 0;
        sc[i] = a.scale ? a.scale[cc] : 1.f; sh[i] = a.shift ? a.shift[cc] : 0.f;
        mu[i] = a.has_bn ? a.mean[cc] : 0.f; is[i] = a.has_bn ? a.invstd[cc] : 1.f;
        if (a.has_bn == 1) { gs[i] = a.gamma[cc] * is[i]; ma[i] = (float)(a.sums[cc] / a.n); mb[i] = (float)(a.sums[a.Cout + cc] / a.n); }
        else if (a.has_bn == 2) { gs[i] = a.gamma[cc] * is[i]; ma[i] = 0.f; mb[i] = 0.f; }
        else { gs[i] = 1.f; ma[i] = 0.f; mb[i] = 0.f; }
    }
    f32x16 acc[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int Hh = a.H >> 1, Wh = a.W >> 1;
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.z), 0, a.z_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy_pool), 0, a.p_bytes, 0x00020000);
    for (int pair = wid; pair < a.pairs_total; pair += nw) {          // pair = b*(H/2) + r: image rows 2r, 2r+1
        const int b = pair / Hh, r = pair - b * Hh;
        const int y0 = 2 * r;
        const int yy0 = y0 + ky - 1, yy1 = yy0 + 1;
        const bool yok0 = jok && (unsigned)yy0 < (unsigned)a.H, yok1 = jok && (unsigned)yy1 < (unsigned)a.H;
        const float* xr0 = a.x + (((size_t)b * a.Cin + c) * a.H + (yok0 ? yy0 : 0)) * a.W;
        const float* xr1 = a.x + (((size_t)b * a.Cin + c) * a.H + (yok1 ? yy1 : 0)) * a.W;
        // z and dy_pool go through buffer descriptors: ONE per-lane byte offset per batch, the 16 pixels of the batch differ in the scalar
        // offset (pointer arithmetic per load held 100 more registers and halved the occupancy).  Host: W % 16 == 0, tensors < 4 GB.
        const unsigned zrow = (unsigned)(((size_t)(b * a.H + y0) * a.W + half) * a.ldz * 4u) + (unsigned)l31 * 4u;
        const unsigned prow = (unsigned)(((size_t)(b * Hh + r) * Wh) * a.ldp * 4u) + (unsigned)l31 * 4u;
        const unsigned zstep = 2u * (unsigned)a.ldz * 4u, z1off = (unsigned)a.W * (unsigned)a.ldz * 4u, pstep = (unsigned)a.ldp * 4u;
        for (int x0 = 0; x0 < a.W; x0 += 16) {                       // 8 windows = 16 pixels per row and batch of loads
            // phase 1: every load of the batch goes out before anything is consumed (40 loads in flight per wave)
            float z0[IB][8], z1[IB][8], dp[IB][8], bv0[8], bv1[8];
            const unsigned zb = zrow + (unsigned)x0 * (unsigned)a.ldz * 4u, pb = prow + (unsigned)(x0 >> 1) * pstep;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int px = x0 + 2 * s + half;
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    z0[i][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, (int)(zb + i * 128u), (int)(s * zstep), 0));
                    z1[i][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, (int)(zb + i * 128u), (int)(s * zstep + z1off), 0));
                    dp[i][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)(pb + i * 128u), (int)(s * pstep), 0));
                }
                const int xx = px + kx - 1;
                const bool xok = (unsigned)xx < (unsigned)a.W;
                bv0[s] = (xok && yok0) ? xr0[xx] : 0.f;
                bv1[s] = (xok && yok1) ? xr1[xx] : 0.f;
            }
            // phase 2: pooling-window arg-max, LeakyReLU derivative, BatchNorm backward -> dz of the two rows -> MFMA
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    const float u0 = z0[i][s] * sc[i] + sh[i], u1 = z1[i][s] * sc[i] + sh[i];
                    const float v0 = u0 > 0.f ? u0 : u0 * a.slope, v1 = u1 > 0.f ? u1 : u1 * a.slope;
                    const float o0 = __shfl_xor(v0, 32), o1 = __shfl_xor(v1, 32);          // the window's other column
                    // scan order q = 0: (row 0, col 0), 1: (row 0, col 1), 2: (row 1, col 0), 3: (row 1, col 1); first maximum wins
                    const float q0 = half ? o0 : v0, q1 = half ? v0 : o0, q2 = half ? o1 : v1, q3 = half ? v1 : o1;
                    float best = q0; int arg = 0;
                    if (q1 > best) { best = q1; arg = 1; }
                    if (q2 > best) { best = q2; arg = 2; }
                    if (q3 > best) { best = q3; arg = 3; }
                    const float g0 = (arg == half) ? dp[i][s] : 0.f, g1 = (arg == 2 + half) ? dp[i][s] : 0.f;
                    const float ge0 = g0 * (u0 > 0.f ? 1.f : a.slope), ge1 = g1 * (u1 > 0.f ? 1.f : a.slope);
                    const float zh0 = (z0[i][s] - mu[i]) * is[i], zh1 = (z1[i][s] - mu[i]) * is[i];
                    // (lanes of channels / pixels that do not exist loaded zeros and hold zero constants only where it matters: mask them)
                    const bool ok = (x0 + 2 * s + half) < a.W && cok[i];
                    const float d0 = ok ? gs[i] * (ge0 - ma[i] - zh0 * mb[i]) : 0.f;
                    const float d1 = ok ? gs[i] * (ge1 - ma[i] - zh1 * mb[i]) : 0.f;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, bv0[s], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, bv1[s], acc[i], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave][co * 33 + l31] = acc[i][r];
        }
    __syncthreads();
    for (int e = t; e < IB * 32 * 32; e += 256) {
        const int co = e >> 5, j = e & 31;
        if (co < a.Cout && j < K) {
            const float v = red[0][co * 33 + j] + red[1][co * 33 + j] + red[2][co * 33 + j] + red[3][co * 33 + j];
            if (a.partial != nullptr) a.partial[(size_t)blockIdx.x * a.Cout * K + (size_t)co * K + j] = v;
            else atomicAdd(a.dw + (size_t)co * K + j, v);
        }
    }
}

}  // namespace

extern "C" int y2_conv0_wgrad_fused(const float* x_nchw, const float* z, const float* scale, const float* shift, const float* mean, const float* invstd,
                                    const float* gamma, float slope, const float* dy_pool, int ldp, const double* sums, float* dw,
                                    int B, int H, int W, int Cin, int Cout, int ldz, int has_bn, y2_stream_t stream) {
    if (!x_nchw || !z || !dy_pool || !dw || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || ldz < Cout || ldp < Cout) return Y2_EINVAL;
    if (has_bn < 0 || has_bn > 2 || (has_bn && (!mean || !invstd || !gamma)) || (has_bn == 1 && !sums)) return Y2_EINVAL;
    if ((H & 1) || (W & 1)) return Y2_EINVAL;
    if (Cin < 1 || Cin > 3 || Cout > 64 || (W & 15)) return Y2_ENOSUP;
    const unsigned long long zb = (unsigned long long)B * H * W * ldz * 4ull, pb = (unsigned long long)B * (H / 2) * (W / 2) * ldp * 4ull;
    if (zb >= 0xffff0000ull) return Y2_ENOSUP;        // 32-bit buffer offsets
    W0FArgs a;
    a.z_bytes = (unsigned)zb; a.p_bytes = (unsigned)pb;
    a.x = x_nchw; a.z = z; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.dy_pool = dy_pool; a.sums = sums; a.dw = dw;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldz = ldz; a.ldp = ldp; a.pairs_total = B * (H / 2); a.has_bn = has_bn; a.slope = slope;
    a.n = (double)B * H * W;
    const int grid = a.pairs_total < 4 * 4 * Y2_NUM_CU ? y2_cdiv(a.pairs_total, 4) : 4 * Y2_NUM_CU;
    a.partial = nullptr;
    if (y2_det.on) {
        if ((size_t)grid * Cout * Cin * 9 * sizeof(float) > y2_det.bytes) return Y2_EINVAL;
        a.partial = y2_det.ws;
    }
    const double flops = 2.0 * (double)B * H * W * 9 * Cin * Cout;
    if (Cout <= 32) Y2_LAUNCH("conv0_wgrad_fused_kernel", flops, (conv0_wgrad_fused_kernel<1>), dim3(grid), dim3(256), 0, y2_s(stream), a);
    else Y2_LAUNCH("conv0_wgrad_fused_kernel", flops, (conv0_wgrad_fused_kernel<2>), dim3(grid), dim3(256), 0, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    if (a.partial != nullptr) return y2_det_reduce_f32(a.partial, grid, (long long)Cout * Cin * 9, (long long)Cout * Cin * 9, nullptr, dw, y2_s(stream));
    return Y2_OK;
}

extern "C" int y2_conv0_wgrad(const float* x_nchw, const float* dz, float* dw, int B, int H, int W, int Cin, int Cout, int ldz, y2_stream_t stream) {
    if (!x_nchw || !dz || !dw || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || ldz < Cout) return Y2_EINVAL;
    if (Cin < 1 || Cin > 3 || Cout > 64) return Y2_ENOSUP;
    W0Args a;
    a.x = x_nchw; a.dz = dz; a.dw = dw; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldz = ldz; a.rows_total = B * H;
    const int grid = a.rows_total < 4 * 4 * Y2_NUM_CU ? y2_cdiv(a.rows_total, 4) : 4 * Y2_NUM_CU;
    a.partial = nullptr;
    if (y2_det.on) {
        if ((size_t)grid * Cout * Cin * 9 * sizeof(float) > y2_det.bytes) return Y2_EINVAL;
        a.partial = y2_det.ws;
    }
    if (Cout <= 32) Y2_LAUNCH("conv0_wgrad_kernel", 2.0 * (double)B * H * W * 9 * Cin * Cout, (conv0_wgrad_kernel<1>), dim3(grid), dim3(256), 0, y2_s(stream), a);
    else Y2_LAUNCH("conv0_wgrad_kernel", 2.0 * (double)B * H * W * 9 * Cin * Cout, (conv0_wgrad_kernel<2>), dim3(grid), dim3(256), 0, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    if (a.partial != nullptr) return y2_det_reduce_f32(a.partial, grid, (long long)Cout * Cin * 9, (long long)Cout * Cin * 9, nullptr, dw, y2_s(stream));
    return Y2_OK;
}

extern "C" int y2_conv_wgrad_ex(const float* x, const float* dz, float* dw, int B, int Hi, int Wi, int Cin, int ldx, int Cout, int ldz,
                                int ksize, int stride, int pad, y2_stream_t stream) {
    return wgrad_impl(x, dz, dw, B, Hi, Wi, Cin, ldx, Cout, ldz, ksize, stride, pad, stream, 1, 0, 0, 0);
}

// `groups` independent 1x1 weight gradients dw_g[Cout][Cin] (+)= sum_m dz_g[m][Cout] * x_g[m][Cin] in one launch; dw pre-zeroed.
// Library-internal (wino.hip).
int y2_internal_wgrad_grouped(const float* x, const float* dz, float* dw, long long M, int Cin, int Cout, int groups, long long gx, long long gz,
                              long long gw, y2_stream_t stream) {
    if (M <= 0 || M > 0x7fffffffLL || groups < 1) return Y2_EINVAL;
    return wgrad_impl(x, dz, dw, 1, 1, (int)M, Cin, Cin, Cout, Cout, 1, 1, 0, stream, groups, gx, gz, gw);
}

// The same with the gradient operand built in the loader (DZRAW): dz_raw = the layer's output gradient [*, zW, ldz] of `dz_bytes` bytes,
// tile_pix = wino_tile_table_kernel's table of the M tiles.  Y2_ENOSUP: the caller falls back to wino_dz_kernel + y2_internal_wgrad_grouped.
int y2_internal_wgrad_grouped_dz(const float* v, const float* dz_raw, const int32_t* tile_pix, float* dw, long long M, int Cin, int Cout, int ldz, int zW,
                                 unsigned long long dz_bytes, long long gx, long long gw, y2_stream_t stream) {
    if (M <= 0 || M > 0x7fffffffLL || tile_pix == nullptr) return Y2_EINVAL;
    const WgradRawDz raw = {tile_pix, zW, ldz, dz_bytes};
    return wgrad_impl(v, dz_raw, dw, 1, 1, (int)M, Cin, Cin, Cout, Cout, 1, 1, 0, stream, 16, gx, 0, gw, &raw);
}

extern "C" int y2_conv_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int ldx, int Cout, int ldz,
                             int ksize, y2_stream_t stream) {
    if (ksize != 1 && ksize != 3) return Y2_ENOSUP;
    return y2_conv_wgrad_ex(x, dz, dw, B, H, W, Cin, ldx, Cout, ldz, ksize, 1, (ksize - 1) / 2, stream);
}
