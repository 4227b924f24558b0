// train.hip — training-side streaming kernels for gfx950: BatchNorm statistics/finalise, BN+LeakyReLU(+MaxPool) forward
// and backward, detection-head decode backward, and the fused region loss (matching, masks, 5 terms, gradient).
//
// All of these are HBM-bound (one or two passes over an activation tensor) or tiny (the loss touches ~100 KB per
// image): 16-B vector accesses along the NHWC channel axis, per-channel reductions staged through LDS atomics and
// finished with one fp64 global atomic per channel per workgroup, wave64 shuffles for scalar reductions.
// -ffp-contract=off (build.sh) keeps the IoU arithmetic identical to the reference's fp32 sequence.
#include "common.h"

namespace {

inline int stream_grid(long long total, int block, int cap_mult = 8) {
    long long g = (total + block - 1) / block;
    const long long cap = (long long)Y2_NUM_CU * cap_mult;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ BN finalise
// nn.BatchNorm2d(momentum=0.01, eps=1e-5) in training mode (model/yolo2.py:58): biased batch variance normalises,
// running stats take the unbiased variance.
__global__ void bn_finalize_kernel(const double* __restrict__ stats, double n, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   float* scale, float* shift, float* mean_out, float* invstd_out, int C, long long* num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;      // nn.BatchNorm2d's step counter, without a launch of its own
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < Y2_STATS_REPL; ++r) { s1 += stats[(size_t)r * 2 * C + c]; s2 += stats[(size_t)r * 2 * C + C + c]; }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float s = gamma[c] * invstd;
    scale[c] = s;
    shift[c] = beta[c] - (float)mean * s;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean != nullptr) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// ------------------------------------------------------------------------------------------------ BN + act forward
struct ActArgs {
    const float* z; const float* scale; const float* shift;   // scale/shift may be NULL (identity)
    float* y; float* y_pool;
    int B, H, W, C, ldz, ldy, coff, ldp, poff, out_mode;
    float slope;
    const float* res; int ldr;     // optional residual added before the activation (model/resnet.py:59,101)
    y2_fastdiv d_wo, d_ho;         // exact 32-bit division by the (pooled) output width / height
};

__device__ __forceinline__ float act1(float z, float sc, float sh, float slope) {
    const float u = z * sc + sh;
    return u > 0.f ? u : u * slope;
}

// The streaming BN/activation kernels run on a grid whose thread count is a multiple of the channel-group count Cg (host:
// act_grid), so a thread keeps ONE channel group for its whole loop (per-channel constants loaded once) and walks pixels
// p, p + step, ...; (b, yo, xo) come from two exact 32-bit fast divisions only where a pool / reorg mapping needs them.
// (The first version did five 64-bit divisions per element and was ALU-bound at ~4 TB/s.)
__device__ __forceinline__ void act_decode(uint32_t p, const y2_fastdiv& d_wo, const y2_fastdiv& d_ho, int Wo, int Ho, int& b, int& yo, int& xo) {
    const uint32_t r = y2_div(p, d_wo);
    xo = (int)(p - r * (uint32_t)Wo);
    const uint32_t bb = y2_div(r, d_ho);
    yo = (int)(r - bb * (uint32_t)Ho);
    b = (int)bb;
}

// one thread = CV channels (4 with 16-B accesses, or 1) of one pixel (POOL = false) or of one 2x2 window (POOL = true)
template <bool POOL, int CV>
__global__ void bn_act_fwd_kernel(const ActArgs a, long long total) {
    const int Cg = a.C / CV;
    const int Wo = POOL ? a.W / 2 : a.W, Ho = POOL ? a.H / 2 : a.H;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;            // multiple of Cg (host: act_grid)
    const int c = (int)(tid % Cg) * CV;
    const uint32_t npix = (uint32_t)(total / Cg), pstep = (uint32_t)(nthreads / Cg);
    float sc[CV], sh[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) { sc[e] = a.scale ? a.scale[c + e] : 1.f; sh[e] = a.shift ? a.shift[c + e] : 0.f; }
    const bool need_yx = POOL || a.out_mode == 1;
    for (uint32_t p = (uint32_t)(tid / Cg); p < npix; p += pstep) {           // p = (b*Ho + yo)*Wo + xo
        int b = 0, yo = 0, xo = 0;
        if (need_yx) act_decode(p, a.d_wo, a.d_ho, Wo, Ho, b, yo, xo);
        float pm[CV];
#pragma unroll
        for (int q = 0; q < (POOL ? 4 : 1); ++q) {
            const int yy = POOL ? 2 * yo + (q >> 1) : yo, xx = POOL ? 2 * xo + (q & 1) : xo;
            const long long pix = need_yx ? ((long long)b * a.H + yy) * a.W + xx : (long long)p;
            float v[CV];
            if (CV == 4) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(a.z + pix * a.ldz + c);
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if (a.res != nullptr) rr = *reinterpret_cast<const f32x4*>(a.res + pix * a.ldr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float u = zz[e] * sc[e] + sh[e] + rr[e]; v[e] = u > 0.f ? u : u * a.slope; }
            } else {
                const float u = a.z[pix * a.ldz + c] * sc[0] + sh[0] + (a.res != nullptr ? a.res[pix * a.ldr + c] : 0.f);
                v[0] = u > 0.f ? u : u * a.slope;
            }
            if (a.y != nullptr) {
                long long o;
                if (a.out_mode == 1) {   // reorg (model/yolo2.py:33-46): channel block ((y&1)*2 + (x&1)) of pixel (y/2, x/2)
                    o = (((long long)b * (a.H >> 1) + (yy >> 1)) * (a.W >> 1) + (xx >> 1)) * a.ldy + a.coff + ((yy & 1) * 2 + (xx & 1)) * a.C + c;
                } else {
                    o = pix * a.ldy + a.coff + c;
                }
                if (CV == 4) { f32x4 w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(a.y + o) = w; }
                else a.y[o] = v[0];
            }
#pragma unroll
            for (int e = 0; e < CV; ++e) pm[e] = (q == 0) ? v[e] : fmaxf(pm[e], v[e]);
        }
        if (POOL && a.y_pool != nullptr) {
            const long long o = (long long)p * a.ldp + a.poff + c;
            if (CV == 4) { f32x4 w = {pm[0], pm[1], pm[2], pm[3]}; *reinterpret_cast<f32x4*>(a.y_pool + o) = w; }
            else a.y_pool[o] = pm[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------ BN + act backward
// g = dL/du (u = BN output before LeakyReLU) from up to two gradient sources: dy_full (grad of the full-resolution
// activation; out_mode 1 = gathered through the reorg mapping) and dy_pool (grad of the 2x2 max-pooled activation, routed
// to the first maximal element of the window like nn.MaxPool2d).  Pass 1 reduces sum(g) and sum(g*zhat) per channel;
// pass 2 writes dz = gamma*invstd*(g - mean(g) - zhat*mean(g*zhat))   (dz = g when there is no BN).
struct ActBwdArgs {
    const float* z; const float* scale; const float* shift; const float* mean; const float* invstd; const float* gamma;
    const float* dy_full; const float* dy_pool;
    const float* dy_full2; int ld2;     // optional second full-resolution gradient source (fan-out of a residual network)
    const float* res; int ldr;          // residual that was added before the activation (decides the ReLU mask)
    float* dres; int lddr;              // optional output: gradient w.r.t. that residual (= gradient of the pre-activation sum)
    double* sums;      // [2C]: sum g, sum g*zhat
    float* partial;    // deterministic mode (pass 1): [threads / Cg][2C] per-thread sums, plain stores (added by det_reduce_rows_kernel)
    float* block_partial;   // pass 1: [blocks][2C] per-WORKGROUP sums, plain stores (added in a fixed order by bn_bwd_block_reduce_kernel) instead of 2C fp64 atomics per workgroup
    float* dz;         // [B,H,W,C] stride ldd
    int B, H, W, C, ldz, ldf, foff, fmode, ldp, poff, ldd;
    float slope;
    double n;
    int has_bn;
    y2_fastdiv d_wo, d_ho;
};

template <bool POOL, int CV, bool APPLY>
__global__ void bn_act_bwd_kernel(const ActBwdArgs a, long long total) {
    extern __shared__ float red[];   // [2*C] block partials (pass 1)
    const int Cg = a.C / CV;
    if (!APPLY) {
        for (int i = threadIdx.x; i < 2 * a.C; i += blockDim.x) red[i] = 0.f;
        __syncthreads();
    }
    const int Wo = POOL ? a.W / 2 : a.W, Ho = POOL ? a.H / 2 : a.H;
    // total threads is a multiple of Cg (host guarantees): every thread keeps the same channel group over its loop
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const int c = (int)(tid % Cg) * CV;
    float sc[CV], sh[CV], mu[CV], is[CV], gs[CV], ma[CV], mb[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) {
        sc[e] = a.scale ? a.scale[c + e] : 1.f; sh[e] = a.shift ? a.shift[c + e] : 0.f;
        mu[e] = a.has_bn ? a.mean[c + e] : 0.f; is[e] = a.has_bn ? a.invstd[c + e] : 1.f;
        if (APPLY && a.has_bn == 1) {
            gs[e] = a.gamma[c + e] * is[e];
            ma[e] = (float)(a.sums[c + e] / a.n);
            mb[e] = (float)(a.sums[a.C + c + e] / a.n);
        } else if (APPLY && a.has_bn == 2) {      // frozen statistics (eval-mode BatchNorm): z-hat does not depend on the batch
            gs[e] = a.gamma[c + e] * is[e]; ma[e] = 0.f; mb[e] = 0.f;
        } else { gs[e] = 1.f; ma[e] = 0.f; mb[e] = 0.f; }
    }
    float s1[CV], s2[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    const uint32_t npix = (uint32_t)(total / Cg), pstep = (uint32_t)(nthreads / Cg);
    const bool need_yx = POOL || a.fmode == 1;
    for (uint32_t p = (uint32_t)(tid / Cg); p < npix; p += pstep) {
        int b = 0, yo = 0, xo = 0;
        if (need_yx) act_decode(p, a.d_wo, a.d_ho, Wo, Ho, b, yo, xo);
        float zv[POOL ? 4 : 1][CV], yv[POOL ? 4 : 1][CV];
        long long pixs[POOL ? 4 : 1];
#pragma unroll
        for (int q = 0; q < (POOL ? 4 : 1); ++q) {
            const int yy = POOL ? 2 * yo + (q >> 1) : yo, xx = POOL ? 2 * xo + (q & 1) : xo;
            pixs[q] = need_yx ? ((long long)b * a.H + yy) * a.W + xx : (long long)p;
            if (CV == 4) {
                const f32x4 zz = *reinterpret_cast<const f32x4*>(a.z + pixs[q] * a.ldz + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) zv[q][e] = zz[e];
            } else zv[q][0] = a.z[pixs[q] * a.ldz + c];
#pragma unroll
            for (int e = 0; e < CV; ++e) yv[q][e] = zv[q][e] * sc[e] + sh[e];   // u (pre-activation)
            if (a.res != nullptr) {
                if (CV == 4) {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + pixs[q] * a.ldr + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[q][e] += rr[e];
                } else yv[q][0] += a.res[pixs[q] * a.ldr + c];
            }
        }
        // pooled gradient -> first maximal activated value in scan order
        float dp[CV];
        int arg[CV];
        if (POOL) {
            const long long o = (long long)p * a.ldp + a.poff + c;
            if (CV == 4) { const f32x4 d = *reinterpret_cast<const f32x4*>(a.dy_pool + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) dp[e] = d[e]; }
            else dp[0] = a.dy_pool[o];
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                float best = 0.f; int bi = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = yv[q][e];
                    const float v = u > 0.f ? u : u * a.slope;
                    if (q == 0 || v > best) { best = v; bi = q; }
                }
                arg[e] = bi;
            }
        }
#pragma unroll
        for (int q = 0; q < (POOL ? 4 : 1); ++q) {
            float g[CV];
#pragma unroll
            for (int e = 0; e < CV; ++e) g[e] = (POOL && arg[e] == q) ? dp[e] : 0.f;
            if (a.dy_full != nullptr) {
                const int yy = POOL ? 2 * yo + (q >> 1) : yo, xx = POOL ? 2 * xo + (q & 1) : xo;
                long long o;
                if (a.fmode == 1) o = (((long long)b * (a.H >> 1) + (yy >> 1)) * (a.W >> 1) + (xx >> 1)) * a.ldf + a.foff + ((yy & 1) * 2 + (xx & 1)) * a.C + c;
                else o = pixs[q] * a.ldf + a.foff + c;
                if (CV == 4) { const f32x4 d = *reinterpret_cast<const f32x4*>(a.dy_full + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] += d[e]; }
                else g[0] += a.dy_full[o];
            }
            if (a.dy_full2 != nullptr) {
                const long long o = pixs[q] * a.ld2 + c;
                if (CV == 4) { const f32x4 d = *reinterpret_cast<const f32x4*>(a.dy_full2 + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] += d[e]; }
                else g[0] += a.dy_full2[o];
            }
            float outv[CV], gev[CV];
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                const float ge = g[e] * (yv[q][e] > 0.f ? 1.f : a.slope);      // through LeakyReLU
                gev[e] = ge;
                const float zh = (zv[q][e] - mu[e]) * is[e];
                if (!APPLY) { s1[e] += ge; s2[e] += ge * zh; }
                else outv[e] = gs[e] * (ge - ma[e] - zh * mb[e]);
            }
            if (APPLY) {
                const long long o = pixs[q] * a.ldd + c;
                if (CV == 4) { f32x4 w = {outv[0], outv[1], outv[2], outv[3]}; *reinterpret_cast<f32x4*>(a.dz + o) = w; }
                else a.dz[o] = outv[0];
                if (a.dres != nullptr) {
                    const long long r = pixs[q] * a.lddr + c;
                    if (CV == 4) { f32x4 w = {gev[0], gev[1], gev[2], gev[3]}; *reinterpret_cast<f32x4*>(a.dres + r) = w; }
                    else a.dres[r] = gev[0];
                }
            }
        }
    }
    if (!APPLY && a.partial != nullptr) {
        float* row = a.partial + (size_t)(tid / Cg) * 2 * a.C;
#pragma unroll
        for (int e = 0; e < CV; ++e) { row[c + e] = s1[e]; row[a.C + c + e] = s2[e]; }
        return;
    }
    if (!APPLY) {
#pragma unroll
        for (int e = 0; e < CV; ++e) { atomicAdd(&red[c + e], s1[e]); atomicAdd(&red[a.C + c + e], s2[e]); }
        __syncthreads();
        if (a.block_partial != nullptr) {
            float* row = a.block_partial + (size_t)blockIdx.x * 2 * a.C;
            for (int i = threadIdx.x; i < 2 * a.C; i += blockDim.x) row[i] = red[i];
            return;
        }
        for (int i = threadIdx.x; i < 2 * a.C; i += blockDim.x) {
            const float v = red[i];
            if (v != 0.f) atomicAdd(a.sums + i, (double)v);
        }
    }
}

// sums[i] += sum over the workgroups' rows of part[row][i] (fp64).  Pass 1 used to end with 2C fp64 atomics PER WORKGROUP on the same 2C addresses (2048
// workgroups x 128 addresses for a 64-channel layer): ~30 microseconds of same-address serialisation per launch, which kept pass 1 at 2.2-4.1 TB/s next to
// pass 2's 5.2-5.5 (profiles/r04_train_b64_traffic_by_kernel.txt).  Block = 64 columns x 16 row lanes over one slice of the rows (gridDim.y slices: a
// single slice per column block was latency-bound at 30 microseconds, 128 dependent loads per thread); one fp64 atomic per column and slice.
__global__ __launch_bounds__(1024) void bn_bwd_block_reduce_kernel(const float* __restrict__ part, int rows, int n, double* __restrict__ sums) {
    __shared__ double acc[16][65];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    double s = 0.0;
    if (col < n)
        for (int r = r0 + lane; r < r1; r += 16) s += (double)part[(size_t)r * n + col];
    acc[lane][threadIdx.x & 63] = s;
    __syncthreads();
    if (lane == 0 && col < n) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += acc[l][threadIdx.x & 63];
        atomicAdd(sums + col, t);
    }
}

// ---- BatchNorm / LeakyReLU backward, second pass, fused with BOTH 4x4-tile Winograd transforms of the gradient it produces (round 6).
// A deep 3x3 layer whose weight gradient runs F(3x3,4x4) and whose data gradient runs F(4x4,3x3) consumes dz only through two transforms:
// V6 = B^T d B on 6x6 patches (the data gradient's input operand) and M6 = G d G^T on the 4x4 tile inside (the weight gradient's "filter" operand).
// The three-kernel form wrote dz (pass 2), read it twice and wrote 2 x 2.25 |dz|; here one thread = (tile, channel) forms dz at the 36 pixels of its
// patch from (z, dy) and the pass-1 sums - the arithmetic of bn_act_bwd_kernel<.., APPLY = true>, operation for operation - and stores both transforms,
// the operations of wino6_in_kernel<false> / <true> in their order: BIT-IDENTICAL operands, no dz tensor, two launches fewer.  Un-pooled gradient source
// with plain addressing only (dy_full, fmode 0); the pooled layers keep the three-kernel form.  Tile grid: Wino6Grid (common.h), like the kernels it replaces.
struct BnWino6Args {
    const float* z; const float* scale; const float* shift; const float* mean; const float* invstd; const float* gamma;
    const float* dy; const double* sums;
    float* v6; float* m6;          // [36][T][C] each; either may be NULL
    float* dz;                     // optional plain gradient [B,H,W,C] (pixel stride ldd): for a consumer that reads it untransformed
    int B, H, W, C, ldz, ldf, foff, ldd, has_bn;
    float slope;
    double n;
    int th, tw, T, tall, wide, gx;
    y2_fastdiv d_c, d_tt, d_tw, d_tall, d_wide;
};

__global__ __launch_bounds__(256) void bn_bwd_wino6_kernel(const BnWino6Args a) {
    constexpr float BT[6][6] = {{1.f, 1.5f, -2.f, -1.5f, 1.f, 0.f}, {0.f, -1.f, -2.5f, -0.5f, 1.f, 0.f}, {0.f, 1.f, 0.5f, -2.5f, 1.f, 0.f},
                                {0.f, -0.5f, -1.f, 0.5f, 1.f, 0.f}, {0.f, 2.f, -1.f, -2.f, 1.f, 0.f}, {0.f, 1.f, 1.5f, -2.f, -1.5f, 1.f}};
    constexpr float GM[6][4] = {{1.f, 0.f, 0.f, 0.f}, {-1.f / 3, -1.f / 3, -1.f / 3, -1.f / 3}, {1.f / 3, -1.f / 3, 1.f / 3, -1.f / 3},
                                {1.f / 15, 2.f / 15, 4.f / 15, 8.f / 15}, {-16.f / 15, 8.f / 15, -4.f / 15, 2.f / 15}, {0.f, 0.f, 0.f, 1.f}};
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = y2_div(idx, a.d_c);
    if (t >= (uint32_t)a.T) return;
    const int c = (int)(idx - t * (uint32_t)a.C);
    // per-channel constants of the apply pass (bn_act_bwd_kernel)
    const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
    const float mu = a.has_bn ? a.mean[c] : 0.f, is = a.has_bn ? a.invstd[c] : 1.f;
    float gs = 1.f, ma = 0.f, mb = 0.f;
    if (a.has_bn == 1) { gs = a.gamma[c] * is; ma = (float)(a.sums[c] / a.n); mb = (float)(a.sums[a.C + c] / a.n); }
    else if (a.has_bn == 2) { gs = a.gamma[c] * is; }
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
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    int colx[6], colb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
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
    // loads first, unconditionally (a pixel that does not exist reads pixel 0 and is discarded by the select below): as conditional loads every one of the
    // 36 pixels was a basic block of its own - load, wait, arithmetic - and the kernel ran at the latency of 36 dependent round trips (3.9 TB/s on the
    // 52x52 layer against 6 TB/s for the kernels it replaces)
    float d[6][6];
#pragma unroll
    for (int half = 0; half < 2; ++half) {          // two batches of 18 pixels: 36 loads in flight, half the address / value registers of one batch of 36
        uint32_t pixs[3][6];
        bool oks[3][6];
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            int yy = y0 + 3 * half + ii, by = 0;
            bool rowok = (unsigned)yy < (unsigned)a.H;
            if (a.tall) {
                rowok = yy >= 0;
                by = rowok ? (int)y2_div((uint32_t)yy, a.d_tall) : 0;
                yy -= by * a.tall;
                rowok = rowok && yy < a.H;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int bb = a.tall ? by * a.gx + colb[j] : b;
                const bool ok = rowok && colx[j] >= 0 && bb < a.B;
                oks[ii][j] = ok;
                pixs[ii][j] = ok ? (uint32_t)((bb * a.H + yy) * a.W + colx[j]) : 0u;
            }
        }
        float zs[3][6], gsrc[3][6];
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                zs[ii][j] = a.z[(size_t)pixs[ii][j] * a.ldz + c];
                gsrc[ii][j] = a.dy[(size_t)pixs[ii][j] * a.ldf + a.foff + c];
            }
#pragma unroll
        for (int ii = 0; ii < 3; ++ii)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int i = 3 * half + ii;
                const float zv = zs[ii][j];
                const float g = 0.f + gsrc[ii][j];
                const float yv = zv * sc + sh;
                const float ge = g * (yv > 0.f ? 1.f : a.slope);
                const float zh = (zv - mu) * is;
                const float v = oks[ii][j] ? gs * (ge - ma - zh * mb) : 0.f;
                if (a.dz != nullptr && oks[ii][j] && i >= 1 && i <= 4 && j >= 1 && j <= 4) a.dz[(size_t)pixs[ii][j] * a.ldd + c] = v;      // (the tile's own pixels: every pixel belongs to one tile)
                d[i][j] = v;
            }
    }
    const size_t plane = (size_t)a.T * a.C;
    if (a.v6 != nullptr) {          // wino6_in_kernel<false>
        float sm[6][6];
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (BT[u][i] != 0.f) v += BT[u][i] * d[i][j];
                sm[u][j] = v;
            }
        float* dst = a.v6 + (size_t)t * a.C + c;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v6 = 0; v6 < 6; ++v6) {
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    if (BT[v6][j] != 0.f) v += BT[v6][j] * sm[u][j];
                dst[(size_t)(6 * u + v6) * plane] = v;
            }
    }
    if (a.m6 != nullptr) {          // wino6_in_kernel<true> on the 4x4 tile inside the patch
        float sm[6][4];
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (GM[u][i] != 0.f) v += GM[u][i] * d[i + 1][j + 1];
                sm[u][j] = v;
            }
        float* dst = a.m6 + (size_t)t * a.C + c;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v6 = 0; v6 < 6; ++v6) {
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (GM[v6][j] != 0.f) v += GM[v6][j] * sm[u][j];
                dst[(size_t)(6 * u + v6) * plane] = v;
            }
    }
}

// nn.MaxPool2d(k, stride, padding) backward on NHWC, gather form: an input element receives the gradient of every window whose
// FIRST maximum (scan order, strict >, like ATen) it is; windows overlap when k > stride (model/resnet.py:114).
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ dy2, float* __restrict__ dx,
                                   int H, int W, int Ho, int Wo, int C, int ldx, int ldy, int lddx, int k, int stride, int pad, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long p = i / C;
        const int xx = (int)(p % W);
        const long long r = p / W;
        const int yy = (int)(r % H);
        const long long b = r / H;
        float g = 0.f;
        const int oy_hi = min(Ho - 1, (yy + pad) / stride), ox_hi = min(Wo - 1, (xx + pad) / stride);
        int oy_lo = (yy + pad - k + stride) / stride; if (yy + pad - k + 1 < 0) oy_lo = 0; if (oy_lo < 0) oy_lo = 0;
        int ox_lo = (xx + pad - k + stride) / stride; if (xx + pad - k + 1 < 0) ox_lo = 0; if (ox_lo < 0) ox_lo = 0;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float best = 0.f; int by = -1, bx = -1;
                for (int ky = 0; ky < k; ++ky) {
                    const int y2 = oy * stride - pad + ky;
                    if ((unsigned)y2 >= (unsigned)H) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        const int x2 = ox * stride - pad + kx;
                        if ((unsigned)x2 >= (unsigned)W) continue;
                        const float v = x[((b * H + y2) * W + x2) * ldx + c];
                        if (by < 0 || v > best) { best = v; by = y2; bx = x2; }
                    }
                }
                if (by == yy && bx == xx) {
                    const long long o = ((b * Ho + oy) * Wo + ox) * ldy + c;
                    g += dy[o];
                    if (dy2 != nullptr) g += dy2[o];
                }
            }
        dx[p * lddx + c] = g;
    }
}

// The same for 4 channels per thread (all strides multiples of 4, 16-byte aligned): one row of the image per blockIdx.y, 16-byte loads, no
// 64-bit divisions.  The stem pool of the ResNets at 608x608 (757 MB of input at batch 32) took 5.3 ms in the scalar form.
__global__ void maxpool_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ dy2, float* __restrict__ dx,
                                    int H, int W, int Ho, int Wo, int C4, int ldx, int ldy, int lddx, int k, int stride, int pad) {
    const int row = blockIdx.y;                    // b * H + yy
    const int b = row / H, yy = row - b * H;
    const int oy_hi = min(Ho - 1, (yy + pad) / stride);
    int oy_lo = (yy + pad - k + stride) / stride; if (yy + pad - k + 1 < 0 || oy_lo < 0) oy_lo = 0;
    const float* xb = x + (size_t)b * H * W * ldx;
    const float* dyb = dy + (size_t)b * Ho * Wo * ldy;
    const float* dy2b = dy2 ? dy2 + (size_t)b * Ho * Wo * ldy : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * C4; i += gridDim.x * blockDim.x) {
        const int xx = i / C4, c = (i - xx * C4) * 4;
        const int ox_hi = min(Wo - 1, (xx + pad) / stride);
        int ox_lo = (xx + pad - k + stride) / stride; if (xx + pad - k + 1 < 0 || ox_lo < 0) ox_lo = 0;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
                int4 at = make_int4(-1, -1, -1, -1);              // flat window position (ky * k + kx) of the first maximum per channel
                int mine = -1;
                for (int ky = 0; ky < k; ++ky) {
                    const int y2 = oy * stride - pad + ky;
                    if ((unsigned)y2 >= (unsigned)H) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        const int x2 = ox * stride - pad + kx;
                        if ((unsigned)x2 >= (unsigned)W) continue;
                        const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)y2 * W + x2) * ldx + c);
                        const int pos = ky * k + kx;
                        if (y2 == yy && x2 == xx) mine = pos;
                        if (at.x < 0 || v.x > best.x) { best.x = v.x; at.x = pos; }
                        if (at.y < 0 || v.y > best.y) { best.y = v.y; at.y = pos; }
                        if (at.z < 0 || v.z > best.z) { best.z = v.z; at.z = pos; }
                        if (at.w < 0 || v.w > best.w) { best.w = v.w; at.w = pos; }
                    }
                }
                if (at.x == mine || at.y == mine || at.z == mine || at.w == mine) {
                    const size_t o = ((size_t)oy * Wo + ox) * ldy + c;
                    float4 d = *reinterpret_cast<const float4*>(dyb + o);
                    if (dy2b != nullptr) { const float4 e = *reinterpret_cast<const float4*>(dy2b + o); d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
                    if (at.x == mine) g.x += d.x;
                    if (at.y == mine) g.y += d.y;
                    if (at.z == mine) g.z += d.z;
                    if (at.w == mine) g.w += d.w;
                }
            }
        *reinterpret_cast<float4*>(dx + ((size_t)row * W + xx) * lddx + c) = g;
    }
}

// nn.MaxPool2d(3, stride 2, padding 1) backward (the stem pool of model/resnet.py:114), LDS-tiled.  The gather kernels above re-derive the arg-max of up to four
// windows PER INPUT PIXEL - 36 16-byte loads and as many compares per thread (1.63 ms for 757 MB of input at batch 32, 608x608: a fifth of the HBM rate).  Here a
// workgroup owns an 8 x 32-pixel x 32-channel tile: the input rows / columns its windows touch ((8 + 3) x (32 + 3) pixels, 128 contiguous bytes per pixel) are
// staged in LDS once, every window's first maximum (scan order, strict >: ATen's) is found once and kept as a byte, and a pixel then looks up the at most four
// windows it belongs to.  x is read once, dy once, dx written once.
constexpr int MP_TH = 8, MP_TW = 32, MP_CH = 32;
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ dy2, float* __restrict__ dx,
                                                            int H, int W, int Ho, int Wo, int cgroups, int ldx, int ldy, int lddx) {
    constexpr int RH = MP_TH + 3, RW = MP_TW + 3;                 // staged rows / columns: y0 - 1 .. y0 + TH + 1
    constexpr int WH = MP_TH / 2 + 1, WW = MP_TW / 2 + 1;         // windows that contain a pixel of the tile
    __shared__ __attribute__((aligned(16))) float xs[RH * RW * MP_CH];
    __shared__ unsigned char arg[WH * WW * MP_CH];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * MP_TW, y0 = blockIdx.y * MP_TH;
    const int b = blockIdx.z / cgroups, c0 = (blockIdx.z - b * cgroups) * MP_CH;
    const float* xb = x + (size_t)b * H * W * ldx + c0;
    // ---- stage the input (pixels outside the image are never looked at below)
    for (int i = t; i < RH * RW * (MP_CH / 4); i += 256) {
        const int c4 = i % (MP_CH / 4), p = i / (MP_CH / 4);
        const int px = p % RW, py = p / RW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = *reinterpret_cast<const float4*>(xb + ((size_t)gy * W + gx) * ldx + 4 * c4);
        *reinterpret_cast<float4*>(xs + (py * RW + px) * MP_CH + 4 * c4) = v;
    }
    __syncthreads();
    // ---- first maximum of every window (oy, ox) = (y0 / 2 + wy, x0 / 2 + wx): input rows 2 oy - 1 .. 2 oy + 1 = staged rows 2 wy .. 2 wy + 2
    for (int i = t; i < WH * WW * MP_CH; i += 256) {
        const int c = i % MP_CH, w = i / MP_CH;
        const int wx = w % WW, wy = w / WW;
        const int oy = y0 / 2 + wy, ox = x0 / 2 + wx;
        int at = 255;
        if (oy < Ho && ox < Wo) {
            float best = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int gy = 2 * oy - 1 + ky;
                if ((unsigned)gy >= (unsigned)H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int gx = 2 * ox - 1 + kx;
                    if ((unsigned)gx >= (unsigned)W) continue;
                    const float v = xs[((2 * wy + ky) * RW + 2 * wx + kx) * MP_CH + c];
                    if (at == 255 || v > best) { best = v; at = ky * 3 + kx; }
                }
            }
        }
        arg[i] = (unsigned char)at;
    }
    __syncthreads();
    // ---- every pixel of the tile: the gradient of each window whose first maximum it is
    const float* dyb = dy + (size_t)b * Ho * Wo * ldy + c0;
    const float* dy2b = dy2 != nullptr ? dy2 + (size_t)b * Ho * Wo * ldy + c0 : nullptr;
    for (int i = t; i < MP_TH * MP_TW * (MP_CH / 4); i += 256) {
        const int c4 = i % (MP_CH / 4), p = i / (MP_CH / 4);
        const int lx = p % MP_TW, ly = p / MP_TW;
        const int yy = y0 + ly, xx = x0 + lx;
        if (yy >= H || xx >= W) continue;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        // windows: oy in {ceil((yy - 1) / 2) .. (yy + 1) / 2}: local wy = oy - y0 / 2
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = (yy + 1) / 2 - a;
            if (oy < 0 || oy >= Ho || 2 * oy - 1 > yy || 2 * oy + 1 < yy) continue;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int ox = (xx + 1) / 2 - bb;
                if (ox < 0 || ox >= Wo || 2 * ox - 1 > xx || 2 * ox + 1 < xx) continue;
                const int mine = (yy - (2 * oy - 1)) * 3 + (xx - (2 * ox - 1));
                const unsigned char* aw = arg + ((oy - y0 / 2) * WW + (ox - x0 / 2)) * MP_CH + 4 * c4;
                const bool m0 = aw[0] == mine, m1 = aw[1] == mine, m2 = aw[2] == mine, m3 = aw[3] == mine;
                if (m0 || m1 || m2 || m3) {
                    const size_t o = ((size_t)oy * Wo + ox) * ldy + 4 * c4;
                    float4 d = *reinterpret_cast<const float4*>(dyb + o);
                    if (dy2b != nullptr) { const float4 e = *reinterpret_cast<const float4*>(dy2b + o); d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w; }
                    if (m0) g[0] += d.x;
                    if (m1) g[1] += d.y;
                    if (m2) g[2] += d.z;
                    if (m3) g[3] += d.w;
                }
            }
        }
        *reinterpret_cast<float4*>(dx + ((size_t)(b * H + yy) * W + xx) * lddx + c0 + 4 * c4) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// per-channel column sums of a [M, C] (stride ld) matrix -> fp64 atomics (conv-bias gradient of blocks without BN)
__global__ void colsum_kernel(const float* __restrict__ x, long long M, int C, int ld, double* out) {
    extern __shared__ float red[];
    for (int i = threadIdx.x; i < C; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    const long long total = M * C;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    const int c = (int)(tid % C);
    float s = 0.f;
    for (long long i = tid; i < total; i += nth) s += x[(i / C) * ld + c];
    atomicAdd(&red[c], s);
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) if (red[i] != 0.f) atomicAdd(out + i, (double)red[i]);
}

__global__ void f64_to_f32_kernel(const double* s, float* d, int n, double mul) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)(s[i] * mul);
}

// ------------------------------------------------------------------------------------------------ decode backward
// Gradient of model.Inference's decode (model/__init__.py:122-135) w.r.t. the head image, from the gradients of
// iou / center_offset / size_norm / logits (the reference's loss detaches yx_min / yx_max, model/__init__.py:142).
__global__ void decode_bwd_kernel(const float* __restrict__ iou, const float* __restrict__ co, const float* __restrict__ d_iou,
                                  const float* __restrict__ d_co, const float* __restrict__ d_sn, const float* __restrict__ d_logits,
                                  float* __restrict__ dfeat, int total, int C) {
    const int E = 5 + C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)total * E; i += (long long)gridDim.x * blockDim.x) {
        const int box = (int)(i / E), k = (int)(i % E);
        float v;
        if (k == 0) { const float s = iou[box]; v = d_iou ? d_iou[box] * s * (1.f - s) : 0.f; }
        else if (k < 3) { const float s = co[2 * (size_t)box + k - 1]; v = d_co ? d_co[2 * (size_t)box + k - 1] * s * (1.f - s) : 0.f; }
        else if (k < 5) v = d_sn ? d_sn[2 * (size_t)box + k - 3] : 0.f;
        else v = d_logits ? d_logits[(size_t)box * C + k - 5] : 0.f;
        dfeat[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ region loss
__device__ __forceinline__ float iou_box(float ymin1, float xmin1, float ymax1, float xmax1,
                                         float ymin2, float xmin2, float ymax2, float xmax2) {
    // utils/iou/torch.py:126-153 operation order
    const float ih = fmaxf(fminf(ymax1, ymax2) - fmaxf(ymin1, ymin2), 0.f);
    const float iw = fmaxf(fminf(xmax1, xmax2) - fmaxf(xmin1, xmin2), 0.f);
    const float inter = ih * iw;
    const float a1 = (ymax1 - ymin1) * (xmax1 - xmin1);
    const float a2 = (ymax2 - ymin2) * (xmax2 - xmin2);
    const float uni = fmaxf((a1 + a2) - inter, 1.1920929e-07f);
    return inter / uni;
}

// model.iou_match (model/__init__.py:59-73): best IoU of every (cell, anchor) slot over the image's GT boxes; first max.
__global__ __launch_bounds__(256) void loss_match_kernel(const float* __restrict__ yx_min, const float* __restrict__ yx_max,
                                                         const float* __restrict__ gt_min, const float* __restrict__ gt_max,
                                                         int n, int N, float* best_iou, int32_t* best_idx, uint8_t* positive, double* sums) {
    __shared__ float g[4 * 256];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    // first kernel of the loss: it also clears what the following kernels accumulate into (no memset launches)
    if (blockIdx.x == 0 && b == 0 && threadIdx.x < 6) sums[threadIdx.x] = 0.0;
    if (i < n) positive[(size_t)b * n + i] = 0;
    float y0 = 0, x0 = 0, y1 = 0, x1 = 0;
    if (i < n) {
        const size_t o = ((size_t)b * n + i) * 2;
        y0 = yx_min[o]; x0 = yx_min[o + 1]; y1 = yx_max[o]; x1 = yx_max[o + 1];
    }
    float best = 0.f; int arg = 0; bool first = true;
    for (int j0 = 0; j0 < N; j0 += 256) {
        const int cnt = min(256, N - j0);
        __syncthreads();
        if (threadIdx.x < cnt) {
            const size_t o = ((size_t)b * N + j0 + threadIdx.x) * 2;
            g[4 * threadIdx.x] = gt_min[o]; g[4 * threadIdx.x + 1] = gt_min[o + 1];
            g[4 * threadIdx.x + 2] = gt_max[o]; g[4 * threadIdx.x + 3] = gt_max[o + 1];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float v = iou_box(y0, x0, y1, x1, g[4 * j], g[4 * j + 1], g[4 * j + 2], g[4 * j + 3]);
            if (first || v > best) { best = v; arg = j0 + j; first = false; }
        }
    }
    if (i < n) { best_iou[(size_t)b * n + i] = best; best_idx[(size_t)b * n + i] = arg; }
}

// model.fit_positive (model/__init__.py:76-95): each valid GT marks (cell of its centre, best-shape anchor).
__global__ void loss_positive_kernel(const float* __restrict__ gt_min, const float* __restrict__ gt_max, const float* __restrict__ anchors,
                                     int B, int N, int rows, int cols, int A, uint8_t* positive) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N;
    const float y0 = gt_min[2 * (size_t)i], x0 = gt_min[2 * (size_t)i + 1], y1 = gt_max[2 * (size_t)i], x1 = gt_max[2 * (size_t)i + 1];
    if (!(y0 < y1 && x0 < x1)) return;                            // :80
    const float cy = (y0 + y1) / 2.f, cx = (x0 + x1) / 2.f;
    const int ci = (int)floorf(cy), cj = (int)floorf(cx);
    if (ci < 0 || ci >= rows || cj < 0 || cj >= cols) return;      // the reference would raise an IndexError here
    float best = 0.f; int arg = 0;
    for (int k = 0; k < A; ++k) {
        const float ah = anchors[2 * k] / 2.f, aw = anchors[2 * k + 1] / 2.f;
        const float v = iou_box(y0 - cy, x0 - cx, y1 - cy, x1 - cx, -ah, -aw, ah, aw);   // :86
        if (k == 0 || v > best) { best = v; arg = k; }
    }
    positive[((size_t)b * rows * cols + (size_t)ci * cols + cj) * A + arg] = 1;
}

struct LossArgs {
    const float* iou; const float* co; const float* sn; const float* logits;      // predictions [B,n], [B,n,2], [B,n,2], [B,n,C]
    const float* best_iou; const int32_t* best_idx; const uint8_t* positive;
    const float* gt_min; const float* gt_max; const int64_t* gt_cls; const float* gt_onehot;   // one of gt_cls / gt_onehot (or neither when C == 0)
    const float* anchors;
    int B, n, N, A, C;
    float threshold;
};

// per-slot targets (model.fill_norm, model/__init__.py:98-103) from the best-IoU-matched GT
__device__ __forceinline__ void slot_targets(const LossArgs& a, int b, int i, float& t_cy, float& t_cx, float& t_h, float& t_w, int& gi) {
    gi = a.best_idx[(size_t)b * a.n + i];
    const size_t o = ((size_t)b * a.N + gi) * 2;
    const float y0 = a.gt_min[o], x0 = a.gt_min[o + 1], y1 = a.gt_max[o], x1 = a.gt_max[o + 1];
    const float cy = (y0 + y1) / 2.f, cx = (x0 + x1) / 2.f;
    t_cy = cy - floorf(cy); t_cx = cx - floorf(cx);
    const int anc = i % a.A;
    t_h = logf((y1 - y0) / a.anchors[2 * anc]);
    t_w = logf((x1 - x0) / a.anchors[2 * anc + 1]);
}

// sums[0..5] = foreground, background, center, size, cls (sum over positives), number of positives   (fp64 atomics)
__global__ __launch_bounds__(256) void loss_fwd_kernel(const LossArgs a, double* sums, float* blocksums) {      // blocksums: deterministic mode, [blocks][6]
    __shared__ float part[6][4];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v[6] = {0, 0, 0, 0, 0, 0};
    if (i < a.n) {
        const size_t s = (size_t)b * a.n + i;
        const float p_iou = a.iou[s];
        const float bi = a.best_iou[s];
        const bool pos = a.positive[s] != 0;
        const bool neg = !pos && bi < a.threshold;                              // model/__init__.py:145
        if (neg) v[1] = p_iou * p_iou;                                          // :152
        if (pos) {
            float t_cy, t_cx, t_h, t_w; int gi;
            slot_targets(a, b, i, t_cy, t_cx, t_h, t_w, gi);
            const float d0 = p_iou - bi; v[0] = d0 * d0;                        // :151
            const float dcy = a.co[2 * s] - t_cy, dcx = a.co[2 * s + 1] - t_cx; v[2] = dcy * dcy + dcx * dcx;   // :154
            const float dh = a.sn[2 * s] - t_h, dw = a.sn[2 * s + 1] - t_w; v[3] = dh * dh + dw * dw;           // :155
            v[5] = 1.f;
            if (a.C > 0) {
                const float* lg = a.logits + s * a.C;
                float mx = lg[0];
                for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, lg[c]);
                float sum = 0.f;
                for (int c = 0; c < a.C; ++c) sum += expf(lg[c] - mx);
                if (a.gt_cls != nullptr) {                                      // :162 cross entropy (summed here, mean taken by the caller)
                    const int k = (int)a.gt_cls[(size_t)b * a.N + gi];
                    v[4] = -(lg[k] - mx - logf(sum));
                } else if (a.gt_onehot != nullptr) {                            // :160 MSE on softmax
                    const float* oh = a.gt_onehot + ((size_t)b * a.N + gi) * a.C;
                    float acc = 0.f;
                    for (int c = 0; c < a.C; ++c) { const float d = expf(lg[c] - mx) / sum - oh[c]; acc += d * d; }
                    v[4] = acc;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) part[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const float x = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        if (blocksums != nullptr) blocksums[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = x;
        else if (x != 0.f) atomicAdd(sums + threadIdx.x, (double)x);
    }
}

// loss[k] = sums[k] / cnt (k < 4); cls: CE mean over positives then / cnt (model/__init__.py:162,164-166), MSE: sum / cnt
__global__ void loss_finalize_kernel(const double* sums, double cnt, int ce, float* out) {
    const int k = threadIdx.x;
    if (k < 4) out[k] = (float)(sums[k] / cnt);
    else if (k == 4) out[4] = ce ? (float)(sums[4] / sums[5] / cnt) : (float)(sums[4] / cnt);   // no positives -> nan, like F.cross_entropy on an empty batch
}

// gradient w.r.t. iou / center_offset / size_norm / logits; w[k] = upstream gradient of loss term k (train.py:348-349 hparams)
__global__ __launch_bounds__(256) void loss_bwd_kernel(const LossArgs a, const double* sums, double cnt, const float* __restrict__ w,
                                                       float* d_iou, float* d_co, float* d_sn, float* d_logits) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const size_t s = (size_t)b * a.n + i;
    const float inv = (float)(1.0 / cnt);
    const float p_iou = a.iou[s];
    const float bi = a.best_iou[s];
    const bool pos = a.positive[s] != 0;
    const bool neg = !pos && bi < a.threshold;
    float g_iou = 0.f, g_cy = 0.f, g_cx = 0.f, g_h = 0.f, g_w = 0.f;
    if (neg) g_iou = w[1] * 2.f * p_iou * inv;
    int gi = 0;
    if (pos) {
        float t_cy, t_cx, t_h, t_w;
        slot_targets(a, b, i, t_cy, t_cx, t_h, t_w, gi);
        g_iou = w[0] * 2.f * (p_iou - bi) * inv;
        g_cy = w[2] * 2.f * (a.co[2 * s] - t_cy) * inv;
        g_cx = w[2] * 2.f * (a.co[2 * s + 1] - t_cx) * inv;
        g_h = w[3] * 2.f * (a.sn[2 * s] - t_h) * inv;
        g_w = w[3] * 2.f * (a.sn[2 * s + 1] - t_w) * inv;
    }
    d_iou[s] = g_iou;
    d_co[2 * s] = g_cy; d_co[2 * s + 1] = g_cx;
    d_sn[2 * s] = g_h; d_sn[2 * s + 1] = g_w;
    if (a.C > 0 && d_logits != nullptr) {
        float* dl = d_logits + s * a.C;
        if (!pos || (a.gt_cls == nullptr && a.gt_onehot == nullptr)) {
            for (int c = 0; c < a.C; ++c) dl[c] = 0.f;
        } else {
            const float* lg = a.logits + s * a.C;
            float mx = lg[0];
            for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, lg[c]);
            float sum = 0.f;
            for (int c = 0; c < a.C; ++c) sum += expf(lg[c] - mx);
            if (a.gt_cls != nullptr) {
                const int k = (int)a.gt_cls[(size_t)b * a.N + gi];
                const float f = w[4] * inv / (float)sums[5];
                for (int c = 0; c < a.C; ++c) dl[c] = f * (expf(lg[c] - mx) / sum - (c == k ? 1.f : 0.f));
            } else {
                const float* oh = a.gt_onehot + ((size_t)b * a.N + gi) * a.C;
                const float f = w[4] * 2.f * inv;
                float dot = 0.f;
                for (int c = 0; c < a.C; ++c) { const float sm = expf(lg[c] - mx) / sum; dot += f * (sm - oh[c]) * sm; }
                for (int c = 0; c < a.C; ++c) { const float sm = expf(lg[c] - mx) / sum; dl[c] = sm * (f * (sm - oh[c]) - dot); }
            }
        }
    }
}

// grid for the BN/activation kernels: blocks of 256 threads, total thread count a multiple of Cg
inline int act_grid(long long total, int Cg) {
    int g = 256, r = Cg;
    while (r) { const int t = g % r; g = r; r = t; }
    const int unit = Cg / g;
    long long want = (total + 256 * 8 - 1) / (256 * 8);
    const long long cap = (long long)Y2_NUM_CU * 8;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)(((want + unit - 1) / unit) * unit);
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int y2_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps,
                              float* scale, float* shift, float* mean, float* invstd, int C, long long* num_batches_tracked, y2_stream_t stream) {
    if (!stats || !gamma || !beta || !scale || !shift || !mean || !invstd || C <= 0 || count <= 0) return Y2_EINVAL;
    Y2_LAUNCH("bn_finalize_kernel", 0.0, bn_finalize_kernel, dim3(y2_cdiv(C, 256)), dim3(256), 0, y2_s(stream), stats, count, gamma, beta, running_mean, running_var,
                       momentum, eps, scale, shift, mean, invstd, C, num_batches_tracked);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_bn_act_fwd_ex(const float* z, const float* scale, const float* shift, float slope, const float* residual, int ldr, float* y, float* y_pool,
                                int B, int H, int W, int C, int ldz, int ldy, int coff, int ldp, int poff, int out_mode, y2_stream_t stream) {
    if (!z || (!y && !y_pool) || B <= 0 || H <= 0 || W <= 0 || C <= 0) return Y2_EINVAL;
    const bool pool = y_pool != nullptr;
    if ((pool || out_mode == 1) && ((H & 1) || (W & 1))) return Y2_EINVAL;
    ActArgs a;
    a.z = z; a.scale = scale; a.shift = shift; a.y = y; a.y_pool = y_pool;
    a.B = B; a.H = H; a.W = W; a.C = C; a.ldz = ldz; a.ldy = ldy; a.coff = coff; a.ldp = ldp; a.poff = poff; a.out_mode = out_mode; a.slope = slope;
    a.res = residual; a.ldr = ldr;
    if (residual != nullptr && ldr < C) return Y2_EINVAL;
    const bool vec = (!residual || (!(ldr & 3) && y2_aligned16(residual))) && !(C & 3) && !(ldz & 3) && (!y || (!(ldy & 3) && !(coff & 3) && y2_aligned16(y))) && (!pool || (!(ldp & 3) && !(poff & 3) && y2_aligned16(y_pool))) && y2_aligned16(z);
    const long long pix = (long long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    if (pix >= 0x7fffffffLL || C > 8192) return Y2_ENOSUP;
    const long long total = pix * (vec ? C / 4 : C);
    const int grid = act_grid(total, vec ? C / 4 : C);
    a.d_wo = y2_make_fastdiv((uint32_t)(pool ? W / 2 : W)); a.d_ho = y2_make_fastdiv((uint32_t)(pool ? H / 2 : H));
    hipStream_t s = y2_s(stream);
    if (pool) { if (vec) Y2_LAUNCH("bn_act_fwd_kernel", 0.0, (bn_act_fwd_kernel<true, 4>), dim3(grid), dim3(256), 0, s, a, total); else Y2_LAUNCH("bn_act_fwd_kernel", 0.0, (bn_act_fwd_kernel<true, 1>), dim3(grid), dim3(256), 0, s, a, total); }
    else { if (vec) Y2_LAUNCH("bn_act_fwd_kernel", 0.0, (bn_act_fwd_kernel<false, 4>), dim3(grid), dim3(256), 0, s, a, total); else Y2_LAUNCH("bn_act_fwd_kernel", 0.0, (bn_act_fwd_kernel<false, 1>), dim3(grid), dim3(256), 0, s, a, total); }
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// park / park_floats: with dz == NULL (pass 1 only), a caller-owned buffer that nothing reads before the caller's own next kernel writes it: the
// per-workgroup partial sums are parked there instead of 2C fp64 atomics per workgroup (what a dense dz buffer is used for otherwise)
static int bn_act_bwd_impl(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                           float slope, const float* dy_full, int ldf, int foff, int fmode, const float* dy_pool, int ldp, int poff,
                           const float* dy_full2, int ld2, const float* residual, int ldr, float* dres, int lddr,
                           double* sums, float* dz, int ldd, int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream, float* park, long long park_floats) {
    if (!z || (!dy_full && !dy_pool) || !sums || B <= 0 || H <= 0 || W <= 0 || C <= 0) return Y2_EINVAL;      // dz == NULL: the sums only (pass 1)
    if (dz == nullptr && dres != nullptr) return Y2_EINVAL;
    if (has_bn < 0 || has_bn > 2 || (has_bn && (!mean || !invstd || !gamma))) return Y2_EINVAL;
    const bool pool = dy_pool != nullptr;
    if ((pool || fmode == 1) && ((H & 1) || (W & 1))) return Y2_EINVAL;
    if (C > 8192) return Y2_ENOSUP;
    ActBwdArgs a;
    a.z = z; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma;
    a.dy_full = dy_full; a.dy_pool = dy_pool; a.sums = sums; a.dz = dz;
    a.B = B; a.H = H; a.W = W; a.C = C; a.ldz = ldz; a.ldf = ldf; a.foff = foff; a.fmode = fmode; a.ldp = ldp; a.poff = poff; a.ldd = ldd;
    a.slope = slope; a.n = (double)B * H * W; a.has_bn = has_bn;
    a.dy_full2 = dy_full2; a.ld2 = ld2; a.res = residual; a.ldr = ldr; a.dres = dres; a.lddr = lddr;
    const bool vec = (!dy_full2 || (!(ld2 & 3) && y2_aligned16(dy_full2))) && (!residual || (!(ldr & 3) && y2_aligned16(residual))) && (!dres || (!(lddr & 3) && y2_aligned16(dres))) && !(C & 3) && !(ldz & 3) && (!dz || (!(ldd & 3) && y2_aligned16(dz))) && y2_aligned16(z) &&
                     (!dy_full || (!(ldf & 3) && !(foff & 3) && y2_aligned16(dy_full))) && (!pool || (!(ldp & 3) && !(poff & 3) && y2_aligned16(dy_pool)));
    const int Cg = vec ? C / 4 : C;
    const long long pix = (long long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    if (pix >= 0x7fffffffLL) return Y2_ENOSUP;
    const long long total = pix * Cg;
    const int grid = act_grid(total, Cg);     // total threads a multiple of Cg
    a.d_wo = y2_make_fastdiv((uint32_t)(pool ? W / 2 : W)); a.d_ho = y2_make_fastdiv((uint32_t)(pool ? H / 2 : H));
    const size_t lds = (size_t)2 * C * sizeof(float);
    hipStream_t s = y2_s(stream);
    a.partial = nullptr;
    // per-workgroup partial sums parked in the (not yet written) dz buffer when it is dense and large enough; else the atomics
    // (pass 1 of OTHER workgroups still reads its inputs while a finished one parks its sums: a dz that aliases any input - the element-wise
    //  in-place use dz == dy_full is legal for this entry point - keeps the atomics)
    const long long npix = (long long)B * H * W;
    auto overlaps = [&](const float* p, long long floats) {
        if (p == nullptr) return false;
        const char* lo = (const char*)dz; const char* hi = lo + (size_t)npix * ldd * sizeof(float);
        const char* q = (const char*)p;
        return q < hi && lo < q + (size_t)floats * sizeof(float);
    };
    const long long opix = pix;      // pooled gradient: one pixel per window
    const bool aliased = dz != nullptr && (overlaps(z, npix * ldz) || overlaps(dy_full, (fmode == 1 ? npix / 4 : npix) * (long long)ldf) || overlaps(dy_pool, opix * ldp) ||
                                           overlaps(dy_full2, npix * ld2) || overlaps(residual, npix * ldr) || overlaps(dres, npix * lddr));
    a.block_partial = (dz != nullptr && !aliased && !y2_det.on && ldd == C && grid > 16 && (long long)grid * 2 * C <= npix * ldd) ? dz : nullptr;
    if (dz == nullptr && park != nullptr && !y2_det.on && grid > 16 && (long long)grid * 2 * C <= park_floats && y2_aligned16(park)) a.block_partial = park;
    const long long prow = (long long)grid * 256 / Cg;            // rows of per-thread partials (grid * 256 is a multiple of Cg)
    if (y2_det.on) {
        if ((size_t)prow * 2 * C * sizeof(float) > y2_det.bytes || prow > 0x7fffffffLL) return Y2_EINVAL;
        a.partial = y2_det.ws;
    }
#define Y2_BWD(POOL, CV)                                                                                             \
    do {                                                                                                             \
        Y2_LAUNCH("bn_act_bwd_kernel", 0.0, (bn_act_bwd_kernel<POOL, CV, false>), dim3(grid), dim3(256), lds, s, a, total);            \
        if (a.partial != nullptr) {                                                                                  \
            const int rc_ = y2_det_reduce_f32(a.partial, (int)prow, (long long)2 * C, (long long)2 * C, sums, nullptr, s);            \
            if (rc_ != Y2_OK) return rc_;                                                                            \
        }                                                                                                            \
        if (a.block_partial != nullptr)                                                                              \
            Y2_LAUNCH("bn_bwd_block_reduce_kernel", 0.0, bn_bwd_block_reduce_kernel, dim3((unsigned)y2_cdiv(2 * C, 64), (unsigned)(grid >= 512 ? 16 : (grid >= 64 ? 4 : 1))), dim3(1024), 0, s, a.block_partial, grid, 2 * C, sums);            \
        if (dz != nullptr) Y2_LAUNCH("bn_act_bwd_kernel", 0.0, (bn_act_bwd_kernel<POOL, CV, true>), dim3(grid), dim3(256), 0, s, a, total);               \
    } while (0)
    if (pool) { if (vec) Y2_BWD(true, 4); else Y2_BWD(true, 1); }
    else { if (vec) Y2_BWD(false, 4); else Y2_BWD(false, 1); }
#undef Y2_BWD
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_bn_act_bwd_ex(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                                float slope, const float* dy_full, int ldf, int foff, int fmode, const float* dy_pool, int ldp, int poff,
                                const float* dy_full2, int ld2, const float* residual, int ldr, float* dres, int lddr,
                                double* sums, float* dz, int ldd, int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream) {
    return bn_act_bwd_impl(z, scale, shift, mean, invstd, gamma, slope, dy_full, ldf, foff, fmode, dy_pool, ldp, poff, dy_full2, ld2, residual, ldr, dres, lddr,
                           sums, dz, ldd, B, H, W, C, ldz, has_bn, stream, nullptr, 0);
}

extern "C" long long y2_wino6_tiles(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return Y2_EINVAL;
    return wino6_grid(B, H, W).T;
}

extern "C" int y2_bn_act_bwd_wino6(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                                   float slope, const float* dy_full, int ldf, int foff, double* sums, float* v6, float* m6, float* dz, int ldd,
                                   int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream) {
    if (!z || !dy_full || !sums || (v6 == nullptr && m6 == nullptr) || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ldz < C || ldf < foff + C) return Y2_EINVAL;
    if (dz != nullptr && ldd < C) return Y2_EINVAL;
    if (y2_det.on) return Y2_ENOSUP;
    const Wino6Grid g6 = wino6_grid(B, H, W);
    if (g6.T * C >= 0xffffffffLL || g6.T > 0x7fffffff || (long long)B * H * W >= 0x7fffffffLL) return Y2_ENOSUP;
    // pass 1 (the sums), its per-workgroup partials parked in the transform buffer this call writes afterwards
    float* park = v6 != nullptr ? v6 : m6;
    if (const int rc = bn_act_bwd_impl(z, scale, shift, mean, invstd, gamma, slope, dy_full, ldf, foff, 0, nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0,
                                       sums, nullptr, 0, B, H, W, C, ldz, has_bn, stream, park, 36LL * g6.T * C)) return rc;
    BnWino6Args a;
    a.z = z; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.dy = dy_full; a.sums = sums;
    a.v6 = v6; a.m6 = m6; a.dz = dz; a.ldd = ldd;
    a.B = B; a.H = H; a.W = W; a.C = C; a.ldz = ldz; a.ldf = ldf; a.foff = foff; a.has_bn = has_bn; a.slope = slope; a.n = (double)B * H * W;
    a.th = g6.th; a.tw = g6.tw; a.T = (int)g6.T; a.tall = g6.tall; a.wide = g6.wide; a.gx = g6.gx;
    a.d_c = y2_make_fastdiv((uint32_t)C); a.d_tt = y2_make_fastdiv((uint32_t)(g6.th * g6.tw)); a.d_tw = y2_make_fastdiv((uint32_t)g6.tw);
    a.d_tall = y2_make_fastdiv((uint32_t)(g6.tall > 0 ? g6.tall : 1)); a.d_wide = y2_make_fastdiv((uint32_t)(g6.wide > 0 ? g6.wide : 1));
    Y2_LAUNCH("bn_bwd_wino6_kernel", 0.0, bn_bwd_wino6_kernel, dim3((unsigned)y2_cdiv(g6.T * C, 256)), dim3(256), 0, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_bn_act_fwd(const float* z, const float* scale, const float* shift, float slope, float* y, float* y_pool,
                             int B, int H, int W, int C, int ldz, int ldy, int coff, int ldp, int poff, int out_mode, y2_stream_t stream) {
    return y2_bn_act_fwd_ex(z, scale, shift, slope, nullptr, 0, y, y_pool, B, H, W, C, ldz, ldy, coff, ldp, poff, out_mode, stream);
}

extern "C" int y2_bn_act_bwd(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                             float slope, const float* dy_full, int ldf, int foff, int fmode, const float* dy_pool, int ldp, int poff,
                             double* sums, float* dz, int ldd, int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream) {
    return y2_bn_act_bwd_ex(z, scale, shift, mean, invstd, gamma, slope, dy_full, ldf, foff, fmode, dy_pool, ldp, poff, nullptr, 0, nullptr, 0, nullptr, 0,
                            sums, dz, ldd, B, H, W, C, ldz, has_bn, stream);
}

extern "C" int y2_colsum(const float* x, long long M, int C, int ld, double* out, y2_stream_t stream) {
    if (!x || !out || M <= 0 || C <= 0 || ld < C || C > 16384) return Y2_EINVAL;
    // total threads multiple of C
    int g = 256, r = C;
    while (r) { const int t = g % r; g = r; r = t; }
    const int unit = C / g;
    long long want = (M * C + 256 * 16 - 1) / (256 * 16);
    const long long cap = (long long)Y2_NUM_CU * 4;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    const int grid = (int)(((want + unit - 1) / unit) * unit);
    Y2_LAUNCH("colsum_kernel", 0.0, colsum_kernel, dim3(grid), dim3(256), (size_t)C * sizeof(float), y2_s(stream), x, M, C, ld, out);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_f64_to_f32(const double* src, float* dst, int n, double mul, y2_stream_t stream) {
    if (!src || !dst || n <= 0) return Y2_EINVAL;
    Y2_LAUNCH("f64_to_f32_kernel", 0.0, f64_to_f32_kernel, dim3(y2_cdiv(n, 256)), dim3(256), 0, y2_s(stream), src, dst, n, mul);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_decode_bwd(const float* iou, const float* center_offset, const float* d_iou, const float* d_center_offset, const float* d_size_norm,
                             const float* d_logits, float* d_feature, int boxes, int C, y2_stream_t stream) {
    if (!iou || !center_offset || !d_feature || boxes <= 0 || C < 0) return Y2_EINVAL;
    const long long total = (long long)boxes * (5 + C);
    Y2_LAUNCH("decode_bwd_kernel", 0.0, decode_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), iou, center_offset, d_iou, d_center_offset, d_size_norm, d_logits,
                       d_feature, boxes, C);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_region_loss_fwd(const float* iou, const float* center_offset, const float* size_norm, const float* logits,
                                  const float* yx_min, const float* yx_max,
                                  const float* gt_yx_min, const float* gt_yx_max, const int64_t* gt_cls, const float* gt_onehot,
                                  const float* anchors, int B, int rows, int cols, int A, int C, int N, float threshold,
                                  float* best_iou, int32_t* best_idx, uint8_t* positive, double* sums, float* loss_out, y2_stream_t stream) {
    if (!iou || !center_offset || !size_norm || !yx_min || !yx_max || !gt_yx_min || !gt_yx_max || !anchors) return Y2_EINVAL;
    if (!best_iou || !best_idx || !positive || !sums || !loss_out) return Y2_EINVAL;
    if (B <= 0 || rows <= 0 || cols <= 0 || A <= 0 || C < 0 || N <= 0) return Y2_EINVAL;
    if (C > 0 && !logits) return Y2_EINVAL;
    const int n = rows * cols * A;
    hipStream_t s = y2_s(stream);
    Y2_LAUNCH("loss_match_kernel", 0.0, loss_match_kernel, dim3(y2_cdiv(n, 256), B), dim3(256), 0, s, yx_min, yx_max, gt_yx_min, gt_yx_max, n, N, best_iou, best_idx, positive, sums);
    Y2_LAUNCH("loss_positive_kernel", 0.0, loss_positive_kernel, dim3(y2_cdiv((long long)B * N, 256)), dim3(256), 0, s, gt_yx_min, gt_yx_max, anchors, B, N, rows, cols, A, positive);
    LossArgs a;
    a.iou = iou; a.co = center_offset; a.sn = size_norm; a.logits = logits; a.best_iou = best_iou; a.best_idx = best_idx; a.positive = positive;
    a.gt_min = gt_yx_min; a.gt_max = gt_yx_max; a.gt_cls = gt_cls; a.gt_onehot = gt_onehot; a.anchors = anchors;
    a.B = B; a.n = n; a.N = N; a.A = A; a.C = C; a.threshold = threshold;
    float* blocksums = nullptr;
    const long long nblk = (long long)y2_cdiv(n, 256) * B;
    if (y2_det.on) {
        if ((size_t)nblk * 6 * sizeof(float) > y2_det.bytes) return Y2_EINVAL;
        blocksums = y2_det.ws;
    }
    Y2_LAUNCH("loss_fwd_kernel", 0.0, loss_fwd_kernel, dim3(y2_cdiv(n, 256), B), dim3(256), 0, s, a, sums, blocksums);
    if (blocksums != nullptr) {
        const int rc = y2_det_reduce_f32(blocksums, (int)nblk, 6, 6, sums, nullptr, s);
        if (rc != Y2_OK) return rc;
    }
    Y2_LAUNCH("loss_finalize_kernel", 0.0, loss_finalize_kernel, dim3(1), dim3(64), 0, s, sums, (double)B * n, gt_cls != nullptr ? 1 : 0, loss_out);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_region_loss_bwd(const float* iou, const float* center_offset, const float* size_norm, const float* logits,
                                  const float* gt_yx_min, const float* gt_yx_max, const int64_t* gt_cls, const float* gt_onehot,
                                  const float* anchors, int B, int rows, int cols, int A, int C, int N, float threshold,
                                  const float* best_iou, const int32_t* best_idx, const uint8_t* positive, const double* sums, const float* weights,
                                  float* d_iou, float* d_center_offset, float* d_size_norm, float* d_logits, y2_stream_t stream) {
    if (!iou || !center_offset || !size_norm || !gt_yx_min || !gt_yx_max || !anchors || !best_iou || !best_idx || !positive || !sums || !weights) return Y2_EINVAL;
    if (!d_iou || !d_center_offset || !d_size_norm) return Y2_EINVAL;
    const int n = rows * cols * A;
    LossArgs a;
    a.iou = iou; a.co = center_offset; a.sn = size_norm; a.logits = logits; a.best_iou = best_iou; a.best_idx = best_idx; a.positive = positive;
    a.gt_min = gt_yx_min; a.gt_max = gt_yx_max; a.gt_cls = gt_cls; a.gt_onehot = gt_onehot; a.anchors = anchors;
    a.B = B; a.n = n; a.N = N; a.A = A; a.C = C; a.threshold = threshold;
    Y2_LAUNCH("loss_bwd_kernel", 0.0, loss_bwd_kernel, dim3(y2_cdiv(n, 256), B), dim3(256), 0, y2_s(stream), a, sums, (double)B * n, weights, d_iou, d_center_offset, d_size_norm, d_logits);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// Re-run the finalisation after the number of positives (sums[5]) has been all-reduced across data-parallel ranks.
extern "C" int y2_region_loss_finalize(const double* sums, double cnt, int cross_entropy, float* loss_out, y2_stream_t stream) {
    if (!sums || !loss_out || cnt <= 0) return Y2_EINVAL;
    Y2_LAUNCH("loss_finalize_kernel", 0.0, loss_finalize_kernel, dim3(1), dim3(64), 0, y2_s(stream), sums, cnt, cross_entropy, loss_out);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_maxpool_bwd(const float* x, const float* dy, const float* dy2, float* dx, int B, int H, int W, int C, int ldx, int ldy, int lddx,
                              int ksize, int stride, int pad, int pad_end, y2_stream_t stream) {
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || pad_end < 0) return Y2_EINVAL;
    const int Ho = (H + pad + pad_end - ksize) / stride + 1, Wo = (W + pad + pad_end - ksize) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return Y2_EINVAL;
    const long long total = (long long)B * H * W * C;
    const bool aligned = !(C & 3) && !(ldx & 3) && !(ldy & 3) && !(lddx & 3) && y2_aligned16(x) && y2_aligned16(dy) && y2_aligned16(dx) && (!dy2 || y2_aligned16(dy2));
    static const bool tiled_env = getenv("Y2_POOL_BWD_TILED") == nullptr || atoi(getenv("Y2_POOL_BWD_TILED")) != 0;
    if (tiled_env && aligned && ksize == 3 && stride == 2 && pad == 1 && !(C % MP_CH) && (long long)B * (C / MP_CH) < 65535 && (long long)H * W * ldx < 0x7fffffffLL) {
        const dim3 grid((unsigned)y2_cdiv(W, MP_TW), (unsigned)y2_cdiv(H, MP_TH), (unsigned)(B * (C / MP_CH)));
        Y2_LAUNCH("maxpool_bwd_kernel", 0.0, maxpool3s2_bwd_kernel, grid, dim3(256), 0, y2_s(stream), x, dy, dy2, dx, H, W, Ho, Wo, C / MP_CH, ldx, ldy, lddx);
        Y2_LAUNCH_CHECK();
        return Y2_OK;
    }
    if (!(C & 3) && !(ldx & 3) && !(ldy & 3) && !(lddx & 3) && y2_aligned16(x) && y2_aligned16(dy) && y2_aligned16(dx) && (!dy2 || y2_aligned16(dy2)) && (long long)B * H < 65535
        && (long long)H * W * ldx < 0x7fffffffLL) {
        const int per_row = W * (C / 4);
        Y2_LAUNCH("maxpool_bwd_kernel", 0.0, maxpool_bwd4_kernel, dim3((per_row + 255) / 256, B * H), dim3(256), 0, y2_s(stream), x, dy, dy2, dx, H, W, Ho, Wo, C / 4, ldx, ldy, lddx, ksize, stride, pad);
        Y2_LAUNCH_CHECK();
        return Y2_OK;
    }
    Y2_LAUNCH("maxpool_bwd_kernel", 0.0, maxpool_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), x, dy, dy2, dx, H, W, Ho, Wo, C, ldx, ldy, lddx, ksize, stride, pad, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
