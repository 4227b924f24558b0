// misc.hip — weight repacking, BatchNorm folding, stand-alone 2x2 max-pool (HBM-bound streaming kernels).
#include "common.h"

namespace {

// dst index -> src index gather; coalesced writes.
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int taps, int mode, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (mode == 0) {          // dst[co][tap][ci] = w[co][ci][tap]
            const int ci = (int)(i % Cin);
            const long long r = i / Cin;
            const int tap = (int)(r % taps);
            const int co = (int)(r / taps);
            dst[i] = w[((long long)co * Cin + ci) * taps + tap];
        } else if (mode == 1) {   // dst[ci][taps-1-tap][co] = w[co][ci][tap]   (dgrad: rotated filter, in/out swapped)
            const int co = (int)(i % Cout);
            const long long r = i / Cout;
            const int tr = (int)(r % taps);
            const int ci = (int)(r / taps);
            dst[i] = w[((long long)co * Cin + ci) * taps + (taps - 1 - tr)];
        } else {                  // mode 2: dst[co][ci][tap] = src[co][tap][ci]   (weight-gradient unpack)
            const int tap = (int)(i % taps);
            const long long r = i / taps;
            const int ci = (int)(r % Cin);
            const int co = (int)(r / Cin);
            dst[i] = w[((long long)co * taps + tap) * Cin + ci];
        }
    }
}

// Multi-tensor weight preparation: one launch for every layer's GEMM operand (y2_prep_weights).  The item table travels in the
// kernel arguments (like the optimizer's pointer table); block b works on the item whose [first_block, next first_block) range holds b.
struct PrepTable {
    y2_prep_item item[Y2_PREP_MAX_ITEMS];
    int first_block[Y2_PREP_MAX_ITEMS + 1];
    int count;
};

__global__ __launch_bounds__(256) void prep_weights_kernel(const PrepTable tb) {
    int it = 0;
    while (it + 1 < tb.count && (int)blockIdx.x >= tb.first_block[it + 1]) ++it;      // uniform scan, count <= 96
    const y2_prep_item& q = tb.item[it];
    const int nblk = tb.first_block[it + 1] - tb.first_block[it];
    const int blk = blockIdx.x - tb.first_block[it];
    const float* __restrict__ w = q.src;
    float* __restrict__ dst = q.dst;
    const int Cout = q.Cout, Cin = q.Cin, taps = q.ksize * q.ksize;
    if (q.mode == Y2_PREP_FPROP || q.mode == Y2_PREP_DGRAD) {
        // one thread per (co, ci): its k*k taps are contiguous in the state_dict layout (36 B for a 3x3 filter), and it scatters them
        // to the k*k tap planes of the GEMM layout.  FPROP walks (co, ci) with ci fastest (stores coalesced over ci), DGRAD with co
        // fastest (stores coalesced over co; its loads are 36-B segments Cin*36 B apart).  The element-per-thread form of
        // pack_weight_kernel reads 4 B every 36 B: this kernel runs once per training step over all 22 layers.
        const long long pairs = (long long)Cout * Cin;
        for (long long i = (long long)blk * 256 + threadIdx.x; i < pairs; i += (long long)nblk * 256) {
            int co, ci;
            if (q.mode == Y2_PREP_FPROP) { ci = (int)(i % Cin); co = (int)(i / Cin); }
            else { co = (int)(i % Cout); ci = (int)(i / Cout); }
            const float* src = w + ((long long)co * Cin + ci) * taps;
            if (taps == 9) {
                float g[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) g[t] = src[t];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (q.mode == Y2_PREP_FPROP) dst[((long long)co * 9 + t) * Cin + ci] = g[t];
                    else dst[((long long)ci * 9 + (8 - t)) * Cout + co] = g[t];
                }
            } else {
                for (int t = 0; t < taps; ++t) {
                    if (q.mode == Y2_PREP_FPROP) dst[((long long)co * taps + t) * Cin + ci] = src[t];
                    else dst[((long long)ci * taps + (taps - 1 - t)) * Cout + co] = src[t];
                }
            }
        }
        return;
    }
    // Winograd F(2x2,3x3) filter transform straight from the state_dict layout: U[p][n][k] = (G g G^T)[p].
    //   WINO_FPROP: n = co, k = ci, g[t] = w[co][ci][t];   WINO_DGRAD: n = ci, k = co, g[t] = w[co][ci][8 - t] (rotated, in/out swapped)
    // Same arithmetic as wino_weight_kernel on the packed operand: bit-identical U.
    const bool dg = q.mode == Y2_PREP_WINO_DGRAD || q.mode == Y2_PREP_WINO6_DGRAD;
    const int N = dg ? Cin : Cout, K = dg ? Cout : Cin;
    const long long total = (long long)N * K;
    if (q.mode == Y2_PREP_WINO6_DGRAD) {
        // F(4x4,3x3) filter operand of the data gradient, U6[p][n = ci][k = co] = (G6 g G6^T)[p] with g[t] = w[co][ci][8 - t]: the same
        // staging as the 2x2-tile item below, the arithmetic of wino6_weight_kernel (csrc/wino.hip) in its order: bit-identical to
        // y2_pack_weight(mode 1) + y2_wino6_weight, without the packed intermediate and without a launch per layer.
        constexpr float GM[6][3] = {{1.f, 0.f, 0.f}, {-1.f / 3, -1.f / 3, -1.f / 3}, {1.f / 3, -1.f / 3, 1.f / 3},
                                    {1.f / 15, 2.f / 15, 4.f / 15}, {-16.f / 15, 8.f / 15, -4.f / 15}, {0.f, 0.f, 1.f}};
        __shared__ float tile6[32][73];
        const int t = threadIdx.x;
        const int tiles_k = (Cout + 31) / 32, tiles_n = (Cin + 7) / 8;
        for (int tl = blk; tl < tiles_k * tiles_n; tl += nblk) {
            const int c0 = (tl % tiles_k) * 32, n0 = (tl / tiles_k) * 8;
            const int nn = min(8, Cin - n0);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int e = t + 256 * j, row = e / 72, col = e % 72;
                if (c0 + row < Cout && col < nn * 9) tile6[row][col] = w[((long long)(c0 + row) * Cin + n0) * 9 + col];
            }
            __syncthreads();
            const int k = c0 + (t & 31), n = n0 + (t >> 5);
            if (k >= Cout || n >= Cin) continue;
            float g[3][3];
#pragma unroll
            for (int x = 0; x < 9; ++x) g[x / 3][x % 3] = tile6[t & 31][(t >> 5) * 9 + 8 - x];
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
            float* d = dst + (long long)n * K + k;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    float v = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (GM[b][j] != 0.f) v += GM[b][j] * sm[a][j];
                    d[(long long)(6 * a + b) * total] = v;
                }
        }
        return;
    }
    if (dg) {
        // the data-gradient operand is the transpose (n = ci, k = co): with one thread per (ci, co) pair and co fastest the stores are
        // coalesced, but every lane's 36-byte load sits Cin * 36 bytes from its neighbour's (28 % of every line used; this item type was
        // half of the kernel's time).  So a workgroup stages a 32 co x 8 ci tile through LDS: rows of 288 contiguous bytes in, pairs out.
        __shared__ float tile[32][73];                  // (73: the 32 rows of a column read fall into 32 different banks)
        const int t = threadIdx.x;
        const int tiles_k = (Cout + 31) / 32, tiles_n = (Cin + 7) / 8;
        for (int tl = blk; tl < tiles_k * tiles_n; tl += nblk) {
            const int c0 = (tl % tiles_k) * 32, n0 = (tl / tiles_k) * 8;
            const int nn = min(8, Cin - n0);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int e = t + 256 * j, row = e / 72, col = e % 72;
                if (c0 + row < Cout && col < nn * 9) tile[row][col] = w[((long long)(c0 + row) * Cin + n0) * 9 + col];
            }
            __syncthreads();
            const int k = c0 + (t & 31), n = n0 + (t >> 5);
            if (k >= Cout || n >= Cin) continue;
            float g[3][3];
#pragma unroll
            for (int x = 0; x < 9; ++x) g[x / 3][x % 3] = tile[t & 31][(t >> 5) * 9 + 8 - x];
            float s[4][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                s[0][j] = g[0][j];
                s[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
                s[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
                s[3][j] = g[2][j];
            }
            float* d = dst + (long long)n * K + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[(4 * r + 0) * total] = s[r][0];
                d[(4 * r + 1) * total] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
                d[(4 * r + 2) * total] = 0.5f * (s[r][0] - s[r][1] + s[r][2]);
                d[(4 * r + 3) * total] = s[r][2];
            }
        }
        return;
    }
    for (long long i = (long long)blk * 256 + threadIdx.x; i < total; i += (long long)nblk * 256) {
        const int k = (int)(i % K);
        const int n = (int)(i / K);
        const float* src = dg ? w + ((long long)k * Cin + n) * 9 : w + ((long long)n * Cin + k) * 9;
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = dg ? src[8 - t] : src[t];
        float s[4][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            s[0][j] = g[0][j];
            s[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
            s[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
            s[3][j] = g[2][j];
        }
        float* d = dst + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d[(4 * r + 0) * total] = s[r][0];
            d[(4 * r + 1) * total] = 0.5f * (s[r][0] + s[r][1] + s[r][2]);
            d[(4 * r + 2) * total] = 0.5f * (s[r][0] - s[r][1] + s[r][2]);
            d[(4 * r + 3) * total] = s[r][2];
        }
    }
}

__global__ void bn_fold_kernel(const float* g, const float* b, const float* m, const float* v, float eps, float* scale, float* shift, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        const float s = g[i] / sqrtf(v[i] + eps);
        scale[i] = s;
        shift[i] = b[i] - m[i] * s;
    }
}

// one thread = 4 channels of one output pixel; 16-B loads/stores
__global__ void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int Ho, int Wo, int C4, int ldx, int ldy, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long p = i / C4;            // output pixel (b*Ho + yo)*Wo + xo
        const int xo = (int)(p % Wo);
        const long long r = p / Wo;            // b*Ho + yo
        const long long in_row = 2 * r;        // b*H + 2*yo  (H = 2*Ho)
        const float* s = x + ((in_row * (2 * Wo)) + 2 * xo) * (long long)ldx + 4 * c4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(s);
        const f32x4 b = *reinterpret_cast<const f32x4*>(s + ldx);
        const f32x4 c = *reinterpret_cast<const f32x4*>(s + (long long)2 * Wo * ldx);
        const f32x4 d = *reinterpret_cast<const f32x4*>(s + (long long)2 * Wo * ldx + ldx);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
        *reinterpret_cast<f32x4*>(y + p * ldy + 4 * c4) = o;
    }
}

// scalar variant for channel counts that are not a multiple of 4 (pruned checkpoints)
__global__ void maxpool2_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, int Ho, int Wo, int C, int ldx, int ldy, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long p = i / C;
        const int xo = (int)(p % Wo);
        const long long r = p / Wo;
        const float* s = x + ((2 * r * (2 * Wo)) + 2 * xo) * (long long)ldx + c;
        const long long row = (long long)2 * Wo * ldx;
        y[p * ldy + c] = fmaxf(fmaxf(s[0], s[ldx]), fmaxf(s[row], s[row + ldx]));
    }
}

// general max-pool on NHWC (nn.MaxPool2d(kernel_size=3, stride=2, padding=1), model/resnet.py:114): padded taps are skipped (-inf)
template <int CV>
__global__ void maxpool_gen_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho, int Wo, int C, int ldx, int ldy,
                                   int k, int stride, int pad, long long total) {
    const int Cg = C / CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cg) * CV;
        const long long p = i / Cg;
        const int xo = (int)(p % Wo);
        const long long r = p / Wo;
        const int yo = (int)(r % Ho);
        const long long b = r / Ho;
        float m[CV];
#pragma unroll
        for (int e = 0; e < CV; ++e) m[e] = -INFINITY;
        for (int ky = 0; ky < k; ++ky) {
            const int yy = yo * stride - pad + ky;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = xo * stride - pad + kx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const float* s = x + ((b * H + yy) * W + xx) * ldx + c;
                if (CV == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(s);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
                } else {
                    m[0] = fmaxf(m[0], s[0]);
                }
            }
        }
        if (CV == 4) { f32x4 o = {m[0], m[1], m[2], m[3]}; *reinterpret_cast<f32x4*>(y + p * ldy + c) = o; }
        else y[p * ldy + c] = m[0];
    }
}

// plugin input boundary for backbones whose first layer goes through the generic conv kernel: NCHW [B,C,H,W] -> NHWC with
// pixel stride ld >= C, padding channels zero-filled
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int ld, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        const long long p = i / ld;            // b*HW + pixel
        const long long b = p / HW;
        const long long pix = p - b * HW;
        y[i] = c < C ? x[(b * C + c) * HW + pix] : 0.f;
    }
}

// Multi-range utility pass (y2_multi): zero fills, fp64 -> fp32 conversions and copies of many small ranges in ONE launch (a training
// step zeroes ~30 accumulation buffers and hands out ~50 affine-parameter gradients; as separate fills / copies they were pure launch latency).
struct MultiTable {
    y2_multi_item item[Y2_MULTI_MAX_ITEMS];
    int first_block[Y2_MULTI_MAX_ITEMS + 1];
    int count;
};

__global__ __launch_bounds__(256) void multi_kernel(const MultiTable tb) {
    int it = 0;
    while (it + 1 < tb.count && (int)blockIdx.x >= tb.first_block[it + 1]) ++it;
    const y2_multi_item& q = tb.item[it];
    const long long nblk = tb.first_block[it + 1] - tb.first_block[it];
    const long long blk = blockIdx.x - tb.first_block[it];
    float* __restrict__ dst = q.dst;
    const long long n = q.n;
    if (q.op == Y2_MULTI_ZERO) {
        const bool v4 = !(reinterpret_cast<uintptr_t>(dst) & 15u);
        const long long n4 = v4 ? n / 4 : 0;
        for (long long i = blk * 256 + threadIdx.x; i < n4; i += nblk * 256) reinterpret_cast<f32x4*>(dst)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (long long i = n4 * 4 + blk * 256 + threadIdx.x; i < n; i += nblk * 256) dst[i] = 0.f;
    } else if (q.op == Y2_MULTI_F64_TO_F32) {
        const double* __restrict__ src = static_cast<const double*>(q.src);
        for (long long i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) dst[i] = (float)src[i];
    } else {
        const float* __restrict__ src = static_cast<const float*>(q.src);
        for (long long i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) dst[i] = src[i];
    }
}

// loss_total = sum_k hparam[k] * loss[k] (train.py:348-349) and its gradient, for the handful of loss terms: one wave
__global__ void small_dot_kernel(const float* __restrict__ v, const float* __restrict__ w, int n, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += v[i] * w[i];       // left to right, like Python's sum() over the dict
        out[0] = s;
    }
}
__global__ void small_scale_kernel(const float* __restrict__ g, const float* __restrict__ w, int n, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < n) out[i] = g[0] * w[i];
}

inline int stream_grid(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = (long long)Y2_NUM_CU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int y2_abi_version(void) { return 2; }
extern "C" const char* y2_build_info(void) { return "libyolo2_hip gfx950 fp32-mfma abi2"; }

extern "C" int y2_pack_weight(const float* w, float* dst, int Cout, int Cin, int ksize, int mode, y2_stream_t stream) {
    if (w == nullptr || dst == nullptr || Cout <= 0 || Cin <= 0 || ksize <= 0 || (mode != 0 && mode != 1)) return Y2_EINVAL;
    const long long total = (long long)Cout * Cin * ksize * ksize;
    Y2_LAUNCH("pack_weight_kernel", 0.0, pack_weight_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), w, dst, Cout, Cin, ksize * ksize, mode, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_prep_weights(const y2_prep_item* items, int32_t count, y2_stream_t stream) {
    if (items == nullptr || count < 0) return Y2_EINVAL;
    for (int lo = 0; lo < count; lo += Y2_PREP_MAX_ITEMS) {
        PrepTable tb;
        tb.count = count - lo < Y2_PREP_MAX_ITEMS ? count - lo : Y2_PREP_MAX_ITEMS;
        int blocks = 0;
        for (int i = 0; i < tb.count; ++i) {
            const y2_prep_item& q = items[lo + i];
            if (q.src == nullptr || q.dst == nullptr || q.Cout <= 0 || q.Cin <= 0 || q.ksize <= 0 || q.mode < Y2_PREP_FPROP || q.mode > Y2_PREP_WINO6_DGRAD) return Y2_EINVAL;
            if ((q.mode == Y2_PREP_WINO_FPROP || q.mode == Y2_PREP_WINO_DGRAD || q.mode == Y2_PREP_WINO6_DGRAD) && q.ksize != 3) return Y2_ENOSUP;
            const long long n = (long long)q.Cout * q.Cin;          // one thread per (co, ci) pair in every mode
            long long nb = (n + 511) / 512;            // ~2 pairs per thread
            if (nb < 1) nb = 1;
            if (nb > 2048) nb = 2048;
            tb.item[i] = q;
            tb.first_block[i] = blocks;
            blocks += (int)nb;
        }
        tb.first_block[tb.count] = blocks;
        if (blocks > 0) Y2_LAUNCH("prep_weights_kernel", 0.0, prep_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, y2_s(stream), tb);
        Y2_LAUNCH_CHECK();
    }
    return Y2_OK;
}

extern "C" int y2_multi(const y2_multi_item* items, int32_t count, y2_stream_t stream) {
    if (items == nullptr || count < 0) return Y2_EINVAL;
    for (int lo = 0; lo < count; lo += Y2_MULTI_MAX_ITEMS) {
        MultiTable tb;
        tb.count = count - lo < Y2_MULTI_MAX_ITEMS ? count - lo : Y2_MULTI_MAX_ITEMS;
        int blocks = 0;
        for (int i = 0; i < tb.count; ++i) {
            const y2_multi_item& q = items[lo + i];
            if (q.dst == nullptr || q.n < 0 || q.op < Y2_MULTI_ZERO || q.op > Y2_MULTI_COPY || (q.op != Y2_MULTI_ZERO && q.src == nullptr)) return Y2_EINVAL;
            long long nb = (q.n + 4095) / 4096;            // ~16 elements per thread
            if (nb < 1) nb = 1;
            if (nb > 1024) nb = 1024;
            tb.item[i] = q;
            tb.first_block[i] = blocks;
            blocks += (int)nb;
        }
        tb.first_block[tb.count] = blocks;
        if (blocks > 0) Y2_LAUNCH("multi_kernel", 0.0, multi_kernel, dim3((unsigned)blocks), dim3(256), 0, y2_s(stream), tb);
        Y2_LAUNCH_CHECK();
    }
    return Y2_OK;
}

extern "C" int y2_small_dot(const float* v, const float* w, int32_t n, float* out, y2_stream_t stream) {
    if (!v || !w || !out || n <= 0 || n > 64) return Y2_EINVAL;
    Y2_LAUNCH("small_dot_kernel", 0.0, small_dot_kernel, dim3(1), dim3(64), 0, y2_s(stream), v, w, n, out);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_small_scale(const float* g, const float* w, int32_t n, float* out, y2_stream_t stream) {
    if (!g || !w || !out || n <= 0 || n > 64) return Y2_EINVAL;
    Y2_LAUNCH("small_scale_kernel", 0.0, small_scale_kernel, dim3(1), dim3(64), 0, y2_s(stream), g, w, n, out);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_unpack_weight_grad(const float* src, float* dst, int Cout, int Cin, int ksize, y2_stream_t stream) {
    if (src == nullptr || dst == nullptr || Cout <= 0 || Cin <= 0 || ksize <= 0) return Y2_EINVAL;
    const long long total = (long long)Cout * Cin * ksize * ksize;
    Y2_LAUNCH("pack_weight_kernel", 0.0, pack_weight_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), src, dst, Cout, Cin, ksize * ksize, 2, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* shift, int C, y2_stream_t stream) {
    if (!gamma || !beta || !mean || !var || !scale || !shift || C <= 0) return Y2_EINVAL;
    Y2_LAUNCH("bn_fold_kernel", 0.0, bn_fold_kernel, dim3(y2_cdiv(C, 256)), dim3(256), 0, y2_s(stream), gamma, beta, mean, var, eps, scale, shift, C);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, y2_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return Y2_EINVAL;
    if ((H & 1) || (W & 1) || ldx < C || ldy < C) return Y2_EINVAL;
    if ((C & 3) || (ldx & 3) || (ldy & 3) || !y2_aligned16(x) || !y2_aligned16(y)) {
        const long long tot = (long long)B * (H / 2) * (W / 2) * C;
        Y2_LAUNCH("maxpool2_scalar_kernel", 0.0, maxpool2_scalar_kernel, dim3(stream_grid(tot, 256)), dim3(256), 0, y2_s(stream), x, y, H / 2, W / 2, C, ldx, ldy, tot);
        Y2_LAUNCH_CHECK();
        return Y2_OK;
    }
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    Y2_LAUNCH("maxpool2_kernel", 0.0, maxpool2_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), x, y, H / 2, W / 2, C / 4, ldx, ldy, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_maxpool_fwd(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, int ksize, int stride, int pad, int pad_end, y2_stream_t stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ksize <= 0 || stride <= 0 || pad < 0 || pad_end < 0 || ldx < C || ldy < C) return Y2_EINVAL;
    const int Ho = (H + pad + pad_end - ksize) / stride + 1, Wo = (W + pad + pad_end - ksize) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return Y2_EINVAL;
    const bool vec = !(C & 3) && !(ldx & 3) && !(ldy & 3) && y2_aligned16(x) && y2_aligned16(y);
    const long long total = (long long)B * Ho * Wo * (vec ? C / 4 : C);
    if (vec) Y2_LAUNCH("maxpool_gen_kernel", 0.0, (maxpool_gen_kernel<4>), dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), x, y, H, W, Ho, Wo, C, ldx, ldy, ksize, stride, pad, total);
    else Y2_LAUNCH("maxpool_gen_kernel", 0.0, (maxpool_gen_kernel<1>), dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), x, y, H, W, Ho, Wo, C, ldx, ldy, ksize, stride, pad, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, int ld, y2_stream_t stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ld < C) return Y2_EINVAL;
    const long long total = (long long)B * H * W * ld;
    Y2_LAUNCH("nchw_to_nhwc_kernel", 0.0, nchw_to_nhwc_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), x, y, C, H * W, ld, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
