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
#include "common.h"

namespace {

struct WinoInArgs {
    const float* x;
    float* v;
    int B, H, W, Cin, ldx, th, tw, T, c4n;
    y2_fastdiv d_c4, d_tt, d_tw;
};

__global__ __launch_bounds__(256) void wino_input_kernel(const WinoInArgs a) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = y2_div(idx, a.d_c4);
    if (t >= (uint32_t)a.T) return;
    const int c4 = (int)(idx - t * (uint32_t)a.c4n);
    const int b = (int)y2_div(t, a.d_tt);
    const int r = (int)t - b * a.th * a.tw;
    const int ty = (int)y2_div((uint32_t)r, a.d_tw);
    const int tx = r - ty * a.tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
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

// dw_packed[co][tap][ci] = (G^T dU G)[tap]
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
        float* dst = dw + ((size_t)co * 9 + 3 * i) * Cin + ci;
        dst[0] = s[i][0] + 0.5f * (s[i][1] + s[i][2]);
        dst[Cin] = 0.5f * (s[i][1] - s[i][2]);
        dst[2 * (size_t)Cin] = 0.5f * (s[i][1] + s[i][2]) + s[i][3];
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

extern "C" int y2_wino_weight(const float* w_packed, float* u, int32_t Cout, int32_t Cin, y2_stream_t stream) {
    if (w_packed == nullptr || u == nullptr || Cout <= 0 || Cin <= 0) return Y2_EINVAL;
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, y2_s(stream), w_packed, u, Cout, Cin);
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
    const int th = (p->H + 1) / 2, tw = (p->W + 1) / 2;
    // Batch chunks bound the workspace (V = 4x the input, M = 4x the output of a chunk); see wino_chunk_bytes().
    const size_t img_bytes = (size_t)16 * th * tw * ((size_t)p->Cin + p->Cout) * sizeof(float);
    int cb = (int)(wino_chunk_bytes() / (img_bytes > 0 ? img_bytes : 1));
    if (cb < 1) cb = 1;
    if (cb > p->B) cb = p->B;
    const int nchunks = y2_cdiv(p->B, cb);
    cb = y2_cdiv(p->B, nchunks);                     // equal chunks
    const long long T = (long long)cb * th * tw;    // tiles of a full chunk
    if (T * (p->Cin / 4) >= 0xffffffffLL || T * (p->Cout / 4) >= 0xffffffffLL || T > 0x7fffffff) return Y2_ENOSUP;
    const size_t vbytes = align256((size_t)16 * T * p->Cin * sizeof(float));
    const size_t mbytes = align256((size_t)16 * T * p->Cout * sizeof(float));

    // stage 2 as a grouped 1x1 "convolution" over an image of 1 x T pixels
    y2_conv_params q = {};
    q.B = 1; q.H = 1; q.W = (int)T; q.Cin = p->Cin; q.ldx = p->Cin; q.Cout = p->Cout; q.ksize = 1;
    q.ldy = p->Cout; q.slope = 1.f; q.tile = p->tile; q.algo = Y2_ALGO_DIRECT;
    if (ws_need != nullptr) {
        float* const dummy = reinterpret_cast<float*>(256);
        q.x = dummy; q.w = dummy; q.y = dummy;
        size_t inner = 0;
        const int rc = y2_internal_conv_grouped(&q, 16, T * p->Cin, (long long)p->Cout * p->Cin, T * p->Cout, stream, &inner);
        if (rc != Y2_OK) return rc;
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
        hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)y2_cdiv(Tc * ia.c4n, 256)), dim3(256), 0, s, ia);

        q.W = (int)Tc;
        q.x = V; q.w = p->w; q.y = M;
        q.workspace = M + mbytes / sizeof(float);
        q.workspace_bytes = (long long)((size_t)p->workspace_bytes - vbytes - mbytes);
        const int rc = y2_internal_conv_grouped(&q, 16, Tc * p->Cin, (long long)p->Cout * p->Cin, Tc * p->Cout, stream, nullptr);
        if (rc != Y2_OK) return rc;

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
        if (p->stats != nullptr) hipLaunchKernelGGL(wino_output_kernel<true>, grid, dim3(nx, ny), 0, s, oa);
        else hipLaunchKernelGGL(wino_output_kernel<false>, grid, dim3(nx, ny), 0, s, oa);
    }
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

// Winograd weight gradient of a 3x3 / stride-1 / same-padding convolution (replaces y2_conv_wgrad for the deep layers).
extern "C" long long y2_wino_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return Y2_EINVAL;
    const long long T = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
    return (long long)(align256((size_t)16 * T * Cin * 4) + align256((size_t)16 * T * Cout * 4) + align256((size_t)16 * Cout * Cin * 4));
}

extern "C" int y2_wino_wgrad(const float* x, const float* dz, float* dw_packed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t ldx,
                             int32_t Cout, int32_t ldz, const float* v_transformed, float* workspace, long long workspace_bytes, y2_stream_t stream) {
    if ((x == nullptr && v_transformed == nullptr) || dz == nullptr || dw_packed == nullptr || workspace == nullptr) return Y2_EINVAL;
    if (v_transformed != nullptr && !y2_aligned16(v_transformed)) return Y2_EALIGN;
    if (x == nullptr) x = v_transformed;      // only the alignment checks below look at it
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldz < Cout) return Y2_EINVAL;
    if ((Cin & 3) || (Cout & 3) || (ldx & 3) || (ldz & 3) || !y2_aligned16(x) || !y2_aligned16(dz) || !y2_aligned16(workspace)) return Y2_EALIGN;
    if (workspace_bytes < y2_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout)) return Y2_EINVAL;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    if (T * (Cin / 4) >= 0xffffffffLL || T * (Cout / 4) >= 0xffffffffLL || T > 0x7fffffff) return Y2_ENOSUP;
    float* V = workspace;
    float* DM = V + align256((size_t)16 * T * Cin * 4) / 4;
    float* DU = DM + align256((size_t)16 * T * Cout * 4) / 4;
    hipStream_t s = y2_s(stream);
    const size_t du_bytes = (size_t)16 * Cout * Cin * 4;
    hipError_t e = hipMemsetAsync(DU, 0, du_bytes, s);       // the grouped reduction adds split partial sums atomically
    if (e != hipSuccess) return -(1000 + (int)e);

    WinoInArgs ia;
    ia.x = x; ia.v = V; ia.B = B; ia.H = H; ia.W = W; ia.Cin = Cin; ia.ldx = ldx; ia.th = th; ia.tw = tw; ia.T = (int)T; ia.c4n = Cin / 4;
    ia.d_c4 = y2_make_fastdiv((uint32_t)ia.c4n); ia.d_tt = y2_make_fastdiv((uint32_t)(th * tw)); ia.d_tw = y2_make_fastdiv((uint32_t)tw);
    if (v_transformed == nullptr) hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)y2_cdiv(T * ia.c4n, 256)), dim3(256), 0, s, ia);
    const float* Vsrc = v_transformed != nullptr ? v_transformed : V;

    WinoDzArgs za;
    za.dz = dz; za.dm = DM; za.B = B; za.H = H; za.W = W; za.Cout = Cout; za.ldz = ldz; za.th = th; za.tw = tw; za.T = (int)T; za.c4n = Cout / 4;
    za.d_c4 = y2_make_fastdiv((uint32_t)za.c4n); za.d_tt = ia.d_tt; za.d_tw = ia.d_tw;
    hipLaunchKernelGGL(wino_dz_kernel, dim3((unsigned)y2_cdiv(T * za.c4n, 256)), dim3(256), 0, s, za);

    const int rc = y2_internal_wgrad_grouped(Vsrc, DM, DU, T, Cin, Cout, 16, T * Cin, T * Cout, (long long)Cout * Cin, stream);
    if (rc != Y2_OK) return rc;
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)y2_cdiv(n, 256)), dim3(256), 0, s, DU, dw_packed, Cout, Cin);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
