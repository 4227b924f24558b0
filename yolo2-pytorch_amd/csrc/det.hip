// det.hip — deterministic mode of libyolo2_hip.so (include/yolo2_hip.h: y2_set_deterministic, y2_colstats_det).
//
// By default three reductions combine partial sums in the order workgroups happen to finish: the fp32 atomics of the split-K
// weight gradient (conv_wgrad.hip), the LDS / fp64 atomics of the BatchNorm-backward sums (train.hip) and the fp64 atomics of
// the BatchNorm statistics and the loss sums.  Results then differ from run to run in the last bits (~1e-6 relative on weight
// gradients).  In deterministic mode every such reduction writes its partials to a caller-provided scratch area and a second
// kernel adds them in a FIXED tree (this file); the forward BatchNorm statistics are taken by a dedicated two-stage column
// reduction over the raw convolution output instead of the atomics in the convolution epilogues.  Same inputs -> same bits, at
// the price of one extra pass over each activation and a few small launches (measured in DESIGN.md).
#include "common.h"

Y2Det y2_det = {0, nullptr, 0};

namespace {

// out[col] = sum over rows of part[row * stride + col], fixed structure: thread (rl, col) adds rows rl, rl + RL, ... in order,
// the RL row-lane partials of a column are then added in order 0 .. RL-1.  Deterministic for a given (R, N).
template <typename TI>
__global__ __launch_bounds__(256) void det_reduce_rows_kernel(const TI* __restrict__ part, int R, long long N, long long stride, double* out_d, float* out_f) {
    constexpr int RL = 8;
    __shared__ double red[RL][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const long long col = (long long)blockIdx.x * 32 + cl;
    double s = 0.0;
    if (col < N)
        for (int r = rl; r < R; r += RL) s += (double)part[(long long)r * stride + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < N) {
        double tot = red[0][cl];
#pragma unroll
        for (int i = 1; i < RL; ++i) tot += red[i][cl];
        if (out_d != nullptr) out_d[col] = tot;
        if (out_f != nullptr) out_f[col] = (float)tot;
    }
}

// stage 1 of the column statistics: block b owns rows [b*chunk, (b+1)*chunk) of x [M, C] (row stride ld) and writes
// part[b][0..C) = sum x, part[b][C..2C) = sum x^2 (fp64).  Thread (rl, cg) walks its rows in order; row lanes are combined in order.
template <int V>
__global__ __launch_bounds__(256) void det_colstats_kernel(const float* __restrict__ x, long long M, int C, int ld, long long chunk, double* __restrict__ part) {
    extern __shared__ double lred[];                  // [RL][2*C] when RL > 1
    const int Cg = C / V;
    const int RL = Cg >= 256 ? 1 : 256 / Cg;
    const long long r0 = (long long)blockIdx.x * chunk, r1 = (r0 + chunk < M) ? r0 + chunk : M;
    double* out = part + (size_t)blockIdx.x * 2 * C;
    if (RL == 1) {
        for (int cg = threadIdx.x; cg < Cg; cg += 256) {
            double s1[V], s2[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
            for (long long r = r0; r < r1; ++r) {
#pragma unroll
                for (int e = 0; e < V; ++e) { const double v = (double)x[r * ld + cg * V + e]; s1[e] += v; s2[e] += v * v; }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) { out[cg * V + e] = s1[e]; out[C + cg * V + e] = s2[e]; }
        }
        return;
    }
    const int rl = threadIdx.x / Cg, cg = threadIdx.x % Cg;
    double s1[V], s2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
    if (rl < RL) {
        for (long long r = r0 + rl; r < r1; r += RL) {
            if (V == 4) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(x + r * ld + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double v = (double)q[e]; s1[e] += v; s2[e] += v * v; }
            } else {
                const double v = (double)x[r * ld + cg]; s1[0] += v; s2[0] += v * v;
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) { lred[(size_t)rl * 2 * C + cg * V + e] = s1[e]; lred[(size_t)rl * 2 * C + C + cg * V + e] = s2[e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        double tot = lred[i];
        for (int q = 1; q < RL; ++q) tot += lred[(size_t)q * 2 * C + i];
        out[i] = tot;
    }
}

}  // namespace

int y2_det_reduce_f32(const float* part, int R, long long N, long long stride, double* out_d, float* out_f, hipStream_t s) {
    if (N <= 0 || R <= 0) return Y2_OK;
    Y2_LAUNCH("det_reduce_rows_kernel", 0.0, det_reduce_rows_kernel<float>, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, s, part, R, N, stride, out_d, out_f);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_set_deterministic(int on, float* workspace, long long workspace_bytes) {
    if (on && (workspace == nullptr || workspace_bytes < (1ll << 20) || !y2_aligned16(workspace))) return Y2_EINVAL;
    y2_det.on = on ? 1 : 0;
    y2_det.ws = on ? workspace : nullptr;
    y2_det.bytes = on ? (size_t)workspace_bytes : 0;
    return Y2_OK;
}

extern "C" int y2_get_deterministic(void) { return y2_det.on; }

extern "C" int y2_colstats_det(const float* x, long long M, int32_t C, int32_t ld, double* stats, float* workspace, long long workspace_bytes, y2_stream_t stream) {
    if (!x || !stats || !workspace || M <= 0 || C <= 0 || ld < C || C > 8192) return Y2_EINVAL;
    long long G = (M + 255) / 256;                     // >= 256 rows per block, at most 1024 blocks
    if (G > 1024) G = 1024;
    if (G < 1) G = 1;
    const long long chunk = (M + G - 1) / G;
    G = (M + chunk - 1) / chunk;
    const size_t need = (size_t)G * 2 * C * sizeof(double);
    if ((size_t)workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 7u)) return Y2_EINVAL;
    double* part = reinterpret_cast<double*>(workspace);
    const bool vec = (C % 4 == 0) && (ld % 4 == 0) && y2_aligned16(x);
    const int V = vec ? 4 : 1;
    const int Cg = C / V;
    const int RL = Cg >= 256 ? 1 : 256 / Cg;
    const size_t lds = RL > 1 ? (size_t)RL * 2 * C * sizeof(double) : 0;
    if (lds > 64 * 1024) return Y2_ENOSUP;
    hipStream_t s = y2_s(stream);
    if (vec) Y2_LAUNCH("det_colstats_kernel", 0.0, det_colstats_kernel<4>, dim3((unsigned)G), dim3(256), lds, s, x, M, C, ld, chunk, part);
    else Y2_LAUNCH("det_colstats_kernel", 0.0, det_colstats_kernel<1>, dim3((unsigned)G), dim3(256), lds, s, x, M, C, ld, chunk, part);
    Y2_LAUNCH("det_reduce_rows_kernel", 0.0, det_reduce_rows_kernel<double>, dim3((unsigned)((2 * C + 31) / 32)), dim3(256), 0, s, part, (int)G, (long long)2 * C, (long long)2 * C, stats, (float*)nullptr);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
