// optim.hip — fused multi-tensor optimizer steps and gradient-norm clipping for the YOLOv2 training path (gfx950).
//
// Replaces the per-tensor update loops of `self.optimizer.step()` (train.py:357; optimizer = eval of the ini lambda,
// config.ini:72: torch.optim.Adam / SGD over the 50.66 M Darknet-19 parameters) and the optional
// `nn.utils.clip_grad_norm` (train.py:352-354).  One launch handles up to Y2_OPT_MAX_TENSORS parameter tensors: the
// pointer table travels in the kernel arguments (no device-side table, nothing to upload, capturable in a hipGraph),
// a block finds its tensor by a scalar search over the running block counts, and streams 4096 elements with 16-B accesses
// (pure HBM work: SGD-momentum 3 reads + 2 writes, Adam 4 reads + 3 writes per element).
// Arithmetic follows torch.optim's single-tensor formulas step by step (-ffp-contract=off keeps them unfused).
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 4096;     // elements per block: 256 threads x 4 floats x 4 passes

struct OptTable {
    float* p[Y2_OPT_MAX_TENSORS];
    float* g[Y2_OPT_MAX_TENSORS];
    float* m[Y2_OPT_MAX_TENSORS];
    float* v[Y2_OPT_MAX_TENSORS];
    long long n[Y2_OPT_MAX_TENSORS];
    int block_end[Y2_OPT_MAX_TENSORS];      // exclusive running count of blocks
    int count;
};

struct OptPlace {
    int t;
    long long base, n;
    bool vec;
};

__device__ __forceinline__ OptPlace opt_place(const OptTable& tb) {
    int t = 0;
    const int b = blockIdx.x;
    while (t < tb.count - 1 && b >= tb.block_end[t]) ++t;       // scalar loop (uniform per block), <= 47 steps
    const int first = t == 0 ? 0 : tb.block_end[t - 1];
    OptPlace pl;
    pl.t = t;
    pl.base = (long long)(b - first) * OPT_CHUNK;
    pl.n = tb.n[t];
    pl.vec = ((reinterpret_cast<uintptr_t>(tb.p[t]) | reinterpret_cast<uintptr_t>(tb.g[t]) | reinterpret_cast<uintptr_t>(tb.m[t]) |
               reinterpret_cast<uintptr_t>(tb.v[t])) & 15u) == 0;
    return pl;
}

// f(i) for every element index i of this block's chunk, 4 consecutive elements per call when aligned (k = how many are valid)
template <typename F>
__device__ __forceinline__ void opt_for_each4(const OptPlace& pl, F f) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const long long i = pl.base + (long long)(pass * 256 + threadIdx.x) * 4;
        if (i >= pl.n) return;
        const int k = (int)(pl.n - i < 4 ? pl.n - i : 4);
        f(i, k);
    }
}

struct SgdHyper { float lr, momentum, dampening, weight_decay; int nesterov, first_step; };

__global__ __launch_bounds__(256) void opt_sgd_kernel(const OptTable tb, const SgdHyper h) {
    const OptPlace pl = opt_place(tb);
    float* p = tb.p[pl.t];
    const float* g = tb.g[pl.t];
    float* buf = tb.m[pl.t];
    opt_for_each4(pl, [&](long long i, int k) {
        float pv[4], gv[4], bv[4];
        const bool v4 = pl.vec && k == 4;
        if (v4) {
            *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(p + i);
            *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(g + i);
            if (buf != nullptr && !h.first_step) *reinterpret_cast<f32x4*>(bv) = *reinterpret_cast<const f32x4*>(buf + i);
        } else {
            for (int e = 0; e < k; ++e) { pv[e] = p[i + e]; gv[e] = g[i + e]; if (buf != nullptr && !h.first_step) bv[e] = buf[i + e]; }
        }
        for (int e = 0; e < k; ++e) {
            float d = gv[e];
            if (h.weight_decay != 0.f) d = d + h.weight_decay * pv[e];
            if (buf != nullptr) {
                const float b = h.first_step ? d : bv[e] * h.momentum + (1.f - h.dampening) * d;
                bv[e] = b;
                d = h.nesterov ? d + h.momentum * b : b;
            }
            pv[e] = pv[e] - h.lr * d;
        }
        if (v4) {
            *reinterpret_cast<f32x4*>(p + i) = *reinterpret_cast<f32x4*>(pv);
            if (buf != nullptr) *reinterpret_cast<f32x4*>(buf + i) = *reinterpret_cast<f32x4*>(bv);
        } else {
            for (int e = 0; e < k; ++e) { p[i + e] = pv[e]; if (buf != nullptr) buf[i + e] = bv[e]; }
        }
    });
}

struct AdamHyper { float lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt; };

__global__ __launch_bounds__(256) void opt_adam_kernel(const OptTable tb, const AdamHyper h) {
    const OptPlace pl = opt_place(tb);
    float* p = tb.p[pl.t];
    const float* g = tb.g[pl.t];
    float* m = tb.m[pl.t];
    float* v = tb.v[pl.t];
    opt_for_each4(pl, [&](long long i, int k) {
        float pv[4], gv[4], mv[4], vv[4];
        const bool v4 = pl.vec && k == 4;
        if (v4) {
            *reinterpret_cast<f32x4*>(pv) = *reinterpret_cast<const f32x4*>(p + i);
            *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(g + i);
            *reinterpret_cast<f32x4*>(mv) = *reinterpret_cast<const f32x4*>(m + i);
            *reinterpret_cast<f32x4*>(vv) = *reinterpret_cast<const f32x4*>(v + i);
        } else {
            for (int e = 0; e < k; ++e) { pv[e] = p[i + e]; gv[e] = g[i + e]; mv[e] = m[i + e]; vv[e] = v[i + e]; }
        }
        for (int e = 0; e < k; ++e) {
            float d = gv[e];
            if (h.weight_decay != 0.f) d = d + h.weight_decay * pv[e];
            mv[e] = mv[e] + (1.f - h.beta1) * (d - mv[e]);              // exp_avg.lerp_(grad, 1 - beta1)
            vv[e] = vv[e] * h.beta2 + (1.f - h.beta2) * (d * d);        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(vv[e]) / h.bc2_sqrt + h.eps;
            pv[e] = pv[e] + (-h.step_size) * (mv[e] / denom);           // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
        if (v4) {
            *reinterpret_cast<f32x4*>(p + i) = *reinterpret_cast<f32x4*>(pv);
            *reinterpret_cast<f32x4*>(m + i) = *reinterpret_cast<f32x4*>(mv);
            *reinterpret_cast<f32x4*>(v + i) = *reinterpret_cast<f32x4*>(vv);
        } else {
            for (int e = 0; e < k; ++e) { p[i + e] = pv[e]; m[i + e] = mv[e]; v[i + e] = vv[e]; }
        }
    });
}

// sum of squares of all gradients (fp64 accumulation: wave reduction, one atomic per block)
__global__ __launch_bounds__(256) void opt_sumsq_kernel(const OptTable tb, double* out) {
    __shared__ double part[4];
    const OptPlace pl = opt_place(tb);
    const float* g = tb.g[pl.t];
    double s = 0.0;
    const bool galigned = (reinterpret_cast<uintptr_t>(g) & 15u) == 0;
    opt_for_each4(pl, [&](long long i, int k) {
        if (galigned && k == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(g + i);
            s += (double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2] + (double)q[3] * q[3];
        } else {
            for (int e = 0; e < k; ++e) s += (double)g[i + e] * g[i + e];
        }
    });
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// grads *= clip_coef,  clip_coef = max_norm / (sqrt(sumsq) + 1e-6) when that is < 1 (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void opt_clip_kernel(const OptTable tb, const double* sumsq, float max_norm) {
    const float coef = max_norm / ((float)sqrt(*sumsq) + 1e-6f);
    if (!(coef < 1.f)) return;
    const OptPlace pl = opt_place(tb);
    float* g = tb.g[pl.t];
    const bool galigned = (reinterpret_cast<uintptr_t>(g) & 15u) == 0;
    opt_for_each4(pl, [&](long long i, int k) {
        if (galigned && k == 4) {
            f32x4 q = *reinterpret_cast<const f32x4*>(g + i);
            q *= coef;
            *reinterpret_cast<f32x4*>(g + i) = q;
        } else {
            for (int e = 0; e < k; ++e) g[i + e] *= coef;
        }
    });
}

int fill_table(OptTable& tb, const y2_opt_tensor* t, int count, bool need_m, bool need_v, long long& blocks) {
    if (t == nullptr || count <= 0 || count > Y2_OPT_MAX_TENSORS) return Y2_EINVAL;
    blocks = 0;
    for (int i = 0; i < Y2_OPT_MAX_TENSORS; ++i) {
        const bool live = i < count;
        tb.p[i] = live ? t[i].param : nullptr;
        tb.g[i] = live ? t[i].grad : nullptr;
        tb.m[i] = live ? t[i].state1 : nullptr;
        tb.v[i] = live ? t[i].state2 : nullptr;
        tb.n[i] = live ? t[i].numel : 0;
        if (live) {
            if (t[i].grad == nullptr || t[i].numel <= 0) return Y2_EINVAL;
            if (t[i].param == nullptr && (need_m || need_v)) return Y2_EINVAL;
            if ((need_m && t[i].state1 == nullptr) || (need_v && t[i].state2 == nullptr)) return Y2_EINVAL;
            blocks += (t[i].numel + OPT_CHUNK - 1) / OPT_CHUNK;
            if (blocks > 0x7fffffffLL) return Y2_EINVAL;
        }
        tb.block_end[i] = (int)blocks;
    }
    tb.count = count;
    return Y2_OK;
}

}  // namespace

extern "C" int y2_opt_sgd(const y2_opt_tensor* tensors, int32_t count, float lr, float momentum, float dampening, float weight_decay,
                          int32_t nesterov, int32_t first_step, y2_stream_t stream) {
    OptTable tb;
    long long blocks;
    const int rc = fill_table(tb, tensors, count, momentum != 0.f, false, blocks);
    if (rc != Y2_OK) return rc;
    for (int i = 0; i < count; ++i)
        if (tensors[i].param == nullptr) return Y2_EINVAL;
    if (momentum == 0.f)
        for (int i = 0; i < Y2_OPT_MAX_TENSORS; ++i) tb.m[i] = nullptr;
    SgdHyper h = {lr, momentum, dampening, weight_decay, nesterov, first_step};
    Y2_LAUNCH("opt_sgd_kernel", 0.0, opt_sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, y2_s(stream), tb, h);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_opt_adam(const y2_opt_tensor* tensors, int32_t count, float lr, float beta1, float beta2, float eps, float weight_decay,
                           int32_t step, y2_stream_t stream) {
    if (step < 1) return Y2_EINVAL;
    OptTable tb;
    long long blocks;
    const int rc = fill_table(tb, tensors, count, true, true, blocks);
    if (rc != Y2_OK) return rc;
    // bias corrections in double on the host, as torch.optim.Adam does with Python floats
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamHyper h = {lr, beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)sqrt(bc2)};
    Y2_LAUNCH("opt_adam_kernel", 0.0, opt_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, y2_s(stream), tb, h);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_opt_grad_sumsq(const y2_opt_tensor* tensors, int32_t count, double* sumsq, y2_stream_t stream) {
    if (sumsq == nullptr) return Y2_EINVAL;
    OptTable tb;
    long long blocks;
    const int rc = fill_table(tb, tensors, count, false, false, blocks);
    if (rc != Y2_OK) return rc;
    Y2_LAUNCH("opt_sumsq_kernel", 0.0, opt_sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, y2_s(stream), tb, sumsq);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_opt_clip_grads(const y2_opt_tensor* tensors, int32_t count, const double* sumsq, float max_norm, y2_stream_t stream) {
    if (sumsq == nullptr || !(max_norm > 0.f)) return Y2_EINVAL;
    OptTable tb;
    long long blocks;
    const int rc = fill_table(tb, tensors, count, false, false, blocks);
    if (rc != Y2_OK) return rc;
    Y2_LAUNCH("opt_clip_kernel", 0.0, opt_clip_kernel, dim3((unsigned)blocks), dim3(256), 0, y2_s(stream), tb, sumsq, max_norm);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
