// detect.hip — detection-head kernels for gfx950: decode, visibility filter, IoU, NMS.
//
// These are HBM/latency-bound integer-and-compare kernels (≈170 KB of traffic per image): the design rules are
// coalesced reads (whole wave reads contiguous boxes, staged through LDS when the per-box record is wide),
// wave64 ballots / reductions instead of serial loops, and no host round trips between stages.
// Compiled with -ffp-contract=off: IoU arithmetic must be bit-identical to the reference's fp32 operation
// sequence (utils/iou/torch.py:34-61) so that NMS survivor indices are bit-exact.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ decode
// model/__init__.py:120-135.  One wave (64 lanes) per 64 consecutive boxes; the 64*(5+C) record floats are
// contiguous in the NHWC head image, so they are loaded coalesced into LDS (record stride E is odd for
// the shipped heads 25 / 85 -> conflict-free per-lane walks) and the softmax row is written back coalesced.
struct DecodeArgs {
    const float* feature; const float* anchors;
    float *iou, *center_offset, *size_norm, *yx_min, *yx_max, *prob, *prob_cls;
    int32_t* cls;
    int total, cells, rows, A, C, E;
};

__global__ __launch_bounds__(64) void decode_kernel(const DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float rec[];   // [64][E]
    const int lane = threadIdx.x;
    const int box0 = blockIdx.x * 64;
    const int nbox = min(64, a.total - box0);
    const int E = a.E, C = a.C;
    const float* src = a.feature + (size_t)box0 * E;
    for (int i = lane; i < nbox * E; i += 64) rec[i] = src[i];
    __syncthreads();
    const int box = box0 + lane;
    if (lane < nbox) {
        const float* f = rec + lane * E;
        const int anc = box % a.A;
        const int cell = (box / a.A) % a.cells;
        const float s0 = 1.f / (1.f + expf(-f[0]));
        const float s1 = 1.f / (1.f + expf(-f[1]));
        const float s2 = 1.f / (1.f + expf(-f[2]));
        // meshgrid quirk, model/__init__.py:53-56: cell k -> (k / rows, k % rows)
        const float cy = (float)(cell / a.rows) + s1;
        const float cx = (float)(cell % a.rows) + s2;
        const float h2 = (expf(f[3]) * a.anchors[2 * anc]) / 2.f;
        const float w2 = (expf(f[4]) * a.anchors[2 * anc + 1]) / 2.f;
        if (a.iou) a.iou[box] = s0;
        if (a.center_offset) { a.center_offset[2 * (size_t)box] = s1; a.center_offset[2 * (size_t)box + 1] = s2; }
        if (a.size_norm) { a.size_norm[2 * (size_t)box] = f[3]; a.size_norm[2 * (size_t)box + 1] = f[4]; }
        if (a.yx_min) { a.yx_min[2 * (size_t)box] = cy - h2; a.yx_min[2 * (size_t)box + 1] = cx - w2; }
        if (a.yx_max) { a.yx_max[2 * (size_t)box] = cy + h2; a.yx_max[2 * (size_t)box + 1] = cx + w2; }
        if (C > 0 && (a.prob || a.prob_cls || a.cls)) {
            float* lg = rec + lane * E + 5;
            float mx = lg[0];
            int arg = 0;
            for (int c = 1; c < C; ++c) { if (lg[c] > mx) { mx = lg[c]; arg = c; } }
            float sum = 0.f;
            for (int c = 0; c < C; ++c) { const float e = expf(lg[c] - mx); lg[c] = e; sum += e; }
            for (int c = 0; c < C; ++c) lg[c] = lg[c] / sum;
            if (a.prob_cls) a.prob_cls[box] = lg[arg];   // first maximal logit == first maximal probability
            if (a.cls) a.cls[box] = arg;
        } else if (C == 0) {
            if (a.prob_cls) a.prob_cls[box] = 1.f;        // detect.get_logits: ones[...,1] when single-class (detect.py:43-48)
            if (a.cls) a.cls[box] = 0;
        }
    }
    __syncthreads();
    if (a.prob && C > 0) {
        float* dst = a.prob + (size_t)box0 * C;
        for (int i = lane; i < nbox * C; i += 64) dst[i] = rec[(i / C) * E + 5 + (i % C)];
    }
}

// ------------------------------------------------------------------------------------------------ filter
// detect.py:52: prob_cls, cls = max(prob, -1): one wave per candidate, coalesced read of its C probabilities,
// wave64 max-reduction carrying the (first) arg-max.
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ prob, int rows, int C, float* prob_cls, int32_t* cls) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = prob + (size_t)row * C;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = p[c];
        if (v > best) { best = v; arg = c; }    // ascending c within a lane -> first max per lane
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(arg, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { prob_cls[row] = best; cls[row] = arg; }
}

// detect.py:53-62: mask + order-preserving compaction.  One workgroup per image; wave ballots + LDS wave offsets.
__global__ __launch_bounds__(256) void compact_kernel(const float* __restrict__ iou, const float* __restrict__ prob_cls, int n, int fix, float thr,
                                                      int32_t* count, int32_t* index) {
    __shared__ int wave_cnt[4];
    __shared__ int base;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + t;
        bool keep = false;
        if (i < n) {
            const float s = iou[(size_t)b * n + i];
            keep = fix ? (s * prob_cls[(size_t)b * n + i]) > thr : s > thr;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) index[(size_t)b * n + off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (t == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (t == 0) count[b] = base;
}

// detect.py:69-79 after NMS (a22): gather the kept boxes of every image and, with `fix`, expand them into (box, class) detections -
// every pair whose score = iou * prob exceeds the class threshold, in row-major (box-major, class-minor) order.  One workgroup per
// image; ordered compaction with wave ballots like compact_kernel.  kept[b][k] = candidate-list position of the k-th survivor.
struct ExpandArgs {
    const float* iou; const float* prob; const float* yx_min; const float* yx_max;      // [B][n], [B][n][C], [B][n][2], [B][n][2]
    const int32_t* cand; const int32_t* keep; const int32_t* keep_count;                 // [B][n], [B][limit], [B]
    float* k_iou; float* k_min; float* k_max;                                            // kept boxes: [B][limit], [B][limit][2] x 2
    float* e_min; float* e_max; float* e_score; long long* e_cls; int32_t* e_count;      // expanded: [B][limit*C][2] x 2, [B][limit*C] x 2, [B]
    int n, C, limit;
    float thr;
};

__global__ __launch_bounds__(256) void expand_classes_kernel(const ExpandArgs a) {
    __shared__ int wave_cnt[4];
    __shared__ int base;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kc = a.keep_count[b];
    if (t == 0) base = 0;
    for (int k = t; k < kc; k += 256) {
        const int src = a.cand != nullptr ? a.cand[(size_t)b * a.n + a.keep[(size_t)b * a.limit + k]] : a.keep[(size_t)b * a.limit + k];
        const size_t o = (size_t)b * a.n + src, d = (size_t)b * a.limit + k;
        a.k_iou[d] = a.iou[o];
        a.k_min[2 * d] = a.yx_min[2 * o]; a.k_min[2 * d + 1] = a.yx_min[2 * o + 1];
        a.k_max[2 * d] = a.yx_max[2 * o]; a.k_max[2 * d + 1] = a.yx_max[2 * o + 1];
    }
    __syncthreads();
    if (a.e_count == nullptr) return;
    const int total = kc * a.C;
    const size_t cap = (size_t)a.limit * a.C;
    for (int i0 = 0; i0 < total; i0 += 256) {
        const int i = i0 + t;
        bool hit = false;
        float sc = 0.f;
        int k = 0, c = 0;
        size_t o = 0;
        if (i < total) {
            k = i / a.C; c = i - k * a.C;
            const int src = a.cand != nullptr ? a.cand[(size_t)b * a.n + a.keep[(size_t)b * a.limit + k]] : a.keep[(size_t)b * a.limit + k];
            o = (size_t)b * a.n + src;
            sc = a.iou[o] * a.prob[o * a.C + c];            // one fp32 multiply, like `iou.unsqueeze(-1) * prob`
            hit = sc > a.thr;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (hit) {
            const size_t d = (size_t)b * cap + off + __popcll(m & ((1ull << lane) - 1ull));
            a.e_min[2 * d] = a.yx_min[2 * o]; a.e_min[2 * d + 1] = a.yx_min[2 * o + 1];
            a.e_max[2 * d] = a.yx_max[2 * o]; a.e_max[2 * d + 1] = a.yx_max[2 * o + 1];
            a.e_score[d] = sc;
            a.e_cls[d] = c;
        }
        __syncthreads();
        if (t == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (t == 0) a.e_count[b] = base;
}

// ------------------------------------------------------------------------------------------------ IoU
// utils/iou/torch.py:34-44,47-61 operation order; every op is a separate fp32 rounding.
__device__ __forceinline__ float iou_one(float ymin1, float xmin1, float ymax1, float xmax1,
                                         float ymin2, float xmin2, float ymax2, float xmax2, float min_union) {
    const float ih = fmaxf(fminf(ymax1, ymax2) - fmaxf(ymin1, ymin2), 0.f);
    const float iw = fmaxf(fminf(xmax1, xmax2) - fmaxf(xmin1, xmin2), 0.f);
    const float inter = ih * iw;
    const float a1 = (ymax1 - ymin1) * (xmax1 - xmin1);
    const float a2 = (ymax2 - ymin2) * (xmax2 - xmin2);
    const float uni = fmaxf((a1 + a2) - inter, min_union);
    return inter / uni;
}

// eval.py:67-75: per prediction the best IoU over the ground-truth boxes and its index (first maximum, like torch.max over the IoU matrix).
__global__ __launch_bounds__(256) void iou_rowmax_kernel(const float* __restrict__ mn1, const float* __restrict__ mx1, const float* __restrict__ mn2, const float* __restrict__ mx2,
                                                         int N1, int N2, float min_union, float* __restrict__ best, long long* __restrict__ which) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N1) return;
    const float y0 = mn1[2 * i], x0 = mn1[2 * i + 1], y1 = mx1[2 * i], x1 = mx1[2 * i + 1];
    float bv = 0.f;
    int bi = 0;
    for (int j = 0; j < N2; ++j) {
        const float v = iou_one(y0, x0, y1, x1, mn2[2 * j], mn2[2 * j + 1], mx2[2 * j], mx2[2 * j + 1], min_union);
        if (j == 0 || v > bv) { bv = v; bi = j; }
    }
    best[i] = bv;
    which[i] = bi;
}

__global__ void iou_matrix_kernel(const float* __restrict__ mn1, const float* __restrict__ mx1, const float* __restrict__ mn2, const float* __restrict__ mx2,
                                  int N1, int N2, float min_union, int mode, float* __restrict__ out, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % N2);
        const long long r = i / N2;          // b*N1 + i1
        const long long b = r / N1;
        const long long c = b * N2 + j;
        if (mode == 0) {
            out[i] = iou_one(mn1[2 * r], mn1[2 * r + 1], mx1[2 * r], mx1[2 * r + 1], mn2[2 * c], mn2[2 * c + 1], mx2[2 * c], mx2[2 * c + 1], min_union);
        } else {
            const float ih = fmaxf(fminf(mx1[2 * r], mx2[2 * c]) - fmaxf(mn1[2 * r], mn2[2 * c]), 0.f);
            const float iw = fmaxf(fminf(mx1[2 * r + 1], mx2[2 * c + 1]) - fmaxf(mn1[2 * r + 1], mn2[2 * c + 1]), 0.f);
            out[i] = ih * iw;
        }
    }
}

__global__ void iou_pair_kernel(const float* __restrict__ mn1, const float* __restrict__ mx1, const float* __restrict__ mn2, const float* __restrict__ mx2,
                                int n, float min_union, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = iou_one(mn1[2 * i], mn1[2 * i + 1], mx1[2 * i], mx1[2 * i + 1], mn2[2 * i], mn2[2 * i + 1], mx2[2 * i], mx2[2 * i + 1], min_union);
}

// ------------------------------------------------------------------------------------------------ NMS
// Stage 1 — utils/postprocess.py:37-38 `score.sort(descending=True)[:limit]` without a sort: the rank of
// candidate i is #{j : s_j > s_i or (s_j == s_i and j < i)} (ties -> lower index first, a total order), computed
// by all threads against LDS-staged score tiles; candidates with rank < limit scatter themselves to order[rank].
// A NaN score has no place in an ordering: it ranks as -inf (after every number; among themselves lower index first), so the
// ranks stay a permutation of 0..n-1 and every order[] slot below min(n, limit) is written.
__device__ __forceinline__ float nms_key(float s) { return s != s ? -__builtin_inff() : s; }

__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ score, const int32_t* __restrict__ cand, const int32_t* __restrict__ n_per, int stride, int limit, int32_t* order) {
    __shared__ float tile[1024];
    const int b = blockIdx.y;
    const int n = n_per[b];
    if ((int)(blockIdx.x * 256) >= n) return;            // uniform per block
    const float* s = score + (size_t)b * stride;
    const int32_t* cd = cand ? cand + (size_t)b * stride : nullptr;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float si = i < n ? nms_key(s[cd ? cd[i] : i]) : 0.f;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        const int cnt = min(1024, n - j0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += 256) tile[j] = nms_key(s[cd ? cd[j0 + j] : j0 + j]);
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const float sj = tile[j];
            rank += (sj > si || (sj == si && (j0 + j) < i)) ? 1 : 0;
        }
    }
    if (i < n && rank < limit) order[(size_t)b * limit + rank] = i;
}

// Stage 2 — utils/postprocess.py:39-48 greedy loop.  One workgroup per image: the L = min(n, limit) boxes in
// descending-score order go to LDS; the L x L "j is suppressed by i" bit matrix (IoU(i,j) <= overlap is FALSE,
// j > i) is built by all 256 threads; wave 0 then replays the reference's serial loop on 64-bit words.
__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ yx_min, const float* __restrict__ yx_max, const int32_t* __restrict__ cand, const int32_t* __restrict__ n_per,
                                                  int stride, float overlap, int limit, const int32_t* __restrict__ order,
                                                  int32_t* keep, int32_t* keep_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = blockIdx.x;
    const int n = n_per[b];
    const int L = min(n, limit);
    const int t = threadIdx.x;
    if (L <= 0) { if (t == 0) keep_count[b] = 0; return; }
    const int words = (L + 63) >> 6;
    float* box = reinterpret_cast<float*>(smem_raw);                                   // [L][4] ymin xmin ymax xmax
    int* idx = reinterpret_cast<int*>(box + 4 * L);                                    // [L]
    unsigned long long* mat = reinterpret_cast<unsigned long long*>(smem_raw + (((size_t)L * 20 + 15) & ~(size_t)15));   // [L][words]
    const int32_t* ord = order + (size_t)b * limit;
    for (int r = t; r < L; r += 256) {
        const int i = ord[r];
        idx[r] = i;
        const size_t g = ((size_t)b * stride + (cand ? cand[(size_t)b * stride + i] : i)) * 2;
        box[4 * r] = yx_min[g]; box[4 * r + 1] = yx_min[g + 1];
        box[4 * r + 2] = yx_max[g]; box[4 * r + 3] = yx_max[g + 1];
    }
    __syncthreads();
    // bit matrix: element (i, w) covers j = 64*w .. 64*w+63
    for (int e = t; e < L * words; e += 256) {
        const int i = e / words, w = e - i * words;
        unsigned long long bits = 0ull;
        const float y0 = box[4 * i], x0 = box[4 * i + 1], y1 = box[4 * i + 2], x1 = box[4 * i + 3];
        const int jend = min(L, 64 * w + 64);
        for (int j = max(64 * w, i + 1); j < jend; ++j) {
            const float v = iou_one(y0, x0, y1, x1, box[4 * j], box[4 * j + 1], box[4 * j + 2], box[4 * j + 3], 1.1920929e-07f);
            if (!(v <= overlap)) bits |= 1ull << (j & 63);    // kept iff iou <= overlap (utils/postprocess.py:48)
        }
        mat[e] = bits;
    }
    __syncthreads();
    if (t < 64) {
        // lane w owns word w of the "removed" set (words <= 16 <= 64)
        unsigned long long removed = 0ull;
        int kept = 0;
        for (int i = 0; i < L; ++i) {
            const unsigned long long wv = __shfl(removed, i >> 6);
            if (!((wv >> (i & 63)) & 1ull)) {          // uniform across the wave
                if (t == 0) keep[(size_t)b * limit + kept] = idx[i];
                ++kept;
                if (t < words) removed |= mat[i * words + t];
            }
        }
        if (t == 0) keep_count[b] = kept;
    }
}

inline int stream_grid(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = (long long)Y2_NUM_CU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int y2_decode(const float* feature, const float* anchors, int B, int rows, int cols, int A, int C,
                         float* iou, float* center_offset, float* size_norm, float* yx_min, float* yx_max, float* prob,
                         float* prob_cls, int32_t* cls, y2_stream_t stream) {
    if (!feature || !anchors || B <= 0 || rows <= 0 || cols <= 0 || A <= 0 || C < 0) return Y2_EINVAL;
    DecodeArgs a;
    a.feature = feature; a.anchors = anchors; a.iou = iou; a.center_offset = center_offset; a.size_norm = size_norm;
    a.yx_min = yx_min; a.yx_max = yx_max; a.prob = prob; a.prob_cls = prob_cls; a.cls = cls;
    a.cells = rows * cols; a.rows = rows; a.A = A; a.C = C; a.E = 5 + C;
    const long long total = (long long)B * a.cells * A;
    if (total > 0x7fffffffLL) return Y2_EINVAL;
    a.total = (int)total;
    const size_t lds = (size_t)64 * a.E * sizeof(float);
    if (lds > 160 * 1024) return Y2_ENOSUP;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(1000 + (int)e);
    }
    Y2_LAUNCH("decode_kernel", 0.0, decode_kernel, dim3(y2_cdiv(total, 64)), dim3(64), lds, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

namespace {
// (a kernel, not hipMemsetAsync: a memset node of a captured hipGraph is not reliably ordered in front of the kernels behind it on this runtime, see wino.hip: zero_fill_kernel)
__global__ void zero_i32_kernel(int32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
}  // namespace

extern "C" int y2_filter_visible(const float* iou, const float* prob, int B, int n, int C, int fix, float thr,
                                 int32_t* count, int32_t* index, float* prob_cls, int32_t* cls, y2_stream_t stream) {
    if (!iou || !count || !index || B <= 0 || n < 0) return Y2_EINVAL;
    if (n == 0) {
        Y2_LAUNCH("zero_i32_kernel", 0.0, zero_i32_kernel, dim3(y2_cdiv(B, 256)), dim3(256), 0, y2_s(stream), count, B);
        Y2_LAUNCH_CHECK();
        return Y2_OK;
    }
    if (prob != nullptr) {
        if (!prob_cls || !cls || C < 1) return Y2_EINVAL;
        const long long rows = (long long)B * n;
        Y2_LAUNCH("rowmax_kernel", 0.0, rowmax_kernel, dim3(y2_cdiv(rows, 4)), dim3(256), 0, y2_s(stream), prob, (int)rows, C, prob_cls, cls);
        Y2_LAUNCH_CHECK();
    } else if (fix && !prob_cls) {
        return Y2_EINVAL;
    }
    Y2_LAUNCH("compact_kernel", 0.0, compact_kernel, dim3(B), dim3(256), 0, y2_s(stream), iou, prob_cls, n, fix, thr, count, index);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_expand_classes(const float* iou, const float* prob, const float* yx_min, const float* yx_max, const int32_t* cand, const int32_t* keep,
                                 const int32_t* keep_count, int B, int n, int C, int limit, float threshold_cls, float* k_iou, float* k_min, float* k_max,
                                 float* e_min, float* e_max, float* e_score, long long* e_cls, int32_t* e_count, y2_stream_t stream) {
    if (!iou || !yx_min || !yx_max || !keep || !keep_count || !k_iou || !k_min || !k_max) return Y2_EINVAL;
    if (B <= 0 || n <= 0 || C < 0 || limit <= 0) return Y2_EINVAL;
    if (e_count != nullptr && (!prob || C <= 0 || !e_min || !e_max || !e_score || !e_cls)) return Y2_EINVAL;
    ExpandArgs a;
    a.iou = iou; a.prob = prob; a.yx_min = yx_min; a.yx_max = yx_max; a.cand = cand; a.keep = keep; a.keep_count = keep_count;
    a.k_iou = k_iou; a.k_min = k_min; a.k_max = k_max; a.e_min = e_min; a.e_max = e_max; a.e_score = e_score; a.e_cls = e_cls; a.e_count = e_count;
    a.n = n; a.C = C; a.limit = limit; a.thr = threshold_cls;
    Y2_LAUNCH("expand_classes_kernel", 0.0, expand_classes_kernel, dim3(B), dim3(256), 0, y2_s(stream), a);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_iou_rowmax(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2, int N1, int N2, float min_union,
                             float* best, long long* which, y2_stream_t stream) {
    if (N1 < 0 || N2 <= 0) return Y2_EINVAL;
    if (N1 == 0) return Y2_OK;
    if (!yx_min1 || !yx_max1 || !yx_min2 || !yx_max2 || !best || !which) return Y2_EINVAL;
    Y2_LAUNCH("iou_rowmax_kernel", 0.0, iou_rowmax_kernel, dim3(y2_cdiv(N1, 256)), dim3(256), 0, y2_s(stream), yx_min1, yx_max1, yx_min2, yx_max2, N1, N2, min_union, best, which);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_iou_matrix(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2,
                             int Bt, int N1, int N2, float min_union, int mode, float* out, y2_stream_t stream) {
    if (Bt < 0 || N1 < 0 || N2 < 0 || (mode != 0 && mode != 1)) return Y2_EINVAL;
    const long long total = (long long)Bt * N1 * N2;
    if (total == 0) return Y2_OK;
    if (!yx_min1 || !yx_max1 || !yx_min2 || !yx_max2 || !out) return Y2_EINVAL;
    Y2_LAUNCH("iou_matrix_kernel", 0.0, iou_matrix_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, y2_s(stream), yx_min1, yx_max1, yx_min2, yx_max2, N1, N2, min_union, mode, out, total);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_iou_pair(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2,
                           int n, float min_union, float* out, y2_stream_t stream) {
    if (n < 0) return Y2_EINVAL;
    if (n == 0) return Y2_OK;
    if (!yx_min1 || !yx_max1 || !yx_min2 || !yx_max2 || !out) return Y2_EINVAL;
    Y2_LAUNCH("iou_pair_kernel", 0.0, iou_pair_kernel, dim3(y2_cdiv(n, 256)), dim3(256), 0, y2_s(stream), yx_min1, yx_max1, yx_min2, yx_max2, n, min_union, out);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}

extern "C" int y2_nms(const float* score, const float* yx_min, const float* yx_max, const int32_t* cand, const int32_t* n, int B, int stride,
                      float overlap, int limit, int32_t* order_ws, int32_t* keep, int32_t* keep_count, y2_stream_t stream) {
    if (!score || !yx_min || !yx_max || !n || !order_ws || !keep || !keep_count) return Y2_EINVAL;
    if (B <= 0 || stride <= 0 || limit <= 0 || limit > 1024) return Y2_EINVAL;
    hipStream_t s = y2_s(stream);
    Y2_LAUNCH("nms_rank_kernel", 0.0, nms_rank_kernel, dim3(y2_cdiv(stride, 256), B), dim3(256), 0, s, score, cand, n, stride, limit, order_ws);
    Y2_LAUNCH_CHECK();
    const int words = (limit + 63) / 64;
    const size_t lds = (((size_t)limit * 20 + 15) & ~(size_t)15) + (size_t)limit * words * 8;
    static size_t attr_lds[Y2_MAX_DEVICES] = {};       // high-water mark per device (function attributes are per device)
    if (lds > 64 * 1024) {
        const int dev = y2_current_device();
        if (dev < 0) return Y2_EINVAL;
        if (lds > attr_lds[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return -(1000 + (int)e);
            attr_lds[dev] = lds;
        }
    }
    Y2_LAUNCH("nms_kernel", 0.0, nms_kernel, dim3(B), dim3(256), lds, s, yx_min, yx_max, cand, n, stride, overlap, limit, order_ws, keep, keep_count);
    Y2_LAUNCH_CHECK();
    return Y2_OK;
}
