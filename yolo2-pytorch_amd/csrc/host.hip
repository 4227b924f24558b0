// host.hip — HOST-memory entry points of libyolo2_hip.so (include/yolo2_hip.h: y2_nms_host, y2_iou_matrix_host, y2_iou_pair_host).
//
// The reference calls utils.postprocess.nms on CPU tensors from its summary worker process (train.py:209, a child forked
// after the GPU was initialised, which must never touch the device) and runs the utils.iou.torch unit tests on CPU tensors
// (utils/iou/torch.py:64-113).  These functions are the product's own host implementation of the SAME algorithms as the
// device kernels in detect.hip — rank by (score desc, index asc), L x L suppression bit matrix, serial greedy replay; the
// identical one-rounding-per-operation fp32 IoU sequence (this file is compiled with -ffp-contract=off like the rest) — so
// GPU and CPU callers get bit-identical keep lists.  No HIP call is made here: safe in a forked child.
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "yolo2_hip.h"

namespace {

inline float iou_host(float ymin1, float xmin1, float ymax1, float xmax1, float ymin2, float xmin2, float ymax2, float xmax2, float min_union) {
    // utils/iou/torch.py:34-61 (same operation order as detect.hip: iou_one)
    const float ih = fmaxf(fminf(ymax1, ymax2) - fmaxf(ymin1, ymin2), 0.f);
    const float iw = fmaxf(fminf(xmax1, xmax2) - fmaxf(xmin1, xmin2), 0.f);
    const float inter = ih * iw;
    const float a1 = (ymax1 - ymin1) * (xmax1 - xmin1);
    const float a2 = (ymax2 - ymin2) * (xmax2 - xmin2);
    const float uni = fmaxf((a1 + a2) - inter, min_union);
    return inter / uni;
}

inline float nms_key(float s) { return s != s ? -INFINITY : s; }   // NaN ranks last (detect.hip: nms_key)

}  // namespace

extern "C" int y2_nms_host(const float* score, const float* yx_min, const float* yx_max, const int32_t* cand, const int32_t* n, int B, int stride,
                           float overlap, int limit, int32_t* keep, int32_t* keep_count) {
    if (!score || !yx_min || !yx_max || !n || !keep || !keep_count) return Y2_EINVAL;
    if (B <= 0 || stride <= 0 || limit <= 0 || limit > 1024) return Y2_EINVAL;
    std::vector<int32_t> order;
    std::vector<float> key, box;
    for (int b = 0; b < B; ++b) {
        const int nb = n[b];
        if (nb < 0 || nb > stride) return Y2_EINVAL;
        const float* s = score + (size_t)b * stride;
        const int32_t* cd = cand ? cand + (size_t)b * stride : nullptr;
        int32_t* kp = keep + (size_t)b * limit;
        // stage 1 (utils/postprocess.py:37-38): the first `limit` of the descending order; ties -> lower index first
        key.resize((size_t)nb);
        order.resize((size_t)nb);
        for (int i = 0; i < nb; ++i) { key[(size_t)i] = nms_key(s[cd ? cd[i] : i]); order[(size_t)i] = i; }
        const int L = nb < limit ? nb : limit;
        std::partial_sort(order.begin(), order.begin() + L, order.end(), [&](int32_t a, int32_t c) {
            const float ka = key[(size_t)a], kc = key[(size_t)c];
            return ka > kc || (ka == kc && a < c);
        });
        // stage 2 (utils/postprocess.py:39-48): the serial greedy loop on a "removed" bit set; row i of the device kernel's
        // suppression matrix is evaluated only when i is kept (the rows of removed boxes are never used there either)
        const int words = (L + 63) >> 6;
        box.resize((size_t)L * 4);
        for (int r = 0; r < L; ++r) {
            const int i = order[(size_t)r];
            const size_t g = ((size_t)b * stride + (size_t)(cd ? cd[i] : i)) * 2;
            box[4 * (size_t)r] = yx_min[g]; box[4 * (size_t)r + 1] = yx_min[g + 1];
            box[4 * (size_t)r + 2] = yx_max[g]; box[4 * (size_t)r + 3] = yx_max[g + 1];
        }
        std::vector<uint64_t> removed((size_t)words, 0ull);
        int kept = 0;
        for (int i = 0; i < L; ++i) {
            if ((removed[(size_t)(i >> 6)] >> (i & 63)) & 1ull) continue;
            kp[kept++] = order[(size_t)i];
            const float y0 = box[4 * (size_t)i], x0 = box[4 * (size_t)i + 1], y1 = box[4 * (size_t)i + 2], x1 = box[4 * (size_t)i + 3];
            for (int j = i + 1; j < L; ++j) {
                const float v = iou_host(y0, x0, y1, x1, box[4 * (size_t)j], box[4 * (size_t)j + 1], box[4 * (size_t)j + 2], box[4 * (size_t)j + 3], 1.1920929e-07f);
                if (!(v <= overlap)) removed[(size_t)(j >> 6)] |= 1ull << (j & 63);     // kept iff iou <= overlap (utils/postprocess.py:48)
            }
        }
        keep_count[b] = kept;
    }
    return Y2_OK;
}

extern "C" int y2_iou_matrix_host(const float* mn1, const float* mx1, const float* mn2, const float* mx2,
                                  int Bt, int N1, int N2, float min_union, int mode, float* out) {
    if (Bt < 0 || N1 < 0 || N2 < 0 || (mode != 0 && mode != 1)) return Y2_EINVAL;
    if ((long long)Bt * N1 * N2 == 0) return Y2_OK;
    if (!mn1 || !mx1 || !mn2 || !mx2 || !out) return Y2_EINVAL;
    for (int b = 0; b < Bt; ++b)
        for (int i = 0; i < N1; ++i) {
            const size_t r = ((size_t)b * N1 + i) * 2;
            for (int j = 0; j < N2; ++j) {
                const size_t c = ((size_t)b * N2 + j) * 2;
                float v;
                if (mode == 0) {
                    v = iou_host(mn1[r], mn1[r + 1], mx1[r], mx1[r + 1], mn2[c], mn2[c + 1], mx2[c], mx2[c + 1], min_union);
                } else {
                    const float ih = fmaxf(fminf(mx1[r], mx2[c]) - fmaxf(mn1[r], mn2[c]), 0.f);
                    const float iw = fmaxf(fminf(mx1[r + 1], mx2[c + 1]) - fmaxf(mn1[r + 1], mn2[c + 1]), 0.f);
                    v = ih * iw;
                }
                out[((size_t)b * N1 + i) * N2 + j] = v;
            }
        }
    return Y2_OK;
}

extern "C" int y2_iou_pair_host(const float* mn1, const float* mx1, const float* mn2, const float* mx2, int n, float min_union, float* out) {
    if (n < 0) return Y2_EINVAL;
    if (n == 0) return Y2_OK;
    if (!mn1 || !mx1 || !mn2 || !mx2 || !out) return Y2_EINVAL;
    for (int i = 0; i < n; ++i)
        out[i] = iou_host(mn1[2 * i], mn1[2 * i + 1], mx1[2 * i], mx1[2 * i + 1], mn2[2 * i], mn2[2 * i + 1], mx2[2 * i], mx2[2 * i + 1], min_union);
    return Y2_OK;
}
