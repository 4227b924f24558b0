/*
 * yolo2_hip.h — C ABI of libyolo2_hip.so: the MI355X (gfx950) YOLOv2 hot path.
 *
 * The reference (ruiminshen/yolo2-pytorch) has no FFI: its "operator API" for this path is a set of
 * Python call signatures.  This header is the boundary underneath the Python mirror of
 * those signatures (yolo2-pytorch_amd/model, utils.postprocess, utils.iou.torch): every entry point
 * names the reference interface (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes; no torch types.  All pointers are DEVICE pointers unless a
 *     parameter says "host".  All floating point is fp32; indices int32 unless stated.
 *   - Every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); nothing here
 *     allocates, frees or synchronises.  The caller owns every buffer and keeps it alive until the
 *     stream has drained (PyTorch caching-allocator stream semantics).
 *   - Return value: 0 on success, a negative Y2_E* code on an argument error, or -(1000+hipError_t)
 *     when a launch failed.  Never throws.
 *   - Stateless and re-entrant per stream: safe under one-process-per-GPU data parallelism.
 *   - Activation layout inside the path is NHWC ("pixel-major"): element (b, y, x, c) of a tensor with
 *     pixel stride `ld` lives at ((b*H + y)*W + x)*ld + c.  The plugin boundary (NCHW in / NCHW out,
 *     model/yolo2.py:125-130) is handled by y2_conv0_fwd (reads NCHW) and by the head writing the
 *     [B, rows, cols, A*(5+C)] image that model/__init__.py:122 obtains with permute(0,2,3,1).
 */
#ifndef YOLO2_HIP_H
#define YOLO2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y2_OK 0
#define Y2_EINVAL (-1)    /* bad size / null pointer */
#define Y2_EALIGN (-2)    /* pointer or stride not aligned as required */
#define Y2_ENOSUP (-3)    /* combination not supported by this build */

typedef void* y2_stream_t; /* hipStream_t */

/* Library/ABI version (bump when a signature changes) and the gfx target it was compiled for. */
int y2_abi_version(void);
const char* y2_build_info(void);

/* ------------------------------------------------------------------------------------------------
 * Weight preparation (replaces nothing in the reference: torch keeps [Cout,Cin,kh,kw], model/yolo2.py:57)
 * ------------------------------------------------------------------------------------------------ */

/* Repack a conv weight [Cout][Cin][k][k] (state_dict layout, model/yolo2.py:57) into the GEMM layouts
 * the kernels read.  mode 0 (fprop): dst[co][tap][ci];  mode 1 (dgrad): dst[ci][k*k-1-tap][co]
 * (filters rotated by 180 deg, in/out swapped).  dst holds Cout*Cin*k*k floats. */
int y2_pack_weight(const float* w, float* dst, int Cout, int Cin, int ksize, int mode, y2_stream_t stream);

/* Inverse of mode 0 for a weight GRADIENT: src[co][tap][ci] -> dst[Cout][Cin][k][k]. */
int y2_unpack_weight_grad(const float* src, float* dst, int Cout, int Cin, int ksize, y2_stream_t stream);

/* Fold eval-mode BatchNorm (nn.BatchNorm2d eps=1e-5, model/yolo2.py:58) into a per-channel affine:
 * scale = gamma / sqrt(var + eps), shift = beta - mean*scale. */
int y2_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
               float* scale, float* shift, int C, y2_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution forward: model.yolo2.Conv2d.forward (model/yolo2.py:61-65) = conv(k, stride 1,
 * pad (k-1)/2) -> per-channel affine (folded BN, or conv bias) -> LeakyReLU(slope), with the
 * following MaxPool2d(2) (model/yolo2.py:79,86,97), the passthrough `reorg` (model/yolo2.py:33-46)
 * and the `torch.cat` (model/yolo2.py:129) expressed as output addressing.
 * ------------------------------------------------------------------------------------------------ */
typedef struct y2_conv_params {
    const float* x;      /* input, NHWC, pixel stride ldx (>= Cin) */
    const float* w;      /* packed weights, y2_pack_weight mode 0: [Cout][k*k][Cin] */
    const float* scale;  /* [Cout] or NULL (= 1) */
    const float* shift;  /* [Cout] or NULL (= 0); conv bias goes here */
    float* y;            /* full-resolution output or NULL */
    float* y_pool;       /* 2x2/stride-2 max-pooled output or NULL (needs even H and W) */
    double* stats;       /* NULL, or [2*Cout] doubles (pre-zeroed): sum and sum of squares of the RAW conv
                            output per channel are atomically accumulated (training-mode BN statistics) */
    int32_t B, H, W;     /* input = output spatial size (stride 1, same padding) */
    int32_t Cin, ldx;
    int32_t Cout;
    int32_t ksize;       /* 1 or 3 */
    int32_t ldy, coff;   /* y pixel stride and channel offset (concat write-through) */
    int32_t ldp, poff;   /* y_pool pixel stride and channel offset */
    int32_t out_mode;    /* 0: y[b,y,x,coff+n];  1: reorg(stride 2): y[b,y/2,x/2, coff + ((y&1)*2+(x&1))*Cout + n] */
    float slope;         /* LeakyReLU negative slope; 1.0f = no activation */
    int32_t tile;        /* 0 = auto; else force a tile config (see conv_fwd.hip; benchmarking only) */
} y2_conv_params;

int y2_conv_fwd(const y2_conv_params* p, y2_stream_t stream);

/* The same for `count` convolutions enqueued back to back (one host call for a whole Darknet stage chain;
 * model/yolo2.py:125-130 runs them as separate nn.Module calls). Stops at the first error. */
int y2_conv_fwd_batch(const y2_conv_params* params, int count, y2_stream_t stream);

/* First layer (model/yolo2.py:78, 'layers1.0'): reads the plugin's NCHW fp32 input [B,Cin<=4,H,W] directly,
 * conv3x3 pad 1 -> affine -> LeakyReLU -> (optional 2x2 max-pool), writes NHWC.
 * w is the UNPACKED state_dict weight [Cout][Cin][3][3]; Cout <= 64.  y (full res, pixel stride ldy) and/or
 * y_pool (pixel stride ldp) may be requested; stats as in y2_conv_params. */
int y2_conv0_fwd(const float* x_nchw, const float* w, const float* scale, const float* shift,
                 float* y, float* y_pool, double* stats,
                 int B, int H, int W, int Cin, int Cout, int ldy, int ldp, float slope, y2_stream_t stream);

/* nn.MaxPool2d(kernel_size=2) (model/yolo2.py:79,86,97) on NHWC; H, W even (16-B vector path when C, ldx, ldy are multiples of 4). */
int y2_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, y2_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Detection head decode: model.Inference.forward after self.dnn(x) (model/__init__.py:120-135),
 * softmax of the class logits (detect.py:152, eval.py:270) and the visibility filter
 * (detect.filter_visible, detect.py:51-63) in one pass over the head image.
 * ------------------------------------------------------------------------------------------------ */
/* feature: [B, cells, A, 5+C] (C may be 0: single class), i.e. the NHWC head image.
 * anchors: [A][2] (height, width) in cell units (utils/__init__.py:78-81).
 * Outputs (any may be NULL): iou [B,cells,A]; center_offset, size_norm, yx_min, yx_max [B,cells,A,2];
 * prob [B,cells,A,C] = softmax(logits); prob_cls [B,cells,A] = max_c prob and cls [B,cells,A] its first arg-max
 * (detect.py:52; 1.0 / 0 when C == 0, detect.py:43-48).  Cell k decodes to (k / rows, k % rows) (model/__init__.py:53-56). */
int y2_decode(const float* feature, const float* anchors, int B, int rows, int cols, int A, int C,
              float* iou, float* center_offset, float* size_norm, float* yx_min, float* yx_max, float* prob,
              float* prob_cls, int32_t* cls, y2_stream_t stream);

/* detect.filter_visible (detect.py:51-63): per image, candidates with score > thr in candidate order, where
 * score = iou*max_c prob (fix != 0, thr = threshold_cls) or iou (fix == 0, thr = threshold).
 * Inputs as produced by y2_decode for one batch: iou [B,n], prob [B,n,C] (C >= 1).
 * Outputs: count[B]; index[B,n] (first count[b] entries = surviving candidate indices, ascending);
 * prob_cls[B,n], cls[B,n] (max prob and its first arg-max for EVERY candidate).
 * If prob == NULL, prob_cls is an INPUT (already computed by y2_decode) and cls is not touched. */
int y2_filter_visible(const float* iou, const float* prob, int B, int n, int C, int fix, float thr,
                      int32_t* count, int32_t* index, float* prob_cls, int32_t* cls, y2_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * IoU: utils.iou.torch (utils/iou/torch.py:24-61, 116-153, 216-233).  Boxes are (y, x) min / max pairs.
 * Bit-exact to the fp32 operation order of the reference (no FMA contraction, IEEE division).
 * ------------------------------------------------------------------------------------------------ */
/* batch_iou_matrix: [Bt,N1,2]x2, [Bt,N2,2]x2 -> out [Bt,N1,N2]; iou_matrix is Bt = 1.  min_union = eps32.
 * mode 0: IoU (utils/iou/torch.py:47-61, 139-153);  mode 1: intersection area only (:24-44, 116-136). */
int y2_iou_matrix(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2,
                  int Bt, int N1, int N2, float min_union, int mode, float* out, y2_stream_t stream);
/* batch_iou_pair: [n,2]x4 -> out [n]. */
int y2_iou_pair(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2,
                int n, float min_union, float* out, y2_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NMS: utils.postprocess.nms (utils/postprocess.py:23-49): class-agnostic greedy NMS on `score`,
 * top-`limit` after a descending sort, keep j iff IoU(head, j) <= overlap.  Batched: image b owns
 * candidates [b*stride, b*stride + n[b]); with cand != NULL (the index list of y2_filter_visible) candidate i of
 * image b is element cand[b*stride + i] of the image's score/box rows (no gather pass).  Ties: lower index first.
 * keep [B, limit] receives, in descending-score order, indices RELATIVE to the image's first candidate;
 * keep_count[B] their number.  limit <= 1024.  One workgroup per image.  order_ws: [B, limit] int32 workspace
 * (receives the top-`limit` candidate indices in descending-score order).
 * ------------------------------------------------------------------------------------------------ */
int y2_nms(const float* score, const float* yx_min, const float* yx_max, const int32_t* cand, const int32_t* n, int B, int stride,
           float overlap, int limit, int32_t* order_ws, int32_t* keep, int32_t* keep_count, y2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO2_HIP_H */
