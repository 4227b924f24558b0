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

/* Training-mode BN statistics are accumulated with fp64 atomics into Y2_STATS_REPL replicated copies of the
 * [sum | sum of squares] vector (copy = tile index mod Y2_STATS_REPL) to keep atomic contention off the critical
 * path; y2_bn_finalize adds the copies.  A stats buffer therefore holds Y2_STATS_REPL * 2 * C doubles. */
#define Y2_STATS_REPL 32

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

/* Multi-tensor form for a whole network: ONE launch prepares every listed operand from the state_dict layout [Cout][Cin][k][k]
 * (a training step re-derives ~90 operands from the freshly updated weights; as separate launches they are launch-latency bound).
 * Modes: Y2_PREP_FPROP / Y2_PREP_DGRAD = y2_pack_weight modes 0 / 1;  Y2_PREP_WINO_FPROP / Y2_PREP_WINO_DGRAD = the Winograd
 * filter transform U[16][Cout][Cin] resp. U[16][Cin][Cout] (rotated, in/out swapped) of a 3x3 filter, bit-identical to
 * y2_pack_weight + y2_wino_weight;  Y2_PREP_WINO6_DGRAD = the F(4x4,3x3) data-gradient operand U6[36][Cin][Cout] (rotated, in/out
 * swapped), bit-identical to y2_pack_weight(mode 1) + y2_wino6_weight.
 * dst sizes: Cout*Cin*k*k floats (packs), 16*Cout*Cin floats (2x2-tile transforms), 36*Cout*Cin floats (4x4-tile transform). */
#define Y2_PREP_FPROP 0
#define Y2_PREP_DGRAD 1
#define Y2_PREP_WINO_FPROP 2
#define Y2_PREP_WINO_DGRAD 3
#define Y2_PREP_WINO6_DGRAD 4
#define Y2_PREP_MAX_ITEMS 96
typedef struct {
    const float* src;
    float* dst;
    int32_t Cout, Cin, ksize, mode;
} y2_prep_item;
int y2_prep_weights(const y2_prep_item* items, int32_t count, y2_stream_t stream);

/* Many small range operations in ONE launch (table in the kernel arguments): the zero fills of a training step's accumulation
 * buffers (autograd's zeros, train.py:350), the fp64 -> fp32 hand-out of the affine-parameter gradients, plain copies.
 * n counts ELEMENTS of dst (fp32); src is const double* for Y2_MULTI_F64_TO_F32, const float* for Y2_MULTI_COPY, ignored for ZERO. */
#define Y2_MULTI_ZERO 0
#define Y2_MULTI_F64_TO_F32 1
#define Y2_MULTI_COPY 2
#define Y2_MULTI_MAX_ITEMS 96
typedef struct {
    const void* src;
    float* dst;
    int64_t n;
    int32_t op, reserved;
} y2_multi_item;
int y2_multi(const y2_multi_item* items, int32_t count, y2_stream_t stream);

/* The weighted loss sum of train.py:348-349, `sum(loss[key] * hparam[key] for key in loss)`, over the n <= 64 loss terms in
 * device memory: out[0] = sum_k v[k]*w[k] (left to right); and its gradient out[k] = g[0]*w[k]. */
int y2_small_dot(const float* v, const float* w, int32_t n, float* out, y2_stream_t stream);
int y2_small_scale(const float* g, const float* w, int32_t n, float* out, y2_stream_t stream);

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
    double* stats;       /* NULL, or [Y2_STATS_REPL][2*Cout] doubles (pre-zeroed): sum and sum of squares of the RAW
                            conv output per channel are atomically accumulated (training-mode BN statistics) */
    int32_t B, H, W;     /* INPUT spatial size; output = (H + 2*pad - k)/stride + 1 (equal to the input for stride 1 / same padding) */
    int32_t Cin, ldx;
    int32_t Cout;
    int32_t ksize;       /* 1..7 (1 and 3 with stride 1 / same padding take the specialised path) */
    int32_t ldy, coff;   /* y pixel stride and channel offset (concat write-through) */
    int32_t ldp, poff;   /* y_pool pixel stride and channel offset */
    int32_t out_mode;    /* 0: y[b,y,x,coff+n];  1: reorg(stride 2): y[b,y/2,x/2, coff + ((y&1)*2+(x&1))*Cout + n] */
    float slope;         /* LeakyReLU negative slope; 1.0f = no activation */
    int32_t tile;        /* 0 = auto; else force a tile config (see conv_fwd.hip; benchmarking only).  With the fused Winograd algorithms:
                          * 3 = the third-generation kernel (32-tile x 64-channel units, two workgroups per CU) */
    float* workspace;    /* optional scratch (16-B aligned) for the split-K remainder scheme, or NULL */
    int64_t workspace_bytes; /* its size; y2_conv_fwd_workspace_bytes() tells how much a problem can use */
    const float* residual; /* optional [B,Ho,Wo,Cout] (pixel stride ldr) added before the activation (model/resnet.py:59,101) */
    int32_t ldr;
    int32_t stride;      /* 0 or 1 = stride 1; 2 = the strided convs of model/resnet.py:31,71,111 */
    int32_t pad_plus1;   /* 0 = "same" padding (k-1)/2; otherwise padding + 1 */
    int32_t transposed;  /* != 0: data gradient of a conv with this (ksize, stride, pad): x = dz [B,H,W,Cin=Cout_fwd], w = y2_pack_weight
                            mode 1 of the forward weight, result [B,out_h,out_w,Cout=Cin_fwd] (fractionally strided convolution) */
    int32_t out_h, out_w; /* transposed only: spatial size of the forward conv's input */
    int32_t algo;        /* Y2_ALGO_DIRECT (0): implicit GEMM, w = y2_pack_weight output.
                            Y2_ALGO_WINOGRAD (1): F(2x2,3x3) for 3x3 / stride 1 / same padding: w = y2_wino_weight output
                            [16][Cout][Cin]; `workspace` is REQUIRED (transformed input + products, y2_conv_fwd_workspace_bytes) */
    int64_t w_plane;     /* Y2_ALGO_WINOGRAD_SPLIT only: distance in ELEMENTS between the three bf16 planes of w (0 = 16*Cout*Cin, the layout
                            y2_split_bf16x3 gives one operand; larger when many operands were split as one array) */
} y2_conv_params;

#define Y2_ALGO_DIRECT 0
#define Y2_ALGO_WINOGRAD 1
#define Y2_ALGO_WINOGRAD_FUSED 2   /* same transforms, but GEMM + output transform in ONE kernel (no product tensor in memory); Cin % 32 == 0 */
#define Y2_ALGO_WINOGRAD_IMPLICIT 3 /* as FUSED, and the input transform B^T d B happens in that kernel's operand loader: the transformed input
                                      (4x the input) never goes to memory either.  Cin % 32 == 0, a batch chunk's input < 1 GB.
                                      Bit-identical results to FUSED.  (Leaves no transformed input behind for y2_wino_wgrad.) */
#define Y2_ALGO_WINOGRAD_SPLIT 4   /* as WINOGRAD (three kernels, fp32 transforms), but the 16 GEMMs run on the bf16 matrix pipe with every fp32 operand
                                      split into three bf16 planes and six plane products per multiply ("bf16x6", csrc/gemm_split.hip): fp32-level
                                      accuracy at 2.67x the fp32-MFMA rate.  w = y2_split_bf16x3 of the y2_wino_weight output; Cin % 32 == 0.
                                      Opt-in precision mode of the Python layer (Y2_SPLIT_BF16=1); never chosen by the library itself. */

#define Y2_ALGO_WINOGRAD_F43 6     /* three kernels like WINOGRAD, on 4x4 output tiles: Winograd F(4x4,3x3), 36 GEMMs, 2.25x fewer multiply-adds than
                                      F(2x2,3x3) on even maps; w = y2_wino6_weight output [36][Cout][Cin]; y only (no pool / statistics).  Error 8-9e-6 x rms
                                      per layer in an fp32 model (F(2x2,3x3): 1.3e-6): meant for GRADIENTS (the training step's data gradients), not
                                      for the inference path, whose tolerance it would use up */
#define Y2_ALGO_WINOGRAD_F43_PRE 7 /* as F43, but x IS the transformed input [36][T][Cin] (ldx == Cin, T = y2_wino6_tiles(B, H, W)), written by
                                      y2_bn_act_bwd_wino6: no input-transform launch, no transformed input in the workspace */
#define Y2_ALGO_WINOGRAD_SPLIT_F16 5 /* as SPLIT with fp16 plane PAIRS (hi = fp16(s x), lo = fp16(s x - hi): 2 x 11 bits + the residual's sign) and three
                                      products (hi x hi, hi x lo, lo x hi): half the matrix instructions and 4 instead of 6 operand bytes of SPLIT.
                                      fp16 has 5 exponent bits: the operands carry fixed power-of-two scales (V x 2^-4, w x 2^8 - pass 256 to
                                      y2_split_f16x2 for w; the product is rescaled exactly) so that |V| up to 10^6 and |w| up to 255 stay finite;
                                      beyond that the result is Inf / NaN, and operands far BELOW 1 (gradients) lose their low plane to fp16's subnormals - unlike every
                                      other algorithm.  For activations (inference, training forward).  Opt-in (Y2_SPLIT_F16=1). */

/* Two fp16 planes of src * scale (scale a power of two): dst = [2][n] fp16 (4 n bytes).  The weight operand of Y2_ALGO_WINOGRAD_SPLIT_F16 (scale 256). */
int y2_split_f16x2(const float* src, void* dst, long long n, float scale, y2_stream_t stream);
/* 1 if any value handed to the fp16 split since the last reset was outside fp16's range (the affected results hold Inf / NaN), else 0; reset != 0
 * clears the flag.  Synchronises the device (a check for the end of an epoch / a benchmark, not for the hot path). */
int y2_split_f16_overflow(int reset);
/* y2_gemm_split for fp16 plane pairs A = [2][groups][M][K], B = [2][groups][N][K]; C = (A B^T) * out_scale. */
int y2_gemm_split_f16(const void* A, const void* B, float* C, long long M, int32_t N, int32_t K, int32_t ldc, int32_t groups, float out_scale, y2_stream_t stream);

/* Three bf16 planes of an fp32 array (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even):
 * dst = [3][n] bf16 (6 n bytes), n % 4 == 0, src 16-B aligned.  The weight operand of Y2_ALGO_WINOGRAD_SPLIT. */
int y2_split_bf16x3(const float* src, void* dst, long long n, y2_stream_t stream);

/* C[g] (M x N, row stride ldc, fp32) = A[g] (M x K) * B[g]^T (N x K) for g < groups on the bf16 matrix pipe, operands as plane triples
 * A = [3][groups][M][K], B = [3][groups][N][K] (y2_split_bf16x3 of the fp32 arrays [groups][M][K] / [groups][N][K]); K % 32 == 0.
 * Two accumulator sets per output (hi x hi products / the five cross products) are added once at the end.
 * The GEMM stage of Y2_ALGO_WINOGRAD_SPLIT, exposed for tests and tools. */
int y2_gemm_split(const void* A, const void* B, float* C, long long M, int32_t N, int32_t K, int32_t ldc, int32_t groups, y2_stream_t stream);

/* Winograd F(2x2,3x3) filter transform U[p][co][ci] = (G g G^T)[p], p = 4*xi + nu, from a packed 3x3 weight
 * (y2_pack_weight mode 0 for the forward conv, mode 1 for the data gradient: [Cout][9][Cin]).  Same role as the cuDNN
 * WINOGRAD algorithm the reference's nn.Conv2d (model/yolo2.py:57) may pick for fp32 3x3 convolutions: 2.25x fewer
 * multiplications, results within fp32 rounding of the direct sum (tests state the tolerance). */
int y2_wino_weight(const float* w_packed, float* u, int32_t Cout, int32_t Cin, y2_stream_t stream);
/* ... and for Y2_ALGO_WINOGRAD_F43: u6[36][Cout][Cin] = G g G^T (G 6x3, interpolation points 0, 1, -1, 2, -1/2, inf). */
int y2_wino6_weight(const float* w_packed, float* u6, int32_t Cout, int32_t Cin, y2_stream_t stream);

/* Winograd form of y2_conv_wgrad for a 3x3 / stride-1 / same-padding convolution: dw_packed[Cout][9][Cin] = (not +=) the
 * weight gradient; 16 reductions over ceil(H/2)*ceil(W/2) tiles per image instead of 9 over H*W pixels.  Cin, Cout, ldx,
 * ldz multiples of 4; workspace of y2_wino_wgrad_workspace_bytes() bytes (transformed input, transformed gradient, dU). */
long long y2_wino_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
/* The same for ONE form (native_layout as y2_wino_wgrad_ex takes it; what that entry point requires): the 4x4-tile form needs 36/64 of the 2x2-tile form's rows. */
long long y2_wino_wgrad_workspace_bytes_ex(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t native_layout);
int y2_wino_wgrad(const float* x, const float* dz, float* dw_packed, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t ldx,
                  int32_t Cout, int32_t ldz, const float* v_transformed, float* workspace, long long workspace_bytes, y2_stream_t stream);
/* ... native_layout bit 0: the result is written as dw[Cout][Cin][3][3] - nn.Conv2d.weight.grad's own layout (model/yolo2.py:57), no
 * y2_unpack_weight_grad pass behind it.  Bit 1: Winograd F(3x3, 4x4) - 36 reductions over 4x4 gradient tiles (1.78x fewer multiply-adds than
 * the 2x2 form on even maps); needs x (not v_transformed); its larger transform constants cost accuracy (1.2-1.4e-5 x rms of the gradient in
 * fp32 against 2.7e-6): for weight gradients only.  Bit 2 (with bit 1): `dz` IS the transformed gradient [36][T][Cout] that
 * y2_bn_act_bwd_wino6 wrote (T = y2_wino6_tiles(B, H, W)); ldz is ignored. */
int y2_wino_wgrad_ex(const float* x, const float* dz, float* dw, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t ldx,
                     int32_t Cout, int32_t ldz, const float* v_transformed, float* workspace, long long workspace_bytes, int32_t native_layout,
                     y2_stream_t stream);
/* v_transformed (optional): the transformed input V[16][B*ceil(H/2)*ceil(W/2)][Cin] of the SAME x, as y2_conv_fwd with
 * algo = Y2_ALGO_WINOGRAD leaves it at the start of its workspace when the batch fits one chunk (V + M <= Y2_WINO_CHUNK_MB,
 * default 4096 MB): the training graph keeps that workspace alive and the weight gradient skips the input transform. */

int y2_conv_fwd(const y2_conv_params* p, y2_stream_t stream);

/* Scratch bytes this problem can use (tiles that do not fill the last round of the 256 CUs are split along K into a
 * caller-provided workspace and combined by a fix-up kernel); 0 = none; negative = argument error. */
long long y2_conv_fwd_workspace_bytes(const y2_conv_params* p);

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

/* General max-pool on NHWC with `pad` rows/cols of -inf before and `pad_end` after each spatial axis:
 * nn.MaxPool2d(3, 2, 1) of model/resnet.py:114 is (3, 2, 1, 1); ConstantPad2d((0,1,0,1), float32.min) + MaxPool2d(2, stride=1)
 * of model/yolo2.py:151-152 (Tiny) is (2, 1, 0, 1).  out = (H + pad + pad_end - k)/stride + 1. */
int y2_maxpool_fwd(const float* x, float* y, int B, int H, int W, int C, int ldx, int ldy, int ksize, int stride, int pad, int pad_end, y2_stream_t stream);

/* Plugin input boundary for backbones whose stem runs through y2_conv_fwd (model/resnet.py:111): NCHW [B,C,H,W] ->
 * NHWC with pixel stride ld >= C, padding channels zero-filled. */
int y2_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, int ld, y2_stream_t stream);

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
/* The tail of detect.postprocess after NMS (detect.py:69-79) for a whole batch in one launch: k_* = the surviving boxes of every image
 * (iou, yx_min, yx_max of candidate cand[b][keep[b][k]], k < keep_count[b]; cand NULL: keep holds box indices) and - e_count != NULL,
 * the `[detect] fix` branch (:73-77) - their expansion into detections: every (box, class) pair with iou * prob > threshold_cls in
 * row-major order: e_min / e_max [B][limit*C][2], e_score, e_cls (int64) [B][limit*C], e_count [B]. */
int y2_expand_classes(const float* iou, const float* prob, const float* yx_min, const float* yx_max, const int32_t* cand, const int32_t* keep,
                      const int32_t* keep_count, int32_t B, int32_t n, int32_t C, int32_t limit, float threshold_cls, float* k_iou, float* k_min, float* k_max,
                      float* e_min, float* e_max, float* e_score, long long* e_cls, int32_t* e_count, y2_stream_t stream);

/* eval.matching (eval.py:67-75): best[i], which[i] = max / first arg-max over j of IoU(box1 i, box2 j) (utils/iou/torch.py:47-61 arithmetic). */
int y2_iou_rowmax(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2, int32_t N1, int32_t N2, float min_union,
                  float* best, long long* which, y2_stream_t stream);

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

/* HOST-memory forms of the same algorithms (all pointers are HOST pointers, no stream, no HIP call: safe in a forked child).
 * utils.postprocess.nms is called on CPU tensors by the reference's summary worker (train.py:209) and the utils.iou.torch unit
 * tests run on CPU tensors (utils/iou/torch.py:64-113).  Same ordering rule (score descending, ties -> lower index, NaN last),
 * same fp32 IoU operation sequence: keep lists and IoU values are bit-identical to the device entry points. */
int y2_nms_host(const float* score, const float* yx_min, const float* yx_max, const int32_t* cand, const int32_t* n, int B, int stride,
                float overlap, int limit, int32_t* keep, int32_t* keep_count);
int y2_iou_matrix_host(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2,
                       int Bt, int N1, int N2, float min_union, int mode, float* out);
int y2_iou_pair_host(const float* yx_min1, const float* yx_max1, const float* yx_min2, const float* yx_max2,
                     int n, float min_union, float* out);

/* ------------------------------------------------------------------------------------------------
 * Deterministic mode (opt-in, process-global; one process per GPU, one stream).  By default the split-K weight gradient, the
 * BatchNorm-backward sums, the BatchNorm statistics and the loss sums combine partial results with atomics, in completion order:
 * run-to-run differences of ~1e-6 relative.  y2_set_deterministic(1, ws, bytes) makes y2_conv_wgrad / _ex / y2_wino_wgrad /
 * y2_conv0_wgrad / y2_bn_act_bwd / _ex / y2_region_loss_fwd write their partials into the scratch area `ws` (device memory, 16-B
 * aligned, >= 1 MiB; 256 MiB covers Darknet-19 at batch 64; a call that needs more returns Y2_EINVAL) and add them in a fixed
 * tree.  y2_conv_fwd / y2_conv0_fwd then refuse `stats` (Y2_ENOSUP): the forward statistics are taken by y2_colstats_det over the
 * raw convolution output z [M, C] (row stride ld): stats = [sum z | sum z^2] in fp64, written (not added) to stats[0 .. 2C) -
 * copy 0 of the Y2_STATS_REPL layout y2_bn_finalize reads (the caller keeps the other copies zero).  `workspace` of
 * y2_colstats_det: min(1024, ceil(M / 256)) * 2 * C doubles.  NOT covered: y2_opt_grad_sumsq (gradient-norm clipping) and y2_colsum.
 * ------------------------------------------------------------------------------------------------ */
int y2_set_deterministic(int on, float* workspace, long long workspace_bytes);
int y2_get_deterministic(void);
int y2_colstats_det(const float* z, long long M, int32_t C, int32_t ld, double* stats, float* workspace, long long workspace_bytes, y2_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py, tools/): y2_prof_enable(1) clears the record table and starts bracketing EVERY kernel launch of
 * this library with a HIP event pair on the launch stream; y2_prof_enable(0) stops.  Record i = (kernel name, milliseconds
 * between its two events, multiply-add FLOPs the launch executes: 2*M*N*K of the GEMM it runs, 0 for non-GEMM kernels).
 * y2_prof_get synchronises that record's end event.  Not for use during hipGraph capture.  Process-global tooling state
 * (the only host state of the library besides per-device attribute caches); one process per GPU.
 * ------------------------------------------------------------------------------------------------ */
int y2_prof_enable(int on);
int y2_prof_count(void);
int y2_prof_get(int i, char* name, int name_cap, float* ms, double* flops);
/* Caller-defined non-negative tag stamped on every record made until the next call (e.g. the layer being executed), and its readback. */
int y2_prof_set_tag(int tag);
int y2_prof_get_tag(int i);

/* ------------------------------------------------------------------------------------------------
 * Training path.  The reference trains through torch autograd (train.py:344-357): conv / BN / LeakyReLU /
 * MaxPool backward are PyTorch's; the kernels below are their MI355X-native equivalents, orchestrated by
 * yolo2-pytorch_amd/model/train_graph.py behind the same Python surface (Darknet.forward, model.loss).
 * ------------------------------------------------------------------------------------------------ */

/* Weight gradient of a stride-1 "same" conv (autograd of nn.Conv2d, model/yolo2.py:57):
 * dw[co][tap][ci] = sum_pixels dz[pixel][co] * x[pixel + tap][ci], written in the PACKED layout [Cout][k*k][Cin]
 * (y2_unpack_weight_grad converts to [Cout,Cin,k,k]).  dw must be zero-filled (split-K partials are added atomically).
 * Requires Cin, ldx, Cout, ldz multiples of 4 and 16-B aligned bases (else Y2_EALIGN). */
int y2_conv_wgrad(const float* x, const float* dz, float* dw, int B, int H, int W, int Cin, int ldx, int Cout, int ldz,
                  int ksize, y2_stream_t stream);

/* The same for any kernel size <= 7, stride and padding (model/resnet.py:31,71,79,111): x [B,Hi,Wi,Cin] is the conv INPUT, dz the
 * gradient of its output [B,(Hi+2p-k)/s+1,(Wi+2p-k)/s+1,Cout]. */
int y2_conv_wgrad_ex(const float* x, const float* dz, float* dw, int B, int Hi, int Wi, int Cin, int ldx, int Cout, int ldz,
                     int ksize, int stride, int pad, y2_stream_t stream);

/* Weight gradient of the first layer (model/yolo2.py:78): x is the plugin's NCHW input [B,Cin<=3,H,W], dz NHWC with pixel
 * stride ldz, Cout <= 64.  dw is the state_dict layout [Cout][Cin][3][3], pre-zeroed (partials are added atomically). */
int y2_conv0_wgrad(const float* x_nchw, const float* dz, float* dw, int B, int H, int W, int Cin, int Cout, int ldz, y2_stream_t stream);

/* The first layer's weight gradient WITHOUT a materialised dz (model/yolo2.py:78-79: Conv2d(3, 32) + MaxPool2d(2)): the BatchNorm / LeakyReLU /
 * max-pool backward of y2_bn_act_bwd's second pass runs in this kernel's loader from (z, dy_pool) and the pass-1 sums (call y2_bn_act_bwd
 * with dz = NULL first: it then computes only `sums`).  Same arguments as y2_bn_act_bwd / y2_conv0_wgrad; H, W even; dw pre-zeroed. */
int y2_conv0_wgrad_fused(const float* x_nchw, const float* z, const float* scale, const float* shift, const float* mean, const float* invstd,
                         const float* gamma, float slope, const float* dy_pool, int32_t ldp, const double* sums, float* dw,
                         int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ldz, int32_t has_bn, y2_stream_t stream);

/* Training-mode nn.BatchNorm2d(momentum 0.01, eps 1e-5) statistics (model/yolo2.py:58): stats = Y2_STATS_REPL copies of
 * [sum z | sum z^2] per channel (fp64, accumulated by y2_conv_fwd / y2_conv0_fwd), count = B*H*W.  Writes the affine (scale, shift) used for
 * normalisation (biased variance), saves mean / invstd for backward and updates the running statistics in place
 * (unbiased variance) when running_mean != NULL; num_batches_tracked (int64 scalar in device memory, may be NULL) is the module's
 * step counter, incremented by one (torch/nn/modules/batchnorm.py, as called at model/yolo2.py:58). */
int y2_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* scale, float* shift, float* mean, float* invstd, int C, long long* num_batches_tracked, y2_stream_t stream);

/* y = LeakyReLU(z*scale + shift) on NHWC (scale/shift NULL = identity), optionally with the following MaxPool2d(2)
 * (y_pool) and/or the reorg/concat output addressing of y2_conv_params (out_mode, ldy, coff). */
int y2_bn_act_fwd(const float* z, const float* scale, const float* shift, float slope, float* y, float* y_pool,
                  int B, int H, int W, int C, int ldz, int ldy, int coff, int ldp, int poff, int out_mode, y2_stream_t stream);

/* Backward of (BN train) -> LeakyReLU -> [MaxPool2d(2)]: from dy_full (gradient of the full-resolution activation;
 * fmode 1: stored reorg'ed in a [B,H/2,W/2,ldf] buffer at channel foff) and/or dy_pool (gradient of the pooled
 * activation, routed to the first maximal window element like nn.MaxPool2d) to dz (gradient of the raw conv output).
 * sums [2C] (pre-zeroed fp64) receives sum(g) = d beta (or d bias) and sum(g*zhat) = d gamma.  has_bn = 0: plain
 * bias + LeakyReLU block (dz = g); dz = NULL: only the sums (pass 1; see y2_conv0_wgrad_fused); has_bn = 2: BatchNorm with FROZEN statistics (eval()-mode module with autograd recording,
 * receptive_field_analyzer.py:67,87): mean / invstd are the running statistics, dz = g * gamma * invstd. */
int y2_bn_act_bwd(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                  float slope, const float* dy_full, int ldf, int foff, int fmode, const float* dy_pool, int ldp, int poff,
                  double* sums, float* dz, int ldd, int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream);

/* Extended forms for residual networks (model/resnet.py:29-104): `residual` (pixel stride ldr) is added before the
 * activation in the forward; the backward takes an optional second full-resolution gradient source dy_full2 (fan-out),
 * uses the residual to rebuild the activation mask and can emit dres = gradient w.r.t. the residual input.
 * Aliasing: dz may be one of the gradient inputs (the op is element-wise); a dense dz (ldd == C) that overlaps NO input
 * additionally serves as scratch for the per-workgroup partial sums of pass 1 (an overlapping dz falls back to atomics). */
int y2_bn_act_fwd_ex(const float* z, const float* scale, const float* shift, float slope, const float* residual, int ldr, float* y, float* y_pool,
                     int B, int H, int W, int C, int ldz, int ldy, int coff, int ldp, int poff, int out_mode, y2_stream_t stream);
int y2_bn_act_bwd_ex(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                     float slope, const float* dy_full, int ldf, int foff, int fmode, const float* dy_pool, int ldp, int poff,
                     const float* dy_full2, int ld2, const float* residual, int ldr, float* dres, int lddr,
                     double* sums, float* dz, int ldd, int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream);

/* Tile count T of the 4x4-tile Winograd forms for a batch of B maps of H x W (per-image grid, or the batch mosaic where that is smaller: csrc/common.h,
 * Wino6Grid) - the row count of the transformed operands [36][T][C] of Y2_ALGO_WINOGRAD_F43(_PRE) / y2_wino_wgrad_ex (bits 1, 2). */
long long y2_wino6_tiles(int32_t B, int32_t H, int32_t W);
/* y2_bn_act_bwd for a block whose gradient dz is consumed ONLY through the two 4x4-tile Winograd transforms (autograd of conv -> BatchNorm(batch
 * statistics) -> LeakyReLU, model/yolo2.py:57-65, in front of a 3x3 convolution whose weight gradient runs F(3x3,4x4) and whose data gradient runs
 * F(4x4,3x3)): pass 1 (sums [2C], pre-zeroed) as y2_bn_act_bwd, then ONE kernel forms dz per pixel and stores v6 = B^T dz B [36][T][C] (the operand of
 * Y2_ALGO_WINOGRAD_F43_PRE) and / or m6 = G dz G^T [36][T][C] (the operand of y2_wino_wgrad_ex bit 2) - bit-identical to y2_bn_act_bwd followed by the
 * transform kernels of those two entry points, without the dz tensor (written once, read twice) and two launches.  Un-pooled gradient source with plain
 * addressing only (dy_full, pixel stride ldf, channel offset foff).  dz: optional plain gradient [B,H,W,C] (pixel stride ldd) for a third consumer, or NULL.
 * The buffer of v6 (else m6) is scratch for pass 1 before it is written.  Not available in deterministic mode. */
int y2_bn_act_bwd_wino6(const float* z, const float* scale, const float* shift, const float* mean, const float* invstd, const float* gamma,
                        float slope, const float* dy_full, int ldf, int foff, double* sums, float* v6, float* m6, float* dz, int ldd,
                        int B, int H, int W, int C, int ldz, int has_bn, y2_stream_t stream);

/* Backward of y2_maxpool_fwd (overlapping windows allowed): dx[B,H,W,C] from dy (+ optional dy2) [B,Ho,Wo,C]; the gradient of a
 * window goes to its first maximum (ATen semantics). */
int y2_maxpool_bwd(const float* x, const float* dy, const float* dy2, float* dx, int B, int H, int W, int C, int ldx, int ldy, int lddx,
                   int ksize, int stride, int pad, int pad_end, y2_stream_t stream);

/* out[c] += sum_m x[m*ld + c] (fp64; conv-bias gradient of the head, model/yolo2.py:112).  out pre-zeroed. */
int y2_colsum(const float* x, long long M, int C, int ld, double* out, y2_stream_t stream);
/* dst[i] = (float)(src[i] * mul) */
int y2_f64_to_f32(const double* src, float* dst, int n, double mul, y2_stream_t stream);

/* Gradient of the decode (model/__init__.py:122-135) w.r.t. the head image [boxes, 5+C] from the gradients of
 * iou [boxes], center_offset / size_norm [boxes,2], logits [boxes,C] (any may be NULL = zero); yx_min / yx_max carry
 * no gradient (the reference's loss detaches them, model/__init__.py:142). */
int y2_decode_bwd(const float* iou, const float* center_offset, const float* d_iou, const float* d_center_offset, const float* d_size_norm,
                  const float* d_logits, float* d_feature, int boxes, int C, y2_stream_t stream);

/* model.loss (model/__init__.py:138-167) = iou_match (:59-73) + fit_positive (:76-95) + fill_norm (:98-103) + the five
 * terms, each divided by cnt = B*cells*A; cls is the MEAN cross entropy over positives (gt_cls int64 [B,N]) or the
 * summed squared error on softmax (gt_onehot [B,N,C]); C = 0: no cls term.  GT boxes [B,N,2] in cell units, zero rows =
 * padding.  Outputs: loss_out[5] = foreground, background, center, size, cls; saved for backward: best_iou [B,n],
 * best_idx [B,n], positive [B,n] (uint8), sums[6] (fp64: the five sums and the number of positives). n = rows*cols*A. */
int y2_region_loss_fwd(const float* iou, const float* center_offset, const float* size_norm, const float* logits,
                       const float* yx_min, const float* yx_max,
                       const float* gt_yx_min, const float* gt_yx_max, const int64_t* gt_cls, const float* gt_onehot,
                       const float* anchors, int B, int rows, int cols, int A, int C, int N, float threshold,
                       float* best_iou, int32_t* best_idx, uint8_t* positive, double* sums, float* loss_out, y2_stream_t stream);

/* Data-parallel training: after sums[5] (number of positives) has been summed over ranks, recompute loss_out so that
 * the mean-over-positives of the cls term uses the GLOBAL count (cnt stays the local B*cells*A). */
int y2_region_loss_finalize(const double* sums, double cnt, int cross_entropy, float* loss_out, y2_stream_t stream);

/* Gradients of sum_k weights[k]*loss[k] (train.py:348-349) w.r.t. iou, center_offset, size_norm, logits. */
int y2_region_loss_bwd(const float* iou, const float* center_offset, const float* size_norm, const float* logits,
                       const float* gt_yx_min, const float* gt_yx_max, const int64_t* gt_cls, const float* gt_onehot,
                       const float* anchors, int B, int rows, int cols, int A, int C, int N, float threshold,
                       const float* best_iou, const int32_t* best_idx, const uint8_t* positive, const double* sums, const float* weights,
                       float* d_iou, float* d_center_offset, float* d_size_norm, float* d_logits, y2_stream_t stream);

/* ---- fused multi-tensor optimizer -------------------------------------------------------------------
 * One launch updates up to Y2_OPT_MAX_TENSORS parameter tensors (the pointer table is passed by value in the kernel
 * arguments).  Replaces the per-tensor loops of `self.optimizer.step()` (train.py:357; optimizer from the ini lambda,
 * config.ini:72) and `nn.utils.clip_grad_norm` (train.py:352-354).  All tensors fp32, contiguous. */
#define Y2_OPT_MAX_TENSORS 48
typedef struct {
    float* param;
    float* grad;
    float* state1;   /* SGD: momentum buffer (NULL when momentum == 0); Adam: exp_avg */
    float* state2;   /* Adam: exp_avg_sq; SGD: NULL */
    int64_t numel;
} y2_opt_tensor;

/* torch.optim.SGD: d = g + wd*p; buf = first_step ? d : momentum*buf + (1-dampening)*d; d = nesterov ? d + momentum*buf : buf; p -= lr*d */
int y2_opt_sgd(const y2_opt_tensor* tensors, int32_t count, float lr, float momentum, float dampening, float weight_decay,
               int32_t nesterov, int32_t first_step, y2_stream_t stream);
/* torch.optim.Adam (no amsgrad), `step` = 1-based step count of these tensors (bias corrections computed in double on the host) */
int y2_opt_adam(const y2_opt_tensor* tensors, int32_t count, float lr, float beta1, float beta2, float eps, float weight_decay,
                int32_t step, y2_stream_t stream);
/* *sumsq (fp64, pre-zeroed, accumulated across calls) += sum of squares of all gradients; then y2_opt_clip_grads scales every
 * gradient by max_norm / (sqrt(*sumsq) + 1e-6) when that is < 1 (torch.nn.utils.clip_grad_norm_, L2).  No host sync. */
int y2_opt_grad_sumsq(const y2_opt_tensor* tensors, int32_t count, double* sumsq, y2_stream_t stream);
int y2_opt_clip_grads(const y2_opt_tensor* tensors, int32_t count, const double* sumsq, float max_norm, y2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO2_HIP_H */
