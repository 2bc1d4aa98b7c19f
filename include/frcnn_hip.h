/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so: the MI355X (gfx950) Faster R-CNN
 * proposal / RoI-head hot path.
 *
 * The reference (csm-kr/faster_rcnn_pytorch) is pure Python and has no FFI; its
 * boundary for this path is the Python call surface of models/model.py + anchor.py
 * and, beneath it, five torchvision entry points.  Every function below names the
 * reference interface (file:line under /root/reference) it replaces; the ctypes
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - extern "C"; plain pointers and sizes; no torch types.  Returns 0 on success,
 *    a negative frcnn_status otherwise; never throws, never allocates device
 *    memory, never synchronises the stream (unless stated).
 *  - Every pointer is a DEVICE pointer unless the name ends in _host.  The caller
 *    owns every buffer, workspaces included.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *  - Variable-length results use a fixed-capacity buffer + a device-side int32
 *    count, so a whole training step can be enqueued without a host round trip.
 *  - Boxes are fp32 xyxy, normalised to [0,1] by image (w,h) unless stated;
 *    indices are int64 (torch.long in the reference), argmax is int32.
 *  - Arithmetic is IEEE binary32 with no FMA contraction (the reference's eager
 *    op chains round after every op); exp / log2 are the fixed +-*-/ sequences
 *    restated in oracle/frcnn_oracle.c, so integer results (sort order, NMS keep
 *    lists, level ids, labels) are bit-exact against the oracle.
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_ABI_VERSION 7

typedef enum {
    FRCNN_OK = 0,
    FRCNN_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, K > capacity ... */
    FRCNN_ERR_UNSUPPORTED = -2,   /* size outside what the kernels are built for */
    FRCNN_ERR_WORKSPACE = -3,     /* workspace too small (see frcnn_workspace_bytes) */
    FRCNN_ERR_LAUNCH = -4         /* hipGetLastError() after a launch was not hipSuccess */
} frcnn_status;

typedef enum { FRCNN_DTYPE_F32 = 0, FRCNN_DTYPE_BF16 = 1 } frcnn_dtype;

/* operation ids for frcnn_workspace_bytes() */
typedef enum {
    FRCNN_OP_TOPK = 1,            /* n1 = N candidates */
    FRCNN_OP_NMS = 2,             /* n1 = K boxes */
    FRCNN_OP_REGION_PROPOSAL = 3, /* n1 = N anchors, n2 = K */
    FRCNN_OP_RPN_TARGETS = 4,     /* n1 = N anchors, n2 = G */
    FRCNN_OP_HEAD_TARGETS = 5,    /* n1 = P + G candidates */
    FRCNN_OP_PREPROCESS = 6,      /* n1 = (h << 32) | w of the source frame, n2 = (oh << 32) | ow of the resized frame */
    FRCNN_OP_HEAD_BWD = 7,        /* n1 = C (frcnn_rpn_head_tail_ml_bwd) */
    FRCNN_OP_RPN_CONV = 8,        /* frcnn_rpn_conv_head_fwd / frcnn_rpn_conv_bwd_data (packed bf16 weights) */
    FRCNN_OP_RPN_CONV_WGRAD = 9,  /* frcnn_rpn_conv_wgrad (per-split partial gradients) */
    FRCNN_OP_RPN_CONV_F32 = 10    /* n1 = C: frcnn_rpn_conv3x3_f32_fwd / _bwd_data / _wgrad (ticket words, transposed weights, slabs) */
} frcnn_op;

/* FRCNN_ABI_VERSION, or FRCNN_ERR_UNSUPPORTED (message in frcnn_last_error) when the objects the library was linked from were compiled
 * against different versions of its internal headers (a stale object; csrc/frcnn_layout.h): a binding must refuse such a library. */
int frcnn_abi_version(void);
/* The same check by itself (FRCNN_OK / FRCNN_ERR_UNSUPPORTED), and the stamp api.o was compiled with: a 64-bit hash of the sizes, field
 * offsets and constants of every structure two translation units share by layout (AnchorDesc, the sample sort's control block + plan). */
int frcnn_layout_check(void);
uint64_t frcnn_layout_stamp(void);
/* thread-local description of the last non-zero status returned on this thread */
const char *frcnn_last_error(void);
size_t frcnn_workspace_bytes(int op, int64_t n1, int64_t n2);

/* ---- anchors -------------------------------------------------------------------------------- */
/* FRCNNAnchorMaker.generate_anchor_base (anchor.py:15-32).  Host computation (done once).       */
int frcnn_anchor_base_host(int base_size, const double *ratios_host, int n_ratios,
                           const double *scales_host, int n_scales, float *out_host /*[nr*ns,4]*/);
/* torchvision AnchorGenerator.generate_anchors for ONE size (models/new_model.py:23-25). Host.  */
int frcnn_tv_base_anchors_host(float size, const float *ratios_host, int n_ratios, float *out_host /*[nr,4]*/);

/* Anchor grid over n_levels feature maps, level-major, position-major (y, x), base-minor:
 *   out[off_l + (y*fw_l + x)*A + a] = (base_l[a] + (x*sw_l, y*sh_l, x*sw_l, y*sh_l)) / (div_w, div_h, div_w, div_h)
 * One level, A = 9, stride 16, div = (W,H)  == FRCNNAnchorMaker._enumerate_shifted_anchor (anchor.py:34-55);
 * five levels, A = 3, strides (H//fh, W//fw) == AnchorGenerator(...)(ImageList, feats) followed by the
 * in-place division at models/new_model.py:46-47.  base_host: [n_levels, A, 4].  div = 1 -> pixels. */
int frcnn_anchor_grid(int n_levels, const int *fh_host, const int *fw_host,
                      const int *stride_h_host, const int *stride_w_host,
                      const float *base_host, int A, float div_w, float div_h,
                      float *out /*[N,4]*/, int64_t N, void *stream);

/* ---- box codec (utils/util.py:15-50) ---------------------------------------------------------- */
/* op: 0 xy_to_cxcy(a) | 1 cxcy_to_xy(a) | 2 decode(a = tcxcy, b = center_anchor) | 3 encode(a = gt_cxcy, b = anc_cxcy) */
int frcnn_box_codec(int op, const float *a, const float *b, int64_t n, float *out, void *stream);

/* find_jaccard_overlap (utils/util.py:66-102; eps = 1e-5) / box_iou (util/box_ops.py:24-37; eps = 0). */
int frcnn_pairwise_iou(const float *set1, int64_t n1, const float *set2, int64_t n2, float eps,
                       float *out /*[n1,n2]*/, void *stream);

/* ---- RegionProposal.forward (models/model.py:12-58, models/new_model.py:49-86) ------------------ */
/* Stage 1: fg softmax + decode + clamp + min-size filter (model.py:20-41).
 * anchors == NULL is not allowed here; see frcnn_region_proposal for the fused anchor-free form.
 * out_scores[i] = softmax(cls[i])[1], or -1 where (w < min_size_norm or h < min_size_norm).       */
int frcnn_proposal_prologue(const float *reg /*[N,4]*/, const float *cls /*[N,2]*/, const float *anchors /*[N,4]*/,
                            int64_t N, float min_size_norm, float *out_boxes /*[N,4]*/, float *out_scores /*[N]*/,
                            void *stream);

/* Stage 2: scores.sort(descending=True)[:K] (model.py:44-49).  Ties are ordered by ascending index
 * (torch's sort is unstable).  Entries with score < 0 never appear.
 * out_count = min(K, #valid).  boxes_in/out_boxes may both be NULL (no gather).                    */
int frcnn_topk_sorted(const float *scores /*[N]*/, const float *boxes_in /*[N,4] or NULL*/, int64_t N, int64_t K,
                      int64_t *out_idx /*[K]*/, float *out_scores /*[K]*/, float *out_boxes /*[K,4] or NULL*/,
                      int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream);

/* Full descending argsort (every score live, negatives included): the sort inside torchvision.ops.nms
 * when the caller's boxes are not pre-sorted (per-class NMS, models/model.py:394).  Ties: ascending index. */
int frcnn_argsort_desc(const float *scores /*[N]*/, const float *boxes_in /*[N,4] or NULL*/, int64_t N,
                       int64_t *out_idx /*[N]*/, float *out_scores /*[N]*/, float *out_boxes /*[N,4] or NULL*/,
                       int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream);

/* Stage 3: torchvision.ops.nms (model.py:53,394) on boxes already in visiting (score-descending) order.
 * Greedy, suppress when inter/(area_i+area_j-inter) > thr (strict).  Writes the first post_k kept
 * positions (ascending) to out_keep, their boxes to out_rois (optional) and the number to out_count.
 * n_boxes_dev (optional) is a device int32 with the live box count (<= K), e.g. frcnn_topk_sorted's count. */
int frcnn_nms(const float *boxes /*[K,4]*/, const int32_t *n_boxes_dev, int64_t K, float iou_threshold, int64_t post_k,
              int64_t *out_keep /*[post_k]*/, float *out_rois /*[post_k,4] or NULL*/, int32_t *out_count,
              void *workspace, size_t workspace_bytes, void *stream);

/* Per-class NMS of FRCNN._suppress (models/model.py:382-402; torchvision batched_nms semantics) in ONE launch pair
 * instead of C-1 nms calls + C-1 device-to-host copies: like frcnn_nms, but box j is suppressed only by a kept box i
 * with cls[i] == cls[j].  boxes / cls are in visiting (score-descending) order.                                      */
int frcnn_nms_classed(const float *boxes /*[K,4]*/, const int32_t *cls /*[K]*/, const int32_t *n_boxes_dev, int64_t K,
                      float iou_threshold, int64_t post_k, int64_t *out_keep, float *out_rois /*or NULL*/, int32_t *out_count,
                      void *workspace, size_t workspace_bytes, void *stream);

/* All of RegionProposal.forward in one call (prologue -> top-K -> NMS -> first P), no host sync.
 * anchors may be NULL when the single-level grid description is given (fh,fw,stride,base9x4 on host):
 * the anchors are then regenerated in registers and never read from HBM (anchor.py:34-55 fused away).
 * nms_level_offsets_host == NULL (default): ONE class-agnostic NMS over the K boxes of all levels -- what the reference does
 *   (models/new_model.py:74-83).
 * nms_level_offsets_host = [n_nms_levels + 1] anchor offsets (level l owns anchors [off[l], off[l+1])): the OPTIONAL per-FPN-level
 *   variant of BASELINE.json configs[3]: after the same global top-K, box j is suppressed only by a kept box of ITS OWN level
 *   (torchvision's RPN batched_nms over level ids), then the first P in score order.  n_nms_levels <= 8.                        */
int frcnn_region_proposal(const float *reg, const float *cls, const float *anchors /*[N,4] or NULL*/, int64_t N,
                          int fh, int fw, int stride, const float *base_host /*[A,4] or NULL*/, int A,
                          float div_w, float div_h,
                          float min_size_norm, int64_t pre_nms_top_k, float iou_threshold, int64_t post_nms_top_k,
                          const int64_t *nms_level_offsets_host /*[n_nms_levels + 1] or NULL*/, int n_nms_levels,
                          float *out_rois /*[P,4]*/, int32_t *out_count,
                          int64_t *out_src_idx /*[P] anchor index of each roi, or NULL*/,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- RPN head tail (models/model.py:79-83; models/new_model.py:109-113) ---------------------------------------- */
/* conv_raw [C,P] = the 3x3 inter_layer convolution WITHOUT its bias (NCHW, P = fh*fw).  Computes
 *   h = relu(conv_raw + b3);  cls = w_cls[n_cls,C] h + b_cls;  reg = w_reg[n_reg,C] h + b_reg
 * on the fp32 matrix cores and stores them as [P, n_cls] / [P, n_reg] row-major, i.e. exactly
 * pred.permute(0,2,3,1).contiguous().view(1,-1,2|4).  C % 64 == 0, n_cls + n_reg <= 64.                          */
int frcnn_rpn_head_tail_fwd(const float *conv_raw, int C, int64_t P, const float *b3,
                            const float *w_cls, const float *b_cls, int n_cls,
                            const float *w_reg, const float *b_reg, int n_reg,
                            float *out_cls, float *out_reg, void *stream);
/* The same for all FPN levels in one launch (models/new_model.py:37-44: the shared head applied per level, outputs
 * concatenated along the anchor axis).  conv_raw_levels / P_levels: HOST arrays (n_levels <= 5) of device pointers
 * [C, P_l] and position counts.  dtype: element type of the conv outputs (FRCNN_DTYPE_F32 | FRCNN_DTYPE_BF16);
 * mfma: FRCNN_DTYPE_F32 = exact fp32 contraction, FRCNN_DTYPE_BF16 = operands rounded to bf16, fp32 accumulate (the
 * mixed-precision configuration of BASELINE.json configs[4]; biases, outputs and everything downstream stay fp32). */
int frcnn_rpn_head_tail_ml_fwd(const void *const *conv_raw_levels, int dtype, int mfma, int C, const int64_t *P_levels, int n_levels,
                               const float *b3, const float *w_cls, const float *b_cls, int n_cls, const float *w_reg,
                               const float *b_reg, int n_reg, float *out_cls, float *out_reg, void *stream);

/* Backward of frcnn_rpn_head_tail_ml_fwd (what autograd derives from models/model.py:79-83 / new_model.py:109-113): given the
 * gradients of the two outputs (g_cls [sum P_l, n_cls], g_reg [sum P_l, n_reg], the outputs' own layout) it writes
 *   d_raw_levels[l] [C, P_l]  gradient of the bias-free 3x3 output (same dtype as conv_raw_levels: 0 = f32, 1 = bf16),
 *   dw_cls [n_cls, C], db_cls [n_cls], dw_reg [n_reg, C], db_reg [n_reg], db3 [C]  (fp32, fully overwritten).
 * C must be 256 or 512; workspace >= frcnn_workspace_bytes(FRCNN_OP_HEAD_BWD, C, 0).  Exact fp32 MFMA; sums in a fixed order. */
int frcnn_rpn_head_tail_ml_bwd(const void *const *conv_raw_levels, void *const *d_raw_levels, int dtype, int C, const int64_t *P_levels,
                               int n_levels, const float *b3, const float *w_cls, int n_cls, const float *w_reg, int n_reg,
                               const float *g_cls, const float *g_reg, float *dw_cls, float *db_cls, float *dw_reg, float *db_reg,
                               float *db3, void *workspace, size_t workspace_bytes, void *stream);

/* The whole FPN RPN head (models/new_model.py:89-114) in the bf16 mixed-precision configuration as one MFMA implicit-GEMM kernel:
 * 3x3 conv (256 -> 256, bf16 operands, fp32 accumulate) -> raw_levels (bf16, the bias-free conv output kept for backward) ->
 * bias + ReLU + both 1x1 heads -> out_cls [sum P_l, n_cls], out_reg [sum P_l, n_reg] (fp32, the layout of frcnn_rpn_head_tail_ml_fwd).
 * feat_levels / raw_levels: [256, H_l, W_l] bf16 NCHW; w3 [256,256,3,3], w_cls [n_cls,256], w_reg [n_reg,256], biases: fp32.
 * workspace >= frcnn_workspace_bytes(FRCNN_OP_RPN_CONV, 0, 0) (the weights re-packed to bf16 on every call). */
int frcnn_rpn_conv_head_fwd(const void *const *feat_levels_bf16, void *const *raw_levels_bf16, const int *H_host, const int *W_host, int n_levels,
                            int C, const float *w3, const float *b3, const float *w_cls, const float *b_cls, int n_cls, const float *w_reg,
                            const float *b_reg, int n_reg, float *out_cls, float *out_reg, void *workspace, size_t workspace_bytes, void *stream);

/* Backward-data of that 3x3 convolution (what autograd derives for `inter_layer`, models/new_model.py:96,109) on the same implicit-GEMM
 * kernel: d_feat[ci] = conv3x3(d_raw, W3 transposed and flipped), bf16 operands, fp32 accumulate, bf16 output.  d_raw_levels /
 * d_feat_levels: [256, H_l, W_l] bf16 NCHW.  workspace >= frcnn_workspace_bytes(FRCNN_OP_RPN_CONV, 0, 0).                         */
int frcnn_rpn_conv_bwd_data(const void *const *d_raw_levels_bf16, void *const *d_feat_levels_bf16, const int *H_host, const int *W_host, int n_levels,
                            int C, const float *w3, void *workspace, size_t workspace_bytes, void *stream);

/* Weight gradient of that convolution: dw3 [256,256,3,3] fp32 (fully overwritten) = sum over levels and positions of
 * d_raw[co](y, x) * feat[ci](y + ky - 1, x + kx - 1), bf16 operands, fp32 accumulate, split over the positions with a fixed-order
 * finalize (bit-reproducible).  workspace >= frcnn_workspace_bytes(FRCNN_OP_RPN_CONV_WGRAD, 0, 0).                                */
int frcnn_rpn_conv_wgrad(const void *const *feat_levels_bf16, const void *const *d_raw_levels_bf16, const int *H_host, const int *W_host, int n_levels,
                         int C, float *dw3, void *workspace, size_t workspace_bytes, void *stream);

/* The RPN head's 3x3 convolution in fp32 -- `self.inter_layer` of models/model.py:68-70,79 (512 -> 512 on the VGG16 map) and
 * models/new_model.py:96-98,109 (256 -> 256 on the FPN levels), WITHOUT its bias (frcnn_rpn_head_tail_* adds it) -- on
 * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate, sums in a fixed order (bit-reproducible; a k-ordered fmaf chain per
 * output inside a K range, K ranges added in range order).  feat_levels / out_levels: [C, H_l, W_l] fp32 NCHW, batch 1; w3 [C, C, 3, 3]
 * in the reference's layout; C a multiple of 128.
 * The three calls share ONE workspace of frcnn_rpn_conv3x3_f32_workspace(H, W, n_levels, C) bytes that is DEDICATED to them and ZERO
 * before the first call (its ticket words are left zero by every call); calls on one workspace must be stream-ordered.
 *   _fwd      : out[co] = sum_ci conv3x3(feat[ci], w3[co][ci]), padding 1.
 *   _bwd_data : d_feat[ci] = sum_co conv3x3(d_out[co], w3[co][ci] flipped): what autograd derives for the input.
 *   _wgrad    : dw3[co][ci][ky][kx] = sum over levels and positions of d_out[co](y, x) * feat[ci](y + ky - 1, x + kx - 1) (fully overwritten). */
size_t frcnn_rpn_conv3x3_f32_workspace(const int *H_host, const int *W_host, int n_levels, int C);
int frcnn_rpn_conv3x3_f32_fwd(const float *const *feat_levels, float *const *out_levels, const int *H_host, const int *W_host, int n_levels, int C,
                              const float *w3, void *workspace, size_t workspace_bytes, void *stream);
int frcnn_rpn_conv3x3_f32_bwd_data(const float *const *d_out_levels, float *const *d_feat_levels, const int *H_host, const int *W_host, int n_levels,
                                   int C, const float *w3, void *workspace, size_t workspace_bytes, void *stream);
int frcnn_rpn_conv3x3_f32_wgrad(const float *const *feat_levels, const float *const *d_out_levels, const int *H_host, const int *W_host, int n_levels,
                                int C, float *dw3, void *workspace, size_t workspace_bytes, void *stream);

/* The same stage for the backbone's own 3x3 convolutions (padding 1, stride 1, batch 1, fp32): the layers the reference takes from torchvision --
 * `self.extractor = nn.Sequential(*list(vgg16.features)[:-1])` models/model.py:279-281 (conv3_x 256 channels on 150 x 250, conv4_x / conv5_x 512
 * channels on 75 x 125 / 37 x 62 at 600 x 1000) and the FPN's output convolutions behind models/new_model.py:372 (256 -> 256 on P2..P5) -- which
 * are 2/3 of the training step's GPU time through the vendor library.  w [Cout, Cin, 3, 3] in the reference's layout.
 *   _fwd      : y[co] = act(bias[co] + sum_ci conv3x3(x[ci], w[co][ci])); bias may be NULL; act = ReLU when relu != 0 (the Conv2d + ReLU(inplace)
 *               pair of vgg16.features in one pass).  Cin a multiple of 32, Cout a multiple of 64 (64-row GEMM tiles where a side is not a multiple of 128).
 *               relu_bits (optional, frcnn_conv3x3_f32_relu_bits_words(H, W, n_levels, Cout) uint16 words, caller-owned): the sign pattern of the
 *               ReLU outputs, one word per (channel, output tile) -- all the backward needs of them, at 1/32 of their bytes.
 *   _bwd_data : dx[ci] = sum_co conv3x3(g[co], w[co][ci] flipped), g = dy where the forward's relu_bits are set (autograd's threshold_backward
 *               folded into the input transform), or g = dy when relu_bits is NULL.  Cin a multiple of 64, Cout of 32.
 *   _wgrad    : dw[co][ci][ky][kx] = sum over levels and positions of g[co](y, x) * x[ci](y + ky - 1, x + kx - 1) (fully overwritten); dbias[co] =
 *               sum of g[co] (NULL: not wanted).  Cin and Cout multiples of 64.
 * Workspace: frcnn_conv3x3_f32_workspace(H, W, n_levels, Cin, Cout) bytes, the same DEDICATED zero-before-first-use block as the RPN calls
 * above (one block sized for the largest layer serves all of them; calls on it must be stream-ordered).  Bit-reproducible.                  */
size_t frcnn_conv3x3_f32_workspace(const int *H_host, const int *W_host, int n_levels, int Cin, int Cout);
/* x_transformed (optional, frcnn_conv3x3_f32_xt_floats(H, W, n_levels, Cin) floats, caller-owned): _fwd leaves the transformed activations
 * B^T d B there instead of in the scratch workspace, and a later _wgrad of the same layer given the same buffer skips transforming them
 * again -- 0.6 GB per VGG16 step on a 288 GB device for one launch less per layer.  NULL: scratch / transform again. */
size_t frcnn_conv3x3_f32_xt_floats(const int *H_host, const int *W_host, int n_levels, int Cin);
/* u_rotated (optional, frcnn_conv3x3_f32_u_floats floats, caller-owned): _fwd also leaves the transformed ROTATED weights there (same launch), and a later
 * _bwd_data of the same layer given the buffer starts without its weight transform.  NULL: _bwd_data transforms the weights itself. */
size_t frcnn_conv3x3_f32_u_floats(const int *H_host, const int *W_host, int n_levels, int Cin, int Cout);
/* dy_transformed (optional, frcnn_conv3x3_f32_xt_floats(H, W, n_levels, Cout) floats, caller-owned): _bwd_data stages the output gradient once and also writes
 * its weight-gradient transform A g A^T there (with want_bias_partials != 0 also the bias gradient's partial sums, in the workspace); a _wgrad of the same
 * layer called NEXT on the same workspace with that buffer skips its own pass over the gradient.  NULL: each call transforms the gradient itself. */
size_t frcnn_conv3x3_f32_relu_bits_words(const int *H_host, const int *W_host, int n_levels, int Cout);
/* relu = 2 (fwd): ReLU + max_pool2d(2, 2) (floor) in the output transform -- the MaxPool2d behind conv1_2 / conv2_2 / conv3_3 / conv4_3 of vgg16.features:
 * y_levels are then [Cout, H/2, W/2], the full-resolution activations are never written, and relu_bits holds per 2 x 2 window the position of the
 * maximum and whether it was positive.  The gradient calls take `pooled` = 1 with those words: dy_levels are [Cout, H/2, W/2] and max_pool2d's and the
 * ReLU's backward happen while the gradient is staged (H, W stay the convolution's own size everywhere).  Needs the 4 x 4 tile: frcnn_conv3x3_f32_tile_size
 * (2 or 4, the tile the calls will use for these shapes). */
int frcnn_conv3x3_f32_tile_size(const int *H_host, const int *W_host, int n_levels);
/* 1 when _fwd (and, with need_grads, _bwd_data and _wgrad) would accept these shapes, else 0: the question a caller's dispatch asks before routing a layer here. */
int frcnn_conv3x3_f32_supported(const int *H_host, const int *W_host, int n_levels, int Cin, int Cout, int need_grads);
/* How the stage's products (M = U . V per Winograd plane; the weight gradient's dU = dM . V^T) are taken -- a process-wide switch, 0 by default:
 *   0  v_mfma_f32_32x32x2_f32 on the fp32 operands;
 *   1  the same fp32 operands cut into three bf16 pieces each in registers (exactly) and six v_mfma_f32_32x32x16_bf16 per 16 k rows, fp32 accumulation:
 *      against float64 as close as mode 0 (tests/test_gpu_ops.py: test_conv3x3_f32_split_products_*), 1.4-1.5 x the rate.  Inputs, outputs, workspaces and
 *      every other kernel of the stage are the same.  Environment: FRCNN_CONV_F32_PRODUCTS=split sets the initial value.
 * Returns the previous mode; mode < 0 only asks. */
int frcnn_conv3x3_f32_products(int mode);
int frcnn_conv3x3_f32_fwd(const float *const *x_levels, float *const *y_levels, const int *H_host, const int *W_host, int n_levels, int Cin, int Cout,
                          const float *w, const float *bias, int relu, unsigned short *relu_bits, float *x_transformed, float *u_rotated, void *workspace,
                          size_t workspace_bytes, void *stream);
int frcnn_conv3x3_f32_bwd_data(const float *const *dy_levels, const unsigned short *relu_bits, float *const *dx_levels, const int *H_host, const int *W_host,
                               int n_levels, int Cin, int Cout, const float *w, const float *u_rotated, int pooled, float *dy_transformed,
                               int want_bias_partials, void *workspace, size_t workspace_bytes, void *stream);
int frcnn_conv3x3_f32_wgrad(const float *const *x_levels, const float *const *dy_levels, const unsigned short *relu_bits, const int *H_host, const int *W_host,
                            int n_levels, int Cin, int Cout, float *dw, float *dbias, const float *x_transformed, int pooled, const float *dy_transformed,
                            void *workspace, size_t workspace_bytes, void *stream);

/* O[m][n] = sum_k A[m][k] * B[n][k], fp32, both operands with k contiguous (the conv stage's weight-gradient GEMM on its own; fixed summation order).  The
 * weight gradient of a 1 x 1 convolution is this product on the NCHW planes as they are: dW [Cout, Cin] = dY [Cout, H*W] . X [Cin, H*W]^T -- the bottlenecks'
 * and the FPN laterals' 1 x 1 convolutions behind models/new_model.py:372.  M, N multiples of 64 (<= 4096); splits >= 1 cuts K into equal pieces with
 * one partial product each, O [splits, M, N] (the caller adds them in order; keeps a long K from being one tile's many slabs); K a multiple of 32 * splits.  workspace: the conv
 * stage's dedicated zero-before-first-use block (>= frcnn_gemm_nt_f32_workspace() bytes; any block sized by frcnn_conv3x3_f32_workspace is larger). */
size_t frcnn_gemm_nt_f32_workspace(void);
int frcnn_gemm_nt_f32(const float *A, const float *B, float *O, int M, int N, int K, int splits, void *workspace, size_t workspace_bytes, void *stream);

/* The backbone's FIRST convolution, nn.Conv2d(3, Cout, 3, padding=1) (+ ReLU): `vgg16.features[0]` + `[1]` behind models/model.py:279-281.  Three input
 * channels are no contraction for the matrix cores: a byte mover on the vector units (csrc/conv_c3.hip).  x [3, H, W], y / dy [Cout, H, W], w [Cout, 3, 3, 3],
 * fp32, batch 1.  relu_bits (optional, ceil(Cout / 64) * H * W uint64 words): the signs of the ReLU outputs, bit co % 64 of word [co / 64][pixel]; _wgrad
 * given the same words counts dy where the bit is set (NULL: dy as it is).  dw [Cout, 3, 3, 3] and dbias [Cout] (or NULL) are fully overwritten; Cout a
 * multiple of 4 and W <= 1700 for _wgrad; workspace >= frcnn_conv3x3_c3_wgrad_workspace(H, Cout) bytes of plain scratch.  No input gradient (the
 * input is the image).  Bit-reproducible. */
int frcnn_conv3x3_c3_fwd(const float *x, float *y, int H, int W, int Cout, const float *w, const float *bias, int relu, unsigned long long *relu_bits,
                         void *stream);
size_t frcnn_conv3x3_c3_wgrad_workspace(int H, int Cout);
int frcnn_conv3x3_c3_wgrad(const float *x, const float *dy, int H, int W, int Cout, const unsigned long long *relu_bits, float *dw, float *dbias,
                           void *workspace, size_t workspace_bytes, void *stream);

/* FrozenBatchNorm2d (+ residual add) (+ ReLU) of torchvision's ResNet bottleneck behind models/new_model.py:372, one pass each way (csrc/affine.hip):
 *   _fwd : y = act((x * scale[c] + shift[c]) [+ res]), the torch form's operations in its order (bit-identical); res may be NULL; act = ReLU when relu != 0.
 *   _bwd : gm = g where y > 0 (relu != 0; y = the forward's output) or g;  dx = gm * scale[c];  dres = gm (NULL: not wanted).
 * x, y, res, g, dx, dres: [C, HW] fp32 (NCHW, batch 1); scale, shift: [C] (the frozen statistics folded as torchvision does). */
int frcnn_affine_act_fwd(const float *x, const float *res, float *y, const float *scale, const float *shift, int C, int HW, int relu, void *stream);
int frcnn_affine_act_bwd(const float *g, const float *y, const float *scale, float *dx, float *dres, int C, int HW, int relu, void *stream);
/* The same passes with bf16 on either side of the fp32 arithmetic (BASELINE configs[4]: bf16 autocast of the backbone; the reference has no mixed-precision
 * mode, torch.autocast over models/new_model.py:372 is what this replaces).  dtype codes: 0 = fp32, 1 = bf16.
 *   _fwd_mixed : x (x_dtype) -> y (y_dtype) = act((x * scale + shift) [+ res fp32]), one rounding on the way out; twin_bf16 (or NULL) receives the same values
 *                rounded to bf16 in the same pass (the next convolutions' input).  Forms: bf16 -> bf16 (no res, no twin); bf16 | fp32 -> fp32 (any).
 *   _bwd_mixed : g, y in y's dtype; g2_bf16 = the gradient that arrived at the twin (or NULL), added to g first; dx in x's dtype; dres fp32 or NULL. */
int frcnn_affine_act_fwd_mixed(const void *x, int x_dtype, const float *res, void *y, int y_dtype, void *twin_bf16, const float *scale, const float *shift,
                               int C, int HW, int relu, void *stream);
int frcnn_affine_act_bwd_mixed(const void *g, int g_dtype, const void *g2_bf16, const void *y, const float *scale, void *dx, int dx_dtype, float *dres,
                               int C, int HW, int relu, void *stream);

/* ---- target makers ------------------------------------------------------------------------------ */
/* RPNTargetMaker.forward: variant 0 = VGG (models/model_.py:186-266), 1 = FPN (models/new_model.py:299-349).
 * Sampling (torch.randperm on the host in the reference, model_.py:228,235):
 *   perm_pos / perm_neg != NULL : consume the reference's permutations (parity mode); their lengths must
 *        equal the positive / negative counts (learn them with a first call and out_counts);
 *   else                        : device Philox4x32-10 keyed by (seed, offset): keep the candidates with the
 *        smallest (key, index).  (seed, offset) come BY VALUE, or -- philox_state_dev != NULL -- from device memory:
 *        philox_state_dev[0] = seed, [1] = offset; the call uses that pair and leaves offset + 1 behind (ABI v4).  A training
 *        step captured in a HIP graph therefore draws fresh samples at every replay; by-value arguments would be frozen in it.
 * out_counts (int32[4], device): {n_pos, n_neg before sampling, error flag, reserved}.
 * workspace: DEDICATED to this entry point and ZERO before the first call.  In device-RNG mode column maxima, labels and (N <= 24 576)
 * sampling run as ONE launch whose workgroups meet at an in-kernel barrier; its arrival counter, last-workgroup ticket and the
 * per-GT maxima live in the workspace and are left zero by every call (no memset node; HIP-graph replays need no clearing).          */
int frcnn_rpn_targets(int variant, const float *anchors /*[N,4]*/, int64_t N, const float *gt /*[G,4]*/, int64_t G,
                      const int64_t *perm_pos, int64_t n_perm_pos, const int64_t *perm_neg, int64_t n_perm_neg,
                      uint64_t seed, uint64_t offset, uint64_t *philox_state_dev /*[2] device, or NULL*/,
                      int64_t *out_cls /*[N]*/, float *out_reg /*[N,4]*/, int32_t *out_counts /*[4]*/,
                      void *workspace, size_t workspace_bytes, void *stream);

/* FastRcnnTargetMaker.forward (models/model_.py:127-179; FPN: models/new_model.py:157-206).
 * rois [P_cap,4] with a device count n_rois_dev (NULL = P_cap rows are live); candidates = cat(rois, gt).
 * variant 0: find_jaccard_overlap(roi, gt) eps 1e-5; 1: box_iou(gt, roi).  label_offset 1 (VGG) / 0 (FPN);
 * max_pos 32 / 128; total 128 / 512.  out_counts = {#pos cand, #neg cand, rows written, error bits}.
 * Failure is never silent and never needs a host sync: rows beyond the number actually sampled (the reference throws
 * there, model.py:340 / new_model.py:182) get the out-of-range class -1 and zero boxes, and a NEGATIVE *n_rois_dev (the
 * proposal stage reporting an aborted NMS scan) marks every row that way; frcnn_detection_loss turns an out-of-range
 * class into a NaN loss.  The error bits are also OR-ed into *sticky_status (device int32, may be NULL), which the
 * caller reads whenever it syncs anyway (logging / checkpoint interval). */
#define FRCNN_HT_ERR_PERM_LENGTH 1      /* perm_pos / perm_neg length != candidate count (parity mode) */
#define FRCNN_HT_ERR_PERM_RANGE 2       /* a permutation entry is out of range */
#define FRCNN_HT_ERR_UPSTREAM_ABORT 4   /* *n_rois_dev < 0 */
#define FRCNN_HT_ERR_SHORT 8            /* fewer than `total` rows could be sampled */
int frcnn_head_targets(int variant, const float *rois, const int32_t *n_rois_dev, int64_t P_cap,
                       const float *gt, const int64_t *gt_label, int64_t G,
                       int64_t label_offset, int64_t max_pos, int64_t total,
                       const int64_t *perm_pos, int64_t n_perm_pos, const int64_t *perm_neg, int64_t n_perm_neg,
                       uint64_t seed, uint64_t offset, uint64_t *philox_state_dev /*[2] device (see frcnn_rpn_targets), or NULL*/,
                       int64_t *out_cls /*[total]*/, float *out_reg /*[total,4]*/, float *out_rois /*[total,4]*/,
                       int64_t *out_keep_index /*[total] or NULL*/, int32_t *out_counts /*[4]*/,
                       int32_t *sticky_status /* device int32 or NULL */, void *stream);

/* ---- RoI pooling ---------------------------------------------------------------------------------- */
/* torchvision.ops.RoIPool((PH,PW), spatial_scale) forward/backward (models/model.py:97,113); one image,
 * feat [C,H,W] fp32 NCHW, rois [R,4] = x1,y1,x2,y2 (the reference pre-multiplies by (fw,fh)). */
int frcnn_roi_pool_fwd(const float *feat, int C, int H, int W, const float *rois, int64_t R, int PH, int PW,
                       float spatial_scale, float *out /*[R,C,PH,PW]*/, int32_t *argmax /*[R,C,PH,PW]*/, void *stream);
/* grad_feat [C,H,W] is fully overwritten (no pre-zeroing needed). */
int frcnn_roi_pool_bwd(const float *grad_out, const int32_t *argmax, int64_t R, int C, int H, int W, int PH, int PW,
                       float *grad_feat, void *stream);
/* The same 7x7 pooling with a library-private 16-bit argmax (0xFFFF = empty bin) for planes of fewer than 65535 pixels that
 * fit the LDS-staged kernel (4*H*W*4 bytes <= 48 KB): what the host layer's autograd pair uses (models/model.py:113 never sees
 * the argmax).  Same `out` / `grad_feat` as the int32 entry points; 2 bytes less per pooled element to write and to read back. */
int frcnn_roi_pool_fwd_a16(const float *feat, int C, int H, int W, const float *rois, int64_t R, float spatial_scale,
                           float *out /*[R,C,7,7]*/, uint16_t *argmax16 /*[R,C,7,7]*/, void *stream);
/* `rois` / `spatial_scale`: the boxes the forward pooled (may be NULL).  With them the backward adds without LDS atomics for every RoI of at
 * least 7 x 7 feature cells; without them, and for smaller RoIs, with ds_add_f32.  Either way the result is bit-reproducible run to run. */
int frcnn_roi_pool_bwd_a16(const float *grad_out, const uint16_t *argmax16, const float *rois /*[R,4] or NULL*/, float spatial_scale,
                           int64_t R, int C, int H, int W, float *grad_feat, void *stream);

/* torchvision.ops.MultiScaleRoIAlign(names, PH, sampling_ratio) (models/new_model.py:127,143): level mapper
 * k = floor(k0 + log2(sqrt(area)/s0) + 1e-6) clamped to [k_min, k_min+n_levels-1]; per level roi_align
 * (aligned = False).  rois in image pixels.  The per-level scales are explicit.            */
int frcnn_roi_level_map(const float *rois, int64_t R, int k_min, int k_max, float s0, int k0, float eps,
                        int32_t *out_level, void *stream);
int frcnn_ms_roi_align_fwd(const float *const *feats_host /*[n_levels] device ptrs*/, const int *H_host, const int *W_host,
                           const float *scales_host, int n_levels, int C, const float *rois, int64_t R,
                           int PH, int PW, int sampling_ratio, int aligned, int k_min, float s0, int k0,
                           float *out /*[R,C,PH,PW]*/, int32_t *out_level /*[R] or NULL*/,
                           const int32_t *order /*[R] or NULL: dispatch order of the RoIs (frcnn_roi_scale_order); results do not depend on it*/,
                           void *stream);
/* The `roi * (w, h, w, h)` of FastRCNNHead.forward (models/new_model.py:136-140) and, in the same launch, the permutation that
 * dispatches the 7x7 / sampling-ratio-2 forward's workgroups largest footprint first: out_rois[i] = rois[i] * mul4 (fp32 products, as
 * torch computes them), out_order = RoI indices by decreasing (staging passes, footprint pixels) of frcnn_ms_roi_align_fwd, ties by
 * index; out_cost (optional, [R]) = the keys.  R <= 4096.  Level table as in frcnn_ms_roi_align_fwd (no feature pointers needed).  */
int frcnn_roi_scale_order(const float *rois, int64_t R, const float *mul4_host, const int *H_host, const int *W_host, const float *scales_host,
                          int n_levels, int aligned, int k_min, float s0, int k0, float *out_rois /*[R,4] or NULL*/, int32_t *out_order /*[R]*/,
                          uint32_t *out_cost /*[R] or NULL*/, void *stream);
/* grad_feats[l] [C,H_l,W_l] are OVERWRITTEN with the gradient of every level (zero where no RoI reaches); the caller does
 * not clear them.  7x7 / sampling_ratio 2: tile-owner gather (per-tile RoI lists, long lists summed by segments in a fixed
 * order), no atomics, bit-reproducible, two launches; workspace >= frcnn_ms_roi_align_bwd_workspace(...) (lists, weight-table
 * records of the (RoI, tile) pairs, partial tiles, tickets; its contents on entry do not matter).  One call at a time per
 * device (the last-workgroup ticket of the lists launch is a word of the library picked by the workspace address).  Other
 * shapes: memset + fp32 atomics inside the library (sum order not fixed, tolerance 1e-4; no workspace needed).             */
size_t frcnn_ms_roi_align_bwd_workspace(const int *H_host, const int *W_host, int n_levels, int C, int64_t R);
int frcnn_ms_roi_align_bwd(const float *grad_out, float *const *grad_feats_host, const int *H_host, const int *W_host,
                           const float *scales_host, int n_levels, int C, const float *rois, int64_t R,
                           int PH, int PW, int sampling_ratio, int aligned, int k_min, float s0, int k0,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---- detection losses (losses/loss.py:5-85) ------------------------------------------------- */
/* FRCNNLoss forward AND the un-normalised input gradients in one pass.  out7 (device): total, rpn_cls, rpn_reg,
 * head_cls, head_reg losses, then 1/#(rpn label >= 0) and 1/R (the scales backward multiplies the gradients by).
 * g_* have the shapes of the predictions.  workspace >= 32 KiB, DEDICATED to this entry point and ZERO before the first call:
 * its first word is the ticket by which the last workgroup to finish adds up the partial sums (one launch, no finalize kernel);
 * that workgroup leaves the ticket zero, so consecutive calls (and HIP-graph replays) need no clearing in between.            */
int frcnn_detection_loss(const float *rpn_cls /*[N,2]*/, const float *rpn_reg /*[N,4]*/, const int64_t *t_rpn_cls /*[N]*/,
                         const float *t_rpn_reg /*[N,4]*/, int64_t N,
                         const float *head_cls /*[R,NC]*/, const float *head_reg /*[R,4]*/, const int64_t *t_cls /*[R]*/,
                         const float *t_reg /*[R,4]*/, int64_t R, int NC,
                         float *out7, float *g_rpn_cls, float *g_rpn_reg, float *g_head_cls, float *g_head_reg,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- input stage in front of the path ------------------------------------------------------
 * One uint8 HWC RGB frame in HBM -> [hflip] -> PIL-bilinear resize to (oh, ow) -> /255 -> (x - mean) / std -> float CHW
 * [3, pad_h, pad_w], zero outside (oh, ow).  Replaces T.RandomHorizontalFlip / T.Resize(800, max 1333) / T.ToTensor /
 * T.Normalize (new_datasets/transforms.py:57-132,238-281; new_datasets/build.py:20-33, datasets/build.py:10-24) and the
 * pad-to-32 collate (new_datasets/coco_dataset.py:49-66).  Bit-identical to Pillow's 8-bit resampler.  mean/std: host,
 * 3 floats.  out_u8 (optional, [oh, ow, 3]) receives the resized uint8 frame; either output may be NULL, not both.  */
int frcnn_preprocess_image(const uint8_t *src_hwc, int h, int w, int flip, int oh, int ow, int pad_h, int pad_w,
                           const float *mean_host, const float *std_host, float *out_chw, uint8_t *out_u8,
                           void *workspace, size_t workspace_bytes, void *stream);
/* Boxes xyxy in source pixels -> hflip (transforms.py:64-68) -> x resize ratios (:113-117) -> / resized (w, h) (:276-280),
 * i.e. the normalised boxes the model takes.  binary32 throughout, like the reference's tensors.                   */
int frcnn_preprocess_boxes(const float *boxes, int64_t n, int w, int h, int flip, int ow, int oh, float *out, void *stream);

/* ---- in-library kernel timing (HIP events on the launch stream) -------------------------------------- */
/* When enabled, every kernel launch made by this library is bracketed by two hipEventRecord on the
 * caller's stream.  frcnn_prof_collect() synchronises those events (call it after the stream is idle)
 * and folds them into per-kernel totals readable with frcnn_prof_get().                                 */
int frcnn_prof_enable(int on);
int frcnn_prof_collect(void);
int frcnn_prof_reset(void);
int frcnn_prof_num_kernels(void);
const char *frcnn_prof_kernel_name(int kernel_id);
int frcnn_prof_get(int kernel_id, double *total_ms, int64_t *launches);
/* per-launch durations (ms, launch order) of one kernel since the last reset: copies min(cap, n), returns n. */
int64_t frcnn_prof_get_samples(int kernel_id, float *out_ms, int64_t cap);

/* Diagnostics: n_workgroups workgroups of 256 threads that do nothing but stay resident for `microseconds` (bounded: <= 100 000) on
 * `stream` -- a stand-in for another tenant's persistent kernels (RCCL's channels during the gradient all-reduce) when testing that the
 * launches with in-kernel hand-offs (nms_kernel, rpn_match_kernel, topk_partition_kernel) still complete and stay exact next to them
 * (tests/test_gpu_ops.py: test_in_kernel_handoffs_next_to_resident_foreign_workgroups).  Not used by the product path.            */
int frcnn_diag_occupy(int n_workgroups, int microseconds, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
