// boxes.hip -- anchor grid, box codec, pairwise IoU, proposal prologue (gfx950).
//
// All four kernels are pure streaming kernels: one anchor / box per lane, 16-byte
// (float4) coalesced loads and stores, nothing staged in LDS because nothing is reused.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "topk_dev.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(boxes);

// ------------------------------------------------------------------------------------------
// anchor of flat index i from the level table (anchor.py:34-55; models/new_model.py:46-47)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 anchor_at(const AnchorDesc &d, int64_t i)
{
    int l = 0;
#pragma unroll
    for (int k = 1; k < FRCNN_MAX_LEVELS; ++k)
        if (k < d.n_levels && i >= d.off[k]) l = k;
    const int local = (int)(i - d.off[l]);
    const int pos = local / d.A;
    const int a = local - pos * d.A;
    const int y = pos / d.fw[l];
    const int x = pos - y * d.fw[l];
    const float sx = (float)(x * d.sw[l]);
    const float sy = (float)(y * d.sh[l]);
    const float *b = d.base[l][a];
    // fp32 add then IEEE fp32 divide == numpy's float64 add/divide + float32 store
    return make_float4((b[0] + sx) / d.div_w, (b[1] + sy) / d.div_h, (b[2] + sx) / d.div_w, (b[3] + sy) / d.div_h);
}

__global__ __launch_bounds__(256) void anchor_grid_kernel(AnchorDesc d, float4 *__restrict__ out, int64_t N)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) out[i] = anchor_at(d, i);
}

int frcnn_fill_anchor_desc(AnchorDesc *d, int n_levels, const int *fh, const int *fw, const int *sh, const int *sw,
                           const float *base, int A, float div_w, float div_h, int64_t *n_total)
{
    FRCNN_REQUIRE(fh && fw && sh && sw && base, "anchor grid: NULL level table");
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS, "anchor grid: n_levels %d not in [1,%d]", n_levels, FRCNN_MAX_LEVELS);
    FRCNN_REQUIRE(A >= 1 && A <= FRCNN_MAX_BASE, "anchor grid: A %d not in [1,%d]", A, FRCNN_MAX_BASE);
    FRCNN_REQUIRE(div_w > 0.f && div_h > 0.f, "anchor grid: divisor must be positive");
    d->n_levels = n_levels; d->A = A; d->div_w = div_w; d->div_h = div_h;
    int64_t off = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        if (l < n_levels) {
            FRCNN_REQUIRE(fh[l] > 0 && fw[l] > 0 && sh[l] >= 0 && sw[l] >= 0, "anchor grid: bad level %d", l);
            d->fh[l] = fh[l]; d->fw[l] = fw[l]; d->sh[l] = sh[l]; d->sw[l] = sw[l];
            d->off[l] = off;
            off += (int64_t)fh[l] * fw[l] * A;
            for (int a = 0; a < A; ++a)
                for (int c = 0; c < 4; ++c) d->base[l][a][c] = base[((size_t)l * A + a) * 4 + c];
        } else {
            d->fh[l] = d->fw[l] = 1; d->sh[l] = d->sw[l] = 0; d->off[l] = off;
        }
    }
    FRCNN_REQUIRE(off < ((int64_t)1 << 31), "anchor grid: %lld anchors exceed int32 indexing", (long long)off);
    *n_total = off;
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_anchor_grid(int n_levels, const int *fh, const int *fw, const int *sh, const int *sw,
                                   const float *base, int A, float div_w, float div_h, float *out, int64_t N, void *stream)
{
    AnchorDesc d;
    int64_t n = 0;
    int rc = frcnn_fill_anchor_desc(&d, n_levels, fh, fw, sh, sw, base, A, div_w, div_h, &n);
    if (rc) return rc;
    FRCNN_REQUIRE(out, "anchor grid: NULL output");
    FRCNN_REQUIRE(n == N, "anchor grid: level table describes %lld anchors, caller passed N=%lld", (long long)n, (long long)N);
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(anchor_grid_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d, (float4 *)out, N);
    FRCNN_CHECK_LAUNCH("anchor_grid_kernel");
    return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------
// box codec (utils/util.py:15-50)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_codec_kernel(int op, const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                        int64_t n, float4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 va = a[i];
    float4 r;
    switch (op) {
    case 0: r = xy_to_cxcy4(va); break;
    case 1: r = cxcy_to_xy4(va); break;
    case 2: r = decode4(va, b[i]); break;
    default: r = encode4(va, b[i]); break;
    }
    out[i] = r;
}

FRCNN_EXPORT int frcnn_box_codec(int op, const float *a, const float *b, int64_t n, float *out, void *stream)
{
    FRCNN_REQUIRE(op >= 0 && op <= 3, "box_codec: unknown op %d", op);
    FRCNN_REQUIRE(n >= 0 && (n == 0 || (a && out)), "box_codec: NULL pointer");
    FRCNN_REQUIRE(op < 2 || n == 0 || b, "box_codec: op %d needs a second operand", op);
    if (n == 0) return FRCNN_OK;
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(box_codec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op, (const float4 *)a,
                 (const float4 *)b, n, (float4 *)out);
    FRCNN_CHECK_LAUNCH("box_codec_kernel");
    return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------
// pairwise IoU [n1,n2] (utils/util.py:66-102, util/box_ops.py:24-37).  One lane per output
// element, n2-minor so the row store is coalesced; set2 is tiny (G boxes) and stays in L1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairwise_iou_kernel(const float4 *__restrict__ s1, int64_t n1,
                                                           const float4 *__restrict__ s2, int64_t n2, float eps, int add_eps,
                                                           float *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n1 * n2) return;
    const int64_t i = e / n2, j = e - i * n2;
    const float4 p = s1[i], q = s2[j];
    out[e] = add_eps ? iou_pair<true>(p, q, eps) : iou_pair<false>(p, q, 0.f);
}

FRCNN_EXPORT int frcnn_pairwise_iou(const float *set1, int64_t n1, const float *set2, int64_t n2, float eps, float *out, void *stream)
{
    FRCNN_REQUIRE(n1 >= 0 && n2 >= 0, "pairwise_iou: negative size");
    if (n1 == 0 || n2 == 0) return FRCNN_OK;
    FRCNN_REQUIRE(set1 && set2 && out, "pairwise_iou: NULL pointer");
    FRCNN_REQUIRE(n1 * n2 < ((int64_t)1 << 40), "pairwise_iou: output too large");
    hipStream_t s = (hipStream_t)stream;
    const int64_t tot = n1 * n2;
    FRCNN_LAUNCH(pairwise_iou_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (const float4 *)set1, n1,
                 (const float4 *)set2, n2, eps, eps != 0.f ? 1 : 0, out);
    FRCNN_CHECK_LAUNCH("pairwise_iou_kernel");
    return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------
// proposal prologue: RegionProposal.forward up to the sort (models/model.py:20-41)
//   score = softmax(cls)[1]; box = clamp(cxcy_to_xy(decode(reg, xy_to_cxcy(anchor))), 0, 1)
//   keep  = (h >= m) & (w >= m); filtered (and NaN) scores are written as -1.
// HAS_ANCHORS = false regenerates the anchor from the level table in registers.
// Algorithmic bytes per anchor: 16 (reg) + 8 (cls) [+16 anchors] in, 16 (box) + 4 (score) out.
// The first lanes also clear the pipeline's control words (topk count, nms count) and the NMS stage's pull counters.
// ------------------------------------------------------------------------------------------
// SS = 512 / SS_LARGE: the first SS / 64 workgroups of the grid draw the top-k stage's splitters instead (topk_dev.h): the score of a
// sampled anchor comes from the same per-anchor function, so it is bit-identical to the one the prologue workgroups store.
template <bool HAS_ANCHORS>
__device__ __forceinline__ float prologue_one(const float4 *__restrict__ reg, const float2 *__restrict__ cls, const float4 *__restrict__ anchors,
                                              const AnchorDesc &d, int64_t i, float min_size, float4 *box)
{
    const float4 t = reg[i];
    const float2 c = cls[i];
    const float4 an = HAS_ANCHORS ? anchors[i] : anchor_at(d, i);
    float4 b = cxcy_to_xy4(decode4(t, xy_to_cxcy4(an)));
    b.x = clamp01(b.x); b.y = clamp01(b.y); b.z = clamp01(b.z); b.w = clamp01(b.w);
    const float ws = b.z - b.x, hs = b.w - b.y;
    const bool keep = (hs >= min_size) && (ws >= min_size);
    const float m = tmax(c.x, c.y);
    const float e0 = det_expf(c.x - m), e1 = det_expf(c.y - m);
    float sc = e1 / (e0 + e1);
    if (!keep || !(sc >= 0.0f)) sc = -1.0f;
    *box = b;
    return sc;
}
template <bool HAS_ANCHORS, int SS>
__global__ __launch_bounds__(256) void proposal_prologue_kernel(const float4 *__restrict__ reg, const float2 *__restrict__ cls,
                                                                const float4 *__restrict__ anchors, AnchorDesc d, int64_t N,
                                                                float min_size, float4 *__restrict__ out_boxes,
                                                                float *__restrict__ out_scores, int32_t *__restrict__ ctrl_zero,
                                                                int n_ctrl, int32_t *__restrict__ zero2, int n_zero2,
                                                                SsCtl *__restrict__ ss_ctl, int ss_stride)
{
    // the sampling workgroups come FIRST in the grid: they are the longest (S decodes + S x 64 compares each) and would otherwise start
    // after every prologue workgroup has been dispatched -- the launch's tail (21 us at FPN size with them last)
    if constexpr (SS > 0) {
        if ((int)blockIdx.x < SS / 64) {
            __shared__ uint4 s_k4[SS / 4];
            __shared__ int s_part[4][64];
            ss_sample_body<SS>([&](int i) { float4 b; return prologue_one<HAS_ANCHORS>(reg, cls, anchors, d, i, min_size, &b); }, (int)N, ss_stride, ss_ctl,
                               (int)blockIdx.x, s_k4, s_part);
            return;
        }
    }
    const int n_wg = (int)gridDim.x - SS / 64;
    const int64_t i = (int64_t)((int)blockIdx.x - SS / 64) * 256 + threadIdx.x;
    if (ctrl_zero && i < n_ctrl) ctrl_zero[i] = 0;
    if (zero2)                                                  // second region: the NMS stage's per-box pull counters + flags
        for (int64_t j = i; j < n_zero2; j += (int64_t)n_wg * 256) zero2[j] = 0;
    if (i >= N) return;
    float4 b;
    const float sc = prologue_one<HAS_ANCHORS>(reg, cls, anchors, d, i, min_size, &b);
    out_boxes[i] = b;
    out_scores[i] = sc;
}

int frcnn_launch_prologue(const float *reg, const float *cls, const float *anchors, const AnchorDesc *d, int64_t N,
                          float min_size, float *out_boxes, float *out_scores, int32_t *ctrl_zero, int n_ctrl, int32_t *zero2, int n_zero2,
                          void *sample_ctl, int64_t pre_k, hipStream_t s)
{
    int S = 0, stride = 0;
    if (sample_ctl) ss_plan(N, pre_k, &S, &stride);
    const dim3 grid((unsigned)((N + 255) / 256) + (unsigned)(S / 64)), block(256);
    AnchorDesc dummy = {};
    const AnchorDesc &dd = anchors ? dummy : *d;
    const float4 *an = (const float4 *)anchors;
#define PRO_ARGS (const float4 *)reg, (const float2 *)cls, an, dd, N, min_size, (float4 *)out_boxes, out_scores, ctrl_zero, n_ctrl, zero2, n_zero2, (SsCtl *)sample_ctl, stride
    if (anchors) {
        if (S == 0) FRCNN_LAUNCH((proposal_prologue_kernel<true, 0>), grid, block, 0, s, PRO_ARGS);
        else if (S == 512) FRCNN_LAUNCH((proposal_prologue_kernel<true, 512>), grid, block, 0, s, PRO_ARGS);
        else FRCNN_LAUNCH((proposal_prologue_kernel<true, SS_LARGE>), grid, block, 0, s, PRO_ARGS);
    } else {
        if (S == 0) FRCNN_LAUNCH((proposal_prologue_kernel<false, 0>), grid, block, 0, s, PRO_ARGS);
        else if (S == 512) FRCNN_LAUNCH((proposal_prologue_kernel<false, 512>), grid, block, 0, s, PRO_ARGS);
        else FRCNN_LAUNCH((proposal_prologue_kernel<false, SS_LARGE>), grid, block, 0, s, PRO_ARGS);
    }
#undef PRO_ARGS
    FRCNN_CHECK_LAUNCH("proposal_prologue_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_proposal_prologue(const float *reg, const float *cls, const float *anchors, int64_t N, float min_size_norm,
                                         float *out_boxes, float *out_scores, void *stream)
{
    FRCNN_REQUIRE(N >= 0, "prologue: negative N");
    if (N == 0) return FRCNN_OK;
    FRCNN_REQUIRE(reg && cls && anchors && out_boxes && out_scores, "prologue: NULL pointer");
    FRCNN_REQUIRE(N < ((int64_t)1 << 31), "prologue: N too large");
    return frcnn_launch_prologue(reg, cls, anchors, nullptr, N, min_size_norm, out_boxes, out_scores, nullptr, 0, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// diagnostics: resident do-nothing workgroups (include/frcnn_hip.h: frcnn_diag_occupy)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_occupy_kernel(unsigned long long ticks)            // ticks of the 100 MHz s_memrealtime counter
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);          // bounded: leaves after `ticks` whatever happens
}

FRCNN_EXPORT int frcnn_diag_occupy(int n_workgroups, int microseconds, void *stream)
{
    FRCNN_REQUIRE(n_workgroups > 0 && n_workgroups <= 4096 && microseconds > 0 && microseconds <= 100000, "diag_occupy: 1 .. 4096 workgroups, 1 .. 100000 us");
    FRCNN_LAUNCH(diag_occupy_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
    FRCNN_CHECK_LAUNCH("diag_occupy_kernel");
    return FRCNN_OK;
}
