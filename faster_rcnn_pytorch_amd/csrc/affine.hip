// affine.hip -- the elementwise tail of a ResNet bottleneck: FrozenBatchNorm2d (a per-channel x * scale + shift with frozen statistics), the residual
// add and the ReLU of torchvision's Bottleneck.forward behind `resnet_fpn_backbone` (models/new_model.py:372), which the torch form runs as three or
// four full-tensor passes forward (mul, add, [add,] clamp) and two or three backward (threshold, mul[, copy]) -- 2.3 ms of elementwise launches per
// 800 x 1344 training step.  One pass each way:
//   affine_act_fwd_kernel   y = act((x * scale[c] + shift[c]) [+ res]),  act = ReLU or identity; the SAME operations in the SAME order as the torch
//                           form (separate multiply and add, no contraction): bit-identical results
//   affine_act_bwd_kernel   gm = g where y > 0 (ReLU) or g;  dx = gm * scale[c];  dres = gm (when the forward had a residual and a ReLU)
// x, y, res, g: [C][HW] fp32 planes (NCHW, batch 1); scale, shift: [C].  Pure HBM movers: thread = four elements 256 apart, grid (pieces, C).
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(affine);

// RES / RELU / DRES are template flags: a run-time `pointer != NULL` test in front of every load of an unrolled batch keeps the loads from being
// issued together (csrc/conv_c3.hip: 170 -> 118 us for the same reason)
template <bool RES>
__global__ __launch_bounds__(256) void affine_act_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res, float *__restrict__ y,
                                                             const float *__restrict__ scale, const float *__restrict__ shift, int HW, int relu)
{
    const int c = blockIdx.y;
    const size_t base = (size_t)c * HW;
    const float s = scale[c], b = shift[c];
    float v[4], r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        v[j] = i < HW ? x[base + i] : 0.0f;
        r[j] = (RES && i < HW) ? res[base + i] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        float o = v[j] * s + b;                                           // compiled without contraction: torch's mul, then add
        if (RES) o = o + r[j];
        if (relu) o = o < 0.0f ? 0.0f : o;                    // NaN stays NaN, as torch.relu (fmaxf would return 0)
        if (i < HW) y[base + i] = o;
    }
}

template <bool RELU, bool DRES>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y, const float *__restrict__ scale,
                                                             float *__restrict__ dx, float *__restrict__ dres, int HW, int relu)
{
    const int c = blockIdx.y;
    const size_t base = (size_t)c * HW;
    const float s = scale[c];
    float v[4], m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        v[j] = i < HW ? g[base + i] : 0.0f;
        m[j] = (RELU && i < HW) ? y[base + i] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        const float gm = m[j] > 0.0f ? v[j] : 0.0f;
        if (i < HW) {
            dx[base + i] = gm * s;
            if (DRES) dres[base + i] = gm;
        }
    }
}

static int af_check(const void *a, const void *b, const void *c, int C, int HW, const char *what)
{
    FRCNN_REQUIRE(a && b && c, "%s: NULL pointer", what);
    FRCNN_REQUIRE(C > 0 && C <= 65535 && HW > 0 && (long long)C * HW < (1ll << 31), "%s: bad size %d x %d", what, C, HW);
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_affine_act_fwd(const float *x_dev, const float *res_dev, float *y_dev, const float *scale_dev, const float *shift_dev, int C, int HW,
                                      int relu, void *stream)
{
    int rc = af_check(x_dev, y_dev, scale_dev, C, HW, "affine_act_fwd");
    if (rc) return rc;
    FRCNN_REQUIRE(shift_dev, "affine_act_fwd: NULL shift");
    const dim3 grid((unsigned)((HW + 1023) / 1024), (unsigned)C);
    if (res_dev) FRCNN_LAUNCH(affine_act_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x_dev, res_dev, y_dev, scale_dev, shift_dev, HW, relu);
    else FRCNN_LAUNCH(affine_act_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x_dev, res_dev, y_dev, scale_dev, shift_dev, HW, relu);
    FRCNN_CHECK_LAUNCH("affine_act_fwd_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_affine_act_bwd(const float *g_dev, const float *y_dev, const float *scale_dev, float *dx_dev, float *dres_dev, int C, int HW, int relu,
                                      void *stream)
{
    int rc = af_check(g_dev, dx_dev, scale_dev, C, HW, "affine_act_bwd");
    if (rc) return rc;
    FRCNN_REQUIRE(!relu || y_dev, "affine_act_bwd: the ReLU's backward needs the forward's output");
    const dim3 grid((unsigned)((HW + 1023) / 1024), (unsigned)C);
    hipStream_t s = (hipStream_t)stream;
    if (relu && dres_dev) FRCNN_LAUNCH((affine_act_bwd_kernel<true, true>), grid, dim3(256), 0, s, g_dev, y_dev, scale_dev, dx_dev, dres_dev, HW, relu);
    else if (relu) FRCNN_LAUNCH((affine_act_bwd_kernel<true, false>), grid, dim3(256), 0, s, g_dev, y_dev, scale_dev, dx_dev, dres_dev, HW, relu);
    else if (dres_dev) FRCNN_LAUNCH((affine_act_bwd_kernel<false, true>), grid, dim3(256), 0, s, g_dev, y_dev, scale_dev, dx_dev, dres_dev, HW, relu);
    else FRCNN_LAUNCH((affine_act_bwd_kernel<false, false>), grid, dim3(256), 0, s, g_dev, y_dev, scale_dev, dx_dev, dres_dev, HW, relu);
    FRCNN_CHECK_LAUNCH("affine_act_bwd_kernel");
    return FRCNN_OK;
}
