// affine.hip -- the elementwise tail of a ResNet bottleneck: FrozenBatchNorm2d (a per-channel x * scale + shift with frozen statistics), the residual
// add and the ReLU of torchvision's Bottleneck.forward behind `resnet_fpn_backbone` (models/new_model.py:372), which the torch form runs as three or
// four full-tensor passes forward (mul, add, [add,] clamp) and two or three backward (threshold, mul[, copy]) -- 2.3 ms of elementwise launches per
// 800 x 1344 training step.  One pass each way:
//   affine_act_fwd_kernel   y = act((x * scale[c] + shift[c]) [+ res]),  act = ReLU or identity; the SAME operations in the SAME order as the torch
//                           form (separate multiply and add, no contraction): bit-identical results
//   affine_act_bwd_kernel   gm = g where y > 0 (ReLU) or g;  dx = gm * scale[c];  dres = gm (when the forward had a residual and a ReLU)
// x, y, res, g: [C][HW] fp32 planes (NCHW, batch 1); scale, shift: [C].  Pure HBM movers: thread = four elements 256 apart, grid (pieces, C).
//
// Mixed precision (round 5; BASELINE configs[4], bf16 autocast): the same two passes with bf16 on either side of the fp32 arithmetic
// (frcnn_affine_act_fwd_mixed / _bwd_mixed).  Under autocast the torch form of a bottleneck's tails was ~20 elementwise / dtype-copy launches per
// block and direction (x.bfloat16() in front of every convolution, addcmul, relu, the promoting x * scale + shift of the norm in front of the
// residual sum, the sum, its ReLU; threshold, mul, copies and the gradient accumulation backward): 3.2 ms of elementwise kernels + 1.2 ms of
// dtype copies in an 11 ms step.  Here an INNER norm reads the convolution's bf16 output and writes bf16 (ONE rounding, as torch.addcmul + relu);
// the norm in front of the residual reads bf16 and the fp32 residual and writes the fp32 stream AND, in the same pass, its bf16 twin -- the next
// block's convolutions read that instead of casting the stream again; backward takes the stream's fp32 gradient and the twin's bf16 gradient
// together (what autograd would accumulate in a pass of its own).
#include <hip/hip_bf16.h>
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


// ---- mixed precision: XT / YT = float or __hip_bfloat16; arithmetic in fp32, one rounding on the way out --------------------------------------
template <typename T> __device__ __forceinline__ float af_ld(const T *p, size_t i);
template <> __device__ __forceinline__ float af_ld<float>(const float *p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float af_ld<__hip_bfloat16>(const __hip_bfloat16 *p, size_t i) { return __bfloat162float(p[i]); }
template <typename T> __device__ __forceinline__ void af_st(T *p, size_t i, float v);
template <> __device__ __forceinline__ void af_st<float>(float *p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void af_st<__hip_bfloat16>(__hip_bfloat16 *p, size_t i, float v) { p[i] = __float2bfloat16(v); }   // round to nearest even; NaN stays NaN

template <typename XT, typename YT, bool RES, bool TWIN>
__global__ __launch_bounds__(256) void affine_act_fwd_mixed_kernel(const XT *__restrict__ x, const float *__restrict__ res, YT *__restrict__ y,
                                                                   __hip_bfloat16 *__restrict__ twin, const float *__restrict__ scale,
                                                                   const float *__restrict__ shift, int HW, int relu)
{
    const int c = blockIdx.y;
    const size_t base = (size_t)c * HW;
    const float s = scale[c], b = shift[c];
    float v[4], r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        v[j] = i < HW ? af_ld<XT>(x, base + i) : 0.0f;
        r[j] = (RES && i < HW) ? res[base + i] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        float o = v[j] * s + b;
        if (RES) o = o + r[j];
        if (relu) o = o < 0.0f ? 0.0f : o;
        if (i < HW) {
            af_st<YT>(y, base + i, o);
            if (TWIN) twin[base + i] = __float2bfloat16(o);
        }
    }
}

// g: gradient of y (GT = y's type); g2: gradient of the bf16 twin (or NULL); y: the forward's output, for the ReLU's mask; dx in x's type
template <typename GT, typename DXT, bool RELU, bool DRES, bool G2>
__global__ __launch_bounds__(256) void affine_act_bwd_mixed_kernel(const GT *__restrict__ g, const __hip_bfloat16 *__restrict__ g2, const GT *__restrict__ y,
                                                                   const float *__restrict__ scale, DXT *__restrict__ dx, float *__restrict__ dres, int HW)
{
    const int c = blockIdx.y;
    const size_t base = (size_t)c * HW;
    const float s = scale[c];
    float v[4], m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        v[j] = i < HW ? af_ld<GT>(g, base + i) : 0.0f;
        if (G2) v[j] += i < HW ? __bfloat162float(g2[base + i]) : 0.0f;
        m[j] = (RELU && i < HW) ? af_ld<GT>(y, base + i) : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * 1024 + j * 256 + threadIdx.x;
        const float gm = m[j] > 0.0f ? v[j] : 0.0f;
        if (i < HW) {
            af_st<DXT>(dx, base + i, gm * s);
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

// dtype codes: 0 = fp32, 1 = bf16.  Forward forms: x bf16 -> y bf16 (an inner norm: no residual, no twin); x bf16 or fp32 -> y fp32 (+ fp32 residual)
// (+ bf16 twin of y).  Anything else: FRCNN_ERR_UNSUPPORTED.
FRCNN_EXPORT int frcnn_affine_act_fwd_mixed(const void *x_dev, int x_dtype, const float *res_dev, void *y_dev, int y_dtype, void *twin_bf16_dev,
                                            const float *scale_dev, const float *shift_dev, int C, int HW, int relu, void *stream)
{
    int rc = af_check(x_dev, y_dev, scale_dev, C, HW, "affine_act_fwd_mixed");
    if (rc) return rc;
    FRCNN_REQUIRE(shift_dev, "affine_act_fwd_mixed: NULL shift");
    const dim3 grid((unsigned)((HW + 1023) / 1024), (unsigned)C);
    hipStream_t s = (hipStream_t)stream;
    typedef __hip_bfloat16 bf;
#define AF_FWD(XT, YT, RES, TWIN)                                                                                                       \
    FRCNN_LAUNCH((affine_act_fwd_mixed_kernel<XT, YT, RES, TWIN>), grid, dim3(256), 0, s, (const XT *)x_dev, res_dev, (YT *)y_dev, \
                 (bf *)twin_bf16_dev, scale_dev, shift_dev, HW, relu)
    const bool res = res_dev != nullptr, twin = twin_bf16_dev != nullptr;
    if (x_dtype == 1 && y_dtype == 1 && !res && !twin) AF_FWD(bf, bf, false, false);
    else if (x_dtype == 1 && y_dtype == 0) {
        if (res && twin) AF_FWD(bf, float, true, true); else if (res) AF_FWD(bf, float, true, false);
        else if (twin) AF_FWD(bf, float, false, true); else AF_FWD(bf, float, false, false);
    } else if (x_dtype == 0 && y_dtype == 0) {
        if (res && twin) AF_FWD(float, float, true, true); else if (res) AF_FWD(float, float, true, false);
        else if (twin) AF_FWD(float, float, false, true); else AF_FWD(float, float, false, false);
    } else
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "affine_act_fwd_mixed: x dtype %d -> y dtype %d (residual %d, twin %d) is not a form the bottleneck uses", x_dtype,
                               y_dtype, (int)res, (int)twin);
#undef AF_FWD
    FRCNN_CHECK_LAUNCH("affine_act_fwd_mixed_kernel");
    return FRCNN_OK;
}

// g / y in y's dtype (g_dtype), g2 = the twin's bf16 gradient or NULL, dx in x's dtype (dx_dtype), dres fp32 or NULL
FRCNN_EXPORT int frcnn_affine_act_bwd_mixed(const void *g_dev, int g_dtype, const void *g2_bf16_dev, const void *y_dev, const float *scale_dev, void *dx_dev,
                                            int dx_dtype, float *dres_dev, int C, int HW, int relu, void *stream)
{
    int rc = af_check(g_dev, dx_dev, scale_dev, C, HW, "affine_act_bwd_mixed");
    if (rc) return rc;
    FRCNN_REQUIRE(!relu || y_dev, "affine_act_bwd_mixed: the ReLU's backward needs the forward's output");
    const dim3 grid((unsigned)((HW + 1023) / 1024), (unsigned)C);
    hipStream_t s = (hipStream_t)stream;
    typedef __hip_bfloat16 bf;
#define AF_BWD(GT, DXT, RELU, DRES, G2)                                                                                              \
    FRCNN_LAUNCH((affine_act_bwd_mixed_kernel<GT, DXT, RELU, DRES, G2>), grid, dim3(256), 0, s, (const GT *)g_dev, (const bf *)g2_bf16_dev, \
                 (const GT *)y_dev, scale_dev, (DXT *)dx_dev, dres_dev, HW)
#define AF_BWD_FLAGS(GT, DXT)                                                                     \
    do {                                                                                          \
        const bool dr = dres_dev != nullptr, g2 = g2_bf16_dev != nullptr;                         \
        if (relu) {                                                                               \
            if (dr && g2) AF_BWD(GT, DXT, true, true, true); else if (dr) AF_BWD(GT, DXT, true, true, false); \
            else if (g2) AF_BWD(GT, DXT, true, false, true); else AF_BWD(GT, DXT, true, false, false);        \
        } else {                                                                                  \
            if (dr && g2) AF_BWD(GT, DXT, false, true, true); else if (dr) AF_BWD(GT, DXT, false, true, false); \
            else if (g2) AF_BWD(GT, DXT, false, false, true); else AF_BWD(GT, DXT, false, false, false);        \
        }                                                                                         \
    } while (0)
    if (g_dtype == 1 && dx_dtype == 1) {
        FRCNN_REQUIRE(!g2_bf16_dev && !dres_dev, "affine_act_bwd_mixed: the bf16 -> bf16 form has neither twin nor residual");
        if (relu) AF_BWD(bf, bf, true, false, false); else AF_BWD(bf, bf, false, false, false);
    } else if (g_dtype == 0 && dx_dtype == 1) AF_BWD_FLAGS(float, bf);
    else if (g_dtype == 0 && dx_dtype == 0) AF_BWD_FLAGS(float, float);
    else return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "affine_act_bwd_mixed: g dtype %d -> dx dtype %d is not a form the bottleneck uses", g_dtype, dx_dtype);
#undef AF_BWD_FLAGS
#undef AF_BWD
    FRCNN_CHECK_LAUNCH("affine_act_bwd_mixed_kernel");
    return FRCNN_OK;
}
