// preprocess.hip -- the input stage in front of the path on gfx950:
//   hflip -> resize -> ToTensor -> Normalize  (new_datasets/transforms.py:57-132,238-281; datasets/build.py:10-24)
//   -> zero pad to a multiple of 32           (new_datasets/coco_dataset.py:49-66)
// for ONE uint8 HWC frame already in HBM, plus the matching box transform.
//
// The reference resizes PIL images, so parity means Pillow's 8-bit separable bilinear resampler: per output index a
// window of (int)ceil(max(scale,1))*2+1 taps whose double-precision triangle weights are normalised and rounded to 22-bit
// fixed point; a horizontal pass rounds to uint8, a vertical pass rounds to uint8 again.  Here:
//   resample_coeffs_kernel : one lane per output column / row computes its window in fp64 (IEEE +,-,*,/ only -> the same
//                            integers as the CPU) ;  launched once per (shape -> shape), the table lives in the workspace
//   resample_h_kernel      : lane = (row, out column): u8 x 3 gathers through the (optionally mirrored) window
//   resample_v_norm_kernel : lane = (out row, out column) of the PADDED frame: vertical window, /255, -mean, /std in
//                            binary32 in the reference's operation order, three coalesced plane stores; zeros in the pad
// Bit-exact against Pillow through the oracle (tests/test_preprocess.py).  HBM-bound in principle (1.5 MB in, 10 MB out
// for 480x640 -> 800x1066) but at these sizes the two passes are launch/latency bound (~10 us).
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(preprocess);
#include <cmath>

#define RS_BITS 22

static int rs_ksize_host(int in_size, int out_size)
{
    const double scale = (double)in_size / (double)out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    return (int)std::ceil(fs) * 2 + 1;
}

__global__ __launch_bounds__(256) void resample_coeffs_kernel(int in_x, int out_x, int ks_x, int32_t *__restrict__ bx, int32_t *__restrict__ kx,
                                                             int in_y, int out_y, int ks_y, int32_t *__restrict__ by, int32_t *__restrict__ ky)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    int in_size, out_size, ks; int32_t *bounds, *kk;
    if (i < out_x) { in_size = in_x; out_size = out_x; ks = ks_x; bounds = bx; kk = kx; }
    else { i -= out_x; if (i >= out_y) return; in_size = in_y; out_size = out_y; ks = ks_y; bounds = by; kk = ky; }
    const double scale = (double)in_size / (double)out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * fs;
    const double center = 0.0 + ((double)i + 0.5) * scale;
    const double ss = 1.0 / fs;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = ((double)(x + xmin) - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    for (int x = 0; x < ks; ++x) {
        double v = 0.0;
        if (x < xmax) {
            double a = ((double)(x + xmin) - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            v = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) v = v / ww;
        }
        kk[(size_t)i * ks + x] = v < 0.0 ? (int32_t)(-0.5 + v * (double)(1 << RS_BITS)) : (int32_t)(0.5 + v * (double)(1 << RS_BITS));
    }
    bounds[2 * i] = xmin;
    bounds[2 * i + 1] = xmax;
}

__device__ __forceinline__ uint8_t rs_clip8(int32_t v)
{
    v >>= RS_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t *__restrict__ src, int h, int w, int flip, int ow, int ks,
                                                        const int32_t *__restrict__ bx, const int32_t *__restrict__ kx, uint8_t *__restrict__ tmp)
{
    const int xx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xx >= ow) return;
    const int xmin = bx[2 * xx], n = bx[2 * xx + 1];
    const int32_t *k = kx + (size_t)xx * ks;
    const uint8_t *row = src + (size_t)y * w * 3;
    int32_t s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        int sx = x + xmin;
        if (flip) sx = w - 1 - sx;
        const int32_t c = k[x];
        s0 += (int32_t)row[sx * 3 + 0] * c;
        s1 += (int32_t)row[sx * 3 + 1] * c;
        s2 += (int32_t)row[sx * 3 + 2] * c;
    }
    uint8_t *o = tmp + ((size_t)y * ow + xx) * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
}

struct NormConst { float mean[3], std[3]; };

__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t *__restrict__ tmp, int oh, int ow, int ph, int pw, int ks,
                                                             const int32_t *__restrict__ by, const int32_t *__restrict__ ky, NormConst nc,
                                                             float *__restrict__ out, uint8_t *__restrict__ out_u8)
{
    const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y;
    if (xx >= pw) return;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
    if (yy < oh && xx < ow) {
        const int ymin = by[2 * yy], n = by[2 * yy + 1];
        const int32_t *k = ky + (size_t)yy * ks;
        int32_t s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < n; ++y) {
            const uint8_t *p = tmp + ((size_t)(y + ymin) * ow + xx) * 3;
            const int32_t c = k[y];
            s0 += (int32_t)p[0] * c; s1 += (int32_t)p[1] * c; s2 += (int32_t)p[2] * c;
        }
        const uint8_t u0 = rs_clip8(s0), u1 = rs_clip8(s1), u2 = rs_clip8(s2);
        if (out_u8) { uint8_t *q = out_u8 + ((size_t)yy * ow + xx) * 3; q[0] = u0; q[1] = u1; q[2] = u2; }
        v0 = ((float)u0 / 255.0f - nc.mean[0]) / nc.std[0];          // F.to_tensor .div(255); F.normalize sub_(mean).div_(std)
        v1 = ((float)u1 / 255.0f - nc.mean[1]) / nc.std[1];
        v2 = ((float)u2 / 255.0f - nc.mean[2]) / nc.std[2];
    }
    if (!out) return;
    const size_t plane = (size_t)ph * pw, o = (size_t)yy * pw + xx;
    out[o] = v0; out[plane + o] = v1; out[2 * plane + o] = v2;
}

__global__ __launch_bounds__(256) void preprocess_boxes_kernel(const float4 *__restrict__ boxes, int64_t n, float fw, int flip, float rw, float rh,
                                                              float fow, float foh, float4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 b = boxes[i];
    if (flip) {                                                       // transforms.py:64-68: boxes[:, [2,1,0,3]] * [-1,1,-1,1] + [w,0,w,0]
        const float nx1 = b.z * -1.0f + fw, nx2 = b.x * -1.0f + fw;
        b.x = nx1; b.z = nx2;
    }
    b.x = b.x * rw; b.y = b.y * rh; b.z = b.z * rw; b.w = b.w * rh;   // :113-117
    out[i] = make_float4(b.x / fow, b.y / foh, b.z / fow, b.w / foh); // :276-280
}

struct PreWs { int32_t *bx, *by, *kx, *ky; uint8_t *tmp; size_t total; int ksx, ksy; };

static PreWs pre_ws_layout(void *base, int h, int w, int oh, int ow)
{
    PreWs p;
    p.ksx = rs_ksize_host(w, ow); p.ksy = rs_ksize_host(h, oh);
    size_t o = 0;
    auto take = [&](size_t bytes) { void *q = base ? (char *)base + o : nullptr; o += align_up(bytes, 256); return q; };
    p.bx = (int32_t *)take((size_t)ow * 8);
    p.by = (int32_t *)take((size_t)oh * 8);
    p.kx = (int32_t *)take((size_t)ow * p.ksx * 4);
    p.ky = (int32_t *)take((size_t)oh * p.ksy * 4);
    p.tmp = (uint8_t *)take((size_t)h * ow * 3);
    p.total = o;
    return p;
}

size_t frcnn_ws_preprocess(int64_t in_hw, int64_t out_hw)
{
    const int h = (int)(in_hw >> 32), w = (int)(in_hw & 0xFFFFFFFF), oh = (int)(out_hw >> 32), ow = (int)(out_hw & 0xFFFFFFFF);
    if (h < 1 || w < 1 || oh < 1 || ow < 1) return 0;
    return pre_ws_layout(nullptr, h, w, oh, ow).total;
}

FRCNN_EXPORT int frcnn_preprocess_image(const uint8_t *src_hwc, int h, int w, int flip, int oh, int ow, int pad_h, int pad_w,
                                        const float *mean_host, const float *std_host, float *out_chw, uint8_t *out_u8,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(h >= 1 && w >= 1 && oh >= 1 && ow >= 1 && pad_h >= oh && pad_w >= ow, "preprocess_image: bad shape %dx%d -> %dx%d pad %dx%d", h, w,
                  oh, ow, pad_h, pad_w);
    FRCNN_REQUIRE(h < (1 << 15) && w < (1 << 15) && pad_h < (1 << 15) && pad_w < (1 << 15), "preprocess_image: frame too large");
    FRCNN_REQUIRE(src_hwc && mean_host && std_host && (out_chw || out_u8) && workspace, "preprocess_image: NULL pointer");
    for (int c = 0; c < 3; ++c) FRCNN_REQUIRE(std_host[c] != 0.0f, "preprocess_image: std[%d] == 0", c);
    PreWs p = pre_ws_layout(workspace, h, w, oh, ow);
    if (workspace_bytes < p.total) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "preprocess_image: workspace %zu < %zu", workspace_bytes, p.total);
    hipStream_t s = (hipStream_t)stream;
    NormConst nc;
    for (int c = 0; c < 3; ++c) { nc.mean[c] = mean_host[c]; nc.std[c] = std_host[c]; }
    FRCNN_LAUNCH(resample_coeffs_kernel, dim3((unsigned)((ow + oh + 255) / 256)), dim3(256), 0, s, w, ow, p.ksx, p.bx, p.kx, h, oh,
                 p.ksy, p.by, p.ky);
    FRCNN_LAUNCH(resample_h_kernel, dim3((unsigned)((ow + 255) / 256), (unsigned)h), dim3(256), 0, s, src_hwc, h, w, flip ? 1 : 0, ow,
                 p.ksx, p.bx, p.kx, p.tmp);
    FRCNN_LAUNCH(resample_v_norm_kernel, dim3((unsigned)((pad_w + 255) / 256), (unsigned)(out_chw ? pad_h : oh)), dim3(256), 0, s,
                 p.tmp, oh, ow, pad_h, pad_w, p.ksy, p.by, p.ky, nc, out_chw, out_u8);
    FRCNN_CHECK_LAUNCH("preprocess kernels");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_preprocess_boxes(const float *boxes, int64_t n, int w, int h, int flip, int ow, int oh, float *out, void *stream)
{
    FRCNN_REQUIRE(n >= 0 && w >= 1 && h >= 1 && ow >= 1 && oh >= 1, "preprocess_boxes: bad argument");
    if (n == 0) return FRCNN_OK;
    FRCNN_REQUIRE(boxes && out, "preprocess_boxes: NULL pointer");
    const float rw = (float)((double)ow / (double)w), rh = (float)((double)oh / (double)h);
    FRCNN_LAUNCH(preprocess_boxes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)boxes, n,
                 (float)w, flip ? 1 : 0, rw, rh, (float)ow, (float)oh, (float4 *)out);
    FRCNN_CHECK_LAUNCH("preprocess_boxes_kernel");
    return FRCNN_OK;
}
