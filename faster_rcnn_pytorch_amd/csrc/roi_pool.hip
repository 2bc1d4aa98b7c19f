// roi_pool.hip -- torchvision.ops.RoIPool forward / backward as called at models/model.py:97,113 (gfx950).
//
// forward : one lane per output element, pw-minor, so the two stores (value + int32 argmax, the
//           25.7 MB that dominate the kernel at R=128, C=512) are perfectly coalesced; the RoI window
//           reads hit L1/L2 (one 37x62 fp32 channel plane is 9 KB).
// backward: one workgroup per channel; the channel's gradient plane lives in LDS, every
//           (roi, bin) of that channel is accumulated with ds_add_f32 and the plane is written
//           once with coalesced stores: no global atomics, no pre-zeroing of grad_feat.
//           Planes larger than the LDS budget fall back to zero-fill + global fp32 atomics.
// Algorithmic bytes: fwd 4*C*H*W + 16R + 8*R*C*PH*PW; bwd the same.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include <cfloat>

struct RoiBins { int sw, sh; float bw, bh; };

__device__ __forceinline__ RoiBins roi_bins(float4 b, float scale, int PH, int PW)
{
    // C round(): half away from zero
    const int sw = (int)roundf(b.x * scale), sh = (int)roundf(b.y * scale);
    const int ew = (int)roundf(b.z * scale), eh = (int)roundf(b.w * scale);
    const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
    RoiBins r;
    r.sw = sw; r.sh = sh;
    r.bw = (float)rw / (float)PW;
    r.bh = (float)rh / (float)PH;
    return r;
}

__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(const float *__restrict__ feat, int C, int H, int W,
                                                           const float4 *__restrict__ rois, int64_t total, int PH, int PW, float scale,
                                                           float *__restrict__ out, int32_t *__restrict__ argmax)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int pw = (int)(e % PW);
    const int ph = (int)((e / PW) % PH);
    const int64_t rc = e / ((int64_t)PW * PH);
    const int c = (int)(rc % C);
    const int r = (int)(rc / C);
    const RoiBins g = roi_bins(rois[r], scale, PH, PW);
    int hs = (int)floorf((float)ph * g.bh) + g.sh;
    int he = (int)ceilf((float)(ph + 1) * g.bh) + g.sh;
    int ws = (int)floorf((float)pw * g.bw) + g.sw;
    int we = (int)ceilf((float)(pw + 1) * g.bw) + g.sw;
    hs = min(max(hs, 0), H); he = min(max(he, 0), H);
    ws = min(max(ws, 0), W); we = min(max(we, 0), W);
    const bool empty = (he <= hs) || (we <= ws);
    float mv = empty ? 0.0f : -FLT_MAX;
    int mi = -1;
    const float *pl = feat + (size_t)c * H * W;
    for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) {
            const float v = pl[h * W + w];
            if (v > mv) { mv = v; mi = h * W + w; }
        }
    out[e] = mv;
    argmax[e] = mi;
}

__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ argmax,
                                                           int R, int C, int HW, int bins, float *__restrict__ grad_feat)
{
    extern __shared__ float plane[];                 // [HW]
    const int c = blockIdx.x;
    for (int i = threadIdx.x; i < HW; i += 256) plane[i] = 0.0f;
    __syncthreads();
    const int n = R * bins;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int r = e / bins, p = e - r * bins;
        const size_t idx = ((size_t)r * C + c) * bins + p;
        const int a = argmax[idx];
        if (a >= 0) atomicAdd(&plane[a], grad_out[idx]);
    }
    __syncthreads();
    float *dst = grad_feat + (size_t)c * HW;
    for (int i = threadIdx.x; i < HW; i += 256) dst[i] = plane[i];
}

__global__ __launch_bounds__(256) void roi_pool_bwd_atomic_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ argmax,
                                                                  int64_t total, int C, int HW, int bins, float *__restrict__ grad_feat)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int a = argmax[e];
    if (a < 0) return;
    const int c = (int)((e / bins) % C);
    atomicAdd(grad_feat + (size_t)c * HW + a, grad_out[e]);
}

FRCNN_EXPORT int frcnn_roi_pool_fwd(const float *feat, int C, int H, int W, const float *rois, int64_t R, int PH, int PW,
                                    float spatial_scale, float *out, int32_t *argmax, void *stream)
{
    FRCNN_REQUIRE(C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "roi_pool_fwd: bad shape");
    if (R == 0) return FRCNN_OK;
    FRCNN_REQUIRE(feat && rois && out && argmax, "roi_pool_fwd: NULL pointer");
    FRCNN_REQUIRE((int64_t)H * W < ((int64_t)1 << 31), "roi_pool_fwd: plane too large for int32 argmax");
    const int64_t total = R * C * PH * PW;
    FRCNN_REQUIRE(total < ((int64_t)1 << 38), "roi_pool_fwd: output too large");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(KID_ROI_POOL_FWD, roi_pool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat, C, H, W,
                 (const float4 *)rois, total, PH, PW, spatial_scale, out, argmax);
    FRCNN_CHECK_LAUNCH("roi_pool_fwd_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_roi_pool_bwd(const float *grad_out, const int32_t *argmax, int64_t R, int C, int H, int W, int PH, int PW,
                                    float *grad_feat, void *stream)
{
    FRCNN_REQUIRE(C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && R >= 0, "roi_pool_bwd: bad shape");
    FRCNN_REQUIRE(grad_feat, "roi_pool_bwd: NULL grad_feat");
    hipStream_t s = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    if (R == 0) {
        if (hipMemsetAsync(grad_feat, 0, (size_t)C * HW * 4, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "roi_pool_bwd: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(grad_out && argmax, "roi_pool_bwd: NULL pointer");
    FRCNN_REQUIRE(R * PH * PW < ((int64_t)1 << 31), "roi_pool_bwd: R*bins too large");
    if (HW * 4 <= 64 * 1024) {
        FRCNN_LAUNCH(KID_ROI_POOL_BWD, roi_pool_bwd_kernel, dim3(C), dim3(256), (size_t)HW * 4, s, grad_out, argmax, (int)R, C, (int)HW,
                     PH * PW, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_pool_bwd_kernel");
    } else {
        if (hipMemsetAsync(grad_feat, 0, (size_t)C * HW * 4, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "roi_pool_bwd: memset failed");
        const int64_t total = R * C * PH * PW;
        FRCNN_LAUNCH(KID_ROI_POOL_BWD, roi_pool_bwd_atomic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, grad_out, argmax,
                     total, C, (int)HW, PH * PW, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_pool_bwd_atomic_kernel");
    }
    return FRCNN_OK;
}
