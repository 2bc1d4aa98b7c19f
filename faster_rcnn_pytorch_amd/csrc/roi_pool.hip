// roi_pool.hip -- torchvision.ops.RoIPool forward / backward as called at models/model.py:97,113 (gfx950).
//
// forward : (7x7, planes that fit LDS -- the shape the reference runs) roi_pool_fwd_lds_kernel: a workgroup stages CB
//           adjacent channel planes in LDS with one coalesced pass, builds the bin tables of its RoIs once, and each lane
//           = (RoI, bin) scans its window in LDS for the CB channels; out + int32 argmax (the 25.7 MB that dominate at
//           R=128, C=512) leave in contiguous CB*49-element runs.  Other shapes: one lane per output element, windows from L1/L2.
// backward: one workgroup per channel group; the gradient planes live in LDS, every (roi, bin) is accumulated with
//           ds_add_f32 and each plane is written once with coalesced stores: no global atomics, no pre-zeroing of
//           grad_feat.  Planes larger than the LDS budget fall back to zero-fill + global fp32 atomics.
// Algorithmic bytes: fwd 4*C*H*W + 16R + 8*R*C*PH*PW; bwd the same.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include <cfloat>

#ifndef ROI_FWD_CB
#define ROI_FWD_CB 4
#endif
#ifndef ROI_FWD_RB
#define ROI_FWD_RB 20
#endif
#ifndef ROI_BWD_CB
#define ROI_BWD_CB 1
#endif

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

// ------------------------------------------------------------------------------------------------
// LDS-staged forward (the shape the reference runs: PHxPW = 7x7, plane <= 64 KB / CB):
// grid (ceil(C/CB), ceil(R/RB)); a block stages CB adjacent channel planes in LDS with one coalesced
// pass, builds the bin-boundary table of its RB RoIs once, then every lane produces outputs
// (roi, channel, bin) with the window scan served from LDS.  For one RoI the CB*49 outputs of the
// block are contiguous in memory (784 B at CB = 4): the stores stay coalesced.
// ------------------------------------------------------------------------------------------------
#ifndef ROI_FWD_BS
#define ROI_FWD_BS 512
#endif
template <int CB, int PH, int PW>
__global__ __launch_bounds__(ROI_FWD_BS) void roi_pool_fwd_lds_kernel(const float *__restrict__ feat, int C, int H, int W,
                                                               const float4 *__restrict__ rois, int R, int RB, float scale,
                                                               float *__restrict__ out, int32_t *__restrict__ argmax)
{
    extern __shared__ float smem[];
    const int HW = H * W;
    float *planes = smem;                                        // [CB][HW]
    int *tab = (int *)(smem + CB * HW);                          // [RB][PH + PW]: (hs | he << 16) x PH, (ws | we << 16) x PW
    constexpr int BINS = PH * PW;
    constexpr int TW = PH + PW;
    constexpr int GROUPS = ROI_FWD_BS / BINS;                           // RoIs processed concurrently by the block
    static_assert(GROUPS >= 1, "PH * PW must fit one block");
    const int c0 = blockIdx.x * CB;
    const int r0 = blockIdx.y * RB;
    const int nr = min(RB, R - r0);
    const int nch = min(CB, C - c0);
    const float *src = feat + (size_t)c0 * HW;
    {   // stage the planes: 8 independent loads in flight per lane
        const int n_stage = nch * HW;
        for (int base = 0; base < n_stage; base += ROI_FWD_BS * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * ROI_FWD_BS + threadIdx.x;
                v[u] = src[min(i, n_stage - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * ROI_FWD_BS + threadIdx.x;
                if (i < n_stage) planes[i] = v[u];
            }
        }
    }
    for (int t = threadIdx.x; t < nr * TW; t += ROI_FWD_BS) {
        const int rl = t / TW, k = t - rl * TW;
        const RoiBins g = roi_bins(rois[r0 + rl], scale, PH, PW);
        if (k < PH) {
            const int hs = (int)floorf((float)k * g.bh) + g.sh, he = (int)ceilf((float)(k + 1) * g.bh) + g.sh;
            tab[t] = min(max(hs, 0), H) | (min(max(he, 0), H) << 16);
        } else {
            const int q = k - PH;
            const int ws = (int)floorf((float)q * g.bw) + g.sw, we = (int)ceilf((float)(q + 1) * g.bw) + g.sw;
            tab[t] = min(max(ws, 0), W) | (min(max(we, 0), W) << 16);
        }
    }
    __syncthreads();
    // lane -> fixed (RoI group, bin); the CB channels of the block are an INNER loop, so the window geometry, the loop
    // control and the clamps are paid once per CB outputs and the CB LDS reads of a pixel are independent
    const int grp = threadIdx.x / BINS;
    const int p = threadIdx.x - grp * BINS;
    const int ph = p / PW, pw = p - ph * PW;
    if (grp >= GROUPS) return;
    size_t e = ((size_t)(r0 + grp) * C + c0) * BINS + p;
    const size_t estep = (size_t)GROUPS * C * BINS;
    for (int rl = grp; rl < nr; rl += GROUPS, e += estep) {
        const int th = tab[rl * TW + ph], tw = tab[rl * TW + PH + pw];
        const int hs = th & 0xFFFF, he = th >> 16, ws = tw & 0xFFFF, we = tw >> 16;
        const bool empty = (he <= hs) || (we <= ws);
        float mv[CB];
        int mi[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) { mv[c] = empty ? 0.0f : -FLT_MAX; mi[c] = -1; }
        for (int h = hs; h < he; ++h) {
            const int rowoff = h * W;
            for (int w = ws; w < we; w += 2) {                   // 2 pixels x CB planes = 2*CB independent LDS reads in flight
                const int w1 = min(w + 1, we - 1);               // a clamped duplicate can never be > the running max
                float v0[CB], v1[CB];
#pragma unroll
                for (int c = 0; c < CB; ++c) { v0[c] = planes[c * HW + rowoff + w]; v1[c] = planes[c * HW + rowoff + w1]; }
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    if (v0[c] > mv[c]) { mv[c] = v0[c]; mi[c] = rowoff + w; }
                    if (v1[c] > mv[c]) { mv[c] = v1[c]; mi[c] = rowoff + w1; }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c < nch) { out[e + (size_t)c * BINS] = mv[c]; argmax[e + (size_t)c * BINS] = mi[c]; }
    }
}

// LDS-accumulating backward, CB adjacent channels per block (contiguous CB*bins runs of grad_out / argmax)
template <int CB>
__global__ __launch_bounds__(512) void roi_pool_bwd_lds_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ argmax,
                                                               int R, int C, int HW, int bins, float *__restrict__ grad_feat)
{
    extern __shared__ float plane[];                 // [CB][HW]
    // adjacent channels share the cache lines at the ends of their 196-byte runs: keep neighbours on ONE XCD (workgroups are
    // dealt round-robin to the 8 XCDs) so that the second touch of such a line is an L2 hit instead of another HBM fetch
    const int nb = gridDim.x;
    const int bx = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int c0 = bx * CB;
    const int nch = min(CB, C - c0);
    for (int i = threadIdx.x; i < nch * HW; i += 512) plane[i] = 0.0f;
    __syncthreads();
    // thread -> (roi slot, position inside the block's contiguous CB*bins run); U independent RoIs in flight per
    // thread so that 2*U global loads are outstanding before the first ds_add (the loads must not be sunk into
    // the `argmax >= 0` branch: that serialises two HBM round trips per element)
    const int run = nch * bins;
    const int rpp = 512 / run;                        // RoIs covered per pass of the block
    const int rsub = threadIdx.x / run, rem = threadIdx.x - rsub * run;
    if (rsub < rpp) {
        const int pl_off = (rem / bins) * HW;
        constexpr int U = 16;
        for (int r0 = rsub; r0 < R; r0 += rpp * U) {
            int a[U];
            float g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rpp;
                const size_t idx = ((size_t)min(r, R - 1) * C + c0) * bins + rem;
                a[u] = __builtin_nontemporal_load(argmax + idx);
                g[u] = __builtin_nontemporal_load(grad_out + idx);
                if (r >= R) a[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (a[u] >= 0) atomicAdd(&plane[pl_off + a[u]], g[u]);
        }
    }
    __syncthreads();
    float *dst = grad_feat + (size_t)c0 * HW;
    for (int i = threadIdx.x; i < nch * HW; i += 512) dst[i] = plane[i];
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
    constexpr int CB = ROI_FWD_CB;
    const size_t plane_bytes = (size_t)CB * H * W * 4;
    if (PH == 7 && PW == 7 && plane_bytes <= 48 * 1024 && R < (1 << 24)) {
        const int RB = ROI_FWD_RB;
        const size_t shmem = plane_bytes + (size_t)RB * (7 + 7) * 4;
        FRCNN_LAUNCH(KID_ROI_POOL_FWD, (roi_pool_fwd_lds_kernel<CB, 7, 7>), dim3((C + CB - 1) / CB, (unsigned)((R + RB - 1) / RB)), dim3(ROI_FWD_BS),
                     shmem, s, feat, C, H, W, (const float4 *)rois, (int)R, RB, spatial_scale, out, argmax);
        FRCNN_CHECK_LAUNCH("roi_pool_fwd_lds_kernel");
        return FRCNN_OK;
    }
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
    if (HW * 4 * ROI_BWD_CB <= 64 * 1024) {
        FRCNN_LAUNCH(KID_ROI_POOL_BWD, roi_pool_bwd_lds_kernel<ROI_BWD_CB>, dim3((C + ROI_BWD_CB - 1) / ROI_BWD_CB), dim3(512), (size_t)HW * 4 * ROI_BWD_CB, s, grad_out, argmax, (int)R, C,
                     (int)HW, PH * PW, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_pool_bwd_lds_kernel");
    } else if (HW * 4 <= 64 * 1024) {
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
