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
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(roi_pool);
#include <cfloat>

#ifndef ROI_FWD_RB
#define ROI_FWD_RB 20
#endif
#ifndef ROI_BWD_CB
#define ROI_BWD_CB 2
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
// LDS-staged forward (the shape the reference runs: PHxPW = 7x7, 4 planes <= 48 KB):
// grid (ceil(C/4), ceil(R/RB)); a block stages FOUR adjacent channel planes in LDS, INTERLEAVED per pixel (float4 = the four
// channels of one pixel), builds the bin-boundary table of its RB RoIs once, then every lane produces outputs (roi, bin) x 4
// channels: one ds_read_b128 per window pixel serves all four channels (round 1 kept the planes separate: four ds_read_b32 and
// ~8 instructions per pixel and channel; the kernel was bound by instruction issue at 18 us).  For one RoI the 4*49 outputs of
// the block are contiguous in memory (784 B): the stores stay coalesced.
// AT = the argmax element: int32_t (the torchvision-shaped ABI) or uint16_t (library-private, planes < 65535 pixels: 6.4 MB less
// to write and, in backward, to read at R = 128, C = 512; 0xFFFF = empty bin).
// ------------------------------------------------------------------------------------------------
#ifndef ROI_FWD_BS
#define ROI_FWD_BS 512
#endif
template <typename AT> __device__ __forceinline__ AT roi_arg_enc(int mi);
template <> __device__ __forceinline__ int32_t roi_arg_enc<int32_t>(int mi) { return mi; }
template <> __device__ __forceinline__ uint16_t roi_arg_enc<uint16_t>(int mi) { return (uint16_t)(mi < 0 ? 0xFFFF : mi); }
template <typename AT> __device__ __forceinline__ int roi_arg_dec(AT a);
template <> __device__ __forceinline__ int roi_arg_dec<int32_t>(int32_t a) { return a; }
template <> __device__ __forceinline__ int roi_arg_dec<uint16_t>(uint16_t a) { return a == 0xFFFF ? -1 : (int)a; }

template <int PH, int PW, typename AT>
__global__ __launch_bounds__(ROI_FWD_BS) void roi_pool_fwd_lds_kernel(const float *__restrict__ feat, int C, int H, int W,
                                                               const float4 *__restrict__ rois, int R, int RB, float scale,
                                                               float *__restrict__ out, AT *__restrict__ argmax)
{
    constexpr int CB = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = H * W;
    float4 *px = (float4 *)smem;                                 // [HW]: the four channels of a pixel
    int *tab = (int *)(smem + CB * HW);                          // [RB][PH + PW]: (hs | he << 16) x PH, (ws | we << 16) x PW
    constexpr int BINS = PH * PW;
    constexpr int TW = PH + PW;
    constexpr int GROUPS = ROI_FWD_BS / BINS;                    // RoIs processed concurrently by the block
    static_assert(GROUPS >= 1, "PH * PW must fit one block");
    const int c0 = blockIdx.x * CB;
    const int r0 = blockIdx.y * RB;
    const int nr = min(RB, R - r0);
    const int nch = min(CB, C - c0);
    const float *src = feat + (size_t)c0 * HW;
    {   // stage the planes interleaved per pixel: a thread owns pixels t, t + BS, ...: four coalesced loads (one per plane), ONE
        // ds_write_b128.  (The first version walked the flat (channel, pixel) index: a division by H * W and a 4-way bank-conflicted
        // 4-byte LDS store per element -- ~540 of the kernel's ~1000 instructions per thread.)
        for (int p0 = threadIdx.x; p0 < HW; p0 += ROI_FWD_BS * 2) {
            const int p1 = p0 + ROI_FWD_BS;
            float4 a, b;
            a.x = src[p0];
            a.y = nch > 1 ? src[HW + p0] : 0.0f;
            a.z = nch > 2 ? src[2 * HW + p0] : 0.0f;
            a.w = nch > 3 ? src[3 * HW + p0] : 0.0f;
            const int q1 = min(p1, HW - 1);
            b.x = src[q1];
            b.y = nch > 1 ? src[HW + q1] : 0.0f;
            b.z = nch > 2 ? src[2 * HW + q1] : 0.0f;
            b.w = nch > 3 ? src[3 * HW + q1] : 0.0f;
            px[p0] = a;
            if (p1 < HW) px[p1] = b;
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
        const float init = empty ? 0.0f : -FLT_MAX;
        float m0 = init, m1 = init, m2 = init, m3 = init;
        int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
        for (int h = hs; h < he; ++h) {
            const int rowoff = h * W;
            for (int w = ws; w < we; w += 2) {                   // two pixels = two independent ds_read_b128 in flight
                const int w1 = min(w + 1, we - 1);               // a clamped duplicate can never be > the running max
                const float4 a = px[rowoff + w], b = px[rowoff + w1];
                const int ia = rowoff + w, ib = rowoff + w1;
                if (a.x > m0) { m0 = a.x; i0 = ia; }
                if (a.y > m1) { m1 = a.y; i1 = ia; }
                if (a.z > m2) { m2 = a.z; i2 = ia; }
                if (a.w > m3) { m3 = a.w; i3 = ia; }
                if (b.x > m0) { m0 = b.x; i0 = ib; }
                if (b.y > m1) { m1 = b.y; i1 = ib; }
                if (b.z > m2) { m2 = b.z; i2 = ib; }
                if (b.w > m3) { m3 = b.w; i3 = ib; }
            }
        }
        out[e] = m0; argmax[e] = roi_arg_enc<AT>(i0);
        if (nch > 1) { out[e + BINS] = m1; argmax[e + BINS] = roi_arg_enc<AT>(i1); }
        if (nch > 2) { out[e + 2 * BINS] = m2; argmax[e + 2 * BINS] = roi_arg_enc<AT>(i2); }
        if (nch > 3) { out[e + 3 * BINS] = m3; argmax[e + 3 * BINS] = roi_arg_enc<AT>(i3); }
    }
}

// LDS-accumulating backward, CB adjacent channels per block (contiguous CB*bins runs of grad_out / argmax)
template <int CB, typename AT>
__global__ __launch_bounds__(512) void roi_pool_bwd_lds_kernel(const float *__restrict__ grad_out, const AT *__restrict__ argmax,
                                                               int R, int C, int HW, int bins, float *__restrict__ grad_feat)
{
    extern __shared__ float plane[];                 // [CB][HW]
    // adjacent channels share the cache lines at the ends of their runs: keep neighbours on ONE XCD (workgroups are
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
#ifndef ROI_BWD_U
#define ROI_BWD_U 8                       // 8 / 16 / 32 RoIs in flight per thread: 24.4 / 25.4 / 29.7 us (HIP events)
#endif
        constexpr int U = ROI_BWD_U;
        for (int r0 = rsub; r0 < R; r0 += rpp * U) {
            int a[U];
            float g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rpp;
                const size_t idx = ((size_t)min(r, R - 1) * C + c0) * bins + rem;
                a[u] = roi_arg_dec<AT>(__builtin_nontemporal_load(argmax + idx));
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

template <typename AT>
static int roi_pool_fwd_launch(const float *feat, int C, int H, int W, const float *rois, int64_t R, int PH, int PW, float spatial_scale,
                               float *out, AT *argmax, hipStream_t s)
{
    const int RB = ROI_FWD_RB;
    const size_t shmem = (size_t)4 * H * W * 4 + (size_t)RB * (7 + 7) * 4;
    FRCNN_LAUNCH((roi_pool_fwd_lds_kernel<7, 7, AT>), dim3((C + 3) / 4, (unsigned)((R + RB - 1) / RB)), dim3(ROI_FWD_BS),
                 shmem, s, feat, C, H, W, (const float4 *)rois, (int)R, RB, spatial_scale, out, argmax);
    FRCNN_CHECK_LAUNCH("roi_pool_fwd_lds_kernel");
    return FRCNN_OK;
}

template <typename AT>
static int roi_pool_bwd_launch(const float *grad_out, const AT *argmax, int64_t R, int C, int64_t HW, int bins, float *grad_feat, hipStream_t s)
{
    FRCNN_LAUNCH((roi_pool_bwd_lds_kernel<ROI_BWD_CB, AT>), dim3((C + ROI_BWD_CB - 1) / ROI_BWD_CB), dim3(512), (size_t)HW * 4 * ROI_BWD_CB, s,
                 grad_out, argmax, (int)R, C, (int)HW, bins, grad_feat);
    FRCNN_CHECK_LAUNCH("roi_pool_bwd_lds_kernel");
    return FRCNN_OK;
}

// Library-private 16-bit argmax pair (the autograd path of the host layer): same results as the int32 entry points.
FRCNN_EXPORT int frcnn_roi_pool_fwd_a16(const float *feat, int C, int H, int W, const float *rois, int64_t R, float spatial_scale,
                                        float *out, uint16_t *argmax16, void *stream)
{
    FRCNN_REQUIRE(C > 0 && H > 0 && W > 0 && R >= 0, "roi_pool_fwd_a16: bad shape");
    if (R == 0) return FRCNN_OK;
    FRCNN_REQUIRE(feat && rois && out && argmax16, "roi_pool_fwd_a16: NULL pointer");
    FRCNN_REQUIRE((int64_t)H * W < 65535 && (size_t)4 * H * W * 4 <= 48 * 1024 && R < (1 << 24),
                  "roi_pool_fwd_a16: plane %dx%d does not fit the 16-bit argmax / LDS path (use frcnn_roi_pool_fwd)", H, W);
    return roi_pool_fwd_launch<uint16_t>(feat, C, H, W, rois, R, 7, 7, spatial_scale, out, argmax16, (hipStream_t)stream);
}

FRCNN_EXPORT int frcnn_roi_pool_bwd_a16(const float *grad_out, const uint16_t *argmax16, int64_t R, int C, int H, int W, float *grad_feat,
                                        void *stream)
{
    FRCNN_REQUIRE(C > 0 && H > 0 && W > 0 && R >= 0, "roi_pool_bwd_a16: bad shape");
    FRCNN_REQUIRE(grad_feat, "roi_pool_bwd_a16: NULL grad_feat");
    hipStream_t s = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    FRCNN_REQUIRE(HW < 65535 && HW * 4 * ROI_BWD_CB <= 64 * 1024, "roi_pool_bwd_a16: plane does not fit the 16-bit argmax / LDS path");
    if (R == 0) {
        if (hipMemsetAsync(grad_feat, 0, (size_t)C * HW * 4, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "roi_pool_bwd_a16: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(grad_out && argmax16, "roi_pool_bwd_a16: NULL pointer");
    FRCNN_REQUIRE(R * 49 < ((int64_t)1 << 31), "roi_pool_bwd_a16: R*bins too large");
    return roi_pool_bwd_launch<uint16_t>(grad_out, argmax16, R, C, HW, 49, grad_feat, s);
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
    const size_t plane_bytes = (size_t)4 * H * W * 4;
    if (PH == 7 && PW == 7 && plane_bytes <= 48 * 1024 && R < (1 << 24))
        return roi_pool_fwd_launch<int32_t>(feat, C, H, W, rois, R, PH, PW, spatial_scale, out, argmax, s);
    FRCNN_LAUNCH(roi_pool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, feat, C, H, W,
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
    // the CB-channel LDS kernel maps a thread to (RoI slot, element of the CB * bins run): it needs at least one whole run per pass of
    // its 512 threads; larger bin grids (e.g. 17 x 17) take the one-channel kernel, which strides over any run length
    if (HW * 4 * ROI_BWD_CB <= 64 * 1024 && (int64_t)PH * PW * ROI_BWD_CB <= 512) {
        return roi_pool_bwd_launch<int32_t>(grad_out, argmax, R, C, HW, PH * PW, grad_feat, s);
    } else if (HW * 4 <= 64 * 1024) {
        FRCNN_LAUNCH(roi_pool_bwd_kernel, dim3(C), dim3(256), (size_t)HW * 4, s, grad_out, argmax, (int)R, C, (int)HW,
                     PH * PW, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_pool_bwd_kernel");
    } else {
        if (hipMemsetAsync(grad_feat, 0, (size_t)C * HW * 4, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "roi_pool_bwd: memset failed");
        const int64_t total = R * C * PH * PW;
        FRCNN_LAUNCH(roi_pool_bwd_atomic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, grad_out, argmax,
                     total, C, (int)HW, PH * PW, grad_feat);
        FRCNN_CHECK_LAUNCH("roi_pool_bwd_atomic_kernel");
    }
    return FRCNN_OK;
}
