// roi_align.hip -- torchvision MultiScaleRoIAlign forward / backward as called at
// models/new_model.py:127,143 (gfx950): level mapper + per-level roi_align in ONE launch.
//
// forward : one lane per output element (pw-minor -> coalesced 4-byte stores, 25.7 MB at R=512,C=256);
//           the level of the RoI is recomputed per lane (sqrt + the deterministic log2) instead of a
//           separate mapper launch + per-level gathers/scatters as torchvision does.
// backward: one lane per output element, 4 samples x 4 corners fp32 atomics into the level's
//           gradient plane (hardware global_atomic_add_f32; order-nondeterministic, tolerance 1e-4).
#include "frcnn_common.h"
#include "frcnn_internal.h"

struct MsLevels {
    int n_levels;
    const float *feat[FRCNN_MAX_LEVELS];
    float *grad[FRCNN_MAX_LEVELS];
    int H[FRCNN_MAX_LEVELS], W[FRCNN_MAX_LEVELS];
    float scale[FRCNN_MAX_LEVELS];
};

// torchvision LevelMapper: floor(k0 + log2(sqrt(area)/s0) + eps) clamped to [k_min,k_max], minus k_min
__device__ __forceinline__ int level_of(float4 b, int k_min, int k_max, float s0, int k0, float eps)
{
    const float area = (b.z - b.x) * (b.w - b.y);
    const float s = __fsqrt_rn(area);
    float t = (float)k0 + det_log2f(s / s0);
    t = t + eps;
    float k = __builtin_floorf(t);
    if (!(k >= (float)k_min)) k = (float)k_min;
    if (k > (float)k_max) k = (float)k_max;
    return (int)k - k_min;
}

__global__ __launch_bounds__(256) void roi_level_map_kernel(const float4 *__restrict__ rois, int64_t R, int k_min, int k_max, float s0,
                                                            int k0, float eps, int32_t *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < R) out[r] = level_of(rois[r], k_min, k_max, s0, k0, eps);
}

struct AlignGeom { float sw, sh, bw, bh; int gw, gh; float cnt; };

__device__ __forceinline__ AlignGeom align_geom(float4 b, float scale, int PH, int PW, int sampling_ratio, bool aligned)
{
    const float off = aligned ? 0.5f : 0.0f;
    AlignGeom g;
    g.sw = b.x * scale - off; g.sh = b.y * scale - off;
    const float ew = b.z * scale - off, eh = b.w * scale - off;
    float rw = ew - g.sw, rh = eh - g.sh;
    if (!aligned) { if (rw < 1.0f) rw = 1.0f; if (rh < 1.0f) rh = 1.0f; }
    g.bh = rh / (float)PH; g.bw = rw / (float)PW;
    g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    g.cnt = (float)max(g.gh * g.gw, 1);
    return g;
}

struct Bilin { int yl, yh, xl, xh; float w1, w2, w3, w4; bool ok; };

__device__ __forceinline__ Bilin bilin_setup(int H, int W, float y, float x)
{
    Bilin s;
    s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    s.yl = yl; s.yh = yh; s.xl = xl; s.xh = xh;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    return s;
}

__global__ __launch_bounds__(256) void roi_align_fwd_kernel(MsLevels L, int C, const float4 *__restrict__ rois, int64_t total, int PH, int PW,
                                                            int sampling_ratio, int aligned, int k_min, float s0, int k0,
                                                            float *__restrict__ out, int32_t *__restrict__ out_level)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int pw = (int)(e % PW);
    const int ph = (int)((e / PW) % PH);
    const int64_t rc = e / ((int64_t)PW * PH);
    const int c = (int)(rc % C);
    const int r = (int)(rc / C);
    const float4 b = rois[r];
    const int l = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
    if (out_level && c == 0 && ph == 0 && pw == 0) out_level[r] = l;
    const int H = L.H[l], W = L.W[l];
    const float *pl = L.feat[l] + (size_t)c * H * W;
    const AlignGeom g = align_geom(b, L.scale[l], PH, PW, sampling_ratio, aligned != 0);
    float acc = 0.0f;
    for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
            const float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
            const Bilin s = bilin_setup(H, W, y, x);
            float v = 0.0f;
            if (s.ok)
                v = s.w1 * pl[s.yl * W + s.xl] + s.w2 * pl[s.yl * W + s.xh] + s.w3 * pl[s.yh * W + s.xl] + s.w4 * pl[s.yh * W + s.xh];
            acc += v;
        }
    }
    out[e] = acc / g.cnt;
}

__global__ __launch_bounds__(256) void roi_align_bwd_kernel(MsLevels L, int C, const float4 *__restrict__ rois, int64_t total, int PH, int PW,
                                                            int sampling_ratio, int aligned, int k_min, float s0, int k0,
                                                            const float *__restrict__ grad_out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int pw = (int)(e % PW);
    const int ph = (int)((e / PW) % PH);
    const int64_t rc = e / ((int64_t)PW * PH);
    const int c = (int)(rc % C);
    const int r = (int)(rc / C);
    const float4 b = rois[r];
    const int l = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
    const int H = L.H[l], W = L.W[l];
    float *pl = L.grad[l] + (size_t)c * H * W;
    const AlignGeom g = align_geom(b, L.scale[l], PH, PW, sampling_ratio, aligned != 0);
    const float go = grad_out[e] / g.cnt;
    for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
            const float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / (float)g.gw;
            const Bilin s = bilin_setup(H, W, y, x);
            if (!s.ok) continue;
            atomicAdd(pl + s.yl * W + s.xl, go * s.w1);
            atomicAdd(pl + s.yl * W + s.xh, go * s.w2);
            atomicAdd(pl + s.yh * W + s.xl, go * s.w3);
            atomicAdd(pl + s.yh * W + s.xh, go * s.w4);
        }
    }
}

FRCNN_EXPORT int frcnn_roi_level_map(const float *rois, int64_t R, int k_min, int k_max, float s0, int k0, float eps, int32_t *out_level,
                                     void *stream)
{
    FRCNN_REQUIRE(R >= 0 && k_max >= k_min && s0 > 0.f, "roi_level_map: bad argument");
    if (R == 0) return FRCNN_OK;
    FRCNN_REQUIRE(rois && out_level, "roi_level_map: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(KID_ROI_LEVEL_MAP, roi_level_map_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, (const float4 *)rois, R, k_min,
                 k_max, s0, k0, eps, out_level);
    FRCNN_CHECK_LAUNCH("roi_level_map_kernel");
    return FRCNN_OK;
}

static int fill_levels(MsLevels *L, const float *const *feats, float *const *grads, const int *H, const int *W, const float *scales, int n_levels)
{
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS, "ms_roi_align: n_levels %d not in [1,%d]", n_levels, FRCNN_MAX_LEVELS);
    FRCNN_REQUIRE(H && W && scales && (feats || grads), "ms_roi_align: NULL level table");
    L->n_levels = n_levels;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        const int k = l < n_levels ? l : 0;
        FRCNN_REQUIRE(H[k] > 0 && W[k] > 0 && scales[k] > 0.f, "ms_roi_align: bad level %d", k);
        FRCNN_REQUIRE((!feats || feats[k]) && (!grads || grads[k]), "ms_roi_align: NULL feature pointer at level %d", k);
        L->feat[l] = feats ? feats[k] : nullptr;
        L->grad[l] = grads ? grads[k] : nullptr;
        L->H[l] = H[k]; L->W[l] = W[k]; L->scale[l] = scales[k];
    }
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_ms_roi_align_fwd(const float *const *feats, const int *H, const int *W, const float *scales, int n_levels, int C,
                                        const float *rois, int64_t R, int PH, int PW, int sampling_ratio, int aligned, int k_min, float s0,
                                        int k0, float *out, int32_t *out_level, void *stream)
{
    FRCNN_REQUIRE(C > 0 && PH > 0 && PW > 0 && R >= 0 && sampling_ratio >= 0 && s0 > 0.f, "ms_roi_align_fwd: bad argument");
    if (R == 0) return FRCNN_OK;
    FRCNN_REQUIRE(rois && out, "ms_roi_align_fwd: NULL pointer");
    MsLevels L;
    int rc = fill_levels(&L, feats, nullptr, H, W, scales, n_levels);
    if (rc) return rc;
    const int64_t total = R * C * PH * PW;
    FRCNN_REQUIRE(total < ((int64_t)1 << 38), "ms_roi_align_fwd: output too large");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(KID_ROI_ALIGN_FWD, roi_align_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, C, (const float4 *)rois,
                 total, PH, PW, sampling_ratio, aligned, k_min, s0, k0, out, out_level);
    FRCNN_CHECK_LAUNCH("roi_align_fwd_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_ms_roi_align_bwd(const float *grad_out, float *const *grad_feats, const int *H, const int *W, const float *scales,
                                        int n_levels, int C, const float *rois, int64_t R, int PH, int PW, int sampling_ratio, int aligned,
                                        int k_min, float s0, int k0, void *stream)
{
    FRCNN_REQUIRE(C > 0 && PH > 0 && PW > 0 && R >= 0 && sampling_ratio >= 0 && s0 > 0.f, "ms_roi_align_bwd: bad argument");
    if (R == 0) return FRCNN_OK;
    FRCNN_REQUIRE(rois && grad_out, "ms_roi_align_bwd: NULL pointer");
    MsLevels L;
    int rc = fill_levels(&L, nullptr, grad_feats, H, W, scales, n_levels);
    if (rc) return rc;
    const int64_t total = R * C * PH * PW;
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(KID_ROI_ALIGN_BWD, roi_align_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, C, (const float4 *)rois,
                 total, PH, PW, sampling_ratio, aligned, k_min, s0, k0, grad_out);
    FRCNN_CHECK_LAUNCH("roi_align_bwd_kernel");
    return FRCNN_OK;
}
