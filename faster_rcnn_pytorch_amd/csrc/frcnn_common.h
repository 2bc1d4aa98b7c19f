// frcnn_common.h -- shared host/device helpers of libfrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/frcnn_hip.h"

#define FRCNN_EXPORT extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------------
// host side: error reporting + per-kernel event timing
// ---------------------------------------------------------------------------------------------
// Per-kernel HIP-event timing: every launch site registers its kernel under the kernel's OWN name (template arguments stripped), so
// the names frcnn_prof_kernel_name() reports are exactly the names rocprofv3 and the PMC passes print -- no hand-kept id table.
#define FRCNN_PROF_MAX_KERNELS 128
int frcnn_prof_register(const char *kernel_expr);

int frcnn_set_error(int code, const char *fmt, ...);
void frcnn_prof_begin(int kid, hipStream_t s);
void frcnn_prof_end(int kid, hipStream_t s);
bool frcnn_prof_on();

struct FrcnnProfScope {
    int kid; hipStream_t s; bool on;
    FrcnnProfScope(int k, hipStream_t st) : kid(k), s(st), on(frcnn_prof_on()) { if (on) frcnn_prof_begin(kid, s); }
    ~FrcnnProfScope() { if (on) frcnn_prof_end(kid, s); }
};

#define FRCNN_LAUNCH(kernel, grid, block, shmem, stream, ...)                              \
    do {                                                                                   \
        static const int _kid = frcnn_prof_register(#kernel);                              \
        FrcnnProfScope _prof(_kid, (stream));                                              \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);               \
    } while (0)

#define FRCNN_CHECK_LAUNCH(what)                                                           \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(_e)); \
    } while (0)

#define FRCNN_REQUIRE(cond, ...)                                                           \
    do { if (!(cond)) return frcnn_set_error(FRCNN_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

#define WAVE 64

// Deterministic exp: the SAME sequence of IEEE binary32 operations as orc_expf()
// (oracle/frcnn_oracle.c).  Compiled with -ffp-contract=off: no FMA is formed.
__device__ __forceinline__ float det_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283935546875f) return __builtin_inff();
    if (x < -103.97283935546875f) return 0.0f;
    const float LOG2E = 1.44269502162933349609375f;
    const float LN2_HI = 0.693145751953125f;
    const float LN2_LO = 1.42860676533018704503775e-06f;
    float t = x * LOG2E;
    float kf = __builtin_floorf(t + 0.5f);
    int k = (int)kf;
    float r = x - kf * LN2_HI;
    r = r - kf * LN2_LO;
    float p = 1.0f / 720.0f;
    p = p * r + 1.0f / 120.0f;
    p = p * r + 1.0f / 24.0f;
    p = p * r + 1.0f / 6.0f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    int k1 = k >> 1;
    int k2 = k - k1;
    float s1 = __uint_as_float((uint32_t)(k1 + 127) << 23);
    float s2 = __uint_as_float((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

// Deterministic log2: same sequence as orc_log2f().
__device__ __forceinline__ float det_log2f(float a)
{
    if (a != a || a < 0.0f) return __builtin_nanf("");
    if (a == 0.0f) return -__builtin_inff();
    if (a == __builtin_inff()) return __builtin_inff();
    uint32_t u = __float_as_uint(a);
    int e = 0;
    if ((u >> 23) == 0) { a = a * 16777216.0f; u = __float_as_uint(a); e = -24; }
    e += (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421353816986083984375f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float q = 1.0f / 9.0f;
    q = q * z + 1.0f / 7.0f;
    q = q * z + 1.0f / 5.0f;
    q = q * z + 1.0f / 3.0f;
    q = q * z + 1.0f;
    float ln_m = (2.0f * s) * q;
    return (float)e + ln_m * 1.44269502162933349609375f;
}

// torch.max / torch.min semantics (NaN propagates) -- utils/util.py:97-98
__device__ __forceinline__ float tmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float tmin(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

__device__ __forceinline__ float4 xy_to_cxcy4(float4 b)
{
    return make_float4((b.z + b.x) / 2.0f, (b.w + b.y) / 2.0f, b.z - b.x, b.w - b.y);
}
__device__ __forceinline__ float4 cxcy_to_xy4(float4 c)
{
    float hw = c.z / 2.0f, hh = c.w / 2.0f;
    return make_float4(c.x - hw, c.y - hh, c.x + hw, c.y + hh);
}
__device__ __forceinline__ float4 decode4(float4 t, float4 a)
{
    return make_float4(t.x * a.z + a.x, t.y * a.w + a.y, det_expf(t.z) * a.z, det_expf(t.w) * a.w);
}
__device__ __forceinline__ float4 encode4(float4 g, float4 a)
{
    return make_float4((g.x - a.x) / a.z, (g.y - a.y) / a.w, logf(g.z / a.z), logf(g.w / a.w));
}

// IoU of utils/util.py:66-102 (add_eps) / util/box_ops.py:24-37 (no eps)
template <bool ADD_EPS>
__device__ __forceinline__ float iou_pair(float4 p, float4 q, float eps)
{
    float lx = tmax(p.x, q.x), ly = tmax(p.y, q.y);
    float ux = tmin(p.z, q.z), uy = tmin(p.w, q.w);
    float w = ux - lx, h = uy - ly;
    if (w < 0.0f) w = 0.0f;
    if (h < 0.0f) h = 0.0f;
    float inter = w * h;
    float a1 = (p.z - p.x) * (p.w - p.y);
    float a2 = (q.z - q.x) * (q.w - q.y);
    float uni = a1 + a2 - inter;
    if (ADD_EPS) uni = uni + eps;
    return inter / uni;
}

// Philox4x32-10 (Salmon et al. 2011): counter-based RNG for on-device target sampling.
__device__ __forceinline__ uint32_t philox_first(uint64_t seed, uint64_t offset, uint32_t stream_id, uint32_t index)
{
    uint32_t c0 = index, c1 = stream_id, c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

#endif  // __HIPCC__
