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
#include <algorithm>
#include <cfloat>

#ifndef ROI_BWD_CB
#define ROI_BWD_CB 2
#endif

#ifndef RP_ABL
#define RP_ABL 0                 // developer ablations of the backward (bit 0: no LDS adds, bit 1: no global loads)
#endif
#ifndef RP_FABL
#define RP_FABL 0                // ... of the forward (bit 0: no stores, bit 1: no window scan, bit 2: no staging loads)
#endif

#ifdef RP_TRACE                        // developer build only (tools/dev/roipool_trace.py): per-wave phase stamps (100 MHz) in a device-global table
__device__ unsigned long long g_rp_trace[2][8192][8];
#define RP_T(kern, slot) do { if ((threadIdx.x & 63) == 0) g_rp_trace[kern][((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 8191][slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) void frcnn_rp_trace_read(void *dst) { hipDeviceSynchronize(); hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rp_trace), sizeof(g_rp_trace)); }
#else
#define RP_T(kern, slot) do {} while (0)
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
// grid (ceil(C/4), S); a workgroup stages FOUR adjacent channel planes in LDS, INTERLEAVED per pixel (float4 = the four channels of one
// pixel), builds the bin-boundary table of its RB = ceil(R / S) RoIs once, then walks its RB * 49 (RoI, bin) tasks, a task per lane and
// pass: one ds_read_b128 per window pixel serves all four channels.  For one RoI the 4*49 outputs of the workgroup are contiguous in memory
// (784 B): the stores stay coalesced.
// Round 5: S is chosen so that the grid is ~one workgroup of 1024 threads per CU (S = 2 at C = 512) instead of 896 short-lived workgroups
// that all staged, scanned and stored in lock step (ablations, HIP-event us: 20.0 whole, 15.8 without the stores, 13.2 without the scan,
// 17.6 without the staging loads): a long-lived workgroup's stores (19 MB, fire and forget) drain under its later passes, and a plane is
// staged 2 times instead of 7.  The window walk reads each pixel once (the two-pixel step re-read a clamped duplicate on odd widths).
// AT = the argmax element: int32_t (the torchvision-shaped ABI) or uint16_t (library-private, planes < 65535 pixels: 6.4 MB less
// to write and, in backward, to read at R = 128, C = 512; 0xFFFF = empty bin).
// ------------------------------------------------------------------------------------------------
#ifndef ROI_FWD_BS
#define ROI_FWD_BS 1024
#endif
#ifndef ROI_FWD_WGS
#define ROI_FWD_WGS 256                      // workgroups the launch aims at (one per CU)
#endif
#ifndef ROI_FWD_EXCLUSIVE
#define ROI_FWD_EXCLUSIVE 1
#endif
#define ROI_FWD_RB_MAX 256                   // RoIs per workgroup at most (bin table: 14 words each)
template <typename AT> __device__ __forceinline__ AT roi_arg_enc(int mi);
template <> __device__ __forceinline__ int32_t roi_arg_enc<int32_t>(int mi) { return mi; }
template <> __device__ __forceinline__ uint16_t roi_arg_enc<uint16_t>(int mi) { return (uint16_t)(mi < 0 ? 0xFFFF : mi); }
template <typename AT> __device__ __forceinline__ int roi_arg_dec(AT a);
template <> __device__ __forceinline__ int roi_arg_dec<int32_t>(int32_t a) { return a; }
template <> __device__ __forceinline__ int roi_arg_dec<uint16_t>(uint16_t a) { return a == 0xFFFF ? -1 : (int)a; }

#define ROI_FWD_TAKE(q, idx)                                   \
    do {                                                       \
        if ((q).x > m0) { m0 = (q).x; i0 = (idx); }            \
        if ((q).y > m1) { m1 = (q).y; i1 = (idx); }            \
        if ((q).z > m2) { m2 = (q).z; i2 = (idx); }            \
        if ((q).w > m3) { m3 = (q).w; i3 = (idx); }            \
    } while (0)

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
    const int c0 = blockIdx.x * CB;
    const int S = gridDim.y, r0 = blockIdx.y;                    // this workgroup's RoIs: r0, r0 + S, r0 + 2 S, ... (interleaved: the sampled RoIs come
    const int nr = (R - r0 + S - 1) / S;                         // sorted by kind, and a contiguous half can hold all the large ones)
    const int nch = min(CB, C - c0);
    const float *src = feat + (size_t)c0 * HW;
    RP_T(0, 0);
    {   // stage the planes interleaved per pixel: a thread owns pixels t, t + BS, ...: four coalesced loads (one per plane), ONE
        // ds_write_b128.  (The first version walked the flat (channel, pixel) index: a division by H * W and a 4-way bank-conflicted
        // 4-byte LDS store per element -- ~540 of the kernel's ~1000 instructions per thread.)
        for (int p0 = threadIdx.x; p0 < HW; p0 += ROI_FWD_BS * 2) {
            const int p1 = p0 + ROI_FWD_BS;
            float4 a, b;
#if RP_FABL & 4                                                              // developer ablation: no staging loads
            a = make_float4((float)p0, 1.0f, 2.0f, 3.0f); b = a; px[p0] = a; if (p1 < HW) px[p1] = b; continue;
#endif
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
        const RoiBins g = roi_bins(rois[r0 + rl * S], scale, PH, PW);
        if (k < PH) {
            const int hs = (int)floorf((float)k * g.bh) + g.sh, he = (int)ceilf((float)(k + 1) * g.bh) + g.sh;
            tab[t] = min(max(hs, 0), H) | (min(max(he, 0), H) << 16);
        } else {
            const int q = k - PH;
            const int ws = (int)floorf((float)q * g.bw) + g.sw, we = (int)ceilf((float)(q + 1) * g.bw) + g.sw;
            tab[t] = min(max(ws, 0), W) | (min(max(we, 0), W) << 16);
        }
    }
    __shared__ int s_next;                                       // the next chunk of 64 tasks nobody has taken yet
    if (threadIdx.x == 0) s_next = ROI_FWD_BS;
    RP_T(0, 1);
    __syncthreads();
    RP_T(0, 2);
    // task = (RoI, bin); a wave takes 64 consecutive tasks at a time (contiguous runs of outputs): its first chunk by position, the next ones from
    // the counter -- windows differ 4x in size between RoIs, and with a fixed assignment the slowest wave finished 40 % after the median one
    for (int t = threadIdx.x, first = 1; t < nr * BINS; first = 0) {
        (void)first;
        const int rl = t / BINS, p = t - rl * BINS;
        const int ph = p / PW, pw = p - ph * PW;
        const int th = tab[rl * TW + ph], tw = tab[rl * TW + PH + pw];
        const int hs = th & 0xFFFF, he = th >> 16, ws = tw & 0xFFFF, we = tw >> 16;
        const bool empty = (he <= hs) || (we <= ws);
        const float init = empty ? 0.0f : -FLT_MAX;
        float m0 = init, m1 = init, m2 = init, m3 = init;
        int i0 = -1, i1 = -1, i2 = -1, i3 = -1;
#if RP_FABL & 2                                                              // developer ablation: no window scan
        if (hs < he && ws < we) { const float4 a = px[hs * W + ws]; m0 = a.x; m1 = a.y; m2 = a.z; m3 = a.w; i0 = i1 = i2 = i3 = hs * W + ws; }
        for (int h = hs; h < hs; ++h) {
#else
        for (int h = hs; h < he; ++h) {
#endif
            const int rowoff = h * W;
            int w = ws;
            for (; w + 1 < we; w += 2) {                         // two pixels = two independent ds_read_b128 in flight
                const float4 a = px[rowoff + w], b = px[rowoff + w + 1];
                ROI_FWD_TAKE(a, rowoff + w);
                ROI_FWD_TAKE(b, rowoff + w + 1);
            }
            if (w < we) {
                const float4 a = px[rowoff + w];
                ROI_FWD_TAKE(a, rowoff + w);
            }
        }
        const size_t e = ((size_t)(r0 + rl * S) * C + c0) * BINS + p;
#if RP_FABL & 1                                                              // developer ablation: no stores (results kept live)
        if (m0 + m1 + m2 + m3 == 12345.678f && i0 + i1 + i2 + i3 == -77) out[e] = m0;
#else
        out[e] = m0; argmax[e] = roi_arg_enc<AT>(i0);
        if (nch > 1) { out[e + BINS] = m1; argmax[e + BINS] = roi_arg_enc<AT>(i1); }
        if (nch > 2) { out[e + 2 * BINS] = m2; argmax[e + 2 * BINS] = roi_arg_enc<AT>(i2); }
        if (nch > 3) { out[e + 3 * BINS] = m3; argmax[e + 3 * BINS] = roi_arg_enc<AT>(i3); }
#endif
#ifdef RP_TRACE
        if (first) RP_T(0, 3);
#endif
        int nx = 0;
        if ((threadIdx.x & 63) == 0) nx = atomicAdd(&s_next, 64);
        t = __builtin_amdgcn_readfirstlane(nx) + (threadIdx.x & 63);
    }
    RP_T(0, 4);
#ifdef RP_TRACE
    __builtin_amdgcn_s_waitcnt(0);                               // vmcnt(0) expcnt(0) lgkmcnt(0): the stores acknowledged
    RP_T(0, 5);
#endif
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


// ------------------------------------------------------------------------------------------------
// Backward with WAVE-PRIVATE gradient planes and NON-ATOMIC adds (round 5; the shape the reference trains: 7x7 bins, 512 x 37 x 62, R = 128).
//
// Measured on gfx950 (tools/dev/micro/lds_atomic_rate.hip): ds_add_f32 costs 1 + 3 cycles PER ACTIVE LANE (192 per full wave-instruction,
// whatever the addresses), a plain ds_read + v_add + ds_write 10; ds_add_rtn_u32 on byte counters (tried for the ranks below) ~55.  The
// shared-plane kernel above spends 14 of its 21 us in its 12 544 lane-adds per workgroup.  Here:
//   * one workgroup per CB adjacent channels (one per CU at C = 512, CB = 2), NW copies of the CB planes in LDS, NW x CB waves: a wave owns
//     ONE plane of one copy and a fixed subset of the RoIs (r = copy, copy + NW, ...).  A piece = its channel's bins of one RoI (lanes < bins).
//   * all loads of a batch of U RoIs are issued up front (2 x U per lane: ONE HBM round trip), the planes are zeroed under them.
//   * several bins of a RoI can have their maximum on the same pixel, but only bins whose windows overlap.  For a RoI of at least PH x PW
//     feature cells (bin sides >= 1, i.e. exactly 1 or >= 1 + 1/7) those are the eight neighbours of a bin, and a pixel lies in the windows
//     of at most a 2 x 2 block of bins: the RANK of an element = how many of its four lower neighbours (left, upper right, up, upper left)
//     hold the same pixel is then distinct inside every group of equal pixels (four lane shuffles, no memory).  Round j = the elements of rank j
//     add with a plain read-modify-write: no two of them share an address, and the wave's LDS operations execute in order.  Rounds beyond
//     the first run only where a wave has such an element.  Smaller RoIs (and every RoI when the caller passes no boxes: the int32 ABI) add
//     with ds_add_f32, 3 cycles per element.
//   * no two waves ever add to the same address and a wave adds RoI after RoI in program order, so the sum behind every pixel has ONE
//     order: the result is bit-reproducible run to run (the shared-plane form adds in arrival order: 1e-6 relative run-to-run noise in every
//     gradient upstream).  The NW copies are summed in wave order on the way out; each plane is written once, coalesced.
// ------------------------------------------------------------------------------------------------
#ifndef ROI_BWD_PU
#define ROI_BWD_PU 16
#endif
// One lane of the wide form holds two consecutive elements e = 2 * lane, 2 * lane + 1 of a RoI's run of 2 * bins elements (channel c0: e < bins,
// channel c0 + 1: e >= bins).  Geometry of such an element: plane offset, and which of its four lower neighbour bins exist.
struct RoiBwdElem { int plane; bool has_l, has_ur, has_up, has_ul; bool in; };
__device__ __forceinline__ RoiBwdElem roi_bwd_elem(int e, int bins, int PW, int HW, int nch)
{
    RoiBwdElem x;
    const int ch = e >= bins ? 1 : 0;
    const int p = e - ch * bins, ph = p / PW, pw = p - ph * PW;
    x.in = e < nch * bins;
    x.plane = ch * HW;
    x.has_l = pw > 0; x.has_ur = ph > 0 && pw < PW - 1; x.has_up = ph > 0; x.has_ul = ph > 0 && pw > 0;
    return x;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void roi_pool_bwd_priv_kernel(const float *__restrict__ grad_out, const uint16_t *__restrict__ argmax,
                                                                    const float4 *__restrict__ rois, float scale,
                                                                    int R, int C, int HW, int PH, int PW, float *__restrict__ grad_feat)
{
    // two adjacent channels per workgroup (C even: a RoI's run of 2 * bins elements starts on an 8-byte boundary), NW copies of the two planes;
    // wave = copy: it owns the RoIs r = copy, copy + NW, ...  and loads a RoI's run with ONE 8-byte and ONE 4-byte load per lane (two gradients,
    // two 16-bit argmaxes): the 4- and 2-byte loads of one element per lane took ~10 us to issue and land (the vector memory path works per
    // instruction, and these were 512 half-empty ones per CU).
    extern __shared__ __attribute__((aligned(16))) float planes[];           // [NW][2 * HW (padded to a multiple of 4)] + [NW][64] (a word per lane)
    constexpr int U = ROI_BWD_PU;                                             // RoIs per wave and batch
    constexpr int CB = 2;
    const int bins = PH * PW;
    const int nb = gridDim.x;
    const int bx = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // neighbours on one XCD
    const int c0 = bx * CB;
    const int nch = min(CB, C - c0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int stride = (CB * HW + 3) & ~3;
    float *mine = planes + (size_t)wave * stride;                            // this wave's copy of the two planes
    float *dummy = planes + (size_t)NW * stride + wave * 64 + lane;          // where a lane that is not part of a round stores
    const bool act = lane < bins;                                            // bins lanes x 2 elements = the run
    const int lb = min(lane, bins - 1);
    const RoiBwdElem x0 = roi_bwd_elem(2 * lb, bins, PW, HW, nch), x1 = roi_bwd_elem(2 * lb + 1, bins, PW, HW, nch);
    uint32_t araw[U];
    float2 g[U];
    bool zeroed = false;
    RP_T(1, 0);
    for (int rb = wave; rb < R; rb += NW * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {                                        // unconditional loads from clamped addresses: nothing between them
            const int r = min(rb + u * NW, R - 1);                           // needs a loaded value, so all 2 * U are in flight together
            const size_t idx = ((size_t)r * C + c0) * bins + 2 * lb;         // (nch = 1, the last odd channel: the second element is masked below,
#if RP_ABL & 2                                                               //  and the launcher keeps such a run inside the buffers)
            araw[u] = ((((unsigned)idx * 2654435761u) >> 20) % 2048u) * 0x10001u; g[u] = make_float2(1.0f, 1.0f);
#else
            araw[u] = *(const uint32_t *)(argmax + idx);
            g[u] = *(const float2 *)(grad_out + idx);
#endif
        }
        RP_T(1, 1);
        if (!zeroed) {                                                       // under the loads in flight
            float4 *z = (float4 *)mine;
            for (int i = lane; i < stride / 4; i += 64) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            zeroed = true;
        }
        RP_T(1, 2);
        unsigned bigbits = 0;                                                // RoI u of this batch spans at least PH x PW cells (wave-uniform)
        if (rois != nullptr) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 b = rois[min(rb + u * NW, R - 1)];
                const int rw = max((int)roundf(b.z * scale) - (int)roundf(b.x * scale) + 1, 1);     // as roi_bins()
                const int rh = max((int)roundf(b.w * scale) - (int)roundf(b.y * scale) + 1, 1);
                bigbits |= (unsigned)(rw >= PW && rh >= PH) << u;
            }
        }
#if !(RP_ABL & 1)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = rb + u * NW < R;
            const int a0r = (int)(araw[u] & 0xFFFFu), a1r = (int)(araw[u] >> 16);
            const bool v0 = act && live && x0.in && a0r != 0xFFFF, v1 = act && live && x1.in && a1r != 0xFFFF;
            const int p0 = x0.plane + (v0 ? a0r : 0), p1 = x1.plane + (v1 ? a1r : 0);
            if ((bigbits >> u) & 1u) {                                       // wave-uniform
                // keys of the lane's two elements in one word (an element that is not there equals nobody: 0x8000 + its own number), and the
                // words of the lanes 1, 3 and 4 below by DPP wave shifts: no LDS traffic (7 ds_bpermute per RoI made the LDS pipe the bound)
                const int k0 = v0 ? p0 : 0x8000 + 2 * lane, k1 = v1 ? p1 : 0x8001 + 2 * lane;
                const int kk = k0 | (k1 << 16);
                const int q1 = __builtin_amdgcn_update_dpp(-1, kk, 0x138, 0xf, 0xf, false);       // wave_shr:1 (lane 0 keeps -1)
                const int q2 = __builtin_amdgcn_update_dpp(-1, q1, 0x138, 0xf, 0xf, false);
                const int q3 = __builtin_amdgcn_update_dpp(-1, q2, 0x138, 0xf, 0xf, false);
                const int q4 = __builtin_amdgcn_update_dpp(-1, q3, 0x138, 0xf, 0xf, false);
                // element e lives in lane e >> 1, slot e & 1 (slot 0 = low half).  With d = PW - 1, PW, PW + 1 and PW = 7 (the only pooled
                // width this kernel is launched for): e0 - 1 = (lane - 1, slot 1); e0 - 6 = (lane - 3, 0); e0 - 7 = (lane - 4, 1);
                // e0 - 8 = (lane - 4, 0); e1 - 1 = e0; e1 - 6 = (lane - 3, 1); e1 - 7 = (lane - 3, 0); e1 - 8 = (lane - 4, 1)
                const int n0_l = (int)((unsigned)q1 >> 16), n0_ur = q3 & 0xFFFF, n0_up = (int)((unsigned)q4 >> 16), n0_ul = q4 & 0xFFFF;
                const int n1_ur = (int)((unsigned)q3 >> 16), n1_up = q3 & 0xFFFF, n1_ul = (int)((unsigned)q4 >> 16);
                const int r0 = v0 ? (int)(x0.has_l && n0_l == k0) + (int)(x0.has_ur && n0_ur == k0) + (int)(x0.has_up && n0_up == k0) + (int)(x0.has_ul && n0_ul == k0) : -1;
                const int r1 = v1 ? (int)(x1.has_l && k0 == k1) + (int)(x1.has_ur && n1_ur == k1) + (int)(x1.has_up && n1_up == k1) + (int)(x1.has_ul && n1_ul == k1) : -1;
                // round j: the elements of rank j add with a plain read-modify-write (no two of them share an address) while the other lanes
                // store to a word of their own: no branch, no wait between the rounds
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j > 0 && !__any(r0 == j || r1 == j)) break;          // wave-uniform: most RoIs have no second element on any pixel
                    const float w0 = mine[p0], w1 = mine[p1];
                    float *d0 = r0 == j ? &mine[p0] : dummy, *d1 = r1 == j ? &mine[p1] : dummy;
                    *d0 = w0 + g[u].x;
                    *d1 = w1 + g[u].y;
                }
            } else {
                if (v0) atomicAdd(&mine[p0], g[u].x);                        // ds_add_f32: 3 cycles per element
                if (v1) atomicAdd(&mine[p1], g[u].y);
            }
        }
#else
        {   // developer ablation: no adds; every loaded value stays live
            float sg = 0.0f; unsigned sa = bigbits;
#pragma unroll
            for (int i = 0; i < U; ++i) { sg += g[i].x + g[i].y; sa += araw[i]; }
            if (sg == 12345.678f && sa == 0x54321u) mine[lane] = sg;
        }
#endif
        RP_T(1, 3);
    }
    if (!zeroed) {
        float4 *z = (float4 *)mine;
        for (int i = lane; i < stride / 4; i += 64) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    RP_T(1, 4);
    __syncthreads();
    RP_T(1, 5);
    float *dst = grad_feat + (size_t)c0 * HW;
    for (int i = threadIdx.x; i < nch * HW; i += 64 * NW) {
        float s = planes[i];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += planes[(size_t)w * stride + i];
        dst[i] = s;
    }
    RP_T(1, 6);
}

template <typename AT>
static int roi_pool_fwd_launch(const float *feat, int C, int H, int W, const float *rois, int64_t R, int PH, int PW, float spatial_scale,
                               float *out, AT *argmax, hipStream_t s)
{
    const int ncg = (C + 3) / 4;
    int S = (ROI_FWD_WGS + ncg - 1) / ncg;                                   // RoI slices: ~one workgroup per CU ...
    S = (int)std::min<int64_t>(std::max<int64_t>(S, (R + ROI_FWD_RB_MAX - 1) / ROI_FWD_RB_MAX), R);   // ... of at most ROI_FWD_RB_MAX RoIs
    const int RB = (int)((R + S - 1) / S);
    S = (int)((R + RB - 1) / RB);
    size_t shmem = (size_t)4 * H * W * 4 + (size_t)RB * (7 + 7) * 4;
#if ROI_FWD_EXCLUSIVE
    // one workgroup per CU: two of these 1024-thread workgroups fit a CU, and the dispatcher does pair them up while other CUs stay empty (trace:
    // median workgroup done at 10.1 us, the last at 14.6); asking for more than half of the CU's LDS makes every workgroup take a CU of its own
    if ((int64_t)ncg * S <= 256 && shmem < 84 * 1024) {
        shmem = 84 * 1024;
        if (hipFuncSetAttribute((const void *)roi_pool_fwd_lds_kernel<7, 7, AT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
            return frcnn_set_error(FRCNN_ERR_LAUNCH, "roi_pool_fwd: cannot opt in to %zu bytes of LDS", shmem);
    }
#endif
    FRCNN_LAUNCH((roi_pool_fwd_lds_kernel<7, 7, AT>), dim3(ncg, S), dim3(ROI_FWD_BS), shmem, s, feat, C, H, W, (const float4 *)rois, (int)R, RB,
                 spatial_scale, out, argmax);
    FRCNN_CHECK_LAUNCH("roi_pool_fwd_lds_kernel");
    return FRCNN_OK;
}

template <int NW>
static int roi_pool_bwd_priv_launch(const float *grad_out, const uint16_t *argmax, const float *rois, float scale, int64_t R, int C, int64_t HW,
                                    int PH, int PW, float *grad_feat, hipStream_t s)
{
    const size_t shmem = (size_t)NW * (((2 * HW + 3) & ~(int64_t)3) + 64) * 4;
    auto kern = roi_pool_bwd_priv_kernel<NW>;
    if (shmem > 64 * 1024 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
        return frcnn_set_error(FRCNN_ERR_LAUNCH, "roi_pool_bwd: cannot opt in to %zu bytes of LDS", shmem);
    FRCNN_LAUNCH((roi_pool_bwd_priv_kernel<NW>), dim3(C / 2), dim3(64 * NW), shmem, s, grad_out, argmax, (const float4 *)rois, scale, (int)R, C,
                 (int)HW, PH, PW, grad_feat);
    FRCNN_CHECK_LAUNCH("roi_pool_bwd_priv_kernel");
    return FRCNN_OK;
}

// `rois` / `scale`: the boxes the forward pooled (NULL = not known)
template <typename AT>
static int roi_pool_bwd_launch(const float *grad_out, const AT *argmax, const float *rois, float scale, int64_t R, int C, int64_t HW, int PH, int PW,
                               float *grad_feat, hipStream_t s)
{
    const int bins = PH * PW;
    if constexpr (sizeof(AT) == 2) {
        // the 16-bit pair (7 x 7 bins): wave-private planes (bit-reproducible, non-atomic adds) wherever C is even, two planes stay below 32768
        // pixels (16-bit keys) and at least four copies of the two planes fit the CU's 160 KB of LDS; the shared-plane form (LDS atomics, arrival order) otherwise
        const int64_t copy_bytes = (((2 * HW + 3) & ~(int64_t)3) + 64) * 4;
        if (getenv("FRCNN_ROI_BWD_SHARED") == nullptr && (C & 1) == 0 && PW == 7 && PH == 7 && 2 * HW < 0x8000) {
            if (8 * copy_bytes <= 160 * 1024) return roi_pool_bwd_priv_launch<8>(grad_out, argmax, rois, scale, R, C, HW, PH, PW, grad_feat, s);
            if (4 * copy_bytes <= 160 * 1024) return roi_pool_bwd_priv_launch<4>(grad_out, argmax, rois, scale, R, C, HW, PH, PW, grad_feat, s);
        }
    }
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

FRCNN_EXPORT int frcnn_roi_pool_bwd_a16(const float *grad_out, const uint16_t *argmax16, const float *rois, float spatial_scale, int64_t R, int C,
                                        int H, int W, float *grad_feat, void *stream)
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
    return roi_pool_bwd_launch<uint16_t>(grad_out, argmax16, rois, spatial_scale, R, C, HW, 7, 7, grad_feat, s);
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
        return roi_pool_bwd_launch<int32_t>(grad_out, argmax, nullptr, 0.0f, R, C, HW, PH, PW, grad_feat, s);
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
