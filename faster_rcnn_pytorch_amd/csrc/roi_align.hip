// roi_align.hip -- torchvision MultiScaleRoIAlign forward / backward as called at
// models/new_model.py:127,143 (gfx950): level mapper + per-level roi_align in ONE launch.
//
// The level of the RoI is recomputed in the kernel (sqrt + the deterministic log2) instead of a separate mapper
// launch + per-level gathers/scatters as torchvision does.
//
// 7x7 bins, sampling_ratio 2 (the only shape the reference uses, new_model.py:127):
//   forward : a workgroup owns (RoI, 32 channels); it stages the RoI's footprint rows of as many channels as fit in
//             25 KB of LDS with coalesced row loads, then a lane owns one bin, sets up its 4 samples x 4 taps ONCE
//             and walks the staged channels (same operation order as the oracle: bit-identical outputs).  Channel groups are pinned to XCDs (blockIdx & 7) so that each XCD's L2 only ever
//             holds its own eighth of the pyramid instead of all 91 MB streaming through all eight.
//   backward: tile-owner GATHER: bilinear scatter is separable, dF = Wy^T (fh x 7) . dOut (7x7) . Wx (7 x fw), so a workgroup that
//             owns a 16 x 8 pixel tile of one level walks the RoIs whose footprint meets the tile IN INDEX ORDER and accumulates
//             the tile in registers; every gradient pixel is written exactly once: no atomics, no memset, bit-reproducible.
//             Fine levels: roi_align_bwd_tile_kernel (32 channels per workgroup, one RoI at a time); coarse levels, where a tile's
//             RoI list is long: roi_align_bwd_coarse_kernel (16 channels, eight RoIs in flight, combined in a fixed order).
// Any other bin/sampling shape takes the generic one-lane-per-output kernels below (memset + fp32 atomics in backward;
// order-nondeterministic, tolerance 1e-4).
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include <cstdlib>
#include <cstring>

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

// ---- 7x7 / sampling_ratio 2 fast paths ---------------------------------------------------------------------------
struct Lin { int lo, hi; float wlo, whi; bool ok; };

__device__ __forceinline__ Lin lin_setup(int D, float v)       // one axis of bilin_setup
{
    Lin s;
    s.ok = !(v < -1.0f || v > (float)D);
    if (v <= 0.0f) v = 0.0f;
    int lo = (int)v, hi;
    if (lo >= D - 1) { hi = lo = D - 1; v = (float)lo; } else hi = lo + 1;
    const float l = v - (float)lo;
    s.lo = lo; s.hi = hi; s.wlo = 1.0f - l; s.whi = l;
    return s;
}

__device__ __forceinline__ void tile_of_block(int b, int R, int n_cg, int *cg, int *r)
{
    if ((n_cg & 7) == 0) { const int j = b >> 3; *cg = (j / R) * 8 + (b & 7); *r = j % R; }   // channel group pinned to an XCD
    else { *cg = b / R; *r = b % R; }
}

#ifndef RA_FWD_CG
#define RA_FWD_CG 32
#endif
#ifndef RA_FWD_WAVES
#define RA_FWD_WAVES 3
#endif
#ifndef RA_FWD_LDS
#define RA_FWD_LDS 8128
#endif
//                 // floats of footprint staging per workgroup (25 KB -> 6 workgroups / CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RA_FWD_WAVES, 8))) void roi_align_fwd77_kernel(MsLevels L, int C, const float4 *__restrict__ rois, int R, int aligned, int k_min,
                                                              float s0, int k0, float *__restrict__ out, int32_t *__restrict__ out_level, int n_cg)
{
    __shared__ float s_f[RA_FWD_LDS];
    int cg, r;
    tile_of_block(blockIdx.x, R, n_cg, &cg, &r);
    const int t = threadIdx.x, grp = t / 49, bin = t - grp * 49;
    const float4 b = rois[r];
    const int l = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
    if (out_level && cg == 0 && t == 0) out_level[r] = l;
    const int ph = bin / 7, pw = bin - ph * 7;
    const int H = L.H[l], W = L.W[l];
    const AlignGeom g = align_geom(b, L.scale[l], 7, 7, 2, aligned != 0);
    // footprint of the RoI on its level (sample coordinates are monotonic in the bin index)
    const Lin ya = lin_setup(H, g.sh + 0.5f * g.bh / 2.0f), yb = lin_setup(H, g.sh + 6.0f * g.bh + 1.5f * g.bh / 2.0f);
    const Lin xa = lin_setup(W, g.sw + 0.5f * g.bw / 2.0f), xb = lin_setup(W, g.sw + 6.0f * g.bw + 1.5f * g.bw / 2.0f);
    const int y0 = ya.lo, x0 = xa.lo, fh = yb.hi - y0 + 1, fw = xb.hi - x0 + 1;
    const int fwp = fw | 1, fp = fh * fwp;
    int yl[4], yh[4], xl[4], xh[4]; float w[4][4]; bool ok[4];
    bool inside = true;                                  // every tap of this lane lies inside the footprint box
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const float y = g.sh + (float)ph * g.bh + ((float)iy + 0.5f) * g.bh / 2.0f;
            const float x = g.sw + (float)pw * g.bw + ((float)ix + 0.5f) * g.bw / 2.0f;
            const Bilin s = bilin_setup(H, W, y, x);
            const int q = iy * 2 + ix;
            ok[q] = s.ok;
            yl[q] = s.yl; yh[q] = s.yh; xl[q] = s.xl; xh[q] = s.xh;
            w[q][0] = s.w1; w[q][1] = s.w2; w[q][2] = s.w3; w[q][3] = s.w4;
            inside = inside && s.yl >= y0 && s.yh < y0 + fh && s.xl >= x0 && s.xh < x0 + fw;
        }
    const int c0 = cg * RA_FWD_CG, c_end = min(C, c0 + RA_FWD_CG);
    const size_t plane = (size_t)H * W;
    int cb = (fh >= 1 && fw >= 1 && fp <= RA_FWD_LDS) ? min(RA_FWD_LDS / fp, RA_FWD_CG) : 0;
    if (__syncthreads_or(!inside && grp < 5)) cb = 0;     // (cannot happen for finite boxes; keeps the LDS path in-bounds regardless)
    int o[4][4];
    {
        const int stride = cb == 0 ? W : fwp, by = cb == 0 ? 0 : y0, bx = cb == 0 ? 0 : x0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q][0] = (yl[q] - by) * stride + xl[q] - bx; o[q][1] = (yl[q] - by) * stride + xh[q] - bx;
            o[q][2] = (yh[q] - by) * stride + xl[q] - bx; o[q][3] = (yh[q] - by) * stride + xh[q] - bx;
        }
    }
    if (cb == 0) {                                       // footprint larger than the staging buffer: gather straight from global
        if (grp >= 5) return;
        for (int c = c0 + grp; c < c_end; c += 5) {
            const float *pl = L.feat[l] + (size_t)c * plane;
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sv = w[q][0] * pl[o[q][0]] + w[q][1] * pl[o[q][1]] + w[q][2] * pl[o[q][2]] + w[q][3] * pl[o[q][3]];
                if (!ok[q]) sv = 0.0f;
                acc += sv;
            }
            out[((size_t)r * C + c) * 49 + bin] = acc * 0.25f;                  // cnt == 4: exact
        }
        return;
    }
    // staging: the pass's (channel, row, column) space is flattened so that consecutive lanes read consecutive columns of a
    // footprint row; 8 independent loads are in flight per lane before the first LDS store.  Divisions by the (uniform)
    // footprint width / height go through float reciprocals with a one-step correction.
    const unsigned ufw = (unsigned)fw, ufh = (unsigned)fh;
    const float rcp_fw = 1.0f / (float)fw, rcp_fh = 1.0f / (float)fh;
    const float *lvl = L.feat[l];
    for (int cbase = c0; cbase < c_end; cbase += cb) {
        const int n = min(cb, c_end - cbase);
        __syncthreads();
        const unsigned total = (unsigned)(n * fh * fw);
        const unsigned gbase = (unsigned)cbase * (unsigned)plane + (unsigned)(y0 * W + x0);
        for (unsigned e0 = 0; e0 < total; e0 += 256 * 8) {
            float v[8]; unsigned lo[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned e = e0 + u * 256 + t;
                unsigned row = (unsigned)((float)e * rcp_fw);             // row = c * fh + y
                unsigned x = e - row * ufw;
                if ((int)x < 0) { --row; x += ufw; } else if (x >= ufw) { ++row; x -= ufw; }
                unsigned c = (unsigned)((float)row * rcp_fh);
                unsigned y = row - c * ufh;
                if ((int)y < 0) { --c; y += ufh; } else if (y >= ufh) { ++c; y -= ufh; }
                lo[u] = e < total ? c * (unsigned)fp + y * (unsigned)fwp + x : 0xFFFFFFFFu;
                v[u] = e < total ? lvl[gbase + c * (unsigned)plane + y * (unsigned)W + x] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (lo[u] != 0xFFFFFFFFu) s_f[lo[u]] = v[u];
        }
        __syncthreads();
        if (grp < 5)
            for (int c = grp; c < n; c += 5) {
                const float *pl = s_f + c * fp;
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float sv = w[q][0] * pl[o[q][0]] + w[q][1] * pl[o[q][1]] + w[q][2] * pl[o[q][2]] + w[q][3] * pl[o[q][3]];
                    if (!ok[q]) sv = 0.0f;
                    acc += sv;
                }
                out[((size_t)r * C + cbase + c) * 49 + bin] = acc * 0.25f;                  // cnt == 4: exact
            }
    }
}

// ---- backward as a tile-owner gather: no atomics, no memset -------------------------------------------------------
// A workgroup owns a 16 x 8 pixel tile of ONE level for 32 channels (lane = (column, channel), 16 row accumulators in
// registers) and writes it exactly once.  (Measured 16x32x8ch 93/246 us, 16x16x16ch 74/191, 16x8x32ch 62/162, 16x4x64ch
// 89/178 -- micro-benchmark / FPN step: narrow tiles keep the per-tile RoI chain short.)  It scans the RoI list (level + footprint recomputed from the box: ~150 instructions per RoI,
// 2 RoIs per lane at R = 512), keeps the ones whose footprint meets the tile IN INDEX ORDER (ballot compaction, so the
// fp32 sum order is fixed: bit-reproducible gradients), and for each of them
//   A: stores the prefetched dOut[r][32 ch][7][7] to LDS and builds the two separable weight tables restricted to the tile,
//      Wy[16][7] and Wx[8][7] (lane = (row|col, bin): two 1-D bilinear set-ups each, gathered, no scatter / zero pass),
//   B: lane = (channel, column) forms its seven T[ph] = sum_pw dOut[c][ph][pw] * Wx[x][pw] in registers and adds
//      sum_ph Wy[y][ph] * T[ph] to its 16 row accumulators (registers) -- one barrier per RoI, LDS tables double-buffered.
#ifndef RT_TH
#define RT_TH 16
#endif
#ifndef RT_TW
#define RT_TW 8
#endif
#define RT_CPS (256 / RT_TW)              // channels per sub-group: lane = (column, channel)
#ifndef RT_NS
#define RT_NS 1                          // channel sub-groups of 8 per workgroup (lane = (column, channel within sub-group))
#endif
#define RT_CB (RT_CPS * RT_NS)
#define RT_LIST 256                      // RoIs are scanned in chunks of this many
#define RT_PF ((RT_CB * 49 + 255) / 256) // dOut elements prefetched per lane

struct TileLevels { int tile0[FRCNN_MAX_LEVELS + 1]; int tiles_x[FRCNN_MAX_LEVELS]; };
struct RoiEnt { int r; float sh, sw, bh, bw; };

template <typename TOUT> __device__ __forceinline__ void store_grad(TOUT *p, float v);
template <> __device__ __forceinline__ void store_grad<float>(float *p, float v) { *p = v; }

template <typename TOUT>
__global__ __launch_bounds__(256) void roi_align_bwd_tile_kernel(MsLevels L, TileLevels TL, int C, const float4 *__restrict__ rois, int R, int aligned,
                                                                 int k_min, float s0, int k0, const float *__restrict__ grad_out, int n_cg)
{
    __shared__ RoiEnt s_list[RT_LIST];
    __shared__ int s_n;
    __shared__ int s_woff[5];
    __shared__ float s_g[2][RT_CB * 49];
    __shared__ __attribute__((aligned(16))) float s_wy[2][RT_TH * 8];
    __shared__ float s_wx[2][RT_TW * 8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // block -> (tile, channel group); groups pinned to XCDs when there are 8 k of them
    int tile, cg;
    if ((n_cg & 7) == 0) { const int j = blockIdx.x >> 3; const int q8 = n_cg >> 3; cg = (j % q8) * 8 + (blockIdx.x & 7); tile = j / q8; }
    else { cg = blockIdx.x % n_cg; tile = blockIdx.x / n_cg; }
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < L.n_levels && tile >= TL.tile0[q]) ? 1 : 0;
    const int tl = tile - TL.tile0[l];
    const int ty0 = (tl / TL.tiles_x[l]) * RT_TH, tx0 = (tl % TL.tiles_x[l]) * RT_TW;
    const int H = L.H[l], W = L.W[l];
    const float scale = L.scale[l];
    const int cx = t % RT_TW, cc = t / RT_TW;             // phase B: lane -> (column cx, channel cc + RT_CPS s of the group)
    const int c0 = cg * RT_CB;
    const int nc = min(RT_CB, C - c0);
    const int ne = nc * 49;                               // valid dOut elements of one RoI for this channel group

    float acc[RT_NS][RT_TH];
#pragma unroll
    for (int sg = 0; sg < RT_NS; ++sg)
#pragma unroll
        for (int y = 0; y < RT_TH; ++y) acc[sg][y] = 0.0f;

    for (int rbase = 0; rbase < R; rbase += RT_LIST) {
        // ---- scan RT_LIST RoIs: which of them touch this tile?
        __syncthreads();                                   // the previous chunk's readers of s_list are done
        const int r = rbase + t;
        bool hit = false;
        RoiEnt e;
        e.r = r; e.sh = e.sw = e.bh = e.bw = 0.0f;
        if (r < R) {
            const float4 b = rois[r];
            const int lr = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
            if (lr == l) {
                const AlignGeom g = align_geom(b, scale, 7, 7, 2, aligned != 0);
                const Lin ya = lin_setup(H, g.sh + 0.5f * g.bh / 2.0f), yb = lin_setup(H, g.sh + 6.0f * g.bh + 1.5f * g.bh / 2.0f);
                const Lin xa = lin_setup(W, g.sw + 0.5f * g.bw / 2.0f), xb = lin_setup(W, g.sw + 6.0f * g.bw + 1.5f * g.bw / 2.0f);
                const int y0 = min(ya.lo, yb.lo), y1 = max(ya.hi, yb.hi), x0 = min(xa.lo, xb.lo), x1 = max(xa.hi, xb.hi);
                hit = y0 < ty0 + RT_TH && y1 >= ty0 && x0 < tx0 + RT_TW && x1 >= tx0;
                e.sh = g.sh; e.sw = g.sw; e.bh = g.bh; e.bw = g.bw;
            }
        }
        const unsigned long long bm = __ballot(hit);
        if (lane == 0) s_woff[wave + 1] = __builtin_popcountll(bm);
        __syncthreads();
        if (t == 0) { s_woff[0] = 0; for (int q = 1; q <= 4; ++q) s_woff[q] += s_woff[q - 1]; s_n = s_woff[4]; }
        __syncthreads();
        if (hit) s_list[s_woff[wave] + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = e;
        __syncthreads();
        const int n = s_n;
        // ---- the RoIs of this chunk that touch the tile, in index order
        float pg[RT_PF];                                   // prefetched dOut elements t + 256 u of the next RoI
#pragma unroll
        for (int u = 0; u < RT_PF; ++u) pg[u] = 0.0f;
        if (n > 0) {
            const float *src = grad_out + ((size_t)s_list[0].r * C + c0) * 49;
#pragma unroll
            for (int u = 0; u < RT_PF; ++u) if (t + 256 * u < ne) pg[u] = src[t + 256 * u];
        }
        for (int i = 0; i < n; ++i) {
            const int buf = i & 1;
            const RoiEnt en = s_list[i];
            // A: dOut tile -> LDS (zero-padded to RT_CB channels), weight tables of this RoI restricted to the tile
#pragma unroll
            for (int u = 0; u < RT_PF; ++u)
                if (t + 256 * u < RT_CB * 49) s_g[buf][t + 256 * u] = t + 256 * u < ne ? pg[u] : 0.0f;
            if (i + 1 < n) {
                const float *src = grad_out + ((size_t)s_list[i + 1].r * C + c0) * 49;
#pragma unroll
                for (int u = 0; u < RT_PF; ++u) if (t + 256 * u < ne) pg[u] = src[t + 256 * u];
            }
            {
                // lanes 0..223: (col = t / 7, bin = t % 7) of Wx ; lanes 0..111 additionally (row, bin) of Wy
                const int bin = t % 7, rc = t / 7;
                if (rc < RT_TW) {
                    const int x = tx0 + rc;
                    float wv = 0.0f;
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const Lin q = lin_setup(W, en.sw + (float)bin * en.bw + ((float)ix + 0.5f) * en.bw / 2.0f);
                        if (q.ok) wv += (q.lo == x ? q.wlo : 0.0f) + (q.hi == x ? q.whi : 0.0f);
                    }
                    s_wx[buf][rc * 8 + bin] = wv;
                }
                if (rc < RT_TH) {
                    const int y = ty0 + rc;
                    float wv = 0.0f;
#pragma unroll
                    for (int iy = 0; iy < 2; ++iy) {
                        const Lin q = lin_setup(H, en.sh + (float)bin * en.bh + ((float)iy + 0.5f) * en.bh / 2.0f);
                        if (q.ok) wv += (q.lo == y ? q.wlo : 0.0f) + (q.hi == y ? q.whi : 0.0f);
                    }
                    s_wy[buf][rc * 8 + bin] = 0.25f * wv;
                }
            }
            __syncthreads();
            // B: per channel sub-group: T[ph] for (cc + 8 sg, cx), then the tile rows the footprint reaches (uniform bounds)
            const Lin fa = lin_setup(H, en.sh + 0.5f * en.bh / 2.0f), fb = lin_setup(H, en.sh + 6.0f * en.bh + 1.5f * en.bh / 2.0f);
            const int ya = __builtin_amdgcn_readfirstlane(max(min(fa.lo, fb.lo) - ty0, 0));
            const int yb = __builtin_amdgcn_readfirstlane(min(max(fa.hi, fb.hi) - ty0, RT_TH - 1));
            float wx[7];
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) wx[pw] = s_wx[buf][cx * 8 + pw];
#pragma unroll
            for (int sg = 0; sg < RT_NS; ++sg) {
                if (RT_NS > 1) asm volatile("" ::: "memory");   // re-read the (broadcast) tables per sub-group instead of pinning 300 registers
                const float *gch = &s_g[buf][(cc + RT_CPS * sg) * 49];
                float T[7];
#pragma unroll
                for (int ph = 0; ph < 7; ++ph) {
                    float a = gch[ph * 7] * wx[0];
#pragma unroll
                    for (int pw = 1; pw < 7; ++pw) a = __builtin_fmaf(gch[ph * 7 + pw], wx[pw], a);
                    T[ph] = a;
                }
#pragma unroll
                for (int y = 0; y < RT_TH; ++y) {
                    if (y >= ya && y <= yb) {                 // scalar branch
                        const float4 w0 = *(const float4 *)&s_wy[buf][y * 8], w1 = *(const float4 *)&s_wy[buf][y * 8 + 4];
                        float a = acc[sg][y];
                        a = __builtin_fmaf(w0.x, T[0], a); a = __builtin_fmaf(w0.y, T[1], a); a = __builtin_fmaf(w0.z, T[2], a);
                        a = __builtin_fmaf(w0.w, T[3], a); a = __builtin_fmaf(w1.x, T[4], a); a = __builtin_fmaf(w1.y, T[5], a);
                        acc[sg][y] = __builtin_fmaf(w1.z, T[6], a);
                    }
                }
            }
            // no barrier here: the next A writes the OTHER buffers; the one after that is fenced by the next barrier
        }
    }
    // ---- the tile is complete: one coalesced store per row and channel
    if (tx0 + cx < W) {
#pragma unroll
        for (int sg = 0; sg < RT_NS; ++sg) {
            const int c = cc + RT_CPS * sg;
            if (c < nc) {
                TOUT *out = (TOUT *)L.grad[l] + ((size_t)(c0 + c) * H + ty0) * W + tx0 + cx;
#pragma unroll
                for (int y = 0; y < RT_TH; ++y)
                    if (ty0 + y < H) store_grad<TOUT>(out + (size_t)y * W, acc[sg][y]);
            }
        }
    }
}

// ---- the same gather for the COARSE levels: 8 channels per workgroup, eight RoIs in flight ------------------------
// (16 channels per workgroup, two sub-groups of 8 per lane.)
// The coarse levels have few tiles and every large RoI touches all of them: with the kernel above (ONE RoI in flight per
// workgroup, 32 channels) a coarse tile is a chain of ~200 RoIs x 0.5 us on an untrained frame -- 194 us in the FPN step against
// 61 us in the micro-benchmark.  Here a workgroup owns a 16 x 8 pixel tile of one level for 16 channels and writes it exactly once.  It scans the RoI list (level +
// footprint recomputed from the box, 2 RoIs per lane at R = 512), keeps the ones whose footprint meets the tile IN INDEX ORDER
// (ballot compaction), and its EIGHT WAVES then walk that list concurrently, wave w taking entries w, w + 8, ... with no workgroup
// barrier in between.  Per RoI a wave
//   A: stores the prefetched dOut[r][8 ch][7][7] to its private LDS tile and builds the two separable weight tables restricted to the
//      tile, Wy[16][7] and Wx[8][7] (lane = (row | col, bin): 1-D bilinear set-ups, gathered, no scatter / zero pass),
//   B: lane = (column, channel) forms its seven T[ph] = sum_pw dOut[c][ph][pw] * Wx[x][pw] in registers and adds
//      sum_ph Wy[y][ph] * T[ph] to its 16 row accumulators (registers).
// At the end the eight private accumulator tiles are added in wave order through LDS: every pixel is written once, the fp32 sum
// order is fixed (bit-reproducible gradients), nothing is cleared beforehand.
// (Used for every level it was 6x slower: 24 000 workgroups x 57 KB of LDS, each scanning the whole RoI list.  It only pays where
// the lists are long.)
#define RC_TH 16
#define RC_TW 8
#define RC_NS 2                          // channel sub-groups of 8 per lane
#define RC_CB (8 * RC_NS)                // channels per workgroup: lane = (column 0..7, channel 0..7 of each sub-group)
#define RC_WAVES 8                       // RoIs in flight per workgroup
#define RC_LIST 512                      // RoIs are scanned in chunks of this many (one per thread)
#define RC_GE (RC_CB * 49)               // dOut elements of one RoI for this channel group
#define RC_PF ((RC_GE + 63) / 64)        // ... prefetched per lane


template <typename TOUT>
__global__ __launch_bounds__(64 * RC_WAVES) void roi_align_bwd_coarse_kernel(MsLevels L, TileLevels TL, int C, const float4 *__restrict__ rois, int R, int aligned,
                                                                          int k_min, float s0, int k0, const float *__restrict__ grad_out, int n_cg)
{
    // the RoI list + the waves' dOut tiles (during the walk) and the waves' accumulator tiles (at the end) share one buffer
    constexpr int U_WALK = (int)sizeof(RoiEnt) / 4 * RC_LIST + RC_WAVES * (RC_GE + 8), U_ACC = (RC_WAVES - 1) * RC_CB * RC_TH * RC_TW;
    __shared__ __attribute__((aligned(16))) float s_u[U_WALK > U_ACC ? U_WALK : U_ACC];
    RoiEnt *s_list = (RoiEnt *)s_u;
    float (*s_g)[RC_GE + 8] = (float (*)[RC_GE + 8])(s_u + sizeof(RoiEnt) / 4 * RC_LIST);
    float (*s_acc)[RC_CB * RC_TH * RC_TW] = (float (*)[RC_CB * RC_TH * RC_TW])s_u;
    __shared__ int s_n;
    __shared__ int s_woff[RC_WAVES + 1];
    __shared__ __attribute__((aligned(16))) float s_wy[RC_WAVES][RC_TH * 8];
    __shared__ float s_wx[RC_WAVES][RC_TW * 8];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // block -> (tile, channel group); groups pinned to XCDs when there are 8 k of them
    int tile, cg;
    if ((n_cg & 7) == 0) { const int j = blockIdx.x >> 3; const int q8 = n_cg >> 3; cg = (j % q8) * 8 + (blockIdx.x & 7); tile = j / q8; }
    else { cg = blockIdx.x % n_cg; tile = blockIdx.x / n_cg; }
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < L.n_levels && tile >= TL.tile0[q]) ? 1 : 0;
    const int tl = tile - TL.tile0[l];
    const int ty0 = (tl / TL.tiles_x[l]) * RC_TH, tx0 = (tl % TL.tiles_x[l]) * RC_TW;
    const int H = L.H[l], W = L.W[l];
    const float scale = L.scale[l];
    const int cx = lane & 7, cc = lane >> 3;              // phase B: lane -> (column cx, channel cc of the group)
    const int c0 = cg * RC_CB;
    const int nc = min(RC_CB, C - c0);
    const int ne = nc * 49;                               // valid dOut elements of one RoI for this channel group
    float *my_g = s_g[wave], *my_wy = s_wy[wave], *my_wx = s_wx[wave];

    float acc[RC_NS][RC_TH];
#pragma unroll
    for (int sg = 0; sg < RC_NS; ++sg)
#pragma unroll
        for (int y = 0; y < RC_TH; ++y) acc[sg][y] = 0.0f;

    for (int rbase = 0; rbase < R; rbase += RC_LIST) {
        // ---- scan RC_LIST RoIs: which of them touch this tile?
        __syncthreads();                                   // the previous chunk's readers of s_list are done
        const int r = rbase + t;
        bool hit = false;
        RoiEnt e;
        e.r = r; e.sh = e.sw = e.bh = e.bw = 0.0f;
        if (r < R) {
            const float4 b = rois[r];
            const int lr = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
            if (lr == l) {
                const AlignGeom g = align_geom(b, scale, 7, 7, 2, aligned != 0);
                const Lin ya = lin_setup(H, g.sh + 0.5f * g.bh / 2.0f), yb = lin_setup(H, g.sh + 6.0f * g.bh + 1.5f * g.bh / 2.0f);
                const Lin xa = lin_setup(W, g.sw + 0.5f * g.bw / 2.0f), xb = lin_setup(W, g.sw + 6.0f * g.bw + 1.5f * g.bw / 2.0f);
                const int y0 = min(ya.lo, yb.lo), y1 = max(ya.hi, yb.hi), x0 = min(xa.lo, xb.lo), x1 = max(xa.hi, xb.hi);
                hit = y0 < ty0 + RC_TH && y1 >= ty0 && x0 < tx0 + RC_TW && x1 >= tx0;
                e.sh = g.sh; e.sw = g.sw; e.bh = g.bh; e.bw = g.bw;
            }
        }
        const unsigned long long bm = __ballot(hit);
        if (lane == 0) s_woff[wave + 1] = __builtin_popcountll(bm);
        __syncthreads();
        if (t == 0) { s_woff[0] = 0; for (int q = 1; q <= RC_WAVES; ++q) s_woff[q] += s_woff[q - 1]; s_n = s_woff[RC_WAVES]; }
        __syncthreads();
        if (hit) s_list[s_woff[wave] + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = e;
        __syncthreads();
        const int n = s_n;
        // ---- my share of the list: entries wave, wave + RC_WAVES, ... (index order inside the share)
        float pg[RC_PF];                                   // prefetched dOut elements lane + 64 u of my next RoI
#pragma unroll
        for (int u = 0; u < RC_PF; ++u) pg[u] = 0.0f;
        if (wave < n) {
            const float *src = grad_out + ((size_t)s_list[wave].r * C + c0) * 49;
#pragma unroll
            for (int u = 0; u < RC_PF; ++u) if (lane + 64 * u < ne) pg[u] = src[lane + 64 * u];
        }
        for (int i = wave; i < n; i += RC_WAVES) {
            const RoiEnt en = s_list[i];
            __builtin_amdgcn_wave_barrier();               // my previous RoI's LDS reads are issued (a wave's ds ops run in order)
            // A: dOut tile -> LDS (zero-padded to RC_CB channels), weight tables of this RoI restricted to the tile
#pragma unroll
            for (int u = 0; u < RC_PF; ++u)
                if (lane + 64 * u < RC_GE) my_g[lane + 64 * u] = lane + 64 * u < ne ? pg[u] : 0.0f;
            if (i + RC_WAVES < n) {
                const float *src = grad_out + ((size_t)s_list[i + RC_WAVES].r * C + c0) * 49;
#pragma unroll
                for (int u = 0; u < RC_PF; ++u) if (lane + 64 * u < ne) pg[u] = src[lane + 64 * u];
            }
            {
                // lanes 0..55: (col = lane / 7, bin = lane % 7) of Wx and rows 0..7 of Wy; then rows 8..15 of Wy
                const int bin = lane % 7, rc = lane / 7;
                if (rc < RC_TW) {
                    const int x = tx0 + rc;
                    float wv = 0.0f;
#pragma unroll
                    for (int ix = 0; ix < 2; ++ix) {
                        const Lin q = lin_setup(W, en.sw + (float)bin * en.bw + ((float)ix + 0.5f) * en.bw / 2.0f);
                        if (q.ok) wv += (q.lo == x ? q.wlo : 0.0f) + (q.hi == x ? q.whi : 0.0f);
                    }
                    my_wx[rc * 8 + bin] = wv;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int yy = rc + 8 * half, y = ty0 + yy;
                        float wy = 0.0f;
#pragma unroll
                        for (int iy = 0; iy < 2; ++iy) {
                            const Lin q = lin_setup(H, en.sh + (float)bin * en.bh + ((float)iy + 0.5f) * en.bh / 2.0f);
                            if (q.ok) wy += (q.lo == y ? q.wlo : 0.0f) + (q.hi == y ? q.whi : 0.0f);
                        }
                        my_wy[yy * 8 + bin] = 0.25f * wy;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // B: T[ph] for (cc, cx), then the tile rows the footprint reaches (uniform bounds)
            const Lin fa = lin_setup(H, en.sh + 0.5f * en.bh / 2.0f), fb = lin_setup(H, en.sh + 6.0f * en.bh + 1.5f * en.bh / 2.0f);
            const int ya = __builtin_amdgcn_readfirstlane(max(min(fa.lo, fb.lo) - ty0, 0));
            const int yb = __builtin_amdgcn_readfirstlane(min(max(fa.hi, fb.hi) - ty0, RC_TH - 1));
            float wx[7];
#pragma unroll
            for (int pw = 0; pw < 7; ++pw) wx[pw] = my_wx[cx * 8 + pw];
#pragma unroll
            for (int sg = 0; sg < RC_NS; ++sg) {
                const float *gch = &my_g[(cc + 8 * sg) * 49];
                float T[7];
#pragma unroll
                for (int ph = 0; ph < 7; ++ph) {
                    float a = gch[ph * 7] * wx[0];
#pragma unroll
                    for (int pw = 1; pw < 7; ++pw) a = __builtin_fmaf(gch[ph * 7 + pw], wx[pw], a);
                    T[ph] = a;
                }
#pragma unroll
                for (int y = 0; y < RC_TH; ++y) {
                    if (y >= ya && y <= yb) {             // scalar branch
                        const float4 w0 = *(const float4 *)&my_wy[y * 8], w1 = *(const float4 *)&my_wy[y * 8 + 4];
                        float a = acc[sg][y];
                        a = __builtin_fmaf(w0.x, T[0], a); a = __builtin_fmaf(w0.y, T[1], a); a = __builtin_fmaf(w0.z, T[2], a);
                        a = __builtin_fmaf(w0.w, T[3], a); a = __builtin_fmaf(w1.x, T[4], a); a = __builtin_fmaf(w1.y, T[5], a);
                        acc[sg][y] = __builtin_fmaf(w1.z, T[6], a);
                    }
                }
            }
        }
    }
    // ---- add the eight private tiles in wave order, then one coalesced store per row and channel
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int sg = 0; sg < RC_NS; ++sg)
#pragma unroll
            for (int y = 0; y < RC_TH; ++y) s_acc[wave - 1][((cc + 8 * sg) * RC_TH + y) * RC_TW + cx] = acc[sg][y];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int sg = 0; sg < RC_NS; ++sg) {
        const int c = cc + 8 * sg;
#pragma unroll
        for (int w = 0; w < RC_WAVES - 1; ++w)
#pragma unroll
            for (int y = 0; y < RC_TH; ++y) acc[sg][y] += s_acc[w][(c * RC_TH + y) * RC_TW + cx];
        if (tx0 + cx < W && c < nc) {
            TOUT *out = (TOUT *)L.grad[l] + ((size_t)(c0 + c) * H + ty0) * W + tx0 + cx;
#pragma unroll
            for (int y = 0; y < RC_TH; ++y)
                if (ty0 + y < H) store_grad<TOUT>(out + (size_t)y * W, acc[sg][y]);
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
    bool small_planes = true;                          // the staged forward indexes a level with 32-bit element offsets
    for (int l = 0; l < n_levels; ++l) small_planes = small_planes && (int64_t)C * H[l] * W[l] < ((int64_t)1 << 30);
    if (PH == 7 && PW == 7 && sampling_ratio == 2 && R < (1 << 24) && small_planes) {
        const int n_cg = (C + RA_FWD_CG - 1) / RA_FWD_CG;
        FRCNN_LAUNCH(KID_ROI_ALIGN_FWD, roi_align_fwd77_kernel, dim3((unsigned)(n_cg * R)), dim3(256), 0, s, L, C, (const float4 *)rois, (int)R,
                     aligned, k_min, s0, k0, out, out_level, n_cg);
        FRCNN_CHECK_LAUNCH("roi_align_fwd77_kernel");
        return FRCNN_OK;
    }
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
    FRCNN_REQUIRE((R == 0 || (rois && grad_out)), "ms_roi_align_bwd: NULL pointer");
    MsLevels L;
    int rc = fill_levels(&L, nullptr, grad_feats, H, W, scales, n_levels);
    if (rc) return rc;
    const int64_t total = R * C * PH * PW;
    hipStream_t s = (hipStream_t)stream;
    if (PH == 7 && PW == 7 && sampling_ratio == 2 && R < (1 << 24)) {
        // fine levels: the 32-channel kernel; coarse levels (fewer than a sixth of the finest level's tiles): the 8-channel kernel
        int real[FRCNN_MAX_LEVELS];
        for (int l = 0; l < FRCNN_MAX_LEVELS; ++l)
            real[l] = l < n_levels ? ((L.W[l] + RT_TW - 1) / RT_TW) * ((L.H[l] + RT_TH - 1) / RT_TH) : 0;
        int first_coarse = n_levels;
        for (int l = n_levels - 1; l >= 1; --l)
            if (real[l] * 6 <= real[0]) first_coarse = l; else break;
        auto fill = [&](TileLevels &T, int lo, int hi) {             // tiles of the levels [lo, hi); the others get none
            int tiles = 0;
            for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
                T.tile0[l] = tiles;
                T.tiles_x[l] = (L.W[l] + RT_TW - 1) / RT_TW;
                if (l >= lo && l < hi) tiles += real[l];
            }
            T.tile0[FRCNN_MAX_LEVELS] = tiles;
            return tiles;
        };
        static_assert(RT_TH == RC_TH && RT_TW == RC_TW, "both tile kernels use the same tile shape");
        TileLevels TF, TC;
        const int tiles_f = fill(TF, 0, first_coarse), tiles_c = fill(TC, first_coarse, n_levels);
        const int n_quads = (C + RT_CB - 1) / RT_CB, n_oct = (C + RC_CB - 1) / RC_CB;
        FRCNN_REQUIRE((int64_t)tiles_f * n_quads < ((int64_t)1 << 31) && (int64_t)tiles_c * n_oct < ((int64_t)1 << 31), "ms_roi_align_bwd: grid too large");
        if (tiles_f > 0)
            FRCNN_LAUNCH(KID_ROI_ALIGN_BWD, (roi_align_bwd_tile_kernel<float>), dim3((unsigned)(tiles_f * n_quads)), dim3(256), 0, s, L, TF, C,
                         (const float4 *)rois, (int)R, aligned, k_min, s0, k0, grad_out, n_quads);
        if (tiles_c > 0)
            FRCNN_LAUNCH(KID_ROI_ALIGN_BWD, (roi_align_bwd_coarse_kernel<float>), dim3((unsigned)(tiles_c * n_oct)), dim3(64 * RC_WAVES), 0, s, L, TC, C,
                         (const float4 *)rois, (int)R, aligned, k_min, s0, k0, grad_out, n_oct);
        FRCNN_CHECK_LAUNCH("roi_align_bwd_tile_kernel");
        return FRCNN_OK;
    }
    for (int l = 0; l < n_levels; ++l)                  // the scatter kernels accumulate: clear the planes first
        if (hipMemsetAsync(L.grad[l], 0, (size_t)C * L.H[l] * L.W[l] * sizeof(float), s) != hipSuccess)
            return frcnn_set_error(FRCNN_ERR_LAUNCH, "ms_roi_align_bwd: memset failed");
    if (R == 0) return FRCNN_OK;
    FRCNN_LAUNCH(KID_ROI_ALIGN_BWD, roi_align_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, C, (const float4 *)rois,
                 total, PH, PW, sampling_ratio, aligned, k_min, s0, k0, grad_out);
    FRCNN_CHECK_LAUNCH("roi_align_bwd_kernel");
    return FRCNN_OK;
}
