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
//             TWO launches: roi_align_bwd_lists_kernel builds the per-tile RoI lists and the (RoI, tile) weight records, and its workgroup
//             that finishes last turns the list lengths into work items (long lists cut into segments, longest first);
//             roi_align_bwd_tile_kernel works the items, and of a cut list's segments the workgroup that finishes last adds the
//             partial tiles in segment order (until round 5 the planning and the adding were launches of their own).
// Any other bin/sampling shape takes the generic one-lane-per-output kernels below (memset + fp32 atomics in backward;
// order-nondeterministic, tolerance 1e-4).
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(roi_align);
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
#ifndef RA_FWD_INFLIGHT
#define RA_FWD_INFLIGHT 8              // channels' loads in flight per lane and (row, column) slot
#endif
#ifndef RA_FWD_LDS
#define RA_FWD_LDS 8128
#endif
//                 // floats of footprint staging per workgroup (25 KB -> 6 workgroups / CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RA_FWD_WAVES, 8))) void roi_align_fwd77_kernel(MsLevels L, int C, const float4 *__restrict__ rois, int R, int aligned, int k_min,
                                                              float s0, int k0, float *__restrict__ out, int32_t *__restrict__ out_level, int n_cg, const int32_t *__restrict__ order)
{
    __shared__ float s_f[RA_FWD_LDS];
    int cg, r;
    tile_of_block(blockIdx.x, R, n_cg, &cg, &r);
    if (order) r = order[r];                            // dispatch slot -> RoI: largest footprint first (roi_scale_order_kernel); the output row is the RoI's own
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
    if (cb > 0) {                                        // equal passes, a multiple of the loads in flight where the buffer allows: 19 + 13 channels
        const int ncg = c_end - c0, npass = (ncg + cb - 1) / cb;     // cost 3 + 2 rounds of dependent loads per pixel, 16 + 16 cost 2 + 2
        int eq = (ncg + npass - 1) / npass;
        eq = (eq + RA_FWD_INFLIGHT - 1) / RA_FWD_INFLIGHT * RA_FWD_INFLIGHT;
        cb = min(cb, eq);
    }
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
    // staging: lane = (column xi = t % 32, row group rg = t / 32); a wave covers two footprint rows of up to 32 columns, the loops
    // run over row blocks, column blocks and channels with purely additive addressing, eight channels' loads in flight per lane.
    // (Round 1 flattened (channel, row, column) into one index and took it apart again per element with reciprocal multiplies and
    // corrections: ~25 instructions per staged float, 1800 per thread and RoI -- the instruction stream, not the fetch, was the
    // kernel's time.)
    const int xi = t & 31, rg = t >> 5;
    const float *lvl = L.feat[l];
    for (int cbase = c0; cbase < c_end; cbase += cb) {
        const int n = min(cb, c_end - cbase);
        __syncthreads();
        // 32-bit element offsets from the level's base (a level has < 2^31 elements: checked by the host)
        const unsigned off0 = (unsigned)cbase * (unsigned)plane + (unsigned)(y0 * W + x0), uplane = (unsigned)plane;
#ifdef RA_FWD_ROWCOL                   // round 2's mapping: lane = (column t % 32, row group t / 32): lanes with column >= fw idle
        for (int yb = rg; yb < fh; yb += 8)
            for (int xx = xi; xx < fw; xx += 32) {
#else                                  // lane = footprint pixel t, t + 256, ..: every lane has work whatever the footprint's width (one
                                       // float division per pixel and lane; rounds of dependent loads: ceil(fh fw / 256) instead of
                                       // ceil(fh / 8) ceil(fw / 32) per eight channels)
        (void)xi; (void)rg;
        const float inv_fw = 1.0f / (float)fw;
        for (int i = t; i < fh * fw; i += 256) {
            {
                int yb = (int)(((float)i + 0.5f) * inv_fw);
                int xx = i - yb * fw;
                if (xx < 0) { --yb; xx += fw; } else if (xx >= fw) { ++yb; xx -= fw; }
#endif
                unsigned off = off0 + (unsigned)(yb * W + xx);
                float *dst = s_f + yb * fwp + xx;
                for (int c = 0; c < n; c += RA_FWD_INFLIGHT, off += RA_FWD_INFLIGHT * uplane) {
                    float v[RA_FWD_INFLIGHT];
#pragma unroll
                    for (int u = 0; u < RA_FWD_INFLIGHT; ++u) v[u] = c + u < n ? lvl[off + (unsigned)u * uplane] : 0.0f;
#pragma unroll
                    for (int u = 0; u < RA_FWD_INFLIGHT; ++u) if (c + u < n) dst[(c + u) * fp] = v[u];
                }
            }
#ifndef RA_FWD_ROWCOL
        }
#endif
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
// Two launches (four until round 5: the plan and the combine had launches of their own):
//  roi_align_bwd_lists_kernel   one workgroup per 16 x 8 pixel tile of any level: scans the RoIs (level + footprint recomputed
//      from the box, ~150 instructions per RoI), keeps the ones whose footprint meets the tile IN INDEX ORDER (ballot
//      compaction) as a list in the workspace.  (Round 1 had every (tile, channel group) workgroup of the main kernel repeat
//      this scan: 8x the work, and four barriers before the first useful load.)  One more workgroup per RoI builds the weight-table
//      RECORDS of its (RoI, tile) pairs (RA_MAXT below).
//      The lists workgroup that finishes LAST (a ticket; list lengths written through) then plans (ra_plan_block): lists -> work items, longest first.
//      A list longer than RS_SPLIT entries is cut into segments.
//  roi_align_bwd_tile_kernel    item workgroup = (tile, segment, 32 channels), one RoI per step.  dOut[r][32 ch][7][7] and the pair's record
//      arrive by LDS-DMA one RoI ahead; then
//        B1: lane (column, channel pair, half of the bin rows): T[bin row] = sum over the bins that reach the column of dOut * Wx -> LDS,
//        B2: lane (column, four rows, two channel pairs): acc[row][pair] += sum over the bin rows that reach the four rows of Wy * T.
//      The tables are SPARSE along both axes: on its own pyramid level a RoI is 14..28 pixels wide, a bin 2..4, and a pixel hears from
//      at most two bins of the seven (the record carries first bin and count per column and per group of four rows; small RoIs on
//      the finest level hear from up to seven and simply loop longer).  Round 2 multiplied all 7 x 7: 161 FMAs per lane and RoI
//      where 2 x 7 + 3 x 8 packed ones do now -- same sums bit for bit, because the skipped terms are exact zeros.
//      An unsplit tile is written exactly once, straight to the gradient plane; a segment writes its partial tile to the workspace.
//      Fill workgroups in the same launch zero the pixels of tiles whose list is empty (whole rows: 256-byte stores).
//      The segment that takes the last ticket of a split tile adds its partial tiles in SEGMENT ORDER and writes the plane (the combine step).
// The fp32 sum order is a function of the RoI list alone: bit-reproducible gradients, no atomics, nothing cleared beforehand.
// Why segments: a workgroup advances one RoI per ~1.2 us (a chain of LDS round trips and two barriers -- not issue- or bandwidth-
// bound: halving the instructions per step or fetching two RoIs ahead did not shorten it, more resident workgroups did), so the
// gather takes as long as the longest list.  Which tiles are hot moves with training: on a fresh RPN the 147 tiles of the stride-8
// level meet 30 RoIs on average and up to 71; 15 SGD steps later the stride-16 level's 44 tiles meet 53 on average and up to 116
// (tools/dev/roi_stats.py).  Round 2's first answer, a second kernel with eight waves per tile for the COARSE levels, guessed the
// hot level from the pyramid shape, paid the scan 16 times per tile and took 67-75 us; together with the fine kernel 137 us per
// step; the dense round-2 form of this design 106, this one 61 (tools/dev/ra_bwd_time.py, ra_trace.py).
// (Tile shape sweep of the dense form, micro-benchmark / FPN step: 16x32x8ch 93/246 us, 16x16x16ch 74/191, 16x8x32ch 62/162, 16x4x64ch 89/178.)
#ifndef RT_TH
#define RT_TH 16
#endif
#ifndef RT_TW
#define RT_TW 8
#endif
#define RT_CPS (256 / RT_TW)              // channels per workgroup: lane = (column, channel)
#define RT_CB RT_CPS
#define RT_PF ((RT_CB * 49 + 255) / 256) // dOut elements prefetched per lane
static_assert(128 + RT_TH * 7 <= 256 && 64 + RT_TW * 7 <= 128, "weight-table lanes do not fit the workgroup");
#ifndef RT_ATTR
#define RT_ATTR __attribute__((amdgpu_waves_per_eu(5, 8)))   // at most 96 registers (the register path; left alone it takes 97 = four workgroups a CU); the DMA path needs 73: six, which is what the LDS holds
#endif
#define RT_GS 49                         // LDS stride of a channel's 49 dOut values
#define RT_TS 112                        // LDS stride of a channel PAIR's 7 x 8 column-reduced values: [bin row][column][2]
typedef float f32x2 __attribute__((ext_vector_type(2)));
static_assert(RT_TH == 16 && RT_TW == 8 && RT_CB == 32, "the pixel-lane phase maps 256 lanes to 16 x 8 pixels x 2 channel halves");
#ifndef RS_SPLIT
#define RS_SPLIT 32                      // a tile's list is cut into ceil(n / RS_SPLIT) segments ... (8 / 12 / 16 / 24 / 32 / 40: 83 / 77 / 74 / 64 / 61 / 64 us for the four launches, RoIs of a trained RPN; every segment pays ~3 us to start and 128 KB of partial tiles)
#endif
#ifndef RS_NSEG
#define RS_NSEG 32                       // ... at most this many
#endif
#ifndef RS_CHUNK
#define RS_CHUNK 64                      // list entries staged in LDS at a time
#endif
#ifndef RT_RING
#define RT_RING 2                        // dOut / record buffers: RoI i is worked on while RT_RING - 1 later ones are in flight (3: no faster per step, and its 31 KB of LDS hold five workgroups a CU where 23.6 KB hold six: 70 against 65 us for the four launches)
#endif

struct TileLevels { int tile0[FRCNN_MAX_LEVELS + 1]; int tiles_x[FRCNN_MAX_LEVELS]; };
struct __attribute__((aligned(16))) RoiEnt { int r; float sh, sw, bh, bw; int rows; int rec; int pad; };   // rows = first | last << 8 tile row the RoI reaches; rec = its record for this tile, -1: none

// The weight tables of one (RoI, tile) pair -- a 1 KB RECORD of 256 dwords:
//   [  0, 64)  Wx[column 8][bin 8]      sum of the bilinear weights the two x samples of the bin put on the column (bin 7 unused)
//   [ 64,192)  Wy[row 16][bin 8]        the same for the rows, times 1 / 4 (the sample count)
//   [192,200)  column c: first bin with a non-zero weight | number of bins up to the last one << 8   (the bins that reach a pixel are
//   [200]      the largest such number over the columns                                               CONSECUTIVE: sample positions grow with the bin)
//   [201,205)  row group g (rows 4 g .. 4 g + 3): first bin that reaches any of the four rows | count << 8
// They depend on the pair alone, and the tile kernel used to rebuild them in each of its C / 32 channel-group workgroups: ~65 of the ~170
// VALU instructions a wave issued per RoI step.  Now one workgroup per RoI, appended to the lists launch, builds every pair's record
// ONCE into a pool indexed (RoI, position of the tile in the RoI's footprint),
// and the tile kernel fetches a record with one global_load_lds_dwordx4 of one wave.  A RoI whose footprint spans more than RA_MAXT
// tiles has no records (rec = -1): its tables are built in the tile kernel by the same function (one wave, straight into LDS).
#define RA_MAXT 16
#define RA_REC 256
#define RA_REC_COL 192
#define RA_REC_NBX 200
#define RA_REC_RG 201

// pixel bounds of the samples of a RoI on its level (the lists kernel's hit test and the record index use the same numbers)
struct RaFoot { int x0, x1, y0, y1; };
__device__ __forceinline__ RaFoot ra_footprint(const AlignGeom &g, int H, int W)
{
    const Lin ya = lin_setup(H, g.sh + 0.5f * g.bh / 2.0f), yb = lin_setup(H, g.sh + 6.0f * g.bh + 1.5f * g.bh / 2.0f);
    const Lin xa = lin_setup(W, g.sw + 0.5f * g.bw / 2.0f), xb = lin_setup(W, g.sw + 6.0f * g.bw + 1.5f * g.bw / 2.0f);
    RaFoot f;
    f.y0 = min(ya.lo, yb.lo); f.y1 = max(ya.hi, yb.hi); f.x0 = min(xa.lo, xb.lo); f.x1 = max(xa.hi, xb.hi);
    return f;
}

// One whole wave builds the record of (RoI geometry, tile at (ty0, tx0)) into rec (global or LDS).
__device__ __forceinline__ void ra_tables_wave(int H, int W, int ty0, int tx0, float sh, float sw, float bh, float bw, float *rec)
{
    const int lane = threadIdx.x & 63, bin = lane % 7, rc = lane / 7;
    unsigned long long bm[3];
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {                 // 0: the 8 columns, 1: rows 0..7, 2: rows 8..15 -- (row | column, bin) = 56 lanes
        const int D = pass == 0 ? W : H;
        const int pix = pass == 0 ? tx0 + rc : ty0 + (pass - 1) * 8 + rc;
        const float s0 = pass == 0 ? sw : sh, bs = pass == 0 ? bw : bh;
        float wv = 0.0f;
        if (lane < 56) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Lin q = lin_setup(D, s0 + (float)bin * bs + ((float)i + 0.5f) * bs / 2.0f);
                if (q.ok) wv += (q.lo == pix ? q.wlo : 0.0f) + (q.hi == pix ? q.whi : 0.0f);
            }
            rec[pass == 0 ? rc * 8 + bin : 64 + ((pass - 1) * 8 + rc) * 8 + bin] = pass == 0 ? wv : 0.25f * wv;
        }
        bm[pass] = __ballot(wv != 0.0f);                   // bit (row | column) * 7 + bin
    }
    if (lane < 8) {
        const unsigned m = (unsigned)(bm[0] >> (lane * 7)) & 127u;
        const int b0 = m ? __builtin_ctz(m) : 0, sp = m ? 32 - __builtin_clz(m) - b0 : 0;
        ((unsigned *)rec)[RA_REC_COL + lane] = (unsigned)b0 | ((unsigned)sp << 8);
        const unsigned long long any1 = __ballot(sp > 1), any2 = __ballot(sp > 2), any3 = __ballot(sp > 3), any4 = __ballot(sp > 4),
                                 any5 = __ballot(sp > 5), any6 = __ballot(sp > 6), any0 = __ballot(sp > 0);
        if (lane == 0)
            ((unsigned *)rec)[RA_REC_NBX] = (any0 != 0) + (any1 != 0) + (any2 != 0) + (any3 != 0) + (any4 != 0) + (any5 != 0) + (any6 != 0);
    }
    if (lane >= 8 && lane < 12) {
        const int g = lane - 8;
        const unsigned m28 = (unsigned)(bm[1 + (g >> 1)] >> ((g & 1) * 28)) & 0xFFFFFFFu;
        const unsigned u = (m28 | (m28 >> 7) | (m28 >> 14) | (m28 >> 21)) & 127u;
        const int p0 = u ? __builtin_ctz(u) : 0, nb = u ? 32 - __builtin_clz(u) - p0 : 0;
        ((unsigned *)rec)[RA_REC_RG + g] = (unsigned)p0 | ((unsigned)nb << 8);
    }
}

template <typename TOUT> __device__ __forceinline__ void store_grad(TOUT *p, float v);
template <> __device__ __forceinline__ void store_grad<float>(float *p, float v) { *p = v; }

__device__ __forceinline__ int ra_nseg(int n, int split) { return n <= split ? 1 : min(RS_NSEG, (n + split - 1) / split); }

__device__ void ra_plan_block(int tiles, int cap_items, const int32_t *__restrict__ cnt, int32_t *__restrict__ tbase, int32_t *__restrict__ tnseg,
                              int4 *__restrict__ items, int32_t *__restrict__ slot);

// cnt[tile] = list length; ent[tile * cap + i] = the i-th RoI (index order) whose footprint meets the tile.  Blocks >= tiles (when pool
// is not NULL): block tiles + r builds the records of RoI r, one footprint tile per wave and round -- see RA_MAXT above.
__global__ __launch_bounds__(256) void roi_align_bwd_lists_kernel(MsLevels L, TileLevels TL, const float4 *__restrict__ rois, int R, int aligned,
                                                                  int k_min, float s0, int k0, int cap, int tiles, int32_t *__restrict__ cnt,
                                                                  RoiEnt *__restrict__ ent, float *__restrict__ pool, int n_cg, int32_t *__restrict__ tix,
                                                                  int *__restrict__ ticket, int cap_items, int32_t *__restrict__ tbase,
                                                                  int32_t *__restrict__ tnseg, int4 *__restrict__ items, int32_t *__restrict__ slot)
{
    __shared__ int s_woff[5];
    __shared__ int s_last;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if ((int)blockIdx.x >= tiles) {                        // block tiles + r: the records of RoI r, wave w those of footprint tiles w, w + 4, ...
        const int r = (int)blockIdx.x - tiles;
        const float4 b = rois[r];
        const int l = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
        const int H = L.H[l], W = L.W[l];
        const AlignGeom g = align_geom(b, L.scale[l], 7, 7, 2, aligned != 0);
        const RaFoot f = ra_footprint(g, H, W);
        const int txa = f.x0 / RT_TW, ntx = f.x1 / RT_TW - txa + 1, tya = f.y0 / RT_TH, nty = f.y1 / RT_TH - tya + 1;
        if (ntx * nty > RA_MAXT) return;
        for (int k = wave; k < ntx * nty; k += 4)
            ra_tables_wave(H, W, (tya + k / ntx) * RT_TH, (txa + k % ntx) * RT_TW, g.sh, g.sw, g.bh, g.bw, pool + ((size_t)r * RA_MAXT + k) * RA_REC);
        return;
    }
    const int tile = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < L.n_levels && tile >= TL.tile0[q]) ? 1 : 0;
    const int tl = tile - TL.tile0[l];
    const int ty0 = (tl / TL.tiles_x[l]) * RT_TH, tx0 = (tl % TL.tiles_x[l]) * RT_TW;
    const int H = L.H[l], W = L.W[l];
    const float scale = L.scale[l];
    RoiEnt *mine = ent + (size_t)tile * cap;
    int base = 0;
    for (int rbase = 0; rbase < R; rbase += 256) {
        const int r = rbase + t;
        bool hit = false;
        RoiEnt e;
        e.r = r; e.sh = e.sw = e.bh = e.bw = 0.0f; e.rows = 0; e.rec = -1; e.pad = 0;
        if (r < R) {
            const float4 b = rois[r];
            const int lr = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
            if (lr == l) {
                const AlignGeom g = align_geom(b, scale, 7, 7, 2, aligned != 0);
                const RaFoot f = ra_footprint(g, H, W);
                hit = f.y0 < ty0 + RT_TH && f.y1 >= ty0 && f.x0 < tx0 + RT_TW && f.x1 >= tx0;
                e.sh = g.sh; e.sw = g.sw; e.bh = g.bh; e.bw = g.bw;
                e.rows = max(f.y0 - ty0, 0) | (min(f.y1 - ty0, RT_TH - 1) << 8);
                const int txa = f.x0 / RT_TW, ntx = f.x1 / RT_TW - txa + 1, tya = f.y0 / RT_TH, nty = f.y1 / RT_TH - tya + 1;
                if (pool && ntx * nty <= RA_MAXT) e.rec = r * RA_MAXT + (ty0 / RT_TH - tya) * ntx + (tx0 / RT_TW - txa);
            }
        }
        const unsigned long long bm = __ballot(hit);
        __syncthreads();                                   // the previous chunk's readers of s_woff are done
        if (lane == 0) s_woff[wave + 1] = __builtin_popcountll(bm);
        __syncthreads();
        if (t == 0) { s_woff[0] = 0; for (int q = 1; q <= 4; ++q) s_woff[q] += s_woff[q - 1]; }
        __syncthreads();
        if (hit) mine[base + s_woff[wave] + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = e;
        base += s_woff[4];
    }
    if (t < n_cg) tix[tile * n_cg + t] = 0;                // the tile kernel's last-segment tickets of this tile (consumed by the NEXT launch)
    if (t == 0) {
        __hip_atomic_store(&cnt[tile], base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // written through: the planning workgroup reads it in THIS launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         // ... and acknowledged before the ticket announces it
        s_last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tiles - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    // the last lists workgroup: every list length is in memory; leave the ticket zero for the next launch and plan
    if (t == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ra_plan_block(tiles, cap_items, cnt, tbase, tnseg, items, slot);
}

// The plan (ra_plan_block below): turns the list lengths into work items.  A non-empty tile q gets nseg(q) consecutive item NUMBERS from tbase[q] (an
// exclusive prefix sum in tile order, so the cut of a list into segments -- and with it the summation order -- is a function of the
// lists alone); an empty tile gets none (its pixels are zero-filled by the fill workgroups of the main launch).  If the items do not
// fit the table the split threshold is doubled until they do (cap >= tiles).  The item RECORDS (tile, first entry, end entry,
// seg | nseg << 8 -- one 16-byte load and a workgroup of the main kernel can start) are laid out LONGEST SEGMENT FIRST (a counting
// sort by length; ties in arrival order, which moves records but no sums): the main kernel's workgroups are dispatched in record order
// and each advances one RoI per step, so this is longest-processing-time-first scheduling over the CUs (worth 0-10 us, most on a
// fresh RPN whose long lists sit on the fine level, at the END of the tile order).  slot[item number] = record position, which
// is also where a segment's partial tile goes (the combine kernel looks it up); unused records carry tile = -1.
__device__ __forceinline__ int ra_items(int n, int split) { return n == 0 ? 0 : ra_nseg(n, split); }
__device__ __forceinline__ int ra_cnt(const int32_t *cnt, int q) { return __hip_atomic_load(&cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // written by other workgroups of this launch
// (Round 5: no launch of its own any more -- the lists workgroup that finishes LAST runs it, 256 threads, reading the list lengths the others wrote
// through with agent-scope loads.)
__device__ void ra_plan_block(int tiles, int cap_items, const int32_t *__restrict__ cnt, int32_t *__restrict__ tbase, int32_t *__restrict__ tnseg,
                              int4 *__restrict__ items, int32_t *__restrict__ slot)
{
    // block-wide exclusive scan of the per-thread item counts: wave scans by shuffles, the wave totals through LDS (two barriers)
    __shared__ int s_wsum[4];
    __shared__ int s_total;
    __shared__ int s_hist[64];                             // bucket b = 63 - min(length, 63)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (tiles + 255) / 256, q0 = t * per, q1 = min(q0 + per, tiles);
    int split = RS_SPLIT;
    int mine = 0, incl = 0;
    // my tiles' list lengths, fetched ONCE and together (each is a trip to the L2 -- the lengths were written through by other workgroups of this launch --
    // and the three passes below used to make it again per tile, one after the other: 13 us for the planning tail where the launch of its own took 6)
    constexpr int PC = 4;
    int cn[PC];
#pragma unroll
    for (int k = 0; k < PC; ++k) cn[k] = q0 + k < q1 ? ra_cnt(cnt, q0 + k) : 0;
    auto each_tile = [&](auto &&f) {
#pragma unroll
        for (int k = 0; k < PC; ++k)
            if (q0 + k < q1) f(q0 + k, cn[k]);
        for (int q = q0 + PC; q < q1; ++q) f(q, ra_cnt(cnt, q));
    };
    // segment sgm of ns over a list of n: entries [n sgm / ns, n (sgm + 1) / ns)  (n <= R < 2^27 -- the launcher checks -- and sgm + 1 <= ns <= RS_NSEG = 32: the products fit 32 bits)
    auto cut = [](int n, int sgm, int ns) { return (int)((unsigned)n * (unsigned)sgm / (unsigned)ns); };
    if (t < 64) s_hist[t] = 0;
    for (;;) {
        mine = 0;
        each_tile([&](int, int n) { mine += ra_items(n, split); });
        incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        __syncthreads();                                   // the previous round's readers of s_wsum / s_total are done
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        if (t == 0) { int a = 0; for (int w = 0; w < 4; ++w) { const int v = s_wsum[w]; s_wsum[w] = a; a += v; } s_total = a; }
        __syncthreads();
        if (s_total <= cap_items) break;
        split *= 2;                                        // (terminates: at split >= max(cnt) every tile has at most one item and tiles <= cap)
    }
    const int base0 = s_wsum[wave] + incl - mine;          // exclusive prefix of my chunk
    const int total = s_total;
    each_tile([&](int, int n) {
        const int ns = ra_items(n, split);
        for (int sgm = 0; sgm < ns; ++sgm) atomicAdd(&s_hist[63 - min(cut(n, sgm + 1, ns) - cut(n, sgm, ns), 63)], 1);
    });
    __syncthreads();
    if (wave == 0) {                                       // exclusive scan of the 64 buckets by one wave
        const int v = s_hist[lane];
        int in = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(in, o); if (lane >= o) in += u; }
        s_hist[lane] = in - v;
    }
    __syncthreads();
    int base = base0;
    each_tile([&](int q, int n) {
        const int ns = ra_items(n, split);
        tbase[q] = base; tnseg[q] = ns;
        for (int sgm = 0; sgm < ns; ++sgm) {
            const int lo = cut(n, sgm, ns), hi = cut(n, sgm + 1, ns);
            const int k = atomicAdd(&s_hist[63 - min(hi - lo, 63)], 1);
            items[k] = make_int4(q, lo, hi, sgm | (ns << 8));
            slot[base + sgm] = k;
        }
        base += ns;
    });
    for (int i = total + t; i < cap_items; i += 256) items[i] = make_int4(-1, 0, 0, 0);
}

// The ticket by which the last lists workgroup is found: a word of the LIBRARY (zero when the code object is loaded, left zero by every launch), picked by
// the workspace's address -- the workspace itself stays plain scratch whose contents on entry do not matter (frcnn_hip.h).  Two calls that run
// CONCURRENTLY on workspaces 64 slots apart would share a word; the path's contract is one call at a time per device (the training thread's stream).
__device__ int g_ra_plan_ticket[64];

#define RF_CH 4                          // channels per fill workgroup
struct FillLevels { int fill0[FRCNN_MAX_LEVELS + 1]; };   // first fill workgroup of level l: (row blocks of l) x (C / RF_CH) each

// 16 bytes per lane, global -> LDS at lds + lane * 16, as INLINE ASSEMBLY: behind the builtin the compiler tracks the transfer and puts
// s_waitcnt vmcnt(0) in front of the next LDS read that might alias it (every read of the other ring buffers) and into __syncthreads() --
// each RoI step then paid the full L2 round trip (1 us a step with no arithmetic at all).  The waits are counted by hand (ra_vm_wait).
__device__ __forceinline__ void ra_dma16(const float *g, const float *lds)
{
    const unsigned l = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(g) : "memory");   // (m0 is a reserved register the compiler sets right in front of the few instructions that read it; no other one here does)
}
template <int N> __device__ __forceinline__ void ra_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ra_barrier()               // workgroup barrier that leaves the LDS-DMA in flight
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#ifdef RT_TRACE      // developer build (tools/dev/ra_trace.py): per workgroup {start, first barrier, end, steps, XCC/CU id, kind} in 10 ns ticks
__device__ unsigned long long g_rt_trace[32768][6];
extern "C" __attribute__((visibility("default"))) void frcnn_ra_trace_read(void *dst) { hipDeviceSynchronize(); hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rt_trace), sizeof(g_rt_trace)); }
#define RT_T(slot) do { if (threadIdx.x == 0 && blockIdx.x < 32768) g_rt_trace[blockIdx.x][slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RT_TV(slot, v) do { if (threadIdx.x == 0 && blockIdx.x < 32768) g_rt_trace[blockIdx.x][slot] = (unsigned long long)(v); } while (0)
#else
#define RT_T(slot) do { } while (0)
#define RT_TV(slot, v) do { } while (0)
#endif
template <typename TOUT, bool DMA>
__global__ __launch_bounds__(256) RT_ATTR void roi_align_bwd_tile_kernel(MsLevels L, TileLevels TL, FillLevels FL, int C, int aligned,
                                                                         const float *__restrict__ grad_out, int n_cg, int cap, int n_item_blocks, int n_first, int n_fill,
                                                                         const int32_t *__restrict__ cnt, const RoiEnt *__restrict__ ent,
                                                                         const int4 *__restrict__ items, const float *__restrict__ pool, float *__restrict__ part,
                                                                         const int32_t *__restrict__ tbase, const int32_t *__restrict__ slot, int32_t *__restrict__ tix)
{
    __shared__ __attribute__((aligned(16))) float s_g[RT_RING][RT_CB * RT_GS];   // ring of three: dOut of the current RoI and of the next two (in flight)
    __shared__ __attribute__((aligned(16))) float s_T[(RT_CB / 2) * RT_TS];   // column-reduced dOut of the current RoI: [channel pair][bin row][column][2]
    __shared__ __attribute__((aligned(16))) float s_rec[RT_RING][RA_REC];   // likewise their weight-table records (layout: RA_REC_* above)
    __shared__ RoiEnt s_list[RS_CHUNK];
    const int t = threadIdx.x;
    // grid order: the first n_first item blocks (one full round of resident workgroups: the longest segments), the fill blocks, the other item blocks
    const int bid = (int)blockIdx.x;
    const bool is_fill = bid >= n_first && bid < n_first + n_fill;
    RT_T(0); RT_TV(2, 0); RT_TV(5, is_fill ? 2 : 1);
    if (is_fill) {
        // ---- fill workgroup: 16 rows x the whole width x RF_CH channels of one level; zero where the owning tile's list is EMPTY
        // (nobody else writes those pixels).  A wave covers 64 consecutive pixels: 256-byte stores, where a tile's own zero-fill
        // would write 32-byte pieces (the empty tiles' stores were 40 of the kernel's 80 us).  These blocks come LAST in the grid:
        // 68 MB of zeros at the FPN shape take 11 us at ~6 TB/s after the item workgroups have drained.  Measured and not better
        // (tools/dev/ra_trace.py): fill blocks first (they hold every slot for 5-10 us and the items start that much later), every
        // block storing a fill unit before or after its own item (gfx9 counts stores in vmcnt: the first s_waitcnt vmcnt(0) of the
        // RoI loop waits for them; at the end they keep the slot from the next item).
        const int f = bid - n_first;
        int l = 0;
#pragma unroll
        for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < L.n_levels && f >= FL.fill0[q]) ? 1 : 0;
        const int ncc = (C + RF_CH - 1) / RF_CH;
        const int fl = f - FL.fill0[l], rbk = fl / ncc, c0 = (fl % ncc) * RF_CH;
        const int H = L.H[l], W = L.W[l];
        const int y0 = rbk * RT_TH, ny = min(RT_TH, H - y0), nch = min(RF_CH, C - c0);
        for (int x = t; x < W; x += 256) {
            if (cnt[TL.tile0[l] + rbk * TL.tiles_x[l] + x / RT_TW] != 0) continue;
            for (int c = 0; c < nch; ++c) {
                TOUT *out = (TOUT *)L.grad[l] + ((size_t)(c0 + c) * H + y0) * W + x;
                for (int y = 0; y < ny; ++y) store_grad<TOUT>(out + (size_t)y * W, 0.0f);
            }
        }
        RT_T(2);
        return;
    }
    // block -> (work item, channel group)  (n_cg = 8: blockIdx % 8 = channel group = XCD, so each L2 holds one eighth of dOut).
    // Records are sorted longest segment first (plan kernel) and workgroups are dispatched in that order.
    const int ib = bid < n_first ? bid : bid - n_fill;
    const int cg = ib % n_cg;
    const int item = ib / n_cg;
    const int4 rec = items[item];
    const int tile = rec.x;
    if (tile < 0) { RT_TV(5, 0); return; }
    const int lo = rec.y, hi = rec.z, nseg = rec.w >> 8;
    RT_TV(3, hi - lo);
    { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); RT_TV(4, (hw & 0xFFFF) | (xcc << 16)); }
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < L.n_levels && tile >= TL.tile0[q]) ? 1 : 0;
    const int tl = tile - TL.tile0[l];
    const int ty0 = (tl / TL.tiles_x[l]) * RT_TH, tx0 = (tl % TL.tiles_x[l]) * RT_TW;
    const int H = L.H[l], W = L.W[l];
    // phase B1: lane -> (column cx, channel pair cp, bin rows 0..3 | 4..6);  phase B2: lane -> (column px, rows 4 rg .. 4 rg + 3, channel pairs cq, cq + 8)
    const int cx = t & 7, cp = (t >> 3) & 15, hs = __builtin_amdgcn_readfirstlane(t >> 7);
    const int px = t & 7, cq = (t >> 3) & 7, rg = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c0 = cg * RT_CB;
    const int nc = min(RT_CB, C - c0);
    const int ne = nc * 49;                               // valid dOut elements of one RoI for this channel group
    const RoiEnt *mine = ent + (size_t)tile * cap;

    f32x2 acc[8];                                         // acc[2 r + p]: row 4 rg + r, channel pair cq + 8 p
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x2{0.0f, 0.0f};
    // dOut[r][c0 .. c0 + 31][7][7] is 6272 contiguous bytes and goes to LDS as it lies.  DMA (C a multiple of 32, 16-byte aligned dOut --
    // the launcher decides): moved by global_load_lds_dwordx4 (six 1 KB pieces + 128 bytes, wave w: pieces w and w + 4, wave 2 the
    // tail), no registers, no ds_write, two RoIs ahead; otherwise through registers, one RoI ahead (zero-padded to 32 channels).  Two
    // instantiations, because a register-path load anywhere in the loop makes the compiler wait for vmcnt(0) at the loop's joins.
    constexpr bool dma = DMA;
    const int wv4 = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;
    auto stage_dma = [&](int r, int buf) {
        const float *src = grad_out + ((size_t)r * C + c0) * 49 + ln * 4;
        ra_dma16(src + wv4 * 256, &s_g[buf][wv4 * 256]);
        if (wv4 < 2) ra_dma16(src + (wv4 + 4) * 256, &s_g[buf][(wv4 + 4) * 256]);
        if (wv4 == 2 && ln < 8) ra_dma16(src + 6 * 256, &s_g[buf][6 * 256]);
    };
    // wave 3 brings the RoI's weight-table record: one global_load_lds_dwordx4 from the pool, or built here when the RoI has none
    auto stage_rec = [&](const RoiEnt &e, int buf) {
        if (wv4 != 3) return;
        const int rec = __builtin_amdgcn_readfirstlane(e.rec);
        if (rec >= 0) ra_dma16(pool + (size_t)rec * RA_REC + ln * 4, &s_rec[buf][0]);
        else
            ra_tables_wave(H, W, ty0, tx0, e.sh, e.sw, e.bh, e.bw, &s_rec[buf][0]);
    };
    // the segment's entries go through LDS in chunks (a global load per RoI in the dependent chain cost 1-2 us each)
    for (int cb = lo; cb < hi; cb += RS_CHUNK) {
        const int m = min(RS_CHUNK, hi - cb);
        __syncthreads();                                   // the previous chunk's readers of s_list and of the table buffers are done
        if (t < m) s_list[t] = mine[cb + t];
        __syncthreads();
        float pg[RT_PF];                                   // register path: prefetched dOut elements t + 256 u of the next RoI
#pragma unroll
        for (int u = 0; u < RT_PF; ++u) pg[u] = 0.0f;
        // dOut and record of the next RT_RING - 1 RoIs are requested while RoI i is worked on
        stage_rec(s_list[0], 0);
        if (dma) stage_dma(s_list[0].r, 0);
        if (RT_RING == 3 && m > 1) { stage_rec(s_list[1], 1); if (dma) stage_dma(s_list[1].r, 1); }
        if (!dma) {
            const float *src = grad_out + ((size_t)s_list[0].r * C + c0) * 49;
#pragma unroll
            for (int u = 0; u < RT_PF; ++u) pg[u] = src[min(t + 256 * u, ne - 1)];   // unconditional (clamped; masked at the LDS store): no exec-masked block per load
        }
        int buf = 0;
        for (int i = 0; i < m; ++i) {
            // A (register path only): dOut tile -> LDS
            if (!dma) {
#pragma unroll
                for (int u = 0; u < RT_PF; ++u)
                    if (t + 256 * u < RT_CB * 49) s_g[buf][t + 256 * u] = t + 256 * u < ne ? pg[u] : 0.0f;
                if (i + 1 < m) {
                    const float *src = grad_out + ((size_t)s_list[i + 1].r * C + c0) * 49;
#pragma unroll
                    for (int u = 0; u < RT_PF; ++u) pg[u] = src[min(t + 256 * u, ne - 1)];
                }
            }
            // RoI i's pieces have landed when at most the requests for RoI i + 1 are outstanding: two per wave (wave 3: its dOut piece
            // and the record -- one if that RoI's tables are built here).  The register path waits for everything.
            {
                const int later = RT_RING == 2 || !dma || i + 1 >= m ? 0 : wv4 != 3 ? 2 : 1 + (__builtin_amdgcn_readfirstlane(s_list[min(i + 1, m - 1)].rec) >= 0 ? 1 : 0);
                if (later == 2) ra_vm_wait<2>();
                else if (later == 1) ra_vm_wait<1>();
                else ra_vm_wait<0>();
            }
            ra_barrier();
#ifdef RT_TRACE
            if (cb == lo && i == 0) RT_T(1);
#endif
            if (i + RT_RING - 1 < m) {                     // into the buffers RoI i - 1 used (their last readers passed this barrier)
                const int nb2 = buf == 0 ? RT_RING - 1 : buf - 1;
                stage_rec(s_list[i + RT_RING - 1], nb2);
                if (dma) stage_dma(s_list[i + RT_RING - 1].r, nb2);
            }
            // B1: lane (column, channel pair, half of the bin rows): T[ph] = sum over the bins that reach my column of dOut[c][ph][bin] *
            // Wx[column][bin], two channels per packed FMA.  The bins that reach a pixel are CONSECUTIVE (sample positions grow with the
            // bin): first bin and count come with the record; the trip count is the largest count of the tile's columns.  On its own pyramid
            // level a RoI is 14..28 pixels wide, a bin 2..4 pixels, and a pixel hears from at most two bins: 2 x 7 products per channel
            // where the dense form did 49.  (Adding the zero-weight terms changes nothing: bit-identical to the dense sums.)
            {
                const unsigned *ri = (const unsigned *)s_rec[buf];
                const unsigned ci = ri[RA_REC_COL + cx];
                const int b0 = ci & 255u, spx = ci >> 8;
                const int nbx = __builtin_amdgcn_readfirstlane(ri[RA_REC_NBX]);
                f32x2 T[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) T[q] = f32x2{0.0f, 0.0f};
                for (int j = 0; j < nbx; ++j) {
                    const int b = min(b0 + j, 6);
                    const float w = j < spx ? s_rec[buf][cx * 8 + b] : 0.0f;
                    const f32x2 w2 = {w, w};
                    const float *gp = &s_g[buf][cp * (2 * RT_GS) + hs * 28 + b];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q < 3 || hs == 0) T[q] = __builtin_elementwise_fma(f32x2{gp[q * 7], gp[q * 7 + RT_GS]}, w2, T[q]);
                }
                f32x2 *tp = (f32x2 *)&s_T[cp * RT_TS + (hs * 32 + cx) * 2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < 3 || hs == 0) tp[q * 8] = T[q];
            }
            ra_barrier();
            // B2: lane (column, rows 4 rg .. 4 rg + 3 with rg = my wave, channel pairs cq and cq + 8): acc[row][pair] += sum over the bin rows
            // that reach ANY of my four rows of Wy[row][bin] * T[pair][bin][column].  Four adjacent rows hear from three bins where
            // each alone hears from two: 3 x 16 bytes of T per lane where one row per lane read 2 x 64.  The bin range is the wave's
            // (scalar); the 4 x 7 weights of the wave's rows come with ONE ds_read_b32 (lane -> (row, bin offset)) and reach the packed
            // FMAs as scalar operands through v_readlane.
            {
                const unsigned gi = __builtin_amdgcn_readfirstlane(((const unsigned *)s_rec[buf])[RA_REC_RG + rg]);
                const int p0 = gi & 255u, nb = gi >> 8;
                if (nb != 0) {                               // else: the footprint misses this wave's four rows
                    const int wr = (t >> 3) & 3, wj = p0 + (t & 7);
                    const float wv = wj <= 6 ? s_rec[buf][64 + (rg * 4 + wr) * 8 + wj] : 0.0f;    // lane (row wr, bin p0 + (lane & 7)); zero outside the row's own bins
                    const f32x2 *tp = (const f32x2 *)&s_T[cq * RT_TS + (p0 * 8 + px) * 2];
                    for (int j = 0; j < nb; ++j) {
                        const f32x2 t0 = tp[j * 8], t1 = tp[j * 8 + 4 * RT_TS];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), r * 8 + j));
                            const f32x2 w2 = {w, w};
                            acc[2 * r] = __builtin_elementwise_fma(w2, t0, acc[2 * r]);
                            acc[2 * r + 1] = __builtin_elementwise_fma(w2, t1, acc[2 * r + 1]);
                        }
                    }
                }
            }
            // no barrier here: the next step's barrier stands between these reads of s_T / s_rec and their next writers
            buf = buf == RT_RING - 1 ? 0 : buf + 1;
        }
    }
    if (nseg > 1) {
        // A segment of a split tile (round 5: no combine launch).  My partial tile goes to the workspace LANE-LINEARLY -- 16 floats per lane as four
        // 16-byte write-through stores, [q][lane][4] -- and is acknowledged before the workgroup takes the tile's ticket; the workgroup that takes the LAST
        // ticket of (tile, channel group) then adds the nseg partial tiles IN SEGMENT ORDER, its own included (read back like the others: the sum must not
        // depend on who came last), and writes the plane.  Same values in the same order as the combine kernel added them: bit-identical gradients.
        typedef float f32x4s __attribute__((ext_vector_type(4)));
        __shared__ int s_last_seg;
        float *dst = part + ((size_t)item * n_cg + cg) * (RT_CB * RT_TH * RT_TW) + t * 4;
        {   // all four stores and the wait in ONE statement, from four DISTINCT register tuples: the compiler does not see that inline asm is a store of
            // more than 64 bits, so it neither keeps the data registers alone for the wait state such a store needs nor knows they are read late -- rebuilt
            // in the same registers for the next piece, the last lanes of every 16 of a piece left with the next piece's values
            const f32x4s v0 = {acc[0].x, acc[0].y, acc[1].x, acc[1].y}, v1 = {acc[2].x, acc[2].y, acc[3].x, acc[3].y};
            const f32x4s v2 = {acc[4].x, acc[4].y, acc[5].x, acc[5].y}, v3 = {acc[6].x, acc[6].y, acc[7].x, acc[7].y};
            asm volatile("global_store_dwordx4 %0, %4, off sc1\n\t"
                         "global_store_dwordx4 %1, %5, off sc1\n\t"
                         "global_store_dwordx4 %2, %6, off sc1\n\t"
                         "global_store_dwordx4 %3, %7, off sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         :: "v"(dst), "v"(dst + 1024), "v"(dst + 2048), "v"(dst + 3072), "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "memory");
        }
        __syncthreads();
        if (t == 0) s_last_seg = (__hip_atomic_fetch_add(&tix[tile * n_cg + cg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nseg - 1) ? 1 : 0;
        __syncthreads();
        if (!s_last_seg) { RT_T(2); return; }
        const int32_t *sl = slot + tbase[tile];                // record position of segment s = where its partial tile lies
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const float *src = part + ((size_t)sl[sgm] * n_cg + cg) * (RT_CB * RT_TH * RT_TW) + t * 4;
            f32x4s v[4];
            // the four loads AND their wait in one statement: data returned to a register from inline asm lands behind the compiler's back, and a copy it
            // places between a load and a separate wait statement reads whatever has arrived by then (seen here: lanes 12-15 of every 16 stale)
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                         "global_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %6, off sc1\n\t"
                         "global_load_dwordx4 %3, %7, off sc1\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                         : "v"(src), "v"(src + 1024), "v"(src + 2048), "v"(src + 3072)
                         : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (sgm == 0) { acc[2 * q] = f32x2{v[q][0], v[q][1]}; acc[2 * q + 1] = f32x2{v[q][2], v[q][3]}; }
                else { acc[2 * q] += f32x2{v[q][0], v[q][1]}; acc[2 * q + 1] += f32x2{v[q][2], v[q][3]}; }
            }
        }
    }
    // ---- the tile is complete: a wave stores eight 32-byte row pieces per instruction
    if (tx0 + px < W) {
        TOUT *out = (TOUT *)L.grad[l] + ((size_t)(c0 + 2 * cq) * H + ty0 + rg * 4) * W + tx0 + px;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (ty0 + rg * 4 + r < H && 2 * cq + 16 * p < nc) store_grad<TOUT>(out + ((size_t)(16 * p) * H + r) * W, acc[2 * r + p].x);
                if (ty0 + rg * 4 + r < H && 2 * cq + 16 * p + 1 < nc) store_grad<TOUT>(out + ((size_t)(16 * p + 1) * H + r) * W, acc[2 * r + p].y);
            }
    }
    RT_T(2);
}

FRCNN_EXPORT int frcnn_roi_level_map(const float *rois, int64_t R, int k_min, int k_max, float s0, int k0, float eps, int32_t *out_level,
                                     void *stream)
{
    FRCNN_REQUIRE(R >= 0 && k_max >= k_min && s0 > 0.f, "roi_level_map: bad argument");
    if (R == 0) return FRCNN_OK;
    FRCNN_REQUIRE(rois && out_level, "roi_level_map: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(roi_level_map_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, (const float4 *)rois, R, k_min,
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

// ---- dispatch order of the forward: RoIs by decreasing staging cost ----------------------------------------------------------------
// roi_align_fwd77's workgroup (RoI, 32 channels) stages the RoI's footprint in one to four passes, or gathers from global memory when
// the footprint does not fit at all: its time is set by the footprint, and with the RoIs in list order the launch ends on whichever
// long workgroups happen to come last (a host-side sort of the list gave 42-54 us against 55-63, profiles/README.md).  This kernel
// replaces the `roi * (w, h, w, h)` elementwise launch of FastRCNNHead.forward (models/new_model.py:136-140): ONE workgroup writes the
// scaled boxes (the same fp32 products) and the permutation `order` = RoI indices by (passes, footprint pixels) descending, index
// ascending -- the forward kernel maps dispatch slot -> order[slot] and still writes every RoI's rows at its own index.
#define RO_MAX 4096
// Grid = ceil(R / 32) workgroups of 256 threads.  Every workgroup computes ALL keys into LDS (R / 256 geometry evaluations per thread),
// then ranks 32 RoIs with eight threads each (every thread compares its RoI's key with an eighth of the keys: broadcast LDS reads; three
// shuffles add the partial ranks).  (One workgroup of 1024 threads ranking one RoI per thread took 18 us in the step: 512 dependent
// iterations per thread on one CU.)
__global__ __launch_bounds__(256) void roi_scale_order_kernel(const float4 *__restrict__ rois, int R, float4 mul, MsLevels L, int aligned, int k_min, float s0,
                                                              int k0, float4 *__restrict__ out_rois, int32_t *__restrict__ out_order, uint32_t *__restrict__ out_cost)
{
    __shared__ uint32_t s_key[RO_MAX];
    for (int i = threadIdx.x; i < R; i += 256) {
        float4 b = rois[i];
        b.x = b.x * mul.x; b.y = b.y * mul.y; b.z = b.z * mul.z; b.w = b.w * mul.w;
        if (out_rois && blockIdx.x == 0) out_rois[i] = b;
        const int l = L.n_levels > 1 ? level_of(b, k_min, k_min + L.n_levels - 1, s0, k0, 1e-6f) : 0;
        const int H = L.H[l], W = L.W[l];
        const AlignGeom g = align_geom(b, L.scale[l], 7, 7, 2, aligned != 0);
        const Lin ya = lin_setup(H, g.sh + 0.5f * g.bh / 2.0f), yb = lin_setup(H, g.sh + 6.0f * g.bh + 1.5f * g.bh / 2.0f);
        const Lin xa = lin_setup(W, g.sw + 0.5f * g.bw / 2.0f), xb = lin_setup(W, g.sw + 6.0f * g.bw + 1.5f * g.bw / 2.0f);
        const int fh = yb.hi - ya.lo + 1, fw = xb.hi - xa.lo + 1;
        const long long fp = (long long)fh * (fw | 1);
        uint32_t key;
        if (!(fh >= 1 && fw >= 1) || fp > RA_FWD_LDS) key = (31u << 20) | (uint32_t)(fp > 0xFFFFF ? 0xFFFFF : (fp < 0 ? 0 : fp));   // gathers from global: the longest
        else {
            const int cb = min(RA_FWD_LDS / (int)fp, RA_FWD_CG);
            key = ((uint32_t)((RA_FWD_CG + cb - 1) / cb) << 20) | (uint32_t)fp;
        }
        s_key[i] = key;
        if (out_cost && blockIdx.x == 0) out_cost[i] = key;
    }
    __syncthreads();
    const int i = (int)blockIdx.x * 32 + (int)(threadIdx.x >> 3), sub = (int)(threadIdx.x & 7);
    const uint32_t k = s_key[min(i, R - 1)];
    int rank = 0;
    for (int j = sub; j < R; j += 8) { const uint32_t q = s_key[j]; rank += (q > k) || (q == k && j < i); }
    rank += __shfl_xor(rank, 1); rank += __shfl_xor(rank, 2); rank += __shfl_xor(rank, 4);
    if (sub == 0 && i < R) out_order[rank] = i;
}

FRCNN_EXPORT int frcnn_roi_scale_order(const float *rois, int64_t R, const float *mul4_host, const int *H, const int *W, const float *scales, int n_levels,
                                       int aligned, int k_min, float s0, int k0, float *out_rois, int32_t *out_order, uint32_t *out_cost, void *stream)
{
    FRCNN_REQUIRE(R >= 0 && s0 > 0.f && mul4_host, "roi_scale_order: bad argument");
    if (R == 0) return FRCNN_OK;
    if (R > RO_MAX) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "roi_scale_order: R=%lld above limit %d", (long long)R, RO_MAX);
    FRCNN_REQUIRE(rois && out_order, "roi_scale_order: NULL pointer");
    MsLevels L;
    static const float *dummy[FRCNN_MAX_LEVELS] = {(const float *)16, (const float *)16, (const float *)16, (const float *)16,
                                                  (const float *)16, (const float *)16, (const float *)16, (const float *)16};   // geometry only: never dereferenced
    int rc = fill_levels(&L, dummy, nullptr, H, W, scales, n_levels);
    if (rc) return rc;
    FRCNN_LAUNCH(roi_scale_order_kernel, dim3((unsigned)((R + 31) / 32)), dim3(256), 0, (hipStream_t)stream, (const float4 *)rois, (int)R,
                 make_float4(mul4_host[0], mul4_host[1], mul4_host[2], mul4_host[3]), L, aligned, k_min, s0, k0, (float4 *)out_rois, out_order, out_cost);
    FRCNN_CHECK_LAUNCH("roi_scale_order_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_ms_roi_align_fwd(const float *const *feats, const int *H, const int *W, const float *scales, int n_levels, int C,
                                        const float *rois, int64_t R, int PH, int PW, int sampling_ratio, int aligned, int k_min, float s0,
                                        int k0, float *out, int32_t *out_level, const int32_t *order, void *stream)
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
        FRCNN_LAUNCH(roi_align_fwd77_kernel, dim3((unsigned)(n_cg * R)), dim3(256), 0, s, L, C, (const float4 *)rois, (int)R,
                     aligned, k_min, s0, k0, out, out_level, n_cg, order);
        FRCNN_CHECK_LAUNCH("roi_align_fwd77_kernel");
        return FRCNN_OK;
    }
    FRCNN_LAUNCH(roi_align_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, C, (const float4 *)rois,
                 total, PH, PW, sampling_ratio, aligned, k_min, s0, k0, out, out_level);
    FRCNN_CHECK_LAUNCH("roi_align_fwd_kernel");
    return FRCNN_OK;
}

// workspace of the 7 x 7 tile gather: list lengths, lists (capacity R per tile), the plan, partial tiles of split tiles
struct RaBwdWs { int32_t *cnt, *tbase, *tnseg, *slot, *tix; int4 *items; RoiEnt *ent; float *pool, *part; int cap_items; size_t total; };
static RaBwdWs carve_ra_bwd(void *ws, int64_t tiles, int64_t R, int n_cg)
{
    RaBwdWs w; char *p = (char *)ws; size_t o = 0;
    auto take = [&](size_t b) { void *r = p ? p + o : nullptr; o += align_up(b, 256); return r; };
    // a RoI meets at most ~15 tiles of its level: at most tiles + 15 R / RS_SPLIT items; the plan kernel coarsens the split if not
    const int64_t cap_items = tiles + 15 * R / RS_SPLIT + 1;
    w.cap_items = (int)cap_items;
    w.cnt = (int32_t *)take((size_t)(tiles + 1) * 4);
    w.tbase = (int32_t *)take((size_t)(tiles + 1) * 4);
    w.tnseg = (int32_t *)take((size_t)(tiles + 1) * 4);
    w.tix = (int32_t *)take((size_t)(tiles + 1) * n_cg * 4);           // last-segment tickets, one per (tile, channel group): zeroed by the lists launch
    w.items = (int4 *)take((size_t)(cap_items + RS_NSEG) * sizeof(int4));
    w.slot = (int32_t *)take((size_t)(cap_items + RS_NSEG) * 4);
    w.ent = (RoiEnt *)take((size_t)tiles * (size_t)(R > 0 ? R : 1) * sizeof(RoiEnt));
    w.pool = (float *)take((size_t)(R > 0 ? R : 1) * RA_MAXT * RA_REC * sizeof(float));
    w.part = (float *)take((size_t)cap_items * n_cg * (RT_CB * RT_TH * RT_TW) * sizeof(float));
    w.total = o;
    return w;
}
static int64_t ra_bwd_tiles(const int *H, const int *W, int n_levels)
{
    int64_t tiles = 0;
    for (int l = 0; l < n_levels; ++l) tiles += (int64_t)((W[l] + RT_TW - 1) / RT_TW) * ((H[l] + RT_TH - 1) / RT_TH);
    return tiles;
}

FRCNN_EXPORT size_t frcnn_ms_roi_align_bwd_workspace(const int *H, const int *W, int n_levels, int C, int64_t R)
{
    if (!H || !W || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || C <= 0 || R < 0) return 0;
    return carve_ra_bwd(nullptr, ra_bwd_tiles(H, W, n_levels), R, (C + RT_CB - 1) / RT_CB).total;
}

FRCNN_EXPORT int frcnn_ms_roi_align_bwd(const float *grad_out, float *const *grad_feats, const int *H, const int *W, const float *scales,
                                        int n_levels, int C, const float *rois, int64_t R, int PH, int PW, int sampling_ratio, int aligned,
                                        int k_min, float s0, int k0, void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(C > 0 && PH > 0 && PW > 0 && R >= 0 && sampling_ratio >= 0 && s0 > 0.f, "ms_roi_align_bwd: bad argument");
    FRCNN_REQUIRE((R == 0 || (rois && grad_out)), "ms_roi_align_bwd: NULL pointer");
    MsLevels L;
    int rc = fill_levels(&L, nullptr, grad_feats, H, W, scales, n_levels);
    if (rc) return rc;
    const int64_t total = R * C * PH * PW;
    hipStream_t s = (hipStream_t)stream;
    if (PH == 7 && PW == 7 && sampling_ratio == 2 && R < (1 << 24)) {
        TileLevels T;
        int tiles = 0;
        for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
            T.tile0[l] = tiles;
            T.tiles_x[l] = (L.W[l] + RT_TW - 1) / RT_TW;
            if (l < n_levels) tiles += T.tiles_x[l] * ((L.H[l] + RT_TH - 1) / RT_TH);
        }
        T.tile0[FRCNN_MAX_LEVELS] = tiles;
        const int n_cg = (C + RT_CB - 1) / RT_CB;
        const size_t need = frcnn_ms_roi_align_bwd_workspace(H, W, n_levels, C, R);
        FRCNN_REQUIRE(workspace != nullptr, "ms_roi_align_bwd: NULL workspace");
        if (workspace_bytes < need) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "ms_roi_align_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
        const RaBwdWs w = carve_ra_bwd(workspace, tiles, R, n_cg);
        const int cap = (int)(R > 0 ? R : 1);
        static const bool use_records = [] { const char *e = getenv("FRCNN_RA_RECORDS"); return !e || atoi(e) != 0; }();
        const int64_t table_blocks = use_records ? R : 0;
        FRCNN_REQUIRE(tiles + table_blocks < ((int64_t)1 << 31) && R * RA_MAXT < ((int64_t)1 << 31), "ms_roi_align_bwd: grid too large");
        static int *ticket_of_dev[64];                         // device address of g_ra_plan_ticket, per device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        int *tickets = __atomic_load_n(&ticket_of_dev[dev], __ATOMIC_RELAXED);
        if (!tickets) {
            if (hipGetSymbolAddress((void **)&tickets, HIP_SYMBOL(g_ra_plan_ticket)) != hipSuccess || !tickets)
                return frcnn_set_error(FRCNN_ERR_LAUNCH, "ms_roi_align_bwd: no address for the plan ticket");
            __atomic_store_n(&ticket_of_dev[dev], tickets, __ATOMIC_RELAXED);
        }
        int *ticket = tickets + (((size_t)workspace >> 12) & 63);
        FRCNN_LAUNCH(roi_align_bwd_lists_kernel, dim3((unsigned)(tiles + table_blocks)), dim3(256), 0, s, L, T, (const float4 *)rois, (int)R,
                     aligned, k_min, s0, k0, cap, tiles, w.cnt, w.ent, use_records ? w.pool : nullptr, n_cg, w.tix, ticket, w.cap_items, w.tbase,
                     w.tnseg, w.items, w.slot);
        FRCNN_CHECK_LAUNCH("roi_align_bwd_lists_kernel");
        FillLevels FLv;
        int64_t fills = 0;
        for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
            FLv.fill0[l] = (int)fills;
            if (l < n_levels) fills += (int64_t)((L.H[l] + RT_TH - 1) / RT_TH) * ((C + RF_CH - 1) / RF_CH);
        }
        FLv.fill0[FRCNN_MAX_LEVELS] = (int)fills;
        const int64_t n_item_blocks = (int64_t)w.cap_items * n_cg;
        FRCNN_REQUIRE(n_item_blocks + fills < ((int64_t)1 << 31), "ms_roi_align_bwd: grid too large");
        static const int resident = [] {                   // workgroups of the tile kernel the device holds at once
            int dev = 0, cus = 256, per = 5;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, roi_align_bwd_tile_kernel<float, true>, 256, 0) != hipSuccess || per < 1) per = 5;
            return cus * per;
        }();
        const int n_first = (int)std::min<int64_t>(n_item_blocks, (int64_t)resident / n_cg * n_cg);
        if (C % RT_CB == 0 && ((size_t)grad_out & 15) == 0)
            FRCNN_LAUNCH((roi_align_bwd_tile_kernel<float, true>), dim3((unsigned)(n_item_blocks + fills)), dim3(256), 0, s, L, T, FLv, C, aligned,
                         grad_out, n_cg, cap, (int)n_item_blocks, n_first, (int)fills, w.cnt, w.ent, w.items, w.pool, w.part, w.tbase, w.slot, w.tix);
        else
            FRCNN_LAUNCH((roi_align_bwd_tile_kernel<float, false>), dim3((unsigned)(n_item_blocks + fills)), dim3(256), 0, s, L, T, FLv, C, aligned,
                         grad_out, n_cg, cap, (int)n_item_blocks, n_first, (int)fills, w.cnt, w.ent, w.items, w.pool, w.part, w.tbase, w.slot, w.tix);
        FRCNN_CHECK_LAUNCH("roi_align_bwd_tile_kernel");
        return FRCNN_OK;
    }
    for (int l = 0; l < n_levels; ++l)                  // the scatter kernels accumulate: clear the planes first
        if (hipMemsetAsync(L.grad[l], 0, (size_t)C * L.H[l] * L.W[l] * sizeof(float), s) != hipSuccess)
            return frcnn_set_error(FRCNN_ERR_LAUNCH, "ms_roi_align_bwd: memset failed");
    if (R == 0) return FRCNN_OK;
    FRCNN_LAUNCH(roi_align_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, C, (const float4 *)rois,
                 total, PH, PW, sampling_ratio, aligned, k_min, s0, k0, grad_out);
    FRCNN_CHECK_LAUNCH("roi_align_bwd_kernel");
    return FRCNN_OK;
}
