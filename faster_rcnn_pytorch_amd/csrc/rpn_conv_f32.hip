// rpn_conv_f32.hip -- 3x3 convolutions (padding 1, stride 1, batch 1) in the reference's own precision (fp32), hand-written for gfx950.  Built for
//   models/model.py:68-70, 79      self.inter_layer = nn.Conv2d(512, 512, 3, padding=1)  on the 37 x 62 VGG16 feature map
//   models/new_model.py:96-98, 109 self.inter_layer = nn.Conv2d(256, 256, 3, padding=1)  on the five FPN levels
// (frcnn_rpn_conv3x3_f32_*: bias-free; the bias, the ReLU and both 1x1 heads stay in rpn_head.hip), and then generalised (frcnn_conv3x3_f32_*:
// Cin != Cout, bias / ReLU / 2 x 2 max-pool in the output transform, their backward in the gradient transforms) for the backbone's own layers
// behind models/model.py:279-281 and models/new_model.py:372, which the vendor library ran on the vector units.
// forward, data gradient and weight gradient on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit for bit a k-ordered fmaf chain,
// no reduced precision; 64 cycles per instruction and SIMD = the fp32 vector rate, 157 TFLOP/s per chip).
//
// TWO forms live in this file.  The DEFAULT one for all three directions is the Winograd F(4x4 | 2x2, 3x3) stage (section "Winograd" below:
// rpn_wino_weight / _input<m,0> / _gemm<false> / _output for forward and data gradient, rpn_wino_input<m,0> + <m,1> / _gemm<true> / _dw for the
// weight gradient): 4 x or 2.25 x fewer MFMAs.  The DIRECT form (9 C deep implicit GEMM, described next; Cin = Cout only) is kept behind
// FRCNN_CONV_F32_DIRECT=1 for the RPN entry points as the A/B partner and the independent check of the stage (tests/test_gpu_ops.py runs it in a
// child process); the stage's GEMM reuses its stream-K skeleton.
//
// rpn_conv3x3_f32_kernel (forward; data gradient = the same kernel on the weights transposed and flipped by rpn_conv_f32_pack_kernel):
//   implicit GEMM  Y[co][p] = sum_{ci, tap} Wt[co][ci * 9 + tap] * X[ci][p + off(tap)],  M = co, N = flat positions p of one level, K = 9 C.
//   Workgroup tile 128 co x 128 consecutive flat positions, 4 waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulators per lane).
//   K runs in chunks of 4 input channels x 9 taps.  The weights of a chunk are 36 CONTIGUOUS floats per output channel in the
//   reference's own [co][ci][3][3] layout (no packing pass for the forward); the activations of a chunk are 4 ci x 3 dy plain copies
//   of 130 consecutive floats of the flattened plane (p0 - 1 + (dy - 1) W ...): a tap's operand is then ONE ds_read_b32 at
//   lane address + compile-time immediate, and the only thing the flat copy gets wrong -- the left / right neighbour of a pixel in
//   the first / last column wraps to the neighbouring row -- is repaired by a v_cndmask with a per-lane edge flag (rows outside the
//   image are zero-filled at staging time).  Both operands are double-buffered in LDS (2 x 25.8 KB), one barrier per chunk; the
//   next chunk's global loads are issued before the current chunk's 72 MFMAs per wave.
//   STREAM-K: the (tile, chunk) units of the whole problem are cut into G = 2 x CUs equal contiguous ranges, one per workgroup
//   (600x1000: 72 tiles x 128 chunks = 9216 units = 18 per workgroup; FPN: 1404 x 64 = 89 856 = 175.5): every matrix pipe gets the
//   same number of MFMAs whatever the shape.  A range that covers only part of a tile's K writes its accumulators to a slab
//   (write-through), takes a ticket on the tile, and the workgroup that completes the tile adds the slabs IN RANGE ORDER and stores
//   the tile: deterministic, no atomics on data, no memset (the ticket words are left zero).
// rpn_conv3x3_f32_wgrad_kernel:
//   dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p + off(tap)]:  M = co, N = ci, K = positions, one accumulator tile per tap.
//   Workgroup = one 32 co x 32 ci x 9 tap output tile; its four WAVES are four K ranges: every wave walks row segments of 16
//   pixels down column strips with its own private LDS ring (dY segment double-buffered, three feature rows + one in flight), nine
//   MFMAs per column pair off one dY fragment read and nine feature fragment reads at immediates, no workgroup barrier in the loop.
//   At the end the four waves' accumulators are added through LDS in wave order and the tile is written once: at 600x1000 (256
//   tiles = one workgroup per CU, one strip per wave) there are NO global partials; at FPN size (64 tiles) four workgroups share a
//   tile through slabs + ticket + fixed-order sum like above.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
FRCNN_LAYOUT_STAMP(rpn_conv_f32);

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CF_MT 128                      // output channels per workgroup tile
#define CF_NT 128                      // flat positions per workgroup tile
#define CF_CI 4                        // input channels per K chunk
#define CF_KK (CF_CI * 9)              // 36 K values per chunk
#define CF_AS 37                       // LDS row stride of the weight tile (floats): odd -> the 32 rows of a fragment read hit 32 banks
#define CF_RS 144                      // LDS row stride of one (ci, dy) activation copy: 130 used; 6 * 144 = 32 (mod 64): the two k halves hit disjoint banks
#define CF_ROWS (CF_CI * 3)
#define CF_SLAB (CF_MT * CF_NT)        // floats per partial slab
#define CF_MAX_TILES 32768             // ticket words in the control block

struct CfLevel { const float *x; float *y; int H, W, HW, tile0; };
struct CfArgs {
    CfLevel lv[FRCNN_MAX_LEVELS];
    int n_levels, C, n_co_tiles, n_pos_tiles, Kc, n_units, G, pos_major;
};
struct CfTile { const float *x; float *y; int W, HW, p0, co0, tile; };

__device__ __forceinline__ CfTile cf_tile(const CfArgs &a, int t)
{
    // tile order: co-major (small maps: the positions of one co tile are neighbours, a weight slice stays in L2) or position-major
    // (large maps: the co tiles of one position tile are neighbours, the activations are re-read while they are still cached)
    int ct, pt;
    if (a.pos_major) { pt = t / a.n_co_tiles; ct = t - pt * a.n_co_tiles; }
    else { ct = t / a.n_pos_tiles; pt = t - ct * a.n_pos_tiles; }
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < a.n_levels && pt >= a.lv[q].tile0) ? 1 : 0;
    CfTile T;
    T.x = a.lv[l].x; T.y = a.lv[l].y; T.W = a.lv[l].W; T.HW = a.lv[l].HW;
    T.p0 = (pt - a.lv[l].tile0) * CF_NT;
    T.co0 = ct * CF_MT;
    T.tile = t;
    return T;
}

__device__ __forceinline__ long long cf_start(int s, int U, int G) { return ((long long)s * U) / G; }

#ifndef CF_WPS
#define CF_WPS 2
#endif
__global__ __launch_bounds__(256, CF_WPS) void rpn_conv3x3_f32_kernel(CfArgs a, const float *__restrict__ w, float *__restrict__ part, int *__restrict__ cnt)
{
    __shared__ float sA[2][CF_MT * CF_AS];
    __shared__ float sB[2][CF_ROWS * CF_RS];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int G = a.G, U = a.n_units, Kc = a.Kc;
    // logical range id: workgroups are dealt round-robin to the 8 XCDs, so the ranges of one XCD are made neighbours (shared weight /
    // activation lines stay in that XCD's L2)
    const int sigma = (G % 8 == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int u0 = (int)cf_start(sigma, U, G), u1 = (int)cf_start(sigma + 1, U, G);
    if (u0 >= u1) return;
    const int Ktot = a.C * 9;

    // Per-thread staging slots, fixed for the whole kernel: five 16-byte pieces of the weight tile (piece idx = tid + 256 q of 128 rows x 9
    // pieces) and seven elements of the activation copies (six passes of two (ci, dy) rows x 128 + the two halo columns of every row).
    // Only 32-bit offsets from wave-uniform bases live in registers (the first version kept 64-bit addresses and recomputed row / column
    // splits per chunk: 253 spilled registers, and every spill reload inside the load sequence waited for the loads issued before it).
    unsigned a_off[5];
    int a_lds[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int idx = tid + 256 * q, row = idx / 9, pc = idx - row * 9;
        a_off[q] = (unsigned)(row * Ktot + pc * 4);
        a_lds[q] = row * CF_AS + pc * 4;
    }
    // activation copies: pass q < 6 covers the (ci, dy) rows 2q and 2q + 1 (thread half `hi`), column j = tid & 127; pass 6 = the two halo
    // columns (-1, 128) of row tid >> 1 for the first 24 threads.  Row -> (ci, dy) are compile-time constants selected by `hi`.
    const int hi = tid >> 7, bj = tid & 127;
    const int h_row = (tid >> 1) % CF_ROWS, h_ci = h_row / 3, h_dy = h_row - 3 * h_ci - 1, h_j = (tid & 1) ? CF_NT : -1;
    const bool a_tail = tid < 128, b_tail = tid < 2 * CF_ROWS;
    float4 ra[5];
    float rb[7];
    unsigned rb_ok = 0;                                              // bit q: element q of the next chunk lies inside the plane (applied at the LDS store)
    auto issue_loads = [&](const CfTile &T, int chunk) {
        const float *wb = w + (size_t)T.co0 * Ktot + chunk * CF_KK;                 // wave-uniform bases, 32-bit lane offsets
        const float *xb = T.x + (size_t)chunk * CF_CI * T.HW;
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (q < 4 || a_tail) ra[q] = *(const float4 *)(wb + a_off[q]);
        unsigned okm = 0;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            int ci, dy, j;
            if (q < 6) { ci = hi ? (2 * q + 1) / 3 : (2 * q) / 3; dy = hi ? (2 * q + 1) % 3 - 1 : (2 * q) % 3 - 1; j = bj; }
            else { ci = h_ci; dy = h_dy; j = h_j; }
            const int f = T.p0 + j + dy * T.W;
            const bool ok = f >= 0 && f < T.HW && (q < 6 || b_tail);
            rb[q] = xb[ok ? (unsigned)(ci * T.HW + f) : 0u];                      // unconditional load at a clamped offset; the zero is selected at the LDS store
            okm |= ok ? (1u << q) : 0u;
        }
        rb_ok = okm;
    };
    auto store_part = [&](int buf, int part) {                      // part 0..4 of the staged chunk -> LDS (one weight piece + one or three activation elements)
        if (part < 4 || a_tail) {
            float *d = &sA[buf][a_lds[part]];
            d[0] = ra[part].x; d[1] = ra[part].y; d[2] = ra[part].z; d[3] = ra[part].w;
        }
        float *db = &sB[buf][hi * CF_RS + bj + 1];
        db[2 * part * CF_RS] = (rb_ok >> part) & 1u ? rb[part] : 0.0f;
        if (part == 4) {
            db[2 * 5 * CF_RS] = (rb_ok >> 5) & 1u ? rb[5] : 0.0f;
            if (b_tail) sB[buf][h_row * CF_RS + h_j + 1] = (rb_ok >> 6) & 1u ? rb[6] : 0.0f;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int part = 0; part < 5; ++part) store_part(buf, part);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // output element (mi, ni, r) of this lane: co = co0 + wm*64 + mi*32 + (r&3) + 8*(r>>2) + 4*lh, p = p0 + wn*64 + ni*32 + li
    auto store_tile = [&](const CfTile &T) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int p = T.p0 + wn * 64 + ni * 32 + li;
            if (p < T.HW) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = T.co0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        T.y[(size_t)co * T.HW + p] = acc[mi][ni][r];
                    }
            }
        }
    };
    auto finish_segment = [&](const CfTile &T, int first_chunk, int n_chunks) {
        if (n_chunks == Kc) { store_tile(T); return; }
#if defined(CF_ABL) && (CF_ABL & 1)                                 // developer ablation: no slabs, no tickets, no reduction (results meaningless)
        if (first_chunk == 0) store_tile(T);
        return;
#endif
        float *slab = part + ((size_t)sigma * 2 + (first_chunk == 0 ? 1 : 0)) * CF_SLAB;
        // register order [wave][mi][ni][r][lane]: 256 contiguous bytes per store, one address per accumulator tile + immediates, write-through
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                float *dst = &slab[(wave * 4 + mi * 2 + ni) * 16 * 64 + lane];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    asm volatile("global_store_dword %0, %1, off offset:%2 sc1" :: "v"(dst), "v"(acc[mi][ni][r]), "n"(r * 256) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the slab has been performed before the ticket announces it
        __syncthreads();
        if (tid == 0) s_last = (__hip_atomic_fetch_add(&cnt[T.tile], n_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + n_chunks == Kc) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        // this workgroup completed the tile: add every range's slab in range order (its own included: the order never depends on who is last)
        const long long lo = (long long)T.tile * Kc, hi = lo + Kc - 1;
        int s_first = (int)((lo * G) / U), s_end = (int)((hi * G) / U);
        while (cf_start(s_first + 1, U, G) <= lo) ++s_first;
        while (cf_start(s_first, U, G) > lo) --s_first;
        while (cf_start(s_end + 1, U, G) <= hi) ++s_end;
        while (cf_start(s_end, U, G) > hi) --s_end;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
        for (int s = s_first; s <= s_end; ++s) {
            const long long st = cf_start(s, U, G);
            const float *sl = part + ((size_t)s * 2 + (st <= lo ? 1 : 0)) * CF_SLAB;
            // all 64 agent-scope loads of the slab in flight (four addresses + immediates), adds behind counted waits (written as
            // acc += load the compiler waits for every load by itself: 64 dependent round trips per slab, 240 us per tile at 600x1000)
            float t[4][16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float *src = &sl[(wave * 4 + g) * 16 * 64 + lane];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    asm volatile("global_load_dword %0, %1, off offset:%2 sc1" : "=v"(t[g][r]) : "v"(src), "n"(r * 256) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(48)" : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[0][2]), "+v"(t[0][3]), "+v"(t[0][4]), "+v"(t[0][5]), "+v"(t[0][6]), "+v"(t[0][7]), "+v"(t[0][8]), "+v"(t[0][9]), "+v"(t[0][10]), "+v"(t[0][11]), "+v"(t[0][12]), "+v"(t[0][13]), "+v"(t[0][14]), "+v"(t[0][15]) :: "memory");     // the adds below depend on this statement
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += t[0][r];
            asm volatile("s_waitcnt vmcnt(32)" : "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[1][2]), "+v"(t[1][3]), "+v"(t[1][4]), "+v"(t[1][5]), "+v"(t[1][6]), "+v"(t[1][7]), "+v"(t[1][8]), "+v"(t[1][9]), "+v"(t[1][10]), "+v"(t[1][11]), "+v"(t[1][12]), "+v"(t[1][13]), "+v"(t[1][14]), "+v"(t[1][15]) :: "memory");     // the adds below depend on this statement
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][1][r] += t[1][r];
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[2][2]), "+v"(t[2][3]), "+v"(t[2][4]), "+v"(t[2][5]), "+v"(t[2][6]), "+v"(t[2][7]), "+v"(t[2][8]), "+v"(t[2][9]), "+v"(t[2][10]), "+v"(t[2][11]), "+v"(t[2][12]), "+v"(t[2][13]), "+v"(t[2][14]), "+v"(t[2][15]) :: "memory");     // the adds below depend on this statement
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][0][r] += t[2][r];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[3][0]), "+v"(t[3][1]), "+v"(t[3][2]), "+v"(t[3][3]), "+v"(t[3][4]), "+v"(t[3][5]), "+v"(t[3][6]), "+v"(t[3][7]), "+v"(t[3][8]), "+v"(t[3][9]), "+v"(t[3][10]), "+v"(t[3][11]), "+v"(t[3][12]), "+v"(t[3][13]), "+v"(t[3][14]), "+v"(t[3][15]) :: "memory");     // the adds below depend on this statement
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][1][r] += t[3][r];
        }
        store_tile(T);
        if (tid == 0) __hip_atomic_store(&cnt[T.tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
    };

    int tile = u0 / Kc, chunk = u0 - tile * Kc;
    CfTile T = cf_tile(a, tile);
    int seg_first = chunk;
    bool el[2], er[2];
    auto edge_flags = [&](const CfTile &Tt) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = (Tt.p0 + wn * 64 + ni * 32 + li) % Tt.W;
            el[ni] = col == 0;
            er[ni] = col == Tt.W - 1;
        }
    };
    edge_flags(T);
    issue_loads(T, chunk);
    store_lds(0);
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
        const int buf = (u - u0) & 1;
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == Kc) { nchunk = 0; ++ntile; }
        CfTile Tn = T;
        const bool more = u + 1 < u1;
        if (more && ntile != tile) Tn = cf_tile(a, ntile);
        const float *pa = &sA[buf][(wm * 64 + li) * CF_AS + 18 * lh];
        const float *pb = &sB[buf][6 * lh * CF_RS + wn * 64 + li];
        // 18 k steps (ci pair cp, tap t), operands of step s + 1 read from LDS before the four MFMAs of step s are issued.  The next
        // chunk's global loads are issued behind the first step's MFMAs and its LDS stores ride in the last five steps (the scheduler
        // is fenced between steps: left alone it puts all loads in front of and all stores behind the 72 MFMAs, where nothing hides them)
        float oa[2][2], ob[2][2];
        auto fetch = [&](int s, int slot) {
            const int cp = s / 9, t = s - cp * 9, dy = t / 3, dx = t % 3 - 1;
#if defined(CF_ABL) && (CF_ABL & 2)                                 // developer ablation: no LDS operand reads
            oa[slot][0] = (float)(s + lane); oa[slot][1] = (float)(s - lane); ob[slot][0] = (float)(t + lane); ob[slot][1] = (float)(dy * lane + dx);
#else
            oa[slot][0] = pa[cp * 9 + t]; oa[slot][1] = pa[32 * CF_AS + cp * 9 + t];
            ob[slot][0] = pb[(cp * 3 + dy) * CF_RS + 1 + dx]; ob[slot][1] = pb[(cp * 3 + dy) * CF_RS + 1 + dx + 32];
#endif
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int slot = s & 1, dx = (s % 9) % 3 - 1;
            if (s + 1 < 18) fetch(s + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);                      // the next step's LDS reads go out BEFORE this step's MFMAs
            float b0 = ob[slot][0], b1 = ob[slot][1];
            if (dx == -1) { b0 = el[0] ? 0.0f : b0; b1 = el[1] ? 0.0f : b1; }
            if (dx == 1) { b0 = er[0] ? 0.0f : b0; b1 = er[1] ? 0.0f : b1; }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot][0], b0, acc[0][0], 0, 0, 0);
#if !(defined(CF_ABL) && (CF_ABL & 32))                             // developer ablations 16 / 32: no LDS stores / no global loads of the next chunk
            if (s == 0 && more) issue_loads(Tn, nchunk);
#endif
#if !(defined(CF_ABL) && (CF_ABL & 16))
            if (s >= 13 && more) store_part(buf ^ 1, s - 13);
#endif
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot][0], b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot][1], b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot][1], b1, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!more || ntile != tile) {
            finish_segment(T, seg_first, chunk + 1 - seg_first);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
            seg_first = 0;
            if (more) { T = Tn; edge_flags(T); }
        }
        tile = ntile; chunk = nchunk;
        __syncthreads();
    }
}

// Wt[ci][co][e] = W[co][ci][8 - e]: the data gradient dX = conv(dY, Wt) runs on the forward kernel.  16 x 16 (co, ci) blocks through LDS:
// 576-byte contiguous runs on both sides, C * C / 256 workgroups (1024 at C = 512).
__global__ __launch_bounds__(256) void rpn_conv_f32_pack_kernel(const float *__restrict__ w, float *__restrict__ wt, int C)
{
    __shared__ float s[16][16 * 9 + 1];
    const unsigned nb = (unsigned)C / 16u, co0 = (blockIdx.x / nb) * 16u, ci0 = (blockIdx.x % nb) * 16u, row = (unsigned)C * 9u;
#pragma unroll
    for (unsigned k = 0; k < 9; ++k) {
        const unsigned e = threadIdx.x + 256u * k, r = e / 144u, q = e - r * 144u;       // r = co, q = ci * 9 + tap
        s[r][q] = w[(co0 + r) * row + ci0 * 9u + q];
    }
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < 9; ++k) {
        const unsigned e = threadIdx.x + 256u * k, r = e / 144u, q = e - r * 144u;       // r = ci, q = co * 9 + tap
        const unsigned co = q / 9u, t = q - co * 9u;
        wt[(ci0 + r) * row + co0 * 9u + q] = s[co][r * 9u + 8u - t];
    }
}

// ---------------------------------------------------------------------------------------------------------------- Winograd F(m x m, 3x3)
// Forward and data gradient as Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A: P = (m + 2)^2 independent [Cout x Cin] . [Cin x T] products over
// the m x m output tiles (T = ceil(H/m) ceil(W/m) per level) instead of one with K = 9 Cin.  m = 2: 16 products, 2.25 x fewer MFMAs
// (fp32: ~3e-7 of the output scale, as good as the direct sum); m = 4: 36 products over a quarter of the tiles, 4 x fewer MFMAs and 0.56 x
// the transformed bytes (fp32: ~5e-6 of the scale -- the 1/24 .. 8 range of its transform constants -- inside the path's 1e-4; maps where
// the tile padding would eat the gain stay with m = 2: wn_pick_m prices both forms by planes x padded tiles).  Four launches, all levels each:
//   rpn_wino_weight_kernel   U[xi][k][m] = (G g G^T)_xi of W[m][k] (forward) or of the flipped W[k][m] (data gradient): once per call
//   rpn_wino_input_kernel    V[xi][k][t] = (B^T d B)_xi of the zero-padded (m + 2)^2 input patch of tile t; rows padded to a multiple of 128 (or 64: wn_padded) tiles
//   rpn_wino_gemm_kernel     M[xi][m][t] = sum_k U[xi][k][m] V[xi][k][t]: stream-K over (xi, 128 x 128 tile, 32-channel chunk) units on
//                            v_mfma_f32_32x32x2_f32, the direct kernel's skeleton without taps: both operand tiles (32 K rows x 128
//                            floats) are staged by LDS-DMA (global_load_lds_dwordx4, no staging registers), no edge selects,
//                            16 k steps per chunk
//   rpn_wino_output_kernel   Y[m][m ty + i][m tx + j] = (A^T M A)_ij (+ bias, ReLU, ReLU + 2 x 2 max-pool; sign / window words)
//   (64 -> 64 channels on 4 x 4 tiles: the last two as ONE launch, rpn_wino_gemm_out64_kernel -- the products stay in the accumulators)
// Weight gradient: dU[xi][co][ci] = sum_t (A dY A^T)_xi[co][t] (B^T d B)_xi[ci][t], dW = G^T dU G: the SAME input transform (of the
// activations) and its sibling for the output gradient write [xi][channel][t] with the tiles contiguous, and the product -- which sums over
// the tiles -- runs on the k-contiguous form of the GEMM (rpn_wino_gemm_kernel<true>: both operand tiles 128 rows x 32 k, 16-byte pieces
// XOR-swizzled at the source side of the LDS-DMA, fragments by ds_read_b128); no transposing pass.
// The transformed operands travel through the workspace (P C Ttot floats each way).
#define WN_KC 32                       // K values per chunk
#ifndef WN_ONE_TILE_NUM
#define WN_ONE_TILE_NUM 3               // one whole tile per workgroup from NUM / DEN of the workgroup slots on
#define WN_ONE_TILE_DEN 4
#endif
struct WnLevel { const float *x; float *y; int H, W, tw, T, off; };
struct WnArgs {
    WnLevel lv[FRCNN_MAX_LEVELS];
    const float *bias;                 // output transform: + bias[c] (or NULL), then ReLU if relu != 0
    // the ReLU's sign pattern as ONE 16-bit word per (channel, output tile): bit i * m + j = output (m ty + i, m tx + j) > 0.  Written by the output
    // transform (bits_out), read by the gradient kernels' input transforms (bits: the gradient counts where the bit is set -- what autograd's
    // threshold_backward computes in a launch of its own, from 1/32 of the bytes of the activations)
    unsigned short *bits_out; const unsigned short *bits;
    int relu, zero_pad;                // zero_pad: the input transforms also write zeros into the padding columns (weight gradient: the product sums over them)
    float *out2;                       // MODE 2 of the input transform: where the second (weight-gradient) transform of the staged gradient goes
    float *db_part;                    // MODE 1 / 2 of the input transform also leaves per-strip sums of its (masked) gradient: the bias gradient's partials (NULL: not wanted)
    int n_levels, C, Ttot;             // C = channels of the side the launch touches
    int tg;                            // the levels' tile counts are padded to multiples of tg (128, or 64: wn_tgran) = the product's tile width on that side
};
// the batched product of the stage, both forms:
//   <false>  O[xi][m][n] = sum_k A[xi][k][m] B[xi][k][n], operands K-major (rows k, 128 columns per tile): forward / data gradient,
//            A = U [k = ci][m = co], B = V [ci][t], O = M [co][t], K = Cin (Cout for the data gradient)
//   <true>   O[xi][m][n] = sum_k A[xi][m][k] B[xi][n][k], operands k-contiguous (128 rows per tile, 32 k per chunk): weight gradient,
//            A = dM [co][t], B = V [ci][t], O = dU [co][ci], K = Ttot
struct WgArgs {
    const float *A, *B; float *O;
    long long sA, sB, sO;              // xi strides (floats)
    int lda, ldb, ldo, n_m_tiles, n_t_tiles, Kc, n_units, G;
    int whole;                         // ranges cut at tile boundaries (no partial tiles, no slabs): when every workgroup gets several tiles
};

template <int M> struct Wn;
template <> struct Wn<2> { static constexpr int A = 4, P = 16, WT = 256, TPB = 1024, LDS = 5632; };
template <> struct Wn<4> { static constexpr int A = 6, P = 36, WT = 128, TPB = 512, LDS = 9472; };

// ---- the 1-D transforms (Lavin & Gray's matrices; points 0, +-1, inf for m = 2 and 0, +-1, +-2, inf for m = 4)
template <int M> __device__ __forceinline__ void wn_bt(const float *d, float *v);       // B^T d: (m + 2) -> (m + 2)
template <> __device__ __forceinline__ void wn_bt<2>(const float *d, float *v)
{
    v[0] = d[0] - d[2]; v[1] = d[1] + d[2]; v[2] = d[2] - d[1]; v[3] = d[1] - d[3];
}
template <> __device__ __forceinline__ void wn_bt<4>(const float *d, float *v)
{
    v[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
    v[1] = (d[3] + d[4]) - 4.0f * (d[1] + d[2]);
    v[2] = 4.0f * (d[1] - d[2]) + (d[4] - d[3]);
    v[3] = 2.0f * (d[3] - d[1]) + (d[4] - d[2]);
    v[4] = 2.0f * (d[1] - d[3]) + (d[4] - d[2]);
    v[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
template <int M> __device__ __forceinline__ void wn_g(const float *g, float *u);        // G g: 3 -> (m + 2)
template <> __device__ __forceinline__ void wn_g<2>(const float *g, float *u)
{
    u[0] = g[0]; u[1] = (g[0] + g[1] + g[2]) * 0.5f; u[2] = (g[0] - g[1] + g[2]) * 0.5f; u[3] = g[2];
}
template <> __device__ __forceinline__ void wn_g<4>(const float *g, float *u)
{
    u[0] = g[0] * 0.25f;
    u[1] = (g[0] + g[1] + g[2]) * (-1.0f / 6.0f);
    u[2] = (g[0] - g[1] + g[2]) * (-1.0f / 6.0f);
    u[3] = g[0] * (1.0f / 24.0f) + g[1] * (1.0f / 12.0f) + g[2] * (1.0f / 6.0f);
    u[4] = g[0] * (1.0f / 24.0f) - g[1] * (1.0f / 12.0f) + g[2] * (1.0f / 6.0f);
    u[5] = g[2];
}
template <int M> __device__ __forceinline__ void wn_at(const float *m, float *y);       // A^T m: (m + 2) -> m
template <> __device__ __forceinline__ void wn_at<2>(const float *m, float *y)
{
    y[0] = m[0] + m[1] + m[2]; y[1] = m[1] - m[2] - m[3];
}
template <> __device__ __forceinline__ void wn_at<4>(const float *m, float *y)
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.0f * d34;
    y[2] = s12 + 4.0f * s34;
    y[3] = d12 + 8.0f * d34 + m[5];
}
template <int M> __device__ __forceinline__ void wn_a(const float *y, float *r);        // A y: m -> (m + 2)   (the transpose of A^T: weight gradient)
template <> __device__ __forceinline__ void wn_a<2>(const float *y, float *r)
{
    r[0] = y[0]; r[1] = y[0] + y[1]; r[2] = y[0] - y[1]; r[3] = -y[1];
}
template <> __device__ __forceinline__ void wn_a<4>(const float *y, float *r)
{
    const float e = y[0] + y[2], o = y[1] + y[3], e4 = y[0] + 4.0f * y[2], o4 = 2.0f * y[1] + 8.0f * y[3];
    r[0] = y[0]; r[1] = e + o; r[2] = e - o; r[3] = e4 + o4; r[4] = e4 - o4; r[5] = y[3];
}
template <int M> __device__ __forceinline__ void wn_gt(const float *u, float *w);       // G^T u: (m + 2) -> 3
template <> __device__ __forceinline__ void wn_gt<2>(const float *u, float *w)
{
    w[0] = u[0] + (u[1] + u[2]) * 0.5f; w[1] = (u[1] - u[2]) * 0.5f; w[2] = (u[1] + u[2]) * 0.5f + u[3];
}
template <> __device__ __forceinline__ void wn_gt<4>(const float *u, float *w)
{
    const float s12 = u[1] + u[2], d12 = u[2] - u[1], s34 = u[3] + u[4], d34 = u[3] - u[4];
    w[0] = u[0] * 0.25f - s12 * (1.0f / 6.0f) + s34 * (1.0f / 24.0f);
    w[1] = d12 * (1.0f / 6.0f) + d34 * (1.0f / 12.0f);
    w[2] = s34 * (1.0f / 6.0f) - s12 * (1.0f / 6.0f) + u[5];
}

// 16 x 16 (m, k) blocks: the data gradient reads W[k][m] (m contiguous) and writes U[m][k] (k contiguous), so its block goes through LDS
// (the plain one-thread-per-element form read 36-byte pieces 18 KB apart: 15 us at C = 512)
// both = 1: ONE launch makes the forward's U (blocks [0, n)) and the data gradient's rotated U (blocks [n, 2n), into U2) -- a forward that will be
// followed by a backward keeps the second for it (frcnn_conv3x3_f32_fwd's u_rotated) and the backward starts one launch shorter
template <int M>
__global__ __launch_bounds__(256) void rpn_wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, float *__restrict__ U2, int Cout, int Cin,
                                                              int tr_only)
{
    // w [Cout][Cin][9].  forward: m = co, k = ci;  data gradient (TR): m = ci, k = co
    constexpr int A = Wn<M>::A;
    __shared__ float s[16][16 * 9 + 1];
    const unsigned nblk = (unsigned)(Cout / 16) * (unsigned)(Cin / 16);
    const bool TR = tr_only || blockIdx.x >= nblk;                        // uniform per block
    const unsigned b = blockIdx.x >= nblk ? blockIdx.x - nblk : blockIdx.x;
    const int Mo = TR ? Cin : Cout, K = TR ? Cout : Cin;
    if (blockIdx.x >= nblk) U = U2;
    const unsigned nb = (unsigned)K / 16u, m0 = (b / nb) * 16u, k0 = (b % nb) * 16u, row = (unsigned)Cin * 9u;
#pragma unroll
    for (unsigned q = 0; q < 9; ++q) {
        const unsigned e = threadIdx.x + 256u * q, r = e / 144u, c = e - r * 144u;
        s[r][c] = TR ? w[(k0 + r) * row + m0 * 9u + c] : w[(m0 + r) * row + k0 * 9u + c];     // rows: k (TR) or m
    }
    __syncthreads();
    const unsigned ki = threadIdx.x >> 4, mi = threadIdx.x & 15u;    // m fastest: U is stored [xi][k][m] (the GEMM's A tile = 32 k rows x 128 m)
    float g[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) g[e] = TR ? s[ki][mi * 9 + (8 - e)] : s[mi][ki * 9 + e];       // data gradient: W[k][m] rotated by 180 degrees
    float t[A][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float col[3] = {g[c], g[3 + c], g[6 + c]};
        float u[A];
        wn_g<M>(col, u);
#pragma unroll
        for (int r = 0; r < A; ++r) t[r][c] = u[r];
    }
    const size_t n = (size_t)Mo * K, o = (size_t)(k0 + ki) * Mo + m0 + mi;
#pragma unroll
    for (int r = 0; r < A; ++r) {
        float u[A];
        wn_g<M>(t[r], u);
#pragma unroll
        for (int q = 0; q < A; ++q) U[(size_t)(r * A + q) * n + o] = u[q];
    }
}

// Staging loads of the input-side transforms.  The loads are inline assembly, unconditional, at clamped addresses (the window's zeros are a select
// behind them), and ONE counted wait follows a whole batch.  Written in C++, with the bounds tests in front of them or folded into a select right
// behind, the compiler put every load into a basic block of its own (s_and_saveexec, or a branch on the wave-uniform row test) with its
// s_waitcnt vmcnt(0) directly behind it: one load in flight per wave and ~36 instructions per load instruction -- at 64 x 600 x 1000 the staging
// took 82 of the transform's 123 us.
__device__ __forceinline__ float wn_ld(const float *base, unsigned off_bytes)
{
    float t;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(t) : "v"(off_bytes), "s"(base) : "memory");
    return t;
}
// wide windows: a wave per row, two rows x NP 64-column pieces in flight (pieces past the row's end fetch its last element again: one cached line)
template <int NP>
__device__ __forceinline__ void wn_stage_wide(float *s, const float *x, int x0, int y0, int nrow, int ncol, int Hs, int Ws, int lane, int wave)
{
    unsigned off[NP];
    bool in[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int xx = x0 + lane + 64 * j;
        off[j] = 4u * (unsigned)min(max(xx, 0), Ws - 1);
        in[j] = (unsigned)xx < (unsigned)Ws;
    }
    for (int r0 = wave; r0 < nrow; r0 += 8) {
        float v[2][NP];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *xr = x + (size_t)min(max(y0 + r0 + 4 * h, 0), Hs - 1) * Ws;             // wave-uniform
#pragma unroll
            for (int j = 0; j < NP; ++j) v[h][j] = wn_ld(xr, off[j]);
        }
        if constexpr (NP == 9)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[0][4]), "+v"(v[0][5]), "+v"(v[0][6]), "+v"(v[0][7]),
                         "+v"(v[0][8]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[1][4]), "+v"(v[1][5]), "+v"(v[1][6]), "+v"(v[1][7]), "+v"(v[1][8]) :: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[0][4]),
                         "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[1][4]) :: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + 4 * h;
            const bool row_in = (unsigned)(y0 + r) < (unsigned)Hs;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (r < nrow && lane + 64 * j < ncol) s[r * ncol + lane + 64 * j] = (row_in && in[j]) ? v[h][j] : 0.0f;
        }
    }
}
// narrow windows flattened: any shape keeps eight loads per lane busy
__device__ __forceinline__ void wn_stage_flat(float *s, const float *x, int x0, int y0, int nrow, int ncol, int Hs, int Ws)
{
    const int n_el = nrow * ncol;
    const float inv_ncol = 1.0f / (float)ncol;
    for (int base = threadIdx.x; base < n_el; base += 256 * 8) {
        float v[8];
        bool in[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = min(base + 256 * j, n_el - 1);
            int r = (int)((float)e * inv_ncol);                               // e / ncol, fixed up below (e < 2^14: the estimate is off by one at most)
            r -= (r * ncol > e) ? 1 : 0;
            r += ((r + 1) * ncol <= e) ? 1 : 0;
            const int q = e - r * ncol, yy = y0 + r, xx = x0 + q;
            in[j] = (unsigned)yy < (unsigned)Hs && (unsigned)xx < (unsigned)Ws;
            v[j] = wn_ld(x, 4u * (unsigned)(min(max(yy, 0), Hs - 1) * Ws + min(max(xx, 0), Ws - 1)));
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (base + 256 * j < n_el) s[base + 256 * j] = in[j] ? v[j] : 0.0f;
    }
}

// The input-side transforms.  Block = (channel, strip); a strip = up to TPB tiles: tile rows x a piece of at most WT tile columns, as many
// whole rows as fit.  The strip's input window is staged in LDS by coalesced loads -- every element is fetched once (plus the halo rows
// that neighbouring strips share); read straight from global memory each patch cost (m + 2)^2 loads of a stride-m pattern that touched
// every line several times, and the transform ran at 2.9 TB/s (load-path bound: more loads in flight per thread made it slower) -- and
// each thread then reads its patches as 8-byte LDS reads.  The ReLU mask of a gradient is applied while staging.
//   MODE 0: V  = B^T d B of the (m + 2)^2 patch with its one-pixel halo (activations; the masked output gradient for the data gradient)
//   MODE 1: dM = A g A^T of the m x m tile of the (masked) output gradient (weight gradient)
//   MODE 2: both of the gradient's transforms from one staging pass: V as MODE 0 and dM (to a.out2) as MODE 1
// Padding columns of a level (tiles past its last) are written as zeros only when a.zero_pad is set (weight gradient: the product sums
// over the tiles); forward / data gradient never read the product's columns there, and a column depends on the same column of V only.
struct WnStrips { int first[FRCNN_MAX_LEVELS + 1]; int segs[FRCNN_MAX_LEVELS]; int rows[FRCNN_MAX_LEVELS]; };   // first strip, strips per tile row, tile rows per strip
// POOL (m = 4, gradients only): the forward fused ReLU + max_pool2d(2, 2) into its output transform (rpn_wino_output_kernel<4, true>), so the
// incoming gradient is at the POOLED resolution and the words carry, per 2 x 2 window of the tile, the position of the maximum and whether
// it was positive (3 bits each): the window staged is the pooled one (half the rows and columns), and a thread rebuilds its full-resolution
// patch from it -- max_pool2d's and the ReLU's backward without the full-resolution gradient ever existing in memory.
template <int M, int MODE, bool POOL>
__global__ __launch_bounds__(256) void rpn_wino_input_kernel(WnArgs a, WnStrips st, float *__restrict__ V)
{
    static_assert(!POOL || M == 4, "pooled gradients: 4 x 4 tiles only");
    constexpr int A = Wn<M>::A, P = Wn<M>::P, WT = Wn<M>::WT, HALO = MODE == 1 ? 0 : 1, IN = MODE == 1 ? M : A;
    constexpr int SM = POOL ? 2 : M, SIN = POOL ? (MODE == 1 ? 2 : 4) : IN;       // staged pixels per tile side; staged patch side
    extern __shared__ __attribute__((aligned(8))) float s[];          // the largest window of the launch's levels (wn_strips): <= Wn<M>::LDS floats
    const int c = blockIdx.y;
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < a.n_levels && (int)blockIdx.x >= st.first[q]) ? 1 : 0;
    const int strip = blockIdx.x - st.first[l], segs = st.segs[l], R = st.rows[l];
    const int H = a.lv[l].H, W = a.lv[l].W, tw = a.lv[l].tw, th = (H + M - 1) / M;
    const int ty0 = (strip / segs) * R, tx0 = (strip % segs) * WT, wt = min(WT, tw - tx0), nr = min(R, th - ty0);
    const int Hs = POOL ? H >> 1 : H, Ws = POOL ? W >> 1 : W;              // the staged tensor's size (pooled: floor, like max_pool2d)
    const float *x = a.lv[l].x + (size_t)c * Hs * Ws;
    const unsigned short *mk = a.bits ? a.bits + (size_t)c * a.Ttot + a.lv[l].off : nullptr;    // this channel's sign words of the level, [ty * tw + tx]
    const int x0 = SM * tx0 - HALO, y0 = SM * ty0 - HALO, ncol = SM * wt + 2 * HALO, nrow = SM * nr + 2 * HALO;      // the strip's input window
    // the sign words of the strip's tiles and of the ring of tiles around it (MODE 0's halo pixels belong to those): (nr + 2) x (wt + 2) words in
    // LDS; a thread then masks its patch with nine (MODE 0) or one (MODE 1) of them at compile-time bit positions.  (Looked up per staged
    // element -- one 2-byte load each -- the mask cost as much as the 4-byte float mask it replaced: the staging is load-instruction bound.)
    __shared__ unsigned short wb[1664];
    if (mk) {
        const int nwc = wt + 2;
        for (int e = threadIdx.x; e < (nr + 2) * nwc; e += 256) {
            const int i = e / nwc, j = e - i * nwc, ty = ty0 - 1 + i, tx = tx0 - 1 + j;
            wb[e] = (ty >= 0 && ty < th && tx >= 0 && tx < tw) ? mk[ty * tw + tx] : (unsigned short)0;
        }
    }
    // Staging: every load of a batch is issued before the first one is used (wn_stage_wide / wn_stage_flat).
    {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (ncol > 320) wn_stage_wide<9>(s, x, x0, y0, nrow, ncol, Hs, Ws, lane, wave);            // <= 514 = m WT + 2 columns
        else if (ncol >= 192) wn_stage_wide<5>(s, x, x0, y0, nrow, ncol, Hs, Ws, lane, wave);
        else wn_stage_flat(s, x, x0, y0, nrow, ncol, Hs, Ws);
    }
    __syncthreads();
    const size_t plane = (size_t)a.C * a.Ttot, col0 = (size_t)c * a.Ttot + a.lv[l].off;
    const int n_tiles = nr * wt;
    float bsum = 0.0f;                                                    // this thread's share of the strip's gradient sum (MODE 1 with a.db_part)
#pragma unroll 1
    for (int u = 0; u < Wn<M>::TPB / 256; ++u) {
        const int e = u * 256 + (int)threadIdx.x;
        if (e >= n_tiles) break;
        const int rr = e / wt, txl = e - rr * wt;
        float d[IN][IN], sp[SIN][SIN];
#pragma unroll
        for (int r = 0; r < SIN; ++r) {
            const float *row = &s[(SM * rr + r) * ncol + SM * txl];        // even offset: ncol and SM are even
#pragma unroll
            for (int q = 0; q < SIN; q += 2) {
                const float2 p = *(const float2 *)(row + q);
                sp[r][q] = p.x; sp[r][q + 1] = p.y;
            }
        }
        if (POOL) {
            // full-resolution pixel (r, q) of the patch = in-tile pixel (iy, ix) of tile (tr, tc) of the 3 x 3 neighbourhood; it gets the pooled
            // gradient of its window iff the window's word says "maximum here, and positive"
            const unsigned short *wc = &wb[(rr + 1) * (wt + 2) + txl + 1];
            unsigned w9[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) w9[i][j] = (MODE == 1 && (i != 1 || j != 1)) ? 0u : (unsigned)wc[(i - 1) * (wt + 2) + (j - 1)];
#pragma unroll
            for (int r = 0; r < IN; ++r)
#pragma unroll
                for (int q = 0; q < IN; ++q) {
                    const int fr = MODE != 1 ? r - 1 : r, fq = MODE != 1 ? q - 1 : q;                 // row / column relative to the tile's first
                    const int tr = fr < 0 ? 0 : (fr < 4 ? 1 : 2), iy = fr < 0 ? 3 : (fr < 4 ? fr : 0);
                    const int tc = fq < 0 ? 0 : (fq < 4 ? 1 : 2), ix = fq < 0 ? 3 : (fq < 4 ? fq : 0);
                    const int k = (iy >> 1) * 2 + (ix >> 1), pos = (iy & 1) * 2 + (ix & 1);
                    const unsigned w3 = (w9[tr][tc] >> (3 * k)) & 7u;
                    const float g = sp[MODE != 1 ? (r + 1) >> 1 : r >> 1][MODE != 1 ? (q + 1) >> 1 : q >> 1];
                    d[r][q] = (w3 == (4u | (unsigned)pos)) ? g : 0.0f;
                }
        } else {
#pragma unroll
            for (int r = 0; r < IN; ++r)
#pragma unroll
                for (int q = 0; q < IN; ++q) d[r][q] = sp[r][q];
        }
        if (!POOL && mk) {
            const unsigned short *wc = &wb[(rr + 1) * (wt + 2) + txl + 1];      // this tile's word; its neighbours at +-1 and +-(wt + 2)
            if (MODE == 1) {
                const unsigned w0 = wc[0];
#pragma unroll
                for (int r = 0; r < IN; ++r)
#pragma unroll
                    for (int q = 0; q < IN; ++q) d[r][q] = ((w0 >> (r * M + q)) & 1u) ? d[r][q] : 0.0f;
            } else {
                unsigned w9[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) w9[i][j] = wc[(i - 1) * (wt + 2) + (j - 1)];
#pragma unroll
                for (int r = 0; r < IN; ++r)
#pragma unroll
                    for (int q = 0; q < IN; ++q) {
                        const int tr = r == 0 ? 0 : (r <= M ? 1 : 2), br = r == 0 ? M - 1 : (r <= M ? r - 1 : 0);      // patch row r = image row m ty - 1 + r
                        const int tc = q == 0 ? 0 : (q <= M ? 1 : 2), bc = q == 0 ? M - 1 : (q <= M ? q - 1 : 0);
                        d[r][q] = ((w9[tr][tc] >> (br * M + bc)) & 1u) ? d[r][q] : 0.0f;
                    }
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < IN; ++r)
#pragma unroll
                for (int q = 0; q < IN; ++q) bsum += d[r][q];               // the tile's share of the bias gradient (masked values, every position once)
        }
        float wv[A][IN];                                                  // columns first: wv[.][q] = T d[.][q]
#pragma unroll
        for (int q = 0; q < IN; ++q) {
            float col[IN], o[A];
#pragma unroll
            for (int r = 0; r < IN; ++r) col[r] = d[r][q];
            if (MODE != 1) wn_bt<M>(col, o); else wn_a<M>(col, o);
#pragma unroll
            for (int r = 0; r < A; ++r) wv[r][q] = o[r];
        }
        const size_t at = col0 + (size_t)(ty0 + rr) * tw + tx0 + txl;
#pragma unroll
        for (int r = 0; r < A; ++r) {
            float o[A];
            if (MODE != 1) wn_bt<M>(wv[r], o); else wn_a<M>(wv[r], o);
#pragma unroll
#if defined(WN_TABL) && (WN_TABL & 2)
            for (int q = 0; q < A; ++q) if (o[q] == 1.2345e-30f) V[(size_t)(r * A + q) * plane + at] = o[q];
#else
            for (int q = 0; q < A; ++q) V[(size_t)(r * A + q) * plane + at] = o[q];      // (as non-temporal stores: 53 -> 93 us at 128 x 300 x 500)
#endif
        }
        if (MODE == 2) {
            // the same staged gradient's SECOND transform: A g A^T of the tile's own m x m pixels (the patch without its halo), for the weight gradient's
            // product -- one pass over the gradient and its words serves both of its consumers
            float w2[A][M];
#pragma unroll
            for (int r = 0; r < M; ++r)
#pragma unroll
                for (int q = 0; q < M; ++q) bsum += d[r + 1][q + 1];        // the order of MODE 1's sum: the bias gradient does not depend on the form
#pragma unroll
            for (int q = 0; q < M; ++q) {
                float col[M], o[A];
#pragma unroll
                for (int r = 0; r < M; ++r) col[r] = d[r + 1][q + 1];
                wn_a<M>(col, o);
#pragma unroll
                for (int r = 0; r < A; ++r) w2[r][q] = o[r];
            }
#pragma unroll
            for (int r = 0; r < A; ++r) {
                float o[A];
                wn_a<M>(w2[r], o);
#pragma unroll
                for (int q = 0; q < A; ++q) a.out2[(size_t)(r * A + q) * plane + at] = o[q];
            }
        }
    }
    // bias gradient (MODE 1 with a.db_part): the windows have no halo, so the strips of a channel partition its positions: strip sums -> partials
    // [channel][strip], which the last launch of the weight gradient (rpn_wino_dw_kernel) adds in strip order.  (A ticket + last-arriver
    // sum inside this launch held every block's resources for the round trips of its thread 0: +30 % on the whole transform.)
    if (MODE != 0 && a.db_part) {
        __shared__ float red[4];
#pragma unroll
        for (int h = 32; h > 0; h >>= 1) bsum += __shfl_down(bsum, h, 64);   // wave sums by lane shuffles (fixed tree), then four values through LDS
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bsum;
        __syncthreads();
        if (threadIdx.x == 0) a.db_part[(size_t)c * st.first[FRCNN_MAX_LEVELS] + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    if (a.zero_pad && strip == 0) {                                       // this level's padding columns of channel c, all planes
        const int T = a.lv[l].T, Tp = (T + a.tg - 1) / a.tg * a.tg, np = Tp - T;
        for (int e = threadIdx.x; e < P * np; e += 256) {
            const int xi = e / np, q = e - xi * np;
            (MODE == 2 ? a.out2 : V)[(size_t)xi * plane + col0 + T + q] = 0.0f;
        }
    }
}

// POOL (m = 4): ReLU + max_pool2d(2, 2) in the same pass -- the tile's four 2 x 2 windows -> the pooled outputs [C][H/2][W/2] (floor) and, per window,
// 3 bits of the word: the position of the (first) maximum and whether it was positive; the full-resolution activations are never written.
// One (channel, tile) of the output side: the tile's P products mv[(r, q)] -> A^T M A (+ bias, ReLU / ReLU + 2 x 2 max-pool), the store and the sign /
// window word.  Shared by rpn_wino_output_kernel (products from memory) and rpn_wino_gemm_out64_kernel (products still in the accumulators).
template <int M, bool POOL>
__device__ __forceinline__ void wn_output_tile(const WnArgs &a, int c, int t, const float (&mv)[Wn<M>::P])
{
    constexpr int A = Wn<M>::A;
    int l = 0;
#pragma unroll
    for (int q = 1; q < FRCNN_MAX_LEVELS; ++q) l += (q < a.n_levels && t >= a.lv[q].off) ? 1 : 0;
    const int tl = t - a.lv[l].off;
    if (tl >= a.lv[l].T) return;
    const size_t at = (size_t)c * a.Ttot + t;
    float s[M][A];                                                        // rows first: s[i][q] = (A^T m[.][q])_i
#pragma unroll
    for (int q = 0; q < A; ++q) {
        float col[A], o[M];
#pragma unroll
        for (int r = 0; r < A; ++r) col[r] = mv[r * A + q];
        wn_at<M>(col, o);
#pragma unroll
        for (int i = 0; i < M; ++i) s[i][q] = o[i];
    }
    const float b = a.bias ? a.bias[c] : 0.0f;
    const int H = a.lv[l].H, W = a.lv[l].W, ty = tl / a.lv[l].tw, tx = tl - ty * a.lv[l].tw;
    unsigned word = 0;
    if (POOL) {
        float v[M][M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            float o[M];
            wn_at<M>(s[i], o);
#pragma unroll
            for (int j = 0; j < M; ++j) { const float t = o[j] + b; v[i][j] = t < 0.0f ? 0.0f : t; }      // keeps a NaN, as torch's relu does (fmaxf drops it)
        }
        const int Hp = H >> 1, Wp = W >> 1;
        float *yp = a.lv[l].y + (size_t)c * Hp * Wp;
#pragma unroll
        for (int wi = 0; wi < 2; ++wi)
#pragma unroll
            for (int wj = 0; wj < 2; ++wj) {
                const int py = 2 * ty + wi, px = 2 * tx + wj;
                if (py < Hp && px < Wp) {                                     // floor mode: a window exists only with all four of its pixels
                    float mx = v[2 * wi][2 * wj];
                    unsigned arg = 0;
                    if (v[2 * wi][2 * wj + 1] > mx || v[2 * wi][2 * wj + 1] != v[2 * wi][2 * wj + 1]) { mx = v[2 * wi][2 * wj + 1]; arg = 1; }      // scan order, strict >: the first maximum, as max_pool2d (which also lets a NaN win)
                    if (v[2 * wi + 1][2 * wj] > mx || v[2 * wi + 1][2 * wj] != v[2 * wi + 1][2 * wj]) { mx = v[2 * wi + 1][2 * wj]; arg = 2; }
                    if (v[2 * wi + 1][2 * wj + 1] > mx || v[2 * wi + 1][2 * wj + 1] != v[2 * wi + 1][2 * wj + 1]) { mx = v[2 * wi + 1][2 * wj + 1]; arg = 3; }
                    yp[(size_t)py * Wp + px] = mx;
                    word |= (arg | (mx > 0.0f ? 4u : 0u)) << (3 * (wi * 2 + wj));
                }
            }
        if (a.bits_out) a.bits_out[at] = (unsigned short)word;
        return;
    }
    float *y = a.lv[l].y + (size_t)c * H * W + (size_t)(M * ty) * W + M * tx;
    // a tile row as ONE store where the map's width allows (W a multiple of m: every tile is whole and every row piece m-float aligned) -- as
    // four dword stores, each wave-instruction wrote every fourth dword of a 1-KB span and the 154 MB of a 64 x 600 x 1000 map left at 2.5 TB/s
    const bool vec = (W % M) == 0 && ((size_t)a.lv[l].y & (4 * M - 1)) == 0, pairs = (W & 1) == 0 && ((size_t)a.lv[l].y & 7) == 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        float o[M];
        wn_at<M>(s[i], o);
        if (M * ty + i < H) {
            float v[M];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                v[j] = o[j] + b;
                if (a.relu) v[j] = v[j] < 0.0f ? 0.0f : v[j];                      // NaN stays NaN (torch.relu / clamp_min)
                if (M * tx + j < W) word |= (v[j] > 0.0f) ? (1u << (i * M + j)) : 0u;
            }
            if (vec) {
                if (M == 4) *(float4 *)(y + (size_t)i * W) = make_float4(v[0], v[1], v[2], v[3]);
                else *(float2 *)(y + (size_t)i * W) = make_float2(v[0], v[1]);
            } else if (M == 4 && pairs) {                                 // an even width: pairs stay whole and 8-byte aligned
#pragma unroll
                for (int j = 0; j < M; j += 2)
                    if (M * tx + j < W) *(float2 *)(y + (size_t)i * W + j) = make_float2(v[j], v[j + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < M; ++j)
                    if (M * tx + j < W) y[(size_t)i * W + j] = v[j];
            }
        }
    }
    if (a.bits_out) a.bits_out[at] = (unsigned short)word;
}

template <int M, bool POOL>
__global__ __launch_bounds__(256) void rpn_wino_output_kernel(WnArgs a, const float *__restrict__ Mp)
{
    static_assert(!POOL || M == 4, "fused pooling: 4 x 4 tiles only");
    constexpr int A = Wn<M>::A, P = Wn<M>::P;
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (t >= a.Ttot) return;
    const size_t plane = (size_t)a.C * a.Ttot, at = (size_t)c * a.Ttot + t;
    float mv[P];
#pragma unroll
    for (int q = 0; q < A; ++q)                                           // (the order of the loads the kernel always had: by column of the plane grid)
#pragma unroll
        for (int r = 0; r < A; ++r) {
#if defined(WN_TABL) && (WN_TABL & 4)
            mv[r * A + q] = (float)(r + q); asm volatile("" : "+v"(mv[r * A + q]));
#else
            mv[r * A + q] = Mp[(size_t)(r * A + q) * plane + at];
#endif
        }
    wn_output_tile<M, POOL>(a, c, t, mv);
}

// dW[co][ci] = G^T dU G: (m + 2)^2 -> 3 x 3, thread = (co, ci); n = Cout * Cin
// blocks past the (co, ci) range: the bias gradient, one block per channel adds the output-gradient transform's strip partials in strip order
template <int M>
__global__ __launch_bounds__(256) void rpn_wino_dw_kernel(const float *__restrict__ dU, float *__restrict__ dw, unsigned n, float *__restrict__ db,
                                                          const float *__restrict__ db_part, int n_strips)
{
    constexpr int A = Wn<M>::A;
    const unsigned dw_blocks = (n + 255u) / 256u;
    if (blockIdx.x >= dw_blocks) {
        __shared__ float red[256];
        const unsigned c = blockIdx.x - dw_blocks;
        float t = 0.0f;
        for (int i = threadIdx.x; i < n_strips; i += 256) t += db_part[(size_t)c * n_strips + i];      // fixed assignment and order: bit-reproducible
        red[threadIdx.x] = t;
        __syncthreads();
#pragma unroll
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) db[c] = red[0];
        return;
    }
    const unsigned o = blockIdx.x * 256u + threadIdx.x;
    if (o >= n) return;
    float t[3][A];                                                        // t[i][q] = (G^T u[.][q])_i
#pragma unroll
    for (int q = 0; q < A; ++q) {
        float col[A], w3[3];
#pragma unroll
        for (int r = 0; r < A; ++r) col[r] = dU[(size_t)(r * A + q) * n + o];
        wn_gt<M>(col, w3);
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i][q] = w3[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float w3[3];
        wn_gt<M>(t[i], w3);
#pragma unroll
        for (int j = 0; j < 3; ++j) dw[(size_t)o * 9 + i * 3 + j] = w3[j];
    }
}

#ifdef WN_STAMP                                                      // developer build (tools/dev/wn_stamps.py): wall-clock stamps (100 MHz) of every workgroup of the last launch
__device__ unsigned long long wn_stamps[4 * 1024];
extern "C" __attribute__((visibility("default"))) int frcnn_debug_wn_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wn_stamps), sizeof(wn_stamps));
}
#define WN_STAMP_AT(i) do { if (threadIdx.x == 0) wn_stamps[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
#else
#define WN_STAMP_AT(i) do { } while (0)
#endif
// MT x NW = the workgroup's output tile (128 or 64 each way: the 64-channel layers of a backbone have a 64-wide side); its four waves sit
// 2 x 2, a wave owns (MT / 2) x (NW / 2) = MI x NI MFMA tiles of 32 x 32.
// SPLIT (round 5; opt-in, frcnn_conv3x3_f32_products): the SAME fp32 operands from the SAME LDS image, but the product on the bf16 matrix cores -- every
// value is cut into three bf16 pieces in registers (v = h + m + l EXACTLY: h = v's upper 16 bits, the residuals are exact fp32 differences), and a tile's
// 16 k rows are six v_mfma_f32_32x32x16_bf16 (l h, h l, m m, m h, h m, h h; the three dropped products are each <= 2^-24 of the term) = 192 matrix-pipe
// cycles where the fp32 instruction needs 8 x 64.  Against float64 the result is as close as the fp32 instruction's (tools/dev/micro/split_product_check.hip,
// profiles/r05_split_product_check.txt: rms error / sum |a b| 1.9e-8 vs 2.4e-8 at K = 256 ... 4096 on mixed-sign data).
typedef short wn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wn_cut8(const float *v, wn_u32x4 &h, wn_u32x4 &m, wn_u32x4 &l)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned x0 = __builtin_bit_cast(unsigned, v[2 * p]), x1 = __builtin_bit_cast(unsigned, v[2 * p + 1]);
        h[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);                // bf16 element 2 p = upper half of x0, 2 p + 1 = upper half of x1
        const float ra = v[2 * p] - __builtin_bit_cast(float, x0 & 0xFFFF0000u), rb = v[2 * p + 1] - __builtin_bit_cast(float, x1 & 0xFFFF0000u);
        const unsigned r0 = __builtin_bit_cast(unsigned, ra), r1 = __builtin_bit_cast(unsigned, rb);
        m[p] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
        const float qa = ra - __builtin_bit_cast(float, r0 & 0xFFFF0000u), qb = rb - __builtin_bit_cast(float, r1 & 0xFFFF0000u);
        l[p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, qb), __builtin_bit_cast(unsigned, qa), 0x07060302u);
    }
}
template <bool NT, int MT, int NW, bool SPLIT>
__global__ __launch_bounds__(256, CF_WPS) void rpn_wino_gemm_kernel(WgArgs a, float *__restrict__ part, int *__restrict__ cnt)
{
    WN_STAMP_AT(0);
    constexpr int MI = MT / 64, NI = NW / 64, NG = MI * NI;          // MFMA tiles per wave
    constexpr int TA = MT / 8, TB = NW / 8;                          // 1-KB DMA transfers per chunk and operand (32 x MT x 4 bytes)
    // <false>: both operand tiles are 32 k rows x MT (NW) floats, row-contiguous: the image global_load_lds_dwordx4 writes (wave base + lane x 16 bytes)
    // <true>:  MT (NW) rows x 32 k (k contiguous in memory): row r's eight 16-byte pieces sit at slots p ^ ((r >> 1) & 7) of its 128 LDS bytes -- the
    //          DMA's LDS image is lane-linear, but which global piece a lane fetches is free -- so that the ds_read_b128 of 16 different rows
    //          (one lane group) hit 16 different bank quads
    __shared__ __attribute__((aligned(16))) float sA[2][WN_KC * MT];
    __shared__ __attribute__((aligned(16))) float sB[2][WN_KC * NW];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int G = a.G, U_ = a.n_units, Kc = a.Kc, lda = a.lda, ldb = a.ldb, ldo = a.ldo;
    const int sigma = (G % 8 == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int u0 = a.whole ? (int)cf_start(sigma, U_ / Kc, G) * Kc : (int)cf_start(sigma, U_, G);
    const int u1 = a.whole ? (int)cf_start(sigma + 1, U_ / Kc, G) * Kc : (int)cf_start(sigma + 1, U_, G);
    if (u0 >= u1) return;
    // tile -> (xi, t tile, m tile): the m tiles of one (xi, t tile) are neighbours (they share the V rows)
    struct Tl { const float *u, *v; float *o; int tile; };
    auto tile_of = [&](int t) {
        const int mt = t % a.n_m_tiles, r = t / a.n_m_tiles, tt = r % a.n_t_tiles, xi = r / a.n_t_tiles;
        Tl T;
        if (NT) {
            T.u = a.A + (size_t)xi * a.sA + (size_t)mt * MT * lda;   // this tile's MT rows, k from 0
            T.v = a.B + (size_t)xi * a.sB + (size_t)tt * NW * ldb;
        } else {
            T.u = a.A + (size_t)xi * a.sA + (size_t)mt * MT;         // rows k, this tile's MT columns
            T.v = a.B + (size_t)xi * a.sB + (size_t)tt * NW;
        }
        T.o = a.O + (size_t)xi * a.sO + (size_t)mt * MT * ldo + (size_t)tt * NW;
        T.tile = t;
        return T;
    };
    // staging by LDS-DMA: a chunk = TA + TB wave transfers of 1 KB, a quarter of them per wave; no staging registers, no ds_write, and the
    // transfers of chunk u + 1 are in flight during all sixteen k steps of chunk u.  (With the chunk staged through registers -- 8 x 16
    // bytes per thread, written to LDS behind the last step -- the GEMM took 533 us at FPN size and 70 at 600x1000; without any staging
    // the same loop takes 366.)
    auto dma16 = [&](const float *g, const float *lds) {
        const unsigned l = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds;
        // (non-temporal where both operands are streamed once from HBM -- the k-contiguous form on 64 x 64 tiles, conv1_2's weight gradient: 211 -> 176 us;
        // on the other shapes the hint cost the product 3-8 %)
        if constexpr (NT && MT == 64 && NW == 64)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(g) : "memory");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(g) : "memory");
    };
    // one wave transfer of a chunk's staging: piece q < (TA + TB) / 4 of this wave (the first TA / 4 belong to the A tile)
    auto issue_piece = [&](const Tl &T, int chunk, int buf, int q) {
        if (NT) {
            const float *ub = T.u + (size_t)chunk * WN_KC, *vb = T.v + (size_t)chunk * WN_KC;
            const unsigned rl = (unsigned)(lane >> 3), sl = (unsigned)lane & 7u;     // transfer d = rows 8d .. 8d + 7, 128 bytes each
            if (q < TA / 4) {
                const int d = wave * (TA / 4) + q;
                const unsigned r = (unsigned)(8 * d) + rl, p = sl ^ ((r >> 1) & 7u);
                dma16(ub + (size_t)r * lda + 4u * p, &sA[buf][d * 256]);
            } else {
                const int d = wave * (TB / 4) + (q - TA / 4);
                const unsigned r = (unsigned)(8 * d) + rl, p = sl ^ ((r >> 1) & 7u);
                dma16(vb + (size_t)r * ldb + 4u * p, &sB[buf][d * 256]);
            }
            return;
        }
        const float *ub = T.u + (size_t)chunk * WN_KC * lda, *vb = T.v + (size_t)chunk * WN_KC * ldb;
        constexpr int RA = 256 / MT, RB = 256 / NW;                  // k rows per transfer: 2 (128-wide tile) or 4 (64-wide)
        if (q < TA / 4) {
            const int d = wave * (TA / 4) + q;
            dma16(ub + (size_t)(RA * d + lane / (MT / 4)) * lda + (unsigned)(lane % (MT / 4)) * 4u, &sA[buf][d * 256]);
        } else {
            const int d = wave * (TB / 4) + (q - TA / 4);
            dma16(vb + (size_t)(RB * d + lane / (NW / 4)) * ldb + (unsigned)(lane % (NW / 4)) * 4u, &sB[buf][d * 256]);
        }
    };
    constexpr int NPIECE = (TA + TB) / 4;                            // 8 on 128 x 128 tiles, 6 / 4 with a 64-wide side
    auto issue_dma = [&](const Tl &T, int chunk, int buf) {
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) issue_piece(T, chunk, buf, q);
    };
    f32x16 acc[MI][NI];
    f32x16 accs[SPLIT ? MI : 1][SPLIT ? NI : 1];                    // SPLIT: the small products' accumulator (see the main loop)
    auto zero_acc = [&]() {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[mi][ni][r] = 0.0f; if constexpr (SPLIT) accs[mi][ni][r] = 0.0f; }
    };
    zero_acc();
    auto store_tile = [&](const Tl &T) {                             // padded columns: no bounds
        int tt = ldo;
        asm volatile("" : "+s"(tt));                                 // opaque here: otherwise the 64 store offsets are hoisted out of the unit loop
                                                                     // and live in 64 registers through it (the staging registers went to scratch)
        if constexpr (!NT) {
            // K-major form: MFMA tile (mi, ni) of the wave holds rows MI i + mi and columns NI j + ni of its quarter (see the operand reads below), so a
            // lane's NI values of a row are neighbours in memory: one 8-byte store per row instead of two dwords 128 bytes apart
            float *o = T.o + (size_t)(wm * (MT / 2) + MI * 4 * lh) * tt + wn * (NW / 2) + NI * li;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *d = o + (size_t)(MI * ((r & 3) + 8 * (r >> 2)) + mi) * tt;
                    if constexpr (NI == 2) *(float2 *)d = make_float2(acc[mi][0][r], acc[mi][1][r]);
                    else d[0] = acc[mi][0][r];
                }
            return;
        }
        float *o = T.o + (size_t)(wm * (MT / 2) + 4 * lh) * tt + wn * (NW / 2) + li;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    o[(size_t)(mi * 32 + (r & 3) + 8 * (r >> 2)) * tt + ni * 32] = acc[mi][ni][r];
    };
    auto finish_segment = [&](const Tl &T, int first_chunk, int n_chunks) {       // as in rpn_conv3x3_f32_kernel
#if defined(WN_ABL) && (WN_ABL & 1)                                 // developer ablation: no output at all (the accumulators stay live through a test that never holds)
        if (acc[0][0][0] != 1.2345e-30f) return;
#endif
        if (n_chunks == Kc) { store_tile(T); return; }
        float *slab = part + ((size_t)sigma * 2 + (first_chunk == 0 ? 1 : 0)) * CF_SLAB;
        // 16-byte write-through stores, [wave][tile][quad][lane][4] (a dword sc1 store is one fabric write per lane-dword: about six times
        // the time per byte of the 16-byte form, MI355X_MICROARCH.md)
        typedef float f32x4s __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float *dst = &slab[(wave * NG + g) * 16 * 64 + lane * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4s v = {acc[g / NI][g % NI][4 * q], acc[g / NI][g % NI][4 * q + 1], acc[g / NI][g % NI][4 * q + 2], acc[g / NI][g % NI][4 * q + 3]};
                asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc1" :: "v"(dst), "v"(v), "n"(q * 1024) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = (__hip_atomic_fetch_add(&cnt[T.tile], n_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + n_chunks == Kc) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        const long long lo = (long long)T.tile * Kc, hi = lo + Kc - 1;
        int s_first = (int)((lo * G) / U_), s_end = (int)((hi * G) / U_);
        while (cf_start(s_first + 1, U_, G) <= lo) ++s_first;
        while (cf_start(s_first, U_, G) > lo) --s_first;
        while (cf_start(s_end + 1, U_, G) <= hi) ++s_end;
        while (cf_start(s_end, U_, G) > hi) --s_end;
        zero_acc();
        constexpr int PER = NG >= 2 ? 2 : 1;                         // accumulator tiles per batch of loads (at most eight 16-byte loads in flight: the
                                                                     // reducer must not push the main loop's registers out to scratch)
        for (int s = s_first; s <= s_end; ++s) {
            const long long st = cf_start(s, U_, G);
            const float *sl = part + ((size_t)s * 2 + (st <= lo ? 1 : 0)) * CF_SLAB;
#pragma unroll
            for (int hf = 0; hf < NG / PER; ++hf) {
                f32x4s t[PER][4];
#pragma unroll
                for (int g = 0; g < PER; ++g) {
                    const float *src = &sl[(wave * NG + hf * PER + g) * 16 * 64 + lane * 4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(t[g][q]) : "v"(src), "n"(q * 1024) : "memory");
                }
#define WN_WAIT(N, g) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(t[g][0]), "+v"(t[g][1]), "+v"(t[g][2]), "+v"(t[g][3]) :: "memory")
                if (PER == 2) WN_WAIT(4, 0); else WN_WAIT(0, 0);
                {
                    constexpr int g0 = 0;
                    const int gg = hf * PER + g0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[gg / NI][gg % NI][r] += t[g0][r >> 2][r & 3];
                }
                if (PER == 2) {
                    WN_WAIT(0, PER - 1);
                    const int gg = hf * PER + 1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[gg / NI][gg % NI][r] += t[PER - 1][r >> 2][r & 3];
                }
#undef WN_WAIT
            }
        }
        store_tile(T);
        if (tid == 0) __hip_atomic_store(&cnt[T.tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    int tile = u0 / Kc, chunk = u0 - tile * Kc;
    Tl T = tile_of(tile);
    int seg_first = chunk;
    issue_dma(T, chunk, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    WN_STAMP_AT(1);
    for (int u = u0; u < u1; ++u) {
        const int buf = (u - u0) & 1;
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == Kc) { nchunk = 0; ++ntile; }
        Tl Tn = T;
        const bool more = u + 1 < u1;
        if (more && ntile != tile) Tn = tile_of(ntile);
        // The next chunk's transfers (every wave is past the barrier that ended the last reads of that buffer).  K-major form: issued BETWEEN this chunk's k
        // steps, one piece behind each of the first NPIECE steps' MFMAs, instead of in front of the loop with the matrix pipe idle behind the barrier: worth 1-2 %
        // (128 -> 128 on 300 x 500: 129.4 -> 127.9 us; 512 -> 512 on 37 x 62: 42.4 -> 40.3) -- the other workgroup of the CU already filled most of that gap.
        // The k-contiguous form keeps them in front (between its steps: 76.8 -> 81.7 us at 64 -> 128 on 300 x 500).
        constexpr bool between = !NT;
#if !(defined(WN_ABL) && (WN_ABL & 2))                             // developer ablation 2: nothing staged behind the first chunk
        if (more && !between) issue_dma(Tn, nchunk, buf ^ 1);
#define WN_PIECE(step) do { if (between && more && (step) < NPIECE) issue_piece(Tn, nchunk, buf ^ 1, (step)); } while (0)
#else
#define WN_PIECE(step) do { } while (0)
#endif
        if constexpr (SPLIT) {
            // lane (i, h) supplies k = 16 kb + 8 h + j (K-major: eight rows of its column) or 16 h + 8 kb + j (k-contiguous: two 16-byte pieces of its row),
            // j = 0 .. 7, to the MFMAs of block kb -- the bf16 instruction's operand map; both operands take the same k, which is all a contraction asks
            float va[MI][8], vb[NI][8];
            typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int kb = 0; kb < WN_KC / 16; ++kb) {
                if constexpr (NT) {
                    const float *ra = &sA[buf][(wm * (MT / 2) + li) * WN_KC], *rb = &sB[buf][(wn * (NW / 2) + li) * WN_KC];
                    const int fsw = (li >> 1) & 7;
#pragma unroll
                    for (int hj = 0; hj < 2; ++hj) {
                        const int o = (((4 * lh + 2 * kb + hj) ^ fsw) << 2);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) { const f32x4 t = *(const f32x4 *)(ra + mi * 32 * WN_KC + o); va[mi][4 * hj] = t[0]; va[mi][4 * hj + 1] = t[1]; va[mi][4 * hj + 2] = t[2]; va[mi][4 * hj + 3] = t[3]; }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) { const f32x4 t = *(const f32x4 *)(rb + ni * 32 * WN_KC + o); vb[ni][4 * hj] = t[0]; vb[ni][4 * hj + 1] = t[1]; vb[ni][4 * hj + 2] = t[2]; vb[ni][4 * hj + 3] = t[3]; }
                    }
                } else {
                    const float *qa = &sA[buf][(16 * kb + 8 * lh) * MT + wm * (MT / 2) + MI * li];
                    const float *qb = &sB[buf][(16 * kb + 8 * lh) * NW + wn * (NW / 2) + NI * li];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if constexpr (MI == 2) { const float2 t = *(const float2 *)(qa + j * MT); va[0][j] = t.x; va[1][j] = t.y; }
                        else va[0][j] = qa[j * MT];
                        if constexpr (NI == 2) { const float2 t = *(const float2 *)(qb + j * NW); vb[0][j] = t.x; vb[1][j] = t.y; }
                        else vb[0][j] = qb[j * NW];
                    }
                }
                wn_u32x4 ah[MI], am[MI], al[MI], bh[NI], bm[NI], bl[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) wn_cut8(va[mi], ah[mi], am[mi], al[mi]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) wn_cut8(vb[ni], bh[ni], bm[ni], bl[ni]);
#define WN_MM(c, x, y) c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wn_bf16x8, x[mi]), __builtin_bit_cast(wn_bf16x8, y[ni]), c[mi][ni], 0, 0, 0)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        // the five small products go to an accumulator of their own (accs, added to acc when the tile's segment ends): the bf16 instruction aligns
                        // its addends to the largest and truncates DOWNWARDS, so small products added to a large sum left a DC error (-1e-8 of the output's
                        // rms at K = 256, -2e-8 at 512: tools/dev/micro/split_dc_check.hip) that the gradient sums over 600 000 positions multiplied by
                        // sqrt(N); apart, the DC is below the noise and the rms error a third of the fp32 instruction's
                        WN_MM(accs, al, bh); WN_MM(accs, ah, bl); WN_MM(accs, am, bm); WN_MM(accs, am, bh); WN_MM(accs, ah, bm); WN_MM(acc, ah, bh);
                    }
#undef WN_MM
#pragma unroll
                for (int q = kb * (NPIECE / 2); q < (kb + 1) * (NPIECE / 2); ++q) WN_PIECE(q);
            }
        } else if (NT) {
            // lane (i = li, half lh) takes k = 16 lh + s at step s: sixteen consecutive floats of its row = four ds_read_b128 per operand row
            // and chunk (the contraction does not care which k goes to which step as long as both operands agree)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const float *ra = &sA[buf][(wm * (MT / 2) + li) * WN_KC], *rb = &sB[buf][(wn * (NW / 2) + li) * WN_KC];
            const int fsw = (li >> 1) & 7;
            f32x4 fa[2][MI], fb[2][NI];                              // [slot][tile]
            auto fetch4 = [&](int j, int slot) {
                const int o = (((4 * lh + j) ^ fsw) << 2);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[slot][mi] = *(const f32x4 *)(ra + mi * 32 * WN_KC + o);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[slot][ni] = *(const f32x4 *)(rb + ni * 32 * WN_KC + o);
            };
            fetch4(0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = j & 1;
                if (j + 1 < 4) fetch4(j + 1, slot ^ 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][mi][e], fb[slot][ni][e], acc[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    WN_PIECE(4 * j + e);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // MFMA tile mi takes rows MI i + mi of the wave's quarter (i = the lane's row index), tile ni columns NI j + ni: a lane's MI (NI) operand
            // values of a k row are neighbours in LDS -- one ds_read_b64 per operand and step where the wave has two tiles on that side
            const float *pa = &sA[buf][lh * MT + wm * (MT / 2) + MI * li];
            const float *pb = &sB[buf][lh * NW + wn * (NW / 2) + NI * li];
            float oa[2][MI], ob[2][NI];
            auto fetch = [&](int s, int slot) {
#if defined(WN_ABL) && (WN_ABL & 4)                                 // developer ablation 4: no operand reads from LDS
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) { oa[slot][mi] = (float)s; asm volatile("" : "+v"(oa[slot][mi])); }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) { ob[slot][ni] = (float)s; asm volatile("" : "+v"(ob[slot][ni])); }
                return;
#endif
                if constexpr (MI == 2) { const float2 t = *(const float2 *)(pa + 2 * s * MT); oa[slot][0] = t.x; oa[slot][1] = t.y; }
                else oa[slot][0] = pa[2 * s * MT];
                if constexpr (NI == 2) { const float2 t = *(const float2 *)(pb + 2 * s * NW); ob[slot][0] = t.x; ob[slot][1] = t.y; }
                else ob[slot][0] = pb[2 * s * NW];
            };
            fetch(0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int slot = s & 1;
                if (s + 1 < 16) fetch(s + 1, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot][mi], ob[slot][ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                WN_PIECE(s);
            }
        }
#undef WN_PIECE
        if (!more || ntile != tile) {
            if (!more) WN_STAMP_AT(2);
            if constexpr (SPLIT) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += accs[mi][ni];
            }
            finish_segment(T, seg_first, chunk + 1 - seg_first);
            if (!more) WN_STAMP_AT(3);
            zero_acc();
            seg_first = 0;
            if (more) T = Tn;
        }
        tile = ntile; chunk = nchunk;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next chunk's transfers have landed (this wave's; the barrier covers the others')
#if !(defined(WN_ABL) && (WN_ABL & 8))                             // developer ablation 8: no barrier between chunks
        __syncthreads();
#endif
    }
}

// ---- the product with the output transform fused in, for 64 output channels and K = 64 (conv1_2 and its data gradient at 600 x 1000) ----
// There the stage is a byte mover: 0.35 GB of product planes written by the GEMM and read back by the output transform, for 11 GFLOP.  Here a workgroup
// owns WO_TB = 32 consecutive tiles x all 64 channels x ALL 36 planes: 36 x 8 accumulator registers per lane (v_mfma_f32_16x16x4_f32: wave w = channels
// 16 w .. 16 w + 15, two 16-tile column blocks), so after the last plane every lane holds the 36 products of its eight (channel, tile) pairs and runs the
// output transform (wn_output_tile: A^T M A, bias, ReLU / pool, store, sign word) on them in registers -- the product planes never exist in memory.
// Operands per plane (U_xi [K][64], V_xi [K][32 tiles]) by LDS-DMA into a ring of WO_NB plane buffers; rows are rotated at the source (16 floats per k
// row of a 4-row MFMA step for U, 16 per pair of rows for V) so that the four k rows a step reads fall into different banks.
// 213 / 199 us (forward / data gradient) against 182 + 104 / 174 + 105 for the product + output transform pair.  What bounds it is the LDS-DMA path: 864 KB
// of transfers per 32 tiles (two thirds of them U_xi, the same 16 KB for every workgroup) at the ~20 GB/s per CU that path gives = 40 of a block's 45 us,
// of which 17 are MFMAs.  Measured on the way: one plane ahead 312 us (every plane waited for its round trip), ring of 4 / 6 buffers 213 / 225, operand
// reads batched a plane ahead of the MFMAs 223, persistent blocks with the next block's planes issued before the output transform 223 (and 522 when the
// 36 x 4 per-lane U addresses were hoisted out of the block loop into scratch), U_xi as plain register loads two planes ahead 208 (the compiler's own
// counted waits for those loads drained the V transfers with them), K = 128 (conv2_1's data gradient) 129 against 94 unfused: not routed here.
typedef float wn_f32x4 __attribute__((ext_vector_type(4)));
#define WO_TB 32
#define WO_NB 4
__device__ __forceinline__ void wn_dma16(const float *g, const float *lds)
{
    const unsigned l = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(g) : "memory");
}
template <int K, bool POOL>
__global__ __launch_bounds__(256, 1) void rpn_wino_gemm_out64_kernel(WnArgs a, const float *__restrict__ U, const float *__restrict__ V)
{
    constexpr int P = 36, MO = 64, NS = K / 4, bufF = K * (MO + WO_TB), NB = WO_NB;
    constexpr int TPW = K / 16 + K / 32;                                  // 1-KB transfers per plane and wave
    extern __shared__ __attribute__((aligned(16))) float wo_s[];          // [NB][K x 64 (U_xi) | K x 32 (V_xi)]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i16 = lane & 15, kk = lane >> 4;
    const int t0 = blockIdx.x * WO_TB;
    const size_t sU = (size_t)K * MO, sV = (size_t)K * a.Ttot;
    auto issue = [&](int xi, int buf) {
        float *sA = wo_s + buf * bufF, *sB = sA + K * MO;
        const float *u = U + (size_t)xi * sU, *v = V + (size_t)xi * sV + t0;
#pragma unroll
        for (int q = 0; q < K / 16; ++q) {                                // U: 1-KB transfer d = k rows 4 d .. 4 d + 3; LDS float p of row r holds column (p - 16 r) mod 64
            const int d = wave + 4 * q, r = lane >> 4, c = ((lane & 15) * 4 - 16 * r) & 63;
            wn_dma16(u + (size_t)(4 * d + r) * MO + c, sA + d * 256);
        }
#pragma unroll
        for (int q = 0; q < K / 32; ++q) {                                // V: transfer d = k rows 8 d .. 8 d + 7 of 32 tiles; row r holds column (p - 16 ((r & 3) >> 1)) mod 32
            const int d = wave + 4 * q, r = lane >> 3, c = ((lane & 7) * 4 - 16 * ((r & 3) >> 1)) & 31;
            wn_dma16(v + (size_t)(8 * d + r) * a.Ttot + c, sB + d * 256);
        }
    };
    wn_f32x4 acc[P][2];
#pragma unroll
    for (int xi = 0; xi < P; ++xi) { acc[xi][0] = (wn_f32x4){0.0f, 0.0f, 0.0f, 0.0f}; acc[xi][1] = (wn_f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
    for (int xi = 0; xi < NB - 1; ++xi) issue(xi, xi);
#pragma unroll
    for (int xi = 0; xi < P; ++xi) {
        // plane xi has landed: this wave's transfers by the counted wait (loads complete in order; the planes issued after xi may still fly), the other
        // waves' by the barrier -- which also says that every wave is done reading the buffer the next issue refills
        {
            const int fl = (P - 1 - xi) < (NB - 2) ? (P - 1 - xi) : (NB - 2);      // planes issued after xi so far (compile-time: the loop is unrolled)
            static_assert(NB - 2 <= 2, "counted waits below");
            if (fl == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TPW) : "memory");
            else if (fl == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
#if !(defined(WO_ABL) && (WO_ABL & 1))                             // developer ablation 1: no transfers behind the first planes
        if (xi + NB - 1 < P) issue(xi + NB - 1, (xi + NB - 1) % NB);
#endif
        const float *sA = wo_s + (xi % NB) * bufF, *sB = sA + K * MO;
        const float *pa = sA + kk * MO + ((wave * 16 + i16 + 16 * kk) & 63);
        const float *pb0 = sB + kk * WO_TB + ((i16 + 16 * (kk >> 1)) & 31), *pb1 = sB + kk * WO_TB + ((16 + i16 + 16 * (kk >> 1)) & 31);
#pragma unroll
        for (int s = 0; s < NS; ++s) {                                    // lane (i16, kk): k = 4 s + kk
            const float av = pa[s * 4 * MO], b0 = pb0[s * 4 * WO_TB], b1 = pb1[s * 4 * WO_TB];
#if defined(WO_ABL) && (WO_ABL & 2)                                 // developer ablation 2: one MFMA step per plane instead of K / 4
            if (s > 0) continue;
#endif
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[xi][1], 0, 0, 0);
        }
    }
    // accumulator register r of column block tb: channel 16 wave + 4 kk + r, tile t0 + 16 tb + i16
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mv[P];
#pragma unroll
            for (int xi = 0; xi < P; ++xi) mv[xi] = acc[xi][tb][r];
            wn_output_tile<4, POOL>(a, wave * 16 + 4 * kk + r, t0 + 16 * tb + i16, mv);
        }
}

// How the stage's products are taken: 0 = v_mfma_f32_32x32x2_f32 (the default), 1 = split bf16 products (rpn_wino_gemm_kernel<.., SPLIT>).  Process-wide;
// FRCNN_CONV_F32_PRODUCTS=split sets the initial value, frcnn_conv3x3_f32_products() changes it between calls (returns the previous value; < 0 only asks).
static std::atomic<int> g_wn_products{[] { const char *e = getenv("FRCNN_CONV_F32_PRODUCTS"); return (e && (!strcmp(e, "split") || !strcmp(e, "1"))) ? 1 : 0; }()};
FRCNN_EXPORT int frcnn_conv3x3_f32_products(int mode)
{
    if (mode < 0) return g_wn_products.load();
    return g_wn_products.exchange(mode ? 1 : 0);
}

// the instantiation a product's tile widths need (host side)
static int wn_launch_gemm(bool nt, int MT, int NW, const WgArgs &g, float *part, int *cnt, hipStream_t s)
{
    const bool split = g_wn_products.load() != 0;
#define WN_GEMM_CASE(NTv, MTv, NWv)                                                                                          \
    if (nt == NTv && MT == MTv && NW == NWv) {                                                                               \
        if (split) FRCNN_LAUNCH((rpn_wino_gemm_kernel<NTv, MTv, NWv, true>), dim3((unsigned)g.G), dim3(256), 0, s, g, part, cnt);   \
        else FRCNN_LAUNCH((rpn_wino_gemm_kernel<NTv, MTv, NWv, false>), dim3((unsigned)g.G), dim3(256), 0, s, g, part, cnt); \
        FRCNN_CHECK_LAUNCH("rpn_wino_gemm_kernel");                                                                          \
        return FRCNN_OK;                                                                                                     \
    }
    WN_GEMM_CASE(false, 128, 128)
    WN_GEMM_CASE(false, 64, 128)
    WN_GEMM_CASE(false, 128, 64)
    WN_GEMM_CASE(false, 64, 64)
    WN_GEMM_CASE(true, 128, 128)
    WN_GEMM_CASE(true, 128, 64)
    WN_GEMM_CASE(true, 64, 128)
    WN_GEMM_CASE(true, 64, 64)
#undef WN_GEMM_CASE
    return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "conv3x3_f32: no GEMM for a %d x %d tile", MT, NW);
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient
#define CW_TC 16                       // columns per row segment (8 MFMA k steps)
#define CW_YS 17                       // LDS row stride of the dY segment [32 co][16]
#define CW_XS 19                       // LDS row stride of a feature row segment [32 ci][18]
#define CW_YB (32 * CW_YS)
#define CW_XB (32 * CW_XS)
#define CW_WAVE_LDS (2 * CW_YB + 4 * CW_XB)

struct CwLevel { const float *x; const float *dy; int H, W, HW, n_strips, unit0; };
struct CwArgs {
    CwLevel lv[FRCNN_MAX_LEVELS];
    int n_levels, C, n_units, S;       // S = workgroups per output tile (K split beyond the four waves)
};

#ifndef CW_WPS
#define CW_WPS 1
#endif
__global__ __launch_bounds__(256, CW_WPS) void rpn_conv3x3_f32_wgrad_kernel(CwArgs a, float *__restrict__ dw, float *__restrict__ part, int *__restrict__ cnt)
{
    __shared__ float s_all[4 * CW_WAVE_LDS];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int C = a.C, nt = C / 32;
    const int tile = (int)blockIdx.x / a.S, split = (int)blockIdx.x - tile * a.S;
    const int co0 = (tile / nt) * 32, ci0 = (tile % nt) * 32;
    float *sY = s_all + wave * CW_WAVE_LDS, *sX = sY + 2 * CW_YB;
    const int q = split * 4 + wave, nq = a.S * 4;
    const int u0 = (int)(((long long)q * a.n_units) / nq), u1 = (int)(((long long)(q + 1) * a.n_units) / nq);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // Staging roles: lane -> (row sr = channel of the tile, half sh): eight consecutive dY columns / nine consecutive feature columns
    // (the 18-wide halo row in two halves) per lane, so a unit's 17 loads and 17 LDS stores are one instruction each: the lane's
    // element offsets are computed once per strip (clamped to the row for columns outside the image; those are stored as zeros).
    const int sr = lane >> 1, sh = lane & 1;
    float *wy = sY + sr * CW_YS + sh * 8;                           // + buf * CW_YB + k
    float *wx = sX + sr * CW_XS + sh * 9;                           // + slot * CW_XB + k
    unsigned yo[8], xo[9];                                          // element offsets inside the row-0 segment of the lane's channel plane
    unsigned ymask = 0, xmask = 0;                                  // bit k: the column exists
    int lvl = 0, strip = 0, row = 0, H = 1, W = 1;
    const float *xg = nullptr, *yg = nullptr;
    auto decode = [&](int u) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < FRCNN_MAX_LEVELS; ++k) l += (k < a.n_levels && u >= a.lv[k].unit0) ? 1 : 0;
        const int v = u - a.lv[l].unit0;
        lvl = l; H = a.lv[l].H; W = a.lv[l].W; xg = a.lv[l].x; yg = a.lv[l].dy;
        strip = v / H; row = v - strip * H;
        const int HW = a.lv[l].HW, c0 = strip * CW_TC;
        ymask = 0; xmask = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c0 + sh * 8 + k;
            const bool ok = c < W;
            yo[k] = (unsigned)((co0 + sr) * HW + (ok ? c : 0));
            ymask |= ok ? (1u << k) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int c = c0 - 1 + sh * 9 + k;
            const bool ok = c >= 0 && c < W;
            xo[k] = (unsigned)((ci0 + sr) * HW + (ok ? c : 0));
            xmask |= ok ? (1u << k) : 0u;
        }
    };
    // two register sets: during unit r the loads of unit r + 2 land in set r & 1 while set (r + 1) & 1 (loaded during unit r - 1) is
    // written to LDS for unit r + 1 -- no copies, a whole unit (72 MFMAs) of flight for every load
    float sy[2][8], sx[2][9];
    auto load_y = [&](float (&d)[8], int r) {                       // r is a row of the image
        const float *b = yg + (size_t)r * W;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = b[yo[k]];
    };
    auto load_x = [&](float (&d)[9], int r) {                       // rows outside the image are read clamped and stored as zeros
        const float *b = xg + (size_t)min(max(r, 0), H - 1) * W;
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = b[xo[k]];
    };
    auto put_y1 = [&](int buf, int k, float v) { wy[buf * CW_YB + k] = (ymask >> k) & 1u ? v : 0.0f; };
    auto put_x1 = [&](int r, int k, float v) { wx[((r + 1) & 3) * CW_XB + k] = ((xmask >> k) & 1u) && r >= 0 && r < H ? v : 0.0f; };
    // one unit = row r of the strip: 8 k steps (column pairs) x 9 taps.  LOAD / WRITE are compile-time: the steady state has no branch.
    auto unit = [&](auto LOADc, auto WRITEc, float (&ly)[8], float (&lx)[9], float (&qy)[8], float (&qx)[9], int r) {
        constexpr bool LOAD = decltype(LOADc)::value, WRITE = decltype(WRITEc)::value;
        const float *py = sY + (r & 1) * CW_YB + li * CW_YS + lh;
        const float *px0 = sX + ((r + 0) & 3) * CW_XB + li * CW_XS + lh;            // feature row r - 1
        const float *px1 = sX + ((r + 1) & 3) * CW_XB + li * CW_XS + lh;
        const float *px2 = sX + ((r + 2) & 3) * CW_XB + li * CW_XS + lh;
        float oa[2], ob[2][9];
        auto fetch = [&](int ks, int slot) {
#if defined(CW_ABL) && (CW_ABL & 1)                                 // developer ablation: no LDS operand reads (results meaningless)
            oa[slot] = (float)(ks + lane);
#pragma unroll
            for (int t = 0; t < 9; ++t) ob[slot][t] = (float)(t + r);
#else
            oa[slot] = py[2 * ks];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3 - 1;
                const float *px = dy == 0 ? px0 : (dy == 1 ? px1 : px2);
                ob[slot][t] = px[2 * ks + dx + 1];
            }
#endif
        };
        fetch(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int slot = ks & 1;
            if (ks + 1 < 8) fetch(ks + 1, slot ^ 1);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot], ob[slot][0], acc[0], 0, 0, 0);
            if (LOAD && ks == 0) { load_y(ly, r + 2); load_x(lx, r + 3); }          // behind the first MFMA, not in front of the unit
            if (WRITE && ks >= 1) {                                                 // unit r + 1's operands -> LDS: two or three stores per step
                put_y1((r + 1) & 1, ks, qy[ks]);
                put_x1(r + 2, ks, qx[ks]);
                if (ks == 1) { put_y1((r + 1) & 1, 0, qy[0]); put_x1(r + 2, 0, qx[0]); }
                if (ks == 7) put_x1(r + 2, 8, qx[8]);
            }
#pragma unroll
            for (int t = 1; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot], ob[slot][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const std::true_type Y{};
    const std::false_type N{};

    if (u0 < u1) {
        decode(u0);
        int u = u0;
        while (u < u1) {
            // ---- a run of rows [row, r_end) of one strip: prime the ring, then one unit per row
            const int r_end = min(H, row + (u1 - u));
            load_y(sy[0], row);
#pragma unroll
            for (int k = 0; k < 8; ++k) put_y1(row & 1, k, sy[0][k]);
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                load_x(sx[0], row + d);
#pragma unroll
                for (int k = 0; k < 9; ++k) put_x1(row + d, k, sx[0][k]);
            }
            int r = row;
#if !(defined(CW_ABL) && (CW_ABL & 2))
            if (r + 1 < r_end) { load_y(sy[1], r + 1); load_x(sx[1], r + 2); }      // in flight: the operands of unit row + 1 (set 1)
            // steady state, two units per trip: sets alternate (unit at even distance from `row` loads set 0 and writes set 1)
            for (; r + 3 < r_end; r += 2) {
                unit(Y, Y, sy[0], sx[0], sy[1], sx[1], r);
                unit(Y, Y, sy[1], sx[1], sy[0], sx[0], r + 1);
            }
            // tail: at most three units left; the parity of (r - row) is even here
            if (r + 2 < r_end) { unit(Y, Y, sy[0], sx[0], sy[1], sx[1], r); unit(N, Y, sy[1], sx[1], sy[0], sx[0], r + 1); unit(N, N, sy[0], sx[0], sy[1], sx[1], r + 2); }
            else if (r + 1 < r_end) { unit(N, Y, sy[0], sx[0], sy[1], sx[1], r); unit(N, N, sy[1], sx[1], sy[0], sx[0], r + 1); }
            else unit(N, N, sy[0], sx[0], sy[1], sx[1], r);
#else
            for (; r < r_end; ++r) unit(N, N, sy[0], sx[0], sy[1], sx[1], r);
#endif
            u += r_end - row;
            if (u < u1) decode(u);
        }
    }
    // ---- the four waves' accumulators added through LDS in wave order, three taps at a time; then the tile (or this workgroup's slab)
    __syncthreads();
    float *dst = a.S > 1 ? part + ((size_t)tile * a.S + split) * (32 * 32 * 9) : nullptr;
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_all[((wave * 3 + tt) * 16 + r) * 64 + lane] = acc[g * 3 + tt][r];
        __syncthreads();
        for (int e = tid; e < 3 * 16 * 64; e += 256) {
            const float v = ((s_all[e] + s_all[3072 + e]) + s_all[2 * 3072 + e]) + s_all[3 * 3072 + e];
            const int tt = e / 1024, r = (e >> 6) & 15, l = e & 63;
            const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), ci = l & 31, t = g * 3 + tt;
            if (dst) __hip_atomic_store(&dst[(co * 32 + ci) * 9 + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef CW_NO_STORE
            else dw[((size_t)(co0 + co) * C + ci0 + ci) * 9 + t] = v;
#else
            else if (v == 12345.678f) dw[0] = v;
#endif
        }
        __syncthreads();
    }
    if (a.S == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(&cnt[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.S - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    for (int e = tid; e < 32 * 32 * 9; e += 256) {
        float v = 0.0f;
        for (int s = 0; s < a.S; ++s) v += __hip_atomic_load(&part[((size_t)tile * a.S + s) * (32 * 32 * 9) + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int co = e / 288, rem = e - co * 288;
        dw[((size_t)(co0 + co) * C + ci0) * 9 + rem] = v;
    }
    if (tid == 0) __hip_atomic_store(&cnt[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------- host side
// workspace (dedicated, ZERO before the first call, left zero by every call): [ticket words | transposed weights | slabs]
struct CfWs { int *cnt; float *wt, *part, *U, *V, *M; size_t total; };
static int cf_ranges()
{
    static int per_dev[64];                                          // 0 = not asked yet; per device: one process may drive several
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int g = __atomic_load_n(&per_dev[dev], __ATOMIC_RELAXED);
    if (!g) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        g = CF_WPS * cus;                                            // CF_WPS workgroups of 51.7 KB LDS per CU
        __atomic_store_n(&per_dev[dev], g, __ATOMIC_RELAXED);
    }
    return g;
}
static CfWs cf_carve(void *ws, int C, size_t u_floats = 0, size_t vm_floats = 0)
{
    // C = max(Cin, Cout) rounded up to 128: the direct kernels' share (Cin = Cout there); u_floats / vm_floats != 0 add the Winograd stage's
    // operand buffers (transformed weights; transformed inputs and products, either of which may be the wider side)
    CfWs w; char *p = (char *)ws; size_t o = 0;
    auto take = [&](size_t b) { void *r = p ? p + o : nullptr; o += align_up(b, 256); return r; };
    w.cnt = (int *)take(CF_MAX_TILES * sizeof(int));
    w.wt = (float *)take((size_t)C * C * 9 * sizeof(float));
    const size_t fwd = (size_t)cf_ranges() * 2 * CF_SLAB * sizeof(float);
    const size_t wg_tiles = (size_t)(C / 32) * (C / 32);           // weight gradient: tiles x workgroups per tile <= max(tiles, CUs) slabs of one tile
    const size_t wg = std::max<size_t>(wg_tiles, (size_t)cf_ranges() / CF_WPS * CW_WPS) * (32 * 32 * 9) * sizeof(float);
    w.part = (float *)take(fwd > wg ? fwd : wg);
    w.U = u_floats ? (float *)take(u_floats * sizeof(float)) : nullptr;
    w.V = vm_floats ? (float *)take(vm_floats * sizeof(float)) : nullptr;
    w.M = vm_floats ? (float *)take(vm_floats * sizeof(float)) : nullptr;
    w.total = o;
    return w;
}
size_t frcnn_ws_rpn_conv_f32(int64_t C) { return (C > 0 && C <= 4096) ? cf_carve(nullptr, (int)C).total : 0; }

static int cf_run(const float *const *in, float *const *out, const int *H, const int *W, int n_levels, int C, const float *w, const CfWs &ws,
                  hipStream_t s)
{
    CfArgs a;
    a.n_levels = n_levels; a.C = C; a.n_co_tiles = C / CF_MT; a.Kc = C / CF_CI;
    int tiles = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        if (l < n_levels) {
            a.lv[l] = {in[l], out[l], H[l], W[l], H[l] * W[l], tiles};
            tiles += (H[l] * W[l] + CF_NT - 1) / CF_NT;
        } else a.lv[l] = {nullptr, nullptr, 1, 1, 1, 1 << 30};
    }
    a.n_pos_tiles = tiles;
    a.pos_major = (long long)tiles * CF_NT * C > 2ll * C * C * 9 ? 1 : 0;      // activations larger than twice the weights
    const long long n_tiles = (long long)tiles * a.n_co_tiles, units = n_tiles * a.Kc;
    FRCNN_REQUIRE(n_tiles <= CF_MAX_TILES && units < (1ll << 31), "rpn_conv3x3_f32: %lld tiles above the limit %d", n_tiles, CF_MAX_TILES);
    a.n_units = (int)units;
    a.G = (int)std::min<long long>(cf_ranges(), units);
    FRCNN_LAUNCH(rpn_conv3x3_f32_kernel, dim3((unsigned)a.G), dim3(256), 0, s, a, w, ws.part, ws.cnt);
    FRCNN_CHECK_LAUNCH("rpn_conv3x3_f32_kernel");
    return FRCNN_OK;
}

// development A/B: FRCNN_CONV_F32_DIRECT=1 keeps the direct (9 C deep) kernel for forward / data gradient
static bool cf_use_direct()
{
    static const bool d = [] { const char *e = getenv("FRCNN_CONV_F32_DIRECT"); return e && atoi(e) != 0; }();
    return d;
}

// ranges of the k-contiguous GEMM (weight gradients): ONE workgroup per CU -- two did not run the matrix pipe any busier and cost this form 8-17 % in
// LDS / DMA contention (123 -> 113 us at 256 x 150 x 250, 60 -> 50 at 512 x 37 x 62); the K-major form keeps two (one per CU: 125 -> 137 us).  A ring of
// three chunk buffers with transfers two chunks ahead (96 KB, one workgroup per CU) was slower in both forms (143 / 121 us).
static int wn_ranges_nt() { return cf_ranges() / CF_WPS; }

// Padded tile total of the levels at tile size M, and the granularity it is padded to: the product's 128-wide tile, or its 64-wide one where that
// shortens the total by 15 % or more (160 tiles of a 37 x 62 map: 192 instead of 256; the narrow tile re-stages its 128 x 32 operand twice as often)
static long long wn_padded(int M, const int *H, const int *W, int n_levels, int *tg)
{
    long long t128 = 0, t64 = 0;
    for (int l = 0; l < n_levels; ++l) {
        const long long t = (long long)((H[l] + M - 1) / M) * ((W[l] + M - 1) / M);
        t128 += (t + 127) / 128 * 128;
        t64 += (t + 63) / 64 * 64;
    }
    const bool narrow = t64 * 100 <= t128 * 85;
    if (tg) *tg = narrow ? 64 : 128;
    return narrow ? t64 : t128;
}

static long long wn_fill(WnArgs *a, int M, const float *const *in, float *const *out, const int *H, const int *W, int n_levels)
{
    long long off = 0;
    wn_padded(M, H, W, n_levels, &a->tg);
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        if (l < n_levels) {
            const int tw = (W[l] + M - 1) / M, th = (H[l] + M - 1) / M;
            a->lv[l] = {in ? in[l] : nullptr, out ? out[l] : nullptr, H[l], W[l], tw, th * tw, (int)off};
            off += ((long long)th * tw + a->tg - 1) / a->tg * a->tg;
        } else a->lv[l] = {nullptr, nullptr, 1, 1, 1, 0, 1 << 30};
    }
    a->bias = nullptr; a->relu = 0; a->zero_pad = 0; a->db_part = nullptr; a->out2 = nullptr; a->bits_out = nullptr; a->bits = nullptr;
    a->n_levels = n_levels; a->C = 0; a->Ttot = (int)off;
    return off;
}

// The tile size of a call: what the stage executes scales with planes x padded tiles (36 planes over a quarter of the tiles against 16), so m = 4
// wherever that product is at least 15 % smaller -- every map of more than a few hundred tiles; small maps, whose 4 x 4 tiles would mostly be
// padding, keep m = 2 and its smaller rounding error.  FRCNN_WINO_M=2|4 forces one (development A/B and the tests' second form)
static int wn_pick_m(const int *H, const int *W, int n_levels)
{
    static const int forced = [] { const char *e = getenv("FRCNN_WINO_M"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 4) ? v : 0; }();
    if (forced) return forced;
    const long long c2 = 16 * wn_padded(2, H, W, n_levels, nullptr), c4 = 36 * wn_padded(4, H, W, n_levels, nullptr);
    return c4 * 100 <= c2 * 85 ? 4 : 2;
}

static bool wn_dims_ok(int Cin, int Cout) { return Cin > 0 && Cout > 0 && Cin % WN_KC == 0 && Cout % WN_KC == 0 && Cin <= 4096 && Cout <= 4096; }

static CfWs wn_carve(void *workspace, int Cin, int Cout, int M, long long Ttot)
{
    const int C = ((std::max(Cin, Cout) + CF_MT - 1) / CF_MT) * CF_MT, P = (M + 2) * (M + 2);
    return cf_carve(workspace, C, (size_t)P * Cin * Cout, (size_t)P * std::max(Cin, Cout) * Ttot);
}

FRCNN_EXPORT size_t frcnn_conv3x3_f32_workspace(const int *H_host, const int *W_host, int n_levels, int Cin, int Cout)
{
    if (!H_host || !W_host || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || !wn_dims_ok(Cin, Cout)) return 0;
    size_t need = 0;
    for (int M = 2; M <= 4; M += 2) {                                // whichever tile size a call picks (the choice may be forced by the environment)
        WnArgs a;
        const long long Ttot = wn_fill(&a, M, nullptr, nullptr, H_host, W_host, n_levels);
        need = std::max(need, wn_carve(nullptr, Cin, Cout, M, Ttot).total);
    }
    return need;
}
FRCNN_EXPORT size_t frcnn_rpn_conv3x3_f32_workspace(const int *H_host, const int *W_host, int n_levels, int C)
{
    if (C <= 0 || C % CF_MT != 0) return 0;
    return frcnn_conv3x3_f32_workspace(H_host, W_host, n_levels, C, C);
}

template <int M>
static int wn_strips(WnStrips *st, const WnArgs &a, const int *H, int n_levels, long long channels, int halo, size_t *lds_bytes, int sm = M)
{
    int n_strips = 0;
    for (int per_block = Wn<M>::TPB; per_block >= 256; per_block /= 2) {       // the largest strips that still give the chip >= 4096 workgroups
        n_strips = 0;
        for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
            st->first[l] = n_strips;
            const int tw = l < n_levels ? a.lv[l].tw : 1, th = l < n_levels ? (H[l] + M - 1) / M : 1;
            st->segs[l] = (tw + Wn<M>::WT - 1) / Wn<M>::WT;
            st->rows[l] = std::min(128, std::max(1, per_block / std::min(tw, Wn<M>::WT)));   // (M R + 2)(M wt + 2) <= Wn<M>::LDS for every wt <= WT
            if (l < n_levels) n_strips += st->segs[l] * ((th + st->rows[l] - 1) / st->rows[l]);
        }
        st->first[FRCNN_MAX_LEVELS] = n_strips;
        if ((long long)n_strips * channels >= 4096) break;
    }
    size_t fl = 0;                                                   // LDS: the largest (M R + 2 halo)(M wt + 2 halo) window of the levels
    for (int l = 0; l < n_levels; ++l) {
        const int wt = std::min(a.lv[l].tw, Wn<M>::WT), th = (H[l] + M - 1) / M, nr = std::min(st->rows[l], th);
        fl = std::max(fl, (size_t)(sm * nr + 2 * halo) * (size_t)(sm * wt + 2 * halo));      // sm = staged pixels per tile side (2 for pooled gradients)
    }
    *lds_bytes = fl * sizeof(float);
    return n_strips;
}

// forward (transposed = false) or data gradient (true) of the convolution with w [Cout][Cin][3][3] through the Winograd domain: four
// launches for all levels.  forward: K = Cin, M = Cout (+ bias, ReLU in the output transform);  data gradient: K = Cout, M = Cin, the
// incoming gradient optionally masked by the forward's ReLU output
template <int M>
static int wn_run(const float *const *in, float *const *out, const unsigned short *bits_in, const int *H, const int *W, int n_levels, int Cin, int Cout,
                  const float *w, bool transposed, const float *bias, int relu, unsigned short *bits_out, float *xt, void *workspace, hipStream_t s,
                  bool pooled = false, const float *u_rot = nullptr, float *dm_out = nullptr, bool want_bias = false)
{
    // relu (forward): 0 none, 1 ReLU, 2 ReLU + max_pool2d(2, 2);  pooled (data gradient): `in` is at the pooled resolution, bits_in are pool words
    constexpr int P = Wn<M>::P;
    const int K = transposed ? Cout : Cin, Mo = transposed ? Cin : Cout;
    WnArgs a;
    const long long Ttot = wn_fill(&a, M, in, out, H, W, n_levels);
    a.bits = bits_in;
    FRCNN_REQUIRE(Ttot < (1ll << 24) && (long long)P * std::max(K, Mo) * Ttot < (1ll << 31) * 4, "conv3x3_f32: %lld output tiles are too many", Ttot);
    const CfWs ws = wn_carve(workspace, Cin, Cout, M, Ttot);
    const int MT = Mo % CF_MT == 0 ? CF_MT : 64;                     // a 64-channel output side (conv1_2, the data gradient of conv2_1): 64-row tiles
    const int n_m_tiles = Mo / MT, n_t_tiles = (int)(Ttot / a.tg), Kc = K / WN_KC;
    const long long n_tiles = (long long)P * n_m_tiles * n_t_tiles, units = n_tiles * Kc;
    FRCNN_REQUIRE(n_tiles <= CF_MAX_TILES && units < (1ll << 31), "conv3x3_f32: %lld tiles above the limit %d", n_tiles, CF_MAX_TILES);
    const unsigned wb = (unsigned)((Mo / 16) * (K / 16));
    const float *Uuse = ws.U;
    if (transposed && u_rot) Uuse = u_rot;                           // the forward already made the rotated transform
    else {
        FRCNN_LAUNCH(rpn_wino_weight_kernel<M>, dim3(!transposed && u_rot ? 2 * wb : wb), dim3(256), 0, s, w, ws.U, (float *)u_rot, Cout, Cin, transposed ? 1 : 0);
        FRCNN_CHECK_LAUNCH("rpn_wino_weight_kernel");
    }
    a.C = K;
    float *Vb = xt ? xt : ws.V;                                      // kept for the weight gradient (zero padding columns included) or scratch
    a.zero_pad = xt ? 1 : 0;
    WnStrips st;
    size_t lds = 0;
    FRCNN_REQUIRE(!pooled || M == 4, "conv3x3_f32: a pooled gradient on the %d x %d tile", M, M);      // (the entry points refuse it first)
    const int n_strips = wn_strips<M>(&st, a, H, n_levels, K, 1, &lds, pooled ? 2 : M);
    if (dm_out) {                                                    // data gradient that also feeds the weight gradient: both transforms, one pass
        a.out2 = dm_out; a.zero_pad = 1;
        if (want_bias) {
            const size_t C9 = (size_t)((std::max(Cin, Cout) + CF_MT - 1) / CF_MT) * CF_MT;
            FRCNN_REQUIRE((size_t)K * n_strips <= C9 * C9 * 9, "conv3x3_f32_bwd_data: %d strips are too many for the bias partials", n_strips);
            a.db_part = ws.wt;
        }
        if (pooled) {
            if constexpr (M == 4) FRCNN_LAUNCH((rpn_wino_input_kernel<4, 2, true>), dim3((unsigned)n_strips, (unsigned)K), dim3(256), lds, s, a, st, Vb);
        } else FRCNN_LAUNCH((rpn_wino_input_kernel<M, 2, false>), dim3((unsigned)n_strips, (unsigned)K), dim3(256), lds, s, a, st, Vb);
    } else if (pooled) {
        if constexpr (M == 4) FRCNN_LAUNCH((rpn_wino_input_kernel<4, 0, true>), dim3((unsigned)n_strips, (unsigned)K), dim3(256), lds, s, a, st, Vb);
    } else FRCNN_LAUNCH((rpn_wino_input_kernel<M, 0, false>), dim3((unsigned)n_strips, (unsigned)K), dim3(256), lds, s, a, st, Vb);
    FRCNN_CHECK_LAUNCH("rpn_wino_input_kernel");
    if constexpr (M == 4) {
        static const bool no_fuse = [] { const char *e = getenv("FRCNN_WINO_NO_FUSE"); return e && atoi(e) != 0; }();
        if (Mo == 64 && K == 64 && !no_fuse) {                           // 64 -> 64 channels: the product with the output transform fused in (no product planes)
            a.C = Mo; a.bias = bias; a.relu = relu; a.bits_out = bits_out;
            const size_t wl = (size_t)WO_NB * K * (64 + WO_TB) * sizeof(float);      // 96 KB
            // per call: the attribute is per device, and a process may drive several (the call is a table lookup in the runtime)
            if (relu == 2) (void)hipFuncSetAttribute((const void *)rpn_wino_gemm_out64_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
            else (void)hipFuncSetAttribute((const void *)rpn_wino_gemm_out64_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
            const dim3 gr((unsigned)(Ttot / WO_TB));
            if (relu == 2) FRCNN_LAUNCH((rpn_wino_gemm_out64_kernel<64, true>), gr, dim3(256), wl, s, a, Uuse, Vb);
            else FRCNN_LAUNCH((rpn_wino_gemm_out64_kernel<64, false>), gr, dim3(256), wl, s, a, Uuse, Vb);
            FRCNN_CHECK_LAUNCH("rpn_wino_gemm_out64_kernel");
            return FRCNN_OK;
        }
    }
    WgArgs g = {Uuse, Vb, ws.M, (long long)Mo * K, (long long)K * Ttot, (long long)Mo * Ttot, Mo, (int)Ttot, (int)Ttot,
                n_m_tiles, n_t_tiles, Kc, (int)units, (int)std::min<long long>(cf_ranges(), units), 0};
    {
        // several tiles per workgroup: cut the ranges at tile boundaries (no partial tiles, no slabs, no tickets) when the rounding costs less
        // than 16 % (128 -> 128 on 300 x 500: 5.2 tiles per workgroup, 132 -> 124 us; 256 -> 256 on 150 x 250: 2.7, 121 -> 118)
        const long long per = (n_tiles + g.G - 1) / g.G;
        if (n_tiles >= 2ll * g.G && per * g.G * 100 <= n_tiles * 116) g.whole = 1;
        // nearly one tile per workgroup slot (432 tiles of 512 -> 512 on 37 x 62): one whole tile each on fewer workgroups, rather than 13.5 chunks each
        // with two partial segments, their slabs and a reduction per tile
        else if (n_tiles <= g.G && n_tiles * WN_ONE_TILE_DEN >= (long long)g.G * WN_ONE_TILE_NUM) { g.G = (int)n_tiles; g.whole = 1; }
    }
    { const int rc = wn_launch_gemm(false, MT, a.tg, g, ws.part, ws.cnt, s); if (rc) return rc; }
    a.C = Mo; a.bias = bias; a.relu = relu; a.bits_out = bits_out;
    if (relu == 2) {
        if constexpr (M == 4) FRCNN_LAUNCH((rpn_wino_output_kernel<4, true>), dim3((unsigned)((Ttot + 255) / 256), (unsigned)Mo), dim3(256), 0, s, a, ws.M);
    } else FRCNN_LAUNCH((rpn_wino_output_kernel<M, false>), dim3((unsigned)((Ttot + 255) / 256), (unsigned)Mo), dim3(256), 0, s, a, ws.M);
    FRCNN_CHECK_LAUNCH("rpn_wino_output_kernel");
    return FRCNN_OK;
}

// in_mult / out_mult: what the GEMM of the call needs of the two channel counts (a side that is tiled by 128, or only chunked by 32)
static int cf_check(const void *const *p0, const void *const *p1, const int *H, const int *W, int n_levels, int Cin, int Cout, int in_mult, int out_mult,
                    const void *w, void *ws, size_t ws_bytes, const char *what)
{
    FRCNN_REQUIRE(p0 && p1 && H && W && w && ws, "%s: NULL pointer", what);
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS, "%s: 1 <= n_levels <= %d", what, FRCNN_MAX_LEVELS);
    if (!wn_dims_ok(Cin, Cout) || Cin % in_mult != 0 || Cout % out_mult != 0)
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "%s: Cin=%d / Cout=%d must be multiples of %d / %d", what, Cin, Cout, in_mult, out_mult);
    for (int l = 0; l < n_levels; ++l) {
        FRCNN_REQUIRE(p0[l] && p1[l] && H[l] > 0 && W[l] > 0, "%s: bad level %d", what, l);
        FRCNN_REQUIRE((long long)H[l] * W[l] * std::max(Cin, Cout) < (1ll << 31), "%s: level %d too large", what, l);
    }
    const size_t need = frcnn_conv3x3_f32_workspace(H, W, n_levels, Cin, Cout);
    if (ws_bytes < need) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", what, ws_bytes, need);
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_conv3x3_f32_fwd(const float *const *x_dev, float *const *y_dev, const int *H_host, const int *W_host, int n_levels, int Cin, int Cout,
                                       const float *w_dev, const float *bias_dev, int relu, unsigned short *relu_bits_dev, float *x_transformed_dev, float *u_rotated_dev,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = cf_check((const void *const *)x_dev, (const void *const *)y_dev, H_host, W_host, n_levels, Cin, Cout, WN_KC, 64, w_dev, workspace, workspace_bytes,
                      "conv3x3_f32_fwd");
    if (rc) return rc;
    FRCNN_REQUIRE(relu >= 0 && relu <= 2, "conv3x3_f32_fwd: relu = %d (0 none, 1 ReLU, 2 ReLU + max_pool2d(2, 2))", relu);
    if (relu == 2 && wn_pick_m(H_host, W_host, n_levels) != 4)
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "conv3x3_f32_fwd: the fused max-pool needs the 4 x 4 tile (frcnn_conv3x3_f32_tile_size)");
    if (wn_pick_m(H_host, W_host, n_levels) == 4)
        return wn_run<4>(x_dev, y_dev, nullptr, H_host, W_host, n_levels, Cin, Cout, w_dev, false, bias_dev, relu, relu_bits_dev, x_transformed_dev, workspace, (hipStream_t)stream,
                         false, u_rotated_dev);
    return wn_run<2>(x_dev, y_dev, nullptr, H_host, W_host, n_levels, Cin, Cout, w_dev, false, bias_dev, relu, relu_bits_dev, x_transformed_dev, workspace, (hipStream_t)stream,
                     false, u_rotated_dev);
}

// 1 when the three calls would accept these shapes (channel multiples, tile / unit counts within the control block, sizes within the 32-bit offsets the
// kernels use), 0 otherwise: what a caller's dispatch (ops.conv3x3_supported) asks before it routes a layer here instead of to the vendor library
FRCNN_EXPORT int frcnn_conv3x3_f32_supported(const int *H_host, const int *W_host, int n_levels, int Cin, int Cout, int need_grads)
{
    if (!H_host || !W_host || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || !wn_dims_ok(Cin, Cout) || Cout % 64 != 0 || (need_grads && Cin % 64 != 0)) return 0;
    const int M = wn_pick_m(H_host, W_host, n_levels), P = (M + 2) * (M + 2), Cm = std::max(Cin, Cout);
    WnArgs a;
    const long long Ttot = wn_fill(&a, M, nullptr, nullptr, H_host, W_host, n_levels);
    for (int l = 0; l < n_levels; ++l)
        if (H_host[l] <= 0 || W_host[l] <= 0 || (long long)H_host[l] * W_host[l] * Cm >= (1ll << 31)) return 0;
    if (Ttot >= (1ll << 24) || (long long)P * Cm * Ttot >= (1ll << 31) * 4) return 0;
    auto side = [](long long c) { return c % CF_MT == 0 ? c / CF_MT : c / 64; };                 // tiles along a channel side (128 wide, or 64)
    auto tiles_ok = [&](long long mt, long long nt, long long k_chunks) {
        const long long n = P * mt * nt;
        return n <= CF_MAX_TILES && n * k_chunks < (1ll << 31);
    };
    if (!tiles_ok(side(Cout), Ttot / a.tg, Cin / WN_KC)) return 0;                               // forward
    if (need_grads && (!tiles_ok(side(Cin), Ttot / a.tg, Cout / WN_KC) || !tiles_ok(side(Cout), side(Cin), Ttot / WN_KC))) return 0;      // data gradient, weight gradient
    return 1;
}

FRCNN_EXPORT int frcnn_conv3x3_f32_tile_size(const int *H_host, const int *W_host, int n_levels)
{
    if (!H_host || !W_host || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS) return 0;
    return wn_pick_m(H_host, W_host, n_levels);
}

FRCNN_EXPORT size_t frcnn_conv3x3_f32_relu_bits_words(const int *H_host, const int *W_host, int n_levels, int Cout)
{
    if (!H_host || !W_host || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || Cout <= 0) return 0;
    WnArgs a;
    return (size_t)Cout * (size_t)wn_fill(&a, wn_pick_m(H_host, W_host, n_levels), nullptr, nullptr, H_host, W_host, n_levels);
}

FRCNN_EXPORT size_t frcnn_conv3x3_f32_u_floats(const int *H_host, const int *W_host, int n_levels, int Cin, int Cout)
{
    if (!H_host || !W_host || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || Cin <= 0 || Cout <= 0) return 0;
    const int M = wn_pick_m(H_host, W_host, n_levels);
    return (size_t)((M + 2) * (M + 2)) * (size_t)Cin * (size_t)Cout;
}

FRCNN_EXPORT size_t frcnn_conv3x3_f32_xt_floats(const int *H_host, const int *W_host, int n_levels, int Cin)
{
    if (!H_host || !W_host || n_levels < 1 || n_levels > FRCNN_MAX_LEVELS || Cin <= 0) return 0;
    const int M = wn_pick_m(H_host, W_host, n_levels);
    WnArgs a;
    return (size_t)((M + 2) * (M + 2)) * (size_t)Cin * (size_t)wn_fill(&a, M, nullptr, nullptr, H_host, W_host, n_levels);
}

FRCNN_EXPORT int frcnn_conv3x3_f32_bwd_data(const float *const *dy_dev, const unsigned short *relu_bits_dev, float *const *dx_dev, const int *H_host, const int *W_host,
                                            int n_levels, int Cin, int Cout, const float *w_dev, const float *u_rotated_dev, int pooled, float *dy_transformed_dev,
                                            int want_bias_partials, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = cf_check((const void *const *)dy_dev, (const void *const *)dx_dev, H_host, W_host, n_levels, Cin, Cout, 64, WN_KC, w_dev, workspace, workspace_bytes,
                      "conv3x3_f32_bwd_data");
    if (rc) return rc;
    if (pooled && (!relu_bits_dev || wn_pick_m(H_host, W_host, n_levels) != 4))
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "conv3x3_f32_bwd_data: a pooled gradient needs the forward's words and the 4 x 4 tile");
    if (wn_pick_m(H_host, W_host, n_levels) == 4)
        return wn_run<4>(dy_dev, dx_dev, relu_bits_dev, H_host, W_host, n_levels, Cin, Cout, w_dev, true, nullptr, 0, nullptr, nullptr, workspace, (hipStream_t)stream,
                         pooled != 0, u_rotated_dev, dy_transformed_dev, want_bias_partials != 0);
    return wn_run<2>(dy_dev, dx_dev, relu_bits_dev, H_host, W_host, n_levels, Cin, Cout, w_dev, true, nullptr, 0, nullptr, nullptr, workspace, (hipStream_t)stream,
                     false, u_rotated_dev, dy_transformed_dev, want_bias_partials != 0);
}

// weight gradient through the Winograd domain: V = B^T d B of the features and dM = A g A^T of the (masked) output gradient, both [xi][channel][t];
// dU = sum over the tiles on the k-contiguous form of the stage's GEMM (K = Ttot); dW = G^T dU G: four launches for all levels (+ one for the
// bias gradient)
template <int M>
static int wn_wgrad(const float *const *feats, const float *const *d_outs, const unsigned short *bits, const int *H, const int *W, int n_levels, int Cin, int Cout,
                    float *dw, float *dbias, const float *xt, void *workspace, hipStream_t s, bool pooled = false, const float *dm_in = nullptr)
{
    constexpr int P = Wn<M>::P;
    WnArgs a;
    const long long Ttot = wn_fill(&a, M, feats, nullptr, H, W, n_levels);
    FRCNN_REQUIRE(Ttot < (1ll << 24), "conv3x3_f32_wgrad: %lld output tiles are too many", Ttot);
    const CfWs ws = wn_carve(workspace, Cin, Cout, M, Ttot);
    const int MT = Cout % CF_MT == 0 ? CF_MT : 64, NW = Cin % CF_NT == 0 ? CF_NT : 64;
    const int mt = Cout / MT, nt = Cin / NW, C9 = ((std::max(Cin, Cout) + CF_MT - 1) / CF_MT) * CF_MT;     // ws.wt holds C9^2 * 9 floats
    const long long n_tiles = (long long)P * mt * nt, Kc = Ttot / WN_KC, units = n_tiles * Kc;
    FRCNN_REQUIRE(n_tiles <= CF_MAX_TILES && units < (1ll << 31), "conv3x3_f32_wgrad: %lld tiles above the limit %d", n_tiles, CF_MAX_TILES);
    WnStrips st;
    a.C = Cin; a.zero_pad = 1;
    size_t lds = 0;
    int n_strips = 0;
    if (!xt) {                                                       // the forward did not keep B^T d B of the activations: transform them again
        n_strips = wn_strips<M>(&st, a, H, n_levels, Cin, 1, &lds);
        FRCNN_LAUNCH((rpn_wino_input_kernel<M, 0, false>), dim3((unsigned)n_strips, (unsigned)Cin), dim3(256), lds, s, a, st, ws.V);
        FRCNN_CHECK_LAUNCH("rpn_wino_input_kernel");
    }
    WnArgs g1 = a;                                                   // the output gradient (and its mask) as the transform's input
    for (int l = 0; l < n_levels; ++l) g1.lv[l].x = d_outs[l];
    g1.bits = bits;
    g1.C = Cout;
    FRCNN_REQUIRE(!pooled || M == 4, "conv3x3_f32_wgrad: a pooled gradient on the %d x %d tile", M, M);
    n_strips = wn_strips<M>(&st, g1, H, n_levels, Cout, dm_in ? 1 : 0, &lds, pooled ? 2 : M);      // (dm_in: the strips of the data gradient's launch that made it)
    const int n_strips_dy = n_strips;
    FRCNN_REQUIRE(!dbias || (size_t)Cout * n_strips_dy <= (size_t)C9 * C9 * 9, "conv3x3_f32_wgrad: %d strips are too many for the bias partials", n_strips_dy);
    if (dbias) g1.db_part = ws.wt;                                   // the bias gradient's strip partials ride in this launch (in the direct form's weight
                                                                     // buffer, idle in this form) and are added up by the last launch
    if (dm_in) {
        // the data gradient's call already transformed the output gradient for this product (and left the bias partials in ws.wt)
    } else if (pooled) {
        if constexpr (M == 4) FRCNN_LAUNCH((rpn_wino_input_kernel<4, 1, true>), dim3((unsigned)n_strips, (unsigned)Cout), dim3(256), lds, s, g1, st, ws.M);
    } else FRCNN_LAUNCH((rpn_wino_input_kernel<M, 1, false>), dim3((unsigned)n_strips, (unsigned)Cout), dim3(256), lds, s, g1, st, ws.M);
    FRCNN_CHECK_LAUNCH("rpn_wino_input_kernel");
    WgArgs g = {dm_in ? dm_in : ws.M, xt ? xt : ws.V, ws.U, Ttot * Cout, Ttot * Cin, (long long)Cout * Cin, (int)Ttot, (int)Ttot, Cin, mt, nt, (int)Kc, (int)units,
                (int)std::min<long long>(wn_ranges_nt(), units), 0};
    { const int rc = wn_launch_gemm(true, MT, NW, g, ws.part, ws.cnt, s); if (rc) return rc; }
    const unsigned n = (unsigned)Cout * (unsigned)Cin;
    FRCNN_LAUNCH(rpn_wino_dw_kernel<M>, dim3((n + 255u) / 256u + (dbias ? (unsigned)Cout : 0u)), dim3(256), 0, s, ws.U, dw, n, dbias, ws.wt, n_strips_dy);
    FRCNN_CHECK_LAUNCH("rpn_wino_dw_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_conv3x3_f32_wgrad(const float *const *x_dev, const float *const *dy_dev, const unsigned short *relu_bits_dev, const int *H_host, const int *W_host,
                                         int n_levels, int Cin, int Cout, float *dw_dev, float *dbias_dev, const float *x_transformed_dev, int pooled,
                                         const float *dy_transformed_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = cf_check((const void *const *)x_dev, (const void *const *)dy_dev, H_host, W_host, n_levels, Cin, Cout, 64, 64, dw_dev, workspace, workspace_bytes,
                      "conv3x3_f32_wgrad");
    if (rc) return rc;
    if (wn_pick_m(H_host, W_host, n_levels) == 4)
        return wn_wgrad<4>(x_dev, dy_dev, relu_bits_dev, H_host, W_host, n_levels, Cin, Cout, dw_dev, dbias_dev, x_transformed_dev, workspace, (hipStream_t)stream,
                           pooled != 0, dy_transformed_dev);
    if (pooled) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "conv3x3_f32_wgrad: a pooled gradient needs the forward's words and the 4 x 4 tile");
    return wn_wgrad<2>(x_dev, dy_dev, relu_bits_dev, H_host, W_host, n_levels, Cin, Cout, dw_dev, dbias_dev, x_transformed_dev, workspace, (hipStream_t)stream, false,
                       dy_transformed_dev);
}

// ---- the stage's k-contiguous GEMM on its own: O[m][n] = sum_k A[m][k] B[n][k] (fp32, fixed summation order).  The weight gradient of a 1 x 1 convolution
// IS this product -- dW[co][ci] = sum_p dY[co][p] X[ci][p], both operands NCHW planes with the positions contiguous -- so the bottlenecks' and the FPN
// laterals' 1 x 1 convolutions behind models/new_model.py:372 get theirs without the vendor path's NCHW -> NHWC transposes (igemm_wrw + batched_transpose:
// 1.3 ms per 800 x 1344 step).  M, N multiples of 64, K a multiple of 32 (a partial last chunk would read past a row).
// splits: the K range cut into that many equal pieces, each a product of its own into O[split][M][N] (the caller adds them in order): a product with few
// output tiles and a very long K (dW [128][512] over 16 800 positions: four tiles) would otherwise hand every tile to ~100 workgroups and leave the
// fixed-order sum of their 64-KB slabs to ONE of them (133 us where the vendor kernel took 67); with 35 splits a tile is shared by three or four.
FRCNN_EXPORT int frcnn_gemm_nt_f32(const float *A_dev, const float *B_dev, float *O_dev, int M, int N, int K, int splits, void *workspace, size_t workspace_bytes,
                                   void *stream)
{
    FRCNN_REQUIRE(A_dev && B_dev && O_dev && workspace, "gemm_nt_f32: NULL pointer");
    if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || M % 64 != 0 || N % 64 != 0 || K % (WN_KC * splits) != 0 || M > 4096 || N > 4096)
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "gemm_nt_f32: M = %d, N = %d must be multiples of 64 (<= 4096), K = %d of %d x splits (%d)", M, N, K, WN_KC, splits);
    const CfWs ws = cf_carve(workspace, CF_MT);
    if (workspace_bytes < ws.total) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "gemm_nt_f32: workspace %zu < %zu bytes", workspace_bytes, ws.total);
    const int MT = M % CF_MT == 0 ? CF_MT : 64, NW = N % CF_NT == 0 ? CF_NT : 64, mt = M / MT, nt = N / NW, Ks = K / splits;
    const long long n_tiles = (long long)splits * mt * nt, Kc = Ks / WN_KC, units = n_tiles * Kc;
    FRCNN_REQUIRE(n_tiles <= CF_MAX_TILES && units < (1ll << 31), "gemm_nt_f32: %lld tiles above the limit %d", n_tiles, CF_MAX_TILES);
    WgArgs g = {A_dev, B_dev, O_dev, Ks, Ks, (long long)M * N, K, K, N, mt, nt, (int)Kc, (int)units, (int)std::min<long long>(wn_ranges_nt(), units), 0};
    return wn_launch_gemm(true, MT, NW, g, ws.part, ws.cnt, (hipStream_t)stream);
}
FRCNN_EXPORT size_t frcnn_gemm_nt_f32_workspace(void) { return cf_carve(nullptr, CF_MT).total; }

// ---- the RPN head's entry points: Cin = Cout = C, no bias (rpn_head.hip adds it), no mask; FRCNN_CONV_F32_DIRECT=1 routes them to the direct kernels
FRCNN_EXPORT int frcnn_rpn_conv3x3_f32_fwd(const float *const *feats_dev, float *const *outs_dev, const int *H_host, const int *W_host, int n_levels, int C,
                                           const float *w3_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cf_use_direct())
        return frcnn_conv3x3_f32_fwd(feats_dev, outs_dev, H_host, W_host, n_levels, C, C, w3_dev, nullptr, 0, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
    int rc = cf_check((const void *const *)feats_dev, (const void *const *)outs_dev, H_host, W_host, n_levels, C, C, CF_MT, CF_MT, w3_dev, workspace, workspace_bytes,
                      "rpn_conv3x3_f32_fwd");
    if (rc) return rc;
    return cf_run(feats_dev, outs_dev, H_host, W_host, n_levels, C, w3_dev, cf_carve(workspace, C), (hipStream_t)stream);
}

FRCNN_EXPORT int frcnn_rpn_conv3x3_f32_bwd_data(const float *const *d_outs_dev, float *const *d_feats_dev, const int *H_host, const int *W_host, int n_levels,
                                                int C, const float *w3_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cf_use_direct())
        return frcnn_conv3x3_f32_bwd_data(d_outs_dev, nullptr, d_feats_dev, H_host, W_host, n_levels, C, C, w3_dev, nullptr, 0, nullptr, 0, workspace, workspace_bytes, stream);
    int rc = cf_check((const void *const *)d_outs_dev, (const void *const *)d_feats_dev, H_host, W_host, n_levels, C, C, CF_MT, CF_MT, w3_dev, workspace,
                      workspace_bytes, "rpn_conv3x3_f32_bwd_data");
    if (rc) return rc;
    const CfWs ws = cf_carve(workspace, C);
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(rpn_conv_f32_pack_kernel, dim3((unsigned)((C / 16) * (C / 16))), dim3(256), 0, s, w3_dev, ws.wt, C);
    FRCNN_CHECK_LAUNCH("rpn_conv_f32_pack_kernel");
    return cf_run(d_outs_dev, d_feats_dev, H_host, W_host, n_levels, C, ws.wt, ws, s);
}

FRCNN_EXPORT int frcnn_rpn_conv3x3_f32_wgrad(const float *const *feats_dev, const float *const *d_outs_dev, const int *H_host, const int *W_host, int n_levels,
                                             int C, float *dw_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!cf_use_direct())
        return frcnn_conv3x3_f32_wgrad(feats_dev, d_outs_dev, nullptr, H_host, W_host, n_levels, C, C, dw_dev, nullptr, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
    int rc = cf_check((const void *const *)feats_dev, (const void *const *)d_outs_dev, H_host, W_host, n_levels, C, C, CF_MT, CF_MT, dw_dev, workspace, workspace_bytes,
                      "rpn_conv3x3_f32_wgrad");
    if (rc) return rc;
    const CfWs ws = cf_carve(workspace, C);
    CwArgs a;
    a.n_levels = n_levels; a.C = C;
    long long units = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        if (l < n_levels) {
            const int ns = (W_host[l] + CW_TC - 1) / CW_TC;
            a.lv[l] = {feats_dev[l], d_outs_dev[l], H_host[l], W_host[l], H_host[l] * W_host[l], ns, (int)units};
            units += (long long)ns * H_host[l];
        } else a.lv[l] = {nullptr, nullptr, 1, 1, 1, 1, 1 << 30};
    }
    FRCNN_REQUIRE(units < (1ll << 30), "rpn_conv3x3_f32_wgrad: too many row segments");
    a.n_units = (int)units;
    const int tiles = (C / 32) * (C / 32);
    static const int cus = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        return c;
    }();
    int S = CW_WPS * cus / tiles;                                   // CW_WPS workgroups per CU: 256 tiles at C = 512, 64 tiles at C = 256
    if (S < 1) S = 1;
    if (S > 16) S = 16;
    while (S > 1 && (long long)S * 4 > units) --S;
    a.S = S;
    FRCNN_LAUNCH(rpn_conv3x3_f32_wgrad_kernel, dim3((unsigned)(tiles * S)), dim3(256), 0, (hipStream_t)stream, a, dw_dev, ws.part, ws.cnt);
    FRCNN_CHECK_LAUNCH("rpn_conv3x3_f32_wgrad_kernel");
    return FRCNN_OK;
}
