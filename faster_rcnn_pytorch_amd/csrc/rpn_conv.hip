// rpn_conv.hip -- the whole RPN head of the FPN model (models/new_model.py:89-114) for the bf16 mixed-precision configuration
// (BASELINE configs[4]) as ONE implicit-GEMM kernel on the bf16 matrix cores:
//     raw = conv3x3(x, W3)            (256 -> 256, stride 1, pad 1, bf16 operands, fp32 accumulate, stored as bf16 for backward)
//     h   = relu(raw + b3)            (fp32 bias; never leaves the registers)
//     cls = Wc h + bc, reg = Wr h + br  (fp32 accumulate, fp32 outputs in the reference's permute(0,2,3,1).view(B,-1,2|4) layout)
// for all pyramid levels in one launch.  MIOpen runs the five 3x3 convolutions at 11.6 % of the dense bf16 peak
// (364 us for 105.6 GFLOP, profiles/r02b_fpn_bf16_kernel_summary.csv: igemm 160 us + NCHW<->NHWC transposes for the largest
// level, im2col + GEMM for the others) and the 256-channel intermediate makes a round trip through HBM before the 1x1 heads.
//
// GEMM view per output tile: D[m = output channel][n = position] = sum_k W3[m][k] X[k][n], k = (input channel, tap).
//   block = 256 threads = 4 waves, tile = 256 channels x (8 rows x 32 columns of one level); wave (wc, wp) owns channels
//   [128 wc, +128) x tile rows [4 wp, +4): 4 x 4 v_mfma_f32_32x32x16_bf16 tiles = 256 accumulator registers per lane.
//   K is walked as 16 input channels at a time (the 10 x 34 halo of the tile for those channels goes to LDS once, [pixel][16 ch],
//   pixel stride 48 bytes: conflict-free ds_read_b128 of "8 consecutive channels of my pixel") x 9 taps (a tap is a shift of
//   the halo address, and an 8 KB slice of the pre-packed weights W3p[chunk][tap][m][16] staged in LDS, row stride 48 bytes).
//   One workgroup barrier per (chunk, kernel row) step = 48 MFMAs per wave; both LDS buffers are double-buffered and filled from
//   registers that were loaded one step (weights) / one chunk (pixels) ahead.
// Epilogue on the accumulators (C/D layout: lane = position, registers = 16 channels): round to bf16 = raw (stored as 64-byte row
//   segments), add b3, ReLU, round to bf16 -- and those registers ARE the B operand of the second product (D2[j][n] += Wh[j][c] h[c][n]):
//   for a 32-channel tile the lane's registers 8 s .. 8 s + 7 hold channels 16 s + {0..3, 8..11} (+4 for the upper half-wave), which
//   is a permutation of a K = 16 slice; the head weights are pre-packed with the same permutation.  The two channel halves are
//   added through LDS, the biases are added and cls / reg are written.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(rpn_conv);
#include <atomic>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define RC3_C 256                       // input = output channels of the head
#define RC3_TH 8
#define RC3_TW 32
#define RC3_HW (RC3_TW + 2)             // halo width
#define RC3_HP ((RC3_TH + 2) * RC3_HW)  // halo pixels: 340
#define RC3_PS 24                       // pixel / weight-row stride in LDS, in bf16 elements (48 bytes)
#define RC3_WSTEP (3 * RC3_C * 16)      // bf16 elements of one step's weights: 3 taps x 256 rows x 16 channels = 24 KB, lane-linear (no padding)
#define RC3_STAGE_ELEMS (2 * RC3_HP * RC3_PS + 3 * RC3_WSTEP)   // bf16 elements of the pixel double buffer + the weight ring of three (106 KB)
#define RC3_WH_ELEMS (8 * 2 * 32 * 2 * 8)                               // packed head weights (16 KB)
#define RC3_LDS_BYTES ((size_t)(RC3_STAGE_ELEMS + RC3_WH_ELEMS) * 2 + RC3_C * 4 + 64 * 4)   // + b3 (1 KB) + the heads' biases: 123 KB of the CU's 160 KB

struct ConvLevels {
    int n_levels;
    const unsigned short *x[FRCNN_MAX_LEVELS];     // [256, H, W] bf16
    unsigned short *raw[FRCNN_MAX_LEVELS];         // [256, H, W] bf16 (bias-free 3x3 output)
    int H[FRCNN_MAX_LEVELS], W[FRCNN_MAX_LEVELS];
    int tile0[FRCNN_MAX_LEVELS + 1];               // first tile of level l
    int tiles_x[FRCNN_MAX_LEVELS];
    int pos0[FRCNN_MAX_LEVELS];                    // first output row of level l in the concatenated cls / reg tensors
};

__device__ __forceinline__ unsigned short f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 is a struct around a union: arrays of it end up in scratch)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// (lo, hi) -> two bf16 in one dword, round to nearest even, NaN stays NaN: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi)
{
    const bf16x2_t r = __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t);
    return *(const unsigned *)&r;
}
__device__ __forceinline__ float bf2f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// Position of W3p element (chunk, tap, row m, channel k of the chunk): [chunk 16][tap 9][row block 8][half g 2][row-in-block 32][8], i.e.
// inside a (tap, 32-row block) the 64 x 16-byte pieces lie in LANE order of the A fragment read (lane = 32 g + row: "8 consecutive
// channels of my row").  The kernel stages the weights with global_load_lds_dwordx4, which writes wave-base + lane x 16 bytes: the LDS
// image of a block is then lane-linear and its fragment read a conflict-free ds_read_b128 without any padding.
__device__ __forceinline__ int rc3_wpos(int chunk, int tap, int m, int k)
{
    return (((chunk * 9 + tap) * 8 + (m >> 5)) * 2 + (k >> 3)) * 256 + (m & 31) * 8 + (k & 7);
}
// W3 [256 out][256 in][3][3] fp32 -> W3p (order: rc3_wpos) bf16 ; Wc [n_cls][256], Wr [n_reg][256] fp32 ->
// Whp[ct 8][s 2][j 32][g 2][e 8] bf16 with channel = 32 ct + 16 s + 4 g + (e & 3) + 8 (e >> 2), rows j >= n_cls + n_reg zero
__global__ __launch_bounds__(256) void rpn_conv_pack_kernel(const float *__restrict__ w3, const float *__restrict__ w_cls, int n_cls,
                                                            const float *__restrict__ w_reg, int n_reg, unsigned short *__restrict__ w3p,
                                                            unsigned short *__restrict__ whp)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 16 * 9 * 256 * 16) {
        const int k = i & 15, m = (i >> 4) & 255, tap = (i >> 12) % 9, chunk = (i >> 12) / 9;
        w3p[rc3_wpos(chunk, tap, m, k)] = f2bf(w3[((size_t)m * RC3_C + chunk * 16 + k) * 9 + tap]);
    }
    if (i < 8 * 2 * 32 * 2 * 8) {
        const int e = i & 7, g = (i >> 3) & 1, j = (i >> 4) & 31, s = (i >> 9) & 1, ct = i >> 10;
        const int c = 32 * ct + 16 * s + 4 * g + (e & 3) + 8 * (e >> 2);
        float v = 0.0f;
        if (j < n_cls) v = w_cls[(size_t)j * RC3_C + c];
        else if (j < n_cls + n_reg) v = w_reg[(size_t)(j - n_cls) * RC3_C + c];
        whp[i] = f2bf(v);
    }
}

// Backward-data of the same convolution IS the same convolution: d_x[ci](y, x) = sum_{co, ky, kx} W3[co][ci][ky][kx] d_raw[co](y + 1 - ky, x + 1 - kx),
// i.e. a 3x3 / pad 1 convolution of d_raw with the weights transposed (in <-> out) and flipped (tap -> 8 - tap):
// W3 [256 out][256 in][3][3] fp32 -> W3p'[chunk 16 of co][tap 9][m = ci 256][16 co] bf16
__global__ __launch_bounds__(256) void rpn_conv_pack_bwd_kernel(const float *__restrict__ w3, unsigned short *__restrict__ w3p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 16 * 9 * 256 * 16) {
        const int k = i & 15, m = (i >> 4) & 255, tap = (i >> 12) % 9, chunk = (i >> 12) / 9;
        w3p[rc3_wpos(chunk, tap, m, k)] = f2bf(w3[((size_t)(chunk * 16 + k) * RC3_C + m) * 9 + (8 - tap)]);
    }
}

#ifdef RC3_TRACE                        // developer build: where a step's cycles go (wave 0 of workgroup 0; tools/dev/conv_trace.py)
__device__ unsigned long long g_rc3_trace[16];
extern "C" __attribute__((visibility("default"))) void frcnn_rc3_trace_read(void *dst) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rc3_trace), sizeof(g_rc3_trace)); }
#define RC3_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
__device__ unsigned long long g_wg_blocks[1024][4];    // wgrad: per workgroup {start, loop end, end (s_memrealtime, 10 ns), row segments}
extern "C" __attribute__((visibility("default"))) void frcnn_wg_blocks_read(void *dst) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wg_blocks), sizeof(g_wg_blocks)); }
#define WG_BLK(slot, v) do { if (threadIdx.x == 0) g_wg_blocks[blockIdx.y * gridDim.x + blockIdx.x][slot] = (v); } while (0)
#define RC3_ACC(slot, a, b) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_rc3_trace[slot] += (b) - (a); } while (0)
#else
#define RC3_STAMP(var) do {} while (0)
#define WG_BLK(slot, v) do {} while (0)
#define RC3_ACC(slot, a, b) do {} while (0)
#endif
// hand-placed vector-memory waits / loads (see the schedule in rpn_conv3x3_head_tile)
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier()                   // workgroup barrier that leaves vector-memory operations in flight (__syncthreads() would drain an LDS-DMA)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ unsigned ld_dword(const unsigned short *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// NPT = position tiles (image rows) per wave: 4 = the full 8-row tile, 2 = a 4-row half tile (ysub = 0 / 4 inside the full tile)
// HEAD = true: the fused RPN head (raw + bias + ReLU + both 1x1 heads); false: the plain convolution, output stored as bf16 (backward-data)
template <bool ODDW, int NPT, bool HEAD = true>
__device__ __forceinline__ void rpn_conv3x3_head_tile(int lvl, int tile, int ysub, const ConvLevels &L, const unsigned short *__restrict__ w3p, const float *__restrict__ b3,
                                                               const unsigned short *__restrict__ whp, const float *__restrict__ b_cls, int n_cls,
                                                               const float *__restrict__ b_reg, int n_reg, float *__restrict__ out_cls,
                                                               float *__restrict__ out_reg)
{
    // dynamic LDS (123 KB): [2][halo pixels x 24] (2 x 16 320 B), then a RING OF THREE weight buffers [3 taps][8 row blocks][64 lanes x 16 B]
    // (3 x 24 576 B, lane-linear: written by global_load_lds_dwordx4, see rc3_wpos); the epilogue reuses the first 64 KB.  The weights
    // of a whole kernel ROW (3 taps) are staged per barrier: one barrier per tap (16 MFMAs per wave between barriers, one wave per
    // SIMD) ran at 12 % of the matrix peak -- no better than MIOpen.
#ifdef RC3_TRACE
    const unsigned long long tt0 = __builtin_readcyclecounter();
    struct TileStamp { unsigned long long t0; __device__ ~TileStamp() { if (blockIdx.x == 0 && threadIdx.x == 0) g_rc3_trace[11] += __builtin_readcyclecounter() - t0; } } tile_stamp{tt0};
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned short s_all[];
    unsigned short (*s_x)[RC3_HP * RC3_PS] = (unsigned short (*)[RC3_HP * RC3_PS])s_all;
    unsigned short (*s_w)[RC3_WSTEP] = (unsigned short (*)[RC3_WSTEP])(s_all + 2 * RC3_HP * RC3_PS);
    // behind them: the packed head weights and b3, copied once per tile so that the epilogue reads them at LDS latency
    unsigned short *s_wh = s_all + RC3_STAGE_ELEMS;
    float *s_b3 = (float *)(s_wh + RC3_WH_ELEMS);
    float *s_bh = s_b3 + RC3_C;                             // the heads' biases (cls, then reg; 64 slots)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, g = lane >> 5;
    const int wc = wave & 1, wp = wave >> 1;               // channel half, tile-row half
    const int H = L.H[lvl], W = L.W[lvl];
    constexpr int TH = 2 * NPT;                             // rows of this (half) tile
    const int tl = tile - L.tile0[lvl];
    const int y0 = (tl / L.tiles_x[lvl]) * RC3_TH + ysub, x0 = (tl % L.tiles_x[lvl]) * RC3_TW;
    if (y0 >= H) return;                                    // the lower half of a tile that ends above it (whole workgroup)
    const unsigned short *xin = L.x[lvl];
    const size_t plane = (size_t)H * W;

    // Pixel staging.  Sub-dword memory operations are slow (global_load_ushort / ds_write_b16 per halo element cost 90 us per round
    // of tiles), so an item = (channel pair, halo row, aligned pixel pair): dword loads (channels 2q and 2q + 1, pixels 2w and
    // 2w + 1 of the row, w counted from the even column x0 - 2), a 2 x 2 transpose in registers, two dword LDS stores
    // ([pixel][channel pair]).  8 pairs x 10 rows x 18 words = 1440 items per 16-channel chunk, <= 6 per thread.
    // The loads are UNCONDITIONAL (out-of-image items read element 0 and are masked at the store): a load under a lane-dependent
    // branch makes the compiler's s_waitcnt bookkeeping give up, and every LDS store of the (earlier issued) weights then waits
    // for the HBM latency of the pixel loads as well -- 35 us per tile.
    // ODDW (a level of odd width, e.g. 13 x 21): element offsets can be odd, so each pair is cut out of the two aligned dwords
    // around it (4 loads per item instead of 2).
    constexpr int XI = (8 * (TH + 2) * 18 + 255) / 256;
    constexpr int XL = ODDW ? 2 : 1;
    unsigned x_off[XI];                                     // element offset of the even pixel inside the chunk (channel 2q); 0 if none
    short x_lds[XI];                                        // LDS element index of (pixel 2w - 1 relative to the halo, channel 2q); see store_x
    unsigned char x_ok[XI];                                 // bit 0 / 1: pixel 2w / 2w + 1 lies inside the image row; bit 2 / 3: inside the halo
    const unsigned end = (unsigned)RC3_C * (unsigned)plane; // elements of the level (< 2^31, checked by the host)
#pragma unroll
    for (int u = 0; u < XI; ++u) {
        const int e = t + 256 * u;
        const int q = e / ((TH + 2) * 18), rem = e - q * ((TH + 2) * 18);
        const int hy = rem / 18, w = rem - hy * 18;
        const int yy = y0 + hy - 1, xe = x0 - 2 + 2 * w;    // even pixel of the pair (x0 is a multiple of 32)
        const bool item = e < 8 * (TH + 2) * 18, row_in = yy >= 0 && yy < H;
        x_off[u] = item && row_in && xe >= 0 && xe < W ? (unsigned)(2 * q) * (unsigned)plane + (unsigned)yy * (unsigned)W + (unsigned)xe : 0u;
        // halo column of pixel xe is hx = xe - (x0 - 1) = 2 w - 1  (w = 0: only the odd pixel is in the halo; w = 17: hx = 33, 34: only the even one)
        x_lds[u] = (short)(item ? (hy * RC3_HW + 2 * w - 1) * RC3_PS + 2 * q : 0);
        x_ok[u] = (unsigned char)((item && row_in && xe >= 0 && xe < W ? 1 : 0) | (item && row_in && xe >= 0 && xe + 1 < W ? 2 : 0) |
                                  (item && w >= 1 ? 4 : 0) | (item && w <= 16 ? 8 : 0));
    }
    struct XV { unsigned a[XL][XI], b[XL][XI]; };
    auto load_x = [&](int chunk, XV &v, int u0, int u1) {
        const unsigned cbase = (unsigned)chunk * 16u * (unsigned)plane;
#pragma unroll
        for (int u = 0; u < XI; ++u) {
            if (u < u0 || u >= u1) continue;
            const unsigned oa = cbase + x_off[u], ob = oa + (unsigned)plane;
            // (inline asm: the compiler must not see these as loads -- with LDS-DMA pieces in flight it would wait vmcnt(0) at their
            // first use; the counted waits are placed by hand, see vm_wait / the schedule below)
            if (!ODDW) {                                    // W even: every offset is even and the pair is one aligned dword
                v.a[0][u] = ld_dword(xin + oa);
                v.b[0][u] = ld_dword(xin + ob);
            } else {                                        // the dwords at (o & ~1) and the next one, clamped to the last dword of the level
                const unsigned ea = oa & ~1u, eb = ob & ~1u;
                v.a[0][u] = ld_dword(xin + ea); v.a[XL - 1][u] = ld_dword(xin + min(ea + 2u, end - 2u));
                v.b[0][u] = ld_dword(xin + eb); v.b[XL - 1][u] = ld_dword(xin + min(eb + 2u, end - 2u));
            }
        }
    };
    auto tie_x = [&](XV &v) {                               // the values of load_x are defined HERE for the compiler: after the wait in front of it
#pragma unroll
        for (int u = 0; u < XI; ++u)
#pragma unroll
            for (int l = 0; l < XL; ++l) { asm volatile("" : "+v"(v.a[l][u])); asm volatile("" : "+v"(v.b[l][u])); }
    };
    auto store_x = [&](int buf, int chunk, const XV &v, int u0, int u1) {
        const unsigned cbase = (unsigned)chunk * 16u * (unsigned)plane;
#pragma unroll
        for (int u = 0; u < XI; ++u) {
            if (u < u0 || u >= u1) continue;
            unsigned va = v.a[0][u], vb = v.b[0][u];
            if (ODDW) {
                const unsigned oa = cbase + x_off[u], ob = oa + (unsigned)plane;
                va = __builtin_amdgcn_alignbit(v.a[XL - 1][u], va, (oa & 1u) * 16u);
                vb = __builtin_amdgcn_alignbit(v.b[XL - 1][u], vb, (ob & 1u) * 16u);
            }
            const unsigned a = (x_ok[u] & 1) ? (va & 0xFFFFu) : 0u, a1 = (x_ok[u] & 2) ? (va >> 16) : 0u;
            const unsigned b = (x_ok[u] & 1) ? (vb & 0xFFFFu) : 0u, b1 = (x_ok[u] & 2) ? (vb >> 16) : 0u;
            unsigned *dst = (unsigned *)(&s_x[buf][0]) + (x_lds[u] >> 1);           // element index is even (2 q): dword aligned
            if (x_ok[u] & 4) dst[0] = a | (b << 16);                                  // pixel 2w     : channels (2q, 2q + 1)
            if (x_ok[u] & 8) dst[RC3_PS / 2] = a1 | (b1 << 16);                       // pixel 2w + 1
        }
    };
    // weights of one (chunk, kernel row) step: 24 KB contiguous in W3p, already in their LDS order: 24 pieces of 1 KB, six per wave, each
    // ONE global_load_lds_dwordx4 (wave-uniform LDS base in M0, lane x 16 bytes) -- no registers, no ds_write, and the piece may stay
    // in flight across barriers (a register-staged step could fly for one step only: the waves spent 30 % of their time in the
    // s_waitcnt in front of the weight stores, PMC SQ_WAIT_INST_ANY)
    auto issue_w_part = [&](int step, int buf, int q0, int q1) {
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const int piece = wave * 6 + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w3p + (size_t)step * RC3_WSTEP + piece * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void *)(&s_w[buf][piece * 512]), 16, 0, 0);
        }
    };
    auto issue_w = [&](int step, int buf) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int piece = wave * 6 + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w3p + (size_t)step * RC3_WSTEP + piece * 512 + lane * 8),
                                             (__attribute__((address_space(3))) void *)(&s_w[buf][piece * 512]), 16, 0, 0);
        }
    };

    f32x16 acc[4][NPT];                                      // [channel tile][position tile]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = (f32x16){0};

    // Main loop: 48 steps = (chunk of 16 input channels) x (kernel row ky), 3 taps kx per step, 16 MFMAs per tap and wave.
    // One wave per SIMD (512 registers), so nothing hides a latency unless the code does:
    //  * the 8 fragments of tap kx + 1 are read from LDS into the OTHER fragment register set while the 16 MFMAs of tap kx run
    //    (the compiler's own schedule reloaded the weight fragment after every 4 MFMAs and waited for it: 44 % MFMA duty);
    //  * the barrier of a step sits before its LAST tap: the staging stores of step + 1 are done by then, so the fragments of
    //    (step + 1, tap 0) are prefetched behind the barrier while the last 16 MFMAs run;
    //  * global loads run two steps (weights) / two chunks (pixels) ahead of their LDS store, in the same registers: store, then
    //    immediately reload.  The body is unrolled over (2 chunks) x (3 rows) with unconditional, clamped loads so that the
    //    s_waitcnt before a weight store counts exactly the pixel loads issued after it.
    bf16x8 fa[2][4], fb[2][NPT];
    // part 0 = the position fragments, 1 / 2 = the first / last two channel-tile fragments, -1 = all.  Inside the loop they are issued
    // in three portions behind consecutive MFMA groups: eight ds_read_b128 in one place cost ~100 cycles of the wave's only issue slot
    // while the matrix pipe holds 32 cycles of work
    auto load_frags = [&](auto P, int chunk, int ky, int kx, int part = -1) {
        constexpr int p = decltype(P)::value;
        const unsigned short *sx = s_x[chunk & 1], *sw = s_w[(chunk * 3 + ky) % 3];
        if (part < 0 || part == 0) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) fb[p][pt] = *(const bf16x8 *)(sx + ((NPT * wp + pt + ky) * RC3_HW + li + kx) * RC3_PS + 8 * g);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            if (part < 0 || part == 1 + (ct >> 1)) fa[p][ct] = *(const bf16x8 *)(sw + ((kx * 8 + 4 * wc + ct) * 64 + lane) * 8);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // Vector-memory schedule (all counted by hand; P = the pixel loads of one chunk and thread).  G(k) = the six DMA pieces of step k's
    // weights, X(c) = chunk c's pixel loads.  Behind barrier k the ring slot of step k is free: X(c + 2) (ky = 0 only) and G(k + 3) are
    // issued from there on -- G in single pieces BETWEEN the 4-MFMA groups of the next three taps (a DMA issue costs ~100 cycles of the
    // wave's issue slot: M0 + address set-up; one group keeps the matrix pipe busy for 128): pieces 0, 1 under tap 2 of step k, pieces
    // 2 .. 5 under taps 0 and 1 of step k + 1.  A step's weights are in flight for two steps.  The pixel store of chunk c + 1 (ky = 0) is
    // spread over the groups of taps 0 and 1 as well.  In program order:
    //   step 3c   : [wait X(c+1)] G(3c+2).2-5 | wait G(3c+1) | barrier | G(3c+3).0-1 X(c+2)
    //   step 3c+1 :               G(3c+3).2-5 | wait G(3c+2) | barrier |        G(3c+4).0-1
    //   step 3c+2 :               G(3c+4).2-5 | wait G(3c+3) | barrier |        G(3c+5).0-1
    // so the operations younger than the awaited ones number 6 / 6 + P / 6 for the weights and 12 for X (8 in chunk 0, from the prologue).
    constexpr int P = 2 * XI * XL;
    XV xv;
    load_x(0, xv, 0, XI);
    issue_w(0, 0);
    if (HEAD) {
        u32x4 hv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) hv[k] = ((const u32x4 *)whp)[t + 256 * k];
        const float bv = b3[t];
#pragma unroll
        for (int k = 0; k < 4; ++k) ((u32x4 *)s_wh)[t + 256 * k] = hv[k];
        s_b3[t] = bv;
        // (16 dependent global loads per lane in the last loop of the epilogue, each behind its own s_waitcnt vmcnt(0), until round 3)
        if (t < 64) s_bh[t] = t < n_cls ? b_cls[t] : (t < n_cls + n_reg ? b_reg[t - n_cls] : 0.0f);
    }
    vm_wait<0>();
    tie_x(xv);
    store_x(0, 0, xv, 0, XI);
    load_x(1, xv, 0, XI);
    issue_w(1, 1);
    issue_w_part(2, 2, 0, 2);
    lds_barrier();
    load_frags(I0{}, 0, 0, 0);
    auto mfma_group = [&](auto S_, int ct) {
        constexpr int p = decltype(S_)::value;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[p][ct], fb[p][pt], acc[ct][pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto do_step = [&](auto P_, auto Q_, auto KY, int chunk) {  // fragments of tap 0 are in set P_ on entry, and in set Q_ = 1 - P_ for the next step
        constexpr int ky = decltype(KY)::value;
        const int step = chunk * 3 + ky;
        const int wnext = min(step + 2, 47), snext = (step + 2) % 3;    // the weights whose pieces 2 .. 5 are still to be issued (clamped at the
                                                                        // end: the last steps re-fetch step 47 into slots nobody reads again)
        RC3_STAMP(t0);
        if (ky == 0) {
            if (chunk == 0) vm_wait<8>(); else vm_wait<12>();           // X(chunk + 1) has landed (long ago)
            tie_x(xv);
        }
        // Order inside a tap: two MFMA groups with one DMA piece behind each, THEN the fragment reads of the next tap, then the other two
        // groups (with the pixel stores at ky = 0).  The compiler puts s_waitcnt lgkmcnt(0) in front of an LDS-DMA issue whenever LDS reads
        // are outstanding (the DMA writes LDS); with the reads issued first the wait landed in front of the tap's first MFMA and exposed
        // their whole latency, twice per step.
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {                                // tap 0 (its fragments were read under the previous tap)
            mfma_group(P_, ct);
            if (ct < 2) issue_w_part(wnext, snext, 2 + ct, 3 + ct);
            else if (ky == 0) store_x((chunk + 1) & 1, min(chunk + 1, 15), xv, (ct - 2) * (XI / 4), (ct - 1) * (XI / 4));
            if (ct >= 1) load_frags(Q_, chunk, ky, 1, ct - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        RC3_STAMP(t1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {                                // tap 1
            mfma_group(Q_, ct);
            if (ct < 2) issue_w_part(wnext, snext, 4 + ct, 5 + ct);
            else if (ky == 0) store_x((chunk + 1) & 1, min(chunk + 1, 15), xv, ct == 2 ? 2 * (XI / 4) : 3 * (XI / 4), ct == 2 ? 3 * (XI / 4) : XI);
            if (ct >= 1) load_frags(P_, chunk, ky, 2, ct - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        RC3_STAMP(t2);
        if (ky == 1) vm_wait<6 + P>(); else vm_wait<6>();               // G(step + 1) has landed
        RC3_STAMP(t3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RC3_STAMP(t4);
        lds_barrier();
        RC3_STAMP(t5);
        RC3_STAMP(t6);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {                                // tap 2 (fragments read before the barrier)
            mfma_group(P_, ct);
            if (ct < 2) issue_w_part(min(step + 3, 47), step % 3, ct, ct + 1);
            else if (ky == 0) load_x(min(chunk + 2, 15), xv, ct == 2 ? 0 : XI / 2, ct == 2 ? XI / 2 : XI);   // X(chunk + 2), behind pieces 0, 1
            if (ct >= 1) { if (ky == 2) load_frags(Q_, chunk + 1, 0, 0, ct - 1); else load_frags(Q_, chunk, ky + 1, 0, ct - 1); }
            __builtin_amdgcn_sched_barrier(0);
        }
        RC3_STAMP(t7);
        RC3_ACC(0, t0, t1); RC3_ACC(1, t1, t2); RC3_ACC(2, t2, t3); RC3_ACC(3, t3, t4); RC3_ACC(4, t4, t5); RC3_ACC(5, t5, t6); RC3_ACC(6, t6, t7); RC3_ACC(7, t0, t7);
        if (ky == 0) RC3_ACC(8, t2, t3);
    };
#ifdef RC3_TRACE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime(), tc0 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) g_rc3_trace[12] += tc0 - tt0;
#endif
    for (int chunk = 0; chunk < 16; chunk += 2) {
        do_step(I0{}, I1{}, I0{}, chunk);
        do_step(I1{}, I0{}, I1{}, chunk);
        do_step(I0{}, I1{}, std::integral_constant<int, 2>{}, chunk);
        do_step(I1{}, I0{}, I0{}, chunk + 1);
        do_step(I0{}, I1{}, I1{}, chunk + 1);
        do_step(I1{}, I0{}, std::integral_constant<int, 2>{}, chunk + 1);
    }
#ifdef RC3_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_rc3_trace[9] += __builtin_amdgcn_s_memrealtime() - tr0; g_rc3_trace[10] += __builtin_readcyclecounter() - tc0; }
#endif
    vm_wait<0>();                                           // the stray DMA pieces and pixel loads of the last steps are done ...
    tie_x(xv);
    lds_barrier();                                          // ... in every wave before the epilogue reuses LDS

    // ---- epilogue: raw (bf16), h = relu(raw + b3) as the B operand of the heads' product.
    // Accumulator register r of a lane is channel cb + (r & 3) + 8 (r >> 2) + 4 g at pixel (yy, x0 + li): registers (r, r + 1) are
    // adjacent channels and convert as one v_cvt_pk_bf16_f32.  Stores of single bf16 are slow, so lanes (li, li ^ 1) swap halves:
    // the even lane stores channel c of pixels (xx, xx + 1), the odd lane channel c + 1 of pixels (xx - 1, xx), one dword each
    // (an odd W breaks the dword alignment: that variant stores bf16 by bf16).
    // Position tile by position tile, so that only ONE 16-register head accumulator is live next to the 256 conv accumulators; every
    // wave leaves its 128-channel partial [32 outputs x 32 positions] in LDS (the staging buffers are free now).
    unsigned short *rawp = L.raw[lvl];
    const bool odd_lane = (li & 1) != 0;
    constexpr bool pair_ok = !ODDW;
    const unsigned sel = odd_lane ? 0x03020706u : 0x05040100u;                 // v_perm_b32 (a = neighbour, b = own): bytes of {a, b} = 7..4, 3..0
    if (!HEAD) {                                              // plain convolution: the accumulators, rounded to bf16, are the output
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int yy = y0 + NPT * wp + pt, xx = x0 + li;
            const bool in = yy < H && xx < W;
            const unsigned pix = (unsigned)yy * (unsigned)W + (unsigned)xx;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int cb = 128 * wc + 32 * ct + 4 * g;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = 2 * k, c = cb + (r & 3) + 8 * (r >> 2);
                    const unsigned own = cvt_pk_bf16(acc[ct][pt][r], acc[ct][pt][r + 1]);
                    if (pair_ok) {
                        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
                        const unsigned d = __builtin_amdgcn_perm(nb, own, sel);
                        if (in) *(unsigned *)(rawp + (unsigned)(c + (odd_lane ? 1 : 0)) * (unsigned)plane + (pix & ~1u)) = d;
                    } else if (in) {
                        rawp[(size_t)c * plane + pix] = (unsigned short)own;
                        rawp[(size_t)(c + 1) * plane + pix] = (unsigned short)(own >> 16);
                    }
                }
            }
        }
        return;
    }
    float *s_part = (float *)s_all;                           // [wc 2][wp 2][pt 4][r 16][lane 64] floats = 64 KB of the 106 KB
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int yy = y0 + NPT * wp + pt, xx = x0 + li;
        const bool in = yy < H && xx < W;
        const unsigned pix = (unsigned)yy * (unsigned)W + (unsigned)xx;
        f32x16 acc2 = (f32x16){0};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int cb = 128 * wc + 32 * ct + 4 * g;
            union { unsigned u[8]; bf16x8 v[2]; } hb;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = 2 * k, c = cb + (r & 3) + 8 * (r >> 2);          // registers r, r + 1 = channels c, c + 1
                const unsigned own = cvt_pk_bf16(acc[ct][pt][r], acc[ct][pt][r + 1]);
                {
                    if (pair_ok) {
                        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
                        const unsigned d = __builtin_amdgcn_perm(nb, own, sel);
                        if (in) *(unsigned *)(rawp + (unsigned)(c + (odd_lane ? 1 : 0)) * (unsigned)plane + (pix & ~1u)) = d;
                    } else if (in) {
                        rawp[(size_t)c * plane + pix] = (unsigned short)own;
                        rawp[(size_t)(c + 1) * plane + pix] = (unsigned short)(own >> 16);
                    }
                }
                const float2 bias = *(const float2 *)(s_b3 + c);
                const float z0 = __uint_as_float(own << 16) + bias.x, z1 = __uint_as_float(own & 0xFFFF0000u) + bias.y;
                hb.u[k] = cvt_pk_bf16(z0 > 0.0f ? z0 : 0.0f, z1 > 0.0f ? z1 : 0.0f);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 wa = *(const bf16x8 *)(s_wh + ((((4 * wc + ct) * 2 + s) * 32 + li) * 2 + g) * 8);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, hb.v[s], acc2, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s_part[(((wc * 2 + wp) * NPT + pt) * 16 + r) * 64 + lane] = acc2[r];
    }
    __syncthreads();
    // ---- add the two channel halves and the biases, store cls / reg: wave (wc, wp) finishes its share (wc) of the position tiles of row half wp
    const int J = n_cls + n_reg;
    float *oc = out_cls + (size_t)L.pos0[lvl] * n_cls, *orr = out_reg + (size_t)L.pos0[lvl] * n_reg;
#pragma unroll
    for (int q = 0; q < NPT / 2; ++q) {
        const int pt = (NPT / 2) * wc + q;
        const int yy = y0 + NPT * wp + pt, xx = x0 + li;
        if (yy >= H || xx >= W) continue;
        const size_t p = (size_t)yy * W + xx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * g;
            if (j < J) {
                const float v = s_part[(((0 * 2 + wp) * NPT + pt) * 16 + r) * 64 + lane] + s_part[(((1 * 2 + wp) * NPT + pt) * 16 + r) * 64 + lane] +
                                s_bh[j];
                if (j < n_cls) oc[p * n_cls + j] = v; else orr[p * n_reg + (j - n_cls)] = v;
            }
        }
    }
}

// One launch for all levels.  Blocks [0, first_split) compute full 8 x 32 tiles; the remaining tiles are cut into two 4-row halves, one
// block each: a tile takes ~80 us, so when the last round of tiles fills at most half of the CUs, halving those tiles halves that
// round (FPN 800 x 1344: 384 tiles on 256 CUs = 256 full tiles + 256 half tiles).  A level of odd width takes the variant that cuts
// its pixel pairs out of two aligned dwords (workgroup-uniform branches: all variants need the same registers and LDS).
__global__ __launch_bounds__(256) void rpn_conv3x3_head_kernel(ConvLevels L, int first_split, const unsigned short *__restrict__ w3p,
                                                               const float *__restrict__ b3, const unsigned short *__restrict__ whp,
                                                               const float *__restrict__ b_cls, int n_cls, const float *__restrict__ b_reg, int n_reg,
                                                               float *__restrict__ out_cls, float *__restrict__ out_reg)
{
    // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Neighbouring tiles share halo rows and the
    // 128-byte lines at their left / right edges, so consecutive tile ids are given to blocks b, b + 8, b + 16, ...: one XCD, dispatched
    // together (PMC: the pixel fetch was 2.4x the input with the identity mapping).
    auto xcd_order = [](int i, int n) { const int n8 = n & ~7; return i < n8 ? (i & 7) * (n8 >> 3) + (i >> 3) : i; };
    const int b = (int)blockIdx.x;
    const bool half = b >= first_split;
    const int hb = half ? xcd_order(b - first_split, (int)gridDim.x - first_split) : 0;
    const int tile = half ? first_split + (hb >> 1) : xcd_order(b, first_split), ysub = half ? 4 * (hb & 1) : 0;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) lvl += (l < L.n_levels && tile >= L.tile0[l]) ? 1 : 0;
    const bool odd = (L.W[lvl] & 1) != 0;
    if (!half) {
        if (odd) rpn_conv3x3_head_tile<true, 4>(lvl, tile, 0, L, w3p, b3, whp, b_cls, n_cls, b_reg, n_reg, out_cls, out_reg);
        else rpn_conv3x3_head_tile<false, 4>(lvl, tile, 0, L, w3p, b3, whp, b_cls, n_cls, b_reg, n_reg, out_cls, out_reg);
    } else {
        if (odd) rpn_conv3x3_head_tile<true, 2>(lvl, tile, ysub, L, w3p, b3, whp, b_cls, n_cls, b_reg, n_reg, out_cls, out_reg);
        else rpn_conv3x3_head_tile<false, 2>(lvl, tile, ysub, L, w3p, b3, whp, b_cls, n_cls, b_reg, n_reg, out_cls, out_reg);
    }
}

// backward-data launch: the same tiling and XCD order as the forward, plain-convolution epilogue
__global__ __launch_bounds__(256) void rpn_conv3x3_bwd_data_kernel(ConvLevels L, int first_split, const unsigned short *__restrict__ w3p)
{
    auto xcd_order = [](int i, int n) { const int n8 = n & ~7; return i < n8 ? (i & 7) * (n8 >> 3) + (i >> 3) : i; };
    const int b = (int)blockIdx.x;
    const bool half = b >= first_split;
    const int hb = half ? xcd_order(b - first_split, (int)gridDim.x - first_split) : 0;
    const int tile = half ? first_split + (hb >> 1) : xcd_order(b, first_split), ysub = half ? 4 * (hb & 1) : 0;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) lvl += (l < L.n_levels && tile >= L.tile0[l]) ? 1 : 0;
    const bool odd = (L.W[lvl] & 1) != 0;
    if (!half) {
        if (odd) rpn_conv3x3_head_tile<true, 4, false>(lvl, tile, 0, L, w3p, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr);
        else rpn_conv3x3_head_tile<false, 4, false>(lvl, tile, 0, L, w3p, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr);
    } else {
        if (odd) rpn_conv3x3_head_tile<true, 2, false>(lvl, tile, ysub, L, w3p, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr);
        else rpn_conv3x3_head_tile<false, 2, false>(lvl, tile, ysub, L, w3p, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr);
    }
}

size_t frcnn_ws_rpn_conv(void) { return (size_t)(16 * 9 * 256 * 16 + 8 * 2 * 32 * 2 * 8) * 2; }

// levels -> ConvLevels + launch geometry shared by the forward and the backward-data launch
static int rpn_conv_levels(const char *who, const void *const *in_levels, void *const *out_levels, const int *H, const int *W, int n_levels,
                           ConvLevels *Lp, int *first_split, unsigned *grid)
{
    int64_t pos0[FRCNN_MAX_LEVELS + 1] = {0};
    for (int k = 0; k < n_levels; ++k) {
        FRCNN_REQUIRE(in_levels[k] && out_levels[k] && H[k] > 0 && W[k] > 0 && (int64_t)H[k] * W[k] * RC3_C < ((int64_t)1 << 31), "%s: bad level %d", who, k);
        FRCNN_REQUIRE((((uintptr_t)in_levels[k] | (uintptr_t)out_levels[k]) & 3) == 0, "%s: level %d is not 4-byte aligned", who, k);
        pos0[k + 1] = pos0[k] + (int64_t)H[k] * W[k];
    }
    FRCNN_REQUIRE(pos0[n_levels] < ((int64_t)1 << 31), "%s: too many positions", who);
    ConvLevels &L = *Lp;
    int64_t tiles = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {              // unused slots repeat level 0 and start past the last tile
        const int k = l < n_levels ? l : 0;
        L.x[l] = (const unsigned short *)in_levels[k]; L.raw[l] = (unsigned short *)out_levels[k];
        L.H[l] = H[k]; L.W[l] = W[k];
        L.tiles_x[l] = (W[k] + RC3_TW - 1) / RC3_TW;
        L.tile0[l] = (int)tiles; L.pos0[l] = (int)pos0[k];
        if (l < n_levels) tiles += (int64_t)L.tiles_x[l] * ((H[k] + RC3_TH - 1) / RC3_TH);
    }
    L.tile0[FRCNN_MAX_LEVELS] = (int)tiles;
    L.n_levels = n_levels;
    FRCNN_REQUIRE(tiles < ((int64_t)1 << 30), "%s: too many tiles", who);
    // tiles of the last, at most half-full round of CUs are split into halves (see the kernel)
    int n_cu = 0;
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    const int64_t rem = tiles % n_cu;
    *first_split = (int)((rem > 0 && 2 * rem <= n_cu) ? tiles - rem : tiles);
    *grid = (unsigned)(*first_split + 2 * (tiles - *first_split));
    return FRCNN_OK;
}

template <typename K>
static int rpn_conv_reserve_lds(K kernel, const char *who, std::atomic<unsigned char> *done)
{   // > 64 KB of dynamic LDS is an opt-in per (function, device)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return frcnn_set_error(FRCNN_ERR_LAUNCH, "%s: no current device", who);
    if (dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
        const hipError_t rc = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RC3_LDS_BYTES);
        if (rc != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", who, (size_t)RC3_LDS_BYTES, hipGetErrorString(rc));
        if (dev < 64) done[dev].store(1, std::memory_order_release);
    }
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_rpn_conv_bwd_data(const void *const *d_raw_levels_bf16, void *const *d_feat_levels_bf16, const int *H, const int *W, int n_levels,
                                         int C, const float *w3, void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(C == RC3_C, "rpn_conv_bwd_data: C=%d (this kernel is built for the FPN head: 256 channels)", C);
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS && d_raw_levels_bf16 && d_feat_levels_bf16 && H && W, "rpn_conv_bwd_data: bad level table");
    FRCNN_REQUIRE(w3 && workspace, "rpn_conv_bwd_data: NULL pointer");
    if (workspace_bytes < frcnn_ws_rpn_conv()) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "rpn_conv_bwd_data: workspace %zu < %zu bytes", workspace_bytes, frcnn_ws_rpn_conv());
    ConvLevels L;
    int first_split = 0;
    unsigned grid = 0;
    int rc = rpn_conv_levels("rpn_conv_bwd_data", d_raw_levels_bf16, d_feat_levels_bf16, H, W, n_levels, &L, &first_split, &grid);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    unsigned short *w3p = (unsigned short *)workspace;
    FRCNN_LAUNCH(rpn_conv_pack_bwd_kernel, dim3(16 * 9 * 256 * 16 / 256), dim3(256), 0, s, w3, w3p);
    FRCNN_CHECK_LAUNCH("rpn_conv_pack_bwd_kernel");
    static std::atomic<unsigned char> done[64];
    rc = rpn_conv_reserve_lds(rpn_conv3x3_bwd_data_kernel, "rpn_conv_bwd_data", done);
    if (rc) return rc;
    FRCNN_LAUNCH(rpn_conv3x3_bwd_data_kernel, dim3(grid), dim3(256), RC3_LDS_BYTES, s, L, first_split, w3p);
    FRCNN_CHECK_LAUNCH("rpn_conv3x3_bwd_data_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_rpn_conv_head_fwd(const void *const *feat_levels_bf16, void *const *raw_levels_bf16, const int *H, const int *W, int n_levels, int C,
                                         const float *w3, const float *b3, const float *w_cls, const float *b_cls, int n_cls, const float *w_reg,
                                         const float *b_reg, int n_reg, float *out_cls, float *out_reg, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    FRCNN_REQUIRE(C == RC3_C, "rpn_conv_head: C=%d (this kernel is built for the FPN head: 256 channels)", C);
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS && feat_levels_bf16 && raw_levels_bf16 && H && W, "rpn_conv_head: bad level table");
    FRCNN_REQUIRE(n_cls > 0 && n_reg > 0 && n_cls + n_reg <= 32, "rpn_conv_head: n_cls + n_reg = %d must be in (0, 32]", n_cls + n_reg);
    FRCNN_REQUIRE(w3 && b3 && w_cls && b_cls && w_reg && b_reg && out_cls && out_reg && workspace, "rpn_conv_head: NULL pointer");
    if (workspace_bytes < frcnn_ws_rpn_conv()) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "rpn_conv_head: workspace %zu < %zu bytes", workspace_bytes, frcnn_ws_rpn_conv());
    int64_t pos0[FRCNN_MAX_LEVELS + 1] = {0};
    for (int k = 0; k < n_levels; ++k) {
        FRCNN_REQUIRE(feat_levels_bf16[k] && raw_levels_bf16[k] && H[k] > 0 && W[k] > 0 && (int64_t)H[k] * W[k] * RC3_C < ((int64_t)1 << 31), "rpn_conv_head: bad level %d", k);
        FRCNN_REQUIRE((((uintptr_t)feat_levels_bf16[k] | (uintptr_t)raw_levels_bf16[k]) & 3) == 0, "rpn_conv_head: level %d is not 4-byte aligned", k);
        pos0[k + 1] = pos0[k] + (int64_t)H[k] * W[k];
    }
    FRCNN_REQUIRE(pos0[n_levels] < ((int64_t)1 << 31), "rpn_conv_head: too many positions");
    hipStream_t s = (hipStream_t)stream;
    unsigned short *w3p = (unsigned short *)workspace, *whp = w3p + 16 * 9 * 256 * 16;
    FRCNN_LAUNCH(rpn_conv_pack_kernel, dim3(16 * 9 * 256 * 16 / 256), dim3(256), 0, s, w3, w_cls, n_cls, w_reg, n_reg, w3p, whp);
    FRCNN_CHECK_LAUNCH("rpn_conv_pack_kernel");
    const size_t lds = RC3_LDS_BYTES;
    {   // > 64 KB of dynamic LDS is an opt-in per (function, device)
        static std::atomic<unsigned char> done[64];
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_conv_head: no current device");
        if (dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            const hipError_t rc = hipFuncSetAttribute((const void *)rpn_conv3x3_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (rc != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_conv_head: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(rc));
            if (dev < 64) done[dev].store(1, std::memory_order_release);
        }
    }
    ConvLevels L;
    int64_t tiles = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {              // unused slots repeat level 0 and start past the last tile
        const int k = l < n_levels ? l : 0;
        L.x[l] = (const unsigned short *)feat_levels_bf16[k]; L.raw[l] = (unsigned short *)raw_levels_bf16[k];
        L.H[l] = H[k]; L.W[l] = W[k];
        L.tiles_x[l] = (W[k] + RC3_TW - 1) / RC3_TW;
        L.tile0[l] = (int)tiles; L.pos0[l] = (int)pos0[k];
        if (l < n_levels) tiles += (int64_t)L.tiles_x[l] * ((H[k] + RC3_TH - 1) / RC3_TH);
    }
    L.tile0[FRCNN_MAX_LEVELS] = (int)tiles;
    L.n_levels = n_levels;
    FRCNN_REQUIRE(tiles < ((int64_t)1 << 30), "rpn_conv_head: too many tiles");
    // tiles of the last, at most half-full round of CUs are split into halves (see the kernel)
    int n_cu = 0;
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    const int64_t rem = tiles % n_cu;
    const int first_split = (int)((rem > 0 && 2 * rem <= n_cu) ? tiles - rem : tiles);
    const unsigned grid = (unsigned)(first_split + 2 * (tiles - first_split));
    FRCNN_LAUNCH(rpn_conv3x3_head_kernel, dim3(grid), dim3(256), lds, s, L, first_split, w3p, b3, whp, b_cls, n_cls, b_reg, n_reg, out_cls, out_reg);
    FRCNN_CHECK_LAUNCH("rpn_conv3x3_head_kernel");
    return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 convolution on the bf16 matrix cores (what autograd derives for `inter_layer.weight`,
// models/new_model.py:96,109):   dW[co][ci][ky][kx] = sum over levels, y, x of  d_raw[co](y, x) * x[ci](y + ky - 1, x + kx - 1).
// GEMM view: M = co, N = ci (one 32 x 32 product per tap), K = positions, 16 consecutive columns of one image row per
// v_mfma_f32_32x32x16_bf16 -- in NCHW both operands of a K step are "8 consecutive pixels of one channel row", the A / B register
// layout of that instruction as the tensors lie in memory.
//   workgroup = 4 waves = (128 output channels) x (32 input channels) x 9 taps, wave w owns output channels [32 w, +32): 9 accumulators;
//   it walks a K range: row segments of 64 pixels, down the rows of one 64-pixel column strip, strip after strip, level after level;
//   per row segment the d_raw tile [128][64] and ONE new feature row [32][66] are staged through LDS (double buffer / ring of 4 rows); the
//   feature rows are kept in THREE copies shifted by -1 / 0 / +1 pixel (built with v_alignbit at staging time), so that every tap is an
//   aligned ds_read_b128 of "8 pixels of my channel"; the kernel row is the ring slot.  9 MFMAs per K step and wave against 1 + 9 reads.
//   grid = (splits of the K range) x (8 input-channel chunks x 2 output-channel halves); every workgroup leaves its partial
//   [128][32][9] in the workspace and rpn_conv_wgrad_finalize_kernel adds the splits in order (bit-reproducible, no atomics).
// MIOpen needs 689 us for the five levels at 800 x 1344 (igemm_wrw + transposes for the large levels, ~41 us of launch skeleton for
// each of the small ones).
// ------------------------------------------------------------------------------------------------------------------------------
#define WG_SEG 64                        // pixels per row segment (4 K steps)
#define WG_AS 72                         // LDS row stride in bf16 elements (144 bytes: 9 x 16, conflict-free ds_read_b128 down a column of rows)
#define WG_NCT 1                         // (accumulator sets per wave; two whole output-channel tiles x nine taps per wave = 288 accumulators spilled: 740 scratch stores)
#define WG_CO (128 * WG_NCT)
#define WG_CI 32
#define WG_MAX_SPLITS 32
struct WgradLevels {
    int n_levels;
    const unsigned short *x[FRCNN_MAX_LEVELS];     // features  [256, H, W] bf16
    const unsigned short *d[FRCNN_MAX_LEVELS];     // d_raw     [256, H, W] bf16
    int H[FRCNN_MAX_LEVELS], W[FRCNN_MAX_LEVELS];
    int seg0[FRCNN_MAX_LEVELS + 1];                // first row segment of level l in the global order (level, strip, row)
    int split0[WG_MAX_SPLITS + 1];                 // first row segment of split s
};
// host-side balancing of the K splits, in HALF fast row segments (measured with tools/dev/wgrad_trace.py: 1.2 us per 16-byte staged row
// segment, 2.4-2.7 per generically staged one of an even-width level, 5-7 on the odd-width level with its strip starts; with the
// round-2 weights 1 / 3 / 3 / 0 the split holding the two smallest levels ran 165 us against 137-150 for the others)
#ifndef WG_FAST_COST
#define WG_FAST_COST 2
#endif
#ifndef WG_SLOW_COST
#define WG_SLOW_COST 5                   // generic staging, even width (dword loads)
#endif
#ifndef WG_ODD_COST
#define WG_ODD_COST 14                   // generic staging, odd width (2-byte loads)
#endif
#ifndef WG_RUN_COST
#define WG_RUN_COST 10                   // starting a column strip: its window of exposed loads
#endif
#define WG_LDS_BYTES ((2 * WG_CO * WG_AS + 3 * 4 * WG_CI * WG_AS) * 2)     // 73 728 + 55 296 = 129 024 bytes of the CU's 160 KB

// one run = consecutive rows [y_first, y_first + n_rows) of one 64-pixel column strip of one level.
// FAST: W % 8 == 0 and 16-byte aligned planes (the two large levels: 94 % of the positions): 16-byte loads / LDS stores, 7 load
// instructions per thread and row instead of 28 dword loads.  Loads run TWO rows ahead of the MFMAs in two alternating register sets
// (one row of MFMAs, ~0.5 us, does not cover an L2 / Infinity-Cache round trip under load: with one row of cover the kernel ran at 14 %
// of the matrix peak, the waves waiting ~80 % of the time for the next row's operands).
template <bool FAST>
__device__ __forceinline__ void rpn_wgrad_run(const unsigned short *__restrict__ xin, const unsigned short *__restrict__ din, int H, int W, int x0,
                                              int y_first, int n_rows, int ci0, int co0, unsigned short (*s_a)[WG_CO * WG_AS],
                                              unsigned short (*s_f)[4][WG_CI * WG_AS], f32x16 (&acc)[WG_NCT][9])
{
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, g = lane >> 5;
    const size_t plane = (size_t)H * W;
    const bool odd_w = (W & 1) != 0;
    constexpr int NA16 = WG_CO * 8 / 256;                              // 16-byte pieces of the d_raw tile per thread (fast path)
    constexpr int NA4 = WG_CO * 32 / 256;                              // dwords per thread (generic path)
    // All loads are UNCONDITIONAL: an item outside the image reads element 0 of its plane and is zeroed at the LDS store (mask bits kept
    // beside the data).  A load under a lane-dependent branch makes the compiler's s_waitcnt bookkeeping give up: it then waits for
    // vmcnt(0) before every LDS store, i.e. for the OTHER register set's just-issued loads too, and the two-rows-ahead prefetch
    // degenerates to one row (the generic path even waited after every single load).
    struct Set { unsigned a[NA4]; unsigned f[12]; unsigned ma, mf; };
    // ---- feature row yy of my 32 channels
    auto load_f = [&](int yy, Set &S) {
        const bool rin = yy >= 0 && yy < H;
        unsigned m = 0u;
        if (FAST) {                                                    // item t: channel t >> 3, 8-pixel piece t & 7
            const int ch = t >> 3, xe = x0 + 8 * (t & 7);
            const unsigned short *pl = xin + (size_t)(ci0 + ch) * plane;
            const size_t ro = (size_t)(rin ? yy : 0) * W;
            const bool in_m = rin && xe < W, in_l = rin && xe >= 2 && xe - 2 < W, in_r = rin && xe + 8 < W;
            const u32x4 mid = *(const u32x4 *)(pl + (in_m ? ro + xe : 0));
            const unsigned left = *(const unsigned *)(pl + (in_l ? ro + xe - 2 : 0)), right = *(const unsigned *)(pl + (in_r ? ro + xe + 8 : 0));
            S.f[0] = mid[0]; S.f[1] = mid[1]; S.f[2] = mid[2]; S.f[3] = mid[3]; S.f[4] = left; S.f[5] = right;
            m = (in_m ? 1u : 0u) | (in_l ? 2u : 0u) | (in_r ? 4u : 0u);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = t + 256 * u, ch = c >> 5, j = c & 31;
                const unsigned short *pl = xin + (size_t)(ci0 + ch) * plane;
                const size_t ro = (size_t)(rin ? yy : 0) * W;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int xe = x0 - 2 + 2 * (j + q);               // even pixel of dword j + q
                    const bool i0 = rin && xe >= 0 && xe < W, i1 = rin && xe + 1 >= 0 && xe + 1 < W;
                    unsigned v;
                    if (!odd_w) v = *(const unsigned *)(pl + (i0 ? ro + xe : 0));      // W even: both pixels in or out together, dword aligned
                    else v = (unsigned)pl[i0 ? ro + xe : 0] | ((unsigned)pl[i1 ? ro + xe + 1 : 0] << 16);
                    S.f[3 * u + q] = v;
                    m |= (i0 ? 1u : 0u) << (2 * (3 * u + q)) | (i1 ? 2u : 0u) << (2 * (3 * u + q));
                }
            }
        }
        S.mf = m;
    };
    // copy kx holds pixel x0 + p + kx - 1 at position p (three aligned reads instead of misaligned ones)
    auto store_f = [&](int yy, const Set &S) {
        const int slot = (yy + 1) & 3;
        if (FAST) {
            const int e = (t >> 3) * WG_AS + 8 * (t & 7);
            const bool km = (S.mf & 1u) != 0u;
            const unsigned m0 = km ? S.f[0] : 0u, m1 = km ? S.f[1] : 0u, m2 = km ? S.f[2] : 0u, m3 = km ? S.f[3] : 0u;
            const unsigned lf = (S.mf & 2u) ? S.f[4] : 0u, rt = (S.mf & 4u) ? S.f[5] : 0u;
            *(u32x4 *)(&s_f[0][slot][e]) = (u32x4){__builtin_amdgcn_alignbit(m0, lf, 16), __builtin_amdgcn_alignbit(m1, m0, 16),
                                                  __builtin_amdgcn_alignbit(m2, m1, 16), __builtin_amdgcn_alignbit(m3, m2, 16)};
            *(u32x4 *)(&s_f[1][slot][e]) = (u32x4){m0, m1, m2, m3};
            *(u32x4 *)(&s_f[2][slot][e]) = (u32x4){__builtin_amdgcn_alignbit(m1, m0, 16), __builtin_amdgcn_alignbit(m2, m1, 16),
                                                  __builtin_amdgcn_alignbit(m3, m2, 16), __builtin_amdgcn_alignbit(rt, m3, 16)};
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = t + 256 * u, ch = c >> 5, j = c & 31;
                const int e = ch * WG_AS + 2 * j;                      // element index of pair j in the row (even: dword aligned)
                unsigned d[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const unsigned mm = (S.mf >> (2 * (3 * u + q))) & 3u;
                    d[q] = S.f[3 * u + q] & ((mm & 1u ? 0x0000FFFFu : 0u) | (mm & 2u ? 0xFFFF0000u : 0u));
                }
                *(unsigned *)(&s_f[0][slot][e]) = __builtin_amdgcn_alignbit(d[1], d[0], 16);       // (x0 + 2j - 1, x0 + 2j)
                *(unsigned *)(&s_f[1][slot][e]) = d[1];                                            // (x0 + 2j,     x0 + 2j + 1)
                *(unsigned *)(&s_f[2][slot][e]) = __builtin_amdgcn_alignbit(d[2], d[1], 16);       // (x0 + 2j + 1, x0 + 2j + 2)
            }
        }
    };
    auto load_a = [&](int yy, Set &S) {
        const bool rin = yy < H;
        const size_t ro = (size_t)(rin ? yy : 0) * W;
        unsigned m = 0u;
        if (FAST) {
#pragma unroll
            for (int u = 0; u < NA16; ++u) {
                const int c = t + 256 * u, r = c >> 3, xe = x0 + 8 * (c & 7);
                const bool in = rin && xe < W;
                const u32x4 v = *(const u32x4 *)(din + (size_t)(co0 + r) * plane + (in ? ro + xe : 0));
                S.a[4 * u] = v[0]; S.a[4 * u + 1] = v[1]; S.a[4 * u + 2] = v[2]; S.a[4 * u + 3] = v[3];
                m |= (in ? 1u : 0u) << u;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NA4; ++u) {
                const int c = t + 256 * u, r = c >> 5, xe = x0 + 2 * (c & 31);
                const unsigned short *pl = din + (size_t)(co0 + r) * plane;
                const bool i0 = rin && xe < W, i1 = rin && xe + 1 < W;
                unsigned v;
                if (!odd_w) v = *(const unsigned *)(pl + (i0 ? ro + xe : 0));
                else v = (unsigned)pl[i0 ? ro + xe : 0] | ((unsigned)pl[i1 ? ro + xe + 1 : 0] << 16);
                if (odd_w && !i1) v &= 0x0000FFFFu;
                S.a[u] = v;
                m |= (i0 ? 1u : 0u) << u;
            }
        }
        S.ma = m;
    };
    auto store_a = [&](int buf, const Set &S) {
        if (FAST) {
#pragma unroll
            for (int u = 0; u < NA16; ++u) {
                const int c = t + 256 * u;
                const bool in = (S.ma >> u) & 1u;
                *(u32x4 *)(&s_a[buf][(c >> 3) * WG_AS + 8 * (c & 7)]) =
                    (u32x4){in ? S.a[4 * u] : 0u, in ? S.a[4 * u + 1] : 0u, in ? S.a[4 * u + 2] : 0u, in ? S.a[4 * u + 3] : 0u};
            }
        } else {
#pragma unroll
            for (int u = 0; u < NA4; ++u) {
                const int c = t + 256 * u;
                *(unsigned *)(&s_a[buf][(c >> 5) * WG_AS + 2 * (c & 31)]) = ((S.ma >> u) & 1u) ? S.a[u] : 0u;
            }
        }
    };
    // Who multiplies what.  With wave w = output-channel tile w x all nine taps every wave read the SAME nine feature fragments per K step:
    // 4 x (1 + 9) x 1 KB = 40 KB of LDS reads per K step and CU, 320 cycles of the LDS pipe against 288 of MFMA.  Now wave (h = w & 1,
    // v = w >> 1) owns the output-channel tiles 2 h and 2 h + 1 for the taps 4 v .. 4 v + 3, plus tap 8 for tile 2 h + v: nine MFMAs as
    // before (accumulator s < 8: tile 2 h + (s >> 2), tap 4 v + (s & 3); s = 8: tile 2 h + v, tap 8), 2 + 5 fragment reads: 28 KB.
    // Every accumulator still sums its K range in the same order: the partials are bit-identical to the former assignment's.
    const int wh = wave & 1, wvv = wave >> 1;
    auto compute = [&](int y, int buf) {
        const unsigned short *sa = s_a[buf] + (64 * wh + li) * WG_AS + 8 * g;
        // one wave per SIMD: nothing overlaps the LDS fragment reads with the MFMAs unless the code does.  The fragments of K step
        // ks + 1 are read into the other register set while the MFMAs of step ks run.
        bf16x8 fa[2][2], fb[2][5];
        auto frags = [&](int p, int ks) {
            fa[p][0] = *(const bf16x8 *)(sa + 16 * ks);
            fa[p][1] = *(const bf16x8 *)(sa + 32 * WG_AS + 16 * ks);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int tap = j < 4 ? 4 * wvv + j : 8, ky = tap / 3, kx = tap - 3 * ky;       // (uniform: wvv is scalar)
                const int slot = (y + ky) & 3;                          // row y + ky - 1 lives in slot (y + ky - 1 + 1) & 3
                fb[p][j] = *(const bf16x8 *)(&s_f[kx][slot][li * WG_AS + 16 * ks + 8 * g]);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < WG_SEG / 16; ++ks) {
            if (ks + 1 < WG_SEG / 16) frags((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8)
                acc[0][s8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][s8 >> 2], fb[ks & 1][s8 & 3], acc[0][s8], 0, 0, 0);
            acc[0][8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wvv ? fa[ks & 1][1] : fa[ks & 1][0], fb[ks & 1][4], acc[0][8], 0, 0, 0);
        }
    };
    // ---- prime: feature rows y_first - 1 .. y_first + 1 and the d_raw row y_first go straight to LDS; the two register sets take
    //      (d_raw y_first + 1, features y_first + 2) and (d_raw y_first + 2, features y_first + 3)
    Set S0, S1;
    {   // all of the window's loads in flight together: ONE exposed round trip per run (a load -> store pair at a time cost three)
        Set S2;
        load_f(y_first - 1, S0); load_f(y_first, S1); load_f(y_first + 1, S2); load_a(y_first, S2);
        __syncthreads();                                                // the previous run's readers are done with LDS
        store_f(y_first - 1, S0); store_f(y_first, S1); store_f(y_first + 1, S2); store_a(0, S2);
    }
    load_a(y_first + 1, S0); load_f(y_first + 2, S0);
    load_a(y_first + 2, S1); load_f(y_first + 3, S1);
    for (int i = 0; i < n_rows; i += 2) {
        {   // even row of the pair: set S0 carries (d_raw y + 1, features y + 2)
            const int y = y_first + i;
            RC3_STAMP(w0);
            __syncthreads();
            RC3_STAMP(w1);
            compute(y, i & 1);
            RC3_STAMP(w2);
            store_a((i + 1) & 1, S0); store_f(y + 2, S0);
            RC3_STAMP(w3);
            load_a(y + 3, S0); load_f(y + 4, S0);
            RC3_STAMP(w4);
            if (blockIdx.y == 0) { RC3_ACC(0, w0, w1); RC3_ACC(1, w1, w2); RC3_ACC(2, w2, w3); RC3_ACC(3, w3, w4); RC3_ACC(4, w0, w0 + 1); }
        }
        if (i + 1 < n_rows) {   // (making only the compute conditional -- so that the compiler can count vmcnt across the back edge instead of
                                // waiting vmcnt(0) in front of the stores -- measured 7 us SLOWER on the same box: 202.5 against 195 us)
            const int y = y_first + i + 1;
            __syncthreads();
            compute(y, (i + 1) & 1);
            store_a(i & 1, S1); store_f(y + 2, S1);
            load_a(y + 3, S1); load_f(y + 4, S1);
        }
    }
}

__global__ __launch_bounds__(256) void rpn_conv3x3_wgrad_kernel(WgradLevels L, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short s_wg[];
    unsigned short (*s_a)[WG_CO * WG_AS] = (unsigned short (*)[WG_CO * WG_AS])s_wg;                              // [2][128 x 72]
    unsigned short (*s_f)[4][WG_CI * WG_AS] = (unsigned short (*)[4][WG_CI * WG_AS])(s_wg + 2 * WG_CO * WG_AS);    // [kx 3][slot 4][32 x 72]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, g = lane >> 5;
    const int split = blockIdx.x, ci0 = ((int)blockIdx.y / (RC3_C / WG_CO)) * WG_CI, co0 = ((int)blockIdx.y % (RC3_C / WG_CO)) * WG_CO;
    f32x16 acc[WG_NCT][9];
#pragma unroll
    for (int ct = 0; ct < WG_NCT; ++ct)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[ct][k] = (f32x16){0};
    // K range of this split: segments [L.split0[split], L.split0[split + 1]) of the global order (level, 64-pixel column strip, row).
    // The host cuts the order into ranges of equal COST: a row segment of a level that takes the generic staging path counts
    // WG_SLOW_COST times (equal COUNTS left all the narrow levels to the last split, whose workgroups then set the kernel's time),
    // and ranges stay contiguous because every new run (strip) costs a window of exposed loads.
    int seg = L.split0[split];
    const int seg_end = L.split0[split + 1];
    WG_BLK(0, __builtin_amdgcn_s_memrealtime()); WG_BLK(3, (unsigned long long)(seg_end - seg));
    while (seg < seg_end) {
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) lvl += (l < L.n_levels && seg >= L.seg0[l]) ? 1 : 0;
        const int H = L.H[lvl], W = L.W[lvl];
        const int rel = seg - L.seg0[lvl];
        const int strip = rel / H, y_first = rel - strip * H;
        const int n_rows = min(H - y_first, seg_end - seg);
        const bool fast = (W & 7) == 0 && ((((uintptr_t)L.x[lvl]) | ((uintptr_t)L.d[lvl])) & 15) == 0;
        if (fast) rpn_wgrad_run<true>(L.x[lvl], L.d[lvl], H, W, strip * WG_SEG, y_first, n_rows, ci0, co0, s_a, s_f, acc);
        else rpn_wgrad_run<false>(L.x[lvl], L.d[lvl], H, W, strip * WG_SEG, y_first, n_rows, ci0, co0, s_a, s_f, acc);
        seg += n_rows;
    }
    // ---- my partial, laid out [split][co][tap][ci]: register r of accumulator s (tile / tap: see compute) is (co = co0 + 32 tile + (r & 3) +
    //      8 (r >> 2) + 4 g, ci = ci0 + li), so the 32 lanes of a half-wave store 128 contiguous bytes (the weight's own [co][ci][tap]
    //      order made every store instruction touch ~18 cache lines: ~30 us of the kernel for 38 MB of partials)
    WG_BLK(1, __builtin_amdgcn_s_memrealtime());
    float *dst = part + (size_t)split * RC3_C * RC3_C * 9;
    const int wh = wave & 1, wvv = wave >> 1;
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) {
        const int tile = s9 < 8 ? 2 * wh + (s9 >> 2) : 2 * wh + wvv, tap = s9 < 8 ? 4 * wvv + (s9 & 3) : 8;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * tile + (r & 3) + 8 * (r >> 2) + 4 * g;
            dst[((size_t)co * 9 + tap) * RC3_C + ci0 + li] = acc[0][s9][r];
        }
    }
    WG_BLK(2, __builtin_amdgcn_s_memrealtime());
}

// dW = sum of the split partials, in split order (fixed: bit-reproducible); partials are [co][tap][ci], the weight is [co][ci][tap]
__global__ __launch_bounds__(256) void rpn_conv_wgrad_finalize_kernel(const float *__restrict__ part, int n_splits, float *__restrict__ dw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                  // index in the partial layout: (co * 9 + tap) * 256 + ci
    if (i >= RC3_C * RC3_C * 9) return;
    float v = 0.0f;
    for (int s0 = 0; s0 < n_splits; s0 += 8) {
        float tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tv[u] = s0 + u < n_splits ? part[(size_t)(s0 + u) * RC3_C * RC3_C * 9 + i] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) v += tv[u];
    }
    const int ci = i & (RC3_C - 1), ct = i >> 8, tap = ct % 9, co = ct / 9;
    dw[((size_t)co * RC3_C + ci) * 9 + tap] = v;
}

size_t frcnn_ws_rpn_conv_wgrad(void) { return (size_t)WG_MAX_SPLITS * RC3_C * RC3_C * 9 * sizeof(float); }

FRCNN_EXPORT int frcnn_rpn_conv_wgrad(const void *const *feat_levels_bf16, const void *const *d_raw_levels_bf16, const int *H, const int *W, int n_levels,
                                      int C, float *dw3, void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(C == RC3_C, "rpn_conv_wgrad: C=%d (this kernel is built for the FPN head: 256 channels)", C);
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS && feat_levels_bf16 && d_raw_levels_bf16 && H && W && dw3 && workspace, "rpn_conv_wgrad: bad argument");
    if (workspace_bytes < frcnn_ws_rpn_conv_wgrad()) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "rpn_conv_wgrad: workspace %zu < %zu bytes", workspace_bytes, frcnn_ws_rpn_conv_wgrad());
    WgradLevels L;
    int64_t segs = 0, cost = 0;
    int lvl_cost[FRCNN_MAX_LEVELS] = {0};
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        const int k = l < n_levels ? l : 0;
        FRCNN_REQUIRE(feat_levels_bf16[k] && d_raw_levels_bf16[k] && H[k] > 0 && W[k] > 0 && (int64_t)H[k] * W[k] * RC3_C < ((int64_t)1 << 31), "rpn_conv_wgrad: bad level %d", k);
        FRCNN_REQUIRE((((uintptr_t)feat_levels_bf16[k] | (uintptr_t)d_raw_levels_bf16[k]) & 3) == 0, "rpn_conv_wgrad: level %d is not 4-byte aligned", k);
        L.x[l] = (const unsigned short *)feat_levels_bf16[k]; L.d[l] = (const unsigned short *)d_raw_levels_bf16[k];
        L.H[l] = H[k]; L.W[l] = W[k];
        L.seg0[l] = (int)segs;
        const bool fast = (W[k] & 7) == 0 && ((((uintptr_t)feat_levels_bf16[k]) | ((uintptr_t)d_raw_levels_bf16[k])) & 15) == 0;
        lvl_cost[l] = fast ? WG_FAST_COST : (W[k] & 1) ? WG_ODD_COST : WG_SLOW_COST;
        if (l < n_levels) {                                         // + WG_RUN_COST per column strip: its window of exposed loads
            const int64_t strips = (W[k] + WG_SEG - 1) / WG_SEG, n = (int64_t)H[k] * strips;
            segs += n; cost += n * lvl_cost[l] + strips * WG_RUN_COST;
        }
    }
    for (int l = n_levels; l <= FRCNN_MAX_LEVELS; ++l) L.seg0[l] = (int)segs;
    L.n_levels = n_levels;
    FRCNN_REQUIRE(segs < ((int64_t)1 << 30), "rpn_conv_wgrad: too many positions");
    // K splits: 16 column groups (8 input-channel chunks x 2 output-channel halves) x splits workgroups = one round of the chip
    int n_cu = 256;
    { int dev = -1; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256; }
    const int groups = (RC3_C / WG_CI) * (RC3_C / WG_CO);           // workgroups per split: input-channel chunks x output-channel parts
    int splits = n_cu / groups;
    if (splits < 1) splits = 1;
    if (splits > WG_MAX_SPLITS) splits = WG_MAX_SPLITS;
    if ((int64_t)splits > segs) splits = (int)segs;
    {   // boundaries of equal cumulative cost
        int s_i = 1, lvl = 0;
        int64_t acc_cost = 0;
        L.split0[0] = 0;
        for (int64_t sg = 0; sg < segs && s_i < splits; ++sg) {
            while (lvl + 1 < n_levels && sg >= L.seg0[lvl + 1]) ++lvl;
            acc_cost += lvl_cost[lvl] + ((sg - L.seg0[lvl]) % L.H[lvl] == 0 ? WG_RUN_COST : 0);
            if (acc_cost * splits >= cost * s_i) L.split0[s_i++] = (int)(sg + 1);
        }
        for (; s_i <= WG_MAX_SPLITS; ++s_i) L.split0[s_i] = (int)segs;
        L.split0[splits] = (int)segs;
    }
    hipStream_t s = (hipStream_t)stream;
    static std::atomic<unsigned char> done[64];
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_conv_wgrad: no current device");
        if (dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            const hipError_t rc = hipFuncSetAttribute((const void *)rpn_conv3x3_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_BYTES);
            if (rc != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_conv_wgrad: cannot reserve %d bytes of LDS: %s", (int)WG_LDS_BYTES, hipGetErrorString(rc));
            if (dev < 64) done[dev].store(1, std::memory_order_release);
        }
    }
    FRCNN_LAUNCH(rpn_conv3x3_wgrad_kernel, dim3((unsigned)splits, (unsigned)groups), dim3(256), WG_LDS_BYTES, s, L, (float *)workspace);
    FRCNN_CHECK_LAUNCH("rpn_conv3x3_wgrad_kernel");
    FRCNN_LAUNCH(rpn_conv_wgrad_finalize_kernel, dim3(RC3_C * RC3_C * 9 / 256), dim3(256), 0, s, (const float *)workspace, splits, dw3);
    FRCNN_CHECK_LAUNCH("rpn_conv_wgrad_finalize_kernel");
    return FRCNN_OK;
}
