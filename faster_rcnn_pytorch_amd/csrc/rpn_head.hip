// rpn_head.hip -- the tail of RegionProposalNetwork.forward (models/model.py:79-83, models/new_model.py:109-113)
// as one MFMA kernel:   h = relu(conv3x3_raw + b3)  ->  cls = Wc h + bc,  reg = Wr h + br  ->  written directly in the
// reference's permute(0,2,3,1).contiguous().view(B,-1,2|4) layout ([P, 2A] and [P, 4A] row-major).
//
// The reference runs this as bias-add, ReLU, two 1x1 convolutions, two bias-adds and two NCHW->NHWC transposing
// copies (10-12 launches, ~60 us on MI355X); here the 1x1 heads are a [P x C] . [C x (2A+4A)] contraction on the
// matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32: D = fma chain in k order, no reduced precision), the
// bias + ReLU of the 3x3 output is applied while the A operand is loaded, and the epilogue stores NHWC.
//   block = 8 waves (4 if C % 128 != 0) = one tile of 32 positions; wave w takes the k-slice [w*C/8, (w+1)*C/8); the partial
//   accumulators are summed through LDS.  A operand: lane l reads raw[k0 + (l>>5)][pos0 + (l&31)] (two 128-B
//   segments per wave-load, straight from the NCHW conv output); B operand: lane l reads W[j = l&31][k0 + (l>>5)]
//   (each lane streams along its own weight row: L1-resident lines).
// The 3x3 convolution itself stays on MIOpen this round (its fp32 igemm already runs the MFMA path at ~100 TF/s).
// All FPN levels go through ONE launch (level table in the kernarg; outputs land at their offsets of the concatenated
// [sum P_l * A, 2|4] tensors), and the mixed-precision configuration (bf16 conv output, bf16 MFMA operands, fp32
// accumulate / bias / outputs so that box regression stays fp32) is a template instance of the same kernel.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(rpn_head);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct HeadLevels {
    int n_levels;
    const void *raw[FRCNN_MAX_LEVELS];       // [C, P_l] conv output of level l (fp32 or bf16)
    int P[FRCNN_MAX_LEVELS];
    int tile0[FRCNN_MAX_LEVELS + 1];         // first 32-position tile of level l in the grid
    int pos0[FRCNN_MAX_LEVELS];              // first output row of level l (concatenation order of new_model.py:42-44)
};

__device__ __forceinline__ float load_raw(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float load_raw(const unsigned short *p, size_t i) { return __uint_as_float((unsigned)p[i] << 16); }

__device__ __forceinline__ short to_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (short)((u >> 16) | 0x40);       // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}

// TIN: element type of the 3x3 output (float, or unsigned short = bf16 bits).  NT: number of 32-wide output tiles (1 when
// n_cls + n_reg <= 32, the FPN head with A = 3).  BF16MM: contract on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16,
// fp32 accumulate, fp32 bias and outputs -- the mixed-precision configuration) instead of the exact-fp32 32x32x2 MFMA.
template <typename TIN, int NT, bool BF16MM, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rpn_head_tail_kernel(HeadLevels L, int C, const float *__restrict__ b3,
                                                            const float *__restrict__ w_cls, const float *__restrict__ b_cls, int n_cls,
                                                            const float *__restrict__ w_reg, const float *__restrict__ b_reg, int n_reg,
                                                            float *__restrict__ out_cls, float *__restrict__ out_reg)
{
    __shared__ float s_acc[WAVES - 1][16 * NT][64];       // partial accumulators of waves 1..: [wave-1][reg][lane]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) lvl += (l < L.n_levels && (int)blockIdx.x >= L.tile0[l]) ? 1 : 0;
    const TIN *raw = (const TIN *)L.raw[lvl];
    const int P = L.P[lvl];
    const int p0 = ((int)blockIdx.x - L.tile0[lvl]) * 32;
    const int li = lane & 31, lk = lane >> 5;
    const int pos = min(p0 + li, P - 1);
    const int kq = C / WAVES;
    const int kbeg = wave * kq, kend = kbeg + kq;
    // weight rows of this lane for the output tiles (j = li, 32 + li); rows >= n_cls + n_reg contribute zeros
    const float *wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = 32 * t + li;
        wrow[t] = j < n_cls ? w_cls + (size_t)j * C : (j < n_cls + n_reg ? w_reg + (size_t)(j - n_cls) * C : nullptr);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};
    if constexpr (!BF16MM) {
        for (int k0 = kbeg; k0 < kend; k0 += 16) {
            // channel of MFMA step u, k-half lk:  k0 + 8 lk + u  (any bijection works as long as A and B agree): the eight
            // weights of a lane are then contiguous -> two 16-byte loads per tile instead of eight scalar ones
            float a[8];
            float4 bw[NT][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                 // 8 + 2 NT independent loads in flight before the first MFMA
                const int k = k0 + 8 * lk + u;
                a[u] = load_raw(raw, (size_t)k * P + pos) + b3[k];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 *wp = (const float4 *)(wrow[t] ? wrow[t] + k0 + 8 * lk : nullptr);
                bw[t][0] = wp ? wp[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                bw[t][1] = wp ? wp[1] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float h = a[u] > 0.0f ? a[u] : 0.0f;    // ReLU of the 3x3 output (bias already added)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 q = bw[t][u >> 2];
                    const float w = (u & 3) == 0 ? q.x : ((u & 3) == 1 ? q.y : ((u & 3) == 2 ? q.z : q.w));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(h, w, acc[t], 0, 0, 0);
                }
            }
        }
    } else {
        for (int k0 = kbeg; k0 < kend; k0 += 16) {        // one 32x32x16 MFMA per tile: lane l holds k = 8 * (l >> 5) + 0..7
            bf16x8 av, bv[NT];
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 8 * lk + u;
                a[u] = load_raw(raw, (size_t)k * P + pos) + b3[k];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 *wp = (const float4 *)(wrow[t] ? wrow[t] + k0 + 8 * lk : nullptr);
                const float4 q0 = wp ? wp[0] : make_float4(0.f, 0.f, 0.f, 0.f), q1 = wp ? wp[1] : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[t][0] = to_bf16_rne(q0.x); bv[t][1] = to_bf16_rne(q0.y); bv[t][2] = to_bf16_rne(q0.z); bv[t][3] = to_bf16_rne(q0.w);
                bv[t][4] = to_bf16_rne(q1.x); bv[t][5] = to_bf16_rne(q1.y); bv[t][6] = to_bf16_rne(q1.z); bv[t][7] = to_bf16_rne(q1.w);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = to_bf16_rne(a[u] > 0.0f ? a[u] : 0.0f);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[t], acc[t], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[wave - 1][16 * t + r][lane] = acc[t][r];
    }
    __syncthreads();
    if (wave == 0) {
        float *oc = out_cls + (size_t)L.pos0[lvl] * n_cls, *orr = out_reg + (size_t)L.pos0[lvl] * n_reg;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 32 * t + li;
            float bias = j < n_cls ? b_cls[j] : (j < n_cls + n_reg ? b_reg[j - n_cls] : 0.0f);
            // an unconditional use of the loaded value in front of the predicated stores: the compiler sinks "v += bias" into each store's
            // branch, and on the merge of a path that waited for the load with one that did not it re-emits s_waitcnt vmcnt(0) in EVERY
            // branch -- which then also waits for the previous branch's store to be acknowledged: 16 dependent round trips per tile
            // (ISA, round 3: most of this kernel's time)
            asm volatile("v_mov_b32 %0, %0" : "+v"(bias));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // C/D layout of the 32x32 MFMA: column j = lane & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int p = p0 + i;
                float v = acc[t][r];
#pragma unroll
                for (int w = 0; w < WAVES - 1; ++w) v += s_acc[w][16 * t + r][lane];
                v += bias;
                if (p < P) {
                    if (j < n_cls) oc[(size_t)p * n_cls + j] = v;
                    else if (j < n_cls + n_reg) orr[(size_t)p * n_reg + (j - n_cls)] = v;
                }
            }
        }
    }
}

template <typename TIN, bool BF16MM>
static void launch_head(int nt, const HeadLevels &L, int tiles, int C, const float *b3, const float *w_cls, const float *b_cls, int n_cls,
                        const float *w_reg, const float *b_reg, int n_reg, float *out_cls, float *out_reg, hipStream_t s)
{
    // 8 waves per 32-position tile when the K slice stays a multiple of 16 (C % 128 == 0): half the dependent load rounds
#define HEAD_LAUNCH(NT_, W_)                                                                                                            \
    FRCNN_LAUNCH((rpn_head_tail_kernel<TIN, NT_, BF16MM, W_>), dim3((unsigned)tiles), dim3(64 * W_), 0, s, L, C, b3, w_cls, \
                 b_cls, n_cls, w_reg, b_reg, n_reg, out_cls, out_reg)
    const bool w8 = (C % 128) == 0;
    if (nt == 1) { if (w8) HEAD_LAUNCH(1, 8); else HEAD_LAUNCH(1, 4); }
    else { if (w8) HEAD_LAUNCH(2, 8); else HEAD_LAUNCH(2, 4); }
#undef HEAD_LAUNCH
}

FRCNN_EXPORT int frcnn_rpn_head_tail_ml_fwd(const void *const *conv_raw_levels, int dtype, int mfma, int C, const int64_t *P_levels, int n_levels,
                                            const float *b3, const float *w_cls, const float *b_cls, int n_cls, const float *w_reg,
                                            const float *b_reg, int n_reg, float *out_cls, float *out_reg, void *stream)
{
    FRCNN_REQUIRE(C > 0 && C % 64 == 0, "rpn_head_tail: C=%d must be a positive multiple of 64", C);
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS && conv_raw_levels && P_levels, "rpn_head_tail: bad level table");
    FRCNN_REQUIRE(dtype == FRCNN_DTYPE_F32 || dtype == FRCNN_DTYPE_BF16, "rpn_head_tail: dtype %d (0 = f32, 1 = bf16)", dtype);
    FRCNN_REQUIRE(mfma == FRCNN_DTYPE_F32 || mfma == FRCNN_DTYPE_BF16, "rpn_head_tail: mfma %d (0 = exact fp32, 1 = bf16 operands)", mfma);
    FRCNN_REQUIRE(n_cls > 0 && n_reg > 0 && n_cls + n_reg <= 64, "rpn_head_tail: n_cls + n_reg = %d must be in (0, 64]", n_cls + n_reg);
    FRCNN_REQUIRE(b3 && w_cls && b_cls && w_reg && b_reg && out_cls && out_reg, "rpn_head_tail: NULL pointer");
    HeadLevels L;
    L.n_levels = n_levels;
    int64_t tiles = 0, pos = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        const int k = l < n_levels ? l : 0;
        FRCNN_REQUIRE(conv_raw_levels[k] && P_levels[k] > 0 && P_levels[k] < ((int64_t)1 << 30), "rpn_head_tail: bad level %d", k);
        L.raw[l] = conv_raw_levels[k]; L.P[l] = (int)P_levels[k];
        L.tile0[l] = (int)tiles; L.pos0[l] = (int)pos;
        if (l < n_levels) { tiles += (P_levels[k] + 31) / 32; pos += P_levels[k]; }
    }
    L.tile0[FRCNN_MAX_LEVELS] = (int)tiles;
    FRCNN_REQUIRE(tiles < ((int64_t)1 << 30), "rpn_head_tail: too many positions");
    hipStream_t s = (hipStream_t)stream;
    const int nt = n_cls + n_reg <= 32 ? 1 : 2;
    if (dtype == FRCNN_DTYPE_F32 && mfma == FRCNN_DTYPE_F32) launch_head<float, false>(nt, L, (int)tiles, C, b3, w_cls, b_cls, n_cls, w_reg, b_reg, n_reg, out_cls, out_reg, s);
    else if (dtype == FRCNN_DTYPE_F32) launch_head<float, true>(nt, L, (int)tiles, C, b3, w_cls, b_cls, n_cls, w_reg, b_reg, n_reg, out_cls, out_reg, s);
    else if (mfma == FRCNN_DTYPE_F32) launch_head<unsigned short, false>(nt, L, (int)tiles, C, b3, w_cls, b_cls, n_cls, w_reg, b_reg, n_reg, out_cls, out_reg, s);
    else launch_head<unsigned short, true>(nt, L, (int)tiles, C, b3, w_cls, b_cls, n_cls, w_reg, b_reg, n_reg, out_cls, out_reg, s);
    FRCNN_CHECK_LAUNCH("rpn_head_tail_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_rpn_head_tail_fwd(const float *conv_raw, int C, int64_t P, const float *b3, const float *w_cls, const float *b_cls,
                                         int n_cls, const float *w_reg, const float *b_reg, int n_reg, float *out_cls, float *out_reg,
                                         void *stream)
{
    const void *lv[1] = {conv_raw};
    return frcnn_rpn_head_tail_ml_fwd(lv, FRCNN_DTYPE_F32, FRCNN_DTYPE_F32, C, &P, 1, b3, w_cls, b_cls, n_cls, w_reg, b_reg, n_reg, out_cls, out_reg,
                                      stream);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Backward of the tail in ONE kernel + a finalize (the reference gets it from autograd as ~8 eager launches per level: bias
// gradients, two transposed 1x1 convolutions, the ReLU mask, the weight gradients, concatenations -- ~10 passes over the
// 256-channel intermediate, 92 MB at FPN size).  With g = [g_cls | g_reg] in [P, J] (J = 2A + 4A <= 64), z = raw + b3, h = relu(z):
//     dz[c, p] = (sum_j W[j, c] g[p, j]) * (z[c, p] > 0)        -> d_raw (the gradient of the bias-free 3x3 output), db3[c] = sum_p dz
//     dW[j, c] = sum_p g[p, j] h[c, p],   db[j] = sum_p g[p, j]
// Block = 8 waves, persistent over 32-position tiles, for 256 channels (blockIdx.y selects the half of a 512-channel head: two
// 32-wide channel tiles per wave needed ~330 registers and spilled 100 of them to scratch -- 69 us for the 72 tiles of config V);
// wave w owns the channels [256 y + 32 w, + 32).  Per tile: the g tile goes to LDS once; dz is a [C x J].[J x 32] product on v_mfma_f32_32x32x2f32 with the
// W^T operand resident in registers for the whole kernel; the mask is applied in the accumulator layout (lane = position, registers =
// 16 channels: raw is read and d_raw written as 128-byte row segments); h passes through a small per-wave LDS tile into the B
// operand of the second product, dW += g^T h, whose accumulators stay in registers across all tiles of the block.  Per-block
// partial sums (dW, db, db3) are added in block order by the finalize kernel: no atomics, bit-reproducible.
// ------------------------------------------------------------------------------------------------------------------------------
#define HB_GS 65                          // row stride of the g tile in LDS (conflict-free column reads)
#define HB_HS 33                          // row stride of the per-wave h tile

struct HeadBwdLevels {
    int n_levels;
    const void *raw[FRCNN_MAX_LEVELS];
    void *d_raw[FRCNN_MAX_LEVELS];
    int P[FRCNN_MAX_LEVELS];
    int tile0[FRCNN_MAX_LEVELS + 1];
    int pos0[FRCNN_MAX_LEVELS];
};

typedef __bf16 hb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hb_cvt_pk_bf16(float lo, float hi)      // one v_cvt_pk_bf16_f32: round to nearest even
{
    const hb_bf16x2 r = __builtin_convertvector((hb_f32x2){lo, hi}, hb_bf16x2);
    return *(const unsigned *)&r;
}
__device__ __forceinline__ void store_raw(float *p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void store_raw(unsigned short *p, size_t i, float v) { p[i] = (unsigned short)to_bf16_rne(v); }

template <typename TIN, int NJ, int CT>
#ifndef HB_NW
#define HB_NW 8                          // waves per workgroup (32 channels each); 4 (two independent workgroups per CU) measured 5 % slower
#endif
#define HB_GQ (2048 / (64 * HB_NW))        // g elements staged per thread
__global__ __launch_bounds__(64 * HB_NW) void rpn_head_tail_bwd_kernel(HeadBwdLevels L, int C, int n_tiles, const float *__restrict__ b3,
                                                                const float *__restrict__ w_cls, int n_cls, const float *__restrict__ w_reg, int n_reg,
                                                                const float *__restrict__ g_cls, const float *__restrict__ g_reg,
                                                                float *__restrict__ part_dw, float *__restrict__ part_db3, float *__restrict__ part_db)
{
    constexpr bool HB_BF16 = sizeof(TIN) == 2;
    __shared__ float s_gbuf[2][32 * HB_GS];
    __shared__ float s_h[HB_NW][32 * HB_HS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 31, lk = lane >> 5;
    const int J = n_cls + n_reg;
    const int cw = (int)blockIdx.y * 32 * HB_NW * CT + wave * 32 * CT;     // first channel of this wave
    // W^T operand of the first product, resident for the whole kernel: A[m = c][k = j] = W[j][c], step s covers j = 2 s + lk
    float wt[CT][NJ * 16];
    float bias[CT][16];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = cw + 32 * ct + li;
#pragma unroll
        for (int s = 0; s < NJ * 16; ++s) {
            const int j = 2 * s + lk;
            wt[ct][s] = j < n_cls ? w_cls[(size_t)j * C + c] : (j < J ? w_reg[(size_t)(j - n_cls) * C + c] : 0.0f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[ct][r] = b3[cw + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lk];
    }
    f32x16 acc_w[NJ][CT];
    float db3[CT][16];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) acc_w[jt][ct] = (f32x16){0};
#pragma unroll
        for (int r = 0; r < 16; ++r) db3[ct][r] = 0.0f;
    }
    float db = 0.0f;                                                // wave 0: lane = output column j
    float *sh = s_h[wave];
    // One workgroup per CU walks ~11 tiles at FPN size; with load -> use inside a tile every tile paid three dependent HBM round trips
    // (7.5 us per tile, 83 us per launch).  So the loads of tile t + 1 (its g rows and this wave's 16 raw values per channel tile) are
    // issued before the MFMAs of tile t, g is double-buffered in LDS: one barrier per tile.
    struct Tile { const TIN *raw; TIN *d_raw; int P, p0; size_t row0; };
    auto tile_of = [&](int tile) {
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < FRCNN_MAX_LEVELS; ++l) lvl += (l < L.n_levels && tile >= L.tile0[l]) ? 1 : 0;
        Tile t;
        t.raw = (const TIN *)L.raw[lvl]; t.d_raw = (TIN *)L.d_raw[lvl]; t.P = L.P[lvl];
        t.p0 = (tile - L.tile0[lvl]) * 32; t.row0 = (size_t)L.pos0[lvl] + t.p0;   // first row of this tile in the concatenated g tensors
        return t;
    };
    float gq[HB_GQ];                                                // g elements threadIdx.x + 64 HB_NW u of the tile (row p = e >> 6, column j = e & 63)
    float zq[CT][16];                                               // raw values of my position for the 16 channels of each channel tile
    auto load_tile = [&](const Tile &t) {
#pragma unroll
        for (int u = 0; u < HB_GQ; ++u) {
            const int e = (int)threadIdx.x + 64 * HB_NW * u, p = e >> 6, j = e & 63;
            float v = 0.0f;
            if (t.p0 + p < t.P) {
                if (j < n_cls) v = g_cls[(t.row0 + p) * n_cls + j];
                else if (j < J) v = g_reg[(t.row0 + p) * n_reg + (j - n_cls)];
            }
            gq[u] = v;
        }
        const int pos = min(t.p0 + li, t.P - 1);
        if (HB_BF16 && (t.P & 1) == 0) {
            // bf16 conv output, even plane size: 2-byte loads are slow (68 us against 52 us for the fp32 instance at FPN size), so lanes
            // (li, li ^ 1) share aligned DWORD loads: the even lane fetches channels r = 0, 2, .. of positions (pos, pos + 1), the odd
            // lane channels r = 1, 3, .. of (pos - 1, pos); they swap halves when the values are used (zq[ct][k] holds the raw dword)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = 2 * k + (li & 1);
                    const size_t o = (size_t)(cw + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lk) * t.P + (pos & ~1);
                    zq[ct][k] = __uint_as_float(*(const unsigned *)((const unsigned short *)t.raw + o));
                }
            return;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) zq[ct][r] = load_raw(t.raw, (size_t)(cw + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lk) * t.P + pos);
    };
    int tile = blockIdx.x;
    Tile cur = tile_of(min(tile, n_tiles - 1));
    if (tile < n_tiles) load_tile(cur);
    for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        float *s_g = s_gbuf[it & 1];
#pragma unroll
        for (int u = 0; u < HB_GQ; ++u) { const int e = (int)threadIdx.x + 64 * HB_NW * u; s_g[(e >> 6) * HB_GS + (e & 63)] = gq[u]; }
        float zr[CT][16];
        const bool paired = HB_BF16 && (cur.P & 1) == 0;            // workgroup-uniform: how load_tile fetched this tile
        if (paired) {
            const bool odd = (li & 1) != 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned own = __float_as_uint(zq[ct][k]);
                    const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
                    const float a = __uint_as_float(odd ? (own & 0xFFFF0000u) : (own << 16));   // my position, channel r = 2 k + odd
                    const float b = __uint_as_float(odd ? (nb & 0xFFFF0000u) : (nb << 16));     // my position, channel r = 2 k + !odd
                    zr[ct][2 * k] = (odd ? b : a) + bias[ct][2 * k];
                    zr[ct][2 * k + 1] = (odd ? a : b) + bias[ct][2 * k + 1];
                }
        } else {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) zr[ct][r] = zq[ct][r] + bias[ct][r];
        }
        __syncthreads();                                            // g of this tile is staged; the other buffer's readers finished a tile ago
        const Tile me = cur;
        if (tile + (int)gridDim.x < n_tiles) { cur = tile_of(tile + (int)gridDim.x); load_tile(cur); }   // in flight under the MFMAs below
        if (wave == 0) {
#pragma unroll 8
            for (int p = 0; p < 32; ++p) db += s_g[p * HB_GS + lane];
        }
        const int P = me.P;
        const int pos = min(me.p0 + li, P - 1);
        const bool pv = me.p0 + li < P;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            // ---- dz = W^T g (accumulator: lane = position li, registers = 16 channels)
            f32x16 acc = (f32x16){0};
#pragma unroll
            for (int s = 0; s < NJ * 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[ct][s], s_g[li * HB_GS + 2 * s + lk], acc, 0, 0, 0);
            // ---- ReLU mask, d_raw, db3, h -> LDS
            __builtin_amdgcn_wave_barrier();                        // the previous c-tile's readers of my h tile are done
            if (paired) {                                           // bf16 d_raw as dwords: the same lane pairing as the loads
                const bool odd = (li & 1) != 0;
                const unsigned sel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32 (a = neighbour, b = own)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = 2 * k;
                    const float d0 = zr[ct][r] > 0.0f ? acc[r] : 0.0f, d1 = zr[ct][r + 1] > 0.0f ? acc[r + 1] : 0.0f;
                    const unsigned own = hb_cvt_pk_bf16(d0, d1);
                    const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);
                    const unsigned d = __builtin_amdgcn_perm(nb, own, sel);
                    const int cl = ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2) + 4 * lk;
                    if (pv) {
                        *(unsigned *)((unsigned short *)me.d_raw + (size_t)(cw + 32 * ct + cl) * P + (pos & ~1)) = d;
                        db3[ct][r] += d0; db3[ct][r + 1] += d1;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sh[((r & 3) + 8 * (r >> 2) + 4 * lk) * HB_HS + li] = zr[ct][r] > 0.0f ? zr[ct][r] : 0.0f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const float dz = zr[ct][r] > 0.0f ? acc[r] : 0.0f;
                    if (pv) { store_raw(me.d_raw, (size_t)(cw + 32 * ct + cl) * P + pos, dz); db3[ct][r] += dz; }
                    sh[cl * HB_HS + li] = zr[ct][r] > 0.0f ? zr[ct][r] : 0.0f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- dW += g^T h : A[m = j][k = p] = g[p][j], B[k = p][n = c] = h[c][p], step s covers p = 2 s + lk
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    acc_w[jt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_g[(2 * s + lk) * HB_GS + 32 * jt + li], sh[li * HB_HS + 2 * s + lk], acc_w[jt][ct], 0, 0, 0);
                }
            }
        }
    }
    // ---- per-block partial sums
    float *pw = part_dw + (size_t)blockIdx.x * 64 * C;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * lk;
                pw[(size_t)j * C + cw + 32 * ct + li] = acc_w[jt][ct][r];
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = db3[ct][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);   // over the 32 positions of my half-wave
            if (li == 0) part_db3[(size_t)blockIdx.x * C + cw + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lk] = v;
        }
    }
    if (wave == 0 && blockIdx.y == 0) part_db[(size_t)blockIdx.x * 64 + lane] = db;
}

// sums the per-block partials in block order: dW rows -> dw_cls / dw_reg, db -> db_cls / db_reg, db3
__global__ __launch_bounds__(256) void rpn_head_tail_bwd_finalize_kernel(const float *__restrict__ part_dw, const float *__restrict__ part_db3,
                                                                         const float *__restrict__ part_db, int nblk, int C, int n_cls, int n_reg,
                                                                         float *__restrict__ dw_cls, float *__restrict__ db_cls,
                                                                         float *__restrict__ dw_reg, float *__restrict__ db_reg, float *__restrict__ db3)
{
    const int J = n_cls + n_reg;
    // 32 outputs per workgroup x 8 groups of partials: thread (g, o) sums partials g, g + 8, g + 16, ... of output o (four loads in
    // flight, neighbouring threads read neighbouring words), the eight group sums are added in group order through LDS.  The order
    // is fixed: bit-reproducible.  (One thread per output walking all 256 partials ran on 19 workgroups: 16 us for 17 MB.)
    __shared__ float s_p[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + o;
    const float *src = nullptr;
    size_t stride = 0;
    if (e < J * C) { src = part_dw + e; stride = (size_t)64 * C; }
    else if (e < J * C + C) { src = part_db3 + (e - J * C); stride = (size_t)C; }
    else if (e < J * C + C + J) { src = part_db + (e - J * C - C); stride = 64; }
    float v = 0.0f;
    if (src) {
        for (int b0 = g; b0 < nblk; b0 += 32) {
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = b0 + 8 * u < nblk ? src[(size_t)(b0 + 8 * u) * stride] : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) v += t[u];
        }
    }
    s_p[g][o] = v;
    __syncthreads();
    if (g != 0 || !src) return;
    v = s_p[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) v += s_p[q][o];
    if (e < J * C) {
        const int j = e / C, c = e - j * C;
        if (j < n_cls) dw_cls[(size_t)j * C + c] = v; else dw_reg[(size_t)(j - n_cls) * C + c] = v;
    } else if (e < J * C + C) {
        db3[e - J * C] = v;
    } else {
        const int j = e - J * C - C;
        if (j < n_cls) db_cls[j] = v; else db_reg[j - n_cls] = v;
    }
}

#define HEAD_BWD_MAX_BLOCKS 256
size_t frcnn_ws_head_bwd(int64_t C) { return (size_t)HEAD_BWD_MAX_BLOCKS * ((size_t)64 * C + C + 64) * sizeof(float); }

FRCNN_EXPORT int frcnn_rpn_head_tail_ml_bwd(const void *const *conv_raw_levels, void *const *d_raw_levels, int dtype, int C, const int64_t *P_levels,
                                            int n_levels, const float *b3, const float *w_cls, int n_cls, const float *w_reg, int n_reg,
                                            const float *g_cls, const float *g_reg, float *dw_cls, float *db_cls, float *dw_reg, float *db_reg,
                                            float *db3, void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(C > 0 && C % 256 == 0 && C <= 512, "rpn_head_tail_bwd: C=%d must be 256 or 512", C);
    FRCNN_REQUIRE(n_levels >= 1 && n_levels <= FRCNN_MAX_LEVELS && conv_raw_levels && d_raw_levels && P_levels, "rpn_head_tail_bwd: bad level table");
    FRCNN_REQUIRE(dtype == FRCNN_DTYPE_F32 || dtype == FRCNN_DTYPE_BF16, "rpn_head_tail_bwd: dtype %d (0 = f32, 1 = bf16)", dtype);
    FRCNN_REQUIRE(n_cls > 0 && n_reg > 0 && n_cls + n_reg <= 64, "rpn_head_tail_bwd: n_cls + n_reg = %d must be in (0, 64]", n_cls + n_reg);
    FRCNN_REQUIRE(b3 && w_cls && w_reg && g_cls && g_reg && dw_cls && db_cls && dw_reg && db_reg && db3 && workspace, "rpn_head_tail_bwd: NULL pointer");
    if (workspace_bytes < frcnn_ws_head_bwd(C))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "rpn_head_tail_bwd: workspace %zu < %zu bytes", workspace_bytes, frcnn_ws_head_bwd(C));
    HeadBwdLevels L;
    L.n_levels = n_levels;
    int64_t tiles = 0, pos = 0;
    for (int l = 0; l < FRCNN_MAX_LEVELS; ++l) {
        const int k = l < n_levels ? l : 0;
        FRCNN_REQUIRE(conv_raw_levels[k] && d_raw_levels[k] && P_levels[k] > 0 && P_levels[k] < ((int64_t)1 << 30), "rpn_head_tail_bwd: bad level %d", k);
        L.raw[l] = conv_raw_levels[k]; L.d_raw[l] = d_raw_levels[k]; L.P[l] = (int)P_levels[k];
        L.tile0[l] = (int)tiles; L.pos0[l] = (int)pos;
        if (l < n_levels) { tiles += (P_levels[k] + 31) / 32; pos += P_levels[k]; }
    }
    L.tile0[FRCNN_MAX_LEVELS] = (int)tiles;
    FRCNN_REQUIRE(tiles < ((int64_t)1 << 30), "rpn_head_tail_bwd: too many positions");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)(tiles < HEAD_BWD_MAX_BLOCKS ? tiles : HEAD_BWD_MAX_BLOCKS);
    float *part_dw = (float *)workspace;
    float *part_db3 = part_dw + (size_t)HEAD_BWD_MAX_BLOCKS * 64 * C;
    float *part_db = part_db3 + (size_t)HEAD_BWD_MAX_BLOCKS * C;
    const int nj = n_cls + n_reg <= 32 ? 1 : 2;
#define HEAD_BWD_LAUNCH(T_, NJ_)                                                                                                            \
    FRCNN_LAUNCH((rpn_head_tail_bwd_kernel<T_, NJ_, 1>), dim3((unsigned)nblk, (unsigned)(C / (32 * HB_NW))), dim3(64 * HB_NW), 0, s, L, C, (int)tiles, \
                 b3, w_cls, n_cls, w_reg, n_reg, g_cls, g_reg, part_dw, part_db3, part_db)
    if (dtype == FRCNN_DTYPE_F32) { if (nj == 1) HEAD_BWD_LAUNCH(float, 1); else HEAD_BWD_LAUNCH(float, 2); }
    else { if (nj == 1) HEAD_BWD_LAUNCH(unsigned short, 1); else HEAD_BWD_LAUNCH(unsigned short, 2); }
#undef HEAD_BWD_LAUNCH
    FRCNN_CHECK_LAUNCH("rpn_head_tail_bwd_kernel");
    const int n_out = (n_cls + n_reg) * C + C + n_cls + n_reg;
    FRCNN_LAUNCH(rpn_head_tail_bwd_finalize_kernel, dim3((unsigned)((n_out + 31) / 32)), dim3(256), 0, s, part_dw, part_db3, part_db,
                 nblk, C, n_cls, n_reg, dw_cls, db_cls, dw_reg, db_reg, db3);
    FRCNN_CHECK_LAUNCH("rpn_head_tail_bwd_finalize_kernel");
    return FRCNN_OK;
}
