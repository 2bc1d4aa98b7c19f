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
            const float bias = j < n_cls ? b_cls[j] : (j < n_cls + n_reg ? b_reg[j - n_cls] : 0.0f);
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
    FRCNN_LAUNCH(KID_RPN_HEAD_TAIL, (rpn_head_tail_kernel<TIN, NT_, BF16MM, W_>), dim3((unsigned)tiles), dim3(64 * W_), 0, s, L, C, b3, w_cls, \
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
