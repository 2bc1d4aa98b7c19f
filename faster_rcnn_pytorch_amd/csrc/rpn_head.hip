// rpn_head.hip -- the tail of RegionProposalNetwork.forward (models/model.py:79-83, models/new_model.py:109-113)
// as one MFMA kernel:   h = relu(conv3x3_raw + b3)  ->  cls = Wc h + bc,  reg = Wr h + br  ->  written directly in the
// reference's permute(0,2,3,1).contiguous().view(B,-1,2|4) layout ([P, 2A] and [P, 4A] row-major).
//
// The reference runs this as bias-add, ReLU, two 1x1 convolutions, two bias-adds and two NCHW->NHWC transposing
// copies (10-12 launches, ~60 us on MI355X); here the 1x1 heads are a [P x C] . [C x (2A+4A)] contraction on the
// matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32: D = fma chain in k order, no reduced precision), the
// bias + ReLU of the 3x3 output is applied while the A operand is loaded, and the epilogue stores NHWC.
//   block = 4 waves = one tile of 32 positions; wave w takes the k-quarter [w*C/4, (w+1)*C/4); the four partial
//   accumulators are summed through LDS.  A operand: lane l reads raw[k0 + (l>>5)][pos0 + (l&31)] (two 128-B
//   segments per wave-load, straight from the NCHW conv output); B operand: lane l reads W[j = l&31][k0 + (l>>5)]
//   (each lane streams along its own weight row: L1-resident lines).
// The 3x3 convolution itself stays on MIOpen this round (its fp32 igemm already runs the MFMA path at ~100 TF/s).
#include "frcnn_common.h"
#include "frcnn_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void rpn_head_tail_kernel(const float *__restrict__ raw, int C, int P, const float *__restrict__ b3,
                                                            const float *__restrict__ w_cls, const float *__restrict__ b_cls, int n_cls,
                                                            const float *__restrict__ w_reg, const float *__restrict__ b_reg, int n_reg,
                                                            float *__restrict__ out_cls, float *__restrict__ out_reg)
{
    __shared__ float s_acc[3][32][64];                    // partial accumulators of waves 1..3: [wave-1][reg (2 tiles x 16)][lane]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int p0 = blockIdx.x * 32;
    const int li = lane & 31, lk = lane >> 5;
    const int pos = min(p0 + li, P - 1);
    const int kq = C / 4;
    const int kbeg = wave * kq, kend = kbeg + kq;
    // weight row of this lane for the two output tiles (j = li and j = 32 + li); rows >= n_cls + n_reg contribute zeros
    const int j0 = li, j1 = 32 + li;
    const float *wrow0 = j0 < n_cls ? w_cls + (size_t)j0 * C : (j0 < n_cls + n_reg ? w_reg + (size_t)(j0 - n_cls) * C : nullptr);
    const float *wrow1 = j1 < n_cls ? w_cls + (size_t)j1 * C : (j1 < n_cls + n_reg ? w_reg + (size_t)(j1 - n_cls) * C : nullptr);
    f32x16 acc0 = {0}, acc1 = {0};
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        float a[8], bw0[8], bw1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                     // 24 independent loads in flight before the first MFMA
            const int k = k0 + 2 * u + lk;
            a[u] = raw[(size_t)k * P + pos] + b3[k];
            bw0[u] = wrow0 ? wrow0[k] : 0.0f;
            bw1[u] = wrow1 ? wrow1[k] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float h = a[u] > 0.0f ? a[u] : 0.0f;    // ReLU of the 3x3 output (bias already added)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(h, bw0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(h, bw1[u], acc1, 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s_acc[wave - 1][r][lane] = acc0[r]; s_acc[wave - 1][16 + r][lane] = acc1[r]; }
    }
    __syncthreads();
    if (wave == 0) {
        const float bias0 = j0 < n_cls ? b_cls[j0] : (j0 < n_cls + n_reg ? b_reg[j0 - n_cls] : 0.0f);
        const float bias1 = j1 < n_cls ? b_cls[j1] : (j1 < n_cls + n_reg ? b_reg[j1 - n_cls] : 0.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // C/D layout of the 32x32 MFMA: column j = lane & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int p = p0 + i;
            const float v0 = acc0[r] + s_acc[0][r][lane] + s_acc[1][r][lane] + s_acc[2][r][lane] + bias0;
            const float v1 = acc1[r] + s_acc[0][16 + r][lane] + s_acc[1][16 + r][lane] + s_acc[2][16 + r][lane] + bias1;
            if (p < P) {
                if (j0 < n_cls) out_cls[(size_t)p * n_cls + j0] = v0;
                else if (j0 < n_cls + n_reg) out_reg[(size_t)p * n_reg + (j0 - n_cls)] = v0;
                if (j1 < n_cls) out_cls[(size_t)p * n_cls + j1] = v1;
                else if (j1 < n_cls + n_reg) out_reg[(size_t)p * n_reg + (j1 - n_cls)] = v1;
            }
        }
    }
}

FRCNN_EXPORT int frcnn_rpn_head_tail_fwd(const float *conv_raw, int C, int64_t P, const float *b3, const float *w_cls, const float *b_cls,
                                         int n_cls, const float *w_reg, const float *b_reg, int n_reg, float *out_cls, float *out_reg,
                                         void *stream)
{
    FRCNN_REQUIRE(C > 0 && C % 64 == 0, "rpn_head_tail: C=%d must be a positive multiple of 64", C);
    FRCNN_REQUIRE(P > 0 && P < ((int64_t)1 << 30), "rpn_head_tail: bad P");
    FRCNN_REQUIRE(n_cls > 0 && n_reg > 0 && n_cls + n_reg <= 64, "rpn_head_tail: n_cls + n_reg = %d must be in (0, 64]", n_cls + n_reg);
    FRCNN_REQUIRE(conv_raw && b3 && w_cls && b_cls && w_reg && b_reg && out_cls && out_reg, "rpn_head_tail: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(KID_RPN_HEAD_TAIL, rpn_head_tail_kernel, dim3((unsigned)((P + 31) / 32)), dim3(256), 0, s, conv_raw, C, (int)P, b3, w_cls, b_cls,
                 n_cls, w_reg, b_reg, n_reg, out_cls, out_reg);
    FRCNN_CHECK_LAUNCH("rpn_head_tail_kernel");
    return FRCNN_OK;
}
