// conv_c3.hip -- the first convolution of the backbone, nn.Conv2d(3, Cout, 3, padding=1) + ReLU: `vgg16.features[0]` + `[1]` behind
// models/model.py:279-281, on the 3 x 600 x 1000 image.  With three input channels it is not a contraction for the matrix cores (K = 27) but a
// byte mover: 7 MB in, 154 MB out, 2 GFLOP.  The vendor path ran it as a Winograd kernel + a bias pass + a ReLU pass forward (175 us) and, backward,
// a threshold pass + an igemm weight gradient behind an NCHW -> NHWC transpose + a 146-us reduction for the bias gradient (370 us).  Here:
//   conv3x3_c3_fwd_kernel    thread = output pixel of a 256-wide row piece: its 27 inputs from an LDS-staged 3 x 3 x 258 window into registers, then
//                            a loop over the output channels with the weights as SCALAR operands (uniform loads) -- y = relu(b + sum w x), one
//                            coalesced store per channel.  fp32 fmaf chain in (ci, ky, kx) order.  The signs of the outputs go out as one 64-bit word
//                            per pixel and 64 channels (5 MB instead of the 154 MB of activations the ReLU's backward would read).
//   conv3x3_c3_wgrad_kernel  dw[co][ci][ky][kx] = sum_p g[co][p] x[ci][p + off], db[co] = sum_p g[co][p], g = dy where the forward's sign word has the
//                            channel's bit.  block = (4 image rows, 16 output channels); wave = 4 channels, its lanes walk a row 256 pixels at a
//                            time with sixteen gradient loads in flight, 4 x 28 running sums per lane, one shuffle reduction per wave at the end ->
//                            partials [block of rows][co][28] (one DPP reduction per wave and block; rows per block chosen per call);  conv3x3_c3_wgrad_finalize_kernel adds them in order (bit-reproducible).
// The input gradient is never needed (the input is the image).
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(conv_c3);

#define C3_TW 256                      // pixels per forward block

__global__ __launch_bounds__(256) void conv3x3_c3_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int H, int W, int Cout,
                                                             const float *__restrict__ w, const float *__restrict__ bias, int relu,
                                                             unsigned long long *__restrict__ bits)
{
    __shared__ float s[9][C3_TW + 4];                                     // [ci * 3 + ky][column x0 - 1 .. x0 + 256]
    extern __shared__ __attribute__((aligned(16))) float sw[];            // [Cout][28]: the 27 weights of a channel + its bias (16-byte rows)
    const int row = blockIdx.y, x0 = blockIdx.x * C3_TW, tid = threadIdx.x;
    for (int e = tid; e < Cout * 28; e += 256) {
        const int co = e / 28, k = e - co * 28;
        sw[e] = k < 27 ? w[(size_t)co * 27 + k] : (bias ? bias[co] : 0.0f);
    }
    {
        float v[10];                                                      // ten loads in flight per thread, then the LDS writes
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int e = tid + 256 * j, r = e / (C3_TW + 2), q = e - r * (C3_TW + 2), ci = r / 3, ky = r - 3 * ci, yy = row - 1 + ky, xx = x0 - 1 + q;
            v[j] = (r < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[((size_t)ci * H + yy) * W + xx] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int e = tid + 256 * j, r = e / (C3_TW + 2), q = e - r * (C3_TW + 2);
            if (r < 9) s[r][q] = v[j];
        }
    }
    __syncthreads();
    const int xx = x0 + tid;
    if (xx >= W) return;
    float v[27];
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v[r * 3 + kx] = s[r][tid + kx];
    float *yo = y + (size_t)row * W + xx;
    const size_t plane = (size_t)H * W;
    unsigned long long word = 0;
    for (int c2 = 0; c2 < Cout; c2 += 2) {                                // two independent chains per pass; the weights come as LDS broadcasts
        const bool two = c2 + 1 < Cout;                                   // (four channels' weights as scalar operands overflowed the SGPR file)
        const float4 *w0 = (const float4 *)&sw[c2 * 28], *w1 = (const float4 *)&sw[(two ? c2 + 1 : c2) * 28];
        float acc0 = sw[c2 * 28 + 27], acc1 = sw[(two ? c2 + 1 : c2) * 28 + 27];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const float4 a = w0[q], c = w1[q];
            const float aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q * 4 + e < 27) { acc0 = fmaf(aw[e], v[q * 4 + e], acc0); acc1 = fmaf(cw[e], v[q * 4 + e], acc1); }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = c2 + j;
            if (co < Cout) {
                const float pre = j ? acc1 : acc0, o = relu && pre < 0.0f ? 0.0f : pre;      // NaN stays NaN (torch.relu)
                yo[(size_t)co * plane] = o;
                word |= o > 0.0f ? 1ull << (co & 63) : 0ull;
                if (bits && ((co & 63) == 63 || co == Cout - 1)) { bits[(size_t)(co >> 6) * plane + (size_t)row * W + xx] = word; word = 0; }
            }
        }
    }
}

// sum over the wave's 64 lanes on DPP row shifts / broadcasts (plain VALU adds, no LDS crossbar), result in lane 63; fixed order
__device__ __forceinline__ float c3_wave_sum(float v)
{
#define C3_DPP_ADD(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    C3_DPP_ADD(0x111, 0xf);    // row_shr:1
    C3_DPP_ADD(0x112, 0xf);    // row_shr:2
    C3_DPP_ADD(0x114, 0xf);    // row_shr:4
    C3_DPP_ADD(0x118, 0xf);    // row_shr:8   -> lane 15 of every row holds the row's sum
    C3_DPP_ADD(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
    C3_DPP_ADD(0x143, 0xc);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
#undef C3_DPP_ADD
    return v;
}

#define C3_WAVES 8                     // waves per block: 32 output channels.  The image rows per block are chosen per call (frcnn_conv3x3_c3_wgrad) so that the launch is ONE round of
                                       // at most one block per CU: with four waves and four rows 600 blocks met 512 slots (two rounds, the second 17 % full: 120 us);
                                       // eight waves x five rows = 240 blocks: 80 us
#define C3_CPW 4                       // output channels per wave: 4 x 28 running sums per lane (two channels per wave, eight waves: 233 us against 189)
// dynamic LDS: 3 x (rows + 2) x (W + 2) floats (the block's image rows of the three input channels with their halo, staged once).  BITS is a
// template flag: with a run-time `bits != NULL` test in front of every sign-word load the kernel took 170 us against 118.  Measured and not kept: all six
// rows of a block staged at once with eight waves and the next batch's loads issued ahead (218 registers, one block per CU: 212 us against 146);
// the 28 sums split over two waves (4 x 14 per lane, four waves per SIMD, the gradient loaded by both: 228 us against 189 -- the loads, not the
// arithmetic, are what the kernel waits for: without the sign words it takes 127).
template <bool BITS>
__global__ __launch_bounds__(64 * C3_WAVES) void conv3x3_c3_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy, int H, int W, int Cout,
                                                                       const unsigned long long *__restrict__ bits, float *__restrict__ part, int rows)
{
    extern __shared__ float s[];                                          // [3][rows + 2][W + 2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ws = W + 2, NT = 64 * C3_WAVES;
    const int co0 = (blockIdx.y * C3_WAVES + wave) * C3_CPW;              // this wave's output channels (a wave past Cout only helps staging)
    const size_t plane = (size_t)H * W;
    // 27 tap sums + the gradient's own sum (tap 27 multiplies by 1.0: the same rounding as a plain add) as 14 PAIRS on packed fp32 (v_pk_fma_f32: two
    // FMAs per lane and instruction: 79 -> 73 us.  The FORWARD's chains as packed pairs were slower, 67 against 57 us, as one chain or as two)
    typedef float c3_f32x2 __attribute__((ext_vector_type(2)));
    c3_f32x2 acc[C3_CPW][14];
#pragma unroll
    for (int j = 0; j < C3_CPW; ++j)
#pragma unroll
        for (int k = 0; k < 14; ++k) acc[j][k] = (c3_f32x2){0.0f, 0.0f};
    // the block's rows + 2 image rows of the three channels, staged ONCE: [ci][row r0 - 1 .. r0 + rows][W + 2] (per image row the staging's
    // five dependent load batches were a third of the kernel's time)
    const int r0 = blockIdx.x * rows, nrow = min(rows, H - r0), RS = rows + 2, n_el = 3 * RS * ws;
    for (int base = tid; base < n_el; base += NT * 8) {                   // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = base + NT * j, rr = e / ws, q = e - rr * ws, ci = rr / RS, ry = rr - RS * ci, yy = r0 - 1 + ry, xx = q - 1;
            v[j] = (e < n_el && yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[((size_t)ci * H + yy) * W + xx] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = base + NT * j;
            if (e < n_el) s[e] = v[j];
        }
    }
    __syncthreads();
    if (co0 >= Cout) return;
    // iteration = (row, 256-pixel piece): 4 pixels x 4 channels of gradient (+ the sign words) per lane, loaded one iteration AHEAD at clamped
    // addresses (unconditional loads; a lane past the row's end counts zero): the next piece's round trip runs under this piece's 432 FMAs
    const int nxb = (W + 255) / 256;
    auto load = [&](int ry, int xb, float (&gq)[4][C3_CPW], unsigned (&mq)[4]) {
        const float *g0 = dy + (size_t)co0 * plane + (size_t)(r0 + ry) * W;
        const unsigned *b0 = BITS ? (const unsigned *)(bits + (size_t)(co0 >> 6) * plane + (size_t)(r0 + ry) * W) + ((co0 & 63) >> 5) : nullptr;      // the 32-bit half that holds this wave's four bits
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xc = min(xb + 64 * i + lane, W - 1);
            mq[i] = BITS ? b0[2 * xc] : ~0u;
#pragma unroll
            for (int j = 0; j < C3_CPW; ++j) gq[i][j] = g0[(size_t)j * plane + xc];
        }
    };
    float g[4][C3_CPW];
    unsigned mb[4];                                                       // the wave's channels' sign bits of each pixel (co0 % C3_CPW == 0: never across words)
    load(0, 0, g, mb);
    for (int ry = 0, xbi = 0;;) {
        const int xb = xbi * 256;
        int ry2 = ry, xbi2 = xbi + 1;
        if (xbi2 == nxb) { xbi2 = 0; ++ry2; }
        const bool more = ry2 < nrow;
        float gn[4][C3_CPW];
        unsigned mn[4];
        if (more) load(ry2, xbi2 * 256, gn, mn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xx = xb + 64 * i + lane;
            if (xb + 64 * i < W) {                                        // uniform
                float v[27];
                const int xc = xx < W ? xx : W - 1;                       // lanes past the row read a valid column (their gradient counts zero)
#pragma unroll
                for (int r = 0; r < 9; ++r)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) v[r * 3 + kx] = s[((r / 3) * RS + ry + r % 3) * ws + xc + kx];      // tap (ci = r / 3, ky = r % 3)
#pragma unroll
                for (int j = 0; j < C3_CPW; ++j) {
                    const float gg = (xx < W && (!BITS || ((mb[i] >> ((co0 & 31) + j)) & 1u))) ? g[i][j] : 0.0f;      // the ReLU's backward from the forward's sign word
#pragma unroll
                    for (int k = 0; k < 14; ++k) {
                        const c3_f32x2 vv = {v[2 * k], k < 13 ? v[2 * k + 1] : 1.0f}, g2 = {gg, gg};
                        acc[j][k] = __builtin_elementwise_fma(g2, vv, acc[j][k]);
                    }
                }
            }
        }
        if (!more) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mb[i] = mn[i];
#pragma unroll
            for (int j = 0; j < C3_CPW; ++j) g[i][j] = gn[i][j];
        }
        ry = ry2; xbi = xbi2;
    }
#pragma unroll
    for (int j = 0; j < C3_CPW; ++j)
#pragma unroll
        for (int k = 0; k < 28; ++k) {
            const float t = c3_wave_sum(acc[j][k >> 1][k & 1]);
            if (lane == 63) part[((size_t)blockIdx.x * Cout + co0 + j) * 28 + k] = t;
        }
}

// thread = (co, k): the rows' partials in row order; k < 27 -> dw[co][k], k = 27 -> db[co]
__global__ __launch_bounds__(256) void conv3x3_c3_wgrad_finalize_kernel(const float *__restrict__ part, int H, int Cout, float *__restrict__ dw,
                                                                        float *__restrict__ db)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= Cout * 28) return;
    const size_t stride = (size_t)Cout * 28;
    float t = 0.0f;
    int r = 0;
    for (; r + 8 <= H; r += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(r + j) * stride + o];
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
    }
    for (; r < H; ++r) t += part[(size_t)r * stride + o];
    const int co = o / 28, k = o - co * 28;
    if (k < 27) dw[(size_t)co * 27 + k] = t;
    else if (db) db[co] = t;
}

static int c3_check(const void *x, const void *y, int H, int W, int Cout, const void *w, const char *what)
{
    FRCNN_REQUIRE(x && y && w, "%s: NULL pointer", what);
    FRCNN_REQUIRE(H > 0 && W > 0 && Cout > 0 && Cout <= 256 && (long long)H * W * Cout < (1ll << 31), "%s: bad size %d x %d x %d", what, Cout, H, W);
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_conv3x3_c3_fwd(const float *x_dev, float *y_dev, int H, int W, int Cout, const float *w_dev, const float *bias_dev, int relu,
                                      unsigned long long *relu_bits_dev, void *stream)
{
    int rc = c3_check(x_dev, y_dev, H, W, Cout, w_dev, "conv3x3_c3_fwd");
    if (rc) return rc;
    FRCNN_LAUNCH(conv3x3_c3_fwd_kernel, dim3((unsigned)((W + C3_TW - 1) / C3_TW), (unsigned)H), dim3(256), (size_t)Cout * 28 * sizeof(float), (hipStream_t)stream, x_dev, y_dev, H, W, Cout,
                 w_dev, bias_dev, relu, relu_bits_dev);
    FRCNN_CHECK_LAUNCH("conv3x3_c3_fwd_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT size_t frcnn_conv3x3_c3_wgrad_workspace(int H, int Cout)
{
    return (H > 0 && Cout > 0) ? (size_t)H * Cout * 28 * sizeof(float) : 0;             // one partial per row block; a block has at least one row
}

FRCNN_EXPORT int frcnn_conv3x3_c3_wgrad(const float *x_dev, const float *dy_dev, int H, int W, int Cout, const unsigned long long *relu_bits_dev,
                                        float *dw_dev, float *dbias_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = c3_check(x_dev, dy_dev, H, W, Cout, dw_dev, "conv3x3_c3_wgrad");
    if (rc) return rc;
    FRCNN_REQUIRE(dw_dev && workspace, "conv3x3_c3_wgrad: NULL pointer");
    if (Cout % 4 != 0 || W > 1700)      // LDS: 3 (rows + 2)(W + 2) floats
        return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "conv3x3_c3_wgrad: Cout = %d must be a multiple of 4 and W = %d at most 1700", Cout, W);
    if (workspace_bytes < frcnn_conv3x3_c3_wgrad_workspace(H, Cout))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "conv3x3_c3_wgrad: workspace %zu < %zu bytes", workspace_bytes, frcnn_conv3x3_c3_wgrad_workspace(H, Cout));
    hipStream_t s = (hipStream_t)stream;
    float *part = (float *)workspace;
    // rows per block: the fewest that keep the launch within one block per CU (one round), as far as the window fits the LDS
    const int groups = (Cout + C3_CPW * C3_WAVES - 1) / (C3_CPW * C3_WAVES);
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int fit = (int)((150 * 1024) / (12 * (size_t)(W + 2))) - 2;
    const int rows = std::max(1, std::min(fit, (int)(((long long)H * groups + cus - 1) / cus)));
    const int nb = (H + rows - 1) / rows;
    const dim3 wg((unsigned)nb, (unsigned)groups);
    const size_t wl = (size_t)3 * (rows + 2) * (W + 2) * sizeof(float);
    // the window may need more than the 64 KB a launch gets without asking (84 KB at 600 x 1000): opt in, per call (the attribute is per device)
    if (relu_bits_dev) (void)hipFuncSetAttribute((const void *)conv3x3_c3_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
    else (void)hipFuncSetAttribute((const void *)conv3x3_c3_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
    if (relu_bits_dev) FRCNN_LAUNCH(conv3x3_c3_wgrad_kernel<true>, wg, dim3(64 * C3_WAVES), wl, s, x_dev, dy_dev, H, W, Cout, relu_bits_dev, part, rows);
    else FRCNN_LAUNCH(conv3x3_c3_wgrad_kernel<false>, wg, dim3(64 * C3_WAVES), wl, s, x_dev, dy_dev, H, W, Cout, relu_bits_dev, part, rows);
    FRCNN_CHECK_LAUNCH("conv3x3_c3_wgrad_kernel");
    FRCNN_LAUNCH(conv3x3_c3_wgrad_finalize_kernel, dim3((unsigned)((Cout * 28 + 255) / 256)), dim3(256), 0, s, part, nb, Cout, dw_dev, dbias_dev);
    FRCNN_CHECK_LAUNCH("conv3x3_c3_wgrad_finalize_kernel");
    return FRCNN_OK;
}
