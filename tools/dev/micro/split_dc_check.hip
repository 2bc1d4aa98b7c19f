// Micro-check (developer tool): does the split-product form carry a DC error?  Many 32 x 32 x K products of mixed-sign data: the MEAN signed error against float64
// (relative to the rms of the outputs) next to the rms error, for (a) v_mfma_f32_32x32x2_f32, (b) six bf16 products into one accumulator, (c) the large product
// (h h) into the main accumulator and the five small ones into a second one added at the end, (d) as (b) with every other 16-row block taken as (-A) . B into a
// second accumulator that is subtracted at the end (a rounding that always errs downwards cancels).
//   hipcc --offload-arch=gfx950 -O3 -o split_dc_check tools/dev/micro/split_dc_check.hip && ./split_dc_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cut8(const float *v, u32x4 &h, u32x4 &m, u32x4 &l)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned x0 = __builtin_bit_cast(unsigned, v[2 * p]), x1 = __builtin_bit_cast(unsigned, v[2 * p + 1]);
        h[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        const float r0 = v[2 * p] - __builtin_bit_cast(float, x0 & 0xFFFF0000u), r1 = v[2 * p + 1] - __builtin_bit_cast(float, x1 & 0xFFFF0000u);
        const unsigned y0 = __builtin_bit_cast(unsigned, r0), y1 = __builtin_bit_cast(unsigned, r1);
        m[p] = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
        const float s0 = r0 - __builtin_bit_cast(float, y0 & 0xFFFF0000u), s1 = r1 - __builtin_bit_cast(float, y1 & 0xFFFF0000u);
        l[p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
    }
}
// one wave per block: block b multiplies A_b [32][K] by B_b [K][32]
__global__ __launch_bounds__(64) void k(const float *__restrict__ A, const float *__restrict__ B, int K, float *__restrict__ C)
{
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    A += (size_t)blockIdx.x * 32 * K; B += (size_t)blockIdx.x * K * 32; C += (size_t)blockIdx.x * 4 * 1024;
    f32x16 c0, c1, c2, c2s, c3, c3n;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c2s[r] = c3[r] = c3n[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 2) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k0 + lh], B[(k0 + lh) * 32 + li], c0, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        float va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { va[j] = A[li * K + k0 + 8 * lh + j]; vb[j] = B[(k0 + 8 * lh + j) * 32 + li]; }
        u32x4 ah, am, al, bh, bm, bl;
        cut8(va, ah, am, al);
        cut8(vb, bh, bm, bl);
#define MM(c, x, y) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)
        MM(c1, al, bh); MM(c1, ah, bl); MM(c1, am, bm); MM(c1, am, bh); MM(c1, ah, bm); MM(c1, ah, bh);
        MM(c2s, al, bh); MM(c2s, ah, bl); MM(c2s, am, bm); MM(c2s, am, bh); MM(c2s, ah, bm); MM(c2, ah, bh);
        if ((k0 >> 4) & 1) {
            u32x4 nh, nm, nl;
#pragma unroll
            for (int p = 0; p < 4; ++p) { nh[p] = ah[p] ^ 0x80008000u; nm[p] = am[p] ^ 0x80008000u; nl[p] = al[p] ^ 0x80008000u; }
            MM(c3n, nl, bh); MM(c3n, nh, bl); MM(c3n, nm, bm); MM(c3n, nm, bh); MM(c3n, nh, bm); MM(c3n, nh, bh);
        } else { MM(c3, al, bh); MM(c3, ah, bl); MM(c3, am, bm); MM(c3, am, bh); MM(c3, ah, bm); MM(c3, ah, bh); }
#undef MM
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        C[0 * 1024 + row * 32 + li] = c0[r];
        C[1 * 1024 + row * 32 + li] = c1[r];
        C[2 * 1024 + row * 32 + li] = c2[r] + c2s[r];
        C[3 * 1024 + row * 32 + li] = c3[r] - c3n[r];
    }
}
int main()
{
    const int NB = 64;
    for (int K : {64, 256, 512}) {
        std::vector<float> A((size_t)NB * 32 * K), B((size_t)NB * K * 32), C((size_t)NB * 4 * 1024);
        srand(3);
        for (auto &x : A) x = 2.0f * rand() / RAND_MAX - 1.0f;
        for (auto &x : B) x = 2.0f * rand() / RAND_MAX - 1.0f;
        float *dA, *dB, *dC;
        (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, dA, dB, K, dC);
        (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double mean[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0}, osq = 0;
        for (int b = 0; b < NB; ++b)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0;
                    for (int q = 0; q < K; ++q) ref += (double)A[((size_t)b * 32 + i) * K + q] * (double)B[((size_t)b * K + q) * 32 + j];
                    osq += ref * ref;
                    for (int v = 0; v < 4; ++v) { const double e = (double)C[((size_t)b * 4 + v) * 1024 + i * 32 + j] - ref; mean[v] += e; sq[v] += e * e; }
                }
        const double n = (double)NB * 1024, orms = sqrt(osq / n);
        const char *nm[] = {"fp32 MFMA", "six bf16, one accumulator", "h h apart from the five small", "alternating sign, two accumulators"};
        for (int v = 0; v < 4; ++v)
            printf("K %3d %-36s mean error / rms(out) %+.2e (its own noise floor %.1e)   rms error / rms(out) %.2e\n", K, nm[v], mean[v] / n / orms, sqrt(sq[v] / n) / orms / sqrt(n), sqrt(sq[v] / n) / orms);
    }
    return 0;
}
