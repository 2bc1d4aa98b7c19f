// Micro-check (developer tool): how exact is an fp32 product taken on the bf16 matrix cores?
// C[32][32] = A[32][K] . B[K][32] three ways on one wave: (a) v_mfma_f32_32x32x2_f32 (what the library runs), (b) every operand cut into three bf16 pieces
// (v = h + m + l exactly: truncation, exact residuals) and six v_mfma_f32_32x32x16_bf16 per 16 k (h h, h m, m h, m m, h l, l h; the three dropped products are
// <= 2^-24 of the term each), (c) two pieces and three products (the 2^-16 form, for scale).  Each against the float64 product of the same fp32 inputs.
//   hipcc --offload-arch=gfx950 -O3 -o split_product_check tools/dev/micro/split_product_check.hip && ./split_product_check
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

// A [32][K] row-major, B [K][32] row-major, C [3][32][32]
__global__ __launch_bounds__(64) void k(const float *__restrict__ A, const float *__restrict__ B, int K, float *__restrict__ C)
{
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    f32x16 c0, c1, c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = 0.0f;
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
        MM(c2, am, bh); MM(c2, ah, bm); MM(c2, ah, bh);
#undef MM
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        C[0 * 1024 + row * 32 + li] = c0[r];
        C[1 * 1024 + row * 32 + li] = c1[r];
        C[2 * 1024 + row * 32 + li] = c2[r];
    }
}

int main()
{
    const int Ks[] = {64, 256, 512, 4096};
    const char *dist[] = {"uniform (-1, 1)", "normal-ish, mixed magnitudes (x 2^(-8..8))", "all positive (0, 1): no cancellation"};
    for (int d = 0; d < 3; ++d)
        for (int K : Ks) {
            std::vector<float> A(32 * K), B(K * 32), C(3 * 1024);
            srand(11 + d);
            auto rnd = [&]() {
                const float u = (float)rand() / (float)RAND_MAX;
                if (d == 0) return 2.0f * u - 1.0f;
                if (d == 2) return u;
                return (2.0f * u - 1.0f) * ldexpf(1.0f, rand() % 17 - 8);
            };
            for (auto &x : A) x = rnd();
            for (auto &x : B) x = rnd();
            float *dA, *dB, *dC;
            (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
            (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
            (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double emax[3] = {0, 0, 0}, erms[3] = {0, 0, 0}, scale = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0, mag = 0;
                    for (int q = 0; q < K; ++q) { ref += (double)A[i * K + q] * (double)B[q * 32 + j]; mag += fabs((double)A[i * K + q] * (double)B[q * 32 + j]); }
                    scale += mag / 1024;
                    for (int v = 0; v < 3; ++v) {
                        const double e = fabs((double)C[v * 1024 + i * 32 + j] - ref) / mag;      // relative to the sum of the terms' magnitudes
                        emax[v] = e > emax[v] ? e : emax[v];
                        erms[v] += e * e / 1024;
                    }
                }
            printf("%-46s K %4d | error / sum|a b|  fp32 MFMA: max %.2e rms %.2e | six bf16 products: max %.2e rms %.2e | three: max %.2e rms %.2e\n", dist[d], K,
                   emax[0], sqrt(erms[0]), emax[1], sqrt(erms[1]), emax[2], sqrt(erms[2]));
            (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
        }
    return 0;
}
