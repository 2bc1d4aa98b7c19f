// Reproducer (developer tool): clang of ROCm 7.2 for gfx950 miscompiles the second of two chained PACKED fp32 subtractions whose subtrahend is a vector built from
// two scalar (bits & mask) values: it emits ONE v_and and `v_pk_add_f32 ... op_sel_hi:[1,0]` -- the first element's masked value is used for both halves --
// so q.y = r.y - (r.x & mask).  Found in the split-product cut (csrc/rpn_conv_f32.hip wn_cut8 uses scalar subtractions for this reason).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_sub_miscompile tools/dev/micro/pk_sub_miscompile.hip && ./pk_sub_miscompile   (expected: mismatches 0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *v, float *o_pk, float *o_sc, int n)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 >= n) return;
    const unsigned x0 = __builtin_bit_cast(unsigned, v[i]), x1 = __builtin_bit_cast(unsigned, v[i + 1]);
    const f32x2 r = f32x2{v[i], v[i + 1]} - f32x2{__builtin_bit_cast(float, x0 & 0xFFFF0000u), __builtin_bit_cast(float, x1 & 0xFFFF0000u)};
    const unsigned r0 = __builtin_bit_cast(unsigned, r.x), r1 = __builtin_bit_cast(unsigned, r.y);
    const f32x2 q = r - f32x2{__builtin_bit_cast(float, r0 & 0xFFFF0000u), __builtin_bit_cast(float, r1 & 0xFFFF0000u)};
    o_pk[i] = r.x; o_pk[i + 1] = r.y; o_pk[n + i] = q.x; o_pk[n + i + 1] = q.y;
    const float ra = v[i] - __builtin_bit_cast(float, x0 & 0xFFFF0000u), rb = v[i + 1] - __builtin_bit_cast(float, x1 & 0xFFFF0000u);
    const float qa = ra - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ra) & 0xFFFF0000u), qb = rb - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rb) & 0xFFFF0000u);
    o_sc[i] = ra; o_sc[i + 1] = rb; o_sc[n + i] = qa; o_sc[n + i + 1] = qb;
}
int main()
{
    const int n = 1 << 16;
    std::vector<float> h(n), a(2 * n), b(2 * n);
    srand(1);
    for (auto &x : h) x = (2.0f * rand() / RAND_MAX - 1.0f) * (rand() % 2 ? 1.0f : 1e-3f);
    float *dv, *da, *db;
    hipMalloc(&dv, n * 4); hipMalloc(&da, 2 * n * 4); hipMalloc(&db, 2 * n * 4);
    hipMemcpy(dv, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, dv, da, db, n);
    hipMemcpy(a.data(), da, 2 * n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, 2 * n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2 * n; ++i) if (a[i] != b[i]) { if (bad < 5) printf("i %d v %.9g packed %.9g scalar %.9g\n", i, h[i % n], a[i], b[i]); ++bad; }
    printf("mismatches %d of %d\n", bad, 2 * n);
    return 0;
}
