// Micro-benchmark (developer tool): what rate of v_mfma_f32_32x32x2_f32 does an MI355X SUSTAIN, and at what shader clock?
// A launch is nothing but matrix instructions: every wave issues `iters` rounds of four independent MFMAs (four 16-register accumulators, operands in
// registers, no memory, no LDS).  Reported per (waves per SIMD, launch length): TFLOP/s over the whole chip by HIP events, the shader clock seen by the kernel
// (s_memtime cycles of workgroup 0 over the event time is NOT it -- s_memtime counts a fixed 100 MHz; the clock is derived from the instruction count:
// one 32x32x2 fp32 MFMA occupies a SIMD's matrix pipe for 64 cycles, 16 passes of 4), and the same after the chip has been kept busy for a while.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate tools/dev/micro/mfma_f32_rate.hip && ./mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float *out, const float *__restrict__ rnd)
{
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.0f;
    // RANDOM: operands that toggle like real data (eight random values per lane and side, a different pair every MFMA; |x| < 1, sums stay finite);
    // otherwise two small constants per lane (the matrix pipe's datapath hardly switches: the least power an MFMA can draw)
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = RANDOM ? rnd[(threadIdx.x * 16 + e + blockIdx.x * 64) & 65535] : (float)(threadIdx.x & 3) * 0.25f;
        b[e] = RANDOM ? rnd[(threadIdx.x * 16 + 8 + e + blockIdx.x * 64) & 65535] : (float)(threadIdx.x & 7) * 0.125f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + q) & 7], b[(u + q * 3) & 7], acc[q], 0, 0, 0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[q][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *d_out, *d_rnd;
    hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4);
    hipMalloc(&d_rnd, 65536 * 4);
    {
        static float h[65536];
        srand(7);
        for (int i = 0; i < 65536; ++i) h[i] = ((float)rand() / (float)RAND_MAX - 0.5f) * 1.9f * (i & 1 ? 1.0f : 1e-3f);   // mixed magnitudes: exponents toggle too
        hipMemcpy(d_rnd, h, sizeof(h), hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d; one v_mfma_f32_32x32x2_f32 = 4096 flop = 64 cycles of a SIMD's matrix pipe; nominal peak at 2.4 GHz: %.1f TFLOP/s\n", cus, cus * 4 * 64.0 * 2.4e9 / 1e12);
    const int per_cu[] = {1, 2, 4};
    const int iters_list[] = {64, 512, 4096, 65536};
    for (int random = 0; random < 2; ++random)
    for (int pc : per_cu)
        for (int iters : iters_list) {
            auto kern = random ? mfma_loop<true> : mfma_loop<false>;
            const int grid = cus * pc;
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters, d_out, d_rnd);
            hipDeviceSynchronize();
            float best = 1e30f, sum = 0.0f;
            const int reps = iters >= 65536 ? 3 : 10;
            for (int r = 0; r < reps; ++r) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, iters, d_out, d_rnd);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best; sum += ms;
            }
            const double mfmas_per_simd = (double)pc * iters * 16.0;           // waves per SIMD x MFMAs per wave
            const double flops = (double)grid * 4 * iters * 16.0 * 4096.0;
            const double mean = sum / reps;
            printf("%s operands  workgroups per CU %d (waves per SIMD %d)  iters %6d : %9.1f us (min %9.1f)  %6.1f TFLOP/s  = matrix pipe busy at %.2f GHz if never idle\n", random ? "random  " : "constant", pc, pc, iters,
                   mean * 1e3, best * 1e3, flops / (mean * 1e-3) / 1e12, mfmas_per_simd * 64.0 / (mean * 1e-3) / 1e9);
        }
    // back to back for ~2 s, then the long launch again: does the rate sag once the part is warm?
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(mfma_loop<true>, dim3(cus * 2), dim3(256), 0, 0, 65536, d_out, d_rnd);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<true>, dim3(cus * 2), dim3(256), 0, 0, 65536, d_out, d_rnd);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("random operands, after 40 long launches back to back: 2 per CU, iters 65536: %9.1f us  %6.1f TFLOP/s\n", ms * 1e3, (double)cus * 2 * 4 * 65536 * 16.0 * 4096.0 / (ms * 1e-3) / 1e12);
    return 0;
}
