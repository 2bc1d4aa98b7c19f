// Micro-benchmark (developer tool): HBM read rate of the RoIPool backward's access pattern on gfx950 -- every workgroup reads, for each of R
// rows, one CHUNK of contiguous bytes at a row stride of 100 352 B (a [R][C][49] fp32 array read per channel group) -- against the chunk size.
//   hipcc --offload-arch=gfx950 -O3 -o build_dbg/chunk_read_rate tools/dev/micro/chunk_read_rate.hip && ./build_dbg/chunk_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// grid = number of chunks per row; block 512 = 8 waves; wave w reads rows w, w + 8, ...; LPR = loads per row and lane of 8 bytes (lanes < chunk / 8 / LPR ...)
template <int U>
__global__ __launch_bounds__(512) void k(const float2 *__restrict__ src, int R, size_t row_stride8, int chunk8, float *out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * chunk8;
    float acc = 0.0f;
    for (int rb = wave; rb < R; rb += 8 * U) {
        float2 v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = min(rb + 8 * u, R - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(lane + 64 * j, chunk8 - 1);
                v[u][j] = (lane + 64 * j < chunk8 || j == 0) ? src[(size_t)r * row_stride8 + base + i] : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[u][j].x + v[u][j].y;
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

int main()
{
    const int R = 128, C = 512;
    const size_t row_bytes = (size_t)C * 49 * 4;
    const size_t total = (size_t)R * row_bytes;
    float2 *d; float *o;
    hipMalloc(&d, total + 4096); hipMalloc(&o, 4096 * 4);
    hipMemset(d, 0, total + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // second buffer to flush caches between repetitions
    float2 *flush; hipMalloc(&flush, 512u << 20);
    for (int chunk : {196, 392, 784, 1568, 3136, 6272}) {
        const int nchunk = (int)(row_bytes / chunk);
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemsetAsync(flush, rep, 512u << 20, 0);           // evict the array from L2 / Infinity Cache
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<16>, dim3(nchunk), dim3(512), 0, 0, d, R, row_bytes / 8, chunk / 8, o);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("chunk %5d B x %4d workgroups: best %6.2f us, mean %6.2f us  -> %5.2f TB/s (12.85 MB, cold caches, event-timed incl. launch)\n", chunk, nchunk, best * 1e3, sum / 5 * 1e3,
               total / (best * 1e-3) * 1e-12);
    }
    return 0;
}
