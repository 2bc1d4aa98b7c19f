// Micro-benchmark (developer tool): cost of LDS float atomics against plain LDS read-modify-write on gfx950.
// One 512-thread workgroup per CU; every wave works on a private 16 KB region.  Reports shader cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate tools/dev/micro/lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int OP>   // 0 ds_add_f32, 1 ds_add_u32, 2 read + add + write (non-atomic), 3 ds_add_rtn_f32
__global__ __launch_bounds__(512) void k(const int *__restrict__ addr, int iters, float *out, long long *cycles, int nactive = 64)
{
    __shared__ float lds[8 * 4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *mine = lds + wave * 4096;
    for (int i = lane; i < 4096; i += 64) mine[i] = 0.0f;
    __syncthreads();
    int a[16];
    for (int i = 0; i < 16; ++i) a[i] = addr[i * 64 + lane];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) { if (lane < nactive) atomicAdd(&mine[a[i]], 1.0f); }
            else if (OP == 1) atomicAdd((unsigned *)&mine[a[i]], 1u);
            else if (OP == 2) { float v = mine[a[i]]; mine[a[i]] = v + 1.0f; }
            else if (OP == 3) acc += atomicAdd(&mine[a[i]], 1.0f);
            else if (OP == 4) acc += __int_as_float(__builtin_amdgcn_ds_bpermute((a[i] & 63) * 4, __float_as_int(acc) + i));      // independent-ish shuffles
            else { const int v = __builtin_amdgcn_ds_bpermute((a[i] & 63) * 4, i + it); acc += (float)(v == a[i]); }
        }
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = mine[lane] + acc;
}

int main()
{
    const int iters = 64;
    int *d_addr; float *d_out; long long *d_cyc;
    hipMalloc(&d_addr, 1024 * 4); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 256 * 8);
    const char *names[] = {"consecutive (conflict-free)", "all lanes one address", "random in 4096", "random in 64 (clustered)", "stride 32 (32-way bank conflict)", "pairs share an address"};
    for (int pat = 0; pat < 6; ++pat) {
        std::vector<int> h(1024);
        srand(1);
        for (int i = 0; i < 16; ++i)
            for (int l = 0; l < 64; ++l) {
                int v;
                switch (pat) {
                case 0: v = (i * 64 + l) % 4096; break;
                case 1: v = i; break;
                case 2: v = rand() % 4096; break;
                case 3: v = i * 64 + rand() % 64; break;
                case 4: v = (l * 32 + i) % 4096; break;
                default: v = (i * 64 + l / 2) % 4096; break;
                }
                h[i * 64 + l] = v;
            }
        hipMemcpy(d_addr, h.data(), 4096, hipMemcpyHostToDevice);
        printf("%-36s", names[pat]);
        for (int op = 0; op < 4; ++op) {
            for (int rep = 0; rep < 2; ++rep) {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc);
                if (op == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc);
            }
            hipDeviceSynchronize();
            long long c[256];
            hipMemcpy(c, d_cyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += c[i];
            // 8 waves x iters x 16 wave-instructions per workgroup share one LDS: cycles per wave-instruction as seen by the CU
            printf("  op%d %7.1f", op, s / 256 / (8.0 * iters * 16));
        }
        printf("   (cycles per wave-instruction per CU; op0 ds_add_f32, op1 ds_add_u32, op2 read+add+write, op3 ds_add_rtn_f32)\n");
    }
    // ds_add_f32 against the number of active lanes (consecutive addresses)
    {
        std::vector<int> h(1024);
        for (int i = 0; i < 1024; ++i) h[i] = i % 4096;
        hipMemcpy(d_addr, h.data(), 4096, hipMemcpyHostToDevice);
        for (int n : {1, 2, 4, 8, 16, 32, 48, 64}) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc, n);
            hipDeviceSynchronize();
            long long c[256];
            hipMemcpy(c, d_cyc, 256 * 8, hipMemcpyDeviceToHost);
            double s2 = 0; for (int i = 0; i < 256; ++i) s2 += c[i];
            printf("ds_add_f32 with %2d active lanes: %7.1f cycles per wave-instruction per CU\n", n, s2 / 256 / (8.0 * iters * 16));
        }
    }
    // lane shuffles (ds_bpermute_b32): op4 = each feeds the next (latency chain), op5 = independent (throughput)
    for (int op = 4; op < 6; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            if (op == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc, 64);
            else hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, d_addr, iters, d_out, d_cyc, 64);
        }
        hipDeviceSynchronize();
        long long c[256];
        hipMemcpy(c, d_cyc, 256 * 8, hipMemcpyDeviceToHost);
        double s2 = 0; for (int i = 0; i < 256; ++i) s2 += c[i];
        printf("ds_bpermute_b32 %s: %7.1f cycles per wave-instruction per CU (8 waves), %7.1f per instruction of one wave\n", op == 4 ? "dependent chain" : "independent", s2 / 256 / (8.0 * iters * 16), s2 / 256 / (iters * 16.0));
    }
    return 0;
}
