// Micro-benchmark (developer tool): what keeps the matrix pipe of a CU at ~65 % in rpn_wino_gemm_kernel's main loop?
// The K-major loop of that kernel on its own -- a 128 x 128 workgroup tile, four waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32 tiles), operands by
// LDS-DMA (global_load_lds_dwordx4, 1 KB per wave transfer) into a ring of NBUF chunk buffers of KC k rows, fragments by ds_read_b64, one barrier per
// chunk -- with each ingredient switchable, one or two workgroups per CU, and nothing else (no tiles to finish, no stream-K, no output).
//   FLAGS  1 LDS-DMA   2 operand reads from LDS   4 barrier per chunk   8 all of a chunk's transfers in front of its steps (else one piece behind each step)
//          64 a tile ends every 256 k rows (8 chunks of 32) as in the kernel at K = 256: 64 accumulators per lane stored (8-byte stores, 128 x 128 floats per workgroup) and zeroed, the chunk's
//          closing s_waitcnt vmcnt(0) then waits for the stores as well   128 the same with TWO accumulator sets: the finished set is stored two 8-byte stores per step
//          behind the next tile's first sixteen steps' MFMAs
//          256 wave w issues its transfer behind the (w + 1)-th MFMA of the step instead of all four waves behind the fourth (64 cycles apart at the texture unit)
//          with 64: 512 the accumulators are not zeroed   1024 not stored   2048 stored as sixteen 16-byte stores (same bytes, the layout of no use) instead of 32 of 8 bytes
//          4096 SPLIT PRODUCTS: the same fp32 operands from the same LDS image, every value cut into three bf16 pieces in registers (exactly: v = h + m + l) and the
//          product taken as six v_mfma_f32_32x32x16_bf16 per accumulator tile and 16 k rows (h h, h m, m h, m m, h l, l h): 6 x 32 matrix-pipe cycles where the
//          fp32 instruction needs 8 x 64
//          32 LDS-DMA with the scalar-base address form (global_load_lds_dwordx4 voffset, s[base]: no 64-bit vector add per piece)
//   hipcc --offload-arch=gfx950 -O3 -o gemm_loop_rate tools/dev/micro/gemm_loop_rate.hip && ./gemm_loop_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const float *g, const float *lds)
{
    const unsigned l = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(g) : "memory");
}
__device__ __forceinline__ void dma16_s(const float *base, unsigned voff, const float *lds)
{
    const unsigned l = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(l)), "v"(voff), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int FLAGS, int KC, int NBUF, int WPE = 2>
__global__ __launch_bounds__(256, WPE) void loop_kernel(const float *__restrict__ A, const float *__restrict__ B, int lda, int ldb, int chunks, size_t wrapA, size_t wrapB, float *out, float *tiles_out)
{
    constexpr int MT = 128, NW = 128, MI = 2, NI = 2;
    constexpr bool DMA = FLAGS & 1, LDSR = FLAGS & 2, BAR = FLAGS & 4, FRONT = FLAGS & 8, SBASE = FLAGS & 32, TILES = FLAGS & 64, TILES2 = FLAGS & 128, STAG = FLAGS & 256, NOZERO = FLAGS & 512, NOSTORE = FLAGS & 1024, ST16 = FLAGS & 2048, SPLIT = FLAGS & 4096, NOMM = FLAGS & 8192, NOCUT = FLAGS & 16384;
    constexpr int PPW = (KC * (MT + NW) * 4 / 1024) / 4;             // 1-KB transfers per wave and chunk: 8 at KC = 32
    constexpr int STEPS = KC / 2;
    __shared__ __attribute__((aligned(16))) float sA[NBUF][KC * MT];
    __shared__ __attribute__((aligned(16))) float sB[NBUF][KC * NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    // every workgroup walks its own column block of A and B down the k rows (wrapping inside the buffers)
    const float *ua = A + (size_t)(blockIdx.x % 2) * MT, *vb = B + (size_t)(blockIdx.x % 16) * NW;
    auto issue_piece = [&](int chunk, int buf, int q) {
        const unsigned row0 = ((unsigned)chunk * KC) & 2047u;         // 2048 k rows, wrapping (no division here: a 64-bit modulo per chunk cost 15 % of the loop)
        const size_t ka = (size_t)row0 * lda, kb = (size_t)row0 * ldb;
        if (q < PPW / 2) {
            const int d = wave * (PPW / 2) + q;
            if (SBASE) dma16_s(ua + ka + (size_t)(2 * d) * lda, (unsigned)((lane / 32) * lda + (lane % 32) * 4) * 4u, &sA[buf][d * 256]);
            else dma16(ua + ka + (size_t)(2 * d + lane / 32) * lda + (unsigned)(lane % 32) * 4u, &sA[buf][d * 256]);
        } else {
            const int d = wave * (PPW / 2) + (q - PPW / 2);
            if (SBASE) dma16_s(vb + kb + (size_t)(2 * d) * ldb, (unsigned)((lane / 32) * ldb + (lane % 32) * 4) * 4u, &sB[buf][d * 256]);
            else dma16(vb + kb + (size_t)(2 * d + lane / 32) * ldb + (unsigned)(lane % 32) * 4u, &sB[buf][d * 256]);
        }
    };
    f32x16 acc[MI][NI], old[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mi][ni][r] = 0.0f; old[mi][ni][r] = 0.0f; }
    // tile t of this workgroup: [128][128] floats, rows as the kernel lays them (row stride 128 here)
    float *tbase = tiles_out + (size_t)blockIdx.x * 4 * MT * NW + (size_t)(wm * (MT / 2) + MI * 4 * lh) * NW + wn * (NW / 2) + NI * li;
    int tile_no = 0, pending = 0;                                    // pending: the old set still has stores to issue (TILES2)
    float *obase = tbase;
    for (int i = tid; i < NBUF * KC * MT; i += 256) { sA[0][i] = (float)(i % 7) * 0.25f; sB[0][i] = (float)(i % 5) * 0.5f; }
    __syncthreads();
    if (DMA) {
#pragma unroll
        for (int c = 0; c < NBUF - 1; ++c)
#pragma unroll
            for (int q = 0; q < PPW; ++q) issue_piece(c, c, q);
        vm_wait<(NBUF - 2) * PPW>();
    }
    __syncthreads();
    int buf = 0;
    for (int u = 0; u < chunks; ++u) {
        const int nb = buf == 0 ? NBUF - 1 : buf - 1;                // the buffer chunk u - 1 used = where chunk u + NBUF - 1 goes
        if (DMA && FRONT) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) issue_piece(u + NBUF - 1, nb, q);
        }
        const float *pa = &sA[buf][lh * MT + wm * (MT / 2) + MI * li];
        const float *pb = &sB[buf][lh * NW + wn * (NW / 2) + NI * li];
        if constexpr (SPLIT) {
            typedef short bf16x8 __attribute__((ext_vector_type(8)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const float *qa = &sA[buf][(8 * lh) * MT + wm * (MT / 2) + MI * li];      // lane (i, h) takes k = 16 kb + 8 h + j, j = 0 .. 7
            const float *qb = &sB[buf][(8 * lh) * NW + wn * (NW / 2) + NI * li];
            unsigned mask = 0xFFFF0000u;
            if (FLAGS & 32768) asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(mask));       // the mask as a scalar register operand instead of a 32-bit literal in every v_and
            auto cut = [mask](const float *v, u32x4 &h, u32x4 &m, u32x4 &l) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const unsigned x0 = __builtin_bit_cast(unsigned, v[2 * p]), x1 = __builtin_bit_cast(unsigned, v[2 * p + 1]);
                    h[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                    const float ra = v[2 * p] - __builtin_bit_cast(float, x0 & mask), rb = v[2 * p + 1] - __builtin_bit_cast(float, x1 & mask);
                    const unsigned r0 = __builtin_bit_cast(unsigned, ra), r1 = __builtin_bit_cast(unsigned, rb);
                    m[p] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
                    const float qa = ra - __builtin_bit_cast(float, r0 & mask), qb = rb - __builtin_bit_cast(float, r1 & mask);
                    l[p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, qb), __builtin_bit_cast(unsigned, qa), 0x07060302u);
                }
            };
#pragma unroll
            for (int kb = 0; kb < KC / 16; ++kb) {
                float va[MI][8], vb[NI][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 t = *(const float2 *)(qa + (16 * kb + j) * MT); va[0][j] = t.x; va[1][j] = t.y;
                    const float2 w = *(const float2 *)(qb + (16 * kb + j) * NW); vb[0][j] = w.x; vb[1][j] = w.y;
                }
                u32x4 ah[MI], am[MI], al[MI], bh[NI], bm[NI], bl[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (NOCUT) { ah[mi] = u32x4{__builtin_bit_cast(unsigned, va[mi][0]), __builtin_bit_cast(unsigned, va[mi][1]), __builtin_bit_cast(unsigned, va[mi][2]), __builtin_bit_cast(unsigned, va[mi][3])}; am[mi] = ah[mi]; al[mi] = u32x4{__builtin_bit_cast(unsigned, va[mi][4]), __builtin_bit_cast(unsigned, va[mi][5]), __builtin_bit_cast(unsigned, va[mi][6]), __builtin_bit_cast(unsigned, va[mi][7])}; }
                    else cut(va[mi], ah[mi], am[mi], al[mi]);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (NOCUT) { bh[ni] = u32x4{__builtin_bit_cast(unsigned, vb[ni][0]), __builtin_bit_cast(unsigned, vb[ni][1]), __builtin_bit_cast(unsigned, vb[ni][2]), __builtin_bit_cast(unsigned, vb[ni][3])}; bm[ni] = bh[ni]; bl[ni] = u32x4{__builtin_bit_cast(unsigned, vb[ni][4]), __builtin_bit_cast(unsigned, vb[ni][5]), __builtin_bit_cast(unsigned, vb[ni][6]), __builtin_bit_cast(unsigned, vb[ni][7])}; }
                    else cut(vb[ni], bh[ni], bm[ni], bl[ni]);
                }
#define MM(x, y) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x[mi]), __builtin_bit_cast(bf16x8, y[ni]), acc[mi][ni], 0, 0, 0)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        if (NOMM) { acc[mi][ni][0] += __builtin_bit_cast(float, al[mi][0] ^ bh[ni][1] ^ am[mi][2] ^ bm[ni][3] ^ ah[mi][1] ^ bl[ni][2]); acc[mi][ni][1] += __builtin_bit_cast(float, al[mi][1] ^ bh[ni][0] ^ am[mi][3] ^ bm[ni][2] ^ ah[mi][0] ^ bl[ni][3] ^ al[mi][2] ^ al[mi][3] ^ am[mi][0] ^ am[mi][1] ^ ah[mi][2] ^ ah[mi][3] ^ bh[ni][2] ^ bh[ni][3] ^ bm[ni][0] ^ bm[ni][1] ^ bl[ni][0] ^ bl[ni][1]); }
                        else { MM(al, bh); MM(ah, bl); MM(am, bm); MM(am, bh); MM(ah, bm); MM(ah, bh); }
                    }
#undef MM
                if (DMA && !FRONT) {
#pragma unroll
                    for (int q = kb * (PPW / (KC / 16)); q < (kb + 1) * (PPW / (KC / 16)); ++q) issue_piece(u + NBUF - 1, nb, q);
                }
            }
        } else {
        float oa[2][MI], ob[2][NI];
        auto fetch = [&](int s, int slot) {
            if (!LDSR) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) { oa[slot][mi] = (float)(s + lane) * 0.001f; asm volatile("" : "+v"(oa[slot][mi])); }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) { ob[slot][ni] = (float)(s - lane) * 0.002f; asm volatile("" : "+v"(ob[slot][ni])); }
                return;
            }
            { const float2 t = *(const float2 *)(pa + 2 * s * MT); oa[slot][0] = t.x; oa[slot][1] = t.y; }
            { const float2 t = *(const float2 *)(pb + 2 * s * NW); ob[slot][0] = t.x; ob[slot][1] = t.y; }
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int slot = s & 1;
            if (s + 1 < STEPS) fetch(s + 1, slot ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[slot][mi], ob[slot][ni], acc[mi][ni], 0, 0, 0);
                    if (STAG) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (DMA && s < PPW && wave == mi * NI + ni) issue_piece(u + NBUF - 1, nb, s);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            if (DMA && !FRONT && !STAG && s < PPW) issue_piece(u + NBUF - 1, nb, s);
            if (TILES2 && pending) {                                 // rows 2 s, 2 s + 1 of the 32 (mi, r) rows of the finished tile
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 2 * s + h, mi = row & 1, r = row >> 1;
                    if (row < 32) *(float2 *)(obase + (size_t)(MI * ((r & 3) + 8 * (r >> 2)) + mi) * NW) = make_float2(old[mi][0][r], old[mi][1][r]);
                }
            }
        }
        }
        if (TILES2 && pending) pending = 0;
        if ((TILES || TILES2) && (u & (256 / KC - 1)) == 256 / KC - 1) {
            float *o = tbase + (size_t)(tile_no & 3) * MT * NW;
            ++tile_no;
            if (TILES && ST16) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; r += 2)
                        *(float4 *)(o + (size_t)(MI * ((r & 3) + 8 * (r >> 2)) + mi) * NW + li * 2) = make_float4(acc[mi][0][r], acc[mi][1][r], acc[mi][0][r + 1], acc[mi][1][r + 1]);
            } else if (TILES && !NOSTORE) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) *(float2 *)(o + (size_t)(MI * ((r & 3) + 8 * (r >> 2)) + mi) * NW) = make_float2(acc[mi][0][r], acc[mi][1][r]);
            } else if (TILES) {
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) old[mi][ni] = acc[mi][ni];
                obase = o; pending = 1;
            }
            if (!NOZERO) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
            }
        }
        if (DMA) vm_wait<(NBUF - 2) * PPW>();                        // chunk u + 1 has landed (this wave's pieces; the barrier covers the others')
        if (BAR) __syncthreads();
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.0f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[mi][ni][r] + old[mi][ni][r];
    out[blockIdx.x * 256 + tid] = s;
}

static float *g_tiles;
template <int FLAGS, int KC, int NBUF, int WPE = 2>
static void run(const char *what, const float *A, const float *B, float *out, int cus, size_t wrapA, size_t wrapB)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pc = (WPE > 2 ? 3 : 1); pc <= WPE; ++pc) {
        if ((size_t)NBUF * KC * 256 * 4 * pc > 160 * 1024) continue;
        const int chunks = 64 * 32 / KC;                             // 2048 k rows: 4096 MFMAs per wave = 262144 matrix-pipe cycles = 109 us at 2.4 GHz (x 2 with two workgroups per CU)
        const int grid = cus * pc;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop_kernel<FLAGS, KC, NBUF, WPE>), dim3(grid), dim3(256), 0, 0, A, B, 256, 2432, chunks, wrapA, wrapB, out, g_tiles);
        hipDeviceSynchronize();
        float sum = 0.0f;
        const int reps = 10;
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((loop_kernel<FLAGS, KC, NBUF, WPE>), dim3(grid), dim3(256), 0, 0, A, B, 256, 2432, chunks, wrapA, wrapB, out, g_tiles);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&sum, e0, e1);
        const double us = sum * 1e3 / reps, ideal = (double)pc * chunks * (KC / 2) * 4 * 64.0 / 2.4e3;
        fflush(stdout); printf("%-58s KC %2d ring %d  workgroups per CU %d : %7.1f us  (matrix pipe alone %6.1f us: %4.1f %%)\n", what, KC, NBUF, pc, us, ideal, 100.0 * ideal / us);
    }
}

int main()
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t nA = (size_t)2048 * 256, nB = (size_t)2048 * 2432;  // U [k][256] and V [k][2432] of one plane, K = 2048 rows (L2-resident like the real operands)
    float *A, *B, *out;
    (void)hipMalloc(&A, nA * 4 + 65536); (void)hipMalloc(&B, nB * 4 + 65536); (void)hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    (void)hipMalloc(&g_tiles, (size_t)cus * 4 * 4 * 128 * 128 * 4);
    (void)hipMemset(A, 0, nA * 4 + 65536); (void)hipMemset(B, 0, nB * 4 + 65536);
    printf("launches back to back (10 per measurement); the matrix pipe alone = MFMAs x 64 cycles at 2.4 GHz\n");
    run<0, 32, 2>("MFMAs only (operands in registers)", A, B, out, cus, nA, nB);
    run<4, 32, 2>("+ barrier per chunk", A, B, out, cus, nA, nB);
    run<2, 32, 2>("+ operand reads from LDS", A, B, out, cus, nA, nB);
    run<6, 32, 2>("+ operand reads + barrier", A, B, out, cus, nA, nB);
    run<5, 32, 2>("+ LDS-DMA (pieces behind the steps) + barrier", A, B, out, cus, nA, nB);
    run<7, 32, 2>("all: DMA behind the steps + reads + barrier  (the kernel)", A, B, out, cus, nA, nB);
    run<7 + 64, 32, 2>("the kernel + a tile stored and zeroed every 8 chunks", A, B, out, cus, nA, nB);
    run<7 + 64 + 512, 32, 2>("  ... stored, not zeroed", A, B, out, cus, nA, nB);
    run<7 + 64 + 1024, 32, 2>("  ... zeroed, not stored", A, B, out, cus, nA, nB);
    run<7 + 64 + 2048, 32, 2>("  ... stored as 16-byte pieces (half the instructions)", A, B, out, cus, nA, nB);
    run<7 + 128, 32, 2>("the kernel + tiles stored behind the next tile's steps", A, B, out, cus, nA, nB);
    run<7 + 256, 32, 2>("the kernel, waves' transfers staggered inside the step", A, B, out, cus, nA, nB);
    run<7 + 256 + 64, 32, 2>("staggered + a tile stored and zeroed every 8 chunks", A, B, out, cus, nA, nB);
    run<7 + 4096, 32, 2>("SPLIT PRODUCTS (6 bf16 MFMAs per tile and 16 k): the kernel's loop", A, B, out, cus, nA, nB);
    run<6 + 4096, 32, 2>("SPLIT PRODUCTS without the LDS-DMA", A, B, out, cus, nA, nB);
    run<6 + 4096 + 8192, 32, 2>("  ... the cutting alone (no MFMAs)", A, B, out, cus, nA, nB);
    run<6 + 4096 + 16384, 32, 2>("  ... the MFMAs alone (no cutting)", A, B, out, cus, nA, nB);
    run<6 + 4096 + 8192 + 32768, 32, 2>("  ... the cutting alone, mask in a scalar register", A, B, out, cus, nA, nB);
    run<7 + 4096 + 32768, 32, 2>("SPLIT PRODUCTS, mask in a scalar register", A, B, out, cus, nA, nB);
    run<7 + 4096 + 64, 32, 2>("SPLIT PRODUCTS + a tile stored and zeroed every 8 chunks", A, B, out, cus, nA, nB);
    run<15, 32, 2>("all, DMA in front of the steps", A, B, out, cus, nA, nB);
    run<7 + 32, 32, 2>("all, LDS-DMA addressed scalar base + lane offset", A, B, out, cus, nA, nB);
    run<7, 16, 2>("all, behind the steps", A, B, out, cus, nA, nB);
    run<7, 16, 2, 4>("all, behind the steps (<= 128 registers)", A, B, out, cus, nA, nB);
    run<7 + 64, 16, 2, 4>("... + a tile stored and zeroed every 8 chunks", A, B, out, cus, nA, nB);
    run<7, 16, 3>("all, behind the steps, two chunks ahead", A, B, out, cus, nA, nB);
    run<7, 16, 4>("all, behind the steps, three chunks ahead", A, B, out, cus, nA, nB);
    run<15, 16, 3>("all, in front, two chunks ahead", A, B, out, cus, nA, nB);
    run<15, 16, 4>("all, in front, three chunks ahead", A, B, out, cus, nA, nB);
    run<7, 32, 3>("all, behind the steps, two chunks ahead", A, B, out, cus, nA, nB);
    run<7, 8, 4>("all, behind the steps, three chunks ahead", A, B, out, cus, nA, nB);
    run<7, 8, 6>("all, behind the steps, five chunks ahead", A, B, out, cus, nA, nB);
    return 0;
}
