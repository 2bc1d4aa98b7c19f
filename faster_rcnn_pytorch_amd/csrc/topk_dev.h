// topk_dev.h -- device pieces of the sample sort (topk.hip) that the proposal prologue (boxes.hip) runs in its own launch: the
// splitter sampling needs S scores only, so S / 64 extra workgroups of the prologue kernel compute them with the prologue's own
// per-anchor function and rank them while the other workgroups decode the anchors -- one launch fewer on the proposal path.
#pragma once
#include "frcnn_common.h"

#define SS_BUCKETS 256
#define SS_MIN_N 4096
#ifndef SS_LARGE
#define SS_LARGE 2048                   // samples for N >= 65536 (1024: the fused prologue launch 23 -> 16 us, but twice the bucket size: topk_bucket 17 -> 35 us)
#endif

#define SS_PAD_BAD 1                    // SsCtl::pad[1]: set by a placement / ranking kernel that found the control block inconsistent (see ss_in_range)

typedef unsigned long long ss_u64;
// bar: the partition launch's grid barrier -- eight per-residue arrival counters and a top counter on their own 64-byte lines, and the
// flag the waiters poll (hundreds of workgroups bumping AND polling one word queue behind each other: 13 us in the RPN target maker)
struct SsCtl { ss_u64 split[SS_BUCKETS]; int cnt[SS_BUCKETS]; int cursor[SS_BUCKETS]; int n_valid; int pad[15]; int bar[18][16]; int flag[16]; };

// Memory-safety fence (round 4): every address the top-k kernels derive from the control block's counts and cursors
// is range-checked against N before it is used.  A control block that was not reset by the sampling workgroup of the SAME build (a stale
// object, a caller's uninitialised workspace) then yields a count of -1 downstream (NMS sees no live box, head_targets raises
// FRCNN_HT_ERR_SHORT, the loss is NaN, check_device_status() reports) -- not a wild store.
__device__ __forceinline__ bool ss_in_range(int base, int len, int N) { return base >= 0 && len >= 0 && (unsigned)base + (unsigned)len <= (unsigned)N; }
__device__ __forceinline__ void ss_mark_bad(SsCtl *ctl) { __hip_atomic_store(&ctl->pad[SS_PAD_BAD], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// order-preserving map float -> uint32 (total order; -0 < +0)
__device__ __forceinline__ uint32_t f2key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ ss_u64 ss_key(float sc, int idx) { return ((ss_u64)f2key(sc) << 32) | (ss_u64)(uint32_t)(~idx); }

// samples per sort and splitter stride: 512 samples for the sizes of one feature map, SS_LARGE above; the stride covers ranks up to
// ~1.5 K (at least), the whole distribution at most
static inline constexpr void ss_plan(int64_t N, int64_t K, int *S_out, int *stride_out)
{
    const int S = N < 65536 ? 512 : SS_LARGE;
    const int full = S / SS_BUCKETS;                                        // stride that spreads 255 splitters over all S samples
    int stride = (int)((3 * K * S + 2 * (SS_BUCKETS - 1) * N - 1) / (2 * (SS_BUCKETS - 1) * N));     // ceil(1.5 K S / (255 N))
    *S_out = S;
    *stride_out = stride < 1 ? 1 : (stride > full ? full : stride);
}

// S evenly spaced samples, rank-sorted by S threads (32-bit keys; the sample position breaks ties exactly like the index would:
// positions grow with the index).  Splitter q (1..255) = the sample of rank q * stride: stride 4 (S = 1024) / 2 (S = 512) covers the
// whole distribution; a smaller stride concentrates the 255 splitters on the best-scored part when K << N (buckets past rank K are
// never ranked), keeping the buckets that matter at ~N / S * stride keys.
// S / 64 workgroups of 256 threads; workgroup rb ranks its 64 samples against ALL S sample keys (staged in LDS), its four waves
// taking every fourth 64-key chunk.  (All samples in one workgroup: 9 us at S = 512, 21 us at S = 1024 -- one CU's VALU; one lone
// wave per 64 samples: 9 / 15 us -- a lone wave issues one instruction every 4-8 cycles.)
// score_at(i) = the score of element i (a load in topk_sample_kernel, the whole per-anchor prologue in the fused launch).
template <int S, class F>
__device__ __forceinline__ void ss_sample_body(F score_at, int N, int stride, SsCtl *__restrict__ ctl, int rb, uint4 *s_k4, int (*s_part)[64])
{
    uint32_t *s_k = (uint32_t *)s_k4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = rb * 64 + lane;
    for (int tq = threadIdx.x; tq < S; tq += 256) s_k[tq] = f2key(score_at((int)(((long long)tq * N) / S)));
    if (rb == 0) { ctl->cnt[threadIdx.x] = 0; ctl->cursor[threadIdx.x] = 0; if (threadIdx.x == 0) { ctl->n_valid = 0; ctl->pad[SS_PAD_BAD] = 0; ctl->flag[0] = 0; ctl->split[0] = ~0ull; } if (threadIdx.x < 18) ctl->bar[threadIdx.x][0] = 0; }
    __syncthreads();
    const int idx = (int)(((long long)t * N) / S);
    const uint32_t k = s_k[t];
    // rank among the samples, (key desc, position asc): whole 64-sample chunks before / after my own chunk need no tie-break
    // (>= / >), only the own chunk evaluates it per lane (same scheme as topk_rank_kernel)
    int rank = 0;
    for (int c = wave; c < S / 64; c += 4) {
        const uint4 *p = s_k4 + c * 16;
        if (c < rb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { const uint4 v = p[q]; rank += (v.x >= k) + (v.y >= k) + (v.z >= k) + (v.w >= k); }
        } else if (c > rb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { const uint4 v = p[q]; rank += (v.x > k) + (v.y > k) + (v.z > k) + (v.w > k); }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint4 v = p[q];
                rank += (v.x > k) || (v.x == k && 4 * q + 0 < lane);
                rank += (v.y > k) || (v.y == k && 4 * q + 1 < lane);
                rank += (v.z > k) || (v.z == k && 4 * q + 2 < lane);
                rank += (v.w > k) || (v.w == k && 4 * q + 3 < lane);
            }
        }
    }
    s_part[wave][lane] = rank;
    __syncthreads();
    if (wave != 0) return;
    rank = s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane];
    // bucket b holds the keys x with split[b] > x >= split[b + 1] (split[0] = +inf, split[256] = -inf): descending ranges.
    // (the 32-bit key back to the composite key: the score's bits are recovered from the order-preserving map)
    if (rank > 0 && rank % stride == 0 && rank / stride < SS_BUCKETS) ctl->split[rank / stride] = ((ss_u64)k << 32) | (ss_u64)(uint32_t)(~idx);
}
