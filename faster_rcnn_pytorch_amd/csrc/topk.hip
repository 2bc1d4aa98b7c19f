// topk.hip -- scores.sort(descending=True)[:K] (models/model.py:44-49) as a chip-wide RANK sort.
//
// Why a rank sort: bs = 1 per GPU means ONE sort of N = 20 646 scores per step.  A radix sort of
// that size is a chain of small dependent launches on a handful of CUs; counting, for every
// element, how many elements beat it is embarrassingly parallel, fills all 256 CUs, needs no
// tie-handling special case (the order (score desc, index asc) is total) and is deterministic.
//   rank(i) = #{ j : key_j > key_i  or (key_j == key_i and j < i) }
// Kernel 1 (topk_rank_kernel): grid (row blocks of 256) x (column segments of SEG); each block
//   stages its segment's keys in LDS (coalesced load, order-preserving float->uint transform) and
//   every lane compares its own key with each staged key (ds_read_b128 broadcast, 2 VALU / pair:
//   v_cmp + v_addc).  Whole 64-column chunks left / right of the wave's own rows use >= / > so the
//   index tie-break costs nothing; only the diagonal chunk evaluates it per lane.
// Kernel 2 (topk_scatter_kernel): sums the per-segment partial ranks and scatters index, score and
//   (optionally) the box to position rank(i) if rank(i) < K: the gather at model.py:48 is fused.
// Work: N^2 pair compares -- used for N < 4096 only; larger N take the sample sort further down.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "topk_dev.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(topk);
#include <cstdlib>

#define TOPK_ROWS 256
#define TOPK_SEG 1024

__global__ __launch_bounds__(TOPK_ROWS) void topk_rank_kernel(const float *__restrict__ scores, int N, int32_t *__restrict__ partial,
                                                              int32_t *__restrict__ count_zero)
{
    __shared__ uint4 seg4[TOPK_SEG / 4];
    uint32_t *seg = (uint32_t *)seg4;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * TOPK_ROWS;
    const int c0 = blockIdx.y * TOPK_SEG;
    if (count_zero && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *count_zero = 0;
#pragma unroll
    for (int t = tid; t < TOPK_SEG; t += TOPK_ROWS) {
        const int j = c0 + t;
        seg[t] = j < N ? f2key(scores[j]) : 0u;       // tail keys are never "greater"
    }
    const int i = r0 + tid;
    const uint32_t ki = i < N ? f2key(scores[i]) : 0xFFFFFFFFu;
    __syncthreads();
    const int wb = r0 + (tid & ~63);                  // first row of this wave
    int rank = 0;
    for (int cc = 0; cc < TOPK_SEG / 64; ++cc) {
        const int cb = c0 + cc * 64;
        if (cb >= N) break;
        const uint4 *p = seg4 + cc * 16;
        if (cb + 64 <= wb) {                          // every column index < every row index: ties count
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint4 k = p[q];
                rank += (k.x >= ki) + (k.y >= ki) + (k.z >= ki) + (k.w >= ki);
            }
        } else if (cb >= wb + 64) {                   // every column index > every row index
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint4 k = p[q];
                rank += (k.x > ki) + (k.y > ki) + (k.z > ki) + (k.w > ki);
            }
        } else {                                      // diagonal chunk: per-lane tie-break
            const int li = i - cb;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint4 k = p[q];
                rank += (k.x > ki) || (k.x == ki && 4 * q + 0 < li);
                rank += (k.y > ki) || (k.y == ki && 4 * q + 1 < li);
                rank += (k.z > ki) || (k.z == ki && 4 * q + 2 < li);
                rank += (k.w > ki) || (k.w == ki && 4 * q + 3 < li);
            }
        }
    }
    if (i < N) partial[(size_t)blockIdx.y * N + i] = rank;
}

__global__ __launch_bounds__(256) void topk_scatter_kernel(const float *__restrict__ scores, const float4 *__restrict__ boxes_in,
                                                           const int32_t *__restrict__ partial, int N, int nseg, int K,
                                                           int proposal_mode, int64_t *__restrict__ out_idx,
                                                           float *__restrict__ out_scores, float4 *__restrict__ out_boxes,
                                                           int32_t *__restrict__ out_count)
{
    __shared__ int s_max[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int mine = 0;
    if (i < N) {
        int rank = 0;
        for (int s = 0; s < nseg; ++s) rank += partial[(size_t)s * N + i];
        const float sc = scores[i];
        const bool valid = proposal_mode ? (sc >= 0.0f) : true;
        if (valid && rank < K) {
            out_idx[rank] = i;
            out_scores[rank] = sc;
            if (out_boxes) out_boxes[rank] = boxes_in[i];
            mine = rank + 1;
        }
    }
    // count = max over selected of (rank + 1): wave max, then block max, then one atomic per block
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = max(mine, __shfl_xor(mine, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (m > 0) atomicMax(out_count, m);
    }
}

// ------------------------------------------------------------------------------------------------
// N >= 4096: SAMPLE SORT on the composite key  key64 = f2key(score) << 32 | ~index  (all keys distinct, order = score descending,
// index ascending, so ties need no special case and cannot unbalance anything):
//   splitters            S = 512 / 2048 evenly spaced samples rank-sorted among themselves, every stride-th kept (255): ss_sample_body in
//                        topk_dev.h -- inside the proposal prologue's launch, or topk_sample_kernel for the generic entry points;
//   topk_partition       every key finds its bucket (binary search over the splitters in LDS) -> bucket sizes, #valid; GRID BARRIER;
//                        exclusive scan of the sizes + one atomic slot per key: buckets become contiguous segments of a scratch array
//                        (order inside a segment is arbitrary); <true>: a second barrier, then the ranking below in the same launch
//                        (topk_count_kernel / topk_place_kernel: the same as two launches, for grids that cannot be co-resident);
//   topk_bucket_kernel   one workgroup per bucket (and part) whose first rank is < K: rank inside the bucket (~80^2 compares at
//                        N = 20 646; the keys are staged in LDS) + the bucket's base = the exact rank; index, score and box are
//                        scattered to that position (the gather at model.py:48 stays fused).
// Work N * (8 + N / 256) compares instead of N^2: the chip-wide rank sort above took 34 us + 11 us (scatter) at N = 20 646 and
// the radix-select pre-filter of round 1 57 us + 6 us at N = 268 569 (four histogram / compaction launches before its rank sort).
// ------------------------------------------------------------------------------------------------
#define SS_PER_THREAD 4                                            // keys per thread of the placing kernel

typedef unsigned long long u64;

// the splitter sampling as a launch of its own (the generic top-k / argsort entry points; the proposal stage runs ss_sample_body
// inside its prologue launch, see topk_dev.h and boxes.hip)
template <int S>
__global__ __launch_bounds__(256) void topk_sample_kernel(const float *__restrict__ scores, int N, int stride, SsCtl *__restrict__ ctl)
{
    __shared__ uint4 s_k4[S / 4];
    __shared__ int s_part[4][64];
    ss_sample_body<S>([&](int i) { return scores[i]; }, N, stride, ctl, (int)blockIdx.x, s_k4, s_part);
}

// number of splitters 1..255 that are > k  =  the bucket of k
__device__ __forceinline__ int ss_bucket(const u64 *s_split, u64 k)
{
    int lo = 0;
#pragma unroll
    for (int step = SS_BUCKETS / 2; step > 0; step >>= 1)
        if (s_split[lo + step] > k) lo += step;
    return lo;
}

__global__ __launch_bounds__(256) void topk_count_kernel(const float *__restrict__ scores, int N, int proposal_mode, SsCtl *__restrict__ ctl)
{
    __shared__ u64 s_split[SS_BUCKETS];
    __shared__ int s_cnt[SS_BUCKETS];
    __shared__ int s_valid;
    s_split[threadIdx.x] = ctl->split[threadIdx.x];
    s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    int valid = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const float sc = scores[i];
        valid += (!proposal_mode || sc >= 0.0f) ? 1 : 0;
        atomicAdd(&s_cnt[ss_bucket(s_split, ss_key(sc, i))], 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    if ((threadIdx.x & 63) == 0 && valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    if (s_cnt[threadIdx.x]) atomicAdd(&ctl->cnt[threadIdx.x], s_cnt[threadIdx.x]);
    if (threadIdx.x == 0 && s_valid) atomicAdd(&ctl->n_valid, s_valid);
}

// exclusive scan of 256 bucket sizes by 256 threads; returns this thread's base
__device__ __forceinline__ int ss_scan256(int v, int *s_w /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) base += q < wave ? s_w[q] : 0;
    return base + inc - v;
}

// Buckets become contiguous segments of `sorted`.  A block first counts its own 1024 keys per bucket in LDS (the returned value
// is the key's slot inside the block's share), reserves the block's share of every non-empty bucket with ONE global atomic per
// bucket, then writes: N / 1024 * (<= 256) global atomics instead of N on 256 hot words.
__global__ __launch_bounds__(256) void topk_place_kernel(const float *__restrict__ scores, int N, SsCtl *__restrict__ ctl, u64 *__restrict__ sorted)
{
    __shared__ u64 s_split[SS_BUCKETS];
    __shared__ int s_base[SS_BUCKETS], s_cnt[SS_BUCKETS];
    __shared__ int s_w[4];
    s_split[threadIdx.x] = ctl->split[threadIdx.x];
    s_cnt[threadIdx.x] = 0;
    s_base[threadIdx.x] = ss_scan256(ctl->cnt[threadIdx.x], s_w);
    __syncthreads();
    u64 k[SS_PER_THREAD];
    int bk[SS_PER_THREAD], slot[SS_PER_THREAD];
#pragma unroll
    for (int e = 0; e < SS_PER_THREAD; ++e) {
        const int i = (blockIdx.x * SS_PER_THREAD + e) * 256 + threadIdx.x;
        bk[e] = -1;
        if (i < N) {
            k[e] = ss_key(scores[i], i);
            bk[e] = ss_bucket(s_split, k[e]);
            slot[e] = atomicAdd(&s_cnt[bk[e]], 1);
        }
    }
    __syncthreads();
    const int c = s_cnt[threadIdx.x];
    if (c) s_base[threadIdx.x] += atomicAdd(&ctl->cursor[threadIdx.x], c);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < SS_PER_THREAD; ++e)
        if (bk[e] >= 0) {
            const int pos = s_base[bk[e]] + slot[e];
            if (ss_in_range(pos, 1, N)) sorted[pos] = k[e];
            else ss_mark_bad(ctl);
        }
}

// topk_count_kernel + topk_place_kernel as ONE launch for grids that are certainly co-resident (<= SS_PART_MAX_WG workgroups of 256
// threads and ~4 KB of LDS): every key's bucket is searched once, the block's histogram goes to the global counts, and a grid barrier
// (two-level arrival counters and a polled flag in the control block, zeroed by the sampling workgroup of the launch before; bounded) separates that
// from the scan + placement.  Between the phases a thread keeps its four keys, buckets and slots in registers.
#define SS_PART_MAX_WG 1024
#define SS_BARRIER_SPINS (1 << 22)
// grid barrier of the partition launch: two-level arrival counters, the waiters poll the flag of generation `gen` (1, 2)
__device__ __forceinline__ void ss_grid_barrier(SsCtl *__restrict__ ctl, int gen)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // what I wrote through / added has been performed before my workgroup arrives
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nb = (int)gridDim.x, x = (int)blockIdx.x & 7;
        int (*bar)[16] = ctl->bar + 9 * (gen - 1);
        bool last;
        if (nb <= 32) last = __hip_atomic_fetch_add(&bar[8][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
        else last = __hip_atomic_fetch_add(&bar[x][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (nb - x + 7) / 8 - 1 &&
                    __hip_atomic_fetch_add(&bar[8][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 8 - 1;
        if (last) __hip_atomic_store(&ctl->flag[0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&ctl->flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen && ++spins < SS_BARRIER_SPINS)
            __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
}

// BUCKETS = true: the ranking of the buckets (topk_bucket_kernel's job) follows behind a second barrier in the same launch: bucket
// b, b + grid, ... by workgroup b while the bucket's first rank is below the output count.  The placed keys are written through
// and read back with agent-scope loads.
template <bool BUCKETS>
__global__ __launch_bounds__(256) void topk_partition_kernel(const float *__restrict__ scores, const float4 *__restrict__ boxes_in, int N, int K,
                                                             int proposal_mode, SsCtl *__restrict__ ctl, u64 *__restrict__ sorted,
                                                             int64_t *__restrict__ out_idx, float *__restrict__ out_scores,
                                                             float4 *__restrict__ out_boxes, int32_t *__restrict__ out_count)
{
    __shared__ u64 s_split[SS_BUCKETS];
    __shared__ int s_base[SS_BUCKETS], s_cnt[SS_BUCKETS];
    __shared__ int s_w[4];
    __shared__ int s_valid;
    __shared__ u64 s_k[BUCKETS ? 1024 : 1];
    s_split[threadIdx.x] = ctl->split[threadIdx.x];
    s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    u64 k[SS_PER_THREAD];
    int bk[SS_PER_THREAD], slot[SS_PER_THREAD];
    int valid = 0;
#pragma unroll
    for (int e = 0; e < SS_PER_THREAD; ++e) {
        const int i = (blockIdx.x * SS_PER_THREAD + e) * 256 + threadIdx.x;
        bk[e] = -1;
        if (i < N) {
            const float sc = scores[i];
            valid += (!proposal_mode || sc >= 0.0f) ? 1 : 0;
            k[e] = ss_key(sc, i);
            bk[e] = ss_bucket(s_split, k[e]);
            slot[e] = atomicAdd(&s_cnt[bk[e]], 1);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    if ((threadIdx.x & 63) == 0 && valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    const int c = s_cnt[threadIdx.x];
    if (c) atomicAdd(&ctl->cnt[threadIdx.x], c);
    if (threadIdx.x == 0 && s_valid) atomicAdd(&ctl->n_valid, s_valid);
    ss_grid_barrier(ctl, 1);
    const int total = __hip_atomic_load(&ctl->cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int excl = ss_scan256(total, s_w);
    s_base[threadIdx.x] = excl;
    __syncthreads();
    if (c) s_base[threadIdx.x] += atomicAdd(&ctl->cursor[threadIdx.x], c);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < SS_PER_THREAD; ++e)
        if (bk[e] >= 0) {
            const int pos = s_base[bk[e]] + slot[e];
            if (!ss_in_range(pos, 1, N)) ss_mark_bad(ctl);         // counts / cursors not reset by this build's sampling workgroup: no wild store
            else if (BUCKETS) __hip_atomic_store(&sorted[pos], k[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else sorted[pos] = k[e];
        }
    if constexpr (BUCKETS) {
        ss_grid_barrier(ctl, 2);
        s_base[threadIdx.x] = excl;                                 // the buckets' first ranks
        s_cnt[threadIdx.x] = total;
        const int n_valid = __hip_atomic_load(&ctl->n_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool bad = __hip_atomic_load(&ctl->pad[SS_PAD_BAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || n_valid < 0 || n_valid > N;
        const int n_out = bad ? -1 : (n_valid < K ? n_valid : K);
        if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = n_out;
        __syncthreads();
        // work item = (bucket, part): the bucket's rows are split over `parts` workgroups, every one of which stages the whole bucket in
        // LDS (one agent-scope load per key: the rows come out of LDS too)
        const int parts = N < 65536 ? 2 : 8;
        for (int it = (int)blockIdx.x; it < SS_BUCKETS * parts; it += (int)gridDim.x) {
            const int bkt = it / parts, part = it - bkt * parts;
            const int base = s_base[bkt], m = s_cnt[bkt];
            if (base >= n_out) break;                               // the buckets are in rank order (n_out = -1: nothing is ranked)
            if (m == 0) continue;
            if (!ss_in_range(base, m, N)) break;
            const u64 *seg = sorted + base;
            const int per = (m + parts - 1) / parts;
            const int e0 = part * per, e1 = min(m, e0 + per);
            auto emit = [&](u64 kk, int r) {
                const int rank = base + r;
                if (rank < n_out) {
                    const int idx = (int)(~(uint32_t)kk);
                    if ((unsigned)idx >= (unsigned)N) return;
                    out_idx[rank] = idx;
                    out_scores[rank] = scores[idx];
                    if (out_boxes) out_boxes[rank] = boxes_in[idx];
                }
            };
            if (m <= 1024) {
                __syncthreads();
                for (int t = threadIdx.x; t < m; t += 256) s_k[t] = __hip_atomic_load(&seg[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                for (int e = e0 + threadIdx.x; e < e1; e += 256) {
                    const u64 kk = s_k[e];
                    int r = 0;
#pragma unroll 8
                    for (int j = 0; j < m; ++j) r += s_k[j] > kk;   // broadcast ds_read_b64
                    emit(kk, r);
                }
                continue;
            }
            for (int r0 = e0; r0 < e1; r0 += 256) {                 // an unlucky sample: row groups x 1024-key chunks
                const int e = r0 + threadIdx.x;
                const u64 kk = e < e1 ? __hip_atomic_load(&seg[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                int r = 0;
                for (int c0 = 0; c0 < m; c0 += 1024) {
                    const int cn = min(1024, m - c0);
                    __syncthreads();
                    for (int t = threadIdx.x; t < cn; t += 256) s_k[t] = __hip_atomic_load(&seg[c0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
#pragma unroll 8
                    for (int j = 0; j < cn; ++j) r += s_k[j] > kk;
                }
                if (e < e1) emit(kk, r);
            }
        }
    }
}

__global__ __launch_bounds__(256) void topk_bucket_kernel(const float *__restrict__ scores, const float4 *__restrict__ boxes_in, int N, int K,
                                                          const SsCtl *__restrict__ ctl, const u64 *__restrict__ sorted,
                                                          int64_t *__restrict__ out_idx, float *__restrict__ out_scores,
                                                          float4 *__restrict__ out_boxes, int32_t *__restrict__ out_count)
{
    __shared__ u64 s_k[1024];
    __shared__ int s_w[4];
    __shared__ int s_b[2];
    const int bkt = blockIdx.x;
    const int part = blockIdx.y, nparts = gridDim.y;                // the bucket's keys (rows) are split over nparts workgroups
    const int mine = ctl->cnt[threadIdx.x];
    const int excl = ss_scan256(mine, s_w);
    if (threadIdx.x == bkt) { s_b[0] = excl; s_b[1] = mine; }
    __syncthreads();
    const int base = s_b[0], m = s_b[1];
    const int n_valid = ctl->n_valid;
    const bool bad = ctl->pad[SS_PAD_BAD] != 0 || n_valid < 0 || n_valid > N;
    const int n_out = bad ? -1 : (n_valid < K ? n_valid : K);
    if (bkt == 0 && part == 0 && threadIdx.x == 0) *out_count = n_out;
    if (base >= n_out || m == 0 || !ss_in_range(base, m, N)) return;
    const u64 *seg = sorted + base;
    const int per = (m + nparts - 1) / nparts;
    const int e0 = part * per, e1 = min(m, e0 + per);              // my rows
    if (e0 >= e1) return;
    auto emit = [&](u64 k, int r) {
        const int rank = base + r;
        if (rank < n_out) {
            const int idx = (int)(~(uint32_t)k);
            if ((unsigned)idx >= (unsigned)N) return;
            out_idx[rank] = idx;
            out_scores[rank] = scores[idx];
            if (out_boxes) out_boxes[rank] = boxes_in[idx];
        }
    };
    if (m <= 1024) {                                               // the usual case: the whole bucket in LDS, one pass
        for (int t = threadIdx.x; t < m; t += 256) s_k[t] = seg[t];
        __syncthreads();
        for (int e = e0 + threadIdx.x; e < e1; e += 256) {         // (waves without a row skip the loop)
            const u64 k = s_k[e];
            int r = 0;
#pragma unroll 8
            for (int j = 0; j < m; ++j) r += s_k[j] > k;           // broadcast ds_read_b64
            emit(k, r);
        }
        return;
    }
    for (int r0 = e0; r0 < e1; r0 += 256) {                        // an unlucky sample: row groups x 1024-key chunks
        const int e = r0 + threadIdx.x;
        const u64 k = e < e1 ? seg[e] : 0ull;
        int r = 0;
        for (int c0 = 0; c0 < m; c0 += 1024) {
            const int cn = min(1024, m - c0);
            __syncthreads();
            for (int t = threadIdx.x; t < cn; t += 256) s_k[t] = seg[c0 + t];
            __syncthreads();
#pragma unroll 8
            for (int j = 0; j < cn; ++j) r += s_k[j] > k;
        }
        if (e < e1) emit(k, r);
    }
}

size_t frcnn_ws_topk(int64_t N)
{
    const int64_t nseg = (N + TOPK_SEG - 1) / TOPK_SEG;
    if (N >= SS_MIN_N) return align_up(sizeof(SsCtl), 256) + align_up((size_t)N * 8, 256);
    return align_up((size_t)(nseg > 0 ? nseg : 1) * (size_t)N * sizeof(int32_t), 256);
}

void *frcnn_topk_sample_ctl(void *ws, int64_t N) { return N >= SS_MIN_N ? ws : nullptr; }

int frcnn_launch_topk(const float *scores, const float *boxes_in, int64_t N, int64_t K, int proposal_mode,
                      int64_t *out_idx, float *out_scores, float *out_boxes, int32_t *out_count,
                      void *ws, size_t ws_bytes, bool sampled, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_topk(N))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "topk: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_topk(N));
    if (N >= SS_MIN_N) {
        SsCtl *ctl = (SsCtl *)ws;
        u64 *sorted = (u64 *)((char *)ws + align_up(sizeof(SsCtl), 256));
        const int gb = (int)((N + 256 * SS_PER_THREAD - 1) / (256 * SS_PER_THREAD));
        if (!sampled) {                                                         // (the proposal prologue has done it otherwise)
            int S, stride;
            ss_plan(N, K, &S, &stride);
            if (S == 512) FRCNN_LAUNCH(topk_sample_kernel<512>, dim3(512 / 64), dim3(256), 0, s, scores, (int)N, stride, ctl);
            else FRCNN_LAUNCH(topk_sample_kernel<SS_LARGE>, dim3(SS_LARGE / 64), dim3(256), 0, s, scores, (int)N, stride, ctl);
            FRCNN_CHECK_LAUNCH("topk_sample_kernel");
        }
        // FRCNN_TOPK_FUSED: 2 = count + place + bucket ranking as ONE launch (two grid barriers), 1 = count + place fused and the ranking
        // as its own launch, 0 = three launches.  The fused forms need every workgroup resident: <= SS_PART_MAX_WG workgroups.
        // Default by size: the one-launch form wins at FPN size (31.6 against 17.0 + 17.5 us, HIP events in the training step) and loses
        // at 600 x 1000 (22.2 against 8.8 + 7.7 us: ten dependent round trips and two barriers for ~20 000 keys).
        static const int fuse_env = [] { const char *e = getenv("FRCNN_TOPK_FUSED"); return e ? atoi(e) : -1; }();
        const int fuse = fuse_env >= 0 ? fuse_env : (N >= 65536 ? 2 : 1);
        if (gb <= SS_PART_MAX_WG && fuse >= 2) {
            const int grid = gb < 256 ? 256 : gb;                               // enough workgroups for the ranking phase (one or two work items each)
            FRCNN_LAUNCH((topk_partition_kernel<true>), dim3(grid), dim3(256), 0, s, scores, (const float4 *)boxes_in, (int)N, (int)K, proposal_mode, ctl, sorted,
                         out_idx, out_scores, (float4 *)out_boxes, out_count);
            FRCNN_CHECK_LAUNCH("topk_partition_kernel");
            return FRCNN_OK;
        }
        if (gb <= SS_PART_MAX_WG && fuse == 1) {
            FRCNN_LAUNCH((topk_partition_kernel<false>), dim3(gb), dim3(256), 0, s, scores, (const float4 *)boxes_in, (int)N, (int)K, proposal_mode, ctl, sorted,
                         out_idx, out_scores, (float4 *)out_boxes, out_count);
            FRCNN_CHECK_LAUNCH("topk_partition_kernel");
        } else {
            FRCNN_LAUNCH(topk_count_kernel, dim3(gb < 1024 ? gb : 1024), dim3(256), 0, s, scores, (int)N, proposal_mode, ctl);
            FRCNN_CHECK_LAUNCH("topk_count_kernel");
            FRCNN_LAUNCH(topk_place_kernel, dim3(gb), dim3(256), 0, s, scores, (int)N, ctl, sorted);
            FRCNN_CHECK_LAUNCH("topk_place_kernel");
        }
        FRCNN_LAUNCH(topk_bucket_kernel, dim3(SS_BUCKETS, N < 65536 ? 4 : 8), dim3(256), 0, s, scores, (const float4 *)boxes_in, (int)N, (int)K, ctl,
                     sorted, out_idx, out_scores, (float4 *)out_boxes, out_count);
        FRCNN_CHECK_LAUNCH("topk_bucket_kernel");
        return FRCNN_OK;
    }
    const int nseg = (int)((N + TOPK_SEG - 1) / TOPK_SEG);
    const int nrow = (int)((N + TOPK_ROWS - 1) / TOPK_ROWS);
    int32_t *partial = (int32_t *)ws;
    FRCNN_LAUNCH(topk_rank_kernel, dim3(nrow, nseg), dim3(TOPK_ROWS), 0, s, scores, (int)N, partial, out_count);
    FRCNN_CHECK_LAUNCH("topk_rank_kernel");
    FRCNN_LAUNCH(topk_scatter_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, scores,
                 (const float4 *)boxes_in, partial, (int)N, nseg, (int)K, proposal_mode, out_idx, out_scores, (float4 *)out_boxes,
                 out_count);
    FRCNN_CHECK_LAUNCH("topk_scatter_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_topk_sorted(const float *scores, const float *boxes_in, int64_t N, int64_t K, int64_t *out_idx,
                                   float *out_scores, float *out_boxes, int32_t *out_count, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(N >= 0 && K >= 0, "topk: negative size");
    FRCNN_REQUIRE(out_count, "topk: NULL out_count");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0 || K == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "topk: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(scores && out_idx && out_scores && workspace, "topk: NULL pointer");
    FRCNN_REQUIRE((boxes_in == nullptr) == (out_boxes == nullptr), "topk: boxes_in and out_boxes must both be given or both NULL");
    FRCNN_REQUIRE(N <= (1 << 22), "topk: N=%lld above the rank-sort limit 4194304", (long long)N);
    return frcnn_launch_topk(scores, boxes_in, N, K, 1, out_idx, out_scores, out_boxes, out_count, workspace, workspace_bytes, false, s);
}

// generic variant used by the nms() op: every score is live (negative scores included)
FRCNN_EXPORT int frcnn_argsort_desc(const float *scores, const float *boxes_in, int64_t N, int64_t *out_idx, float *out_scores,
                                    float *out_boxes, int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(N >= 0 && out_count, "argsort: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "argsort: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(scores && out_idx && out_scores && workspace, "argsort: NULL pointer");
    FRCNN_REQUIRE((boxes_in == nullptr) == (out_boxes == nullptr), "argsort: boxes_in and out_boxes must both be given or both NULL");
    FRCNN_REQUIRE(N <= (1 << 22), "argsort: N too large");
    return frcnn_launch_topk(scores, boxes_in, N, N, 0, out_idx, out_scores, out_boxes, out_count, workspace, workspace_bytes, false, s);
}
