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
// Work: N^2 pair compares (426 M at N = 20 646 -> ~11 us of VALU on 1024 SIMDs).
#include "frcnn_common.h"
#include "frcnn_internal.h"

#define TOPK_ROWS 256
#define TOPK_SEG 1024

// order-preserving map float -> uint32 (total order; -0 < +0)
__device__ __forceinline__ uint32_t f2key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}

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

size_t frcnn_ws_topk(int64_t N)
{
    const int64_t nseg = (N + TOPK_SEG - 1) / TOPK_SEG;
    return align_up((size_t)(nseg > 0 ? nseg : 1) * (size_t)N * sizeof(int32_t), 256);
}

int frcnn_launch_topk(const float *scores, const float *boxes_in, int64_t N, int64_t K, int proposal_mode,
                      int64_t *out_idx, float *out_scores, float *out_boxes, int32_t *out_count,
                      void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_topk(N))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "topk: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_topk(N));
    const int nseg = (int)((N + TOPK_SEG - 1) / TOPK_SEG);
    const int nrow = (int)((N + TOPK_ROWS - 1) / TOPK_ROWS);
    int32_t *partial = (int32_t *)ws;
    FRCNN_LAUNCH(KID_TOPK_RANK, topk_rank_kernel, dim3(nrow, nseg), dim3(TOPK_ROWS), 0, s, scores, (int)N, partial, out_count);
    FRCNN_CHECK_LAUNCH("topk_rank_kernel");
    FRCNN_LAUNCH(KID_TOPK_SCATTER, topk_scatter_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, scores,
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
    return frcnn_launch_topk(scores, boxes_in, N, K, 1, out_idx, out_scores, out_boxes, out_count, workspace, workspace_bytes, s);
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
    return frcnn_launch_topk(scores, boxes_in, N, N, 0, out_idx, out_scores, out_boxes, out_count, workspace, workspace_bytes, s);
}
