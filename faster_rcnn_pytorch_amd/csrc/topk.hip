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

// ------------------------------------------------------------------------------------------------
// Large N with K << N (FPN: N = 268 569, K = 4000): the O(N^2) rank sort would be ~72 G compares, so a two-level
// radix histogram (11 + 11 bits of the order-preserving key) first finds a 22-bit prefix threshold that at least K
// keys reach; those M >= K candidates (M - K = ties inside one 2^-13-relative score band, typically tens) are
// compacted and rank-sorted among themselves with the explicit (key desc, index asc) order.
//   topk_hist_kernel<0>  ->  topk_hist_kernel<1>  ->  topk_compact_kernel  ->  topk_rank_cand_kernel  ->  topk_scatter_cand_kernel
// Every kernel re-derives what it needs from the previous histogram in its prologue (2048 bins, one block scan), so
// there is no single-block "pick the digit" launch in between.
// ------------------------------------------------------------------------------------------------
#define SEL_BINS 2048

struct SelCtl { unsigned hist1[SEL_BINS]; unsigned hist2[SEL_BINS]; int m; int pad[15]; };

// descending search: returns the bin holding the `want`-th largest key (1-based) and the number of keys in higher bins;
// if the histogram holds fewer than `want` keys, returns bin 0 (everything qualifies).  256 threads.
__device__ __forceinline__ void sel_find_bin(const unsigned *__restrict__ hist, int want, int *s_tmp /*[8]*/, int *bin, int *above)
{
    const int t = threadIdx.x;                       // thread t owns descending bins d = 8t .. 8t+7  (bin = 2047 - d)
    unsigned c[8];
    int local = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { c[q] = hist[SEL_BINS - 1 - (8 * t + q)]; local += (int)c[q]; }
    // block-wide exclusive scan of `local` over 256 threads
    const int lane = t & 63, wave = t >> 6;
    int inc = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    __syncthreads();
    if (lane == 63) s_tmp[wave] = inc;
    if (t == 0) { s_tmp[4] = 0; s_tmp[5] = 0; s_tmp[6] = 0; }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wave) base += s_tmp[w];
    const int excl = base + inc - local;
    if (excl < want && want <= excl + local) {       // exactly one thread (if the total reaches `want`)
        int acc = excl, q = 0;
        for (; q < 7; ++q) {
            if (acc + (int)c[q] >= want) break;
            acc += (int)c[q];
        }
        s_tmp[4] = SEL_BINS - 1 - (8 * t + q);
        s_tmp[5] = acc;
        s_tmp[6] = 1;
    }
    __syncthreads();
    *bin = s_tmp[6] ? s_tmp[4] : 0;
    *above = s_tmp[6] ? s_tmp[5] : 0;
    __syncthreads();
}

template <int LEVEL>
__global__ __launch_bounds__(256) void topk_hist_kernel(const float *__restrict__ scores, int N, int K, int proposal_mode, SelCtl *__restrict__ ctl)
{
    __shared__ unsigned s_hist[SEL_BINS];
    __shared__ int s_tmp[8];
    int bin1 = 0, above1 = 0;
    if (LEVEL == 1) sel_find_bin(ctl->hist1, K, s_tmp, &bin1, &above1);
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) s_hist[i] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) {
        const float sc = scores[i];
        if (proposal_mode && !(sc >= 0.0f)) continue;
        const uint32_t k = f2key(sc);
        if (LEVEL == 0) atomicAdd(&s_hist[k >> 21], 1u);
        else if ((int)(k >> 21) == bin1) atomicAdd(&s_hist[(k >> 10) & (SEL_BINS - 1)], 1u);
    }
    __syncthreads();
    unsigned *dst = LEVEL == 0 ? ctl->hist1 : ctl->hist2;
    for (int i = threadIdx.x; i < SEL_BINS; i += 256)
        if (s_hist[i]) atomicAdd(&dst[i], s_hist[i]);
}

__global__ __launch_bounds__(256) void topk_compact_kernel(const float *__restrict__ scores, int N, int K, int proposal_mode,
                                                           SelCtl *__restrict__ ctl, uint32_t *__restrict__ cand_key,
                                                           int32_t *__restrict__ cand_idx, int32_t *__restrict__ cand_rank)
{
    __shared__ int s_tmp[8];
    int bin1, above1, bin2, above2;
    sel_find_bin(ctl->hist1, K, s_tmp, &bin1, &above1);
    sel_find_bin(ctl->hist2, K - above1, s_tmp, &bin2, &above2);
    const uint32_t thr22 = ((uint32_t)bin1 << 11) | (uint32_t)bin2;      // keep keys whose top 22 bits reach this
    for (int i0 = blockIdx.x * 256; i0 < N; i0 += gridDim.x * 256) {
        const int i = i0 + threadIdx.x;
        bool c = false;
        uint32_t k = 0u;
        if (i < N) {
            const float sc = scores[i];
            k = f2key(sc);
            c = (!proposal_mode || sc >= 0.0f) && (k >> 10) >= thr22;
        }
        const unsigned long long bm = __ballot(c);
        if (bm != 0ull) {
            int base = 0;
            if ((threadIdx.x & 63) == 0) base = atomicAdd(&ctl->m, __builtin_popcountll(bm));
            base = __shfl(base, 0);
            if (c) {
                const int slot = base + __builtin_popcountll(bm & ((1ull << (threadIdx.x & 63)) - 1ull));
                cand_key[slot] = k;
                cand_idx[slot] = i;
                cand_rank[slot] = 0;
            }
        }
    }
}

// rank among the M candidates (M on the device); persistent grid over (256-row block) x (1024-column segment) tiles
__global__ __launch_bounds__(256) void topk_rank_cand_kernel(const SelCtl *__restrict__ ctl, const uint32_t *__restrict__ cand_key,
                                                             const int32_t *__restrict__ cand_idx, int32_t *__restrict__ cand_rank)
{
    __shared__ uint32_t s_k[TOPK_SEG];
    __shared__ int32_t s_i[TOPK_SEG];
    const int M = ctl->m;
    const int nrow = (M + TOPK_ROWS - 1) / TOPK_ROWS, nseg = (M + TOPK_SEG - 1) / TOPK_SEG;
    for (int tile = blockIdx.x; tile < nrow * nseg; tile += gridDim.x) {
        const int rb = tile / nseg, sg = tile - rb * nseg;
        const int c0 = sg * TOPK_SEG;
        const int cn = min(TOPK_SEG, M - c0);
        __syncthreads();
        for (int t = threadIdx.x; t < cn; t += 256) { s_k[t] = cand_key[c0 + t]; s_i[t] = cand_idx[c0 + t]; }
        __syncthreads();
        const int r = rb * TOPK_ROWS + threadIdx.x;
        if (r < M) {
            const uint32_t kr = cand_key[r];
            const int32_t ir = cand_idx[r];
            int rank = 0;
            for (int t = 0; t < cn; ++t) {
                const uint32_t k = s_k[t];
                rank += (k > kr) || (k == kr && s_i[t] < ir);
            }
            if (rank) atomicAdd(&cand_rank[r], rank);
        }
    }
}

__global__ __launch_bounds__(256) void topk_scatter_cand_kernel(const SelCtl *__restrict__ ctl, const float *__restrict__ scores,
                                                                const float4 *__restrict__ boxes_in, const int32_t *__restrict__ cand_idx,
                                                                const int32_t *__restrict__ cand_rank, int K, int64_t *__restrict__ out_idx,
                                                                float *__restrict__ out_scores, float4 *__restrict__ out_boxes,
                                                                int32_t *__restrict__ out_count)
{
    const int M = ctl->m;
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = M < K ? M : K;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < M; r += gridDim.x * 256) {
        const int rank = cand_rank[r];
        if (rank < K) {
            const int i = cand_idx[r];
            out_idx[rank] = i;
            out_scores[rank] = scores[i];
            if (out_boxes) out_boxes[rank] = boxes_in[i];
        }
    }
}

static inline bool topk_use_select(int64_t N, int64_t K) { return N > 32768 && K * 4 <= N; }

size_t frcnn_ws_topk(int64_t N)
{
    const int64_t nseg = (N + TOPK_SEG - 1) / TOPK_SEG;
    const size_t direct = N > 32768 ? 0 : align_up((size_t)(nseg > 0 ? nseg : 1) * (size_t)N * sizeof(int32_t), 256);
    const size_t direct_big = align_up((size_t)(nseg > 0 ? nseg : 1) * (size_t)N * sizeof(int32_t), 256);
    const size_t select = align_up(sizeof(SelCtl), 256) + 3 * align_up((size_t)N * 4, 256);
    // N > 32768: the select path needs `select`; the direct path (K close to N) needs `direct_big`
    return N > 32768 ? (direct_big > select ? direct_big : select) : direct;
}

int frcnn_launch_topk(const float *scores, const float *boxes_in, int64_t N, int64_t K, int proposal_mode,
                      int64_t *out_idx, float *out_scores, float *out_boxes, int32_t *out_count,
                      void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_topk(N))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "topk: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_topk(N));
    if (topk_use_select(N, K)) {
        char *p = (char *)ws;
        SelCtl *ctl = (SelCtl *)p; p += align_up(sizeof(SelCtl), 256);
        uint32_t *cand_key = (uint32_t *)p; p += align_up((size_t)N * 4, 256);
        int32_t *cand_idx = (int32_t *)p; p += align_up((size_t)N * 4, 256);
        int32_t *cand_rank = (int32_t *)p;
        if (hipMemsetAsync(ctl, 0, sizeof(SelCtl), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "topk: memset failed");
        const int gb = (int)((N + 2047) / 2048) < 1024 ? (int)((N + 2047) / 2048) : 1024;
        FRCNN_LAUNCH(KID_TOPK_RANK, topk_hist_kernel<0>, dim3(gb), dim3(256), 0, s, scores, (int)N, (int)K, proposal_mode, ctl);
        FRCNN_CHECK_LAUNCH("topk_hist_kernel<0>");
        FRCNN_LAUNCH(KID_TOPK_RANK, topk_hist_kernel<1>, dim3(gb), dim3(256), 0, s, scores, (int)N, (int)K, proposal_mode, ctl);
        FRCNN_CHECK_LAUNCH("topk_hist_kernel<1>");
        FRCNN_LAUNCH(KID_TOPK_RANK, topk_compact_kernel, dim3(gb), dim3(256), 0, s, scores, (int)N, (int)K, proposal_mode, ctl, cand_key,
                     cand_idx, cand_rank);
        FRCNN_CHECK_LAUNCH("topk_compact_kernel");
        FRCNN_LAUNCH(KID_TOPK_RANK, topk_rank_cand_kernel, dim3(1024), dim3(256), 0, s, ctl, cand_key, cand_idx, cand_rank);
        FRCNN_CHECK_LAUNCH("topk_rank_cand_kernel");
        FRCNN_LAUNCH(KID_TOPK_SCATTER, topk_scatter_cand_kernel, dim3(256), dim3(256), 0, s, ctl, scores, (const float4 *)boxes_in, cand_idx,
                     cand_rank, (int)K, out_idx, out_scores, (float4 *)out_boxes, out_count);
        FRCNN_CHECK_LAUNCH("topk_scatter_cand_kernel");
        return FRCNN_OK;
    }
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
