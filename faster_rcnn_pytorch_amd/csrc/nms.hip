// nms.hip -- torchvision.ops.nms as the reference calls it (models/model.py:53,394), gfx950.
//
// Two kernels, no host round trip (torchvision copies the 18 MB mask to the host and scans there):
//  nms_mask_kernel : chip-wide.  Upper-triangular 64x64 tiles of the suppression matrix; one wave per
//      tile, lane = row box, the 64 column boxes arrive through the scalar path (wave-uniform
//      addresses), each lane builds one uint64 word with 64 IoU tests.  The decision
//      inter/(a_i+a_j-inter) > thr is taken WITHOUT the IEEE division on the fast path: if inter is
//      outside a 2^-20 relative band around thr*union the comparison is already decided; only inside
//      the band is the exact division evaluated, so results are bit-identical to the oracle.
//  nms_scan_kernel : one 1024-thread workgroup.  Wave 0 resolves each 64-box block with scalar bit
//      tricks (s_ff1 + v_readlane on the diagonal words), then all 16 waves OR the mask rows of the
//      boxes just kept into an LDS-resident `removed` bit vector (ds_or_b64).  Emits the first post_k
//      kept positions, their boxes and the count; stops as soon as post_k boxes are kept.
#include "frcnn_common.h"
#include "frcnn_internal.h"

#define NMS_MAX_BLOCKS 4096            // K <= 262144

__device__ __forceinline__ bool nms_suppress(float4 a, float area_a, float4 b, float area_b, float thr)
{
    const float xx1 = a.x > b.x ? a.x : b.x;
    const float yy1 = a.y > b.y ? a.y : b.y;
    const float xx2 = a.z < b.z ? a.z : b.z;
    const float yy2 = a.w < b.w ? a.w : b.w;
    float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
    float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
    const float inter = w * h;
    const float uni = area_a + area_b - inter;
    // fast path: inter vs thr*uni with a guard band (relative 2^-20 >> the 2^-23 of two roundings)
    const float p = thr * uni;
    const float d = inter - p;
    const float band = __builtin_fabsf(p) * 9.5367431640625e-07f;
    if (uni > 0.0f && __builtin_fabsf(d) > band) return d > 0.0f;
    return inter / uni > thr;          // exact IEEE division (also the NaN / inf / zero-area cases)
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_dev, int K,
                                                       float thr, int nblk, unsigned long long *__restrict__ mask)
{
    const int n = n_dev ? min(*n_dev, K) : K;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int rb = blockIdx.y;
    const int cb = blockIdx.x * 4 + wave;
    if (cb < rb || cb >= nblk) return;
    if (rb * 64 >= n) return;                                   // dead rows: the scan never reads them
    const int row = rb * 64 + lane;
    const float4 a = boxes[min(row, K - 1)];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    unsigned long long bits = 0ull;
    const int c0 = cb * 64;
    if (c0 < n) {
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const int col = c0 + j;                             // wave-uniform
            const float4 b = boxes[min(col, K - 1)];            // scalar load
            const float area_b = (b.z - b.x) * (b.w - b.y);
            const bool s = nms_suppress(a, area_a, b, area_b, thr) && col > row && col < n;
            bits |= s ? (1ull << j) : 0ull;
        }
    }
    if (row < K) mask[(size_t)row * nblk + cb] = bits;
}

__global__ __launch_bounds__(1024) void nms_scan_kernel(const float4 *__restrict__ boxes, const int32_t *__restrict__ n_dev, int K,
                                                        int nblk, const unsigned long long *__restrict__ mask, int post_k,
                                                        int64_t *__restrict__ out_keep, float4 *__restrict__ out_rois,
                                                        const int64_t *__restrict__ src_map, int64_t *__restrict__ out_src,
                                                        int32_t *__restrict__ out_count)
{
    extern __shared__ unsigned long long removed[];             // [nblk]
    __shared__ unsigned long long s_kept;
    __shared__ int s_total;
    __shared__ int s_rows[64];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int n = n_dev ? min(*n_dev, K) : K;
    const int nb = (n + 63) >> 6;
    for (int w = tid; w < nb; w += 1024) removed[w] = 0ull;
    if (tid == 0) { s_total = 0; s_kept = 0ull; }
    __syncthreads();
    int total = 0;
    for (int b = 0; b < nb; ++b) {
        if (wave == 0) {
            const int row = b * 64 + lane;
            const unsigned long long d = row < n ? mask[(size_t)row * nblk + b] : 0ull;   // diagonal word of my row
            const int live = n - b * 64;
            const unsigned long long valid = live >= 64 ? ~0ull : ((1ull << live) - 1ull);
            const unsigned long long rem = removed[b];
            // NB: the readfirstlane/readlane builtins return int: go through unsigned or the low word sign-extends
            unsigned long long alive = ~(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(rem >> 32)) << 32) |
                                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)rem)) & valid;
            unsigned long long kept = 0ull;
            int cnt = 0;
            const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
            while (alive != 0ull && total + cnt < post_k) {                                // wave-uniform loop
                const int i = __builtin_ctzll(alive);
                kept |= 1ull << i;
                ++cnt;
                const unsigned long long di = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(dlo, i) |
                                              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, i) << 32);
                alive &= ~(di | (1ull << i));
            }
            if ((kept >> lane) & 1ull) {
                const int pos = total + __builtin_popcountll(kept & ((1ull << lane) - 1ull));
                out_keep[pos] = row;
                if (out_rois) out_rois[pos] = boxes[row];
                if (out_src) out_src[pos] = src_map ? src_map[row] : (int64_t)row;
                s_rows[pos - total] = lane;
            }
            if (lane == 0) { s_kept = kept; s_total = total + cnt; }
        }
        __syncthreads();
        const unsigned long long kept = s_kept;
        total = s_total;
        if (total >= post_k) break;
        const int nrows = __builtin_popcountll(kept);
        if (nrows > 0 && b + 1 < nb) {
            for (int ri = wave; ri < nrows; ri += 16) {
                const size_t rowbase = (size_t)(b * 64 + s_rows[ri]) * nblk;
                for (int w = b + 1 + lane; w < nb; w += 64) {
                    const unsigned long long v = mask[rowbase + w];
                    if (v) atomicOr(&removed[w], v);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) *out_count = total < post_k ? total : post_k;
}

size_t frcnn_ws_nms(int64_t K)
{
    const int64_t nblk = (K + 63) / 64;
    return align_up((size_t)K * (size_t)nblk * 8, 256);
}

int frcnn_launch_nms(const float *boxes, const int32_t *n_boxes_dev, int64_t K, float thr, int64_t post_k,
                     int64_t *out_keep, float *out_rois, const int64_t *src_map, int64_t *out_src, int32_t *out_count,
                     void *ws, size_t ws_bytes, hipStream_t s)
{
    if (ws_bytes < frcnn_ws_nms(K))
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "nms: workspace %zu < %zu bytes", ws_bytes, frcnn_ws_nms(K));
    const int nblk = (int)((K + 63) / 64);
    if (nblk > NMS_MAX_BLOCKS) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "nms: K=%lld above limit %d", (long long)K, NMS_MAX_BLOCKS * 64);
    unsigned long long *mask = (unsigned long long *)ws;
    FRCNN_LAUNCH(KID_NMS_MASK, nms_mask_kernel, dim3((nblk + 3) / 4, nblk), dim3(256), 0, s, (const float4 *)boxes, n_boxes_dev, (int)K,
                 thr, nblk, mask);
    FRCNN_CHECK_LAUNCH("nms_mask_kernel");
    FRCNN_LAUNCH(KID_NMS_SCAN, nms_scan_kernel, dim3(1), dim3(1024), (size_t)nblk * 8, s, (const float4 *)boxes, n_boxes_dev, (int)K, nblk,
                 mask, (int)post_k, out_keep, (float4 *)out_rois, src_map, out_src, out_count);
    FRCNN_CHECK_LAUNCH("nms_scan_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_nms(const float *boxes, const int32_t *n_boxes_dev, int64_t K, float iou_threshold, int64_t post_k,
                           int64_t *out_keep, float *out_rois, int32_t *out_count, void *workspace, size_t workspace_bytes,
                           void *stream)
{
    FRCNN_REQUIRE(K >= 0 && post_k >= 0, "nms: negative size");
    FRCNN_REQUIRE(out_count, "nms: NULL out_count");
    hipStream_t s = (hipStream_t)stream;
    if (K == 0 || post_k == 0) {
        if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "nms: memset failed");
        return FRCNN_OK;
    }
    FRCNN_REQUIRE(boxes && out_keep && workspace, "nms: NULL pointer");
    return frcnn_launch_nms(boxes, n_boxes_dev, K, iou_threshold, post_k, out_keep, out_rois, nullptr, nullptr, out_count, workspace,
                            workspace_bytes, s);
}
