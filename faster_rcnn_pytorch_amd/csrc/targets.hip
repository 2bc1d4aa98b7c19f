// targets.hip -- RPNTargetMaker.forward and FastRcnnTargetMaker.forward on the device (gfx950).
//
// Reference: models/model_.py:186-266 (VGG RPN), models/new_model.py:299-349 (FPN RPN),
//            models/model_.py:127-179 (VGG head), models/new_model.py:157-206 (FPN head).
// The reference builds an [n_anchor, G] IoU matrix with ~25 eager launches, syncs the host 4-6
// times (boolean indexing, `if n_pos > 128`) and draws torch.randperm on the CPU.  Here:
//   rpn_colmax_kernel : per-GT best anchor, packed (iou_bits << 32 | ~index) + atomicMax (LDS, then global)
//   rpn_label_kernel  : per-anchor max/argmax over the G boxes held in SGPRs (scalar loads), label,
//                       encode(), counts -- the IoU matrix is never materialised
//   rpn_sample_kernel : one workgroup; either consumes the reference's permutations (parity mode) or
//                       selects by smallest Philox key with an LDS radix select (no host sync)
//   head_targets_kernel: one workgroup does IoU + ordered compaction + sampling + encode for <= 4096
//                       candidates and writes the fixed [total] rows.
#include "frcnn_common.h"
#include "frcnn_internal.h"

#define EPS_JACCARD 1e-5f

// IoU of candidate box `b` against gt `g` in the operand order of the reference variant
__device__ __forceinline__ float iou_variant(int variant, float4 b, float4 g)
{
    return variant == 1 ? iou_pair<false>(g, b, 0.f) : iou_pair<true>(b, g, EPS_JACCARD);
}
__device__ __forceinline__ bool anchor_inside(float4 a) { return a.x >= 0.0f && a.y >= 0.0f && a.z <= 1.0f && a.w <= 1.0f; }

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rpn_colmax_kernel(int variant, const float4 *__restrict__ anchors, int N,
                                                         const float4 *__restrict__ gt, int G,
                                                         unsigned long long *__restrict__ colkey, int32_t *__restrict__ counts_zero)
{
    extern __shared__ unsigned long long s_key[];      // [G]
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (counts_zero && blockIdx.x == 0 && threadIdx.x < 4) counts_zero[threadIdx.x] = 0;
    for (int g = threadIdx.x; g < G; g += 256) s_key[g] = 0ull;
    __syncthreads();
    if (i < N) {
        const float4 a = anchors[i];
        if (variant == 1 || anchor_inside(a)) {
            for (int g = 0; g < G; ++g) {
                const float v = iou_variant(variant, a, gt[g]);
                if (v >= 0.0f) {        // NaN never wins
                    const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
                    // cheap pre-test against the block's current best to keep LDS atomics rare
                    if (key > s_key[g]) atomicMax(&s_key[g], key);
                }
            }
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256)
        if (s_key[g] != 0ull) atomicMax(&colkey[g], s_key[g]);
}

__global__ __launch_bounds__(256) void rpn_label_kernel(int variant, const float4 *__restrict__ anchors, int N,
                                                        const float4 *__restrict__ gt, int G,
                                                        const unsigned long long *__restrict__ colkey,
                                                        int64_t *__restrict__ out_cls, float4 *__restrict__ out_reg,
                                                        int8_t *__restrict__ label8, int32_t *__restrict__ counts)
{
    __shared__ int s_cnt[2][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int lab = -1;
    if (i < N) {
        const float4 a = anchors[i];
        float4 reg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (variant == 1 || anchor_inside(a)) {
            float best = -__builtin_inff();
            int arg = 0;
            bool match = false;
            for (int g = 0; g < G; ++g) {
                const float v = iou_variant(variant, a, gt[g]);
                if (v > best) { best = v; arg = g; }
                const unsigned long long ck = colkey[g];
                if (variant == 1) match |= (v == __uint_as_float((unsigned)(ck >> 32))) && ck != 0ull;
                else match |= (0xFFFFFFFFu - (unsigned)ck) == (unsigned)i && ck != 0ull;
            }
            if (best < 0.3f) lab = 0;
            if (match) lab = 1;
            if (best >= 0.7f) lab = 1;
            reg = encode4(xy_to_cxcy4(gt[arg]), xy_to_cxcy4(a));
        }
        out_cls[i] = lab;
        out_reg[i] = reg;
        label8[i] = (int8_t)lab;
    }
    // block counts of positives / negatives
    const unsigned long long bp = __ballot(lab == 1), bn = __ballot(lab == 0);
    if ((threadIdx.x & 63) == 0) { s_cnt[0][threadIdx.x >> 6] = __builtin_popcountll(bp); s_cnt[1][threadIdx.x >> 6] = __builtin_popcountll(bn); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int c = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
        if (c) atomicAdd(&counts[threadIdx.x], c);
    }
}

// ------------------------------------------------------------------------------------------------
// block-wide helpers for the single-workgroup kernels (1024 threads = 16 waves)
// ------------------------------------------------------------------------------------------------
// exclusive prefix of `v` over the block in thread order; *total = block sum.  s_w: 17 ints of LDS.
__device__ __forceinline__ int block_excl_scan_1024(int v, int *s_w, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();                       // protect s_w reuse
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = s_w[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

// Radix select over 32-bit keys of the flagged elements: finds T = the `keep`-th smallest key
// (1-based) and `rem` = how many elements with key == T still belong to the kept set.
// Elements are owned chunk-wise: thread t owns [t*chunk, min((t+1)*chunk, n)).
template <typename KeyFn, typename FlagFn>
__device__ void block_radix_select(int n, int chunk, int keep, KeyFn key_of, FlagFn is_cand, unsigned *s_hist /*[256]*/,
                                   unsigned *s_pref /*[2]*/, unsigned *outT, int *outRem)
{
    unsigned prefix = 0u;
    int remaining = keep;
    const int lo = threadIdx.x * chunk, hi = min(lo + chunk, n);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (threadIdx.x < 256) s_hist[threadIdx.x] = 0u;
        __syncthreads();
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = lo; i < hi; ++i)
            if (is_cand(i)) {
                const unsigned k = key_of(i);
                if ((k & himask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
            }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0;
            unsigned d = 0;
            for (; d < 256u; ++d) {
                const int c = (int)s_hist[d];
                if (acc + c >= remaining) break;
                acc += c;
            }
            s_pref[0] = prefix | (d << shift);
            s_pref[1] = (unsigned)(remaining - acc);
        }
        __syncthreads();
        prefix = s_pref[0];
        remaining = (int)s_pref[1];
        __syncthreads();
    }
    *outT = prefix;
    *outRem = remaining;
}

// ------------------------------------------------------------------------------------------------
// rpn_sample_kernel: models/model_.py:225-236
//   if n_pos > 128: label[pos_indices[perm[128:]]] = -1
//   if n_neg > 256 - n_pos: label[neg_indices[perm[(256 - min(n_pos,128)):]]] = -1
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rpn_sample_kernel(int N, int8_t *__restrict__ label8, int64_t *__restrict__ out_cls,
                                                          const int64_t *__restrict__ perm_pos, int n_perm_pos,
                                                          const int64_t *__restrict__ perm_neg, int n_perm_neg,
                                                          unsigned long long seed, unsigned long long offset,
                                                          int32_t *__restrict__ list, unsigned *__restrict__ keys,
                                                          int32_t *__restrict__ counts)
{
    __shared__ int s_w[17];
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_pref[2];
    const int n_pos = counts[0], n_neg = counts[1];
    const int np_eff = min(n_pos, 128);
    const bool drop_pos = n_pos > 128;
    const bool drop_neg = n_neg > 256 - n_pos;
    if (!drop_pos && !drop_neg) return;
    const bool host_mode = (perm_pos != nullptr) || (perm_neg != nullptr);
    const int chunk = (N + 1023) / 1024;
    const int lo = threadIdx.x * chunk, hi = min(lo + chunk, N);

    for (int cls_id = 1; cls_id >= 0; --cls_id) {          // positives first (reference order), then negatives
        const bool drop = cls_id == 1 ? drop_pos : drop_neg;
        if (!drop) continue;
        const int n_c = cls_id == 1 ? n_pos : n_neg;
        const int keep = cls_id == 1 ? 128 : 256 - np_eff;
        const int8_t want = (int8_t)cls_id;
        if (host_mode) {
            const int64_t *perm = cls_id == 1 ? perm_pos : perm_neg;
            const int n_perm = cls_id == 1 ? n_perm_pos : n_perm_neg;
            if (perm == nullptr || n_perm != n_c) {         // uniform
                if (threadIdx.x == 0) counts[2] = 1;
                continue;
            }
            int c = 0;
            for (int i = lo; i < hi; ++i) c += label8[i] == want;
            int tot;
            int base = block_excl_scan_1024(c, s_w, &tot);
            for (int i = lo; i < hi; ++i)
                if (label8[i] == want) list[base++] = i;
            __syncthreads();
            for (int j = keep + threadIdx.x; j < n_c; j += 1024) {
                const int64_t p = perm[j];
                if (p >= 0 && p < n_c) { const int i = list[p]; out_cls[i] = -1; }
                else counts[2] = 2;
            }
            __syncthreads();
            // label8 is stale for the demoted entries from here on; the negative pass only looks at label 0
        } else {
            const unsigned stream_id = (unsigned)cls_id;
            for (int i = lo; i < hi; ++i)
                if (label8[i] == want) keys[i] = philox_first(seed, offset, stream_id, (unsigned)i);
            __syncthreads();
            unsigned T; int rem;
            block_radix_select(N, chunk, keep, [&](int i) { return keys[i]; }, [&](int i) { return label8[i] == want; }, s_hist, s_pref, &T, &rem);
            // ties at T: the first `rem` in index order stay
            int c = 0;
            for (int i = lo; i < hi; ++i) c += (label8[i] == want && keys[i] == T);
            int tot;
            int base = block_excl_scan_1024(c, s_w, &tot);
            for (int i = lo; i < hi; ++i)
                if (label8[i] == want) {
                    const unsigned k = keys[i];
                    bool kept = k < T;
                    if (k == T) { kept = base < rem; ++base; }
                    if (!kept) out_cls[i] = -1;
                }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// head_targets_kernel (one workgroup of 1024 threads, n = n_rois + G <= HT_MAX candidates)
// ------------------------------------------------------------------------------------------------
#define HT_MAX 4096
#define HT_ROWS_MAX 1024

__global__ __launch_bounds__(1024) void head_targets_kernel(int variant, const float4 *__restrict__ rois, const int32_t *__restrict__ n_rois_dev,
                                                            int P_cap, const float4 *__restrict__ gt, const int64_t *__restrict__ gt_label, int G,
                                                            int label_offset, int max_pos, int total,
                                                            const int64_t *__restrict__ perm_pos, int n_perm_pos,
                                                            const int64_t *__restrict__ perm_neg, int n_perm_neg,
                                                            unsigned long long seed, unsigned long long offset,
                                                            int64_t *__restrict__ out_cls, float4 *__restrict__ out_reg,
                                                            float4 *__restrict__ out_rois, int64_t *__restrict__ out_keep,
                                                            int32_t *__restrict__ counts)
{
    __shared__ int s_w[17];
    __shared__ short s_arg[HT_MAX];
    __shared__ signed char s_flag[HT_MAX];          // 1 pos cand, 0 neg cand, -1 neither
    __shared__ unsigned short s_list[2][HT_MAX];    // [0] pos, [1] neg candidate ids (< HT_MAX), ascending
    __shared__ int s_err;
    __shared__ unsigned s_key[HT_MAX];
    __shared__ int s_row[HT_ROWS_MAX];
    const int tid = threadIdx.x;
    const int n_rois = n_rois_dev ? min(max(*n_rois_dev, 0), P_cap) : P_cap;
    const int n = n_rois + G;

    for (int j = tid; j < total; j += 1024) s_row[j] = -1;
    if (tid == 0) s_err = 0;
    // phase 1+2: IoU max/argmax and ordered compaction, 1024 candidates per round
    int npc = 0, nnc = 0;
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = k0 + tid;
        int flag = -1;
        if (k < n) {
            const float4 b = k < n_rois ? rois[k] : gt[k - n_rois];
            float best = -__builtin_inff();
            int arg = 0;
            for (int g = 0; g < G; ++g) {
                const float v = iou_variant(variant, b, gt[g]);
                if (v > best) { best = v; arg = g; }
            }
            s_arg[k] = (short)arg;
            if (best >= 0.5f) flag = 1;
            else if (best < 0.5f && best >= 0.0f) flag = 0;
            s_flag[k] = (signed char)flag;
        }
        int tp, tn;
        const int bp = block_excl_scan_1024(flag == 1, s_w, &tp);
        if (flag == 1) s_list[0][npc + bp] = (unsigned short)k;
        const int bn = block_excl_scan_1024(flag == 0, s_w, &tn);
        if (flag == 0) s_list[1][nnc + bn] = (unsigned short)k;
        npc += tp; nnc += tn;
    }
    __syncthreads();
    const int n_pos = min(npc, max_pos);
    const int n_neg = min(total - n_pos, nnc);
    const bool host_mode = (perm_pos != nullptr) || (perm_neg != nullptr);
    if (host_mode) {
        if (perm_pos == nullptr || perm_neg == nullptr || n_perm_pos != npc || n_perm_neg != nnc) { if (tid == 0) s_err = 1; }
        else {
            for (int j = tid; j < n_pos + n_neg; j += 1024) {
                const int64_t p = j < n_pos ? perm_pos[j] : perm_neg[j - n_pos];
                const int lim = j < n_pos ? npc : nnc;
                if (p >= 0 && p < lim) s_row[j] = s_list[j < n_pos ? 0 : 1][p];
                else s_err = 2;
            }
        }
    } else {
        // rank by (philox key, candidate id) inside each list; rank r < quota -> output row
        for (int which = 0; which < 2; ++which) {
            const int m = which == 0 ? npc : nnc;
            const int quota = which == 0 ? n_pos : n_neg;
            const int rowbase = which == 0 ? 0 : n_pos;
            __syncthreads();
            for (int q = tid; q < m; q += 1024) s_key[q] = philox_first(seed, offset, 2u + (unsigned)which, (unsigned)s_list[which][q]);
            __syncthreads();
            if (quota > 0)
                for (int q = tid; q < m; q += 1024) {
                    const unsigned kq = s_key[q];
                    int rank = 0;
                    for (int o = 0; o < m; ++o) {
                        const unsigned ko = s_key[o];
                        rank += (ko < kq) || (ko == kq && o < q);
                    }
                    if (rank < quota) s_row[rowbase + rank] = s_list[which][q];
                }
        }
    }
    __syncthreads();
    // phase 4: rows
    for (int j = tid; j < total; j += 1024) {
        const int k = s_row[j];
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f), reg = box;
        int64_t cls = 0;
        if (k >= 0) {
            box = k < n_rois ? rois[k] : gt[k - n_rois];
            const int arg = s_arg[k];
            if (j < n_pos) cls = gt_label[arg] + label_offset;
            const float4 e = encode4(xy_to_cxcy4(gt[arg]), xy_to_cxcy4(box));
            reg = make_float4((e.x - 0.0f) / 0.1f, (e.y - 0.0f) / 0.1f, (e.z - 0.0f) / 0.2f, (e.w - 0.0f) / 0.2f);
        }
        out_cls[j] = cls;
        out_reg[j] = reg;
        out_rois[j] = box;
        if (out_keep) out_keep[j] = k;
    }
    if (tid == 0) { const int err = s_err; counts[0] = npc; counts[1] = nnc; counts[2] = err ? 0 : n_pos + n_neg; counts[3] = err; }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
struct RpnWs { unsigned long long *colkey; int8_t *label8; int32_t *list; unsigned *keys; size_t total; };
static RpnWs carve_rpn(void *ws, int64_t N, int64_t G)
{
    RpnWs w; char *p = (char *)ws; size_t o = 0;
    auto take = [&](size_t b) { void *r = p ? p + o : nullptr; o += align_up(b, 256); return r; };
    w.colkey = (unsigned long long *)take((size_t)G * 8);
    w.label8 = (int8_t *)take((size_t)N);
    w.list = (int32_t *)take((size_t)N * 4);
    w.keys = (unsigned *)take((size_t)N * 4);
    w.total = o;
    return w;
}
size_t frcnn_ws_rpn_targets(int64_t N, int64_t G) { return carve_rpn(nullptr, N, G).total; }
size_t frcnn_ws_head_targets(int64_t) { return 256; }

FRCNN_EXPORT int frcnn_rpn_targets(int variant, const float *anchors, int64_t N, const float *gt, int64_t G,
                                   const int64_t *perm_pos, int64_t n_perm_pos, const int64_t *perm_neg, int64_t n_perm_neg,
                                   uint64_t seed, uint64_t offset, int64_t *out_cls, float *out_reg, int32_t *out_counts,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(variant == 0 || variant == 1, "rpn_targets: variant must be 0 (VGG) or 1 (FPN)");
    FRCNN_REQUIRE(N > 0 && N < ((int64_t)1 << 31), "rpn_targets: bad N");
    FRCNN_REQUIRE(G > 0, "rpn_targets: G must be >= 1 (the reference fails on an image without boxes)");
    FRCNN_REQUIRE(G <= 4096, "rpn_targets: G=%lld above limit 4096", (long long)G);
    FRCNN_REQUIRE(anchors && gt && out_cls && out_reg && out_counts && workspace, "rpn_targets: NULL pointer");
    FRCNN_REQUIRE(n_perm_pos >= 0 && n_perm_neg >= 0 && n_perm_pos < ((int64_t)1 << 31) && n_perm_neg < ((int64_t)1 << 31), "rpn_targets: bad perm length");
    RpnWs w = carve_rpn(workspace, N, G);
    if (workspace_bytes < w.total) return frcnn_set_error(FRCNN_ERR_WORKSPACE, "rpn_targets: workspace %zu < %zu bytes", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(w.colkey, 0, (size_t)G * 8, s) != hipSuccess) return frcnn_set_error(FRCNN_ERR_LAUNCH, "rpn_targets: memset failed");
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    FRCNN_LAUNCH(KID_RPN_COLMAX, rpn_colmax_kernel, grid, block, (size_t)G * 8, s, variant, (const float4 *)anchors, (int)N, (const float4 *)gt,
                 (int)G, w.colkey, out_counts);
    FRCNN_CHECK_LAUNCH("rpn_colmax_kernel");
    FRCNN_LAUNCH(KID_RPN_LABEL, rpn_label_kernel, grid, block, 0, s, variant, (const float4 *)anchors, (int)N, (const float4 *)gt, (int)G,
                 w.colkey, out_cls, (float4 *)out_reg, w.label8, out_counts);
    FRCNN_CHECK_LAUNCH("rpn_label_kernel");
    FRCNN_LAUNCH(KID_RPN_SAMPLE, rpn_sample_kernel, dim3(1), dim3(1024), 0, s, (int)N, w.label8, out_cls, perm_pos, (int)n_perm_pos, perm_neg,
                 (int)n_perm_neg, (unsigned long long)seed, (unsigned long long)offset, w.list, w.keys, out_counts);
    FRCNN_CHECK_LAUNCH("rpn_sample_kernel");
    return FRCNN_OK;
}

FRCNN_EXPORT int frcnn_head_targets(int variant, const float *rois, const int32_t *n_rois_dev, int64_t P_cap, const float *gt,
                                    const int64_t *gt_label, int64_t G, int64_t label_offset, int64_t max_pos, int64_t total,
                                    const int64_t *perm_pos, int64_t n_perm_pos, const int64_t *perm_neg, int64_t n_perm_neg,
                                    uint64_t seed, uint64_t offset, int64_t *out_cls, float *out_reg, float *out_rois,
                                    int64_t *out_keep_index, int32_t *out_counts, void *workspace, size_t workspace_bytes, void *stream)
{
    (void)workspace; (void)workspace_bytes;
    FRCNN_REQUIRE(variant == 0 || variant == 1, "head_targets: variant must be 0 (VGG) or 1 (FPN)");
    FRCNN_REQUIRE(P_cap >= 0 && G > 0, "head_targets: need P_cap >= 0 and G >= 1");
    if (P_cap + G > HT_MAX) return frcnn_set_error(FRCNN_ERR_UNSUPPORTED, "head_targets: P+G=%lld above limit %d", (long long)(P_cap + G), HT_MAX);
    FRCNN_REQUIRE(total > 0 && total <= HT_ROWS_MAX && max_pos >= 0 && max_pos <= total, "head_targets: need 0 < total <= %d and 0 <= max_pos <= total", HT_ROWS_MAX);
    FRCNN_REQUIRE((P_cap == 0 || rois) && gt && gt_label && out_cls && out_reg && out_rois && out_counts, "head_targets: NULL pointer");
    FRCNN_REQUIRE(n_perm_pos >= 0 && n_perm_neg >= 0, "head_targets: bad perm length");
    hipStream_t s = (hipStream_t)stream;
    FRCNN_LAUNCH(KID_HEAD_TARGETS, head_targets_kernel, dim3(1), dim3(1024), 0, s, variant, (const float4 *)rois, n_rois_dev, (int)P_cap,
                 (const float4 *)gt, gt_label, (int)G, (int)label_offset, (int)max_pos, (int)total, perm_pos, (int)n_perm_pos, perm_neg,
                 (int)n_perm_neg, (unsigned long long)seed, (unsigned long long)offset, out_cls, (float4 *)out_reg, (float4 *)out_rois,
                 out_keep_index, out_counts);
    FRCNN_CHECK_LAUNCH("head_targets_kernel");
    return FRCNN_OK;
}
