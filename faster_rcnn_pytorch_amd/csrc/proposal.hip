// proposal.hip -- RegionProposal.forward (models/model.py:12-58; FPN: models/new_model.py:49-86) as ONE
// C call that never touches the host.  Launches at 600x1000 (N = 20 646): proposal_prologue (decode + scores, with the sample sort's
// splitter sampling in its first workgroups) -> topk_partition (count + place behind a grid barrier) -> topk_bucket (ranks, gathers the
// boxes) -> nms_kernel (relation + resolution + outputs); at FPN size the bucket ranking rides in the partition launch: three launches.
// The reference's version is ~20 eager launches, a full torch.sort, three boolean-index host syncs and torchvision's NMS (device mask ->
// host scan).  Counts stay on the device (the top-k count feeds the NMS kernel through a device int32), so the step is
// graph-capturable; the prologue also clears the NMS stage's zero region: no memset node.
#include "frcnn_common.h"
#include "frcnn_internal.h"
#include "frcnn_layout.h"
FRCNN_LAYOUT_STAMP(proposal);

struct ProposalWs {
    float *boxes, *scores, *sscores, *sboxes;
    int64_t *sidx, *keep;
    int32_t *ctrl, *lvl;
    void *topk_ws, *nms_ws;
    size_t topk_bytes, nms_bytes, total;
};

static ProposalWs carve(void *ws, int64_t N, int64_t K, int64_t P)
{
    ProposalWs w;
    char *p = (char *)ws;
    size_t o = 0;
    auto take = [&](size_t bytes) { void *r = p ? p + o : nullptr; o += align_up(bytes, 256); return r; };
    w.boxes = (float *)take((size_t)N * 16);
    w.scores = (float *)take((size_t)N * 4);
    w.sidx = (int64_t *)take((size_t)K * 8);
    w.sscores = (float *)take((size_t)K * 4);
    w.sboxes = (float *)take((size_t)K * 16);
    w.keep = (int64_t *)take((size_t)P * 8);
    w.ctrl = (int32_t *)take(256);
    w.lvl = (int32_t *)take((size_t)K * 4);
    w.topk_bytes = frcnn_ws_topk(N);
    w.topk_ws = take(w.topk_bytes);
    w.nms_bytes = frcnn_ws_nms(K);
    w.nms_ws = take(w.nms_bytes);
    w.total = o;
    return w;
}

size_t frcnn_ws_region_proposal(int64_t N, int64_t K, int64_t P) { return carve(nullptr, N, K, P).total; }

// per-level NMS option: the level id of every sorted box from its anchor index (level l owns anchors [off[l], off[l + 1]))
struct LevelOffsets { int n; int64_t off[FRCNN_MAX_LEVELS + 1]; };
__global__ __launch_bounds__(256) void level_ids_kernel(const int64_t *__restrict__ sidx, int K, LevelOffsets lo, int32_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const int64_t a = sidx[i];                 // rows past the live count hold stale indices: their level is never read (nms: me < n)
    int l = 0;
    for (int q = 1; q < lo.n; ++q) l += a >= lo.off[q];
    out[i] = l;
}

FRCNN_EXPORT int frcnn_region_proposal(const float *reg, const float *cls, const float *anchors, int64_t N, int fh, int fw, int stride,
                                       const float *base_host, int A, float div_w, float div_h, float min_size_norm,
                                       int64_t pre_nms_top_k, float iou_threshold, int64_t post_nms_top_k,
                                       const int64_t *nms_level_offsets_host, int n_nms_levels, float *out_rois,
                                       int32_t *out_count, int64_t *out_src_idx, void *workspace, size_t workspace_bytes, void *stream)
{
    FRCNN_REQUIRE(!nms_level_offsets_host || (n_nms_levels >= 1 && n_nms_levels <= FRCNN_MAX_LEVELS), "region_proposal: 1 <= n_nms_levels <= %d", FRCNN_MAX_LEVELS);
    FRCNN_REQUIRE(N > 0 && pre_nms_top_k > 0 && post_nms_top_k > 0, "region_proposal: sizes must be positive");
    FRCNN_REQUIRE(reg && cls && out_rois && out_count && workspace, "region_proposal: NULL pointer");
    FRCNN_REQUIRE(N < ((int64_t)1 << 22), "region_proposal: N=%lld above the rank-sort limit", (long long)N);
    const int64_t K = pre_nms_top_k < N ? pre_nms_top_k : N;
    const int64_t P = post_nms_top_k;
    ProposalWs w = carve(workspace, N, K, K);
    if (workspace_bytes < w.total)
        return frcnn_set_error(FRCNN_ERR_WORKSPACE, "region_proposal: workspace %zu < %zu bytes", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    AnchorDesc d;
    if (!anchors) {
        FRCNN_REQUIRE(base_host, "region_proposal: neither anchors nor a grid description given");
        int64_t n = 0;
        int rc = frcnn_fill_anchor_desc(&d, 1, &fh, &fw, &stride, &stride, base_host, A, div_w, div_h, &n);
        if (rc) return rc;
        FRCNN_REQUIRE(n == N, "region_proposal: grid describes %lld anchors, N=%lld", (long long)n, (long long)N);
    }
    int32_t *nz_ptr = nullptr;
    int nz_n = 0;
    frcnn_nms_zero_region(w.nms_ws, K, &nz_ptr, &nz_n);            // cleared by the prologue kernel: no memset node in the pipeline
    void *sample_ctl = frcnn_topk_sample_ctl(w.topk_ws, N);        // the sample sort's splitters are drawn inside the prologue launch
    int rc = frcnn_launch_prologue(reg, cls, anchors, &d, N, min_size_norm, w.boxes, w.scores, w.ctrl, 8, nz_ptr, nz_n, sample_ctl, K, s);
    if (rc) return rc;
    rc = frcnn_launch_topk(w.scores, w.boxes, N, K, 1, w.sidx, w.sscores, w.sboxes, w.ctrl, w.topk_ws, w.topk_bytes, sample_ctl != nullptr, s);
    if (rc) return rc;
    const int32_t *lvl = nullptr;
    if (nms_level_offsets_host) {                                  // optional per-level NMS (not the reference's behaviour)
        LevelOffsets lo;
        lo.n = n_nms_levels;
        for (int l = 0; l <= n_nms_levels; ++l) lo.off[l] = nms_level_offsets_host[l];
        FRCNN_LAUNCH(level_ids_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, s, w.sidx, (int)K, lo, w.lvl);
        FRCNN_CHECK_LAUNCH("level_ids_kernel");
        lvl = w.lvl;
    }
    return frcnn_launch_nms(w.sboxes, lvl, w.ctrl, K, iou_threshold, P < K ? P : K, w.keep, out_rois, w.sidx, out_src_idx, out_count, w.nms_ws,
                            w.nms_bytes, true, s);
}
