// frcnn_internal.h -- structures and launchers shared between the .hip files.
#pragma once
#include "frcnn_common.h"

#define FRCNN_MAX_LEVELS 8
#define FRCNN_MAX_BASE 16

// Level table of an anchor grid, passed to kernels by value (kernarg).
struct AnchorDesc {
    int n_levels, A;
    int fh[FRCNN_MAX_LEVELS], fw[FRCNN_MAX_LEVELS], sh[FRCNN_MAX_LEVELS], sw[FRCNN_MAX_LEVELS];
    int64_t off[FRCNN_MAX_LEVELS];
    float base[FRCNN_MAX_LEVELS][FRCNN_MAX_BASE][4];
    float div_w, div_h;
};

int frcnn_fill_anchor_desc(AnchorDesc *d, int n_levels, const int *fh, const int *fw, const int *sh, const int *sw,
                           const float *base, int A, float div_w, float div_h, int64_t *n_total);

// sample_ctl != nullptr: S / 64 extra workgroups of the same launch sample the top-k stage's splitters (topk_dev.h) for pre_k
int frcnn_launch_prologue(const float *reg, const float *cls, const float *anchors, const AnchorDesc *d, int64_t N,
                          float min_size, float *out_boxes, float *out_scores, int32_t *ctrl_zero, int n_ctrl, int32_t *zero2, int n_zero2,
                          void *sample_ctl, int64_t pre_k, hipStream_t s);

// top-K: count must be zero before topk_scatter runs; zero_count = true lets the rank kernel clear it.
size_t frcnn_ws_topk(int64_t N);
// sampled: the splitters are already in the workspace's control block (frcnn_topk_sample_ctl(ws, N); nullptr below the sample sort's size)
void *frcnn_topk_sample_ctl(void *ws, int64_t N);
int frcnn_launch_topk(const float *scores, const float *boxes_in, int64_t N, int64_t K, int proposal_mode,
                      int64_t *out_idx, float *out_scores, float *out_boxes, int32_t *out_count,
                      void *ws, size_t ws_bytes, bool sampled, hipStream_t s);

size_t frcnn_ws_nms(int64_t K);
// pre_zeroed: the caller has already cleared the region frcnn_nms_zero_region() describes (e.g. inside an earlier kernel of the
// same stream); otherwise frcnn_launch_nms clears it with a memset node.
int frcnn_launch_nms(const float *boxes, const int32_t *cls, const int32_t *n_boxes_dev, int64_t K, float thr, int64_t post_k,
                     int64_t *out_keep, float *out_rois, const int64_t *src_map, int64_t *out_src, int32_t *out_count,
                     void *ws, size_t ws_bytes, bool pre_zeroed, hipStream_t s);
void frcnn_nms_zero_region(void *ws, int64_t K, int32_t **ptr, int *n_ints);
