"""Host-side operator layer: the reference's box/anchor functions and the five torchvision entry
points it calls, re-implemented on libfrcnn_hip.so (hand-written gfx950 kernels behind a C ABI).

Mirrors (file:line under the reference):
  utils/util.py:15-102   cxcy_to_xy, xy_to_cxcy, encode, decode, find_jaccard_overlap
  util/box_ops.py:24-37  box_iou
  torchvision.ops.nms            as called at models/model.py:53,394
  torchvision.ops.RoIPool        as called at models/model.py:97,113
  torchvision.ops.MultiScaleRoIAlign   as called at models/new_model.py:127,143
  torchvision AnchorGenerator    as called at models/new_model.py:23-25,46

PyTorch is plumbing here (device memory, streams, autograd glue).  Every op requires contiguous
fp32 tensors on a HIP device and raises otherwise: there is no CPU path in the product.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


# --------------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------------
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the frcnn ops run only on a HIP device (no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


_WS = {}


def _workspace(device, nbytes):
    """Grow-only per-(device, stream) scratch buffer owned by the caller side (torch allocator)."""
    if torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture the scratch is an ordinary allocation of the graph's own pool (it lives and dies with the graph).  A cached
        # buffer keyed by the capture stream was first allocated inside the FIRST captured configuration's pool and then baked into the graphs of
        # a later one, after that pool had been released: a memory fault on replay (bench.py, a graph run of one model followed by one of another)
        return torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


_CTRL = {}
_CTRL_IN_GRAPHS = {}          # key -> buffers whose address a captured graph holds (kept alive when the cache moves on to a larger one)


def _ctrl_workspace(device, tag, nbytes):
    """A persistent, ZERO-INITIALISED workspace per (op, device) for the ops whose kernels keep a control word between calls (a
    last-workgroup ticket that the kernel itself resets: include/frcnn_hip.h).  Never shared with the scratch of other ops; one per
    device, not per stream (calls of one op on one device are stream-ordered in the training step), so that a HIP-graph capture on its
    own capture stream finds the buffer the warm-up created instead of allocating -- and zero-filling -- a new one inside the graph.

    Under capture (the conv stage keeps U / V / M here: hundreds of MB sized by the largest layer seen) a missing or too small buffer is
    NOT cached: torch.zeros would allocate it inside the capturing graph's private pool, and the cache would hand that address to eager
    calls and to later graphs after the pool is gone (the fault ops._workspace had in round 4).  The graph gets a zeroed buffer of its own
    pool instead (its memset node replays with it; the kernels leave the control words zero anyway).  And a cached buffer that a capture
    has seen stays alive when the cache later grows past it."""
    key = (tag, device.index)
    buf = _CTRL.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if buf is None or buf.numel() < nbytes:
        if capturing:
            return torch.zeros(int(nbytes), dtype=torch.uint8, device=device)     # graph-private: lives and dies with the graph's pool
        buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)          # (a buffer a graph replays into stays referenced by _CTRL_IN_GRAPHS)
        _CTRL[key] = buf
    if capturing and not any(b is buf for b in _CTRL_IN_GRAPHS.setdefault(key, [])):
        _CTRL_IN_GRAPHS[key].append(buf)
    return buf


_CONST = {}


def const_tensor(values, device, dtype=torch.float32):
    """A small constant on the device, created once per (values, device).  torch.tensor(list, device='cuda') inside a training step
    is a BLOCKING host-to-device copy: the host waits for everything already enqueued on the stream (the whole backbone) before it
    can go on enqueueing -- 2.5 ms of lost run-ahead per FPN step."""
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _CONST.get(key)
    if t is None:
        t = torch.tensor(list(key[0]), dtype=dtype, device=device)
        _CONST[key] = t
    return t


def _host_i32(v):
    return np.ascontiguousarray(v, dtype=np.int32)


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------------
# anchors
# --------------------------------------------------------------------------------------------
def anchor_base(base_size=16, ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32)):
    """FRCNNAnchorMaker.generate_anchor_base (anchor.py:15-32) -> np.float32 [len(r)*len(s), 4]."""
    r = np.ascontiguousarray(ratios, dtype=np.float64)
    s = np.ascontiguousarray(anchor_scales, dtype=np.float64)
    out = np.empty((len(r) * len(s), 4), np.float32)
    check(lib.frcnn_anchor_base_host(base_size, _np_ptr(r), len(r), _np_ptr(s), len(s), _np_ptr(out)), "anchor_base")
    return out


def tv_base_anchors(size, ratios=(0.5, 1.0, 2.0)):
    r = np.ascontiguousarray(ratios, dtype=np.float32)
    out = np.empty((len(r), 4), np.float32)
    check(lib.frcnn_tv_base_anchors_host(float(size), _np_ptr(r), len(r), _np_ptr(out)), "tv_base_anchors")
    return out


def anchor_grid(feat_shapes, strides, base, div_w, div_h, device):
    """Device anchor grid over levels.  feat_shapes [(fh,fw)], strides [(sh,sw)], base [L,A,4] (host)."""
    fh = _host_i32([s[0] for s in feat_shapes])
    fw = _host_i32([s[1] for s in feat_shapes])
    sh = _host_i32([s[0] for s in strides])
    sw = _host_i32([s[1] for s in strides])
    base = np.ascontiguousarray(base, dtype=np.float32).reshape(len(fh), -1, 4)
    A = base.shape[1]
    N = int((fh.astype(np.int64) * fw).sum() * A)
    out = torch.empty((N, 4), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        check(lib.frcnn_anchor_grid(len(fh), _np_ptr(fh), _np_ptr(fw), _np_ptr(sh), _np_ptr(sw), _np_ptr(base), A,
                                    float(div_w), float(div_h), _ptr(out), N, _stream()), "anchor_grid")
    return out


# --------------------------------------------------------------------------------------------
# box codec + IoU (utils/util.py)
# --------------------------------------------------------------------------------------------
def _codec(op, a, b=None):
    a = _req(a, name="boxes")
    shape = a.shape
    a2 = a.reshape(-1, 4)
    b2 = None
    if b is not None:
        b2 = _req(b, name="anchors").reshape(-1, 4)
        if b2.shape != a2.shape:
            raise ValueError("box codec: operand shapes differ: %s vs %s" % (tuple(a2.shape), tuple(b2.shape)))
    out = torch.empty_like(a2)
    with torch.cuda.device(a.device):
        check(lib.frcnn_box_codec(op, _ptr(a2), _ptr(b2), a2.shape[0], _ptr(out), _stream()), "box_codec")
    return out.reshape(shape)


def xy_to_cxcy(xy):
    return _codec(0, xy)


def cxcy_to_xy(cxcy):
    return _codec(1, cxcy)


def decode(tcxcy, center_anchor):
    return _codec(2, tcxcy, center_anchor)


def encode(gt_cxywh, anc_cxywh):
    return _codec(3, gt_cxywh, anc_cxywh)


def _pairwise(s1, s2, eps):
    s1 = _req(s1, name="set_1")
    s2 = _req(s2, name="set_2")
    out = torch.empty((s1.shape[0], s2.shape[0]), dtype=torch.float32, device=s1.device)
    with torch.cuda.device(s1.device):
        check(lib.frcnn_pairwise_iou(_ptr(s1), s1.shape[0], _ptr(s2), s2.shape[0], float(eps), _ptr(out), _stream()), "pairwise_iou")
    return out


def find_jaccard_overlap(set_1, set_2, eps=1e-5):
    return _pairwise(set_1, set_2, eps)


def box_iou(boxes1, boxes2):
    """util/box_ops.py:24-37 -- returns (iou, union) like the reference; union is recomputed in torch."""
    iou = _pairwise(boxes1, boxes2, 0.0)
    a1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    a2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    union = a1[:, None] + a2 - wh[..., 0] * wh[..., 1]
    return iou, union


def box_area(boxes):
    """torchvision.ops.boxes.box_area as imported at util/box_ops.py:6: (x2 - x1) * (y2 - y1)."""
    boxes = _req(boxes, name="boxes")
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_cxcywh_to_xyxy(x):
    """util/box_ops.py:9-13 (same arithmetic as utils/util.py cxcy_to_xy)."""
    return cxcy_to_xy(x)


def box_xyxy_to_cxcywh(x):
    """util/box_ops.py:16-20 (same arithmetic as utils/util.py xy_to_cxcy)."""
    return xy_to_cxcy(x)


# --------------------------------------------------------------------------------------------
# proposal stage (models/model.py:12-58)
# --------------------------------------------------------------------------------------------
def proposal_prologue(reg, cls, anchors, min_size_norm):
    reg = _req(reg, name="reg").reshape(-1, 4)
    cls = _req(cls, name="cls").reshape(-1, 2)
    anchors = _req(anchors, name="anchors").reshape(-1, 4)
    N = reg.shape[0]
    if cls.shape[0] != N or anchors.shape[0] != N:
        raise ValueError("prologue: N mismatch")
    boxes = torch.empty((N, 4), dtype=torch.float32, device=reg.device)
    scores = torch.empty((N,), dtype=torch.float32, device=reg.device)
    with torch.cuda.device(reg.device):
        check(lib.frcnn_proposal_prologue(_ptr(reg), _ptr(cls), _ptr(anchors), N, float(min_size_norm), _ptr(boxes), _ptr(scores),
                                          _stream()), "proposal_prologue")
    return boxes, scores


def topk_sorted(scores, K, boxes=None, all_live=False):
    """Returns (idx[K] i64, sorted_scores[K], sorted_boxes[K,4] | None, count (device int32 tensor))."""
    scores = _req(scores, name="scores").reshape(-1)
    N = scores.shape[0]
    K = int(min(K, N))
    dev = scores.device
    boxes = None if boxes is None else _req(boxes, name="boxes").reshape(-1, 4)
    idx = torch.empty((K,), dtype=torch.int64, device=dev)
    ssc = torch.empty((K,), dtype=torch.float32, device=dev)
    sbx = None if boxes is None else torch.empty((K, 4), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    nb = _lib.workspace_bytes(_lib.OP_TOPK, N)
    ws = _workspace(dev, nb)
    with torch.cuda.device(dev):
        if all_live:
            if K != N:
                raise ValueError("all_live sort returns all N entries")
            check(lib.frcnn_argsort_desc(_ptr(scores), _ptr(boxes), N, _ptr(idx), _ptr(ssc), _ptr(sbx), _ptr(cnt), _ptr(ws), nb, _stream()),
                  "argsort_desc")
        else:
            check(lib.frcnn_topk_sorted(_ptr(scores), _ptr(boxes), N, K, _ptr(idx), _ptr(ssc), _ptr(sbx), _ptr(cnt), _ptr(ws), nb, _stream()),
                  "topk_sorted")
    return idx, ssc, sbx, cnt


def nms_sorted(boxes, iou_threshold, post_k=None, n_boxes=None, want_rois=False):
    """NMS over boxes already in visiting order.  No host sync: returns (keep[post_k] i64, rois|None, count int32[1])."""
    boxes = _req(boxes, name="boxes").reshape(-1, 4)
    K = boxes.shape[0]
    post_k = K if post_k is None else int(min(post_k, K))
    dev = boxes.device
    keep = torch.empty((max(post_k, 1),), dtype=torch.int64, device=dev)
    rois = torch.empty((max(post_k, 1), 4), dtype=torch.float32, device=dev) if want_rois else None
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    nb = _lib.workspace_bytes(_lib.OP_NMS, K)
    ws = _workspace(dev, nb)
    if n_boxes is not None:
        n_boxes = _req(n_boxes, torch.int32, "n_boxes")
    with torch.cuda.device(dev):
        check(lib.frcnn_nms(_ptr(boxes), _ptr(n_boxes), K, float(iou_threshold), post_k, _ptr(keep), _ptr(rois), _ptr(cnt), _ptr(ws), nb,
                            _stream()), "nms")
    return keep, rois, cnt


def nms(boxes, scores, iou_threshold):
    """Drop-in for torchvision.ops.nms(boxes, scores, iou_threshold) -> int64 indices of kept boxes,
    sorted by decreasing score (models/model.py:53,394).  Like torchvision it returns a
    variable-length tensor, which costs one host sync (the count)."""
    boxes = _req(boxes, name="boxes").reshape(-1, 4)
    scores = _req(scores, name="scores").reshape(-1)
    if boxes.shape[0] != scores.shape[0]:
        raise ValueError("nms: boxes and scores differ in length")
    if boxes.shape[0] == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    idx, _, sboxes, _ = topk_sorted(scores, boxes.shape[0], boxes, all_live=True)
    keep, _, cnt = nms_sorted(sboxes, iou_threshold)
    n = host_count(cnt)
    return idx[keep[:n]]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms semantics: NMS is performed independently per class id in `idxs`, in ONE launch pair
    (class-aware suppression mask + scan).  Returns int64 indices of the kept boxes sorted by decreasing score.
    Replaces the per-class python loop of FRCNN._suppress (models/model.py:382-402): one host sync instead of C-1."""
    boxes = _req(boxes, name="boxes").reshape(-1, 4)
    scores = _req(scores, name="scores").reshape(-1)
    idxs = _req(idxs.to(torch.int32), torch.int32, "idxs").reshape(-1)
    n = boxes.shape[0]
    if scores.shape[0] != n or idxs.shape[0] != n:
        raise ValueError("batched_nms: boxes, scores and idxs differ in length")
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order, _, sboxes, _ = topk_sorted(scores, n, boxes, all_live=True)
    scls = idxs[order].contiguous()
    dev = boxes.device
    keep = torch.empty((n,), dtype=torch.int64, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    nb = _lib.workspace_bytes(_lib.OP_NMS, n)
    ws = _workspace(dev, nb)
    with torch.cuda.device(dev):
        check(lib.frcnn_nms_classed(_ptr(sboxes), _ptr(scls), None, n, float(iou_threshold), n, _ptr(keep), None, _ptr(cnt), _ptr(ws), nb,
                                    _stream()), "nms_classed")
    return order[keep[:host_count(cnt)]]


def region_proposal(reg, cls, anchors, min_size_norm, pre_nms_top_k, iou_threshold, post_nms_top_k, grid=None, want_src=False,
                    nms_level_offsets=None):
    """RegionProposal.forward in one enqueue (no host sync).
    anchors: [N,4] device tensor, or None with grid=(fh, fw, stride, base[A,4] host, div_w, div_h).
    nms_level_offsets: None = one global class-agnostic NMS (the reference, new_model.py:74-83); a list [0, n_0, n_0 + n_1, ..., N]
    of per-level anchor offsets = the optional per-FPN-level NMS (boxes only compete inside their level; BASELINE configs[3]).
    Returns (rois [P,4] fixed capacity, count int32[1] device, src_idx [P] | None)."""
    reg = _req(reg, name="reg").reshape(-1, 4)
    cls = _req(cls, name="cls").reshape(-1, 2)
    N = reg.shape[0]
    dev = reg.device
    P = int(post_nms_top_k)
    K = int(min(pre_nms_top_k, N))
    rois = torch.zeros((P, 4), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    src = torch.empty((P,), dtype=torch.int64, device=dev) if want_src else None
    nb = _lib.workspace_bytes(_lib.OP_REGION_PROPOSAL, N, K)
    ws = _workspace(dev, nb)
    if anchors is not None:
        anchors = _req(anchors, name="anchors").reshape(-1, 4)
        if anchors.shape[0] != N:
            raise ValueError("region_proposal: anchors/reg length mismatch")
        fh = fw = stride = A = 0
        base_p, dw, dh = None, 1.0, 1.0
    else:
        fh, fw, stride, base, dw, dh = grid
        base = np.ascontiguousarray(base, dtype=np.float32)
        A = base.shape[0]
        base_p = _np_ptr(base)
    lo_p, n_lv = None, 0
    if nms_level_offsets is not None:
        lo = np.ascontiguousarray(nms_level_offsets, dtype=np.int64)
        if lo.ndim != 1 or len(lo) < 2 or lo[0] != 0 or lo[-1] != N or (np.diff(lo) < 0).any():
            raise ValueError("region_proposal: nms_level_offsets must be ascending anchor offsets [0, ..., N]")
        lo_p, n_lv = _np_ptr(lo), len(lo) - 1
    with torch.cuda.device(dev):
        check(lib.frcnn_region_proposal(_ptr(reg), _ptr(cls), _ptr(anchors), N, int(fh), int(fw), int(stride), base_p, int(A),
                                        float(dw), float(dh), float(min_size_norm), K, float(iou_threshold), P, lo_p, n_lv,
                                        _ptr(rois), _ptr(cnt), _ptr(src), _ptr(ws), nb, _stream()), "region_proposal")
    return rois, cnt, src


# --------------------------------------------------------------------------------------------
# RPN head tail (models/model.py:79-83): bias + ReLU + both 1x1 heads + NHWC store on the fp32 matrix cores
# --------------------------------------------------------------------------------------------
class _RPNHeadTailFn(torch.autograd.Function):
    """All levels in one launch.  args: (mfma_bf16, b3, w_cls, b_cls, w_reg, b_reg, *raws); raws are the bias-free 3x3 outputs
    [1, C, fh_l, fw_l], all fp32 or all bf16."""

    @staticmethod
    def forward(ctx, mfma_bf16, b3, w_cls, b_cls, w_reg, b_reg, *raws):
        if not raws:
            raise ValueError("rpn_head_tail: no feature level")
        dt = raws[0].dtype
        if dt not in (torch.float32, torch.bfloat16):
            raise TypeError("rpn_head_tail: conv output must be float32 or bfloat16, got %s" % dt)
        raws = [_req(r, dt, "conv_raw") for r in raws]
        for r in raws:
            if r.dim() != 4 or r.shape[0] != 1 or r.shape[1] != raws[0].shape[1]:
                raise ValueError("rpn_head_tail: every level must be [1,C,fh,fw] (batch 1 per GPU) with the same C")
        Cc = raws[0].shape[1]
        Ps = [r.shape[2] * r.shape[3] for r in raws]
        b3, b_cls, b_reg = _req(b3, name="b3"), _req(b_cls, name="b_cls"), _req(b_reg, name="b_reg")
        wc = _req(w_cls, name="w_cls").reshape(w_cls.shape[0], -1)
        wr = _req(w_reg, name="w_reg").reshape(w_reg.shape[0], -1)
        n_cls, n_reg = wc.shape[0], wr.shape[0]
        if wc.shape[1] != Cc or wr.shape[1] != Cc or b3.numel() != Cc:
            raise ValueError("rpn_head_tail: channel mismatch")
        dev = raws[0].device
        Pt = sum(Ps)
        out_cls = torch.empty((1, Pt * n_cls // 2, 2), dtype=torch.float32, device=dev)
        out_reg = torch.empty((1, Pt * n_reg // 4, 4), dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * len(raws))(*[r.data_ptr() for r in raws])
        pl = (C.c_int64 * len(raws))(*Ps)
        with torch.cuda.device(dev):
            check(lib.frcnn_rpn_head_tail_ml_fwd(ptrs, 1 if dt == torch.bfloat16 else 0, 1 if mfma_bf16 else 0, Cc, pl, len(raws), _ptr(b3),
                                                 _ptr(wc), _ptr(b_cls), n_cls, _ptr(wr), _ptr(b_reg), n_reg, _ptr(out_cls), _ptr(out_reg),
                                                 _stream()), "rpn_head_tail_ml_fwd")
        ctx.save_for_backward(b3, wc, wr, *raws)
        ctx.w_shapes = (w_cls.shape, w_reg.shape)
        return out_cls, out_reg

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        b3, wc, wr = ctx.saved_tensors[:3]
        raws = ctx.saved_tensors[3:]
        n_cls, n_reg = wc.shape[0], wr.shape[0]
        Cc = wc.shape[1]
        dev = wc.device
        if Cc in (256, 512):
            # one fused MFMA kernel + a finalize for all levels (csrc/rpn_head.hip): d_raw, dW, db, db3
            g_cls = _req(g_cls.reshape(-1, n_cls), name="grad_cls")
            g_reg = _req(g_reg.reshape(-1, n_reg), name="grad_reg")
            d_raws = [torch.empty_like(r) for r in raws]
            dwc, dwr = torch.empty_like(wc), torch.empty_like(wr)
            dbc = torch.empty((n_cls,), dtype=torch.float32, device=dev)
            dbr = torch.empty((n_reg,), dtype=torch.float32, device=dev)
            db3 = torch.empty((Cc,), dtype=torch.float32, device=dev)
            nb = _lib.workspace_bytes(_lib.OP_HEAD_BWD, Cc)
            ws = _workspace(dev, nb)
            ptrs = (C.c_void_p * len(raws))(*[r.data_ptr() for r in raws])
            dptrs = (C.c_void_p * len(raws))(*[r.data_ptr() for r in d_raws])
            pl = (C.c_int64 * len(raws))(*[r.shape[2] * r.shape[3] for r in raws])
            with torch.cuda.device(dev):
                check(lib.frcnn_rpn_head_tail_ml_bwd(ptrs, dptrs, 1 if raws[0].dtype == torch.bfloat16 else 0, Cc, pl, len(raws), _ptr(b3), _ptr(wc),
                                                     n_cls, _ptr(wr), n_reg, _ptr(g_cls), _ptr(g_reg), _ptr(dwc), _ptr(dbc), _ptr(dwr), _ptr(dbr),
                                                     _ptr(db3), _ptr(ws), nb, _stream()), "rpn_head_tail_ml_bwd")
            return (None, db3, dwc.reshape(ctx.w_shapes[0]), dbc, dwr.reshape(ctx.w_shapes[1]), dbr, *d_raws)
        raise _lib.FrcnnError("rpn_head_tail backward: the fused kernel is built for the reference's head widths C = 256 (FPN) and 512 (VGG), "
                              "got C = %d (forward-only use is fine)" % Cc)


class _RPNConvHeadFn(torch.autograd.Function):
    """The whole FPN RPN head in the bf16 mixed-precision configuration (csrc/rpn_conv.hip): 3x3 conv + bias + ReLU + both 1x1
    heads for all levels in one MFMA implicit-GEMM launch.  args: (w3, b3, w_cls, b_cls, w_reg, b_reg, *feats bf16 [1,256,h,w]).
    Backward: the fused tail backward (d_raw, head gradients), then the convolution's data gradient (the forward kernel on transposed,
    flipped weights) and weight gradient (split-K MFMA kernel) -- no MIOpen kernel is left in the head."""

    @staticmethod
    def forward(ctx, w3, b3, w_cls, b_cls, w_reg, b_reg, *feats):
        feats = [_req(f, torch.bfloat16, "feature map") for f in feats]
        Cc = feats[0].shape[1]
        for f in feats:
            if f.dim() != 4 or f.shape[0] != 1 or f.shape[1] != Cc:
                raise ValueError("rpn_conv_head: every level must be [1,C,h,w] (batch 1 per GPU) with the same C")
        w3c = _req(w3, name="w3")
        b3, b_cls, b_reg = _req(b3, name="b3"), _req(b_cls, name="b_cls"), _req(b_reg, name="b_reg")
        wc = _req(w_cls, name="w_cls").reshape(w_cls.shape[0], -1)
        wr = _req(w_reg, name="w_reg").reshape(w_reg.shape[0], -1)
        n_cls, n_reg = wc.shape[0], wr.shape[0]
        dev = feats[0].device
        raws = [torch.empty_like(f) for f in feats]
        Pt = sum(f.shape[2] * f.shape[3] for f in feats)
        out_cls = torch.empty((1, Pt * n_cls // 2, 2), dtype=torch.float32, device=dev)
        out_reg = torch.empty((1, Pt * n_reg // 4, 4), dtype=torch.float32, device=dev)
        H = _host_i32([f.shape[2] for f in feats])
        W = _host_i32([f.shape[3] for f in feats])
        fp = (C.c_void_p * len(feats))(*[f.data_ptr() for f in feats])
        rp = (C.c_void_p * len(feats))(*[r.data_ptr() for r in raws])
        nb = _lib.workspace_bytes(_lib.OP_RPN_CONV, 0)
        ws = _workspace(dev, nb)
        with torch.cuda.device(dev):
            check(lib.frcnn_rpn_conv_head_fwd(fp, rp, _np_ptr(H), _np_ptr(W), len(feats), Cc, _ptr(w3c), _ptr(b3), _ptr(wc), _ptr(b_cls), n_cls,
                                              _ptr(wr), _ptr(b_reg), n_reg, _ptr(out_cls), _ptr(out_reg), _ptr(ws), nb, _stream()), "rpn_conv_head_fwd")
        ctx.save_for_backward(w3c, b3, wc, wr, *feats, *raws)
        ctx.n = len(feats)
        ctx.w_shapes = (w_cls.shape, w_reg.shape)
        return out_cls, out_reg

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        w3, b3, wc, wr = ctx.saved_tensors[:4]
        feats = ctx.saved_tensors[4:4 + ctx.n]
        raws = ctx.saved_tensors[4 + ctx.n:]
        n_cls, n_reg = wc.shape[0], wr.shape[0]
        Cc = wc.shape[1]
        dev = wc.device
        g_cls = _req(g_cls.reshape(-1, n_cls), name="grad_cls")
        g_reg = _req(g_reg.reshape(-1, n_reg), name="grad_reg")
        d_raws = [torch.empty_like(r) for r in raws]
        dwc, dwr = torch.empty_like(wc), torch.empty_like(wr)
        dbc = torch.empty((n_cls,), dtype=torch.float32, device=dev)
        dbr = torch.empty((n_reg,), dtype=torch.float32, device=dev)
        db3 = torch.empty((Cc,), dtype=torch.float32, device=dev)
        nb = _lib.workspace_bytes(_lib.OP_HEAD_BWD, Cc)
        ws = _workspace(dev, nb)
        ptrs = (C.c_void_p * len(raws))(*[r.data_ptr() for r in raws])
        dptrs = (C.c_void_p * len(raws))(*[r.data_ptr() for r in d_raws])
        pl = (C.c_int64 * len(raws))(*[r.shape[2] * r.shape[3] for r in raws])
        with torch.cuda.device(dev):
            check(lib.frcnn_rpn_head_tail_ml_bwd(ptrs, dptrs, 1, Cc, pl, len(raws), _ptr(b3), _ptr(wc), n_cls, _ptr(wr), n_reg, _ptr(g_cls), _ptr(g_reg),
                                                 _ptr(dwc), _ptr(dbc), _ptr(dwr), _ptr(dbr), _ptr(db3), _ptr(ws), nb, _stream()), "rpn_head_tail_ml_bwd")
        # data gradient of the 3x3 convolution: the same implicit-GEMM kernel on the transposed, flipped weights, all levels in one launch
        d_feats = [None] * len(feats)
        if any(ctx.needs_input_grad[6:]):
            d_feats = [torch.empty_like(f) for f in feats]
            H = _host_i32([f.shape[2] for f in feats])
            W = _host_i32([f.shape[3] for f in feats])
            fptrs = (C.c_void_p * len(feats))(*[t.data_ptr() for t in d_feats])
            nbc = _lib.workspace_bytes(_lib.OP_RPN_CONV, 0)
            wsc = _workspace(dev, nbc)
            with torch.cuda.device(dev):
                check(lib.frcnn_rpn_conv_bwd_data(dptrs, fptrs, _np_ptr(H), _np_ptr(W), len(feats), Cc, _ptr(w3), _ptr(wsc), nbc, _stream()), "rpn_conv_bwd_data")
        # weight gradient: the hand-written split-K MFMA kernel, all levels in one launch + a fixed-order finalize
        dw3 = rpn_conv_wgrad(feats, d_raws)
        return (dw3, db3, dwc.reshape(ctx.w_shapes[0]), dbc, dwr.reshape(ctx.w_shapes[1]), dbr, *d_feats)


def rpn_conv_bwd_data(d_raws, w3):
    """Data gradient of the shared 3x3 RPN convolution (models/new_model.py:96) for all levels in one launch: d_raws = bf16
    [1,256,h,w] gradients of the bias-free conv output, w3 = the fp32 weight [256,256,3,3]; returns bf16 gradients of the inputs."""
    d_raws = [_req(d, torch.bfloat16, "d_raw") for d in d_raws]
    w3 = _req(w3, name="w3")
    dev = d_raws[0].device
    outs = [torch.empty_like(d) for d in d_raws]
    H = _host_i32([d.shape[2] for d in d_raws])
    W = _host_i32([d.shape[3] for d in d_raws])
    ip = (C.c_void_p * len(d_raws))(*[d.data_ptr() for d in d_raws])
    op = (C.c_void_p * len(d_raws))(*[o.data_ptr() for o in outs])
    nb = _lib.workspace_bytes(_lib.OP_RPN_CONV, 0)
    ws = _workspace(dev, nb)
    with torch.cuda.device(dev):
        check(lib.frcnn_rpn_conv_bwd_data(ip, op, _np_ptr(H), _np_ptr(W), len(d_raws), d_raws[0].shape[1], _ptr(w3), _ptr(ws), nb, _stream()), "rpn_conv_bwd_data")
    return outs


def rpn_conv_wgrad(feats, d_raws):
    """Weight gradient of the shared 3x3 RPN convolution summed over the levels: feats / d_raws = bf16 [1,256,h,w]; returns fp32 [256,256,3,3]."""
    feats = [_req(f, torch.bfloat16, "feature map") for f in feats]
    d_raws = [_req(d, torch.bfloat16, "d_raw") for d in d_raws]
    dev = feats[0].device
    dw = torch.empty((256, 256, 3, 3), dtype=torch.float32, device=dev)
    H = _host_i32([f.shape[2] for f in feats])
    W = _host_i32([f.shape[3] for f in feats])
    fp = (C.c_void_p * len(feats))(*[f.data_ptr() for f in feats])
    dp = (C.c_void_p * len(feats))(*[d.data_ptr() for d in d_raws])
    nb = _lib.workspace_bytes(_lib.OP_RPN_CONV_WGRAD, 0)
    ws = _workspace(dev, nb)
    with torch.cuda.device(dev):
        check(lib.frcnn_rpn_conv_wgrad(fp, dp, _np_ptr(H), _np_ptr(W), len(feats), feats[0].shape[1], _ptr(dw), _ptr(ws), nb, _stream()), "rpn_conv_wgrad")
    return dw


class _Conv3x3Bf16C256Fn(torch.autograd.Function):
    """A plain 256 -> 256 3x3 convolution (stride 1, padding 1, no bias) on bf16 [1,256,h,w] maps with an fp32 [256,256,3,3] weight, from the RPN head's own
    bf16 MFMA kernels (csrc/rpn_conv.hip): the forward IS frcnn_rpn_conv_bwd_data on the transposed, flipped weight (that kernel is the head's implicit-GEMM
    main loop without the head: out[b] = sum_a,t W'[a][b][t] in[a](shifted by the flipped tap) -- with W'[a][b][ky][kx] = W[b][a][2-ky][2-kx] that is conv(x, W));
    backward = frcnn_rpn_conv_bwd_data on W itself and frcnn_rpn_conv_wgrad.  fp32 accumulate, bf16 outputs, fp32 weight gradient.  Used for the FPN's output
    convolutions on its two largest levels under bf16 autocast, where MIOpen took 0.2 ms forward and 0.6 ms backward for the stride-4 level alone."""

    @staticmethod
    def forward(ctx, x, weight):
        x = _req(x, torch.bfloat16, "x")
        weight = _req(weight, name="weight")
        if x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != 256 or tuple(weight.shape) != (256, 256, 3, 3):
            raise ValueError("conv3x3_bf16_c256: x must be bf16 [1,256,h,w] and weight fp32 [256,256,3,3]")
        w_fwd = weight.detach().flip(2, 3).transpose(0, 1).contiguous()
        y = rpn_conv_bwd_data([x], w_fwd)[0]
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = rpn_conv_bwd_data([g], weight.detach())[0] if ctx.needs_input_grad[0] else None
        dw = rpn_conv_wgrad([x], [g]) if ctx.needs_input_grad[1] else None
        return dx, dw


def conv3x3_bf16_c256(x, weight, bias=None):
    """nn.Conv2d(256, 256, 3, padding=1) on one bf16 [1,256,h,w] map through the library's bf16 MFMA kernels (see _Conv3x3Bf16C256Fn); bias added in bf16."""
    y = _Conv3x3Bf16C256Fn.apply(x, weight)
    if bias is not None:
        y = y + bias.to(torch.bfloat16).reshape(1, -1, 1, 1)
    return y


def conv3x3_bf16_c256_supported(x, weight):
    """bf16 autocast, one bf16 [1,256,h,w] HIP map of at least 100 x 168 positions (below that the kernel's 256-channel x 8 x 32-position tiles do not fill the
    chip and the vendor convolution is as fast), an fp32 [256,256,3,3] weight."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 256 and tuple(weight.shape) == (256, 256, 3, 3)
            and weight.dtype == torch.float32 and x.shape[2] * x.shape[3] >= 100 * 168 and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)


# --------------------------------------------------------------------------------------------
# the RPN head's 3x3 convolution in fp32 (models/model.py:68-70,79; models/new_model.py:96-98,109) on the fp32 matrix cores
# --------------------------------------------------------------------------------------------
def conv3x3_f32_products(mode=None):
    """How the fp32 conv stage takes its products (process-wide; csrc/rpn_conv_f32.hip, frcnn_conv3x3_f32_products): "native" = the fp32 matrix instruction (default),
    "split" = the same fp32 operands cut exactly into three bf16 pieces each and six bf16 matrix instructions per 16 k rows with fp32 accumulation -- as
    close to float64 as "native", 1.4-1.5 x the rate.  Returns the previous mode; None only asks."""
    names = ("native", "split")
    if mode is None:
        return names[lib.frcnn_conv3x3_f32_products(-1)]
    if mode not in names:
        raise ValueError("conv3x3_f32_products: mode must be one of %s" % (names,))
    return names[lib.frcnn_conv3x3_f32_products(names.index(mode))]


CONV_TRACE = None        # a list while a caller (bench.py) records which calls the fp32 conv stage gets in one step: dicts kind / Cin / Cout / shapes / mask / bias


def _conv_trace(what, Cin, Cout, H, W, mask=False, bias=False, cached=False, relu_bits=False, pooled=False, urot=False):
    if CONV_TRACE is not None:
        kind = "wgrad" if what.endswith("wgrad") else ("bwd_data" if what.endswith("bwd_data") else "fwd")
        CONV_TRACE.append({"kind": kind, "Cin": int(Cin), "Cout": int(Cout), "shapes": [(int(h), int(w)) for h, w in zip(H, W)], "mask": bool(mask),
                           "bias": bool(bias), "cached": bool(cached), "relu_bits": bool(relu_bits), "pooled": bool(pooled), "urot": bool(urot)})


def _conv_f32_call(fn, what, ins, outs, C_, w_or_dw):
    dev = ins[0].device
    H = _host_i32([t.shape[2] for t in ins])
    W = _host_i32([t.shape[3] for t in ins])
    ip = (C.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
    op = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    nb = int(lib.frcnn_rpn_conv3x3_f32_workspace(_np_ptr(H), _np_ptr(W), len(ins), C_))
    if nb == 0:
        raise _lib.FrcnnError("%s: C = %d is outside what the fp32 conv kernels are built for (a multiple of 128)" % (what, C_))
    ws = _ctrl_workspace(dev, "rpn_conv_f32", nb)     # ticket words zero on first use, left zero by every call; slabs + transposed weights behind them
    with torch.cuda.device(dev):
        check(fn(ip, op, _np_ptr(H), _np_ptr(W), len(ins), C_, _ptr(w_or_dw), _ptr(ws), ws.numel(), _stream()), what)
    _conv_trace(what, C_, C_, H, W)


def _conv_f32_levels(ts, name):
    ts = [_req(t, torch.float32, name) for t in ts]
    if not ts:
        raise ValueError("rpn_conv3x3: no feature level")
    Cc = ts[0].shape[1]
    for t in ts:
        if t.dim() != 4 or t.shape[0] != 1 or t.shape[1] != Cc:
            raise ValueError("rpn_conv3x3: every level must be [1,C,h,w] (batch 1 per GPU) with the same C")
    return ts, Cc


def rpn_conv3x3_fwd(feats, w3):
    """Bias-free 3x3 convolution (padding 1) of every level with the shared weight w3 [C,C,3,3]: fp32 [1,C,h,w] -> fp32 [1,C,h,w]."""
    feats, Cc = _conv_f32_levels(feats, "feature map")
    w3 = _req(w3, name="w3")
    if tuple(w3.shape) != (Cc, Cc, 3, 3):
        raise ValueError("rpn_conv3x3: weight must be [C,C,3,3] with C = %d" % Cc)
    outs = [torch.empty_like(f) for f in feats]
    _conv_f32_call(lib.frcnn_rpn_conv3x3_f32_fwd, "rpn_conv3x3_f32_fwd", feats, outs, Cc, w3)
    return outs


def rpn_conv3x3_bwd_data(d_outs, w3):
    """Gradient of rpn_conv3x3_fwd with respect to its inputs (the same kernel on the transposed, flipped weights)."""
    d_outs, Cc = _conv_f32_levels(d_outs, "d_out")
    w3 = _req(w3, name="w3")
    outs = [torch.empty_like(d) for d in d_outs]
    _conv_f32_call(lib.frcnn_rpn_conv3x3_f32_bwd_data, "rpn_conv3x3_f32_bwd_data", d_outs, outs, Cc, w3)
    return outs


def rpn_conv3x3_wgrad(feats, d_outs):
    """Gradient of rpn_conv3x3_fwd with respect to w3, summed over the levels: fp32 [C,C,3,3]."""
    feats, Cc = _conv_f32_levels(feats, "feature map")
    d_outs, _ = _conv_f32_levels(d_outs, "d_out")
    if len(d_outs) != len(feats) or any(a.shape != b.shape for a, b in zip(feats, d_outs)):
        raise ValueError("rpn_conv3x3_wgrad: feats and d_outs differ in shape")
    dw = torch.empty((Cc, Cc, 3, 3), dtype=torch.float32, device=feats[0].device)
    _conv_f32_call(lib.frcnn_rpn_conv3x3_f32_wgrad, "rpn_conv3x3_f32_wgrad", feats, d_outs, Cc, dw)
    return dw


class _RPNConv3x3F32Fn(torch.autograd.Function):
    """args: (w3, *feats) -> the bias-free conv outputs of all levels (one launch); backward = data gradient + weight gradient kernels."""

    @staticmethod
    def forward(ctx, w3, *feats):
        outs = rpn_conv3x3_fwd(feats, w3)
        ctx.save_for_backward(w3, *feats)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        w3 = ctx.saved_tensors[0]
        feats = ctx.saved_tensors[1:]
        g = [x.contiguous() for x in g]
        d_feats = rpn_conv3x3_bwd_data(g, w3) if any(ctx.needs_input_grad[1:]) else [None] * len(feats)
        dw = rpn_conv3x3_wgrad(feats, g) if ctx.needs_input_grad[0] else None
        return (dw, *d_feats)


def rpn_conv3x3_supported(feats, w3):
    """Whether rpn_conv3x3 takes these levels (C a multiple of 128, sizes within the stage's control block): else the caller keeps torch's conv2d."""
    C_ = int(w3.shape[0])
    if C_ % 128 != 0 or tuple(w3.shape) != (C_, C_, 3, 3) or len(feats) > 5:
        return False
    Hh, Wh = _host_i32([f.shape[2] for f in feats]), _host_i32([f.shape[3] for f in feats])
    return bool(lib.frcnn_conv3x3_f32_supported(_np_ptr(Hh), _np_ptr(Wh), len(feats), C_, C_, 1))


def rpn_conv3x3(feats, w3):
    """`inter_layer` without its bias for a list of fp32 levels [1,C,h,w]; differentiable (hand-written forward / backward kernels)."""
    return list(_RPNConv3x3F32Fn.apply(w3, *feats))


# ---- the same stage for the backbone's 3x3 convolutions (Cin != Cout, bias + ReLU in the output transform, ReLU's backward in the input transforms)
def _conv3x3_call(fn, what, H, W, n, Cin, Cout, dev, args_of, mask=False, bias=False, cached=False, relu_bits=False, pooled=False, urot=False):
    _conv_trace(what, Cin, Cout, H, W, mask, bias, cached, relu_bits, pooled, urot)
    Hh, Wh = _host_i32(H), _host_i32(W)
    nb = int(lib.frcnn_conv3x3_f32_workspace(_np_ptr(Hh), _np_ptr(Wh), n, Cin, Cout))
    if nb == 0:
        raise _lib.FrcnnError("%s: Cin = %d / Cout = %d is outside what the fp32 conv stage is built for (multiples of 32)" % (what, Cin, Cout))
    ws = _ctrl_workspace(dev, "rpn_conv_f32", nb)     # ONE block for every fp32 3x3 of the step, sized by the largest layer seen
    with torch.cuda.device(dev):
        check(fn(*args_of(_np_ptr(Hh), _np_ptr(Wh), _ptr(ws), ws.numel(), _stream())), what)


def _ptr_list(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _conv3x3_levels(ts, Cc, name):
    ts = [_req(t, torch.float32, name) for t in ts]
    if not ts:
        raise ValueError("conv3x3: no level")
    for t in ts:
        if t.dim() != 4 or t.shape[0] != 1 or t.shape[1] != Cc:
            raise ValueError("conv3x3: every %s level must be [1,%d,h,w] (batch 1 per GPU)" % (name, Cc))
    return ts


def conv3x3_fwd(xs, w, bias=None, relu=False, keep_transformed=False, want_bits=False, pool=False, u_rotated=None):
    """y_l = act(bias + conv3x3(x_l, w)), padding 1, for a list of fp32 levels [1,Cin,h,w] sharing w [Cout,Cin,3,3] (frcnn_conv3x3_f32_fwd).
    keep_transformed: also return the transformed activations (a flat fp32 tensor) for conv3x3_wgrad(..., x_transformed=).
    want_bits (with relu): also return the ReLU's sign pattern, one int16 word per (channel, output tile), for the gradient calls' relu_bits=.
    pool (with relu): max_pool2d(2, 2) in the same pass -- ys are [1,Cout,h//2,w//2] and the words describe the pooling windows.
    u_rotated: a buffer from conv3x3_u_buffer() that also receives the data gradient's weight transform (for conv3x3_bwd_data(..., u_rotated=)).
    Returns ys, or (ys, x_transformed | None, relu_bits | None) when either extra is asked for."""
    w = _req(w, name="w")
    Cout, Cin = int(w.shape[0]), int(w.shape[1])
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("conv3x3: weight must be [Cout,Cin,3,3]")
    xs = _conv3x3_levels(xs, Cin, "input")
    if bias is not None:
        bias = _req(bias, name="bias")
        if tuple(bias.shape) != (Cout,):
            raise ValueError("conv3x3: bias must be [Cout]")
    if pool and not relu:
        raise ValueError("conv3x3_fwd: pool needs relu=True")
    ys = [torch.empty((1, Cout, x.shape[2] // 2, x.shape[3] // 2) if pool else (1, Cout, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device) for x in xs]
    xp, yp = _ptr_list(xs), _ptr_list(ys)
    xt = bits = None
    Hh, Wh = _host_i32([x.shape[2] for x in xs]), _host_i32([x.shape[3] for x in xs])
    if keep_transformed:
        xt = torch.empty((int(lib.frcnn_conv3x3_f32_xt_floats(_np_ptr(Hh), _np_ptr(Wh), len(xs), Cin)),), dtype=torch.float32, device=xs[0].device)
    if want_bits:
        if not relu:
            raise ValueError("conv3x3_fwd: want_bits needs relu=True")
        bits = torch.empty((int(lib.frcnn_conv3x3_f32_relu_bits_words(_np_ptr(Hh), _np_ptr(Wh), len(xs), Cout)),), dtype=torch.int16, device=xs[0].device)
    _conv3x3_call(lib.frcnn_conv3x3_f32_fwd, "conv3x3_f32_fwd", [x.shape[2] for x in xs], [x.shape[3] for x in xs], len(xs), Cin, Cout, xs[0].device,
                  lambda H, W, ws, nws, st: (xp, yp, H, W, len(xs), Cin, Cout, _ptr(w), _ptr(bias), (2 if pool else 1) if relu else 0, _ptr(bits), _ptr(xt), _ptr(u_rotated), ws, nws, st),
                  bias=bias is not None, relu_bits=bits is not None, pooled=bool(pool), urot=u_rotated is not None)
    return (ys, xt, bits) if (keep_transformed or want_bits) else ys


def conv3x3_dy_buffer(hw, Cout, device):
    """An uninitialised buffer for the weight-gradient transform of an output gradient over maps of sizes hw = [(h, w), ...] (the convolution's own sizes)."""
    Hh, Wh = _host_i32([h for h, _ in hw]), _host_i32([w_ for _, w_ in hw])
    return torch.empty((int(lib.frcnn_conv3x3_f32_xt_floats(_np_ptr(Hh), _np_ptr(Wh), len(hw), int(Cout))),), dtype=torch.float32, device=device)


def conv3x3_u_buffer(xs, w):
    """An uninitialised buffer for the rotated weight transform of these levels (frcnn_conv3x3_f32_u_floats)."""
    Hh, Wh = _host_i32([x.shape[2] for x in xs]), _host_i32([x.shape[3] for x in xs])
    n = int(lib.frcnn_conv3x3_f32_u_floats(_np_ptr(Hh), _np_ptr(Wh), len(xs), int(w.shape[1]), int(w.shape[0])))
    return torch.empty((n,), dtype=torch.float32, device=w.device)


def conv3x3_bwd_data(dys, w, relu_bits=None, pooled_from=None, u_rotated=None, dy_transformed=None, want_bias_partials=False):
    """Input gradient of conv3x3_fwd; relu_bits = the forward's words (the gradient counts where the ReLU output was > 0) or None.
    pooled_from = [(h, w), ...]: the forward ran with pool=True on maps of that size; dys are at the pooled resolution.
    dy_transformed: a buffer from conv3x3_dy_buffer() that also receives the gradient's weight-gradient transform, for a conv3x3_wgrad(...,
    dy_transformed=) called next (want_bias_partials: that call will also want the bias gradient)."""
    w = _req(w, name="w")
    Cout, Cin = int(w.shape[0]), int(w.shape[1])
    dys = _conv3x3_levels(dys, Cout, "d_out")
    hw = _conv3x3_full_sizes(dys, pooled_from)
    relu_bits = _conv3x3_bits(relu_bits, hw, Cout)
    dxs = [torch.empty((1, Cin, h, w_), dtype=torch.float32, device=d.device) for d, (h, w_) in zip(dys, hw)]
    gp, xp = _ptr_list(dys), _ptr_list(dxs)
    _conv3x3_call(lib.frcnn_conv3x3_f32_bwd_data, "conv3x3_f32_bwd_data", [h for h, _ in hw], [w_ for _, w_ in hw], len(dys), Cin, Cout,
                  dys[0].device, lambda H, W, ws, nws, st: (gp, _ptr(relu_bits), xp, H, W, len(dys), Cin, Cout, _ptr(w), _ptr(u_rotated), 1 if pooled_from else 0, _ptr(dy_transformed),
                                            1 if want_bias_partials else 0, ws, nws, st),
                  mask=relu_bits is not None, pooled=bool(pooled_from), urot=u_rotated is not None, cached=dy_transformed is not None)
    return dxs


def _conv3x3_full_sizes(dys, pooled_from):
    """The convolution's own output sizes: those of dys, or -- for a pooled gradient -- the ones given (checked against dys)."""
    if not pooled_from:
        return [(int(d.shape[2]), int(d.shape[3])) for d in dys]
    hw = [(int(h), int(w_)) for h, w_ in pooled_from]
    if len(hw) != len(dys) or any((h // 2, w_ // 2) != (int(d.shape[2]), int(d.shape[3])) for d, (h, w_) in zip(dys, hw)):
        raise ValueError("conv3x3: pooled gradient shapes do not match pooled_from")
    return hw


def _conv3x3_bits(relu_bits, hw, Cout):
    if relu_bits is None:
        return None
    relu_bits = _req(relu_bits, torch.int16, "relu_bits")
    Hh, Wh = _host_i32([h for h, _ in hw]), _host_i32([w_ for _, w_ in hw])
    if relu_bits.numel() != int(lib.frcnn_conv3x3_f32_relu_bits_words(_np_ptr(Hh), _np_ptr(Wh), len(hw), Cout)):
        raise ValueError("conv3x3: relu_bits does not belong to these shapes")
    return relu_bits


def conv3x3_wgrad(xs, dys, relu_bits=None, want_bias=False, x_transformed=None, pooled=False, dy_transformed=None):
    """(dw [Cout,Cin,3,3], dbias [Cout] | None) of conv3x3_fwd, summed over the levels.  x_transformed: what conv3x3_fwd(..., keep_transformed=True)
    returned for the same xs (the activations are then not transformed again)."""
    Cin, Cout = int(xs[0].shape[1]), int(dys[0].shape[1])
    xs = _conv3x3_levels(xs, Cin, "input")
    dys = _conv3x3_levels(dys, Cout, "d_out")
    if len(xs) != len(dys) or any(((a.shape[2] // 2, a.shape[3] // 2) if pooled else tuple(a.shape[2:])) != tuple(b.shape[2:]) for a, b in zip(xs, dys)):
        raise ValueError("conv3x3_wgrad: inputs and d_out differ in shape")
    relu_bits = _conv3x3_bits(relu_bits, [(int(x.shape[2]), int(x.shape[3])) for x in xs], Cout)
    dev = xs[0].device
    dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=dev)
    db = torch.empty((Cout,), dtype=torch.float32, device=dev) if want_bias else None
    xp, gp = _ptr_list(xs), _ptr_list(dys)
    _conv3x3_call(lib.frcnn_conv3x3_f32_wgrad, "conv3x3_f32_wgrad", [x.shape[2] for x in xs], [x.shape[3] for x in xs], len(xs), Cin, Cout, dev,
                  lambda H, W, ws, nws, st: (xp, gp, _ptr(relu_bits), H, W, len(xs), Cin, Cout, _ptr(dw), _ptr(db), _ptr(x_transformed), 1 if pooled else 0, _ptr(dy_transformed),
                                            ws, nws, st),
                  mask=relu_bits is not None, bias=want_bias, cached=x_transformed is not None, pooled=pooled, urot=dy_transformed is not None)
    return dw, db


CONV3X3_MIN_WORK = 1 << 26         # Cin * Cout * positions: below this the stage's fixed costs (four launches per direction, partial-tile reduction) lose to
                                   # the vendor kernel (tools/dev/conv_layers_time.py: since the tile totals are padded to 64 where that pays, 256 -> 256 on 25 x 42
                                   # wins too -- 38.7 against 41.8 us forward, 92 against 108 with the gradients; smaller maps were not measured and stay)


def conv3x3_supported(x, weight, need_input_grad=None):
    """Whether `conv3x3` takes this layer: fp32 on a HIP device, batch 1, [Cout,Cin,3,3] with the multiples the stage's GEMMs need (forward
    Cin % 32, Cout % 64; the gradients Cin % 64 as well).  Everything else stays with torch's own convolution (the backbone is PyTorch code
    in the north-star's own words; this stage takes the layers where it wins)."""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1):
        return False
    Cout, Cin = int(weight.shape[0]), int(weight.shape[1])
    if tuple(weight.shape[2:]) != (3, 3) or Cout % 64 != 0 or Cin % 32 != 0 or Cin * Cout * x.shape[2] * x.shape[3] < CONV3X3_MIN_WORK:
        return False
    grads = torch.is_grad_enabled() and (weight.requires_grad or (x.requires_grad if need_input_grad is None else need_input_grad))
    Hh, Wh = _host_i32([x.shape[2]]), _host_i32([x.shape[3]])
    return bool(lib.frcnn_conv3x3_f32_supported(_np_ptr(Hh), _np_ptr(Wh), 1, Cin, Cout, 1 if grads else 0))       # channel multiples and the control block's limits


class _Conv3x3F32Fn(torch.autograd.Function):
    """args: (act, w, bias | None, x) -> act(bias + conv3x3(x, w)), act = 0 none / 1 ReLU / 2 ReLU + max_pool2d(2, 2); the activation's backward
    rides in the gradient kernels' transforms."""

    @staticmethod
    def forward(ctx, relu, w, bias, x):
        # the transformed activations are kept for the weight gradient when there will be one (0.6 GB per VGG16 step; HBM is 288 GB)
        keep = bool(ctx.needs_input_grad[1]) and int(w.shape[1]) % 64 == 0
        grads = any(ctx.needs_input_grad[1:])
        pool = relu == 2
        urot = conv3x3_u_buffer([x], w) if ctx.needs_input_grad[3] else None       # the backward's weight transform rides in the forward's launch (A/B: +1.1 % images/s)
        if keep or (relu and grads):
            ys, xt, bits = conv3x3_fwd([x], w, bias, bool(relu), keep_transformed=keep, want_bits=bool(relu) and grads, pool=pool, u_rotated=urot)
        else:
            ys, xt, bits = conv3x3_fwd([x], w, bias, bool(relu), pool=pool, u_rotated=urot), None, None
        ctx.relu = bool(relu)
        ctx.pool = pool
        ctx.has_bias = bias is not None
        ctx.save_for_backward(w, x, bits, xt, urot)        # the ReLU's backward needs the sign words, not the activations
        return ys[0]

    @staticmethod
    def backward(ctx, g):
        w, x, mask, xt, urot = ctx.saved_tensors
        g = g.contiguous()
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_b = bool(ctx.has_bias and ctx.needs_input_grad[2])
        # both gradients: the data gradient's call stages dy once and leaves its weight-gradient transform (and the bias partials) for the call behind it
        dyt = conv3x3_dy_buffer([tuple(x.shape[2:])], int(w.shape[0]), g.device) if (ctx.needs_input_grad[3] and want_w) else None
        dx = None
        if ctx.needs_input_grad[3]:
            dx = conv3x3_bwd_data([g], w, mask, [tuple(x.shape[2:])] if ctx.pool else None, u_rotated=urot, dy_transformed=dyt, want_bias_partials=want_b)[0]
        dw = db = None
        if want_w:
            dw, db = conv3x3_wgrad([x], [g], mask, want_bias=want_b, x_transformed=xt, pooled=ctx.pool, dy_transformed=dyt)
            if not ctx.needs_input_grad[1]:
                dw = None
        return None, dw, db, dx


def conv3x3(x, weight, bias=None, relu=False, pool=False):
    """nn.Conv2d(Cin, Cout, 3, padding=1) [+ nn.ReLU [+ nn.MaxPool2d(2, 2)]] on one fp32 [1,Cin,h,w] map through the hand-written Winograd stage;
    differentiable."""
    if pool and not relu:
        raise ValueError("conv3x3: pool needs relu=True")
    return _Conv3x3F32Fn.apply((2 if pool else 1) if relu else 0, weight, bias, x)


def conv3x3_pool_supported(x):
    """Whether conv3x3(..., pool=True) can take a map of x's size: the fused max-pool lives in the 4 x 4 tile's output transform."""
    Hh, Wh = _host_i32([x.shape[2]]), _host_i32([x.shape[3]])
    return int(lib.frcnn_conv3x3_f32_tile_size(_np_ptr(Hh), _np_ptr(Wh), 1)) == 4


# ---- 1 x 1 convolutions: torch's forward and input gradient, the weight gradient on the conv stage's k-contiguous GEMM (no NHWC transposes)
def _gemm_nt_splits(M, N, K):
    """How many equal pieces to cut K into: the largest divisor of K / 32 up to 64 that still leaves a piece of >= 8 chunks, once the product has fewer
    than 128 output tiles (few tiles + a long K = one workgroup adding ~100 slabs per tile)."""
    tiles = (M // (128 if M % 128 == 0 else 64)) * (N // (128 if N % 128 == 0 else 64))
    kc = K // 32
    if tiles >= 128:
        return 1
    best = 1
    for d in range(2, 65):
        if kc % d == 0 and kc // d >= 8 and tiles * d <= 2048:
            best = d
    return best


def gemm_nt(a, b):
    """a [M,K] . b [N,K]^T -> [M,N] in fp32 on the matrix cores (frcnn_gemm_nt_f32): M, N multiples of 64, K of 32; bit-reproducible."""
    a, b = _req(a, name="a"), _req(b, name="b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("gemm_nt: a [M,K] and b [N,K]")
    M, N, K = int(a.shape[0]), int(b.shape[0]), int(a.shape[1])
    if K % 32 != 0:
        raise ValueError("gemm_nt: K must be a multiple of 32")
    sp = _gemm_nt_splits(M, N, K)
    if CONV_TRACE is not None:
        CONV_TRACE.append({"kind": "gemm_nt", "M": M, "N": N, "K": K, "splits": sp})
    out = torch.empty((sp, M, N), dtype=torch.float32, device=a.device)
    ws = _ctrl_workspace(a.device, "rpn_conv_f32", int(lib.frcnn_gemm_nt_f32_workspace()))
    with torch.cuda.device(a.device):
        check(lib.frcnn_gemm_nt_f32(_ptr(a), _ptr(b), _ptr(out), M, N, K, sp, _ptr(ws), ws.numel(), _stream()), "gemm_nt_f32")
    return out[0] if sp == 1 else out.sum(0)                  # the pieces in order (a deterministic reduction)


class _Conv1x1Fn(torch.autograd.Function):
    """args: (w [Cout,Cin,1,1], bias [Cout] | None, x [1,Cin,h,w]) -> conv2d(x, w, bias), stride 1: forward, input gradient and bias gradient are
    torch's, dW = dY . X^T runs here."""

    @staticmethod
    def forward(ctx, w, bias, x):
        ctx.save_for_backward(w, x)
        ctx.has_bias = bias is not None
        return torch.nn.functional.conv2d(x, w, bias)

    @staticmethod
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        g = g.contiguous()
        dx = db = None
        want_b = ctx.has_bias and ctx.needs_input_grad[1]
        if ctx.needs_input_grad[2] or want_b:
            dx, _, db = torch.ops.aten.convolution_backward(g, x, w, [int(w.shape[0])] if ctx.has_bias else None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                            [bool(ctx.needs_input_grad[2]), False, bool(want_b)])
        dw = None
        if ctx.needs_input_grad[0]:
            Cout, Cin, HW = int(w.shape[0]), int(w.shape[1]), int(x.shape[2] * x.shape[3])
            dw = gemm_nt(g.reshape(Cout, HW), x.contiguous().reshape(Cin, HW)).reshape(Cout, Cin, 1, 1)
        return dw, db, dx


def conv1x1(x, weight, bias=None):
    """nn.Conv2d(Cin, Cout, 1) on one fp32 [1,Cin,h,w] map with the weight gradient on the library's GEMM; differentiable."""
    return _Conv1x1Fn.apply(weight, bias, x)


def conv1x1_supported(x, weight):
    """fp32 on a HIP device, batch 1, a trainable [Cout,Cin,1,1] weight with Cout, Cin multiples of 64 and h * w a multiple of 32 (the GEMM's K chunk)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1 and not torch.is_autocast_enabled()
            and tuple(weight.shape[2:]) == (1, 1) and weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0 and weight.shape[0] <= 4096
            and weight.shape[1] <= 4096 and (x.shape[2] * x.shape[3]) % 32 == 0 and torch.is_grad_enabled() and weight.requires_grad)


# ---- FrozenBatchNorm2d (+ residual) (+ ReLU) of a ResNet bottleneck in one pass each way (csrc/affine.hip)
AFFINE_TRACE = None      # a dict while a caller (bench.py) records one step: kernel name -> [launches, bytes the launch must move (every tensor it reads or writes, once)]


def _affine_trace(kernel, *tensors):
    if AFFINE_TRACE is not None:
        e = AFFINE_TRACE.setdefault(kernel, [0, 0])
        e[0] += 1
        e[1] += sum(int(t.numel()) * t.element_size() for t in tensors if t is not None)


class _AffineActFn(torch.autograd.Function):
    """args: (relu, scale [C], shift [C], x [1,C,h,w], res | None) -> act((x * scale + shift) [+ res]); scale / shift are frozen (no gradient)."""

    @staticmethod
    def forward(ctx, relu, scale, shift, x, res):
        x = _req(x, name="x")
        Cc, HW = int(x.shape[1]), int(x.shape[2] * x.shape[3])
        if res is not None:
            res = _req(res, name="res")
            if res.shape != x.shape:
                raise ValueError("affine_act: residual and input differ in shape")
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.frcnn_affine_act_fwd(_ptr(x), _ptr(res), _ptr(y), _ptr(scale), _ptr(shift), Cc, HW, 1 if relu else 0, _stream()), "affine_act_fwd")
        _affine_trace("affine_act_fwd_kernel", x, res, y, scale, shift)
        ctx.relu, ctx.has_res = bool(relu), res is not None
        ctx.save_for_backward(scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        scale, y = ctx.saved_tensors
        g = g.contiguous()
        Cc, HW = int(g.shape[1]), int(g.shape[2] * g.shape[3])
        need_x, need_r = ctx.needs_input_grad[3], ctx.has_res and ctx.needs_input_grad[4]
        if not need_x and not need_r:
            return None, None, None, None, None
        dx = torch.empty_like(g)
        dres = torch.empty_like(g) if (need_r and ctx.relu) else None     # without a ReLU the residual's gradient is g itself
        with torch.cuda.device(g.device):
            check(lib.frcnn_affine_act_bwd(_ptr(g), _ptr(y), _ptr(scale), _ptr(dx), _ptr(dres), Cc, HW, 1 if ctx.relu else 0, _stream()), "affine_act_bwd")
        _affine_trace("affine_act_bwd_kernel", g, y, dx, dres, scale)
        return None, None, None, (dx if need_x else None), ((dres if ctx.relu else g) if need_r else None)


def affine_act(x, scale, shift, res=None, relu=False):
    """FrozenBatchNorm2d(x) [+ res] [-> ReLU] for an fp32 [1,C,h,w] map: scale / shift = the norm's folded [C] (or [1,C,1,1]) statistics."""
    return _AffineActFn.apply(bool(relu), scale.reshape(-1), shift.reshape(-1), x, res)


def affine_act_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1 and not torch.is_autocast_enabled()


class _AffineActMixedFn(torch.autograd.Function):
    """args: (relu, out_bf16, want_twin, scale [C], shift [C], x [1,C,h,w] bf16 | fp32, res fp32 | None) -> y (bf16 when out_bf16, else fp32) or (y, twin):
    the mixed-precision form of _AffineActFn (csrc/affine.hip, frcnn_affine_act_*_mixed): fp32 arithmetic, bf16 on either side, and optionally the fp32
    output's bf16 twin written in the same pass; backward adds the twin's gradient to the output's before the mask."""

    @staticmethod
    def forward(ctx, relu, out_bf16, want_twin, scale, shift, x, res):
        if x.dtype not in (torch.bfloat16, torch.float32) or not x.is_cuda:
            raise ValueError("affine_act_mixed: x must be a bf16 or fp32 HIP tensor")
        x = x.contiguous()
        Cc, HW = int(x.shape[1]), int(x.shape[2] * x.shape[3])
        if res is not None:
            res = _req(res, name="res")
            if res.shape != x.shape:
                raise ValueError("affine_act_mixed: residual and input differ in shape")
        ctx.set_materialize_grads(False)                      # an output nobody used sends None, not a tensor of zeros to read
        y = torch.empty_like(x, dtype=torch.bfloat16 if out_bf16 else torch.float32)
        twin = torch.empty_like(x, dtype=torch.bfloat16) if want_twin else None
        with torch.cuda.device(x.device):
            check(lib.frcnn_affine_act_fwd_mixed(_ptr(x), 1 if x.dtype == torch.bfloat16 else 0, _ptr(res), _ptr(y), 1 if out_bf16 else 0, _ptr(twin),
                                                 _ptr(scale), _ptr(shift), Cc, HW, 1 if relu else 0, _stream()), "affine_act_fwd_mixed")
        _affine_trace("affine_act_fwd_mixed_kernel", x, res, y, twin, scale, shift)
        ctx.relu, ctx.has_res, ctx.x_bf16, ctx.twin = bool(relu), res is not None, x.dtype == torch.bfloat16, bool(want_twin)
        ctx.save_for_backward(scale, y if relu else None)
        if want_twin:
            return y, twin
        return y

    @staticmethod
    def backward(ctx, g, g2=None):
        scale, y = ctx.saved_tensors
        need_x, need_r = ctx.needs_input_grad[5], ctx.has_res and ctx.needs_input_grad[6]
        if not need_x and not need_r:
            return None, None, None, None, None, None, None
        if g is None and g2 is None:
            return None, None, None, None, None, None, None
        if g is None:                                         # only the twin was used downstream
            g = torch.zeros_like(g2, dtype=torch.float32)
        g = g.contiguous()
        if g2 is not None:
            g2 = g2.contiguous()
        Cc, HW = int(g.shape[1]), int(g.shape[2] * g.shape[3])
        dx = torch.empty_like(g, dtype=torch.bfloat16 if ctx.x_bf16 else torch.float32)
        dres = torch.empty_like(g, dtype=torch.float32) if (need_r and (ctx.relu or g2 is not None or g.dtype != torch.float32)) else None
        with torch.cuda.device(g.device):
            check(lib.frcnn_affine_act_bwd_mixed(_ptr(g), 1 if g.dtype == torch.bfloat16 else 0, _ptr(g2), _ptr(y), _ptr(scale), _ptr(dx),
                                                 1 if ctx.x_bf16 else 0, _ptr(dres), Cc, HW, 1 if ctx.relu else 0, _stream()), "affine_act_bwd_mixed")
        _affine_trace("affine_act_bwd_mixed_kernel", g, g2, y, dx, dres, scale)
        return None, None, None, None, None, (dx if need_x else None), ((dres if dres is not None else g) if need_r else None)


def affine_act_mixed(x, scale, shift, res=None, relu=False, out_bf16=False, twin=False):
    """FrozenBatchNorm2d(x) [+ res] [-> ReLU] with bf16 on either side (bf16 autocast): x bf16 | fp32 [1,C,h,w], res fp32; out_bf16: y in bf16 (an inner
    norm); twin: also return y's bf16 copy, written in the same pass -> (y, y_bf16)."""
    return _AffineActMixedFn.apply(bool(relu), bool(out_bf16), bool(twin), scale.reshape(-1), shift.reshape(-1), x, res)


def affine_act_mixed_supported(x):
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4 and x.shape[0] == 1 and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)


# ---- the backbone's first convolution: three input channels, a byte mover on the vector units (csrc/conv_c3.hip)
def conv3x3_c3_fwd(x, w, bias=None, relu=False, want_bits=False):
    """act(bias + conv3x3(x, w)), padding 1: x fp32 [1,3,h,w], w [Cout,3,3,3] -> [1,Cout,h,w] (frcnn_conv3x3_c3_fwd); want_bits: also the signs of
    the outputs as int64 words [ceil(Cout/64), h, w] for conv3x3_c3_wgrad."""
    x, w = _req(x, name="x"), _req(w, name="w")
    if x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != 3 or w.dim() != 4 or tuple(w.shape[1:]) != (3, 3, 3):
        raise ValueError("conv3x3_c3: x must be [1,3,h,w] and w [Cout,3,3,3]")
    Cout, H, W = int(w.shape[0]), int(x.shape[2]), int(x.shape[3])
    if bias is not None:
        bias = _req(bias, name="bias")
    y = torch.empty((1, Cout, H, W), dtype=torch.float32, device=x.device)
    bits = torch.empty(((Cout + 63) // 64, H, W), dtype=torch.int64, device=x.device) if want_bits else None
    with torch.cuda.device(x.device):
        check(lib.frcnn_conv3x3_c3_fwd(_ptr(x), _ptr(y), H, W, Cout, _ptr(w), _ptr(bias), 1 if relu else 0, _ptr(bits), _stream()), "conv3x3_c3_fwd")
    return (y, bits) if want_bits else y


def conv3x3_c3_wgrad(x, dy, relu_bits=None, want_bias=True):
    """(dw [Cout,3,3,3], dbias [Cout] | None) of conv3x3_c3_fwd; relu_bits: the forward's sign words (dy counts where the output was > 0)."""
    x, dy = _req(x, name="x"), _req(dy, name="dy")
    Cout, H, W = int(dy.shape[1]), int(x.shape[2]), int(x.shape[3])
    if x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != 3 or tuple(dy.shape) != (1, Cout, H, W):
        raise ValueError("conv3x3_c3_wgrad: x must be [1,3,h,w] and dy [1,Cout,h,w]")
    if relu_bits is not None:
        relu_bits = _req(relu_bits, torch.int64, "relu_bits")
        if tuple(relu_bits.shape) != ((Cout + 63) // 64, H, W):
            raise ValueError("conv3x3_c3_wgrad: relu_bits does not belong to these shapes")
    dw = torch.empty((Cout, 3, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    ws = _workspace(x.device, int(lib.frcnn_conv3x3_c3_wgrad_workspace(H, Cout)))
    with torch.cuda.device(x.device):
        check(lib.frcnn_conv3x3_c3_wgrad(_ptr(x), _ptr(dy), H, W, Cout, _ptr(relu_bits), _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), _stream()), "conv3x3_c3_wgrad")
    return dw, db


class _Conv3x3C3Fn(torch.autograd.Function):
    """args: (relu, w, bias | None, x) with x the image (no gradient flows into it)."""

    @staticmethod
    def forward(ctx, relu, w, bias, x):
        grads = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])
        if relu and grads:
            y, bits = conv3x3_c3_fwd(x, w, bias, True, want_bits=True)
        else:
            y, bits = conv3x3_c3_fwd(x, w, bias, relu), None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, bits)
        return y

    @staticmethod
    def backward(ctx, g):
        x, bits = ctx.saved_tensors
        if ctx.needs_input_grad[3]:
            raise RuntimeError("conv3x3_c3: no input gradient (the input is the image)")
        dw, db = conv3x3_c3_wgrad(x, g.contiguous(), bits, want_bias=ctx.has_bias)
        return None, dw, db, None


def conv3x3_c3(x, weight, bias=None, relu=False):
    """nn.Conv2d(3, Cout, 3, padding=1) [+ nn.ReLU] on the fp32 image [1,3,h,w]; differentiable in weight and bias."""
    return _Conv3x3C3Fn.apply(bool(relu), weight, bias, x)


def conv3x3_c3_supported(x, weight):
    """fp32 on a HIP device, batch 1, [Cout,3,3,3] with Cout % 4 == 0 and Cout <= 256 (the forward keeps Cout * 28 weights in LDS), the input
    itself needing no gradient, rows of at most 1700 pixels.  Anything else stays with torch's convolution."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 3
            and tuple(weight.shape[1:]) == (3, 3, 3) and weight.shape[0] % 4 == 0 and weight.shape[0] <= 256 and x.shape[3] <= 1700
            and not x.requires_grad)


def rpn_conv_head_levels(feats, w3, b3, w_cls, b_cls, w_reg, b_reg):
    """(pred_cls [1, sum P_l * A, 2], pred_reg [1, sum P_l * A, 4]) of the shared FPN RPN head on bf16 feature maps (models/new_model.py:37-44,
    89-114): one launch for conv3x3 + ReLU + both heads on the bf16 matrix cores; box regression outputs stay fp32."""
    return _RPNConvHeadFn.apply(w3, b3, w_cls, b_cls, w_reg, b_reg, *feats)


def rpn_head_tail(conv_raw, b3, w_cls, b_cls, w_reg, b_reg):
    """(pred_cls [1, P*A, 2], pred_reg [1, P*A, 4]) from the bias-free 3x3 output; see include/frcnn_hip.h."""
    return _RPNHeadTailFn.apply(False, b3, w_cls, b_cls, w_reg, b_reg, conv_raw)


def rpn_head_tail_levels(conv_raws, b3, w_cls, b_cls, w_reg, b_reg, mfma="f32"):
    """The shared RPN head tail over all FPN levels (models/new_model.py:37-44,109-113) in one launch: outputs are the
    torch.cat(dim=1) of the per-level results.  mfma='bf16' contracts bf16-rounded operands with fp32 accumulation
    (mixed-precision configuration); biases and outputs are fp32 either way."""
    if mfma not in ("f32", "bf16"):
        raise ValueError("mfma must be 'f32' or 'bf16'")
    return _RPNHeadTailFn.apply(mfma == "bf16", b3, w_cls, b_cls, w_reg, b_reg, *conv_raws)


# --------------------------------------------------------------------------------------------
# target makers
# --------------------------------------------------------------------------------------------
def host_count(cnt, what="nms"):
    """Read a device-side count on the host (a sync).  The library reports an aborted scan (a bounded spin ran out: should never
    happen, see nms.hip) as -1: fail loudly instead of slicing with it."""
    n = int(cnt.item())
    if n < 0:
        raise _lib.FrcnnError("%s: the device-side scan aborted (count = %d); results are invalid" % (what, n))
    return n


def describe_status(bits):
    """Names of the _lib.HT_ERR_* bits set in a status word."""
    names = ((_lib.HT_ERR_PERM_LENGTH, "permutation length mismatch"), (_lib.HT_ERR_PERM_RANGE, "permutation entry out of range"),
             (_lib.HT_ERR_UPSTREAM_ABORT, "the proposal stage reported an aborted NMS scan (count < 0)"),
             (_lib.HT_ERR_SHORT, "fewer RoI samples than requested (the reference throws here)"))
    return [n for b, n in names if bits & b]


class DeviceStatus(object):
    """Sticky device-side error word of one model (one int32 per device): target makers OR their error bits into it without a
    host sync; `check()` reads it (a sync -- call it where the training loop syncs anyway: logging / checkpoint interval)."""

    def __init__(self):
        self._words = {}

    def word(self, device):
        w = self._words.get(device)
        if w is None:
            w = torch.zeros((1,), dtype=torch.int32, device=device)
            self._words[device] = w
        return w

    def check(self, reset=True):
        for dev, w in self._words.items():
            bits = int(w.item())
            if bits:
                if reset:
                    w.zero_()
                raise _lib.FrcnnError("device status %d on %s since the last check: %s" % (bits, dev, "; ".join(describe_status(bits))))


def _perm(p, dev):
    if p is None:
        return None, 0
    p = torch.as_tensor(p, dtype=torch.int64).to(dev).contiguous()
    return p, p.numel()


def philox_state(seed, offset, device):
    """Device-resident RNG stream of the target makers: int64[2] = (seed, offset) bit patterns.  Every sampling call that is
    handed this tensor uses the pair it finds and leaves offset + 1 behind (include/frcnn_hip.h), so a captured HIP graph of
    the training step draws new samples at each replay."""
    m = (1 << 64) - 1
    to_i64 = lambda v: (v & m) - (1 << 64) if (v & m) >= (1 << 63) else (v & m)      # noqa: E731
    return torch.tensor([to_i64(int(seed)), to_i64(int(offset))], dtype=torch.int64, device=device)


def _philox(state):
    if state is None:
        return None
    return _req(state, torch.int64, "philox_state")


def rpn_targets(anchors, gt, variant=0, perm_pos=None, perm_neg=None, seed=0, offset=0, philox_state=None):
    """RPNTargetMaker.forward.  Returns (cls[N] i64, reg[N,4], counts int32[4] device = n_pos, n_neg, err, -).
    philox_state (int64[2] on the device, see ops.philox_state) overrides (seed, offset) and is advanced by the call."""
    anchors = _req(anchors, name="anchors").reshape(-1, 4)
    gt = _req(gt, name="gt").reshape(-1, 4)
    N, G = anchors.shape[0], gt.shape[0]
    dev = anchors.device
    cls = torch.empty((N,), dtype=torch.int64, device=dev)
    reg = torch.empty((N, 4), dtype=torch.float32, device=dev)
    counts = torch.empty((4,), dtype=torch.int32, device=dev)
    pp, npp = _perm(perm_pos, dev)
    pn, npn = _perm(perm_neg, dev)
    nb = _lib.workspace_bytes(_lib.OP_RPN_TARGETS, N, G)
    ws = _ctrl_workspace(dev, "rpn_targets", nb)          # holds the fused kernel's barrier / ticket words: zero on first use, left zero by every call
    with torch.cuda.device(dev):
        check(lib.frcnn_rpn_targets(int(variant), _ptr(anchors), N, _ptr(gt), G, _ptr(pp), npp, _ptr(pn), npn, int(seed), int(offset),
                                    _ptr(_philox(philox_state)), _ptr(cls), _ptr(reg), _ptr(counts), _ptr(ws), ws.numel(), _stream()), "rpn_targets")
    return cls, reg, counts


def head_targets(rois, gt, gt_label, n_rois=None, variant=0, label_offset=1, max_pos=32, total=128,
                 perm_pos=None, perm_neg=None, seed=0, offset=0, want_keep=False, status=None, philox_state=None):
    """FastRcnnTargetMaker.forward.  Returns (cls[total] i64, reg[total,4], sample_rois[total,4], keep|None, counts int32[4]).
    counts[3] holds the _lib.HT_ERR_* bits; they are also OR-ed into `status` (device int32[1], sticky across steps) if given."""
    rois = _req(rois, name="rois").reshape(-1, 4)
    gt = _req(gt, name="gt").reshape(-1, 4)
    gt_label = _req(gt_label, torch.int64, "gt_label").reshape(-1)
    dev = rois.device
    if n_rois is not None:
        n_rois = _req(n_rois, torch.int32, "n_rois")
    cls = torch.empty((total,), dtype=torch.int64, device=dev)
    reg = torch.empty((total, 4), dtype=torch.float32, device=dev)
    srois = torch.empty((total, 4), dtype=torch.float32, device=dev)
    keep = torch.empty((total,), dtype=torch.int64, device=dev) if want_keep else None
    counts = torch.empty((4,), dtype=torch.int32, device=dev)
    pp, npp = _perm(perm_pos, dev)
    pn, npn = _perm(perm_neg, dev)
    if status is not None:
        status = _req(status, torch.int32, "status")
    with torch.cuda.device(dev):
        check(lib.frcnn_head_targets(int(variant), _ptr(rois), _ptr(n_rois), rois.shape[0], _ptr(gt), _ptr(gt_label), gt.shape[0],
                                     int(label_offset), int(max_pos), int(total), _ptr(pp), npp, _ptr(pn), npn, int(seed), int(offset),
                                     _ptr(_philox(philox_state)), _ptr(cls), _ptr(reg), _ptr(srois), _ptr(keep), _ptr(counts), _ptr(status), _stream()),
              "head_targets")
    return cls, reg, srois, keep, counts


# --------------------------------------------------------------------------------------------
# detection losses (losses/loss.py:5-85)
# --------------------------------------------------------------------------------------------
class _DetLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rpn_cls, rpn_reg, head_cls, head_reg, t_rpn_cls, t_rpn_reg, t_cls, t_reg):
        shapes = (rpn_cls.shape, rpn_reg.shape, head_cls.shape, head_reg.shape)
        rc = _req(rpn_cls, name="pred_rpn_cls").reshape(-1, 2)
        rr = _req(rpn_reg, name="pred_rpn_reg").reshape(-1, 4)
        hc = _req(head_cls, name="pred_fast_rcnn_cls")
        hr = _req(head_reg, name="pred_fast_rcnn_reg").reshape(-1, 4)
        trc = _req(t_rpn_cls, torch.int64, "target_rpn_cls").reshape(-1)
        trr = _req(t_rpn_reg, name="target_rpn_reg").reshape(-1, 4)
        tc = _req(t_cls, torch.int64, "target_fast_rcnn_cls").reshape(-1)
        tr = _req(t_reg, name="target_fast_rcnn_reg").reshape(-1, 4)
        N, R, NC = rc.shape[0], hc.shape[0], hc.shape[1]
        if rr.shape[0] != N or trc.shape[0] != N or trr.shape[0] != N or hr.shape[0] != R or tc.shape[0] != R or tr.shape[0] != R:
            raise ValueError("detection_loss: shape mismatch")
        dev = rc.device
        out = torch.empty((7,), dtype=torch.float32, device=dev)
        g = [torch.empty_like(t) for t in (rc, rr, hc, hr)]
        ws = _ctrl_workspace(dev, "det_loss", 32768)        # zero on first use; the kernel's last workgroup leaves its ticket zero
        with torch.cuda.device(dev):
            check(lib.frcnn_detection_loss(_ptr(rc), _ptr(rr), _ptr(trc), _ptr(trr), N, _ptr(hc), _ptr(hr), _ptr(tc), _ptr(tr), R, NC,
                                           _ptr(out), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(g[3]), _ptr(ws), ws.numel(), _stream()), "detection_loss")
        ctx.save_for_backward(out, *g)
        ctx.shapes = shapes
        ctx.set_materialize_grads(False)        # the four terms are usually reported, not differentiated: no zero tensors (a fill launch each) for them
        return out[0], out[1], out[2], out[3], out[4]

    @staticmethod
    def backward(ctx, g_total, g1, g2, g3, g4):
        out, g_rc, g_rr, g_hc, g_hr = ctx.saved_tensors
        sh = ctx.shapes
        if g_total is not None and g1 is None and g2 is None and g3 is None and g4 is None:
            # the training step: only the total is differentiated.  Three launches (the two normalisers times the upstream gradient, then one
            # multi-tensor multiply per normaliser) where the general form below takes seventeen; the same products, bit for bit
            # (x + 0 = x, and the factors are formed in the same order)
            t = g_total * out[5:7]
            d_rpn = torch._foreach_mul([g_rc, g_rr], t[0])
            d_head = torch._foreach_mul([g_hc, g_hr], t[1])
            return d_rpn[0].reshape(sh[0]), d_rpn[1].reshape(sh[1]), d_head[0].reshape(sh[2]), d_head[1].reshape(sh[3]), None, None, None, None
        z = out.new_zeros(())
        g_total, g1, g2, g3, g4 = (z if t is None else t for t in (g_total, g1, g2, g3, g4))
        s_rpn, s_head = out[5], out[6]
        return ((g_rc * ((g_total + g1) * s_rpn)).reshape(sh[0]), (g_rr * ((g_total + g2) * s_rpn)).reshape(sh[1]),
                (g_hc * ((g_total + g3) * s_head)).reshape(sh[2]), (g_hr * ((g_total + g4) * s_head)).reshape(sh[3]), None, None, None, None)


def detection_loss(pred, target):
    """FRCNNLoss.forward (losses/loss.py:71-85): (total, rpn_cls, rpn_reg, fast_rcnn_cls, fast_rcnn_reg)."""
    return _DetLossFn.apply(pred[0], pred[1], pred[2], pred[3], target[0], target[1], target[2], target[3])


# --------------------------------------------------------------------------------------------
# RoIPool (models/model.py:97,113)
# --------------------------------------------------------------------------------------------
class _RoIPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, PH, PW, scale):
        feat = _req(feat, name="features")
        rois = _req(rois, name="rois").reshape(-1, 4)
        if feat.dim() != 4 or feat.shape[0] != 1:
            raise ValueError("RoIPool: features must be [1,C,H,W] (the reference trains with batch 1 per GPU)")
        _, Cc, H, W = feat.shape
        R = rois.shape[0]
        out = torch.empty((R, Cc, PH, PW), dtype=torch.float32, device=feat.device)
        # the argmax never leaves this Function: 16 bits are enough for planes below 65535 pixels (the library's private pair)
        a16 = PH == 7 and PW == 7 and H * W < 65535 and 16 * H * W <= 48 * 1024 and R < (1 << 24)
        arg = torch.empty((R, Cc, PH, PW), dtype=torch.uint16 if a16 else torch.int32, device=feat.device)
        with torch.cuda.device(feat.device):
            if a16:
                check(lib.frcnn_roi_pool_fwd_a16(_ptr(feat), Cc, H, W, _ptr(rois), R, float(scale), _ptr(out), _ptr(arg), _stream()), "roi_pool_fwd_a16")
            else:
                check(lib.frcnn_roi_pool_fwd(_ptr(feat), Cc, H, W, _ptr(rois), R, PH, PW, float(scale), _ptr(out), _ptr(arg), _stream()),
                      "roi_pool_fwd")
        ctx.save_for_backward(arg, rois)
        ctx.shape = (Cc, H, W, PH, PW, R, float(scale))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        arg, rois = ctx.saved_tensors
        Cc, H, W, PH, PW, R, scale = ctx.shape
        grad_out = _req(grad_out, name="grad_out")
        gf = torch.empty((1, Cc, H, W), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            if arg.dtype == torch.uint16:
                # (the boxes let the backward add without LDS atomics wherever a RoI spans at least 7 x 7 cells)
                check(lib.frcnn_roi_pool_bwd_a16(_ptr(grad_out), _ptr(arg), _ptr(rois), scale, R, Cc, H, W, _ptr(gf), _stream()), "roi_pool_bwd_a16")
            else:
                check(lib.frcnn_roi_pool_bwd(_ptr(grad_out), _ptr(arg), R, Cc, H, W, PH, PW, _ptr(gf), _stream()), "roi_pool_bwd")
        return gf, None, None, None, None      # no gradient through box coordinates


def roi_pool_with_argmax(features, rois, output_size=(7, 7), spatial_scale=1.0):
    """The torchvision-shaped forward: (out [R,C,PH,PW], argmax int32 [R,C,PH,PW], -1 = empty bin), no autograd (tests, parity)."""
    feat = _req(features, name="features")
    rois = _req(rois, name="rois").reshape(-1, 4)
    _, Cc, H, W = feat.shape
    PH, PW = output_size
    R = rois.shape[0]
    out = torch.empty((R, Cc, PH, PW), dtype=torch.float32, device=feat.device)
    arg = torch.empty((R, Cc, PH, PW), dtype=torch.int32, device=feat.device)
    with torch.cuda.device(feat.device):
        check(lib.frcnn_roi_pool_fwd(_ptr(feat), Cc, H, W, _ptr(rois), R, PH, PW, float(spatial_scale), _ptr(out), _ptr(arg), _stream()), "roi_pool_fwd")
    return out, arg


def roi_pool(features, rois, output_size=(7, 7), spatial_scale=1.0):
    if isinstance(rois, (list, tuple)):
        if len(rois) != 1:
            raise ValueError("roi_pool: one image per call (batch 1 per GPU)")
        rois = rois[0]
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    return _RoIPoolFn.apply(features, rois, int(output_size[0]), int(output_size[1]), float(spatial_scale))


class RoIPool(torch.nn.Module):
    """torchvision.ops.RoIPool(output_size, spatial_scale) with the call convention of models/model.py:113."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale

    def forward(self, input, rois):
        return roi_pool(input, rois, self.output_size, self.spatial_scale)


# --------------------------------------------------------------------------------------------
# MultiScaleRoIAlign (models/new_model.py:127,143)
# --------------------------------------------------------------------------------------------
def roi_level_map(rois, k_min=2, k_max=5, s0=224.0, k0=4, eps=1e-6):
    rois = _req(rois, name="rois").reshape(-1, 4)
    out = torch.empty((rois.shape[0],), dtype=torch.int32, device=rois.device)
    with torch.cuda.device(rois.device):
        check(lib.frcnn_roi_level_map(_ptr(rois), rois.shape[0], k_min, k_max, float(s0), k0, float(eps), _ptr(out), _stream()), "roi_level_map")
    return out


def _level_tables(feats, scales):
    n = len(feats)
    ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in feats])
    H = _host_i32([f.shape[-2] for f in feats])
    W = _host_i32([f.shape[-1] for f in feats])
    sc = np.ascontiguousarray(scales, dtype=np.float32)
    return ptrs, H, W, sc


class _MsRoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, PH, PW, sampling_ratio, aligned, scales, k_min, s0, k0, order, *feats):
        feats = [_req(f, name="feature map") for f in feats]
        rois = _req(rois, name="rois").reshape(-1, 4)
        Cc = feats[0].shape[1]
        for f in feats:
            if f.dim() != 4 or f.shape[0] != 1 or f.shape[1] != Cc:
                raise ValueError("MultiScaleRoIAlign: feature maps must be [1,C,H,W] with a common C")
        R = rois.shape[0]
        if order is not None:
            order = _req(order, torch.int32, "order").reshape(-1)
            if order.shape[0] != R:
                raise ValueError("MultiScaleRoIAlign: order must hold one entry per RoI")
        out = torch.empty((R, Cc, PH, PW), dtype=torch.float32, device=rois.device)
        ptrs, H, W, sc = _level_tables(feats, scales)
        with torch.cuda.device(rois.device):
            check(lib.frcnn_ms_roi_align_fwd(ptrs, _np_ptr(H), _np_ptr(W), _np_ptr(sc), len(feats), Cc, _ptr(rois), R, PH, PW,
                                             sampling_ratio, int(aligned), k_min, float(s0), k0, _ptr(out), None, _ptr(order), _stream()), "ms_roi_align_fwd")
        ctx.save_for_backward(rois)
        ctx.meta = (PH, PW, sampling_ratio, aligned, tuple(scales), k_min, s0, k0, [tuple(f.shape) for f in feats])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        PH, PW, sampling_ratio, aligned, scales, k_min, s0, k0, shapes = ctx.meta
        grad_out = _req(grad_out, name="grad_out")
        grads = [torch.empty(s, dtype=torch.float32, device=grad_out.device) for s in shapes]     # overwritten by the library
        ptrs, H, W, sc = _level_tables(grads, scales)
        with torch.cuda.device(grad_out.device):
            nb = lib.frcnn_ms_roi_align_bwd_workspace(_np_ptr(H), _np_ptr(W), len(grads), shapes[0][1], rois.shape[0])
            ws = _workspace(grad_out.device, max(nb, 256))
            check(lib.frcnn_ms_roi_align_bwd(_ptr(grad_out), ptrs, _np_ptr(H), _np_ptr(W), _np_ptr(sc), len(grads), shapes[0][1], _ptr(rois),
                                             rois.shape[0], PH, PW, sampling_ratio, int(aligned), k_min, float(s0), k0, _ptr(ws), ws.numel(),
                                             _stream()), "ms_roi_align_bwd")
        return (None,) * 10 + tuple(grads)


def roi_scale_order(rois, mul, feat_shapes, scales=(0.25, 0.125, 0.0625, 0.03125), aligned=False, canonical_scale=224.0, canonical_level=4,
                    want_cost=False):
    """(rois * mul, order[, cost]) in one launch: the `roi * (w, h, w, h)` of FastRCNNHead.forward (models/new_model.py:136-140) and the
    dispatch order (largest footprint first) that ms_roi_align(..., order=order) hands to the forward kernel.  feat_shapes = the (h, w) of
    the pyramid levels the pooler reads; results of the pooling do not depend on `order`."""
    rois = _req(rois, name="rois").reshape(-1, 4)
    R = rois.shape[0]
    dev = rois.device
    out = torch.empty_like(rois)
    order = torch.empty((R,), dtype=torch.int32, device=dev)
    cost = torch.empty((R,), dtype=torch.int32, device=dev) if want_cost else None
    H = _host_i32([s[0] for s in feat_shapes])
    W = _host_i32([s[1] for s in feat_shapes])
    sc = np.ascontiguousarray(scales, dtype=np.float32)
    m = np.ascontiguousarray(mul, dtype=np.float32)
    if m.shape != (4,):
        raise ValueError("roi_scale_order: mul must have four entries")
    k_min = int(round(-np.log2(scales[0])))
    with torch.cuda.device(dev):
        check(lib.frcnn_roi_scale_order(_ptr(rois), R, _np_ptr(m), _np_ptr(H), _np_ptr(W), _np_ptr(sc), len(feat_shapes), int(aligned), k_min,
                                        float(canonical_scale), int(canonical_level), _ptr(out), _ptr(order), _ptr(cost), _stream()), "roi_scale_order")
    return (out, order, cost) if want_cost else (out, order)


def ms_roi_align(feats, rois, output_size=7, sampling_ratio=2, scales=(0.25, 0.125, 0.0625, 0.03125), aligned=False,
                 canonical_scale=224.0, canonical_level=4, order=None):
    k_min = int(round(-np.log2(scales[0])))
    PH, PW = (output_size, output_size) if isinstance(output_size, int) else output_size
    feats = [f if f.dtype == torch.float32 else f.float() for f in feats]      # under autocast: pooling runs in fp32 (as torchvision's does)
    rois = rois.float()
    return _MsRoIAlignFn.apply(rois, int(PH), int(PW), int(sampling_ratio), bool(aligned), tuple(float(s) for s in scales), k_min,
                               float(canonical_scale), int(canonical_level), order, *feats)


def infer_scales_like_torchvision(feat_shapes, image_shapes):
    """torchvision.ops.poolers: `_setup_scales` / `_infer_scale` as published (>= 0.13):
        original = (max over images of shape[0], max over images of shape[1])     # torchvision MEANS (h, w)
        scale_l  = 2 ** round(log2(feat_l.shape[-2] / original[0]))               # only the FIRST axis decides
    The reference passes image_shapes = [(w, h)] (models/new_model.py:143), so under torchvision the FEATURE
    HEIGHT is divided by the IMAGE WIDTH: an 800x1344 frame gives 200/1344 -> 2^-3, i.e. levels 1/8 .. 1/64 and k_min = 3
    instead of 1/4 .. 1/32 and k_min = 2.  Pure host arithmetic on shapes (fp32 log2 / round like torch.tensor(x).log2().round())."""
    o0 = max(int(sh[0]) for sh in image_shapes)
    scales = []
    for fh, _fw in feat_shapes:
        approx = np.float32(float(fh) / float(o0))
        scales.append(2.0 ** float(np.round(np.log2(approx))))
    return tuple(scales)


class MultiScaleRoIAlign(torch.nn.Module):
    """torchvision.ops.MultiScaleRoIAlign(featmap_names, output_size, sampling_ratio) with the call convention of
    models/new_model.py:143: forward(features: dict, [rois in image pixels], image_shapes).

    scales=None (default): the per-level scales are the true strides 1/4 .. 1/32 whatever image_shapes says.  The reference
        passes (w, h) where torchvision expects (h, w), so torchvision's inference from image_shapes picks the
        WRONG pyramid level scales for non-square frames; the default corrects that.
    scales='reference': reproduce what the reference actually computes under torchvision, i.e. infer the scales (and with
        them the level mapper's k_min / k_max) from image_shapes exactly as torchvision does, swapped axes included
        (`infer_scales_like_torchvision`).  Needed to evaluate checkpoints TRAINED with the reference on non-square frames
        at the pooling geometry they were trained with.
    scales=(s0, s1, ...): explicit."""

    def __init__(self, featmap_names, output_size, sampling_ratio, scales=None):
        super().__init__()
        self.featmap_names = list(featmap_names)
        self.output_size = output_size
        self.sampling_ratio = sampling_ratio
        if isinstance(scales, str):
            if scales != "reference":
                raise ValueError("scales must be None, 'reference' or a tuple of floats")
            self.scales = "reference"
        else:
            self.scales = tuple(scales) if scales is not None else tuple(2.0 ** -(2 + i) for i in range(len(self.featmap_names)))

    def forward(self, x, boxes, image_shapes=None, order=None):
        if isinstance(boxes, (list, tuple)):
            if len(boxes) != 1:
                raise ValueError("MultiScaleRoIAlign: one image per call (batch 1 per GPU)")
            boxes = boxes[0]
        feats = [x[k] for k in self.featmap_names]
        scales = self.scales
        if scales == "reference":
            if not image_shapes:
                raise ValueError("MultiScaleRoIAlign(scales='reference') needs image_shapes (as passed at new_model.py:143)")
            scales = infer_scales_like_torchvision([tuple(f.shape[-2:]) for f in feats], image_shapes)
        return ms_roi_align(feats, boxes, self.output_size, self.sampling_ratio, scales, order=order)


# --------------------------------------------------------------------------------------------
# AnchorGenerator (models/new_model.py:23-25,46)
# --------------------------------------------------------------------------------------------
class ImageList(object):
    """torchvision.models.detection.image_list.ImageList as the reference builds it at models/new_model.py:46:
    ImageList(x, [(w, h)]).  AnchorGenerator reads only `.tensors` (strides come from the padded batch tensor's shape),
    so the swapped (w, h) in `.image_sizes` is harmless there."""

    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device):
        return ImageList(self.tensors.to(device), self.image_sizes)


class AnchorGenerator(torch.nn.Module):
    """torchvision.models.detection.rpn.AnchorGenerator(sizes, aspect_ratios) restricted to what the reference
    uses: one size per level, a shared ratio tuple.  Called as the reference does (models/new_model.py:46):
    anchor_generator(ImageList(x, [(w, h)]), features) -> [anchors[N,4] in pixels]; `features` may be the backbone's
    OrderedDict or a list of maps; an (H, W) pair is accepted in place of the ImageList.
    Anchors depend only on shapes, so they are cached per (image, feature) shape and stay resident in HBM.  The reference
    divides the returned tensor IN PLACE by (w, h, w, h) (new_model.py:47): the list therefore holds a clone of the cached grid."""

    def __init__(self, sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),) * 5):
        super().__init__()
        for s in sizes:
            if len(s) != 1:
                raise ValueError("AnchorGenerator: one size per level (as the reference configures it)")
        self.sizes = tuple(float(s[0]) for s in sizes)
        self.aspect_ratios = tuple(tuple(float(r) for r in ar) for ar in aspect_ratios)
        if len(set(self.aspect_ratios)) != 1:
            raise ValueError("AnchorGenerator: all levels must share the aspect ratios")
        self._cache = {}

    def grid(self, image_hw, feat_shapes, device, normalise=False):
        H, W = int(image_hw[0]), int(image_hw[1])
        key = (H, W, tuple(feat_shapes), str(device), normalise)
        a = self._cache.get(key)
        if a is None:
            base = np.stack([tv_base_anchors(s, self.aspect_ratios[0]) for s in self.sizes[:len(feat_shapes)]])
            strides = [(H // fh, W // fw) for fh, fw in feat_shapes]
            a = anchor_grid(feat_shapes, strides, base, W if normalise else 1.0, H if normalise else 1.0, device)
            self._cache[key] = a
        return a

    def forward(self, image_list, feature_maps):
        if hasattr(feature_maps, "values"):
            feature_maps = list(feature_maps.values())
        if hasattr(image_list, "tensors"):                       # ImageList-shaped (torchvision's, or ops.ImageList)
            t = image_list.tensors
            image_hw = tuple(t.shape[-2:])
            sizes = getattr(image_list, "image_sizes", None)
            n_img = len(sizes) if sizes is not None else (t.shape[0] if t.dim() == 4 else 1)     # one anchor set per image, as torchvision
        else:
            image_hw, n_img = tuple(int(v) for v in image_list), 1
        shapes = [tuple(f.shape[-2:]) for f in feature_maps]
        a = self.grid(image_hw, shapes, feature_maps[0].device)
        return [a.clone() for _ in range(n_img)]
