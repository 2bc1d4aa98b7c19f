"""ctypes front-end of the CPU ORACLE (oracle/frcnn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under faster_rcnn_pytorch_amd/ imports it.
All arrays are numpy; boxes fp32 [n,4] xyxy normalised, indices int64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "frcnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_expf.restype = C.c_float
        _lib.orc_expf.argtypes = [C.c_float]
        _lib.orc_log2f.restype = C.c_float
        _lib.orc_log2f.argtypes = [C.c_float]
        _lib.orc_fg_softmax.restype = C.c_float
        _lib.orc_fg_softmax.argtypes = [C.c_float, C.c_float]
        for name in ("orc_tv_anchor_grid", "orc_topk_sorted", "orc_nms", "orc_region_proposal", "orc_head_targets"):
            getattr(_lib, name).restype = C.c_int64
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc, name):
    if rc < 0:
        raise RuntimeError("oracle %s failed: %d" % (name, rc))
    return rc


# -- deterministic math -----------------------------------------------------
def expf(x):
    L = lib()
    x = _f32(x)
    return np.array([L.orc_expf(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def log2f(x):
    L = lib()
    x = _f32(x)
    return np.array([L.orc_log2f(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


# -- anchors ------------------------------------------------------------------
def anchor_base(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    r = np.asarray(ratios, dtype=np.float64)
    s = np.asarray(scales, dtype=np.float64)
    out = np.empty((len(r) * len(s), 4), np.float32)
    _chk(lib().orc_anchor_base(C.c_int(base_size), _p(r), C.c_int(len(r)), _p(s), C.c_int(len(s)), _p(out)), "anchor_base")
    return out


def anchor_grid(H, W, base=None, base_size=16):
    base = anchor_base(base_size) if base is None else _f32(base)
    A = base.shape[0]
    out = np.empty(((H // base_size) * (W // base_size) * A, 4), np.float32)
    _chk(lib().orc_anchor_grid(C.c_int(H), C.c_int(W), C.c_int(base_size), _p(base), C.c_int(A), _p(out)), "anchor_grid")
    return out


def tv_base_anchors(size, ratios=(0.5, 1.0, 2.0)):
    r = _f32(ratios)
    out = np.empty((len(r), 4), np.float32)
    _chk(lib().orc_tv_base_anchors(C.c_float(size), _p(r), C.c_int(len(r)), _p(out)), "tv_base_anchors")
    return out


def tv_anchor_grid(H, W, feat_shapes, sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1.0, 2.0), normalise=True):
    fh = np.ascontiguousarray([s[0] for s in feat_shapes], dtype=np.int32)
    fw = np.ascontiguousarray([s[1] for s in feat_shapes], dtype=np.int32)
    sz = _f32(sizes)
    r = _f32(ratios)
    n = int(sum(int(a) * int(b) for a, b in zip(fh, fw)) * len(r))
    out = np.empty((n, 4), np.float32)
    got = _chk(lib().orc_tv_anchor_grid(C.c_int(H), C.c_int(W), C.c_int(len(fh)), _p(fh), _p(fw), _p(sz), _p(r),
                                        C.c_int(len(r)), C.c_int(1 if normalise else 0), _p(out)), "tv_anchor_grid")
    assert got == n
    return out


# -- box codec ------------------------------------------------------------------
def xy_to_cxcy(xy):
    xy = _f32(xy)
    out = np.empty_like(xy)
    _chk(lib().orc_xy_to_cxcy(_p(xy), C.c_int64(xy.shape[0]), _p(out)), "xy_to_cxcy")
    return out


def cxcy_to_xy(c):
    c = _f32(c)
    out = np.empty_like(c)
    _chk(lib().orc_cxcy_to_xy(_p(c), C.c_int64(c.shape[0]), _p(out)), "cxcy_to_xy")
    return out


def decode(t, center_anchor, use_libm=False):
    t, a = _f32(t), _f32(center_anchor)
    out = np.empty_like(t)
    _chk(lib().orc_decode(_p(t), _p(a), C.c_int64(t.shape[0]), _p(out), C.c_int(int(use_libm))), "decode")
    return out


def encode(gt_cxcy, anc_cxcy):
    g, a = _f32(gt_cxcy), _f32(anc_cxcy)
    out = np.empty_like(g)
    _chk(lib().orc_encode(_p(g), _p(a), C.c_int64(g.shape[0]), _p(out)), "encode")
    return out


def pairwise_iou(s1, s2, eps=1e-5):
    """eps=1e-5: find_jaccard_overlap (utils/util.py:66); eps=0: box_iou (util/box_ops.py:24)."""
    s1, s2 = _f32(s1), _f32(s2)
    out = np.empty((s1.shape[0], s2.shape[0]), np.float32)
    _chk(lib().orc_pairwise_iou(_p(s1), C.c_int64(s1.shape[0]), _p(s2), C.c_int64(s2.shape[0]), C.c_float(eps), _p(out)), "iou")
    return out


# -- proposal stage ---------------------------------------------------------------
def fg_softmax(cls):
    cls = _f32(cls)
    L = lib()
    return np.array([L.orc_fg_softmax(float(a), float(b)) for a, b in cls], dtype=np.float32)


def proposal_prologue(reg, cls, anchor, min_size_norm):
    reg, cls, anchor = _f32(reg), _f32(cls), _f32(anchor)
    N = reg.shape[0]
    boxes = np.empty((N, 4), np.float32)
    scores = np.empty((N,), np.float32)
    nv = C.c_int64(0)
    _chk(lib().orc_proposal_prologue(_p(reg), _p(cls), _p(anchor), C.c_int64(N), C.c_float(min_size_norm),
                                     _p(boxes), _p(scores), C.byref(nv)), "prologue")
    return boxes, scores, nv.value


def topk_sorted(scores, K):
    scores = _f32(scores)
    idx = np.empty((max(K, 1),), np.int64)
    sc = np.empty((max(K, 1),), np.float32)
    k = _chk(lib().orc_topk_sorted(_p(scores), C.c_int64(scores.shape[0]), C.c_int64(K), _p(idx), _p(sc)), "topk")
    return idx[:k].copy(), sc[:k].copy()


def nms(boxes, thr, order=None):
    """Greedy NMS over boxes visited in `order` (None = as given, i.e. already score-sorted)."""
    boxes = _f32(boxes)
    K = boxes.shape[0]
    keep = np.empty((max(K, 1),), np.int64)
    o = None if order is None else _i64(order)
    nk = _chk(lib().orc_nms(_p(boxes), _p(o), C.c_int64(K), C.c_float(thr), _p(keep)), "nms")
    return keep[:nk].copy()


def region_proposal(reg, cls, anchor, min_size_norm, K, thr, P):
    reg, cls, anchor = _f32(reg), _f32(cls), _f32(anchor)
    rois = np.empty((max(P, 1), 4), np.float32)
    src = np.empty((max(P, 1),), np.int64)
    n = _chk(lib().orc_region_proposal(_p(reg), _p(cls), _p(anchor), C.c_int64(reg.shape[0]), C.c_float(min_size_norm),
                                       C.c_int64(K), C.c_float(thr), C.c_int64(P), _p(rois), _p(src)), "region_proposal")
    return rois[:n].copy(), src[:n].copy()


# -- target makers ------------------------------------------------------------------
def rpn_targets(anchor, gt, perm_pos=None, perm_neg=None, variant=0):
    """Returns (cls[N] i64, reg[N,4] f32, (n_pos, n_neg) before sampling)."""
    anchor, gt = _f32(anchor), _f32(gt)
    N, G = anchor.shape[0], gt.shape[0]
    cls = np.empty((N,), np.int64)
    reg = np.empty((N, 4), np.float32)
    counts = np.zeros((2,), np.int64)
    pp = None if perm_pos is None else _i64(perm_pos)
    pn = None if perm_neg is None else _i64(perm_neg)
    _chk(lib().orc_rpn_targets(C.c_int(variant), _p(anchor), C.c_int64(N), _p(gt), C.c_int64(G),
                               _p(pp), C.c_int64(0 if pp is None else len(pp)),
                               _p(pn), C.c_int64(0 if pn is None else len(pn)),
                               _p(cls), _p(reg), _p(counts)), "rpn_targets")
    return cls, reg, (int(counts[0]), int(counts[1]))


def head_target_counts(rois, gt, gt_label, variant=0):
    rois, gt, gl = _f32(rois).reshape(-1, 4), _f32(gt), _i64(gt_label)
    counts = np.zeros((2,), np.int64)
    _chk(lib().orc_head_targets(C.c_int(variant), _p(rois), C.c_int64(rois.shape[0]), _p(gt), _p(gl), C.c_int64(gt.shape[0]),
                                C.c_int64(0), C.c_int64(0), C.c_int64(0), None, C.c_int64(0), None, C.c_int64(0),
                                None, None, None, None, _p(counts)), "head_targets(counts)")
    return int(counts[0]), int(counts[1])


def head_targets(rois, gt, gt_label, perm_pos, perm_neg, variant=0, label_offset=1, max_pos=32, total=128):
    """Returns (cls[rows], reg[rows,4], sample_rois[rows,4], keep_index[rows])."""
    rois, gt, gl = _f32(rois).reshape(-1, 4), _f32(gt), _i64(gt_label)
    pp, pn = _i64(perm_pos), _i64(perm_neg)
    cls = np.empty((total,), np.int64)
    reg = np.empty((total, 4), np.float32)
    srois = np.empty((total, 4), np.float32)
    keep = np.empty((total,), np.int64)
    counts = np.zeros((2,), np.int64)
    rows = _chk(lib().orc_head_targets(C.c_int(variant), _p(rois), C.c_int64(rois.shape[0]), _p(gt), _p(gl),
                                       C.c_int64(gt.shape[0]), C.c_int64(label_offset), C.c_int64(max_pos), C.c_int64(total),
                                       _p(pp), C.c_int64(len(pp)), _p(pn), C.c_int64(len(pn)),
                                       _p(cls), _p(reg), _p(srois), _p(keep), _p(counts)), "head_targets")
    return cls[:rows].copy(), reg[:rows].copy(), srois[:rows].copy(), keep[:rows].copy()


# -- RoI pooling ----------------------------------------------------------------------
def roi_pool_fwd(feat, rois, PH=7, PW=7, scale=1.0):
    """feat [C,H,W] fp32 (single image), rois [R,4] -> (out [R,C,PH,PW], argmax int32)."""
    feat, rois = _f32(feat), _f32(rois).reshape(-1, 4)
    Cc, H, W = feat.shape
    R = rois.shape[0]
    out = np.empty((R, Cc, PH, PW), np.float32)
    arg = np.empty((R, Cc, PH, PW), np.int32)
    _chk(lib().orc_roi_pool_fwd(_p(feat), C.c_int(Cc), C.c_int(H), C.c_int(W), _p(rois), C.c_int64(R), C.c_int(PH), C.c_int(PW),
                                C.c_float(scale), _p(out), _p(arg)), "roi_pool_fwd")
    return out, arg


def roi_pool_bwd(grad_out, argmax, C_, H, W):
    g = _f32(grad_out)
    a = np.ascontiguousarray(argmax, dtype=np.int32)
    R, Cc, PH, PW = g.shape
    assert Cc == C_
    gf = np.empty((Cc, H, W), np.float32)
    _chk(lib().orc_roi_pool_bwd(_p(g), _p(a), C.c_int64(R), C.c_int(Cc), C.c_int(H), C.c_int(W), C.c_int(PH), C.c_int(PW), _p(gf)), "roi_pool_bwd")
    return gf


def roi_align_fwd(feat, rois, PH=7, PW=7, scale=1.0, sampling_ratio=2, aligned=False, level=None, sel=0, out=None):
    feat, rois = _f32(feat), _f32(rois).reshape(-1, 4)
    Cc, H, W = feat.shape
    R = rois.shape[0]
    if out is None:
        out = np.zeros((R, Cc, PH, PW), np.float32)
    lv = None if level is None else np.ascontiguousarray(level, dtype=np.int32)
    _chk(lib().orc_roi_align_fwd(_p(feat), C.c_int(Cc), C.c_int(H), C.c_int(W), _p(rois), C.c_int64(R), _p(lv), C.c_int(sel),
                                 C.c_int(PH), C.c_int(PW), C.c_float(scale), C.c_int(sampling_ratio), C.c_int(int(aligned)), _p(out)),
         "roi_align_fwd")
    return out


def roi_align_bwd(grad_out, feat_shape, rois, scale=1.0, sampling_ratio=2, aligned=False, level=None, sel=0):
    g, rois = _f32(grad_out), _f32(rois).reshape(-1, 4)
    R, Cc, PH, PW = g.shape
    _, H, W = feat_shape
    gf = np.zeros((Cc, H, W), np.float32)
    lv = None if level is None else np.ascontiguousarray(level, dtype=np.int32)
    _chk(lib().orc_roi_align_bwd(_p(g), C.c_int(Cc), C.c_int(H), C.c_int(W), _p(rois), C.c_int64(R), _p(lv), C.c_int(sel),
                                 C.c_int(PH), C.c_int(PW), C.c_float(scale), C.c_int(sampling_ratio), C.c_int(int(aligned)), _p(gf)),
         "roi_align_bwd")
    return gf


def roi_level_map(rois, k_min=2, k_max=5, s0=224.0, k0=4, eps=1e-6):
    rois = _f32(rois).reshape(-1, 4)
    out = np.empty((rois.shape[0],), np.int32)
    _chk(lib().orc_roi_level_map(_p(rois), C.c_int64(rois.shape[0]), C.c_int(k_min), C.c_int(k_max), C.c_float(s0), C.c_int(k0),
                                 C.c_float(eps), _p(out)), "roi_level_map")
    return out


def ms_roi_align(feats, rois, scales=(0.25, 0.125, 0.0625, 0.03125), PH=7, PW=7, sampling_ratio=2):
    """MultiScaleRoIAlign (models/new_model.py:127,143): feats = list of [C,H,W]; rois in image pixels."""
    rois = _f32(rois).reshape(-1, 4)
    k_min = int(round(-np.log2(scales[0])))
    k_max = int(round(-np.log2(scales[-1])))
    lv = roi_level_map(rois, k_min, k_max)
    out = np.zeros((rois.shape[0], feats[0].shape[0], PH, PW), np.float32)
    for l, (f, s) in enumerate(zip(feats, scales)):
        roi_align_fwd(f, rois, PH, PW, s, sampling_ratio, False, lv, l, out)
    return out, lv


# -- losses ------------------------------------------------------------------------------
def frcnn_loss(pred, target):
    rpn_cls, rpn_reg, head_cls, head_reg = [_f32(np.asarray(p)) for p in pred]
    t_rpn_cls, t_rpn_reg, t_cls, t_reg = target
    rpn_cls = rpn_cls.reshape(-1, 2)
    rpn_reg = rpn_reg.reshape(-1, 4)
    t_rpn_cls, t_cls = _i64(t_rpn_cls), _i64(t_cls)
    t_rpn_reg, t_reg = _f32(t_rpn_reg), _f32(t_reg)
    out = np.zeros((5,), np.float32)
    _chk(lib().orc_frcnn_loss(_p(rpn_cls), _p(rpn_reg), _p(t_rpn_cls), _p(t_rpn_reg), C.c_int64(rpn_cls.shape[0]),
                              _p(head_cls), _p(head_reg), _p(t_cls), _p(t_reg), C.c_int64(head_cls.shape[0]),
                              C.c_int(head_cls.shape[1]), _p(out)), "frcnn_loss")
    return out


# -- (f)3 input stage ----------------------------------------------------------------------
def preprocess_image(img_u8_hwc, out_hw, pad_hw=None, flip=False, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """hflip -> PIL-bilinear resize -> to_tensor -> normalize -> zero pad.  Returns (resized uint8 HWC, float32 CHW padded)."""
    img = np.ascontiguousarray(img_u8_hwc, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 3
    oh, ow = out_hw
    ph, pw = pad_hw if pad_hw is not None else out_hw
    u8 = np.empty((oh, ow, 3), np.uint8)
    out = np.empty((3, ph, pw), np.float32)
    m, s = _f32(mean), _f32(std)
    _chk(lib().orc_preprocess_image(_p(img), C.c_int(h), C.c_int(w), C.c_int(int(flip)), C.c_int(oh), C.c_int(ow), C.c_int(ph), C.c_int(pw),
                                    _p(m), _p(s), _p(u8), _p(out)), "preprocess_image")
    return u8, out


def preprocess_boxes(boxes, src_wh, out_wh, flip=False):
    b = _f32(boxes).reshape(-1, 4)
    out = np.empty_like(b)
    _chk(lib().orc_preprocess_boxes(_p(b), C.c_int64(b.shape[0]), C.c_int(src_wh[0]), C.c_int(src_wh[1]), C.c_int(int(flip)),
                                    C.c_int(out_wh[0]), C.c_int(out_wh[1]), _p(out)), "preprocess_boxes")
    return out
