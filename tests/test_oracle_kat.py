"""Known-answer tests for the torchvision-resident ops the reference calls (nms, RoIPool,
MultiScaleRoIAlign, AnchorGenerator).  torchvision is not available -> "parity unpinned":
the oracle is checked against hand-derived answers and independent numpy brute force."""
import numpy as np
import pytest

from oracle import oracle as orc


# ---------------------------------------------------------------- NMS (models/model_.py:53)
def brute_nms(boxes, thr):
    boxes = boxes.astype(np.float32)
    n = len(boxes)
    alive = np.ones(n, bool)
    keep = []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        lt = np.maximum(boxes[i, :2], boxes[i + 1:, :2])
        rb = np.minimum(boxes[i, 2:], boxes[i + 1:, 2:])
        wh = np.maximum(rb - lt, np.float32(0))
        inter = wh[:, 0] * wh[:, 1]
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = inter / (area[i] + area[i + 1:] - inter)
        alive[i + 1:] &= ~(iou > np.float32(thr))
    return np.array(keep, np.int64)


def test_nms_hand_cases():
    # boxes 0 and 1: inter 0.5*1=0.5, union 1.5 -> IoU 1/3; box 2 disjoint
    b = np.array([[0, 0, 1, 1], [0.5, 0, 1.5, 1], [2, 2, 3, 3]], np.float32)
    assert list(orc.nms(b, 0.3)) == [0, 2]
    assert list(orc.nms(b, 0.34)) == [0, 1, 2]
    # threshold is strict: IoU exactly 0.5 (inter 1, union 2) is NOT suppressed at thr 0.5
    b = np.array([[0, 0, 2, 1], [1, 0, 3, 1]], np.float32)      # inter 1, union 3 -> 1/3
    b2 = np.array([[0, 0, 2, 1], [0, 0, 1, 1]], np.float32)     # inter 1, union 2 -> exactly 0.5
    assert list(orc.nms(b2, 0.5)) == [0, 1]
    assert list(orc.nms(b2, np.nextafter(np.float32(0.5), np.float32(0)))) == [0]
    assert list(orc.nms(b, 1 / 3 - 1e-3)) == [0]
    # chain: 0 suppresses 1, so 1 cannot suppress 2
    c = np.array([[0, 0, 1, 1], [0.2, 0, 1.2, 1], [0.4, 0, 1.4, 1]], np.float32)  # IoU(0,1)=.667 IoU(1,2)=.667 IoU(0,2)=.4286
    assert list(orc.nms(c, 0.5)) == [0, 2]
    # zero-area boxes: 0/0 = NaN > thr is False -> never suppressed
    z = np.array([[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5]], np.float32)
    assert list(orc.nms(z, 0.1)) == [0, 1]
    assert list(orc.nms(np.zeros((0, 4), np.float32), 0.5)) == []


def test_nms_order_argument():
    b = np.array([[0, 0, 1, 1], [0.1, 0, 1.1, 1], [2, 2, 3, 3]], np.float32)
    assert list(orc.nms(b, 0.5, order=[1, 0, 2])) == [1, 2]


@pytest.mark.parametrize("seed", range(6))
def test_nms_vs_brute_force(seed):
    rng = np.random.RandomState(seed)
    n = 700
    c = rng.rand(n, 2).astype(np.float32) * 0.6 + 0.2
    wh = (rng.rand(n, 2).astype(np.float32) * 0.3 + 0.02)
    b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    for thr in (0.3, 0.7):
        assert np.array_equal(orc.nms(b, thr), brute_nms(b, thr))


# ---------------------------------------------------------------- RoIPool (models/model_.py:97,113)
def test_roi_pool_ramp_known_answer():
    # 1 channel 4x4 ramp 0..15; RoI covering the whole map, 2x2 output:
    # start=0,end=3 -> roi 4x4, bin 2x2 -> maxima 5,7,13,15 at flat indices 5,7,13,15
    f = np.arange(16, dtype=np.float32).reshape(1, 4, 4)
    out, arg = orc.roi_pool_fwd(f, [[0, 0, 3, 3]], 2, 2, 1.0)
    assert out.reshape(-1).tolist() == [5, 7, 13, 15]
    assert arg.reshape(-1).tolist() == [5, 7, 13, 15]
    # 7x7 output from a 4x4 RoI: bins overlap; bin(ph) = [floor(ph*4/7), ceil((ph+1)*4/7))
    out, arg = orc.roi_pool_fwd(f, [[0, 0, 3, 3]], 7, 7, 1.0)
    hs = [int(np.floor(p * 4 / 7)) for p in range(7)]
    he = [int(np.ceil((p + 1) * 4 / 7)) for p in range(7)]
    exp = np.array([[f[0, hs[i]:he[i], hs[j]:he[j]].max() for j in range(7)] for i in range(7)])
    assert np.array_equal(out[0, 0], exp)
    # rounding is half away from zero: 0.5 -> 1, 2.5 -> 3  => RoI rows/cols 1..3
    out, arg = orc.roi_pool_fwd(f, [[0.5, 0.5, 2.5, 2.5]], 1, 1, 1.0)
    assert out.item() == 15 and arg.item() == 15
    out, _ = orc.roi_pool_fwd(f, [[0.49, 0.49, 2.49, 2.49]], 1, 1, 1.0)    # -> 0..2
    assert out.item() == 10
    # RoI fully outside the map: all bins empty -> 0, argmax -1
    out, arg = orc.roi_pool_fwd(f, [[10, 10, 12, 12]], 2, 2, 1.0)
    assert (out == 0).all() and (arg == -1).all()
    # first strict maximum wins on ties
    out, arg = orc.roi_pool_fwd(np.ones((1, 4, 4), np.float32), [[0, 0, 3, 3]], 1, 1, 1.0)
    assert out.item() == 1 and arg.item() == 0
    # spatial_scale
    out, _ = orc.roi_pool_fwd(f, [[0, 0, 6, 6]], 1, 1, 0.5)
    assert out.item() == 15


def test_roi_pool_bwd_scatter():
    rng = np.random.RandomState(0)
    f = rng.randn(3, 9, 11).astype(np.float32)
    rois = np.array([[0, 0, 10, 8], [2.2, 1.7, 7.9, 6.1], [5, 5, 5, 5]], np.float32)
    out, arg = orc.roi_pool_fwd(f, rois, 7, 7, 1.0)
    go = rng.randn(*out.shape).astype(np.float32)
    gf = orc.roi_pool_bwd(go, arg, 3, 9, 11)
    exp = np.zeros((3, 99), np.float64)
    for r in range(3):
        for c in range(3):
            for p in range(49):
                a = arg[r, c].reshape(-1)[p]
                if a >= 0:
                    exp[c, a] += go[r, c].reshape(-1)[p]
    assert np.allclose(gf.reshape(3, -1), exp, atol=1e-5)


# ---------------------------------------------------------------- RoIAlign (models/new_model.py:127,143)
def np_bilinear(pl, y, x):
    H, W = pl.shape
    if y < -1 or y > H or x < -1 or x > W:
        return 0.0
    y = max(y, 0.0)
    x = max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    return (1 - ly) * (1 - lx) * pl[yl, xl] + (1 - ly) * lx * pl[yl, xh] + ly * (1 - lx) * pl[yh, xl] + ly * lx * pl[yh, xh]


def np_roi_align(f, roi, PH, PW, scale, sr):
    C, H, W = f.shape
    sw, sh, ew, eh = [v * scale for v in roi]
    rw, rh = max(ew - sw, 1.0), max(eh - sh, 1.0)
    bh, bw = rh / PH, rw / PW
    gh = sr if sr > 0 else int(np.ceil(rh / PH))
    gw = sr if sr > 0 else int(np.ceil(rw / PW))
    out = np.zeros((C, PH, PW))
    for c in range(C):
        for ph in range(PH):
            for pw in range(PW):
                acc = 0.0
                for iy in range(gh):
                    y = sh + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        x = sw + pw * bw + (ix + 0.5) * bw / gw
                        acc += np_bilinear(f[c].astype(np.float64), y, x)
                out[c, ph, pw] = acc / max(gh * gw, 1)
    return out


def test_roi_align_linear_ramp_exact():
    # f(y,x) = 2x + 3y is reproduced exactly by bilinear interpolation; the mean of the 2x2
    # samples of a bin is f at the bin centre (aligned=False: no half-pixel shift)
    H, W = 12, 16
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    f = (2 * xx + 3 * yy).astype(np.float32)[None]
    roi = np.array([[2.0, 1.0, 9.0, 8.0]], np.float32)               # 7x7 px -> bins of 1 px
    out = orc.roi_align_fwd(f, roi, 7, 7, 1.0, 2)
    cy = 1.0 + np.arange(7) + 0.5
    cx = 2.0 + np.arange(7) + 0.5
    exp = 3 * cy[:, None] + 2 * cx[None, :]
    assert np.allclose(out[0, 0], exp, atol=1e-4)


@pytest.mark.parametrize("sr", [2, 0])
def test_roi_align_vs_numpy(sr):
    rng = np.random.RandomState(1)
    f = rng.randn(2, 10, 13).astype(np.float32)
    rois = np.array([[0, 0, 50, 38], [-8, -6, 20, 15], [30, 20, 60, 50], [10, 10, 10.5, 10.2], [49, 37, 80, 70]], np.float32)
    out = orc.roi_align_fwd(f, rois, 7, 7, 0.25, sr)
    for r in range(len(rois)):
        assert np.allclose(out[r], np_roi_align(f, rois[r], 7, 7, 0.25, sr), atol=2e-5)


def test_roi_align_bwd_is_adjoint_of_fwd():
    rng = np.random.RandomState(2)
    f = rng.randn(2, 10, 13).astype(np.float32)
    rois = np.array([[0, 0, 50, 38], [-8, -6, 20, 15], [30, 20, 60, 50]], np.float32)
    out = orc.roi_align_fwd(f, rois, 7, 7, 0.25, 2)
    go = rng.randn(*out.shape).astype(np.float32)
    gf = orc.roi_align_bwd(go, f.shape, rois, 0.25, 2)
    # <fwd(f), go> == <f, bwd(go)> because fwd is linear in f
    assert abs(float((out.astype(np.float64) * go).sum()) - float((f.astype(np.float64) * gf).sum())) < 1e-3


def test_level_mapper_known_answers():
    def sq(s):
        return [0, 0, s, s]
    rois = np.array([sq(50), sq(111.9), sq(112), sq(223.9), sq(224), sq(447.9), sq(448), sq(895), sq(896), sq(2000), sq(0)], np.float32)
    lv = orc.roi_level_map(rois)          # k = floor(4 + log2(s/224) + 1e-6) in [2,5], minus 2
    assert lv.tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 3, 3, 0]


def test_ms_roi_align_routes_by_level():
    rng = np.random.RandomState(3)
    feats = [rng.randn(2, 64 >> l, 96 >> l).astype(np.float32) for l in range(4)]
    rois = np.array([[10, 10, 60, 70], [0, 0, 200, 240], [20, 30, 380, 250], [5, 5, 17, 13]], np.float32)
    out, lv = orc.ms_roi_align(feats, rois)
    assert lv.tolist() == [0, 1, 2, 0]
    for r in range(4):
        s = 0.25 / (1 << lv[r])
        assert np.allclose(out[r], np_roi_align(feats[lv[r]], rois[r], 7, 7, s, 2), atol=2e-5)


# ---------------------------------------------------------------- AnchorGenerator (models/new_model.py:23-25,46-47)
def test_tv_anchor_generator_known_answers():
    # size 32, ratios (0.5,1,2): h=32*sqrt(r), w=32/sqrt(r), round(+-w/2, +-h/2)
    b = orc.tv_base_anchors(32.0)
    assert b.tolist() == [[-23, -11, 23, 11], [-16, -16, 16, 16], [-11, -23, 11, 23]]
    assert orc.tv_base_anchors(512.0).tolist() == [[-362, -181, 362, 181], [-256, -256, 256, 256], [-181, -362, 181, 362]]
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    a = orc.tv_anchor_grid(800, 1344, shapes, normalise=False)
    assert a.shape == (268569, 4)                                     # N at 800x1344
    assert a[0].tolist() == [-23, -11, 23, 11]
    assert a[3].tolist() == [4 - 23, -11, 4 + 23, 11]                 # x-minor, stride 4
    assert a[336 * 3].tolist() == [-23, 4 - 11, 23, 4 + 11]           # next row
    last = a[-1]                                                      # level 5: strides (800//13, 1344//21) = (61, 64)
    assert last.tolist() == [20 * 64 - 181, 12 * 61 - 362, 20 * 64 + 181, 12 * 61 + 362]
    n = orc.tv_anchor_grid(800, 1344, shapes, normalise=True)
    assert np.array_equal(n, a / np.array([1344, 800, 1344, 800], np.float32))


# ---------------------------------------------------------------- head targets (models/model_.py:127-179)
def test_head_targets_restated_in_numpy():
    rng = np.random.RandomState(5)
    gt = np.array([[0.1, 0.1, 0.5, 0.6], [0.4, 0.3, 0.9, 0.9]], np.float32)
    lab = np.array([11, 14], np.int64)
    c = rng.rand(300, 2) * 0.8 + 0.1
    wh = rng.rand(300, 2) * 0.5 + 0.05
    rois = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)
    rois[:10] = gt[0] + rng.randn(10, 4).astype(np.float32) * 0.01
    npc, nnc = orc.head_target_counts(rois, gt, lab)
    allr = np.concatenate([rois, gt])
    iou = orc.pairwise_iou(allr, gt, 1e-5)
    mx, am = iou.max(1), iou.argmax(1)
    assert npc == int((mx >= 0.5).sum()) and nnc == int(((mx < 0.5) & (mx >= 0)).sum())
    pp, pn = rng.permutation(npc), rng.permutation(nnc)
    cls, reg, srois, keep = orc.head_targets(rois, gt, lab, pp, pn)
    n_pos = min(npc, 32)
    pos_idx = np.nonzero(mx >= 0.5)[0][pp[:n_pos]]
    neg_idx = np.nonzero((mx < 0.5) & (mx >= 0))[0][pn[:128 - n_pos]]
    ki = np.concatenate([pos_idx, neg_idx])
    assert np.array_equal(keep, ki) and len(cls) == 128
    ecls = lab[am][ki] + 1
    ecls[n_pos:] = 0
    assert np.array_equal(cls, ecls) and np.array_equal(srois, allr[ki])
    e = orc.encode(orc.xy_to_cxcy(gt[am][ki]), orc.xy_to_cxcy(allr[ki])) / np.array([0.1, 0.1, 0.2, 0.2], np.float32)
    assert np.allclose(reg, e, atol=1e-6)
    assert keep[:n_pos].max() >= 0 and (300 in keep or 301 in keep or npc > 32)   # gt rows are candidates too


def test_nms_nan_inf_boxes_kat():
    """torchvision semantics for non-finite boxes: a NaN coordinate -> NaN area -> every IoU NaN -> `> thr` false:
    the box is kept and suppresses nothing.  Hand case + the numpy brute force above."""
    nan = np.nan
    b = np.array([[0, 0, 1, 1], [0, 0, 1, nan], [0.05, 0, 1.05, 1], [nan, nan, nan, nan], [0, 0, 1, np.inf], [0.01, 0, 1.01, 1]], np.float32)
    # 0 kept; 1 NaN kept; 2 suppressed by 0 (IoU 0.905); 3 NaN kept; 4 area inf: IoU with 0 = 1/inf = 0 -> kept, suppresses nothing; 5 killed by 0
    assert orc.nms(b, 0.5).tolist() == [0, 1, 3, 4]
    rng = np.random.RandomState(4)
    bb = (rng.rand(400, 4) * 0.5).astype(np.float32)
    bb[:, 2:] += bb[:, :2]
    bb[rng.choice(400, 60, replace=False), rng.randint(0, 4, 60)] = nan
    bb[rng.choice(400, 20, replace=False), 2] = np.inf
    with np.errstate(invalid="ignore"):
        assert np.array_equal(orc.nms(bb, 0.4), brute_nms(bb, 0.4))


# ---------------------------------------------------------------- FRCNN.predict post-processing (models/model.py:368-402)
def test_predict_post_known_answer():
    """Hand-derived: 3 RoIs, background + 2 classes, threshold 0.05.
    class 1: RoI 0 (0.60) and RoI 1 (0.70) overlap with IoU 0.82 > 0.3 -> RoI 1 wins; RoI 2 (0.20) is disjoint -> kept after it.
    class 2: RoI 0 (0.03) is under the threshold; RoI 2 (0.50) then RoI 1 (0.10) are disjoint -> both kept, score order.
    Output is class-major with labels l - 1."""
    from oracle import model_ref
    rois = np.array([[0.10, 0.10, 0.50, 0.50], [0.12, 0.12, 0.52, 0.52], [0.60, 0.60, 0.90, 0.90]], np.float32)
    raw = np.repeat(rois[:, None, :], 3, axis=1).reshape(3, 12)                 # every class predicts its RoI unchanged
    prob = np.array([[0.37, 0.60, 0.03], [0.20, 0.70, 0.10], [0.30, 0.20, 0.50]], np.float32)
    b, l, s = model_ref.ref_suppress(raw, prob, 3, 0.05)
    assert l.tolist() == [0, 0, 1, 1] and l.dtype == np.int32 and b.dtype == np.float32 and s.dtype == np.float32
    assert np.array_equal(s, np.array([0.70, 0.20, 0.50, 0.10], np.float32))
    assert np.array_equal(b, rois[[1, 2, 2, 1]])
    # zero deltas decode to the RoI itself (up to the rounding of xy -> cxcy -> xy); softmax of equal logits is uniform
    hb, hl, hs, pred, p = model_ref.ref_predict_post(np.zeros((3, 3), np.float32), np.zeros((3, 12), np.float32), rois, 3, 0.05)
    assert np.allclose(p, 1 / 3) and np.abs(pred.reshape(3, 3, 4) - rois[:, None, :]).max() < 1e-6
    assert hl.tolist() == [0, 0, 1, 1] and np.allclose(hs, 1 / 3)               # ties: ascending RoI index, RoI 1 falls to RoI 0
    assert np.abs(hb - rois[[0, 2, 0, 2]]).max() < 1e-6
    # a delta of log(2) on w doubles the width around the same centre; x is clamped to [0, 1]
    reg = np.zeros((1, 12), np.float32)
    reg[0, 4 + 2] = np.log(2.0) / 0.2                                           # class 1, dw, un-normalised by 0.2
    _, _, _, pred, _ = model_ref.ref_predict_post(np.zeros((1, 3), np.float32), reg, np.array([[0.6, 0.2, 1.0, 0.4]], np.float32), 3, 0.05)
    assert np.abs(pred.reshape(3, 4)[1] - np.array([0.4, 0.2, 1.0, 0.4], np.float32)).max() < 1e-6


def test_philox_reference_matches_random123_known_answers():
    """oracle/philox_ref.py (the restatement of csrc/frcnn_common.h: philox_first that the device-sampling parity tests are built on)
    against the Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors)."""
    from oracle import philox_ref as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(int(x) for x in P.philox4x32_10(np.array(ctr), key)) == out
    # vectorised form = element-wise form; first word; 64-bit seed / offset split
    idx = np.array([0, 1, 77, 2 ** 31 + 5])
    a = P.philox_first((7 << 32) | 3, (9 << 32) | 1, 2, idx)
    for i, x in zip(idx, a):
        assert int(x) == int(P.philox4x32_10(np.array([i, 2, 1, 9]), (3, 7))[0])
    perm = P.sampling_perm(1, 0, 0, np.arange(50) * 3)
    assert sorted(perm.tolist()) == list(range(50))
