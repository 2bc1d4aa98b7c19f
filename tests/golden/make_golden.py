"""Generates tests/golden/*.npz by IMPORTING the reference's torch-only modules
(anchor.py, utils/util.py, losses/loss.py) from /root/reference in the build
container.  The reference itself never travels: only these input/output vectors
are committed.  Re-run:  python tests/golden/make_golden.py

What is NOT here (and why): models/model_.py / new_model.py / util/box_ops.py need
torchvision (+cv2), which is not installed -> "parity unpinned" for nms, RoIPool,
RoIAlign, AnchorGenerator (see oracle/frcnn_oracle.c header).
"""
import hashlib
import os
import sys

import numpy as np
import torch

REF = os.environ.get("FRCNN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
from anchor import FRCNNAnchorMaker  # noqa: E402
from utils.util import (cxcy_to_xy, decode, encode, find_jaccard_overlap,  # noqa: E402
                        xy_to_cxcy)
from losses.loss import FRCNNLoss  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rand_boxes(g, n, lo=0.02, hi=0.6):
    c = torch.rand(n, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(n, 2, generator=g) * (hi - lo) + lo
    return torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)


def main():
    torch.set_num_threads(1)
    am = FRCNNAnchorMaker()

    # ---- anchors (anchor.py:15-55) -------------------------------------------------
    d = {"anchor_base": am.anchor_base}
    for (h, w) in [(600, 1000), (800, 800), (800, 1344), (880, 960), (160, 240), (37, 50)]:
        a = am._enumerate_shifted_anchor((h, w))
        inside = int(((a[:, 0] >= 0) & (a[:, 1] >= 0) & (a[:, 2] <= 1) & (a[:, 3] <= 1)).sum())
        key = "%dx%d" % (h, w)
        d[key + "_shape"] = np.array(a.shape)
        d[key + "_inside"] = np.array(inside)
        d[key + "_sha256"] = np.array(sha(a))
        d[key + "_head"] = a[:128].copy()
        d[key + "_tail"] = a[-128:].copy()
        d[key + "_stride97"] = a[::97].copy()
        if (h, w) in [(160, 240), (37, 50)]:
            d[key + "_full"] = a
    np.savez_compressed(os.path.join(OUT, "anchors.npz"), **d)

    # ---- box codec + IoU (utils/util.py:15-102) ----------------------------------------
    g = torch.Generator().manual_seed(1234)
    xy = rand_boxes(g, 512)
    cx = xy_to_cxcy(xy)
    t = torch.randn(512, 4, generator=g) * torch.tensor([0.1, 0.1, 0.2, 0.2])
    t[:8, 2:] = torch.tensor([[-20.0, 20.0], [50.0, -50.0], [88.0, -88.0], [0.0, 0.0],
                              [1e-3, -1e-3], [5.0, -5.0], [10.0, -10.0], [3.3, -3.3]])
    anc = xy_to_cxcy(rand_boxes(g, 512))
    dec = decode(t, anc)
    gt = xy_to_cxcy(rand_boxes(g, 512))
    enc = encode(gt, anc)
    s1 = rand_boxes(g, 96)
    s2 = rand_boxes(g, 7)
    s1[0] = s2[0]                                   # identical boxes
    s1[1] = torch.tensor([0.3, 0.3, 0.3, 0.3])      # zero-area
    s2[1] = torch.tensor([0.3, 0.3, 0.3, 0.3])      # zero-area vs zero-area (eps keeps it finite)
    s1[2] = torch.tensor([0.0, 0.0, 1.0, 1.0])      # contains everything
    s1[3] = torch.tensor([0.9, 0.9, 0.95, 0.95])    # probably disjoint
    iou = find_jaccard_overlap(s1, s2)
    rmax, rarg = iou.max(dim=1)
    cmax, carg = iou.max(dim=0)
    logits = torch.randn(512, 2, generator=g) * 3
    logits[:4] = torch.tensor([[0.0, 0.0], [30.0, -30.0], [-30.0, 30.0], [100.0, 100.5]])
    fg = torch.softmax(logits, dim=-1)[..., 1]
    np.savez_compressed(os.path.join(OUT, "codec.npz"),
                        xy=xy.numpy(), xy_to_cxcy=cx.numpy(), cxcy_to_xy=cxcy_to_xy(cx).numpy(),
                        t=t.numpy(), anc_cxcy=anc.numpy(), decode=dec.numpy(),
                        gt_cxcy=gt.numpy(), encode=enc.numpy(),
                        s1=s1.numpy(), s2=s2.numpy(), jaccard=iou.numpy(),
                        row_max=rmax.numpy(), row_arg=rarg.numpy(), col_max=cmax.numpy(), col_arg=carg.numpy(),
                        logits=logits.numpy(), fg_softmax=fg.numpy())

    # ---- RegionProposal.forward body before NMS (models/model_.py:19-49), rebuilt from the
    #      reference's own functions; inputs are tie-free so the unstable sort is unambiguous ----
    H, W = 160, 240
    anchor = torch.from_numpy(am._enumerate_shifted_anchor((H, W)))
    N = anchor.shape[0]
    reg = torch.randn(N, 4, generator=g) * torch.tensor([0.1, 0.1, 0.2, 0.2])
    reg[::50, 2:] = -9.0                           # some boxes collapse below min_size
    cls = torch.randn(N, 2, generator=g) * 2
    score = torch.softmax(cls, dim=-1)[..., 1]
    assert torch.unique(score).numel() == N, "fixture must be tie-free"
    roi = cxcy_to_xy(decode(reg, xy_to_cxcy(anchor))).clamp(0, 1)
    ws = roi[:, 2] - roi[:, 0]
    hs = roi[:, 3] - roi[:, 1]
    min_size = 1
    keep = (hs >= (min_size / 1000)) & (ws >= (min_size / 1000))
    roi_k = roi[keep]
    sc_k = score[keep]
    ssc, sidx = sc_k.sort(descending=True)
    K = 600
    orig_idx = torch.arange(N)[keep][sidx[:K]]
    np.savez_compressed(os.path.join(OUT, "proposal_pre_nms.npz"),
                        H=H, W=W, reg=reg.numpy(), cls=cls.numpy(), anchor=anchor.numpy(),
                        roi_all=roi.numpy(), keep=keep.numpy(), score_all=score.numpy(),
                        K=K, top_roi=roi_k[sidx[:K]].numpy(), top_score=ssc[:K].numpy(),
                        top_orig_idx=orig_idx.numpy())

    # ---- RPN/head target building blocks on the reference's smoke boxes (models/model.py:413-416) ----
    boxes = torch.tensor([[79.8867, 286.8000, 329.7450, 444.0000],
                          [11.8980, 13.2000, 596.6006, 596.4000]]) / 800
    a800 = torch.from_numpy(am._enumerate_shifted_anchor((800, 800)))
    ak = (a800[:, 0] >= 0) & (a800[:, 1] >= 0) & (a800[:, 2] <= 1) & (a800[:, 3] <= 1)
    ain = a800[ak]
    iou = find_jaccard_overlap(ain, boxes)
    rmax, rarg = iou.max(dim=1)
    cmax, carg = iou.max(dim=0)
    label = -1 * torch.ones(ain.size(0))
    label[rmax < 0.3] = 0
    label[carg] = 1
    label[rmax >= 0.7] = 1
    tg = encode(xy_to_cxcy(boxes[rarg]), xy_to_cxcy(ain))
    np.savez_compressed(os.path.join(OUT, "smoke_targets.npz"),
                        boxes=boxes.numpy(), n_inside=int(ak.sum()), inside_idx=torch.arange(a800.size(0))[ak].numpy(),
                        row_max=rmax.numpy(), row_arg=rarg.numpy(), col_max=cmax.numpy(), col_arg=carg.numpy(),
                        label_pre_sample=label.numpy(), n_pos=int((label == 1).sum()), n_neg=int((label == 0).sum()),
                        tg=tg.numpy())

    # ---- losses (losses/loss.py:5-85) ---------------------------------------------------
    crit = FRCNNLoss(None)
    N, R, NC = 2000, 128, 21
    p_rc = torch.randn(1, N, 2, generator=g)
    p_rr = torch.randn(1, N, 4, generator=g) * 0.3
    p_hc = torch.randn(R, NC, generator=g)
    p_hr = torch.randn(R, 4, generator=g) * 0.5
    t_rc = torch.full((N,), -1, dtype=torch.long)
    idx = torch.randperm(N, generator=g)
    t_rc[idx[:200]] = 0
    t_rc[idx[200:256]] = 1
    t_rr = torch.randn(N, 4, generator=g) * 0.3
    t_hc = torch.zeros(R, dtype=torch.long)
    t_hc[:32] = torch.randint(1, NC, (32,), generator=g)
    t_hr = torch.randn(R, 4, generator=g)
    out = crit((p_rc, p_rr, p_hc, p_hr), (t_rc, t_rr, t_hc, t_hr))
    np.savez_compressed(os.path.join(OUT, "loss.npz"),
                        p_rpn_cls=p_rc.numpy(), p_rpn_reg=p_rr.numpy(), p_head_cls=p_hc.numpy(), p_head_reg=p_hr.numpy(),
                        t_rpn_cls=t_rc.numpy(), t_rpn_reg=t_rr.numpy(), t_head_cls=t_hc.numpy(), t_head_reg=t_hr.numpy(),
                        losses=np.array([float(o) for o in out], dtype=np.float32))
    make_full_size_proposal_goldens(am)
    print("golden vectors written to", OUT)


# ---- FULL-SIZE pre-NMS proposal bodies (VERDICT r1 "next" 1a): models/model_.py:19-49 at 600x1000 (N = 20 646,
#      K = 12 000) and models/new_model.py:49-76 at 800x1344 (N = 268 569, K = 4 000), executed with the REFERENCE's own
#      decode / xy_to_cxcy / cxcy_to_xy + torch.softmax / clamp / sort, in both synthetic RPN-output regimes of SURVEY 8d.
#      Inputs are not stored: tests regenerate them from the seed (torch CPU generator) and check their sha256.
FULL_CASES = [
    # name, (H, W), seed, regime, K, min_size (model_.py:15 -> 1; new_model.py:21 -> 10)
    ("V_init", (600, 1000), 101, "init", 12000, 1),
    ("V_trained", (600, 1000), 102, "trained", 12000, 1),
    ("V_trained_test", (600, 1000), 105, "trained", 6000, 1),
    ("F_init", (800, 1344), 104, "init", 4000, 10),
    ("F_trained", (800, 1344), 103, "trained", 4000, 10),
]
FPN_SHAPES = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]


def full_case_inputs(seed, regime, N):
    """SURVEY 8d regimes.  init: what normal_init(., 0.01) heads produce; trained: fg-bg logit ~ N(-2, 2), deltas N(0, .1/.2)."""
    g = torch.Generator().manual_seed(seed)
    if regime == "init":
        reg = torch.randn(N, 4, generator=g) * 0.02
        cls = torch.randn(N, 2, generator=g) * 0.02
    else:
        reg = torch.randn(N, 4, generator=g) * torch.tensor([0.1, 0.1, 0.2, 0.2])
        d = torch.randn(N, generator=g) * 2 - 2
        cls = torch.stack([torch.zeros(N), d], 1)
    return reg.contiguous(), cls.contiguous()


def make_full_size_proposal_goldens(am):
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import oracle as orc      # ONLY for the FPN anchor grid: torchvision's AnchorGenerator is not importable
    d = {}
    for name, (H, W), seed, regime, K, min_size in FULL_CASES:
        if name.startswith("V"):
            anchor = torch.from_numpy(am._enumerate_shifted_anchor((H, W)))                 # anchor.py:34-55 (reference)
        else:
            anchor = torch.from_numpy(orc.tv_anchor_grid(H, W, FPN_SHAPES, normalise=True))  # input of this fixture, not pinned by it
        N = anchor.shape[0]
        reg, cls = full_case_inputs(seed, regime, N)
        score = torch.softmax(cls, dim=-1)[..., 1]                                          # model_.py:20
        roi = cxcy_to_xy(decode(reg, xy_to_cxcy(anchor))).clamp(0, 1)                       # model_.py:31-33
        ws = roi[:, 2] - roi[:, 0]
        hs = roi[:, 3] - roi[:, 1]
        keep = (hs >= (min_size / 1000)) & (ws >= (min_size / 1000))                        # model_.py:36-38
        roi_k, sc_k = roi[keep], score[keep]
        ssc, sidx = sc_k.sort(descending=True)                                              # model_.py:44 (unstable; ties: see tests)
        k = min(K, len(sidx))
        orig_idx = torch.arange(N)[keep][sidx[:k]]
        d[name + "_meta"] = np.array([H, W, seed, K, min_size, N, int(keep.sum()), k], np.int64)
        d[name + "_regime"] = np.array(regime)
        d[name + "_sha_reg"] = np.array(sha(reg.numpy()))
        d[name + "_sha_cls"] = np.array(sha(cls.numpy()))
        d[name + "_sha_anchor"] = np.array(sha(anchor.numpy()))
        d[name + "_sha_keep"] = np.array(sha(keep.numpy()))
        d[name + "_top_idx"] = orig_idx.numpy().astype(np.int32)
        d[name + "_top_score"] = ssc[:k].numpy()
        d[name + "_top_roi_s16"] = roi_k[sidx[:k]][::16].numpy()
        d[name + "_roi_all_s101"] = roi[::101].numpy()
        d[name + "_score_all_s101"] = score[::101].numpy()
        d[name + "_n_distinct_top_scores"] = np.array(int(torch.unique(ssc[:k]).numel()))
        print(name, "N", N, "kept", int(keep.sum()), "k", k, "distinct top scores", int(torch.unique(ssc[:k]).numel()))
    np.savez_compressed(os.path.join(OUT, "proposal_full.npz"), **d)


if __name__ == "__main__":
    main()
