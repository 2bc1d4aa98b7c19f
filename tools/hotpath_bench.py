#!/usr/bin/env python
"""Micro-benchmark of the hand-written hot-path kernels alone (no backbone), SURVEY 8d protocol:
synthetic RPN outputs in two regimes (init-like / trained-like), config V (VGG 600x1000) shapes,
20 warm-up + N timed iterations, per-kernel average from HIP events on the launch stream."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faster_rcnn_pytorch_amd import _lib, ops  # noqa: E402
from faster_rcnn_pytorch_amd.anchor import FRCNNAnchorMaker  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--regime", default="both")
    ap.add_argument("--config", default="V", choices=["V", "F"], help="V: VGG16 600x1000; F: ResNet50-FPN 800x1344")
    args = ap.parse_args()
    if args.config == "F":
        return fpn(args)
    dev = "cuda:0"
    H, W, C = 600, 1000, 512
    am = FRCNNAnchorMaker()
    anchors = am.device_anchors((H, W), dev)
    N = anchors.shape[0]
    fh, fw = H // 16, W // 16
    res = {}
    for regime in (["init", "trained"] if args.regime == "both" else [args.regime]):
        rng = np.random.RandomState(0)
        if regime == "init":
            reg = (rng.randn(N, 4) * 0.02).astype(np.float32)
            cls = (rng.randn(N, 2) * 0.02).astype(np.float32)
        else:
            reg = (rng.randn(N, 4) * np.array([0.1, 0.1, 0.2, 0.2])).astype(np.float32)
            cls = np.stack([np.zeros(N, np.float32), (rng.randn(N) * 2 - 2).astype(np.float32)], 1)
        reg, cls = torch.from_numpy(reg).to(dev), torch.from_numpy(cls).to(dev)
        G = 6
        c = rng.rand(G, 2) * 0.7 + 0.15
        wh = rng.rand(G, 2) * 0.52 + 0.08
        gt = torch.from_numpy(np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)).to(dev)
        lab = torch.from_numpy(rng.randint(0, 20, G).astype(np.int64)).to(dev)
        feat = torch.randn(1, C, fh, fw, device=dev, requires_grad=True)
        scale = torch.tensor([fw, fh, fw, fh], dtype=torch.float32, device=dev)
        grid = am.grid_desc((H, W))

        def one(i):
            rois, cnt, _ = ops.region_proposal(reg, cls, None, 1 / 1000, 12000, 0.7, 2000, grid=grid)
            ops.rpn_targets(anchors, gt, seed=1, offset=i)
            tc, tr, srois, _, _ = ops.head_targets(rois, gt, lab, n_rois=cnt, seed=1, offset=i)
            out = ops.roi_pool(feat, srois * scale, (7, 7), 1.0)
            out.backward(out)
            return cnt
        for i in range(20):
            cnt = one(i)
        torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(True)
        for i in range(args.iters):
            one(20 + i)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        rep = _lib.prof_report()
        res[regime] = {"n_rois": int(cnt.item()), "kernels_us_per_frame": {k: round(ms / args.iters * 1e3, 2) for k, (ms, n) in rep.items()},
                       "sum_us": round(sum(ms / args.iters * 1e3 for ms, n in rep.values()), 1)}
    print(json.dumps(res, indent=1))


def fpn(args):
    """Config F: 5 levels at 800x1344, N = 268 569, K/P = 4000/1000, 512 RoIs, C = 256, MultiScaleRoIAlign."""
    dev = "cuda:0"
    H, W, C = 800, 1344, 256
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    ag = ops.AnchorGenerator()
    anchors = ag.grid((H, W), shapes, dev, normalise=True)
    N = anchors.shape[0]
    rng = np.random.RandomState(0)
    reg = torch.from_numpy((rng.randn(N, 4) * np.array([0.1, 0.1, 0.2, 0.2])).astype(np.float32)).to(dev)
    cls = torch.from_numpy(np.stack([np.zeros(N, np.float32), (rng.randn(N) * 2 - 2).astype(np.float32)], 1)).to(dev)
    G = 6
    c = rng.rand(G, 2) * 0.7 + 0.15
    wh = rng.rand(G, 2) * 0.52 + 0.08
    gt = torch.from_numpy(np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.randint(1, 91, G).astype(np.int64)).to(dev)
    feats = [torch.randn(1, C, h, w, device=dev, requires_grad=True) for h, w in shapes[:4]]
    scale = torch.tensor([W, H, W, H], dtype=torch.float32, device=dev)

    def one(i):
        rois, cnt, _ = ops.region_proposal(reg, cls, anchors, 10 / 1000, 4000, 0.7, 1000)
        ops.rpn_targets(anchors, gt, variant=1, seed=1, offset=i)
        tc, tr, srois, _, _ = ops.head_targets(rois, gt, lab, n_rois=cnt, variant=1, label_offset=0, max_pos=128, total=512, seed=1, offset=i)
        out = ops.ms_roi_align(feats, srois * scale, 7, 2)
        out.backward(out)
        return cnt
    for i in range(5):
        cnt = one(i)
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    for i in range(args.iters):
        one(5 + i)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    rep = _lib.prof_report()
    print(json.dumps({"config": "F", "n_rois": int(cnt.item()), "kernels_us_per_frame": {k: round(ms / args.iters * 1e3, 2) for k, (ms, n) in rep.items()},
                      "sum_us": round(sum(ms / args.iters * 1e3 for ms, n in rep.values()), 1)}, indent=1))


if __name__ == "__main__":
    main()
