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
    args = ap.parse_args()
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
        res[regime] = {"n_rois": int(cnt.item()), "kernels_us": {k: round(ms / n * 1e3, 2) for k, (ms, n) in rep.items()},
                       "sum_us": round(sum(ms / n * 1e3 for ms, n in rep.values()), 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
