#!/usr/bin/env python
"""Training-step timing of the ResNet-50-FPN mirror (BASELINE.json configs[3] and [4]) on ONE GPU: synthetic 800x1344
frames (COCO shape padded to /32), bs = 1, fp32 and bf16 autocast.  Not the bench line (bench.py measures configs[1]);
prints one JSON object with ms/step, images/s and the library's per-kernel HIP-event times for both precisions.

    python tools/fpn_bench.py --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_rcnn_pytorch_amd import _lib  # noqa: E402
from faster_rcnn_pytorch_amd.loss import FRCNNLoss  # noqa: E402
from faster_rcnn_pytorch_amd.new_model import FRCNN  # noqa: E402

H, W = 800, 1344


def frame(step):
    g = torch.Generator().manual_seed(7000 + step)
    x = torch.randn(1, 3, H, W, generator=g)
    G = int(torch.randint(1, 9, (1,), generator=g))
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    return x, torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1), torch.randint(1, 91, (G,), generator=g)


def run(amp, steps, warmup, dev):
    torch.manual_seed(0)
    model = FRCNN(num_classes=91, sampling="device").to(dev).train()
    crit = FRCNNLoss(None)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)
    frames = [tuple(t.to(dev) for t in frame(i)) for i in range(4)]

    def step(i):
        x, b, l = frames[i % len(frames)]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            pred, target = model(x, [b], [l])
        loss = crit(tuple(p.float() for p in pred), target)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    rep = _lib.prof_report()
    print("[fpn_bench] amp=%s %.2f ms/step" % (amp, dt / steps * 1e3), file=sys.stderr, flush=True)
    return {"ms_per_step": round(dt / steps * 1e3, 3), "images_per_s": round(steps / dt, 3), "final_loss": round(float(loss), 4),
            "hot_path_us_per_frame": {k: round(ms / steps * 1e3, 1) for k, (ms, n) in rep.items()},
            "hot_path_sum_us": round(sum(ms for ms, n in rep.values()) / steps * 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = {"workload": "ResNet-50-FPN Faster R-CNN train step, synthetic %dx%d, bs=1, 1 GPU, device sampling, fused SGD" % (H, W),
           "f32": run(False, a.steps, a.warmup, dev), "bf16_autocast": run(True, a.steps, a.warmup, dev)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
