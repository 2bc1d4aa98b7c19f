"""Developer tool: time the NMS stage on the bench-like frame (12 000 boxes of an untrained RPN at 600x1000)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops, _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
DEV = "cuda:0"
g = torch.Generator().manual_seed(3)
K = int(os.environ.get("K", "12000"))
BENCH = os.environ.get("BENCH_BOXES")
# boxes like decoded anchors of an untrained RPN: 9 anchor shapes on a stride-16 grid, jittered, clipped; scores sorted descending
cx = torch.rand(K, generator=g) * 1000; cy = torch.rand(K, generator=g) * 600
sz = torch.tensor([128., 256., 512.])[torch.randint(0, 3, (K,), generator=g)]; ar = torch.tensor([0.5, 1., 2.])[torch.randint(0, 3, (K,), generator=g)]
w = sz * ar.sqrt() * torch.exp(torch.randn(K, generator=g) * 0.1); h = sz / ar.sqrt() * torch.exp(torch.randn(K, generator=g) * 0.1)
b = torch.stack([(cx - w / 2).clamp(0, 1000), (cy - h / 2).clamp(0, 600), (cx + w / 2).clamp(0, 1000), (cy + h / 2).clamp(0, 600)], 1).to(DEV)
s = torch.sort(torch.rand(K, generator=g), descending=True)[0].to(DEV)
if BENCH:                                  # real bench-frame boxes (tools/dev/dump_bench_boxes.py), already score-sorted
    import numpy as np
    bb = np.load(BENCH)[int(os.environ.get("FRAME", "0"))]
    b = torch.from_numpy(bb).to(DEV); K = b.shape[0]
    s = torch.linspace(1, 0, K).to(DEV)
POST = os.environ.get("POST_K")             # POST_K=2000: the proposal stage's call (boxes already sorted, outputs written by the NMS kernel itself)
def run():
    if POST:
        k, r, c = ops.nms_sorted(b, 0.7, post_k=int(POST), want_rois=True)
        return k[:int(c.item())] if False else k
    return ops.nms(b, s, 0.7)
for _ in range(20): keep = run()
torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
for _ in range(50): keep = run()
torch.cuda.synchronize(); _lib.prof_enable(False)
print(os.environ.get("FRCNN_HIP_LIB", "default"), "kept", int(keep.numel()), {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.prof_report().items()})
