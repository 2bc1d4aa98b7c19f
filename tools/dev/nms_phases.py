"""Developer tool: phase times of nms_resolve_kernel (needs a library built with -DNMS_DEBUG, loaded through FRCNN_HIP_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops
from oracle import oracle as orc
dev = "cuda:0"
anchor = orc.anchor_grid(600, 1000); N = len(anchor)
for name in ("init", "trained"):
    rng = np.random.RandomState(0)
    if name == "init":
        reg = (rng.randn(N, 4) * 0.02).astype(np.float32); cls = (rng.randn(N, 2) * 0.02).astype(np.float32)
    else:
        reg = (rng.randn(N, 4) * np.array([0.1, 0.1, 0.2, 0.2])).astype(np.float32)
        cls = np.stack([np.zeros(N, np.float32), (rng.randn(N) * 2 - 2).astype(np.float32)], 1)
    boxes, scores, nv = orc.proposal_prologue(reg, cls, anchor, 1 / 1000)
    idx, sc = orc.topk_sorted(scores, 12000)
    b = torch.from_numpy(boxes[idx]).to(dev)
    for it in range(3):
        keep, _, cnt = ops.nms_sorted(b, 0.7, post_k=2000)
    torch.cuda.synchronize()
    K, nblk = 12000, 188
    ws = ops._WS[(0, torch.cuda.current_stream().cuda_stream)]
    import math
    al = lambda v: (v + 255) // 256 * 256
    off = al(K * nblk * 8 + 256) + al(K * 3 * 8) + al((nblk + 1) * 8) * 2
    d = ws[off:off + 32].view(torch.int32).cpu().tolist()
    print(name, "cnt", int(cnt), "abort", d[0], "max iterations of a wave", d[2], "wave-iterations total", d[3], "longest wave %.1f us" % (d[4] / 100))
