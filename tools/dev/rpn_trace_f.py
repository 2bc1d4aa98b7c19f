import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import oracle as orc
from faster_rcnn_pytorch_amd import ops, _lib
rng = np.random.RandomState(2)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
anchor = orc.tv_anchor_grid(800, 1344, shapes, normalise=True)
def _gt(rng, G):
    c = rng.rand(G, 2) * 0.7 + 0.15
    wh = rng.rand(G, 2) * 0.52 + 0.08
    return np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)
T = lambda a: torch.from_numpy(a).cuda()
for G in (1, 8):
    gt = _gt(rng, G)
    ta, tg = T(anchor), T(gt)
    for k in range(5): cls, reg, counts = ops.rpn_targets(ta, tg, seed=1, offset=k, variant=1) if 'variant' in ops.rpn_targets.__code__.co_varnames else ops.rpn_targets(ta, tg, seed=1, offset=k)
    torch.cuda.synchronize()
    buf = np.zeros(16, np.uint64)
    _lib.lib.frcnn_rpn_trace_read(buf.ctypes.data_as(C.c_void_p))
    t = buf.astype(np.int64); us = lambda k: (t[k] - t[0]) / 100.0
    print("N", anchor.shape[0], "G", G, "counts", counts.cpu().tolist(), "| colmax done %.2f, barrier passed %.2f, labels done %.2f, hist flushed %.2f | last WG: ticket %.2f, counts/reset %.2f, end %.2f" % (us(1), us(2), us(7), us(3), us(4), us(5), us(6)))
