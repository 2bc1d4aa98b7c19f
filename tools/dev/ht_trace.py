import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
from faster_rcnn_pytorch_amd import ops, _lib
g = torch.Generator().manual_seed(1)
for (P, G, total, maxpos, variant) in ((2000, 6, 128, 32, 0), (1000, 6, 512, 128, 1)):
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15; wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    gt = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1).cuda()
    lab = torch.randint(0, 20, (G,), generator=g).cuda()
    # proposals: jittered copies of the GT boxes + random boxes
    idx = torch.randint(0, G, (P,), generator=g)
    rois = (gt.cpu()[idx] + torch.randn(P, 4, generator=g) * 0.08).clamp(0, 1)
    rois = torch.cat([torch.minimum(rois[:, :2], rois[:, 2:]), torch.maximum(rois[:, :2], rois[:, 2:]) + 0.01], 1).clamp(0, 1).cuda()
    cnt = torch.tensor([P], dtype=torch.int32).cuda()
    for k in range(5):
        out = ops.head_targets(rois, gt, lab, n_rois=cnt, variant=variant, label_offset=1, max_pos=maxpos, total=total, seed=3, offset=k)
    torch.cuda.synchronize()
    buf = np.zeros(16, np.uint64)
    _lib.lib.frcnn_rpn_trace_read(buf.ctypes.data_as(C.c_void_p))
    t = buf.astype(np.int64); us = lambda k: (t[k] - t[8]) / 100.0
    print("P", P, "total", total, "| iou+compaction %.2f | pos: select %.2f..%.2f, compacted %.2f | neg: select %.2f..%.2f, compacted %.2f | ranked %.2f, rows written %.2f" % (us(9), us(10), us(11), us(12), us(13), us(14), us(15), us(6), us(7)))
