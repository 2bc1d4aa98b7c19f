"""Debug: RoIAlign backward at config F ('spread' RoIs) vs the C oracle -- where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops
from oracle import oracle as orc
orc.build()
rng = np.random.RandomState(11)
C, R, Wimg, Himg = 256, 512, 1344, 800
shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
feats = [rng.randn(C, h, w).astype(np.float32) for h, w in shapes]
c = rng.rand(R, 2).astype(np.float32); wh = (rng.rand(R, 2) * 0.97 + 0.01).astype(np.float32)
rois = np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32) * np.array([Wimg, Himg, Wimg, Himg], np.float32)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out_o, lv = orc.ms_roi_align(feats, rois)
fts = [T(f[None]).requires_grad_(True) for f in feats]
m = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
out = m({str(i): f for i, f in enumerate(fts)}, [T(rois)], [(Wimg, Himg)])
go = rng.randn(*out_o.shape).astype(np.float32)
out.backward(T(go))
for l, f in enumerate(fts):
    gf_o = orc.roi_align_bwd(go, feats[l].shape, rois, 0.25 / (1 << l), 2, False, lv, l)
    g = f.grad[0].cpu().numpy()
    bad = ~np.isclose(g, gf_o, rtol=1e-4, atol=1e-4 * np.abs(gf_o).max())
    H, W = shapes[l]
    tiles = np.zeros(((H + 15) // 16, (W + 7) // 8), int)
    ys, xs = np.nonzero(bad.any(0))
    for y, x in zip(ys, xs): tiles[y // 16, x // 8] += 1
    print("level", l, "RoIs", int((lv == l).sum()), "bad elements", int(bad.sum()), "of", bad.size, "bad channels", np.nonzero(bad.any((1, 2)))[0][:10], "bad tiles", int((tiles > 0).sum()), "of", tiles.size,
          "first bad tiles", list(zip(*np.nonzero(tiles)))[:6], "finite", bool(np.isfinite(g).all()))
l = 3
gf_o = orc.roi_align_bwd(go, feats[l].shape, rois, 0.25 / (1 << l), 2, False, lv, l)
g = fts[l].grad[0].cpu().numpy()
bad = ~np.isclose(g, gf_o, rtol=1e-4, atol=1e-4 * np.abs(gf_o).max())
cs, ys, xs = np.nonzero(bad)
for i in range(0, min(len(cs), 4000), 400):
    c_, y, x = cs[i], ys[i], xs[i]
    print("c %d y %d x %d  got %.6g want %.6g ratio %.4f | neighbour channel c^2: got %.6g want %.6g" % (c_, y, x, g[c_, y, x], gf_o[c_, y, x], g[c_, y, x] / gf_o[c_, y, x], g[c_ ^ 2, y, x], gf_o[c_ ^ 2, y, x]))
print("rows of bad elements in their tile (y % 16):", np.bincount(ys % 16, minlength=16), " cols (x % 8):", np.bincount(xs % 8, minlength=8))
