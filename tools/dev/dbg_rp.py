import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops
from oracle import oracle as orc
from test_gpu_ops import rpn_outputs, T
rng = np.random.RandomState(11)
H, W = 600, 1000
anchor = orc.anchor_grid(H, W); N = anchor.shape[0]
reg, cls = rpn_outputs(rng, N, "init")
K, P = 12000, 2000
grid = (H // 16, W // 16, 16, ops.anchor_base(), W, H)
tr, tc, ta = T(reg), T(cls), T(anchor)
ref = None
for it, kind in enumerate("aabbabab"):
    if kind == "a": rois, cnt, src = ops.region_proposal(tr, tc, ta, 1 / 1000, K, 0.7, P, want_src=True)
    else: rois, cnt, src = ops.region_proposal(tr, tc, None, 1 / 1000, K, 0.7, P, grid=grid, want_src=True)
    torch.cuda.synchronize()
    n = int(cnt.item())
    if ref is None: ref = src.clone()
    print(it, kind, "count", n, "src mismatches vs first call", int((src[:n] != ref[:n]).sum()))
# inspect the workspace of the last call: scores (N f32 at align256(N*16)), sidx (K i64 next)
dev = tr.device
ws = ops._workspace(dev, 0)
al = lambda x: (x + 255) // 256 * 256
o_scores = al(N * 16); o_sidx = o_scores + al(N * 4); o_ss = o_sidx + al(K * 8); o_sb = o_ss + al(K * 4)
scores = ws[o_scores:o_scores + N * 4].view(torch.float32)
sidx = ws[o_sidx:o_sidx + K * 8].view(torch.int64)
ref = torch.sort(scores, descending=True, stable=True).indices[:K]
nv = int((scores >= 0).sum())
m = min(nv, K)
print("valid", nv, "topk index mismatches", int((sidx[:m] != ref[:m]).sum()), "first bad", int(torch.nonzero(sidx[:m] != ref[:m])[0]) if (sidx[:m] != ref[:m]).any() else -1)
o_keep = o_sb + al(K * 16); o_ctrl = o_keep + al(K * 8); o_lvl = o_ctrl + 256; o_topk = o_lvl + al(K * 4)
ctl = ws[o_topk:o_topk + 2048 + 1024 + 1024 + 64]
split = ctl[:2048].view(torch.int64); cnt = ctl[2048:3072].view(torch.int32); cur = ctl[3072:4096].view(torch.int32); tail = ctl[4096:4160].view(torch.int32)
print("cnt sum", int(cnt.sum()), "cursor sum", int(cur.sum()), "n_valid", int(tail[0]), "barrier words", int(tail[1]), "max bucket", int(cnt.max()))
print("splitters descending:", bool((split[1:256].cpu().numpy().astype(np.uint64)[:-1] > split[1:256].cpu().numpy().astype(np.uint64)[1:]).all()))
