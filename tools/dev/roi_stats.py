"""Developer tool: per-level RoI counts and per-tile RoI list lengths of the RoIAlign backward at the bench's FPN frames."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from faster_rcnn_pytorch_amd import ops
from faster_rcnn_pytorch_amd.new_model import FRCNN
cfg = bench.CONFIGS["fpn"]
dev = torch.device("cuda:0")
model = FRCNN(num_classes=cfg["num_classes"], sampling="device", seed=1234).to(dev)
cap = {}
orig = ops.ms_roi_align
def spy(feats, boxes, out, sr, scales, *a, **k):
    cap["rois"] = boxes[0].detach().float().cpu().numpy() if isinstance(boxes, (list, tuple)) else boxes.detach().float().cpu().numpy()
    cap["scales"] = scales; cap["shapes"] = [tuple(f.shape[-2:]) for f in feats]
    return orig(feats, boxes, out, sr, scales, *a, **k)
ops.ms_roi_align = spy
from faster_rcnn_pytorch_amd.loss import FRCNNLoss
crit = FRCNNLoss()
opt = torch.optim.SGD(model.parameters(), lr=2e-3, momentum=0.9, weight_decay=5e-4)
STEPS = int(os.environ.get("STEPS", "30"))
for step in range(STEPS):
    x, bbox, label = bench.synth_frame(cfg, 0, step % 8)
    model.train()
    pred, target = model(x.to(dev), [bbox.to(dev)], [label.to(dev)])
    loss = crit(pred, target)[0]
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    if step not in (0, 5, 15, STEPS - 1):
        continue
    r = cap["rois"]; sc = cap["scales"]; shp = cap["shapes"]
    w = r[:, 2] - r[:, 0]; h = r[:, 3] - r[:, 1]
    k = np.floor(4 + np.log2(np.sqrt(np.maximum(w * h, 1e-12)) / 224 + 1e-6)).clip(2, 5).astype(int) - 2
    print("step", step, "R", len(r), "per level", np.bincount(k, minlength=4), "scales", [round(float(s), 4) for s in sc], shp)
    for l in range(len(shp)):
        H, W = shp[l]; s = float(sc[l]); rr = r[k == l] * s
        ty, tx = (H + 15) // 16, (W + 7) // 8
        cnt = np.zeros((ty, tx), int)
        for b in rr:
            x0, y0, x1, y1 = b; x0 = int(np.floor(max(x0 - 1, 0))); y0 = int(np.floor(max(y0 - 1, 0))); x1 = int(min(np.ceil(x1 + 1), W - 1)); y1 = int(min(np.ceil(y1 + 1), H - 1))
            cnt[y0 // 16:y1 // 16 + 1, x0 // 8:x1 // 8 + 1] += 1
        print("  level", l, "tiles", ty * tx, "list len: max", cnt.max(), "mean", round(cnt.mean(), 1), "sum", cnt.sum(), "p90", int(np.percentile(cnt, 90)))
