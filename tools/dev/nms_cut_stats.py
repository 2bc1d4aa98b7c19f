"""Developer tool: where in the score order does the post_k-th kept box of the proposal NMS lie?  (bench frames of the VGG mirror at random init and after
STEPS SGD steps; synthetic anchor-like boxes.)  Everything below that rank cannot change the stage's outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from faster_rcnn_pytorch_amd import ops
from faster_rcnn_pytorch_amd.model import FRCNN
from faster_rcnn_pytorch_amd.loss import FRCNNLoss
dev = "cuda:0"
torch.manual_seed(0)
m = FRCNN(num_classes=21, sampling="device", seed=1234).to(dev)
cfg = bench.CONFIGS["vgg"]
crit = FRCNNLoss()
opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
def stats(tag):
    with torch.no_grad():
        for i in range(3):
            x, b, l = bench.synth_frame(cfg, 0, i)
            f = m.extractor(x.to(dev))
            cls, reg = m.rpn(f)
            anchor = m.anchor_maker.device_anchors((600, 1000), dev)
            boxes, scores = ops.proposal_prologue(reg[0], cls[0], anchor, 1 / 1000)
            idx, ssc, sbx, cnt = ops.topk_sorted(scores, 12000, boxes)
            keep, _, c = ops.nms_sorted(sbx, 0.7)
            k = keep[:int(c.item())].cpu().numpy()
            print(tag, "frame", i, "valid", int(cnt.item()), "kept", len(k), "rank of the 2000th kept:", int(k[1999]) if len(k) >= 2000 else None,
                  "of the 300th:", int(k[299]) if len(k) >= 300 else None)
stats("init")
for step in range(int(os.environ.get("STEPS", "40"))):
    x, bbox, label = bench.synth_frame(cfg, 0, step % 8)
    m.train()
    pred, target = m(x.to(dev), [bbox.to(dev)], [label.to(dev)])
    loss = crit(pred, target)[0]
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
m.eval()
stats("after %s steps" % os.environ.get("STEPS", "40"))
