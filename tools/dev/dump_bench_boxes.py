"""Developer tool: the score-sorted pre-NMS boxes of bench.py's frames (VGG mirror, random init, frame 0..n) -> gpurun_out/bench_boxes_v.npy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from faster_rcnn_pytorch_amd import ops
from faster_rcnn_pytorch_amd.model import FRCNN
dev = "cuda:0"
torch.manual_seed(0)
m = FRCNN(num_classes=21, sampling="device", seed=1234).to(dev)
cfg = bench.CONFIGS["vgg"]
out = []
with torch.no_grad():
    for i in range(4):
        x, b, l = bench.synth_frame(cfg, 0, i)
        f = m.extractor(x.to(dev))
        cls, reg = m.rpn(f)
        anchor = m.anchor_maker.device_anchors((600, 1000), dev)
        boxes, scores = ops.proposal_prologue(reg[0], cls[0], anchor, 1 / 1000)
        idx, ssc, sbx, cnt = ops.topk_sorted(scores, 12000, boxes)
        out.append(sbx.cpu().numpy())
        keep, _, c = ops.nms_sorted(sbx, 0.7)
        print("frame", i, "valid", int(cnt.item()), "kept", int(c.item()))
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/bench_boxes_v.npy", np.stack(out))
