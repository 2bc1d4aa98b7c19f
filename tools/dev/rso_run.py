import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops
dev = "cuda:0"
r = torch.rand(512, 4, device=dev); r[:, 2:] = r[:, :2] + torch.rand(512, 2, device=dev) * 0.4
for _ in range(20):
    ops.roi_scale_order(r, (1344., 800., 1344., 800.), [(200, 336), (100, 168), (50, 84), (25, 42)])
torch.cuda.synchronize()
