"""Launches the fp32 3x3 RPN conv kernels a few times at one bench shape (argv[1] = V | F): a target for rocprofv3 --pmc / --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops
DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "V"
C_, shapes = (512, [(37, 62)]) if which == "V" else (256, [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)])
feats = [torch.randn(1, C_, h, w, device=DEV) for h, w in shapes]
w = torch.randn(C_, C_, 3, 3, device=DEV) * 0.02
g = [torch.randn_like(f) for f in feats]
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    ops.rpn_conv3x3_fwd(feats, w); ops.rpn_conv3x3_wgrad(feats, g)
torch.cuda.synchronize()
