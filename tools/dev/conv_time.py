import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
feats = [torch.randn(1, 256, h, w, device=DEV).bfloat16() for h, w in shapes]
w3 = torch.randn(256, 256, 3, 3, device=DEV) * 0.01; b3 = torch.zeros(256, device=DEV)
wc = torch.randn(6, 256, 1, 1, device=DEV) * 0.02; bc = torch.zeros(6, device=DEV); wr = torch.randn(12, 256, 1, 1, device=DEV) * 0.02; br = torch.zeros(12, device=DEV)
x = torch.randn(4096, 4096, device=DEV)
for i in range(30):
    y = x @ x                                   # keep the clocks up
    with torch.no_grad():
        ops.rpn_conv_head_levels(feats, w3, b3, wc, bc, wr, br)
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
for i in range(30):
    y = x @ x
    with torch.no_grad():
        ops.rpn_conv_head_levels(feats, w3, b3, wc, bc, wr, br)
torch.cuda.synchronize(); _lib.prof_enable(False)
print(os.environ.get("FRCNN_HIP_LIB", "default"), {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.prof_report().items()})
