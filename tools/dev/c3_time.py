"""Times of the first-convolution kernels (csrc/conv_c3.hip) at 3 x 600 x 1000 -> 64, library HIP-event brackets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops, _lib
dev = torch.device("cuda:0")
x = torch.randn(1, 3, 600, 1000, device=dev); w = torch.randn(64, 3, 3, 3, device=dev) * 0.3; b = torch.randn(64, device=dev) * 0.1
dy = torch.randn(1, 64, 600, 1000, device=dev)
y, bits = ops.conv3x3_c3_fwd(x, w, b, True, want_bits=True)
for _ in range(2):
    ops.conv3x3_c3_fwd(x, w, b, True, want_bits=True); ops.conv3x3_c3_wgrad(x, dy, bits)
torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
for _ in range(10):
    ops.conv3x3_c3_fwd(x, w, b, True, want_bits=True); ops.conv3x3_c3_wgrad(x, dy, bits)
torch.cuda.synchronize(); _lib.prof_enable(False)
print({k: round(sorted(v)[len(v) // 2] * 1e3, 1) for k, v in _lib.prof_samples().items()})
_lib.prof_reset(); _lib.prof_enable(True)
for _ in range(10):
    ops.conv3x3_c3_wgrad(x, dy, None)
torch.cuda.synchronize(); _lib.prof_enable(False)
print("no bits:", {k: round(sorted(v)[len(v) // 2] * 1e3, 1) for k, v in _lib.prof_samples().items()})
