"""Developer tool: time RoIPool forward/backward (int32-argmax ABI and the a16 autograd pair) at config V."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops, _lib
dev = "cuda:0"
rng = np.random.RandomState(0)
C, H, W, R = 512, 37, 62, 128
feat = torch.randn(1, C, H, W, device=dev, requires_grad=True)
c = rng.rand(R, 2) * 0.7 + 0.15; wh = rng.rand(R, 2) * 0.52 + 0.08
rois = torch.from_numpy(np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32) * np.array([W, H, W, H], np.float32)).to(dev)
for it in range(10):
    out = ops.roi_pool(feat, rois, (7, 7), 1.0); out.backward(out)
    ops.roi_pool_with_argmax(feat.detach(), rois, (7, 7), 1.0)
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
for it in range(50):
    out = ops.roi_pool(feat, rois, (7, 7), 1.0); out.backward(out)
torch.cuda.synchronize(); _lib.prof_enable(False)
print(os.environ.get("FRCNN_HIP_LIB", "default"), {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.prof_report().items()})
