"""RoIPool forward / backward at config V through the autograd function: us per launch (HIP events) and run-to-run reproducibility of the
backward (hash of the feature gradient over repeated launches).  FRCNN_HIP_LIB selects a variant build; FRCNN_ROI_BWD_SHARED=1 the shared-plane backward."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
C, H, W = 512, 37, 62
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
feat = torch.randn(1, C, H, W, generator=g).to(DEV).requires_grad_(True)
# RoIs like the bench's sampled proposals: centres uniform, sides 0.1 .. 0.7 of the image
LO = float(os.environ.get("ROI_MIN_SIDE", "0.1"))          # 0.1: one RoI in seven has a side under 7 cells; 0.3: none
c = torch.rand(R, 2, generator=g); wh = torch.rand(R, 2, generator=g) * (0.7 - LO) + LO
rois = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1) * torch.tensor([W * 16.0, H * 16.0, W * 16.0, H * 16.0])
rois = rois.to(DEV)
go = torch.randn(R, C, 7, 7, generator=g).to(DEV)
hashes = set()
def step(check=False):
    feat.grad = None
    out = ops.roi_pool(feat, rois, (7, 7), 1 / 16.0)
    out.backward(go)
    if check:
        hashes.add(hashlib.sha256(feat.grad.cpu().numpy().tobytes()).hexdigest()[:12])
for _ in range(10): step(True)
torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
for _ in range(100): step()
torch.cuda.synchronize(); _lib.prof_enable(False)
print(os.environ.get("FRCNN_HIP_LIB", "default"), "shared" if os.environ.get("FRCNN_ROI_BWD_SHARED") else "private", "R=%d min side %.2f" % (R, LO),
      {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.prof_report().items()}, "distinct gradient hashes over 10 runs:", len(hashes))
