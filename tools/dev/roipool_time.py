import os, sys, torch
sys.path.insert(0, "/root/repo")
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
C, H, W, R = 512, 37, 62, 128
feat = torch.randn(1, C, H, W, generator=g).to(DEV).requires_grad_(True)
# RoIs like the bench's sampled proposals: centres uniform, sides 0.1 .. 0.7 of the image
c = torch.rand(R, 2, generator=g); wh = torch.rand(R, 2, generator=g) * 0.6 + 0.1
rois = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1) * torch.tensor([W * 16.0, H * 16.0, W * 16.0, H * 16.0])
rois = rois.to(DEV)
def step():
    out = ops.roi_pool(feat, rois, (7, 7), 1 / 16.0)
    out.backward(torch.ones_like(out))
for _ in range(10): step()
torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
for _ in range(50): step()
torch.cuda.synchronize(); _lib.prof_enable(False)
print(os.environ.get("FRCNN_HIP_LIB", "default"), {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.prof_report().items()})
