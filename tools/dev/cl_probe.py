"""Probe: does this PyTorch/MIOpen run the FPN top-down path in channels_last at no extra cost? (times fwd+bwd of the FPN module alone)"""
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from faster_rcnn_pytorch_amd.new_model import FeaturePyramidNetwork
torch.manual_seed(0)
dev = "cuda:0"
shapes = [(256, 200, 336), (512, 100, 168), (1024, 50, 84), (2048, 25, 42)]
def run(cl, dtype):
    fpn = FeaturePyramidNetwork().to(dev)
    if cl: fpn = fpn.to(memory_format=torch.channels_last)
    xs = {str(i): torch.randn(1, *s, device=dev, requires_grad=True) for i, s in enumerate(shapes)}
    def step():
        inp = {k: (v.contiguous(memory_format=torch.channels_last) if cl else v) for k, v in xs.items()}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
            out = fpn(inp)
        loss = sum(o.float().sum() for o in out.values())
        loss.backward()
        return out
    for _ in range(5): out = step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20 * 1e3
    o = out["0"]
    print("channels_last=%s dtype=%s: %.3f ms/iter; P2 is_contiguous(cl)=%s strides=%s" % (cl, dtype, dt, o.is_contiguous(memory_format=torch.channels_last), o.stride()))
for dtype in ("f32", "bf16"):
    for cl in (False, True):
        run(cl, dtype)
