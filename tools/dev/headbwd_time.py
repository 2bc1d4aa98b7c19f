"""Developer tool: time the fused RPN head tail backward at FPN size (fp32 and bf16 conv outputs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
for dt in (torch.float32, torch.bfloat16):
    raws = [torch.randn(1, 256, h, w, device=DEV).to(dt).requires_grad_(True) for h, w in shapes]
    b3 = torch.zeros(256, device=DEV, requires_grad=True)
    wc = (torch.randn(6, 256, 1, 1, device=DEV) * 0.02).requires_grad_(True); bc = torch.zeros(6, device=DEV, requires_grad=True)
    wr = (torch.randn(12, 256, 1, 1, device=DEV) * 0.02).requires_grad_(True); br = torch.zeros(12, device=DEV, requires_grad=True)
    x = torch.randn(4096, 4096, device=DEV)
    def step():
        c, r = ops.rpn_head_tail_levels(raws, b3, wc, bc, wr, br, mfma="bf16" if dt == torch.bfloat16 else "f32")
        (c.sum() + r.sum()).backward()
    for i in range(10):
        y = x @ x; step()
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for i in range(20):
        y = x @ x; step()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    print(os.environ.get("FRCNN_HIP_LIB", "default"), dt, {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.prof_report().items()})
