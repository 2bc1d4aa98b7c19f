"""Developer tool: the conv stage as the model calls it (ops.conv3x3 under autograd: bias + ReLU [+ fused max-pool], sign / window words, kept transforms) --
input gradient and weight gradient of one layer against float64 with the DEVICE's decisions, native vs split products: relative scale and rms of the distance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from faster_rcnn_pytorch_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
for spec in os.environ.get("SHAPES", "256,256,150,250,0;256,256,150,250,1;128,128,300,500,1;512,512,75,125,1").split(";"):
    Cin, Cout, H, W, pool = (int(v) for v in spec.split(","))
    x0 = torch.relu(torch.randn(1, Cin, H, W, generator=g))
    w0 = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b0 = torch.randn(Cout, generator=g) * 0.1
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    dy = torch.randn(1, Cout, Ho, Wo, generator=g) * float(os.environ.get("DYSCALE", "1"))
    for mode in ("native", "split"):
        ops.conv3x3_f32_products(mode)
        x = x0.to(DEV).requires_grad_(True); w = w0.to(DEV).requires_grad_(True); b = b0.to(DEV).requires_grad_(True)
        y = ops.conv3x3(x, w, b, relu=True, pool=bool(pool))
        y.backward(dy.to(DEV))
        # float64 with the device's decisions: mask = where the device output is positive (and, pooled, which element of the window it took)
        xd = x0.double().requires_grad_(True); wd = w0.double().requires_grad_(True)
        pre = F.conv2d(xd, wd, b0.double(), padding=1)
        if pool:
            # the device's window choice: the position whose pre-activation equals the pooled output most closely, gated by output > 0
            yd = y.detach().double().cpu()
            up = F.interpolate(yd, scale_factor=2, mode="nearest")
            pc = pre[:, :, :2 * Ho, :2 * Wo]
            sel = ((pc.detach() - up).abs() < 1e-4 * up.abs().clamp_min(1e-3)) & (up > 0)
            # one winner per window
            ref_y = (pc * sel).reshape(1, Cout, Ho, 2, Wo, 2).sum((3, 5)) / sel.reshape(1, Cout, Ho, 2, Wo, 2).sum((3, 5)).clamp_min(1)
        else:
            ref_y = pre * (y.detach().double().cpu() > 0)
        ref_y.backward(dy.double())
        def stat(o, r):
            d = o.double().cpu() - r
            return "scale %+.2e rms %.2e max %.2e" % (float((d * r).sum() / (r * r).sum()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()), float(d.abs().max() / r.abs().max()))
        print("%d->%d %dx%d pool %d %-6s dx %s | dw %s" % (Cin, Cout, H, W, pool, mode, stat(x.grad, xd.grad), stat(w.grad, wd.grad)))
    ops.conv3x3_f32_products("native")
