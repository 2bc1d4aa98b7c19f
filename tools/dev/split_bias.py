"""Developer tool: is the split-product form of the conv stage BIASED against the native one?  Per direction of one layer: the relative scale
<d, ref> / <ref, ref> of d = out - ref64 (a systematic shrink or growth) next to the rms distance, for both product forms, against float64 on the CPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from faster_rcnn_pytorch_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
for Cin, Cout, H, W in [tuple(int(v) for v in s.split(",")) for s in os.environ.get("SHAPES", "256,256,150,250;512,512,75,125;128,128,300,500").split(";")]:
    x = torch.relu(torch.randn(1, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    dy = torch.randn(1, Cout, H, W, generator=g) * (torch.rand(1, Cout, H, W, generator=g) > 0.5)
    y64 = F.conv2d(x.double(), w.double(), None, padding=1)
    dx64 = F.conv_transpose2d(dy.double(), w.double(), None, padding=1)
    dw64 = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)
    for mode in ("native", "split"):
        ops.conv3x3_f32_products(mode)
        y = ops.conv3x3_fwd([x.to(DEV)], w.to(DEV), None, False)[0].double().cpu()
        dx = ops.conv3x3_bwd_data([dy.to(DEV)], w.to(DEV), None)[0].double().cpu()
        dw, _ = ops.conv3x3_wgrad([x.to(DEV)], [dy.to(DEV)], None, want_bias=True)
        dw = dw.double().cpu()
        def stat(o, r):
            d = o - r
            return "scale %+.2e rms %.2e" % (float((d * r).sum() / (r * r).sum()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()))
        print("%d->%d %dx%d %-6s fwd %s | bwd_data %s | wgrad %s" % (Cin, Cout, H, W, mode, stat(y, y64), stat(dx, dx64), stat(dw, dw64)))
    ops.conv3x3_f32_products("native")
