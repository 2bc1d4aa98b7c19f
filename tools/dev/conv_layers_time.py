"""Per-layer A/B of the fp32 Winograd stage against torch's convolution (MIOpen) on the backbone's 3x3 shapes: forward (+ bias + ReLU), and
forward + backward.  Prints µs per call (HIP events over REP calls) and the largest difference relative to the output scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from faster_rcnn_pytorch_amd import ops

REP = int(os.environ.get("REP", "10"))
LAYERS = [  # name, Cin, Cout, H, W, needs dx
    ("vgg conv1_2", 64, 64, 600, 1000, True), ("vgg conv2_1", 64, 128, 300, 500, True), ("vgg conv2_2", 128, 128, 300, 500, True),
    ("vgg conv3_1", 128, 256, 150, 250, False), ("vgg conv3_2", 256, 256, 150, 250, True),
    ("vgg conv4_1", 256, 512, 75, 125, True), ("vgg conv4_2", 512, 512, 75, 125, True),
    ("vgg conv5_1", 512, 512, 37, 62, True),
    ("fpn P2", 256, 256, 200, 336, True), ("fpn P3", 256, 256, 100, 168, True), ("fpn P4", 256, 256, 50, 84, True), ("fpn P5", 256, 256, 25, 42, True),
    ("res layer2", 128, 128, 100, 168, True), ("res layer3", 256, 256, 50, 84, True), ("res layer4", 512, 512, 25, 42, True),
]
ONLY = os.environ.get("ONLY")


def timed(fn):
    """GPU time per call: the call is captured in a HIP graph (after warm-up) and replayed, so the host's enqueue time is not in the figure."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("GRAPH", "1") != "0":
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        fn = gr.replay
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, Cin, Cout, H, W, need_dx in LAYERS:
        if ONLY and ONLY not in name:
            continue
        x = torch.randn(1, Cin, H, W, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
        dy = torch.randn(1, Cout, H, W, generator=g).to(dev)
        frozen = False
        with torch.no_grad():
            ref = torch.relu(F.conv2d(x, w, b, padding=1))
            got = ops.conv3x3(x, w, b, relu=True)
            err = float((ref - got).abs().max() / ref.abs().max())
            t_ref = timed(lambda: torch.relu_(F.conv2d(x, w, b, padding=1)))
            t_got = timed(lambda: ops.conv3x3(x, w, b, relu=True))
        line = "%-12s %3d->%3d %3dx%3d  fwd torch %7.1f  ours %7.1f us (err %.1e)" % (name, Cin, Cout, H, W, t_ref, t_got, err)
        if not frozen:
            xr = x.clone().requires_grad_(need_dx)
            wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)

            def step(ours):
                for t in (xr, wr, br):
                    t.grad = None
                y = ops.conv3x3(xr, wr, br, relu=True) if ours else torch.relu_(F.conv2d(xr, wr, br, padding=1))
                y.backward(dy)
            step(False)
            gr = [t.grad.clone() for t in (wr, br)] + ([xr.grad.clone()] if need_dx else [])
            step(True)
            go = [t.grad.clone() for t in (wr, br)] + ([xr.grad.clone()] if need_dx else [])
            errs = [float((a - c).abs().max() / a.abs().max()) for a, c in zip(gr, go)]
            t_ref = timed(lambda: step(False))
            t_got = timed(lambda: step(True))
            line += "  | fwd+bwd torch %7.1f  ours %7.1f us (err %s)" % (t_ref, t_got, " ".join("%.1e" % e for e in errs))
        print(line, flush=True)


if __name__ == "__main__":
    main()
