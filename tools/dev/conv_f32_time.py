"""Times the fp32 3x3 RPN conv kernels (forward / data gradient / weight gradient) at the two bench shapes against MIOpen's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, C_, shapes in (("V", 512, [(37, 62)]), ("F", 256, [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)])):
    feats = [torch.randn(1, C_, h, w, device=DEV) for h, w in shapes]
    w = torch.randn(C_, C_, 3, 3, device=DEV) * 0.02
    g = [torch.randn_like(f) for f in feats]
    flop = sum(2 * C_ * C_ * 9 * h * w_ for h, w_ in shapes)
    t = timeit(lambda: ops.rpn_conv3x3_fwd(feats, w))
    print("%s fwd   %8.1f us  %6.1f TFLOP/s (%.0f %% of 157.3)" % (name, t, flop / t * 1e-6, flop / t * 1e-6 / 157.3 * 100))
    t = timeit(lambda: ops.rpn_conv3x3_bwd_data(g, w))
    print("%s bwd_d %8.1f us  %6.1f TFLOP/s (%.0f %%)" % (name, t, flop / t * 1e-6, flop / t * 1e-6 / 157.3 * 100))
    t = timeit(lambda: ops.rpn_conv3x3_wgrad(feats, g))
    print("%s wgrad %8.1f us  %6.1f TFLOP/s (%.0f %%)" % (name, t, flop / t * 1e-6, flop / t * 1e-6 / 157.3 * 100))
    t = timeit(lambda: [F.conv2d(f, w, None, padding=1) for f in feats])
    print("%s MIOpen fwd   %8.1f us" % (name, t))
    t = timeit(lambda: [torch.ops.aten.convolution_backward(gg, f, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]) for f, gg in zip(feats, g)])
    print("%s MIOpen bwd_d %8.1f us" % (name, t))
    t = timeit(lambda: [torch.ops.aten.convolution_backward(gg, f, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]) for f, gg in zip(feats, g)])
    print("%s MIOpen wgrad %8.1f us" % (name, t))
_lib.prof_reset(); _lib.prof_enable(True)
for name, C_, shapes in (("V", 512, [(37, 62)]), ("F", 256, [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)])):
    feats = [torch.randn(1, C_, h, w, device=DEV) for h, w in shapes]
    w = torch.randn(C_, C_, 3, 3, device=DEV) * 0.02
    g = [torch.randn_like(f) for f in feats]
    flop = sum(2 * C_ * C_ * 9 * h * w_ for h, w_ in shapes)
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(20):
        ops.rpn_conv3x3_fwd(feats, w); ops.rpn_conv3x3_bwd_data(g, w); ops.rpn_conv3x3_wgrad(feats, g)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    for k, v in sorted(_lib.prof_samples().items()):
        v = sorted(v)
        med = v[len(v) // 2] * 1e3
        print("%s %-34s n=%3d median %8.1f us min %8.1f  -> %6.1f TFLOP/s (%.0f %%)" % (name, k, len(v), med, v[0] * 1e3, flop / med * 1e-6, flop / med * 1e-6 / 157.3 * 100))
