"""Per-kernel times of the fp32 Winograd stage at one layer shape (the library's own HIP-event brackets): forward, data gradient, weight gradient.
usage: python tools/dev/wino_kernels_time.py [Cin Cout H W] ...   (default: the VGG16 layers at 600x1000)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops, _lib

REP = int(os.environ.get("REP", "10"))
SHAPES = [(128, 128, 300, 500), (256, 256, 150, 250), (512, 512, 75, 125), (512, 512, 37, 62)]


def main():
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)] or SHAPES
    dev = torch.device("cuda:0")
    for Cin, Cout, H, W in shapes:
        x = torch.randn(1, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
        b = torch.randn(Cout, device=dev)
        dy = torch.randn(1, Cout, H, W, device=dev)
        bits = ops.conv3x3_fwd([x], w, b, True, want_bits=True)[2]
        print("%d -> %d on %d x %d" % (Cin, Cout, H, W))
        for name, fn in (("fwd", lambda: ops.conv3x3_fwd([x], w, b, True)), ("bwd_data", lambda: ops.conv3x3_bwd_data([dy], w, bits)),
                         ("wgrad", lambda: ops.conv3x3_wgrad([x], [dy], bits, want_bias=True))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(REP):
                fn()
            torch.cuda.synchronize()
            _lib.prof_enable(False)
            s = _lib.prof_samples()
            tot = sum(sum(v) for v in s.values()) * 1e3 / REP
            print("  %-9s %7.1f us : " % (name, tot) + "  ".join("%s %.1f" % (k.replace("rpn_wino_", "").replace("_kernel", ""), sorted(v)[len(v) // 2] * 1e3 * (len(v) // REP))
                                                                for k, v in s.items()), flush=True)


if __name__ == "__main__":
    main()
