"""Where a launch of rpn_wino_gemm_kernel spends its time: wall-clock stamps per workgroup (variant build -DWN_STAMP, tools/dev/build_variant.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops, _lib

dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:]] or [512, 512, 75, 125]
Cin, Cout, H, W = a
x = torch.randn(1, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
dy = torch.randn(1, Cout, H, W, device=dev)
lib = _lib.lib
lib.frcnn_debug_wn_stamps.argtypes = [C.c_void_p]
for name, fn in (("fwd", lambda: ops.conv3x3_fwd([x], w)), ("wgrad", lambda: ops.conv3x3_wgrad([x], [dy]))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros(4096, np.uint64)
    lib.frcnn_debug_wn_stamps(buf.ctypes.data_as(C.c_void_p))
    t = buf.reshape(1024, 4)[:512].astype(np.float64) * 0.01          # us
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    print("%s  workgroups %d   launch span (first entry -> last exit) %.1f us" % (name, len(t), t[:, 3].max() - base))
    for lbl, v in (("entry after first entry", t[:, 0] - base), ("entry -> first data", t[:, 1] - t[:, 0]), ("main loop", t[:, 2] - t[:, 1]),
                   ("last segment (slab / ticket / reduce / store)", t[:, 3] - t[:, 2]), ("exit before last exit", t[:, 3].max() - t[:, 3])):
        print("   %-46s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (lbl, v.min(), np.median(v), np.percentile(v, 90), v.max()))
    d = t[:, 2] - t[:, 1]
    b = np.arange(len(d))
    print("   main loop by XCD (blockIdx & 7):", " ".join("%.1f" % np.median(d[(b & 7) == k]) for k in range(8)))
    print("   main loop by eighth of blockIdx >> 3:", " ".join("%.1f" % np.median(d[((b >> 3) * 8 // 64) == k]) for k in range(8)))
    print("   start skew by XCD:", " ".join("%.2f" % np.median((t[:, 0] - base)[(b & 7) == k]) for k in range(8)))
    o = np.argsort(d)
    print("   fastest blocks:", [(int(i), round(float(d[i]), 1)) for i in o[:8]], " slowest:", [(int(i), round(float(d[i]), 1)) for i in o[-8:]])
