"""Per-wave phase stamps of the RoIPool kernels from a -DRP_TRACE build (tools/dev/build_variant.sh roi_pool trace -DRP_TRACE):
FRCNN_HIP_LIB=build_dbg/trace/libfrcnn_hip.so python tools/dev/roipool_trace.py
forward  slots: 0 entry, 1 staging + bin table issued/landed, 2 barrier passed, 3 first task stored, 4 last task stored, 5 stores acknowledged
backward slots: 0 entry, 1 loads issued, 2 plane zeroed, 3 ranks known (= loads landed + shuffles), 4 adds done, 5 barrier passed, 6 reduced + stored"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
C_, H, W, R = 512, 37, 62, 128
LO = float(os.environ.get("ROI_MIN_SIDE", "0.1"))
feat = torch.randn(1, C_, H, W, generator=g).to(DEV).requires_grad_(True)
c = torch.rand(R, 2, generator=g); wh = torch.rand(R, 2, generator=g) * (0.7 - LO) + LO
rois = (torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1) * torch.tensor([W * 16.0, H * 16.0, W * 16.0, H * 16.0])).to(DEV)
go = torch.randn(R, C_, 7, 7, generator=g).to(DEV)
for _ in range(5):
    feat.grad = None
    out = ops.roi_pool(feat, rois, (7, 7), 1 / 16.0); out.backward(go)
torch.cuda.synchronize()
buf = np.zeros((2, 8192, 8), np.uint64)
_lib.lib.frcnn_rp_trace_read(buf.ctypes.data_as(C.c_void_p))
for kern, name, nslot in ((0, "forward", 6), (1, "backward", 7)):
    t = buf[kern].astype(np.int64)
    t = t[t[:, 0] > 0][:, :nslot]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print("%s: %d waves; kernel span %.2f us (first entry -> last stamp)" % (name, len(t), us.max()))
    print("  slot:      " + " ".join("%7d" % i for i in range(nslot)))
    for label, f in (("min", np.min), ("median", np.median), ("p90", lambda a, axis: np.percentile(a, 90, axis=axis)), ("max", np.max)):
        print("  %-9s  " % label + " ".join("%7.2f" % v for v in f(us, axis=0)))
    d = np.diff(us, axis=1)
    print("  phase median (us): " + " ".join("%7.2f" % v for v in np.median(d, axis=0)))
