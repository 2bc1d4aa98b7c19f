"""Developer tool: per-workgroup timeline of roi_align_bwd_tile_kernel from a -DRT_TRACE build (tools/dev/build_variant.sh roi_align ra_trace -DRT_TRACE).
FRCNN_HIP_LIB=build_dbg/ra_trace/libfrcnn_hip.so KEY=rois29 python tools/dev/ra_trace.py   (RoI sets: build_dbg/fpn_rois.npz, written by ra_bwd_time.py)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from faster_rcnn_pytorch_amd.ops import lib, _ptr, _np_ptr, _level_tables, _stream, check
dev = torch.device("cuda:0")
d = np.load(os.path.join(ROOT, "build_dbg", "fpn_rois.npz"))
scales = [float(s) for s in d["scales"]]; shapes = [tuple(int(v) for v in s) for s in d["shapes"]]
r = d[os.environ.get("KEY", "rois29")]
rois = torch.from_numpy(r).to(dev)
g = torch.randn((len(r), 256, 7, 7), device=dev)
grads = [torch.empty(s, device=dev) for s in shapes]
ptrs, H, W, sc = _level_tables(grads, scales)
nb = lib.frcnn_ms_roi_align_bwd_workspace(_np_ptr(H), _np_ptr(W), 4, 256, len(r))
ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
def run():
    check(lib.frcnn_ms_roi_align_bwd(_ptr(g), ptrs, _np_ptr(H), _np_ptr(W), _np_ptr(sc), 4, 256, _ptr(rois), len(r), 7, 7, 2, 0, 2, 224.0, 4,
                                     _ptr(ws), ws.numel(), _stream()), "bwd")
for _ in range(5): run()
torch.cuda.synchronize()
buf = np.zeros((32768, 6), np.uint64)
lib.frcnn_ra_trace_read.argtypes = [C.c_void_p]
lib.frcnn_ra_trace_read(buf.ctypes.data_as(C.c_void_p))
b = buf.astype(np.int64)
item = b[b[:, 5] == 1]; fill = b[b[:, 5] == 2]
t0 = min(item[:, 0].min(), fill[:, 0].min())
us = lambda v: (v - t0) / 100.0
print("item workgroups %d (steps: total %d, max %d), fill workgroups %d, idle records %d" % (len(item), item[:, 3].sum(), item[:, 3].max(), len(fill), (b[:, 5] == 0).sum()))
print("item: first start %.2f, last start %.2f, last end %.2f us | fill: first start %.2f, last start %.2f, last end %.2f us" % (
    us(item[:, 0].min()), us(item[:, 0].max()), us(item[:, 2].max()), us(fill[:, 0].min()), us(fill[:, 0].max()), us(fill[:, 2].max())))
dur = (item[:, 2] - item[:, 0]) / 100.0; fb = (item[:, 1] - item[:, 0]) / 100.0
print("item: start -> first barrier: median %.2f p90 %.2f us | per step after it: median %.2f p10 %.2f p90 %.2f us" % (
    np.median(fb), np.percentile(fb, 90), *np.percentile(((item[:, 2] - item[:, 1]) / 100.0) / np.maximum(item[:, 3], 1), [50, 10, 90])))
for lo, hi in ((1, 4), (5, 12), (13, 20), (21, 64)):
    m = (item[:, 3] >= lo) & (item[:, 3] <= hi)
    if m.any(): print("  steps %2d..%2d: %4d workgroups, start median %.1f us, duration median %.1f us, us/step %.2f" % (lo, hi, m.sum(), np.median(us(item[m, 0])), np.median(dur[m]), np.median(dur[m] / item[m, 3])))
print("starts per 5 us:", np.histogram(us(item[:, 0]), bins=np.arange(0, 80, 5))[0])
print("ends   per 5 us:", np.histogram(us(item[:, 2]), bins=np.arange(0, 80, 5))[0])
cu = item[:, 4]
ids, cnts = np.unique(cu, return_counts=True)
print("distinct (xcc, hw_id) values %d; item workgroups per value: min %d max %d" % (len(ids), cnts.min(), cnts.max()))
