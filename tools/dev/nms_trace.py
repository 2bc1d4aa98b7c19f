"""Developer tool: per-wave timeline of nms_kernel from a -DNMS_TRACE build (tools/dev/build_variant.sh nms trace -DNMS_TRACE).
FRCNN_HIP_LIB=build_dbg/trace/libfrcnn_hip.so [BENCH_BOXES=build_dbg/bench_boxes_v.npy] python tools/dev/nms_trace.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from faster_rcnn_pytorch_amd import ops, _lib
DEV = "cuda:0"
bb = np.load(os.environ.get("BENCH_BOXES", "build_dbg/bench_boxes_v.npy"))[int(os.environ.get("FRAME", "0"))]
K = int(os.environ.get("K", str(len(bb))))
b = torch.from_numpy(bb[:K]).to(DEV)
for _ in range(5): ops.nms_sorted(b, 0.7)
torch.cuda.synchronize()
L = _lib.lib
L.frcnn_nms_trace_clear()
keep, _, cnt = ops.nms_sorted(b, 0.7, post_k=int(os.environ['POST_K']) if 'POST_K' in os.environ else None, want_rois=True)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), np.uint64)
L.frcnn_nms_trace_read(buf.ctypes.data_as(C.c_void_p))
nblk = (K + 63) // 64
res = buf[0:4 * nblk:4].astype(np.int64); tiles = buf[4096:4096 + nblk].astype(np.int64)
t0 = min(res[:, 0].min(), tiles[:, 0].min())
us = lambda v: (v - t0) / 100.0
print("kept", int(cnt.item()), "nblk", nblk)
print("blk | entry row_ready first_fill decided | sweeps fills || tile row: first_start last_end")
for bl in list(range(0, nblk, max(1, nblk // 24))) + [nblk - 1]:
    r = res[bl]; t = tiles[bl]
    print("%3d | %6.2f %6.2f %6.2f %6.2f | %5d %4d || %6.2f %6.2f" % (bl, us(r[0]), us(r[1]), us(r[2]), us(r[3]), r[4], r[5], us(t[0]), us(t[1])))
e = buf[8100].astype(np.int64)
if e[0]: print("emit by the last workgroup: ticket %.2f | bitmap loaded + scanned %.2f, list built %.2f, outputs stored %.2f (us)" % (us(e[0]), us(e[1]), us(e[2]), us(e[3])))
sw = np.zeros((4096, 16), np.uint64)
L.frcnn_nms_sweep_read(sw.ctypes.data_as(C.c_void_p))
for bl in (0, 7, 49, 105, 126, 187):
    if bl < nblk:
        v = sw[bl].astype(np.int64); v = v[v > 0]
        print("sweeps of block %d at us:" % bl, " ".join("%.2f" % us(x) for x in v))
print("last tile end %.2f us; last decided %.2f us; max (decided - row_ready) %.2f us at block %d" % (us(tiles[:, 1].max()), us(res[:, 3].max()), ((res[:, 3] - res[:, 1]) / 100.0).max(), int(np.argmax(res[:, 3] - res[:, 1]))))
