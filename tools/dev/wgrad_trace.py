"""Developer tool: where a row segment's cycles go in rpn_conv3x3_wgrad_kernel (-DRC3_TRACE build: s_memtime stamps of thread 0 of
workgroup (0, 0), even rows only).  tools/dev/build_variant.sh rpn_conv wgtrace -DRC3_TRACE; FRCNN_HIP_LIB=build_dbg/wgtrace/libfrcnn_hip.so python tools/dev/wgrad_trace.py"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
from faster_rcnn_pytorch_amd import ops, _lib
g = torch.Generator().manual_seed(0)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
feats = [torch.randn(1, 256, h, w, generator=g).bfloat16().cuda() for h, w in shapes]
draws = [torch.randn(1, 256, h, w, generator=g).bfloat16().cuda() for h, w in shapes]
for _ in range(3): ops.rpn_conv_wgrad(feats, draws)
torch.cuda.synchronize()
b0 = np.zeros(16, np.uint64); _lib.lib.frcnn_rc3_trace_read(b0.ctypes.data_as(C.c_void_p))
ops.rpn_conv_wgrad(feats, draws); torch.cuda.synchronize()
b1 = np.zeros(16, np.uint64); _lib.lib.frcnn_rc3_trace_read(b1.ctypes.data_as(C.c_void_p))
d = (b1 - b0).astype(np.int64); n = max(1, int(d[4]))
print("even rows traced: %d; s_memtime ticks per row segment (x ~0.52 ns):" % n)
for i, name in enumerate(["barrier (incl. waiting for the other waves)", "compute: 4 K steps x 9 MFMAs + fragment reads", "wait for the set's loads + LDS stores", "issue the next loads"]):
    print("  %-50s %8.0f" % (name, d[i] / n))
print("  sum %.0f ticks = %.2f us" % (d[:4].sum() / n, d[:4].sum() / n * 0.52e-3))
blk = np.zeros((1024, 4), np.uint64); _lib.lib.frcnn_wg_blocks_read(blk.ctypes.data_as(C.c_void_p))
b = blk.astype(np.int64); b = b[b[:, 0] > 0]
t0 = b[:, 0].min()
print("workgroups %d; start %.1f..%.1f us; loop end %.1f..%.1f us; end %.1f..%.1f us" % (len(b), (b[:,0].min()-t0)/100, (b[:,0].max()-t0)/100, (b[:,1].min()-t0)/100, (b[:,1].max()-t0)/100, (b[:,2].min()-t0)/100, (b[:,2].max()-t0)/100))
nsp = len(b) // 16
loop = (b[:, 1] - b[:, 0]) / 100.0
for sp in range(nsp):
    m = b[sp::nsp] if False else b[[g * nsp + sp for g in range(16)]]
    l = (m[:, 1] - m[:, 0]) / 100.0
    print("  split %2d: %3d row segments, loop %.1f..%.1f us (%.2f us per segment), epilogue %.1f us" % (sp, m[0, 3], l.min(), l.max(), l.mean() / max(1, m[0, 3]), ((m[:, 2] - m[:, 1]) / 100.0).mean()))
