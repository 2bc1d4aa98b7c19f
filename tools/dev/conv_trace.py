import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
from faster_rcnn_pytorch_amd import ops, _lib
torch.manual_seed(0)
shapes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
feats = [torch.randn(1, 256, h, w, device="cuda").to(torch.bfloat16) for h, w in shapes]
w3 = torch.randn(256, 256, 3, 3, device="cuda") * 0.01; b3 = torch.zeros(256, device="cuda")
wc = torch.randn(3, 256, device="cuda") * 0.01; bc = torch.zeros(3, device="cuda"); wr = torch.randn(12, 256, device="cuda") * 0.01; br = torch.zeros(12, device="cuda")
WHICH = os.environ.get('WHICH', 'bwd')
run = (lambda: ops.rpn_conv_bwd_data(feats, w3)) if WHICH == 'bwd' else (lambda: ops.rpn_conv_head_levels(feats, w3, b3, wc, bc, wr, br))
for _ in range(3): out = run()
torch.cuda.synchronize()
buf0 = np.zeros(16, np.uint64); _lib.lib.frcnn_rc3_trace_read(buf0.ctypes.data_as(C.c_void_p))
out = run()
torch.cuda.synchronize()
buf = np.zeros(16, np.uint64); _lib.lib.frcnn_rc3_trace_read(buf.ctypes.data_as(C.c_void_p))
d = (buf - buf0).astype(np.int64)
names = ["frags(tap1)+mfma(tap0)", "frags(tap2)+mfma(tap1)", "vm waits (+pixel store at ky=0)", "lgkmcnt(0)", "barrier", "issue DMA/loads + frags(next tap0)", "mfma(tap2)", "whole step"]
print("workgroup 0, wave 0, one tile = 48 steps; cycles per step (s_memtime):")
for i, n in enumerate(names): print("  %-40s %8.0f  (%.1f %%)" % (n, d[i] / 48.0, 100.0 * d[i] / max(1, d[7])))
print("  of the vm waits, in the 16 ky=0 steps: %.0f per such step" % (d[8] / 16.0))

print("main loop of the tile: %.2f us by s_memrealtime (100 MHz), %d s_memtime ticks -> %.3f ticks per ns" % (d[9] / 100.0, d[10], d[10] / (d[9] * 10.0)))
print("whole tile function %d ticks: prologue %d, main loop %d (stamps included), epilogue %d" % (d[11], d[12], d[10], d[11] - d[12] - d[10]))
