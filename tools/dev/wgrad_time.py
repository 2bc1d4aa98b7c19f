import sys, torch
sys.path.insert(0, "/root/repo")
from faster_rcnn_pytorch_amd import ops
dev="cuda:0"
g=torch.Generator().manual_seed(0)
allsh=[(200,336),(100,168),(50,84),(25,42),(13,21)]
def t(shapes):
    feats=[torch.randn(1,256,h,w,generator=g).bfloat16().to(dev) for h,w in shapes]
    draws=[torch.randn(1,256,h,w,generator=g).bfloat16().to(dev) for h,w in shapes]
    for _ in range(5): ops.rpn_conv_wgrad(feats, draws)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.rpn_conv_wgrad(feats, draws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/30*1e3
for sh in [allsh, allsh[:1], allsh[:2], allsh[1:2], allsh[2:3], allsh[3:4], allsh[4:5], allsh[2:]]:
    print(sh, "%.1f us (incl. finalize)" % t(sh))
