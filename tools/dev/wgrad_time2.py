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
    res=[]
    for rep in range(3):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.rpn_conv_wgrad(feats, draws)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1)/30*1e3)
    return res
for sh in [allsh[2:], allsh[2:4], allsh[3:], allsh, allsh[2:]]:
    print(sh, ["%.1f" % v for v in t(sh)])
