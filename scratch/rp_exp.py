import sys, numpy as np, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_rcnn_pytorch_amd import ops, _lib
dev='cuda:0'
C,H,W,R=512,37,62,128
feat=torch.randn(1,C,H,W,device=dev)
def bench(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    _lib.prof_enable(False)
    rep=_lib.prof_report()
    return s.elapsed_time(e)/n*1e3, {k: round(v[0]/v[1]*1e3,2) for k,v in rep.items()}
rng=np.random.RandomState(0)
def rois(kind):
    if kind=='tiny':
        c=rng.rand(R,2)*np.array([W-2,H-2])+1
        r=np.concatenate([c,c],1)
    elif kind=='full':
        r=np.tile(np.array([[0,0,W-1,H-1]]),(R,1))
    else:
        c=rng.rand(R,2)*0.7+0.15; wh=rng.rand(R,2)*0.5+0.05
        r=np.clip(np.concatenate([c-wh/2,c+wh/2],1),0,1)*np.array([W,H,W,H])
    return torch.from_numpy(r.astype(np.float32)).to(dev)
x=torch.randn(R,C,7,7,device=dev); y=torch.empty_like(x)
print('copy 12.85MB:', bench(lambda: y.copy_(x))[0], 'us')
z=torch.empty(R,C,7,7,dtype=torch.int32,device=dev)
print('fill 2x12.85MB:', bench(lambda: (y.fill_(1.0), z.fill_(1)))[0], 'us')
for kind in ['tiny','rand','full']:
    r=rois(kind)
    ft=feat.clone().requires_grad_(True)
    out=ops.roi_pool(ft, r, (7,7), 1.0)
    g=torch.randn_like(out)
    def f():
        o=ops.roi_pool(ft, r, (7,7), 1.0)
        o.backward(g)
    t,rep=bench(f)
    print(kind, 'wall/iter us', round(t,1), rep)
