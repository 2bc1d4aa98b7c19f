import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_rcnn_pytorch_amd import ops, _lib
from faster_rcnn_pytorch_amd.anchor import FRCNNAnchorMaker
dev='cuda:0'
H,W=600,1000
am=FRCNNAnchorMaker(); N=(H//16)*(W//16)*9
lib=C.CDLL(_lib.LIB_PATH)
def run(reg, cls, name):
    reg=torch.from_numpy(reg).to(dev); cls=torch.from_numpy(cls).to(dev)
    grid=am.grid_desc((H,W))
    for _ in range(3): ops.region_proposal(reg, cls, None, 1/1000, 12000, 0.7, 2000, grid=grid)
    torch.cuda.synchronize()
    z=(C.c_uint64*64)()
    lib.frcnn_dbg_read(z); a=np.array(list(z),dtype=np.int64)
    rois,cnt,_=ops.region_proposal(reg, cls, None, 1/1000, 12000, 0.7, 2000, grid=grid)
    torch.cuda.synchronize()
    lib.frcnn_dbg_read(z); b=np.array(list(z),dtype=np.int64)
    d=b-a; nb=d[40]
    print(name, 'kept', int(cnt.item()), 'blocks', nb)
    print('   busy/blk :', ' '.join('%5d'%(d[w]/nb) for w in range(16)))
    print('   wait/blk :', ' '.join('%5d'%(d[16+w]/nb) for w in range(16)))
rng=np.random.RandomState(0)
run((rng.randn(N,4)*0.02).astype(np.float32), (rng.randn(N,2)*0.02).astype(np.float32), 'init   ')
