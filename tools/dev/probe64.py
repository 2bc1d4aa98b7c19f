import sys; sys.path.insert(0,".")
import torch
from faster_rcnn_pytorch_amd import ops, _lib
dev=torch.device("cuda:0")
for Cin,Cout,H,W in ((64,128,600,1000),(64,128,300,500)):
    x=torch.randn(1,Cin,H,W,device=dev); w=torch.randn(Cout,Cin,3,3,device=dev)*0.02; b=torch.randn(Cout,device=dev)
    for _ in range(2): ops.conv3x3_fwd([x],w,b,True)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(5): ops.conv3x3_fwd([x],w,b,True)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    s=_lib.prof_samples(); print(Cin,Cout,H,W,{k:round(sorted(v)[len(v)//2]*1e3,1) for k,v in s.items()})
