"""Which torch ops (with shapes) the bf16 FPN step spends its GPU time in: torch.profiler over one eager step (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from faster_rcnn_pytorch_amd.new_model import FRCNN
from faster_rcnn_pytorch_amd.loss import FRCNNLoss
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["fpn"]
torch.manual_seed(0)
model = FRCNN(num_classes=91, sampling="device", seed=1).to(dev)
crit = FRCNNLoss(None)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=2e-3, momentum=0.9, weight_decay=1e-4, fused=True)
x, b, l = bench.synth_frame(cfg, 0, 0)
x, b, l = x.to(dev), b.to(dev), l.to(dev)
AMP = os.environ.get("AMP", "bf16") == "bf16"
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=AMP):
        pred, target = model(x, [b], [l])
    pred = tuple(p.float() for p in pred)
    loss = crit(pred, target)[0]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=int(sys.argv[1]) if len(sys.argv) > 1 else 60, max_name_column_width=48, max_shapes_column_width=70))
