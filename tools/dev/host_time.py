"""Developer tool: host enqueue time vs GPU time of one training step (is the step host-bound?)."""
import sys, os, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from faster_rcnn_pytorch_amd.loss import FRCNNLoss
cfgname = sys.argv[1] if len(sys.argv) > 1 else "fpn"
amp = len(sys.argv) > 2 and sys.argv[2] == "bf16"
cfg = bench.CONFIGS[cfgname]
if cfgname == "vgg":
    from faster_rcnn_pytorch_amd.model import FRCNN
else:
    from faster_rcnn_pytorch_amd.new_model import FRCNN
dev = torch.device("cuda:0")
model = FRCNN(num_classes=cfg["num_classes"], sampling="device", seed=1234).to(dev)
crit = FRCNNLoss(None)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=2e-3, momentum=0.9, weight_decay=1e-4, fused=True)
frames = [tuple(t.to(dev) for t in bench.synth_frame(cfg, 0, i)) for i in range(8)]
def step(i):
    x, b, l = frames[i % 8]
    if amp:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred, target = model(x, [b], [l])
        pred = tuple(p.float() for p in pred)
    else:
        pred, target = model(x, [b], [l])
    loss = crit(pred, target)[0]
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for i in range(15): step(i)
torch.cuda.synchronize(); gc.collect(); gc.freeze(); gc.disable()
host, total = [], []
for i in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(i); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(cfgname, "bf16" if amp else "f32", "host enqueue median %.2f ms; step (sync each) median %.2f ms" % (host[15], total[15]))
if os.environ.get("CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(20): step(i)
    torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(45)
