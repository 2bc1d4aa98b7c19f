"""Which piece of parallel.GraphStep moves a gradient off the eager loss.backward() (one process, no collective)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import parallel
from faster_rcnn_pytorch_amd.loss import FRCNNLoss
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_graph_ddp import synth

mirror = sys.argv[1] if len(sys.argv) > 1 else "vgg"
dev = torch.device("cuda:0")
if mirror == "vgg":
    from faster_rcnn_pytorch_amd.model import FRCNN
    nc, lo, hi, H, W = 21, 0, 20, 320, 480
else:
    from faster_rcnn_pytorch_amd.new_model import FRCNN
    nc, lo, hi, H, W = 91, 1, 91, 320, 448
    torch.backends.cudnn.deterministic = True
crit = FRCNNLoss(None)


def fresh():
    torch.manual_seed(0)
    m = FRCNN(num_classes=nc, sampling="device", seed=10).to(dev)
    return m


frames = [tuple(t.to(dev) for t in synth(50 + i, H, W, 3, lo, hi)) for i in range(1)]
names = None


def grads_eager():
    m = fresh()
    x, b, l = frames[0]
    pred, target = m(x, [b], [l])
    loss = crit(pred, target)[0]
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, float(loss)


def grads_gs(graphs):
    m = fresh()
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.0, momentum=0.0)
    keep = torch.zeros(1, device=dev)

    def forward_loss(f):
        x, b, l = frames[f]
        pred, target = m(x, [b], [l])
        return crit(pred, target), pred
    gs = parallel.GraphStep(m, opt, forward_loss, 1, dev, record=lambda f, ls: keep.copy_(ls[0].detach().reshape(1)), graphs=graphs, **m.graph_stages())
    gs.capture()
    m.sampler.reseed(10, 1)
    gs.step(0)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, float(keep)


g0, l0 = grads_eager()
g0b, l0b = grads_eager()
g1, l1 = grads_gs(False)
g2, l2 = grads_gs(True)
print("losses", l0, l0b, l1, l2)
for n in g0:
    d = [float((g0[n] - g[n]).abs().max()) for g in (g0b, g1, g2)]
    if any(d):
        print("%-40s eager-rerun %.3e  pieces %.3e  graphs %.3e   scale %.3e" % (n, d[0], d[1], d[2], float(g0[n].abs().max())))
print("done")
