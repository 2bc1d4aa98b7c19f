"""Developer tool: the VGG extractor's backward as a chain -- the gradient reaching every stage input, native products vs split products in the data gradient only
(forward and weight gradients native in both runs): relative distance (max over the tensor / its scale), scale bias, and where it grows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops
from faster_rcnn_pytorch_amd.model import FRCNN
DEV = "cuda:0"
torch.manual_seed(0)
m = FRCNN(num_classes=21, sampling="device", seed=1234).to(DEV)
ext = m.extractor
g = torch.Generator().manual_seed(1)
x0 = torch.randn(1, 3, 600, 1000, generator=g).to(DEV)
up = None
def run(where):
    global up
    if where: os.environ["TMP_SPLIT_WHERE"] = where
    else: os.environ.pop("TMP_SPLIT_WHERE", None)
    caps = []
    orig = ops.conv3x3
    def spy(x, w, b=None, relu=False, pool=False):
        x.register_hook(lambda gr, i=len(caps): caps.append((i, gr.detach().clone())))
        return orig(x, w, b, relu=relu, pool=pool)
    ops.conv3x3 = spy
    import faster_rcnn_pytorch_amd.model as M
    for p in ext.parameters(): p.grad = None
    x = x0.clone().requires_grad_(True)
    f = ext(x)
    if up is None: up = torch.randn(f.shape, generator=torch.Generator().manual_seed(2)).to(DEV)
    f.backward(up)
    ops.conv3x3 = orig
    ops.conv3x3_f32_products("native")
    return caps, {n: p.grad.detach().clone() for n, p in ext.named_parameters()}
ca, ga = run(None)
cb, gb = run("bwd_data")
cc, gc = run(None)
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
def sc(a, b): return float(((a - b) * b).sum() / (b * b).sum())
print("gradient reaching the input of conv stage k (in backward order), split-bwd vs native | native vs native again")
for (i, a), (_, b), (_, c) in zip(ca, cb, cc):
    print("stage %2d shape %-22s rel max %.2e scale %+.2e | %.1e" % (i, tuple(a.shape), rel(b, a), sc(b, a), rel(c, a)))
for n in ga:
    print("%-12s rel max %.2e scale %+.2e" % (n, rel(gb[n], ga[n]), sc(gb[n], ga[n])))
