"""Where do two FPN models with identical weights first differ in predict()?  (round 4: checkpoint round-trip test)"""
import sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd.new_model import FRCNN
DEV = "cuda:0"
torch.backends.cudnn.deterministic = os.environ.get("DET", "0") == "1"
print("cudnn.deterministic =", torch.backends.cudnn.deterministic)
torch.manual_seed(3)
src = FRCNN(num_classes=91, sampling="device", seed=5).to(DEV)
with torch.no_grad():
    src.frcnn_head.cls_head.weight.normal_(0, 1e-3)
    src.frcnn_head.cls_head.bias.zero_()
torch.manual_seed(99)
dst = FRCNN(num_classes=91, sampling="device", seed=5).to(DEV)
dst.load_state_dict(src.state_dict())
g = torch.Generator().manual_seed(77)
x = torch.randn(1, 3, 320, 448, generator=g).to(DEV)
caps = {}
def hook(tag, name):
    def f(m, i, o):
        outs = o.values() if isinstance(o, dict) else (o if isinstance(o, (tuple, list)) else [o])
        caps[(tag, name)] = [t.detach().clone() for t in outs if torch.is_tensor(t)]
    return f
for tag, m in (("src", src), ("dst", dst)):
    m.eval()
    for name, mod in (("body", m.backbone.body), ("backbone", m.backbone), ("rpn", m.rpn), ("head", m.frcnn_head), ("roi_pool", m.frcnn_head.roi_pool)):
        mod.register_forward_hook(hook(tag, name))
outs = {}
for rep in range(2):
    for tag, m in (("src", src), ("dst", dst)):
        outs[(tag, rep)] = m.predict(x, 0.005)
        for name in ("body", "backbone", "rpn", "roi_pool", "head"):
            caps[(tag, rep, name)] = caps.get((tag, name))
for name in ("body", "backbone", "rpn", "roi_pool", "head"):
    for (a, b) in ((("src", 0), ("src", 1)), (("src", 0), ("dst", 0)), (("dst", 0), ("dst", 1))):
        ta, tb = caps[a + (name,)], caps[b + (name,)]
        if ta is None or tb is None:
            print(name, "not captured"); continue
        same = all(x_.shape == y_.shape and torch.equal(x_, y_) for x_, y_ in zip(ta, tb))
        md = max([float((x_.float() - y_.float()).abs().max()) if x_.shape == y_.shape else -1 for x_, y_ in zip(ta, tb)] + [0])
        print(name, a, b, "equal" if same else "DIFFER", md)
print([o[0].shape[0] for o in outs.values()])
