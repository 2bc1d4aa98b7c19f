import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_rcnn_pytorch_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(21)
C_, A = 256, 3
shapes = [(40, 56), (9, 33)]
feats = [(torch.randn(1, C_, h, w, generator=g)).bfloat16().to(DEV) for h, w in shapes]
w3 = (torch.randn(C_, C_, 3, 3, generator=g) * 0.01).to(DEV)
b3 = (torch.randn(C_, generator=g) * 0.1).to(DEV)
wc, bc = (torch.randn(2 * A, C_, 1, 1, generator=g) * 0.02).to(DEV), (torch.randn(2 * A, generator=g) * 0.1).to(DEV)
wr, br = (torch.randn(4 * A, C_, 1, 1, generator=g) * 0.02).to(DEV), (torch.randn(4 * A, generator=g) * 0.1).to(DEV)
fin = [f.clone().requires_grad_(True) for f in feats]
params = [t.clone().requires_grad_(True) for t in (w3, b3, wc, bc, wr, br)]
cls, reg = ops.rpn_conv_head_levels(fin, *params)
saved = cls.grad_fn.saved_tensors
raws = saved[4 + len(feats):]
for f, r in zip(feats, raws):
    ref = torch.nn.functional.conv2d(f.double(), w3.bfloat16().double(), None, padding=1)
    err = (r.double() - ref).abs()
    print("raw", tuple(r.shape), "max err", float(err.max()), "max |ref|", float(ref.abs().max()), "per-channel max err top:", torch.topk(err.amax(dim=(0, 2, 3)), 5))
    bad = (err > 0.02).nonzero()
    print("  #bad", len(bad), bad[:10].tolist())
# w3 gradient: fused path vs the unfused autocast-style path (MIOpen bf16 conv + tail kernel) vs float64
gc, gr = torch.randn(cls.shape, generator=g).to(DEV), torch.randn(reg.shape, generator=g).to(DEV)
((cls * gc).sum() + (reg * gr).sum()).backward()
p2 = [t.clone().requires_grad_(True) for t in (w3, b3, wc, bc, wr, br)]
raws2 = [torch.nn.functional.conv2d(f, p2[0].bfloat16(), None, padding=1) for f in feats]
c2, r2 = ops.rpn_head_tail_levels(raws2, p2[1], p2[2], p2[3], p2[4], p2[5], mfma="bf16")
((c2 * gc).sum() + (r2 * gr).sum()).backward()
rp = [t.clone().double().requires_grad_(True) for t in (w3.bfloat16().double(), b3, wc.bfloat16().float(), bc, wr.bfloat16().float(), br)]
cc, rr = [], []
for f in feats:
    raw = torch.nn.functional.conv2d(f.double(), rp[0], None, padding=1)
    h = torch.relu(raw + rp[1][None, :, None, None])
    cc.append(torch.nn.functional.conv2d(h, rp[2], rp[3]).permute(0, 2, 3, 1).contiguous().view(1, -1, 2))
    rr.append(torch.nn.functional.conv2d(h, rp[4], rp[5]).permute(0, 2, 3, 1).contiguous().view(1, -1, 4))
((torch.cat(cc, 1) * gc.double()).sum() + (torch.cat(rr, 1) * gr.double()).sum()).backward()
ref = rp[0].grad
print("dW3 max |ref|", float(ref.abs().max()), "fused err", float((params[0].grad.double() - ref).abs().max()), "unfused (MIOpen bf16) err", float((p2[0].grad.double() - ref).abs().max()))
