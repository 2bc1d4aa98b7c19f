"""Golden vectors for the RPN head's 3x3 convolution in the reference's own precision and on the reference's CPU path.

`self.inter_layer = nn.Conv2d(512, 512, kernel_size=3, padding=1)` + `normal_init(m, 0, 0.01)` (models/model.py:68-77; the FPN head:
models/new_model.py:96-104 with 256 channels) executes, on the CPU, as torch's own fp32 conv2d and its autograd -- the class body
cannot be imported here (torchvision / cv2), the layer it builds can be restated in two lines.  Inputs are regenerated from the seed
at test time (same torch build in the container and on the GPU box; their sha256 is stored and checked), the expected outputs are
strided samples of the CPU results: forward, gradient with respect to the input, gradient with respect to the weight.
Re-run:  python tests/golden/make_golden_rpn_conv.py"""
import hashlib
import os

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = {"vgg600x1000": (512, [(37, 62)]), "fpn_small": (256, [(50, 84), (25, 42), (13, 21)])}
STRIDE = 89


def inputs(name):
    C, shapes = CASES[name]
    g = torch.Generator().manual_seed(20241004 + len(name))
    w = torch.empty(C, C, 3, 3).normal_(0, 0.01, generator=g)                  # normal_init(m, 0, 0.01)
    feats = [torch.randn(1, C, h, ww, generator=g).relu_() for h, ww in shapes]   # backbone outputs are post-ReLU
    gouts = [torch.randn(1, C, h, ww, generator=g) * 0.1 for h, ww in shapes]
    return C, shapes, w, feats, gouts


def sha(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(np.ascontiguousarray(t.numpy()).tobytes())
    return h.hexdigest()


def main():
    torch.set_num_threads(8)
    d = {}
    for name in CASES:
        C, shapes, w, feats, gouts = inputs(name)
        w = w.requires_grad_(True)
        feats = [f.requires_grad_(True) for f in feats]
        outs = [torch.nn.functional.conv2d(f, w, None, padding=1) for f in feats]          # the bias (zero at init) is added by the head tail
        torch.autograd.backward(outs, gouts)
        d[name + "_inputs_sha256"] = np.array(sha([w.detach()] + [f.detach() for f in feats] + gouts))
        for k, o in enumerate(outs):
            d["%s_out%d" % (name, k)] = o.detach().reshape(-1)[::STRIDE].numpy().copy()
            d["%s_dx%d" % (name, k)] = feats[k].grad.reshape(-1)[::STRIDE].numpy().copy()
        d[name + "_dw"] = w.grad.reshape(-1)[::STRIDE].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "rpn_conv.npz"), **d)
    print({k: (v.shape, v.dtype) for k, v in d.items()})


if __name__ == "__main__":
    main()
