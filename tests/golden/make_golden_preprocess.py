#!/usr/bin/env python
"""Golden vectors for the input stage, generated HERE with Pillow -- the library the reference's
transforms resize with (new_datasets/transforms.py:106 F.resize on a PIL image == Image.resize(size[::-1], BILINEAR)) --
and with the torch expressions of ToTensor / Normalize / the box transforms (transforms.py:64-68,113-117,238-240,
256-281).  new_datasets/transforms.py itself cannot be imported (torchvision is not installed), so the sizes of
get_size_with_aspect_ratio are checked by hand-derived known answers in tests/test_preprocess.py instead.

    python tests/golden/make_golden_preprocess.py      # writes tests/golden/preprocess.npz
"""
import hashlib
import os

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def pil_pipeline(img, out_hw, flip):
    pil = Image.fromarray(img, "RGB")
    if flip:
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT)                       # F.hflip
    pil = pil.resize((out_hw[1], out_hw[0]), Image.BILINEAR)               # F.resize(image, (h, w))
    u8 = np.array(pil)
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().float().div(255)                   # F.to_tensor
    t = t.sub(torch.tensor(MEAN)[:, None, None]).div(torch.tensor(STD)[:, None, None])        # F.normalize
    return u8, t.numpy()


def box_pipeline(boxes, w, h, ow, oh, flip):
    b = torch.from_numpy(boxes)
    if flip:
        b = b[:, [2, 1, 0, 3]] * torch.as_tensor([-1, 1, -1, 1]) + torch.as_tensor([w, 0, w, 0])
    rw, rh = float(ow) / float(w), float(oh) / float(h)
    b = b * torch.as_tensor([rw, rh, rw, rh])
    b = b / torch.tensor([ow, oh, ow, oh], dtype=torch.float32)
    return b.numpy()


def main():
    rng = np.random.RandomState(2024)
    out = {}
    cases = [("up", 37, 53, 61, 88, False), ("down", 120, 90, 64, 48, False), ("down_flip", 97, 131, 40, 54, True),
             ("same_w", 50, 80, 100, 80, False), ("strong_down", 200, 300, 23, 31, False), ("tall", 64, 40, 90, 50, True)]
    for name, h, w, oh, ow, flip in cases:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        if name == "up":
            img[:8] = 255
            img[8:16] = 0                                                # saturated rows: clip8 at both ends
        u8, f = pil_pipeline(img, (oh, ow), flip)
        boxes = (rng.rand(5, 4) * np.array([w, h, w, h])).astype(np.float32)
        out[name + "_img"], out[name + "_u8"], out[name + "_f32"] = img, u8, f
        out[name + "_meta"] = np.array([h, w, oh, ow, int(flip)], np.int64)
        out[name + "_boxes"], out[name + "_boxes_out"] = boxes, box_pipeline(boxes, w, h, ow, oh, flip)
    # full-size cases: only digests of the Pillow result (images are regenerated from the seed in the test)
    for name, h, w, oh, ow, flip, seed in [("voc", 375, 500, 800, 1066, False, 7), ("coco", 480, 640, 800, 1066, True, 8),
                                           ("wide", 300, 1000, 399, 1333, False, 9)]:
        img = np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)
        u8, f = pil_pipeline(img, (oh, ow), flip)
        out[name + "_meta"] = np.array([h, w, oh, ow, int(flip), seed], np.int64)
        out[name + "_sha_u8"] = np.frombuffer(hashlib.sha256(u8.tobytes()).digest(), np.uint8)
        out[name + "_sha_f32"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(f).tobytes()).digest(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print("wrote preprocess.npz with", len(out), "arrays; Pillow", Image.__version__ if hasattr(Image, "__version__") else "")


if __name__ == "__main__":
    main()
