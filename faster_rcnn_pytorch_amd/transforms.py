"""Device-side input stage: the reference's per-sample transforms and pad-to-32 collate for ONE frame
that is already in HBM as uint8 HWC, on the kernels of csrc/preprocess.hip.

Mirrors (file:line under the reference):
  new_datasets/transforms.py:57-72    hflip (image + boxes)
  new_datasets/transforms.py:76-132   resize(image, target, size, max_size) incl. get_size_with_aspect_ratio
  new_datasets/transforms.py:238-281  ToTensor, Normalize (boxes / (w, h))
  new_datasets/build.py:20-33, datasets/build.py:10-24   Compose([RandomHorizontalFlip, Resize(800, 1333), ToTensor, Normalize])
  new_datasets/coco_dataset.py:49-66  batched_tensor_from_tensor_list (zero pad to a multiple of 32)
JPEG decoding and the dataset classes stay out of scope; the flip coin is the caller's (random.random() < p).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .ops import _ptr, _req, _stream, _workspace, _np_ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_size_with_aspect_ratio(image_size, size, max_size=None):
    """transforms.py:79-99.  image_size = (w, h) as PIL reports it; returns (oh, ow)."""
    w, h = int(image_size[0]), int(image_size[1])
    if max_size is not None:
        lo, hi = float(min(w, h)), float(max(w, h))
        if hi / lo * size > max_size:
            size = int(round(max_size * lo / hi))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def get_size(image_size, size, max_size=None):
    """transforms.py:101-105: a (w, h) tuple is taken as is (reversed), a scalar is the shorter side."""
    if isinstance(size, (list, tuple)):
        return tuple(size[::-1])
    return get_size_with_aspect_ratio(image_size, size, max_size)


def padded_size(h, w, size_divisible=32):
    """coco_dataset.py:57-60."""
    stride = float(size_divisible)
    return int(math.ceil(float(h) / stride) * stride), int(math.ceil(float(w) / stride) * stride)


def preprocess_image(img, out_hw, pad_hw=None, flip=False, mean=IMAGENET_MEAN, std=IMAGENET_STD, want_u8=False):
    """uint8 HWC [h, w, 3] on a HIP device -> float32 [3, pad_h, pad_w] (and optionally the resized uint8 [oh, ow, 3])."""
    img = _req(img, torch.uint8, "img")
    if img.dim() != 3 or img.shape[2] != 3:
        raise ValueError("img must be [h, w, 3] uint8 (HWC RGB), got %s" % (tuple(img.shape),))
    h, w = int(img.shape[0]), int(img.shape[1])
    oh, ow = int(out_hw[0]), int(out_hw[1])
    ph, pw = (oh, ow) if pad_hw is None else (int(pad_hw[0]), int(pad_hw[1]))
    dev = img.device
    out = torch.empty((3, ph, pw), dtype=torch.float32, device=dev)
    u8 = torch.empty((oh, ow, 3), dtype=torch.uint8, device=dev) if want_u8 else None
    m = np.ascontiguousarray(mean, dtype=np.float32)
    s = np.ascontiguousarray(std, dtype=np.float32)
    nb = _lib.workspace_bytes(_lib.OP_PREPROCESS, (h << 32) | w, (oh << 32) | ow)
    ws = _workspace(dev, nb)
    with torch.cuda.device(dev):
        check(lib.frcnn_preprocess_image(_ptr(img), h, w, int(bool(flip)), oh, ow, ph, pw, _np_ptr(m), _np_ptr(s), _ptr(out), _ptr(u8),
                                         _ptr(ws), nb, _stream()), "preprocess_image")
    return (out, u8) if want_u8 else out


def preprocess_boxes(boxes, src_wh, out_wh, flip=False):
    """xyxy boxes in source pixels -> the normalised boxes the model takes (hflip, resize ratios, / resized (w, h))."""
    boxes = _req(boxes, torch.float32, "boxes").reshape(-1, 4)
    out = torch.empty_like(boxes)
    with torch.cuda.device(boxes.device):
        check(lib.frcnn_preprocess_boxes(_ptr(boxes), boxes.shape[0], int(src_wh[0]), int(src_wh[1]), int(bool(flip)), int(out_wh[0]),
                                         int(out_wh[1]), _ptr(out), _stream()), "preprocess_boxes")
    return out


class DeviceInputStage:
    """Compose([RandomHorizontalFlip, Resize(size, max_size), ToTensor, Normalize]) + the batch-1 collate, on the device.

    stage(img_u8_hwc, boxes_xyxy_pixels=None, flip=False) -> (x [1, 3, PH, PW] float32, boxes normalised or None,
    {'size': (oh, ow), 'padded': (PH, PW), 'orig_size': (h, w)}).  size_divisible=None skips the pad (the VOC loader)."""

    def __init__(self, size=800, max_size=1333, mean=IMAGENET_MEAN, std=IMAGENET_STD, size_divisible=32):
        self.size, self.max_size, self.mean, self.std, self.size_divisible = size, max_size, tuple(mean), tuple(std), size_divisible

    def __call__(self, img, boxes=None, flip=False):
        h, w = int(img.shape[0]), int(img.shape[1])
        oh, ow = get_size((w, h), self.size, self.max_size)
        ph, pw = padded_size(oh, ow, self.size_divisible) if self.size_divisible else (oh, ow)
        x = preprocess_image(img, (oh, ow), (ph, pw), flip, self.mean, self.std)
        b = preprocess_boxes(boxes, (w, h), (ow, oh), flip) if boxes is not None else None
        return x[None], b, {"size": (oh, ow), "padded": (ph, pw), "orig_size": (h, w)}
