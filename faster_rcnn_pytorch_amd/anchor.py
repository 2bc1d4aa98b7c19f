"""FRCNNAnchorMaker -- host-side mirror of the reference's anchor.py:7-55.

Same constructor, same attributes (`anchor_base` [9,4] np.float32), same method
`_enumerate_shifted_anchor((H, W)) -> np.float32 [fh*fw*9, 4]` (normalised xyxy), but the grid is
produced by the HIP anchor_grid kernel and cached per image shape in HBM: the reference recomputes it
with numpy on the host and copies 330 KB to the device on EVERY forward (models/model.py:310-312).
`device_anchors()` hands the resident tensor to the model; `grid_desc()` feeds the fused proposal
prologue, which regenerates anchors in registers and never reads them from memory at all.
"""
import numpy as np

from . import ops


class FRCNNAnchorMaker(object):
    def __init__(self, base_size=16, ratios=[0.5, 1, 2], anchor_scales=[8, 16, 32]):
        self.base_size = base_size
        self.ratios = ratios
        self.anchor_scales = anchor_scales
        self.anchor_base = self.generate_anchor_base()
        self._cache = {}

    def generate_anchor_base(self):
        return ops.anchor_base(self.base_size, self.ratios, self.anchor_scales)       # anchor.py:15-32

    def feature_size(self, origin_image_size):
        h, w = int(origin_image_size[0]), int(origin_image_size[1])
        return h // self.base_size, w // self.base_size

    def device_anchors(self, origin_image_size, device):
        h, w = int(origin_image_size[0]), int(origin_image_size[1])
        key = (h, w, str(device))
        a = self._cache.get(key)
        if a is None:
            fh, fw = self.feature_size((h, w))
            a = ops.anchor_grid([(fh, fw)], [(self.base_size, self.base_size)], self.anchor_base[None], w, h, device)
            self._cache[key] = a
        return a

    def grid_desc(self, origin_image_size):
        """(fh, fw, stride, base[A,4], div_w, div_h) for ops.region_proposal(grid=...)."""
        h, w = int(origin_image_size[0]), int(origin_image_size[1])
        fh, fw = self.feature_size((h, w))
        return fh, fw, self.base_size, self.anchor_base, float(w), float(h)

    def _enumerate_shifted_anchor(self, origin_image_size, device="cuda"):
        """Reference signature (anchor.py:34-55): returns a host numpy array."""
        return self.device_anchors(origin_image_size, device).cpu().numpy().astype(np.float32)
