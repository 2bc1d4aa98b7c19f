"""faster_rcnn_pytorch_amd -- MI355X-native (gfx950) Faster R-CNN proposal / RoI-head hot path.

Drop-in for the path behind models/model.py + anchor.py of csm-kr/faster_rcnn_pytorch:
hand-written HIP kernels behind the C ABI of include/frcnn_hip.h (libfrcnn_hip.so), plus the
Python host code that mirrors the reference's module API.  Importing the package loads the
shared library and fails loudly if it has not been built.
"""
from . import _lib  # noqa: F401  (raises ImportError when libfrcnn_hip.so is missing)

__all__ = ["_lib"]
