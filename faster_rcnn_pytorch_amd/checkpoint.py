"""Checkpoint compatibility with the reference.

The reference writes  {'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict'}  with torch.save to
<log_dir>/<name>/saves/<name>.<epoch>.pth.tar (train.py:80-85), resumes from it (utils/util.py:142-155) and, for the
published weights, strips DistributedDataParallel's 'module.' from every key before load_state_dict
(models/model_.py:305-312).  The mirrors in model.py / new_model.py keep the reference's sub-module names -- including
the VGG head's classifier being registered twice ('classifier.*' and 'fast_rcnn_head.classifier.*', model_.py:282-297)
-- so those files load unchanged; this module is the small amount of host logic around that.
"""
import os
from collections import OrderedDict

import torch


def checkpoint_path(log_dir, name, epoch):
    """train.py:75-85 / utils/util.py:145: <log_dir>/<name>/saves/<name>.<epoch>.pth.tar ('best' is used by test.py:163)."""
    return os.path.join(log_dir, name, "saves", "%s.%s.pth.tar" % (name, epoch))


def strip_module_prefix(state_dict):
    """models/model_.py:308-311: n.replace('module.', '') on every key (DDP / DataParallel wrappers)."""
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k.replace("module.", "")] = v
    return out


def save_checkpoint(path, epoch, model, optimizer=None, scheduler=None):
    """Write the reference's checkpoint dict.  `model` may be DDP-wrapped: like the reference (train.py:81) its keys are
    saved as they are, so a file written from a wrapped model carries 'module.' exactly as the reference's files do."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = {"epoch": epoch, "model_state_dict": model.state_dict()}
    if optimizer is not None:
        ckpt["optimizer_state_dict"] = optimizer.state_dict()
    if scheduler is not None:
        ckpt["scheduler_state_dict"] = scheduler.state_dict()
    torch.save(ckpt, path)
    return path


def load_reference_checkpoint(model, ckpt, optimizer=None, scheduler=None, map_location="cpu", strict=True):
    """Load a reference .pth.tar (path or already-loaded dict) into a mirror model.

    Accepts the full checkpoint dict or a bare state dict; strips 'module.' unless the target itself is wrapped
    (then keys are matched as given).  Returns the stored epoch (or None).  Missing / unexpected keys raise as
    load_state_dict does when strict."""
    if isinstance(ckpt, (str, bytes, os.PathLike)):
        ckpt = torch.load(ckpt, map_location=map_location, weights_only=False)
    state = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    wrapped = hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module)
    target = model.module if wrapped else model
    target.load_state_dict(strip_module_prefix(state), strict=strict)
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return ckpt.get("epoch") if isinstance(ckpt, dict) else None


def resume(log_dir, name, start_epoch, model, optimizer=None, scheduler=None, map_location="cpu"):
    """utils/util.py:142-155: start_epoch != 0 -> load <name>.<start_epoch-1>.pth.tar; returns True when something was loaded."""
    if start_epoch == 0:
        return False
    load_reference_checkpoint(model, checkpoint_path(log_dir, name, start_epoch - 1), optimizer, scheduler, map_location)
    return True
