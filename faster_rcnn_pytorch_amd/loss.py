"""Detection loss with the interface of the reference's losses/loss.py:64-85 (`FRCNNLoss(opts)(pred, target)` -> the
5-tuple total, rpn_cls, rpn_reg, fast_rcnn_cls, fast_rcnn_reg), SURVEY 8(f) rank 1.

On fp32 HIP tensors the four terms AND their gradients come from one fused kernel (`ops.detection_loss`,
csrc/loss.hip).  Anything else (CPU tensors in the gloo tests, non-fp32 predictions) takes `detection_loss_torch`, which
states the same four formulas (loss.py:20-40 RPN, :43-61 head) once, as masked sums: the reference's boolean-mask indexing
`pred_reg[target_cls > 0]` forces a nonzero() host sync per step and is algebraically the same thing.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _masked_smooth_l1(pred, target, labels, beta):
    """sum over the rows with label > 0 of smooth-L1(pred - target; beta), divided by the number of rows with label >= 0."""
    d = (pred - target).abs()
    per_elem = torch.where(d >= beta, d - 0.5 * beta, d * d * (0.5 / beta))
    positive = (labels > 0).to(per_elem.dtype)[:, None]
    return (per_elem * positive).sum() / (labels >= 0).sum()


def detection_loss_torch(pred, target):
    rpn_cls, rpn_reg, head_cls, head_reg = (p.squeeze(0) if p.dim() == 3 else p for p in pred)
    t_rpn_cls, t_rpn_reg, t_head_cls, t_head_reg = target
    l_rpn_cls = F.cross_entropy(rpn_cls, t_rpn_cls, ignore_index=-1)          # anchors labelled -1 are not sampled
    l_rpn_reg = _masked_smooth_l1(rpn_reg, t_rpn_reg, t_rpn_cls, 1.0 / 9.0)     # beta 1/9, normalised by #(label >= 0)
    l_head_cls = F.cross_entropy(head_cls, t_head_cls)
    l_head_reg = _masked_smooth_l1(head_reg, t_head_reg, t_head_cls, 1.0)       # beta 1, normalised by R
    return l_rpn_cls + l_rpn_reg + l_head_cls + l_head_reg, l_rpn_cls, l_rpn_reg, l_head_cls, l_head_reg


class FRCNNLoss(nn.Module):
    def __init__(self, opts=None):
        super().__init__()
        self.opts = opts

    def forward(self, pred, target):
        if all(t.is_cuda and t.dtype == torch.float32 for t in pred):
            from . import ops
            return ops.detection_loss(pred, target)                             # 2 launches instead of ~15 + two host syncs
        return detection_loss_torch(pred, target)


def build_loss(opts=None):
    return FRCNNLoss(opts)
