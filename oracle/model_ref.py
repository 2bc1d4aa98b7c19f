"""CPU ORACLE of the model-level path (test infrastructure; see oracle/frcnn_oracle.c header).

A torch-CPU + oracle-C restatement of models/model_.py:304-344 (FRCNN.forward, VGG16), used
  * by tests/ and __graft_entry__.smoke() to check the HIP-backed model stage by stage on identical
    inputs (the conv/FC layers are plain torch on both sides; the path stages go through oracle.py);
  * by bench.py's cpu_baseline leg (kind "port"): the whole training step on the host cores.
It draws torch.randperm from the CPU default generator in the reference's order (RPN maker first,
model_.py:324 then :328).  Never imported by faster_rcnn_pytorch_amd/.
"""
import numpy as np
import torch
import torch.nn as nn

from . import oracle as orc


def path_forward(features, rpn_cls, rpn_reg, bbox, label, hw, mode="train"):
    """Steps 2,4,5,6 + the RoIPool of step 7 of models/model_.py:304-344 on numpy inputs.
    features [C,fh,fw]; rpn_cls [N,2]; rpn_reg [N,4]; bbox [G,4]; label [G]; hw = (H, W)."""
    H, W = int(hw[0]), int(hw[1])
    anchor = orc.anchor_grid(H, W)                                             # model_.py:310
    K, P = (12000, 2000) if mode == "train" else (6000, 300)
    rois, src = orc.region_proposal(rpn_reg, rpn_cls, anchor, 1 / 1000, K, 0.7, P)   # model_.py:318
    out = {"anchor": anchor, "rois": rois, "src": src}
    if mode != "train":
        return out
    _, _, (n_pos, n_neg) = orc.rpn_targets(anchor, bbox)                       # model_.py:324
    pp = torch.randperm(n_pos).numpy() if n_pos > 128 else None
    pn = torch.randperm(n_neg).numpy() if n_neg > 256 - n_pos else None
    t_rpn_cls, t_rpn_reg, _ = orc.rpn_targets(anchor, bbox, pp, pn)
    npc, nnc = orc.head_target_counts(rois, bbox, label)                       # model_.py:328
    hp = torch.randperm(npc).numpy()
    hn = torch.randperm(nnc).numpy()
    t_cls, t_reg, srois, keep = orc.head_targets(rois, bbox, label, hp, hn)
    fh, fw = features.shape[1:]
    scaled = srois * np.array([fw, fh, fw, fh], np.float32)                     # model_.py:107-109
    pool, argmax = orc.roi_pool_fwd(features, scaled, 7, 7, 1.0)               # model_.py:113
    out.update(t_rpn_cls=t_rpn_cls, t_rpn_reg=t_rpn_reg, t_cls=t_cls, t_reg=t_reg, sample_rois=srois, keep=keep,
               pool=pool, argmax=argmax, counts=(n_pos, n_neg, npc, nnc))
    return out


def fpn_path(feats4, shapes5, rpn_cls, rpn_reg, bbox, label, hw, mode="train"):
    """models/new_model.py:391-418 hot-path stages on numpy inputs: feats4 = levels '0'..'3' as [C,h,w]; shapes5 = (h,w) of all 5 maps."""
    H, W = int(hw[0]), int(hw[1])
    anchor = orc.tv_anchor_grid(H, W, shapes5, normalise=True)                 # new_model.py:46-47
    K, P = (4000, 1000) if mode == "train" else (2000, 1000)
    rois, src = orc.region_proposal(rpn_reg, rpn_cls, anchor, 10 / 1000, K, 0.7, P)   # new_model.py:49-84
    out = {"anchor": anchor, "rois": rois, "src": src}
    if mode != "train":
        return out
    _, _, (n_pos, n_neg) = orc.rpn_targets(anchor, bbox, variant=1)            # new_model.py:400
    pp = torch.randperm(n_pos).numpy() if n_pos > 128 else None
    pn = torch.randperm(n_neg).numpy() if n_neg > 256 - n_pos else None
    t_rpn_cls, t_rpn_reg, _ = orc.rpn_targets(anchor, bbox, pp, pn, variant=1)
    npc, nnc = orc.head_target_counts(rois, bbox, label, variant=1)            # new_model.py:404
    hp, hn = torch.randperm(npc).numpy(), torch.randperm(nnc).numpy()
    t_cls, t_reg, srois, keep = orc.head_targets(rois, bbox, label, hp, hn, variant=1, label_offset=0, max_pos=128, total=512)
    scaled = srois * np.array([W, H, W, H], np.float32)                        # new_model.py:136-140
    pool, lv = orc.ms_roi_align(feats4, scaled)                                # new_model.py:143
    out.update(t_rpn_cls=t_rpn_cls, t_rpn_reg=t_rpn_reg, t_cls=t_cls, t_reg=t_reg, sample_rois=srois, keep=keep, pool=pool, level=lv)
    return out


class _RefRoIPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois):
        f = feat.detach().numpy()[0]
        out, arg = orc.roi_pool_fwd(f, rois.numpy(), 7, 7, 1.0)
        ctx.arg = arg
        ctx.shape = f.shape
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, go):
        C, H, W = ctx.shape
        return torch.from_numpy(orc.roi_pool_bwd(go.contiguous().numpy(), ctx.arg, C, H, W))[None], None


def _vgg16_features():
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    layers, c = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
            c = v
    return layers


class RefFRCNN(nn.Module):
    """CPU restatement of models/model_.py:269-344 with the same sub-module names, so a state_dict from the
    HIP-backed model loads directly."""

    def __init__(self, num_classes=21):
        super().__init__()
        self.num_classes = num_classes
        self.extractor = nn.Sequential(*_vgg16_features()[:-1])
        self.classifier = nn.Sequential(nn.Linear(25088, 4096), nn.ReLU(inplace=True), nn.Linear(4096, 4096), nn.ReLU(inplace=True))
        self.rpn = nn.Module()
        self.rpn.inter_layer = nn.Conv2d(512, 512, 3, padding=1)
        self.rpn.cls_layer = nn.Conv2d(512, 18, 1)
        self.rpn.reg_layer = nn.Conv2d(512, 36, 1)
        self.fast_rcnn_head = nn.Module()
        self.fast_rcnn_head.cls_head = nn.Linear(4096, num_classes)
        self.fast_rcnn_head.reg_head = nn.Linear(4096, num_classes * 4)
        self.fast_rcnn_head.classifier = self.classifier

    def forward(self, x, bbox, label):
        features = self.extractor(x)
        h = torch.relu(self.rpn.inter_layer(features))
        pred_cls = self.rpn.cls_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 2)
        pred_reg = self.rpn.reg_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 4)
        bbox = bbox[0] if isinstance(bbox, (list, tuple)) else bbox
        label = label[0] if isinstance(label, (list, tuple)) else label
        p = path_forward(features.detach().numpy()[0], pred_cls.detach().numpy()[0], pred_reg.detach().numpy()[0],
                         bbox.numpy(), label.numpy().astype(np.int64), x.shape[2:])
        fh, fw = features.shape[2:]
        scaled = torch.from_numpy(p["sample_rois"] * np.array([fw, fh, fw, fh], np.float32))
        pool = _RefRoIPool.apply(features, scaled)
        z = self.classifier(pool.view(pool.size(0), -1))
        head_cls = self.fast_rcnn_head.cls_head(z)
        head_reg = self.fast_rcnn_head.reg_head(z).reshape(128, -1, 4)
        t_cls = torch.from_numpy(p["t_cls"])
        head_reg = head_reg[torch.arange(128), t_cls]
        return (pred_cls, pred_reg, head_cls, head_reg), \
               (torch.from_numpy(p["t_rpn_cls"]), torch.from_numpy(p["t_rpn_reg"]), t_cls, torch.from_numpy(p["t_reg"]))


class MaskPopper(nn.Module):
    """A stand-in for nn.ReLU on the CPU side of the same-weights tests: multiplies by the next mask of a list recorded on ANOTHER run of the same
    network (the device's), in call order, instead of deciding `x > 0` itself.  Every ReLU / max-pool / RoIPool-argmax is a DECISION: two fp32
    evaluations that agree to 1e-6 still disagree about the sign of ~1e-5 of the pre-activations (those within rounding of zero), and one such
    element moves a weight gradient -- a sum of random-sign terms -- by a whole dy * x: a fraction f of flipped decisions shows as ~sqrt(f) of the
    gradient's scale (measured 1e-3 .. 7e-3 between ANY two fp32 evaluations, the reference's CPU path against float64 included).  With the
    decisions handed over, what remains is linear arithmetic and can be compared at 1e-4.  `flips` counts how many decisions this run would have taken
    differently, and how far from zero the largest such pre-activation was (relative to the tensor's scale)."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.k, self.flips, self.total, self.worst = masks, 0, 0, 0, 0.0

    def forward(self, x):
        m = self.masks[self.k]
        self.k += 1
        assert tuple(m.shape) == tuple(x.shape), "decision %d was recorded for %s, this run has %s" % (self.k - 1, tuple(m.shape), tuple(x.shape))
        with torch.no_grad():
            d = m != (x > 0)
            n = int(d.sum())
            self.flips += n
            self.total += x.numel()
            if n:
                self.worst = max(self.worst, float(x[d].abs().max()) / max(float(x.abs().max()), 1e-30))
        return x * m.to(x.dtype)


def transplanted_roi_pool(features, argmax):
    """RoIPool with the bins' argmax handed in (int [R,C,7,7] flat indices into a channel plane, -1 = empty bin): out[r,c,b] = feat[c, argmax[r,c,b]]."""
    C_ = features.shape[1]
    f = features[0].reshape(C_, -1)
    idx = argmax.permute(1, 0, 2, 3).reshape(C_, -1)
    vals = f.gather(1, idx.clamp(min=0).long()) * (idx >= 0).to(f.dtype)
    R = argmax.shape[0]
    return vals.reshape(C_, R, argmax.shape[2], argmax.shape[3]).permute(1, 0, 2, 3).contiguous()


def vgg_extractor_with_decisions(ref, x, decisions):
    """ref.extractor (nn.Sequential(*vgg16.features[:-1]), models/model.py:279-281) on the CPU with every ReLU sign and max-pool selection taken
    from `decisions`, one entry per convolution in order: ("mask", bool [1,C,H,W]) for conv + ReLU, ("pool", sel [1,C,H,W]) for conv + ReLU +
    MaxPool2d(2, 2) where sel marks the ONE pixel of each window that carries it (none: the window's maximum was not positive)."""
    import torch.nn.functional as F
    mods, h, i, k, flips = list(ref.extractor), x, 0, 0, []
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            y = m(h)
            kind, dec = decisions[k]
            k += 1
            own = (y > 0)
            if kind == "pool":
                assert isinstance(mods[i + 1], nn.ReLU) and isinstance(mods[i + 2], nn.MaxPool2d)
                Hp, Wp = y.shape[2] // 2, y.shape[3] // 2
                with torch.no_grad():
                    act = y.clamp_min(0)
                    _, idx = F.max_pool2d(act, 2, 2, return_indices=True)
                    own_sel = torch.zeros(act.shape[1], act.shape[2] * act.shape[3], dtype=torch.bool).scatter_(
                        1, idx.view(act.shape[1], -1), (F.max_pool2d(act, 2, 2) > 0).view(act.shape[1], -1)).view_as(act)
                    flips.append((int((own_sel != dec).sum()) // 2, y.numel()))
                h = F.avg_pool2d(y * dec.to(y.dtype), 2) * 4                  # exactly the selected value of each window (one non-zero term, powers of two)
                assert h.shape[2:] == (Hp, Wp)
                i += 3
            else:
                assert isinstance(mods[i + 1], nn.ReLU)
                flips.append((int((own != dec).sum()), y.numel()))
                h = y * dec.to(y.dtype)
                i += 2
        else:
            h = m(h)                                                       # a max-pool the device did not fuse: on its own values on either side
            i += 1
    assert k == len(decisions)
    return h, flips


def ref_forward_fixed_vgg(ref, x, sample_rois, t_cls, decisions=None):
    """The reference's CPU form of models/model_.py:304-341 with the SAMPLED RoIs and head classes handed in (from another run of the same step)
    instead of re-derived: extractor -> RPN head (:307,:315), RoIPool of the given RoIs -> classifier -> heads -> class-row gather (:335-341).
    What the same-weights tests compare the HIP-backed model's features / predictions / parameter gradients with: every value and gradient
    depends only on the network arithmetic, not on the sort / NMS / sampling decisions (which have their own bit-exact tests).
    decisions (optional) = dict(extractor=[...] for vgg_extractor_with_decisions, rpn=bool mask of the RPN's ReLU, argmax=RoIPool argmax,
    classifier=[two bool masks]): the ReLU / pooling decisions of that other run (see MaskPopper).
    Returns (features, (rpn_cls [1,N,2], rpn_reg [1,N,4], head_cls [128,C], head_reg [128,4]), flips | None)."""
    flips = None
    if decisions is None:
        features = ref.extractor(x)
        h = torch.relu(ref.rpn.inter_layer(features))
    else:
        features, fl = vgg_extractor_with_decisions(ref, x, decisions["extractor"])
        pop = MaskPopper([decisions["rpn"]])
        h = pop(ref.rpn.inter_layer(features))
        flips = {"extractor": fl, "rpn": (pop.flips, pop.total)}
    pred_cls = ref.rpn.cls_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 2)
    pred_reg = ref.rpn.reg_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 4)
    fh, fw = features.shape[2:]
    scaled = sample_rois.to(torch.float32) * torch.tensor([fw, fh, fw, fh], dtype=torch.float32)      # model_.py:107-109, fp32 like the reference
    if decisions is None:
        pool = _RefRoIPool.apply(features, scaled)
        z = ref.classifier(pool.view(pool.size(0), -1))
    else:
        pool = transplanted_roi_pool(features, decisions["argmax"])
        pop = MaskPopper(decisions["classifier"])
        z = pool.view(pool.size(0), -1)
        for m in ref.classifier:
            z = pop(z) if isinstance(m, nn.ReLU) else m(z)
        flips["classifier"] = (pop.flips, pop.total)
    head_cls = ref.fast_rcnn_head.cls_head(z)
    R = sample_rois.shape[0]
    head_reg = ref.fast_rcnn_head.reg_head(z).reshape(R, -1, 4)[torch.arange(R), t_cls.clamp(min=0)]
    return features, (pred_cls, pred_reg, head_cls, head_reg), flips


def ref_forward_fixed_fpn(ref, x, sample_rois, t_cls, decisions=None):
    """models/new_model.py:391-412 on the CPU with the sampled RoIs / head classes handed in (see ref_forward_fixed_vgg).  decisions (optional) =
    dict(backbone=[bool masks of the body's ReLUs in call order], rpn=[one mask per level], classifier=[two masks]): the caller has already put
    a MaskPopper over the first list in place of the body's nn.ReLU modules; the other two are applied here.
    Returns ([five feature maps], (rpn_cls [1,N,2], rpn_reg [1,N,4], head_cls [512,C], head_reg [512,4]), flips | None)."""
    features = ref.backbone(x)
    feats = list(features.values())
    cls, reg = [], []
    pop_r = MaskPopper(decisions["rpn"]) if decisions else None
    for f in feats:
        raw = ref.rpn_head.inter_layer(f)
        h = pop_r(raw) if decisions else torch.relu(raw)
        cls.append(ref.rpn_head.cls_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 2))
        reg.append(ref.rpn_head.reg_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 4))
    pred_cls, pred_reg = torch.cat(cls, dim=1), torch.cat(reg, dim=1)
    H, W = x.shape[2:]
    scaled = sample_rois.to(torch.float32) * torch.tensor([W, H, W, H], dtype=torch.float32)          # new_model.py:136-140
    pool = _RefMsRoIAlign.apply(scaled, *feats[:4])
    z = pool.view(pool.size(0), -1)
    pop_c = MaskPopper(decisions["classifier"]) if decisions else None
    for m in ref.classifier:
        z = pop_c(z) if (decisions and isinstance(m, nn.ReLU)) else m(z)
    head_cls = ref.cls_head(z)
    R = sample_rois.shape[0]
    head_reg = ref.reg_head(z).reshape(R, -1, 4)[torch.arange(R), t_cls.clamp(min=0)]
    flips = {"rpn": (pop_r.flips, pop_r.total), "classifier": (pop_c.flips, pop_c.total)} if decisions else None
    return feats, (pred_cls, pred_reg, head_cls, head_reg), flips


class _RefMsRoIAlign(torch.autograd.Function):
    """MultiScaleRoIAlign (models/new_model.py:127,143) through the C oracle, with its backward (per level scatter)."""

    @staticmethod
    def forward(ctx, rois, *feats):
        fs = [f.detach().numpy()[0] for f in feats]
        out, lv = orc.ms_roi_align(fs, rois.numpy())
        ctx.rois, ctx.lv, ctx.shapes = rois.numpy(), lv, [f.shape for f in fs]
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, go):
        g = go.contiguous().numpy()
        scales = (0.25, 0.125, 0.0625, 0.03125)
        grads = [torch.from_numpy(orc.roi_align_bwd(g, ctx.shapes[l], ctx.rois, scales[l], 2, False, ctx.lv, l))[None] for l in range(len(ctx.shapes))]
        return (None,) + tuple(grads)


class RefFRCNNFPN(nn.Module):
    """CPU restatement of models/new_model.py:366-418 (ResNet-50-FPN training forward).  The backbone is handed in (a plain
    torch module, e.g. resnet_fpn_backbone or the build's from-scratch definition of it): conv layers are torch on both sides,
    only the path stages go through the oracle."""

    def __init__(self, backbone, num_classes=91):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = backbone
        self.classifier = nn.Sequential(nn.Linear(12544, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 1024), nn.ReLU(inplace=True))
        self.rpn_head = nn.Module()
        self.rpn_head.inter_layer = nn.Conv2d(256, 256, 3, padding=1)
        self.rpn_head.cls_layer = nn.Conv2d(256, 6, 1)
        self.rpn_head.reg_layer = nn.Conv2d(256, 12, 1)
        self.cls_head = nn.Linear(1024, num_classes)
        self.reg_head = nn.Linear(1024, num_classes * 4)

    def forward(self, x, boxes, labels):
        features = self.backbone(x)                                             # new_model.py:394
        feats = list(features.values())
        cls, reg = [], []
        for f in feats:                                                         # new_model.py:37-44
            h = torch.relu(self.rpn_head.inter_layer(f))
            cls.append(self.rpn_head.cls_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 2))
            reg.append(self.rpn_head.reg_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 4))
        pred_cls, pred_reg = torch.cat(cls, dim=1), torch.cat(reg, dim=1)
        boxes = boxes[0] if isinstance(boxes, (list, tuple)) else boxes
        labels = labels[0] if isinstance(labels, (list, tuple)) else labels
        H, W = x.shape[2:]
        p = fpn_path([f.detach().numpy()[0] for f in feats[:4]], [tuple(f.shape[-2:]) for f in feats], pred_cls.detach().numpy()[0],
                     pred_reg.detach().numpy()[0], boxes.numpy(), labels.numpy().astype(np.int64), (H, W))
        scaled = torch.from_numpy(p["sample_rois"] * np.array([W, H, W, H], np.float32))
        pool = _RefMsRoIAlign.apply(scaled, *feats[:4])
        z = self.classifier(pool.view(pool.size(0), -1))
        head_cls = self.cls_head(z)
        t_cls = torch.from_numpy(p["t_cls"])
        head_reg = self.reg_head(z).reshape(512, -1, 4)[torch.arange(512), t_cls]
        return (pred_cls, pred_reg, head_cls, head_reg), \
               (torch.from_numpy(p["t_rpn_cls"]), torch.from_numpy(p["t_rpn_reg"]), t_cls, torch.from_numpy(p["t_reg"]))


def ref_suppress(raw_cls_bbox, raw_prob, num_classes, thres):
    """FRCNN._suppress (models/model.py:382-402, models/new_model.py:445-470) as the reference writes it: a Python loop over the
    classes 1 .. C-1 (0 = background is skipped), per class the score mask `prob > thres`, torchvision nms(0.3) on the masked
    boxes (nms visits boxes by descending score; ties by ascending index, the build's definition of torch's unstable sort), and
    the CLASS-MAJOR concatenation of boxes / (l - 1) labels / scores.  numpy in, numpy out."""
    boxes = np.ascontiguousarray(raw_cls_bbox, np.float32).reshape(-1, num_classes, 4)
    prob = np.ascontiguousarray(raw_prob, np.float32)
    bbox, label, score = [], [], []
    for l in range(1, num_classes):                                            # model.py:388
        cls_bbox_l = boxes[:, l, :]
        prob_l = prob[:, l]
        mask = prob_l > np.float32(thres)                                      # model.py:391
        cls_bbox_l = np.ascontiguousarray(cls_bbox_l[mask])
        prob_l = prob_l[mask]
        order = np.argsort(-prob_l, kind="stable")
        keep = orc.nms(cls_bbox_l, 0.3, order) if len(prob_l) else np.zeros((0,), np.int64)   # model.py:394
        bbox.append(cls_bbox_l[keep])
        label.append((l - 1) * np.ones((len(keep),)))                          # model.py:396
        score.append(prob_l[keep])
    return (np.concatenate(bbox, axis=0).astype(np.float32), np.concatenate(label, axis=0).astype(np.int32),
            np.concatenate(score, axis=0).astype(np.float32))


def ref_predict_post(head_cls, head_reg, rois, num_classes, thres, prob=None):
    """The post-processing half of FRCNN.predict (models/model.py:368-380, models/new_model.py:431-443) on numpy inputs:
    softmax over the head's class logits (torch CPU, as the reference's eager op), regression * (0.1, 0.1, 0.2, 0.2)
    (SURVEY Q10), every class decoded against its RoI (decode / xy_to_cxcy / cxcy_to_xy of utils/util.py), clamp to [0, 1],
    then _suppress.  `prob` (optional) overrides the softmax output: tests pass the device's own softmax there to separate the
    last-bit differences of two softmax implementations from the logic under test.
    Returns (bbox [M,4] f32, label [M] i32, score [M] f32, raw_bbox [R, C*4], prob [R, C])."""
    head_cls = np.ascontiguousarray(head_cls, np.float32)
    head_reg = np.ascontiguousarray(head_reg, np.float32).reshape(-1, num_classes, 4)
    rois = np.ascontiguousarray(rois, np.float32).reshape(-1, 4)
    if prob is None:
        prob = torch.softmax(torch.from_numpy(head_cls), dim=-1).numpy()       # model.py:369
    t = head_reg * np.array([0.1, 0.1, 0.2, 0.2], np.float32)                  # model.py:372
    r = np.ascontiguousarray(np.broadcast_to(rois.reshape(-1, 1, 4), t.shape)).reshape(-1, 4)   # model.py:373
    pred = orc.cxcy_to_xy(orc.decode(np.ascontiguousarray(t.reshape(-1, 4)), orc.xy_to_cxcy(r)))   # model.py:374-375
    pred = np.clip(pred.reshape(-1, num_classes * 4), np.float32(0), np.float32(1))                 # model.py:377-378 (NaN stays NaN)
    bbox, label, score = ref_suppress(pred, prob, num_classes, thres)
    return bbox, label, score, pred, prob


def ref_loss(pred, target):
    """losses/loss.py:5-85 in torch (CPU), written as the reference writes it (boolean-mask indexing)."""
    import torch.nn.functional as F
    rc, rr, hc, hr = pred
    trc, trr, thc, thr = target

    def sl1(p, t, beta):
        x = (p - t).abs()
        return torch.where(x >= beta, x - 0.5 * beta, 0.5 * x ** 2 / beta)
    l1 = F.cross_entropy(rc.squeeze(0), trc, ignore_index=-1)
    l2 = sl1(rr.squeeze(0)[trc > 0], trr[trc > 0], 1 / 9).sum() / (trc >= 0).sum()
    l3 = F.cross_entropy(hc, thc)
    l4 = sl1(hr[thc > 0], thr[thc > 0], 1.0).sum() / (thc >= 0).sum()
    return l1 + l2 + l3 + l4, l1, l2, l3, l4
