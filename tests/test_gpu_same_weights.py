"""Same weights, same frame, both sides END TO END (VERDICT r4 "next" 1): ONE state_dict is loaded into the HIP-backed mirror and into the
reference's CPU form (oracle/model_ref.RefFRCNN / RefFRCNNFPN: torch-CPU fp32 convolutions and linear layers -- what models/model.py:279-281,
304-316 and models/new_model.py:372,391-400 run on a CPU -- + the C oracle's RoIPool / MultiScaleRoIAlign), at the sizes bench.py runs.

The model-level tests in test_gpu_model.py push the GPU model's OWN features through the oracle's path stages: they pin the sort / NMS /
target / pooling decisions, but a wrong extractor could not fail them.  Here nothing is shared but the weights, the frame, the sampled RoIs and
the targets (taken from the device run; they have their own bit-exact tests).  Compared:
  * the extractor's features (13 Winograd-stage layers with fused ReLU / max-pool words at 600 x 1000; the FPN's five maps at 800 x 1344),
  * the RPN's softmax scores and box regressions -- the tensors BASELINE.json's north_star bounds at 1e-4 of the reference's CPU path,
  * the head's outputs, the four losses,
  * after ONE loss.backward(): every extractor / backbone, RPN, classifier and head parameter gradient.

Gradients are compared twice.  (i) Each side taking its own ReLU / max-pool / RoIPool-argmax decisions: two fp32 evaluations that agree to 1e-6
still disagree about ~1e-5 of those decisions (pre-activations within rounding of zero), and a fraction f of flipped decisions moves a weight
gradient -- a sum of random-sign terms -- by ~sqrt(f) of its scale.  That is a property of the network, not of an implementation: the test
measures the same distance between the reference's CPU path and a float64 evaluation, and asserts that the device is no further from float64
than a small multiple of that.  (ii) With the device's decisions handed to the CPU run (oracle/model_ref.MaskPopper & co): what remains is
linear arithmetic, and THERE every gradient must agree within 1e-4 of its scale; the number of decisions the CPU would have taken differently,
and how close to zero they all were, is asserted too.
The measured maxima are written to gpurun_out/same_weights_<config>.json (quoted in DESIGN.md section 2)."""
import copy
import json
import os

import pytest
import torch

from oracle import model_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4            # north_star: box / score tensors within 1e-4 of the reference's CPU path; the same bound, relative to each tensor's scale, on gradients


def synth(seed, H, W, G, label_lo=0, label_hi=20):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 3, H, W, generator=g)
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    labels = torch.randint(label_lo, label_hi, (G,), generator=g)
    return x, boxes, labels


def rel(a, b):
    """max |a - b| over the scale of b (its largest magnitude)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _report(name, rec):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "same_weights_%s.json" % name), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("\n[same-weights %s] " % name + json.dumps({k: v for k, v in rec.items() if not isinstance(v, dict)}))


def _grads(named):
    return {n: p.grad.detach().clone() for n, p in named if p.requires_grad and p.grad is not None}


def _threads():
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))


def _decode_pool_words(bits, Cout, H, W):
    """The fused conv + ReLU + max-pool's words (one uint16 per channel and 4 x 4 tile; window k = wi * 2 + wj: bits 3k, 3k + 1 = position of
    the maximum in scan order, bit 3k + 2 = maximum > 0) -> bool [1,Cout,H,W]: the pixel of each 2 x 2 window that carries its gradient."""
    th, tw = (H + 3) // 4, (W + 3) // 4
    words = bits.cpu().view(Cout, -1)[:, :th * tw].to(torch.int32).bitwise_and(0xFFFF).view(Cout, th, tw)
    sel = torch.zeros(Cout, 4 * th, 4 * tw, dtype=torch.bool)
    for k in range(4):
        w3 = (words >> (3 * k)) & 7
        for pos in range(4):
            sel[:, (k // 2) * 2 + pos // 2::4, (k % 2) * 2 + pos % 2::4] = (w3 == (4 | pos))
    return sel[:, :H, :W][None].contiguous()


def _loss_list(t):
    return [float(v.detach()) for v in t]


@pytest.fixture(params=["native", "split"])
def products(request):
    """Both forms of the fp32 conv stage's products (ops.conv3x3_f32_products): the fp32 matrix instruction (the default) and the opt-in split form --
    the same end-to-end bounds hold for either; the split form's maxima go to gpurun_out/same_weights_<config>_split_products.json."""
    from faster_rcnn_pytorch_amd import ops
    prev = ops.conv3x3_f32_products(request.param)
    yield request.param
    ops.conv3x3_f32_products(prev)


def test_vgg16_600x1000_same_weights_features_rpn_outputs_and_gradients_vs_the_cpu_form(monkeypatch, products):
    from faster_rcnn_pytorch_amd import ops
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    from faster_rcnn_pytorch_amd.model import FRCNN
    H, W = 600, 1000
    torch.manual_seed(0)
    m = FRCNN(num_classes=21, sampling="host").to(DEV)
    with torch.no_grad():                                      # spread the RPN's outputs (N(0, 0.01) heads give scores ~0.5 everywhere)
        m.rpn.cls_layer.weight.mul_(30)
        m.rpn.reg_layer.weight.mul_(10)
    ref = model_ref.RefFRCNN(21)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})       # ONE state_dict, both sides
    x, boxes, labels = synth(1, H, W, 4)

    # ---- the device run, with every ReLU / pooling decision it takes recorded on the way
    dec, stash, cap = [], {}, {}
    o_fwd, o_c, o_c3, o_rpn = ops.conv3x3_fwd, ops.conv3x3, ops.conv3x3_c3, ops.rpn_conv3x3

    def spy_fwd(xs, w, bias=None, relu=False, **kw):
        r = o_fwd(xs, w, bias, relu, **kw)
        if kw.get("pool"):
            stash["bits"] = r[2]
        return r

    def spy_c(x_, weight, bias=None, relu=False, pool=False):
        y = o_c(x_, weight, bias, relu, pool)
        assert relu
        dec.append(("pool", _decode_pool_words(stash.pop("bits"), weight.shape[0], x_.shape[2], x_.shape[3])) if pool else ("mask", (y.detach() > 0).cpu()))
        return y

    def spy_c3(x_, weight, bias=None, relu=False):
        y = o_c3(x_, weight, bias, relu)
        dec.append(("mask", (y.detach() > 0).cpu()))
        return y

    def spy_rpn(feats, w3):
        raws = o_rpn(feats, w3)
        stash["rpn_raw"] = raws[0].detach()
        return raws
    for name, fn in (("conv3x3_fwd", spy_fwd), ("conv3x3", spy_c), ("conv3x3_c3", spy_c3), ("rpn_conv3x3", spy_rpn)):
        monkeypatch.setattr(ops, name, fn)
    cls_masks = []
    hooks = [m.extractor.register_forward_hook(lambda mod, i, o: cap.__setitem__("feat", o)),
             m.fast_rcnn_target_maker.register_forward_hook(lambda mod, i, o: cap.__setitem__("srois", o[2].detach())),
             m.fast_rcnn_head.roi_pool.register_forward_hook(lambda mod, i, o: cap.__setitem__("pool", (i[1][0].detach(), o.detach()))),
             m.classifier[1].register_forward_hook(lambda mod, i, o: cls_masks.append((o.detach() > 0).cpu())),
             m.classifier[3].register_forward_hook(lambda mod, i, o: cls_masks.append((o.detach() > 0).cpu()))]
    m.train()
    torch.manual_seed(101)
    pred, target = m(x.to(DEV), [boxes.to(DEV)], [labels.to(DEV)])
    for h in hooks:
        h.remove()
    monkeypatch.undo()
    assert len(dec) == 13 and [k for k, _ in dec].count("pool") == 4 and len(cls_masks) == 2       # all 13 layers ran on the library, 4 with the fused pool
    loss = FRCNNLoss(None)(pred, target)
    m.zero_grad(set_to_none=True)
    loss[0].backward()
    g_dev = _grads(m.named_parameters())
    scaled_rois, pool_dev = cap["pool"]
    pool_chk, argmax = ops.roi_pool_with_argmax(cap["feat"].detach(), scaled_rois)
    assert torch.equal(pool_chk, pool_dev)                      # the argmax handed over is the one behind the model's own pooled tensor
    decisions = {"extractor": dec, "rpn": ((stash["rpn_raw"] + m.rpn.inter_layer.bias.detach().view(1, -1, 1, 1)) > 0).cpu(),
                 "argmax": argmax.cpu(), "classifier": cls_masks}

    # ---- the reference's CPU form on the same weights, frame, sampled RoIs and targets: (i) its own decisions, (ii) the device's
    _threads()
    t_cpu = [t.detach().cpu() for t in target]
    feat_c, pred_c, _ = model_ref.ref_forward_fixed_vgg(ref, x, cap["srois"].cpu(), t_cpu[2])
    loss_c = model_ref.ref_loss(pred_c, t_cpu)
    loss_c[0].backward()
    g_cpu = _grads(ref.named_parameters())
    ref.zero_grad(set_to_none=True)
    feat_t, pred_t, flips = model_ref.ref_forward_fixed_vgg(ref, x, cap["srois"].cpu(), t_cpu[2], decisions=decisions)
    loss_t = model_ref.ref_loss(pred_t, t_cpu)
    loss_t[0].backward()
    g_tr = _grads(ref.named_parameters())
    # ---- float64, own decisions: the exact values both fp32 paths approximate (forward AND gradients)
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad(set_to_none=True)
    f64, pred64, _ = _vgg_f64(ref64, x, cap["srois"].cpu(), t_cpu)
    g_64 = _grads(ref64.named_parameters())
    sm = lambda t: torch.softmax(t.detach().double().cpu(), dim=-1)          # noqa: E731
    n_flip = sum(f for f, _ in flips["extractor"]) + flips["rpn"][0] + flips["classifier"][0]
    n_dec = sum(n for _, n in flips["extractor"]) + flips["rpn"][1] + flips["classifier"][1]
    rec = {
        "frame": "%dx%d seed 1, 4 boxes" % (H, W),
        "features_rel_dev_vs_cpu": rel(cap["feat"], feat_c), "features_rel_dev_vs_f64": rel(cap["feat"], f64), "features_rel_cpu_vs_f64": rel(feat_c, f64),
        "rpn_score_abs_dev_vs_cpu": float((sm(pred[0]) - sm(pred_c[0])).abs().max()), "rpn_score_abs_dev_vs_f64": float((sm(pred[0]) - sm(pred64[0])).abs().max()),
        "rpn_score_abs_cpu_vs_f64": float((sm(pred_c[0]) - sm(pred64[0])).abs().max()),
        "rpn_reg_rel_dev_vs_cpu": rel(pred[1], pred_c[1]), "rpn_reg_rel_dev_vs_f64": rel(pred[1], pred64[1]), "rpn_reg_rel_cpu_vs_f64": rel(pred_c[1], pred64[1]),
        "rpn_reg_abs_dev_vs_cpu": float((pred[1].detach().cpu() - pred_c[1].detach()).abs().max()),
        "head_cls_rel_dev_vs_cpu": rel(pred[2], pred_c[2]), "head_reg_rel_dev_vs_cpu": rel(pred[3], pred_c[3]),
        "loss_dev": _loss_list(loss), "loss_cpu": _loss_list(loss_c), "loss_cpu_with_device_decisions": _loss_list(loss_t),
        "decisions": n_dec, "decisions_the_cpu_takes_differently": n_flip,
        "flips_per_layer": [f for f, _ in flips["extractor"]] + [flips["rpn"][0], flips["classifier"][0]],
        "grad_rel_own_decisions_dev_vs_cpu": {n: rel(g_dev[n], g_cpu[n]) for n in g_cpu},
        "grad_rel_own_decisions_cpu_vs_f64": {n: rel(g_cpu[n], g_64[n]) for n in g_cpu},
        "grad_rel_own_decisions_dev_vs_f64": {n: rel(g_dev[n], g_64[n]) for n in g_cpu},
        "grad_rel_device_decisions_dev_vs_cpu": {n: rel(g_dev[n], g_tr[n]) for n in g_cpu},
        "features_rel_device_decisions_dev_vs_cpu": rel(cap["feat"], feat_t),
    }
    for k in ("grad_rel_own_decisions_dev_vs_cpu", "grad_rel_own_decisions_cpu_vs_f64", "grad_rel_own_decisions_dev_vs_f64", "grad_rel_device_decisions_dev_vs_cpu"):
        rec[k + "_max"] = max(rec[k].values())
    _report("vgg" if products == "native" else "vgg_split_products", rec)
    assert set(g_dev) == set(g_cpu) == set(g_tr) and len(g_cpu) == 13 * 2 + 3 * 2 + 2 * 2 + 4      # extractor, RPN, heads, classifier (ONE module under two names: listed once)
    assert rec["features_rel_dev_vs_cpu"] < TOL
    assert rec["rpn_score_abs_dev_vs_cpu"] < TOL and rec["rpn_reg_rel_dev_vs_cpu"] < TOL
    assert rec["head_cls_rel_dev_vs_cpu"] < TOL and rec["head_reg_rel_dev_vs_cpu"] < TOL
    for a, b in zip(rec["loss_dev"], rec["loss_cpu"]):
        assert abs(a - b) < TOL * max(1.0, abs(b))
    # (ii) the arithmetic, decisions pinned: every parameter gradient within 1e-4 of its scale; few decisions differ, all at rounding-level pre-activations
    assert rec["grad_rel_device_decisions_dev_vs_cpu_max"] < TOL, sorted(rec["grad_rel_device_decisions_dev_vs_cpu"].items(), key=lambda kv: -kv[1])[:5]
    assert n_flip < 1e-4 * n_dec
    # (i) own decisions: no further from the exact gradient than a small multiple of what the reference's CPU path is itself
    # (measured: device 7.2e-3, CPU path 1.6e-3 -- 64 flipped decisions of 164 M against float64-rounded-differently ones; each flip is a discrete event,
    # so the ratio of two maxima is noisy: the bound is an order of magnitude, the rigorous statement is (ii))
    assert rec["grad_rel_own_decisions_dev_vs_f64_max"] < 10 * rec["grad_rel_own_decisions_cpu_vs_f64_max"] + TOL
    assert rec["grad_rel_own_decisions_dev_vs_cpu_max"] < 5e-2


def _vgg_f64(ref64, x, srois, t_cpu):
    """Forward + loss + backward of the float64 copy (its own decisions; RoIPool by torch ops: the C oracle is binary32)."""
    features = ref64.extractor(x.double())
    h = torch.relu(ref64.rpn.inter_layer(features))
    pred_cls = ref64.rpn.cls_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 2)
    pred_reg = ref64.rpn.reg_layer(h).permute(0, 2, 3, 1).contiguous().view(1, -1, 4)
    fh, fw = features.shape[2:]
    scaled = (srois.to(torch.float32) * torch.tensor([fw, fh, fw, fh], dtype=torch.float32)).numpy()
    pool = _roi_pool_f64(features, scaled)
    z = ref64.classifier(pool.view(pool.size(0), -1))
    head_cls = ref64.fast_rcnn_head.cls_head(z)
    R = srois.shape[0]
    head_reg = ref64.fast_rcnn_head.reg_head(z).reshape(R, -1, 4)[torch.arange(R), t_cpu[2].clamp(min=0)]
    pred = (pred_cls, pred_reg, head_cls, head_reg)
    tg = [t_cpu[0], t_cpu[1].double(), t_cpu[2], t_cpu[3].double()]
    model_ref.ref_loss(pred, tg)[0].backward()
    return features, pred, None


def _roi_pool_f64(features, scaled):
    """torchvision RoIPool's bins (round(x * scale), bin [floor(p w / 7), ceil((p + 1) w / 7)) clipped, max, empty -> 0) on a float64
    map with autograd: adaptive max over each bin."""
    import math
    import numpy as np
    _, C_, H, W = features.shape
    out = []
    for r in scaled:
        x1, y1, x2, y2 = [int(np.floor(np.float32(v) + np.float32(0.5))) if v >= 0 else int(np.ceil(np.float32(v) - np.float32(0.5))) for v in r]   # roundf
        rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        rows = []
        for ph in range(7):
            hs, he = min(max(int(math.floor(ph * rh / 7.0)) + y1, 0), H), min(max(int(math.ceil((ph + 1) * rh / 7.0)) + y1, 0), H)
            cols = []
            for pw in range(7):
                ws, we = min(max(int(math.floor(pw * rw / 7.0)) + x1, 0), W), min(max(int(math.ceil((pw + 1) * rw / 7.0)) + x1, 0), W)
                if he <= hs or we <= ws:
                    cols.append(features.new_zeros(C_))
                else:
                    cols.append(features[0, :, hs:he, ws:we].amax(dim=(1, 2)))
            rows.append(torch.stack(cols, 1))
        out.append(torch.stack(rows, 1))
    return torch.stack(out, 0)


def test_resnet50_fpn_800x1344_same_weights_features_rpn_outputs_and_gradients_vs_the_cpu_form(monkeypatch, products):
    from faster_rcnn_pytorch_amd import ops
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    from faster_rcnn_pytorch_amd.new_model import BackboneWithFPN, Bottleneck, FRCNN
    H, W = 800, 1344
    torch.manual_seed(0)
    m = FRCNN(num_classes=91, sampling="host").to(DEV)
    with torch.no_grad():
        m.rpn.rpn_head.cls_layer.weight.mul_(30)               # (box deltas left at their initial scale: ~820 proposals survive NMS, the sampler needs 512)
    ref = model_ref.RefFRCNNFPN(BackboneWithFPN(trainable_layers=3), 91)          # on CPU tensors the backbone's modules run their plain torch forms
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}                  # noqa: E731
    ref.backbone.load_state_dict(cpu(m.backbone.state_dict()))
    ref.classifier.load_state_dict(cpu(m.classifier.state_dict()))
    ref.rpn_head.load_state_dict(cpu(m.rpn.rpn_head.state_dict()))
    ref.cls_head.load_state_dict(cpu(m.frcnn_head.cls_head.state_dict()))
    ref.reg_head.load_state_dict(cpu(m.frcnn_head.reg_head.state_dict()))
    x, boxes, labels = synth(5, H, W, 3, 1, 91)

    body_masks, cls_masks, stash, cap = [], [], {}, {}
    o_aff, o_c, o_rpn = ops.affine_act, ops.conv3x3, ops.rpn_conv3x3

    def spy_aff(x_, scale, shift, res=None, relu=False):
        y = o_aff(x_, scale, shift, res, relu)
        if relu:
            body_masks.append((y.detach() > 0).cpu())
        return y

    def spy_c(x_, weight, bias=None, relu=False, pool=False):
        y = o_c(x_, weight, bias, relu, pool)
        if relu:
            body_masks.append((y.detach() > 0).cpu())
        return y

    def spy_rpn(feats, w3):
        raws = o_rpn(feats, w3)
        stash["rpn_raw"] = [r.detach() for r in raws]
        return raws
    for name, fn in (("affine_act", spy_aff), ("conv3x3", spy_c), ("rpn_conv3x3", spy_rpn)):
        monkeypatch.setattr(ops, name, fn)
    hooks = [m.backbone.register_forward_hook(lambda mod, i, o: cap.__setitem__("feats", list(o.values()))),
             m.frcnn_target_maker.register_forward_hook(lambda mod, i, o: cap.__setitem__("srois", o[2].detach())),
             m.classifier[1].register_forward_hook(lambda mod, i, o: cls_masks.append((o.detach() > 0).cpu())),
             m.classifier[3].register_forward_hook(lambda mod, i, o: cls_masks.append((o.detach() > 0).cpu()))]
    m.train()
    torch.manual_seed(205)
    pred, target = m(x.to(DEV), boxes.to(DEV), labels.to(DEV))
    for h in hooks:
        h.remove()
    monkeypatch.undo()
    assert len(body_masks) == 1 + 16 * 3 and len(cls_masks) == 2          # the stem's ReLU + three per bottleneck, all through the library's fused passes
    loss = FRCNNLoss(None)(pred, target)
    m.zero_grad(set_to_none=True)
    loss[0].backward()
    names = {"backbone.": "backbone.", "classifier.": "classifier.", "rpn.rpn_head.": "rpn_head.", "frcnn_head.cls_head.": "cls_head.", "frcnn_head.reg_head.": "reg_head."}
    g_dev = {}
    for n, g in _grads(m.named_parameters()).items():
        for a, b in names.items():
            if n.startswith(a):
                g_dev[b + n[len(a):]] = g
    b3 = m.rpn.rpn_head.inter_layer.bias.detach().view(1, -1, 1, 1)
    decisions = {"rpn": [((r + b3) > 0).cpu() for r in stash["rpn_raw"]], "classifier": cls_masks}

    _threads()
    t_cpu = [t.detach().cpu() for t in target]
    feats_c, pred_c, _ = model_ref.ref_forward_fixed_fpn(ref, x, cap["srois"].cpu(), t_cpu[2])
    loss_c = model_ref.ref_loss(pred_c, t_cpu)
    loss_c[0].backward()
    g_cpu = _grads(ref.named_parameters())
    ref.zero_grad(set_to_none=True)
    # float64 forward (own decisions) BEFORE the body's ReLUs are replaced
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        f64 = list(ref64.backbone(x.double()).values())
        c64, r64 = [], []
        for f in f64:
            h = torch.relu(ref64.rpn_head.inter_layer(f))
            c64.append(ref64.rpn_head.cls_layer(h).permute(0, 2, 3, 1).reshape(1, -1, 2))
            r64.append(ref64.rpn_head.reg_layer(h).permute(0, 2, 3, 1).reshape(1, -1, 4))
        cls64, reg64 = torch.cat(c64, 1), torch.cat(r64, 1)
    del ref64
    # (ii) the device's decisions: one MaskPopper in place of every nn.ReLU of the body (the stem's, then three calls per bottleneck, in order)
    pop = model_ref.MaskPopper(body_masks)
    ref.backbone.body.relu = pop
    for mod in ref.backbone.body.modules():
        if isinstance(mod, Bottleneck):
            mod.relu = pop
    feats_t, pred_t, flips = model_ref.ref_forward_fixed_fpn(ref, x, cap["srois"].cpu(), t_cpu[2], decisions=decisions)
    assert pop.k == len(body_masks)
    loss_t = model_ref.ref_loss(pred_t, t_cpu)
    loss_t[0].backward()
    g_tr = _grads(ref.named_parameters())
    sm = lambda t: torch.softmax(t.detach().double().cpu(), dim=-1)          # noqa: E731
    n_flip = pop.flips + flips["rpn"][0] + flips["classifier"][0]
    n_dec = pop.total + flips["rpn"][1] + flips["classifier"][1]
    rec = {
        "frame": "%dx%d seed 5, 3 boxes" % (H, W),
        "features_rel_dev_vs_cpu": max(rel(a, b) for a, b in zip(cap["feats"], feats_c)),
        "features_rel_dev_vs_f64": max(rel(a, b) for a, b in zip(cap["feats"], f64)),
        "features_rel_cpu_vs_f64": max(rel(a, b) for a, b in zip(feats_c, f64)),
        "features_rel_device_decisions_dev_vs_cpu": max(rel(a, b) for a, b in zip(cap["feats"], feats_t)),
        "rpn_score_abs_dev_vs_cpu": float((sm(pred[0]) - sm(pred_c[0])).abs().max()), "rpn_score_abs_dev_vs_f64": float((sm(pred[0]) - sm(cls64)).abs().max()),
        "rpn_score_abs_cpu_vs_f64": float((sm(pred_c[0]) - sm(cls64)).abs().max()),
        "rpn_reg_rel_dev_vs_cpu": rel(pred[1], pred_c[1]), "rpn_reg_rel_dev_vs_f64": rel(pred[1], reg64), "rpn_reg_rel_cpu_vs_f64": rel(pred_c[1], reg64),
        "rpn_reg_abs_dev_vs_cpu": float((pred[1].detach().cpu() - pred_c[1].detach()).abs().max()),
        "head_cls_rel_dev_vs_cpu": rel(pred[2], pred_c[2]), "head_reg_rel_dev_vs_cpu": rel(pred[3], pred_c[3]),
        "loss_dev": _loss_list(loss), "loss_cpu": _loss_list(loss_c), "loss_cpu_with_device_decisions": _loss_list(loss_t),
        "decisions": n_dec, "decisions_the_cpu_takes_differently": n_flip, "largest_flipped_preactivation_rel": pop.worst,
        "grad_rel_own_decisions_dev_vs_cpu": {n: rel(g_dev[n], g_cpu[n]) for n in g_cpu},
        "grad_rel_device_decisions_dev_vs_cpu": {n: rel(g_dev[n], g_tr[n]) for n in g_cpu},
    }
    for k in ("grad_rel_own_decisions_dev_vs_cpu", "grad_rel_device_decisions_dev_vs_cpu"):
        rec[k + "_max"] = max(rec[k].values())
    _report("fpn" if products == "native" else "fpn_split_products", rec)
    assert set(g_dev) == set(g_cpu) == set(g_tr) and len(g_cpu) == 72
    assert rec["features_rel_dev_vs_cpu"] < TOL
    assert rec["rpn_score_abs_dev_vs_cpu"] < TOL and rec["rpn_reg_rel_dev_vs_cpu"] < TOL
    assert rec["head_cls_rel_dev_vs_cpu"] < TOL and rec["head_reg_rel_dev_vs_cpu"] < TOL
    for a, b in zip(rec["loss_dev"], rec["loss_cpu"]):
        assert abs(a - b) < TOL * max(1.0, abs(b))
    assert rec["grad_rel_device_decisions_dev_vs_cpu_max"] < TOL, sorted(rec["grad_rel_device_decisions_dev_vs_cpu"].items(), key=lambda kv: -kv[1])[:5]
    assert n_flip < 1e-4 * n_dec and pop.worst < 1e-4
    assert rec["grad_rel_own_decisions_dev_vs_cpu_max"] < 5e-2
