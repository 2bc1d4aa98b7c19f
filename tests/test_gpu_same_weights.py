"""Same weights, same frame, both sides END TO END (VERDICT r4 "next" 1): ONE state_dict is loaded into the HIP-backed mirror and into the
reference's CPU form (oracle/model_ref.RefFRCNN / RefFRCNNFPN: torch-CPU fp32 convolutions and linear layers -- what models/model.py:279-281,
304-316 and models/new_model.py:372,391-400 run on a CPU -- + the C oracle's RoIPool / MultiScaleRoIAlign), at the sizes bench.py runs.

The model-level tests in test_gpu_model.py push the GPU model's OWN features through the oracle's path stages: they pin the sort / NMS /
target / pooling decisions, but a wrong extractor could not fail them.  Here nothing is shared but the weights, the frame, the sampled RoIs and
the targets (taken from the device run; they have their own bit-exact tests): compared are
  * the extractor's features (13 Winograd-stage layers with fused ReLU / max-pool words at 600 x 1000; the FPN's five maps at 800 x 1344),
  * the RPN's softmax scores and box regressions -- the tensors BASELINE.json's north_star bounds at 1e-4 of the reference's CPU path,
  * the head's outputs, the four losses,
  * after ONE loss.backward(): every extractor / backbone, RPN and head parameter gradient.
A float64 evaluation of the same network (forward) prices both fp32 paths: the CPU path is itself 1e-6-ish away from the exact values.
The measured maxima are written to gpurun_out/same_weights_<config>.json (quoted in DESIGN.md section 2)."""
import copy
import json
import os

import numpy as np
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
    """max |a - b| over the scale of b (its largest magnitude, at least 1e-30)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _report(name, rec):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "same_weights_%s.json" % name), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("\n[same-weights %s] " % name + json.dumps({k: v for k, v in rec.items() if not isinstance(v, dict)}))
    worst = sorted(rec["grad_rel"].items(), key=lambda kv: -kv[1])[:6]
    print("[same-weights %s] largest gradient differences (of each tensor's scale): %s" % (name, worst))


def _grads(named):
    return {n: p.grad.detach().clone() for n, p in named if p.requires_grad and p.grad is not None}


def test_vgg16_600x1000_same_weights_features_rpn_outputs_and_gradients_vs_the_cpu_form():
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
    cap = {}
    h1 = m.extractor.register_forward_hook(lambda mod, i, o: cap.__setitem__("feat", o))
    h2 = m.fast_rcnn_target_maker.register_forward_hook(lambda mod, i, o: cap.__setitem__("srois", o[2].detach()))
    m.train()
    torch.manual_seed(101)
    pred, target = m(x.to(DEV), [boxes.to(DEV)], [labels.to(DEV)])
    h1.remove(); h2.remove()
    loss = FRCNNLoss(None)(pred, target)
    m.zero_grad(set_to_none=True)
    loss[0].backward()
    g_dev = _grads(m.named_parameters())
    # the reference's CPU form on the same weights, frame, sampled RoIs and targets
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    t_cpu = [t.detach().cpu() for t in target]
    feat_c, pred_c = model_ref.ref_forward_fixed_vgg(ref, x, cap["srois"].cpu(), t_cpu[2])
    loss_c = model_ref.ref_loss(pred_c, t_cpu)
    loss_c[0].backward()
    g_cpu = _grads(ref.named_parameters())
    # float64 forward of extractor + RPN head: the exact values both fp32 paths approximate
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        f64 = ref64.extractor(x.double())
        h64 = torch.relu(ref64.rpn.inter_layer(f64))
        cls64 = ref64.rpn.cls_layer(h64).permute(0, 2, 3, 1).reshape(1, -1, 2)
        reg64 = ref64.rpn.reg_layer(h64).permute(0, 2, 3, 1).reshape(1, -1, 4)
    sm = lambda t: torch.softmax(t.detach().double().cpu(), dim=-1)          # noqa: E731
    rec = {
        "frame": "%dx%d seed 1, 4 boxes" % (H, W),
        "features_rel_dev_vs_cpu": rel(cap["feat"], feat_c), "features_rel_dev_vs_f64": rel(cap["feat"], f64), "features_rel_cpu_vs_f64": rel(feat_c, f64),
        "rpn_score_abs_dev_vs_cpu": float((sm(pred[0]) - sm(pred_c[0])).abs().max()), "rpn_score_abs_dev_vs_f64": float((sm(pred[0]) - sm(cls64)).abs().max()),
        "rpn_score_abs_cpu_vs_f64": float((sm(pred_c[0]) - sm(cls64)).abs().max()),
        "rpn_reg_rel_dev_vs_cpu": rel(pred[1], pred_c[1]), "rpn_reg_rel_dev_vs_f64": rel(pred[1], reg64), "rpn_reg_rel_cpu_vs_f64": rel(pred_c[1], reg64),
        "rpn_reg_abs_dev_vs_cpu": float((pred[1].detach().cpu() - pred_c[1].detach()).abs().max()),
        "head_cls_rel_dev_vs_cpu": rel(pred[2], pred_c[2]), "head_reg_rel_dev_vs_cpu": rel(pred[3], pred_c[3]),
        "loss_dev": [float(v.detach()) for v in loss], "loss_cpu": [float(v.detach()) for v in loss_c],
        "grad_rel": {n: rel(g_dev[n], g_cpu[n]) for n in g_cpu},
    }
    rec["grad_rel_max"] = max(rec["grad_rel"].values())
    _report("vgg", rec)
    assert set(g_dev) == set(g_cpu) and len(g_cpu) == 13 * 2 + 3 * 2 + 2 * 2 + 4      # extractor, RPN, heads, classifier (ONE module under two names: listed once)
    assert rec["features_rel_dev_vs_cpu"] < TOL
    assert rec["rpn_score_abs_dev_vs_cpu"] < TOL and rec["rpn_reg_rel_dev_vs_cpu"] < TOL
    assert rec["head_cls_rel_dev_vs_cpu"] < TOL and rec["head_reg_rel_dev_vs_cpu"] < TOL
    for a, b in zip(rec["loss_dev"], rec["loss_cpu"]):
        assert abs(a - b) < TOL * max(1.0, abs(b))
    assert rec["grad_rel_max"] < TOL, sorted(rec["grad_rel"].items(), key=lambda kv: -kv[1])[:5]


def test_resnet50_fpn_800x1344_same_weights_features_rpn_outputs_and_gradients_vs_the_cpu_form():
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    from faster_rcnn_pytorch_amd.new_model import BackboneWithFPN, FRCNN
    H, W = 800, 1344
    torch.manual_seed(0)
    m = FRCNN(num_classes=91, sampling="host").to(DEV)
    with torch.no_grad():
        m.rpn.rpn_head.cls_layer.weight.mul_(30)
        m.rpn.rpn_head.reg_layer.weight.mul_(2)
    ref = model_ref.RefFRCNNFPN(BackboneWithFPN(trainable_layers=3), 91)          # on CPU tensors the backbone's modules run their plain torch forms
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}                  # noqa: E731
    ref.backbone.load_state_dict(cpu(m.backbone.state_dict()))
    ref.classifier.load_state_dict(cpu(m.classifier.state_dict()))
    ref.rpn_head.load_state_dict(cpu(m.rpn.rpn_head.state_dict()))
    ref.cls_head.load_state_dict(cpu(m.frcnn_head.cls_head.state_dict()))
    ref.reg_head.load_state_dict(cpu(m.frcnn_head.reg_head.state_dict()))
    x, boxes, labels = synth(5, H, W, 3, 1, 91)
    cap = {}
    h1 = m.backbone.register_forward_hook(lambda mod, i, o: cap.__setitem__("feats", list(o.values())))
    h2 = m.frcnn_target_maker.register_forward_hook(lambda mod, i, o: cap.__setitem__("srois", o[2].detach()))
    m.train()
    torch.manual_seed(205)
    pred, target = m(x.to(DEV), boxes.to(DEV), labels.to(DEV))
    h1.remove(); h2.remove()
    loss = FRCNNLoss(None)(pred, target)
    m.zero_grad(set_to_none=True)
    loss[0].backward()
    names = {"backbone.": "backbone.", "classifier.": "classifier.", "rpn.rpn_head.": "rpn_head.", "frcnn_head.cls_head.": "cls_head.", "frcnn_head.reg_head.": "reg_head."}
    g_dev = {}
    for n, g in _grads(m.named_parameters()).items():
        for a, b in names.items():
            if n.startswith(a):
                g_dev[b + n[len(a):]] = g
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    t_cpu = [t.detach().cpu() for t in target]
    feats_c, pred_c = model_ref.ref_forward_fixed_fpn(ref, x, cap["srois"].cpu(), t_cpu[2])
    loss_c = model_ref.ref_loss(pred_c, t_cpu)
    loss_c[0].backward()
    g_cpu = _grads(ref.named_parameters())
    ref64 = copy.deepcopy(ref).double()
    with torch.no_grad():
        f64 = list(ref64.backbone(x.double()).values())
        c64, r64 = [], []
        for f in f64:
            h = torch.relu(ref64.rpn_head.inter_layer(f))
            c64.append(ref64.rpn_head.cls_layer(h).permute(0, 2, 3, 1).reshape(1, -1, 2))
            r64.append(ref64.rpn_head.reg_layer(h).permute(0, 2, 3, 1).reshape(1, -1, 4))
        cls64, reg64 = torch.cat(c64, 1), torch.cat(r64, 1)
    sm = lambda t: torch.softmax(t.detach().double().cpu(), dim=-1)          # noqa: E731
    rec = {
        "frame": "%dx%d seed 5, 3 boxes" % (H, W),
        "features_rel_dev_vs_cpu": max(rel(a, b) for a, b in zip(cap["feats"], feats_c)),
        "features_rel_dev_vs_f64": max(rel(a, b) for a, b in zip(cap["feats"], f64)),
        "features_rel_cpu_vs_f64": max(rel(a, b) for a, b in zip(feats_c, f64)),
        "rpn_score_abs_dev_vs_cpu": float((sm(pred[0]) - sm(pred_c[0])).abs().max()), "rpn_score_abs_dev_vs_f64": float((sm(pred[0]) - sm(cls64)).abs().max()),
        "rpn_score_abs_cpu_vs_f64": float((sm(pred_c[0]) - sm(cls64)).abs().max()),
        "rpn_reg_rel_dev_vs_cpu": rel(pred[1], pred_c[1]), "rpn_reg_rel_dev_vs_f64": rel(pred[1], reg64), "rpn_reg_rel_cpu_vs_f64": rel(pred_c[1], reg64),
        "rpn_reg_abs_dev_vs_cpu": float((pred[1].detach().cpu() - pred_c[1].detach()).abs().max()),
        "head_cls_rel_dev_vs_cpu": rel(pred[2], pred_c[2]), "head_reg_rel_dev_vs_cpu": rel(pred[3], pred_c[3]),
        "loss_dev": [float(v.detach()) for v in loss], "loss_cpu": [float(v.detach()) for v in loss_c],
        "grad_rel": {n: rel(g_dev[n], g_cpu[n]) for n in g_cpu},
    }
    rec["grad_rel_max"] = max(rec["grad_rel"].values())
    _report("fpn", rec)
    assert set(g_dev) == set(g_cpu) and len(g_cpu) == 72
    assert rec["features_rel_dev_vs_cpu"] < TOL
    assert rec["rpn_score_abs_dev_vs_cpu"] < TOL and rec["rpn_reg_rel_dev_vs_cpu"] < TOL
    assert rec["head_cls_rel_dev_vs_cpu"] < TOL and rec["head_reg_rel_dev_vs_cpu"] < TOL
    for a, b in zip(rec["loss_dev"], rec["loss_cpu"]):
        assert abs(a - b) < TOL * max(1.0, abs(b))
    assert rec["grad_rel_max"] < TOL, sorted(rec["grad_rel"].items(), key=lambda kv: -kv[1])[:5]
