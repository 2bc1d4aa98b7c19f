"""Checkpoint compatibility: key names are those of the reference's modules, written down here by
reading models/model_.py:269-297 (VGG16 FRCNN) -- torchvision's vgg16.features indices of the 13 convs, the RPN's three
convs, the head's two linears and the classifier registered under two names."""
import torch

from faster_rcnn_pytorch_amd import checkpoint as ck
from faster_rcnn_pytorch_amd.model import FRCNN

VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]


def reference_vgg_keys():
    keys = []
    for i in VGG_CONV_IDX:
        keys += ["extractor.%d.weight" % i, "extractor.%d.bias" % i]
    keys += ["classifier.0.weight", "classifier.0.bias", "classifier.2.weight", "classifier.2.bias"]
    for l in ("inter_layer", "cls_layer", "reg_layer"):
        keys += ["rpn.%s.weight" % l, "rpn.%s.bias" % l]
    for l in ("cls_head", "reg_head"):
        keys += ["fast_rcnn_head.%s.weight" % l, "fast_rcnn_head.%s.bias" % l]
    keys += ["fast_rcnn_head.classifier.0.weight", "fast_rcnn_head.classifier.0.bias",
             "fast_rcnn_head.classifier.2.weight", "fast_rcnn_head.classifier.2.bias"]
    return keys


def test_state_dict_keys_and_shapes_are_the_references():
    m = FRCNN(num_classes=21)
    sd = m.state_dict()
    assert sorted(sd.keys()) == sorted(reference_vgg_keys())
    assert sd["rpn.inter_layer.weight"].shape == (512, 512, 3, 3) and sd["rpn.inter_layer.bias"].shape == (512,)
    assert sd["rpn.cls_layer.weight"].shape == (18, 512, 1, 1) and sd["rpn.reg_layer.weight"].shape == (36, 512, 1, 1)
    assert sd["fast_rcnn_head.cls_head.weight"].shape == (21, 4096) and sd["fast_rcnn_head.reg_head.weight"].shape == (84, 4096)
    assert sd["classifier.0.weight"].data_ptr() == sd["fast_rcnn_head.classifier.0.weight"].data_ptr()     # one module, two names


def test_load_ddp_prefixed_reference_checkpoint_and_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    src = FRCNN(num_classes=21)
    ref_sd = {"module." + k: torch.randn(v.shape, generator=g) * 0.01 for k, v in src.state_dict().items()}
    for a, b in (("classifier.0", "fast_rcnn_head.classifier.0"), ("classifier.2", "fast_rcnn_head.classifier.2")):
        for p in ("weight", "bias"):
            ref_sd["module.%s.%s" % (b, p)] = ref_sd["module.%s.%s" % (a, p)]      # the reference saves the alias with equal values
    path = ck.checkpoint_path(str(tmp_path), "frcnn", 7)
    import os
    os.makedirs(os.path.dirname(path))
    torch.save({"epoch": 7, "model_state_dict": ref_sd, "optimizer_state_dict": {}, "scheduler_state_dict": {}}, path)
    assert path.endswith(os.path.join("frcnn", "saves", "frcnn.7.pth.tar"))

    m = FRCNN(num_classes=21)
    assert ck.load_reference_checkpoint(m, path) == 7
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref_sd["module." + k]), k
    assert m.classifier[0].weight is m.fast_rcnn_head.classifier[0].weight

    # resume(): optimizer + scheduler state survive a save/load cycle in the reference's format
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=5e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.1)
    for _ in range(4):
        opt.step()
        sch.step()
    ck.save_checkpoint(ck.checkpoint_path(str(tmp_path), "run", 3), 3, m, opt, sch)
    m2 = FRCNN(num_classes=21)
    opt2 = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=5e-4)
    sch2 = torch.optim.lr_scheduler.StepLR(opt2, step_size=3, gamma=0.1)
    assert ck.resume(str(tmp_path), "run", 4, m2, opt2, sch2) and not ck.resume(str(tmp_path), "run", 0, m2)
    assert sch2.last_epoch == 4 and abs(opt2.param_groups[0]["lr"] - 1e-4) < 1e-12
    for k, v in m2.state_dict().items():
        assert torch.equal(v, m.state_dict()[k])


def test_unexpected_or_missing_keys_raise_like_the_reference():
    m = FRCNN(num_classes=21)
    sd = dict(m.state_dict())
    sd.pop("rpn.cls_layer.bias")
    try:
        ck.load_reference_checkpoint(m, {"model_state_dict": sd})
    except RuntimeError as e:
        assert "rpn.cls_layer.bias" in str(e)
    else:
        raise AssertionError("missing key must raise (strict load_state_dict, utils/util.py:149)")


# ---- ResNet-50-FPN model (models/new_model.py:352-385) ---------------------------------------------------------------
# Key names written down from the published module structure, independently of faster_rcnn_pytorch_amd/new_model.py:
#   backbone = torchvision resnet_fpn_backbone('resnet50', trainable_layers=3)  (new_model.py:372)
#       .body = IntermediateLayerGetter(resnet50 up to layer4): conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3},
#               layer{l}.0.downsample.{0 conv, 1 bn}; every norm is FrozenBatchNorm2d: 4 BUFFERS weight / bias / running_mean /
#               running_var and no num_batches_tracked
#       .fpn  = FeaturePyramidNetwork: inner_blocks.{0..3}.0 (1x1, bias) and layer_blocks.{0..3}.0 (3x3, bias) -- the ".0" is the
#               Conv2dNormActivation wrapper of torchvision >= 0.13 (the reference needs >= 0.13: ResNet50_Weights, new_model.py:13)
#   classifier.{0,2} (new_model.py:373-376) aliased as frcnn_head.classifier.{0,2} (new_model.py:385)
#   rpn.rpn_head.{inter_layer, cls_layer, reg_layer} (new_model.py:20, 96-108); frcnn_head.{cls_head, reg_head} (new_model.py:122-123)
RESNET50_BLOCKS = {1: 3, 2: 4, 3: 6, 4: 3}
FBN = ("weight", "bias", "running_mean", "running_var")


def reference_fpn_keys():
    keys = ["backbone.body.conv1.weight"] + ["backbone.body.bn1." + b for b in FBN]
    for l, nblk in RESNET50_BLOCKS.items():
        for i in range(nblk):
            p = "backbone.body.layer%d.%d." % (l, i)
            for j in (1, 2, 3):
                keys.append(p + "conv%d.weight" % j)
                keys += [p + "bn%d.%s" % (j, b) for b in FBN]
            if i == 0:
                keys.append(p + "downsample.0.weight")
                keys += [p + "downsample.1." + b for b in FBN]
    for blk in ("inner_blocks", "layer_blocks"):
        for i in range(4):
            keys += ["backbone.fpn.%s.%d.0.weight" % (blk, i), "backbone.fpn.%s.%d.0.bias" % (blk, i)]
    for pre in ("classifier.", "frcnn_head.classifier."):
        keys += [pre + "0.weight", pre + "0.bias", pre + "2.weight", pre + "2.bias"]
    for l in ("inter_layer", "cls_layer", "reg_layer"):
        keys += ["rpn.rpn_head.%s.weight" % l, "rpn.rpn_head.%s.bias" % l]
    for l in ("cls_head", "reg_head"):
        keys += ["frcnn_head.%s.weight" % l, "frcnn_head.%s.bias" % l]
    return keys


def test_fpn_state_dict_keys_and_shapes_are_the_references():
    from faster_rcnn_pytorch_amd.new_model import FRCNN as FRCNN_FPN
    m = FRCNN_FPN(num_classes=91)
    sd = m.state_dict()
    want = reference_fpn_keys()
    assert len(want) == len(set(want)) == 1 + 4 + 16 * (3 * 5) + 4 * 5 + 16 + 8 + 6 + 4
    assert sorted(sd.keys()) == sorted(want)
    assert sd["backbone.body.conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["backbone.body.layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["backbone.body.layer4.2.conv3.weight"].shape == (2048, 512, 1, 1)
    assert sd["backbone.body.layer3.0.conv2.weight"].shape == (256, 256, 3, 3)           # stride on the 3x3 (torchvision v1.5 form)
    assert sd["backbone.fpn.inner_blocks.3.0.weight"].shape == (256, 2048, 1, 1)
    assert sd["backbone.fpn.layer_blocks.0.0.weight"].shape == (256, 256, 3, 3)
    assert sd["rpn.rpn_head.inter_layer.weight"].shape == (256, 256, 3, 3)
    assert sd["rpn.rpn_head.cls_layer.weight"].shape == (6, 256, 1, 1) and sd["rpn.rpn_head.reg_layer.weight"].shape == (12, 256, 1, 1)
    assert sd["classifier.0.weight"].shape == (1024, 12544) and sd["classifier.2.weight"].shape == (1024, 1024)
    assert sd["frcnn_head.cls_head.weight"].shape == (91, 1024) and sd["frcnn_head.reg_head.weight"].shape == (364, 1024)
    assert sd["classifier.0.weight"].data_ptr() == sd["frcnn_head.classifier.0.weight"].data_ptr()
    # trainable_layers=3: layer2..4 train, conv1 / layer1 are frozen (torchvision _resnet_fpn_extractor)
    req = {n: p.requires_grad for n, p in m.named_parameters()}
    assert not req["backbone.body.conv1.weight"] and not req["backbone.body.layer1.0.conv1.weight"]
    assert req["backbone.body.layer2.0.conv1.weight"] and req["backbone.body.layer4.2.conv3.weight"] and req["backbone.fpn.inner_blocks.0.0.weight"]


def test_fpn_checkpoint_loads_with_module_prefix():
    from faster_rcnn_pytorch_amd.new_model import FRCNN as FRCNN_FPN
    g = torch.Generator().manual_seed(1)
    m = FRCNN_FPN(num_classes=91)
    ref_sd = {"module." + k: torch.randn(v.shape, generator=g) * 0.01 for k, v in m.state_dict().items()}
    for a, b in (("classifier.0", "frcnn_head.classifier.0"), ("classifier.2", "frcnn_head.classifier.2")):
        for p in ("weight", "bias"):
            ref_sd["module.%s.%s" % (b, p)] = ref_sd["module.%s.%s" % (a, p)]
    assert ck.load_reference_checkpoint(m, {"epoch": 3, "model_state_dict": ref_sd}) == 3
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref_sd["module." + k]), k
