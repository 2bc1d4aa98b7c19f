"""ResNet-50-FPN Faster R-CNN -- host-side mirror of the reference's models/new_model.py on the HIP hot path.

Same classes / signatures as the reference (models/new_model.py:17-470):
    FRCNN(num_classes).forward(x, boxes, labels) -> ((rpn_cls[1,N,2], rpn_reg[1,N,4], head_cls[512,C], head_reg[512,4]),
                                                      (tgt_rpn_cls[N], tgt_rpn_reg[N,4], tgt_cls[512], tgt_reg[512,4]))
    FRCNN.predict(x, opts) -> (bbox, label, score)
Hot-path stages go through libfrcnn_hip (FPN variants): multi-level anchor grid (torchvision AnchorGenerator,
new_model.py:23-25,46-47), ONE global proposal stage over all levels (new_model.py:49-86), tie-inclusive RPN
matching (new_model.py:299-349), 512-sample head targets with raw labels (new_model.py:157-206), MultiScaleRoIAlign
(new_model.py:127,143).  The backbone is a from-scratch torch definition of torchvision's resnet_fpn_backbone('resnet50',
trainable_layers=3) -- FrozenBatchNorm2d, 5 output maps '0','1','2','3','pool' -- with random init (weights need the network).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .model import _Sampler, _unwrap, normal_init


class FrozenBatchNorm2d(nn.Module):
    """torchvision.ops.misc.FrozenBatchNorm2d: fixed statistics and affine parameters (buffers)."""

    def __init__(self, n, eps=1e-5, inner=False):
        super().__init__()
        self.eps = eps
        self.inner = inner                                   # bn1 / bn2 of a bottleneck: feeds ReLU + the next convolution, not the residual sum
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

        self._affine_key, self._affine, self._affine_bf16 = None, None, None

    def affine(self):
        """(scale, bias) of torchvision's forward, computed once per state of the four buffers: they are frozen, and recomputing
        them costs five tiny kernels per layer per step (53 layers: ~270 launches and ~1 ms of host enqueue in the FPN step).
        Same operations on the same values, so the result is bit-identical to evaluating them in forward."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b.data_ptr(), b._version) for b in bufs)
        if key != self._affine_key:
            with torch.no_grad():
                scale = (self.weight * (self.running_var + self.eps).rsqrt()).reshape(1, -1, 1, 1)
                bias = self.bias.reshape(1, -1, 1, 1) - self.running_mean.reshape(1, -1, 1, 1) * scale
            self._affine_key, self._affine = key, (scale, bias)
        return self._affine

    def affine_bf16(self):
        scale, bias = self.affine()
        if self._affine_bf16 is None or self._affine_bf16[0] is not scale:
            self._affine_bf16 = (scale, scale.bfloat16(), bias.bfloat16())
        return self._affine_bf16[1], self._affine_bf16[2]

    def forward(self, x):
        if self.inner and x.dtype == torch.bfloat16:
            # bf16 autocast (configs[4]; the reference has no mixed-precision mode): an INNER norm stays in bf16 -- one addcmul (fp32
            # arithmetic, one rounding) whose output the next convolution consumes as it is.  x * scale + bias would promote the
            # activation to fp32 and autocast would cast it back in front of the convolution: two full-tensor kernels more per layer,
            # forward and backward.  The norms in front of the residual sum (bn3, downsample) keep promoting: the residual stream
            # stays fp32 (keeping ALL norms in bf16 moved the RPN outputs 13-15 % from the fp32 model instead of < 8 %).
            scale, bias = self.affine_bf16()
            return torch.addcmul(bias, x, scale)
        # (One fused F.batch_norm for bf16 activations was tried: it keeps the residual stream in bf16 -- RPN outputs 13-15 % off the fp32
        # model instead of < 8 % on a random-init network -- and its host dispatch costs more than these two ops: 14.6 vs 13.3 ms enqueue.)
        scale, bias = self.affine()
        return x * scale + bias


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes, inner=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes, inner=True)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def _norm(self, bn, x, res=None, relu=False):
        """bn(x) [+ res] [-> ReLU]: one fused pass on the fp32 device path (ops.affine_act: the same operations in the same order), the torch form otherwise."""
        if ops.affine_act_supported(x):
            scale, shift = bn.affine()
            return ops.affine_act(x, scale, shift, res, relu)
        out = bn(x)
        if res is not None:
            out = out + res
        return self.relu(out) if relu else out

    @staticmethod
    def _c1(conv, x):
        """A 1 x 1 convolution: torch's forward and input gradient; its weight gradient dW = dY . X^T on the library's k-contiguous GEMM where the
        shapes allow (the vendor path transposes both operands to NHWC first)."""
        if conv.stride == (1, 1) and conv.bias is None and ops.conv1x1_supported(x, conv.weight):
            return ops.conv1x1(x, conv.weight)
        return conv(x)

    def _forward_mixed(self, x):
        """bf16 autocast (BASELINE configs[4]): the three frozen norms with their ReLUs / the residual sum as ONE pass each (ops.affine_act_mixed, csrc/affine.hip):
        the inner norms read and write bf16, the norm in front of the residual sum reads the convolution's bf16 output and the fp32 stream and writes the fp32
        stream AND its bf16 twin, which the next block's convolutions read (`_frcnn_bf16` on the tensor) instead of casting the stream again.  Same values as the
        torch form under autocast (fp32 arithmetic, one rounding per tensor); ~20 elementwise / dtype-copy launches per block and direction less."""
        xb = getattr(x, "_frcnn_bf16", None)
        if xb is None:
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        s1, b1 = self.bn1.affine()
        s2, b2 = self.bn2.affine()
        s3, b3 = self.bn3.affine()
        out = ops.affine_act_mixed(self.conv1(xb), s1, b1, relu=True, out_bf16=True)
        out = ops.affine_act_mixed(self.conv2(out), s2, b2, relu=True, out_bf16=True)
        if self.downsample is not None:
            sd, bd = self.downsample[1].affine()
            idt = ops.affine_act_mixed(self.downsample[0](xb), sd, bd)
        else:
            idt = x if x.dtype == torch.float32 else x.float()
        y, yb = ops.affine_act_mixed(self.conv3(out), s3, b3, res=idt, relu=True, twin=True)
        y._frcnn_bf16 = yb
        return y

    def forward(self, x):
        if ops.affine_act_mixed_supported(x):
            return self._forward_mixed(x)
        idt = x
        out = self._norm(self.bn1, self._c1(self.conv1, x), relu=True)
        if self.conv2.stride == (1, 1) and not torch.is_autocast_enabled() and ops.conv3x3_supported(out, self.conv2.weight):
            # conv2 + bn2 + ReLU in one stage call: the frozen norm's scale folded into the weight (s * conv(x, w) = conv(x, s w): one small
            # elementwise launch, differentiable), its shift as the bias, the ReLU in the output transform and its backward in the gradient
            # kernels -- instead of the vendor convolution + two norm passes + the ReLU, forward and backward
            scale, shift = self.bn2.affine()
            out = ops.conv3x3(out, self.conv2.weight * scale.reshape(-1, 1, 1, 1), shift.reshape(-1), relu=True)
        else:
            out = self._norm(self.bn2, self.conv2(out), relu=True)
        if self.downsample is not None:
            idt = self._norm(self.downsample[1], self.downsample[0](x))
        return self._norm(self.bn3, self._c1(self.conv3, out), res=idt, relu=True)


class ResNet50Body(nn.Module):
    """IntermediateLayerGetter(resnet50, {'layer1':'0', ..., 'layer4':'3'})."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.inplanes = 64
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.conv1(x)
        if ops.affine_act_mixed_supported(x):
            scale, shift = self.bn1.affine()
            x = self.maxpool(ops.affine_act_mixed(x, scale, shift, relu=True))     # bf16 autocast: the stem's norm + ReLU in one pass, fp32 out
        elif ops.affine_act_supported(x):
            scale, shift = self.bn1.affine()
            x = self.maxpool(ops.affine_act(x, scale, shift, None, True))
        else:
            x = self.maxpool(self.relu(self.bn1(x)))
        out = OrderedDict()
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            x = layer(x)
            out[str(i)] = x
        return out


class FeaturePyramidNetwork(nn.Module):
    """torchvision.ops.FeaturePyramidNetwork([256,512,1024,2048], 256, extra_blocks=LastLevelMaxPool())."""

    def __init__(self, in_channels_list=(256, 512, 1024, 2048), out_channels=256):
        super().__init__()
        self.inner_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(c, out_channels, 1)) for c in in_channels_list])
        self.layer_blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, padding=1)) for _ in in_channels_list])
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        names, xs = list(x.keys()), list(x.values())
        def layer(idx, t):
            conv = self.layer_blocks[idx][0]
            if not torch.is_autocast_enabled() and ops.conv3x3_supported(t, conv.weight):
                return ops.conv3x3(t, conv.weight, conv.bias)            # the 256 -> 256 output convolution on the fp32 Winograd stage
            if ops.conv3x3_bf16_c256_supported(t, conv.weight):
                return ops.conv3x3_bf16_c256(t, conv.weight, conv.bias)  # bf16 autocast, the two largest levels: the RPN head's bf16 MFMA kernels
            return self.layer_blocks[idx](t)
        def inner(idx, t):
            conv = self.inner_blocks[idx][0]
            if torch.is_autocast_enabled() and getattr(t, "_frcnn_bf16", None) is not None:
                return self.inner_blocks[idx](t._frcnn_bf16)         # the stream's bf16 twin, written by the last bottleneck's fused norm (no cast pass)
            if ops.conv1x1_supported(t, conv.weight):
                return ops.conv1x1(t, conv.weight, conv.bias)            # the lateral 1 x 1: weight gradient on the library's GEMM
            return self.inner_blocks[idx](t)
        last = inner(-1, xs[-1])
        results = [layer(-1, last)]
        for idx in range(len(xs) - 2, -1, -1):
            lat = inner(idx, xs[idx])
            if lat.dtype == torch.bfloat16 and last.dtype == torch.bfloat16:
                # bf16 autocast: interpolate is on autocast's fp32 list, which turned the whole top-down pathway (and, behind it, a cast pass in front of
                # every output convolution) into fp32; nearest-neighbour upsampling copies values, so it runs outside autocast and the sum stays bf16
                with torch.autocast("cuda", enabled=False):
                    last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            else:
                last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
            results.insert(0, layer(idx, last))
        names.append("pool")
        results.append(F.max_pool2d(results[-1], 1, 2, 0))           # LastLevelMaxPool
        return OrderedDict(zip(names, results))


class BackboneWithFPN(nn.Module):
    def __init__(self, trainable_layers=3):
        super().__init__()
        self.body = ResNet50Body()
        self.fpn = FeaturePyramidNetwork()
        self.out_channels = 256
        train = ["layer4", "layer3", "layer2", "layer1", "conv1"][:trainable_layers]
        for name, p in self.body.named_parameters():
            if all(not name.startswith(t) for t in train):
                p.requires_grad_(False)

    def forward(self, x):
        return self.fpn(self.body(x))


class RPNHead(nn.Module):
    """models/new_model.py:89-114."""

    def __init__(self, in_channels=256, out_channels=256):
        super().__init__()
        num_anchors = 3
        self.inter_layer = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.cls_layer = nn.Conv2d(in_channels, num_anchors * 2, kernel_size=1)
        self.reg_layer = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=1)
        normal_init(self.inter_layer, 0, 0.01)
        normal_init(self.cls_layer, 0, 0.01)
        normal_init(self.reg_layer, 0, 0.01)
        self.fused_bf16_conv = True          # under bf16 autocast: the hand-written implicit-GEMM head (csrc/rpn_conv.hip) instead of MIOpen + tail

    def forward_levels(self, feats):
        """All levels at once: the 3x3 without its bias (fp32: the hand-written MFMA kernel; bf16 tensors outside the fused path: MIOpen), then ONE MFMA kernel for bias + ReLU + both 1x1 heads
        + the NHWC layout + the concatenation (new_model.py:37-44).  Under bf16 autocast the 3x3 outputs are bf16 and the
        heads contract on the bf16 matrix cores with fp32 accumulate; predictions (hence box regression) stay fp32."""
        f0 = feats[0]
        n_out = self.cls_layer.out_channels + self.reg_layer.out_channels
        if (self.fused_bf16_conv and f0.is_cuda and f0.size(0) == 1 and f0.size(1) == 256 and n_out <= 32 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype('cuda') == torch.bfloat16):
            # mixed-precision configuration: conv3x3 + bias + ReLU + both heads of all levels in ONE bf16 MFMA implicit-GEMM launch
            with torch.autocast("cuda", enabled=False):
                return ops.rpn_conv_head_levels([f.to(torch.bfloat16) for f in feats], self.inter_layer.weight, self.inter_layer.bias,
                                                self.cls_layer.weight, self.cls_layer.bias, self.reg_layer.weight, self.reg_layer.bias)
        if f0.is_cuda and f0.size(0) == 1 and f0.dtype in (torch.float32, torch.bfloat16):
            if f0.dtype == torch.float32 and not torch.is_autocast_enabled() and ops.rpn_conv3x3_supported(feats, self.inter_layer.weight):
                raws = ops.rpn_conv3x3(list(feats), self.inter_layer.weight)      # fp32 MFMA implicit GEMM, all levels in one launch (csrc/rpn_conv_f32.hip)
            else:
                raws = [torch.nn.functional.conv2d(f, self.inter_layer.weight, None, padding=1) for f in feats]
            if raws[0].dtype in (torch.float32, torch.bfloat16) and all(r.dtype == raws[0].dtype for r in raws):
                with torch.autocast("cuda", enabled=False):
                    return ops.rpn_head_tail_levels(raws, self.inter_layer.bias, self.cls_layer.weight, self.cls_layer.bias,
                                                    self.reg_layer.weight, self.reg_layer.bias,
                                                    mfma="bf16" if raws[0].dtype == torch.bfloat16 else "f32")
        cls, reg = zip(*[self.forward(f) for f in feats])                          # reference form (batch > 1)
        return torch.cat(cls, dim=1), torch.cat(reg, dim=1)

    def forward(self, features):
        batch_size = features.size(0)
        x = torch.relu(self.inter_layer(features))
        pred_cls = self.cls_layer(x)
        pred_reg = self.reg_layer(x)
        pred_reg = pred_reg.permute(0, 2, 3, 1).contiguous().view(batch_size, -1, 4)
        pred_cls = pred_cls.permute(0, 2, 3, 1).contiguous().view(batch_size, -1, 2)
        return pred_cls, pred_reg


class RegionProposalNetwork(nn.Module):
    """models/new_model.py:17-86."""

    def __init__(self):
        super().__init__()
        self.min_size = 10
        # False (default) = ONE class-agnostic NMS over the boxes of all levels: the reference (new_model.py:74-83).
        # True = the per-FPN-level variant BASELINE.json configs[3] words (torchvision's RPN): boxes only compete inside their level.
        self.per_level_nms = False
        self.rpn_head = RPNHead()
        self.anchor_generator = ops.AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)),
                                                    aspect_ratios=((0.5, 1.0, 2.0),) * 5)

    @staticmethod
    def top_k(mode):
        return (2000, 1000) if mode == "test" else (4000, 1000)                   # new_model.py:54-58

    def propose(self, x, features, mode):
        """Asynchronous form: (pred_rpn_cls [N,2], pred_rpn_reg [N,4], rois [P,4] fixed, count int32[1], anchors [N,4])."""
        feats = list(features.values())
        cls, reg = self.rpn_head.forward_levels(feats)
        pred_rpn_cls = cls.float().flatten(0, -2)
        pred_rpn_reg = reg.float().reshape(-1, 4)
        h, w = x.shape[2:]
        shapes = [tuple(f.shape[-2:]) for f in feats]
        anchor = self.anchor_generator.grid((h, w), shapes, x.device, normalise=True)   # new_model.py:46-47, cached in HBM
        pre, post = self.top_k(mode)
        lvl = None
        if self.per_level_nms:
            A = self.rpn_head.cls_layer.out_channels // 2
            lvl = np.concatenate([[0], np.cumsum([fh * fw * A for fh, fw in shapes])])
        rois, cnt, _ = ops.region_proposal(pred_rpn_reg.detach(), pred_rpn_cls.detach(), anchor, self.min_size / 1000, pre, 0.7, post,
                                           nms_level_offsets=lvl)
        return pred_rpn_cls, pred_rpn_reg, rois, cnt, anchor

    def forward(self, x, features, mode):
        pred_rpn_cls, pred_rpn_reg, rois, cnt, anchor = self.propose(x, features, mode)
        return pred_rpn_cls, pred_rpn_reg, rois[:ops.host_count(cnt, 'region_proposal')], anchor


class FRCNNHead(nn.Module):
    """models/new_model.py:117-150."""

    def __init__(self, num_classes, roi_size, classifier):
        super().__init__()
        self.num_classes = num_classes
        self.cls_head = nn.Linear(1024, num_classes)
        self.reg_head = nn.Linear(1024, num_classes * 4)
        self.roi_pool = ops.MultiScaleRoIAlign(featmap_names=["0", "1", "2", "3"], output_size=roi_size, sampling_ratio=2)
        self.classifier = classifier
        normal_init(self.cls_head, 0, 0.01)
        normal_init(self.reg_head, 0, 0.001)

    def forward(self, features, roi, img_shape):
        h, w = img_shape
        feats = [features[k] for k in self.roi_pool.featmap_names]
        if roi.is_cuda and roi.dtype == torch.float32 and self.roi_pool.scales != "reference" and roi.shape[0] <= 4096 and not roi.requires_grad:
            # new_model.py:136-140 (roi * (w, h, w, h): image pixels) and, in the same launch, the order that dispatches the pooling's
            # workgroups largest footprint first (the pooled rows do not depend on it)
            scaled_roi, order = ops.roi_scale_order(roi, (w, h, w, h), [tuple(f.shape[-2:]) for f in feats], self.roi_pool.scales)
        else:
            scaled_roi, order = roi * ops.const_tensor((w, h, w, h), roi.device), None
        pool = self.roi_pool(features, [scaled_roi], [(w, h)], order=order)
        x = self.classifier(pool.view(pool.size(0), -1))
        return self.cls_head(x), self.reg_head(x)


class FRCNNTargetMaker(nn.Module):
    """models/new_model.py:153-206: box_iou (no eps), labels used raw, <= 128 positives of 512 samples."""

    def __init__(self, sampler=None):
        super().__init__()
        self.sampler = sampler or _Sampler()

    def forward(self, boxes, labels, rois, n_rois=None):
        boxes = _unwrap(boxes)
        labels = _unwrap(labels).to(torch.int64)
        s = self.sampler
        kw = dict(n_rois=n_rois, variant=1, label_offset=0, max_pos=128, total=512)
        if s.sampling == "host":
            c = ops.head_targets(rois, boxes, labels, **kw)[4].cpu().tolist()
            perm_pos, perm_neg = torch.randperm(c[0]), torch.randperm(c[1])        # new_model.py:175,180
            cls, reg, srois, _, counts = ops.head_targets(rois, boxes, labels, perm_pos=perm_pos, perm_neg=perm_neg, **kw)
            if counts.cpu().tolist()[2] != 512:
                raise RuntimeError("FRCNNTargetMaker: fewer than 512 samples (assert at new_model.py:183)")
        else:
            cls, reg, srois, _, _ = ops.head_targets(rois, boxes, labels, philox_state=s.state(rois.device),
                                                     status=s.status.word(rois.device), **kw)
        return cls, reg, srois


class RPNTargetMaker(nn.Module):
    """models/new_model.py:299-349: no inside filter, tie-inclusive low-quality matches."""

    def __init__(self, sampler=None):
        super().__init__()
        self.sampler = sampler or _Sampler()

    def forward(self, boxes, anchors):
        boxes = _unwrap(boxes)
        s = self.sampler
        if s.sampling == "host":
            n_pos, n_neg = ops.rpn_targets(anchors, boxes, variant=1)[2].cpu().tolist()[:2]
            perm_pos = torch.randperm(n_pos) if n_pos > 128 else None
            perm_neg = torch.randperm(n_neg) if n_neg > 256 - n_pos else None
            cls, reg, counts = ops.rpn_targets(anchors, boxes, variant=1, perm_pos=perm_pos, perm_neg=perm_neg)
            if counts.cpu().tolist()[2] != 0:
                raise RuntimeError("RPNTargetMaker: permutation length mismatch")
        else:
            cls, reg, _ = ops.rpn_targets(anchors, boxes, variant=1, philox_state=s.state(anchors.device))
        return cls, reg


class FRCNN(nn.Module):
    """models/new_model.py:366-470."""

    def __init__(self, num_classes, sampling="device", seed=0):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = BackboneWithFPN(trainable_layers=3)
        self.classifier = nn.Sequential(nn.Linear(in_features=12544, out_features=1024), nn.ReLU(inplace=True),
                                        nn.Linear(in_features=1024, out_features=1024), nn.ReLU(inplace=True))
        self.sampler = _Sampler(sampling, seed)
        self.rpn = RegionProposalNetwork()
        self.rpn_target_maker = RPNTargetMaker(self.sampler)
        self.frcnn_target_maker = FRCNNTargetMaker(self.sampler)
        self.frcnn_head = FRCNNHead(num_classes=num_classes, roi_size=7, classifier=self.classifier)
        self.last_proposal_count = None

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def check_device_status(self):
        """As models.model.FRCNN.check_device_status: raises if a device-side failure (aborted NMS scan, fewer than 512 RoI
        samples -- the reference asserts there, new_model.py:182) was recorded since the last call; one host sync."""
        self.sampler.status.check()

    def graph_stages(self):
        """Where parallel.GraphStep cuts the backward (see models.model.FRCNN.graph_stages): behind the multi-scale RoIAlign."""
        return {"cut_module": self.frcnn_head.roi_pool, "late_modules": (self.classifier, self.frcnn_head.cls_head, self.frcnn_head.reg_head)}

    def forward(self, x, boxes, labels):
        features = self.backbone(x)                                                # new_model.py:394
        pred_rpn_cls, pred_rpn_reg, rois, n_rois, anchors = self.rpn.propose(x, features, "train")
        self.last_proposal_count = n_rois                                          # device int32[1]: proposals that survived NMS (<= 1000)
        target_rpn_cls, target_rpn_reg = self.rpn_target_maker(boxes=boxes, anchors=anchors)
        target_fast_rcnn_cls, target_fast_rcnn_reg, sample_rois = self.frcnn_target_maker(boxes=boxes, labels=labels, rois=rois,
                                                                                          n_rois=n_rois)
        pred_fast_rcnn_cls, pred_fast_rcnn_reg = self.frcnn_head(features, sample_rois, x.shape[2:])
        pred_fast_rcnn_reg = pred_fast_rcnn_reg.reshape(512, -1, 4)
        # row i keeps the regression of its target class (reference: pred[arange(512), cls], advanced indexing).  torch.gather selects the
        # same elements with one kernel forward and one backward; the indexing form costs an arange, an index kernel and a rocprim sort
        # + scatter in backward (~30 us and 7 launches per step).
        pred_fast_rcnn_reg = torch.gather(pred_fast_rcnn_reg, 1, target_fast_rcnn_cls.clamp(min=0).view(-1, 1, 1).expand(-1, 1, 4)).squeeze(1)   # (-1 = unsampled row: its loss is NaN anyway)
        return (pred_rpn_cls.unsqueeze(0), pred_rpn_reg.unsqueeze(0), pred_fast_rcnn_cls, pred_fast_rcnn_reg), \
               (target_rpn_cls, target_rpn_reg, target_fast_rcnn_cls, target_fast_rcnn_reg)

    @torch.no_grad()
    def predict(self, x, opts):
        threshold = float(getattr(opts, "thres", opts))
        features = self.backbone(x)
        _, _, rois, n_rois, _ = self.rpn.propose(x, features, "test")
        rois = rois[:ops.host_count(n_rois, 'region_proposal')]
        pred_fast_rcnn_cls, pred_fast_rcnn_reg = self.frcnn_head(features, rois, x.shape[2:])
        pred_cls = torch.softmax(pred_fast_rcnn_cls, dim=-1)
        pred_fast_rcnn_reg = pred_fast_rcnn_reg.reshape(-1, self.num_classes, 4) * ops.const_tensor((0.1, 0.1, 0.2, 0.2), x.device)
        rois = rois.reshape(-1, 1, 4).expand_as(pred_fast_rcnn_reg)
        pred_bbox = ops.cxcy_to_xy(ops.decode(pred_fast_rcnn_reg.reshape(-1, 4).contiguous(), ops.xy_to_cxcy(rois.reshape(-1, 4).contiguous())))
        pred_bbox = pred_bbox.reshape(-1, self.num_classes * 4).clamp(min=0, max=1)
        return self._suppress(pred_bbox, pred_cls, threshold)

    def _suppress(self, raw_cls_bbox, raw_prob, threshold):
        """models/model.py:382-402 (per-class score mask + nms(0.3), class 0 = background skipped, results concatenated
        class by class) as ONE class-aware NMS: 2 host syncs in total instead of 2 per class."""
        R = raw_prob.shape[0]
        boxes = raw_cls_bbox.reshape((R, self.num_classes, 4))[:, 1:, :]
        prob = raw_prob[:, 1:]
        ridx, cidx = (prob > threshold).nonzero(as_tuple=True)                    # sync 1: number of candidates
        cand_box = boxes[ridx, cidx].contiguous()
        cand_score = prob[ridx, cidx].contiguous()
        keep = ops.batched_nms(cand_box, cand_score, cidx, 0.3)                   # sync 2: number kept (score-descending)
        bbox = cand_box[keep].cpu().numpy()
        label = cidx[keep].cpu().numpy()                                          # 0-based: l - 1 in the reference
        score = cand_score[keep].cpu().numpy()
        order = np.argsort(label, kind="stable")                                  # class-major, score-descending inside a class
        bbox = torch.from_numpy(bbox[order].astype(np.float32))
        label = torch.from_numpy(label[order].astype(np.int32))
        score = torch.from_numpy(score[order].astype(np.float32))
        return bbox, label, score
