"""VGG16 Faster R-CNN -- host-side mirror of the reference's models/model.py (authoritative variant:
models/model_.py) on top of the HIP hot path.

Same classes, constructor arguments, sub-module names and forward()/predict() signatures as the
reference (models/model.py:12-402), so train.py:31 / test.py:60 work unchanged:
    FRCNN(num_classes).forward(x, bbox, label)
        -> (rpn_cls[1,N,2], rpn_reg[1,N,4], head_cls[R,C], head_reg[R,4]),
           (tgt_rpn_cls[N] i64, tgt_rpn_reg[N,4], tgt_cls[R] i64, tgt_reg[R,4])
    FRCNN.predict(x, opts | threshold) -> (bbox[M,4] f32, label[M] i32, score[M] f32)

What changed underneath (each step cites the reference lines it replaces):
  * anchors: cached in HBM per image shape / regenerated in registers (model.py:310-312 numpy + H2D)
  * RegionProposal: one enqueue of 5 HIP kernels, fixed-capacity output + device count
    (model.py:17-58: ~20 launches, 3 host syncs, torchvision nms with a host scan)
  * target makers: 3 + 1 HIP kernels, sampling by device Philox keys (model.py:127-266: ~45 launches,
    7-9 host syncs, CPU randperm).  sampling='host' reproduces the reference's RNG stream exactly
    (torch.randperm on the CPU generator, RPN maker first: SURVEY Q4) at the price of the same syncs.
  * RoIPool: HIP forward/backward (torchvision.ops.RoIPool, model.py:97,113)
  * RPN head: the 3x3 convolution on the fp32 matrix cores (Winograd stage, csrc/rpn_conv_f32.hip), bias + ReLU + both 1x1 heads + the
    NHWC layout in one more MFMA kernel (csrc/rpn_head.hip) (model.py:68-83: three MIOpen calls + two transposing copies)
  * extractor: its thirteen stride-1 3x3 convolutions run on the same fp32 stage, each with the ReLU (and, where the 4 x 4 tile is used,
    the 2 x 2 max-pool) behind it in the same call (VGGExtractor below; csrc/conv_c3.hip for the three-channel first layer).  Layers the
    stage does not take (autocast, batch > 1) run as the torch modules they are.
The FC head (classifier / cls_head / reg_head), the optimizer and the loss's reductions over torch tensors stay on PyTorch-ROCm (hipBLASLt).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .anchor import FRCNNAnchorMaker


def _unwrap(t):
    """bbox/label calling convention: a list with one tensor (model_.py:130-131,188) or a bare tensor (train.py:18-19)."""
    return t[0] if isinstance(t, (list, tuple)) else t


def normal_init(m, mean, stddev):
    m.weight.data.normal_(mean, stddev)
    m.bias.data.zero_()


def vgg16_features():
    """torchvision.models.vgg16().features with identical layer indices (state_dict keys extractor.N.*);
    random init here: pretrained weights need the network."""
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    layers, c_in = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            c_in = v
    for m in layers:
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(m.bias, 0)
    return layers


class VGGExtractor(nn.Sequential):
    """`nn.Sequential(*list(vgg16.features)[:-1])` (models/model.py:279-281) with the same children, indices and state_dict keys.  Its forward walks
    the children like nn.Sequential does, except that a 3x3 convolution the fp32 Winograd stage takes (ops.conv3x3_supported: 128-multiples of
    channels, batch 1, fp32 on the device) runs there TOGETHER with the ReLU behind it -- bias and ReLU in the output transform, the ReLU's
    backward in the gradient kernels' transforms -- instead of three vendor / elementwise launches forward and four backward; a 2 x 2 max-pool
    behind such a pair joins the same call where the stage uses its 4 x 4 tile; the first convolution (three input channels) has kernels of
    its own.  Every other child (small maps' pools; any layer under autocast or with batch > 1) runs as it is."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride == (1, 1) and m.dilation == (1, 1)
                    and m.groups == 1 and m.bias is not None and not torch.is_autocast_enabled() and ops.conv3x3_supported(x, m.weight)):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                p = mods[i + 2] if fuse and i + 2 < len(mods) else None
                pool = (isinstance(p, nn.MaxPool2d) and p.kernel_size in (2, (2, 2)) and p.stride in (2, (2, 2)) and p.padding in (0, (0, 0))
                        and p.dilation in (1, (1, 1)) and not p.ceil_mode and not p.return_indices and ops.conv3x3_pool_supported(x))
                x = ops.conv3x3(x, m.weight, m.bias, relu=fuse, pool=pool)      # conv + ReLU (+ the 2 x 2 max-pool behind it) in one stage call
                i += 3 if pool else (2 if fuse else 1)
                continue
            if (isinstance(m, nn.Conv2d) and m.in_channels == 3 and m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride == (1, 1)
                    and m.dilation == (1, 1) and m.groups == 1 and not torch.is_autocast_enabled() and ops.conv3x3_c3_supported(x, m.weight)):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ops.conv3x3_c3(x, m.weight, m.bias, relu=fuse)              # features[0] (+ [1]): three input channels, a byte mover (csrc/conv_c3.hip)
                i += 2 if fuse else 1
                continue
            x = m(x)
            i += 1
        return x


class RegionProposal(nn.Module):
    """models/model.py:12-58."""

    def __init__(self):
        super().__init__()
        self.min_size = 1
        self.nms_threshold = 0.7

    @staticmethod
    def top_k(mode):
        return (6000, 300) if mode == "test" else (12000, 2000)              # model.py:24-28

    def propose(self, cls, reg, anchor, mode, grid=None, want_src=False):
        """Asynchronous form: (rois [P,4] fixed capacity, count int32[1] on device, src | None)."""
        pre, post = self.top_k(mode)
        return ops.region_proposal(reg.detach().float(), cls.detach().float(), anchor, self.min_size / 1000, pre, self.nms_threshold, post,
                                   grid=grid, want_src=want_src)

    def forward(self, cls, reg, anchor, mode):
        """Reference signature: returns roi_tensor [n,4] (variable length -> one host sync for n)."""
        rois, cnt, _ = self.propose(cls, reg, anchor, mode)
        return rois[:ops.host_count(cnt, 'region_proposal')]


class RegionProposalNetwork(nn.Module):
    """models/model.py:61-84."""

    def __init__(self, in_channels=512, out_channels=512):
        super().__init__()
        num_anchors = 9
        self.inter_layer = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.cls_layer = nn.Conv2d(in_channels, num_anchors * 2, kernel_size=1)
        self.reg_layer = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=1)
        normal_init(self.inter_layer, 0, 0.01)
        normal_init(self.cls_layer, 0, 0.01)
        normal_init(self.reg_layer, 0, 0.01)

    def forward(self, features):
        if (features.is_cuda and features.dtype == torch.float32 and features.size(0) == 1 and not torch.is_autocast_enabled()
                and ops.rpn_conv3x3_supported([features], self.inter_layer.weight)):
            # the 3x3 without its bias on the fp32 matrix cores (csrc/rpn_conv_f32.hip: forward, data and weight gradient hand-written);
            # bias + ReLU + both 1x1 heads + the NHWC layout in one more MFMA kernel
            raw = ops.rpn_conv3x3([features], self.inter_layer.weight)[0]
            return ops.rpn_head_tail(raw, self.inter_layer.bias, self.cls_layer.weight, self.cls_layer.bias,
                                     self.reg_layer.weight, self.reg_layer.bias)
        batch_size = features.size(0)                                              # reference form (autocast / batch > 1)
        x = torch.relu(self.inter_layer(features))
        pred_cls = self.cls_layer(x)
        pred_reg = self.reg_layer(x)
        pred_reg = pred_reg.permute(0, 2, 3, 1).contiguous().view(batch_size, -1, 4)
        pred_cls = pred_cls.permute(0, 2, 3, 1).contiguous().view(batch_size, -1, 2)
        return pred_cls, pred_reg


class FastRCNNHead(nn.Module):
    """models/model.py:87-120."""

    def __init__(self, num_classes, roi_size, classifier):
        super().__init__()
        self.num_classes = num_classes
        self.cls_head = nn.Linear(4096, num_classes)
        self.reg_head = nn.Linear(4096, num_classes * 4)
        self.roi_pool = ops.RoIPool(output_size=(roi_size, roi_size), spatial_scale=1.)
        self.classifier = classifier
        normal_init(self.cls_head, 0, 0.01)
        normal_init(self.reg_head, 0, 0.001)

    def forward(self, features, roi):
        f_height, f_width = features.size()[2:]
        scale = ops.const_tensor((f_width, f_height, f_width, f_height), roi.device)   # cached: torch.tensor(list, device=cuda) is a blocking copy
        scaled_roi = roi * scale                                               # model.py:107-109
        pool = self.roi_pool(features.float(), [scaled_roi])           # the hot path computes in fp32 (also under autocast)
        x = pool.view(pool.size(0), -1)
        x = self.classifier(x)
        return self.cls_head(x), self.reg_head(x)


class _Sampler(object):
    """Sampling policy shared by the two target makers."""

    def __init__(self, sampling="device", seed=0):
        if sampling not in ("device", "host"):
            raise ValueError("sampling must be 'device' (Philox on the GPU, no sync) or 'host' (reference RNG stream)")
        self.sampling = sampling
        self.seed = int(seed)
        # sticky device-side error word: the sync-free ('device') target makers OR their failure bits into it; the training
        # loop reads it where it syncs anyway (FRCNN.check_device_status()).  The same failures also make the loss NaN.
        self.status = ops.DeviceStatus()
        self._state = {}

    def state(self, device):
        """The device-resident Philox stream (seed, offset) of this model on `device`: each target-maker call uses the pair it
        finds and leaves offset + 1 behind, ON THE DEVICE -- no host-side counter rides in the launch arguments, so the whole
        step can be captured in a HIP graph and still draws fresh samples at every replay."""
        key = str(device)
        st = self._state.get(key)
        if st is None:
            st = ops.philox_state(self.seed, 1, device)
            self._state[key] = st
        return st

    def reseed(self, seed=None, offset=1):
        """Restart the stream (tests; deterministic replays): a device-side copy into the existing state tensors, no sync."""
        if seed is not None:
            self.seed = int(seed)
        for key, st in self._state.items():
            st.copy_(ops.philox_state(self.seed, offset, st.device))


class FastRcnnTargetMaker(nn.Module):
    """models/model_.py:123-179."""

    def __init__(self, sampler=None):
        super().__init__()
        self.sampler = sampler or _Sampler()

    def forward(self, bbox, label, rois, n_rois=None):
        bbox = _unwrap(bbox)
        label = _unwrap(label).to(torch.int64)
        s = self.sampler
        if s.sampling == "host":
            c = ops.head_targets(rois, bbox, label, n_rois=n_rois)[4].cpu().tolist()     # learn the candidate counts
            perm_pos = torch.randperm(c[0])                                                # model_.py:149
            perm_neg = torch.randperm(c[1])                                                # model_.py:155
            cls, reg, srois, _, counts = ops.head_targets(rois, bbox, label, n_rois=n_rois, perm_pos=perm_pos, perm_neg=perm_neg)
            if counts.cpu().tolist()[2] != 128:
                raise RuntimeError("FastRcnnTargetMaker: fewer than 128 samples (the reference fails here too, model_.py:340)")
        else:
            cls, reg, srois, _, _ = ops.head_targets(rois, bbox, label, n_rois=n_rois, philox_state=s.state(rois.device),
                                                     status=s.status.word(rois.device))
        return cls, reg, srois


class RPNTargetMaker(nn.Module):
    """models/model_.py:182-266."""

    def __init__(self, sampler=None):
        super().__init__()
        self.sampler = sampler or _Sampler()

    def forward(self, bbox, anchor):
        bbox = _unwrap(bbox)
        s = self.sampler
        if s.sampling == "host":
            n_pos, n_neg = ops.rpn_targets(anchor, bbox)[2].cpu().tolist()[:2]
            perm_pos = torch.randperm(n_pos) if n_pos > 128 else None                      # model_.py:225-229
            perm_neg = torch.randperm(n_neg) if n_neg > 256 - n_pos else None              # model_.py:231-236
            cls, reg, counts = ops.rpn_targets(anchor, bbox, perm_pos=perm_pos, perm_neg=perm_neg)
            if counts.cpu().tolist()[2] != 0:
                raise RuntimeError("RPNTargetMaker: permutation length mismatch")
        else:
            cls, reg, _ = ops.rpn_targets(anchor, bbox, philox_state=s.state(anchor.device))
        return cls, reg


class FRCNN(nn.Module):
    """models/model.py:269-402 (VGG16 Faster R-CNN)."""

    def __init__(self, num_classes=81, pretrained=False, sampling="device", seed=0):
        super().__init__()
        if pretrained:
            raise RuntimeError("pretrained weights need the network (gdown / torchvision hub); load a state_dict instead")
        self.num_classes = num_classes
        self.extractor = VGGExtractor(*vgg16_features()[:-1])                  # model.py:279-281: drop the last max-pool
        self.classifier = nn.Sequential(nn.Linear(in_features=25088, out_features=4096), nn.ReLU(inplace=True),
                                        nn.Linear(in_features=4096, out_features=4096), nn.ReLU(inplace=True))
        self.sampler = _Sampler(sampling, seed)
        self.rpn = RegionProposalNetwork()
        self.rp = RegionProposal()
        self.anchor_maker = FRCNNAnchorMaker()
        self.rpn_target_maker = RPNTargetMaker(self.sampler)
        self.fast_rcnn_target_maker = FastRcnnTargetMaker(self.sampler)
        self.fast_rcnn_head = FastRCNNHead(num_classes=num_classes, roi_size=7, classifier=self.classifier)
        self.last_proposal_count = None

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def check_device_status(self):
        """Raises if a device-side failure was recorded since the last call (an aborted NMS scan upstream of the head target
        maker, or fewer than 128 RoI samples -- where the reference throws, model.py:340).  One host sync: call it at the
        logging / checkpoint interval.  The same failures turn the step's loss into NaN, so a caller that prints loss.item()
        sees them even without this call."""
        self.sampler.status.check()

    def graph_stages(self):
        """Where parallel.GraphStep cuts the backward: everything behind the RoI pooling (the FC head: 120 M of the 137 M parameters) has its
        gradient first, and its all-reduce runs under the backward of the rest."""
        return {"cut_module": self.fast_rcnn_head.roi_pool, "late_modules": (self.classifier, self.fast_rcnn_head.cls_head, self.fast_rcnn_head.reg_head)}

    def forward(self, x, bbox, label):
        hw = x.size()[2:]
        # 5. rpn targets depend only on the anchors and the ground truth (model.py:324), not on the network: they are enqueued first.
        #    (Round 1 ran them on a second HIP stream "underneath" the backbone.  Measured A/B on one box, four runs: the fork / join
        #    and the single-workgroup sampler competing with MIOpen's kernels cost more than the 48 us they hide: 14.77-14.80 ms per step
        #    with the side stream, 14.63-14.67 ms without; the same experiment in the FPN mirror: 18.64 vs 18.17 ms.)
        anchor = self.anchor_maker.device_anchors(hw, x.device)           # model.py:310-312: resident in HBM
        target_rpn_cls, target_rpn_reg = self.rpn_target_maker(bbox=bbox, anchor=anchor)
        # 1. extract features                                                  model.py:307
        features = self.extractor(x)
        # 3. forward rpn                                                       model.py:315
        pred_rpn_cls, pred_rpn_reg = self.rpn(features)
        # 4. propose regions -> fixed-capacity rois + device count (anchors regenerated in registers)   model.py:318
        rois, n_rois, _ = self.rp.propose(pred_rpn_cls.squeeze(0), pred_rpn_reg.squeeze(0), None, "train",
                                          grid=self.anchor_maker.grid_desc(hw))
        self.last_proposal_count = n_rois                                     # device int32[1]: how many proposals survived NMS (<= 2000)
        # 6. fast rcnn targets                                                 model.py:328
        target_fast_rcnn_cls, target_fast_rcnn_reg, sample_rois = self.fast_rcnn_target_maker(bbox=bbox, label=label, rois=rois,
                                                                                              n_rois=n_rois)
        # 7. fast rcnn head                                                    model.py:335
        pred_fast_rcnn_cls, pred_fast_rcnn_reg = self.fast_rcnn_head(features, sample_rois)
        # 8. regression row of the target class                                model.py:340-341
        pred_fast_rcnn_reg = pred_fast_rcnn_reg.reshape(128, -1, 4)
        # row i keeps the regression of its target class (reference: pred[arange(128), cls], advanced indexing).  torch.gather selects the
        # same elements with one kernel forward and one backward; the indexing form costs an arange, an index kernel and a rocprim sort
        # + scatter in backward (~30 us and 7 launches per step).
        pred_fast_rcnn_reg = torch.gather(pred_fast_rcnn_reg, 1, target_fast_rcnn_cls.clamp(min=0).view(-1, 1, 1).expand(-1, 1, 4)).squeeze(1)   # (-1 = unsampled row: its loss is NaN anyway)
        return (pred_rpn_cls, pred_rpn_reg, pred_fast_rcnn_cls, pred_fast_rcnn_reg), \
               (target_rpn_cls, target_rpn_reg, target_fast_rcnn_cls, target_fast_rcnn_reg)

    @torch.no_grad()
    def predict(self, x, opts):
        """opts: an object with .thres (model.py:346) or a bare float threshold (model_.py:346)."""
        threshold = float(getattr(opts, "thres", opts))
        features = self.extractor(x)
        hw = x.size()[2:]
        pred_rpn_cls, pred_rpn_reg = self.rpn(features)
        rois, n_rois, _ = self.rp.propose(pred_rpn_cls.squeeze(0), pred_rpn_reg.squeeze(0), None, "test",
                                          grid=self.anchor_maker.grid_desc(hw))
        rois = rois[:ops.host_count(n_rois, 'region_proposal')]
        pred_fast_rcnn_cls, pred_fast_rcnn_reg = self.fast_rcnn_head(features, rois)
        pred_cls = torch.softmax(pred_fast_rcnn_cls, dim=-1)                    # model.py:369
        pred_fast_rcnn_reg = pred_fast_rcnn_reg.reshape(-1, self.num_classes, 4)
        pred_fast_rcnn_reg = pred_fast_rcnn_reg * ops.const_tensor((0.1, 0.1, 0.2, 0.2), x.device)   # model.py:372
        rois = rois.reshape(-1, 1, 4).expand_as(pred_fast_rcnn_reg)
        pred_bbox = ops.decode(pred_fast_rcnn_reg.reshape(-1, 4).contiguous(), ops.xy_to_cxcy(rois.reshape(-1, 4).contiguous()))
        pred_bbox = ops.cxcy_to_xy(pred_bbox)
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


def build_model(opts=None, num_classes=21, device=None, **kw):
    """models/build.py:7-19: construct, move to the device, wrap in DDP when torch.distributed is initialised."""
    from .parallel import wrap_ddp
    num_classes = getattr(opts, "num_classes", num_classes) if opts is not None else num_classes
    model = FRCNN(num_classes=num_classes, **kw)
    if device is not None:
        model = model.to(device)
        model = wrap_ddp(model, torch.device(device))
    return model
