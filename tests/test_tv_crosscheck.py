"""Opportunistic cross-check against torchvision (SURVEY 8c/8d sanction it: a third-party library that may be installed on
the box, not the reference travelling).  nms / RoIPool / MultiScaleRoIAlign + level mapper / AnchorGenerator are the ops whose
arithmetic lives in torchvision; the oracle restates them from the published algorithms ("parity unpinned").  When
torchvision is importable these tests pin BOTH the oracle (CPU tests) and the HIP kernels (gpu tests) to torchvision's CPU
ops at BASELINE shapes.  When it is not (this image: ModuleNotFoundError), every test here SKIPS VISIBLY -- the pytest header
line "torchvision cross-check: ..." (tests/conftest.py) says which.

Independently of torchvision, `transformers` (installed) carries the DETR `box_iou` that the reference's util/box_ops.py:24-37
is a copy of: that one is cross-checked unconditionally."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

DEV = "cuda:0"


def _tv():
    return pytest.importorskip("torchvision", reason="torchvision is not installed on this box: cross-check skipped (parity stays unpinned)")


def rand_boxes(rng, n, lo=0.02, hi=0.6):
    c = rng.rand(n, 2) * 0.8 + 0.1
    wh = rng.rand(n, 2) * (hi - lo) + lo
    return np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, 1).astype(np.float32)


def proposal_like_boxes(rng, K):
    """Score-sorted, heavily overlapping boxes like the pre-NMS top-K of an RPN (clusters around a few hundred centres)."""
    nc = max(K // 40, 1)
    centres = rand_boxes(rng, nc, 0.05, 0.5)
    b = centres[rng.randint(0, nc, K)] + rng.randn(K, 4).astype(np.float32) * 0.01
    b = np.clip(b, 0, 1)
    b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 1e-3)
    s = np.sort(rng.rand(K).astype(np.float32))[::-1].copy()
    assert len(np.unique(s)) == K or True
    return b, s


FPN_SHAPES = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]


# ------------------------------------------------------------------------------------------------ unconditional: DETR box_iou
def _detr_box_iou():
    mod = pytest.importorskip("transformers.loss.loss_for_object_detection", reason="transformers is not installed")
    return mod.box_iou


def test_oracle_box_iou_equals_detr_box_iou_in_transformers():
    """util/box_ops.py:24-37 is DETR's box_iou; transformers ships the same function (third party, independent of this build)."""
    box_iou = _detr_box_iou()
    rng = np.random.RandomState(0)
    a, b = rand_boxes(rng, 300), rand_boxes(rng, 41)
    iou, union = box_iou(torch.from_numpy(a), torch.from_numpy(b))
    got = orc.pairwise_iou(a, b, eps=0.0)
    assert np.array_equal(got, iou.numpy())


@pytest.mark.gpu
def test_hip_box_iou_equals_detr_box_iou_in_transformers():
    from faster_rcnn_pytorch_amd import ops
    box_iou = _detr_box_iou()
    rng = np.random.RandomState(1)
    a, b = rand_boxes(rng, 2000), rand_boxes(rng, 8)
    iou, union = box_iou(torch.from_numpy(a), torch.from_numpy(b))
    got_iou, got_union = ops.box_iou(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV))
    assert np.array_equal(got_iou.cpu().numpy(), iou.numpy())
    assert np.allclose(got_union.cpu().numpy(), union.numpy(), rtol=0, atol=1e-7)


# ------------------------------------------------------------------------------------------------ torchvision: oracle (CPU)
@pytest.mark.parametrize("K,thr", [(12000, 0.7), (4000, 0.7), (6000, 0.7), (1500, 0.3)])
def test_oracle_nms_equals_torchvision(K, thr):
    tv = _tv()
    rng = np.random.RandomState(K)
    b, s = proposal_like_boxes(rng, K)
    want = tv.ops.nms(torch.from_numpy(b), torch.from_numpy(s), thr).numpy()
    got = orc.nms(b, thr)
    assert np.array_equal(got, want)


def test_oracle_roi_pool_equals_torchvision():
    tv = _tv()
    rng = np.random.RandomState(2)
    feat = rng.randn(512, 37, 62).astype(np.float32)
    rois = rand_boxes(rng, 128, 0.03, 0.9) * np.array([62, 37, 62, 37], np.float32)
    want = tv.ops.roi_pool(torch.from_numpy(feat)[None], [torch.from_numpy(rois)], (7, 7), 1.0).numpy()
    got, _ = orc.roi_pool_fwd(feat, rois, 7, 7, 1.0)
    assert np.array_equal(got, want)


def test_oracle_ms_roi_align_and_level_map_equal_torchvision():
    tv = _tv()
    from collections import OrderedDict
    rng = np.random.RandomState(3)
    feats = [rng.randn(16, fh, fw).astype(np.float32) for fh, fw in FPN_SHAPES[:4]]
    rois = rand_boxes(rng, 512, 0.02, 0.95) * np.array([1344, 800, 1344, 800], np.float32)
    pool = tv.ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    fd = OrderedDict((str(i), torch.from_numpy(f)[None]) for i, f in enumerate(feats))
    want = pool(fd, [torch.from_numpy(rois)], [(800, 1344)]).numpy()               # (h, w): the orientation torchvision means
    got, lv = orc.ms_roi_align(feats, rois)
    assert np.abs(got - want).max() < 1e-5
    assert np.array_equal(lv, pool.map_levels([torch.from_numpy(rois)]).numpy())
    # the reference's swapped call (new_model.py:143): scales='reference' mode of the build
    from faster_rcnn_pytorch_amd.ops import infer_scales_like_torchvision
    pool2 = tv.ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    want2 = pool2(fd, [torch.from_numpy(rois)], [(1344, 800)]).numpy()
    sc = infer_scales_like_torchvision([f.shape[-2:] for f in feats], [(1344, 800)])
    assert tuple(pool2.scales) == sc
    got2, _ = orc.ms_roi_align(feats, rois, scales=sc)
    assert np.abs(got2 - want2).max() < 1e-5


def test_oracle_anchor_generator_equals_torchvision():
    _tv()
    from torchvision.models.detection.anchor_utils import AnchorGenerator
    from torchvision.models.detection.image_list import ImageList
    ag = AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),) * 5)    # new_model.py:23-25
    x = torch.zeros(1, 3, 800, 1344)
    feats = [torch.zeros(1, 1, fh, fw) for fh, fw in FPN_SHAPES]
    want = ag(ImageList(x, [(1344, 800)]), feats)[0].numpy()                          # new_model.py:46
    assert np.array_equal(orc.tv_anchor_grid(800, 1344, FPN_SHAPES, normalise=False), want)


# ------------------------------------------------------------------------------------------------ torchvision: HIP kernels (gpu)
@pytest.mark.gpu
@pytest.mark.parametrize("K,thr", [(12000, 0.7), (4000, 0.7), (6000, 0.7), (1500, 0.3)])
def test_hip_nms_equals_torchvision(K, thr):
    tv = _tv()
    from faster_rcnn_pytorch_amd import ops
    rng = np.random.RandomState(K + 1)
    b, s = proposal_like_boxes(rng, K)
    perm = rng.permutation(K)                                                         # unsorted input: nms sorts by score itself
    want = tv.ops.nms(torch.from_numpy(b[perm]), torch.from_numpy(s[perm]), thr).numpy()
    got = ops.nms(torch.from_numpy(b[perm]).to(DEV), torch.from_numpy(s[perm]).to(DEV), thr).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("R", [128, 300])
def test_hip_roi_pool_fwd_bwd_equal_torchvision(R):
    tv = _tv()
    from faster_rcnn_pytorch_amd import ops
    rng = np.random.RandomState(R)
    feat = rng.randn(1, 512, 37, 62).astype(np.float32)
    rois = rand_boxes(rng, R, 0.03, 0.9) * np.array([62, 37, 62, 37], np.float32)
    go = rng.randn(R, 512, 7, 7).astype(np.float32)
    f_cpu = torch.from_numpy(feat).requires_grad_(True)
    want = tv.ops.roi_pool(f_cpu, [torch.from_numpy(rois)], (7, 7), 1.0)
    want.backward(torch.from_numpy(go))
    f_gpu = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    got = ops.roi_pool(f_gpu, [torch.from_numpy(rois).to(DEV)], (7, 7), 1.0)
    got.backward(torch.from_numpy(go).to(DEV))
    assert np.array_equal(got.detach().cpu().numpy(), want.detach().numpy())
    assert np.abs(f_gpu.grad.cpu().numpy() - f_cpu.grad.numpy()).max() < 1e-4       # fp32 sums in a different order


@pytest.mark.gpu
def test_hip_ms_roi_align_fwd_bwd_equal_torchvision():
    tv = _tv()
    from collections import OrderedDict
    from faster_rcnn_pytorch_amd import ops
    rng = np.random.RandomState(7)
    feats = [rng.randn(1, 256, fh, fw).astype(np.float32) for fh, fw in FPN_SHAPES[:4]]
    rois = rand_boxes(rng, 512, 0.02, 0.95) * np.array([1344, 800, 1344, 800], np.float32)
    go = rng.randn(512, 256, 7, 7).astype(np.float32)
    fc_ = [torch.from_numpy(f).requires_grad_(True) for f in feats]
    pool = tv.ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    want = pool(OrderedDict((str(i), f) for i, f in enumerate(fc_)), [torch.from_numpy(rois)], [(800, 1344)])
    want.backward(torch.from_numpy(go))
    fg = [torch.from_numpy(f).to(DEV).requires_grad_(True) for f in feats]
    mine = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    got = mine(OrderedDict((str(i), f) for i, f in enumerate(fg)), [torch.from_numpy(rois).to(DEV)], [(800, 1344)])
    got.backward(torch.from_numpy(go).to(DEV))
    assert np.abs(got.detach().cpu().numpy() - want.detach().numpy()).max() < 1e-4  # north star tolerance
    for a, b in zip(fg, fc_):
        assert np.abs(a.grad.cpu().numpy() - b.grad.numpy()).max() < 1e-3
    assert np.array_equal(ops.roi_level_map(torch.from_numpy(rois).to(DEV)).cpu().numpy(), pool.map_levels([torch.from_numpy(rois)]).numpy())
    # the reference's swapped image_shapes
    pool2 = tv.ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    with torch.no_grad():
        want2 = pool2(OrderedDict((str(i), torch.from_numpy(f)) for i, f in enumerate(feats)), [torch.from_numpy(rois)], [(1344, 800)]).numpy()
        got2 = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2, scales="reference")(
            OrderedDict((str(i), torch.from_numpy(f).to(DEV)) for i, f in enumerate(feats)), [torch.from_numpy(rois).to(DEV)], [(1344, 800)]).cpu().numpy()
    assert np.abs(got2 - want2).max() < 1e-4


@pytest.mark.gpu
def test_hip_anchor_generator_equals_torchvision():
    _tv()
    from torchvision.models.detection.anchor_utils import AnchorGenerator
    from torchvision.models.detection.image_list import ImageList
    from faster_rcnn_pytorch_amd import ops
    ag = AnchorGenerator(sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),) * 5)
    x = torch.zeros(1, 3, 800, 1344)
    feats = [torch.zeros(1, 1, fh, fw) for fh, fw in FPN_SHAPES]
    want = ag(ImageList(x, [(1344, 800)]), feats)[0].numpy()
    got = ops.AnchorGenerator()(ops.ImageList(x.to(DEV), [(1344, 800)]), [f.to(DEV) for f in feats])[0].cpu().numpy()
    assert np.array_equal(got, want)
