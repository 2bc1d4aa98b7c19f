"""CPU-only tests of the host logic: losses vs the reference's golden values, data-parallel plumbing
with gloo (world_size 2), sharding."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loss_matches_reference_golden(golden):
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    g = golden("loss")                                                      # produced by the reference's losses/loss.py
    t = lambda k: torch.from_numpy(g[k])                                    # noqa: E731
    out = FRCNNLoss(None)((t("p_rpn_cls"), t("p_rpn_reg"), t("p_head_cls"), t("p_head_reg")),
                          (t("t_rpn_cls"), t("t_rpn_reg"), t("t_head_cls"), t("t_head_reg")))
    got = np.array([float(o) for o in out], np.float32)
    assert np.allclose(got, g["losses"], rtol=2e-6, atol=1e-6)


def test_loss_gradient_equals_boolean_mask_form(golden):
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    from oracle.model_ref import ref_loss
    g = golden("loss")
    ps = [torch.from_numpy(g[k]).clone().requires_grad_(True) for k in ("p_rpn_cls", "p_rpn_reg", "p_head_cls", "p_head_reg")]
    ts = [torch.from_numpy(g[k]) for k in ("t_rpn_cls", "t_rpn_reg", "t_head_cls", "t_head_reg")]
    FRCNNLoss(None)(ps, ts)[0].backward()
    g1 = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    ref_loss(ps, ts)[0].backward()
    for a, p in zip(g1, ps):
        assert torch.allclose(a, p.grad, atol=1e-7)


def test_shard_indices_partition():
    from faster_rcnn_pytorch_amd.parallel import shard_indices
    for world in (1, 2, 3, 8):
        parts = [shard_indices(29, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(29))
        assert all(p == list(range(r, 29, world)) for r, p in enumerate(parts))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from faster_rcnn_pytorch_amd import parallel
    r, lr, w, dev = parallel.init_for_distributed(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    ddp = parallel.wrap_ddp(model, dev)
    # each rank sees a different "image" (its shard); after backward the gradients must be the rank-average
    xs = torch.arange(4 * 8, dtype=torch.float32).reshape(4, 8) / 10
    mine = parallel.shard_indices(4, r, w)
    loss = ddp(xs[mine]).pow(2).mean()
    loss.backward()
    grad = model[0].weight.grad.clone()
    t = parallel.max_over_ranks(1.0 + r, dev)
    s = parallel.sum_over_ranks(1.0 + r, dev)
    parallel.barrier()
    q.put((r, grad.numpy(), t, s))
    dist.destroy_process_group()


def test_ddp_gloo_world2_gradient_average_and_timing_rule():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.allclose(res[0][1], res[1][1])                               # all-reduced gradients agree
    assert res[0][2] == res[1][2] == 2.0 and res[0][3] == res[1][3] == 3.0  # MAX / SUM over ranks
    # and equal the average of the per-shard gradients computed in one process
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    xs = torch.arange(4 * 8, dtype=torch.float32).reshape(4, 8) / 10
    gs = []
    for r in range(2):
        model.zero_grad()
        model(xs[list(range(r, 4, 2))]).pow(2).mean().backward()
        gs.append(model[0].weight.grad.clone())
    assert np.allclose(res[0][1], ((gs[0] + gs[1]) / 2).numpy(), atol=1e-6)


def _ddp_struct_worker(r, world, port, q):
    os.environ.update(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from faster_rcnn_pytorch_amd import parallel
    from faster_rcnn_pytorch_amd.model import FRCNN
    _, _, w, dev = parallel.init_for_distributed(backend="gloo")
    torch.manual_seed(0)
    model = FRCNN(num_classes=21)                      # module construction needs no GPU (only forward does)
    net = parallel.wrap_ddp(model, dev)
    rep = parallel.ddp_report(net)
    # a gradient for every parameter, produced through autograd so that DDP's per-parameter hooks fire: rank-dependent values
    net.require_backward_grad_sync = True
    net.reducer.prepare_for_backward([])
    loss = sum((p * float(1 + r + i % 3)).sum() for i, p in enumerate(model.parameters()))
    loss.backward()
    g = [float(p.grad.flatten()[0]) for p in model.parameters()]
    q.put((r, rep, g, parallel.ddp_report(net)))
    parallel.shutdown()


def test_ddp_wraps_every_parameter_once_with_the_aliased_classifier():
    """the VGG mirror registers `classifier` under two names (models/model.py:282,298).  DDP must hook each of the
    40 parameter tensors exactly once (44 names), reduce 548 MB per step, and average the per-parameter gradients across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 2000
    procs = [ctx.Process(target=_ddp_struct_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r, rep, g, rep2 in res:
        assert rep["num_parameter_tensors"] == rep["unique_trainable_parameters"] == 40
        assert rep["registered_names_with_aliases"] == 44                       # classifier.{0,2}.{weight,bias} twice
        assert rep["total_parameter_size_bytes"] == rep["unique_trainable_bytes"] == 548312956
        assert rep["bucket_cap_bytes"] == 100 * 1024 * 1024 and rep["gradient_as_bucket_view"] and not rep["find_unused_parameters"]
        assert sum(rep["bucket_sizes"]) == 548312956
    # every parameter's gradient was all-reduced exactly once: mean over ranks of (1 + r + i % 3)
    for i in range(40):
        exp = ((1 + 0 + i % 3) + (1 + 1 + i % 3)) / 2
        assert res[0][2][i] == res[1][2][i] == exp


def test_torchvision_scale_inference_with_the_references_swapped_shapes():
    """models/new_model.py:143 passes image_shapes=[(w, h)] where torchvision's pooler reads (h, w): under
    torchvision the FEATURE HEIGHT is divided by the IMAGE WIDTH.  ops.MultiScaleRoIAlign(scales='reference') reproduces that."""
    from faster_rcnn_pytorch_amd import ops
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    # landscape COCO frame 800 x 1344 as the reference passes it: [(w, h)] = [(1344, 800)] -> 200/1344 = 0.149 -> 2^-3
    assert ops.infer_scales_like_torchvision(shapes, [(1344, 800)]) == (1 / 8, 1 / 16, 1 / 32, 1 / 64)
    # the same frame with the axes the way torchvision means them: the true strides
    assert ops.infer_scales_like_torchvision(shapes, [(800, 1344)]) == (1 / 4, 1 / 8, 1 / 16, 1 / 32)
    # square frames are unaffected by the swap
    sq = [(200, 200), (100, 100), (50, 50), (25, 25)]
    assert ops.infer_scales_like_torchvision(sq, [(800, 800)]) == (1 / 4, 1 / 8, 1 / 16, 1 / 32)
    # portrait 1344 x 800 passed as (w, h) = (800, 1344): 336/800 = 0.42 -> 2^-1
    pt = [(336, 200), (168, 100), (84, 50), (42, 25)]
    assert ops.infer_scales_like_torchvision(pt, [(800, 1344)]) == (1 / 2, 1 / 4, 1 / 8, 1 / 16)
    m = ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
    assert m.scales == (1 / 4, 1 / 8, 1 / 16, 1 / 32)
    assert ops.MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2, scales="reference").scales == "reference"
    with pytest.raises(ValueError):
        ops.MultiScaleRoIAlign(["0"], 7, 2, scales="torchvision")


def test_device_status_names():
    from faster_rcnn_pytorch_amd import _lib, ops
    assert ops.describe_status(0) == []
    names = ops.describe_status(_lib.HT_ERR_UPSTREAM_ABORT | _lib.HT_ERR_SHORT)
    assert len(names) == 2 and "aborted NMS scan" in names[0] and "fewer RoI samples" in names[1]
    st = ops.DeviceStatus()
    w = st.word(torch.device("cpu"))
    st.check()                                                             # clean: no raise
    w |= _lib.HT_ERR_UPSTREAM_ABORT
    with pytest.raises(_lib.FrcnnError, match="aborted NMS scan"):
        st.check()
    st.check()                                                             # check() cleared the word


def test_frozen_batchnorm_cache_follows_its_buffers():
    """new_model.FrozenBatchNorm2d caches (scale, bias) per state of its four frozen buffers: the cached form must equal
    torchvision's expression bit for bit, and a load_state_dict / in-place update / .to() must invalidate it."""
    from faster_rcnn_pytorch_amd.new_model import FrozenBatchNorm2d
    g = torch.Generator().manual_seed(5)
    bn = FrozenBatchNorm2d(8)

    def expect(x):
        scale = (bn.weight * (bn.running_var + bn.eps).rsqrt()).reshape(1, -1, 1, 1)
        bias = bn.bias.reshape(1, -1, 1, 1) - bn.running_mean.reshape(1, -1, 1, 1) * scale
        return x * scale + bias
    x = torch.randn(2, 8, 5, 7, generator=g)
    assert torch.equal(bn(x), expect(x))
    key0 = bn._affine_key
    assert torch.equal(bn(x), expect(x)) and bn._affine_key == key0          # second call: served from the cache
    sd = {k: torch.rand(8, generator=g) + 0.5 for k in ("weight", "bias", "running_mean", "running_var")}
    bn.load_state_dict(sd)
    assert torch.equal(bn(x), expect(x)) and bn._affine_key != key0          # load_state_dict copies in place: version bump
    key1 = bn._affine_key
    with torch.no_grad():
        bn.running_var.mul_(2.0)
    assert torch.equal(bn(x), expect(x)) and bn._affine_key != key1
    bn = bn.double()                                                           # new buffer tensors: new data pointers
    assert torch.equal(bn(x.double()), expect(x.double()))


def test_const_tensor_is_cached_per_values_and_device():
    from faster_rcnn_pytorch_amd import ops
    a = ops.const_tensor((1, 2, 3, 4), torch.device("cpu"))
    b = ops.const_tensor([1.0, 2.0, 3.0, 4.0], torch.device("cpu"))
    c = ops.const_tensor((1, 2, 3, 5), torch.device("cpu"))
    assert a is b and a is not c and a.tolist() == [1.0, 2.0, 3.0, 4.0] and a.dtype == torch.float32


def test_backbone_wrappers_are_the_reference_modules_off_the_fp32_device_path():
    """model.VGGExtractor and the FPN's Bottleneck / FeaturePyramidNetwork route layers to the library's conv stage only for fp32 tensors on a HIP
    device; anywhere else (here: the CPU) they must BE the reference's modules: the extractor the plain nn.Sequential of its children (same
    indices, same state-dict keys as `nn.Sequential(*list(vgg16.features)[:-1])`, models/model.py:279-281), the bottleneck torchvision's
    conv-norm-relu chain; and the eligibility predicates must say no without touching the library."""
    import torch
    import torch.nn as nn
    from faster_rcnn_pytorch_amd import ops
    from faster_rcnn_pytorch_amd.model import VGGExtractor, vgg16_features
    from faster_rcnn_pytorch_amd.new_model import Bottleneck, FrozenBatchNorm2d
    torch.manual_seed(0)
    layers = vgg16_features()[:-1]
    ext = VGGExtractor(*layers)
    assert len(ext) == 30 and isinstance(ext[0], nn.Conv2d) and isinstance(ext[4], nn.MaxPool2d) and isinstance(ext[29], nn.ReLU)
    conv_idx = [i for i, m in enumerate(ext) if isinstance(m, nn.Conv2d)]
    assert conv_idx == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]                       # torchvision's vgg16.features indices
    assert sorted(ext.state_dict()) == sorted("%d.%s" % (i, k) for i in conv_idx for k in ("weight", "bias"))
    x = torch.randn(1, 3, 64, 96)
    assert torch.equal(ext(x), nn.Sequential(*layers)(x))
    assert not ops.conv3x3_supported(x, ext[2].weight) and not ops.conv3x3_c3_supported(x, ext[0].weight) and not ops.affine_act_supported(x)
    blk = Bottleneck(64, 16, 1, nn.Sequential(nn.Conv2d(64, 64, 1, bias=False), FrozenBatchNorm2d(64)))
    for bn in (blk.bn1, blk.bn2, blk.bn3, blk.downsample[1]):
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    xb = torch.randn(1, 64, 20, 24)
    out = blk(xb)
    ref = torch.relu(blk.bn1(blk.conv1(xb)))
    ref = torch.relu(blk.bn2(blk.conv2(ref)))
    ref = torch.relu(blk.bn3(blk.conv3(ref)) + blk.downsample(xb))
    assert torch.equal(out, ref)


# ------------------------------------------------------------------------------------------------ parallel.GraphStep (N > 1, graph-submitted step)
class _TinyDet(torch.nn.Module):
    """A stand-in with the detection models' shape: a trunk, an `rpn` branch off the trunk, a pooling module (the cut) and a head behind it."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16))
        self.rpn = torch.nn.Linear(16, 6)
        self.pool = torch.nn.Tanh()
        self.head = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))

    def forward(self, x):
        f = self.trunk(x)
        r = self.rpn(f)
        return (r[:, :2], r[:, 2:], self.head(self.pool(f)))


def _graphstep_worker(r, world, port, q):
    os.environ.update(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from faster_rcnn_pytorch_amd import parallel
    _, _, w, dev = parallel.init_for_distributed(backend="gloo")
    xs = torch.arange(6 * 8, dtype=torch.float32).reshape(6, 8).sin()
    frames = [xs[i:i + 1] for i in parallel.shard_indices(6, r, w)]         # rank r: frames r, r + W, ...

    def losses_of(pred):
        a, b, c = pred
        return (a.pow(2).mean() + b.abs().mean() + c.pow(2).mean(),)

    def run(kind):
        torch.manual_seed(0 if kind != "graphstep" else 100 + r)            # GraphStep must broadcast rank 0's weights itself
        model = _TinyDet()
        opt_of = lambda m: torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.05, momentum=0.9, weight_decay=1e-4)   # noqa: E731
        if kind == "ddp":
            from torch.nn.parallel import DistributedDataParallel as DDP
            net, opt = DDP(model, find_unused_parameters=False), opt_of(model)
            for i in range(3):
                loss = losses_of(net(frames[i % len(frames)]))[0]
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
        else:
            opt = opt_of(model)
            if kind == "graphstep":
                torch.manual_seed(0)
                ref = _TinyDet()                                            # rank 0 starts from the seed-0 values DDP's run started from;
                if r == 0:                                                  # rank 1 from other values, which the broadcast must replace
                    model.load_state_dict(ref.state_dict())

            def forward_loss(f):
                pred = model(frames[f])
                return losses_of(pred), pred
            gs = parallel.GraphStep(model, opt, forward_loss, len(frames), dev, cut_module=model.pool, late_modules=(model.head,), graphs=False)
            assert gs.comm_bytes[0] == 4 * sum(p.numel() for p in model.head.parameters())
            for i in range(3):
                gs.step(i)
        return [p.detach().clone().numpy() for p in model.parameters()]
    a, b = run("ddp"), run("graphstep")
    q.put((r, a, b))
    parallel.shutdown()


def test_graph_step_pieces_equal_ddp_bit_for_bit_over_gloo_world2():
    """The N > 1 form of the graph-submitted step (parallel.GraphStep) with its pieces run eagerly on the CPU: broadcast of rank 0's weights,
    backward cut at the pooling module and at the RPN outputs, head / trunk gradients in two flat buffers, pre-scaled sum over the ranks,
    optimizer on the flat views -- three steps, every parameter equal to DistributedDataParallel's BIT FOR BIT on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 13) % 2000
    procs = [ctx.Process(target=_graphstep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r, a, b in res:
        assert len(a) == len(b) == 10
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    for x, y in zip(res[0][2], res[1][2]):
        assert np.array_equal(x, y)                                         # replicas identical
