"""The data-parallel step submitted as HIP graphs (parallel.GraphStep; VERDICT r4 "next" 2; models/build.py:8-14 + train.py:31-37):
graph A (forward + loss + the FC head's backward) | all-reduce | graph B (the trunk's backward) | all-reduce | the optimizer's graph.

  * two ranks on the one GPU, gloo standing in for RCCL: after three steps every parameter of every replica equals what EAGER
    DistributedDataParallel leaves behind, bit for bit (same frames, same device-side sampling stream);
  * one process, the ResNet-50-FPN mirror: the cut backward (autograd.grad in two pieces) equals the single loss.backward() bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synth(seed, H, W, G, lo=0, hi=20):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, 3, H, W, generator=g)
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    labels = torch.randint(lo, hi, (G,), generator=g)
    return x, boxes, labels


def _snapshot(model, opt):
    return [p.detach().clone() for p in model.parameters()], [b.detach().clone() for b in model.buffers()]


def _restore(model, opt, snap):
    """In place (the graphs hold the addresses): weights back, momentum buffers to zero (= a fresh optimizer: 0.9 * 0 + g == g exactly)."""
    with torch.no_grad():
        for p, s in zip(model.parameters(), snap[0]):
            p.copy_(s)
        for b, s in zip(model.buffers(), snap[1]):
            b.copy_(s)
        for st in opt.state.values():
            if st.get("momentum_buffer") is not None:
                st["momentum_buffer"].zero_()


def _run(mirror, kind, rank, world, device, steps=3, n_frames=2, H=320, W=480):
    """`steps` training steps of one replica; returns (parameters, losses of the steps).  kind: 'ddp' = eager DistributedDataParallel
    (or the bare model in one process), 'graphstep' = parallel.GraphStep."""
    from faster_rcnn_pytorch_amd import parallel
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    if mirror == "vgg":
        from faster_rcnn_pytorch_amd.model import FRCNN
        nc, lo, hi = 21, 0, 20
    else:
        from faster_rcnn_pytorch_amd.new_model import FRCNN
        nc, lo, hi = 91, 1, 91
    torch.manual_seed(0)                                                       # identical initial weights everywhere
    model = FRCNN(num_classes=nc, sampling="device", seed=10 + rank).to(device)
    crit = FRCNNLoss(None)
    frames = [tuple(t.to(device) for t in synth(50 + rank * 100 + i, H, W, 3, lo, hi)) for i in range(n_frames)]
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, weight_decay=1e-4, fused=True)
    losses = []
    if kind == "ddp":
        net = parallel.wrap_ddp(model, device)
        for i in range(steps):
            x, b, l = frames[i % n_frames]
            pred, target = net(x, [b], [l])
            loss = crit(pred, target)[0]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    else:
        keep = torch.zeros(n_frames, device=device)

        def forward_loss(f):
            x, b, l = frames[f]
            pred, target = model(x, [b], [l])
            return crit(pred, target), pred

        def record(f, ls):
            keep[f:f + 1].copy_(ls[0].detach().reshape(1))
        gs = parallel.GraphStep(model, opt, forward_loss, n_frames, device, record=record, **model.graph_stages())
        snap = _snapshot(model, opt)
        gs.capture()                                                           # its warm-up passes step the weights and the sampling stream:
        _restore(model, opt, snap)                                             # back to the initial state, in place
        model.sampler.reseed(10 + rank, 1)
        assert gs.report()["graphs"] == 2 * n_frames + 1
        for i in range(steps):
            gs.step(i)
            losses.append(float(keep[i % n_frames]))
    torch.cuda.synchronize()
    model.check_device_status()
    return [p.detach().cpu().numpy() for p in model.parameters()], losses


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", 0)
    a = _run("vgg", "ddp", rank, world, device)
    b = _run("vgg", "graphstep", rank, world, device)
    q.put((rank, a, b))
    dist.barrier()
    dist.destroy_process_group()


def test_graph_submitted_ddp_step_equals_eager_ddp_bit_for_bit_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, (pa, la), (pb, lb) in res:
        assert np.isfinite(la).all() and la == lb                               # the same local losses, step by step
        assert len(pa) == len(pb) == 40
        for x, y in zip(pa, pb):
            assert np.array_equal(x, y)                                        # every parameter, bit for bit
    assert res[0][1][1] != res[1][1][1]                                        # different frames per rank ...
    for x, y in zip(res[0][2][0], res[1][2][0]):
        assert np.array_equal(x, y)                                            # ... identical replicas


@pytest.mark.parametrize("mirror", ["vgg", "fpn"])
def test_graph_step_single_process_equals_the_eager_step_bit_for_bit(mirror, monkeypatch):
    """World size 1 (no collective): the cut backward + flat gradient buffers + captured optimizer against the plain eager step."""
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)           # the vendor convolutions of the ResNet body, run to run
    device = torch.device(DEV)
    H, W = (320, 480) if mirror == "vgg" else (320, 448)
    pa, la = _run(mirror, "ddp", 0, 1, device, H=H, W=W)
    pb, lb = _run(mirror, "graphstep", 0, 1, device, H=H, W=W)
    assert np.isfinite(la).all() and la == lb
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)
