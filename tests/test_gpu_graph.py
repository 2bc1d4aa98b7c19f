"""HIP-graph capture of the hot path ("HIP-graph the whole proposal path").

proposal stage -> RPN targets -> head targets -> RoIPool forward -> a small head -> the fused detection loss -> backward
(RoIPool backward, loss gradients) is captured ONCE in a torch.cuda.graph and replayed on new frames.  Nothing on the path
syncs the host, counts stay on the device, and the sampling RNG stream lives in device memory (ops.philox_state): a replay draws
NEW samples, exactly those an eager run draws from the same stream position -- everything is compared bit for bit."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, W, C, G, NC = 600, 1000, 512, 4, 21


def _frame(seed):
    g = torch.Generator().manual_seed(seed)
    fh, fw = H // 16, W // 16
    N = fh * fw * 9
    feat = torch.randn(1, C, fh, fw, generator=g)
    reg = torch.randn(N, 4, generator=g) * torch.tensor([0.1, 0.1, 0.2, 0.2])
    cls = torch.stack([torch.zeros(N), torch.randn(N, generator=g) * 2 - 2], 1)
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    gt = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    lab = torch.randint(0, 20, (G,), generator=g)
    return feat, reg, cls, gt, lab


class _Path(torch.nn.Module):
    """The hand-written part of FRCNN.forward (models/model.py:310-341) with a small linear head standing in for the FC layers."""

    def __init__(self, ops, am):
        super().__init__()
        self.ops, self.am = ops, am
        g = torch.Generator().manual_seed(0)
        self.w_cls = torch.nn.Parameter(torch.randn(C * 49, NC, generator=g) * 0.01)
        self.w_reg = torch.nn.Parameter(torch.randn(C * 49, NC * 4, generator=g) * 0.01)
        self.status = ops.DeviceStatus()

    def forward(self, feat, reg, cls, gt, lab, state):
        ops = self.ops
        anchor = self.am.device_anchors((H, W), feat.device)
        t_rpn_cls, t_rpn_reg, _ = ops.rpn_targets(anchor, gt, philox_state=state)
        rois, cnt, _ = ops.region_proposal(reg.detach(), cls.detach(), None, 1 / 1000, 12000, 0.7, 2000, grid=self.am.grid_desc((H, W)))
        t_cls, t_reg, srois, keep, _ = ops.head_targets(rois, gt, lab, n_rois=cnt, philox_state=state, status=self.status.word(feat.device), want_keep=True)
        fh, fw = feat.shape[2:]
        pool = ops.roi_pool(feat, srois * ops.const_tensor((fw, fh, fw, fh), feat.device), (7, 7), 1.0)
        x = pool.view(128, -1)
        head_cls = x @ self.w_cls
        head_reg = (x @ self.w_reg).reshape(128, -1, 4)
        head_reg = torch.gather(head_reg, 1, t_cls.clamp(min=0).view(-1, 1, 1).expand(-1, 1, 4)).squeeze(1)
        losses = ops.detection_loss((cls.unsqueeze(0), reg.unsqueeze(0), head_cls, head_reg), (t_rpn_cls, t_rpn_reg, t_cls, t_reg))
        return losses, (t_rpn_cls, t_cls, keep, cnt, srois)


def test_hot_path_graph_replay_matches_eager_with_fresh_samples():
    from faster_rcnn_pytorch_amd import ops
    from faster_rcnn_pytorch_amd.anchor import FRCNNAnchorMaker
    path = _Path(ops, FRCNNAnchorMaker()).to(DEV)
    frames = [_frame(31), _frame(32), _frame(31)]                 # frame 2 repeats frame 0: same inputs, later stream position
    static = [t.to(DEV).clone() for t in frames[0]]
    static[0].requires_grad_(True); static[1].requires_grad_(True); static[2].requires_grad_(True)
    state = ops.philox_state(777, 1, DEV)

    def run_once():
        for t in static[:3]:
            t.grad = None
        path.zero_grad(set_to_none=True)
        losses, aux = path(*static, state)
        losses[0].backward()
        return losses, aux

    # warm-up on a side stream (allocators, workspaces, constants), as torch.cuda.graph asks
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run_once()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    # ---- eager pass over the three frames from stream position 100
    def load(f):
        with torch.no_grad():
            for s_t, t in zip(static, f):
                s_t.copy_(t.to(DEV))

    def snapshot(losses, aux):
        return ([float(l.detach()) for l in losses], [a.clone() for a in aux], static[0].grad.clone(), static[1].grad.clone(), static[2].grad.clone(),
                path.w_cls.grad.clone(), path.w_reg.grad.clone())
    state.copy_(ops.philox_state(777, 100, DEV))
    eager = []
    for f in frames:
        load(f)
        eager.append(snapshot(*run_once()))
    torch.cuda.synchronize()
    assert state.cpu().tolist()[1] == 100 + 2 * 3                 # two sampling calls per step

    # ---- capture once, replay on the same three frames from the same stream position
    load(frames[0])
    for t in static[:3]:
        t.grad = None
    path.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_losses, g_aux = path(*static, state)
        g_losses[0].backward()
    state.copy_(ops.philox_state(777, 100, DEV))                  # (the capture itself enqueued nothing)
    replayed = []
    for f in frames:
        load(f)
        graph.replay()
        replayed.append(snapshot(g_losses, g_aux))
    torch.cuda.synchronize()
    assert state.cpu().tolist()[1] == 100 + 2 * 3
    path.status.check()

    for e, r in zip(eager, replayed):
        assert e[0] == r[0]                                                        # the five losses, bit for bit
        for a, b in zip(e[1], r[1]):
            assert torch.equal(a, b)                                               # RPN labels, head labels, sampled RoI ids, count, RoIs
        assert torch.equal(e[3], r[3]) and torch.equal(e[4], r[4])                 # d loss / d RPN outputs (fused loss kernel: fixed-order sums)
        # RoIPool backward accumulates with LDS float atomics (ds_add_f32) whose order inside a workgroup is not fixed: two EAGER runs
        # differ in the last bits just the same.  d/d features and the head-weight gradients (GEMMs downstream of it) within 1e-6 relative.
        for k in (2, 5, 6):
            scale = float(e[k].abs().max())
            assert float((e[k] - r[k]).abs().max()) <= 1e-6 * scale + 1e-12, (k, float((e[k] - r[k]).abs().max()), scale)
    # fresh samples per replay: frames 0 and 2 are identical inputs at different stream positions
    assert torch.equal(replayed[0][1][3], replayed[2][1][3])                       # same proposals ...
    assert not torch.equal(replayed[0][1][2], replayed[2][1][2])                   # ... different RoI samples
    assert not torch.equal(replayed[0][1][0], replayed[2][1][0])                   # and different RPN anchor samples
    # and the replayed proposals are the oracle's
    f = frames[1]
    ro, _ = orc.region_proposal(f[1].numpy(), f[2].numpy(), orc.anchor_grid(H, W), 1 / 1000, 12000, 0.7, 2000)
    assert int(replayed[1][1][3].item()) == len(ro)


def test_scratch_of_a_captured_op_does_not_outlive_its_graph():
    """Two captures in one process, the first graph destroyed (and its pool released) before the second is made -- what `bench.py` does when it times
    one model as graph replays and then another.  An op's scratch buffer (ops._workspace) must come from the CAPTURING graph's pool: cached per stream --
    torch captures every graph on the same capture stream -- the buffer of the first capture stayed in the cache, its pool went away with the first
    graph, and the second graph replayed on a dangling address (a memory fault in round 4).  Both replays are compared with the eager sort."""
    from faster_rcnn_pytorch_amd import ops
    g0 = torch.Generator().manual_seed(5)
    outs = []
    for n in (20000, 9000):                                              # two "models": different sizes, the second's scratch fits in the first's
        scores = torch.randn(n, generator=g0).to(DEV)
        ref_idx = ops.topk_sorted(scores, 2000)[0].clone()         # eager warm-up: every persistent buffer exists before the capture
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.topk_sorted(scores, 2000)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            idx = ops.topk_sorted(scores, 2000)[0]
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        outs.append(bool(torch.equal(idx, ref_idx)))
        del g, idx
        torch.cuda.synchronize()
        torch.cuda.empty_cache()                                         # the first graph's private pool is released here
    assert outs == [True, True]
    key = (torch.device(DEV).index, torch.cuda.graphs.graph.default_capture_stream.cuda_stream) if getattr(torch.cuda.graphs.graph, "default_capture_stream", None) else None
    assert key is None or key not in ops._WS                             # nothing was cached under the capture stream


def test_conv_stage_workspace_grown_under_capture_is_graph_private():
    """the conv stage keeps ticket words + U / V / M in ONE persistent block per device (ops._ctrl_workspace, tag 'rpn_conv_f32'), sized by
    the largest layer seen.  A capture that needs MORE than the eager warm-up left there (a layer or image size first met under capture) must not put a
    buffer of its own private pool into that cache: after the graph is gone, eager calls and later captures would run on a released address.  Two
    captures at growing shapes without an eager warm-up at those shapes, the first graph destroyed before the second is made; then an eager call at the
    larger shape.  Every result equals the eager result computed up front on a small cache... and the cache never holds a capture-time allocation."""
    from faster_rcnn_pytorch_amd import ops
    g0 = torch.Generator().manual_seed(9)
    w = (torch.randn(128, 128, 3, 3, generator=g0) * 0.03).to(DEV)
    b = torch.randn(128, generator=g0).to(DEV)
    xs = [torch.randn(1, 128, h, wd, generator=g0).to(DEV) for h, wd in ((40, 56), (96, 132), (150, 250))]
    ops._CTRL.pop(("rpn_conv_f32", torch.device(DEV).index), None)         # as in a fresh process
    refs = []
    for x in xs:                                                            # eager references, each computed on a cache that is dropped again
        refs.append(ops.conv3x3_fwd([x], w, b, True)[0].clone())
        torch.cuda.synchronize()
        ops._CTRL.pop(("rpn_conv_f32", torch.device(DEV).index), None)
    ops.conv3x3_fwd([xs[0]], w, b, True)                                    # the only warm-up: the SMALLEST shape
    torch.cuda.synchronize()
    cached = ops._CTRL[("rpn_conv_f32", torch.device(DEV).index)]
    n_cached = cached.numel()
    for i in (1, 2):                                                        # growing shapes, each first met under capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = ops.conv3x3_fwd([xs[i]], w, b, True)[0]
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, refs[i])
        assert ops._CTRL[("rpn_conv_f32", torch.device(DEV).index)] is cached and cached.numel() == n_cached      # the cache did not take the graph's buffer
        del g, y
        torch.cuda.synchronize()
        torch.cuda.empty_cache()                                            # the graph's private pool is released here
    assert torch.equal(ops.conv3x3_fwd([xs[2]], w, b, True)[0], refs[2])   # eager at the large shape: grows the cache with an ordinary allocation
    assert ops._CTRL[("rpn_conv_f32", torch.device(DEV).index)].numel() > n_cached
    g = torch.cuda.CUDAGraph()                                              # and a capture that FINDS a large enough cached block uses it
    with torch.cuda.graph(g):
        y = ops.conv3x3_fwd([xs[1]], w, b, True)[0]
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, refs[1])
