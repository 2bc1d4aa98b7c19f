"""Data-parallel plumbing: one process per GPU, torch.distributed (backend 'nccl' == RCCL on ROCm)
over xGMI; mirrors the reference's utils/__init__.py:5-25 (init) and models/build.py:7-19 (DDP wrap).

The proposal / RoI path shards naturally (one image per GPU, no exchange); the only steady-state
collective is DDP's bucketed gradient all-reduce, overlapped with backward.  Bucket size: the VGG16
gradient is 548 MB fp32 and xGMI is point-to-point (per-link bound ring), so larger buckets than
DDP's 25 MB default amortise the per-collective latency; classifier.0 (411 MB) finishes its backward
first, which lets its reduction overlap the whole conv backward.
"""
import os

import torch
import torch.distributed as dist


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_for_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world, device)."""
    rank, local_rank, world = dist_env()
    use_cuda = torch.cuda.is_available()
    if backend is None:
        # FRCNN_DIST_BACKEND=gloo: rehearsal of the N > 1 code path where RCCL cannot run (several ranks sharing the one GPU of a test box:
        # RCCL refuses two ranks on one device); never set in a measured run -- bench.py prints the backend it used
        backend = os.environ.get("FRCNN_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    if use_cuda:
        if backend == "gloo" and local_rank >= torch.cuda.device_count():
            local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "23456")       # the reference's fixed port (utils/__init__.py:16)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        dist.barrier()
    device = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    return rank, local_rank, world, device


def shutdown():
    """Tear the process group down (quiet exit under torchrun); no-op for a single process."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def shard_indices(n_items, rank, world):
    """DistributedSampler-style partition (new_datasets/build.py:72): rank r takes r, r+W, r+2W, ..."""
    return list(range(rank, n_items, world))


def wrap_ddp(model, device, bucket_cap_mb=100):
    """models/build.py:8-14: DDP, find_unused_parameters=False.  No-op for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [device.index] if device.type == "cuda" else None
    # broadcast_buffers=False: the only buffers of the two mirrors are the FROZEN batch-norm statistics of the ResNet-50-FPN backbone,
    # identical on every rank by construction; broadcasting them in every forward costs a collective per step and, being an in-place
    # copy, would invalidate the cached scale / shift of every FrozenBatchNorm2d (new_model.py) each iteration.
    return DDP(model, device_ids=ids, find_unused_parameters=False, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
               broadcast_buffers=False)


def ddp_report(net):
    """What DDP's reducer was built over, from its own logging record: the number of parameter tensors it hooked (each
    parameter ONCE -- the VGG mirror registers `classifier` twice, as FRCNN.classifier and fast_rcnn_head.classifier, like the
    reference, models/model.py:282,298), their bytes, the bucket cap and the bucket sizes (initial assignment; DDP re-buckets
    in gradient-ready order after the first backward and reports that as rebuilt_bucket_sizes).  None for an unwrapped model."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if not isinstance(net, DDP):
        return None
    d = net._get_ddp_logging_data()

    def sizes(key):
        v = d.get(key)
        return [int(x) for x in str(v).replace(",", " ").split()] if v not in (None, "") else None
    unique = {id(p): p for p in net.module.parameters() if p.requires_grad}
    return {"num_parameter_tensors": int(d.get("num_parameter_tensors", -1)), "total_parameter_size_bytes": int(d.get("total_parameter_size_bytes", -1)),
            "unique_trainable_parameters": len(unique), "unique_trainable_bytes": int(sum(p.numel() * p.element_size() for p in unique.values())),
            "registered_names_with_aliases": len(list(net.module.named_parameters(remove_duplicate=False))),
            "bucket_cap_bytes": int(d.get("bucket_cap_bytes", -1)), "bucket_sizes": sizes("bucket_sizes"),
            "rebuilt_bucket_sizes": sizes("rebuilt_bucket_sizes"), "gradient_as_bucket_view": bool(d.get("gradient_as_bucket_view", 0)),
            "find_unused_parameters": bool(d.get("find_unused_parameters", 0))}


def max_over_ranks(value, device):
    """MAX of a python float over all ranks (bench.py's timing rule)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device):
    """The python float of every rank, as a list indexed by rank (bench.py prints per-rank step times)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def sum_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class GraphStep(object):
    """The data-parallel training step of train.py:31-37 under models/build.py:8-14, submitted as HIP graphs instead of ~400 (VGG16) /
    ~2500 (ResNet-50-FPN) launches enqueued from Python, with the gradient all-reduce issued BETWEEN the replays (RCCL is not captured):

        replay A[f]   forward + loss + the backward of the head (FC layers behind the RoI pooling: 88 % of VGG16's gradient bytes),
                      those gradients written into one flat buffer
        all-reduce    of that buffer, asynchronously on the process group's own stream      -+  these two overlap: the
        replay B[f]   the rest of the backward (RoI pooling, RPN, backbone) -> a second flat buffer  -+  collective hides behind the trunk
        all-reduce    of the second buffer
        replay U      the optimizer step, reading p.grad = views into the two flat buffers

    A[f] / B[f] exist once per resident frame f (the number of ground-truth boxes is a launch argument), U once.  The backward is cut at
    the output of `cut_module` (the RoI pooling) and at the RPN's predictions: A takes d loss / d {those tensors, head parameters} with
    torch.autograd.grad, B continues from them -- the same nodes the single backward() runs, each once, so the gradients are the
    ones DDP computes.  Like DDP's reducer the average is taken by scaling BEFORE the sum (here: the backward is seeded with 1 / world,
    exact for a power-of-two world, so the replicas match eager DDP bit for bit there; within rounding otherwise).
    Parameters are broadcast from rank 0 at construction, as DDP does.

    `forward_loss(f) -> (losses, pred)`: losses[0] is the scalar to minimise, pred the model's prediction tuple (pred[0], pred[1] = the
    RPN's class / box predictions).  `record(f, losses)` runs inside the capture right after the forward (copy what must survive the
    shared memory pool).  With graphs=False every piece runs eagerly (CPU tensors, gloo: the world-size-2 tests) through the same code."""

    def __init__(self, model, optimizer, forward_loss, n_frames, device, cut_module=None, late_modules=(), record=None, graphs=True):
        self.model, self.opt, self.forward_loss, self.n, self.device = model, optimizer, forward_loss, n_frames, device
        self.record = record
        self.graphs = bool(graphs) and device.type == "cuda"
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        params = [p for p in model.parameters() if p.requires_grad]                       # parameters() lists an aliased module's tensors once
        late_ids = {id(p) for m in late_modules for p in m.parameters() if p.requires_grad} if cut_module is not None else set()
        self.late = [p for p in params if id(p) in late_ids]
        self.early = [p for p in params if id(p) not in late_ids]
        self.cut_module = cut_module if self.late else None
        if self.world > 1:
            with torch.no_grad():
                for p in params:
                    dist.broadcast(p.data, 0)
                for b in model.buffers():
                    dist.broadcast(b.data, 0)
        self.flat, self.views = [], {}
        for group in (self.late, self.early):
            if not group:
                self.flat.append(None)
                continue
            assert all(p.dtype == group[0].dtype for p in group), "one dtype per flat buffer"
            buf = torch.zeros(sum(p.numel() for p in group), dtype=group[0].dtype, device=device)
            off = 0
            for p in group:
                self.views[id(p)] = buf[off:off + p.numel()].view_as(p)
                off += p.numel()
            self.flat.append(buf)
        for p in params:
            p.grad = self.views[id(p)]                                                    # for good: the optimizer graph reads these addresses
        self.seed = torch.full((), 1.0 / self.world, dtype=torch.float32, device=device)
        self._cut = []
        self.gA, self.gB, self.gU = [], [], None
        self.comm_bytes = [0 if b is None else b.numel() * b.element_size() for b in self.flat]

    # ---- the pieces (eager or under capture)
    def _stage_a(self, f):
        h = None
        if self.cut_module is not None:
            del self._cut[:]
            h = self.cut_module.register_forward_hook(lambda mod, i, o: self._cut.append(o))
        try:
            losses, pred = self.forward_loss(f)
        finally:
            if h is not None:
                h.remove()
        if self.record is not None:
            self.record(f, losses)
        total = losses[0]
        seed = self.seed.to(total.dtype)
        if self.cut_module is None:
            grads = torch.autograd.grad([total], self.early, [seed], allow_unused=True)
            self._store(self.early, grads)
            return None
        cuts = [t for t in list(self._cut) + [pred[0], pred[1]] if torch.is_tensor(t) and t.requires_grad]
        del self._cut[:]
        grads = torch.autograd.grad([total], cuts + self.late, [seed], retain_graph=True, allow_unused=True)
        self._store(self.late, grads[len(cuts):])
        live = [(t, g) for t, g in zip(cuts, grads[:len(cuts)]) if g is not None]
        return [t for t, _ in live], [g for _, g in live]

    def _stage_b(self, carry):
        roots, seeds = carry
        grads = torch.autograd.grad(roots, self.early, seeds, allow_unused=True)
        self._store(self.early, grads)

    def _store(self, params, grads):
        if any(g is None for g in grads):
            # torch's optimizers skip a parameter whose .grad is None (no weight decay, no momentum); a captured optimizer step cannot, so --
            # like DDP with find_unused_parameters=False (models/build.py:12) -- every trainable parameter must take part in the loss
            raise RuntimeError("GraphStep: %d trainable parameter(s) received no gradient" % sum(g is None for g in grads))
        if params:
            torch._foreach_copy_([self.views[id(p)] for p in params], list(grads))

    def _reduce(self, k):
        if self.world > 1 and self.flat[k] is not None:
            return dist.all_reduce(self.flat[k], op=dist.ReduceOp.SUM, async_op=True)
        return None

    # ---- capture
    def capture(self, warm=1):
        """Eager warm-up passes on a side stream (lazy initialisation, optimizer state), then the graphs.  The warm-up steps UPDATE the
        weights; callers that need the initial weights (tests) restore them in place afterwards."""
        if not self.graphs:
            return self
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                for f in range(self.n):
                    self._eager(f)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = None
        for f in range(self.n):
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, pool=pool):
                carry = self._stage_a(f)
            pool = ga.pool()
            gb = None
            if carry is not None:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, pool=pool):
                    self._stage_b(carry)
            del carry
            self.gA.append(ga)
            self.gB.append(gb)
        self.gU = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gU, pool=pool):
            self.opt.step()
        self.sync_state()
        return self

    def sync_state(self):
        """Rank 0's parameters, buffers and optimizer state (momentum) to every rank, in place: whatever ran before (warm-up passes on each
        rank's own frames) the replicas are identical from here on."""
        if self.world == 1:
            return
        with torch.no_grad():
            ts = [p.data for p in self.late + self.early] + [b.data for b in self.model.buffers()]
            for p in self.late + self.early:
                ts += [v for _, v in sorted(self.opt.state.get(p, {}).items()) if torch.is_tensor(v)]
            for t in ts:
                dist.broadcast(t, 0)

    def _eager(self, f):
        carry = self._stage_a(f)
        w0 = self._reduce(0)
        if carry is not None:
            self._stage_b(carry)
        w1 = self._reduce(1)
        for w in (w0, w1):
            if w is not None:
                w.wait()
        self.opt.step()

    def step(self, i):
        f = i % self.n
        if not self.graphs:
            return self._eager(f)
        self.gA[f].replay()
        w0 = self._reduce(0)                            # on the process group's stream, behind what the replay enqueued
        if self.gB[f] is not None:
            self.gB[f].replay()
        w1 = self._reduce(1)
        for w in (w0, w1):
            if w is not None:
                w.wait()                                # the current stream waits for the collective; the host does not
        self.gU.replay()

    def report(self):
        return {"submission": "HIP graphs: A (forward + loss + head backward) | all-reduce | B (trunk backward) | all-reduce | optimizer" if self.graphs else "eager pieces",
                "world": self.world, "head_gradient_bytes": self.comm_bytes[0], "trunk_gradient_bytes": self.comm_bytes[1],
                "parameter_tensors": len(self.late) + len(self.early), "graphs": (len(self.gA) + sum(g is not None for g in self.gB) + 1) if self.graphs else 0}
