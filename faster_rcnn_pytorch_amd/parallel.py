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
        backend = "nccl" if use_cuda else "gloo"
    if use_cuda:
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
