#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its config[1]/[2]: training images/sec of VGG16 Faster R-CNN on
synthetic 600x1000 frames, batch 1 per GPU, the proposal / RoI-head path on the hand-written HIP kernels.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward (backbone -> RPN -> proposals -> targets -> RoIPool -> head) + loss + backward + SGD on one
frame per GPU.  Frames and boxes are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
`roofline` describes the dominant hand-written kernel of the hot path, timed live with HIP events on the launch
stream (libfrcnn_hip's frcnn_prof_* facility) inside the timed region; `cpu_baseline` is the oracle's CPU
restatement of the same training step (oracle/model_ref.py) on a bounded number of steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

H, W = 600, 1000                      # BASELINE.json configs[1]
NUM_CLASSES = 21                      # VOC: 20 + background (models/model.py:141)
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9        # wave64 VALU issue: lanes per second, one op each (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def synth_frame(rank, step):
    """x ~ randn(1,3,600,1000); G ~ U{1..8}; centres U(.15,.85)^2, sides U(.08,.6); labels U{0..19}."""
    g = torch.Generator().manual_seed(1000 + rank * 10 ** 6 + step)
    x = torch.randn(1, 3, H, W, generator=g)
    G = int(torch.randint(1, 9, (1,), generator=g))
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    labels = torch.randint(0, 20, (G,), generator=g)
    return x, boxes, labels


def algorithmic_bytes(kernel, N, K, P, R, C, fh, fw, G):
    """Algorithmic HBM bytes per launch (SURVEY 8d / DESIGN.md 'kernels')."""
    nblk = (K + 63) // 64
    return {
        "proposal_prologue_kernel": 44 * N,                               # reg 16N + cls 8N in, boxes 16N + scores 4N out
        "topk_rank_kernel": 4 * N,                                        # scores in (partials are workspace traffic)
        "topk_scatter_kernel": 4 * N + 16 * N + 28 * K,                   # scores + boxes in, idx/score/box out
        "nms_mask_kernel": 16 * K + 8 * K * nblk // 2,                    # boxes in + upper-triangle mask out
        "nms_scan_kernel": 16 * K + 8 * P + 16 * P,                       # compulsory: boxes in, keep + rois out
        "rpn_colmax_kernel": 16 * (N + G),
        "rpn_label_kernel": 16 * (N + G) + 24 * N,                        # anchors + gt in, cls i64 + reg out
        "rpn_sample_kernel": 9 * N,
        "head_targets_kernel": 16 * (P + G) + 44 * R,
        "roi_pool_fwd_kernel": 4 * C * fh * fw + 16 * R + 8 * R * C * 49,  # features + rois in, out + argmax out
        "roi_pool_bwd_kernel": 8 * R * C * 49 + 4 * C * fh * fw,           # grad_out + argmax in, grad_feat out
        "rpn_head_tail_kernel": 4 * C * fh * fw + 4 * 54 * C + 4 * 54 * fh * fw,   # conv output + weights in, cls + reg out
    }.get(kernel)


T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames kept resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket library kernels with HIP events")
    ap.add_argument("--lr", type=float, default=2e-3)          # config.py:24
    ap.add_argument("--miopen-search", action="store_true", help="torch.backends.cudnn.benchmark=True (exhaustive MIOpen find)")
    ap.add_argument("--channels-last", action="store_true", help="run the backbone in NHWC memory format")
    ap.add_argument("--no-fused-sgd", action="store_true")
    ap.add_argument("--amp", default="none", choices=["none", "bf16"],
                    help="autocast the torch layers (backbone / RPN convs / FC head); NOT the default: the reference trains in fp32")
    args = ap.parse_args()

    from faster_rcnn_pytorch_amd import _lib, parallel
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    from faster_rcnn_pytorch_amd.model import FRCNN

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    rank, local_rank, world, device = parallel.init_for_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    torch.backends.cudnn.benchmark = bool(args.miopen_search)

    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)

    torch.manual_seed(0)
    model = FRCNN(num_classes=NUM_CLASSES, sampling="device", seed=1234 + rank).to(device)
    if args.channels_last:
        model.extractor = model.extractor.to(memory_format=torch.channels_last)
    net = parallel.wrap_ddp(model, device)
    crit = FRCNNLoss(None)
    opt = torch.optim.SGD(net.parameters(), lr=args.lr, momentum=0.9, weight_decay=1e-4,        # main.py:55-60
                          fused=not args.no_fused_sgd)   # same update rule, one multi-tensor kernel

    frames = []
    for i in range(args.frames):
        x, b, l = synth_frame(rank, i)
        x = x.to(device)
        if args.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        frames.append((x, b.to(device), l.to(device)))
    torch.cuda.synchronize()

    def step(i):
        x, b, l = frames[i % len(frames)]
        if args.amp == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred, target = net(x, [b], [l])
            pred = tuple(p.float() for p in pred)
        else:
            pred, target = net(x, [b], [l])
        loss = crit(pred, target)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    # Initialisation pass, outside the W / K accounting: the first steps of a process select and compile MIOpen kernels, grow the
    # caching allocator and the library workspaces (seconds, not steady state).  The W warm-up steps below then run warm.
    log("model + %d frames resident; initialisation pass" % len(frames))
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    log("warm-up")
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    if not args.no_kernel_events:
        _lib.prof_reset()
        _lib.prof_enable(True)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    dt = parallel.max_over_ranks(dt, device)
    log("timed region: %d steps in %.3f s" % (args.steps, dt))
    final_loss = float(loss.detach())
    kernels = {} if args.no_kernel_events else _lib.prof_report()

    if rank != 0:
        parallel.shutdown()
        return
    ms_per_step = dt / args.steps * 1e3
    value = world * args.steps / dt
    N = (H // 16) * (W // 16) * 9
    fh, fw = H // 16, W // 16
    shape = dict(N=N, K=12000, P=2000, R=128, C=512, fh=fh, fw=fw, G=8)
    # HBM traffic per launch from the rocprofv3 PMC passes (tools/pmc_traffic.sh; separate FETCH_SIZE / WRITE_SIZE runs,
    # gfx950 fetch correction): measured once per round on the same shapes and committed under profiles/
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hotpath_pmc_traffic.json")) as f:
            pmc = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        pmc = {}

    def pmc_traffic(name):
        stem = name[:-len("_kernel")] if name.endswith("_kernel") else name
        for k, v in pmc.items():
            if k.startswith(stem):
                return v["traffic_bytes"]
        return None
    # the compulsory bytes of NMS / top-k are negligible, so those two are ALSO priced against the fp32 VALU issue
    # peak (256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3 T lane-ops/s): pair IoUs x ~21 VALU ops, rank compares x 2
    valu_ops = {"nms_mask_kernel": (shape["K"] * (shape["K"] - 1) // 2) * 21, "topk_rank_kernel": shape["N"] * shape["N"] * 2}
    per_kernel = {}
    for name, (ms, n) in kernels.items():
        us = ms / n * 1e3
        ab = algorithmic_bytes(name, **shape)
        per_kernel[name] = {"avg_us": round(us, 2), "launches": n, "algorithmic_bytes": ab,
                            "GB_s": round(ab / us * 1e-3, 2) if ab else None, "pmc_traffic_bytes": pmc_traffic(name)}
        if name in valu_ops:
            per_kernel[name]["valu_lane_ops"] = valu_ops[name]
            per_kernel[name]["valu_frac_of_39.3T"] = round(valu_ops[name] / (us * 1e-6) / VALU_PEAK_LANE_OPS, 3)
        if name == "nms_mask_kernel":
            per_kernel[name]["pair_iou_per_s"] = round(shape["K"] * (shape["K"] - 1) / 2 / (us * 1e-6), 0)
    roofline = None
    if per_kernel:
        dom = max(per_kernel, key=lambda k: per_kernel[k]["avg_us"])
        d = per_kernel[dom]
        roofline = {"kernel": dom, "bound": "hbm", "achieved": d["GB_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(d["GB_s"] / HBM_PEAK_GBS, 5) if d["GB_s"] else None, "traffic": pmc_traffic(dom),
                    "traffic_source": "profiles/r01_hotpath_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                    "avg_launch_us": d["avg_us"], "algorithmic_bytes": d["algorithmic_bytes"]}
    hot_us = sum(v["avg_us"] for v in per_kernel.values())

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args.cpu_steps, args.lr)

    out = {
        "metric": "train images/sec (VGG16 Faster R-CNN, 600x1000, bs=1/GPU)", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.amp == "none" else "bf16(torch layers)+f32(hot path)", "data": "synthetic",
        "config": {"workload": "VGG16 Faster R-CNN train step, synthetic 600x1000 frames, bs=1/GPU, HIP proposal/RoI path "
                               "(N=20646 anchors, pre/post NMS 12000/2000, 128 RoIs, RoIPool 7x7 on 512x37x62)",
                   "global_batch": world, "parallelism": "dp%d" % world, "sampling": "device-philox"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "hot_path": {"sum_kernel_us_per_img": round(hot_us, 1), "proposals_per_s": round(value * 2000, 1),
                     # BASELINE.json's second figure: NMS (mask + scan) + RoI pooling forward/backward, HIP-event us per image
                     "nms_plus_roi_us_per_img": round(sum(v["avg_us"] for k, v in per_kernel.items() if k.startswith(("nms_", "roi_"))), 1),
                     # SURVEY 8(d) "proposals/s" unit: one image's proposal stage = prologue -> top-k -> NMS -> P rois
                     "proposal_stage_us_per_img": round(sum(v["avg_us"] for k, v in per_kernel.items()
                                                            if k.startswith(("proposal_prologue", "topk_", "nms_"))), 1),
                     "kernels": per_kernel},
        "final_loss": round(final_loss, 4),
    }
    st = out["hot_path"]["proposal_stage_us_per_img"]
    out["hot_path"]["proposal_stage_proposals_per_s"] = round(2000 / (st * 1e-6), 0) if st else None
    print(json.dumps(out), flush=True)
    parallel.shutdown()


def cpu_baseline(steps, lr):
    """The oracle's CPU restatement of the same training step (kind 'port'), bounded sample."""
    from oracle import oracle as orc
    from oracle.model_ref import RefFRCNN, ref_loss
    orc.build()
    torch.manual_seed(0)
    ref = RefFRCNN(NUM_CLASSES)
    for m in ref.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            torch.nn.init.zeros_(m.bias)
    for m in (ref.rpn.inter_layer, ref.rpn.cls_layer, ref.rpn.reg_layer):
        m.weight.data.normal_(0, 0.01)
        m.bias.data.zero_()
    opt = torch.optim.SGD(ref.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    # the GPU box gives one GPU's share of the host: 16 cores; more threads than that oversubscribes the cgroup
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    print("[bench] cpu baseline on %d threads (affinity %d, cpu_count %s)" % (cores, avail, os.cpu_count()), file=sys.stderr, flush=True)

    def one(i):
        x, b, l = synth_frame(0, i)
        pred, target = ref(x, [b], [l])
        loss = ref_loss(pred, target)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    one(0)                                   # warm-up (allocator, thread pool)
    print("[bench] cpu baseline warm-up step done", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for i in range(steps):
        one(1 + i)
        print("[bench] cpu baseline step %d done (%.1f s)" % (i, time.perf_counter() - t0), file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps (fwd+loss+bwd+SGD) of oracle/model_ref.RefFRCNN on the same synthetic 600x1000 "
                      "frames, torch CPU %d threads + oracle C path, %.1f s" % (steps, cores, dt)}


if __name__ == "__main__":
    main()
