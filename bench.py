#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: training images/sec of Faster R-CNN on synthetic frames, batch 1 per GPU, the
proposal / RoI-head path on the hand-written HIP kernels.

    python bench.py                          # configs[1]: VGG16, 600x1000, 1 GPU (the headline configuration) + an `also` block
                                             #   with short driver-timed runs of configs[3] / configs[4] (ResNet-50-FPN fp32 / bf16)
    python bench.py --config fpn             # configs[3]: ResNet-50-FPN, 800x1344 (COCO shape padded to /32), fp32
    python bench.py --config fpn --amp bf16  # configs[4]: bf16 autocast on the torch layers + bf16 MFMA RPN head, fp32 box path
    python bench.py --config fpn --graph     # the same step captured once per resident frame in a HIP graph and replayed
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward (backbone -> RPN -> proposals -> targets -> RoIPool / RoIAlign -> head) + loss + backward + SGD on one
frame per GPU.  Frames and boxes are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
  roofline     : the hand-written kernel with the largest average launch time in the timed region, timed live with HIP events
                 on its launch stream (libfrcnn_hip's frcnn_prof_* facility; kernels are reported under their own names, the
                 ones rocprofv3 prints); `bound` says what actually bounds it ("latency" for the single-workgroup sequential
                 kernels, "valu" for the pair-IoU kernels whose compulsory bytes are negligible, "hbm" for the streaming ones,
                 "mfma" for the conv head); `hbm_kernel` is the largest HBM-bound kernel beside it.
  cpu_baseline : the oracle's CPU restatement of the same training step (oracle/model_ref.py) on a bounded number of steps.
  hot_path     : per-kernel mean / median / p10 / p90 launch time, bytes, GB/s, PMC traffic; proposals/s from the
                 device-side proposal counts of the timed steps.
  also         : (default VGG run on one GPU only) the same record for short runs of --config fpn and --config fpn --amp bf16, so
                 that BASELINE's "RoIAlign + NMS us/img" and configs[3]/[4] are timed by whoever runs this command.
"""
import argparse
import gc
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# MIOpen's first-call search times every applicable solver, including the naive reference convolutions (seconds per layer at these
# shapes, and never the winner).  Leaving them out only shortens the initialisation pass; the steady state is unchanged.
for _k in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")

# PyTorch TunableOp: the first time a GEMM shape is seen (the initialisation pass below, outside the W / K accounting) every rocBLAS /
# hipBLASLt solution for it is timed and the fastest is kept for the rest of the process (~8 s for the dozen shapes of a step).  The
# default heuristic picks a 64x64 macro-tile for the classifier's 128 x 25088 x 4096 GEMMs that streams the 411 MB weight at 1.46 TB/s
# (282 us, three of them per step); the tuned pick takes 190 us: 14.57 -> 14.29 ms per VGG step.  --no-tunableop (or
# FRCNN_BENCH_TUNABLEOP=0) turns it off; the JSON line says which it was and whether the picks were tuned now or read from a file.
TUNABLEOP = os.environ.get("FRCNN_BENCH_TUNABLEOP", "1") != "0" and "--no-tunableop" not in sys.argv
TUNABLEOP_FILE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "frcnn_bench_tunableop_%d.csv")
TUNABLEOP_CACHED = bool(glob.glob(TUNABLEOP_FILE.replace("%d", "*")))
if TUNABLEOP:
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", TUNABLEOP_FILE)

import torch  # noqa: E402

# MI355X_MICROARCH.md:53-54,473: 256 CUs x 4 SIMDs, each SIMD a 32-lane fp32 datapath (a wave64 v_fma_f32 issues in 2 cycles),
# 2.4 GHz -> 256 * 4 * 32 * 2.4e9 = 78.6 T lane-ops/s (= the 157.3 TFLOP/s fp32 vector peak / 2 flops per FMA).
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s HBM3E spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, MI355X_MICROARCH.md (never the 2:1-sparsity figure)
MFMA_PEAK_F32_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32 / 16x16x4_f32: the fp32 vector rate (MI355X_MICROARCH.md:42,430; 155 measured)

CONFIGS = {
    "vgg": dict(H=600, W=1000, num_classes=21, label_lo=0, label_hi=20,          # labels U{0..19} (model.py:141 adds 1)
                shape=dict(N=20646, K=12000, P=2000, R=128, C=512, G=8, feat_bytes=4 * 512 * 37 * 62, A=9, P_head=37 * 62, img_px=600 * 1000, C1=64),
                metric="train images/sec (VGG16 Faster R-CNN, 600x1000, bs=1/GPU)",
                workload="VGG16 Faster R-CNN train step, synthetic 600x1000 frames, bs=1/GPU, HIP proposal/RoI path "
                         "(N=20646 anchors, pre/post NMS 12000/2000, 128 RoIs, RoIPool 7x7 on 512x37x62)"),
    "fpn": dict(H=800, W=1344, num_classes=91, label_lo=1, label_hi=91,          # raw COCO ids 1..90
                shape=dict(N=268569, K=4000, P=1000, R=512, C=256, G=8,
                           feat_bytes=4 * 256 * (200 * 336 + 100 * 168 + 50 * 84 + 25 * 42), A=3,
                           P_head=200 * 336 + 100 * 168 + 50 * 84 + 25 * 42 + 13 * 21),
                metric="train images/sec (ResNet-50-FPN Faster R-CNN, 800x1344, bs=1/GPU)",
                workload="ResNet-50-FPN Faster R-CNN train step, synthetic 800x1344 frames (COCO 800x1333 padded to /32), bs=1/GPU, "
                         "HIP proposal/RoI path (N=268569 anchors over 5 levels, pre/post NMS 4000/1000, 512 RoIs, "
                         "MultiScaleRoIAlign 7x7 on 256 x {200x336,100x168,50x84,25x42})"),
}

# What bounds each hand-written kernel; keys are the kernels' own names.  Anything not listed: "latency".
BOUND = {"nms_kernel": "valu", "nms_filter_kernel": "valu",
         "roi_pool_fwd_lds_kernel": "hbm", "roi_pool_bwd_lds_kernel": "hbm", "roi_pool_bwd_priv_kernel": "hbm", "roi_pool_fwd_kernel": "hbm", "roi_pool_bwd_kernel": "hbm",
         "roi_align_fwd77_kernel": "hbm", "roi_align_fwd_nhwc_kernel": "hbm", "roi_align_bwd_tile_kernel": "hbm", "roi_align_bwd_nhwc_kernel": "hbm",
         "rpn_conv3x3_head_kernel": "mfma", "rpn_conv3x3_bwd_data_kernel": "mfma",
         "rpn_conv3x3_wgrad_kernel": "mfma", "rpn_conv_pack_kernel": "hbm",
         "rpn_conv3x3_f32_kernel": "mfma", "rpn_conv3x3_f32_bwd_data_kernel": "mfma", "rpn_conv3x3_f32_wgrad_kernel": "mfma",
         "rpn_conv_f32_pack_kernel": "hbm", "rpn_wino_gemm_kernel": "mfma", "rpn_wino_input_kernel": "hbm", "rpn_wino_output_kernel": "hbm",
         "rpn_wino_weight_kernel": "hbm", "rpn_wino_dw_kernel": "hbm", "rpn_wino_gemm_out64_kernel": "hbm",
         "conv3x3_c3_fwd_kernel": "hbm", "conv3x3_c3_wgrad_kernel": "hbm",
         "affine_act_fwd_kernel": "hbm", "affine_act_bwd_kernel": "hbm", "affine_act_fwd_mixed_kernel": "hbm", "affine_act_bwd_mixed_kernel": "hbm"}
F32_MFMA_KERNELS = ("rpn_conv3x3_f32_kernel", "rpn_conv3x3_f32_bwd_data_kernel", "rpn_conv3x3_f32_wgrad_kernel", "rpn_wino_gemm_kernel")
WINO_STAGE = ("rpn_wino_weight_kernel", "rpn_wino_input_kernel", "rpn_wino_gemm_kernel", "rpn_wino_output_kernel",
              "rpn_wino_dw_kernel", "rpn_wino_gemm_out64_kernel")      # forward / data gradient: weight, input, gemm, output (64 -> 64 channels: the
                                                                        # last two as ONE launch, gemm_out64); weight gradient: input x 2, gemm, dw


def synth_frame(cfg, rank, step):
    """x ~ randn(1,3,H,W); G ~ U{1..8}; centres U(.15,.85)^2, sides U(.08,.6); labels uniform over the classes."""
    g = torch.Generator().manual_seed(1000 + rank * 10 ** 6 + step)
    x = torch.randn(1, 3, cfg["H"], cfg["W"], generator=g)
    G = int(torch.randint(1, 9, (1,), generator=g))
    c = torch.rand(G, 2, generator=g) * 0.7 + 0.15
    wh = torch.rand(G, 2, generator=g) * 0.52 + 0.08
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    labels = torch.randint(cfg["label_lo"], cfg["label_hi"], (G,), generator=g)
    return x, boxes, labels


def algorithmic_bytes(kernel, N, K, P, R, C, G, feat_bytes, A, P_head, img_px=0, C1=0):
    """Algorithmic HBM bytes per launch (SURVEY 8d / DESIGN.md 'kernels'); None where the figure would say nothing."""
    nblk = (K + 63) // 64
    pooled = R * C * 49
    return {
        "proposal_prologue_kernel": 44 * N,                               # reg 16N + cls 8N in, boxes 16N + scores 4N out
        "topk_count_kernel": 4 * N, "topk_place_kernel": 4 * N + 8 * N, "topk_partition_kernel": 8 * N + 8 * N,
        "topk_bucket_kernel": 8 * N + 16 * K + 28 * K,                    # placed keys + the K gathered boxes in, idx/score/box out
        "nms_kernel": 16 * K + 8 * K + 8 * 2 * nblk,                      # compulsory: boxes in, a word per box + the two bitmaps (negligible)
        "nms_filter_kernel": 16 * K + 16 * K,                             # boxes in, compacted survivors out
        "nms_emit_kernel": 8 * nblk + 16 * P + 8 * P + 16 * P,            # bitmap + kept boxes in, keep + rois out
        "rpn_colmax_kernel": 16 * (N + G), "rpn_label_kernel": 16 * (N + G) + 24 * N, "rpn_match_kernel": 2 * 16 * (N + G) + 24 * N + 5 * N,   # + label byte and Philox key out
        "rpn_sample_kernel": 9 * N, "rpn_apply_kernel": 5 * N + 8 * 256,       # label + key in, ~256 demotions out (sampler's second half, FPN size)
        "head_targets_kernel": 16 * (P + G) + 44 * R,
        "roi_pool_fwd_lds_kernel": feat_bytes + 16 * R + 6 * pooled,      # features + rois in, out (fp32) + argmax (16-bit, the autograd pair) out
        "roi_pool_bwd_lds_kernel": 6 * pooled + feat_bytes,               # grad_out + 16-bit argmax in, grad_feat out
        "roi_pool_bwd_priv_kernel": 6 * pooled + feat_bytes,              # (the same bytes: the round-5 kernel on wave-private planes)
        "roi_align_fwd77_kernel": 4 * pooled + feat_bytes,                # out + (at most) the four pooled levels in
        "roi_align_fwd_nhwc_kernel": 4 * pooled + feat_bytes,
        "roi_align_bwd_tile_kernel": 4 * pooled + feat_bytes,             # grad_out in + every gradient pixel written once
        "roi_align_bwd_nhwc_kernel": 4 * pooled + feat_bytes,
        "rpn_head_tail_kernel": 4 * C * P_head + 4 * 6 * A * C + 4 * 6 * A * P_head,   # conv output + weights in, cls + reg out
        "rpn_conv_f32_pack_kernel": 2 * 4 * 9 * C * C,                    # the 3x3 weights in, transposed + flipped out
        "conv3x3_c3_fwd_kernel": (4 * 3 + 4 * C1 + 8 * ((C1 + 63) // 64)) * img_px or None,        # image in, activations + sign words out
        "conv3x3_c3_wgrad_kernel": (4 * 3 + 4 * C1 + 8 * ((C1 + 63) // 64)) * img_px or None,      # image + gradient + sign words in
    }.get(kernel)


def algorithmic_flops(kernel, N, K, P, R, C, G, feat_bytes, A, P_head, img_px=0, C1=0):
    """Algorithmic flops per launch of the MFMA-bound kernels (useful positions only: no tile padding, no halo)."""
    conv = 2 * C * 9 * C * P_head                                         # C -> C 3x3 on every RPN position (model.py:68-70, new_model.py:96-98)
    return {"rpn_conv3x3_head_kernel": conv + 2 * C * 6 * A * P_head,     # raw = conv3x3 + both 1x1 heads
            "rpn_conv3x3_bwd_data_kernel": conv, "rpn_conv3x3_wgrad_kernel": conv,
            "rpn_conv3x3_f32_kernel": conv, "rpn_conv3x3_f32_bwd_data_kernel": conv, "rpn_conv3x3_f32_wgrad_kernel": conv}.get(kernel)
    # (rpn_wino_gemm_kernel serves several layers per image: build_record prices it from the traced calls of one step, wino_work)


def wino_tiling(shapes):
    """(m, padded tile total) of a stage call over these (H, W) levels: the library's rule (csrc/rpn_conv_f32.hip wn_padded / wn_pick_m; the
    CPU suite checks this mirror against the library's own answers).  Per level the m x m tiles are padded to the product's tile width -- 128, or 64
    where that shortens the total by 15 % or more --, and m = 4 wherever 36 planes x its total is at least 15 % below 16 planes x the 2 x 2 total."""
    def padded(m):
        t = [(-(-h // m)) * (-(-w // m)) for h, w in shapes]
        t128, t64 = sum(-(-x // 128) * 128 for x in t), sum(-(-x // 64) * 64 for x in t)
        return t64 if t64 * 100 <= t128 * 85 else t128
    m = 4 if 36 * padded(4) * 100 <= 16 * padded(2) * 85 else 2
    return m, padded(m)


def wino_work(calls):
    """Per-image algorithmic totals of the fp32 Winograd stage's kernels from the calls ONE training step makes (ops.CONV_TRACE: the RPN
    convolution and every backbone layer the stage takes, each forward / data gradient / weight gradient): kernel -> launches, bytes,
    flops; plus the flop count of the convolutions served (18 Cin Cout per position and direction).  Tiles are padded per level (wino_tiling:
    that padding is executed, so the GEMM is priced on it); transforms move the activations once and the 16 planes once."""
    tot = {k: {"launches": 0, "bytes": 0, "flops": 0} for k in WINO_STAGE}
    conv_flops = 0

    def add(k, b=0, f=0):
        tot[k]["launches"] += 1; tot[k]["bytes"] += b; tot[k]["flops"] += f
    for c in calls:
        if c["kind"] == "gemm_nt":                                          # a 1 x 1 convolution's weight gradient on the stage's GEMM: dW = dY . X^T
            add("rpn_wino_gemm_kernel", 0, 2 * c["M"] * c["N"] * c["K"])
            conv_flops += 2 * c["M"] * c["N"] * c["K"]
            continue
        Cin, Cout = c["Cin"], c["Cout"]
        HW = sum(h * w for h, w in c["shapes"])
        m, Tp = wino_tiling(c["shapes"])                                # the library's choice of the output tile and its padded tile total
        P = (m + 2) ** 2
        bits = 2 * Cout * Tp if c.get("mask") else 0                    # the ReLU's sign words: one uint16 per (channel, tile)
        HWo = sum((h // 2) * (w // 2) for h, w in c["shapes"]) if c.get("pooled") else HW     # fused max-pool: outputs / incoming gradients at the pooled size
        conv_flops += 18 * Cin * Cout * HW
        # 64 -> 64 channels on 4 x 4 tiles, forward and data gradient: the product and the output transform are one launch (csrc: rpn_wino_gemm_out64_kernel;
        # it reads U and V and writes the outputs: the product planes do not exist)
        fused = m == 4 and Cin == 64 and Cout == 64 and c["kind"] in ("fwd", "bwd_data")
        if not fused:
            add("rpn_wino_gemm_kernel", 0, 2 * P * Cin * Cout * Tp)
        if c["kind"] == "wgrad":
            if not c.get("cached"):
                add("rpn_wino_input_kernel", 4 * Cin * HW + 4 * P * Cin * Tp)             # B^T d B of the activations, unless the forward kept it
            if not c.get("urot"):                                           # (`urot` on a weight-gradient call: the data gradient's launch already made it)
                add("rpn_wino_input_kernel", 4 * Cout * HWo + bits + 4 * P * Cout * Tp)   # A g A^T of the (masked, possibly pooled) output gradient
            add("rpn_wino_dw_kernel", 4 * (P + 9) * Cin * Cout)                           # (the bias gradient rides in the output gradient's transform)
        else:
            K, M = (Cin, Cout) if c["kind"] == "fwd" else (Cout, Cin)
            if c["kind"] == "fwd":                                          # with `urot` the forward's launch also writes the data gradient's transform,
                add("rpn_wino_weight_kernel", 4 * ((2 * P if c.get("urot") else P) + 9) * Cin * Cout)
            elif not c.get("urot"):                                         # and the data gradient has no weight launch of its own
                add("rpn_wino_weight_kernel", 4 * (P + 9) * Cin * Cout)
            add("rpn_wino_input_kernel", 4 * K * (HWo if c["kind"] == "bwd_data" else HW) + (bits if c["kind"] == "bwd_data" else 0)
                + 4 * P * K * Tp * (2 if c["kind"] == "bwd_data" and c.get("cached") else 1))       # (`cached` on a data gradient: both transforms in one pass)
            out_bytes = 4 * M * (HWo if c["kind"] == "fwd" else HW) + (2 * M * Tp if c["kind"] == "fwd" and c.get("relu_bits") else 0)
            if fused:
                add("rpn_wino_gemm_out64_kernel", 4 * P * K * M + 4 * P * K * Tp + out_bytes, 2 * P * Cin * Cout * Tp)
            else:
                add("rpn_wino_output_kernel", 4 * P * M * Tp + out_bytes)
    return tot, conv_flops


def percentiles(v):
    v = sorted(v)
    n = len(v)
    if n == 0:
        return None, None, None
    q = lambda f: v[min(n - 1, max(0, int(round(f * (n - 1)))))]      # noqa: E731
    return q(0.5), q(0.1), q(0.9)


T0 = time.perf_counter()


def load_pmc(config, amp):
    """HBM traffic per launch from the rocprofv3 PMC passes over THIS command (tools/pmc_traffic.sh: separate FETCH_SIZE /
    WRITE_SIZE runs of bench.py itself, gfx950 fetch correction); measured per round and committed under profiles/.  Keys are
    kernel names; lookup is by EXACT name (a kernel the profile does not list has traffic null)."""
    pmc_cfg = "fpn_bf16" if (config == "fpn" and amp == "bf16") else config
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % pmc_cfg)))
    if not cands:
        return {}, None
    try:
        with open(cands[-1]) as f:
            return json.load(f)["kernels"], os.path.relpath(cands[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return {}, None


def timed_region(step, steps, warmup, device, armed=None, before_step=None):
    """The bench contract's timing rule, for any device (the world-size-2 gloo test runs it on the CPU): W untimed warm-up steps, then
    EXACTLY K steps bracketed by a barrier + device synchronisation on both sides; the job's time is the MAX over ranks.  Returns
    {dt (max over ranks, s), dt_local, per_rank_ms (ms per step of every rank, indexed by rank), step_ms (rank-local, from device
    events at the step boundaries; [] on the CPU), last (what the last step returned)}."""
    from faster_rcnn_pytorch_amd import parallel
    cuda = device.type == "cuda"
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    for i in range(warmup):
        step(i)
        sync()
    if armed:
        armed()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if cuda else None   # step boundaries on the main stream (no sync)
    parallel.barrier()
    sync()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        if marks:
            marks[i].record()
        if before_step:
            before_step(i)
        last = step(warmup + i)
    if marks:
        marks[steps].record()
    sync()
    parallel.barrier()
    sync()
    dt_local = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt_local, device)
    per_rank_ms = [round(v / steps * 1e3, 3) for v in parallel.gather_over_ranks(dt_local, device)]
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)] if marks else []
    return {"dt": dt, "dt_local": dt_local, "per_rank_ms": per_rank_ms, "step_ms": step_ms, "last": last}


def run_config(args, config, amp, steps, warmup, graph, rank, world, device, with_cpu):
    """One bench record (the JSON object described in the module docstring) for one configuration."""
    from faster_rcnn_pytorch_amd import _lib, ops, parallel
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    cfg = CONFIGS[config]

    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s%s: %s" % (time.perf_counter() - T0, config, "+" + amp if amp != "none" else "", msg), file=sys.stderr, flush=True)

    torch.manual_seed(0)
    if config == "vgg":
        from faster_rcnn_pytorch_amd.model import FRCNN
    else:
        from faster_rcnn_pytorch_amd.new_model import FRCNN
    model = FRCNN(num_classes=cfg["num_classes"], sampling="device", seed=1234 + rank).to(device)
    if args.channels_last:
        bb = "extractor" if config == "vgg" else "backbone"
        setattr(model, bb, getattr(model, bb).to(memory_format=torch.channels_last))
    # (--graph at N > 1: parallel.GraphStep owns the gradient exchange; DDP's reducer hooks are not captured, so the model stays bare)
    net = model if (graph and world > 1) else parallel.wrap_ddp(model, device)
    crit = FRCNNLoss(None)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=args.lr, momentum=0.9, weight_decay=1e-4,   # main.py:55-60
                          fused=not args.no_fused_sgd)   # same update rule, one multi-tensor kernel

    frames = []
    for i in range(args.frames):
        x, b, l = synth_frame(cfg, rank, i)
        x = x.to(device)
        if args.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        frames.append((x, b.to(device), l.to(device)))
    torch.cuda.synchronize()
    counts = []                                       # device-side proposal counts of the timed steps (read after the timed region)

    def body(fi):
        x, b, l = frames[fi]
        if amp == "bf16":
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

    def eager_step(i):
        loss = body(i % len(frames))
        counts.append(model.last_proposal_count)
        return loss

    # Initialisation pass, outside the W / K accounting: the first steps of a process select and compile MIOpen kernels, grow the
    # caching allocator and the library workspaces (seconds, not steady state).  The W warm-up steps below then run warm.
    log("model + %d frames resident; initialisation pass" % len(frames))
    for i in range(3):
        eager_step(i)
    torch.cuda.synchronize()
    ops.CONV_TRACE = []                              # which calls the fp32 conv stage gets in one step (shapes only; for the roofline's flop / byte totals)
    ops.AFFINE_TRACE = {}                            # likewise the norm / residual / ReLU passes: launches and the bytes each must move
    eager_step(3)
    conv_calls, ops.CONV_TRACE = ops.CONV_TRACE, None
    affine_calls, ops.AFFINE_TRACE = ops.AFFINE_TRACE, None
    torch.cuda.synchronize()
    step = eager_step
    graphs = gstep = None
    if graph and world > 1:
        # N > 1: the step as HIP graphs with the gradient all-reduce issued between the replays (parallel.GraphStep): graph A = forward + loss +
        # the FC head's backward, its all-reduce under graph B = the trunk's backward, second all-reduce, the optimizer's graph.  `net` is the
        # bare model here (GraphStep broadcasts rank 0's weights and owns the gradient buffers; DDP's hooks are not captured).
        log("capturing the data-parallel step graphs (%d frames)" % len(frames))
        keep_loss = torch.zeros((len(frames),), dtype=torch.float32, device=device)
        keep_cnt = torch.zeros((len(frames),), dtype=torch.int32, device=device)

        def forward_loss(fi):
            x, b, l = frames[fi]
            if amp == "bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    pred, target = model(x, [b], [l])
                pred = tuple(p.float() for p in pred)
            else:
                pred, target = model(x, [b], [l])
            return crit(pred, target), pred

        def record(fi, losses):
            keep_loss[fi:fi + 1].copy_(losses[0].detach().reshape(1))
            keep_cnt[fi:fi + 1].copy_(model.last_proposal_count)
        gstep = parallel.GraphStep(model, opt, forward_loss, len(frames), device, record=record, **model.graph_stages()).capture()
        graphs = [(None, keep_loss[fi], keep_cnt[fi:fi + 1]) for fi in range(len(frames))]

        def graph_step(i):
            gstep.step(i)
            return graphs[i % len(graphs)][1]
        step = graph_step
    elif graph:
        # One HIP graph per resident frame (the number of ground-truth boxes, a launch argument, differs per frame), all in one
        # memory pool: forward + loss + backward + SGD captured once, replayed as a single submission.  Nothing on the path syncs
        # the host; the sampling RNG stream lives in device memory (ops.philox_state), so every replay draws fresh samples.
        log("capturing %d step graphs" % len(frames))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(len(frames)):
                body(i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs, pool = [], None
        # the graphs share one memory pool, so a graph's outputs live only until the next graph replays: the two scalars read after
        # the timed region (loss, proposal count) are copied to persistent tensors INSIDE the captured step
        keep_loss = torch.zeros((len(frames),), dtype=torch.float32, device=device)
        keep_cnt = torch.zeros((len(frames),), dtype=torch.int32, device=device)
        for fi in range(len(frames)):
            g = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g, pool=pool):
                loss = body(fi)
                keep_loss[fi:fi + 1].copy_(loss.detach().reshape(1))
                keep_cnt[fi:fi + 1].copy_(model.last_proposal_count)
            pool = g.pool()
            graphs.append((g, keep_loss[fi], keep_cnt[fi:fi + 1]))

        def graph_step(i):
            g, loss, cnt = graphs[i % len(graphs)]
            g.replay()
            return loss
        step = graph_step
    # The FPN step enqueues ~2500 launches from Python; a generation-2 garbage collection in the middle of a step stalls the
    # enqueue for 70-100 ms (3 of 60 timed steps read 90-115 ms against a 19.4 ms median).  Everything allocated so far is
    # long-lived: freeze it and keep the collector off while steps are timed (reference counting still frees the step's tensors).
    if not args.gc_on:
        gc.collect(); gc.freeze(); gc.disable()
    events = not args.no_kernel_events and not graph
    ms0 = {}

    def armed():                                      # between the warm-up and the timed region
        log("warm-up done (%d steps)" % warmup)
        if events:
            _lib.prof_reset()
        del counts[:]
        ms0.update(torch.cuda.memory_stats(device))

    def before(i):
        if events:
            _lib.prof_enable(i % args.event_every == 0)    # the HIP-event brackets cost ~0.2 ms per step: sample every n-th timed step
    log("warm-up")
    tr = timed_region(step, steps, warmup, device, armed=armed, before_step=before)
    _lib.prof_enable(False)
    dt, per_rank_ms, step_ms, loss = tr["dt"], tr["per_rank_ms"], tr["step_ms"], tr["last"]
    log("timed region: %d steps in %.3f s" % (steps, dt))
    gc.enable()
    ms1 = torch.cuda.memory_stats(device)
    allocator = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")}
    final_loss = float(loss.detach())
    model.check_device_status()                      # sticky device-side error word (aborted scan / short sample): raises if set
    n_sampled = len(range(0, steps, args.event_every))               # timed steps whose kernels were bracketed
    if graph and not args.no_kernel_events:
        # a captured graph cannot carry the per-kernel event brackets: the live kernel times of --graph come from a short EAGER pass
        # of the same steps right after the timed region (same process, same frames, warm)
        _lib.prof_reset()
        _lib.prof_enable(True)
        n_sampled = min(8, steps)
        for i in range(n_sampled):
            eager_step(i)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
    if graphs:
        counts[:] = [g[2] for g in graphs]                           # persistent copies written inside the captured steps
    samples = {} if args.no_kernel_events else _lib.prof_samples()
    n_props = [int(c.item()) for c in counts if c is not None]       # device counts, read after the timed region
    backend = torch.distributed.get_backend() if world > 1 else None
    world_seen = torch.distributed.get_world_size() if world > 1 else 1
    ddp = gstep.report() if gstep is not None else parallel.ddp_report(net)

    if rank != 0:
        return None
    pmc, pmc_src = load_pmc(config, amp)
    cpu = cpu_baseline(config, cfg, args.cpu_steps, args.lr) if with_cpu else None
    return build_record(config, amp, world=world, steps=steps, warmup=warmup, dt=dt, per_rank_ms=per_rank_ms, step_ms=step_ms,
                        samples=samples, n_sampled=n_sampled, n_props=n_props, graph=graph, pmc=pmc, pmc_src=pmc_src, cpu=cpu,
                        allocator=allocator, ddp=ddp, backend=backend, world_seen=world_seen, final_loss=final_loss, gc_on=args.gc_on, conv_calls=conv_calls,
                        affine_calls=affine_calls)


def build_record(config, amp, *, world, steps, warmup, dt, per_rank_ms, step_ms, samples, n_sampled, n_props, graph, pmc, pmc_src, cpu,
                 allocator, ddp, backend, world_seen, final_loss, gc_on=False, conv_calls=None, affine_calls=None):
    """The FULL bench record (written to bench_detail.json) from plain measured values: no GPU, no torch.  `samples` = kernel name ->
    list of launch times in ms (frcnn_prof_samples), `dt` = max-over-ranks seconds of the timed region, `n_props` = device-side
    proposal counts of the timed steps.  compact_record() cuts it down to the line the driver parses."""
    cfg = CONFIGS[config]
    mean_props = sum(n_props) / len(n_props) if n_props else None
    ms_per_step = dt / steps * 1e3
    value = world * steps / dt
    shape = cfg["shape"]

    def pmc_traffic(name):
        return pmc[name]["traffic_bytes"] if name in pmc else None
    # the compulsory bytes of NMS are negligible, so it is ALSO priced against the fp32 VALU issue peak: pair IoUs x 16
    # VALU ops (counted in the ISA of nms.hip's pair loop).  The pair count is the FULL K (K - 1) / 2 of torchvision's mask
    # kernel (the cascade of nms.hip only runs above 16 384 boxes, i.e. never in a training step).  16 = the checked form's count; the
    # packed path the bulk of the tiles takes ISSUES 11 per pair: `frac_on_issued_instructions` prices the kernel on those.
    valu_ops = {"nms_kernel": (shape["K"] * (shape["K"] - 1) // 2) * 16}
    per_kernel = {}
    for name, v in samples.items():
        n = len(v)
        us = sum(v) / n * 1e3
        med, p10, p90 = (x * 1e3 for x in percentiles(v))
        ab = algorithmic_bytes(name, **shape)
        per_kernel[name] = {"bound": BOUND.get(name, "latency"), "avg_us": round(us, 2), "median_us": round(med, 2), "p10_us": round(p10, 2),
                            "p90_us": round(p90, 2), "launches": n, "launches_per_img": round(n / max(n_sampled, 1), 2),
                            "us_per_img": round(sum(v) * 1e3 / max(n_sampled, 1), 2),
                            "algorithmic_bytes": ab, "GB_s": round(ab / us * 1e-3, 2) if ab else None,
                            "hbm_frac": round(ab / us * 1e-3 / HBM_PEAK_GBS, 5) if ab else None, "pmc_traffic_bytes": pmc_traffic(name)}
        af = algorithmic_flops(name, **shape)
        if af:
            per_kernel[name]["algorithmic_flops"] = af
            per_kernel[name]["TFLOP_s"] = round(af / us * 1e-6, 2)
            per_kernel[name]["mfma_peak_TFLOP_s"] = MFMA_PEAK_F32_TFLOPS if name in F32_MFMA_KERNELS else MFMA_PEAK_BF16_TFLOPS
            per_kernel[name]["mfma_frac"] = round(af / us * 1e-6 / per_kernel[name]["mfma_peak_TFLOP_s"], 5)
    # Winograd convolution = a stage of four launches per call (forward / data gradient: weight and input transforms, the GEMM, the output
    # transform; weight gradient: two transposed transforms, the GEMM, G^T dU G): the convolution's flop count is priced over the time of
    # ALL the stage's launches per call, reported under the GEMM's name; the GEMM's own MFMA utilisation beside it
    if "rpn_wino_gemm_kernel" in per_kernel and conv_calls:
        # the stage serves several layers per image (round 4: the backbone's 3x3 convolutions too), so every figure is a PER-IMAGE total
        # from the traced calls of one step over the per-image time of the kernel's launches; "per launch" = that total / launches
        tot, conv_flops = wino_work(conv_calls)
        matched = True
        for k, t in tot.items():
            if k not in per_kernel:
                continue
            d = per_kernel[k]
            if t["launches"] == 0 or abs(d["launches_per_img"] - t["launches"]) > 0.01:
                matched = False                                      # the sampled steps did not make the traced calls: no figure rather than a wrong one
                d["algorithmic_bytes"] = d["GB_s"] = d["hbm_frac"] = None
                continue
            if t["bytes"]:
                d["algorithmic_bytes"] = round(t["bytes"] / t["launches"])
                d["algorithmic_bytes_per_img"] = t["bytes"]
                d["GB_s"] = round(t["bytes"] / d["us_per_img"] * 1e-3, 2)
                d["hbm_frac"] = round(t["bytes"] / d["us_per_img"] * 1e-3 / HBM_PEAK_GBS, 5)
        d = per_kernel["rpn_wino_gemm_kernel"]
        stage_us = sum(per_kernel[k]["us_per_img"] for k in tot if k in per_kernel)
        own = tot["rpn_wino_gemm_kernel"]["flops"]
        d["algorithmic_flops_per_img"] = own
        d["algorithmic_flops"] = round(own / max(tot["rpn_wino_gemm_kernel"]["launches"], 1))
        d["mfma_peak_TFLOP_s"] = MFMA_PEAK_F32_TFLOPS
        d["TFLOP_s"] = round(own / d["us_per_img"] * 1e-6, 2) if matched else None
        d["mfma_frac"] = round(own / d["us_per_img"] * 1e-6 / MFMA_PEAK_F32_TFLOPS, 5) if matched else None
        d["stage_calls_per_img"] = len(conv_calls)
        d["stage_us_per_img"] = round(stage_us, 2)
        d["stage_us_per_call"] = round(stage_us / len(conv_calls), 2)
        d["stage_kernels"] = [k for k in tot if k in per_kernel]
        d["conv_flops_per_img"] = conv_flops
        d["conv_equivalent_TFLOP_s"] = round(conv_flops / stage_us * 1e-6, 2) if matched else None
        d["layers"] = sorted({"%dx%d %s" % (c["Cin"], c["Cout"], "+".join("%dx%d" % s for s in c["shapes"])) if c["kind"] != "gemm_nt"
                              else "1x1 wgrad %dx%dx%d" % (c["M"], c["N"], c["K"]) for c in conv_calls})
        d["note"] = ("Winograd F(2x2,3x3), all launches of one image (the RPN convolution and the backbone layers in `layers`, forward + data gradient "
                     "+ weight gradient): the GEMM is priced on the flops it executes (32 Cin Cout per padded 2x2 tile, 2.25x fewer than the "
                     "convolutions it serves); conv_equivalent_TFLOP_s = the convolutions' own flop count over the time of ALL the stage's launches")
    # the norm / residual / ReLU passes (FPN configs) run on maps of many sizes: per-image byte totals from the traced step over the per-image time
    for k, (launches, nbytes) in (affine_calls or {}).items():
        if k in per_kernel and launches and abs(per_kernel[k]["launches_per_img"] - launches) < 0.01:
            d = per_kernel[k]
            d["algorithmic_bytes"] = round(nbytes / launches)
            d["algorithmic_bytes_per_img"] = nbytes
            d["GB_s"] = round(nbytes / d["us_per_img"] * 1e-3, 2)
            d["hbm_frac"] = round(nbytes / d["us_per_img"] * 1e-3 / HBM_PEAK_GBS, 5)
    # the NMS stage is several launches of nms_kernel (+ filter / emit): its VALU figure is priced on the stage's time per image
    nms_us = sum(v["us_per_img"] for k, v in per_kernel.items() if k.startswith("nms_"))
    if "nms_kernel" in per_kernel and nms_us > 0:
        d = per_kernel["nms_kernel"]
        d["valu_lane_ops_full_pair_count"] = valu_ops["nms_kernel"]
        d["nms_stage_us_per_img"] = round(nms_us, 2)
        d["valu_frac_of_78.6T"] = round(valu_ops["nms_kernel"] / (nms_us * 1e-6) / VALU_PEAK_LANE_OPS, 3)
        d["pair_iou_per_s"] = round(shape["K"] * (shape["K"] - 1) / 2 / (nms_us * 1e-6), 0)
        d["valu_frac_on_issued_11_per_pair"] = round(valu_ops["nms_kernel"] / 16 * 11 / (nms_us * 1e-6) / VALU_PEAK_LANE_OPS, 3)

    def roofline_of(name):
        d = per_kernel[name]
        if d["bound"] == "valu" and d.get("valu_lane_ops_full_pair_count"):
            # priced against the fp32 VALU issue peak: its compulsory HBM bytes are negligible (the GB/s figure is kept beside it)
            return {"kernel": name, "bound": "valu", "achieved": round(d["valu_lane_ops_full_pair_count"] / (d["nms_stage_us_per_img"] * 1e-6) * 1e-12, 3),
                    "peak": round(VALU_PEAK_LANE_OPS * 1e-12, 1), "unit": "T lane-ops/s", "frac": d["valu_frac_of_78.6T"],
                    "frac_on_issued_instructions": d["valu_frac_on_issued_11_per_pair"],
                    "traffic": d["pmc_traffic_bytes"], "avg_launch_us": d["avg_us"], "median_launch_us": d["median_us"],
                    "nms_stage_us_per_img": d["nms_stage_us_per_img"], "valu_lane_ops": d["valu_lane_ops_full_pair_count"],
                    "algorithmic_bytes": d["algorithmic_bytes"], "GB_s": d["GB_s"], "hbm_frac": d["hbm_frac"],
                    "note": "all NMS launches of one image over the full K(K-1)/2 pair count x 16 VALU (the checked form; 11 issued on the packed path)"}
        if d["bound"] == "mfma" and d.get("algorithmic_flops"):
            r = {"kernel": name, "bound": "mfma", "achieved": d["TFLOP_s"], "peak": d["mfma_peak_TFLOP_s"], "unit": "TFLOP/s",
                 "frac": d["mfma_frac"], "traffic": d["pmc_traffic_bytes"], "avg_launch_us": d["avg_us"],
                 "median_launch_us": d["median_us"], "algorithmic_flops": d["algorithmic_flops"]}
            for k in ("stage_us_per_call", "stage_us_per_img", "stage_calls_per_img", "stage_kernels", "conv_equivalent_TFLOP_s", "conv_flops_per_call",
                      "conv_flops_per_img", "algorithmic_flops_per_img", "layers", "note"):
                if k in d:
                    r[k] = d[k]
            return r
        return {"kernel": name, "bound": d["bound"], "achieved": d["GB_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": d["hbm_frac"], "traffic": d["pmc_traffic_bytes"], "avg_launch_us": d["avg_us"],
                "median_launch_us": d["median_us"], "algorithmic_bytes": d["algorithmic_bytes"]}
    roofline = None
    if per_kernel:
        dom = max(per_kernel, key=lambda k: per_kernel[k]["us_per_img"])
        roofline = roofline_of(dom)
        roofline["traffic_source"] = (pmc_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py, separate passes)") if pmc_src else None
        if roofline["bound"] not in ("hbm", "mfma", "valu"):
            roofline["note"] = ("this kernel is %s-bound: its compulsory HBM bytes are negligible, so the HBM fraction says nothing about "
                                "its quality; see hbm_kernel for the largest HBM-bound kernel" % roofline["bound"])
        hbm = [k for k in per_kernel if per_kernel[k]["bound"] == "hbm" and per_kernel[k]["algorithmic_bytes"]]
        if hbm:
            roofline["hbm_kernel"] = roofline_of(max(hbm, key=lambda k: per_kernel[k]["us_per_img"]))
        if graph:
            roofline["measured_in"] = "an eager pass of %d steps after the timed graph replays (a captured graph cannot carry the event brackets)" % n_sampled
    hot_us = sum(v["us_per_img"] for v in per_kernel.values())
    smed, sp10, sp90 = percentiles(step_ms)
    st = sum(v["us_per_img"] for k, v in per_kernel.items() if k.startswith(("proposal_prologue", "topk_", "nms_", "level_ids")))
    record = {
        "metric": cfg["metric"], "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if amp == "none" else "bf16(torch layers, RPN head MFMA)+f32(box path)", "data": "synthetic",
        "config": {"workload": cfg["workload"], "global_batch": world, "parallelism": "dp%d" % world, "sampling": "device-philox",
                   "submission": ("HIP graphs per resident frame (forward + head backward | all-reduce | trunk backward | all-reduce | optimizer), replayed" if (graph and world > 1)
                                  else "one HIP graph per resident frame, replayed" if graph else "eager (one launch per kernel)")},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "conv_calls": conv_calls,           # every call of the fp32 conv stage in one step (ops.CONV_TRACE): what wino_work() prices
        "step_ms": ({"median": round(smed, 3), "p10": round(sp10, 3), "p90": round(sp90, 3), "max": round(max(step_ms), 3),
                     "slowest_steps": sorted(range(len(step_ms)), key=lambda i: -step_ms[i])[:3],
                     "source": "HIP events at the step boundaries on the main stream (rank 0)"} if step_ms else None),
        "allocator_in_timed_region": allocator,
        "conditions": {
            "gemm_selection": ("PyTorch TunableOp: rocBLAS / hipBLASLt solution timed and picked per GEMM shape during the initialisation pass"
                               if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else "library default heuristics"),
            "tunableop_results": (("read from a results file of an earlier run + tuned now for new shapes" if TUNABLEOP_CACHED else "tuned in this process")
                                  if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else None),
            "python_gc": "on" if gc_on else "frozen after initialisation, off while the timed steps run (--gc-on to leave it on)",
            "off_switches": "--no-tunableop --gc-on give the untuned, collector-on figure"},
        "distributed": {"backend": backend, "world_size_seen_by_rank0": world_seen, "per_rank_ms_per_step": per_rank_ms,
                        "min_ms": min(per_rank_ms), "max_ms": max(per_rank_ms), "ddp": ddp},
        "hot_path": {"sum_kernel_us_per_img": round(hot_us, 1),
                     "launches_per_img": round(sum(v["launches_per_img"] for v in per_kernel.values()), 1),
                     # proposals actually produced: mean of the device-side counts of the timed steps (capacity P is an upper bound)
                     "mean_proposals_per_img": round(mean_props, 1) if mean_props is not None else None,
                     "proposals_per_s": round(value * mean_props, 1) if mean_props is not None else None,
                     # BASELINE.json's second figure: NMS + RoI pooling forward/backward, HIP-event us per image
                     "nms_plus_roi_us_per_img": round(sum(v["us_per_img"] for k, v in per_kernel.items() if k.startswith(("nms_", "roi_"))), 1),
                     # SURVEY 8(d) "proposals/s" unit: one image's proposal stage = prologue -> top-k -> NMS -> P rois
                     "proposal_stage_us_per_img": round(st, 1),
                     "proposal_stage_proposals_per_s": round(mean_props / (st * 1e-6), 0) if (st and mean_props is not None) else None,
                     "kernels": per_kernel},
        "final_loss": round(final_loss, 4), "device_status": 0,
    }
    return record


COMPACT_LIMIT = 4096          # the driver keeps a ~10 KB tail of stdout: the line it parses stays well below that (tests/test_bench_record.py)


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d} if d is not None else None


def compact_record(full, also=()):
    """The ONE JSON line rank 0 prints last on stdout: the contract's keys + roofline + cpu_baseline + the hot-path totals, and a
    compact `also` list; per-kernel records, percentiles, conditions, allocator counters go to bench_detail.json (emit())."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    out["config"] = _pick(full["config"], ("workload", "global_batch", "parallelism", "submission", "sampling", "products"))
    cond = full.get("conditions") or {}
    # what the number was measured under, in three words (the sentences are in bench_detail.json): how the vendor GEMMs were picked and whether
    # Python's collector could interrupt the enqueue (--no-tunableop --gc-on give the other figure)
    out["conditions"] = {"tunableop": ("off" if cond.get("tunableop_results") is None else ("cached" if cond["tunableop_results"].startswith("read") else "tuned")),
                         "gc": "on" if cond.get("python_gc") == "on" else "frozen"}
    rk = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us")
    roof = _pick(full.get("roofline"), rk + ("frac_on_issued_instructions", "stage_us_per_call", "conv_equivalent_TFLOP_s"))
    if roof is not None and full["roofline"].get("hbm_kernel"):
        roof["hbm_kernel"] = _pick(full["roofline"]["hbm_kernel"], rk)
    out["roofline"] = roof
    out["cpu_baseline"] = full.get("cpu_baseline")
    out["hot_path"] = _pick(full.get("hot_path"), ("sum_kernel_us_per_img", "launches_per_img", "mean_proposals_per_img", "proposals_per_s",
                                                   "nms_plus_roi_us_per_img", "proposal_stage_us_per_img"))
    if full["n_gpus"] > 1:
        dd = full["distributed"]
        out["distributed"] = {"backend": dd["backend"], "per_rank_ms_per_step": dd["per_rank_ms_per_step"],
                              "ddp": _pick(dd.get("ddp"), ("num_parameter_tensors", "total_parameter_size_bytes", "bucket_cap_bytes",
                                                                   "parameter_tensors", "head_gradient_bytes", "trunk_gradient_bytes", "graphs"))}
    if also:
        out["also"] = []
        for rec in also:
            e = {"config": rec["config"]["workload"].split(",")[0], "dtype": rec["dtype"].split("(")[0], "submission": rec["config"]["submission"].split(" (")[0],
                 "value": rec["value"], "ms_per_step": rec["ms_per_step"], "roofline": _pick(rec.get("roofline"), ("kernel", "bound", "frac", "avg_launch_us")),
                 "nms_plus_roi_us_per_img": (rec.get("hot_path") or {}).get("nms_plus_roi_us_per_img")}
            if rec.get("eager_submission"):
                e["eager_value"] = rec["eager_submission"]["value"]
            if rec["config"].get("products"):
                e["products"] = "split: 6 bf16 MFMAs per product on exactly cut fp32 operands"
            out["also"].append(e)
    out["detail"] = "bench_detail.json"
    return out


def emit(full, also=()):
    """Full records -> bench_detail.json (repo root, and gpurun_out/ when it exists) + a per-kernel table on stderr; the compact line ->
    stdout, LAST."""
    detail = dict(full)
    if also:
        detail["also"] = list(also)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(detail, f, indent=1)
            except OSError as e:
                print("[bench] could not write %s/bench_detail.json: %s" % (d, e), file=sys.stderr)
    for rec in [full] + list(also):
        ks = (rec.get("hot_path") or {}).get("kernels") or {}
        print("[bench] %s | %s | %s" % (rec["config"]["workload"].split(",")[0], rec["dtype"], rec["config"]["submission"]), file=sys.stderr)
        for name, v in sorted(ks.items(), key=lambda kv: -kv[1]["us_per_img"]):
            print("[bench]   %-36s %5.2f /img  avg %8.2f us  %-7s %s" % (name, v["launches_per_img"], v["avg_us"], v["bound"],
                  ("%.0f GB/s" % v["GB_s"]) if v.get("GB_s") else (("%.0f TFLOP/s" % v["TFLOP_s"]) if v.get("TFLOP_s") else "")), file=sys.stderr)
    line = json.dumps(compact_record(full, also))
    if len(line) >= COMPACT_LIMIT:                     # never let the line outgrow the driver's capture again: shed the optional parts
        slim = compact_record(full, ())
        slim["also_dropped"] = "line would exceed %d bytes; see bench_detail.json" % COMPACT_LIMIT
        line = json.dumps(slim)
    sys.stderr.flush()
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--config", default="vgg", choices=sorted(CONFIGS), help="vgg = BASELINE configs[1]/[2] (headline); fpn = configs[3]/[4]")
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames kept resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket library kernels with HIP events")
    ap.add_argument("--event-every", type=int, default=10, help="bracket the library kernels of every n-th timed step (live roofline samples)")
    ap.add_argument("--lr", type=float, default=2e-3)          # config.py:24
    ap.add_argument("--channels-last", action="store_true", help="run the backbone in NHWC memory format")
    ap.add_argument("--no-fused-sgd", action="store_true")
    ap.add_argument("--no-tunableop", action="store_true", help="library default GEMM heuristics instead of PyTorch TunableOp")
    ap.add_argument("--gc-on", action="store_true", help="leave Python's cyclic garbage collector on during the timed region")
    ap.add_argument("--graph", action="store_true", help="capture the whole step (one HIP graph per resident frame) and time replays")
    ap.add_argument("--no-also", action="store_true", help="headline run only: skip the short FPN fp32 / bf16 runs of the `also` block")
    ap.add_argument("--also-steps", type=int, default=20)
    ap.add_argument("--products", default="native", choices=["native", "split"],
                    help="how the fp32 conv stage takes its products (ops.conv3x3_f32_products): native = the fp32 matrix instruction (default); split = the same "
                         "fp32 operands cut exactly into three bf16 pieces, six bf16 matrix instructions per 16 k rows, fp32 accumulation (as close to float64)")
    ap.add_argument("--amp", default="none", choices=["none", "bf16"],
                    help="autocast the torch layers (backbone / RPN convs / FC head); NOT the default: the reference trains in fp32")
    args = ap.parse_args()

    from faster_rcnn_pytorch_amd import parallel
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    rank, local_rank, world, device = parallel.init_for_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    # MIOpen: False = immediate mode (the perf database's pick, a GEMM fallback where it has none), True = time every applicable solver per shape
    # in the initialisation pass.  FRCNN_BENCH_MIOPEN_FIND=1 switches the search on (an experiment switch; the record's `conditions` say which ran)
    torch.backends.cudnn.benchmark = os.environ.get("FRCNN_BENCH_MIOPEN_FIND", "0") == "1"

    def release():                                    # the finished configuration's model / optimizer / graphs, before the next one is built
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()
    from faster_rcnn_pytorch_amd import ops as _ops
    _ops.conv3x3_f32_products(args.products)

    def tag_products(rec):
        if rec is not None and _ops.conv3x3_f32_products() == "split":
            rec["config"]["products"] = ("fp32 conv stage: every product as six bf16 matrix instructions on operands cut exactly into three bf16 pieces, fp32 accumulation "
                                         "(tests: as close to float64 as the fp32 matrix instruction)")
        return rec
    out = tag_products(run_config(args, args.config, args.amp, args.steps, args.warmup, args.graph, rank, world, device,
                                  with_cpu=(not args.no_cpu_baseline and world == 1)))
    release()
    also = []
    if rank == 0 and world == 1 and not args.no_also and args.config == "vgg" and args.amp == "none" and not args.graph:
        # BASELINE.json configs[3] / configs[4] and its "RoIAlign + NMS us/img" figure, timed by the same command (short runs).
        # The FPN step enqueues ~2500 launches from Python and is host-bound when submitted launch by launch (its eager figure moves
        # +-8 % between boxes); both entries are therefore timed as HIP-graph replays (--graph), with the eager figure beside it.
        # First the headline configuration itself as HIP-graph replays: the same step, submitted as one graph per resident frame instead of ~400 launches
        # (the headline `value` stays the eager figure -- the N > 1 runs are eager, DDP's bucket hooks are not captured -- with this one beside it).
        rec = run_config(args, "vgg", "none", args.also_steps, 5, True, rank, world, device, with_cpu=False)
        rec["config"]["note"] = "the headline configuration as graph replays: short run inside the headline command, %d timed steps, 5 warm-up" % args.also_steps
        rec["eager_submission"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "step_ms": out["step_ms"]}
        also.append(rec)
        release()
        if args.products == "native":
            # the same again with the conv stage's products on the bf16 matrix cores (opt-in, --products split): not the headline, shown beside it
            _ops.conv3x3_f32_products("split")
            try:
                rec = tag_products(run_config(args, "vgg", "none", args.also_steps, 5, True, rank, world, device, with_cpu=False))
            finally:
                _ops.conv3x3_f32_products("native")
            rec["config"]["note"] = "the headline configuration as graph replays with --products split: short run inside the headline command, %d timed steps, 5 warm-up" % args.also_steps
            also.append(rec)
            release()
        for amp in ("none", "bf16"):
            rec = run_config(args, "fpn", amp, args.also_steps, 5, True, rank, world, device, with_cpu=False)
            rec["config"]["note"] = "short run inside the headline command: %d timed steps, 5 warm-up" % args.also_steps
            release()
            eager = run_config(args, "fpn", amp, args.also_steps, 5, False, rank, world, device, with_cpu=False)
            rec["eager_submission"] = {"value": eager["value"], "ms_per_step": eager["ms_per_step"], "step_ms": eager["step_ms"]}
            also.append(rec)
            release()
    if rank == 0:
        emit(out, also)
    parallel.shutdown()


def cpu_baseline(config, cfg, steps, lr):
    """The oracle's CPU restatement of the same training step (kind 'port'), bounded sample."""
    from oracle import oracle as orc
    from oracle import model_ref
    orc.build()
    torch.manual_seed(0)
    if config == "vgg":
        ref = model_ref.RefFRCNN(cfg["num_classes"])
        heads = (ref.rpn.inter_layer, ref.rpn.cls_layer, ref.rpn.reg_layer)
        steps = 8 if steps is None else steps
    else:
        from faster_rcnn_pytorch_amd.new_model import BackboneWithFPN      # the backbone is plain torch on both sides
        ref = model_ref.RefFRCNNFPN(BackboneWithFPN(trainable_layers=3), cfg["num_classes"])
        heads = (ref.rpn_head.inter_layer, ref.rpn_head.cls_layer, ref.rpn_head.reg_layer)
        steps = 4 if steps is None else steps
    if config == "vgg":
        for m in ref.modules():
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                torch.nn.init.zeros_(m.bias)
    for m in heads:
        m.weight.data.normal_(0, 0.01)
        m.bias.data.zero_()
    opt = torch.optim.SGD([p for p in ref.parameters() if p.requires_grad], lr=lr, momentum=0.9, weight_decay=1e-4)
    # the GPU box gives one GPU's share of the host: 16 cores; more threads than that oversubscribes the cgroup
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    print("[bench] cpu baseline on %d threads (affinity %d, cpu_count %s)" % (cores, avail, os.cpu_count()), file=sys.stderr, flush=True)

    def one(i):
        x, b, l = synth_frame(cfg, 0, i)
        pred, target = ref(x, [b], [l])
        loss = model_ref.ref_loss(pred, target)[0]
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
    name = "RefFRCNN" if config == "vgg" else "RefFRCNNFPN"
    return {"value": round(steps / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps (fwd+loss+bwd+SGD) of oracle/model_ref.%s on the same synthetic %dx%d "
                      "frames, torch CPU %d threads + oracle C path, %.1f s" % (steps, name, cfg["H"], cfg["W"], cores, dt)}


if __name__ == "__main__":
    main()
