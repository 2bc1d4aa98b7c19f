"""bench.py's record assembly without a GPU: the line rank 0 prints must stay parsable by the driver (round 3's grew to 27 KB and
`BENCH_r03.json.parsed` came back null), and the N > 1 plumbing (max-over-ranks rule, per-rank times, the `distributed` block) must run
end to end before the first real 8-GPU launch (world-size-2 gloo, stub model)."""
import json
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V_KERNELS = ["nms_kernel", "rpn_match_kernel", "roi_pool_bwd_lds_kernel", "rpn_head_tail_bwd_kernel", "roi_pool_fwd_lds_kernel",
             "head_targets_kernel", "rpn_head_tail_kernel", "topk_partition_kernel", "proposal_prologue_kernel", "topk_bucket_kernel",
             "det_loss_kernel", "rpn_head_tail_bwd_finalize_kernel", "rpn_conv3x3_f32_kernel", "rpn_conv3x3_f32_bwd_data_kernel",
             "rpn_conv3x3_f32_wgrad_kernel"]
F_KERNELS = ["rpn_conv3x3_wgrad_kernel", "rpn_conv3x3_head_kernel", "rpn_conv3x3_bwd_data_kernel", "roi_align_fwd77_kernel",
             "rpn_head_tail_bwd_kernel", "roi_align_bwd_tile_kernel", "nms_kernel", "rpn_match_kernel", "topk_partition_kernel",
             "head_targets_kernel", "proposal_prologue_kernel", "det_loss_kernel", "rpn_apply_kernel", "rpn_conv_wgrad_finalize_kernel",
             "roi_align_bwd_lists_kernel", "rpn_conv_pack_bwd_kernel",
             "rpn_head_tail_bwd_finalize_kernel", "rpn_conv_pack_kernel"]


def _canned(bench, config, amp, names, world=1, graph=False):
    samples = {n: [0.01 * (i + 1) + 0.001 * j for j in range(15)] for i, n in enumerate(names)}
    pmc = {n: {"traffic_bytes": 12345678 + i} for i, n in enumerate(names)}
    cpu = {"value": 0.6456, "unit": "images/s", "cores": 16, "kind": "port",
           "sample": "8 full training steps (fwd+loss+bwd+SGD) of oracle/model_ref.RefFRCNN on the same synthetic 600x1000 frames, torch CPU 16 threads + oracle C path, 12.4 s"}
    ddp = None
    if world > 1:
        ddp = {"num_parameter_tensors": 40, "total_parameter_size_bytes": 548312956, "unique_trainable_parameters": 40,
               "unique_trainable_bytes": 548312956, "registered_names_with_aliases": 44, "bucket_cap_bytes": 104857600,
               "bucket_sizes": [1, 2, 3], "rebuilt_bucket_sizes": [3, 2, 1], "gradient_as_bucket_view": True, "find_unused_parameters": False}
    return bench.build_record(config, amp, world=world, steps=60, warmup=12, dt=0.85 * world, per_rank_ms=[14.123 + r for r in range(world)],
                              step_ms=[14.1 + 0.01 * i for i in range(60)], samples=samples, n_sampled=15, n_props=[783, 801, 779], graph=graph,
                              pmc=pmc, pmc_src="profiles/r04_pmc_traffic_%s.json" % config, cpu=cpu if world == 1 else None,
                              allocator={"num_device_alloc": 0, "num_device_free": 0, "num_alloc_retries": 0, "num_ooms": 0}, ddp=ddp,
                              backend="nccl" if world > 1 else None, world_seen=world, final_loss=1.2345)


def test_compact_line_stays_under_4k_with_every_optional_block():
    import bench
    full = _canned(bench, "vgg", "none", V_KERNELS)
    also = [_canned(bench, "vgg", "none", V_KERNELS, graph=True)]            # the headline configuration as graph replays, then the two FPN entries
    also[0]["eager_submission"] = {"value": full["value"], "ms_per_step": full["ms_per_step"], "step_ms": full["step_ms"]}
    for amp in ("none", "bf16"):
        rec = _canned(bench, "fpn", amp, F_KERNELS, graph=True)
        rec["eager_submission"] = {"value": 55.123, "ms_per_step": 18.141, "step_ms": rec["step_ms"]}
        also.append(rec)
    line = json.dumps(bench.compact_record(full, also))
    assert len(line) < bench.COMPACT_LIMIT, len(line)
    assert len(json.dumps(full)) > bench.COMPACT_LIMIT                        # the full record is what goes to bench_detail.json
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "hot_path", "also"):
        assert k in out, k
    assert out["value"] == round(60 / 0.85, 3) and out["n_gpus"] == 1 and out["dtype"] == "f32" and "workload" in out["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us"):
        assert k in out["roofline"], k
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-3
    assert out["roofline"]["traffic"] is not None
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    for k in ("sum_kernel_us_per_img", "launches_per_img", "mean_proposals_per_img", "proposals_per_s", "nms_plus_roi_us_per_img",
              "proposal_stage_us_per_img"):
        assert out["hot_path"][k] is not None, k
    # what the figure was measured under rides on the line itself (the driver keeps only the parsed line)
    assert out["conditions"]["tunableop"] in ("tuned", "cached", "off") and out["conditions"]["gc"] in ("frozen", "on")
    assert out["config"]["sampling"] == "device-philox"
    assert [a["dtype"] for a in out["also"]] == ["f32", "f32", "bf16"] and all(a["roofline"]["kernel"] for a in out["also"])
    assert out["also"][0]["eager_value"] == out["value"] and out["also"][0]["submission"].startswith("one HIP graph")


def test_emit_prints_the_compact_line_last_and_writes_the_detail_file(tmp_path, capsys, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _canned(bench, "vgg", "none", V_KERNELS)
    bench.emit(full, [_canned(bench, "fpn", "none", F_KERNELS, graph=True)])
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    assert len(last) < bench.COMPACT_LIMIT and json.loads(last)["roofline"]["kernel"]
    detail = json.load(open(tmp_path / "bench_detail.json"))
    assert set(V_KERNELS) == set(detail["hot_path"]["kernels"]) and len(detail["also"]) == 1
    assert "rpn_conv3x3_f32_kernel" in cap.err                               # the per-kernel table goes to stderr


def test_multi_gpu_record_is_compact_and_carries_the_distributed_block():
    import bench
    full = _canned(bench, "vgg", "none", V_KERNELS, world=8)
    line = json.dumps(bench.compact_record(full))
    out = json.loads(line)
    assert len(line) < bench.COMPACT_LIMIT
    assert out["n_gpus"] == 8 and out["value"] == round(8 * 60 / (0.85 * 8), 3) and out["config"]["parallelism"] == "dp8"
    assert out["distributed"]["backend"] == "nccl" and len(out["distributed"]["per_rank_ms_per_step"]) == 8
    assert out["distributed"]["ddp"]["total_parameter_size_bytes"] == 548312956 and out["cpu_baseline"] is None


def _bench_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import time
    import bench
    from faster_rcnn_pytorch_amd import parallel
    r, _, w, dev = parallel.init_for_distributed(backend="gloo")
    torch.manual_seed(0)
    net = parallel.wrap_ddp(torch.nn.Linear(16, 4), dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    calls = []

    def step(i):
        calls.append(i)
        loss = net(torch.full((2, 16), float(r + i))).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        time.sleep(0.01 * (1 + 2 * r))                                        # rank 1 is the slow one: the job's time is ITS time
        return loss
    armed = []
    tr = bench.timed_region(step, 5, 2, dev, armed=lambda: armed.append(len(calls)))
    rec = None
    ddp = parallel.ddp_report(net)
    if r == 0:
        rec = bench.build_record("vgg", "none", world=w, steps=5, warmup=2, dt=tr["dt"], per_rank_ms=tr["per_rank_ms"], step_ms=tr["step_ms"],
                                 samples={"nms_kernel": [0.07, 0.071], "roi_pool_bwd_lds_kernel": [0.02, 0.021]}, n_sampled=2, n_props=[700],
                                 graph=False, pmc={}, pmc_src=None, cpu=None, allocator={}, ddp=ddp,
                                 backend=torch.distributed.get_backend(), world_seen=torch.distributed.get_world_size(), final_loss=float(tr["last"]))
        rec = bench.compact_record(rec)
    q.put((r, calls, armed, tr["dt"], tr["dt_local"], tr["per_rank_ms"], float(net.module.weight.detach().sum()), rec))
    parallel.shutdown()


def test_bench_timing_rule_and_record_assembly_world2_gloo():
    """timed_region + build_record + compact_record under torch.distributed with two ranks (gloo on the CPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 13) % 2000
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, c0, a0, dt0, l0, pr0, w0, rec), (_, c1, a1, dt1, l1, pr1, w1, none) = res
    assert c0 == c1 == list(range(7)) and a0 == a1 == [2]                    # W warm-up steps, then EXACTLY K timed ones
    assert dt0 == dt1 and dt0 >= max(l0, l1) - 1e-9 and l1 >= 5 * 0.03        # MAX over ranks, on both ranks
    assert pr0 == pr1 and len(pr0) == 2 and abs(pr0[1] - l1 / 5 * 1e3) < 1e-2
    assert abs(w0 - w1) < 1e-6                                                # DDP kept the replicas in step
    assert none is None and rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 2
    assert abs(rec["value"] - 2 * 5 / dt0) < 1e-2 and abs(rec["ms_per_step"] - dt0 / 5 * 1e3) < 1e-2
    assert rec["distributed"]["backend"] == "gloo" and rec["distributed"]["per_rank_ms_per_step"] == pr0
    assert rec["distributed"]["ddp"]["num_parameter_tensors"] == 2
    assert len(json.dumps(rec)) < 4096


def test_winograd_stage_accounting_in_the_roofline():
    """The fp32 3x3 convolutions (the RPN's and the backbone layers the stage takes) run as Winograd stages of several launches: the GEMM
    kernel is priced on the flops IT executes per image -- 2 (m + 2)^2 Cin Cout per padded m x m tile, summed over the calls one step
    makes (ops.CONV_TRACE) -- over the per-image time of its launches: a fraction of the fp32 MFMA peak, never above it; the convolutions'
    own flop count over the time of ALL the stage's launches rides beside it."""
    import bench
    rpn = {"Cin": 512, "Cout": 512, "shapes": [(13, 21)], "mask": False, "bias": False, "cached": False}          # 4 x 6 tiles of 4 x 4 would be padding: m = 2, 7 x 11 = 77 -> 128
    c4 = {"Cin": 512, "Cout": 512, "shapes": [(75, 125)], "mask": True, "bias": True, "cached": True}            # 19 x 32 = 608 tiles: m = 4
    calls = [dict(rpn, kind=k) for k in ("fwd", "bwd_data", "wgrad")] + [dict(c4, kind="fwd", mask=False), dict(c4, kind="bwd_data"), dict(c4, kind="wgrad")]
    tot, conv = bench.wino_work(calls)
    assert tot["rpn_wino_gemm_kernel"]["launches"] == 6 and tot["rpn_wino_input_kernel"]["launches"] == 4 + 2 + 1     # rpn wgrad transforms x, the cached one does not
    own = 3 * 2 * 16 * 512 * 512 * 128 + 3 * 2 * 36 * 512 * 512 * 640                                            # 77 -> 128 and 608 -> 640 padded tiles
    assert tot["rpn_wino_gemm_kernel"]["flops"] == own and conv == 3 * 18 * 512 * 512 * (13 * 21 + 75 * 125)
    us = {"rpn_wino_gemm_kernel": 0.090, "rpn_wino_input_kernel": 0.020, "rpn_wino_output_kernel": 0.012, "rpn_wino_weight_kernel": 0.008,
          "rpn_wino_dw_kernel": 0.008, "nms_kernel": 0.070}
    n = {k: tot[k]["launches"] for k in tot}
    n["nms_kernel"] = 1
    samples = {k: [us[k]] * (n[k] * 5) for k in us}                          # five bracketed steps
    kw = dict(world=1, steps=20, warmup=4, dt=0.2, per_rank_ms=[10.0], step_ms=[10.0] * 20, n_sampled=5, n_props=[780], graph=False, pmc={},
              pmc_src=None, cpu=None, allocator={}, ddp=None, backend=None, world_seen=1, final_loss=1.0)
    rec = bench.build_record("vgg", "none", samples=samples, conv_calls=calls, **kw)
    r = rec["roofline"]
    assert r["kernel"] == "rpn_wino_gemm_kernel" and r["bound"] == "mfma" and r["peak"] == 157.3
    assert abs(r["achieved"] - own / (6 * 90.0) * 1e-6) < 0.05 and 0 < r["frac"] < 1
    stage = 6 * 90 + 7 * 20 + 4 * 12 + 4 * 8 + 2 * 8
    assert abs(r["stage_us_per_img"] - stage) < 0.05 and r["stage_calls_per_img"] == 6
    assert abs(r["conv_equivalent_TFLOP_s"] - conv / stage * 1e-6) < 0.05
    hk = r["hbm_kernel"]
    assert hk["kernel"] == "rpn_wino_input_kernel" and abs(hk["achieved"] - tot["rpn_wino_input_kernel"]["bytes"] / (7 * 20.0) * 1e-3) < 0.5
    c = bench.compact_record(rec)["roofline"]
    assert c["conv_equivalent_TFLOP_s"] == r["conv_equivalent_TFLOP_s"] and c["frac"] == r["frac"]
    # a step that did not make the traced calls gets no figure rather than a wrong one
    short = dict(samples, rpn_wino_gemm_kernel=[0.09] * 25)
    r2 = bench.build_record("vgg", "none", samples=short, conv_calls=calls, **kw)["roofline"]
    assert r2["kernel"] != "rpn_wino_gemm_kernel" or r2.get("frac") is None


def test_norm_passes_are_priced_on_the_bytes_the_traced_step_moved():
    """The fused norm / residual / ReLU passes run on maps of many sizes: their HBM figure is the traced step's byte total (ops.AFFINE_TRACE) over the
    per-image time of the launches -- and absent when the sampled steps made a different number of launches."""
    import bench
    samples = {"affine_act_fwd_mixed_kernel": [0.010] * (53 * 5), "nms_kernel": [0.040] * 5}
    kw = dict(world=1, steps=20, warmup=4, dt=0.2, per_rank_ms=[10.0], step_ms=[10.0] * 20, n_sampled=5, n_props=[780], graph=False, pmc={},
              pmc_src=None, cpu=None, allocator={}, ddp=None, backend=None, world_seen=1, final_loss=1.0)
    nbytes = 53 * 20_000_000
    r = bench.build_record("fpn", "bf16", samples=samples, affine_calls={"affine_act_fwd_mixed_kernel": [53, nbytes]}, **kw)["roofline"]
    assert r["kernel"] == "affine_act_fwd_mixed_kernel" and r["bound"] == "hbm"
    assert abs(r["achieved"] - nbytes / (53 * 10.0) * 1e-3) < 0.5 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4 and r["algorithmic_bytes"] == 20_000_000
    r2 = bench.build_record("fpn", "bf16", samples=samples, affine_calls={"affine_act_fwd_mixed_kernel": [50, nbytes]}, **kw)["roofline"]
    assert r2["achieved"] is None


def test_stage_accounting_of_pooled_calls_and_1x1_weight_gradients():
    """wino_work() prices what ops.CONV_TRACE records: a forward that pooled writes a quarter of the pixels (+ the window words), its gradient calls read the
    gradient at the pooled size (+ the words); a 1 x 1 convolution's weight gradient is one GEMM launch of 2 M N K flops and nothing else."""
    import bench
    lay = {"Cin": 64, "Cout": 64, "shapes": [(600, 1000)], "bias": True}
    T = 150 * 250                                                            # 4 x 4 tiles: >= 512 -> m = 4, P = 36
    Tp = -(-T // 128) * 128
    fwd = dict(lay, kind="fwd", mask=False, cached=False, relu_bits=True, pooled=True)
    bwd = dict(lay, kind="bwd_data", mask=True, cached=False, relu_bits=False, pooled=True)
    wg = dict(lay, kind="wgrad", mask=True, cached=True, relu_bits=False, pooled=True)
    tot, conv = bench.wino_work([fwd, bwd, wg, {"kind": "gemm_nt", "M": 128, "N": 512, "K": 16800, "splits": 35}])
    # 64 -> 64 channels on 4 x 4 tiles: forward and data gradient run the product and the output transform as ONE launch (rpn_wino_gemm_out64_kernel: U and V in,
    # the outputs out, no product planes); the weight gradient and the 1 x 1 gradient stay on the GEMM
    assert tot["rpn_wino_gemm_kernel"]["launches"] == 2 and tot["rpn_wino_output_kernel"]["launches"] == 0 and tot["rpn_wino_gemm_out64_kernel"]["launches"] == 2
    assert tot["rpn_wino_gemm_kernel"]["flops"] == 2 * 36 * 64 * 64 * Tp + 2 * 128 * 512 * 16800
    assert tot["rpn_wino_gemm_out64_kernel"]["flops"] == 2 * 2 * 36 * 64 * 64 * Tp
    assert conv == 3 * 18 * 64 * 64 * 600 * 1000 + 2 * 128 * 512 * 16800
    words = 2 * 64 * Tp
    operands = 4 * 36 * 64 * 64 + 4 * 36 * 64 * Tp
    assert tot["rpn_wino_gemm_out64_kernel"]["bytes"] == (operands + 4 * 64 * 300 * 500 + words) + (operands + 4 * 64 * 600 * 1000)
    other = dict(lay, Cout=128)                                              # any other channel pair keeps the two launches
    t2, _ = bench.wino_work([dict(other, kind="fwd", mask=False, cached=False, relu_bits=True, pooled=False)])
    assert t2["rpn_wino_gemm_out64_kernel"]["launches"] == 0 and t2["rpn_wino_gemm_kernel"]["launches"] == 1 and t2["rpn_wino_output_kernel"]["launches"] == 1
    assert tot["rpn_wino_input_kernel"]["launches"] == 3                      # x (forward), pooled dy twice; the weight gradient's x transform was kept
    assert tot["rpn_wino_input_kernel"]["bytes"] == (4 * 64 * 600 * 1000 + 4 * 36 * 64 * Tp) + 2 * (4 * 64 * 300 * 500 + words + 4 * 36 * 64 * Tp)


def test_the_bench_mirror_of_the_stage_tiling_is_the_library_rule():
    """bench.wino_tiling (tile size and padded tile total of a stage call: what the GEMM's flops are priced on) against the library's own answers
    (frcnn_conv3x3_f32_tile_size; frcnn_conv3x3_f32_relu_bits_words with one channel = the padded total) over the shapes of both configurations, the
    FPN's five-level call and a sweep of small maps -- including the 37 x 62 map whose 160 4 x 4 tiles are padded to 192 (64-wide product tiles), not 256."""
    import numpy as np
    import bench
    from faster_rcnn_pytorch_amd import _lib
    lib = _lib.lib
    cases = [[(600, 1000)], [(300, 500)], [(150, 250)], [(75, 125)], [(37, 62)], [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)],
             [(200, 336)], [(100, 168)], [(50, 84)], [(25, 42)], [(13, 21)]] + [[(h, w)] for h in (1, 5, 16, 23, 31, 40) for w in (3, 17, 32, 47, 64)]
    for shapes in cases:
        H = np.ascontiguousarray([h for h, _ in shapes], dtype=np.int32)
        W = np.ascontiguousarray([w for _, w in shapes], dtype=np.int32)
        m = int(lib.frcnn_conv3x3_f32_tile_size(H.ctypes.data, W.ctypes.data, len(shapes)))
        tp = int(lib.frcnn_conv3x3_f32_relu_bits_words(H.ctypes.data, W.ctypes.data, len(shapes), 1))
        assert (m, tp) == bench.wino_tiling(shapes), shapes
    assert bench.wino_tiling([(37, 62)]) == (4, 192) and bench.wino_tiling([(75, 125)]) == (4, 640) and bench.wino_tiling([(13, 21)]) == (2, 128)

