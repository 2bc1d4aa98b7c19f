"""Developer tool: RoIAlign backward alone at the bench's FPN RoIs (captured after STEPS SGD steps): bin-size statistics and the time
of its launches (two since round 5; four before).  FRCNN_HIP_LIB=build_dbg/<variant>/libfrcnn_hip.so selects a variant build."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from faster_rcnn_pytorch_amd import ops
dev = torch.device("cuda:0")
path = os.path.join(ROOT, "build_dbg", "fpn_rois.npz")
if not os.path.exists(path) or os.environ.get("RECAPTURE"):
    from faster_rcnn_pytorch_amd.new_model import FRCNN
    from faster_rcnn_pytorch_amd.loss import FRCNNLoss
    cfg = bench.CONFIGS["fpn"]
    model = FRCNN(num_classes=cfg["num_classes"], sampling="device", seed=1234).to(dev)
    cap = {}
    orig = ops.ms_roi_align
    def spy(feats, boxes, out, sr, scales, *a, **k):
        cap["rois"] = (boxes[0] if isinstance(boxes, (list, tuple)) else boxes).detach().float().cpu().numpy()
        cap["scales"] = np.asarray(scales, np.float32); cap["shapes"] = np.asarray([tuple(f.shape) for f in feats])
        return orig(feats, boxes, out, sr, scales, *a, **k)
    ops.ms_roi_align = spy
    crit = FRCNNLoss()
    opt = torch.optim.SGD(model.parameters(), lr=2e-3, momentum=0.9, weight_decay=5e-4)
    keep = {}
    for step in range(30):
        x, bbox, label = bench.synth_frame(cfg, 0, step % 8)
        model.train()
        pred, target = model(x.to(dev), [bbox.to(dev)], [label.to(dev)])
        loss = crit(pred, target)[0]
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        if step in (0, 15, 29): keep["rois%d" % step] = cap["rois"].copy()
    ops.ms_roi_align = orig
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, scales=cap["scales"], shapes=cap["shapes"], **keep)
    del model, opt
d = np.load(path)
scales = [float(s) for s in d["scales"]]; shapes = [tuple(int(v) for v in s) for s in d["shapes"]]
print("scales", scales, "shapes", shapes)
for key in os.environ.get("KEYS", "rois0,rois15,rois29").split(","):
    r = d[key]
    w = r[:, 2] - r[:, 0]; h = r[:, 3] - r[:, 1]
    k = np.floor(4 + np.log2(np.sqrt(np.maximum(w * h, 1e-12)) / 224 + 1e-6)).clip(2, 5).astype(int) - 2
    sc = np.asarray(scales)[k]
    bw = np.maximum(w * sc, 1) / 7; bh = np.maximum(h * sc, 1) / 7
    nb = lambda b: np.where(b >= 4 / 3, 2, np.where(b >= 1, 3, np.where(b >= 0.8, 4, 7)))
    nbx, nby = nb(bw), nb(bh)
    print(key, "R", len(r), "per level", np.bincount(k, minlength=4), "| bins per pixel (x,y) <=2,<=2: %.2f  <=3,<=3: %.2f | mean nbx*nby %.2f" % (
        np.mean((nbx <= 2) & (nby <= 2)), np.mean((nbx <= 3) & (nby <= 3)), np.mean(nbx * nby)),
        "| bw pct 10/50/90: %.2f %.2f %.2f" % tuple(np.percentile(bw, [10, 50, 90])))
    import ctypes as C
    from faster_rcnn_pytorch_amd.ops import lib, _ptr, _np_ptr, _level_tables, _stream, check
    rois = torch.from_numpy(r).to(dev)
    g = torch.randn((len(r), 256, 7, 7), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    grads = [torch.empty(s, device=dev) for s in shapes]
    ptrs, H, W, sc = _level_tables(grads, scales)
    nb = lib.frcnn_ms_roi_align_bwd_workspace(_np_ptr(H), _np_ptr(W), 4, 256, len(r))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    def run():
        check(lib.frcnn_ms_roi_align_bwd(_ptr(g), ptrs, _np_ptr(H), _np_ptr(W), _np_ptr(sc), 4, 256, _ptr(rois), len(r), 7, 7, 2, 0, 2, 224.0, 4,
                                         _ptr(ws), ws.numel(), _stream()), "bwd")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    import hashlib
    feats = [torch.randn(s, device=dev, generator=torch.Generator(device=dev).manual_seed(6)) for s in shapes]
    fptrs, _, _, _ = _level_tables(feats, scales)
    out = torch.empty((len(r), 256, 7, 7), device=dev)
    frois = rois
    if os.environ.get('SORT'):                               # experiment: largest footprints first (longest-processing-time-first over the CUs)
        fa = (w * sc_np) * (h * sc_np) if False else None
        area = ((r[:, 2] - r[:, 0]) * np.asarray(scales)[k]) * ((r[:, 3] - r[:, 1]) * np.asarray(scales)[k])
        frois = torch.from_numpy(r[np.argsort(-area)].copy()).to(dev)
    order = None
    if os.environ.get('ORDER'):                              # the library's own device-side order (round 4): rois are already in pixels -> mul = 1
        _, order = ops.roi_scale_order(rois, (1.0, 1.0, 1.0, 1.0), [s[-2:] for s in shapes], scales)
    def fwd():
        check(lib.frcnn_ms_roi_align_fwd(fptrs, _np_ptr(H), _np_ptr(W), _np_ptr(sc), 4, 256, _ptr(frois), len(r), 7, 7, 2, 0, 2, 224.0, 4, _ptr(out), None, _ptr(order), _stream()), "fwd")
    for _ in range(5): fwd()
    torch.cuda.synchronize()
    tf = []
    for _ in range(5):
        e0.record()
        for _ in range(20): fwd()
        e1.record(); torch.cuda.synchronize()
        tf.append(e0.elapsed_time(e1) * 1e3 / 20)
    print("   forward: median %.1f us  min %.1f us   sha1 %s" % (float(np.median(tf)), min(tf), hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]))
    h = hashlib.sha1(b"".join(x.cpu().numpy().tobytes() for x in grads)).hexdigest()[:12]
    print("   backward (launches back to back): median %.1f us  min %.1f us   sha1 %s" % (float(np.median(ts)), min(ts), h))
# the ordering kernel itself (round 4)
for _ in range(3): ops.roi_scale_order(rois, (1.0, 1.0, 1.0, 1.0), [s[-2:] for s in shapes], scales)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.roi_scale_order(rois, (1.0, 1.0, 1.0, 1.0), [s[-2:] for s in shapes], scales)
e1.record(); torch.cuda.synchronize()
print("roi_scale_order: %.1f us per call (host-bound loop, includes the launch)" % (e0.elapsed_time(e1) * 1e3 / 50))
