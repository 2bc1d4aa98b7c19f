"""ctypes binding of libfrcnn_hip.so (include/frcnn_hip.h).

There is NO fallback: if the shared library is missing the import fails loudly, and every op
in ops.py refuses non-GPU tensors.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C faster_rcnn_pytorch_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FRCNN_HIP_LIB") or os.path.join(_HERE, "lib", "libfrcnn_hip.so")     # override: kernel tuning builds only

OK = 0
ABI_VERSION = 7
HT_ERR_PERM_LENGTH, HT_ERR_PERM_RANGE, HT_ERR_UPSTREAM_ABORT, HT_ERR_SHORT = 1, 2, 4, 8
OP_TOPK, OP_NMS, OP_REGION_PROPOSAL, OP_RPN_TARGETS, OP_HEAD_TARGETS, OP_PREPROCESS, OP_HEAD_BWD, OP_RPN_CONV, OP_RPN_CONV_WGRAD, OP_RPN_CONV_F32 = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10

_vp, _i, _i64, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); must list EVERY symbol include/frcnn_hip.h declares (tests check this)
SIGNATURES = {
    "frcnn_abi_version": (_i, []),
    "frcnn_layout_check": (_i, []),
    "frcnn_layout_stamp": (_u64, []),
    "frcnn_last_error": (C.c_char_p, []),
    "frcnn_workspace_bytes": (_sz, [_i, _i64, _i64]),
    "frcnn_anchor_base_host": (_i, [_i, _vp, _i, _vp, _i, _vp]),
    "frcnn_tv_base_anchors_host": (_i, [_f, _vp, _i, _vp]),
    "frcnn_anchor_grid": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _i64, _vp]),
    "frcnn_box_codec": (_i, [_i, _vp, _vp, _i64, _vp, _vp]),
    "frcnn_pairwise_iou": (_i, [_vp, _i64, _vp, _i64, _f, _vp, _vp]),
    "frcnn_proposal_prologue": (_i, [_vp, _vp, _vp, _i64, _f, _vp, _vp, _vp]),
    "frcnn_topk_sorted": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_argsort_desc": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_nms": (_i, [_vp, _vp, _i64, _f, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_nms_classed": (_i, [_vp, _vp, _vp, _i64, _f, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_region_proposal": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _f, _f, _f, _i64, _f, _i64, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_head_tail_fwd": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "frcnn_rpn_head_tail_ml_fwd": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "frcnn_rpn_head_tail_ml_bwd": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv_head_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv_bwd_data": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv3x3_f32_workspace": (_sz, [_vp, _vp, _i, _i]),
    "frcnn_gemm_nt_f32_workspace": (_sz, []),
    "frcnn_gemm_nt_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "frcnn_affine_act_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "frcnn_affine_act_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "frcnn_affine_act_fwd_mixed": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "frcnn_affine_act_bwd_mixed": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "frcnn_conv3x3_c3_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "frcnn_conv3x3_c3_wgrad_workspace": (_sz, [_i, _i]),
    "frcnn_conv3x3_c3_wgrad": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_conv3x3_f32_workspace": (_sz, [_vp, _vp, _i, _i, _i]),
    "frcnn_conv3x3_f32_xt_floats": (_sz, [_vp, _vp, _i, _i]),
    "frcnn_conv3x3_f32_relu_bits_words": (_sz, [_vp, _vp, _i, _i]),
    "frcnn_conv3x3_f32_u_floats": (_sz, [_vp, _vp, _i, _i, _i]),
    "frcnn_conv3x3_f32_products": (_i, [_i]),
    "frcnn_conv3x3_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_conv3x3_f32_tile_size": (_i, [_vp, _vp, _i]),
    "frcnn_conv3x3_f32_supported": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "frcnn_conv3x3_f32_bwd_data": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "frcnn_conv3x3_f32_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv3x3_f32_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv3x3_f32_bwd_data": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_conv3x3_f32_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_rpn_targets": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_head_targets": (_i, [_i, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _u64, _u64, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "frcnn_roi_pool_fwd": (_i, [_vp, _i, _i, _i, _vp, _i64, _i, _i, _f, _vp, _vp, _vp]),
    "frcnn_roi_pool_bwd": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "frcnn_roi_pool_fwd_a16": (_i, [_vp, _i, _i, _i, _vp, _i64, _f, _vp, _vp, _vp]),
    "frcnn_roi_pool_bwd_a16": (_i, [_vp, _vp, _vp, _f, _i64, _i, _i, _i, _vp, _vp]),
    "frcnn_roi_level_map": (_i, [_vp, _i64, _i, _i, _f, _i, _f, _vp, _vp]),
    "frcnn_ms_roi_align_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "frcnn_roi_scale_order": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "frcnn_ms_roi_align_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "frcnn_ms_roi_align_bwd_workspace": (_sz, [_vp, _vp, _i, _i, _i64]),
    "frcnn_detection_loss": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_preprocess_image": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_preprocess_boxes": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    "frcnn_diag_occupy": (_i, [_i, _i, _vp]),
    "frcnn_prof_enable": (_i, [_i]),
    "frcnn_prof_collect": (_i, []),
    "frcnn_prof_reset": (_i, []),
    "frcnn_prof_num_kernels": (_i, []),
    "frcnn_prof_kernel_name": (C.c_char_p, [_i]),
    "frcnn_prof_get": (_i, [_i, _vp, _vp]),
    "frcnn_prof_get_samples": (_i64, [_i, _vp, _i64]),
}


class FrcnnError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfrcnn_hip.so not found at %s -- the HIP extension is mandatory (no CPU fallback). "
            "Build it: make -C faster_rcnn_pytorch_amd/csrc   (or __graft_entry__.build())" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    ver = lib.frcnn_abi_version()
    if ver < 0:                          # the library's own objects disagree about a shared layout (csrc/frcnn_layout.h): never run its kernels
        raise ImportError("libfrcnn_hip.so refuses to load: %s" % (lib.frcnn_last_error() or b"").decode())
    if ver != ABI_VERSION:
        raise ImportError("libfrcnn_hip.so ABI version %d, expected %d (rebuild: make -C faster_rcnn_pytorch_amd/csrc)" % (ver, ABI_VERSION))
    return lib


lib = _load()


def check(rc, what=""):
    if rc != OK:
        msg = lib.frcnn_last_error()
        raise FrcnnError("%s failed with status %d: %s" % (what or "libfrcnn_hip", rc, msg.decode() if msg else ""))


def workspace_bytes(op, n1, n2=0):
    return int(lib.frcnn_workspace_bytes(op, n1, n2))


def diag_occupy(n_workgroups, microseconds, stream):
    """Resident do-nothing workgroups on `stream` (a torch.cuda.Stream): see include/frcnn_hip.h."""
    check(lib.frcnn_diag_occupy(int(n_workgroups), int(microseconds), C.c_void_p(stream.cuda_stream)))


def prof_enable(on=True):
    check(lib.frcnn_prof_enable(1 if on else 0))


def prof_reset():
    check(lib.frcnn_prof_reset())


def prof_report():
    """{kernel_name: (total_ms, launches)} for kernels launched since the last reset (syncs the events)."""
    check(lib.frcnn_prof_collect())
    out = {}
    for k in range(lib.frcnn_prof_num_kernels()):
        ms, n = C.c_double(0), C.c_int64(0)
        check(lib.frcnn_prof_get(k, C.byref(ms), C.byref(n)))
        if n.value:
            out[lib.frcnn_prof_kernel_name(k).decode()] = (ms.value, n.value)
    return out


def prof_samples():
    """{kernel_name: [per-launch ms, ...]} since the last reset (syncs the events)."""
    check(lib.frcnn_prof_collect())
    out = {}
    for k in range(lib.frcnn_prof_num_kernels()):
        n = int(lib.frcnn_prof_get_samples(k, None, 0))
        if n > 0:
            buf = (C.c_float * n)()
            lib.frcnn_prof_get_samples(k, buf, n)
            out[lib.frcnn_prof_kernel_name(k).decode()] = list(buf)
    return out
