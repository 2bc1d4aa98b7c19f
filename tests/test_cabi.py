"""CPU-only checks of the drop-in boundary: libfrcnn_hip.so loads without a GPU and exports exactly
the symbols include/frcnn_hip.h declares; argument validation works without touching the device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "frcnn_hip.h")


@pytest.fixture(scope="module")
def L():
    so = os.path.join(ROOT, "faster_rcnn_pytorch_amd", "lib", "libfrcnn_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    from faster_rcnn_pytorch_amd import _lib
    return _lib


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound(L):
    decl = declared_symbols()
    assert len(decl) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH]).decode()
    exported = set(re.findall(r" T (frcnn_[a-z0-9_]+)", out))
    assert set(decl) <= exported, sorted(set(decl) - exported)
    assert exported <= set(decl), "exported but undeclared: %s" % sorted(exported - set(decl))
    assert set(L.SIGNATURES) == set(decl)


def test_abi_version_and_error_reporting(L):
    assert L.lib.frcnn_abi_version() == L.ABI_VERSION == 7
    rc = L.lib.frcnn_nms(None, None, 10, 0.5, 10, None, None, None, None, 0, None)      # NULL out_count
    assert rc == -1 and b"nms" in L.lib.frcnn_last_error()
    with pytest.raises(L.FrcnnError):
        L.check(L.lib.frcnn_box_codec(9, None, None, 1, None, None), "box_codec")
    rc = L.lib.frcnn_rpn_targets(0, None, 10, None, 0, None, 0, None, 0, 0, 0, None, None, None, None, None, 0, None)
    assert rc == -1 and b"G must be >= 1" in L.lib.frcnn_last_error()                    # the reference crashes on G = 0 too


def test_workspace_sizes(L):
    assert L.workspace_bytes(L.OP_NMS, 12000) >= 12000 * 188 * 8
    assert L.workspace_bytes(L.OP_TOPK, 20646) >= 20646 * 4
    assert L.workspace_bytes(L.OP_REGION_PROPOSAL, 20646, 12000) > L.workspace_bytes(L.OP_NMS, 12000)
    assert L.workspace_bytes(99, 1, 1) == 0


def test_host_side_anchor_bases_match_reference(L, golden):
    from faster_rcnn_pytorch_amd import ops
    assert np.array_equal(ops.anchor_base(), golden("anchors")["anchor_base"])          # anchor.py:15-32
    assert ops.tv_base_anchors(32.0).tolist() == [[-23, -11, 23, 11], [-16, -16, 16, 16], [-11, -23, 11, 23]]


def test_ops_refuse_cpu_tensors(L):
    import torch
    from faster_rcnn_pytorch_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.xy_to_cxcy(torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.nms(torch.zeros(4, 4), torch.zeros(4), 0.5)


def test_layout_stamp_of_every_object_agrees(L):
    """csrc/frcnn_layout.h: every translation unit registers a compile-time hash of the layouts it shares with the others (AnchorDesc,
    the sample sort's control block and plan); the library checks them at load."""
    assert L.lib.frcnn_layout_check() == 0
    assert L.lib.frcnn_layout_stamp() != 0


def test_a_stale_object_makes_the_library_refuse_to_load(tmp_path):
    """The class of fault behind round 3's 14:49 abort (docs/HISTORY.md): one object compiled against another version of a shared
    header.  Here topk.hip is compiled with a skewed stamp and linked with the other, current objects: frcnn_abi_version() must return
    FRCNN_ERR_UNSUPPORTED and name the object, and the Python binding must refuse the library (ImportError), before any kernel runs."""
    csrc = os.path.join(ROOT, "faster_rcnn_pytorch_amd", "csrc")
    obj = os.path.join(ROOT, "faster_rcnn_pytorch_amd", "lib", "obj")
    if not os.path.exists(os.path.join(obj, "api.o")):
        import __graft_entry__ as g
        g.build()
    flags = "--offload-arch=gfx950 -O1 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off".split()
    skew = str(tmp_path / "topk_skew.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DFRCNN_LAYOUT_TEST_SKEW=7", "-c", os.path.join(csrc, "topk.hip"), "-o", skew])
    objs = [skew if f == "topk.o" else os.path.join(obj, f) for f in sorted(os.listdir(obj)) if f.endswith(".o")]
    so = str(tmp_path / "libfrcnn_skew.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    lib = C.CDLL(so)
    lib.frcnn_last_error.restype = C.c_char_p
    assert lib.frcnn_layout_check() == -2 and lib.frcnn_abi_version() == -2           # FRCNN_ERR_UNSUPPORTED
    msg = lib.frcnn_last_error().decode()
    assert "'topk'" in msg and "stale object" in msg
    code = "import os; os.environ['FRCNN_HIP_LIB'] = %r\nfrom faster_rcnn_pytorch_amd import _lib" % so
    r = subprocess.run([os.sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "ImportError" in r.stderr and "refuses to load" in r.stderr and "'topk'" in r.stderr
