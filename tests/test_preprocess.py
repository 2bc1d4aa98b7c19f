"""Input stage.  CPU: the oracle's restatement of Pillow's resampler + ToTensor/Normalize/box
transforms against golden vectors generated with Pillow and torch (tests/golden/make_golden_preprocess.py), and the host
size logic against hand-derived answers.  GPU: the HIP kernels against the oracle and the same goldens, bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import oracle as orc

SMALL = ("up", "down", "down_flip", "same_w", "strong_down", "tall")
LARGE = ("voc", "coco", "wide")


@pytest.fixture(scope="module")
def pre():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz"))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


# ------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_pillow_and_torch_golden(pre, name):
    h, w, oh, ow, flip = pre[name + "_meta"]
    u8, f = orc.preprocess_image(pre[name + "_img"], (oh, ow), None, bool(flip))
    assert np.array_equal(u8, pre[name + "_u8"])                                  # Pillow's uint8 result: bit-exact
    assert np.array_equal(f, pre[name + "_f32"])                                  # to_tensor + normalize: bit-exact
    b = orc.preprocess_boxes(pre[name + "_boxes"], (w, h), (ow, oh), bool(flip))
    assert np.array_equal(b, pre[name + "_boxes_out"])


@pytest.mark.parametrize("name", LARGE)
def test_oracle_matches_pillow_digest_full_size(pre, name):
    h, w, oh, ow, flip, seed = pre[name + "_meta"]
    img = np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)
    u8, f = orc.preprocess_image(img, (oh, ow), None, bool(flip))
    assert _sha(u8) == pre[name + "_sha_u8"].tobytes() and _sha(f) == pre[name + "_sha_f32"].tobytes()


def test_oracle_zero_pad():
    img = np.random.RandomState(0).randint(0, 256, (20, 30, 3)).astype(np.uint8)
    _, f = orc.preprocess_image(img, (40, 60), (64, 64))
    _, g = orc.preprocess_image(img, (40, 60))
    assert np.array_equal(f[:, :40, :60], g) and (f[:, 40:] == 0).all() and (f[:, :, 60:] == 0).all()


def test_size_logic_known_answers():
    from faster_rcnn_pytorch_amd import transforms as T
    # (w, h) -> (oh, ow), worked by hand from transforms.py:79-99
    assert T.get_size_with_aspect_ratio((640, 480), 800, 1333) == (800, 1066)      # 800 * 640 / 480 = 1066.67 -> int
    assert T.get_size_with_aspect_ratio((500, 375), 800, 1333) == (800, 1066)
    assert T.get_size_with_aspect_ratio((480, 640), 800, 1333) == (1066, 800)
    assert T.get_size_with_aspect_ratio((1000, 300), 800, 1333) == (400, 1333)     # 1000/300*800 > 1333 -> size = round(399.9) = 400; ow = int(400*1000/300)
    assert T.get_size_with_aspect_ratio((800, 600), 600, 1000) == (600, 800)       # shorter side already == size
    assert T.get_size_with_aspect_ratio((333, 500), 800, 1333) == (1201, 800)
    assert T.get_size((640, 480), (300, 200)) == (200, 300)                         # explicit (w, h) -> (h, w)
    assert T.padded_size(800, 1066) == (800, 1088) and T.padded_size(800, 1344) == (800, 1344) and T.padded_size(1, 1) == (32, 32)


# ------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from faster_rcnn_pytorch_amd import transforms
    return transforms


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL)
def test_hip_matches_pillow_golden(T, pre, name):
    h, w, oh, ow, flip = (int(v) for v in pre[name + "_meta"])
    img = torch.from_numpy(pre[name + "_img"]).cuda()
    f, u8 = T.preprocess_image(img, (oh, ow), None, bool(flip), want_u8=True)
    assert np.array_equal(u8.cpu().numpy(), pre[name + "_u8"])
    assert np.array_equal(f.cpu().numpy(), pre[name + "_f32"])
    b = T.preprocess_boxes(torch.from_numpy(pre[name + "_boxes"]).cuda(), (w, h), (ow, oh), bool(flip))
    assert np.array_equal(b.cpu().numpy(), pre[name + "_boxes_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", LARGE)
def test_hip_full_size_matches_pillow_digest_and_oracle(T, pre, name):
    h, w, oh, ow, flip, seed = (int(v) for v in pre[name + "_meta"])
    img = np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)
    ph, pw = T.padded_size(oh, ow)
    f, u8 = T.preprocess_image(torch.from_numpy(img).cuda(), (oh, ow), (ph, pw), bool(flip), want_u8=True)
    assert _sha(u8.cpu().numpy()) == pre[name + "_sha_u8"].tobytes()
    _, fo = orc.preprocess_image(img, (oh, ow), (ph, pw), bool(flip))
    assert np.array_equal(f.cpu().numpy(), fo)
    assert _sha(f[:, :oh, :ow].cpu().numpy()) == pre[name + "_sha_f32"].tobytes()


@pytest.mark.gpu
def test_device_input_stage_end_to_end(T):
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    boxes = (rng.rand(7, 4) * np.array([640, 480, 640, 480])).astype(np.float32)
    stage = T.DeviceInputStage()
    x, b, info = stage(torch.from_numpy(img).cuda(), torch.from_numpy(boxes).cuda(), flip=True)
    assert info == {"size": (800, 1066), "padded": (800, 1088), "orig_size": (480, 640)} and x.shape == (1, 3, 800, 1088)
    _, fo = orc.preprocess_image(img, (800, 1066), (800, 1088), True)
    assert np.array_equal(x[0].cpu().numpy(), fo)
    assert np.array_equal(b.cpu().numpy(), orc.preprocess_boxes(boxes, (640, 480), (1066, 800), True))
    with pytest.raises(RuntimeError):
        T.preprocess_image(torch.from_numpy(img), (800, 1066))                     # CPU tensor: refused, no fallback
    with pytest.raises(ValueError):
        T.preprocess_image(torch.zeros(3, 8, 8, dtype=torch.uint8).cuda(), (8, 8))  # CHW is not accepted silently
