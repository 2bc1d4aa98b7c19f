"""Test infrastructure (oracle/): Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) in
numpy, and the permutations the library's DEVICE sampling mode is equivalent to.  Pinned by the Random123 known-answer vectors
(tests/test_oracle_kat.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.

The library (csrc/frcnn_common.h: philox_first) draws, for element `index` of sampling stream `stream_id`, the FIRST output word of
Philox4x32-10 with counter (index, stream_id, offset_lo, offset_hi) and key (seed_lo, seed_hi), and keeps the candidates with the
smallest (key, position) pairs.  The reference samples with torch.randperm (models/model.py:225-236, 318-345): "keep perm[:k]" of a
candidate list; the device mode is therefore the reference's algorithm run with perm = argsort of the keys, which is what
sampling_perm() returns -- fed to the oracle's host-permutation path it yields the exact expected outputs."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: array [..., 4] of uint32, key: (k0, k1) -> array [..., 4] of uint32."""
    c = np.asarray(counter, dtype=np.uint64)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & _MASK
        n1 = p1 & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & _MASK
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], -1).astype(np.uint32)


def philox_first(seed, offset, stream_id, index):
    """First output word for every element of `index` (csrc/frcnn_common.h: philox_first)."""
    index = np.asarray(index, dtype=np.uint64)
    ctr = np.zeros(index.shape + (4,), np.uint64)
    ctr[..., 0] = index & _MASK
    ctr[..., 1] = np.uint64(stream_id)
    ctr[..., 2] = np.uint64(int(offset) & 0xFFFFFFFF)
    ctr[..., 3] = np.uint64((int(offset) >> 32) & 0xFFFFFFFF)
    return philox4x32_10(ctr, (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF))[..., 0]


def sampling_perm(seed, offset, stream_id, elements):
    """The permutation of range(len(elements)) (positions in the candidate list, which is in ascending element order) that the
    device mode is equivalent to: candidates ordered by (Philox key of the element, position)."""
    elements = np.asarray(elements, dtype=np.int64)
    keys = philox_first(seed, offset, stream_id, elements)
    return np.lexsort((np.arange(len(elements)), keys)).astype(np.int64)
