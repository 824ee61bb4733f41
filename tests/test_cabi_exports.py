"""CPU: the C-ABI library loads and exports every symbol include/retinanet_mi355x.h declares (no compute)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "retinanet_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from retinanet_mi355x import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(_hip.SIGNATURES), set(names) ^ set(_hip.SIGNATURES)


def test_host_helpers_without_gpu():
    """rn_anchor_count / rn_anchor_base_boxes are host code: check them against the oracle on CPU."""
    import numpy as np
    from oracle import anchors as oanchors
    from retinanet_mi355x import _hip
    lib = _hip.load()
    assert lib.rn_anchor_count(1080, 1920) == 389205
    assert lib.rn_anchor_count(512, 512) == oanchors.num_anchors(512, 512)
    buf = (ctypes.c_double * 180)()
    lib.rn_anchor_base_boxes(ctypes.cast(buf, ctypes.c_void_p))
    got = np.frombuffer(buf, dtype=np.float64).reshape(5, 9, 4)
    for li, lvl in enumerate(oanchors.PYRAMID_LEVELS):
        assert np.array_equal(got[li], oanchors.base_boxes(2 ** (lvl + 2)))       # bit-exact fp64
    assert lib.rn_version().startswith(b"retinanet_mi355x")


def test_ops_refuse_cpu_tensors():
    import torch
    from retinanet_mi355x import ops
    with pytest.raises(RuntimeError):
        ops.decode_dir(torch.zeros(1, 4, 4), torch.zeros(1, 4, 12))
    with pytest.raises(RuntimeError):
        ops.anchors(64, 64, "cpu")
