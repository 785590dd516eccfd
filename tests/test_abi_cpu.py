"""CPU-side checks of the boundary: the shared library loads, exports every symbol that
include/detgeom.h declares, the ctypes table covers them, argument validation works without a
GPU, and the product refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "detgeom.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from vision_amd import _abi
    lib = _abi.load()
    syms = declared_symbols()
    assert len(syms) >= 26
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in detgeom.h but not exported"
        assert s in _abi.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_abi.SIGNATURES) == set(syms)
    assert b"gfx950" in lib.dg_version()


def test_argument_validation_without_gpu():
    from vision_amd import _abi
    lib = _abi.load()
    # n == 0 is a no-op, negative sizes and null pointers are rejected before any launch
    assert lib.dg_iou_elementwise_f32(None, None, 0, 1e-7, None, None) == _abi.OK
    assert lib.dg_iou_elementwise_f32(None, None, -1, 1e-7, None, None) == _abi.EINVAL
    assert "n < 0" in _abi.last_error()
    assert lib.dg_iou_pairwise_f32(None, 4, None, 4, 1e-7, None, None) == _abi.EINVAL
    assert lib.dg_roipool_fwd_f32(None, 1, 8, 4, 4, None, 1, 0, 1, None, None, None, None) == _abi.EINVAL
    assert "output size" in _abi.last_error()
    assert lib.dg_roipool_fwd_f32(None, 1, 8, 4, 4, None, 1, 5, 1, None, None, None, None) == _abi.EINVAL      # null pointers
    assert lib.dg_roipool_fwd_a16_f32(None, 1, 8, 400, 400, None, 0, 7, 1, None, None, None, None) == _abi.OK   # n == 0
    assert lib.dg_kmeans_iou_step(None, 10, None, 17, None, None, None, 0, None) == _abi.EINVAL
    with pytest.raises(ValueError):
        _abi.check(_abi.EINVAL)
    # workspace sizing is pure host arithmetic
    assert lib.dg_nms_workspace_bytes(1, 0) > 0
    n64k = lib.dg_nms_workspace_bytes(1, 65536)
    assert n64k >= 65536 * 1024 * 8                       # the N x N/64 bitmask
    assert lib.dg_nms_workspace_bytes(256, 2000) >= 256 * 2048 * 32 * 8
    assert lib.dg_nms_batched_f32(None, None, None, 1, 128, 0.5, 1, None, None, None, 0, None) == _abi.EINVAL


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a host without a GPU")
def test_no_cpu_fallback():
    from vision_amd import _abi, yolo_v1, fast_rcnn, cluster
    import numpy as np
    with pytest.raises(_abi.DetGeomError):
        yolo_v1.get_iou(torch.zeros(2, 4), torch.zeros(2, 4))
    with pytest.raises(_abi.DetGeomError):
        yolo_v1.nms(torch.zeros(2, 4), torch.zeros(2), 0.3)
    with pytest.raises(_abi.DetGeomError):
        fast_rcnn.slowroipool((7, 7)).foward(torch.zeros(1, 4, 8, 8), np.array([[0, 0, 1, 1.0]]), np.array([0]))
    with pytest.raises(_abi.DetGeomError):
        cluster.kmean(5).fit(np.ones((10, 2), np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vision_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "liboracle" not in text, f
