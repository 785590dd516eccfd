"""The allocation-free NMS entry point (dg_nms_batched_noalloc_f32: workspace = a device array inside the library)
against dg_nms_batched_f32 with a caller workspace, through the C ABI: two calls in a row, the second smaller,
keep lists and counts bit-identical; sizes beyond the built-in buffer are refused.  Collected last (test_zz_*)."""
import numpy as np
import pytest
import torch

from tools import workloads as wl

pytestmark = pytest.mark.gpu


def _both(lib, _abi, ops, boxes, scores, counts, thr, quirk):
    dev = boxes.device
    B, N = scores.shape
    st = _abi.stream_ptr(dev)
    out = []
    for noalloc in (False, True):
        keep = torch.full((B, N), -7, dtype=torch.int64, device=dev)
        kc = torch.full((B,), -7, dtype=torch.int32, device=dev)
        cp = _abi.ptr(counts) if counts is not None else None
        if noalloc:
            rc = lib.dg_nms_batched_noalloc_f32(_abi.ptr(boxes), _abi.ptr(scores), cp, B, N, thr, quirk, _abi.ptr(keep), _abi.ptr(kc), st)
        else:
            nbytes = lib.dg_nms_workspace_bytes(B, N)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            rc = lib.dg_nms_batched_f32(_abi.ptr(boxes), _abi.ptr(scores), cp, B, N, thr, quirk, _abi.ptr(keep), _abi.ptr(kc),
                                        _abi.ptr(ws), nbytes, st)
        _abi.check(rc)
        torch.cuda.synchronize()
        out.append((keep.cpu().numpy(), kc.cpu().numpy()))
    return out


def test_nms_noalloc_matches_workspace_entry_point():
    from oracle import oracle as orc
    from vision_amd import _abi, ops
    orc.build()
    lib = _abi.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert lib.dg_nms_noalloc_max_bytes() == 16 << 20
    # first call: 6 ragged images of up to 2000 boxes; second: smaller (3 x 700), other threshold; third: one 9000-box image
    cases = [(6, 2000, 0.7, 1, True), (3, 700, 0.3, 1, False), (1, 9000, 0.5, 0, False), (2, 40, 0.5, 1, False)]
    for B, N, thr, quirk, ragged in cases:
        bs = [wl.frcnn_proposals(100 + i, N) for i in range(B)]
        boxes = torch.from_numpy(np.stack([b for b, _ in bs])).to(dev)
        scores = torch.from_numpy(np.stack([s for _, s in bs])).to(dev)
        counts = torch.tensor([N - 137 * i for i in range(B)], dtype=torch.int32, device=dev) if ragged else None
        (k0, c0), (k1, c1) = _both(lib, _abi, ops, boxes, scores, counts, thr, quirk)
        assert np.array_equal(c0, c1), (B, N, c0, c1)
        assert c0.min() >= 0
        for i in range(B):
            assert np.array_equal(k0[i, :c0[i]], k1[i, :c1[i]]), (B, N, i)
            ni = N - 137 * i if ragged else N                     # and both against the oracle (yolo/yolo_v1/yolo.py:188-203)
            want = orc.nms(bs[i][0][:ni], bs[i][1][:ni], thr, bool(quirk))
            assert np.array_equal(k1[i, :c1[i]], want), (B, N, i)


def test_nms_noalloc_rejects_oversize():
    from vision_amd import _abi
    lib = _abi.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert lib.dg_nms_workspace_bytes(1, 20000) > lib.dg_nms_noalloc_max_bytes()
    t = torch.zeros(16, device=dev)
    rc = lib.dg_nms_batched_noalloc_f32(_abi.ptr(t), _abi.ptr(t), None, 1, 20000, 0.5, 1, _abi.ptr(t), _abi.ptr(t), _abi.stream_ptr(dev))
    assert rc == _abi.ENOMEM
    assert "built-in buffer" in _abi.last_error()


def test_nms_noalloc_under_graph_capture():
    """The callers the entry point is for: a HIP graph capture, where hipMalloc is not allowed.  Captured once, replayed
    on new boxes written into the same buffers; sparse and dense boxes (both reduce kernels)."""
    from oracle import oracle as orc
    from vision_amd import _abi, ops
    orc.build()
    lib = _abi.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for n, dense, thr in ((2000, False, 0.5), (3000, True, 0.5), (6000, False, 0.5)):
        boxes = torch.zeros((1, n, 4), dtype=torch.float32, device=dev)
        scores = torch.zeros((1, n), dtype=torch.float32, device=dev)
        keep = torch.zeros((1, n), dtype=torch.int64, device=dev)
        kc = torch.zeros(1, dtype=torch.int32, device=dev)

        def call():
            rc = lib.dg_nms_batched_noalloc_f32(_abi.ptr(boxes), _abi.ptr(scores), None, 1, n, thr, 1, _abi.ptr(keep), _abi.ptr(kc),
                                                _abi.stream_ptr(dev))
            assert rc == 0, _abi.last_error()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            call()                                           # first call on the device looks the scratch symbol up: outside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            call()
        for seed in (5, 6):
            b, s = wl.stress_boxes(seed, n, dense=dense)
            boxes.copy_(torch.from_numpy(b)[None]); scores.copy_(torch.from_numpy(s)[None])
            g.replay()
            torch.cuda.synchronize()
            k1, c1 = ops.nms_batched(boxes, scores, None, thr, True)
            assert int(kc.item()) == int(c1.item())
            assert torch.equal(keep[0, :int(kc.item())], k1[0, :int(c1.item())])
            assert np.array_equal(keep[0, :int(kc.item())].cpu().numpy(), orc.nms(b, s, thr, True))
