"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Bars (BASELINE.json north_star): NMS keep indices, RoI argmax, k-means
assignments bit-exact; IoU / pooled values / centroids within 1e-6 relative (we assert bit-exact
where the arithmetic is IEEE-identical and say so)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tools import workloads as wl  # noqa: E402


@pytest.fixture(scope="module")
def va():
    import vision_amd
    from vision_amd import _abi, ops, yolo_v1, yolo_v2, rcnn_utils, fast_rcnn, cluster
    _abi.load()
    assert torch.cuda.is_available()
    return vision_amd


def test_single_hip_runtime(va):
    """libdetgeom must share torch's HIP runtime (one libamdhip64 mapped), or streams would not be valid."""
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(libs) == 1, libs
    assert any("libdetgeom.so" in line for line in open("/proc/self/maps"))


# ------------------------------------------------------------------ IoU
def test_iou_elementwise_golden(va, golden):
    from vision_amd import yolo_v1, yolo_v2
    g = golden("iou_tensor")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    got = yolo_v1.get_iou(a, b)
    assert got.dtype == torch.float32 and not got.is_cuda
    assert np.array_equal(got.numpy(), g["iou_eps1e7"])          # bit-exact
    assert np.array_equal(yolo_v2.get_iou(a.cuda(), b.cuda()).cpu().numpy(), g["iou_eps1e7"])
    assert np.array_equal(yolo_v1.Loss(2, 5).get_iou(a, b).numpy(), g["iou_eps1e6"])
    got64 = yolo_v1.get_iou(a.double(), b.double())
    assert got64.dtype == torch.float64
    assert np.array_equal(got64.numpy(), g["iou_f64"])
    nan = yolo_v1.get_iou(torch.from_numpy(g["a_nan"]), torch.from_numpy(g["b_nan"])).numpy()
    assert np.array_equal(nan, g["iou_nan"], equal_nan=True)
    assert yolo_v1.get_iou(torch.zeros(0, 4), torch.zeros(0, 4)).shape == (0,)


@pytest.mark.parametrize("n,m", [(1, 1), (63, 65), (64, 1024), (130, 1027), (257, 2052)])
def test_iou_pairwise_vs_oracle(va, oracle, n, m):
    from vision_amd import ops
    a, b = wl.random_boxes(n, n), wl.random_boxes(1000 + m, m)
    got = ops.iou_pairwise(torch.from_numpy(a), torch.from_numpy(b)).cpu().numpy()
    assert np.array_equal(got, oracle.iou_pairwise(a, b))        # bit-exact


def test_iou_scalar_golden(va, golden):
    from vision_amd import rcnn_utils
    g = golden("iou_scalar")
    iou, flags = rcnn_utils.iou_many(g["rois"], g["grt"])
    assert np.array_equal(iou, g["iou"]) and not flags.any()
    assert rcnn_utils.get_iou([0, 0, 10, 10], np.array([10.0, 0, 20, 10])) == 0.0
    assert rcnn_utils.get_iou(list(g["rois"][0]), g["grt"]) == g["iou"][0]
    for bb in g["bad_boxes"]:
        with pytest.raises(AssertionError):
            rcnn_utils.get_iou(bb, g["grt"])


# ------------------------------------------------------------------ NMS
def test_nms_golden(va, golden):
    from vision_amd import yolo_v1
    g = golden("nms")
    names = sorted(k[:-5] for k in g.files if k.endswith("_keep"))
    for name in names:
        keep = yolo_v1.nms(torch.from_numpy(g[name + "_boxes"]), torch.from_numpy(g[name + "_scores"]), float(g[name + "_thr"]))
        assert keep.dtype == torch.int64
        assert np.array_equal(keep.numpy(), g[name + "_keep"]), name
    assert yolo_v1.nms([], [], 0.3).numel() == 0


@pytest.mark.parametrize("n,dense,thr", [(4096, True, 0.5), (4097, False, 0.5), (10000, False, 0.3), (20000, True, 0.7)])
def test_nms_large_vs_oracle(va, oracle, n, dense, thr):
    """exercises the global bitonic sort and the 1024-thread reduce"""
    from vision_amd import yolo_v1
    b, s = wl.stress_boxes(n, n, dense=dense)
    keep = yolo_v1.nms(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), thr)
    assert keep.is_cuda
    assert np.array_equal(keep.cpu().numpy(), oracle.nms(b, s, thr)), (n, dense, thr)


@pytest.mark.parametrize("B,C,per,a16", [(16, 256, 40, True), (64, 72, 12, False), (64, 72, 12, True), (1, 6, 300, False), (1, 6, 300, True),
                                         (1, 64, 300, True)])
def test_roipool_backward_slice_shapes_vs_oracle(va, oracle, B, C, per, a16):
    """The max backward's instantiations that the other tests' shapes do not reach (a kernel trace of the test run against the
    compiled instantiations found them): 16-channel slices with a uint16 argmax (B * C / 16 >= 256), 8-channel slices (a channel
    count that is a multiple of 8 but not of 16, B * C / 8 >= 512) with either argmax width, and the single-image kernel that takes
    the RoIs in order itself on a channel count that rules out the 16-byte path, and with the uint16 argmax."""
    from vision_amd import ops
    rng = np.random.default_rng(B * 1000 + C)
    H, W = 38, 50
    n = B * per
    img = rng.standard_normal((B, C, H, W), dtype=np.float32)
    rois = np.concatenate([wl.frcnn_rois(100 + i, per) for i in range(B)])
    idx = np.repeat(np.arange(B), per).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, B, H, W, x.device)
    assert int(status.item()) == 0
    out, arg = ops.roipool_forward(x, win, "max", argmax_dtype=torch.int16 if a16 else torch.int32)
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal((ops.argmax_i32(arg) if a16 else arg).cpu().numpy(), warg)
    gout = rng.standard_normal((n, C, 7, 7), dtype=np.float32)
    gin = ops.roipool_backward(torch.from_numpy(gout).cuda(), arg, win, img.shape, "max")
    wgin = oracle.roipool_bwd(gout, warg, img.shape, rois, idx, 7, "max")
    np.testing.assert_allclose(gin.cpu().numpy(), wgin, rtol=1e-4, atol=1e-4)          # fp32 sums in a different order


def test_abi_entry_points_the_wrappers_do_not_use(va, oracle):
    """Entry points of include/detgeom.h that vision_amd's own wrappers route around (tools/abi_coverage.py lists what a test run
    reached): the NMS phases called one after the other (what bench.py times; the same keep list as the one-call entry point and as
    the oracle, on the pruned path, the dense path and a small image), dg_roipool_fwd_f32 / dg_roipool_bwd_f32 (the 7 x 7, int32-argmax
    entry points; the wrappers go through the *_any_* ones) against the oracle, dg_version, dg_comm_rank / dg_comm_size."""
    from vision_amd import ops, _abi, cluster
    lib = _abi.load()
    dev = torch.device("cuda", 0)
    sp = _abi.stream_ptr(dev)
    assert b"gfx950" in lib.dg_version()
    for n, dense, thr in ((20000, False, 0.5), (9000, True, 0.3), (2000, False, 0.7), (300, True, 0.5)):
        b, s = wl.stress_boxes(3 + n, n, dense=dense)
        bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
        ws, nb = ops.nms_workspace(dev, 1, n)
        keep = torch.full((1, n), -1, dtype=torch.int64, device=dev); kc = torch.zeros((1,), dtype=torch.int32, device=dev)
        _abi.check(lib.dg_nms_phase_sort(_abi.ptr(bt), _abi.ptr(st), None, 1, n, _abi.ptr(ws), nb, sp))
        for _ in range(2):                                       # the later phases may be run again on a sorted workspace
            _abi.check(lib.dg_nms_phase_mask(None, 1, n, thr, _abi.ptr(ws), nb, sp))
            _abi.check(lib.dg_nms_phase_reduce(None, 1, n, 1, _abi.ptr(keep), _abi.ptr(kc), _abi.ptr(ws), nb, sp))
        want = oracle.nms(b, s, thr, True)
        assert int(kc[0]) == len(want) and np.array_equal(keep[0, :len(want)].cpu().numpy(), want), (n, dense, thr)
    # 7 x 7 max pooling with int32 argmax through the dedicated entry points
    rng = np.random.default_rng(4)
    B, Cn, H, W, n = 2, 128, 38, 50, 300
    img = rng.standard_normal((B, Cn, H, W), dtype=np.float32)
    rois = np.concatenate([wl.frcnn_rois(i, n // B) for i in range(B)])
    idx = np.repeat(np.arange(B), n // B).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, B, H, W, dev)
    nhwc, nanf = ops.to_nhwc(x)
    out = torch.empty((n, Cn, 7, 7), dtype=torch.float32, device=dev); arg = torch.empty((n, Cn, 7, 7), dtype=torch.int32, device=dev)
    _abi.check(lib.dg_roipool_fwd_f32(_abi.ptr(nhwc), B, Cn, H, W, _abi.ptr(win), n, 7, _abi.POOL_MAX, _abi.ptr(nanf), _abi.ptr(out), _abi.ptr(arg), sp))
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(arg.cpu().numpy(), warg)
    gout = torch.from_numpy(rng.standard_normal((n, Cn, 7, 7), dtype=np.float32)).cuda()
    gin = torch.full((B, Cn, H, W), 7.0, dtype=torch.float32, device=dev)          # overwritten, not accumulated
    nbw = lib.dg_roipool_bwd_workspace_bytes(B, n)
    wsb = ops.workspace(dev, nbw)
    _abi.check(lib.dg_roipool_bwd_f32(_abi.ptr(gout), _abi.ptr(arg), _abi.ptr(win), n, B, Cn, H, W, 7, _abi.POOL_MAX, _abi.ptr(gin), _abi.ptr(wsb), nbw, sp))
    wgin = oracle.roipool_bwd(gout.cpu().numpy(), warg, img.shape, rois, idx, 7, "max")
    np.testing.assert_allclose(gin.cpu().numpy(), wgin, rtol=1e-4, atol=1e-4)          # fp32 sums in a different order
    # a communicator's rank and size (a single-rank RCCL group, made without a process group)
    comm = cluster.DgComm.create(None, dev)
    try:
        assert lib.dg_comm_rank(comm.handle) == 0 and lib.dg_comm_size(comm.handle) == 1
    finally:
        comm.close()


@pytest.mark.parametrize("n", [300, 2000, 9000])
def test_nms_counts_outside_the_batch_are_clamped(va, oracle, n):
    """counts[i] is held to [0, nmax] inside the kernels: a negative count is an empty image, one beyond nmax the whole row of
    boxes -- never a read outside the image's boxes (small, mid-size and pruned-path images)."""
    from vision_amd import ops
    B = 4
    boxes = np.zeros((B, n, 4), np.float32); scores = np.zeros((B, n), np.float32)
    for i in range(B):
        boxes[i], scores[i] = wl.stress_boxes(40 + i, n, dense=bool(i & 1))
    counts = np.array([n + 12345, -7, n // 2, 2147483647], np.int32)
    keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts), 0.5, True)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for i, c in enumerate((n, 0, n // 2, n)):
        want = oracle.nms(boxes[i, :c], scores[i, :c], 0.5, True) if c else np.zeros(0, np.int64)
        assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (n, i)


@pytest.mark.parametrize("n,dense,thr", [(70001, False, 0.5), (100000, True, 0.6), (270000, True, 0.3)])
def test_nms_beyond_65536_boxes_vs_oracle(va, oracle, n, dense, thr):
    """Images of more than 65536 boxes: the global bitonic network, the gather, the dense mask and the serial chain over more than
    1024 row blocks (the pruned path and the chunk sort end at 65536); beyond 262144 boxes the chain's widest instantiation
    (sixteen words of `removed` per thread, an 8.6 GB mask)."""
    from vision_amd import ops
    b, s = wl.stress_boxes(5, n, dense=dense)
    keep, kc = ops.nms_batched(torch.from_numpy(b).cuda()[None], torch.from_numpy(s).cuda()[None], None, thr, True)
    assert np.array_equal(keep[0, :int(kc[0])].cpu().numpy(), oracle.nms(b, s, thr, True))


@pytest.mark.parametrize("switch", [("DG_NMS_FP_SYNC", "1"), ("DG_NMS_BAND_PARTS", "1"), ("DG_NMS_BAND_PARTS", "5")])
def test_nms_pruned_path_switches_vs_oracle(va, oracle, switch, monkeypatch):
    """The pruned path's opt-in variants: the fixed point with a grid barrier per round (k_nms_resolve_mc, DG_NMS_FP_SYNC=1; the
    default is the free-running k_nms_resolve_async) and other splits of a row tile's band over workgroups (DG_NMS_BAND_PARTS);
    ragged batches of sparse and dense sets, both quirk settings, against the oracle."""
    from vision_amd import ops
    monkeypatch.setenv(*switch)
    for n, dense, thr in ((5000, False, 0.5), (20000, True, 0.6), (9000, True, 0.3), (33000, False, 0.4)):
        B = 2
        counts = np.array([n, n - 1234], np.int32)
        boxes = np.zeros((B, n, 4), np.float32); scores = np.zeros((B, n), np.float32)
        for i in range(B):
            boxes[i], scores[i] = wl.stress_boxes(77 + n + i, n, dense=dense)
        for quirk in (True, False):
            keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts), thr, quirk)
            keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
            for i in range(B):
                want = oracle.nms(boxes[i, :counts[i]], scores[i, :counts[i]], thr, quirk)
                assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (switch, n, thr, quirk, i)


@pytest.mark.parametrize("dense", [False, True])
def test_nms_band_reduce_forced_vs_oracle(va, oracle, dense, monkeypatch):
    """k_nms_reduce_band (the reduce of images whose compact mask form overflows) on every kind of input: with
    DG_NMS_FORCE_FALLBACK=1 no image "fits" the compact form, so sparse and dense, ragged, single-block and multi-super-block
    images all go through it; both quirk settings; against the oracle."""
    from vision_amd import ops
    monkeypatch.setenv("DG_NMS_FORCE_FALLBACK", "1")
    for n, thr in ((449, 0.5), (513, 0.3), (1100, 0.7), (4097, 0.5), (9000, 0.3), (20000, 0.6)):
        B = 3
        counts = np.array([n, max(n - 70, 1), min(450, n - 1)], np.int32)
        boxes = np.zeros((B, n, 4), np.float32); scores = np.zeros((B, n), np.float32)
        for i in range(B):
            boxes[i], scores[i] = wl.stress_boxes(1000 + n + i, n, dense=dense)
        for quirk in (True, False):
            keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts), thr, quirk)
            keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
            for i in range(B):
                want = oracle.nms(boxes[i, :counts[i]], scores[i, :counts[i]], thr, quirk)
                assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (n, thr, quirk, i)


@pytest.mark.parametrize("fb", ["0", "2"])
def test_nms_small_image_rounds_and_serial_vs_oracle(va, oracle, fb, monkeypatch):
    """k_nms_resolve_small (images of 449 .. 4096 boxes: the non-zero mask words in LDS, the boxes decided in rounds): sparse
    and dense sets, ragged batches, both quirk settings, a chain that needs hundreds of rounds (so the round limit hands over to
    the serial chain), masks with more non-zero words than the registers hold (the same rounds with the words streamed from
    the mask), and with DG_NMS_FORCE_FALLBACK=2 every image through the in-kernel serial chain; against the oracle."""
    from vision_amd import ops
    monkeypatch.setenv("DG_NMS_FORCE_FALLBACK", fb)
    cases = []
    for n, dense, thr in ((449, False, 0.5), (700, True, 0.3), (2000, False, 0.7), (2000, True, 0.3), (4096, True, 0.1), (4096, False, 0.6)):
        bb = np.stack([wl.stress_boxes(500 + n + i, n, dense=dense)[0] for i in range(3)])
        ss = np.stack([wl.stress_boxes(500 + n + i, n, dense=dense)[1] for i in range(3)])
        cases.append((bb, ss, np.array([n, max(n - 70, 1), min(450, n - 1)], np.int32), thr))
    b, s = wl.frcnn_proposals(1, 2000)
    cases.append((b[None], s[None], None, 0.7)); cases.append((b[None], s[None], None, 0.3))
    m = 3000                                                     # a ladder: every box suppresses exactly the next one
    x = np.arange(m, dtype=np.float32) * 0.45
    lad = np.stack([x, np.zeros(m, np.float32), x + 1.0, np.ones(m, np.float32)], 1)
    cases.append((lad[None], np.linspace(1, 0.1, m, dtype=np.float32)[None], None, 0.3))
    for boxes, scores, counts, thr in cases:
        for quirk in (True, False):
            keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(),
                                        torch.from_numpy(counts) if counts is not None else None, thr, quirk)
            keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
            for i in range(boxes.shape[0]):
                ni = counts[i] if counts is not None else boxes.shape[1]
                want = oracle.nms(boxes[i, :ni], scores[i, :ni], thr, quirk)
                assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (boxes.shape, thr, quirk, i)


@pytest.mark.parametrize("prune,rounds", [("1", None), ("2", None), ("2", "0"), ("2", "1"), ("2", "3"), ("0", None)])
def test_nms_pruned_path_vs_oracle(va, oracle, prune, rounds, monkeypatch):
    """The pruned path of large images (x-bucket band pair search, 64 pair lists by target rank, fixed-point keep list on 64
    workgroups) and its fallbacks: DG_NMS_PRUNE=2 takes it whatever the candidate count, DG_NMS_FP_ROUNDS cuts the fixed point
    off after that many rounds (the serial chain over the grouped pairs finishes), DG_NMS_PRUNE=0 = the dense mask only.
    Sparse and dense sets, degenerate boxes, ragged batches, a ladder (one round per box), identical boxes (every pair
    suppresses); against the oracle."""
    from vision_amd import ops
    monkeypatch.setenv("DG_NMS_PRUNE", prune)
    if rounds is not None:
        monkeypatch.setenv("DG_NMS_FP_ROUNDS", rounds)
    cases = []
    for n, dense in ((20000, False), (9000, True), (4200, False), (5000, True)):
        b, s = wl.stress_boxes(3 + n, n, dense=dense)
        cases.append((b[None], s[None], None))
    cases.append((wl.random_boxes(5, 9000)[None], wl.tiefree_scores(6, 9000)[None], None))
    bb = np.stack([wl.stress_boxes(40 + i, 6000)[0] for i in range(3)]); ss = np.stack([wl.stress_boxes(40 + i, 6000)[1] for i in range(3)])
    cases.append((bb, ss, np.array([6000, 4500, 4100], np.int32)))
    m = 5000
    x = np.arange(m, dtype=np.float32) * 0.45
    cases.append((np.stack([x, np.zeros(m, np.float32), x + 1.0, np.ones(m, np.float32)], 1)[None], np.linspace(1, 0.1, m, dtype=np.float32)[None], None))
    cases.append((np.tile(np.array([[0.1, 0.1, 0.5, 0.5]], np.float32), (4500, 1))[None], wl.tiefree_scores(9, 4500)[None], None))
    nf = wl.stress_boxes(8, 6000)[0].copy(); nf[17, 2] = np.inf; nf[3000] = np.nan                 # non-finite boxes: dense mask
    cases.append((nf[None], wl.tiefree_scores(10, 6000)[None], None))
    for boxes, scores, counts in cases:
        bt, st = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        ct = torch.from_numpy(counts) if counts is not None else None
        for thr in (0.3, 0.6):
            for quirk in (True, False):
                keep, cnt = ops.nms_batched(bt, st, ct, thr, quirk)
                keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
                info = ops.nms_path_info(bt.device, boxes.shape[0], boxes.shape[1])
                if prune == "0" or not np.isfinite(boxes).all():
                    assert info[0] == 1, info
                elif prune == "2":
                    lcap = (-(-boxes.shape[1] // 64)) ** 2                          # entries one of the image's 64 pair lists holds
                    assert info[0] == 0 or info[3] > lcap - 4, info                 # pruned unless a pair list overflowed
                    if info[0] == 0 and rounds == "0" and info[3] > 0:
                        assert info[1] == 0, info                                   # the serial chain finished the image
                for i in range(boxes.shape[0]):
                    ni = counts[i] if counts is not None else boxes.shape[1]
                    want = oracle.nms(boxes[i, :ni], scores[i, :ni], thr, quirk)
                    assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (boxes.shape, thr, quirk, i, info)


def test_nms_band_narrowed_by_threshold_vs_oracle(va, oracle, monkeypatch):
    """k_nms_band only looks at partners whose left edge lies within (1 - thr) of a box's width (IoU <= ix / w): sets where that
    limit is delicate -- coordinates far from the origin (the limit's rounding), boxes too thin for a normal area (full band
    kept), thresholds close to 0 and to 1, exactly 1 and above, exact IoU values of the set, everything in one x-bucket, integer
    grids with many equal left edges, inverted boxes; the pruned path forced (DG_NMS_PRUNE=2) and free; against the oracle."""
    from vision_amd import ops
    rng = np.random.default_rng(77)
    n = 5200
    sets = []
    b0, s0 = wl.stress_boxes(91, n, dense=True)
    sets.append(("dense", b0, s0))
    far = b0.copy(); far[:, [0, 2]] += np.float32(4096.0); far[:, [1, 3]] -= np.float32(300.0)
    sets.append(("far from the origin", far, s0))
    thin = wl.stress_boxes(92, n, dense=False)[0].copy()
    thin[::3, 2] = thin[::3, 0] + np.float32(1e-12); thin[1::3, 3] = thin[1::3, 1] + np.float32(1e-30)
    sets.append(("thin", thin, s0))
    xy = rng.integers(0, 60, (n, 2)).astype(np.float32); wh = rng.integers(1, 12, (n, 2)).astype(np.float32)
    grid = np.concatenate([xy, xy + wh], 1)
    sets.append(("integer grid", grid, s0))
    one = b0.copy(); one[:, 0] = np.float32(0.25); one[:, 2] = one[:, 0] + rng.uniform(0.01, 0.3, n).astype(np.float32)
    sets.append(("one left edge", one, s0))
    inv = wl.random_boxes(93, n).copy(); inv[::7] = inv[::7][:, [2, 1, 0, 3]]; inv[3::11] = inv[3::11][:, [0, 3, 2, 1]]
    sets.append(("inverted", inv, s0))
    near = b0[rng.integers(0, 400, n)] + rng.uniform(-0.004, 0.004, (n, 4)).astype(np.float32)      # heavy clusters of near-duplicates
    sets.append(("near-duplicates", near.astype(np.float32), s0))
    for prune in ("2", "1"):
        monkeypatch.setenv("DG_NMS_PRUNE", prune)
        for name, b, s in sets:
            vals = np.unique(oracle.iou_pairwise(b[:48], b[:512]))
            vals = vals[(vals > 0.02) & (vals < 1)]
            thrs = [0.05, 0.5, 0.7, 0.95, 0.999, 1.0, 1.5]
            if len(vals):
                v = vals[rng.integers(0, len(vals))]
                thrs += [float(v), float(np.nextafter(v, np.float32(2)))]
            bt, st = torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda()
            for thr in thrs:
                keep, cnt = ops.nms_batched(bt, st, None, thr, True)
                got = keep[0, :int(cnt[0])].cpu().numpy()
                want = oracle.nms(b, s, thr, True)
                assert np.array_equal(got, want), (name, prune, thr, ops.nms_path_info(bt.device, 1, n), got.size, len(want))


def test_nms_threshold_edges_vs_oracle(va, oracle):
    """The mask phase tests inter/den < thr without dividing: thresholds that are exactly an IoU value of the set,
    its float neighbours, 0, negatives, inf and NaN must all give the oracle's (dividing) answer."""
    from vision_amd import ops
    rng = np.random.default_rng(11)
    for n in (700, 4500):                                  # one-wave reduce and the summary-driven reduce
        xy = rng.integers(0, 40, (n, 2)).astype(np.float32)
        wh = rng.integers(1, 9, (n, 2)).astype(np.float32)
        b = np.concatenate([xy, xy + wh], 1)
        s = wl.tiefree_scores(3, n)
        vals = np.unique(oracle.iou_pairwise(b[:64], b[:256]))
        vals = vals[(vals > 0) & (vals < 1)]
        pick = vals[rng.choice(len(vals), 6, replace=False)]
        thrs = []
        for v in pick:
            thrs += [v, np.nextafter(v, np.float32(2)), np.nextafter(v, np.float32(-1))]
        thrs += [np.float32(0), np.float32(-0.25), np.float32(1), np.float32(np.inf), np.float32(np.nan), np.float32(1e-45)]
        bt, st = torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda()
        for thr in thrs:
            for quirk in (True, False):
                keep, cnt = ops.nms_batched(bt, st, None, float(thr), quirk)
                got = keep[0, :int(cnt[0])].cpu().numpy()
                assert np.array_equal(got, oracle.nms(b, s, float(thr), quirk)), (n, thr, quirk)


def test_nms_nonfinite_boxes_vs_oracle(va, oracle):
    from vision_amd import ops
    n = 4500
    b, s = wl.stress_boxes(21, n, dense=True)
    rng = np.random.default_rng(4)
    idx = rng.choice(n, 40, replace=False)
    b[idx[:10], 2] = np.inf
    b[idx[10:20], 0] = np.nan
    b[idx[20:30]] = np.float32(3e38) * np.array([-1, -1, 1, 1], np.float32)
    b[idx[30:40], 3] = -np.inf
    with np.errstate(all="ignore"):
        want = oracle.nms(b, s, 0.5, True)
    keep, cnt = ops.nms_batched(torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda(), None, 0.5, True)
    assert np.array_equal(keep[0, :int(cnt[0])].cpu().numpy(), want)


def test_nms_batched_ragged_vs_oracle(va, oracle):
    from vision_amd import ops
    B, N = 9, 300
    rng = np.random.default_rng(5)
    counts = np.array([0, 1, 2, 64, 65, 128, 299, 300, 37], np.int32)
    boxes = np.stack([wl.random_boxes(50 + i, N) for i in range(B)])
    scores = np.stack([wl.tiefree_scores(60 + i, N) for i in range(B)])
    for quirk in (True, False):
        keep, cnt = ops.nms_batched(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(counts), 0.4, quirk)
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        for i in range(B):
            want = oracle.nms(boxes[i, :counts[i]], scores[i, :counts[i]], 0.4, quirk)
            assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (i, quirk)


def test_nms_batched_ragged_long_segments_vs_oracle(va, oracle):
    """several images with more than 4096 boxes each and ragged counts: per-image offsets of the compact-mask reduce,
    sparse and dense images side by side (a dense image falls back to the summary-driven reduce on its own)."""
    from vision_amd import ops
    N = 6000
    counts = np.array([0, 1, 70, 4097, 6000, 5000], np.int32)
    boxes = np.zeros((len(counts), N, 4), np.float32); scores = np.zeros((len(counts), N), np.float32)
    for i in range(len(counts)):
        b, s = wl.stress_boxes(100 + i, N, dense=(i == 5))
        boxes[i], scores[i] = b, s
    for thr, quirk in ((0.5, True), (0.3, False)):
        keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts), thr, quirk)
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        for i in range(len(counts)):
            want = oracle.nms(boxes[i, :counts[i]], scores[i, :counts[i]], thr, quirk)
            assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (i, thr, quirk)
    # short segments inside a long-segment layout, on a workspace a full-length run has just used: the summary words
    # past a short segment's end are stale then and must not be looked at (found by tools/stress_random.py)
    ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), None, 0.5, True)
    short = np.array([305, 1422, 2153, 3102, 64, 4095], np.int32)
    keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(short), 0.5, True)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for i in range(len(short)):
        want = oracle.nms(boxes[i, :short[i]], scores[i, :short[i]], 0.5, True)
        assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), (i, int(short[i]))


def test_nms_frcnn_top(va, golden):
    from vision_amd import fast_rcnn
    t = golden("nms_frcnn_top")
    r, s = fast_rcnn.nms(list(t["rects"]), list(t["scores"]))
    assert np.array_equal(r, t["best_rect"]) and s == float(t["best_score"])


def test_nms_frcnn_top_ties_and_near_ties(va, oracle):
    """fast_rcnn.py:601-610 takes the maximum on the scores as given (float64).  Near ties (distinct doubles that are one
    fp32 value) must pick the true maximum; a NaN score ranks first, as argsort places NaNs last; among exactly equal
    maxima the reference's pick is whatever numpy's unstable argsort does -- ours is the highest index."""
    from vision_amd import fast_rcnn
    rng = np.random.default_rng(5)
    rects = [rng.integers(0, 200, 4) for _ in range(8)]
    cases = {
        "near": [0.6, 0.9000000001, 0.9000000002, 0.9, 0.1, 0.9000000001, 0.3, 0.2],
        "tie": [0.6, 0.9, 0.2, 0.9, 0.1, 0.9, 0.3, 0.2],
        "nan": [0.6, 0.9, float("nan"), 0.95, 0.1, 0.9, 0.3, 0.2],
        "neg": [-3.0, -1.5, -1.5, -2.0, -9.0, -1.5000001, -7.0, -8.0],
    }
    for name, sc in cases.items():
        r, s = fast_rcnn.nms(rects, sc)
        wr, ws = oracle.nms_frcnn_top(rects, sc)
        assert (s == ws) or (np.isnan(s) and np.isnan(ws)), name
        if name in ("near", "nan"):
            assert np.array_equal(r, wr), name
        else:
            # equal maxima: numpy's argsort is unstable (its order among equal keys differs between builds), so the
            # reference's pick is implementation-defined; ours is the highest index
            tied = [i for i, v in enumerate(sc) if v == max(sc)]
            assert np.array_equal(r, rects[tied[-1]]), name
            assert any(np.array_equal(wr, rects[i]) for i in tied), name
    big = rng.random(100003)
    big[[17, 5000, 99999]] = 2.0 + np.array([0.0, 1e-13, 0.0])
    r, s = fast_rcnn.nms(np.arange(4 * 100003).reshape(-1, 4), big)
    assert s == big[5000] and np.array_equal(r, np.arange(4 * 100003).reshape(-1, 4)[5000])
    with pytest.raises(IndexError):
        fast_rcnn.nms([], [])


def test_nms_two_streams_do_not_share_scratch(va, oracle):
    """ops.* take their scratch per (device, stream): two NMS calls in flight on two streams must both be right
    (with one buffer per device they raced on masks and tickets)."""
    from vision_amd import ops, _abi
    n = 6000
    sets = [wl.frcnn_proposals(40 + i, n) for i in range(2)]
    want = [oracle.nms(b, s, 0.5) for b, s in sets]
    dev = torch.device("cuda", 0)
    tens = [(torch.from_numpy(b).to(dev)[None], torch.from_numpy(s).to(dev)[None]) for b, s in sets]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    for rep in range(10):
        outs = []
        for (b, s), st in zip(tens, streams):
            with torch.cuda.stream(st):
                outs.append(ops.nms_batched(b, s, None, 0.5, True))
        torch.cuda.synchronize()
        for (keep, cnt), w in zip(outs, want):
            assert int(cnt[0]) == len(w) and np.array_equal(keep[0, :len(w)].cpu().numpy(), w), rep
    keys = [k for k in _abi._ws if k[1] in (streams[0].cuda_stream, streams[1].cuda_stream)]
    assert len(keys) == 2 and _abi._ws[keys[0]].data_ptr() != _abi._ws[keys[1]].data_ptr()


# ------------------------------------------------------------------ decode / evaluate / ap
def test_decode_golden(va, golden):
    from vision_amd import yolo_v1
    g = golden("decode_v1")
    for tag in "abc":
        ct, pt = g[tag + "_thr"]
        for i in range(g["output"].shape[0]):
            r = yolo_v1.decode(torch.from_numpy(g["output"][i]), torch.from_numpy(g["target"][i]), ct, pt)
            box, lab, conf, score, tb, tl = r
            want = g[f"{tag}_box_{i}"]
            if len(want) == 0:
                assert box == [] and lab == []
                continue
            assert box.dtype == torch.float32 and lab.dtype == torch.int64 and tb.dtype == torch.float64
            assert np.array_equal(box.numpy(), want), (tag, i)
            assert np.array_equal(lab.numpy(), g[f"{tag}_label_{i}"])
            assert np.array_equal(conf.numpy(), g[f"{tag}_conf_{i}"])
            assert np.array_equal(score.numpy(), g[f"{tag}_score_{i}"])
            assert np.array_equal(tb.numpy(), g[f"{tag}_tbox_{i}"]) and tl == int(g[f"{tag}_tlabel_{i}"])
    r = yolo_v1.decode(torch.from_numpy(g["low_output"]), torch.from_numpy(g["target"][0]), 0.2, 0.1)
    assert r[0] == [] and r[3] == []


def test_evaluate_and_ap_golden(va, golden):
    from vision_amd import yolo_v1
    g = golden("evaluate_v1")
    det, tgt = yolo_v1.evaluate_batch(torch.from_numpy(g["output"]).cuda(), torch.from_numpy(g["target"]).cuda(), 5,
                                      prob_threshold=0.01, nms_threshold=0.3)
    for c in range(5):
        assert np.array_equal(np.array(det[c], np.float64).reshape(-1, 6), g[f"detects_{c}"]), c
        assert np.array_equal(np.array(tgt[c], np.float64).reshape(-1, 6), g[f"targets_{c}"]), c
    assert np.array_equal(np.array(yolo_v1.ap(det, tgt, 0.05)), g["aps_iou005"])                 # float64, bit for bit
    assert np.array_equal(np.array(yolo_v1.ap(det, tgt, 0.5)), g["aps_iou05"])


def _ap_case(seed, n_img, n_cls, per_img, empty_classes=()):
    """detections / targets in the reference's dict-of-rows form: confidence ties, several targets per (class, image),
    images without a target, classes without targets or without detections."""
    rng = np.random.default_rng(seed)
    detects = {c: [] for c in range(n_cls)}
    targets = {c: [] for c in range(n_cls)}
    for i in range(n_img):
        for _ in range(int(rng.integers(0, 3))):
            c = int(rng.integers(0, n_cls))
            if c in empty_classes:
                continue
            xy = rng.uniform(0, 0.6, 2); wh = rng.uniform(0.1, 0.4, 2)
            targets[c].append([i, 1] + np.concatenate([xy, xy + wh]).tolist())        # float64 rows, as bbox_target.tolist()
        for _ in range(int(rng.integers(0, per_img + 1))):
            c = int(rng.integers(0, n_cls))
            if targets[c] and targets[c][-1][0] == i and rng.random() < 0.6:           # near the ground truth
                g = np.array(targets[c][-1][2:])
                b = g + rng.normal(0, 0.04, 4)
            else:
                xy = rng.uniform(0, 0.6, 2); b = np.concatenate([xy, xy + rng.uniform(0.05, 0.4, 2)])
            conf = np.float32(rng.integers(1, 200) / 200.0)                            # many exact ties
            detects[c].append([i, float(conf)] + [float(np.float32(v)) for v in b])
    return detects, targets


@pytest.mark.parametrize("seed,n_img,n_cls,per_img,empty", [(0, 40, 3, 3, ()), (1, 3000, 6, 4, (4,)), (2, 500, 20, 30, (0, 7))])
def test_ap_vs_oracle(va, oracle, seed, n_img, n_cls, per_img, empty):
    """bit-exact float64 AP per class, at thresholds that sit exactly on an IoU value (float32 >=, as torch compares)."""
    from vision_amd import yolo_v1
    det, tgt = _ap_case(seed, n_img, n_cls, per_img, empty)
    if seed == 2:
        det[3] = []                                                                    # a class with targets but no detections
    some = [d for c in det for d in det[c] if any(t[0] == d[0] for t in tgt[c])][:50]
    edge = 0.5
    for d in some:                                                                     # an IoU that really occurs
        c = next(c for c in det if d in det[c])
        g = [t for t in tgt[c] if t[0] == d[0]][-1]
        v = float(oracle.iou_elementwise(np.array([d[2:]], np.float32), np.array([g[2:]], np.float32))[0])
        if 0.05 < v < 0.95:
            edge = v
            break
    for thr in (0.01, 0.05, 0.5, 0.7, edge, float(np.nextafter(np.float32(edge), np.float32(1)))):
        got = yolo_v1.ap(det, tgt, thr)
        want = oracle.average_precision(det, tgt, thr)
        assert len(got) == len(want)
        assert np.array_equal(np.array(got), np.array(want, np.float64), equal_nan=True), (thr, got, want)


def test_evaluate_and_ap_device_resident(va, oracle):
    """decode -> NMS -> per-class rows -> AP without leaving the device equals the list-based path and the oracle."""
    from vision_amd import yolo_v1
    out, tgt = wl.yolo1_batch(5, 64)
    o, t = torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda()
    det_l, tgt_l = yolo_v1.evaluate_batch(o, t, 5, prob_threshold=0.01, nms_threshold=0.3)
    flat = yolo_v1.evaluate_batch_device(o, t, 5, prob_threshold=0.01, nms_threshold=0.3)
    assert flat[0].is_cuda and flat[0].shape[0] == sum(len(v) for v in det_l.values())
    want_det, want_tgt = oracle.evaluate_batch(out, tgt, 5, 0.01, 0.3)
    for thr in (0.05, 0.5):
        want = oracle.average_precision(want_det, want_tgt, thr)
        assert np.array_equal(np.array(yolo_v1.ap_device(*flat, 5, 64, thr)), np.array(want), equal_nan=True)
        assert np.array_equal(np.array(yolo_v1.ap(det_l, tgt_l, thr)), np.array(want), equal_nan=True)


def test_ap_degenerate(va, oracle):
    from vision_amd import yolo_v1
    assert yolo_v1.ap({0: [], 1: []}, {0: [], 1: []}, 0.5) == oracle.average_precision({0: [], 1: []}, {0: [], 1: []}, 0.5)
    det = {0: [[0, 0.9, 0.1, 0.1, 0.5, 0.5]], 1: [[5, 0.3, 0.0, 0.0, 1.0, 1.0]]}
    tgt = {0: [[0, 1, 0.1, 0.1, 0.5, 0.5], [0, 1, 0.6, 0.6, 0.9, 0.9]], 1: []}       # the later target of image 0 wins (:321)
    got, want = yolo_v1.ap(det, tgt, 0.5), oracle.average_precision(det, tgt, 0.5)
    assert np.array_equal(np.array(got), np.array(want), equal_nan=True), (got, want)


@pytest.mark.parametrize("fused", [True, False])
def test_decode_nms_config2_vs_oracle(va, oracle, fused):
    """config 2 at full size (batch 64); ties between the two slots of a cell are ordered lower index first.
    fused: decode + NMS in one launch (dg_yolo1_decode_nms_f32) / the decode launch + the general batched NMS."""
    from vision_amd import yolo_v1
    out, tgt = wl.yolo1_batch(0, 64)
    box, label, conf, score, count, keep, kcnt, tbox, tlabel = [
        t.cpu().numpy() for t in yolo_v1.decode_nms_batched(torch.from_numpy(out), torch.from_numpy(tgt), 0.2, 0.01, 0.3, fused=fused)]
    for i in range(64):
        b, l, c, s, tb, tl = oracle.yolo1_decode(out[i], tgt[i], 0.2, 0.01)
        assert count[i] == len(b) and np.array_equal(box[i, :count[i]], b) and np.array_equal(label[i, :count[i]], l)
        want = oracle.nms(b, s, 0.3)
        assert kcnt[i] == len(want) and np.array_equal(keep[i, :kcnt[i]], want), i
        assert np.array_equal(tbox[i], tb) and tlabel[i] == tl


# ------------------------------------------------------------------ RoI pool
def _nms_per_class_oracle(oracle, boxes, scores, labels, C, thr, quirk):
    """the spec of per-class mode: the oracle's (= the reference's) nms applied to each class's boxes in list order"""
    keep, per = [], []
    for c in range(C):
        idx = np.flatnonzero(labels == c)
        k = oracle.nms(boxes[idx], scores[idx], thr, quirk) if idx.size else np.zeros(0, np.int64)
        keep.append(idx[k])
        per.append(len(k))
    return np.concatenate(keep) if keep else np.zeros(0, np.int64), per


def test_decode_nms_graph_replay(va, oracle):
    """DecodeNmsGraph: the fused launch captured once, replayed on new inputs written into its static buffers."""
    from vision_amd import yolo_v1
    gr = yolo_v1.DecodeNmsGraph(16, 5, 0.2, 0.01, 0.3)
    for seed in (3, 4):
        out, tgt = wl.yolo1_batch(seed, 16)
        gr.output.copy_(torch.from_numpy(out)); gr.target.copy_(torch.from_numpy(tgt))
        got = [x.cpu().numpy() for x in gr()]
        want = [x.cpu().numpy() for x in yolo_v1.decode_nms_batched(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), 0.2, 0.01, 0.3, fused=False)]
        cnt, kc = want[4], want[6]
        assert np.array_equal(got[4], cnt) and np.array_equal(got[6], kc)
        for i in range(16):
            assert np.array_equal(got[0][i, :cnt[i]], want[0][i, :cnt[i]]) and np.array_equal(got[5][i, :kc[i]], want[5][i, :kc[i]])
        assert np.array_equal(got[7], want[7]) and np.array_equal(got[8], want[8])


@pytest.mark.parametrize("per_class", [False, True])
@pytest.mark.parametrize("C", [5, 20, 1])
def test_decode_nms_fused_equals_unfused_and_oracle(va, oracle, C, per_class):
    """dg_yolo1_decode_nms_f32 (one launch, candidates in LDS) against the decode launch + the general batched NMS, bit for
    bit, and against the oracle's decode + nms: thresholds that keep nothing / everything (98 candidates: both waves of
    rows), thresholds sitting exactly on IoUs of the set and their float neighbours, 0, inf, NaN, both quirk settings, a
    batch without targets, images with no / one / two candidates."""
    from vision_amd import ops
    rng = np.random.default_rng(100 + C + 7 * per_class)
    B = 48
    out = rng.random((B, 7, 7, 10 + C), dtype=np.float32)
    out[..., 2:4] *= 0.6; out[..., 7:9] *= 0.6                       # sizes that overlap their neighbours
    out[0] = 0.0                                                       # nothing passes
    out[1] = 0.0; out[1, 3, 4, :] = 0.9                                # one cell: two candidates with equal scores
    out[2] = 0.0; out[2, 0, 0, [0, 1, 2, 3, 4]] = 0.8; out[2, 0, 0, 10] = 0.7   # exactly one candidate
    out[3, ..., 10:] = 0.95; out[3, ..., 4] = 1.0; out[3, ..., 9] = 1.0          # all 98 pass with equal scores
    tgt = np.zeros_like(out)
    for i in range(B):
        tgt[i, rng.integers(0, 7), rng.integers(0, 7), :] = rng.random(10 + C, dtype=np.float32) + 0.1
    # IoUs of image 5's candidates give thresholds that sit exactly on a quotient
    b5, _, _, s5, _, _ = oracle.yolo1_decode(out[5], tgt[5], 0.2, 0.01)
    ious = oracle.iou_pairwise(b5[:20], b5[:20], 1e-7).ravel()
    ious = ious[(ious > 0.05) & (ious < 0.9)]
    thrs = [0.3, 0.0, float("inf"), float("nan"), -1.0, 1e-30]
    for q in ious[:4]:
        thrs += [float(q), float(np.nextafter(np.float32(q), np.float32(0))), float(np.nextafter(np.float32(q), np.float32(1)))]
    o, t = torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda()
    saw_full = False
    for n_thr, thr in enumerate(thrs):
        for quirk in (True, False):
            for ct, pt in ((0.2, 0.01), (0.5, 0.3)):
                tt = t if (n_thr % 2 == 0) else None
                f = ops.yolo1_decode_batched(o, tt, ct, pt, thr, per_class, quirk)
                box, label, conf, score, count, tbox, tlabel, keep, kcnt, ccnt = f
                u = ops.yolo1_decode_batched(o, tt, ct, pt)
                assert torch.equal(count, u[4])
                cnt = count.cpu().numpy()
                for a, b in zip(f[:4], u[:4]):                       # box, label, conf, score: the rows the decode wrote
                    a, b = a.cpu().numpy(), b.cpu().numpy()
                    assert all(np.array_equal(a[i, :cnt[i]], b[i, :cnt[i]]) for i in range(B))
                if tt is not None:
                    assert np.array_equal(tbox.cpu().numpy(), u[5].cpu().numpy(), equal_nan=True)
                    assert torch.equal(tlabel, u[6])
                if per_class:
                    k2, c2, cc2, st = ops.nms_batched_per_class(u[0], u[3], u[1], C, u[4], thr, quirk)
                    assert torch.equal(ccnt, cc2)
                else:
                    k2, c2 = ops.nms_batched(u[0], u[3], u[4], thr, quirk)
                assert torch.equal(kcnt, c2), (thr, quirk)
                keep, k2, kc = keep.cpu().numpy(), k2.cpu().numpy(), kcnt.cpu().numpy()
                for i in range(B):
                    assert np.array_equal(keep[i, :kc[i]], k2[i, :kc[i]]), (thr, quirk, i)
                saw_full |= int(count.max()) == 98
        if n_thr == 0 and not per_class:                               # and directly against the oracle
            box, score, count = [x.cpu().numpy() for x in (f[0], f[3], f[4])]
            for i in range(0, B, 5):
                want = oracle.nms(box[i, :count[i]], score[i, :count[i]], thr, False)
                assert kc[i] == len(want) and np.array_equal(keep[i, :kc[i]], want)
    assert saw_full


@pytest.mark.parametrize("fused", [True, False])
def test_decode_nms_config2_per_class_vs_oracle(va, oracle, fused):
    """BASELINE configs[1] as worded: YOLOv1 decode + PER-CLASS NMS, batch 64.  Parity unpinned vs the reference (its nms is
    class-agnostic, yolo.py:282-298): checked against the oracle's nms applied per class.  Both box slots of a cell share the
    class score, so this input has tied keys by construction: ties are "lower index first" on both sides."""
    from vision_amd import yolo_v1
    out, tgt = wl.yolo1_batch(0, 64)
    box, label, conf, score, count, keep, kcnt, _, _ = yolo_v1.decode_nms_batched(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(),
                                                                                  0.2, 0.01, 0.3, per_class=True, fused=fused)
    box, label, score, count, keep, kcnt = [t.cpu().numpy() for t in (box, label, score, count, keep, kcnt)]
    assert count.max() > 60
    for i in range(64):
        m = count[i]
        want, _ = _nms_per_class_oracle(oracle, box[i, :m], score[i, :m], label[i, :m], 5, 0.3, True)
        assert kcnt[i] == len(want) and np.array_equal(keep[i, :kcnt[i]], want), i


@pytest.mark.parametrize("quirk", [True, False])
def test_nms_per_class_ragged_vs_oracle(va, oracle, quirk):
    from vision_amd import ops
    B, N, C = 5, 3000, 20
    rng = np.random.default_rng(77)
    props = [wl.frcnn_proposals(300 + i, N) for i in range(B)]
    boxes = np.stack([b for b, _ in props]); scores = np.stack([s for _, s in props])
    labels = rng.integers(0, C, (B, N)).astype(np.int64)
    labels[1] = 7                                            # one class only
    labels[2, :] = np.arange(N) % 3 + 17                     # classes 17..19 only
    counts = np.array([N, N, 2999, 1, 0], np.int32)
    keep, kcnt, ccnt, status = ops.nms_batched_per_class(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(),
                                                         torch.from_numpy(labels).cuda(), C, torch.from_numpy(counts).cuda(), 0.5, quirk)
    keep, kcnt, ccnt = keep.cpu().numpy(), kcnt.cpu().numpy(), ccnt.cpu().numpy()
    assert int(status.item()) == 0
    for i in range(B):
        m = counts[i]
        want, per = _nms_per_class_oracle(oracle, boxes[i, :m], scores[i, :m], labels[i, :m], C, 0.5, quirk)
        assert kcnt[i] == len(want) and np.array_equal(keep[i, :kcnt[i]], want), i
        assert list(ccnt[i]) == per
    # a label outside [0, C) is reported and its box ignored
    labels[0, 5] = C
    labels[0, 6] = -1
    keep2, kcnt2, _, status2 = ops.nms_batched_per_class(torch.from_numpy(boxes[:1]).cuda(), torch.from_numpy(scores[:1]).cuda(),
                                                         torch.from_numpy(labels[:1]).cuda(), C, None, 0.5, quirk)
    assert int(status2.item()) == 1
    ok = np.ones(N, bool); ok[5] = ok[6] = False
    idx = np.flatnonzero(ok)
    want, _ = _nms_per_class_oracle(oracle, boxes[0][idx], scores[0][idx], labels[0][idx], C, 0.5, quirk)
    assert np.array_equal(keep2.cpu().numpy()[0, :int(kcnt2[0])], idx[want])


def test_roipool_golden(va, golden):
    from vision_amd import fast_rcnn
    g = golden("roipool")
    img = torch.from_numpy(g["img"])
    m = fast_rcnn.slowroipool((7, 7), mode="max")
    out = m.foward(img, g["rois"], g["idx"])
    assert out.shape == (96, 8, 7, 7) and not out.is_cuda
    assert np.array_equal(out.numpy(), g["max_out"])                         # bit-exact
    assert np.array_equal(m.argmax.cpu().numpy(), g["max_argmax"])           # bit-exact
    avg = fast_rcnn.slowroipool((7, 7), mode="avg").foward(img.cuda(), g["rois"], g["idx"])
    np.testing.assert_allclose(avg.cpu().numpy(), g["avg_out"], rtol=1e-6, atol=1e-7)
    # backward through autograd, both modes
    for mode, key in (("max", "max_grad_in"), ("avg", "avg_grad_in")):
        x = img.cuda().requires_grad_(True)
        y = fast_rcnn.slowroipool((7, 7), mode=mode)(x, g["rois"], g["idx"])
        y.backward(torch.from_numpy(g["grad_out"]).cuda())
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[key], rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):
        m.foward(img, g["empty_rois"], g["idx"][:2])


def test_roipool_config1_vs_oracle(va, oracle):
    """config 1 shape: 512x38x50 map, 2000 RoIs (oracle takes ~2 s)"""
    from vision_amd import ops
    img = wl.frcnn_feature_map(0)
    rois = wl.frcnn_rois(0)
    idx = np.zeros(2000, np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, 1, 38, 50, x.device)
    assert int(status.item()) == 0
    out, arg = ops.roipool_forward(x, win, "max")
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(arg.cpu().numpy(), warg)
    rng = np.random.default_rng(1)
    gout = rng.standard_normal(want.shape, dtype=np.float32)
    gin = ops.roipool_backward(torch.from_numpy(gout).cuda(), arg, win, img.shape, "max").cpu().numpy()
    wgin = oracle.roipool_bwd(gout, warg, img.shape, rois, idx, 7, "max")
    np.testing.assert_allclose(gin, wgin, rtol=1e-4, atol=1e-4)     # fp32 sums of ~50 terms in a different order


def test_roi_windows_out_of_domain_vs_oracle(va, oracle):
    """dg_roi_windows on RoIs the reference never meant to see -- far outside [0, 1], reversed, on cell boundaries, a hair below
    zero, negative stops (python wraps them too), NaN, infinities, negative image indices (counted from the last image) --
    against the oracle's window, which tests/test_oracle_golden.py pins to numpy's conversion and python's slice normalisation."""
    import ctypes as C
    from vision_amd import ops
    lib = oracle.lib()
    rng = np.random.default_rng(8)
    special = np.array([-1e-18, 1e-18, 1 - 1e-16, 1 + 1e-15, 0.0, 1.0, -1.0, 2.0, 1e12, -1e12, 1e300, -1e300, np.nan, np.inf, -np.inf, -0.4, -0.02])
    for H, W in ((38, 50), (1, 1), (7, 300), (233, 3)):
        n = 3000
        kind = rng.integers(0, 4, (n, 4))
        grid = np.stack([rng.integers(-2, W + 3, n) / W, rng.integers(-2, H + 3, n) / H, rng.integers(-2, W + 3, n) / W, rng.integers(-2, H + 3, n) / H], 1)
        rois = np.where(kind == 0, rng.uniform(-1.5, 2.5, (n, 4)), np.where(kind == 1, grid, np.where(kind == 2, rng.uniform(0, 1, (n, 4)),
                        special[rng.integers(0, len(special), (n, 4))])))
        rois = np.ascontiguousarray(rois, np.float64)
        idx = rng.integers(-4, 4, n).astype(np.int64)
        win, status = ops.roi_windows(rois, idx, 4, H, W, torch.device("cuda", 0))
        win = win.cpu().numpy()
        any_empty = False
        for r in range(n):
            a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            rc = lib.orc_roi_window(rois[r].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(H), C.c_int(W), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
            if rc != 0:
                any_empty = True
                continue
            assert tuple(int(v) for v in win[r]) == (a.value, b.value, c.value, d.value, int(idx[r]) % 4), (H, W, rois[r])
        assert int(status.item()) == (1 if any_empty else 0)
    _, status = ops.roi_windows(np.array([[0.1, 0.1, 0.5, 0.5]]), np.array([-5], np.int64), 4, 38, 50, torch.device("cuda", 0))
    assert int(status.item()) == 2                                # an image index python would refuse as well


@pytest.mark.parametrize("with_nan", [False, True, "inf", "inf+nan"])
def test_roipool_every_crop_width_with_ties(va, oracle, with_nan):
    """Every crop width 1..50 (the width-specialised, the generic staged and the direct code paths) at assorted heights,
    on a map quantised to a few levels so that almost every bin has tied maxima (first one must win, as in ATen),
    with both signs of zero, and optionally NaNs (a NaN always wins) and infinities (a bin of -inf alone keeps its first
    cell: the running maximum starts at -inf and only a strictly larger value replaces it)."""
    from vision_amd import ops
    rng = np.random.default_rng(17)
    B, C, H, W = 2, 128, 38, 50
    levels = np.array([-2.0, -1.0, -0.0, 0.0, 0.5, 1.0], np.float32)
    if isinstance(with_nan, str):
        levels = np.array([-np.inf, -np.inf, -np.inf, -1.0, -0.0, 0.0, np.inf], np.float32)
    img = levels[rng.integers(0, len(levels), (B, C, H, W))]
    if isinstance(with_nan, str):
        img[:, :, :, 30:] = np.where(rng.random((B, C, H, W - 30)) < 0.9, -np.inf, img[:, :, :, 30:])   # whole bins of -inf
    if with_nan is True or with_nan == "inf+nan":
        img[rng.random(img.shape) < 0.002] = np.nan
    wins = []
    for lw in range(1, W + 1):
        for lh in sorted(set(int(v) for v in rng.integers(1, H + 1, 5)) | {1 + lw % 7, H if lw % 9 == 0 else 7}):
            x1 = int(rng.integers(0, W - lw + 1)); y1 = int(rng.integers(0, H - lh + 1))
            wins.append((x1, y1, x1 + lw, y1 + lh))
    wins = np.array(wins, np.int64)
    rois = np.stack([(wins[:, 0] + 0.25) / W, (wins[:, 1] + 0.25) / H, (wins[:, 2] - 0.25) / W, (wins[:, 3] - 0.25) / H], 1)
    idx = rng.integers(0, B, len(wins)).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, B, H, W, x.device)
    assert int(status.item()) == 0 and np.array_equal(win.cpu().numpy()[:, :4], wins)
    out, arg = ops.roipool_forward(x, win, "max")
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    got = out.cpu().numpy()
    bad = (got.view(np.uint32) != want.view(np.uint32)) & ~np.isnan(want)                                  # bits: -0 stays -0
    assert not bad.any(), sorted(set((wins[:, 2] - wins[:, 0])[bad.reshape(len(wins), -1).any(1)]))
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(arg.cpu().numpy(), warg)
    # the uint16-argmax kernel is an instantiation of its own (strips share their boundary row there): same bits, same indices
    out16, arg16 = ops.roipool_forward(x, win, "max", argmax_dtype=torch.int16)
    got16 = out16.cpu().numpy()
    assert np.array_equal(got16.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)])
    assert np.array_equal(np.isnan(got16), np.isnan(want))
    assert np.array_equal(ops.argmax_i32(arg16).cpu().numpy(), warg)


def test_roipool_avg_every_crop_width(va, oracle):
    """avg mode (the reference class as written, fast_rcnn.py:354): every crop width 1..50 -- the width-specialised
    staged path up to 24, the direct path beyond -- against the oracle and against torch's AdaptiveAvgPool2d."""
    from vision_amd import ops
    rng = np.random.default_rng(23)
    B, C, H, W = 2, 128, 38, 50
    img = rng.standard_normal((B, C, H, W), dtype=np.float32)
    wins = []
    for lw in range(1, W + 1):
        for lh in sorted(set(int(v) for v in rng.integers(1, H + 1, 4)) | {1 + lw % 7, H if lw % 9 == 0 else 7}):
            x1 = int(rng.integers(0, W - lw + 1)); y1 = int(rng.integers(0, H - lh + 1))
            wins.append((x1, y1, x1 + lw, y1 + lh))
    wins = np.array(wins, np.int64)
    rois = np.stack([(wins[:, 0] + 0.25) / W, (wins[:, 1] + 0.25) / H, (wins[:, 2] - 0.25) / W, (wins[:, 3] - 0.25) / H], 1)
    idx = rng.integers(0, B, len(wins)).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, B, H, W, x.device)
    out, arg = ops.roipool_forward(x, win, "avg")
    assert arg is None
    want, _ = oracle.roipool_fwd(img, rois, idx, 7, "avg")
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-6)          # fp32 sums in a different order
    for r in range(0, len(wins), 17):
        x1, y1, x2, y2 = wins[r]
        ref = torch.nn.functional.adaptive_avg_pool2d(x[idx[r]:idx[r] + 1, :, y1:y2, x1:x2], 7)[0]
        torch.testing.assert_close(out[r], ref, rtol=1e-6, atol=1e-6)


def _every_width_rois(H, W, rng, per_width=2):
    """normalised RoIs whose crops take every width 1..W and many heights on an HxW map"""
    rois = []
    for w in range(1, W + 1):
        for _ in range(per_width):
            h = int(rng.integers(1, H + 1))
            x1 = int(rng.integers(0, W - w + 1)); y1 = int(rng.integers(0, H - h + 1))
            rois.append([(x1 + 0.3) / W, (y1 + 0.3) / H, (x1 + w - 0.3) / W, (y1 + h - 0.3) / H])
    return np.array(rois, np.float64)


@pytest.mark.parametrize("B,C,H,W,reps", [(2, 70, 38, 50, 2), (2, 128, 38, 50, 2), (1, 6, 70, 90, 2), (1, 8, 38, 50, 45), (3, 4, 20, 100, 3)])
def test_roipool_avg_backward_vs_oracle(va, oracle, B, C, H, W, reps):
    """avg-mode backward (the reference class as written + autograd) against the oracle on every crop width: channel counts
    that leave a partial slice (C % 4 != 0: the planar cb < 4 path), maps wider and taller than 64 cells (the row-segment
    walk and the integer-division bin ranges), and one image with >= 2048 RoIs (the RoI list is split over gridDim.z and the
    slices are added with global atomics)."""
    from vision_amd import ops
    rng = np.random.default_rng(B * 1000 + C)
    rois = np.concatenate([_every_width_rois(H, W, rng) for _ in range(reps)])
    n = rois.shape[0]
    idx = rng.integers(0, B, n).astype(np.int64)
    img_shape = (B, C, H, W)
    gout = rng.standard_normal((n, C, 7, 7)).astype(np.float32)
    dev = torch.device("cuda", 0)
    win, st = ops.roi_windows(rois, idx, B, H, W, dev)
    assert int(st.item()) == 0
    got = ops.roipool_backward(torch.from_numpy(gout).to(dev), None, win, img_shape, "avg").cpu().numpy()
    want = oracle.roipool_bwd(gout, None, img_shape, rois, idx, 7, "avg")
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    # and the forward of the same set (values only: avg has no argmax)
    img = rng.standard_normal(img_shape).astype(np.float32)
    out, _ = ops.roipool_forward(torch.from_numpy(img).to(dev), win, "avg")
    wout, _ = oracle.roipool_fwd(img, rois, idx, 7, "avg")
    np.testing.assert_allclose(out.cpu().numpy(), wout, rtol=1e-5, atol=1e-5)


def test_roipool_other_output_sizes_golden(va, golden):
    """slowroipool(output_size) for sizes other than (7,7), through the drop-in class with autograd, against vectors
    generated from the reference class (rcnn/fast_rcnn.py:351-377) and torch's autograd."""
    from vision_amd import fast_rcnn
    g = golden("roipool_sizes")
    for tag in ("s3x5", "s14", "s1", "s2x9"):
        P = tuple(int(v) for v in g[f"{tag}_size"])
        size = P if P[0] != P[1] else P[0]
        for mode in ("max", "avg"):
            x = torch.from_numpy(g["img"]).cuda().requires_grad_(True)
            m = fast_rcnn.slowroipool(size, mode=mode)
            y = m.foward(x, g["rois"], g["idx"])
            assert tuple(y.shape) == g[f"{tag}_{mode}_out"].shape
            if mode == "max":
                assert np.array_equal(y.detach().cpu().numpy(), g[f"{tag}_max_out"]), tag
            else:
                np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}_avg_out"], rtol=1e-6, atol=1e-6)
            y.backward(torch.from_numpy(g[f"{tag}_grad_out"]).cuda())
            np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{tag}_{mode}_grad_in"], rtol=1e-5, atol=1e-5)
        m = fast_rcnn.slowroipool(size, mode="max")
        m.foward(torch.from_numpy(g["img"]).cuda(), g["rois"], g["idx"])
        assert np.array_equal(m.argmax.cpu().numpy(), g[f"{tag}_max_argmax"]), tag


@pytest.mark.parametrize("H,W,P,mode", [(300, 260, 7, "max"), (300, 260, 7, "avg"), (150, 200, 7, "max"), (130, 120, 7, "avg"),
                                        (300, 260, (4, 6), "max"), (38, 50, 5, "max"), (200, 250, 7, "max"), (200, 250, 7, "avg")])
def test_roipool_large_maps_and_sizes_vs_oracle(va, oracle, H, W, P, mode):
    """maps beyond the tuned kernels' limits (forward: 65535 cells; backward plane in LDS: 32768 cells in max mode, 14336 in
    avg mode) and other output sizes, forward and backward against the oracle."""
    from vision_amd import ops
    rng = np.random.default_rng(H * W)
    B, C = 2, 6
    img = rng.standard_normal((B, C, H, W)).astype(np.float32)
    n = 60
    xy = rng.uniform(0, 0.7, (n, 2)); wh = rng.uniform(0.01, 0.3, (n, 2))
    rois = np.concatenate([xy, xy + wh], 1)
    rois[0] = [0, 0, 1, 1]
    idx = rng.integers(0, B, n).astype(np.int64)
    dev = torch.device("cuda", 0)
    win, st = ops.roi_windows(rois, idx, B, H, W, dev)
    assert int(st.item()) == 0
    out, arg = ops.roipool_forward(torch.from_numpy(img).to(dev), win, mode, P=P)
    wout, warg = oracle.roipool_fwd(img, rois, idx, P, mode)
    if mode == "max":
        assert np.array_equal(out.cpu().numpy(), wout) and np.array_equal(arg.cpu().numpy(), warg)
    else:
        np.testing.assert_allclose(out.cpu().numpy(), wout, rtol=1e-5, atol=1e-5)
    gout = rng.standard_normal(wout.shape).astype(np.float32)
    gin = ops.roipool_backward(torch.from_numpy(gout).to(dev), arg, win, img.shape, mode, P=P).cpu().numpy()
    want = oracle.roipool_bwd(gout, warg, img.shape, rois, idx, P, mode)
    np.testing.assert_allclose(gin, want, rtol=1e-4, atol=1e-4)


def test_roipool_band_tuned_forward_generic_backward(va, oracle):
    """32768 < H*W <= 65535 in max mode: the tuned forward (uint16 argmax under autograd) paired with the generic
    global-atomic backward, which then reads a 16-bit index -- through slowroipool with requires_grad, against the oracle."""
    from vision_amd import fast_rcnn
    H, W = 200, 250
    rng = np.random.default_rng(50000)
    B, C = 2, 6
    img = rng.standard_normal((B, C, H, W)).astype(np.float32)
    n = 80
    xy = rng.uniform(0, 0.9, (n, 2)); wh = rng.uniform(0.004, 0.09, (n, 2))        # crops of 1..23 cells: the tuned strips' widths
    rois = np.concatenate([xy, xy + wh], 1)
    rois[0] = [0.0, 0.0, 0.05, 1.0]                                                 # the full height: argmax up to 49,999 (> int16)
    idx = rng.integers(0, B, n).astype(np.int64)
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    assert warg.max() > 40000
    x = torch.from_numpy(img).cuda().requires_grad_(True)
    y = fast_rcnn.slowroipool(7, mode="max").foward(x, rois, idx)
    assert np.array_equal(y.detach().cpu().numpy(), want)
    gout = rng.standard_normal(want.shape).astype(np.float32)
    y.backward(torch.from_numpy(gout).cuda())
    np.testing.assert_allclose(x.grad.cpu().numpy(), oracle.roipool_bwd(gout, warg, img.shape, rois, idx, 7, "max"), rtol=1e-4, atol=1e-4)


def test_roipool_odd_channels_and_batch(va, oracle):
    from vision_amd import ops
    rng = np.random.default_rng(2)
    img = rng.standard_normal((3, 70, 20, 24), dtype=np.float32)      # C not a multiple of 4 nor 64
    img[0, 3, 5, 5] = np.nan
    rois = wl.frcnn_rois(9, 50)
    idx = rng.integers(0, 3, 50).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, 3, 20, 24, x.device)
    for mode in ("max", "avg"):
        out, arg = ops.roipool_forward(x, win, mode)
        want, warg = oracle.roipool_fwd(img, rois, idx, 7, mode)
        if mode == "max":
            assert np.array_equal(out.cpu().numpy(), want, equal_nan=True) and np.array_equal(arg.cpu().numpy(), warg)
        else:
            np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-7)


def test_yolo1_encode_golden(va, golden):
    from vision_amd import yolo_v1
    g = golden("yolo1_encode")
    t5 = yolo_v1.encode_batched(g["bb"][:30], g["label"][:30], 5, dtype=torch.float64)
    t20 = yolo_v1.encode_batched(torch.from_numpy(g["bb"][30:]).cuda(), g["label"][30:], 20, dtype=torch.float64)
    assert t5.is_cuda and np.array_equal(t5.cpu().numpy(), g["target_c5"]) and np.array_equal(t20.cpu().numpy(), g["target_c20"])
    bb = g["bb"][3].copy()
    one = yolo_v1.encode(bb, int(g["label"][3]), 5)
    assert one.dtype == np.float64 and np.array_equal(one, g["target_c5"][3]) and np.array_equal(bb, g["bb_after"][3])
    with pytest.raises(IndexError):
        yolo_v1.encode(np.array([0.9, 0.9, 0.4, 0.4]), 1, 5)          # centre in cell 7: NumPy raises too
    with pytest.raises(IndexError):
        yolo_v1.encode(np.array([0.1, 0.1, 0.2, 0.2]), 6, 5)


# ------------------------------------------------------------------ YOLOv1 loss (fused geometry)
def test_yolo1_loss_golden_and_autograd(va, golden, oracle):
    from vision_amd import yolo_v1, ops
    g = golden("yolo1_loss")
    for name in ("b16c5", "b4c20", "b1c5"):
        out, tgt = g[name + "_modely"], g[name + "_targety"]
        C = out.shape[-1] - 10
        x = torch.from_numpy(out).cuda().requires_grad_(True)
        crit = yolo_v1.Loss(2, C)
        loss = crit(x, torch.from_numpy(tgt).cuda())
        (3.0 * loss).backward()
        assert abs(loss.item() - g[name + "_loss_f64"]) <= 1e-6 * abs(g[name + "_loss_f64"]), name      # 1e-6 relative
        np.testing.assert_allclose(x.grad.cpu().numpy(), 3.0 * g[name + "_grad"], rtol=1e-5, atol=3e-6, err_msg=name)
        # against the oracle: same op order -> tight
        ol, ocomp, ograd = oracle.yolo1_loss(out, tgt, quirk=True)
        loss6, grad = ops.yolo1_loss(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), True)
        assert abs(loss6[0].item() - ol) <= 2e-7 * abs(ol)
        np.testing.assert_allclose(loss6[1:].cpu().numpy(), ocomp, rtol=1e-6)
        np.testing.assert_allclose(grad.cpu().numpy(), ograd, rtol=2e-6, atol=1e-7)
        # CPU tensors in -> CPU scalar out, like the reference
        lc = crit(torch.from_numpy(out), torch.from_numpy(tgt))
        assert not lc.is_cuda and abs(lc.item() - loss.item()) == 0.0
    assert abs(yolo_v1.Loss.get_iou(yolo_v1.Loss, torch.tensor([[0., 0, 1, 1]]), torch.tensor([[0., .5, 1, 1.5]])).item() - 1 / 3) < 1e-6


def test_yolo1_loss_large_batch_no_quirk_vs_oracle(va, oracle):
    from vision_amd import ops
    out, tgt = wl.yolo1_batch(21, batch=700, num_classes=20)
    rng = np.random.default_rng(3)
    out[..., 2:4] = rng.uniform(-0.2, 0.9, out[..., 2:4].shape)
    tgt[5, 3, 3, 4] = -1.0                                    # neither object nor background: contributes nothing
    for quirk in (True, False):
        ol, ocomp, ograd = oracle.yolo1_loss(out, tgt, quirk=quirk)
        loss6, grad = ops.yolo1_loss(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), quirk)
        assert abs(loss6[0].item() - ol) <= 2e-7 * abs(ol), quirk
        np.testing.assert_allclose(grad.cpu().numpy(), ograd, rtol=2e-6, atol=1e-7)
    assert np.all(grad.cpu().numpy()[5, 3, 3] == 0)


# ------------------------------------------------------------------ YOLOv2 / R-CNN box regression
def test_yolo2_get_point_golden(va, golden, oracle):
    from vision_amd import yolo_v2
    g = golden("yolo2_get_point")
    pts, enc = yolo_v2.get_point(56, torch.from_numpy(g["bbox"]))
    assert pts.dtype == torch.int64
    assert np.array_equal(pts.numpy(), g["points"])
    # bit for bit, like the oracle (round 3: the 1e-6 tolerance this line used to carry was never needed -- fmodf is exact
    # on the device too, and the division by the cell is IEEE)
    assert np.array_equal(enc.numpy().view(np.uint32), g["enc"].view(np.uint32))
    p1, e1 = yolo_v2.get_point(56, torch.from_numpy(g["bbox"][0]))
    assert p1.shape == (2,) and e1.shape == (4,)
    xyxy = yolo_v2.decode_v2(pts, enc).numpy()
    assert np.array_equal(xyxy, oracle.yolo2_decode_cells(pts.numpy(), enc.numpy()))
    rng = np.random.default_rng(0)
    pred = rng.normal(0, 1, (13, 13, 5, 10)).astype(np.float32)
    anchors = rng.uniform(0.05, 0.5, (5, 2)).astype(np.float32)
    bx, cf = yolo_v2.decode_anchor(torch.from_numpy(pred), torch.from_numpy(anchors))
    wb, wc = oracle.yolo2_decode_anchor(pred, anchors)
    np.testing.assert_allclose(bx.numpy(), wb, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cf.numpy(), wc, rtol=1e-6, atol=1e-6)


def test_rcnn_boxreg_golden(va, golden):
    from vision_amd import rcnn_utils
    g = golden("rcnn_boxreg")
    t = rcnn_utils.calcul(torch.from_numpy(g["positive"]), torch.from_numpy(g["grt"]))
    np.testing.assert_allclose(t.numpy(), g["t_target"], rtol=1e-6, atol=1e-6)
    for variant, key in ((0, "g_rect_rcnn"), (1, "g_rect_frcnn")):
        got = rcnn_utils.decode_boxes(torch.from_numpy(g["p_rect"]), torch.from_numpy(g["tset"]), variant)
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-6, atol=1e-4)


def test_label_proposals_quota_walk_vs_oracle(va, oracle):
    """dg_rcnn_label_quota_f64 (the ordered quota walk of do_ss as a device scan) against the oracle's literal loop: quotas
    that fill early, late or never, min_num 0, overlapping thresholds (a proposal taken twice), more than one 1024-proposal
    chunk, and a degenerate proposal before / after the point where the walk stops (AssertionError only if visited)."""
    from vision_amd import rcnn_utils
    rng = np.random.default_rng(21)
    bbox = [40, 30, 120, 90]
    for n in (1, 7, 1024, 1025, 5000):
        xy = rng.integers(0, 160, (n, 2)); wh = rng.integers(4, 140, (n, 2))
        rects = np.concatenate([xy, wh], 1)
        rects[rng.integers(0, n, max(n // 8, 1))] = [45, 35, 110, 85]          # some clear positives
        for min_num, pos, neg, mult, full in ((5, 0.1, 0.1, 2, False), (64, 0.5, 0.1, 1, True), (0, 0.1, 0.1, 2, False),
                                              (3, 0.1, 0.3, 2, False), (100000, 0.1, 0.1, 2, False), (1, 0.5, 0.1, 1, True)):
            want = oracle.label_proposals(rects, bbox, 3, min_num, pos, neg, mult, full)
            got = rcnn_utils.label_proposals(rects, bbox, 3, min_num, pos, neg, mult, full)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[1], want[1]), (n, min_num, pos, neg)
    # a zero-width proposal: the reference asserts inside get_iou when the walk reaches it, and never sees it after the stop
    rects = np.tile(np.array([[45, 35, 110, 85], [0, 0, 5, 5]]), (40, 1))       # alternating positive / negative
    want = oracle.label_proposals(rects, bbox, 3, 2, 0.1, 0.1, 2)
    bad_late = rects.copy(); bad_late[60] = [10, 10, 0, 5]
    got = rcnn_utils.label_proposals(bad_late, bbox, 3, 2, 0.1, 0.1, 2)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    bad_early = rects.copy(); bad_early[3] = [10, 10, 0, 5]
    with pytest.raises(AssertionError):
        oracle.label_proposals(bad_early, bbox, 3, 2, 0.1, 0.1, 2)
    with pytest.raises(AssertionError):
        rcnn_utils.label_proposals(bad_early, bbox, 3, 2, 0.1, 0.1, 2)


def test_label_proposals_golden(va, golden):
    from vision_amd import rcnn_utils
    g = golden("do_ss_quota")
    rois, grt, labels = rcnn_utils.label_proposals(g["rects"], g["bbox"], int(g["label"]), int(g["min_num"]), 0.1, 0.1, 2)
    assert np.array_equal(rois, g["rcnn_rois"]) and np.array_equal(labels, g["rcnn_labels"]) and np.array_equal(grt, g["rcnn_grt"])
    rois, grt, labels = rcnn_utils.label_proposals(g["rects"], g["bbox"], int(g["label"]), int(g["min_num"]), 0.5, 0.1, 1, True)
    assert np.array_equal(rois, g["frcnn_rois"]) and np.array_equal(labels, g["frcnn_labels"])


@pytest.mark.parametrize("C,n", [(512, 2000), (70, 300), (128, 1), (8, 50), (4, 50)])
def test_roipool_uint16_argmax_equals_int32(va, C, n):
    """the 16-bit argmax layout (vector and scalar store paths, partial channel blocks) holds the same indices, and the
    backward from it gives the same gradient bits' worth of result as from the int32 one."""
    from vision_amd import ops
    rng = np.random.default_rng(C + n)
    img = torch.from_numpy(rng.standard_normal((2, C, 38, 50), dtype=np.float32)).cuda()
    rois = np.concatenate([wl.frcnn_rois(1, n), wl.frcnn_rois(2, n)])
    rois[::7, 2:] = np.minimum(1.0, rois[::7, :2] + 0.7)                       # some wide crops (direct path)
    idx = np.repeat(np.arange(2), n).astype(np.int64)
    win, st = ops.roi_windows(rois, idx, 2, 38, 50, img.device)
    out32, a32 = ops.roipool_forward(img, win, "max")
    out16, a16 = ops.roipool_forward(img, win, "max", argmax_dtype=torch.int16)
    assert a16.dtype == torch.int16 and torch.equal(out32, out16)
    assert torch.equal(ops.argmax_i32(a16), a32)
    g = torch.from_numpy(rng.standard_normal(tuple(out32.shape), dtype=np.float32)).cuda()
    g32 = ops.roipool_backward(g, a32, win, img.shape, "max")
    g16 = ops.roipool_backward(g, a16, win, img.shape, "max")
    assert torch.allclose(g32, g16, rtol=1e-4, atol=1e-3)                       # LDS accumulation order is not fixed


@pytest.mark.parametrize("C,n", [(128, 33000), (64, 65537), (128, 32767)])
def test_roipool_forward_launch_shapes_vs_oracle(va, oracle, C, n):
    """task counts just above / below the point where the forward goes from 2048 persistent workgroups to about 8
    block-strided tasks per workgroup (65536 tasks): every task still runs exactly once."""
    from vision_amd import ops
    rng = np.random.default_rng(C + n)
    B, H, W = 3, 38, 50
    img = rng.standard_normal((B, C, H, W), dtype=np.float32)
    rois = np.concatenate([wl.frcnn_rois(5 + i, 2000) for i in range((n + 1999) // 2000)])[:n]
    idx = rng.integers(0, B, n).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, st = ops.roi_windows(rois, idx, B, H, W, x.device)
    out = torch.full((n, C, 7, 7), float("nan"), device="cuda")
    arg = torch.full((n, C, 7, 7), -1, dtype=torch.int32, device="cuda")
    ops.roipool_forward(x, win, "max", out=out, argmax=arg)
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(arg.cpu().numpy(), warg)


def test_roipool_uint16_argmax_every_width_and_height(va, oracle):
    """every crop width 1..24 x every crop height 1..38 through the uint16-argmax kernel (strip-sharing code: the cases differ
    by height mod 7) on a tie-rich map with both zeros, against the oracle."""
    from vision_amd import ops
    rng = np.random.default_rng(23)
    B, C, H, W = 1, 64, 38, 50
    levels = np.array([-1.0, -0.0, 0.0, 0.5, 1.0, 2.0], np.float32)
    img = levels[rng.integers(0, len(levels), (B, C, H, W))]
    wins = np.array([(x1, y1, x1 + lw, y1 + lh) for lw in range(1, 25) for lh in range(1, H + 1)
                     for x1, y1 in [(int(rng.integers(0, W - lw + 1)), int(rng.integers(0, H - lh + 1)))]], np.int64)
    rois = np.stack([(wins[:, 0] + 0.25) / W, (wins[:, 1] + 0.25) / H, (wins[:, 2] - 0.25) / W, (wins[:, 3] - 0.25) / H], 1)
    idx = np.zeros(len(wins), np.int64)
    x = torch.from_numpy(img).cuda()
    win, status = ops.roi_windows(rois, idx, B, H, W, x.device)
    assert int(status.item()) == 0 and np.array_equal(win.cpu().numpy()[:, :4], wins)
    want, warg = oracle.roipool_fwd(img, rois, idx, 7, "max")
    out16, arg16 = ops.roipool_forward(x, win, "max", argmax_dtype=torch.int16)
    assert np.array_equal(out16.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(ops.argmax_i32(arg16).cpu().numpy(), warg)
    out32, arg32 = ops.roipool_forward(x, win, "max")
    assert np.array_equal(out32.cpu().numpy().view(np.uint32), want.view(np.uint32)) and np.array_equal(arg32.cpu().numpy(), warg)


# ------------------------------------------------------------------ k-means
@pytest.mark.parametrize("n,k", [(1000, 5), (200000, 5), (50000, 9)])
def test_kmeans_vs_oracle(va, oracle, n, k):
    from vision_amd import cluster
    wh = wl.anchor_wh(3, n)
    km = cluster.kmean(n_clusters=k, random_state=42).fit(wh)
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, k, 42)
    assert np.array_equal(km.labels_, labels)                     # bit-exact assignments
    assert np.array_equal(km.cluster_centers_, cent)              # bit-exact centroids (integer sums)
    assert km.n_iter_ == n_iter and abs(km.inertia_ - inertia) < 1e-9
    assert np.array_equal(km.predict(wh), labels)


def test_kmeans_labels_are_copied_on_first_access(va, oracle):
    """fit() leaves the labels on the device (uint8, `labels_device_`); `labels_` is sklearn's int32 ndarray, made on first
    access and cached; a second fit gives the object new labels; before any fit both are None."""
    from vision_amd import cluster
    km = cluster.kmean(n_clusters=5, random_state=42)
    assert km.labels_ is None and km.labels_device_ is None
    wh = wl.anchor_wh(4, 30000)
    km.fit(wh, fixed_iters=4)
    assert km._labels_host is None and km.labels_device_.dtype == torch.uint8 and km.labels_device_.is_cuda
    _, labels, _, _ = oracle.kmeans_fit(wh, 5, 42, fixed_iters=4)
    got = km.labels_
    assert got.dtype == np.int32 and np.array_equal(got, labels) and km.labels_ is got
    assert np.array_equal(km.labels_device_.cpu().numpy().astype(np.int32), labels)
    wh2 = wl.anchor_wh(5, 20000)
    km.fit(wh2, fixed_iters=3)
    _, labels2, _, _ = oracle.kmeans_fit(wh2, 5, 42, fixed_iters=3)
    assert np.array_equal(km.labels_, labels2) and np.array_equal(km.fit_predict(wh, fixed_iters=4), labels)


@pytest.mark.parametrize("iters", [1, 2, 7])
def test_kmeans_fixed_iterations_vs_oracle(va, oracle, iters):
    """the device-side loop stopped after a fixed number of iterations (the bench's mode): centroids, labels and the
    inertia of the LAST iteration (the only one that forms it) equal the oracle's."""
    from vision_amd import cluster
    wh = wl.anchor_wh(11, 300001)
    km = cluster.kmean(n_clusters=5, random_state=42).fit(wh, fixed_iters=iters)
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, 5, 42, fixed_iters=iters)
    assert np.array_equal(km.labels_, labels) and np.array_equal(km.cluster_centers_, cent)
    assert km.n_iter_ == iters and abs(km.inertia_ - inertia) < 1e-9


@pytest.mark.parametrize("n,k,iters", [(300001, 5, None), (250000, 9, 12), (500000, 3, 7), (400000, 16, 5), (1000, 5, None), (260000, 7, 9)])
def test_kmeans_prepared_fit_vs_oracle(va, oracle, n, k, iters, monkeypatch):
    """The prepared fit (dg_kmeans_prepare + dg_kmeans_iou_run_prepared: boxes grouped by (w, h) cell once, whole pieces of a
    cell labelled at once where the bounds on the IoUs leave no doubt, the rest box by box): centroids, labels, n_iter_ and
    inertia bit for bit those of the oracle's Lloyd loop and of the box-by-box loop; boxes outside the grid (zero, NaN-free
    negatives, tiny, larger than 1) sit in the catch-all cells and are always walked.  No reference counterpart (parity
    unpinned: the 1-IoU k-means is build-defined, spec in oracle.c)."""
    from vision_amd import cluster
    wh = wl.anchor_wh(n % 97, n)
    if n == 500000:
        wh[::1000] = [3.0, 0.5]; wh[5] = [0.0, 0.2]; wh[7] = [1e-4, 1e-4]; wh[11] = [-0.3, 0.4]
    x = torch.from_numpy(wh).cuda()
    monkeypatch.setenv("DG_KMEANS_CELLS", "1")
    a = cluster.kmean(k, 42).fit(x, fixed_iters=iters)
    assert a._backend.cells is not None
    monkeypatch.setenv("DG_KMEANS_CELLS", "0")
    b = cluster.kmean(k, 42).fit(x, fixed_iters=iters)
    assert b._backend.cells is None
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, k, 42, fixed_iters=iters) if iters else oracle.kmeans_fit(wh, k, 42)
    for km in (a, b):
        assert np.array_equal(km.cluster_centers_.view(np.uint32), cent.view(np.uint32))
        assert np.array_equal(km.labels_, labels) and km.n_iter_ == n_iter and abs(km.inertia_ - inertia) < 1e-9
    # a second chunk on the same prepared workspace continues the fit: history and centroids as the oracle's longer run
    if iters:
        monkeypatch.setenv("DG_KMEANS_CELLS", "1")
        c = cluster.kmean(k, 42).fit(x, fixed_iters=iters)
        c._run_iters(0, 3)
        torch.cuda.synchronize()
        cent2, labels2, _, _ = oracle.kmeans_fit(wh, k, 42, fixed_iters=iters + 3)
        assert np.array_equal(c._backend.cent.cpu().numpy().view(np.uint32), cent2.view(np.uint32))
        assert np.array_equal(c._backend.labels.cpu().numpy().astype(np.int32), labels2)
        h_acc = np.zeros(3 * k + 2, np.int64); lab = np.full(n, 255, np.uint8); ce = oracle.kmeans_init(wh, k, 42)
        hist = []
        for _ in range(iters + 3):
            acc = oracle.kmeans_step(wh, ce, lab); hist.append(int(acc[3 * k])); ce = oracle.kmeans_update(acc, ce)
        assert list(c._backend.history_host(3)) == hist[iters:iters + 3]
        # the changed counts from the very first iteration (the fit prepares without labels: every box counts as changed once)
        assert list(a._backend.history_host(iters)) == hist[:iters]
        # grouping again in mid-fit, this time with the labels the fit has reached, then continuing: the same three iterations
        e = cluster.kmean(k, 42).fit(x, fixed_iters=iters)
        e._backend.reprepare()
        e._run_iters(0, 3)
        torch.cuda.synchronize()
        assert np.array_equal(e._backend.cent.cpu().numpy().view(np.uint32), cent2.view(np.uint32))
        assert np.array_equal(e._backend.labels.cpu().numpy().astype(np.int32), labels2)
        assert list(e._backend.history_host(3)) == hist[iters:iters + 3]


@pytest.mark.parametrize("n,k,iters", [(300001, 5, 12), (250000, 9, None), (5000, 16, 9), (1200000, 5, 20), (220000, 3, 8), (210000, 7, 8)])
def test_kmeans_persistent_kernel_vs_oracle(va, oracle, n, k, iters, monkeypatch):
    """DG_KMEANS_PERSIST=1: all cell iterations of a chunk in one launch with a grid barrier per iteration (k_kc_persist; opt-in,
    measured slower than a launch per iteration): same centroids, labels, n_iter_, inertia and changed counts as the oracle, for a
    fixed number of iterations and for a fit that runs until nothing changes (whose chunks run over cells only until the last)."""
    from vision_amd import cluster
    wh = wl.anchor_wh(n % 89, n)
    if n == 5000:
        wh[::250] = [3.0, 0.5]; wh[3] = [-0.3, 0.4]; wh[9] = [0.0, 0.2]
    x = torch.from_numpy(wh).cuda()
    monkeypatch.setenv("DG_KMEANS_CELLS", "1")
    monkeypatch.setenv("DG_KMEANS_PERSIST", "1")
    km = cluster.kmean(k, 42).fit(x, fixed_iters=iters)
    assert km._backend.cells is not None
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, k, 42, fixed_iters=iters) if iters else oracle.kmeans_fit(wh, k, 42)
    assert np.array_equal(km.cluster_centers_.view(np.uint32), cent.view(np.uint32))
    assert np.array_equal(km.labels_, labels) and km.n_iter_ == n_iter and abs(km.inertia_ - inertia) < 1e-9
    if iters:
        lab = np.full(n, 255, np.uint8); ce = oracle.kmeans_init(wh, k, 42); hist = []
        for _ in range(iters):
            acc = oracle.kmeans_step(wh, ce, lab); hist.append(int(acc[3 * k])); ce = oracle.kmeans_update(acc, ce)
        assert list(km._backend.history_host(iters)) == hist


@pytest.mark.parametrize("k", [3, 9, 15])
def test_kmeans_boxes_outside_the_domain_vs_oracle(va, oracle, k, monkeypatch):
    """Boxes the 1-IoU distance was not made for (a negative side, both negative, zero, far larger than 1, tiny): the specification
    (oracle.c) is still an argmin over fp32 expressions, and both device loops follow it bit for bit -- the assignment pre-filter's
    shortcut only holds where every key inter / (area + A_c) is at most 1/2 (a box (-0.3, 0.4) made keys beyond the pole of
    x / (1 - x) and was assigned wrongly among 15 centroids until the filter refused such keys)."""
    from vision_amd import cluster
    n = 5000
    wh = wl.anchor_wh(41, n)
    odd = np.array([[3.0, 0.5], [0.0, 0.2], [1e-4, 1e-4], [-0.3, 0.4], [0.5, 7.0], [-0.2, -0.3], [0.4, -0.1], [-2.0, 0.05]], np.float32)
    idx = np.random.default_rng(3).choice(n, 40, replace=False)
    wh[idx] = odd[np.arange(40) % odd.shape[0]]
    x = torch.from_numpy(wh).cuda()
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, k, 42, fixed_iters=12)
    for cells in ("1", "0"):
        monkeypatch.setenv("DG_KMEANS_CELLS", cells)
        km = cluster.kmean(k, 42).fit(x, fixed_iters=12)
        assert (km._backend.cells is not None) == (cells == "1")
        assert np.array_equal(km.cluster_centers_.view(np.uint32), cent.view(np.uint32))
        assert np.array_equal(km.labels_, labels) and abs(km.inertia_ - inertia) < 1e-9


@pytest.mark.parametrize("max_iter", [5, 20, 40])
def test_kmeans_prepared_fit_stopped_by_max_iter_vs_oracle(va, oracle, max_iter, monkeypatch):
    """A fit that runs until nothing changes issues its intermediate chunks over cells only (no pass over the boxes: dg_kmeans_iou_run_prepared
    with labels NULL) and growing (8, 8, 16, 32 iterations); one that max_iter stops short of convergence must still end in a full
    chunk: labels and inertia of the LAST iteration's assignment (against the centroids before the last update), as the oracle's loop."""
    from vision_amd import cluster
    n = 250000
    wh = wl.anchor_wh(5, n)
    x = torch.from_numpy(wh).cuda()
    monkeypatch.setenv("DG_KMEANS_CELLS", "1")
    km = cluster.kmean(5, 42, max_iter=max_iter).fit(x)
    assert km._backend.cells is not None
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, 5, 42, max_iter=max_iter)
    assert n_iter == max_iter == km.n_iter_                       # this set needs more than 40 iterations
    assert np.array_equal(km.cluster_centers_.view(np.uint32), cent.view(np.uint32))
    assert np.array_equal(km.labels_, labels) and abs(km.inertia_ - inertia) < 1e-9


def test_kmeans_sharded_equals_single(va, oracle):
    """fake all-reduce: shards stepped one after another on one device, integer partial sums added."""
    from vision_amd import ops
    wh = wl.anchor_wh(4, 100001)
    k = 5
    cent0 = oracle.kmeans_init(wh, k, 42)
    x = torch.from_numpy(wh).cuda()
    cent = torch.from_numpy(cent0).cuda()
    labels = torch.full((wh.shape[0],), 255, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(17, dtype=torch.int64, device="cuda")
    ops.kmeans_step(x, cent, labels, acc)
    total = torch.zeros_like(acc)
    lab2 = torch.full((wh.shape[0],), 255, dtype=torch.uint8, device="cuda")
    bounds = [0, 30000, 30001, 77777, 100001]
    for a, b in zip(bounds[:-1], bounds[1:]):
        part = torch.zeros_like(acc)
        ops.kmeans_step(x[a:b], cent, lab2[a:b], part)
        total += part
    assert torch.equal(total, acc) and torch.equal(labels, lab2)
    want = oracle.kmeans_step(wh, cent0, np.full(wh.shape[0], 255, np.uint8))
    assert np.array_equal(acc.cpu().numpy(), want)


def test_kmeans_sharded_iter_two_shards_vs_oracle(va, oracle):
    """The sharded loop's device path proper: dg_kmeans_iou_iter (centroid update folded into the next step's prologue)
    on two shards of one device, their int64 sums added by hand (the all-reduce), against the oracle's fixed-iteration fit."""
    from vision_amd import cluster
    n, k, iters = 70001, 5, 6
    wh = wl.anchor_wh(12, n)
    init = oracle.kmeans_init(wh, k, 42)
    cut = 26001                                   # odd: the second shard starts 8-byte- but not 16-byte-aligned
    x = torch.from_numpy(wh).cuda()
    shards = []
    for lo, hi in ((0, cut), (cut, n)):
        be = cluster._HipBackend()
        be.prepare(x[lo:hi], init, k, iters)
        shards.append(be)
    for be in shards:
        be.begin()
    for t in range(iters):
        for be in shards:
            be.iterate(t > 0, t + 1 == iters, t - 1)
        total = shards[0].reduce_buffer() + shards[1].reduce_buffer()       # what the all-reduce does
        for be in shards:
            be.reduce_buffer().copy_(total)
    for be in shards:
        be.update(iters - 1)
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, k, 42, fixed_iters=iters)
    hist = None
    for be, (lo, hi) in zip(shards, ((0, cut), (cut, n))):
        c, l, acc = be.results()
        assert np.array_equal(c, cent)
        assert np.array_equal(l, labels[lo:hi])
        assert abs(float(acc[3 * k + 1]) / 2.0 ** 32 - inertia) < 1e-9
        h = be.history_host(iters)
        assert hist is None or np.array_equal(h, hist)
        hist = h
    assert hist[0] == n                           # every label changes in the first iteration


@pytest.mark.parametrize("graph", ["1", "0"])
def test_kmeans_sharded_loop_single_rank_group_vs_oracle(va, oracle, graph, monkeypatch):
    """kmean's sharded loop as the product runs it: launch + RCCL all-reduce per iteration (a single-rank group here: one
    GPU), once captured in a HIP graph and once eager, to convergence and for a fixed count, against the oracle."""
    import socket
    import torch.distributed as dist
    from vision_amd import cluster
    monkeypatch.setenv("DG_KMEANS_GRAPH", graph)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        wh = wl.anchor_wh(5, 150001)
        x = torch.from_numpy(wh).cuda()
        km = cluster.kmean(5, 42, sharded=True, comm="torch").fit(x)
        cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, 5, 42)
        assert np.array_equal(km.cluster_centers_, cent) and np.array_equal(km.labels_, labels)
        assert km.n_iter_ == n_iter and abs(km.inertia_ - inertia) < 1e-9
        assert km.comm_state.startswith("torch")
        assert km.graph_state == ("captured" if graph == "1" else "eager: graph capture disabled or host-side all-reduce"), km.graph_state
        km3 = cluster.kmean(5, 42, sharded=True, comm="torch").fit(x, fixed_iters=11)
        c3, l3, n3, i3 = oracle.kmeans_fit(wh, 5, 42, fixed_iters=11)
        assert np.array_equal(km3.cluster_centers_, c3) and np.array_equal(km3.labels_, l3) and km3.n_iter_ == 11
        km3._run_iters(0, 11)                     # replay (or eager re-run) continues the fit: 22 iterations
        torch.cuda.synchronize()
        c22, _, _, _ = oracle.kmeans_fit(wh, 5, 42, fixed_iters=22)
        assert np.array_equal(km3._backend.cent.cpu().numpy(), c22)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph", ["1", "0"])
@pytest.mark.parametrize("with_group", [False, True])
def test_kmeans_sharded_loop_library_comm_vs_oracle(va, oracle, graph, with_group, monkeypatch):
    """The sharded loop issued by the library: dg_comm_* (RCCL bound inside libdetgeom.so, a single-rank communicator here:
    one GPU) + dg_kmeans_iou_run_sharded -- eager first, captured in a HIP graph on the second run of a chunk, replayed
    after -- to convergence and for fixed counts, against the oracle.  with_group: the id travels through torch.distributed."""
    import socket
    import torch.distributed as dist
    from vision_amd import _abi, cluster
    monkeypatch.setenv("DG_KMEANS_GRAPH", graph)
    if with_group:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        wh = wl.anchor_wh(6, 150001)
        x = torch.from_numpy(wh).cuda()
        km = cluster.kmean(5, 42, sharded=True).fit(x)                     # comm=None: the library's communicator
        cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, 5, 42)
        assert km.comm_state == "so", km.comm_state
        assert np.array_equal(km.cluster_centers_, cent) and np.array_equal(km.labels_, labels)
        assert km.n_iter_ == n_iter and abs(km.inertia_ - inertia) < 1e-9
        km.close()
        km3 = cluster.kmean(5, 42, sharded=True, comm="so").fit(x, fixed_iters=11)           # first run of this chunk: eager
        c3, l3, n3, i3 = oracle.kmeans_fit(wh, 5, 42, fixed_iters=11)
        assert np.array_equal(km3.cluster_centers_, c3) and np.array_equal(km3.labels_, l3) and km3.n_iter_ == 11
        assert abs(km3.inertia_ - i3) < 1e-9
        assert km3.graph_state == ("armed" if graph == "1" else "eager: graph capture disabled"), km3.graph_state
        for total in (22, 33):                    # second run captures and replays, third replays: the fit continues
            km3._run_iters(0, 11)
            torch.cuda.synchronize()
            assert km3.graph_state == ("captured" if graph == "1" else "eager: graph capture disabled"), km3.graph_state
            cN, _, _, _ = oracle.kmeans_fit(wh, 5, 42, fixed_iters=total)
            assert np.array_equal(km3._backend.cent.cpu().numpy(), cN)
        # on a user stream (capturable as it is) and through the bare all-reduce
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            km4 = cluster.kmean(5, 42, sharded=True, comm=km3._comm).fit(x, fixed_iters=3)
            buf = torch.arange(40, dtype=torch.int64, device="cuda")
            km3._comm.allreduce_i64(buf)
        st.synchronize()
        c4, l4, _, _ = oracle.kmeans_fit(wh, 5, 42, fixed_iters=3)
        assert np.array_equal(km4.cluster_centers_, c4) and np.array_equal(km4.labels_, l4)
        assert torch.equal(buf.cpu(), torch.arange(40))
        km3.close()
        with pytest.raises(ValueError):
            _abi.check(_abi.load().dg_comm_allreduce_i64(None, None, 4, None))
    finally:
        if with_group:
            dist.destroy_process_group()


def _km_adversarial_boxes(seed, n, cent):
    """boxes that stress the assignment pre-filter: exact and near ties, tiny overlaps, copies of the centroids,
    zero-size, negative and very large boxes, next to ordinary ones."""
    rng = np.random.default_rng(seed)
    wh = np.exp(rng.normal(0.0, 1.0, (n, 2))).astype(np.float32)
    m = n // 8
    wh[:m] = np.exp(rng.normal(0.0, 4.0, (m, 2))).astype(np.float32)                      # IoU far below 1/32 for many
    s = np.exp(rng.normal(0.0, 1.0, m)).astype(np.float32)
    wh[m:2 * m] = np.stack([s, s], 1)                                                       # exact ties for mirrored centroids
    small = np.flatnonzero(cent.max(1) < 1e6)                   # x * 2^32 has to stay inside int64
    pick = small[rng.integers(0, small.size, m)]
    wh[2 * m:3 * m] = cent[pick] * (1 + rng.integers(-2, 3, (m, 2)) * np.float32(2.0 ** -23))  # within ulps of a centroid
    wh[3 * m:3 * m + 64] = 0.0
    wh[3 * m + 64:3 * m + 128] = np.float32(1e6) * (1 + rng.random((64, 2), dtype=np.float32))  # past the fast conversion range
    wh[3 * m + 128:3 * m + 192] *= -1.0
    wh[3 * m + 192:3 * m + 256] = np.float32(1e-7)
    return wh[rng.permutation(n)]


@pytest.mark.parametrize("name,cent", [
    ("mirrored", [[0.5, 2.0], [2.0, 0.5], [1.0, 1.0], [3.0, 3.0], [0.25, 0.25]]),
    ("near_duplicate", [[1.0, 1.0], [1.0, 1.0 + 2.0 ** -20], [1.0 + 2.0 ** -22, 1.0], [2.0, 0.5], [0.5, 2.0]]),
    ("duplicate", [[1.0, 1.0], [1.0, 1.0], [2.0, 2.0], [2.0, 2.0], [0.5, 0.5]]),
    ("huge_centroid", [[1.0, 1.0], [2.0 ** 31, 2.0 ** 31], [2.0, 0.5], [0.5, 2.0], [3.0, 3.0]]),
    ("k4", [[0.5, 2.0], [2.0, 0.5], [1.0, 1.0], [3.0, 3.0]]),
    ("k9", [[0.5, 2.0], [2.0, 0.5], [1.0, 1.0], [3.0, 3.0], [0.25, 0.25], [1.0, 3.0], [3.0, 1.0], [0.7, 0.7], [1.5, 1.5]]),
    ("k12", [[0.5, 2.0], [2.0, 0.5], [1.0, 1.0], [3.0, 3.0], [0.25, 0.25], [1.0, 3.0], [3.0, 1.0], [0.7, 0.7], [1.5, 1.5],
             [1.5, 1.5], [0.1, 5.0], [5.0, 0.1]]),
])
def test_kmeans_step_adversarial(va, oracle, name, cent):
    """the approximate pre-filter may only ever skip work, never change a label, a sum or the inertia."""
    from vision_amd import ops
    cent = np.asarray(cent, np.float32)
    k = cent.shape[0]
    wh = _km_adversarial_boxes(len(name), 1_200_003, cent)
    x = torch.from_numpy(wh).cuda()
    labels = torch.full((wh.shape[0],), 255, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(3 * k + 2, dtype=torch.int64, device="cuda")
    ops.kmeans_step(x, torch.from_numpy(cent).cuda(), labels, acc)
    lab = np.full(wh.shape[0], 255, np.uint8)
    want = oracle.kmeans_step(wh, cent, lab)
    assert np.array_equal(labels.cpu().numpy(), lab)
    assert np.array_equal(acc.cpu().numpy(), want)
    # an 8-byte-offset shard start (head peel) and an odd tail give the same answer
    acc2 = torch.zeros_like(acc)
    lab2 = torch.full((wh.shape[0] - 1,), 255, dtype=torch.uint8, device="cuda")
    ops.kmeans_step(x[1:], torch.from_numpy(cent).cuda(), lab2, acc2)
    lab1 = np.full(1, 255, np.uint8)
    first = oracle.kmeans_step(wh[:1], cent, lab1)
    assert np.array_equal(lab2.cpu().numpy(), lab[1:])
    got = acc2.cpu().numpy() + first
    got[3 * k] = want[3 * k]                                     # 'changed' counts are per call
    assert np.array_equal(got, want)
