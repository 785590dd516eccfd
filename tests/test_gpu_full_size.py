"""Parity at BASELINE.json's full sizes: through size-independent properties, and (round 3) against the C oracle itself where
it finishes in tens of seconds -- the 64k NMS keep lists, the 30-iteration fit on 10 M boxes, sampled rows of the 64k x 64k
IoU matrix.  Every check goes through the product path (C ABI); torch on the GPU is only the checker."""
import numpy as np
import pytest
import torch

from tools import workloads as wl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def va():
    from vision_amd import _abi
    _abi.load()
    return _abi


@pytest.mark.parametrize("dense,thr", [(False, 0.5), (False, 0.7), (True, 0.5), (True, 0.7)])
def test_nms_64k_properties(va, dense, thr):
    """config 5: keep list sorted by score; no kept pair reaches thr; every dropped box is covered by a kept box that
    outranks it (so the list is the greedy one: the greedy solution is unique given the order); NMS of the kept set
    keeps all of it (idempotence; the quirk can only take the last one)."""
    from vision_amd import ops
    b, s = wl.stress_boxes(0, 65536, dense=dense)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    keep, cnt = ops.nms_batched(bt[None], st[None], None, thr, False)
    k = keep[0, :int(cnt[0])]
    assert k.numel() > 0 and k.unique().numel() == k.numel()
    ks = st[k]
    assert bool((ks[1:] < ks[:-1]).all())                                  # descending (scores are tie-free)
    kb = bt[k]
    thr32 = torch.tensor(thr, dtype=torch.float32).item()
    # kept x kept, in row chunks
    rank = torch.empty(65536, dtype=torch.int64, device="cuda"); rank[torch.argsort(st, descending=True)] = torch.arange(65536, device="cuda")
    covered = torch.zeros(65536, dtype=torch.bool, device="cuda")
    for lo in range(0, k.numel(), 4096):
        iou = ops.iou_pairwise(kb[lo:lo + 4096], kb)                       # rows: kept chunk, cols: all kept
        idx = torch.arange(lo, min(lo + 4096, k.numel()), device="cuda")
        iou[torch.arange(idx.numel(), device="cuda"), idx] = 0.0
        assert bool((iou < thr32).all()), "two kept boxes overlap at or above the threshold"
        alli = ops.iou_pairwise(kb[lo:lo + 4096], bt)                      # rows: kept chunk, cols: every box
        hit = (~(alli < thr32)) & (rank[k[lo:lo + 4096]][:, None] < rank[None, :])
        covered |= hit.any(0)
    dropped = torch.ones(65536, dtype=torch.bool, device="cuda"); dropped[k] = False
    assert bool(covered[dropped].all()), "a dropped box is not suppressed by any higher-ranked kept box"
    keep2, cnt2 = ops.nms_batched(kb[None], ks[None], None, thr, False)
    assert int(cnt2[0]) == k.numel() and bool((keep2[0, :k.numel()] == torch.arange(k.numel(), device="cuda")).all())
    # the reference's numel()>1 rule drops at most the last kept box and nothing else
    keepq, cntq = ops.nms_batched(bt[None], st[None], None, thr, True)
    kq = keepq[0, :int(cntq[0])]
    assert int(cntq[0]) in (k.numel(), k.numel() - 1) and bool((kq == k[:kq.numel()]).all())


def test_pairwise_64k_properties(va):
    """config 5: the 64k x 64k matrix is symmetric bit for bit, its diagonal is a/(a+eps), entries are in [0,1]."""
    from vision_amd import ops
    b, _ = wl.stress_boxes(0, 65536, dense=True)
    bt = torch.from_numpy(b).cuda()
    m = ops.iou_pairwise(bt, bt)
    assert m.shape == (65536, 65536)
    for lo in range(0, 65536, 8192):
        blk = m[lo:lo + 8192]
        assert bool((blk >= 0).all()) and bool((blk <= 1).all())
        assert bool((blk == m[:, lo:lo + 8192].T).all())
    area = ((bt[:, 2] - bt[:, 0]) * (bt[:, 3] - bt[:, 1])).abs()
    assert bool((m.diagonal() == area / (((area + area) - area) + 1e-7)).all())


def test_roipool_batch_properties(va):
    """config 3 shape at 64 images x 2000 RoIs: each pooled value is the map value at its argmax, the argmax lies in
    its bin, nothing in the bin is larger; the backward conserves the gradient mass per (image, channel) and is linear."""
    from vision_amd import ops
    images = 64
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    img = torch.randn((images, 512, 38, 50), device=dev)
    rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i) for i in range(images)])).to(dev)
    idx = torch.arange(images, device=dev).repeat_interleave(2000)
    win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
    assert int(st.item()) == 0
    out, arg = ops.roipool_forward(img, win, "max")
    n = win.shape[0]
    for lo in range(0, n, 16000):
        hi = min(n, lo + 16000)
        w_ = win[lo:hi].long()
        a = arg[lo:hi].long()
        planes = img[w_[:, 4]]                                              # (m,512,38,50) view by gather would be huge: index flat
        flat = img.view(images, 512, -1)
        got = torch.gather(flat[w_[:, 4]], 2, a.view(hi - lo, 512, 49)).view(hi - lo, 512, 7, 7)
        del planes
        assert bool((got == out[lo:hi]).all())
        ay, ax = a // 50, a % 50
        i7 = torch.arange(7, device=dev)
        Lh, Lw = (w_[:, 3] - w_[:, 1])[:, None], (w_[:, 2] - w_[:, 0])[:, None]
        hs = w_[:, 1:2] + (i7[None] * Lh) // 7; he = w_[:, 1:2] + ((i7[None] + 1) * Lh + 6) // 7
        ws = w_[:, 0:1] + (i7[None] * Lw) // 7; we = w_[:, 0:1] + ((i7[None] + 1) * Lw + 6) // 7
        assert bool(((ay >= hs[:, None, :, None]) & (ay < he[:, None, :, None]) & (ax >= ws[:, None, None, :]) & (ax < we[:, None, None, :])).all())
    # bin maximum against torch's own adaptive max pool on a sample of RoIs
    for r in (0, 777, 64001, n - 1):
        x1, y1, x2, y2, b = [int(v) for v in win[r]]
        ref = torch.nn.functional.adaptive_max_pool2d(img[b:b + 1, :, y1:y2, x1:x2], 7)[0]
        assert bool((ref == out[r]).all())
    g1 = torch.randn_like(out); g2 = torch.randn_like(out)
    gi1 = ops.roipool_backward(g1, arg, win, img.shape, "max").clone()
    gi2 = ops.roipool_backward(g2, arg, win, img.shape, "max").clone()
    g12 = g1 + 2.0 * g2
    gi12 = ops.roipool_backward(g12, arg, win, img.shape, "max")
    torch.testing.assert_close(gi12, gi1 + 2.0 * gi2, rtol=1e-4, atol=1e-3)       # fp32 sums of ~100 terms, different order
    mass_out = torch.zeros((images, 512), device=dev, dtype=torch.float64)
    mass_out.index_add_(0, idx, g1.double().sum((2, 3)))
    torch.testing.assert_close(gi1.double().sum((2, 3)), mass_out, rtol=1e-5, atol=1e-2)


def test_kmeans_10m_properties(va):
    """config 4: one Lloyd step on 10 M boxes -- labels equal torch's fp32 argmin of the same formula (same op order,
    first minimum wins), counts add up to N, the fixed-point sums equal an exact integer re-summation."""
    from vision_amd import ops
    wh_np = wl.anchor_wh(0, 10_000_000)
    wh = torch.from_numpy(wh_np).cuda()
    cent = wh[torch.tensor([5, 77, 1234, 99999, 5000000], device="cuda")].clone()
    labels = torch.full((wh.shape[0],), 255, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(17, dtype=torch.int64, device="cuda")
    ops.kmeans_step(wh, cent, labels, acc)
    w, h = wh[:, 0:1], wh[:, 1:2]
    inter = torch.minimum(w, cent[None, :, 0]) * torch.minimum(h, cent[None, :, 1])
    uni = ((w * h) + (cent[None, :, 0] * cent[None, :, 1])) - inter
    d = 1.0 - inter / uni
    want = d.argmin(1)
    # argmin's tie order is not promised by torch: compare through the distance of the chosen centroid and the index rule
    dl = d.gather(1, labels.long()[:, None])[:, 0]
    assert bool((dl == d.min(1).values).all())
    first = (d == d.min(1, keepdim=True).values).float().argmax(1)
    assert bool((labels.long() == first).all())
    del want
    a = acc.cpu().numpy()
    assert a[10:15].sum() == wh.shape[0] and a[15] == wh.shape[0]             # counts, changed (all labels were 255)
    fx = torch.round(wh.double() * 4294967296.0).long()
    for c in range(5):
        sel = labels == c
        assert int(fx[sel, 0].sum()) == a[c] and int(fx[sel, 1].sum()) == a[5 + c] and int(sel.sum()) == a[10 + c]
    assert int(torch.round(dl.double() * 4294967296.0).long().sum()) == a[16]


def test_config3_full_size_roipool_256_images(va):
    """BASELINE configs[2] at its full size, 256 images x 2000 RoIs (512,000 RoIs; out + argmax + gradient = 154 GB, element
    offsets pass 2^32): forward value = map[argmax] with the argmax inside its bin and equal to torch's own adaptive max
    pool on sampled RoIs of every 16th image; backward equal to a torch scatter-add of the same (gradient, argmax) pairs
    for EVERY image (fp32 sums in another order: 1e-4), i.e. also mass-conserving."""
    from vision_amd import ops
    images, n_rois = 256, 2000
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    img = torch.randn((images, 512, 38, 50), device=dev)
    rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i, n_rois) for i in range(images)])).to(dev)
    idx = torch.arange(images, device=dev).repeat_interleave(n_rois)
    win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
    assert int(st.item()) == 0
    out, arg = ops.roipool_forward(img, win, "max")
    n = win.shape[0]
    assert n == images * n_rois and out.numel() > 2 ** 32
    flat = img.view(images, 512, -1)
    i7 = torch.arange(7, device=dev)
    for b in range(images):
        lo, hi = b * n_rois, (b + 1) * n_rois
        a = arg[lo:hi].long().view(n_rois, 512, 49)
        got = torch.gather(flat[b].unsqueeze(0).expand(n_rois, -1, -1), 2, a)
        assert bool((got == out[lo:hi].view(n_rois, 512, 49)).all()), b
        w_ = win[lo:hi].long()
        assert bool((w_[:, 4] == b).all())
        a = a.view(n_rois, 512, 7, 7)
        ay, ax = a // 50, a % 50
        Lh, Lw = (w_[:, 3] - w_[:, 1])[:, None], (w_[:, 2] - w_[:, 0])[:, None]
        hs = w_[:, 1:2] + (i7[None] * Lh) // 7; he = w_[:, 1:2] + ((i7[None] + 1) * Lh + 6) // 7
        ws = w_[:, 0:1] + (i7[None] * Lw) // 7; we = w_[:, 0:1] + ((i7[None] + 1) * Lw + 6) // 7
        assert bool(((ay >= hs[:, None, :, None]) & (ay < he[:, None, :, None]) & (ax >= ws[:, None, None, :]) & (ax < we[:, None, None, :])).all()), b
        if b % 16 == 15:
            for r in (lo, lo + 777, hi - 1):
                x1, y1, x2, y2, bb = [int(v) for v in win[r]]
                ref, ri = torch.nn.functional.adaptive_max_pool2d(img[bb:bb + 1, :, y1:y2, x1:x2], 7, return_indices=True)
                assert bool((ref[0] == out[r]).all())
                # torch's index is relative to the crop
                ry, rx = ri[0] // (x2 - x1) + y1, ri[0] % (x2 - x1) + x1
                assert bool((ry * 50 + rx == arg[r].long()).all())
    g = torch.randn_like(out)
    gin = ops.roipool_backward(g, arg, win, img.shape, "max")
    for b in range(images):
        lo, hi = b * n_rois, (b + 1) * n_rois
        want = torch.zeros((512, 38 * 50), device=dev, dtype=torch.float64)
        want.scatter_add_(1, arg[lo:hi].long().view(n_rois, 512, 49).permute(1, 0, 2).reshape(512, -1),
                          g[lo:hi].view(n_rois, 512, 49).permute(1, 0, 2).reshape(512, -1).double())
        torch.testing.assert_close(gin[b].view(512, -1).double(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("thr", [0.7, 0.3])
def test_config3_full_size_batched_nms_256_images_vs_oracle(va, thr):
    """BASELINE configs[2]: batched NMS over 256 images x 2000 proposals in one call, every image against the C oracle."""
    from oracle import oracle as orc
    from vision_amd import ops
    orc.build()
    images, n = 256, 2000
    props = [wl.frcnn_proposals(i, n) for i in range(images)]
    boxes = torch.from_numpy(np.stack([b for b, _ in props])).cuda()
    scores = torch.from_numpy(np.stack([s for _, s in props])).cuda()
    keep, cnt = ops.nms_batched(boxes, scores, None, thr, True)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for i, (b, s) in enumerate(props):
        want = orc.nms(b, s, thr)
        assert cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want), i


# ---------------------------------------------------------------- round 3: the oracle itself at full size
@pytest.mark.parametrize("dense,thr,quirk", [(False, 0.5, True), (False, 0.7, True), (False, 0.3, False), (True, 0.5, True), (True, 0.7, False)])
def test_nms_64k_vs_oracle(va, dense, thr, quirk):
    """config 5 (yolo/yolo_v1/yolo.py:188-203 on 65,536 boxes): the keep list, bit for bit, against the C oracle; sparse sets
    and dense sets alike take the pruned pair search + fixed-point keep list at these thresholds, and the test checks that."""
    from oracle import oracle as orc
    from vision_amd import ops
    orc.build()
    b, s = wl.stress_boxes(0, 65536, dense=dense)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    keep, cnt = ops.nms_batched(bt[None], st[None], None, thr, quirk)
    got = keep[0, :int(cnt[0])].cpu().numpy()
    mode, fpdone, rounds, npairs, cpb, est = ops.nms_path_info(bt.device, 1, 65536)
    # all five take the pruned path (the dense set has about 70 suppressing pairs per box at 0.5, under the 128 the estimate allows)
    assert mode == 0 and fpdone == 1 and 0 < rounds < 64 and npairs > 0, (mode, fpdone, rounds, npairs, cpb, est)
    assert 0.5 * npairs <= est * 65536 + 65536 and est * 65536 <= 2 * npairs + 65536, (npairs, est)   # the sampled estimate is about right
    want = orc.nms(b, s, thr, quirk)
    assert len(got) == len(want) and np.array_equal(got, want)


def test_nms_64k_fallbacks_vs_oracle(va, monkeypatch):
    """The same 64k sparse set through the pruned path's serial fallback (the fixed point cut off after 2 rounds) and with
    the pruned path forced onto dense boxes: same keep lists as the oracle."""
    from oracle import oracle as orc
    from vision_amd import ops
    orc.build()
    b, s = wl.stress_boxes(0, 65536, dense=False)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    want = orc.nms(b, s, 0.5, True)
    monkeypatch.setenv("DG_NMS_FP_ROUNDS", "2")
    keep, cnt = ops.nms_batched(bt[None], st[None], None, 0.5, True)
    assert ops.nms_path_info(bt.device, 1, 65536)[:2] == (0, 0)
    assert np.array_equal(keep[0, :int(cnt[0])].cpu().numpy(), want)
    monkeypatch.delenv("DG_NMS_FP_ROUNDS")
    monkeypatch.setenv("DG_NMS_PRUNE", "2")
    b, s = wl.stress_boxes(0, 65536, dense=True)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda()
    keep, cnt = ops.nms_batched(bt[None], st[None], None, 0.7, True)
    assert ops.nms_path_info(bt.device, 1, 65536)[:2] == (0, 1)
    assert np.array_equal(keep[0, :int(cnt[0])].cpu().numpy(), orc.nms(b, s, 0.7, True))
    # 0.3 on the dense set: about 400 pairs per box -- the estimate sends the image to the dense mask (unless forced: above)
    monkeypatch.setenv("DG_NMS_PRUNE", "1")
    keep, cnt = ops.nms_batched(bt[None], st[None], None, 0.3, True)
    info = ops.nms_path_info(bt.device, 1, 65536)
    assert info[0] == 1 and info[5] > 128, info
    assert np.array_equal(keep[0, :int(cnt[0])].cpu().numpy(), orc.nms(b, s, 0.3, True))


def test_pairwise_64k_rows_vs_oracle(va):
    """config 5: 2,048 sampled rows of the 64k x 64k IoU matrix (yolo/yolo_v1/modules.py:73-98 per entry), bit for bit."""
    from oracle import oracle as orc
    from vision_amd import ops
    orc.build()
    b, _ = wl.stress_boxes(0, 65536, dense=True)
    bt = torch.from_numpy(b).cuda()
    m = ops.iou_pairwise(bt, bt)
    rows = np.sort(np.random.default_rng(5).choice(65536, 2048, replace=False))
    got = m[torch.from_numpy(rows).cuda()].cpu().numpy()
    want = orc.iou_pairwise(b[rows], b)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_kmeans_10m_fit30_vs_oracle(va):
    """config 4 as bench.py times it: kmean(5, 42).fit(wh_10M, fixed_iters=30) against the oracle's 30 Lloyd iterations --
    labels, centroids, inertia, bit for bit (integer sums; spec in oracle.c, no reference counterpart: parity unpinned)."""
    from oracle import oracle as orc
    from vision_amd.cluster import kmean
    orc.build()
    wh = wl.anchor_wh(0, 10_000_000)
    km = kmean(n_clusters=5, random_state=42).fit(torch.from_numpy(wh).cuda(), fixed_iters=30)
    cent, labels, n_iter, inertia = orc.kmeans_fit(wh, 5, 42, fixed_iters=30)
    assert km.n_iter_ == n_iter == 30
    assert np.array_equal(np.asarray(km.cluster_centers_, np.float32).view(np.uint32), cent.view(np.uint32))
    got_labels = km.labels_.cpu().numpy() if isinstance(km.labels_, torch.Tensor) else np.asarray(km.labels_)
    assert np.array_equal(got_labels.astype(np.int32), labels)
    assert float(km.inertia_) == inertia
