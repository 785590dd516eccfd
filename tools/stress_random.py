"""Randomised differential test of every kernel family against the oracle: random shapes, seeds, thresholds,
channel counts, NaN maps, ragged batches.  One-off (minutes of GPU time), complements tests/ with breadth."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
from oracle import oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2024)
fail = []
t0 = time.time()
def chk(name, ok, info=""):
    if not ok:
        fail.append((name, info)); print("MISMATCH", name, info, flush=True)

for it in range(rounds):
    # ---- RoI pooling: random map size, channels, batch, windows incl. wide/tall, optional NaN
    B = int(rng.integers(1, 4)); Cc = int(rng.choice([3, 64, 70, 128, 192])); H = int(rng.integers(8, 41)); W = int(rng.integers(8, 56))
    img = rng.standard_normal((B, Cc, H, W), dtype=np.float32)
    if rng.random() < 0.3: img = np.round(img * 2) / 2                      # ties
    if rng.random() < 0.3: img[rng.random(img.shape) < 0.003] = np.nan
    n = int(rng.integers(1, 300))
    x1 = rng.integers(0, W, n); y1 = rng.integers(0, H, n)
    x2 = np.minimum(W, x1 + rng.integers(1, W + 1, n)); y2 = np.minimum(H, y1 + rng.integers(1, H + 1, n))
    rois = np.stack([(x1 + 0.3) / W, (y1 + 0.3) / H, (x2 - 0.3) / W, (y2 - 0.3) / H], 1)
    idx = rng.integers(0, B, n).astype(np.int64)
    x = torch.from_numpy(img).cuda()
    win, st = ops.roi_windows(rois, idx, B, H, W, x.device)
    P = 7 if rng.random() < 0.7 else [3, (2, 5), 10, 1, (9, 4)][int(rng.integers(5))]      # mostly the tuned 7x7 path
    for mode in ("max", "avg"):
        out, arg = ops.roipool_forward(x, win, mode, P=P)
        want, warg = oracle.roipool_fwd(img, rois, idx, P, mode)
        if mode == "max":
            chk("roi fwd max", np.array_equal(out.cpu().numpy(), want, equal_nan=True) and np.array_equal(arg.cpu().numpy(), warg), (B, Cc, H, W, n, P))
        else:
            chk("roi fwd avg", np.allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5, equal_nan=True), (B, Cc, H, W, n))
        if not np.isnan(img).any():
            g = rng.standard_normal(want.shape, dtype=np.float32)
            gin = ops.roipool_backward(torch.from_numpy(g).cuda(), arg, win, img.shape, mode, P=P).cpu().numpy()
            wgin = oracle.roipool_bwd(g, warg, img.shape, rois, idx, P, mode)
            chk("roi bwd " + mode, np.allclose(gin, wgin, rtol=1e-4, atol=1e-3), (B, Cc, H, W, n))
    # ---- NMS: ragged batch, random sizes crossing the kernel-selection boundaries, random threshold
    nb = int(rng.integers(1, 5)); N = int(rng.choice([5, 98, 447, 449, 700, 2000, 4100, 9000]))
    counts = rng.integers(0, N + 1, nb).astype(np.int32); counts[0] = N
    boxes = np.zeros((nb, N, 4), np.float32); scores = np.zeros((nb, N), np.float32)
    for i in range(nb):
        b, s = wl.stress_boxes(int(rng.integers(1 << 30)), N, dense=bool(rng.integers(2)))
        boxes[i], scores[i] = b, s
    thr = float(rng.choice([0.05, 0.3, 0.5, 0.7, 0.95])); quirk = bool(rng.integers(2))
    keep, cnt = ops.nms_batched(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts), thr, quirk)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for i in range(nb):
        wantk = oracle.nms(boxes[i, :counts[i]], scores[i, :counts[i]], thr, quirk)
        chk("nms", cnt[i] == len(wantk) and np.array_equal(keep[i, :cnt[i]], wantk), (N, int(counts[i]), thr, quirk))
    # ---- per-class NMS on the same boxes: the oracle's nms applied per class
    Ccls = int(rng.choice([1, 3, 20, 80]))
    labels = rng.integers(0, Ccls, (nb, N)).astype(np.int64)
    keep, cnt, ccnt, stt = ops.nms_batched_per_class(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(labels).cuda(),
                                                     Ccls, torch.from_numpy(counts), thr, quirk)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for i in range(nb):
        m = int(counts[i]); wk = []
        for c in range(Ccls):
            ii = np.flatnonzero(labels[i, :m] == c)
            if ii.size:
                wk.append(ii[oracle.nms(boxes[i, ii], scores[i, ii], thr, quirk)])
        wk = np.concatenate(wk) if wk else np.zeros(0, np.int64)
        chk("nms per class", cnt[i] == len(wk) and np.array_equal(keep[i, :cnt[i]], wk), (N, m, Ccls, thr, quirk))
    # ---- IoU pairwise / elementwise
    a = wl.random_boxes(int(rng.integers(1 << 30)), int(rng.integers(1, 400))); b = wl.random_boxes(int(rng.integers(1 << 30)), int(rng.integers(1, 3000)))
    chk("pairwise", np.array_equal(ops.iou_pairwise(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy(), oracle.iou_pairwise(a, b)))
    # ---- k-means step: random n, k, unaligned shard start
    k = int(rng.integers(1, 17)); nk = int(rng.integers(1, 300000)); off = int(rng.integers(0, 3))
    wh = wl.anchor_wh(int(rng.integers(1 << 30)), nk + off)
    cent = wh[rng.choice(nk, k, replace=(nk < k))].copy()
    lab = rng.integers(0, k, nk).astype(np.uint8); lab0 = lab.copy()
    acc = oracle.kmeans_step(wh[off:], cent, lab)
    whd = torch.from_numpy(wh).cuda()[off:]
    l2 = torch.from_numpy(lab0).cuda(); a2 = torch.zeros(3 * k + 2, dtype=torch.int64, device="cuda")
    ops.kmeans_step(whd, torch.from_numpy(cent).cuda(), l2, a2)
    chk("kmeans", np.array_equal(a2.cpu().numpy(), acc) and np.array_equal(l2.cpu().numpy(), lab), (nk, k, off))
    # ---- the sharded Lloyd loop on the device (dg_kmeans_iou_iter / _finish; no process group: one shard), fixed iterations
    if it % 4 == 0:
        from vision_amd import cluster
        k2 = int(rng.choice([2, 5, 9, 13])); n2 = int(rng.integers(k2 + 1, 120000)); its = int(rng.integers(1, 12))
        wh2 = wl.anchor_wh(int(rng.integers(1 << 30)), n2)
        km = cluster.kmean(k2, 42, sharded=True).fit(torch.from_numpy(wh2).cuda(), fixed_iters=its)
        c2, l2_, n_it, inert = oracle.kmeans_fit(wh2, k2, 42, fixed_iters=its)
        chk("kmeans sharded loop", np.array_equal(km.cluster_centers_, c2) and np.array_equal(km.labels_, l2_) and abs(km.inertia_ - inert) < 1e-9, (n2, k2, its))
    # ---- YOLOv1 decode + loss
    bs = int(rng.integers(1, 40)); Cn = int(rng.choice([1, 5, 20]))
    out, tgt = wl.yolo1_batch(int(rng.integers(1 << 30)), bs, Cn)
    ol, ocomp, ograd = oracle.yolo1_loss(out, tgt, quirk=True)
    loss6, grad = ops.yolo1_loss(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), True)
    chk("loss", abs(loss6[0].item() - ol) <= 3e-7 * abs(ol) and np.allclose(grad.cpu().numpy(), ograd, rtol=3e-6, atol=1e-7), (bs, Cn))
    # ---- fused YOLOv1 decode + NMS (one launch) against the oracle's decode followed by its nms (class-agnostic / per class)
    thr1 = float(rng.choice([0.05, 0.3, 0.5, 0.9])); q1 = bool(rng.integers(2)); pc = bool(rng.integers(2))
    ct, pt = float(rng.choice([0.2, 0.5, 0.9])), float(rng.choice([0.01, 0.1, 0.4]))
    if rng.random() < 0.5: out = rng.random(out.shape, dtype=np.float32)            # dense candidates (up to 98 per image)
    f = ops.yolo1_decode_batched(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), ct, pt, thr1, pc, q1)
    fb, fl, fs, fc, fk, fkc = [f[i].cpu().numpy() for i in (0, 1, 3, 4, 7, 8)]
    for i in range(bs):
        b_, l_, c_, s_, tb_, tl_ = oracle.yolo1_decode(out[i], tgt[i], ct, pt)
        ok = fc[i] == len(b_) and (len(b_) == 0 or (np.array_equal(fb[i, :fc[i]], b_) and np.array_equal(fl[i, :fc[i]], l_)))
        if ok and len(b_):
            b_, l_, s_ = np.asarray(b_, np.float32), np.asarray(l_), np.asarray(s_, np.float32)
            if pc:
                wk = [ii[oracle.nms(b_[ii], s_[ii], thr1, q1)] for ii in (np.flatnonzero(l_ == c) for c in range(Cn)) if ii.size]
                wk = np.concatenate(wk) if wk else np.zeros(0, np.int64)
            else:
                wk = oracle.nms(b_, s_, thr1, q1)
            ok = fkc[i] == len(wk) and np.array_equal(fk[i, :fkc[i]], wk)
        elif ok:
            ok = fkc[i] == 0
        chk("fused decode+nms", ok, (bs, Cn, i, thr1, q1, pc, ct, pt))
    # ---- VOC AP: random detections / targets, thresholds incl. one that sits exactly on an occurring IoU
    import tests.test_gpu_parity as T
    from vision_amd import yolo_v1
    det, tg = T._ap_case(int(rng.integers(1 << 30)), int(rng.integers(1, 400)), int(rng.integers(1, 8)), int(rng.integers(0, 12)),
                         empty_classes=(0,) if rng.random() < 0.3 else ())
    for thr in (0.05, 0.5, float(rng.uniform(0.01, 0.9))):
        got, want = yolo_v1.ap(det, tg, thr), oracle.average_precision(det, tg, thr)
        chk("ap", len(got) == len(want) and np.array_equal(np.array(got), np.array(want, np.float64), equal_nan=True), thr)
    print(f"round {it}: failures {len(fail)}  ({time.time() - t0:.0f} s)", flush=True)
print("TOTAL FAILURES", len(fail), fail[:10])
sys.exit(1 if fail else 0)
