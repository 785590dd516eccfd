"""Randomised differential test of the large-image NMS (4097..65536 boxes: pruned pair search, fixed-point keep list, their
fallbacks and the dense mask) against the oracle: random sizes, ragged batches, box mixtures (sparse, dense, unit-square with
degenerate rows, clustered, duplicated, scaled), thresholds (round values and exact IoU values of the set), both quirk
settings, random DG_NMS_PRUNE / DG_NMS_FP_ROUNDS.  usage: python tools/stress_nms.py [rounds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
from oracle import oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
fails, paths, t0 = 0, {}, time.time()

def boxes_of(kind, seed, n):
    if kind == 0: return wl.stress_boxes(seed, n, dense=False)
    if kind == 1: return wl.stress_boxes(seed, n, dense=True)
    if kind == 2: return wl.random_boxes(seed, n), wl.tiefree_scores(seed + 1, n)
    r = np.random.default_rng(seed)
    if kind == 3:                                          # clusters in x: many boxes per bucket, long runs of partners
        cx = r.choice(r.uniform(0, 1000, 12), n) + r.normal(0, 3, n); cy = r.uniform(0, 1000, n)
        wh = r.uniform(2, 40, (n, 2))
        b = np.stack([cx - wh[:, 0] / 2, cy - wh[:, 1] / 2, cx + wh[:, 0] / 2, cy + wh[:, 1] / 2], 1).astype(np.float32)
    elif kind == 4:                                        # every box several times over (IoU exactly 1 among copies)
        m = max(1, n // 3)
        base = wl.stress_boxes(seed, m, dense=False)[0]
        b = base[r.integers(0, m, n)]
    else:                                                  # wide flat boxes: everything overlaps in x, little in y
        x1 = r.uniform(0, 100, n); y1 = r.uniform(0, 5000, n)
        b = np.stack([x1, y1, x1 + r.uniform(400, 900, n), y1 + r.uniform(1, 6, n)], 1).astype(np.float32)
    return b, wl.tiefree_scores(seed + 1, n)

for it in range(rounds):
    nb = int(rng.integers(1, 4))
    N = int(rng.choice([4097, 4160, 4200, 5000, 6500, 8192, 8193, 12000, 16384, 20000, 33000]))
    if N > 20000: nb = 1
    counts = rng.integers(65, N + 1, nb).astype(np.int32); counts[0] = N
    if rng.random() < 0.2 and nb > 1: counts[-1] = int(rng.integers(0, 65))
    boxes = np.zeros((nb, N, 4), np.float32); scores = np.zeros((nb, N), np.float32)
    kinds = []
    for i in range(nb):
        kind = int(rng.integers(0, 6)); kinds.append(kind)
        b, s = boxes_of(kind, int(rng.integers(1 << 30)), N)
        boxes[i], scores[i] = b, s
    if rng.random() < 0.5:
        thr = float(rng.choice([0.05, 0.3, 0.5, 0.7, 0.95]))
    else:                                                   # exactly an IoU value of the set, or a float neighbour
        v = oracle.iou_pairwise(boxes[0, :64], boxes[0, :512]).ravel(); v = v[(v > 0) & (v < 1)]
        thr = float(v[rng.integers(len(v))]) if len(v) else 0.5
        thr = float(np.nextafter(np.float32(thr), np.float32(rng.choice([0.0, 1.0, thr]))))
    quirk = bool(rng.integers(2))
    os.environ["DG_NMS_PRUNE"] = str(rng.choice([1, 1, 2]))
    os.environ["DG_NMS_FP_ROUNDS"] = str(rng.choice([128, 128, 0, 1, 2, 5]))
    bt, st_, ct = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), torch.from_numpy(counts)
    keep, cnt = ops.nms_batched(bt, st_, ct, thr, quirk)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for i in range(nb):
        info = ops.nms_path_info(bt.device, nb, N, i)
        paths[(info[0], info[1])] = paths.get((info[0], info[1]), 0) + 1
        want = oracle.nms(boxes[i, :counts[i]], scores[i, :counts[i]], thr, quirk)
        if not (cnt[i] == len(want) and np.array_equal(keep[i, :cnt[i]], want)):
            fails += 1
            print("MISMATCH", it, i, N, int(counts[i]), kinds[i], thr, quirk, os.environ["DG_NMS_PRUNE"], os.environ["DG_NMS_FP_ROUNDS"], info, int(cnt[i]), len(want), flush=True)
    if it % 10 == 9:
        print(f"round {it + 1}: {fails} mismatches, paths (mode, fixed point) {paths}, {time.time() - t0:.0f} s", flush=True)
print("TOTAL MISMATCHES", fails, "paths", paths)
sys.exit(1 if fails else 0)
