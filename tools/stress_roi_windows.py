"""Randomised differential test of dg_roi_windows (rcnn/fast_rcnn.py:366-369 + python's slice clamping and negative-index wrap)
against the oracle's orc_roi_window: RoIs far outside [0, 1], reversed, on exact cell boundaries (k / W), a hair below zero
(floor -> -1 -> wraps to the last column; a negative stop wraps too), NaN and infinities (numpy's astype(int) makes INT64_MIN of them on x86-64: bound 0).
usage: python tools/stress_roi_windows.py [rounds] [seed]"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from oracle import oracle
oracle.build()
lib = oracle.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for it in range(rounds):
    H, W = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    n = int(rng.integers(1, 400))
    kind = rng.integers(0, 4, (n, 4))
    grid = np.stack([rng.integers(-2, W + 3, n) / W, rng.integers(-2, H + 3, n) / H, rng.integers(-2, W + 3, n) / W, rng.integers(-2, H + 3, n) / H], 1)
    rois = np.where(kind == 0, rng.uniform(-1.5, 2.5, (n, 4)), np.where(kind == 1, grid, np.where(kind == 2, rng.uniform(0, 1, (n, 4)),
                    rng.choice(np.array([-1e-18, 1e-18, 1 - 1e-16, 1 + 1e-15, 0.0, 1.0, -1.0, 2.0, 1e12, -1e12, 1e300, -1e300, np.nan, np.inf, -np.inf, -0.4, -0.02]), (n, 4)))))
    rois = np.ascontiguousarray(rois, np.float64)
    idx = rng.integers(-3, 3, n).astype(np.int64)                # negative: counts from the last image, as img[roi_idx[i]] does
    win, status = ops.roi_windows(rois, idx, 3, H, W, torch.device("cuda", 0))
    win = win.cpu().numpy(); st = int(status.item())
    any_empty = False
    for r in range(n):
        x1, y1, x2, y2 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = lib.orc_roi_window(rois[r].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(H), C.c_int(W), C.byref(x1), C.byref(y1), C.byref(x2), C.byref(y2))
        if rc != 0:
            any_empty = True
            continue
        want = (x1.value, y1.value, x2.value, y2.value, int(idx[r]) % 3)
        if tuple(int(v) for v in win[r]) != want:
            bad += 1
            if bad <= 10: print("MISMATCH", H, W, rois[r].tolist(), "got", win[r].tolist(), "want", want)
    if bool(st & 1) != any_empty:
        bad += 1
        if bad <= 10: print("STATUS MISMATCH", H, W, st, any_empty)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
