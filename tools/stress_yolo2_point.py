"""Randomised differential test of yolo_v2.get_point (yolo/yolo_v2/utils.py:228-234: torch's floor-divide and remainder on fp32)
against the oracle AND against torch's own operators on the CPU: centres on exact cell multiples, negative, far outside the image,
tiny, huge, with infinities and NaN.  usage: python tools/stress_yolo2_point.py [rounds] [seed]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import yolo_v2
from oracle import oracle
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
bad = 0
# (centres beyond ~2^24 cells are left out: there torch's vectorised fmod (Sleef) and its scalar tail give different results for
#  the same value -- NaN against the exact remainder at 3.4e38 -- so the reference itself has no single answer)
special = np.array([0.0, -0.0, 1.0 / 13, 2.0 / 13, 1.0, -1.0 / 13, 1e-30, -1e-30, 1e5, -1e5, 7e5, np.inf, -np.inf, np.nan, 12.0 / 13, 13.0 / 13], np.float32)
cell = torch.tensor(1.0 / 13.0, dtype=torch.float32)      # what the reference's 1/13 becomes next to an fp32 tensor
for it in range(rounds):
    n = int(rng.integers(1, 5000))
    kind = rng.integers(0, 4, (n, 4))
    k13 = (rng.integers(-30, 40, (n, 4)) / 13.0).astype(np.float32)
    b = np.where(kind == 0, rng.uniform(-2, 3, (n, 4)), np.where(kind == 1, k13, np.where(kind == 2, rng.uniform(0, 1, (n, 4)),
                 special[rng.integers(0, len(special), (n, 4))]))).astype(np.float32)
    pts, enc = yolo_v2.get_point(56, torch.from_numpy(b).cuda())
    pts, enc = pts.cpu().numpy(), enc.cpu().numpy()
    opts, oenc = oracle.yolo2_get_point(b)
    t = torch.from_numpy(b)
    center = t[:, :2] + t[:, 2:] / 2
    tq = torch.floor_divide(center, 1 / 13)
    tr = torch.remainder(center, 1 / 13) / (1 / 13)
    # the integer cast of a non-finite / huge quotient is platform-defined: compare the points where the quotient fits int64
    fin = np.isfinite(tq.numpy()) & (np.abs(tq.numpy()) < 9e18)
    nn = ~(np.isnan(enc) & np.isnan(oenc))                                    # (a NaN's sign and payload are not part of the contract)
    ok_o = np.array_equal(pts[fin], opts[fin]) and np.array_equal(enc.view(np.uint32)[nn], oenc.view(np.uint32)[nn])
    ok_t = np.array_equal(pts[fin], tq.numpy().astype(np.int64)[fin]) and np.array_equal(enc[:, :2], tr.numpy(), equal_nan=True)
    if not (ok_o and ok_t):
        bad += 1
        if bad <= 5:
            d = np.nonzero((enc[:, :2].view(np.uint32) != tr.numpy().view(np.uint32)) & ~(np.isnan(enc[:, :2]) & np.isnan(tr.numpy())))
            do = np.nonzero((enc.view(np.uint32) != oenc.view(np.uint32)) & nn)
            print("  vs oracle: points equal", np.array_equal(pts[fin], opts[fin]), "enc diffs", [(b[i].tolist(), int(j), float(enc[i, j]), float(oenc[i, j])) for i, j in zip(*do)][:4])
            dp = np.nonzero((pts != opts) & fin)
            print("  point diffs", [(b[i].tolist(), int(j), int(pts[i, j]), int(opts[i, j])) for i, j in zip(*dp)][:4])
            print("round", it, "oracle", ok_o, "torch", ok_t, "first diffs", [(b[i].tolist(), enc[i, j], float(tr[i, j])) for i, j in zip(*d)][:3])
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
