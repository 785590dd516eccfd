"""The pruned NMS path (x-bucket band search + fixed-point keep list) against the dense-mask path of the same library, in
one process: DG_NMS_PRUNE / DG_NMS_FP_ROUNDS are read on every call.  Keep lists must be identical."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi
from tools import workloads as wl

dev = torch.device("cuda", 0)

def run(b, s, counts, thr, quirk, prune, rounds=None):
    os.environ["DG_NMS_PRUNE"] = str(prune)
    if rounds is None: os.environ.pop("DG_NMS_FP_ROUNDS", None)
    else: os.environ["DG_NMS_FP_ROUNDS"] = str(rounds)
    k, c = ops.nms_batched(b, s, counts, thr, quirk)
    torch.cuda.synchronize()
    c = c.cpu().numpy(); k = k.cpu().numpy()
    return [k[i, :c[i]].copy() for i in range(len(c))]

def same(a, b): return all(np.array_equal(x, y) for x, y in zip(a, b))

cases = []
for n in (65536, 20000, 5000, 4200):
    for dense in (False, True):
        b, s = wl.stress_boxes(3 + n, n, dense=dense)
        cases.append((f"stress n={n} dense={dense}", b[None], s[None], None))
b = wl.random_boxes(5, 9000); s = wl.tiefree_scores(6, 9000); cases.append(("random_boxes 9000 (degenerate rows)", b[None], s[None], None))
bb = np.stack([wl.stress_boxes(40 + i, 6000)[0] for i in range(3)]); ss = np.stack([wl.stress_boxes(40 + i, 6000)[1] for i in range(3)])
cases.append(("ragged 3x6000", bb, ss, np.array([6000, 4500, 4100], np.int32)))
# a ladder: box i overlaps only i+1 strongly (a chain: the fixed point needs many rounds)
n = 5000; x = np.arange(n, dtype=np.float32) * 0.45; lad = np.stack([x, np.zeros(n, np.float32), x + 1.0, np.ones(n, np.float32)], 1)
cases.append(("ladder 5000", lad[None], np.linspace(1, 0.1, n, dtype=np.float32)[None], None))
# all boxes identical / all share x1
same_b = np.tile(np.array([[0.1, 0.1, 0.5, 0.5]], np.float32), (4500, 1)); cases.append(("identical 4500", same_b[None], wl.tiefree_scores(9, 4500)[None], None))
bad = 0
for name, b, s, counts in cases:
    bt, st = torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)
    ct = torch.from_numpy(counts).to(dev) if counts is not None else None
    for thr in (0.3, 0.5, 0.7):
        for quirk in (True, False):
            ref = run(bt, st, ct, thr, quirk, 0)
            for prune, rounds in ((1, None), (2, None), (2, 0), (2, 1), (2, 3)):
                got = run(bt, st, ct, thr, quirk, prune, rounds)
                ok = same(ref, got)
                if not ok:
                    bad += 1
                    print("MISMATCH", name, thr, quirk, prune, rounds, [len(x) for x in ref], [len(x) for x in got], flush=True)
    print(name, "kept", [len(x) for x in ref], "ok" if not bad else f"bad so far {bad}", flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
