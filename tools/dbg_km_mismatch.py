"""Debug: replay tools/stress_kmeans.py's random stream up to a round, then step the box loop and the oracle side by side and print
the first boxes whose labels differ.  usage: python tools/dbg_km_mismatch.py <round> [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
from oracle import oracle as orc
orc.build()
target = int(sys.argv[1]); rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
def boxes(kind, seed, n):
    r = np.random.default_rng(seed)
    if kind == 0: return wl.anchor_wh(seed % 97, n)
    if kind == 1: return r.uniform(0.01, 1.0, (n, 2)).astype(np.float32)
    if kind == 2:
        base = r.uniform(0.02, 0.9, (int(r.integers(20, 400)), 2)).astype(np.float32)
        return base[r.integers(0, base.shape[0], n)]
    wh = wl.anchor_wh(seed % 97, n)
    idx = r.integers(0, n, max(1, n // 500))
    wh[idx] = r.choice(np.array([[3.0, 0.5], [0.0, 0.2], [1e-4, 1e-4], [-0.3, 0.4], [0.5, 7.0]], np.float32), idx.shape[0])
    return wh
for rnd in range(target + 1):
    n = int(rng.choice([1000, 5000, 70001, 250000, 600000, 1500000])); k = int(rng.integers(1, 17)); kind = int(rng.integers(0, 4))
    wh = boxes(kind, int(rng.integers(0, 1 << 30)), n)
    if rng.random() < 0.5: kw = {"fixed_iters": int(rng.integers(1, 45))}
    else: kw = {"max_iter": int(rng.choice([1, 7, 8, 9, 16, 17, 31, 32, 33, 64, 300]))}
print("round", target, "n", n, "k", k, "kind", kind, kw)
cent = orc.kmeans_init(wh, k, 42)
print("init centroids", cent.tolist())
lab = np.full(n, 255, np.uint8)
x = torch.from_numpy(wh).cuda(); labd = torch.full((n,), 255, dtype=torch.uint8, device="cuda")
for it in range(40):
    acc = orc.kmeans_step(wh, cent, lab)
    accd = torch.zeros(3 * k + 2, dtype=torch.int64, device="cuda")
    ops.kmeans_step(x, torch.from_numpy(cent).cuda(), labd, accd)
    lg = labd.cpu().numpy()
    d = np.nonzero(lg != lab)[0]
    if d.size or not np.array_equal(accd.cpu().numpy(), acc):
        print("iteration", it, ":", d.size, "labels differ; acc equal:", np.array_equal(accd.cpu().numpy(), acc))
        print("centroids", cent.tolist())
        for i in d[:8]:
            w, h = wh[i]
            ds = [1.0 - (min(w, c[0]) * min(h, c[1])) / ((w * h + c[0] * c[1]) - min(w, c[0]) * min(h, c[1])) for c in cent.astype(np.float64)]
            print("  box", i, wh[i].tolist(), "oracle", int(lab[i]), "gpu", int(lg[i]), "d(f64)", [round(float(v), 7) for v in ds])
        break
    cent = orc.kmeans_update(acc, cent)
else:
    print("no difference in 40 iterations")
