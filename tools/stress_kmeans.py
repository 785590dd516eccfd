"""Randomised differential test of kmean.fit's prepared path (boxes grouped by (w, h) cell; pieces "unset" at the start, count-only
queue entries, chunks that grow and run over cells only, the closing plain step) against the box loop of the same library and, for
the smaller sets, the oracle: random sizes, k, seeds, box mixtures (the anchor set, uniform, a few hundred distinct boxes many times
over, boxes outside the grid), run until convergence with a random max_iter or for a fixed number of iterations.
usage: python tools/stress_kmeans.py [rounds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
from oracle import oracle as orc
orc.build()

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
bad, t0 = 0, time.time()

def boxes(kind, seed, n):
    r = np.random.default_rng(seed)
    if kind == 0: return wl.anchor_wh(seed % 97, n)
    if kind == 1: return r.uniform(0.01, 1.0, (n, 2)).astype(np.float32)
    if kind == 2:                                        # few distinct boxes: whole cells of identical boxes, many exact ties between iterations
        base = r.uniform(0.02, 0.9, (int(r.integers(20, 400)), 2)).astype(np.float32)
        return base[r.integers(0, base.shape[0], n)]
    wh = wl.anchor_wh(seed % 97, n)                      # with boxes outside the grid: the catch-all cells
    idx = r.integers(0, n, max(1, n // 500))
    wh[idx] = r.choice(np.array([[3.0, 0.5], [0.0, 0.2], [1e-4, 1e-4], [-0.3, 0.4], [0.5, 7.0], [-0.2, -0.3], [0.4, -0.1], [1e6, 0.3], [6e5, 7e5], [2e-30, 0.5], [-1e6, 2.0]], np.float32), idx.shape[0])
    return wh

def fit(x, k, cells, **kw):
    os.environ["DG_KMEANS_CELLS"] = "1" if cells else "0"
    mi = kw.pop("max_iter", 300)
    return cluster.kmean(k, 42, max_iter=mi).fit(x, **kw)

for rnd in range(rounds):
    n = int(rng.choice([1000, 5000, 70001, 250000, 600000, 1500000]))
    k = int(rng.integers(1, 17))
    kind = int(rng.integers(0, 4))
    wh = boxes(kind, int(rng.integers(0, 1 << 30)), n)
    x = torch.from_numpy(wh).cuda()
    if rng.random() < 0.5:
        kw = {"fixed_iters": int(rng.integers(1, 45))}
    else:
        kw = {"max_iter": int(rng.choice([1, 7, 8, 9, 16, 17, 31, 32, 33, 64, 300]))}
    a, b = fit(x, k, True, **dict(kw)), fit(x, k, False, **dict(kw))
    assert a._backend.cells is not None and b._backend.cells is None
    ok = (np.array_equal(a.cluster_centers_.view(np.uint32), b.cluster_centers_.view(np.uint32)) and np.array_equal(a.labels_, b.labels_)
          and a.n_iter_ == b.n_iter_ and a.inertia_ == b.inertia_)
    same = ok
    why = "" if ok else (" [cells vs boxes: centroids %s labels %s (%d differ) n_iter %s inertia %s]" % (
        np.array_equal(a.cluster_centers_.view(np.uint32), b.cluster_centers_.view(np.uint32)), np.array_equal(a.labels_, b.labels_),
        int((a.labels_ != b.labels_).sum()), a.n_iter_ == b.n_iter_, a.inertia_ == b.inertia_))
    o = ""
    if n <= 250000:
        c, l, ni, inert = orc.kmeans_fit(wh, k, 42, **kw)
        ok2 = (np.array_equal(a.cluster_centers_.view(np.uint32), c.view(np.uint32)) and np.array_equal(a.labels_, l) and a.n_iter_ == ni
               and abs(a.inertia_ - inert) < 1e-9)
        o = " oracle " + ("ok" if ok2 else "MISMATCH")
        if not ok2:
            for nm, km in (("cells", a), ("boxes", b)):
                o += " [%s vs oracle: centroids %s labels %s (%d differ) n_iter %s inertia %s]" % (
                    nm, np.array_equal(km.cluster_centers_.view(np.uint32), c.view(np.uint32)), np.array_equal(km.labels_, l),
                    int((km.labels_ != l).sum()), km.n_iter_ == ni, abs(km.inertia_ - inert) < 1e-9)
        ok = ok and ok2
    if not ok:
        bad += 1
    print(f"round {rnd}: n {n} k {k} kind {kind} {kw}: n_iter {a.n_iter_} {'cells==boxes' if same else 'MISMATCH' + why}{o}, {time.time() - t0:.0f} s", flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
