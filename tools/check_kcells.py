"""The prepared (cell) k-means fit against the box-by-box loop of the same library and the oracle: centroids, labels, history,
inertia bit for bit; then timings."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import cluster, _abi
from tools import workloads as wl
from oracle import oracle as orc
orc.build()

def fit(wh, cells, iters=None, k=5):
    os.environ["DG_KMEANS_CELLS"] = "1" if cells else "0"
    km = cluster.kmean(k, 42)
    km.fit(torch.from_numpy(wh).cuda(), fixed_iters=iters)
    return km

bad = 0
for n, k, iters in ((300001, 5, None), (1000, 5, None), (250000, 9, 12), (2_000_000, 5, 30), (500000, 3, 7), (400000, 16, 5)):
    wh = wl.anchor_wh(n % 97, n)
    if n == 500000:
        wh[::1000] = [3.0, 0.5]; wh[5] = [0.0, 0.2]; wh[7] = [1e-4, 1e-4]      # outside the grid: catch-all cells
    a, b = fit(wh, True, iters, k), fit(wh, False, iters, k)
    same = (np.array_equal(a.cluster_centers_, b.cluster_centers_) and np.array_equal(a.labels_, b.labels_) and a.n_iter_ == b.n_iter_
            and a.inertia_ == b.inertia_)
    o = ""
    if n <= 2_000_000:
        c, l, ni, inert = orc.kmeans_fit(wh, k, 42, fixed_iters=iters) if iters else orc.kmeans_fit(wh, k, 42)
        ok2 = np.array_equal(a.cluster_centers_, c) and np.array_equal(a.labels_, l) and a.n_iter_ == ni and abs(a.inertia_ - inert) < 1e-9
        o = " oracle " + ("ok" if ok2 else "MISMATCH")
        bad += not ok2
    print(n, k, iters, "cells==boxes" if same else "MISMATCH", "n_iter", a.n_iter_, o, flush=True)
    bad += not same
wh = wl.anchor_wh(0, 10_000_000)
x = torch.from_numpy(wh).cuda()
for cells in (False, True):
    os.environ["DG_KMEANS_CELLS"] = "1" if cells else "0"
    km = cluster.kmean(5, 42)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    km.fit(x, fixed_iters=30)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    km._run_iters(0, 30); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        ev0.record(); km._run_iters(0, 30); ev1.record(); torch.cuda.synchronize(); ts.append(ev0.elapsed_time(ev1) / 30 * 1e3)
    print("10M cells=%d: fit(30) wall %.2f ms; steady %.2f us/iter (min of 5 runs of 30)" % (cells, (t1 - t0) * 1e3, min(ts)), flush=True)
print("BAD", bad)
sys.exit(1 if bad else 0)
