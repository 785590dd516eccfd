import os, sys, torch, numpy as np
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
x = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
os.environ["DG_KMEANS_CELLS"] = "1"
ref = None
for persist, g in (("0", "0"), ("1", "0"), ("1", "512"), ("1", "256"), ("1", "128")):
    os.environ["DG_KMEANS_PERSIST"] = persist; os.environ["DG_KMEANS_PERSIST_G"] = g
    km = cluster.kmean(5, 42); km.fit(x, fixed_iters=30)
    if ref is None: ref = km.cluster_centers_.copy()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        ev0.record(); km._run_iters(0, 30); ev1.record(); torch.cuda.synchronize(); ts.append(ev0.elapsed_time(ev1) / 30 * 1e3)
    print("persist", persist, "G cap", g, "us/iter %.2f" % min(ts), "same" if np.array_equal(ref, km.cluster_centers_) else "DIFF", flush=True)
