"""rocprofv3 --kernel-trace workload: three whole 30-iteration fits of 10 M boxes from Python (PK_CELLS=0: the box loop), a device
synchronisation between them; read with `tools/rocpd_summary.py timeline <db> 60`.  Prints the wall time of each fit."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
os.environ["DG_KMEANS_CELLS"] = os.environ.get("PK_CELLS", "1")
x = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
init = cluster.kmean.init_centers(x.cpu().numpy(), 5, 42)
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    km = cluster.kmean(5, 42); km.fit(x, init_centers=init, fixed_iters=30); torch.cuda.synchronize()
    print("fit wall ms", (time.perf_counter() - t0) * 1e3, km.n_iter_, km.inertia_)
