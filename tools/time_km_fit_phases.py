"""Where a whole 30-iteration fit of 10 M boxes spends its wall time: backend set-up (allocations + the one-time grouping), the
iterations, reading the results back -- a device synchronisation after each phase (so the sum is a little above fit()'s own time).
TK_CELLS=0: the box loop."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from vision_amd import cluster, _abi
from tools import workloads as wl
os.environ["DG_KMEANS_CELLS"] = os.environ.get("TK_CELLS", "1")
x = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
init = cluster.kmean.init_centers(x.cpu().numpy(), 5, 42)
sync = torch.cuda.synchronize
for rep in range(5):
    km = cluster.kmean(5, 42)
    be = km._backend
    sync(); t0 = time.perf_counter()
    be.prepare(x, init, 5, 30, cells=None)
    t1h = time.perf_counter(); sync(); t1 = time.perf_counter()
    km._run_iters(0, 30)
    t2h = time.perf_counter(); sync(); t2 = time.perf_counter()
    cent, labels, acc = be.results_lazy()
    sync(); t3 = time.perf_counter()
    print(f"rep {rep}: set-up {1e3*(t1-t0):.3f} ms (host returns after {1e3*(t1h-t0):.3f}), iterations {1e3*(t2-t1):.3f} (host {1e3*(t2h-t1):.3f}), "
          f"results {1e3*(t3-t2):.3f}, sum {1e3*(t3-t0):.3f}")
    sync(); t0 = time.perf_counter()
    km2 = cluster.kmean(5, 42); km2.fit(x, init_centers=init, fixed_iters=30); sync()
    print(f"        whole fit {1e3*(time.perf_counter()-t0):.3f} ms")
