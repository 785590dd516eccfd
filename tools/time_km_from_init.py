"""30 Lloyd iterations of 10 M boxes timed with HIP events two ways: FROM THE INITIAL CENTROIDS (state reset and the boxes regrouped
before every timed run, untimed: what a fit runs) and as a replay from the converged state (what `_run_iters` after a fit runs).
The prepared iterations depend on the state -- pieces move between "decided as a whole" and "walked" while the centroids move --
the box loop does not.  TK_CELLS=0: the box loop."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
os.environ["DG_KMEANS_CELLS"] = os.environ.get("TK_CELLS", "1")
iters = int(os.environ.get("TK_ITERS", "30"))
x = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
init = cluster.kmean.init_centers(x.cpu().numpy(), 5, 42)
km = cluster.kmean(5, 42); km.fit(x, init_centers=init, fixed_iters=iters)
be = km._backend
init_dev = torch.from_numpy(init).cuda().reshape(be.cent.shape)
def ev(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    return a.elapsed_time(b)
def reset():
    be.cent.copy_(init_dev); be.labels.fill_(255)
    if be.cells is not None: be.reprepare(fresh=True)
fi = []
for _ in range(5):
    reset(); fi.append(ev(lambda: km._run_iters(0, iters)))
cent_fit = be.cent.cpu().numpy().copy()
rp = [ev(lambda: km._run_iters(0, iters)) for _ in range(5)]
print(f"path {'prepared' if be.cells is not None else 'box loop'}: from the initial centroids {min(fi)/iters*1e3:.2f} us per iteration (runs {[round(v,3) for v in fi]} ms), "
      f"replay from the state after {iters} {min(rp)/iters*1e3:.2f} us ({[round(v,3) for v in rp]})")
print("centroids after the timed fit equal the first fit's:", np.array_equal(cent_fit, km.cluster_centers_))
