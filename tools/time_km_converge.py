"""A fit that runs until no label changes (kmean's default mode: chunks of iterations between two host checks of the changed
counts), 10 M boxes, wall time; TK_CELLS=0: the box loop; TK_N: boxes; TK_CHUNK: fixed chunks of that many iterations instead of the growing ones."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
os.environ["DG_KMEANS_CELLS"] = os.environ.get("TK_CELLS", "1")
if os.environ.get("TK_CHUNK"):
    cluster._CHUNK = cluster._CHUNK_MAX = int(os.environ["TK_CHUNK"])
n = int(os.environ.get("TK_N", "10000000"))
x = torch.from_numpy(wl.anchor_wh(0, n)).cuda()
init = cluster.kmean.init_centers(x.cpu().numpy(), 5, 42)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    km = cluster.kmean(5, 42); km.fit(x, init_centers=init); torch.cuda.synchronize()
    print(f"chunks {cluster._CHUNK}..{cluster._CHUNK_MAX}: fit to convergence {1e3*(time.perf_counter()-t0):.3f} ms, n_iter_ {km.n_iter_}, inertia {km.inertia_:.6f}")
