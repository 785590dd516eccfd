"""rocprofv3 --kernel-trace workload: the 10 M-box 30-iteration fit, prepared (cells) path; PK_CELLS=0 for the box loop."""
import os, sys
import torch
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
os.environ["DG_KMEANS_CELLS"] = os.environ.get("PK_CELLS", "1")
x = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
km = cluster.kmean(5, 42)
km.fit(x, fixed_iters=30)
for _ in range(3): km._run_iters(0, 30)
torch.cuda.synchronize()
print(km.n_iter_, km.inertia_)
