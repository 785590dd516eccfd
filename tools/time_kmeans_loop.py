"""Host-side cost of the sharded Lloyd loop: the world>1 code path (step, all-reduce, update per iteration from Python)
on one GPU with the all-reduce stubbed out, against the device-side loop of the single-rank path."""
import sys, time
import torch
sys.path.insert(0, ".")
from vision_amd import cluster
from tools import workloads as wl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000         # default: one rank's shard at 8 GPUs
wh = torch.from_numpy(wl.anchor_wh(0, n)).cuda()
init = cluster.kmean.init_centers(wh.cpu().numpy(), 5, 42)
km = cluster.kmean(5, 42); km.fit(wh, init_centers=init, fixed_iters=30)
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / 30 * 1e6
print("device-side loop  us/iter", round(t(lambda: km._run_iters(0, 30)), 1))
km._world = lambda: 2
km._allreduce = lambda acc: None
print("python loop (no collective) us/iter", round(t(lambda: km._run_iters(0, 30)), 1))
