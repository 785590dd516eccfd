"""k-means step / run time against the shard size: the intercept is the fixed cost of a launch (prologue, block reduce, tickets,
the last workgroup's fold), the slope the per-box cost.  usage: python tools/time_kmeans_sizes.py"""
import sys
import torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi
from tools import workloads as wl

wh_all = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
k = 5
cent = wh_all[:k].clone()


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (256, 65536, 625_000, 1_250_000, 2_500_000, 5_000_000, 10_000_000):
    wh = wh_all[:n]
    lab = torch.zeros(n, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(3 * k + 2, dtype=torch.int64, device="cuda")
    hist = torch.zeros(64, dtype=torch.int64, device="cuda")
    c2 = cent.clone()
    t_step = timed(lambda: ops.kmeans_step(wh, cent, lab, acc))
    t_run = timed(lambda: ops.kmeans_run(wh, c2, lab, acc, hist, 0, 10), 20) / 10
    print(f"n={n:>9}  step (inertia) {t_step:7.2f} us   run (no inertia) {t_run:7.2f} us/iter   run: {12.0 * n / t_run / 1e6 / 8e3 * 100:5.1f} % of 8 TB/s", flush=True)
