"""Kernel-only time of one k-means step on 10 M boxes (config 4)."""
import sys
import torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
for k in (5, 9):
    wh = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda(); c = wh[:k].clone()
    l = torch.zeros(wh.shape[0], dtype=torch.uint8, device="cuda"); a = torch.zeros(3 * k + 2, dtype=torch.int64, device="cuda")
    for _ in range(3): ops.kmeans_step(wh, c, l, a)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): ops.kmeans_step(wh, c, l, a)
    e1.record(); torch.cuda.synchronize(); print(f"k={k} step ms {e0.elapsed_time(e1) / 20:.4f}")
