"""RoI forward throughput by crop width class (exact <=16, generic staged 17..24, direct >24), 16 images x 2000 RoIs."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
images, n = 16, 2000
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
nhwc, nanf = ops.to_nhwc(img)
rng = np.random.default_rng(0)
for name, (lo, hi) in {"w 1..16": (0.02, 0.30), "w 17..24": (0.34, 0.46), "w 25..40": (0.5, 0.78)}.items():
    wh = rng.uniform(lo, hi, (images * n, 2)); xy = rng.uniform(0, 1 - hi, (images * n, 2))
    rois = torch.from_numpy(np.concatenate([xy, xy + wh], 1)).to(dev)
    idx = torch.arange(images, device=dev).repeat_interleave(n)
    win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
    out, arg = ops.roipool_forward(img, win, "max", nhwc=nhwc, nan_flag=nanf)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5): ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    lw = (win[:, 2] - win[:, 0]).float()
    print(f"{name}: widths {int(lw.min())}..{int(lw.max())}  {ms:.3f} ms  {images * n / ms / 1e3:.2f} M RoIs/s", flush=True)
# the unbounded proposal mix of bench.py's config3_wide_rois (w,h ~ U(0.1,0.9))
from tools import workloads as wl
rois = torch.from_numpy(np.concatenate([wl.frcnn_rois_wide(i, n) for i in range(images)])).to(dev)
idx = torch.arange(images, device=dev).repeat_interleave(n)
win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
out, arg = ops.roipool_forward(img, win, "max", nhwc=nhwc, nan_flag=nanf)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(5): ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"mix U(0.1,0.9): {ms:.3f} ms  {images * n / ms / 1e3:.2f} M RoIs/s", flush=True)
