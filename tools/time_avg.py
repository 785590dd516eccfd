"""RoI pooling in avg mode (the reference class as written, fast_rcnn.py:354) vs max mode, 32 images x 2000 RoIs."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
images = 32
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i) for i in range(images)])).to(dev)
idx = torch.arange(images, device=dev).repeat_interleave(2000)
win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
nhwc, nanf = ops.to_nhwc(img)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it
for mode in ("max", "avg"):
    out, arg = ops.roipool_forward(img, win, mode, nhwc=nhwc, nan_flag=nanf)
    g = torch.randn_like(out)
    print(mode, "fwd ms", round(t(lambda: ops.roipool_forward(img, win, mode, out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)), 3),
          "bwd ms", round(t(lambda: ops.roipool_backward(g, arg, win, img.shape, mode)), 3), flush=True)
