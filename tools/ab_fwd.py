"""A/B: forward/backward (int32 argmax) time at config-3 size for the library given by run_with_lib."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
images = 256
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i) for i in range(images)])).to(dev)
idx = torch.arange(images, device=dev).repeat_interleave(2000)
win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
nhwc, nanf = ops.to_nhwc(img)
out, arg = ops.roipool_forward(img, win, "max", nhwc=nhwc, nan_flag=nanf)
def t(fn, it=int(__import__("os").environ.get("AB_IT", "6"))):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it
print("fwd ms", round(t(lambda: ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)), 3),
      "bwd ms", round(t(lambda: ops.roipool_backward(out, arg, win, img.shape, "max")), 3), flush=True)
