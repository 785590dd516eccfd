"""RoI max-pool forward / backward with int32 vs uint16 argmax, 64 images x 2000 RoIs (config-3 shapes)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
images = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
for dt in (torch.int32, torch.int16):
    out, arg = ops.roipool_forward(img, win, "max", nhwc=nhwc, nan_flag=nanf, argmax_dtype=dt)
    g = torch.randn_like(out)
    print(dt, "fwd ms", round(t(lambda: ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)), 3),
          "bwd ms", round(t(lambda: ops.roipool_backward(g, arg, win, img.shape, "max")), 3), flush=True)
