"""Experiment: does grouping RoIs of an image by crop width help the forward kernel (instruction-cache locality)?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
images = 128
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
nhwc, nanf = ops.to_nhwc(img)
def run(sort):
    rs = []
    for i in range(images):
        r = wl.frcnn_rois(i)
        if sort:
            w = np.ceil(r[:, 2] * 50) - np.floor(r[:, 0] * 50)
            r = r[np.argsort(w, kind="stable")]
        rs.append(r)
    rois = torch.from_numpy(np.concatenate(rs)).to(dev)
    idx = torch.arange(images, device=dev).repeat_interleave(2000)
    win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
    out, arg = ops.roipool_forward(img, win, "max", nhwc=nhwc, nan_flag=nanf)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5):
        ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5
print("unsorted ms", run(False)); print("sorted by width ms", run(True)); print("unsorted ms", run(False))
