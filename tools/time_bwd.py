"""RoI max-pool backward timing at config-3 size (256 images x 2000 RoIs), int32 argmax."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
images = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i) for i in range(images)])).to(dev)
idx = torch.arange(images, device=dev).repeat_interleave(2000)
win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
out, arg = ops.roipool_forward(img, win, "max")
gin = torch.empty_like(img)
def run(): ops.roipool_backward(out, arg, win, img.shape, "max", grad_in=gin)
for _ in range(3): run()
ref = gin.clone()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
print(f"bwd median {ts[5]:.3f} ms  min {ts[0]:.3f}  ({images * 2000 * 200704 / ts[5] / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
