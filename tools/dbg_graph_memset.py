"""Does a hipMemsetAsync captured in a HIP graph run on every replay?  (ROCm 7.2: the small memset of the k-means loop did not
on the second replay, when the device had been synchronised before the capture.)  The RoI backward zeroes grad_in with
hipMemsetAsync when an image's RoIs are split over several workgroups (S > 1): captured with torch.cuda.graph, replayed."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
B, C, H, W, n = 2, 64, 38, 50, 400
img = torch.randn(B, C, H, W, device="cuda")
if True:
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 0.6, (n, 2)); wh = rng.uniform(0.1, 0.4, (n, 2))
    rois = np.concatenate([xy, xy + wh], 1); idx = np.sort(rng.integers(0, B, n))
win, status = ops.roi_windows(rois, idx, B, H, W, img.device)
out, argmax = ops.roipool_forward(img, win, "max", 7)
g = torch.randn_like(out)
grad = torch.empty_like(img)
ops.roipool_backward(g, argmax, win, img.shape, "max", 7, grad_in=grad)
torch.cuda.synchronize()
ref = grad.clone()
st = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(st):
    ops.roipool_backward(g, argmax, win, img.shape, "max", 7, grad_in=grad)     # warm-up on the capture stream
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        ops.roipool_backward(g, argmax, win, img.shape, "max", 7, grad_in=grad)
for i in range(4):
    gr.replay(); torch.cuda.synchronize()
    d = (grad - ref).abs().max().item()
    print("replay", i, "max |diff| vs eager:", d, "ratio of sums", (grad.sum() / ref.sum()).item(), flush=True)
