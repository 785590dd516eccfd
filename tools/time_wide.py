"""RoI pooling on the unbounded proposal mix (bench.py's config3_wide_rois) and on config 1's single image, timed on their own."""
import json, sys
sys.path.insert(0, ".")
import torch
import bench
class D: world, rank, dev = 1, 0, torch.device("cuda:0")
torch.cuda.set_device(0)
print(json.dumps(bench.extra_wide_rois(D()), indent=1))
if hasattr(bench, "extra_single_image"):
    r = bench.extra_single_image(D())
    print(json.dumps({k: v for k, v in r.items() if "roipool" in k or "hbm" in k or "bwd" in k}, indent=1))
