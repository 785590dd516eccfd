"""One-off stress: many seeds of the RoI forward (max, argmax bit-exact) and of NMS against the oracle.
Timing-dependent faults (an LDS-DMA / ds_read race was found this way) need many samples to show."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
from oracle import oracle
bad = 0
t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    img = wl.frcnn_feature_map(seed)[:, :128]                  # 128 channels keep the oracle at ~0.5 s
    rois = wl.frcnn_rois(seed); idx = np.zeros(2000, np.int64)
    x = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    win, st = ops.roi_windows(rois, idx, 1, 38, 50, x.device)
    want, warg = oracle.roipool_fwd(np.ascontiguousarray(img), rois, idx, 7, "max")
    for rep in range(3):                                        # same input, different timing
        out, arg = ops.roipool_forward(x, win, "max")
        ok = np.array_equal(out.cpu().numpy(), want) and np.array_equal(arg.cpu().numpy(), warg)
        bad += not ok
    b, s = wl.stress_boxes(seed, 12000, dense=bool(seed & 1))
    for thr in (0.3, 0.6):
        keep, cnt = ops.nms_batched(torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda(), None, thr, True)
        ok = np.array_equal(keep[0, :int(cnt[0])].cpu().numpy(), oracle.nms(b, s, thr, True))
        bad += not ok
    print(f"seed {seed}: mismatches so far {bad}  ({time.time() - t0:.0f} s)", flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
