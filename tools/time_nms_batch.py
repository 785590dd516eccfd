import sys, numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
for B, N in ((8, 16384), (32, 8192), (2, 65536)):
    bb = np.stack([wl.stress_boxes(i, N, dense=False)[0] for i in range(B)]); ss = np.stack([wl.stress_boxes(i, N, dense=False)[1] for i in range(B)])
    bt, st = torch.from_numpy(bb).cuda(), torch.from_numpy(ss).cuda()
    keep = torch.empty((B, N), dtype=torch.int64, device="cuda"); kc = torch.empty((B,), dtype=torch.int32, device="cuda")
    for thr in (0.5,):
        for _ in range(5): ops.nms_batched(bt, st, None, thr, True, keep=keep, keep_count=kc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.nms_batched(bt, st, None, thr, True, keep=keep, keep_count=kc)
        e1.record(); torch.cuda.synchronize()
        print(B, N, thr, "ms/call", e0.elapsed_time(e1) / 20, "kept", kc[:3].tolist(), [ops.nms_path_info(bt.device, B, N, i)[:4] for i in (0, B - 1)], flush=True)
