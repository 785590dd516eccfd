"""One workload for rocprofv3 --kernel-trace: the 64k NMS (sparse / dense, PN_THR) through dg_nms_batched_f32, PN_IT times."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
n = int(os.environ.get("PN_N", "65536")); dense = os.environ.get("PN_DENSE", "0") == "1"; thr = float(os.environ.get("PN_THR", "0.5"))
if os.environ.get("PN_FRCNN", "0") == "1": b, s = wl.frcnn_proposals(1, n)
else: b, s = wl.stress_boxes(7, n, dense=dense)
bt, st = torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda()
keep = torch.empty((1, n), dtype=torch.int64, device="cuda"); kc = torch.empty((1,), dtype=torch.int32, device="cuda")
for _ in range(int(os.environ.get("PN_IT", "20"))):
    ops.nms_batched(bt, st, None, thr, True, keep=keep, keep_count=kc)
torch.cuda.synchronize()
print("kept", int(kc.item()), ops.nms_path_info(bt.device, 1, n))
