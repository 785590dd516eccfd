"""Whole-call timing of dg_nms_batched_f32 on config 5's sets (64k boxes), same process, interleaved A/B of an environment
switch the library reads per call: TNC_VAR (default DG_NMS_FORK) = 0 / 1."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl

var = os.environ.get("TNC_VAR", "DG_NMS_FORK")
vals = os.environ.get("TNC_VALS", "0,1").split(",")
sizes = [int(x) for x in os.environ.get("TNC_SIZES", "65536,20000,8192").split(",")]

def med(fn, reps=60, warm=10):
    ts = []
    for i in range(reps + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= warm: ts.append(a.elapsed_time(b))
    return float(np.median(ts))

for n in sizes:
    for dense in (False, True):
        for thr in (0.5, 0.7):
            b, s = wl.stress_boxes(7, n, dense=dense)
            bt, st = torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda()
            keep = torch.empty((1, n), dtype=torch.int64, device="cuda"); kc = torch.empty((1,), dtype=torch.int32, device="cuda")
            out, ref = {}, None
            for rnd in range(2):
                for v in vals:
                    os.environ[var] = v
                    t = med(lambda: ops.nms_batched(bt, st, None, thr, True, keep=keep, keep_count=kc))
                    out.setdefault(v, []).append(t)
                    k = keep[0, :int(kc.item())].cpu().numpy()
                    if ref is None: ref = k
                    assert np.array_equal(ref, k), "keep lists differ between the settings"
            info = ops.nms_path_info(bt.device, 1, n)
            print(f"n={n} dense={dense} thr={thr}: " + "  ".join(f"{var}={v}: {min(out[v])*1e3:.1f} us" for v in vals) + f"  kept {ref.size} info {info}", flush=True)
os.environ.pop(var, None)
