"""Phase timings of the 64k-box NMS (config 5) — sort / mask / reduce, sparse and dense sets, with the pruned path
(DG_NMS_PRUNE=1, the default) and without it (0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi
from tools import workloads as wl
from vision_amd._abi import load, check, ptr, stream_ptr

def phases(b, s, thr, reps=10):
    dev = b.device
    B, N = s.shape
    ws, nbytes = ops.nms_workspace(dev, B, N)
    keep = torch.empty((B, N), dtype=torch.int64, device=dev); cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = load(); st = stream_ptr(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = np.zeros(3)
    for it in range(reps + 2):
        ev[0].record()
        check(lib.dg_nms_phase_sort(ptr(b), ptr(s), None, B, N, ptr(ws), nbytes, st)); ev[1].record()
        check(lib.dg_nms_phase_mask(None, B, N, thr, ptr(ws), nbytes, st)); ev[2].record()
        check(lib.dg_nms_phase_reduce(None, B, N, 1, ptr(keep), ptr(cnt), ptr(ws), nbytes, st)); ev[3].record()
        torch.cuda.synchronize()
        if it >= 2: acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    return acc / reps, cnt.cpu().numpy(), ops.nms_path_info(dev, B, N, 0)

sizes = [int(x) for x in os.environ.get("TN_SIZES", "65536,20000").split(",")]
for n in sizes:
    for dense in (False, True):
        for thr in (0.3, 0.5, 0.7):
            b, s = wl.stress_boxes(7, n, dense=dense)
            for prune in ([int(x) for x in os.environ.get("TN_PRUNE", "0,1").split(",")]):
                os.environ["DG_NMS_PRUNE"] = str(prune)
                t, c, info = phases(torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda(), thr)
                print(f"n={n} dense={dense} thr={thr} prune={prune}: sort {t[0]:.3f} mask {t[1]:.3f} reduce {t[2]:.3f} ms total {t.sum():.3f}  kept {c[0]} path(mode,fp,rounds)={info}", flush=True)
os.environ["DG_NMS_PRUNE"] = "1"
b = np.stack([wl.random_boxes(50 + i, 2000) for i in range(256)]); s = np.stack([wl.tiefree_scores(60 + i, 2000) for i in range(256)])
t, c, _ = phases(torch.from_numpy(b).cuda(), torch.from_numpy(s).cuda(), 0.7)
print(f"batched 256x2000: sort {t[0]:.3f} mask {t[1]:.3f} reduce {t[2]:.3f} ms", flush=True)
b, s = wl.frcnn_proposals(1, 2000)
for thr in (0.7, 0.3):
    t, c, _ = phases(torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda(), thr, reps=50)
    print(f"one image 2000 thr {thr}: sort {t[0]*1e3:.1f} mask {t[1]*1e3:.1f} reduce {t[2]*1e3:.1f} us total {t.sum()*1e3:.1f} us", flush=True)
