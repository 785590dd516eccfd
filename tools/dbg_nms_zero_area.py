"""Check: several zero-area boxes that do not overlap (IoU 0 / 0 = NaN, which does not pass `iou < thr`: the later one is dropped)
on the large-image NMS paths against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
from oracle import oracle
bad = 0
for n in (5000, 20000, 65536, 2000, 300):
    for nz in (2, 40):
        b, s = wl.stress_boxes(3, n, dense=False)
        r = np.random.default_rng(n + nz)
        idx = r.choice(n, nz, replace=False)
        kinds = r.integers(0, 3, nz)
        for i, kd in zip(idx, kinds):
            if kd == 0: b[i, 2] = b[i, 0]; b[i, 3] = b[i, 1]          # a point
            elif kd == 1: b[i, 2] = b[i, 0]                            # a vertical line
            else: b[i, 3] = b[i, 1]                                    # a horizontal line
        for thr in (0.5, 0.7):
            for prune in ("1", "2", "0"):
                os.environ["DG_NMS_PRUNE"] = prune
                keep, kc = ops.nms_batched(torch.from_numpy(b).cuda()[None], torch.from_numpy(s).cuda()[None], None, thr, True)
                got = keep[0, :int(kc[0])].cpu().numpy()
                want = oracle.nms(b, s, thr, True)
                ok = np.array_equal(got, want)
                bad += not ok
                info = ops.nms_path_info(torch.device("cuda", 0), 1, n)
                print(f"n {n} zero-area {nz} thr {thr} DG_NMS_PRUNE={prune}: kept {len(want)} {'ok' if ok else 'MISMATCH (got %d)' % len(got)} info {info}")
print("BAD", bad)
