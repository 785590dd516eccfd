"""Time of the device AP (ordering + k_ap_gt_table + k_voc_ap) on flat device arrays, against the oracle's Python loop."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from vision_amd import ops
from oracle import oracle
import test_gpu_parity as T
for n_img, n_cls, per in ((3000, 6, 4), (20000, 20, 40)):
    det, tgt = T._ap_case(1, n_img, n_cls, per)
    d = np.array([[c] + r for c in det for r in det[c]], np.float64).reshape(-1, 7)
    t = np.array([[c] + r for c in tgt for r in tgt[c]], np.float64).reshape(-1, 7)
    dev = dict(device="cuda")
    args = (torch.tensor(d[:, 1], dtype=torch.int32, **dev), torch.tensor(d[:, 0], dtype=torch.int64, **dev), torch.tensor(d[:, 2], **dev),
            torch.tensor(d[:, 3:7], dtype=torch.float32, **dev), torch.tensor(t[:, 1], dtype=torch.int32, **dev),
            torch.tensor(t[:, 0], dtype=torch.int32, **dev), torch.tensor(t[:, 3:7], dtype=torch.float32, **dev), n_cls, n_img, 0.5)
    for _ in range(3): ops.voc_ap(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out, _n = ops.voc_ap(*args)
    torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter(); want = oracle.average_precision(det, tgt, 0.5) if len(d) < 200000 else None; cpu = time.perf_counter() - t0
    print(f"images {n_img} classes {n_cls} detections {len(d)}: device {gpu * 1e3:.3f} ms" + (f", oracle loop {cpu:.2f} s" if want else ""))
