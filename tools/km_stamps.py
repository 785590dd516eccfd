"""Diagnostic: where workgroup 0 spends a k-means launch (build: make -C vision_amd/csrc stamps; run through
tools/run_with_lib.py vision_amd/libdetgeom_stamps.so tools/km_stamps.py)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi
from tools import workloads as wl
lib = _abi.load()
wh_all = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).cuda()
for n in (256, 65536, 1_250_000, 10_000_000):
    wh = wh_all[:n]
    k = 5
    cent = wh_all[:k].clone()
    lab = torch.zeros(n, dtype=torch.uint8, device="cuda")
    acc = torch.zeros(17, dtype=torch.int64, device="cuda")
    hist = torch.zeros(64, dtype=torch.int64, device="cuda")
    for _ in range(5):
        ops.kmeans_run(wh, cent, lab, acc, hist, 0, 10)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    assert lib.dg_debug_km_stamps(out) == 0
    t = np.array(list(out), np.int64)
    d = (t[1:6] - t[:5]) * 10
    print(f"n={n:>9}: entry->loads+barrier {d[0]} ns, centroids {d[1]} ns, main loop {d[2]} ns, wave/block reduce {d[3]} ns, atomics issued {d[4]} ns; total {(t[5]-t[0])*10} ns")
    blk = (C.c_ulonglong * 2048)()
    assert lib.dg_debug_km_blocks(blk) == 0
    b = np.array(list(blk), np.int64).reshape(1024, 2)[:int(t[7])]
    t0 = b[:, 0].min()
    st, en = (b[:, 0] - t0) * 10, (b[:, 1] - t0) * 10
    print(f"      {len(b)} workgroups: start min/med/max {st.min()}/{int(np.median(st))}/{st.max()} ns; end min/med/max {en.min()}/{int(np.median(en))}/{en.max()} ns; "
          f"duration min/med/max {(en-st).min()}/{int(np.median(en-st))}/{(en-st).max()} ns")
    q = [0, 10, 20, 30, 33, 40, 50, 60, 66, 70, 80, 90, 100]
    print("      start-time percentiles (ns):", {p: int(np.percentile(st, p)) for p in q})
    print("      started within 1 us:", int((st < 1000).sum()), "of", len(st))
    if len(b) >= 64:
        dur = en - st
        print("      median duration by blockIdx % 8:", [int(np.median(dur[r::8])) for r in range(8)])
        print("      median duration by blockIdx // (nb/8):", [int(np.median(x)) for x in np.array_split(dur, 8)])
        order = np.argsort(dur)
        print("      slowest 12 blocks:", [(int(i), int(dur[i])) for i in order[-12:]])
        print("      fastest 12 blocks:", [(int(i), int(dur[i])) for i in order[:12]])
