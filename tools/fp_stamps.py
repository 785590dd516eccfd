"""Per-round time stamps of k_nms_resolve_fp (library built with -DDG_FP_STAMPS): python tools/run_with_lib.py vision_amd/libdetgeom_fpstamps.so tools/fp_stamps.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi
from tools import workloads as wl
n = 65536
for thr in (0.5, 0.3):
    b, s = wl.stress_boxes(7, n)
    bt, st = torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda()
    for _ in range(3): k, c = ops.nms_batched(bt, st, None, thr, True)
    torch.cuda.synchronize()
    ws, nbytes = ops.nms_workspace(bt.device, 1, n)
    # layout: keys (Np2*8), 6 planes of Np*4 (each 256-aligned), then mask
    off = n * 8 + 6 * n * 4
    ecap = n * 32
    dbg = ws[off + ecap * 4: off + ecap * 4 + 64 * 8].view(torch.int64).cpu().numpy()
    t0 = dbg[61]
    print("thr", thr, "info", ops.nms_path_info(bt.device, 1, n))
    print(" round0 pass: %.1f us" % ((dbg[0] - t0) / 100.0))
    r = 0
    while r < 30 and dbg[1 + 2 * r] > 0 and (r == 0 or dbg[2 + 2 * r] > 0 or dbg[2 * r] > 0):
        nxt = dbg[1 + 2 * (r + 1)] if dbg[2 + 2 * r] > 0 else dbg[63]
        print("  round %d: live %d, %.1f us" % (r + 1, dbg[2 + 2 * r], (nxt - dbg[1 + 2 * r]) / 100.0))
        if dbg[2 + 2 * r] == 0: break
        r += 1
    print(" tail (count, quirk, keep list): %.1f us; total %.1f us" % ((dbg[62] - dbg[63]) / 100.0, (dbg[62] - t0) / 100.0))
