import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import vision_amd._abi as abi
abi.LIB_PATH = os.path.abspath("vision_amd/libexp_st.so")
from vision_amd import ops
from tools import workloads as wl
for dense, thr in ((False, 0.5), (True, 0.5)):
    b, s = wl.stress_boxes(7, 65536, dense=dense)
    bt, st = torch.from_numpy(b)[None].cuda(), torch.from_numpy(s)[None].cuda()
    for _ in range(5): keep, cnt = ops.nms_batched(bt, st, None, thr, True)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * (64 * 8))()
    abi.load().dg_debug_nms_stamps(out)
    a = np.array(out[:], dtype=np.float64).reshape(64, 8)
    print("dense", dense, "thr", thr, "(cycles of the 100 MHz/shader counter per workgroup; cols: publish+count, snapshot, idle polls, list pass, tail/total, loop total, idle count, passes)")
    for g in range(64):
        print("  wg %2d" % g, " ".join("%9.0f" % v for v in a[g]))
    print("  mean ", " ".join("%9.0f" % v for v in a.mean(0)))
