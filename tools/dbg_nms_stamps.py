"""Where k_nms_resolve_async spends its cycles (in-kernel stamps of a diagnostic build).  Build the variant first:
   EXTRA_FLAGS=-DDG_NMS_STAMPS bash tools/build_variant.sh st nms.hip vision_amd/csrc/nms.hip
then run this on the GPU box.  Columns per workgroup of image 0, shader cycles: ending phases (sole suppressors; publish + wait
for the stores; arrival + wait for all; packed counts + scans), whole kernel after round 0, the loop of passes, the loop's
publish + count phase, passes."""
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
    print("dense", dense, "thr", thr, "(shader cycles per workgroup; cols: ending phases 0..3, whole, loop, loop publish+count, passes)")
    for g in range(64):
        print("  wg %2d" % g, " ".join("%9.0f" % v for v in a[g]))
    print("  mean ", " ".join("%9.0f" % v for v in a.mean(0)))
