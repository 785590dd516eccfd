"""A/B of the 64k NMS phases (mask, reduce) between builds of the library, in ONE process on the SAME workspace, launches
interleaved:  python tools/ab_nms.py vision_amd/libdetgeom.so vision_amd/libexp_x.so [...]
AB_N (65536), AB_FRCNN (1: Fast R-CNN proposals), AB_DENSE (0/1: wh U(0.05,0.3) instead of U(0.005,0.05)), AB_THR (0.5).  Checks that the keep lists agree."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import _abi as abi
from tools import workloads as wl

n = int(os.environ.get("AB_N", "65536")); dense = os.environ.get("AB_DENSE", "0") == "1"; thr = float(os.environ.get("AB_THR", "0.5"))
it = int(os.environ.get("AB_IT", "20")); rounds = int(os.environ.get("AB_ROUNDS", "7"))
dev = torch.device("cuda", 0)
if os.environ.get("AB_FRCNN", "0") == "1":                  # config 1's proposals (AB_N = 2000) instead of the 64k stress boxes
    b, s = wl.frcnn_proposals(1, n)
else:
    b, s = wl.stress_boxes(0, n, dense=dense)
boxes, scores = torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev)
base = abi.load()
wsb = base.dg_nms_workspace_bytes(1, n)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
keep = torch.empty((1, n), dtype=torch.int64, device=dev); kc = torch.zeros((1,), dtype=torch.int32, device=dev)
st = abi.stream_ptr(dev)
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    for name in ("dg_nms_phase_sort", "dg_nms_phase_mask", "dg_nms_phase_reduce"):
        fn = getattr(lib, name); fn.restype, fn.argtypes = abi.SIGNATURES[name]
    libs.append((path, lib))
def sort(lib): assert lib.dg_nms_phase_sort(abi.ptr(boxes), abi.ptr(scores), None, 1, n, abi.ptr(ws), wsb, st) == 0
def mask(lib): assert lib.dg_nms_phase_mask(None, 1, n, thr, abi.ptr(ws), wsb, st) == 0
def red(lib): assert lib.dg_nms_phase_reduce(None, 1, n, 1, abi.ptr(keep), abi.ptr(kc), abi.ptr(ws), wsb, st) == 0
ref = None
for path, lib in libs:
    ws.zero_(); sort(lib); mask(lib); red(lib); torch.cuda.synchronize()
    k = keep[0, :int(kc.item())].cpu().numpy().copy()
    if ref is None: ref = k
    print(path, "kept", len(k), "same" if np.array_equal(k, ref) else "DIFFERENT", flush=True)
def timed(fn, lib):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(lib); a.record()
    for _ in range(it): fn(lib)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it
tm = {p: [] for p, _ in libs}; tr = {p: [] for p, _ in libs}
for r in range(rounds):
    for path, lib in libs:
        sort(lib)
        tm[path].append(timed(mask, lib)); tr[path].append(timed(red, lib))
for p in tm:
    print(f"{p}: mask median {np.median(tm[p]):.4f} ms min {min(tm[p]):.4f} | reduce median {np.median(tr[p]):.4f} ms min {min(tr[p]):.4f}", flush=True)
