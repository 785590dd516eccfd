"""A/B of the 64k x 64k pairwise IoU between builds of the library, ONE process, SAME buffers, launches interleaved:
    python tools/ab_pairwise.py vision_amd/libdetgeom.so vision_amd/libexp_x.so [...]      (AB_N, default 65536)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import _abi as abi
from tools import workloads as wl

n = int(os.environ.get("AB_N", "65536")); it = int(os.environ.get("AB_IT", "10")); rounds = int(os.environ.get("AB_ROUNDS", "7"))
dev = torch.device("cuda", 0)
b, _ = wl.stress_boxes(0, n)
boxes = torch.from_numpy(b).to(dev)
out = torch.empty((n, n), dtype=torch.float32, device=dev)
st = abi.stream_ptr(dev)
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    fn = lib.dg_iou_pairwise_f32; fn.restype, fn.argtypes = abi.SIGNATURES["dg_iou_pairwise_f32"]
    libs.append((path, fn))
def launch(fn): assert fn(abi.ptr(boxes), n, abi.ptr(boxes), n, 1e-7, abi.ptr(out), st) == 0
ref = None
for path, fn in libs:
    out.zero_(); launch(fn); torch.cuda.synchronize()
    sig = out.view(torch.int32).sum(dtype=torch.int64).item()
    if ref is None: ref = sig
    print(path, "checksum", "same" if sig == ref else "DIFFERENT", flush=True)
tm = {p: [] for p, _ in libs}
for r in range(rounds):
    for path, fn in libs:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(fn); a.record()
        for _ in range(it): launch(fn)
        e.record(); torch.cuda.synchronize()
        tm[path].append(a.elapsed_time(e) / it)
for p, t in tm.items():
    print(f"{p}: median {np.median(t):.4f} ms min {min(t):.4f}  ({4.0 * n * n / np.median(t) / 1e9:.2f} TB/s)", flush=True)
