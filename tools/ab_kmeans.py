"""A/B of the k-means step (10 M boxes, k = 5; device-side loop of 30 iterations) between builds of the library, ONE process,
SAME buffers, runs interleaved:  python tools/ab_kmeans.py vision_amd/libdetgeom.so vision_amd/libexp_x.so [...]
Checks that labels and sums agree."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import _abi as abi, cluster
from tools import workloads as wl

n = int(os.environ.get("AB_N", "10000000")); k = 5; iters = 30; rounds = int(os.environ.get("AB_ROUNDS", "7"))
dev = torch.device("cuda", 0)
wh_h = wl.anchor_wh(0, n)
init = cluster.kmean.init_centers(wh_h, k, 42)
wh = torch.from_numpy(wh_h).to(dev)
st = abi.stream_ptr(dev)
base = abi.load()
wsb = base.dg_kmeans_workspace_bytes(n, k)
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    fn = lib.dg_kmeans_iou_run; fn.restype, fn.argtypes = abi.SIGNATURES["dg_kmeans_iou_run"]
    libs.append((path, fn))
def run(fn):
    cent = torch.from_numpy(init).to(dev).clone()
    labels = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    acc = torch.zeros(3 * k + 2, dtype=torch.int64, device=dev); hist = torch.zeros(iters, dtype=torch.int64, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    assert fn(abi.ptr(wh), n, abi.ptr(cent), k, abi.ptr(labels), abi.ptr(acc), abi.ptr(hist), 0, iters, abi.ptr(ws), wsb, st) == 0
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3, (labels.cpu().numpy(), acc.cpu().numpy(), cent.cpu().numpy(), hist.cpu().numpy())
ref = None
for path, fn in libs:
    run(fn); _, out = run(fn)
    if ref is None: ref = out
    print(path, "same" if all(np.array_equal(a, b) for a, b in zip(out, ref)) else "DIFFERENT", flush=True)
tm = {p: [] for p, _ in libs}
for r in range(rounds):
    for path, fn in libs: tm[path].append(run(fn)[0])
for p, t in tm.items(): print(f"{p}: median {np.median(t):.2f} us/iter min {min(t):.2f}", flush=True)
