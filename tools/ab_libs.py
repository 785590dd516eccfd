"""A/B of the 7x7 max forward (AB_OP=fwd, default; fwd16 = uint16 argmax) or backward (AB_OP=bwd / bwd16) between builds of the library, in ONE process on the SAME buffers, launches interleaved
(run-to-run and buffer-placement noise on a box is several per cent, more than most kernel changes):
    python tools/ab_libs.py vision_amd/libdetgeom.so vision_amd/libexp_x.so [...]
Prints per library the median / min of `rounds` timings of `it` launches each, and checks that the outputs agree."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi as abi
from tools import workloads as wl

images = int(os.environ.get("AB_IMAGES", "256")); it = int(os.environ.get("AB_IT", "10")); rounds = int(os.environ.get("AB_ROUNDS", "7"))
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i) for i in range(images)])).to(dev)
idx = torch.arange(images, device=dev).repeat_interleave(2000)
win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
nhwc, nanf = ops.to_nhwc(img)
n = win.shape[0]
out = torch.empty((n, 512, 7, 7), device=dev); arg = torch.empty((n, 512, 7, 7), dtype=torch.int32, device=dev)
op = os.environ.get("AB_OP", "fwd")
sym = {"fwd": "dg_roipool_fwd_f32", "bwd": "dg_roipool_bwd_f32", "fwd16": "dg_roipool_fwd_a16_f32", "bwd16": "dg_roipool_bwd_a16_f32"}[op]
if op.endswith("16"): arg = torch.empty((n, 512, 7, 7), dtype=torch.int16, device=dev)
libs = []
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    fn = getattr(lib, sym)
    fn.restype, fn.argtypes = abi.SIGNATURES[sym]
    libs.append((path, fn))
if op.startswith("bwd"):
    ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)
    del nhwc
    gin = torch.empty((images, 512, 38, 50), device=dev)
    wsb = abi.load().dg_roipool_bwd_workspace_bytes(images, n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
def launch(fn):
    if op.startswith("fwd"):
        rc = fn(abi.ptr(nhwc), images, 512, 38, 50, abi.ptr(win), n, 7, abi.POOL_MAX, abi.ptr(nanf), abi.ptr(out), abi.ptr(arg), abi.stream_ptr(dev))
    else:
        rc = fn(abi.ptr(out), abi.ptr(arg), abi.ptr(win), n, images, 512, 38, 50, 7, abi.POOL_MAX, abi.ptr(gin), abi.ptr(ws), wsb, abi.stream_ptr(dev))
    assert rc == 0, rc
ref = None
for path, fn in libs:
    if op.startswith("fwd"):
        out.zero_(); arg.zero_(); launch(fn); torch.cuda.synchronize()
        sig = (out.view(torch.int32).sum(dtype=torch.int64).item(), arg.view(torch.uint16 if op == "fwd16" else torch.int32).sum(dtype=torch.int64).item())
    else:
        gin.zero_(); launch(fn); torch.cuda.synchronize()
        sig = (gin.double().sum().item(), gin.double().abs().sum().item())
    if ref is None: ref = sig
    print(path, "checksum", "same" if sig == ref else f"DIFFERENT {sig} vs {ref}", flush=True)
times = {p: [] for p, _ in libs}
for r in range(rounds):
    for path, fn in libs:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        launch(fn); a.record()
        for _ in range(it): launch(fn)
        b.record(); torch.cuda.synchronize()
        times[path].append(a.elapsed_time(b) / it)
for p, t in times.items():
    print(f"{p}: median {np.median(t):.3f} ms  min {min(t):.3f}  max {max(t):.3f}", flush=True)
