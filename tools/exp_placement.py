"""Experiment: does the forward kernel's time depend on where `out` and `argmax` sit relative to each other?
One arena, the two outputs carved out of it at different skews; same library, same inputs."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops, _abi as abi
from tools import workloads as wl

images = int(os.environ.get("AB_IMAGES", "64")); it = 10
dev = torch.device("cuda", 0)
img = torch.randn((images, 512, 38, 50), device=dev)
rois = torch.from_numpy(np.concatenate([wl.frcnn_rois(i) for i in range(images)])).to(dev)
idx = torch.arange(images, device=dev).repeat_interleave(2000)
win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
nhwc, nanf = ops.to_nhwc(img)
n = win.shape[0]
nbytes = n * 512 * 49 * 4
lib = abi.load()
arena = torch.empty(2 * nbytes + (256 << 20), dtype=torch.uint8, device=dev)
base = arena.data_ptr()
print("arena base %x  out bytes %d (mod 4096 = %d, mod 2MiB = %d)" % (base, nbytes, nbytes % 4096, nbytes % (2 << 20)), flush=True)
def run(off_out, off_arg):
    po, pa = C.c_void_p(base + off_out), C.c_void_p(base + off_arg)
    def launch():
        rc = lib.dg_roipool_fwd_f32(abi.ptr(nhwc), images, 512, 38, 50, abi.ptr(win), n, 7, abi.POOL_MAX, abi.ptr(nanf), po, pa, abi.stream_ptr(dev))
        assert rc == 0
    launch(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): launch()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for skew in [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 2 << 20, 4 << 20, 16 << 20, 64 << 20, 0]:
    t = run(0, nbytes + skew)
    print(f"skew {skew:>9d}: {t:.3f} ms  ({n / t / 1e3:.2f} M RoIs/s)", flush=True)
# separate torch allocations, a few times
for k in range(4):
    o = torch.empty(nbytes, dtype=torch.uint8, device=dev); a_ = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    base2 = o.data_ptr()
    def launch():
        lib.dg_roipool_fwd_f32(abi.ptr(nhwc), images, 512, 38, 50, abi.ptr(win), n, 7, abi.POOL_MAX, abi.ptr(nanf), abi.ptr(o), abi.ptr(a_), abi.stream_ptr(dev))
    launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): launch()
    e1.record(); torch.cuda.synchronize()
    print(f"separate allocation {k}: out {o.data_ptr():x} arg {a_.data_ptr():x} delta {a_.data_ptr() - o.data_ptr()}: {e0.elapsed_time(e1) / it:.3f} ms", flush=True)
    keep = (o, a_) if k % 2 == 0 else None
