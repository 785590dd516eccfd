"""Launch-only time of dg_yolo1_decode_nms_f32 (config 2: batch 64, C = 5): outputs allocated once, the C entry point called
back to back.  python tools/time_yolo_fused.py"""
import os, sys
sys.path.insert(0, '.')
import torch
import bench
from vision_amd import _abi, ops
from tools import workloads as wl
out, tgt = wl.yolo1_batch(0, 64)
dev = torch.device('cuda', 0)
o, t = torch.from_numpy(out).to(dev), torch.from_numpy(tgt).to(dev)
box, label, conf, score, count, tbox, tlabel, keep, kcnt, ccnt = ops.yolo1_decode_batched(o, t, 0.2, 0.01, 0.3, True, True)
print("candidates/image", float(count.float().mean()), "kept", float(kcnt.float().mean()))
lib, p, st = _abi.load(), _abi.ptr, _abi.stream_ptr(dev)
for _ in (0,):
    for pc in (0, 1):
        def raw():
            lib.dg_yolo1_decode_nms_f32(p(o), p(t), 64, 5, 0.2, 0.01, 0.3, pc, 1, p(box), p(label), p(conf), p(score),
                                        p(count), p(keep), p(kcnt), p(ccnt), p(tbox), p(tlabel), st)
        us = min(bench.ev_time(raw, 200, 20) for _ in range(3)) * 1e3
        print("per_class", pc, "us per launch", round(us, 2))
