"""Profiling driver: runs one hot-path phase a few times (used under rocprofv3).
usage: python3 tools/prof_roi.py {fwd|bwd|fwdwide|kmeans|kmeans_prepared|kmeans_shard|nms64k|nms64k_dense|nms256|nms2000|pairwise} [images] [a32]   (RoI pooling: uint16 argmax unless a32)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vision_amd import ops
from tools import workloads as wl  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
images = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
if what in ("fwd", "bwd", "fwdwide"):
    img = torch.randn((images, 512, 38, 50), device=dev)
    gen = wl.frcnn_rois_wide if what == "fwdwide" else wl.frcnn_rois
    rois = torch.from_numpy(np.concatenate([gen(i) for i in range(images)])).to(dev)
    idx = torch.arange(images, device=dev).repeat_interleave(2000)
    win, st = ops.roi_windows(rois, idx, images, 38, 50, dev)
    nhwc, nanf = ops.to_nhwc(img)
    adt = torch.int32 if (len(sys.argv) > 3 and sys.argv[3] == "a32") else torch.int16
    out, arg = ops.roipool_forward(img, win, "max", nhwc=nhwc, nan_flag=nanf, argmax_dtype=adt)
    for _ in range(3):
        if what in ("fwd", "fwdwide"):
            ops.roipool_forward(img, win, "max", out=out, argmax=arg, nhwc=nhwc, nan_flag=nanf)
        else:
            ops.roipool_backward(out, arg, win, img.shape, "max")
elif what in ("kmeans", "kmeans_shard"):
    wh = torch.from_numpy(wl.anchor_wh(0, 10_000_000 if what == "kmeans" else 1_250_000)).to(dev)
    cent = wh[:5].clone()
    labels = torch.zeros(wh.shape[0], dtype=torch.uint8, device=dev)
    acc = torch.zeros(17, dtype=torch.int64, device=dev)
    for _ in range(4):
        ops.kmeans_step(wh, cent, labels, acc)
elif what == "kmeans_prepared":
    from vision_amd import cluster
    wh = torch.from_numpy(wl.anchor_wh(0, 10_000_000)).to(dev)
    km = cluster.kmean(5, 42)
    km.fit(wh, init_centers=cluster.kmean.init_centers(wl.anchor_wh(0, 10_000_000), 5, 42), fixed_iters=30)
    km._run_iters(0, 30)
elif what in ("nms64k", "nms64k_dense"):
    b, s = wl.stress_boxes(0, 65536, dense=what == "nms64k_dense")
    for _ in range(3):
        ops.nms_batched(torch.from_numpy(b).to(dev).reshape(1, -1, 4), torch.from_numpy(s).to(dev).reshape(1, -1), None, 0.5, True)
elif what == "nms256":
    props = [wl.frcnn_proposals(i, 2000) for i in range(256)]
    bb = torch.from_numpy(np.stack([b for b, _ in props])).to(dev)
    ss = torch.from_numpy(np.stack([s for _, s in props])).to(dev)
    for _ in range(3):
        ops.nms_batched(bb, ss, None, 0.7, True)
elif what == "nms2000":
    b, s = wl.frcnn_proposals(0, 2000)
    bt, st = torch.from_numpy(b).to(dev).reshape(1, -1, 4), torch.from_numpy(s).to(dev).reshape(1, -1)
    for _ in range(5):
        ops.nms_batched(bt, st, None, 0.7, True)
elif what == "pairwise":
    b, s = wl.stress_boxes(0, 65536, dense=False)
    bt = torch.from_numpy(b).to(dev)
    for _ in range(3):
        ops.iou_pairwise(bt, bt)
torch.cuda.synchronize()
print("done", what)
