#!/usr/bin/env python3
"""Reference-proper CPU timings (SURVEY.md 8(d) "CPU baseline plan (1)").

Runs in the BUILD CONTAINER only: it needs /root/reference.  The reference's hot-path functions are extracted exactly as
oracle/gen_golden.py extracts them (ast nodes, or text lines for rcnn/fast_rcnn.py, which does not parse) and are timed
on the BASELINE configs' synthetic inputs: warm-up 3 + median of 20 runs, torch intra-op threads stated.  No reference
source is stored; the numbers go into BASELINE.md section 2 (between the markers) and profiles/r02_reference_cpu.json.
The reference's Python never travels to the GPU box: bench.py times the C oracle there instead (cpu_baseline.kind "port").

    python tools/time_reference_cpu.py [--threads 8] [--runs 20]
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as gg  # noqa: E402  (extraction helpers only)
from tools import workloads as wl  # noqa: E402

BEGIN, END = "<!-- reference-cpu-timings:begin -->", "<!-- reference-cpu-timings:end -->"


def median_time(fn, runs, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--runs", type=int, default=20)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    rows = []

    def add(path, inp, fn, units, unit_name, runs=a.runs):
        with contextlib.redirect_stdout(io.StringIO()):
            med, best = median_time(fn, runs)
        rows.append({"path": path, "input": inp, "median_ms": med * 1e3, "best_ms": best * 1e3,
                     "rate": units / med, "unit": unit_name + "/s"})
        print(f"{path:60s} {med * 1e3:10.2f} ms   {units / med:12.4g} {unit_name}/s", flush=True)

    # ---------------- config 1: Fast R-CNN, one image, 2000 proposals
    ns = gg.extract_lines("rcnn/fast_rcnn.py", 351, 377, gg.base_ns())
    Pool = ns["slowroipool"]
    img = torch.from_numpy(wl.frcnn_feature_map(0, 1))
    rois, idx = wl.frcnn_rois(0, 2000), np.zeros(2000, np.int64)
    m = Pool((7, 7))
    add("slowroipool.foward, rcnn/fast_rcnn.py:351-377 (AdaptiveAvgPool2d, as written)", "1x512x38x50, 2000 RoIs",
        lambda: m.foward(img, rois, idx), 2000, "RoIs")
    mm = Pool((7, 7))
    mm.maxpool = nn.AdaptiveMaxPool2d((7, 7), return_indices=False)
    add("same class with AdaptiveMaxPool2d substituted (north_star: max)", "1x512x38x50, 2000 RoIs",
        lambda: mm.foward(img, rois, idx), 2000, "RoIs")
    timg = img.clone().requires_grad_(True)
    gout = torch.randn(2000, 512, 7, 7)

    def fwd_bwd():
        timg.grad = None
        mm.foward(timg, rois, idx).backward(gout)
    add("max variant forward + autograd backward", "1x512x38x50, 2000 RoIs", fwd_bwd, 2000, "RoIs", runs=max(5, a.runs // 4))

    nms = gg._ref_nms()
    b, s = wl.frcnn_proposals(0, 2000)
    tb, ts = torch.from_numpy(b), torch.from_numpy(s)
    for thr in (0.7, 0.3):
        kept = len(nms(tb, ts, thr))
        add(f"nms, yolo/yolo_v1/yolo.py:188-203, thr {thr} ({kept} kept)", "2000 proposals", lambda: nms(tb, ts, thr), 2000, "boxes")
    ns3 = gg.extract_defs("rcnn/utils.py", {"get_iou"}, gg.base_ns())
    giou = ns3["get_iou"]
    rects = (wl.frcnn_rois(0, 2000) * [50, 38, 50, 38]).round().astype(int)
    rects[:, 2:] = np.maximum(rects[:, 2:], rects[:, :2] + 1)
    grt = np.array([10.0, 8.0, 30.0, 25.0])
    add("scalar get_iou loop, rcnn/utils.py:50-67 as called by rcnn/rcnn.py:35-40", "2000 proposals vs 1 ground truth",
        lambda: [giou(list(r), grt) for r in rects], 2000, "boxes")

    # ---------------- config 2: YOLOv1 decode + NMS (+ per-class split of the survivors), batch 64
    ns2 = gg.extract_defs("yolo/yolo_v1/modules.py", {"get_iou", "encode"}, gg.base_ns())
    gg.extract_defs("yolo/yolo_v1/yolo.py", {"decode", "nms"}, ns2)
    out, tgt = wl.yolo1_batch(0, 64)
    to, tt = torch.from_numpy(out), torch.from_numpy(tgt)

    def decode_nms_batch():
        for i in range(64):
            box, lab, conf, score, _, _ = ns2["decode"](to[i], tt[i], 0.2, 0.01)
            if len(box):
                ids = ns2["nms"](box, score, 0.3)
                b_, l_, c_ = box[ids], lab[ids], conf[ids]
                for c in range(5):                         # the split of yolo.py:289-298
                    sel = l_ == c
                    if sel.sum():
                        torch.cat([c_[sel].view(-1, 1), b_[sel].view(-1, 4)], dim=1)
    add("decode + nms + per-class split per image, yolo/yolo_v1/yolo.py:277-298", "batch 64 x (7,7,15)", decode_nms_batch, 64 * 98, "boxes")

    # ---------------- config 4: the reference has no 1-IoU k-means; its own call is sklearn's Euclidean KMeans on 100x2
    try:
        from sklearn.cluster import KMeans
        X = wl.anchor_wh(0, 100000).astype(np.float64)
        add("sklearn KMeans(5, random_state=42) as clustering/cluster_prac.py:38-40 calls it (Euclidean, NOT 1-IoU)",
            "100,000 (w,h) boxes", lambda: KMeans(n_clusters=5, random_state=42, n_init=1).fit(X), 1, "fits", runs=5)
    except Exception as e:  # noqa: BLE001
        print("sklearn timing skipped:", e)

    meta = {"cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__, "runs": a.runs, "warmup": 3}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump({"meta": meta, "rows": rows}, open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json"), "w"), indent=1)
    table = [BEGIN, "",
             f"Measured by `tools/time_reference_cpu.py` in the build container: {meta['cpus']} vCPUs, torch {meta['torch']} CPU path, "
             f"{meta['torch_threads']} intra-op threads, warm-up 3 + median of {a.runs} runs, reference functions executed by "
             "function-level extraction.", "",
             "| path | input | median | rate |", "|---|---|---|---|"]
    for r in rows:
        table.append(f"| {r['path']} | {r['input']} | {r['median_ms']:.2f} ms | {r['rate']:.4g} {r['unit']} |")
    table += ["", END]
    p = os.path.join(ROOT, "BASELINE.md")
    s = open(p).read()
    if BEGIN in s:
        s = s[:s.index(BEGIN)] + "\n".join(table) + s[s.index(END) + len(END):]
    else:
        marker = "## 3. Roofline targets"
        s = s.replace(marker, "### 2b. Reference-proper CPU timings (round 2)\n\n" + "\n".join(table) + "\n\n" + marker)
    open(p, "w").write(s)
    print("BASELINE.md updated")


if __name__ == "__main__":
    main()
