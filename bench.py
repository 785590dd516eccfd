#!/usr/bin/env python3
"""bench.py — box-geometry hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself (a
`python -m torch.distributed.run` child, decided before any GPU call) and exits with the child's code; under
torchrun (WORLD_SIZE set) it is one of the ranks.  A line whose n_gpus differs from --gpus is never printed.

Primary line (ONE JSON line on rank 0): BASELINE.json configs[2] — Fast R-CNN batch of 256 images x 2000 RoIs on
512x38x50 maps: RoI max-pool forward (+argmax), backward, and batched NMS (thr 0.7).  The path shards by image with
no data-path collective; every rank processes a full 256-image batch ("weak": per-GPU work fixed; `--scaling strong`
splits ONE 256-image batch over the ranks instead, and N > 1 runs report that as extra.config3_strong_scaling).
A "step" is one pass of that hot path over the batch; value = RoIs processed by all ranks / time.
`roofline` prices the dominant kernel (k_roipool_fwd) against 8 TB/s HBM; `cpu_baseline` times the
CPU oracle (a port of the reference's algorithm) on a bounded sample on rank 0.
`extra` carries the other BASELINE configs: single image (1), YOLOv1 decode+NMS batch 64 (2),
anchor k-means 10M boxes sharded with an int64 all-reduce (4), 64k pairwise IoU + NMS (5).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ROI_BYTES = 200704                # out + argmax bytes per RoI (512 ch x 49 x (4+4)), int32 argmax as BASELINE configs[2] sizes it
ROI_BYTES_A16 = 150528            # the uint16-argmax layout slowroipool's autograd path uses (512 ch x 49 x (4+2)): side measurement
MAP_BYTES = 3891200               # one 512x38x50 fp32 map


VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 at 2.4 GHz = 7.86e13 lane-ops/s (157.3 TFLOP/s fp32 = 2 flops each)


def ev_time(fn, iters, warm=1):
    """mean of `iters` back-to-back runs timed with one HIP event pair on the current stream (ms)."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


def ev_median(fn, iters=50, warm=20):
    """20 warm-up + median of >= 50 launches, each bracketed by its own HIP event pair on the launch stream (ms)."""
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a torch.distributed.run child (this process has
    made no GPU call) and hand back its exit code; the ranks' stdout/stderr are this process's."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


class Dist:
    def __init__(self, n, stub=False):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != n:
            raise SystemExit(f"bench.py: --gpus {n} but WORLD_SIZE={self.world}: refusing to report a run whose n_gpus differs from --gpus")
        # DG_BENCH_REHEARSE=1: every rank on cuda:0 with gloo -- a functional rehearsal of the N>1 path on a one-GPU box
        # stub (DG_BENCH_STUB=1): no GPU at all, gloo -- exercises the launch / timing / reporting scaffolding on the CPU
        rehearse = self.rehearse = stub or os.environ.get("DG_BENCH_REHEARSE") == "1"
        self.stub = stub
        self.backend = "none"
        if stub:
            self.dev = torch.device("cpu")
        else:
            dev_index = 0 if rehearse else self.local
            torch.cuda.set_device(dev_index)
            self.dev = torch.device("cuda", dev_index)
        if self.world > 1:
            self.backend = "gloo" if rehearse else "nccl"
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
            seen = dist.get_world_size()
            if seen != n:
                raise SystemExit(f"bench.py: process group has {seen} ranks, --gpus {n}")
            print(f"[bench rank {self.rank}] process group up: backend={self.backend} world_size={seen} device={self.dev}", file=sys.stderr, flush=True)

    def barrier(self):
        if self.world > 1:
            dist.barrier()
        if not self.stub:
            torch.cuda.synchronize()

    def max(self, x):
        if self.world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.rehearse else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, steps, warmup):
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        return self.max(time.perf_counter() - t0)


# ------------------------------------------------------------------ config 3 (primary) / config 1
class FrcnnBatch:
    def __init__(self, d, images_total, n_rois=2000, C=512, H=38, W=50, nms_thr=0.7, roi_gen=None):
        from vision_amd import ops, shard
        from tools import workloads as wl
        self.ops, self.d = ops, d
        first, last = shard.shard_range(images_total, d.rank, d.world)     # contiguous image range of this rank
        per = last - first
        self.B, self.n_rois, self.shape = per, n_rois, (per, C, H, W)
        dev = d.dev
        g = torch.Generator(device=dev).manual_seed(1234 + d.rank)
        self.img = torch.randn((per, C, H, W), generator=g, device=dev, dtype=torch.float32)
        roi_gen = roi_gen or wl.frcnn_rois
        rois = np.concatenate([roi_gen(first + i, n_rois) for i in range(per)])
        self.rois = torch.from_numpy(rois).to(dev)
        self.idx = torch.arange(per, device=dev, dtype=torch.int64).repeat_interleave(n_rois)
        self.boxes = self.rois.to(torch.float32).reshape(per, n_rois, 4).contiguous()
        sc = np.stack([wl.tiefree_scores(first + i + 7, n_rois) for i in range(per)])
        self.scores = torch.from_numpy(sc).to(dev)
        n = per * n_rois
        self.n = n
        self.out = torch.empty((n, C, 7, 7), dtype=torch.float32, device=dev)
        self.arg = torch.empty((n, C, 7, 7), dtype=torch.int32, device=dev)
        try:
            self.gout = torch.randn((n, C, 7, 7), generator=g, device=dev, dtype=torch.float32)
            self.gout_alias = False
        except torch.OutOfMemoryError:
            self.gout, self.gout_alias = self.out, True      # pooled values reused as the incoming gradient
        self.gin = torch.empty((per, C, H, W), dtype=torch.float32, device=dev)
        self.keep = torch.empty((per, n_rois), dtype=torch.int64, device=dev)
        self.kcnt = torch.empty((per,), dtype=torch.int32, device=dev)
        self.nms_thr = nms_thr
        self.status = None

    # phases -------------------------------------------------------------
    def windows(self):
        self.win, self.status = self.ops.roi_windows(self.rois, self.idx, self.B, self.shape[2], self.shape[3], self.d.dev)

    def stage(self):
        self.nhwc, self.nanf = self.ops.to_nhwc(self.img)

    def fwd(self):
        self.ops.roipool_forward(self.img, self.win, "max", out=self.out, argmax=self.arg, nhwc=self.nhwc, nan_flag=self.nanf)

    def bwd(self):
        self.ops.roipool_backward(self.gout, self.arg, self.win, self.shape, "max", grad_in=self.gin)

    def nms(self):
        self.ops.nms_batched(self.boxes, self.scores, None, self.nms_thr, True, keep=self.keep, keep_count=self.kcnt)

    def step(self):
        self.windows(); self.stage(); self.fwd(); self.bwd(); self.nms()

    def check(self):
        assert int(self.status.item()) == 0, "empty RoI in synthetic workload"


def cpu_baseline_frcnn(images=12, n_rois=2000):
    """CPU oracle (scalar C port of the reference algorithm), one core, same step on a bounded sample."""
    from oracle import oracle as orc
    from tools import workloads as wl
    orc.build()
    img = wl.frcnn_feature_map(0, images)
    rois = np.concatenate([wl.frcnn_rois(i, n_rois) for i in range(images)])
    idx = np.repeat(np.arange(images), n_rois).astype(np.int64)
    gout = np.random.default_rng(0).standard_normal((images * n_rois, 512, 7, 7), dtype=np.float32)
    t0 = time.perf_counter()
    out, arg = orc.roipool_fwd(img, rois, idx, 7, "max")
    orc.roipool_bwd(gout, arg, img.shape, rois, idx, 7, "max")
    for i in range(images):
        b, s = wl.frcnn_proposals(i, n_rois)
        orc.nms(b, s, 0.7)
    dt = time.perf_counter() - t0
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    ref = None
    try:    # the reference's own Python, timed in the build container by tools/time_reference_cpu.py (numbers only travel)
        rj = json.load(open(os.path.join(ROOT, "profiles", "r02_reference_cpu.json")))
        rows = {r["path"]: r["median_ms"] for r in rj["rows"]}
        step_ms = ([v for k, v in rows.items() if k.startswith("max variant forward + autograd backward")][0] +
                   [v for k, v in rows.items() if k.startswith("nms") and "thr 0.7" in k][0])
        ref = {"value": 2000 / step_ms * 1e3, "unit": "RoIs/s", "cores": rj["meta"]["torch_threads"],
               "what": "reference functions (slowroipool with AdaptiveMaxPool2d fwd + autograd bwd, nms thr 0.7) on one config-1 image, "
                       "build container, median of %d" % rj["meta"]["runs"]}
    except Exception:  # noqa: BLE001
        pass
    return {"value": images * n_rois / dt, "unit": "RoIs/s", "cores": 1, "kind": "port", "reference_python_in_build_container": ref,
            "host": {"os_cpu_count": os.cpu_count(), "usable_cpus": usable, "torch_num_threads": torch.get_num_threads()},
            "sample": f"{images} images x {n_rois} RoIs of the same step (C oracle: RoI max-pool fwd+bwd + NMS 0.7), {dt:.1f} s, "
                      "single thread; the reference's own Python is timed in the build container only (BASELINE.md section 2, "
                      "tools/time_reference_cpu.py): it cannot travel to the GPU box"}


def counters():
    """per-launch PMC figures summarised from profiles/ (tools/pmc_summary.py): {kernel: {size: {...}}}"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:  # noqa: BLE001
        return {}


def valu_frac(kernel, size, ms):
    """VALU lane-operations per second of a launch (SQ_INSTS_VALU x 64 lanes, from the committed PMC pass) over the chip's
    VALU issue peak: the yardstick for the kernels that are VALU-bound, where an HBM fraction says nothing."""
    insts = counters().get("valu_insts_per_launch", {}).get(kernel, {}).get(str(size))
    if not insts:
        return None
    return insts * 64.0 / (ms * 1e-3) / VALU_PEAK_LANE_OPS


# ------------------------------------------------------------------ extras
def extra_single_image(d):
    w = FrcnnBatch(d if d.world == 1 else _solo(d), 1)
    w.windows(); w.stage()
    r = {}
    t = ev_median(w.fwd, 100, 20); r["roipool_fwd_ms"] = t
    r["roipool_fwd_rois_per_s"] = 2000 / t * 1e3
    r["roipool_fwd_hbm_frac"] = (2000 * ROI_BYTES + MAP_BYTES) / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS
    t = ev_median(w.bwd, 100, 20); r["roipool_bwd_ms"] = t
    r["roipool_bwd_hbm_frac"] = (2000 * ROI_BYTES + MAP_BYTES) / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS
    for thr in (0.7, 0.3):
        w.nms_thr = thr
        t = ev_median(w.nms, 100, 20)
        r[f"nms2000_thr{thr}_ms"] = t
        r[f"nms2000_thr{thr}_boxes_per_s"] = 2000 / t * 1e3
    r["note"] = ("config 1; NMS at N=2000 is launch-latency-bound (3 kernels: LDS bitonic sort, mask, fixed-point keep list from the "
                 "mask's non-zero words); median of 100 launches after 20 warm-up")
    return r


class _solo:
    """view of a Dist as a single-rank job (for the replicas-only extras)"""
    def __init__(self, d):
        self.world, self.rank, self.dev = 1, 0, d.dev


def extra_wide_rois(d, images=32):
    """RoI pooling on a proposal mix with no size bound (w,h ~ U(0.1,0.9) of the map: crops up to 46 x 35 cells): the
    primary workload's RoIs (w,h ~ U(0.02,0.30)) are at most 16 cells wide."""
    from tools import workloads as wl
    w = FrcnnBatch(_solo(d), images, roi_gen=wl.frcnn_rois_wide)
    w.windows(); w.stage()
    tf, tb = ev_median(w.fwd, 20, 5), ev_median(w.bwd, 20, 5)
    w.check()
    by = w.n * ROI_BYTES + w.B * MAP_BYTES
    return {"images": images, "rois": w.n, "roipool_fwd_ms": tf, "roipool_bwd_ms": tb,
            "fwd_rois_per_s": w.n / tf * 1e3, "bwd_rois_per_s": w.n / tb * 1e3,
            "fwd_hbm_frac": by / (tf * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bwd_hbm_frac": by / (tb * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "note": "output bytes per RoI are the same as in the primary workload; a wide crop reads up to 1610 cells x 512 ch "
                    "(3.3 MB, from L2) per 200 KB written, so this mix is bound by the argmax scan, not by HBM"}


def extra_yolo(d):
    from vision_amd import yolo_v1, ops
    from tools import workloads as wl
    out, tgt = wl.yolo1_batch(0, 64)
    o, t = torch.from_numpy(out).to(d.dev), torch.from_numpy(tgt).to(d.dev)
    ms = ev_median(lambda: yolo_v1.decode_nms_batched(o, t, 0.2, 0.01, 0.3), 100, 20)
    ms_unfused = ev_median(lambda: yolo_v1.decode_nms_batched(o, t, 0.2, 0.01, 0.3, fused=False), 100, 20)
    big, bigt = o.repeat(64, 1, 1, 1), t.repeat(64, 1, 1, 1)             # 4096 images: past the launch-latency regime
    ms_big = ev_median(lambda: yolo_v1.decode_nms_batched(big, bigt, 0.2, 0.01, 0.3), 50, 10)
    ms_loss = ev_median(lambda: ops.yolo1_loss(o, t, True, True), 100, 20)
    # the launch alone (outputs allocated once, the C entry point called directly): what a caller replaying a HIP graph sees
    from vision_amd import _abi
    box, label, conf, score, count, tbox, tlabel, keep, kcnt, ccnt = ops.yolo1_decode_batched(o, t, 0.2, 0.01, 0.3, True, True)
    lib, p, st = _abi.load(), _abi.ptr, _abi.stream_ptr(d.dev)

    def raw(per_class):
        _abi.check(lib.dg_yolo1_decode_nms_f32(p(o), p(t), 64, 5, 0.2, 0.01, 0.3, per_class, 1, p(box), p(label), p(conf), p(score),
                                               p(count), p(keep), p(kcnt), p(ccnt), p(tbox), p(tlabel), st))
    us_raw = min(ev_time(lambda: raw(0), 200, 20) for _ in range(3)) * 1e3
    us_raw_pc = min(ev_time(lambda: raw(1), 200, 20) for _ in range(3)) * 1e3
    gr = yolo_v1.DecodeNmsGraph(64, 5, 0.2, 0.01, 0.3, device=d.dev)
    gr.output.copy_(o); gr.target.copy_(t)
    us_graph = min(ev_time(gr, 200, 20) for _ in range(3)) * 1e3
    r = {"decode_nms_b64_ms": ms, "launch_only_us": us_raw, "graph_replay_us": us_graph,
         "graph_replay_boxes_per_s": 64 * 98 / us_graph * 1e6, "boxes_per_s": 64 * 98 / ms * 1e3, "launches": 1,
         "decode_nms_b64_unfused_ms": ms_unfused, "unfused_launches": 4,
         "decode_nms_b4096_ms": ms_big, "b4096_boxes_per_s": 4096 * 98 / ms_big * 1e3,
         "loss_fwd_bwd_b64_ms": ms_loss, "loss_cells_per_s": 64 * 49 / ms_loss * 1e3,
         "note": "config 2: ~0.4 MB moved: launch-latency-bound, so decode and NMS are ONE launch (dg_yolo1_decode_nms_f32: an image's "
                 "<= 98 candidates stay in LDS; unfused = the decode launch + the general batched NMS); NMS here is class-agnostic per image as in the reference "
                 "(yolo.py:282-298 splits by class AFTER the NMS); loss = fused YOLOv1 loss forward + gradient (2 kernels; the "
                 "reference issues ~60 torch kernels + autograd)"}
    if hasattr(yolo_v1, "decode_nms_batched") and "per_class" in yolo_v1.decode_nms_batched.__code__.co_varnames:
        ms_pc = ev_median(lambda: yolo_v1.decode_nms_batched(o, t, 0.2, 0.01, 0.3, per_class=True), 100, 20)
        ms_pc_u = ev_median(lambda: yolo_v1.decode_nms_batched(o, t, 0.2, 0.01, 0.3, per_class=True, fused=False), 100, 20)
        r["config2_per_class"] = {"decode_nms_b64_ms": ms_pc, "boxes_per_s": 64 * 98 / ms_pc * 1e3, "launches": 1, "launch_only_us": us_raw_pc,
                                  "decode_nms_b64_unfused_ms": ms_pc_u, "unfused_launches": 6,
                                  "note": "BASELINE configs[1] as worded: per-class NMS (boxes of different classes never suppress each "
                                          "other).  No reference counterpart (the reference's NMS is class-agnostic): parity unpinned, "
                                          "checked against the oracle's NMS applied per class"}
    return r


def extra_stress(d, n=65536):
    from vision_amd import ops, _abi
    from tools import workloads as wl
    r = {}
    b, s = wl.stress_boxes(0, n, dense=False)
    bt, st = torch.from_numpy(b).to(d.dev), torch.from_numpy(s).to(d.dev)
    out = torch.empty((n, n), dtype=torch.float32, device=d.dev)
    ms = ev_median(lambda: ops.iou_pairwise(bt, bt, out=out), 50, 20)
    r["pairwise_64k_ms"] = ms
    r["pairwise_pairs_per_s"] = n * n / ms * 1e3
    r["pairwise_hbm_frac"] = (4.0 * n * n + 32 * n) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    del out
    lib = _abi.load()
    for dense in (False, True):
        b, s = wl.stress_boxes(0, n, dense=dense)
        bt, st = torch.from_numpy(b).to(d.dev).reshape(1, n, 4), torch.from_numpy(s).to(d.dev).reshape(1, n)
        keep = torch.empty((1, n), dtype=torch.int64, device=d.dev)
        kc = torch.empty((1,), dtype=torch.int32, device=d.dev)
        ws, nb = ops.nms_workspace(d.dev, 1, n)
        sp = _abi.stream_ptr(d.dev)
        for thr in (0.5, 0.7):
            tag = f"nms64k_{'dense' if dense else 'sparse'}_thr{thr}"
            t_sort = ev_median(lambda: lib.dg_nms_phase_sort(_abi.ptr(bt), _abi.ptr(st), None, 1, n, _abi.ptr(ws), nb, sp), 50, 20)
            t_mask = ev_median(lambda: lib.dg_nms_phase_mask(None, 1, n, thr, _abi.ptr(ws), nb, sp), 50, 20)
            t_red = ev_median(lambda: lib.dg_nms_phase_reduce(None, 1, n, 1, _abi.ptr(keep), _abi.ptr(kc), _abi.ptr(ws), nb, sp), 50, 20)
            tot = t_sort + t_mask + t_red
            t_all = ev_median(lambda: ops.nms_batched(bt, st, None, thr, True, keep=keep, keep_count=kc), 50, 20)
            mode, fpdone, rounds, npairs, cpb, est = ops.nms_path_info(d.dev, 1, n)
            kept = int(kc.item())
            a_n = 20.0 * n + n * n / 4.0 + 8.0 * kept            # A(N) = 20 N + N^2/8 written + N^2/8 read + 8 K
            r[tag] = {"sort_ms": t_sort, "mask_ms": t_mask, "reduce_ms": t_red, "whole_call_ms": t_all, "kept": kept,
                      "boxes_per_s": n / t_all * 1e3,
                      "path": ("pruned: x-bucket band pair search (band narrowed by the threshold) + free-running fixed-point keep list on 64 workgroups" if mode == 0 else "dense mask"),
                      "fixed_point_rounds": rounds if mode == 0 and fpdone else None, "suppressing_pairs": npairs if mode == 0 else None,
                      "candidates_per_box": cpb, "estimated_pairs_per_box": est,
                      "survey_A_N_frac": a_n / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "mask_hbm_frac": ((n * n / 16.0 + 20 * n) / (t_mask * 1e-3) / 1e9 / HBM_PEAK_GBPS) if mode == 1 else None,
                      "mask_valu_frac": valu_frac("k_nms_mask", n, t_mask) if mode == 1 else None}
    r["note"] = ("config 5 (replicas only).  survey_A_N_frac = SURVEY 8(d)'s A(N) (which prices an N x N mask written and read) over "
                 "the whole call's time and 8 TB/s: the pruned path never forms that mask -- pairs of boxes that do not overlap cannot "
                 "suppress, so it searches a band of the x-bucket order and moves a few MB -- which is why this is an equivalent rate, "
                 "not memory traffic; the path is chosen per image from the candidate pairs per box (x-bucket order) and, once the "
                 "threshold is known, from the suppressing pairs of 1024 sampled boxes (more than 128 per box: dense mask, k_nms_mask, "
                 "VALU-bound: mask_valu_frac = SQ_INSTS_VALU x 64 / time over the chip's 7.86e13 lane-ops/s); phases are timed one by one (sort includes the x-bucket sort of the pruned "
                 "path), whole_call_ms is dg_nms_batched_f32 as one call; medians of 50 launches after 20 warm-up")
    return r


def extra_kmeans(d, n_total=10_000_000, iters=30):
    from vision_amd import cluster, shard as sh
    from tools import workloads as wl
    wh_all = wl.anchor_wh(0, n_total)
    init = cluster.kmean.init_centers(wh_all, 5, 42)
    lo, hi = sh.shard_range(n_total, d.rank, d.world)
    per = hi - lo
    shard = torch.from_numpy(wh_all[lo:hi]).to(d.dev)
    loops = None
    if d.world > 1 and not d.rehearse:
        # every way to issue the sharded loop, most conservative first (launch + torch.distributed all-reduce per iteration from
        # Python, the same captured with torch.cuda.graph, then the library's own communicator and C-side loop -- never run on
        # more than one rank before an 8-GPU node sees it -- eager and captured): the watchdog in main() publishes what has
        # finished should a later one hang on this node
        loops = {}
        for name, comm, graph in (("torch_eager", "torch", "0"), ("torch_graph", "torch", "1"), ("so_eager", "so", "0"), ("so_graph", "so", "1")):
            os.environ["DG_KMEANS_GRAPH"] = graph
            try:                                   # a way that fails alike on every rank (no librccl.so, say) must not cost the others
                ke = cluster.kmean(5, 42, comm=comm)
                ke.fit(shard, init_centers=init, fixed_iters=iters)
                ke._run_iters(0, iters)                                   # the second run of a chunk is the one that captures
                sec_e = min(Dist.timed(d, lambda: ke._run_iters(0, iters), 1, 0) for _ in range(5))
                loops[name] = {"iters_per_s": iters / sec_e, "ms_per_iter": sec_e / iters * 1e3, "state": ke.graph_state, "comm": ke.comm_state}
                ke.close()
                del ke
            except Exception as e:  # noqa: BLE001
                loops[name] = {"error": f"{type(e).__name__}: {e}"}
            PARTIAL["config4_kmeans_10M"] = {"loops": dict(loops), "ranks": d.world, "note": "a later way of issuing the loop did not finish"}
        os.environ.pop("DG_KMEANS_GRAPH", None)
    km = cluster.kmean(5, 42, reduce_device="cpu" if d.rehearse else None)   # rehearsal: gloo reduces host tensors
    km.fit(shard, init_centers=init, fixed_iters=iters)     # warm-up (RCCL channels, graph capture); leaves the device state
    single = d.world == 1 and not d.rehearse
    replay = None
    if single:
        # one device: the timed iterations are A FIT'S -- from the initial centroids, labels unset, boxes regrouped (the reset is
        # not timed; the grouping's time is reported apart as prepare_device_ms).  The prepared iterations depend on the state
        # (pieces move between "decided as a whole" and "walked" while the centroids move), so a replay from the state the fit
        # left is a different, slightly lighter workload: reported beside it as `replay_from_fitted_state`.
        be = km._backend
        init_dev = torch.from_numpy(np.asarray(init, np.float32)).to(d.dev).reshape(be.cent.shape)

        def reset():
            be.cent.copy_(init_dev)
            be.labels.fill_(255)
            if getattr(be, "cells", None) is not None:
                be.reprepare(fresh=True)                        # as fit() does: no labels yet

        def fit_iters_wall():
            reset()
            return Dist.timed(d, lambda: km._run_iters(0, iters), 1, 0)

        def fit_iters_dev():
            reset()
            return ev_time(lambda: km._run_iters(0, iters), 1, 0)

        sec = min(fit_iters_wall() for _ in range(5))
        ms = min(fit_iters_dev() for _ in range(3)) / iters
        fitted_again = bool(np.array_equal(be.cent.cpu().numpy().reshape(-1), np.asarray(km.cluster_centers_).reshape(-1)))
        replay = {"ms_per_iter": min(Dist.timed(d, lambda: km._run_iters(0, iters), 1, 0) for _ in range(5)) / iters * 1e3,
                  "device_ms_per_iter": min(ev_time(lambda: km._run_iters(0, iters), 3, 1) for _ in range(3)) / iters}
    else:
        sec = min(Dist.timed(d, lambda: km._run_iters(0, iters), 1, 0) for _ in range(5))   # the Lloyd iterations proper (box loop: state-independent)
        # device time of one iteration on this rank's shard (HIP events around the same 30 iterations: no host sync in it)
        ms = min(ev_time(lambda: km._run_iters(0, iters), 3, 1) for _ in range(3)) / iters
    r = {"iters_per_s": iters / sec, "ms_per_iter": sec / iters * 1e3, "boxes_total": n_total, "ranks": d.world,
         "sharded_loop": km.graph_state if d.world > 1 else "single device: dg_kmeans_iou_run",
         "comm": km.comm_state if d.world > 1 else None,
         "device_ms_per_iter_this_shard": ms, "iter_hbm_frac": (12.0 * per) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
         "iter_valu_frac": valu_frac("k_kmeans_step", per, ms) if d.world > 1 else None,   # (one device: the prepared path, see box_loop)
         "centers": km.cluster_centers_.round(6).tolist(), "loops": loops,
         "best_iters_per_s": max([iters / sec] + [v["iters_per_s"] for v in (loops or {}).values() if "iters_per_s" in v]),
         "note": "config 4, strong scaling: 10M boxes sharded; per iteration ONE launch (the centroid update is the prologue of "
                 "the next step) + ONE all-reduce of the 2 KiB partial-sums buffer, issued by the library itself (dg_comm_* / "
                 "dg_kmeans_iou_run_sharded; comm says which path ran; `loops` times the other ways of issuing it, best_iters_per_s the fastest); "
                 "timed: best of 5 runs of 30 iterations, device-resident; the inertia sum is formed by the last iteration of a "
                 "run only, on one rank and on many alike; device_ms_per_iter = HIP-event time of the same 30 iterations / 30 "
                 "(k_kmeans_step is VALU-bound: iter_valu_frac = SQ_INSTS_VALU x 64 / time over the chip's 7.86e13 lane-ops/s)"}
    km.close()                                   # frees the library-side communicator, if this fit made one (collective)
    if d.world == 1 and not d.rehearse:
        # round 3: the single-device fit runs PREPARED (boxes grouped by (w,h) cell once per fit, iterations over pieces of
        # cells; dg_kmeans_prepare / dg_kmeans_iou_run_prepared) -- the figures above are that path.  The box-by-box loop
        # (k_kmeans_step every iteration, what the sharded fit still runs) beside it, with its HBM fraction at SURVEY's
        # 12 B/box and at the bytes the kernel really moves (8 B box + 1 B old label + 1 B new label = 10 B/box)
        r["path"] = "prepared (cells)" if getattr(km._backend, "cells", None) is not None else "box loop"
        r["timed"] = ("the 30 iterations of a fit from the initial centroids (state reset and boxes regrouped before every timed run, "
                      "untimed), best of 5; replay_from_fitted_state = the same 30 launches started from the state the fit left")
        r["replay_from_fitted_state"] = replay
        r["timed_fit_reaches_the_fit_centroids"] = fitted_again
        r["iter_hbm_frac_note"] = ("the prepared iterations do not read the boxes they decide piece-wise, so SURVEY's 12 B/box over "
                                   "their time is an equivalent rate, not memory traffic")
        try:
            os.environ["DG_KMEANS_CELLS"] = "0"
            kb = cluster.kmean(5, 42)
            kb.fit(shard, init_centers=init, fixed_iters=iters)
            msb = min(ev_time(lambda: kb._run_iters(0, iters), 3, 1) for _ in range(3)) / iters
            r["box_loop"] = {"device_ms_per_iter": msb, "iters_per_s": 1e3 / msb, "iter_hbm_frac_12B": (12.0 * per) / (msb * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "iter_hbm_frac_own_10B": (10.0 * per) / (msb * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "iter_valu_frac": valu_frac("k_kmeans_step", per, msb),
                             "same_centers": bool(np.array_equal(kb.cluster_centers_, km.cluster_centers_))}
            # a whole fit from Python, wall clock (workspace allocation, the one-time grouping by cell, 30 iterations, labels
            # widened, history read back), second fit of each kind in this process; device time of the grouping alone beside it
            walls = {}
            for cells in ("1", "0"):
                os.environ["DG_KMEANS_CELLS"] = cells
                best = None
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    kp = cluster.kmean(5, 42); kp.fit(shard, init_centers=init, fixed_iters=iters); torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) * 1e3
                    best = dt if best is None else min(best, dt)
                walls[cells] = best
            # and a fit that runs until no label changes (kmean's default mode: chunks of iterations between two host checks of the
            # changed counts; the prepared fit's chunks grow and run over cells only until the last one), wall clock, best of 3
            conv = {}
            for cells in ("1", "0"):
                os.environ["DG_KMEANS_CELLS"] = cells
                best, kc = None, None
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    kc = cluster.kmean(5, 42); kc.fit(shard, init_centers=init); torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) * 1e3
                    best = dt if best is None else min(best, dt)
                conv["prepared" if cells == "1" else "box_loop"] = {"wall_ms": best, "n_iter": kc.n_iter_, "inertia": kc.inertia_}
            r["fit_to_convergence"] = conv
            r["fit_30_iters_wall_ms_incl_prepare"] = walls["1"]
            r["fit_30_iters_wall_ms_box_loop"] = walls["0"]
            os.environ["DG_KMEANS_CELLS"] = "1"
            kp = cluster.kmean(5, 42); kp.fit(shard, init_centers=init, fixed_iters=iters)
            be = kp._backend
            if getattr(be, "cells", None) is not None and hasattr(be, "reprepare"):
                be.labels.fill_(255)
                r["prepare_device_ms"] = min(ev_time(lambda: be.reprepare(fresh=True), 3, 1) for _ in range(3))   # a fit's: no labels yet
                r["prepare_device_ms_with_labels"] = min(ev_time(be.reprepare, 3, 1) for _ in range(3))            # re-grouping mid-fit
        except Exception as e:  # noqa: BLE001
            r["box_loop"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            os.environ.pop("DG_KMEANS_CELLS", None)
    if d.world == 1 and not d.rehearse:
        # what the sharded loop costs per iteration on a 1/8 shard (1.25 M boxes), all-reduce included, on this one GPU:
        # a single-rank RCCL group runs the same launch + all-reduce sequence the 8-GPU job runs on every rank
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=d.dev)
            sub = torch.from_numpy(wh_all[:n_total // 8]).to(d.dev)
            for comm in ("so", "torch"):           # the library's own communicator and C-side loop / torch.distributed from Python
                for mode in ("1", "0"):
                    os.environ["DG_KMEANS_GRAPH"] = mode
                    k8 = cluster.kmean(5, 42, sharded=True, comm=comm)
                    k8.fit(sub, init_centers=init, fixed_iters=iters)
                    s8 = min(Dist.timed(d, lambda: k8._run_iters(0, iters), 1, 0) for _ in range(5))
                    r["shard_1p25M_loop_%s_%s" % (comm, "graph" if mode == "1" else "eager")] = {
                        "us_per_iter": s8 / iters * 1e6, "state": k8.graph_state, "comm": k8.comm_state}
                    k8.close()
            os.environ["DG_KMEANS_GRAPH"] = "1"
            k1 = cluster.kmean(5, 42)                 # the same shard on the single-device loop: kernel time without a collective
            k1.sharded = False
            k1.fit(sub, init_centers=init, fixed_iters=iters)
            r["shard_1p25M_device_us_per_iter_no_collective"] = min(ev_time(lambda: k1._run_iters(0, iters), 3, 1) for _ in range(3)) / iters * 1e3
            os.environ.pop("DG_KMEANS_GRAPH", None)
            dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001
            r["shard_1p25M_loop_error"] = f"{type(e).__name__}: {e}"
    return r


T0 = time.perf_counter()
PARTIAL = {}          # side results a section publishes before a step that could hang (read by the watchdog)


def start_watchdog(d, line, extra, seconds):
    """The side configurations run after the primary measurement.  Should one of them hang (a collective that never
    returns, say), every rank leaves after `seconds`: rank 0 prints the primary line with what the extras had produced and
    `"extras_hung": true`, and every rank exits with status 3 (the hang stays visible to the driver)."""
    import threading

    def fire():
        try:
            if d.rank == 0:
                ex = dict(extra)
                for k, v in PARTIAL.items():
                    ex.setdefault(k, v)
                ex["watchdog"] = f"side configurations did not finish within {seconds} s; primary line printed by the watchdog"
                out = dict(line)
                out["extra"] = ex
                out["extras_hung"] = True
                out.setdefault("roofline", None)
                out.setdefault("cpu_baseline", None)
                sys.stdout.write(json.dumps(out, default=str) + "\n")
                sys.stdout.flush()
            print(f"[bench rank {d.rank}] watchdog: leaving after {seconds} s in the side configurations", file=sys.stderr, flush=True)
        except BaseException as e:  # noqa: BLE001
            print(f"[bench rank {d.rank}] watchdog failed to print the line: {e!r}", file=sys.stderr, flush=True)
        finally:
            os._exit(3)      # the primary line is out, but a hang in the side configurations must not look like a clean run
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def trace(d, msg):
    if os.environ.get("DG_BENCH_TRACE") == "1":
        print(f"[bench rank {d.rank} t={time.perf_counter() - T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


class StubBatch:
    """DG_BENCH_STUB=1: a trivial CPU workload with the FrcnnBatch surface, so that the launcher, the process group, the
    barriers / max-over-ranks timing and the reporting run where there is no GPU (tests/test_bench_launch.py).  It computes
    nothing of the hot path and its line says so ("data": "stub")."""
    n_rois = 2000

    def __init__(self, d, images_total):
        from vision_amd import shard
        lo, hi = shard.shard_range(images_total, d.rank, d.world)
        self.B, self.n = hi - lo, (hi - lo) * self.n_rois
        self.x = np.arange(self.n, dtype=np.float64)

    def step(self):
        self.x = np.sqrt(self.x + 1.0)

    def check(self):
        assert np.isfinite(self.x).all()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=256, help="images per batch (config 3: 256)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank runs a full --images batch; strong: ONE --images batch split over the ranks")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))            # no GPU call has been made in this process
    stub = os.environ.get("DG_BENCH_STUB") == "1"
    d = Dist(a.gpus, stub=stub)
    images_total = a.images * d.world if a.scaling == "weak" else a.images
    if stub:
        w = StubBatch(d, images_total)
    else:
        from vision_amd import _abi
        _abi.load()
        trace(d, "init done")
        w = FrcnnBatch(d, images_total)
    trace(d, "workload resident")
    sec = d.timed(w.step, a.steps, a.warmup)
    trace(d, "timed region done")
    w.check()
    total_rois = images_total * w.n_rois
    ms_step = sec / a.steps * 1e3
    value = total_rois * a.steps / sec
    line = {"metric": "RoIs/sec RoI-pool fwd+bwd + batched NMS (Fast R-CNN batch)", "value": value, "unit": "RoIs/s",
            "n_gpus": d.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "stub" if stub else "synthetic",
            "config": {"workload": f"configs[2]: Fast R-CNN batch {a.images} images x 2000 RoIs{' per GPU' if a.scaling == 'weak' else ''}, "
                                   "512x38x50 maps, RoI max-pool 7x7 fwd(+argmax)+bwd + batched NMS thr 0.7, sharded by image",
                       "images_per_gpu": w.B, "global_batch_images": images_total, "rois_per_image": 2000,
                       "parallelism": f"image-shard x{d.world}, no data-path collective", "backend": d.backend,
                       "world_size_seen": dist.get_world_size() if d.world > 1 else 1}}
    if stub:
        finish(d, line)
        return

    # dominant kernel and the other phases: 20 warm-up + median of 50 launches, HIP events on the launch stream (rank 0's shard)
    reps = dict(iters=50, warm=20) if w.B >= 8 else dict(iters=100, warm=20)
    t_fwd = ev_median(w.fwd, **reps)
    phases = {"windows_ms": ev_median(w.windows, **reps), "nhwc_stage_ms": ev_median(w.stage, **reps), "roipool_fwd_ms": t_fwd,
              "roipool_bwd_ms": ev_median(w.bwd, **reps), "nms_batched_ms": ev_median(w.nms, **reps),
              "timing": f"median of {reps['iters']} launches after {reps['warm']} warm-up, one HIP event pair per launch"}
    fwd_bytes = w.n * ROI_BYTES + w.B * MAP_BYTES + w.n * 20
    achieved = fwd_bytes / (t_fwd * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_roipool_fwd_bytes_per_launch", {}).get(str(w.n))
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_roipool_fwd<max>", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "bytes_per_launch": fwd_bytes, "avg_launch_ms": t_fwd,
                "bwd_frac": (w.n * ROI_BYTES + w.B * MAP_BYTES) / (phases["roipool_bwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    gout_alias = w.gout_alias
    # the same step with the argmax kept as uint16 (what slowroipool's autograd path does; H*W <= 65535): every rank runs it
    a16 = None
    if not a.no_extras:
        err = None
        try:
            w.arg = torch.empty(tuple(w.out.shape), dtype=torch.int16, device=d.dev)
        except torch.OutOfMemoryError as e:
            err = str(e)
        # every rank takes part in the timed region's barriers or none does: a rank that skipped them alone would pair its
        # next collective with the others' barriers
        if d.max(0.0 if err is None else 1.0) > 0.0:
            a16 = {"error": err or "another rank ran out of memory"}
        else:
            sec16 = d.timed(w.step, a.steps, 1)
            f16, b16 = ev_median(w.fwd, 20, 5), ev_median(w.bwd, 20, 5)
            a16 = {"value": total_rois * a.steps / sec16, "unit": "RoIs/s", "ms_per_step": sec16 / a.steps * 1e3,
                   "roipool_fwd_ms": f16, "roipool_bwd_ms": b16, "bytes_per_roi": ROI_BYTES_A16,
                   "fwd_hbm_frac": (w.n * ROI_BYTES_A16 + w.B * MAP_BYTES + w.n * 20) / (f16 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "bwd_hbm_frac": (w.n * ROI_BYTES_A16 + w.B * MAP_BYTES) / (b16 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "note": "same step, argmax stored as uint16 instead of int32: the primary line keeps the int32 layout "
                           "BASELINE configs[2] sizes (51.4 GB of argmax); fractions here use this layout's own bytes"}
    del w
    torch.cuda.empty_cache()
    extra = {"phases_ms_rank0": phases, "grad_out_aliases_out": gout_alias, "config3_uint16_argmax": a16}
    line["roofline"] = roofline
    dog = start_watchdog(d, line, extra, int(os.environ.get("DG_BENCH_EXTRAS_TIMEOUT", "240")))

    def side(name, fn):
        # a side configuration that fails is reported as such; it must not cost the primary line
        trace(d, f"{name} ...")
        try:
            extra[name] = fn(d)
        except Exception as e:  # noqa: BLE001
            extra[name] = {"error": f"{type(e).__name__}: {e}"}
            print(f"[bench rank {d.rank}] {name} failed: {e!r}", file=sys.stderr, flush=True)

    # ONE 256-image batch split over the ranks (BASELINE configs[2] as worded), next to the weak-scaling primary line
    if d.world > 1 and a.scaling == "weak" and not a.no_extras:
        def strong(d):
            ws, err = None, None
            try:
                ws = FrcnnBatch(d, a.images)
            except torch.OutOfMemoryError as e:
                err = str(e)
            if d.max(0.0 if err is None else 1.0) > 0.0:       # all ranks or none enter the timed region's barriers
                return {"error": err or "another rank ran out of memory"}
            sec_s = d.timed(ws.step, a.steps, a.warmup)
            ws.check()
            return {"value": a.images * ws.n_rois * a.steps / sec_s, "unit": "RoIs/s", "ms_per_step": sec_s / a.steps * 1e3,
                    "images_per_gpu": ws.B, "global_batch_images": a.images, "scaling": "strong"}
        side("config3_strong_scaling", strong)
        torch.cuda.empty_cache()

    # what a plain device-to-device copy reaches on this box (read + write bytes), as SURVEY 8(d) asks: the second yardstick
    src = torch.empty(1 << 30, dtype=torch.float32, device=d.dev)
    dst = torch.empty_like(src)
    t_copy = ev_median(lambda: dst.copy_(src), 20, 5)
    copy_gbps = 2 * src.numel() * 4 / (t_copy * 1e-3) / 1e9
    # and what a plain device fill reaches: the forward kernel writes 99 % of its bytes, so the store rate of the box is its
    # real ceiling (tools/ubench/store_bw.hip: 5.4-5.7 TB/s for every pattern and launch shape tried)
    t_fill = ev_median(lambda: dst.fill_(1.0), 20, 5)
    fill_gbps = dst.numel() * 4 / (t_fill * 1e-3) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    roofline["d2d_copy_GBps"] = copy_gbps
    roofline["frac_of_d2d_copy"] = achieved / copy_gbps
    roofline["device_fill_GBps"] = fill_gbps
    roofline["frac_of_device_fill"] = achieved / fill_gbps

    if not a.no_extras:
        if d.rank == 0:
            side("config1_single_image", extra_single_image)
            side("config2_yolo1_b64", extra_yolo)
            side("config3_wide_rois", extra_wide_rois)
            side("config5_stress_64k", extra_stress)
            torch.cuda.empty_cache()
        trace(d, "extras on rank 0 done / waiting")
        d.barrier()
        side("config4_kmeans_10M", extra_kmeans)
        trace(d, "k-means done")
    cpu = None
    if d.rank == 0 and d.world == 1 and not a.no_cpu:
        cpu = cpu_baseline_frcnn()
    dog.cancel()
    line.update({"roofline": roofline, "cpu_baseline": cpu, "extra": extra})
    finish(d, line)


def finish(d, line):
    if d.rank == 0:
        assert line["n_gpus"] == d.world
        print(json.dumps(line))
    sys.stdout.flush()
    if d.world > 1:
        # leave together; a teardown that does not come back within 30 s must not hold the whole job
        import threading
        d.barrier()
        trace(d, "final barrier passed")
        t = threading.Thread(target=dist.destroy_process_group, daemon=True)
        t.start()
        t.join(30.0)
        trace(d, "process group destroyed" if not t.is_alive() else "destroy_process_group still running: exiting anyway")
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
