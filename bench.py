#!/usr/bin/env python3
"""bench.py — box-geometry hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Primary line (ONE JSON line on rank 0): BASELINE.json configs[2] — Fast R-CNN batch of 256 images x
2000 RoIs on 512x38x50 maps: RoI max-pool forward (+argmax), backward, and batched NMS (thr 0.7),
sharded by image over the ranks with no data-path collective (total work fixed => "strong").
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


def ev_time(fn, iters, warm=1):
    """median-free mean of `iters` runs timed with HIP events on the current stream (ms)."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / iters


class Dist:
    def __init__(self, n):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != n and self.world > 1:
            raise SystemExit(f"--gpus {n} but WORLD_SIZE={self.world}")
        # DG_BENCH_REHEARSE=1: every rank on cuda:0 with gloo -- a functional rehearsal of the N>1 path on a one-GPU box
        rehearse = self.rehearse = os.environ.get("DG_BENCH_REHEARSE") == "1"
        dev_index = 0 if rehearse else self.local
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        if self.world > 1:
            if rehearse:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)

    def barrier(self):
        if self.world > 1:
            dist.barrier()
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
    def __init__(self, d, images_total, n_rois=2000, C=512, H=38, W=50, nms_thr=0.7):
        from vision_amd import ops, shard
        from tools import workloads as wl
        self.ops, self.d = ops, d
        first, last = shard.shard_range(images_total, d.rank, d.world)     # contiguous image range of this rank
        per = last - first
        self.B, self.n_rois, self.shape = per, n_rois, (per, C, H, W)
        dev = d.dev
        g = torch.Generator(device=dev).manual_seed(1234 + d.rank)
        self.img = torch.randn((per, C, H, W), generator=g, device=dev, dtype=torch.float32)
        rois = np.concatenate([wl.frcnn_rois(first + i, n_rois) for i in range(per)])
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
    return {"value": images * n_rois / dt, "unit": "RoIs/s", "cores": 1, "kind": "port",
            "sample": f"{images} images x {n_rois} RoIs of the same step (C oracle: RoI max-pool fwd+bwd + NMS 0.7), {dt:.1f} s"}


# ------------------------------------------------------------------ extras
def extra_single_image(d):
    w = FrcnnBatch(d if d.world == 1 else _solo(d), 1)
    w.windows(); w.stage()
    r = {}
    t = ev_time(w.fwd, 20, 3); r["roipool_fwd_ms"] = t
    r["roipool_fwd_rois_per_s"] = 2000 / t * 1e3
    r["roipool_fwd_hbm_frac"] = (2000 * ROI_BYTES + MAP_BYTES) / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS
    t = ev_time(w.bwd, 20, 3); r["roipool_bwd_ms"] = t
    r["roipool_bwd_hbm_frac"] = (2000 * ROI_BYTES + MAP_BYTES) / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS
    for thr in (0.7, 0.3):
        w.nms_thr = thr
        t = ev_time(w.nms, 20, 3)
        r[f"nms2000_thr{thr}_ms"] = t
        r[f"nms2000_thr{thr}_boxes_per_s"] = 2000 / t * 1e3
    r["note"] = "config 1; NMS at N=2000 is launch-latency-bound (3 kernels)"
    return r


class _solo:
    """view of a Dist as a single-rank job (for the replicas-only extras)"""
    def __init__(self, d):
        self.world, self.rank, self.dev = 1, 0, d.dev


def extra_yolo(d):
    from vision_amd import yolo_v1
    from tools import workloads as wl
    out, tgt = wl.yolo1_batch(0, 64)
    o, t = torch.from_numpy(out).to(d.dev), torch.from_numpy(tgt).to(d.dev)
    ms = ev_time(lambda: yolo_v1.decode_nms_batched(o, t, 0.2, 0.01, 0.3), 50, 5)
    from vision_amd import ops
    ms_loss = ev_time(lambda: ops.yolo1_loss(o, t, True, True), 50, 5)
    return {"decode_nms_b64_ms": ms, "boxes_per_s": 64 * 98 / ms * 1e3,
            "loss_fwd_bwd_b64_ms": ms_loss, "loss_cells_per_s": 64 * 49 / ms_loss * 1e3,
            "note": "config 2: 4 kernels, ~0.4 MB moved: launch-latency-bound; NMS is class-agnostic per image as in the reference; "
                    "loss = fused YOLOv1 loss forward + gradient (2 kernels; the reference issues ~60 torch kernels + autograd)"}


def extra_stress(d, n=65536):
    from vision_amd import ops, _abi
    from tools import workloads as wl
    r = {}
    b, s = wl.stress_boxes(0, n, dense=False)
    bt, st = torch.from_numpy(b).to(d.dev), torch.from_numpy(s).to(d.dev)
    out = torch.empty((n, n), dtype=torch.float32, device=d.dev)
    ms = ev_time(lambda: ops.iou_pairwise(bt, bt, out=out), 5, 2)
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
            t_sort = ev_time(lambda: lib.dg_nms_phase_sort(_abi.ptr(bt), _abi.ptr(st), None, 1, n, _abi.ptr(ws), nb, sp), 5, 1)
            t_mask = ev_time(lambda: lib.dg_nms_phase_mask(None, 1, n, thr, _abi.ptr(ws), nb, sp), 5, 1)
            t_red = ev_time(lambda: lib.dg_nms_phase_reduce(None, 1, n, 1, _abi.ptr(keep), _abi.ptr(kc), _abi.ptr(ws), nb, sp), 3, 1)
            tot = t_sort + t_mask + t_red
            r[tag] = {"sort_ms": t_sort, "mask_ms": t_mask, "reduce_ms": t_red, "kept": int(kc.item()),
                      "boxes_per_s": n / tot * 1e3,
                      "mask_hbm_frac": (n * n / 16.0 + 20 * n) / (t_mask * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    r["note"] = "config 5 (replicas only); mask bytes = upper-triangular N^2/16 written; reduce is a serial chain"
    return r


def extra_kmeans(d, n_total=10_000_000, iters=30):
    from vision_amd import cluster, shard as sh
    from tools import workloads as wl
    wh_all = wl.anchor_wh(0, n_total)
    init = cluster.kmean.init_centers(wh_all, 5, 42)
    lo, hi = sh.shard_range(n_total, d.rank, d.world)
    per = hi - lo
    shard = torch.from_numpy(wh_all[lo:hi]).to(d.dev)
    del wh_all
    km = cluster.kmean(5, 42, reduce_device="cpu" if d.rehearse else None)   # rehearsal: gloo reduces host tensors
    km.fit(shard, init_centers=init, fixed_iters=iters)     # warm-up (RCCL channels, workspace); leaves the device state
    sec = Dist.timed(d, lambda: km._run_iters(0, iters), 1, 0)   # the Lloyd iterations proper (step [+ all-reduce] + update)
    # kernel-only rate of the step on this shard
    be = km._backend
    ms = ev_time(be.step, 20, 3)
    return {"iters_per_s": iters / sec, "ms_per_iter": sec / iters * 1e3, "boxes_total": n_total, "ranks": d.world,
            "step_kernel_ms_per_shard": ms, "step_hbm_frac": (12.0 * per) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "centers": km.cluster_centers_.round(6).tolist(),
            "note": "config 4, strong scaling: 10M boxes sharded, one 17 x int64 all-reduce per iteration; "
                    "timed: 30 iterations of step [+ all-reduce] + centroid update, device-resident; on one rank the device-side loop forms the inertia sum in the last iteration only (a Lloyd iteration does not use it), the sharded loop in every step"}


T0 = time.perf_counter()


def trace(d, msg):
    if os.environ.get("DG_BENCH_TRACE") == "1":
        print(f"[bench rank {d.rank} t={time.perf_counter() - T0:7.2f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=256, help="images in the global batch (config 3: 256)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    d = Dist(a.gpus)
    from vision_amd import _abi
    _abi.load()

    trace(d, "init done")
    w = FrcnnBatch(d, a.images)
    trace(d, "workload resident")
    sec = d.timed(w.step, a.steps, a.warmup)
    trace(d, "timed region done")
    w.check()
    total_rois = a.images * w.n_rois
    ms_step = sec / a.steps * 1e3
    value = total_rois * a.steps / sec

    # dominant kernel, timed live with events on the launch stream (rank 0's shard)
    t_fwd = ev_time(w.fwd, max(3, a.steps), 1)
    phases = {"windows_ms": ev_time(w.windows, 3, 1), "nhwc_stage_ms": ev_time(w.stage, 3, 1), "roipool_fwd_ms": t_fwd,
              "roipool_bwd_ms": ev_time(w.bwd, 3, 1), "nms_batched_ms": ev_time(w.nms, 3, 1)}
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
        try:
            w.arg = torch.empty(tuple(w.out.shape), dtype=torch.int16, device=d.dev)
            sec16 = d.timed(w.step, a.steps, 1)
            f16, b16 = ev_time(w.fwd, 3, 1), ev_time(w.bwd, 3, 1)
            a16 = {"value": total_rois * a.steps / sec16, "unit": "RoIs/s", "ms_per_step": sec16 / a.steps * 1e3,
                   "roipool_fwd_ms": f16, "roipool_bwd_ms": b16, "bytes_per_roi": ROI_BYTES_A16,
                   "fwd_hbm_frac": (w.n * ROI_BYTES_A16 + w.B * MAP_BYTES + w.n * 20) / (f16 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "bwd_hbm_frac": (w.n * ROI_BYTES_A16 + w.B * MAP_BYTES) / (b16 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "note": "same step, argmax stored as uint16 instead of int32: the primary line keeps the int32 layout "
                           "BASELINE configs[2] sizes (51.4 GB of argmax); fractions here use this layout's own bytes"}
        except torch.OutOfMemoryError as e:
            a16 = {"error": str(e)}
    del w
    torch.cuda.empty_cache()
    # what a plain device-to-device copy reaches on this box (read + write bytes), as SURVEY 8(d) asks: the second yardstick
    src = torch.empty(1 << 30, dtype=torch.float32, device=d.dev)
    dst = torch.empty_like(src)
    t_copy = ev_time(lambda: dst.copy_(src), 5, 2)
    copy_gbps = 2 * src.numel() * 4 / (t_copy * 1e-3) / 1e9
    del src, dst
    torch.cuda.empty_cache()
    roofline["d2d_copy_GBps"] = copy_gbps
    roofline["frac_of_d2d_copy"] = achieved / copy_gbps

    extra = {"phases_ms_rank0": phases, "grad_out_aliases_out": gout_alias, "config3_uint16_argmax": a16}
    def side(name, fn):
        # a side configuration that fails is reported as such; it must not cost the primary line
        try:
            extra[name] = fn(d)
        except Exception as e:  # noqa: BLE001
            extra[name] = {"error": f"{type(e).__name__}: {e}"}
            print(f"[bench rank {d.rank}] {name} failed: {e!r}", file=sys.stderr, flush=True)

    if not a.no_extras:
        if d.rank == 0:
            side("config1_single_image", extra_single_image)
            side("config2_yolo1_b64", extra_yolo)
            side("config5_stress_64k", extra_stress)
            torch.cuda.empty_cache()
        trace(d, "extras on rank 0 done / waiting")
        d.barrier()
        side("config4_kmeans_10M", extra_kmeans)
        trace(d, "k-means done")
    cpu = None
    if d.rank == 0 and d.world == 1 and not a.no_cpu:
        cpu = cpu_baseline_frcnn()

    if d.rank == 0:
        line = {"metric": "RoIs/sec RoI-pool fwd+bwd + batched NMS (Fast R-CNN batch)", "value": value, "unit": "RoIs/s",
                "n_gpus": d.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"configs[2]: Fast R-CNN batch {a.images} images x 2000 RoIs, 512x38x50 maps, "
                                       "RoI max-pool 7x7 fwd(+argmax)+bwd + batched NMS thr 0.7, sharded by image",
                           "global_batch_images": a.images, "rois_per_image": 2000, "parallelism": f"image-shard x{d.world}"},
                "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
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
