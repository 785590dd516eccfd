"""Randomised differential test of the fused YOLOv1 decode + NMS launch (and the unfused decode) on predictions OUTSIDE the domain
a trained head produces: negative and huge box sizes (inverted / enormous boxes), centre offsets far outside the cell, confidences
and class probabilities below 0 and above 1, a few infinities -- against the oracle's decode followed by its nms.
usage: python tools/stress_decode_odd.py [rounds] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vision_amd import ops
from tools import workloads as wl
from oracle import oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
fail, t0 = [], time.time()
odd = np.array([-1.0, -0.3, -1e-3, 0.0, 2.5, 40.0, 1e4, -1e4, 1e19, -1e19], np.float32)
for it in range(rounds):
    bs = int(rng.integers(1, 24)); Cn = int(rng.choice([1, 5, 20]))
    out, tgt = wl.yolo1_batch(int(rng.integers(1 << 30)), bs, Cn)
    if rng.random() < 0.6: out = rng.random(out.shape, dtype=np.float32)
    m = rng.random(out.shape) < float(rng.choice([0.01, 0.05, 0.3]))
    out = np.where(m, odd[rng.integers(0, len(odd), out.shape)], out).astype(np.float32)
    with_inf = rng.random() < 0.3
    if with_inf:
        out[rng.random(out.shape) < 0.003] = np.inf
        out[rng.random(out.shape) < 0.003] = -np.inf
    thr1 = float(rng.choice([0.05, 0.3, 0.5, 0.9])); q1 = bool(rng.integers(2)); pc = bool(rng.integers(2))
    ct, pt = float(rng.choice([0.2, 0.5, 0.9])), float(rng.choice([0.01, 0.1, 0.4]))
    f = ops.yolo1_decode_batched(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), ct, pt, thr1, pc, q1)
    fb, fl, fs, fc, fk, fkc = [f[i].cpu().numpy() for i in (0, 1, 3, 4, 7, 8)]
    for i in range(bs):
        b_, l_, c_, s_, tb_, tl_ = oracle.yolo1_decode(out[i], tgt[i], ct, pt)
        why = ""
        ok = fc[i] == len(b_)
        if not ok: why = f"candidates {fc[i]} vs {len(b_)}"
        if ok and len(b_):
            b_, l_, s_ = np.asarray(b_, np.float32), np.asarray(l_), np.asarray(s_, np.float32)
            if not np.array_equal(fb[i, :fc[i]], b_, equal_nan=True): ok, why = False, "boxes"
            elif not np.array_equal(fl[i, :fc[i]], l_): ok, why = False, "labels"
            elif not np.array_equal(fs[i, :fc[i]], s_, equal_nan=True): ok, why = False, "scores"
        if ok and len(b_):
            if pc:
                wk = [ii[oracle.nms(b_[ii], s_[ii], thr1, q1)] for ii in (np.flatnonzero(l_ == c) for c in range(Cn)) if ii.size]
                wk = np.concatenate(wk) if wk else np.zeros(0, np.int64)
            else:
                wk = oracle.nms(b_, s_, thr1, q1)
            if not (fkc[i] == len(wk) and np.array_equal(fk[i, :fkc[i]], wk)): ok, why = False, f"keep list ({fkc[i]} vs {len(wk)})"
        elif ok and fkc[i] != 0:
            ok, why = False, "keep count without candidates"
        if not ok:
            fail.append((it, i, bs, Cn, thr1, q1, pc, ct, pt, with_inf, why))
    if it % 10 == 9: print(f"round {it}: failures {len(fail)} ({time.time() - t0:.0f} s)", flush=True)
print("TOTAL FAILURES", len(fail))
for x in fail[:12]: print(x)
sys.exit(1 if fail else 0)
