#!/usr/bin/env python3
"""Golden-vector generator  (TEST INFRASTRUCTURE — never imported by the product).

Runs in the build container only (needs /root/reference).  It pulls the *named
function bodies* of the reference's hot path out of the reference files with
`ast` (or by line range for rcnn/fast_rcnn.py, which does not parse), executes
them under the installed torch/numpy on seeded synthetic inputs, and stores the
inputs + outputs as small .npz fixtures in tests/golden/.  No reference source
text is written anywhere: fixtures are data only.

    python oracle/gen_golden.py            # regenerates tests/golden/*.npz

Reference symbols executed:
  yolo/yolo_v1/modules.py : get_iou (73-98), encode (151-174), Loss (198-306: get_iou, forward + autograd)
  yolo/yolo_v1/yolo.py    : decode (115-186), nms (188-203), ap (306-367),
                            evaluate body (273-299, as text with a stub model)
  yolo/yolo_v2/utils.py   : get_point (228-234), get_iou (238-263)
  rcnn/utils.py           : get_iou scalar (50-67), calcul (153-171)
  rcnn/rcnn.py            : do_ss quota loop (27-65, text), box decode (287-310, text)
  rcnn/fast_rcnn.py       : slowroipool (351-377, text), nms (590-610, text)
  clustering/cluster_prac.py : lines 11-41 (sklearn demo; API shape only)
"""
import ast
import io
import os
import sys
import textwrap
import contextlib

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import workloads as wl  # noqa: E402
from oracle import oracle as orc  # noqa: E402

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def _read(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return f.read()


def extract_defs(rel, names, ns):
    """exec only the wanted top-level FunctionDef/ClassDef nodes of a parseable file."""
    tree = ast.parse(_read(rel))
    got = set()
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), rel, "exec"), ns)
            got.add(node.name)
    missing = set(names) - got
    assert not missing, (rel, missing)
    return ns


def extract_method(rel, cls, meth, ns):
    tree = ast.parse(_read(rel))
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == meth:
                    exec(compile(ast.Module([sub], []), rel, "exec"), ns)
                    return ns[meth]
    raise KeyError((rel, cls, meth))


def extract_lines(rel, a, b, ns, dedent=True):
    """exec text lines a..b (1-based, inclusive) of a file."""
    lines = _read(rel).splitlines()[a - 1:b]
    code = "\n".join(lines)
    if dedent:
        code = textwrap.dedent(code)
    exec(compile(code, f"{rel}:{a}-{b}", "exec"), ns)
    return ns


def base_ns():
    return {"np": np, "torch": torch, "nn": nn}


# ------------------------------------------------------------------ YOLOv1 loss
def _loss_inputs(seed, batch, num_classes, objects_per_image):
    """modely with realistic + adversarial entries, targety with 1..k object cells per image."""
    rng = np.random.default_rng(seed)
    D = 10 + num_classes
    out = rng.uniform(0, 1, (batch, 7, 7, D)).astype(np.float32)
    out[..., 2:4] = rng.uniform(-0.3, 0.9, (batch, 7, 7, 2))      # negative w/h: sign*sqrt|.| and abs-area paths
    out[..., 7:9] = rng.uniform(-0.3, 0.9, (batch, 7, 7, 2))
    tgt = np.zeros_like(out)
    for b in range(batch):
        for _ in range(objects_per_image):
            xy = rng.uniform(0.02, 0.6, 2); wh = rng.uniform(0.05, 0.38, 2)
            lab = int(rng.integers(1, num_classes + 1))
            t = orc.yolo1_encode(np.concatenate([xy, wh]), lab, num_classes).astype(np.float32)
            cell = t[..., 4] > 0
            tgt[b][cell] = t[cell]
    # a few object cells whose predictions sit far from the target: both IoUs 0 (tie -> slot 1 is responsible)
    obj = np.argwhere(tgt[..., 4] > 0)
    for (b, i, j) in obj[:: max(1, len(obj) // 6)]:
        out[b, i, j, 0:2] = [0.0, 0.0]; out[b, i, j, 2:4] = [0.001, 0.001]
        out[b, i, j, 5:7] = [1.0, 1.0]; out[b, i, j, 7:9] = [0.001, 0.001]
    return out, tgt


def gen_encode():
    ns = extract_defs("yolo/yolo_v1/modules.py", {"encode"}, base_ns())
    rng = np.random.default_rng(8)
    bbs, labs, tgts, centres = [], [], [], []
    for k in range(40):
        C = 5 if k < 30 else 20
        bb = np.concatenate([rng.uniform(0.0, 0.62, 2), rng.uniform(0.02, 0.38, 2)])
        if k == 7:
            bb = np.array([0.0, 0.0, 2.0 / 7.0, 2.0 / 7.0])        # centre exactly on a cell corner
        lab = int(rng.integers(1, C + 1))
        b2 = bb.copy()
        t = ns["encode"](b2, lab, C)
        bbs.append(bb); labs.append(lab); centres.append(b2)
        tgts.append(t if C == 5 else None)
        if C == 20:
            tgts[-1] = t
    np.savez_compressed(os.path.join(OUT, "yolo1_encode.npz"), bb=np.array(bbs), label=np.array(labs, np.int64),
                        bb_after=np.array(centres), target_c5=np.array(tgts[:30]), target_c20=np.array(tgts[30:]))


def gen_loss():
    ns = base_ns()
    extract_defs("yolo/yolo_v1/modules.py", {"get_iou", "Loss"}, ns)
    cases = {}
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # Loss.__init__ moves its cell grid to the GPU; there is none here
    try:
        for name, (seed, batch, C, k) in {"b16c5": (3, 16, 5, 1), "b4c20": (4, 4, 20, 3), "b1c5": (5, 1, 5, 2)}.items():
            out, tgt = _loss_inputs(seed, batch, C, k)
            m = ns["Loss"](2, C)
            x = torch.from_numpy(out).clone().requires_grad_(True)
            loss = m.forward(x, torch.from_numpy(tgt).clone())
            loss.backward()
            cases[name + "_modely"] = out
            cases[name + "_targety"] = tgt
            cases[name + "_loss"] = loss.detach().numpy()
            cases[name + "_grad"] = x.grad.numpy()
            # same loss in float64 (tells summation-order noise from real differences)
            x64 = torch.from_numpy(out).double().requires_grad_(True)
            l64 = m.forward(x64, torch.from_numpy(tgt).double())
            cases[name + "_loss_f64"] = l64.detach().numpy()
    finally:
        torch.Tensor.cuda = saved_cuda
    np.savez_compressed(os.path.join(OUT, "yolo1_loss.npz"), **cases)
    print("yolo1_loss:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in cases.items() if k.endswith("_loss")},
          [float(cases[n + "_loss"]) for n in ("b16c5", "b4c20", "b1c5")])



# ------------------------------------------------------------------ IoU
def gen_iou():
    ns = extract_defs("yolo/yolo_v1/modules.py", {"get_iou"}, base_ns())
    ns2 = extract_defs("yolo/yolo_v2/utils.py", {"get_iou"}, base_ns())
    loss_iou = extract_method("yolo/yolo_v1/modules.py", "Loss", "get_iou", base_ns())
    a = wl.random_boxes(1, 300)
    b = wl.random_boxes(2, 300)
    b[10:20] = a[10:20]                      # identical boxes -> a/(a+eps)
    a[30] = [0, 0, 1, 1]; b[30] = [1, 0, 2, 1]   # touching -> 0
    a[31] = [0, 0, 1, 1]; b[31] = [0, 0.5, 1, 1.5]  # 1/3
    a[32] = [0, 0, 1, 1]; b[32] = [2, 2, 3, 3]   # disjoint
    ta, tb = torch.from_numpy(a), torch.from_numpy(b)
    i1 = ns["get_iou"](ta.clone(), tb.clone()).numpy()
    i1b = ns2["get_iou"](ta.clone(), tb.clone()).numpy()
    assert np.array_equal(i1, i1b)
    i2 = loss_iou(None, ta.clone(), tb.clone()).numpy()
    # float64 inputs keep float64
    i1d = ns["get_iou"](ta.double(), tb.double()).numpy()
    # NaN / inf rows
    an, bn = a[:8].copy(), b[:8].copy()
    an[0, 0] = np.nan; bn[1, 3] = np.nan; an[2, 2] = np.inf; bn[3, 0] = -np.inf
    inan = ns["get_iou"](torch.from_numpy(an), torch.from_numpy(bn)).numpy()
    np.savez(os.path.join(OUT, "iou_tensor.npz"), a=a, b=b, iou_eps1e7=i1, iou_eps1e6=i2,
             iou_f64=i1d, a_nan=an, b_nan=bn, iou_nan=inan)

    # scalar (rcnn/utils.py:50-67)
    nss = extract_defs("rcnn/utils.py", {"get_iou"}, base_ns())
    rng = np.random.default_rng(3)
    rois = np.stack([rng.integers(0, 300, 64), rng.integers(0, 200, 64)], 1)
    rois = np.concatenate([rois, rois + rng.integers(1, 200, (64, 2))], 1).astype(np.int64)
    grt = np.array([100.5, 50.25, 260.75, 180.0])
    res = np.array([nss["get_iou"](list(map(int, r)), grt) for r in rois], np.float64)
    # touching + assertion cases
    touch = nss["get_iou"]([0, 0, 10, 10], np.array([10.0, 0, 20, 10]))
    bad = []
    for bb in ([5, 5, 5, 9], [5, 5, 9, 5], [9, 5, 5, 9]):
        try:
            nss["get_iou"](bb, grt); bad.append(0)
        except AssertionError:
            bad.append(1)
    np.savez(os.path.join(OUT, "iou_scalar.npz"), rois=rois, grt=grt, iou=res,
             touch=np.float64(touch), bad_boxes=np.array([[5, 5, 5, 9], [5, 5, 9, 5], [9, 5, 5, 9]]),
             bad_raises=np.array(bad))


# ------------------------------------------------------------------ NMS
def _ref_nms():
    ns = extract_defs("yolo/yolo_v1/modules.py", {"get_iou"}, base_ns())
    extract_defs("yolo/yolo_v1/yolo.py", {"nms"}, ns)
    return ns["nms"]


def gen_nms():
    nms = _ref_nms()
    cases = {}

    def run(name, boxes, scores, thr):
        boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
        scores = np.asarray(scores, np.float32)
        keep = nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy().astype(np.int64)
        cases[name + "_boxes"] = boxes
        cases[name + "_scores"] = scores
        cases[name + "_thr"] = np.float64(thr)
        cases[name + "_keep"] = keep

    run("empty", np.zeros((0, 4)), np.zeros((0,)), 0.3)
    run("one", [[0, 0, 1, 1]], [0.9], 0.3)
    run("two_disjoint", [[0, 0, 1, 1], [2, 2, 3, 3]], [0.9, 0.8], 0.3)
    run("two_overlap", [[0, 0, 1, 1], [0, 0, 1, 0.9]], [0.9, 0.8], 0.3)
    run("dup_then_disjoint", [[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3], [4, 4, 5, 5]],
        [0.9, 0.8, 0.7, 0.6], 0.3)
    run("all_identical", np.tile([[0.1, 0.1, 0.5, 0.5]], (17, 1)), wl.tiefree_scores(5, 17), 0.5)
    run("last_survivor_suppressed_by_km", [[0, 0, 1, 1], [2, 2, 3, 3], [2, 2, 3, 2.9]],
        [0.9, 0.8, 0.7], 0.3)
    # iou exactly at the threshold is suppressed (strict <): two unit squares, overlap 1/3 -> thr=fp32(1/3)
    run("iou_eq_thr", [[0, 0, 1, 1], [0, 0.5, 1, 1.5], [5, 5, 6, 6]], [0.9, 0.8, 0.1],
        float(np.float32(1.0) / np.float32(3.0)))
    for n in (63, 64, 65, 98, 129, 500):
        b = wl.random_boxes(100 + n, n)
        run(f"rand{n}", b, wl.tiefree_scores(200 + n, n), 0.3)
    b, s = wl.frcnn_proposals(0, 2000)
    for thr in (0.1, 0.3, 0.5, 0.7):
        run(f"frcnn2000_thr{int(thr * 10)}", b, s, thr)
    b, s = wl.stress_boxes(0, 5000, dense=True)
    run("dense5000_thr5", b, s, 0.5)
    b, s = wl.stress_boxes(1, 5000, dense=False)
    run("sparse5000_thr5", b, s, 0.5)
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **cases)

    # degenerate rcnn/fast_rcnn.py:590-610 "nms" (returns the top-score rect only)
    ns = extract_lines("rcnn/fast_rcnn.py", 590, 610, base_ns())
    rng = np.random.default_rng(9)
    rects = [rng.integers(0, 200, 4).astype(np.int64) for _ in range(8)]
    scores = [float(x) for x in rng.uniform(0.6, 1.0, 8)]
    r, s = ns["nms"](rects, scores)
    np.savez(os.path.join(OUT, "nms_frcnn_top.npz"), rects=np.array(rects), scores=np.array(scores),
             best_rect=np.asarray(r), best_score=np.float64(s))


# ------------------------------------------------------------------ decode / evaluate / ap
def gen_decode():
    ns = extract_defs("yolo/yolo_v1/modules.py", {"get_iou", "encode"}, base_ns())
    extract_defs("yolo/yolo_v1/yolo.py", {"decode", "nms", "ap"}, ns)
    # our synthetic target builder must agree with the reference's encode
    rng = np.random.default_rng(4)
    for _ in range(20):
        bb = np.concatenate([rng.uniform(0.05, 0.6, 2), rng.uniform(0.05, 0.35, 2)])
        lab = int(rng.integers(1, 6))
        assert np.array_equal(ns["encode"](bb.copy(), lab, 5), orc.yolo1_encode(bb, lab, 5))

    out, tgt = wl.yolo1_batch(0, batch=16)
    d = {"output": out, "target": tgt}
    for tag, (ct, pt) in {"a": (0.2, 0.01), "b": (0.2, 0.1), "c": (0.5, 0.3)}.items():
        counts = []
        for i in range(out.shape[0]):
            with contextlib.redirect_stdout(io.StringIO()):
                r = ns["decode"](torch.from_numpy(out[i]), torch.from_numpy(tgt[i]), ct, pt)
            box, lab, conf, score, tb, tl = r
            if isinstance(box, list):
                assert len(box) == 0
                box = np.zeros((0, 4), np.float32); lab = np.zeros((0,), np.int64)
                conf = np.zeros((0,), np.float32); score = np.zeros((0,), np.float32)
            else:
                assert box.dtype == torch.float32 and lab.dtype == torch.int64
                box, lab, conf, score = box.numpy(), lab.numpy(), conf.numpy(), score.numpy()
            assert tb.dtype == torch.float64
            counts.append(len(box))
            d[f"{tag}_box_{i}"] = box; d[f"{tag}_label_{i}"] = lab
            d[f"{tag}_conf_{i}"] = conf; d[f"{tag}_score_{i}"] = score
            d[f"{tag}_tbox_{i}"] = tb.numpy(); d[f"{tag}_tlabel_{i}"] = np.int64(tl)
        d[f"{tag}_thr"] = np.array([ct, pt]); d[f"{tag}_counts"] = np.array(counts)
    # an all-below-threshold image (returns python lists)
    low = (out[0] * 0.01).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        r = ns["decode"](torch.from_numpy(low), torch.from_numpy(tgt[0]), 0.2, 0.1)
    assert r[0] == [] and r[1] == []
    d["low_output"] = low
    np.savez_compressed(os.path.join(OUT, "decode_v1.npz"), **d)

    # evaluate (yolo.py:273-299): body executed as text with the model replaced by a stub
    num_classes = 5
    out64, tgt64 = wl.yolo1_batch(1, batch=32, one_slot=True)   # tie-free NMS keys

    class _Stub:
        def eval(self):
            return self

        def __call__(self, image):
            return torch.from_numpy(out64)

    ns.update(fortest=_Stub(), num_classes=num_classes,
              detects={i: [] for i in range(num_classes)}, targets={i: [] for i in range(num_classes)})
    extract_lines("yolo/yolo_v1/yolo.py", 273, 299, ns)
    with contextlib.redirect_stdout(io.StringIO()):
        detects, targets = ns["evaluate"](None, torch.from_numpy(tgt64), prob_threshold=0.01, nms_threshold=0.3)
    e = {"output": out64, "target": tgt64}
    for c in range(num_classes):
        e[f"detects_{c}"] = np.array(detects[c], np.float64).reshape(-1, 6)
        e[f"targets_{c}"] = np.array(targets[c], np.float64).reshape(-1, 6)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        aps = ns["ap"](detects, targets, 0.05)
    e["aps_iou005"] = np.array(aps, np.float64)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        aps = ns["ap"](detects, targets, 0.5)
    e["aps_iou05"] = np.array(aps, np.float64)
    np.savez_compressed(os.path.join(OUT, "evaluate_v1.npz"), **e)


# ------------------------------------------------------------------ RoI pool
def gen_roipool():
    ns = extract_lines("rcnn/fast_rcnn.py", 351, 377, base_ns())
    Pool = ns["slowroipool"]
    rng = np.random.default_rng(6)
    B, C, H, W = 2, 8, 38, 50
    img = wl.frcnn_feature_map(3, B, C, H, W)
    rois = wl.frcnn_rois(3, 96)
    rois[0] = [0.0, 0.0, 1.0, 1.0]            # whole map
    rois[1] = [0.5, 0.5, 0.5001, 0.5001]      # 1x1 crop
    rois[2] = [0.1, 0.1, 0.1 + 3 / 50, 0.1 + 2 / 38 - 1e-9]  # tiny crop (bins repeat)
    rois[3] = [0.02, 0.0, 0.98, 0.05]         # wide & flat
    idx = rng.integers(0, B, 96).astype(np.int64)
    timg = torch.from_numpy(img).requires_grad_(True)
    gout = rng.standard_normal((96, C, 7, 7)).astype(np.float32)

    # as written: AdaptiveAvgPool2d
    m = Pool((7, 7))
    y = m.foward(timg, rois, idx)
    y.backward(torch.from_numpy(gout))
    d = dict(img=img, rois=rois, idx=idx, grad_out=gout, avg_out=y.detach().numpy(),
             avg_grad_in=timg.grad.numpy().copy())
    timg.grad = None
    # north_star variant: same class with AdaptiveMaxPool2d(return_indices=True) substituted
    m2 = Pool((7, 7))
    m2.maxpool = nn.AdaptiveMaxPool2d((7, 7))
    y2 = m2.foward(timg, rois, idx)
    y2.backward(torch.from_numpy(gout))
    d.update(max_out=y2.detach().numpy(), max_grad_in=timg.grad.numpy().copy())
    # argmax from torch (index inside the crop) converted to feature-map flat index y*W+x
    x1 = np.floor(rois[:, 0] * W).astype(int); x2 = np.ceil(rois[:, 2] * W).astype(int)
    y1 = np.floor(rois[:, 1] * H).astype(int); y2_ = np.ceil(rois[:, 3] * H).astype(int)
    arg = np.zeros((96, C, 7, 7), np.int32)
    pool_i = nn.AdaptiveMaxPool2d((7, 7), return_indices=True)
    for i in range(96):
        crop = timg.detach()[idx[i], :, y1[i]:y2_[i], x1[i]:x2[i]].unsqueeze(0)
        v, ind = pool_i(crop)
        assert torch.equal(v[0], y2.detach()[i])
        cw = x2[i] - x1[i]
        ind = ind[0].numpy()
        arg[i] = ((ind // cw) + y1[i]) * W + (ind % cw) + x1[i]
    d["max_argmax"] = arg
    # empty crop -> RuntimeError in the reference
    bad = rois[:2].copy(); bad[1] = [0.5, 0.5, 0.5, 0.5]
    try:
        m2.foward(timg, bad, idx[:2]); raised = 0
    except RuntimeError:
        raised = 1
    d["empty_rois"] = bad; d["empty_raises"] = np.int64(raised)
    np.savez_compressed(os.path.join(OUT, "roipool.npz"), **d)


def gen_roipool_sizes():
    """slowroipool(output_size) for output sizes other than (7,7): the reference passes output_size to the adaptive pool
    (rcnn/fast_rcnn.py:352-355), so any int or pair is legal.  As written (avg) and with AdaptiveMaxPool2d substituted,
    forward + autograd gradients, on a small map with crops from 1 cell up to the whole map."""
    ns = extract_lines("rcnn/fast_rcnn.py", 351, 377, base_ns())
    Pool = ns["slowroipool"]
    rng = np.random.default_rng(16)
    B, C, H, W = 2, 5, 23, 31
    img = rng.standard_normal((B, C, H, W)).astype(np.float32)
    img[0, 0, 3:6, 4:9] = 1.5                       # ties inside bins
    n = 40
    xy = rng.uniform(0, 0.8, (n, 2)); wh = rng.uniform(0.02, 0.9, (n, 2))
    rois = np.clip(np.concatenate([xy, xy + wh], 1), 0, 1)
    rois[0] = [0, 0, 1, 1]
    rois[1] = [0.5, 0.5, 0.5001, 0.5001]
    idx = rng.integers(0, B, n).astype(np.int64)
    d = dict(img=img, rois=rois, idx=idx)
    x1 = np.floor(rois[:, 0] * W).astype(int); x2 = np.ceil(rois[:, 2] * W).astype(int)
    y1 = np.floor(rois[:, 1] * H).astype(int); y2_ = np.ceil(rois[:, 3] * H).astype(int)
    for tag, size in {"s3x5": (3, 5), "s14": 14, "s1": 1, "s2x9": (2, 9)}.items():
        PH, PW = size if isinstance(size, tuple) else (size, size)
        gout = rng.standard_normal((n, C, PH, PW)).astype(np.float32)
        timg = torch.from_numpy(img).requires_grad_(True)
        m = Pool(size)
        y = m.foward(timg, rois, idx)
        y.backward(torch.from_numpy(gout))
        d[f"{tag}_size"] = np.array([PH, PW]); d[f"{tag}_grad_out"] = gout
        d[f"{tag}_avg_out"] = y.detach().numpy(); d[f"{tag}_avg_grad_in"] = timg.grad.numpy().copy()
        timg.grad = None
        m2 = Pool(size)
        m2.maxpool = nn.AdaptiveMaxPool2d(size)
        y2 = m2.foward(timg, rois, idx)
        y2.backward(torch.from_numpy(gout))
        d[f"{tag}_max_out"] = y2.detach().numpy(); d[f"{tag}_max_grad_in"] = timg.grad.numpy().copy()
        arg = np.zeros((n, C, PH, PW), np.int32)
        pool_i = nn.AdaptiveMaxPool2d(size, return_indices=True)
        for i in range(n):
            crop = timg.detach()[idx[i], :, y1[i]:y2_[i], x1[i]:x2[i]].unsqueeze(0)
            v, ind = pool_i(crop)
            assert torch.equal(v[0], y2.detach()[i])
            cw = x2[i] - x1[i]
            ind = ind[0].numpy()
            arg[i] = ((ind // cw) + y1[i]) * W + (ind % cw) + x1[i]
        d[f"{tag}_max_argmax"] = arg
    np.savez_compressed(os.path.join(OUT, "roipool_sizes.npz"), **d)


# ------------------------------------------------------------------ YOLOv2 get_point, R-CNN calcul/decode, do_ss
def gen_misc():
    ns = extract_defs("yolo/yolo_v2/utils.py", {"get_point"}, base_ns())
    rng = np.random.default_rng(8)
    bbs = np.concatenate([rng.uniform(0, 0.6, (64, 2)), rng.uniform(0.02, 0.39, (64, 2))], 1).astype(np.float32)
    pts, enc = [], []
    for bb in bbs:
        p, e = ns["get_point"](56, torch.from_numpy(bb))
        pts.append(p.numpy()); enc.append(e.numpy())
    np.savez(os.path.join(OUT, "yolo2_get_point.npz"), bbox=bbs, points=np.array(pts), enc=np.array(enc))

    ns = extract_defs("rcnn/utils.py", {"calcul", "get_iou"}, base_ns())
    pos = np.stack([rng.integers(0, 200, 40), rng.integers(0, 200, 40)], 1)
    pos = np.concatenate([pos, pos + rng.integers(8, 150, (40, 2))], 1).astype(np.float32)
    grt = np.array([60.0, 40.0, 210.0, 190.0], np.float32)
    t = ns["calcul"](torch.from_numpy(pos), torch.from_numpy(grt)).numpy()
    # rcnn/rcnn.py:287-310 (inline box decode), executed as text
    tset = torch.from_numpy(rng.normal(0, 0.3, (5, 8, 4)).astype(np.float32))
    p_rect = torch.from_numpy(pos.reshape(5, 8, 4).astype(np.int32))
    ns2 = base_ns(); ns2.update(test={"p_rect": p_rect}, tset=tset)
    extract_lines("rcnn/rcnn.py", 287, 310, ns2)
    g = ns2["test_g_rect"].numpy()
    # rcnn/fast_rcnn.py:543-560 variant (centre = (x1+x2)/2)
    ns3 = base_ns(); ns3.update(test={"p_rect": p_rect})

    class _C:
        def __len__(self):
            return 5
    ns3.update(c=_C(), tset=torch.cat([tset, tset], 1).reshape(5 * 16 * 4))
    extract_lines("rcnn/fast_rcnn.py", 543, 560, ns3)
    g2 = ns3["test_g_rect"].numpy()
    np.savez(os.path.join(OUT, "rcnn_boxreg.npz"), positive=pos, grt=grt, t_target=t,
             tset=tset.numpy(), p_rect=p_rect.numpy(), g_rect_rcnn=g, g_rect_frcnn=g2)

    # do_ss quota loop, rcnn/rcnn.py:27-65, body as text (selective search itself is cv2: not reproduced)
    def run_do_ss(rel, a, b, rects, bbox, label, min_num):
        body = _read(rel).splitlines()[a - 1:b]
        code = "def _quota(rects, dict, min_num):\n" + "\n".join(body)
        nsx = base_ns(); nsx["get_iou"] = ns["get_iou"]; nsx["img_rgb"] = np.zeros((1,))
        exec(compile(code, f"{rel}:{a}-{b}", "exec"), nsx)
        return nsx["_quota"](rects, {"bbox": list(bbox), "labels": np.array([label])}, min_num)

    rects = np.stack([rng.integers(0, 300, 400), rng.integers(0, 200, 400),
                      rng.integers(5, 200, 400), rng.integers(5, 200, 400)], 1).astype(np.int32)
    bbox = [100.0, 50.0, 160.0, 130.0]      # xywh
    r1 = run_do_ss("rcnn/rcnn.py", 27, 65, rects, bbox, 3, 5)
    r2 = run_do_ss("rcnn/fast_rcnn.py", 115, 155, rects, bbox, 3, 5)
    np.savez(os.path.join(OUT, "do_ss_quota.npz"), rects=rects, bbox=np.array(bbox), label=np.int64(3),
             min_num=np.int64(5),
             rcnn_rois=r1[0], rcnn_grt=r1[1], rcnn_labels=r1[2],
             frcnn_rois=r2[1], frcnn_grt=r2[2], frcnn_labels=r2[3])
    # a run that never fills the positive quota returns four empty lists (fast_rcnn.py:153-154)
    r3 = run_do_ss("rcnn/fast_rcnn.py", 115, 155, rects[:30], bbox, 3, 50)
    assert r3 == ([], [], [], [])


# ------------------------------------------------------------------ clustering demo (API shape only)
def gen_cluster():
    try:
        import sklearn  # noqa: F401
    except Exception as e:  # pragma: no cover
        print("sklearn missing, skipping cluster fixture:", e); return
    lines = _read("clustering/cluster_prac.py").splitlines()
    code = "\n".join(lines[10:41])
    ns = {}
    import pandas as pd
    from sklearn.preprocessing import StandardScaler as sts
    from sklearn.cluster import KMeans as kmean
    import random
    ns.update(pd=pd, np=np, sts=sts, kmean=kmean, random=random, os=os)
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(code, "cluster_prac.py:11-41", "exec"), ns)
    km = ns["model_km"]
    X = ns["dset_sts"][["연봉", "연차"]].values
    np.savez(os.path.join(OUT, "cluster_demo.npz"), X=X, labels=np.asarray(ns["dset_sts"]["kmean"].values),
             centers=km.cluster_centers_, n_iter=np.int64(km.n_iter_),
             sklearn_version=np.array(sklearn.__version__))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    which = sys.argv[1:] or ["iou", "nms", "decode", "roipool", "roipool_sizes", "misc", "cluster", "loss", "encode"]
    for w in which:
        print("generating", w, flush=True)
        globals()["gen_" + w]()
    print("done ->", OUT)
