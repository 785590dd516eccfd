"""ctypes/NumPy front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by vision_amd/.
Every function takes and returns NumPy arrays on the host.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORACLE_SO") or os.path.join(_HERE, "liboracle.so")      # ORACLE_SO: e.g. the sanitizer build (make asan)

OK, EINVAL, EEMPTYROI, EASSERT = 0, -1, -2, -3


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if os.environ.get("ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_yolo1_decode.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


# ------------------------------------------------------------------ IoU
def iou_elementwise(bb1, bb2, eps=1e-7):
    if np.asarray(bb1).dtype == np.float64:
        a, b = np.ascontiguousarray(bb1, np.float64), np.ascontiguousarray(bb2, np.float64)
        out = np.empty(a.shape[0], np.float64)
        lib().orc_iou_elementwise_f64(_p(a), _p(b), C.c_int64(a.shape[0]), C.c_double(eps), _p(out))
        return out
    a, b = _f32(bb1), _f32(bb2)
    out = np.empty(a.shape[0], np.float32)
    lib().orc_iou_elementwise_f32(_p(a), _p(b), C.c_int64(a.shape[0]), C.c_float(eps), _p(out))
    return out


def iou_pairwise(a, b, eps=1e-7):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().orc_iou_pairwise_f32(_p(a), C.c_int64(a.shape[0]), _p(b), C.c_int64(b.shape[0]), C.c_float(eps), _p(out))
    return out


def iou_scalar(bb1, bb2):
    a = np.ascontiguousarray(bb1, np.float64)
    b = np.ascontiguousarray(bb2, np.float64)
    out = C.c_double(0)
    st = lib().orc_iou_scalar_f64(_p(a), _p(b), C.byref(out))
    if st == EASSERT:
        raise AssertionError("get_iou: degenerate box or iou out of [0,1]")
    return out.value


# ------------------------------------------------------------------ NMS
def nms(boxes, scores, threshold=0.1, quirk_drop_last=True):
    b, s = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    n = b.shape[0]
    keep = np.empty(max(n, 1), np.int64)
    cnt = C.c_int64(0)
    st = lib().orc_nms_f32(_p(b), _p(s), C.c_int64(n), C.c_double(threshold), C.c_int(int(quirk_drop_last)),
                           _p(keep), C.byref(cnt))
    assert st == OK
    return keep[:cnt.value].copy()


def nms_frcnn_top(rects, scores):
    """rcnn/fast_rcnn.py:590-610: argsort(scores)[::-1], no suppression, returns element 0."""
    rects = np.array(rects)
    scores = np.array(scores)
    idx = np.argsort(scores)[::-1]
    return rects[idx][0], scores[idx][0]


# ------------------------------------------------------------------ YOLOv1 decode
def yolo1_decode(output, target=None, conf_threshold=0.2, prob_threshold=0.1):
    o = _f32(output)
    Cn = o.shape[-1] - 10
    t = _f32(target) if target is not None else None
    box = np.empty((98, 4), np.float32); lab = np.empty(98, np.int64)
    conf = np.empty(98, np.float32); score = np.empty(98, np.float32)
    tb = np.zeros(4, np.float64); tl = C.c_int64(0)
    m = lib().orc_yolo1_decode(_p(o), _p(t) if t is not None else None, C.c_int64(Cn),
                               C.c_double(conf_threshold), C.c_double(prob_threshold),
                               _p(box), _p(lab), _p(conf), _p(score), _p(tb), C.byref(tl))
    return box[:m].copy(), lab[:m].copy(), conf[:m].copy(), score[:m].copy(), tb, tl.value


# ------------------------------------------------------------------ RoI pool
def _pool_size(P):
    return (int(P[0]), int(P[1])) if isinstance(P, (tuple, list)) else (int(P), int(P))


def roipool_fwd(img, rois, roi_idx, P=7, mode="max"):
    img = _f32(img)
    B, Cc, H, W = img.shape
    PH, PW = _pool_size(P)
    rois = np.ascontiguousarray(rois, np.float64).reshape(-1, 4)
    idx = np.ascontiguousarray(roi_idx, np.int64)
    n = rois.shape[0]
    out = np.empty((n, Cc, PH, PW), np.float32)
    arg = np.empty((n, Cc, PH, PW), np.int32) if mode == "max" else None
    st = lib().orc_roipool_fwd2_f32(_p(img), C.c_int64(B), C.c_int64(Cc), C.c_int(H), C.c_int(W), _p(rois), _p(idx),
                                    C.c_int64(n), C.c_int(PH), C.c_int(PW), C.c_int(1 if mode == "max" else 0), _p(out),
                                    _p(arg) if arg is not None else None)
    if st == EEMPTYROI:
        raise RuntimeError("empty RoI crop")
    assert st == OK, st
    return out, arg


def roipool_bwd(grad_out, argmax, img_shape, rois, roi_idx, P=7, mode="max"):
    B, Cc, H, W = img_shape
    PH, PW = _pool_size(P)
    g = _f32(grad_out)
    rois = np.ascontiguousarray(rois, np.float64).reshape(-1, 4)
    idx = np.ascontiguousarray(roi_idx, np.int64)
    gin = np.zeros((B, Cc, H, W), np.float32)
    a = np.ascontiguousarray(argmax, np.int32) if argmax is not None else None
    st = lib().orc_roipool_bwd2_f32(_p(g), _p(a) if a is not None else None, C.c_int64(B), C.c_int64(Cc), C.c_int(H),
                                    C.c_int(W), _p(rois), _p(idx), C.c_int64(rois.shape[0]), C.c_int(PH), C.c_int(PW),
                                    C.c_int(1 if mode == "max" else 0), _p(gin))
    assert st == OK, st
    return gin


# ------------------------------------------------------------------ YOLOv2
def yolo2_get_point(bbox):
    b = _f32(bbox).reshape(-1, 4)
    pts = np.empty((b.shape[0], 2), np.int64); enc = np.empty((b.shape[0], 4), np.float32)
    lib().orc_yolo2_get_point(_p(b), C.c_int64(b.shape[0]), _p(pts), _p(enc))
    return pts, enc


def yolo2_decode_cells(points, enc):
    p = np.ascontiguousarray(points, np.int64).reshape(-1, 2); e = _f32(enc).reshape(-1, 4)
    out = np.empty((p.shape[0], 4), np.float32)
    lib().orc_yolo2_decode_cells(_p(p), _p(e), C.c_int64(p.shape[0]), _p(out))
    return out


def yolo2_decode_anchor(pred, anchors):
    p = _f32(pred); G, A = p.shape[0], p.shape[2]; Cn = p.shape[3] - 5
    an = _f32(anchors)
    xyxy = np.empty((G * G * A, 4), np.float32); conf = np.empty(G * G * A, np.float32)
    lib().orc_yolo2_decode_anchor(_p(p), C.c_int64(G), C.c_int64(A), C.c_int64(Cn), _p(an), _p(xyxy), _p(conf))
    return xyxy, conf


# ------------------------------------------------------------------ R-CNN box regression
def rcnn_calcul(positive, grt):
    p = _f32(positive).reshape(-1, 4); g = _f32(grt)
    t = np.empty_like(p)
    lib().orc_rcnn_calcul_f32(_p(p), C.c_int64(p.shape[0]), _p(g), _p(t))
    return t


def rcnn_decode(p_rect, tset, variant=0):
    p = np.ascontiguousarray(p_rect, np.int32).reshape(-1, 4); t = _f32(tset).reshape(-1, 4)
    g = np.empty((p.shape[0], 4), np.float32)
    lib().orc_rcnn_decode_f32(_p(p), _p(t), C.c_int64(p.shape[0]), C.c_int(variant), _p(g))
    return g


def yolo1_loss(modely, targety, quirk=True, want_grad=True):
    """modules.py:234-306 -> (loss fp32, comp[5] float64 = xy, wh, obj, noobj, label sums, grad or None)"""
    y = _f32(modely); t = _f32(targety)
    n, D = y.shape[0], y.shape[-1]
    comp = np.zeros(5, np.float64); loss = np.zeros(1, np.float32)
    grad = np.empty_like(y) if want_grad else None
    lib().orc_yolo1_loss(_p(y), _p(t), C.c_int64(n), C.c_int(D - 10), C.c_int(int(quirk)), _p(comp), _p(loss),
                         _p(grad) if want_grad else None)
    return loss[0], comp, grad


# ------------------------------------------------------------------ proposal labelling (host logic)
def label_proposals(rects_xywh, bbox_xywh, label, min_num, pos_thr=0.1, neg_thr=0.1, neg_quota_mult=2,
                    require_full=False):
    """rcnn/rcnn.py:27-65 (pos_thr=neg_thr=0.1, negatives 2*min_num) and
    rcnn/fast_rcnn.py:115-155 (pos 0.5 / neg 0.1, negatives min_num, returns empties unless both
    quotas were seen full). Literal loop."""
    grt = np.array(bbox_xywh, np.float64)
    grt[2] += grt[0]; grt[3] += grt[1]
    count_t = count_f = 0
    fflag = bflag = 0
    rois, labels = [], []
    for box in rects_xywh:
        roi = [int(v) for v in box]
        roi[2] += roi[0]; roi[3] += roi[1]
        iou = iou_scalar(roi, grt)
        if count_t < min_num:
            if iou >= pos_thr:
                rois.append(roi); labels.append(int(label)); count_t += 1
        else:
            fflag = 1
        if count_f < min_num * neg_quota_mult:
            if iou < neg_thr:
                rois.append(roi); labels.append(0); count_f += 1
        else:
            bflag = 1
        if fflag == 1 and bflag == 1:
            break
    if require_full and (fflag == 0 or bflag == 0):
        return np.zeros((0, 4), np.int64), grt, np.zeros((0,), np.int64)
    return np.array(rois, np.int64).reshape(-1, 4), grt, np.array(labels, np.int64)


# ------------------------------------------------------------------ evaluate / AP (host logic)
def evaluate_batch(output, target, num_classes, prob_threshold=0.005, nms_threshold=0.3):
    """yolo/yolo_v1/yolo.py:273-299 after the model call: decode -> nms -> per-class rows."""
    detects = {i: [] for i in range(num_classes)}
    targets = {i: [] for i in range(num_classes)}
    for i in range(len(output)):
        box, lab, conf, score, tb, tl = yolo1_decode(output[i], target[i], 0.2, prob_threshold)
        targets[tl - 1].append([i, 1] + tb.tolist())
        keep = nms(box, score, nms_threshold)
        if len(keep) > 0:
            box, lab, conf = box[keep], lab[keep], conf[keep]
            for c in range(num_classes):
                sel = lab == c
                for cf, bx in zip(conf[sel], box[sel]):
                    detects[c].append([i, float(cf)] + [float(v) for v in bx])
    return detects, targets


def average_precision(detects, targets, iou_thr=0.01):
    """yolo/yolo_v1/yolo.py:306-367 (VOC-style AP; note :314-315 appends 0.0 AND continues
    computing for an empty class, so an empty class contributes two entries)."""
    aps = []
    for c in targets.keys():
        if len(targets[c]) == 0:
            aps.append(0.0)
        n_grt = len(targets[c])
        grts = {i[0]: [i[2:]] for i in targets[c]}
        hit = {k: np.zeros(1) for k in grts}
        det = sorted(list(detects[c]), key=lambda x: x[1], reverse=True)
        TP = np.zeros(len(det)); FP = np.zeros(len(det))
        for d in range(len(det)):
            img = det[d][0]
            grt = grts[img] if img in grts else []
            iou_max = -float("inf"); gmax = 0
            for e, g in enumerate(grt):
                iou = iou_elementwise(np.array([det[d][2:]], np.float32), np.array([g], np.float32))[0]
                if iou > iou_max:
                    iou_max = iou; gmax = e
            if iou_max >= iou_thr:
                if hit[img][gmax] == 0:
                    TP[d] = 1; hit[img][gmax] = 1
                else:
                    FP[d] = 1
            else:
                FP[d] = 1
        tp, fp = np.cumsum(TP), np.cumsum(FP)
        with np.errstate(all="ignore"):
            recall = tp / n_grt
            prec = np.divide(tp, tp + fp)
        recall = np.concatenate(([0.0], recall, [1.0]))
        prec = np.concatenate(([0.0], prec, [0.0]))
        for i in range(prec.size - 1, 0, -1):
            prec[i - 1] = max(prec[i - 1], prec[i])
        ap = 0.0
        for i in range(prec.size - 1):
            ap += (recall[i + 1] - recall[i]) * prec[i + 1]
        aps.append(ap)
    return aps


# ------------------------------------------------------------------ YOLOv1 target builder
def yolo1_encode(bb, label, num_classes, S=7, B=2):
    """encode(bb, label, num_classes): yolo/yolo_v1/modules.py:151-174, float64 like the reference (NumPy restatement;
    pinned by tests/golden/yolo1_encode.npz, which the reference's own encode produced)."""
    bb = np.asarray(bb, np.float64).copy()
    N = 5 * B + num_classes
    bb[:2] += bb[2:] / 2                                         # :156 centre
    t = np.zeros((S, S, N))
    cxy, wh = bb[:2], bb[2:]
    ij = (cxy * S).astype(int)                                   # :160
    i, j = ij
    xy_cell = (cxy - ij * (1 / S)) / (1 / S)                     # :163-164
    for b in range(B):
        t[j, i, b * 5:(b + 1) * 5] = np.hstack([xy_cell, wh, 1])
    t[j, i, B * 5 + label - 1] = 1
    return t


# ------------------------------------------------------------------ anchor k-means (spec in oracle.c)
def kmeans_init(wh, k, random_state):
    perm = np.random.RandomState(random_state).permutation(wh.shape[0])
    return np.ascontiguousarray(wh[perm[:k]], np.float32).copy()


def kmeans_step(wh, cent, labels, acc=None):
    wh = _f32(wh); cent = _f32(cent)
    k = cent.shape[0]
    if acc is None:
        acc = np.zeros(3 * k + 2, np.int64)
    lib().orc_kmeans_step(_p(wh), C.c_int64(wh.shape[0]), _p(cent), C.c_int(k), _p(labels), _p(acc))
    return acc


def kmeans_update(acc, cent):
    cent = _f32(cent).copy()
    lib().orc_kmeans_update(_p(acc), C.c_int(cent.shape[0]), _p(cent))
    return cent


def kmeans_fit(wh, k=5, random_state=42, max_iter=300, fixed_iters=None):
    """Full Lloyd loop. Returns centers (k,2) f32, labels int32, n_iter, inertia (float)."""
    wh = _f32(wh)
    cent = kmeans_init(wh, k, random_state)
    labels = np.full(wh.shape[0], 255, np.uint8)
    n_iter = 0
    acc = None
    iters = fixed_iters if fixed_iters is not None else max_iter
    for it in range(iters):
        acc = kmeans_step(wh, cent, labels)
        n_iter = it + 1
        if fixed_iters is None and acc[3 * k] == 0:
            break
        cent = kmeans_update(acc, cent)
    inertia = float(acc[3 * k + 1]) / 4294967296.0 if acc is not None else 0.0
    return cent, labels.astype(np.int32), n_iter, inertia
