"""Tensor-level front end of the HIP kernels: GPU tensors in, GPU tensors out, async on the
current stream.  The reference-named drop-ins (yolo_v1.py, yolo_v2.py, rcnn_utils.py,
fast_rcnn.py, cluster.py) are thin layers over these."""
import torch

from . import _abi
from ._abi import check, load, ptr, stream_ptr, workspace


def _f32c(t, dev):
    return _abi.to_dev(t, torch.float32, dev)


# ------------------------------------------------------------------ IoU
def iou_elementwise(bb1, bb2, eps=1e-7):
    dev = _abi.device_of(bb1, bb2)
    f64 = isinstance(bb1, torch.Tensor) and bb1.dtype == torch.float64
    dt = torch.float64 if f64 else torch.float32
    a, b = _abi.to_dev(bb1, dt, dev).reshape(-1, 4), _abi.to_dev(bb2, dt, dev).reshape(-1, 4)
    if a.shape != b.shape:
        raise ValueError(f"get_iou: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    out = torch.empty(a.shape[0], dtype=dt, device=dev)
    fn = load().dg_iou_elementwise_f64 if f64 else load().dg_iou_elementwise_f32
    with torch.cuda.device(dev):
        check(fn(ptr(a), ptr(b), a.shape[0], eps, ptr(out), stream_ptr(dev)))
    return out


def iou_pairwise(a, b, eps=1e-7, out=None):
    dev = _abi.device_of(a, b)
    a, b = _f32c(a, dev).reshape(-1, 4), _f32c(b, dev).reshape(-1, 4)
    if out is None:
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_iou_pairwise_f32(ptr(a), a.shape[0], ptr(b), b.shape[0], eps, ptr(out), stream_ptr(dev)))
    return out


def iou_rois_vs_gt(rois, grt):
    """float64 IoU of n xyxy rois against one box + per-roi assert flags (rcnn/utils.py:50-67)."""
    dev = _abi.device_of(rois, grt)
    r = _abi.to_dev(rois, torch.float64, dev).reshape(-1, 4)
    g = _abi.to_dev(grt, torch.float64, dev).reshape(4)
    out = torch.empty(r.shape[0], dtype=torch.float64, device=dev)
    flags = torch.empty(r.shape[0], dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_iou_rois_vs_gt_f64(ptr(r), ptr(g), r.shape[0], ptr(out), ptr(flags), stream_ptr(dev)))
    return out, flags


# ------------------------------------------------------------------ NMS
def nms_workspace(dev, batch, nmax):
    nbytes = load().dg_nms_workspace_bytes(batch, nmax)
    return workspace(dev, nbytes), nbytes


def nms_path_info(dev, batch, nmax, img=0):
    """(mode, fpdone, rounds, npairs, cand_per_box, est_pairs_per_box) of image `img` in the LAST NMS call that used this
    stream's workspace: mode 0 = pruned pair search, 1 = dense mask; fpdone 1 = the fixed-point kernel produced the keep list.
    Synchronous (diagnostic)."""
    import ctypes as C
    ws, nbytes = nms_workspace(dev, batch, nmax)
    out = (C.c_int32 * 6)()
    with torch.cuda.device(dev):
        check(load().dg_nms_path_info(ptr(ws), nbytes, batch, nmax, img, out))
    return tuple(out)


def nms_batched(boxes, scores, counts=None, threshold=0.1, quirk_drop_last=True, keep=None, keep_count=None):
    """boxes (B,N,4) fp32, scores (B,N), counts (B) int32 or None -> keep (B,N) int64, keep_count (B) int32."""
    dev = _abi.device_of(boxes, scores)
    boxes, scores = _f32c(boxes, dev), _f32c(scores, dev)
    B, N = scores.shape
    if boxes.shape != (B, N, 4):
        raise ValueError("nms_batched: boxes must be (B,N,4) matching scores (B,N)")
    if counts is not None:
        counts = _abi.to_dev(counts, torch.int32, dev)
    if keep is None:
        keep = torch.empty((B, N), dtype=torch.int64, device=dev)
    if keep_count is None:
        keep_count = torch.empty((B,), dtype=torch.int32, device=dev)
    ws, nbytes = nms_workspace(dev, B, N)
    with torch.cuda.device(dev):
        check(load().dg_nms_batched_f32(ptr(boxes), ptr(scores), ptr(counts), B, N, float(threshold), int(quirk_drop_last),
                                        ptr(keep), ptr(keep_count), ptr(ws), nbytes, stream_ptr(dev)))
    return keep, keep_count


def nms_batched_per_class(boxes, scores, labels, num_classes, counts=None, threshold=0.1, quirk_drop_last=True):
    """Per-class NMS (no reference counterpart: its nms is class-agnostic, yolo.py:188-203): the reference's nms applied to
    each class's boxes of an image separately.  labels (B,N) int64 in [0,num_classes).
    -> keep (B,N) int64 (classes' keep lists concatenated in class order), keep_count (B), class_count (B,num_classes)."""
    dev = _abi.device_of(boxes, scores)
    boxes, scores = _f32c(boxes, dev), _f32c(scores, dev)
    labels = _abi.to_dev(labels, torch.int64, dev)
    B, N = scores.shape
    if boxes.shape != (B, N, 4) or labels.shape != (B, N):
        raise ValueError("nms_batched_per_class: boxes (B,N,4), scores (B,N), labels (B,N)")
    if counts is not None:
        counts = _abi.to_dev(counts, torch.int32, dev)
    keep = torch.empty((B, N), dtype=torch.int64, device=dev)
    keep_count = torch.empty((B,), dtype=torch.int32, device=dev)
    class_count = torch.empty((B, int(num_classes)), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    nbytes = load().dg_nms_per_class_workspace_bytes(B, N, int(num_classes))
    ws = workspace(dev, nbytes)
    with torch.cuda.device(dev):
        check(load().dg_nms_batched_per_class_f32(ptr(boxes), ptr(scores), ptr(labels), ptr(counts), B, N, int(num_classes),
                                                  float(threshold), int(quirk_drop_last), ptr(keep), ptr(keep_count),
                                                  ptr(class_count), ptr(status), ptr(ws), nbytes, stream_ptr(dev)))
    return keep, keep_count, class_count, status


# ------------------------------------------------------------------ YOLOv1 decode
def yolo1_decode_batched(output, target=None, conf_threshold=0.2, prob_threshold=0.1, nms_threshold=None, per_class=False,
                         quirk_drop_last=True):
    """decode of a batch; with nms_threshold the NMS runs in the same launch (dg_yolo1_decode_nms_f32) and the result
    gains (keep (B,98) int64, keep_count (B) int32, class_count (B,C) int32)."""
    dev = _abi.device_of(output, target)
    o = _f32c(output, dev)
    if o.dim() == 3:
        o = o.unsqueeze(0)
    B = o.shape[0]
    if tuple(o.shape[1:3]) != (7, 7) or o.shape[3] < 11:
        raise ValueError("decode: output must be (...,7,7,10+C)")
    C_ = o.shape[3] - 10
    t = None
    if target is not None:
        t = _f32c(target, dev).reshape(o.shape)
    box = torch.empty((B, 98, 4), dtype=torch.float32, device=dev)
    label = torch.empty((B, 98), dtype=torch.int64, device=dev)
    conf = torch.empty((B, 98), dtype=torch.float32, device=dev)
    score = torch.empty((B, 98), dtype=torch.float32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    tbox = torch.empty((B, 4), dtype=torch.float64, device=dev) if t is not None else None
    tlabel = torch.empty((B,), dtype=torch.int64, device=dev) if t is not None else None
    if nms_threshold is not None:
        keep = torch.empty((B, 98), dtype=torch.int64, device=dev)
        kcnt = torch.empty((B,), dtype=torch.int32, device=dev)
        ccnt = torch.empty((B, C_), dtype=torch.int32, device=dev) if per_class else None
        with torch.cuda.device(dev):
            check(load().dg_yolo1_decode_nms_f32(ptr(o), ptr(t), B, C_, float(conf_threshold), float(prob_threshold),
                                                 float(nms_threshold), int(bool(per_class)), int(bool(quirk_drop_last)), ptr(box),
                                                 ptr(label), ptr(conf), ptr(score), ptr(count), ptr(keep), ptr(kcnt), ptr(ccnt),
                                                 ptr(tbox), ptr(tlabel), stream_ptr(dev)))
        return box, label, conf, score, count, tbox, tlabel, keep, kcnt, ccnt
    with torch.cuda.device(dev):
        check(load().dg_yolo1_decode_f32(ptr(o), ptr(t), B, C_, float(conf_threshold), float(prob_threshold), ptr(box), ptr(label),
                                         ptr(conf), ptr(score), ptr(count), ptr(tbox), ptr(tlabel), stream_ptr(dev)))
    return box, label, conf, score, count, tbox, tlabel


def yolo1_encode_batched(bb, label, num_classes):
    """bb (n,4) {left, top, w, h}, label (n) in 1..C -> target (n,7,7,10+C) float64 on the GPU (modules.py:151-174)."""
    dev = _abi.device_of(bb, label)
    b = _abi.to_dev(bb, torch.float64, dev).reshape(-1, 4)
    l = _abi.to_dev(label, torch.int64, dev).reshape(-1)
    if l.shape[0] != b.shape[0]:
        raise ValueError("encode: one label per box")
    n = b.shape[0]
    target = torch.empty((n, 7, 7, 10 + num_classes), dtype=torch.float64, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_yolo1_encode_f64(ptr(b), ptr(l), n, num_classes, ptr(target), ptr(status), stream_ptr(dev)))
    return target, status


# ------------------------------------------------------------------ YOLOv1 loss (fused geometry)
def yolo1_loss(modely, targety, quirk_slot2=True, want_grad=True):
    """(n,7,7,10+C) fp32 x2 -> (loss6 device tensor [loss, xy, wh, obj, noobj, label], grad or None)."""
    dev = _abi.device_of(modely, targety)
    y, t = _f32c(modely, dev), _f32c(targety, dev)
    if y.dim() != 4 or tuple(y.shape[1:3]) != (7, 7) or y.shape[3] < 10 or y.shape != t.shape:
        raise ValueError("Loss: modely/targety must both be (n,7,7,10+C)")
    n, C_ = y.shape[0], y.shape[3] - 10
    loss6 = torch.empty((6,), dtype=torch.float32, device=dev)
    grad = torch.empty_like(y) if want_grad else None
    nbytes = load().dg_yolo1_loss_workspace_bytes(n)
    ws = workspace(dev, nbytes)
    with torch.cuda.device(dev):
        check(load().dg_yolo1_loss_f32(ptr(y), ptr(t), n, C_, int(quirk_slot2), ptr(loss6), ptr(grad), ptr(ws), nbytes, stream_ptr(dev)))
    return loss6, grad


def voc_ap(det_img, det_cls, det_conf, det_box, tgt_img, tgt_cls, tgt_box, num_classes, num_images, iou_thr):
    """VOC-style AP of yolo.py:306-367 for every class in one launch.  Detections and targets are flat device (or host)
    arrays in the reference's list order; returns (ap float64 (C,), n_grt int32 (C,)) on the device.
    The ordering (class, then confidence descending, stable) is two stable torch sorts; the rest is k_voc_ap."""
    dev = _abi.device_of(det_box, tgt_box)
    C_, I = int(num_classes), int(num_images)
    dimg = _abi.to_dev(det_img, torch.int32, dev).reshape(-1)
    dcls = _abi.to_dev(det_cls, torch.int64, dev).reshape(-1)
    dconf = _abi.to_dev(det_conf, torch.float64, dev).reshape(-1)
    dbox = _f32c(det_box, dev).reshape(-1, 4)
    timg = _abi.to_dev(tgt_img, torch.int32, dev).reshape(-1)
    tcls = _abi.to_dev(tgt_cls, torch.int32, dev).reshape(-1)
    tbox = _f32c(tgt_box, dev).reshape(-1, 4)
    n = dimg.shape[0]
    if not (dcls.shape[0] == dconf.shape[0] == dbox.shape[0] == n) or not (timg.shape[0] == tcls.shape[0] == tbox.shape[0]):
        raise ValueError("ap: ragged detection / target arrays")
    if n:
        by_conf = torch.sort(dconf, descending=True, stable=True).indices
        by_cls = torch.sort(dcls[by_conf], stable=True).indices
        order = by_conf[by_cls].to(torch.int32)
        counts = torch.bincount(dcls.clamp(0, C_ - 1), minlength=C_)
    else:
        order = torch.empty(0, dtype=torch.int32, device=dev)
        counts = torch.zeros(C_, dtype=torch.int64, device=dev)
    cls_off = torch.zeros(C_ + 1, dtype=torch.int64, device=dev)
    cls_off[1:] = torch.cumsum(counts, 0)
    out = torch.empty(C_, dtype=torch.float64, device=dev)
    n_grt = torch.empty(C_, dtype=torch.int32, device=dev)
    nbytes = load().dg_voc_ap_workspace_bytes(n, C_, I)
    ws = workspace(dev, nbytes)
    with torch.cuda.device(dev):
        check(load().dg_voc_ap_f32(ptr(dimg), ptr(dbox), ptr(order), ptr(cls_off), n, ptr(timg), ptr(tcls), ptr(tbox),
                                   timg.shape[0], C_, I, float(iou_thr), ptr(out), ptr(n_grt), ptr(ws), nbytes, stream_ptr(dev)))
    return out, n_grt


# ------------------------------------------------------------------ RoI pooling
def roi_windows(rois, roi_idx, B, H, W, dev):
    r = _abi.to_dev(rois, torch.float64, dev).reshape(-1, 4)
    idx = _abi.to_dev(roi_idx, torch.int64, dev).reshape(-1)
    if idx.shape[0] != r.shape[0]:
        raise ValueError("roi_idx must have one entry per RoI")
    win = torch.empty((r.shape[0], 5), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_roi_windows(ptr(r), ptr(idx), r.shape[0], B, H, W, ptr(win), ptr(status), stream_ptr(dev)))
    return win, status


def to_nhwc(img, out=None, nan_flag=None):
    """(B,C,H,W) fp32 -> ((B,H,W,C) staging copy, nan_flag).  nan_flag is a device int32 that is non-zero iff
    the map holds a NaN (None for a tensor that already is channels_last: no copy, NaN state unknown)."""
    B, C_, H, W = img.shape
    if img.is_contiguous(memory_format=torch.channels_last) and not img.is_contiguous():
        return img.permute(0, 2, 3, 1), None       # same bytes, viewed NHWC
    img = img.contiguous()
    if out is None:
        out = torch.empty((B, H, W, C_), dtype=torch.float32, device=img.device)
    if nan_flag is None:
        nan_flag = torch.zeros((1,), dtype=torch.int32, device=img.device)
    else:
        nan_flag.zero_()
    with torch.cuda.device(img.device):
        check(load().dg_nchw_to_nhwc_f32(ptr(img), B, C_, H, W, ptr(out), ptr(nan_flag), stream_ptr(img.device)))
    return out, nan_flag


def _pool_size(P):
    if isinstance(P, (tuple, list)):
        if len(P) != 2:
            raise ValueError("output_size must be an int or a (PH, PW) pair")
        return int(P[0]), int(P[1])
    return int(P), int(P)


def roipool_forward(img, win, mode="max", P=7, out=None, argmax=None, nhwc=None, nan_flag=None, argmax_dtype=torch.int32):
    """P: output size, an int or (PH, PW) (rcnn/fast_rcnn.py:352-355 passes it to the adaptive pool).
    argmax: int32 (default) or, with argmax_dtype=torch.int16 / an int16 tensor passed in, the uint16 bit pattern of the
    same index (widen with `argmax_i32`): half the index bytes in the forward and the backward (7 x 7, H*W <= 65535 only)."""
    dev = img.device
    B, C_, H, W = img.shape
    PH, PW = _pool_size(P)
    n = win.shape[0]
    if nhwc is None:
        nhwc, nan_flag = to_nhwc(img)
    m = _abi.POOL_MAX if mode == "max" else _abi.POOL_AVG
    if out is None:
        out = torch.empty((n, C_, PH, PW), dtype=torch.float32, device=dev)
    if m == _abi.POOL_MAX and argmax is None:
        argmax = torch.empty((n, C_, PH, PW), dtype=argmax_dtype, device=dev)
    a16 = m == _abi.POOL_MAX and argmax.dtype == torch.int16
    if m == _abi.POOL_MAX and argmax.dtype not in (torch.int32, torch.int16):
        raise ValueError("roipool_forward: argmax must be int32 or int16 (uint16 bit pattern)")
    with torch.cuda.device(dev):
        if a16:
            if (PH, PW) != (7, 7):
                raise ValueError("roipool_forward: the uint16 argmax exists for 7 x 7 outputs only")
            check(load().dg_roipool_fwd_a16_f32(ptr(nhwc), B, C_, H, W, ptr(win), n, 7, m, ptr(nan_flag), ptr(out), ptr(argmax), stream_ptr(dev)))
        else:
            check(load().dg_roipool_fwd_any_f32(ptr(nhwc), B, C_, H, W, ptr(win), n, PH, PW, m, ptr(nan_flag), ptr(out),
                                                ptr(argmax) if m == _abi.POOL_MAX else None, stream_ptr(dev)))
    return out, (argmax if m == _abi.POOL_MAX else None)


def argmax_i32(argmax):
    """int32 view of an argmax tensor of either width (the 16-bit form holds the uint16 bit pattern)."""
    return argmax if argmax.dtype == torch.int32 else argmax.to(torch.int32) & 0xFFFF


def roipool_backward(grad_out, argmax, win, img_shape, mode="max", P=7, grad_in=None):
    dev = grad_out.device
    B, C_, H, W = img_shape
    PH, PW = _pool_size(P)
    n = win.shape[0]
    g = grad_out.contiguous()
    if grad_in is None:
        grad_in = torch.empty((B, C_, H, W), dtype=torch.float32, device=dev)
    nbytes = load().dg_roipool_bwd_workspace_bytes(B, n)
    ws = workspace(dev, nbytes)
    m = _abi.POOL_MAX if mode == "max" else _abi.POOL_AVG
    with torch.cuda.device(dev):
        if argmax is not None and argmax.dtype == torch.int16:
            check(load().dg_roipool_bwd_a16_f32(ptr(g), ptr(argmax), ptr(win), n, B, C_, H, W, PH, m, ptr(grad_in), ptr(ws), nbytes, stream_ptr(dev)))
        else:
            check(load().dg_roipool_bwd_any_f32(ptr(g), ptr(argmax), ptr(win), n, B, C_, H, W, PH, PW, m, ptr(grad_in), ptr(ws), nbytes,
                                                stream_ptr(dev)))
    return grad_in


# ------------------------------------------------------------------ k-means
def kmeans_step(wh, cent, labels, acc):
    dev = wh.device
    n, k = wh.shape[0], cent.shape[0]
    nbytes = load().dg_kmeans_workspace_bytes(n, k)
    ws = workspace(dev, nbytes)
    with torch.cuda.device(dev):
        check(load().dg_kmeans_iou_step(ptr(wh), n, ptr(cent), k, ptr(labels), ptr(acc), ptr(ws), nbytes, stream_ptr(dev)))
    return acc


def kmeans_run(wh, cent, labels, acc, history, slot0, iters):
    """iters Lloyd iterations on one device without returning to Python in between."""
    dev = wh.device
    n, k = wh.shape[0], cent.shape[0]
    nbytes = load().dg_kmeans_workspace_bytes(n, k)
    ws = workspace(dev, nbytes)
    with torch.cuda.device(dev):
        check(load().dg_kmeans_iou_run(ptr(wh), n, ptr(cent), k, ptr(labels), ptr(acc), ptr(history), slot0, iters, ptr(ws), nbytes,
                                       stream_ptr(dev)))


def kmeans_finish(part, k, cent=None, acc_out=None, history=None, slot=0):
    """fold a partial-sums buffer (all-reduced when sharded): acc_out <- sums, cent moved by them, history[slot] <- changed"""
    dev = part.device
    with torch.cuda.device(dev):
        check(load().dg_kmeans_finish(ptr(part), k, ptr(cent), ptr(acc_out), ptr(history), slot, stream_ptr(dev)))


def kmeans_labels_i32(labels_u8):
    out = torch.empty(labels_u8.shape, dtype=torch.int32, device=labels_u8.device)
    with torch.cuda.device(labels_u8.device):
        check(load().dg_kmeans_labels_i32(ptr(labels_u8), labels_u8.numel(), ptr(out), stream_ptr(labels_u8.device)))
    return out
