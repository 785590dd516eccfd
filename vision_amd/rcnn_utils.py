"""Drop-ins for rcnn/utils.py and the proposal-labelling loop of rcnn/rcnn.py / rcnn/fast_rcnn.py.

  get_iou(bb1, bb2)                   rcnn/utils.py:50-67 (scalar, float64, asserts, no eps)
  calcul(positive, grt)               rcnn/utils.py:153-171
  decode_boxes(p_rect, tset, variant) rcnn/rcnn.py:287-310 (0) / rcnn/fast_rcnn.py:543-560 (1)
  label_proposals(...)                quota loop of do_ss: rcnn/rcnn.py:27-65, rcnn/fast_rcnn.py:115-155
"""
import numpy as np
import torch

from . import _abi, ops
from ._abi import check, load, ptr, stream_ptr


def get_iou(bb1, bb2):
    """Box-vs-box IoU as a Python float; AssertionError where the reference asserts."""
    iou, flags = ops.iou_rois_vs_gt(np.asarray(bb1, np.float64).reshape(1, 4), np.asarray(bb2, np.float64))
    assert int(flags[0].item()) == 0, "get_iou: degenerate box or iou outside [0,1]"
    return float(iou[0].item())


def iou_many(rois_xyxy, grt_xyxy):
    """All proposals of an image against its ground truth in one launch (the loop at rcnn/rcnn.py:35-40).
    Returns (iou float64 ndarray, flags int32 ndarray)."""
    iou, flags = ops.iou_rois_vs_gt(np.asarray(rois_xyxy, np.float64), np.asarray(grt_xyxy, np.float64))
    return iou.cpu().numpy(), flags.cpu().numpy()


def calcul(positive, grt):
    dev = _abi.device_of(positive, grt)
    p = _abi.to_dev(positive, torch.float32, dev).reshape(-1, 4)
    g = _abi.to_dev(grt, torch.float32, dev).reshape(4)
    t = torch.empty_like(p)
    with torch.cuda.device(dev):
        check(load().dg_rcnn_calcul_f32(ptr(p), p.shape[0], ptr(g), ptr(t), stream_ptr(dev)))
    return t if (isinstance(positive, torch.Tensor) and positive.is_cuda) else t.cpu()


def decode_boxes(p_rect, tset, variant=0):
    dev = _abi.device_of(p_rect, tset)
    shape = tuple(torch.as_tensor(tset).shape)
    p = _abi.to_dev(p_rect, torch.int32, dev).reshape(-1, 4)
    t = _abi.to_dev(tset, torch.float32, dev).reshape(-1, 4)
    g = torch.empty_like(t)
    with torch.cuda.device(dev):
        check(load().dg_rcnn_decode_f32(ptr(p), ptr(t), p.shape[0], int(variant), ptr(g), stream_ptr(dev)))
    g = g.reshape(shape)
    return g if (isinstance(tset, torch.Tensor) and tset.is_cuda) else g.cpu()


def label_proposals(rects_xywh, bbox_xywh, label, min_num, pos_thr=0.1, neg_thr=0.1, neg_quota_mult=2, require_full=False):
    """The labelling part of do_ss: IoU of every proposal vs the ground truth on the GPU in one launch,
    then the reference's ordered quota walk (first min_num positives / first quota negatives, stop when
    both quotas have been seen full).  rcnn.py: (0.1, 0.1, 2, False); fast_rcnn.py: (0.5, 0.1, 1, True).
    Returns (rois (k,4) int, grt (4,) float64 xyxy, labels (k,) int)."""
    grt = np.array(bbox_xywh, np.float64)
    grt[2] += grt[0]
    grt[3] += grt[1]
    rects = np.asarray(rects_xywh).astype(np.int64).reshape(-1, 4)
    rois = rects.copy()
    rois[:, 2] += rois[:, 0]
    rois[:, 3] += rois[:, 1]
    n = len(rois)
    if n == 0:
        return np.zeros((0, 4), np.int_), grt, np.zeros((0,), np.int_)
    # IoU of every proposal against the ground truth, then the ordered quota walk, both on the GPU (dg_rcnn_label_quota_f64)
    iou, flags = ops.iou_rois_vs_gt(rois.astype(np.float64), grt)
    dev = iou.device
    sel = torch.empty(2 * n, dtype=torch.int32, device=dev)
    lab = torch.empty(2 * n, dtype=torch.int32, device=dev)
    out = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_rcnn_label_quota_f64(ptr(iou), ptr(flags), n, int(min_num), int(min_num * neg_quota_mult), float(pos_thr),
                                             float(neg_thr), int(label), ptr(sel), ptr(lab), ptr(out), stream_ptr(dev)))
    count, status = (int(v) for v in out.cpu())
    assert not (status & 1), "get_iou: degenerate box or iou outside [0,1]"
    if require_full and (status & 6) != 6:
        return np.zeros((0, 4), np.int_), grt, np.zeros((0,), np.int_)
    sel, labels = sel[:count].cpu().numpy(), lab[:count].cpu().numpy()
    return rois[sel].astype(np.int_).reshape(-1, 4), grt, labels.astype(np.int_)
