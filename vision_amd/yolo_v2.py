"""Drop-ins for yolo/yolo_v2/utils.py.

  get_iou(bb1, bb2)                 yolo/yolo_v2/utils.py:238-263 (byte-identical to the v1 copy)
  get_point(output_size, bbox)      yolo/yolo_v2/utils.py:228-234 (13x13 encoder; output_size is ignored there too)
  decode_v2(points, enc)            NOT in the reference: exact inverse of get_point (parity unpinned)
  decode_anchor(pred, anchors)      NOT in the reference: standard YOLOv2 anchor decode (parity unpinned)
"""
import torch

from . import _abi
from ._abi import check, load, ptr, stream_ptr
from .yolo_v1 import get_iou, _back  # noqa: F401  (same function in the reference)


def get_point(output_size=56, bbox=[0.1, 0.2, 0.3, 0.4]):  # noqa: B006  (the reference's signature, utils.py:228)
    if isinstance(bbox, (list, tuple)):
        # the reference evaluates bbox[2:]/2 (:229): a plain list -- its own default included -- raises there
        raise TypeError("unsupported operand type(s) for /: 'list' and 'int'")
    single = torch.as_tensor(bbox).dim() == 1
    dev = _abi.device_of(bbox)
    b = _abi.to_dev(bbox, torch.float32, dev).reshape(-1, 4)
    n = b.shape[0]
    points = torch.empty((n, 2), dtype=torch.int64, device=dev)
    enc = torch.empty((n, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_yolo2_get_point_f32(ptr(b), n, ptr(points), ptr(enc), stream_ptr(dev)))
    if single:
        points, enc = points[0], enc[0]
    return _back(points, bbox), _back(enc, bbox)


def decode_v2(points, enc):
    dev = _abi.device_of(points, enc)
    p = _abi.to_dev(points, torch.int64, dev).reshape(-1, 2)
    e = _abi.to_dev(enc, torch.float32, dev).reshape(-1, 4)
    out = torch.empty((p.shape[0], 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_yolo2_decode_cells_f32(ptr(p), ptr(e), p.shape[0], ptr(out), stream_ptr(dev)))
    if torch.as_tensor(enc).dim() == 1:
        out = out[0]
    return _back(out, enc)


def decode_anchor(pred, anchors):
    """pred (G,G,A,5+C) fp32, anchors (A,2) as fractions of the image -> (xyxy (G*G*A,4), conf)."""
    dev = _abi.device_of(pred)
    p = _abi.to_dev(pred, torch.float32, dev)
    G, A, C = p.shape[0], p.shape[2], p.shape[3] - 5
    an = _abi.to_dev(anchors, torch.float32, dev).reshape(A, 2)
    xyxy = torch.empty((G * G * A, 4), dtype=torch.float32, device=dev)
    conf = torch.empty((G * G * A,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(load().dg_yolo2_decode_anchor_f32(ptr(p), G, A, C, ptr(an), ptr(xyxy), ptr(conf), stream_ptr(dev)))
    return _back(xyxy, pred), _back(conf, pred)
