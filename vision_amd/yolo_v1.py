"""Drop-ins for the box geometry of yolo/yolo_v1 (same names, arguments and return types as the
reference; the arithmetic runs in HIP kernels on the MI355X).

  get_iou(bb1, bb2)                       yolo/yolo_v1/modules.py:73-98
  Loss.get_iou(self, bb1, bb2)            yolo/yolo_v1/modules.py:208-228   (eps 1e-6)
  Loss(num_bboxes, num_classes)(modely, targety)   yolo/yolo_v1/modules.py:234-306   (fused forward + gradient)
  encode(bb, label, num_classes)          yolo/yolo_v1/modules.py:151-174   (+ encode_batched)
  decode(output, target, conf_threshold, prob_threshold)   yolo/yolo_v1/yolo.py:115-186
  nms(out_box, out_score, threshold)      yolo/yolo_v1/yolo.py:188-203
  evaluate_batch(...)                     yolo/yolo_v1/yolo.py:273-299 after the model call
  ap(detects, targets, iou_threds)        yolo/yolo_v1/yolo.py:306-367

Results come back on the device of the inputs (CPU tensors are copied to the GPU and back);
nothing here computes on the CPU except the list bookkeeping the reference also does in Python.
"""
import numpy as np
import torch

from . import ops


def _back(t, like):
    return t if (isinstance(like, torch.Tensor) and like.is_cuda) else t.cpu()


def get_iou(bb1, bb2):
    return _back(ops.iou_elementwise(bb1, bb2, 1e-7), bb1)


class _Yolo1LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, modely, targety, quirk):
        loss6, grad = ops.yolo1_loss(modely.detach(), targety, quirk, want_grad=modely.requires_grad)
        ctx.grad = grad
        ctx.in_device, ctx.in_dtype = modely.device, modely.dtype
        ctx.components = loss6[1:]
        return loss6[0].to(modely.device) if not modely.is_cuda else loss6[0]

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad * gout.to(ctx.grad.device)
        return g.to(device=ctx.in_device, dtype=ctx.in_dtype), None, None


class Loss(torch.nn.Module):
    """Reference Loss (modules.py:198-306) with the forward pass -- in-loss decode, IoU, responsible-box select,
    the five terms -- and its gradient fused into one HIP kernel.  `quirk_slot2=True` keeps modules.py:264
    (slot 2's half-size read from columns 5:7); the IoU target is not detached, as in the reference."""

    def __init__(self, num_bboxes=2, num_classes=5, quirk_slot2=True):
        super().__init__()
        if num_bboxes != 2:
            raise ValueError("Loss: the reference's geometry is written for 2 boxes per cell (modules.py:253-265)")
        self.S = 7
        self.num_bboxes = num_bboxes
        self.num_classes = num_classes
        self.quirk_slot2 = quirk_slot2

    def get_iou(self, bb1, bb2):
        return _back(ops.iou_elementwise(bb1, bb2, 1e-6), bb1)

    def forward(self, modely, targety):
        modely = torch.as_tensor(modely)
        if modely.shape[-1] != 5 * self.num_bboxes + self.num_classes:
            raise ValueError("Loss: last dimension must be 5*num_bboxes + num_classes")
        return _Yolo1LossFn.apply(modely, torch.as_tensor(targety), self.quirk_slot2)


def encode(bb, label, num_classes):
    """Target builder (modules.py:151-174): bb = {left, top, w, h} -> (7,7,10+C) float64 ndarray.  Like the reference it
    turns `bb[:2]` into the box centre IN PLACE when bb is an ndarray."""
    src = np.array(bb, np.float64)
    target, status = ops.yolo1_encode_batched(src.reshape(1, 4), np.array([label], np.int64), num_classes)
    st = int(status.item())
    if st:
        raise IndexError("encode: box centre outside the 7x7 grid" if st & 1 else "encode: label outside 1..num_classes")
    if isinstance(bb, np.ndarray):
        bb[:2] += bb[2:] / 2
    return target[0].cpu().numpy()


def encode_batched(bbs, labels, num_classes, dtype=torch.float32):
    """(n,4) boxes + (n) labels -> (n,7,7,10+C) targets on the GPU in one launch (what a collate step would call)."""
    target, status = ops.yolo1_encode_batched(bbs, labels, num_classes)
    st = int(status.item())
    if st:
        raise IndexError("encode: box centre outside the 7x7 grid" if st & 1 else "encode: label outside 1..num_classes")
    return target.to(dtype)


def decode(output, target, conf_threshold=0.2, prob_threshold=0.1):
    """One image.  Returns (out_box (M,4) f32, out_label (M,) i64, out_conf, out_score, bbox_target (4,) f64,
    label_target int); when nothing passes, the first four are empty Python lists (yolo.py:184-186)."""
    box, label, conf, score, count, tbox, tlabel = ops.yolo1_decode_batched(output, target, conf_threshold, prob_threshold)
    m = int(count[0].item())
    bbox_target = _back(tbox[0], output)
    label_target = int(tlabel[0].item())
    if m == 0:
        return [], [], [], [], bbox_target, label_target
    return (_back(box[0, :m], output), _back(label[0, :m], output), _back(conf[0, :m], output),
            _back(score[0, :m], output), bbox_target, label_target)


def nms(out_box, out_score, threshold=0.1):
    """Class-agnostic greedy NMS, keep indices (int64) in descending-score order, including the
    reference's quirk that the candidate left alone at the end is not kept."""
    if len(out_box) == 0:
        return torch.LongTensor([])
    dev_in = out_box
    boxes = torch.as_tensor(out_box)
    scores = torch.as_tensor(out_score)
    keep, cnt = ops.nms_batched(boxes.reshape(1, -1, 4), scores.reshape(1, -1), None, threshold, True)
    k = int(cnt[0].item())
    if k < 0:           # the pruned path's 64 workgroups never all arrived, with more pairs than its serial fallback holds (nms.hip)
        from ._abi import DetGeomError
        raise DetGeomError("nms: the keep list was not produced (keep_count = -1): the GPU did not run the 64 workgroups of "
                               "the fixed-point kernel together; set DG_NMS_PRUNE=0 to take the dense-mask path")
    return _back(keep[0, :k], dev_in)


def decode_nms_batched(output, target=None, conf_threshold=0.2, prob_threshold=0.1, nms_threshold=0.3, per_class=False,
                       fused=True):
    """Fused path for a whole batch (config 2): decode -> NMS in ONE launch (the image's candidates stay in LDS;
    fused=False: the decode launch followed by the general batched NMS -- same results, four launches).
    Returns device tensors (box, label, conf, score, count, keep, keep_count, tbox, tlabel).
    per_class=False is the reference (class-agnostic NMS per image, yolo.py:282; the split by class comes after it,
    :289-298).  per_class=True is BASELINE.json configs[1] as worded -- the same nms applied to each class's boxes
    separately; keep then lists the classes' survivors in class order (no reference counterpart)."""
    if fused:
        box, label, conf, score, count, tbox, tlabel, keep, kcnt, _ = ops.yolo1_decode_batched(
            output, target, conf_threshold, prob_threshold, nms_threshold, per_class, True)
        return box, label, conf, score, count, keep, kcnt, tbox, tlabel
    box, label, conf, score, count, tbox, tlabel = ops.yolo1_decode_batched(output, target, conf_threshold, prob_threshold)
    if per_class:
        keep, kcnt, _, _ = ops.nms_batched_per_class(box, score, label, torch.as_tensor(output).shape[-1] - 10, count,
                                                     nms_threshold, True)
    else:
        keep, kcnt = ops.nms_batched(box, score, count, nms_threshold, True)
    return box, label, conf, score, count, keep, kcnt, tbox, tlabel


class DecodeNmsGraph:
    """decode -> NMS of a fixed-shape batch as ONE HIP-graph replay (serving / evaluation loops: the launch is a few
    microseconds of kernel time, so a Python call per batch -- ten output allocations and a ctypes call -- costs more
    than the work).  The caller writes predictions into `.output` (and `.target`), calls the object, and reads the
    results from the same tensors every time: (box, label, conf, score, count, keep, keep_count, tbox, tlabel), as
    decode_nms_batched returns them.  Results are overwritten by the next call."""

    def __init__(self, batch, num_classes=5, conf_threshold=0.2, prob_threshold=0.1, nms_threshold=0.3, per_class=False,
                 with_target=True, device=None):
        from . import _abi
        dev = torch.device(device) if device is not None else _abi.require_gpu()
        self.output = torch.zeros((batch, 7, 7, 10 + num_classes), dtype=torch.float32, device=dev)
        self.target = torch.zeros_like(self.output) if with_target else None
        args = (self.output, self.target, conf_threshold, prob_threshold, nms_threshold, per_class)
        side = torch.cuda.Stream(dev)                     # capture needs a non-default stream; warm up on it first
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            decode_nms_batched(*args)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.results = decode_nms_batched(*args)

    def __call__(self):
        self.graph.replay()
        return self.results


def evaluate_batch(output, target, num_classes=5, prob_threshold=0.005, nms_threshold=0.3, detects=None, targets=None):
    """yolo.py:273-299 minus the model call: `output` is the (B,7,7,10+C) prediction.  Appends rows
    [img, conf, x1,y1,x2,y2] to detects[c] and [img, 1, x1,y1,x2,y2] to targets[label-1]."""
    detects = detects if detects is not None else {i: [] for i in range(num_classes)}
    targets = targets if targets is not None else {i: [] for i in range(num_classes)}
    box, label, conf, score, count, keep, kcnt, tbox, tlabel = decode_nms_batched(
        output, target, 0.2, prob_threshold, nms_threshold)
    box, label, conf = box.cpu(), label.cpu(), conf.cpu()
    keep, kcnt, tbox, tlabel = keep.cpu(), kcnt.cpu(), tbox.cpu(), tlabel.cpu()
    for i in range(box.shape[0]):
        targets[int(tlabel[i]) - 1].append([i, 1] + tbox[i].tolist())
        k = int(kcnt[i])
        if k == 0:
            continue
        ids = keep[i, :k]
        b, l, c = box[i][ids], label[i][ids], conf[i][ids]
        for cls in range(num_classes):
            sel = l == cls
            if sel.sum() == 0:
                continue
            pred = torch.cat([c[sel].view(-1, 1), b[sel].view(-1, 4)], dim=1)
            for row in pred:
                detects[cls].append([i] + row.tolist())
    return detects, targets


def evaluate_batch_device(output, target, num_classes=5, prob_threshold=0.005, nms_threshold=0.3, image_offset=0):
    """The rows evaluate_batch appends, left on the device as flat tensors in the reference's list order (image
    ascending, then NMS rank): (det_img int32, det_cls int64, det_conf fp32, det_box (n,4) fp32, tgt_img int32,
    tgt_cls int32, tgt_box (B,4) float64).  No host round trip; feed several batches' results, concatenated, to ap_device."""
    box, label, conf, score, count, keep, kcnt, tbox, tlabel = decode_nms_batched(
        output, target, 0.2, prob_threshold, nms_threshold)
    B, N = keep.shape
    dev = keep.device
    sel = torch.arange(N, device=dev)[None, :] < kcnt[:, None]
    img = torch.arange(B, device=dev)[:, None].expand(B, N)[sel]
    idx = keep[sel]
    return ((img + image_offset).to(torch.int32), label[img, idx], conf[img, idx], box[img, idx],
            torch.arange(image_offset, image_offset + B, device=dev, dtype=torch.int32), (tlabel - 1).to(torch.int32), tbox)


def ap_device(det_img, det_cls, det_conf, det_box, tgt_img, tgt_cls, tgt_box, num_classes, num_images, iou_threds=0.01):
    """ap() on flat device arrays (evaluate_batch_device's layout): the same list the reference returns."""
    out, n_grt = ops.voc_ap(det_img, det_cls, det_conf, det_box, tgt_img, tgt_cls, tgt_box, num_classes, num_images,
                            np.float32(iou_threds))
    out, n_grt = out.cpu().numpy(), n_grt.cpu().numpy()
    aps = []
    for i in range(int(num_classes)):
        if n_grt[i] == 0:
            aps.append(0.0)
        aps.append(float(out[i]))
    return aps


def ap(detects, targets, iou_threds=0.01):
    """VOC-style AP (yolo.py:306-367) for the reference's dict-of-row-lists.  The rows are flattened in list order
    and handed to the device (ops.voc_ap: ordering, IoU vs the image's ground truth, first-hit matching, cumulative
    sums, precision envelope and the area sum all run there); the host only rebuilds the list.  Like the reference,
    an empty class appends 0.0 and then still appends its AP (:314-315)."""
    keys = list(targets.keys())
    pos = {k: i for i, k in enumerate(keys)}
    det_rows = [(pos[c], r) for c in keys for r in detects[c]]
    tgt_rows = [(pos[c], r) for c in keys for r in targets[c]]
    images = sorted({r[0] for _, r in det_rows} | {r[0] for _, r in tgt_rows})
    img_id = {v: i for i, v in enumerate(images)}           # any hashable image key, as in the reference's dict
    d = np.array([[c, img_id[r[0]]] + [float(v) for v in r[1:6]] for c, r in det_rows], np.float64).reshape(-1, 7)
    t = np.array([[c, img_id[r[0]]] + [float(v) for v in r[2:6]] for c, r in tgt_rows], np.float64).reshape(-1, 6)
    out, n_grt = ops.voc_ap(d[:, 1].astype(np.int32), d[:, 0].astype(np.int64), d[:, 2], d[:, 3:7].astype(np.float32),
                            t[:, 1].astype(np.int32), t[:, 0].astype(np.int32), t[:, 2:6].astype(np.float32),
                            max(len(keys), 1), len(images), np.float32(iou_threds))
    out, n_grt = out.cpu().numpy(), n_grt.cpu().numpy()
    aps = []
    for i in range(len(keys)):
        if n_grt[i] == 0:
            aps.append(0.0)
        aps.append(float(out[i]))
    return aps
