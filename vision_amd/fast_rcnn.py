"""Drop-ins for rcnn/fast_rcnn.py.

  slowroipool(output_size).foward(img, roi_ary, roi_idx)   rcnn/fast_rcnn.py:351-377
  nms(rects, scores) -> (rect, score)                      rcnn/fast_rcnn.py:590-610

As written the reference pools with AdaptiveAvgPool2d stored in a field called `maxpool`
(:354) and never defines backward (autograd).  `mode="avg"` (the default: an import swap must
reproduce the class as written) is that; `mode="max"` is the RoI MAX-pooling north_star asks
for -- the same module with AdaptiveMaxPool2d substituted -- and also yields the argmax used
by the backward scatter.
"""
import numpy as np
import torch

from . import ops


class _RoIPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, win, mode, size):
        small = tuple(size) == (7, 7) and img.shape[2] * img.shape[3] <= 65535
        out, argmax = ops.roipool_forward(img, win, mode, P=size,
                                          argmax_dtype=torch.int16 if small else torch.int32)   # uint16 indices: half the bytes
        ctx.mode, ctx.img_shape, ctx.size = mode, tuple(img.shape), tuple(size)
        ctx.save_for_backward(win, argmax if argmax is not None else win)
        ctx.argmax = argmax
        return out

    @staticmethod
    def backward(ctx, grad_out):
        win, arg = ctx.saved_tensors
        g = ops.roipool_backward(grad_out.contiguous(), arg if ctx.mode == "max" else None, win, ctx.img_shape, ctx.mode, P=ctx.size)
        return g, None, None, None


class slowroipool(torch.nn.Module):
    def __init__(self, output_size, mode="avg"):
        super().__init__()
        size = tuple(int(v) for v in output_size) if isinstance(output_size, (tuple, list)) else (int(output_size), int(output_size))
        if len(size) != 2 or min(size) < 1:
            raise ValueError("slowroipool: output_size must be a positive int or a pair of them")
        if mode not in ("max", "avg"):
            raise ValueError("mode must be 'max' or 'avg'")
        self.size = size
        self.mode = mode
        self.argmax = None            # (n,C,PH,PW) int32 of the last max-mode forward (flat y*W+x in the map)

    def foward(self, img, roi_ary, roi_idx):          # sic: the reference's spelling (:357)
        was_cpu = not img.is_cuda
        dev = ops._abi.device_of(img)
        x = img.to(device=dev, dtype=torch.float32)
        B, C, H, W = x.shape
        win, status = ops.roi_windows(np.asarray(roi_ary, np.float64) if not isinstance(roi_ary, torch.Tensor) else roi_ary,
                                      np.asarray(roi_idx, np.int64) if not isinstance(roi_idx, torch.Tensor) else roi_idx,
                                      B, H, W, dev)
        st = int(status.item())
        if st & 1:
            raise RuntimeError("slowroipool: a RoI crop is empty (x2<=x1 or y2<=y1 after floor/ceil)")
        if st & 2:
            raise IndexError("slowroipool: roi_idx out of range")
        if x.requires_grad:
            out = _RoIPoolFn.apply(x, win, self.mode, self.size)
        else:
            out, self.argmax = ops.roipool_forward(x, win, self.mode, P=self.size)
        return out.cpu() if was_cpu else out

    forward = foward


def nms(rects, scores):
    """rcnn/fast_rcnn.py:590-610.  The reference's function of this name does no suppression: it orders the rects by
    np.argsort(scores)[::-1] and returns element 0 -- the rect of the maximal score, taken on the scores as given
    (float64).  That is what runs here (dg_top_score_f64): exact on scores that differ below fp32 resolution; equal
    maxima resolve to the highest index (in the reference the order among equal keys is whatever numpy's unstable
    argsort produces, so its pick among exactly tied scores is implementation-defined)."""
    rects_a = np.array(rects)
    scores_a = np.array(scores)
    if scores_a.size == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")     # what nms_rects[0] raises in the reference
    dev = ops._abi.device_of(None)
    sc = ops._abi.to_dev(scores_a.astype(np.float64).reshape(-1), torch.float64, dev)
    top = torch.empty((1,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        ops.check(ops.load().dg_top_score_f64(ops.ptr(sc), sc.shape[0], ops.ptr(top), ops.stream_ptr(dev)))
    top = int(top.item())
    return rects_a[top], scores_a[top]
