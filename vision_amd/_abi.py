"""ctypes binding of libdetgeom.so (include/detgeom.h) + device plumbing.

PyTorch is used only for device memory, streams and (elsewhere) torch.distributed.
There is NO CPU fallback: if the library is missing, or no ROCm device is visible,
every product call raises (vision_amd never imports oracle/).
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdetgeom.so")

OK, EINVAL, EEMPTYROI, ENOMEM, EHIP, ERCCL = 0, -1, -2, -3, -4, -5
COMM_ID_BYTES = 128
POOL_AVG, POOL_MAX = 0, 1

_vp, _i64, _i32, _f32, _f64, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/detgeom.h declares
SIGNATURES = {
    "dg_version": (C.c_char_p, []),
    "dg_last_error": (C.c_int, [C.c_char_p, _sz]),
    "dg_iou_elementwise_f32": (C.c_int, [_vp, _vp, _i64, _f32, _vp, _vp]),
    "dg_iou_elementwise_f64": (C.c_int, [_vp, _vp, _i64, _f64, _vp, _vp]),
    "dg_iou_pairwise_f32": (C.c_int, [_vp, _i64, _vp, _i64, _f32, _vp, _vp]),
    "dg_iou_rois_vs_gt_f64": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "dg_nms_workspace_bytes": (_sz, [_i64, _i64]),
    "dg_nms_batched_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f32, C.c_int, _vp, _vp, _vp, _sz, _vp]),
    "dg_nms_noalloc_max_bytes": (_sz, []),
    "dg_nms_batched_noalloc_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _f32, C.c_int, _vp, _vp, _vp]),
    "dg_nms_per_class_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dg_nms_batched_per_class_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, C.c_int, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dg_nms_path_info": (C.c_int, [_vp, _sz, _i64, _i64, _i64, _vp]),
    "dg_top_score_f64": (C.c_int, [_vp, _i64, _vp, _vp]),
    "dg_nms_phase_sort": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp]),
    "dg_nms_phase_mask": (C.c_int, [_vp, _i64, _i64, _f32, _vp, _sz, _vp]),
    "dg_nms_phase_reduce": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, _vp, _vp, _sz, _vp]),
    "dg_yolo1_decode_f32": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dg_yolo1_decode_nms_f32": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _f32, _f32, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp]),
    "dg_yolo2_get_point_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "dg_yolo2_decode_cells_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "dg_yolo2_decode_anchor_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "dg_roi_windows": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "dg_nchw_to_nhwc_f32": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "dg_roipool_fwd_f32": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp]),
    "dg_roipool_fwd_any_f32": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp]),
    "dg_roipool_bwd_any_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, C.c_int, _vp, _vp, _sz, _vp]),
    "dg_roipool_fwd_a16_f32": (C.c_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp]),
    "dg_roipool_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "dg_roipool_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, C.c_int, _vp, _vp, _sz, _vp]),
    "dg_roipool_bwd_a16_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, C.c_int, _vp, _vp, _sz, _vp]),
    "dg_kmeans_workspace_bytes": (_sz, [_i64, _i64]),
    "dg_kmeans_iou_step": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dg_kmeans_part_elems": (_sz, [_i64]),
    "dg_kmeans_iou_iter": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, C.c_int, _vp, _i64, _vp]),
    "dg_kmeans_finish": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "dg_kmeans_iou_run": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp]),
    "dg_kmeans_labels_i32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "dg_kmeans_prepared_workspace_bytes": (_sz, [_i64, _i64]),
    "dg_kmeans_prepare": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "dg_kmeans_iou_run_prepared": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _sz, _vp]),
    "dg_comm_unique_id": (C.c_int, [_vp]),
    "dg_comm_init": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "dg_comm_rank": (C.c_int, [_vp]),
    "dg_comm_size": (C.c_int, [_vp]),
    "dg_comm_allreduce_i64": (C.c_int, [_vp, _vp, _i64, _vp]),
    "dg_comm_destroy": (C.c_int, [_vp]),
    "dg_comm_graph_state": (C.c_int, [_vp]),
    "dg_kmeans_iou_run_sharded": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _sz, C.c_int, _vp]),
    "dg_rcnn_calcul_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "dg_rcnn_decode_f32": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp]),
    "dg_rcnn_label_quota_f64": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _f64, _f64, _i32, _vp, _vp, _vp, _vp]),
    "dg_yolo1_encode_f64": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "dg_yolo1_loss_workspace_bytes": (C.c_size_t, [_i64]),
    "dg_yolo1_loss_f32": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]),
    "dg_voc_ap_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dg_voc_ap_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _sz, _vp]),
}

_lib = None
_lock = threading.Lock()


class DetGeomError(RuntimeError):
    pass


def load():
    """Load libdetgeom.so (after torch, so the process keeps a single HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise DetGeomError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or make -C vision_amd/csrc). There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    load().dg_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(status, exc=None):
    if status == OK:
        return
    msg = last_error()
    if status == EINVAL:
        raise (exc or ValueError)(msg)
    if status == ENOMEM:
        raise MemoryError(msg)
    raise DetGeomError(f"libdetgeom status {status}: {msg}")


def require_gpu():
    if not torch.cuda.is_available():
        raise DetGeomError("vision_amd needs a ROCm device (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def device_of(*tensors):
    """Device the kernels run on: the device of the first GPU tensor, else the current GPU."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return require_gpu()


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# scratch, grown on demand (the C ABI allocates nothing).  One buffer per (device, stream): calls in flight on different
# streams must not share masks, tickets or RoI lists.  A buffer is allocated while its stream
# is current, so the caching allocator hands a replaced buffer's memory out again only in that stream's order.
_ws = {}
_WS_MAX_ENTRIES = 64


def workspace(device, nbytes):
    stream = torch.cuda.current_stream(device)
    key = (device.index, stream.cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            buf.record_stream(stream)            # kernels already queued on this stream may still be using it
        if len(_ws) >= _WS_MAX_ENTRIES:          # streams come and go: do not keep their buffers for ever
            for k in [k for k in _ws if k != key]:
                _ws.pop(k)                       # its memory returns to the allocator in its own stream's order
        with torch.cuda.device(device):
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def to_dev(x, dtype, device):
    """Host arrays / CPU tensors are copied in; GPU tensors are used in place (contiguous)."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    return x.to(device=device, dtype=dtype).contiguous()
