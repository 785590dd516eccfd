"""Anchor k-means with the 1-IoU distance behind the sklearn-style API the reference uses
(clustering/cluster_prac.py:38-40: kmean(n_clusters=5, random_state=42).fit / .fit_predict).

The reference itself runs sklearn's Euclidean KMeans on a toy table; the 1-IoU Lloyd iteration is
build-defined (spec: oracle/oracle.c) and therefore "parity unpinned".
Multi-GPU: every rank holds a contiguous shard of the boxes; per iteration ONE launch (the centroid
update is the prologue of the next step, dg_kmeans_iou_iter) and ONE all-reduce of the (3k+2) int64
accumulator over RCCL / xGMI.  Two ways to issue it:
  comm="so"    (default on GPUs): the library owns an RCCL communicator (dg_comm_*) and
               issues the whole chunk -- launches and all-reduces -- from one C call
               (dg_kmeans_iou_run_sharded); DG_KMEANS_GRAPH=1 captures it in a HIP graph from its second run on;
  comm="torch": launch + torch.distributed.all_reduce per iteration from Python, the chunk captured
               with torch.cuda.graph when the process group allows it (and the gloo / host-side
               rehearsal path, reduce_device="cpu").
Integer sums make the result bit-identical for every world size and either way.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _abi, ops

_CHUNK = 8          # iterations launched between two host checks of the changed-count history
_CHUNK_MAX = 32     # the prepared fit's chunks grow (8, 8, 16, 32, 32, ...) -- the end of a chunk costs what seven of its iterations do
_CELLS_MIN_BOXES = 200_000   # single-device fits of at least this many boxes take the prepared (cell) path


class DgComm:
    """RCCL communicator owned by libdetgeom.so (include/detgeom.h dg_comm_*): one per process and device."""

    def __init__(self, handle, rank, world, device):
        self.handle, self.rank, self.world, self.device = handle, rank, world, device

    @classmethod
    def create(cls, group=None, device=None):
        """Collective over the ranks of `group` (torch.distributed carries the 128-byte id from rank 0 to the others);
        without an initialised process group: a single-rank communicator."""
        lib = _abi.load()
        dev = device if device is not None else _abi.require_gpu()
        have_pg = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if have_pg else 0
        world = dist.get_world_size(group) if have_pg else 1
        uid = C.create_string_buffer(_abi.COMM_ID_BYTES)
        if rank == 0:
            _abi.check(lib.dg_comm_unique_id(uid))
        if world > 1:
            box = [uid.raw]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            with torch.cuda.device(dev):
                dist.broadcast_object_list(box, src=src, group=group)
            uid = C.create_string_buffer(box[0], _abi.COMM_ID_BYTES)
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _abi.check(lib.dg_comm_init(C.byref(h), uid, rank, world))
        return cls(h, rank, world, dev)

    def allreduce_i64(self, t):
        assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()
        with torch.cuda.device(self.device):
            _abi.check(_abi.load().dg_comm_allreduce_i64(self.handle, _abi.ptr(t), t.numel(), _abi.stream_ptr(self.device)))

    def graph_state(self):
        return _abi.load().dg_comm_graph_state(self.handle)

    def close(self):
        if self.handle is not None and self.handle.value:
            h, self.handle = self.handle, None
            _abi.check(_abi.load().dg_comm_destroy(h))


class _HipBackend:
    """Device side of the Lloyd loop: HIP kernels through the C ABI (the only backend the product has)."""

    def prepare(self, X, init_centers, k, iters, cells=None):
        dev = _abi.device_of(X)
        lib = _abi.load()
        self.wh = _abi.to_dev(X, torch.float32, dev).reshape(-1, 2)
        self.cent = _abi.to_dev(np.asarray(init_centers, np.float32), torch.float32, dev).reshape(k, 2).clone()
        n = self.wh.shape[0]
        # single device, many boxes: the prepared fit (boxes grouped by (w, h) cell once, iterations over pieces of cells;
        # same bits, include/detgeom.h).  DG_KMEANS_CELLS=0 keeps the box-by-box loop, =1 forces the prepared one.
        # The grouping (0.2 ms of device time for 10 M boxes) is issued FIRST: the small allocations and fills below each cost
        # the host 10-15 us, and the device would sit idle for all of them otherwise.
        self.cells = None
        mode = os.environ.get("DG_KMEANS_CELLS", "")
        if cells is not False and mode != "0" and n >= 1 and (mode == "1" or cells is True or n >= _CELLS_MIN_BOXES):
            nb = lib.dg_kmeans_prepared_workspace_bytes(n, k)
            self.cells = (torch.empty(nb, dtype=torch.uint8, device=dev), nb)
            self.reprepare(fresh=True)
        self.labels = torch.full((n,), 255, dtype=torch.uint8, device=dev)
        # one zeroed buffer for acc and the history, and no fill for what is zeroed before use anyway (`part`: begin() does it)
        zero = torch.zeros(3 * k + 2 + max(iters, 1), dtype=torch.int64, device=dev)
        self.acc, self.history = zero[:3 * k + 2], zero[3 * k + 2:]
        # single device: workspace of dg_kmeans_iou_run; sharded: three partial-sums buffers in rotation (include/detgeom.h),
        # `part[i]` is what gets all-reduced
        self.ws_bytes = lib.dg_kmeans_workspace_bytes(n, k)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.part = torch.empty((3, lib.dg_kmeans_part_elems(k)), dtype=torch.int64, device=dev)
        self.cur = None               # index of the buffer the last launch added into
        self._bound = None

    def reprepare(self, fresh=False):
        """group the boxes (and the current labels) by (w, h) cell again: dg_kmeans_prepare on this fit's workspace.
        fresh: no box has a label yet (self.labels is all 255, as prepare() leaves it) -- the library is told so (labels = NULL)
        and does not move ten million label bytes that all say the same"""
        dev = self.wh.device
        with torch.cuda.device(dev):
            _abi.check(_abi.load().dg_kmeans_prepare(_abi.ptr(self.wh), self.wh.shape[0], self.cent.shape[0],
                                                     None if fresh else _abi.ptr(self.labels),
                                                     _abi.ptr(self.cells[0]), self.cells[1], _abi.stream_ptr(dev)))

    def run_local(self, slot0, iters, light=False):
        """single device: `iters` launches issued by one C call (dg_kmeans_iou_run).  light (prepared fit only): the iterations
        over cells alone -- centroids and history, no labels / inertia / acc (include/detgeom.h)"""
        dev = self.wh.device
        if self.cells is not None:
            with torch.cuda.device(dev):
                _abi.check(_abi.load().dg_kmeans_iou_run_prepared(_abi.ptr(self.wh), self.wh.shape[0], _abi.ptr(self.cent),
                                                                  self.cent.shape[0], None if light else _abi.ptr(self.labels),
                                                                  None if light else _abi.ptr(self.acc),
                                                                  _abi.ptr(self.history), slot0, iters, _abi.ptr(self.cells[0]),
                                                                  self.cells[1], _abi.stream_ptr(dev)))
            return
        with torch.cuda.device(dev):
            _abi.check(_abi.load().dg_kmeans_iou_run(_abi.ptr(self.wh), self.wh.shape[0], _abi.ptr(self.cent), self.cent.shape[0],
                                                     _abi.ptr(self.labels), _abi.ptr(self.acc), _abi.ptr(self.history), slot0, iters,
                                                     _abi.ptr(self.ws), self.ws_bytes, _abi.stream_ptr(dev)))

    def run_sharded(self, comm, slot0, iters, use_graph):
        """sharded, comm="so": `iters` x [launch; RCCL all-reduce] + finish issued by one C call (dg_kmeans_iou_run_sharded)"""
        dev = self.wh.device
        with torch.cuda.device(dev):
            _abi.check(_abi.load().dg_kmeans_iou_run_sharded(comm.handle, _abi.ptr(self.wh), self.wh.shape[0], _abi.ptr(self.cent),
                                                             self.cent.shape[0], _abi.ptr(self.labels), _abi.ptr(self.acc),
                                                             _abi.ptr(self.history), slot0, iters, _abi.ptr(self.ws), self.ws_bytes,
                                                             int(use_graph), _abi.stream_ptr(dev)))

    # The sharded loop issues one launch + one all-reduce per iteration from Python: with the kernel at a few microseconds
    # per shard the host side matters, so the ABI calls are bound once with raw pointers.
    def _bind(self):
        lib = _abi.load()
        dev = self.wh.device
        n, k = self.wh.shape[0], self.cent.shape[0]
        st = _abi.stream_ptr(dev)
        head = (_abi.ptr(self.wh), n, _abi.ptr(self.cent), k, _abi.ptr(self.labels))
        hist = _abi.ptr(self.history)
        parts = [_abi.ptr(self.part[i]) for i in range(3)]
        iter_fn, fin_fn, check = lib.dg_kmeans_iou_iter, lib.dg_kmeans_finish, _abi.check
        cent, acc = _abi.ptr(self.cent), _abi.ptr(self.acc)
        self._bound = (dev, st.value,
                       lambda prev, out, zero, inertia, slot: check(iter_fn(*head, parts[prev] if prev is not None else None,
                                                                            parts[out], parts[zero], inertia, hist, slot, st)),
                       lambda cur, slot: check(fin_fn(parts[cur], k, cent, acc, hist, slot, st)))

    def _fast(self):
        b = getattr(self, "_bound", None)
        if b is None or b[0] != self.wh.device or b[1] != _abi.stream_ptr(self.wh.device).value:
            self._bind()
            b = self._bound
        return b

    def begin(self):
        """start of a chunk of sharded iterations: all three buffers zero, no previous sums"""
        self.part.zero_()
        self.cur = None

    def iterate(self, apply_prev, inertia, slot):
        """one sharded iteration as one launch: [centroids <- the previous (all-reduced) buffer; history[slot] <- its changed
        count;] step; this shard's sums are ADDED into the next buffer of the rotation (`reduce_buffer()`)"""
        prev = self.cur if apply_prev else None
        out = 0 if self.cur is None else (self.cur + 1) % 3
        with torch.cuda.device(self.wh.device):
            self._fast()[2](prev, out, (out + 1) % 3, int(inertia), slot)
        self.cur = out

    def reduce_buffer(self):
        """the tensor to all-reduce (SUM) across ranks after iterate()"""
        return self.part[self.cur]

    def update(self, slot):
        """after the last iteration of a chunk (and its all-reduce): fold, move the centroids, history[slot] <- changed"""
        with torch.cuda.device(self.wh.device):
            self._fast()[3](self.cur, slot)

    def step(self):
        """plain step (kernel + fold): acc <- this shard's sums with the inertia, centroids untouched"""
        ops.kmeans_step(self.wh, self.cent, self.labels, self.acc)

    def history_host(self, upto):
        h = self.history[:upto].cpu().numpy()
        if (h < 0).any():       # the persistent kernel's grid barrier gave up (include/detgeom.h); never seen, but not silent
            raise _abi.DetGeomError("k-means: the persistent iteration kernel abandoned its grid barrier (history holds -1); "
                                    "set DG_KMEANS_PERSIST=0")
        return h

    def results(self):
        return (self.cent.cpu().numpy(), ops.kmeans_labels_i32(self.labels).cpu().numpy(), self.acc.cpu().numpy())

    def results_lazy(self):
        """centroids and accumulator on the host; the labels stay on the device (uint8) until somebody asks for them"""
        return (self.cent.cpu().numpy(), self.labels, self.acc.cpu().numpy())


class kmean:
    def __init__(self, n_clusters=5, random_state=42, max_iter=300, group=None, reduce_device=None, sharded=None, comm=None):
        """sharded: None = the sharded loop iff the process group has more than one rank; True forces it (with a
        single-rank group it then runs launch + all-reduce per iteration, which is how its device path is tested).
        comm: "so" = the library's own RCCL communicator and C-side loop, "torch" = torch.distributed.all_reduce from
        Python, None = "so" unless the all-reduce runs on the host (reduce_device), the process group is not RCCL, or
        DG_KMEANS_COMM=torch; a DgComm instance is used as it is (and stays the caller's to close)."""
        if not 1 <= n_clusters <= 16:
            raise ValueError("n_clusters must be in [1,16]")
        self.n_clusters, self.random_state, self.max_iter = n_clusters, random_state, max_iter
        self.group = group
        self.reduce_device = reduce_device      # all-reduce on another device (e.g. "cpu" with gloo)
        self.sharded = sharded
        self.comm = comm
        self._comm = comm if isinstance(comm, DgComm) else None
        self.comm_state = "unused"              # "so" / "torch: <why>" once the sharded loop has run
        self._side = None
        self._backend = _HipBackend()
        self._graphs = {}
        self.graph_state = "unused"             # "captured" / "eager: <why>" once the sharded loop has run
        self.cluster_centers_ = None
        self._labels_dev, self._labels_host = None, None
        self.n_iter_ = 0
        self.inertia_ = None

    # -- distributed plumbing -------------------------------------------------------------
    def _world(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _is_sharded(self):
        return self._world() > 1 if self.sharded is None else bool(self.sharded)

    def _allreduce(self, acc):
        if not (dist.is_available() and dist.is_initialized()):
            return
        if self.reduce_device is not None:
            t = acc.to(self.reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            acc.copy_(t)
        else:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)

    @staticmethod
    def init_centers(X_all, k, random_state):
        """first k rows of a RandomState(random_state) permutation of ALL boxes (host, deterministic)."""
        X_all = np.asarray(X_all, np.float32)
        perm = np.random.RandomState(random_state).permutation(X_all.shape[0])
        return np.ascontiguousarray(X_all[perm[:k]]).copy()

    # -- results --------------------------------------------------------------------------
    @property
    def labels_(self):
        """int32 ndarray as sklearn's.  Widening 10 M labels and copying them to the host costs ten times the 30 Lloyd
        iterations that produced them, so it happens on first access, not in fit(); the device tensor it is made from is
        this fit's own (uint8, `labels_device_`).  None before the first fit."""
        if self._labels_host is None and self._labels_dev is not None:
            self._labels_host = ops.kmeans_labels_i32(self._labels_dev).cpu().numpy()
        return self._labels_host

    @labels_.setter
    def labels_(self, value):
        self._labels_host, self._labels_dev = value, None

    @property
    def labels_device_(self):
        """this fit's labels where they were computed: uint8 tensor on the GPU (no copy)"""
        return self._labels_dev

    # -- the fit --------------------------------------------------------------------------
    def fit(self, X, init_centers=None, fixed_iters=None):
        """X: this rank's shard, (n,2) w,h in (0,1].  With more than one rank pass `init_centers`
        (computed from the full set, e.g. kmean.init_centers on rank 0 and broadcast)."""
        k = self.n_clusters
        if init_centers is None:
            if self._world() > 1:
                raise ValueError("sharded fit needs init_centers drawn from the full box set")
            src = X.cpu().numpy() if isinstance(X, torch.Tensor) else X
            init_centers = self.init_centers(src, k, self.random_state)
        iters = fixed_iters if fixed_iters is not None else self.max_iter
        be = self._backend
        self._graphs = {}
        sharded = self._is_sharded()
        be.prepare(X, init_centers, k, min(iters, _CHUNK if sharded else max(_CHUNK, _CHUNK_MAX)) if fixed_iters is None else iters,
                   cells=False if sharded else None)
        done, n_iter = 0, 0
        while done < iters:
            # one device: a chunk ends in an iteration over the boxes (labels, inertia), two small launches and a read-back of the
            # history -- about 0.1 ms, seven prepared iterations -- while an iteration past convergence costs one: the chunks
            # grow (8, 8, 16, 32, 32, ...).  The box loop's chunk ends in a read-back only and its iterations cost twice as much:
            # fixed chunks (measured: growing ones 5.2 -> 5.3 ms for 156 iterations); sharded: fixed too (one captured graph)
            prepared = getattr(be, "cells", None) is not None
            grow = _CHUNK if not prepared or done < 2 * _CHUNK else (2 * _CHUNK if done < 4 * _CHUNK else max(_CHUNK, _CHUNK_MAX))
            chunk = min(grow if fixed_iters is None else iters, iters - done)
            # a prepared chunk that cannot be the fit's last one (iterations remain after it) skips the pass over the boxes: if
            # it turns out to hold the first iteration that changes nothing, the centroids have stopped moving and one plain step
            # at them gives the labels and the inertia that the chunk's last iteration would have given
            light = prepared and fixed_iters is None and done + chunk < iters
            # a converged iteration (changed == 0) recomputes identical centroids, so running the
            # rest of the chunk is idempotent; every chunk writes history[0:chunk], read back here.
            self._run_iters(0, chunk, light=light)
            if fixed_iters is None:
                z = np.nonzero(be.history_host(chunk) == 0)[0]
                if z.size:
                    n_iter = done + int(z[0]) + 1
                    if light:
                        be.step()
                    break
            done += chunk
            n_iter = done
        cent, labels, acc = be.results_lazy() if hasattr(be, "results_lazy") else be.results()
        self.cluster_centers_ = cent
        if isinstance(labels, torch.Tensor):
            self._labels_dev, self._labels_host = labels, None
        else:                                           # a stand-in backend (tests) hands over host labels
            self._labels_dev, self._labels_host = None, np.asarray(labels)
        self.n_iter_ = n_iter
        self.inertia_ = float(acc[3 * k + 1]) / 4294967296.0
        return self

    def _sharded_chunk(self, slot0, count):
        """count iterations: [launch, all-reduce] x count, then the finish.  The inertia is formed by the
        chunk's last launch only, exactly as dg_kmeans_iou_run does on a single device."""
        be = self._backend
        be.begin()
        for j in range(count):
            be.iterate(j > 0, j + 1 == count, slot0 + j - 1)
            self._allreduce(be.reduce_buffer())
        be.update(slot0 + count - 1)

    def _so_comm(self):
        """the library-side communicator for this fit, or None (then torch.distributed carries the all-reduce)"""
        if self._comm is not None:
            return self._comm
        if self.comm_state.startswith("torch"):
            return None
        want = self.comm if self.comm is not None else os.environ.get("DG_KMEANS_COMM") or None
        have_pg = dist.is_available() and dist.is_initialized()
        if want == "torch" or self.reduce_device is not None or not self._backend.acc.is_cuda:
            self.comm_state = "torch: asked for, or host-side all-reduce"
            return None
        if want is None and have_pg and dist.get_backend(self.group) != "nccl":
            self.comm_state = "torch: process group is not RCCL"
            return None
        try:
            self._comm = DgComm.create(self.group, self._backend.acc.device)
            self._own_comm = True
            self.comm_state = "so"
        except Exception as e:  # noqa: BLE001  (librccl.so absent: symmetric on every rank)
            if want == "so":
                raise
            self.comm_state = f"torch: dg_comm_init failed ({type(e).__name__}: {e})"
        return self._comm

    def close(self):
        """frees the communicator this object created (collective: every rank calls it)"""
        if getattr(self, "_own_comm", False) and self._comm is not None:
            self._comm.close()
            self._comm, self._own_comm = None, False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # destroying an RCCL communicator is collective: a finalizer runs at a different moment on every rank (GC order,
        # interpreter shutdown, after the process group or librccl itself is gone), so it is NOT done here.  Call close()
        # on every rank, or use the object as a context manager; a communicator left open is only reported.
        if getattr(self, "_own_comm", False) and getattr(self, "_comm", None) is not None:
            try:
                import warnings
                warnings.warn("vision_amd.cluster.kmean: the library-side communicator was not closed (call close() on every "
                              "rank); it is left to process exit", ResourceWarning, stacklevel=2)
            except Exception:  # noqa: BLE001
                pass

    def _run_so(self, comm, slot0, count):
        be = self._backend
        dev = be.wh.device
        # issued from C the host keeps ahead of the device without a graph (measured on a 1.25 M-box shard: 8.5 us per iteration
        # eager, 10.2 us replayed on a side stream), so capture is opt-in here: DG_KMEANS_GRAPH=1
        use_graph = os.environ.get("DG_KMEANS_GRAPH", "0") == "1"
        cur = torch.cuda.current_stream(dev)
        if use_graph and cur.cuda_stream == 0:
            # the legacy default stream cannot be captured: the chunk runs on a side stream ordered after / before it
            if self._side is None:
                self._side = torch.cuda.Stream(dev)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                be.run_sharded(comm, slot0, count, True)
            cur.wait_stream(self._side)
        else:
            be.run_sharded(comm, slot0, count, use_graph)
        if not use_graph:
            self.graph_state = "eager: graph capture disabled"
        else:
            self.graph_state = {2: "captured", 1: "armed"}.get(comm.graph_state(), "eager: " + _abi.last_error())

    def _run_iters(self, slot0, count, light=False):
        """`count` Lloyd iterations, asynchronous on the current stream (history slots slot0..slot0+count-1)."""
        be = self._backend
        if not self._is_sharded():
            if light:
                be.run_local(slot0, count, light=True)
            else:
                be.run_local(slot0, count)
            return
        comm = self._so_comm()
        if comm is not None:
            self._run_so(comm, slot0, count)
            return
        # HIP graph of the whole chunk (kernels + RCCL all-reduces): replays cost the host one launch.  Only when the
        # all-reduce runs on the device (not the gloo rehearsal path) and DG_KMEANS_GRAPH != 0; any capture failure
        # falls back to eager launches for the rest of this fit.
        use_graph = (self.reduce_device is None and be.acc.is_cuda and os.environ.get("DG_KMEANS_GRAPH", "1") != "0"
                     and not str(self.graph_state).startswith("eager"))
        if not use_graph:
            if self.graph_state == "unused":
                self.graph_state = "eager: graph capture disabled or host-side all-reduce"
            self._sharded_chunk(slot0, count)
            return
        key = (slot0, count, be.wh.data_ptr())
        g = self._graphs.get(key)
        if g is None:
            self._sharded_chunk(slot0, count)            # warm-up outside capture (RCCL channels, occupancy look-ups)
            try:
                torch.cuda.synchronize(be.wh.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL's watchdog thread keeps polling its events
                    self._sharded_chunk(slot0, count)
                self._graphs[key] = g
                self.graph_state = "captured"
            except Exception as e:  # noqa: BLE001  (capture of a collective is backend-dependent)
                self.graph_state = f"eager: capture failed ({type(e).__name__}: {e})"
                be._bound = None
            return                                        # the warm-up run WAS this chunk
        g.replay()

    def fit_predict(self, X, **kw):
        return self.fit(X, **kw).labels_

    def predict(self, X):
        dev = _abi.device_of(X)
        wh = _abi.to_dev(X, torch.float32, dev).reshape(-1, 2)
        cent = _abi.to_dev(self.cluster_centers_, torch.float32, dev)
        labels = torch.zeros((wh.shape[0],), dtype=torch.uint8, device=dev)
        acc = torch.zeros(3 * self.n_clusters + 2, dtype=torch.int64, device=dev)
        ops.kmeans_step(wh, cent, labels, acc)
        return ops.kmeans_labels_i32(labels).cpu().numpy()


AnchorKMeans = kmean
