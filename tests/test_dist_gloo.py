"""world_size-2 (and 3) CPU tests of the multi-GPU logic over the gloo backend.

The kernels themselves need a GPU, so here the device side of the Lloyd loop is replaced by a
test-only backend that steps with the CPU oracle; what is under test is the product's host logic:
shard ranges, the per-iteration integer all-reduce, convergence bookkeeping — and that any
world size gives bit-identical centroids/labels (integer sums are associative)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vision_amd import cluster, shard
from tools import workloads as wl


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 256, 257, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    sel, (lo, hi) = shard.shard_rois_by_image(np.array([0, 0, 1, 2, 2, 3]), 1, 2, 4)
    assert (lo, hi) == (2, 4) and list(sel) == [3, 4, 5]
    with pytest.raises(ValueError):
        shard.shard_range(4, 2, 2)


class OracleBackend:
    """test-only stand-in for cluster._HipBackend (same methods), stepping with oracle/oracle.c"""

    def prepare(self, X, init_centers, k, iters, cells=None):
        from oracle import oracle as orc
        self.orc, self.k = orc, k
        self.wh = np.ascontiguousarray(X, np.float32)
        self.cent = np.array(init_centers, np.float32).reshape(k, 2).copy()
        self.lab = np.full(self.wh.shape[0], 255, np.uint8)
        self.acc = torch.zeros(3 * k + 2, dtype=torch.int64)
        self.history = np.zeros(max(iters, 1), np.int64)

    def step(self):
        a = self.orc.kmeans_step(self.wh, self.cent, self.lab)
        self.acc.copy_(torch.from_numpy(a))

    def update(self, slot):
        a = self.acc.numpy()
        self.cent = self.orc.kmeans_update(a, self.cent)
        self.history[slot] = a[3 * self.k]

    def begin(self):
        pass

    def reduce_buffer(self):
        return self.acc

    def iterate(self, apply_prev, inertia, slot):
        if apply_prev:
            self.update(slot)
        self.step()
        if not inertia:
            self.acc[3 * self.k + 1] = 0

    def run_local(self, slot0, iters):
        for t in range(iters):
            self.step(); self.update(slot0 + t)

    def history_host(self, upto):
        return self.history[:upto]

    def results(self):
        return self.cent.copy(), self.lab.astype(np.int32), self.acc.numpy().copy()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wh = wl.anchor_wh(7, n)
        lo, hi = shard.shard_range(n, rank, world)
        init = cluster.kmean.init_centers(wh, 5, 42)
        km = cluster.kmean(5, 42)
        km._backend = OracleBackend()
        km.fit(wh[lo:hi], init_centers=init)
        with pytest.raises(ValueError):
            cluster.kmean(5, 42).fit(wh[lo:hi])          # sharded fit without global init centres
        np.savez(os.path.join(out, f"r{rank}.npz"), cent=km.cluster_centers_, labels=km.labels_, n_iter=km.n_iter_,
                 inertia=km.inertia_, lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_kmeans_sharded_gloo_equals_single_process(world, tmp_path):
    from oracle import oracle as orc
    orc.build()
    n = 30011
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    wh = wl.anchor_wh(7, n)
    cent, labels, n_iter, inertia = orc.kmeans_fit(wh, 5, 42)
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(z["cent"], cent)                               # bit-identical centroids on every rank
        assert np.array_equal(z["labels"], labels[int(z["lo"]):int(z["hi"])])
        assert int(z["n_iter"]) == n_iter and abs(float(z["inertia"]) - inertia) < 1e-9


def test_single_process_loop_with_oracle_backend():
    """world = 1 path of the same host loop (chunked, history-based n_iter_)"""
    from oracle import oracle as orc
    wh = wl.anchor_wh(9, 5000)
    km = cluster.kmean(5, 42)
    km._backend = OracleBackend()
    km.fit(wh)
    cent, labels, n_iter, inertia = orc.kmeans_fit(wh, 5, 42)
    assert np.array_equal(km.cluster_centers_, cent) and np.array_equal(km.labels_, labels) and km.n_iter_ == n_iter
    km2 = cluster.kmean(5, 42)
    km2._backend = OracleBackend()
    km2.fit(wh, fixed_iters=3)
    c3, l3, n3, _ = orc.kmeans_fit(wh, 5, 42, fixed_iters=3)
    assert np.array_equal(km2.cluster_centers_, c3) and km2.n_iter_ == 3
