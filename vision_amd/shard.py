"""How the path partitions over the GPUs of one node: one process per GPU.

  * batched NMS, YOLO decode and RoI pooling shard by IMAGE: rank r owns a contiguous range of
    images; RoIs follow their image; there is no data-path collective (results are per image);
  * anchor k-means shards BOXES: contiguous shards, centroids replicated, one all-reduce of the
    (3k+2) int64 accumulator per iteration (vision_amd/cluster.py);
  * single-image NMS and the 64k pairwise matrix do not shard: replicas only.
"""


def shard_range(n, rank, world):
    """Contiguous range [lo, hi) of n items owned by `rank`: the first n % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rois_by_image(roi_idx, rank, world, n_images):
    """Indices of the RoIs whose image belongs to `rank`, and the image range; roi_idx: 1-D integer array."""
    import numpy as np
    lo, hi = shard_range(n_images, rank, world)
    roi_idx = np.asarray(roi_idx)
    sel = np.nonzero((roi_idx >= lo) & (roi_idx < hi))[0]
    return sel, (lo, hi)
