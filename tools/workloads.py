"""Seeded synthetic inputs for the five BASELINE.json configs.

Pure NumPy on the host: tests, the golden-vector generator and bench.py all draw
their inputs from here so that "same seeded inputs" means the same bytes everywhere.
Nothing in this file computes a result of the hot path.
"""
import numpy as np

# ---------------------------------------------------------------- scores
def tiefree_scores(seed, n):
    """fp32 scores in (0,1), pairwise distinct (the reference sort is unstable,
    yolo/yolo_v1/yolo.py:194, so fixtures must be tie-free)."""
    rng = np.random.default_rng(seed)
    if n == 0:
        return np.zeros((0,), np.float32)
    s = (rng.permutation(n).astype(np.float64) + rng.uniform(0.25, 0.75, n)) / n
    s = s.astype(np.float32)
    assert np.unique(s).size == n
    return s


# ---------------------------------------------------------------- config 1 / 3
def frcnn_rois(seed, n=2000):
    """Normalised xyxy RoIs, float64, all non-empty after floor/ceil on a 38x50 map."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0.0, 0.7, (n, 2))
    wh = rng.uniform(0.02, 0.30, (n, 2))
    return np.concatenate([xy, xy + wh], axis=1)


def frcnn_rois_wide(seed, n=2000):
    """Realistic proposal mix without a size bound (rcnn/fast_rcnn.py:366-369 puts none on the RoI): w,h ~ U(0.1,0.9) of
    the map, i.e. crops of 4..46 x 4..35 cells on a 38x50 map; all non-empty."""
    rng = np.random.default_rng(5000 + seed)
    wh = rng.uniform(0.1, 0.9, (n, 2))
    xy = rng.uniform(0.0, 1.0, (n, 2)) * (1.0 - wh)
    return np.concatenate([xy, xy + wh], axis=1)


def frcnn_feature_map(seed, b=1, c=512, h=38, w=50):
    rng = np.random.default_rng(1000 + seed)
    return rng.standard_normal((b, c, h, w), dtype=np.float32)


def frcnn_proposals(seed, n=2000):
    """boxes fp32 xyxy (the RoIs rounded to fp32) + tie-free scores."""
    return frcnn_rois(seed, n).astype(np.float32), tiefree_scores(seed + 7, n)


# ---------------------------------------------------------------- config 2
def yolo1_batch(seed, batch=64, num_classes=5, one_slot=False):
    """output ~ U(0,1) (sigmoid range) (batch,7,7,10+C) fp32 and a one-object target.

    Both box slots of a cell share the cell's class score, which is the NMS sort key
    (yolo/yolo_v1/yolo.py:168,173,194): with both slots alive the reference's unstable
    sort sees ties and its order is implementation-defined.  `one_slot=True` zeroes the
    confidence of one random slot per cell so that parity fixtures are tie-free."""
    rng = np.random.default_rng(seed)
    out = rng.uniform(0, 1, (batch, 7, 7, 10 + num_classes)).astype(np.float32)
    if one_slot:
        dead = rng.integers(0, 2, (batch, 7, 7))
        out[..., 4][dead == 0] = 0.0
        out[..., 9][dead == 1] = 0.0
    # one-object targets in the layout decode() expects (yolo/yolo_v1/yolo.py:133-152): the object's cell carries, in both
    # box slots, the centre's offset inside the cell in cell units, the size and confidence 1, plus a one-hot class
    # (draw order per image: corner, size, label)
    tgt = np.zeros_like(out)
    for b in range(batch):
        xy = rng.uniform(0.05, 0.6, 2)
        wh = rng.uniform(0.05, 0.35, 2)
        lab = int(rng.integers(1, num_classes + 1))
        centre = xy + wh / 2
        col, row = (centre * 7).astype(int)
        slot = np.concatenate([(centre - np.array([col, row]) * (1 / 7)) / (1 / 7), wh, [1.0]])
        tgt[b, row, col, 0:5] = slot
        tgt[b, row, col, 5:10] = slot
        tgt[b, row, col, 10 + lab - 1] = 1
    return out, tgt


# ---------------------------------------------------------------- config 4
def anchor_wh(seed, n):
    rng = np.random.default_rng(seed)
    wh = np.exp(rng.normal(-1.5, 0.6, (n, 2)))
    return np.clip(wh, 1e-3, 1.0).astype(np.float32)


# ---------------------------------------------------------------- config 5
def stress_boxes(seed, n=65536, dense=False):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0.0, 0.95, (n, 2))
    wh = rng.uniform(0.05, 0.3, (n, 2)) if dense else rng.uniform(0.005, 0.05, (n, 2))
    boxes = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
    return boxes, tiefree_scores(seed + 11, n)


# ---------------------------------------------------------------- generic
def random_boxes(seed, n, degenerate=True):
    """Unit-square boxes; with `degenerate` a few inverted / zero-area / duplicate ones."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 0.8, (n, 2))
    wh = rng.uniform(0.01, 0.4, (n, 2))
    b = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)
    if degenerate and n >= 8:
        b[1] = b[0]                       # duplicate
        b[2, 2:] = b[2, :2]               # zero area
        b[3] = b[3, [2, 3, 0, 1]]         # inverted corners
        b[4] = [0, 0, 1, 1]
        b[5] = [0, 0, 1, 1]
    return b
