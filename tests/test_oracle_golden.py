"""Pins the CPU oracle (oracle/oracle.c) against vectors produced by executing the
reference's own function bodies (oracle/gen_golden.py -> tests/golden/*.npz)."""
import numpy as np
import pytest


def test_iou_tensor(golden, oracle):
    g = golden("iou_tensor")
    assert np.array_equal(oracle.iou_elementwise(g["a"], g["b"], 1e-7), g["iou_eps1e7"])
    assert np.array_equal(oracle.iou_elementwise(g["a"], g["b"], 1e-6), g["iou_eps1e6"])
    assert np.array_equal(oracle.iou_elementwise(g["a"].astype(np.float64), g["b"].astype(np.float64), 1e-7),
                          g["iou_f64"])
    assert np.array_equal(oracle.iou_elementwise(g["a_nan"], g["b_nan"]), g["iou_nan"], equal_nan=True)
    # hand-derivable known answers
    one = np.array([[0, 0, 1, 1]], np.float32)
    assert oracle.iou_elementwise(one, one)[0] == np.float32(1.0) / (np.float32(1.0) + np.float32(1e-7))
    assert oracle.iou_elementwise(one, one + 5)[0] == 0.0
    pw = oracle.iou_pairwise(g["a"][:40], g["b"][:50])
    for i in range(40):
        assert np.array_equal(pw[i], oracle.iou_elementwise(np.repeat(g["a"][i:i + 1], 50, 0), g["b"][:50]))


def test_iou_scalar(golden, oracle):
    g = golden("iou_scalar")
    got = np.array([oracle.iou_scalar(r, g["grt"]) for r in g["rois"]])
    assert np.array_equal(got, g["iou"])
    assert oracle.iou_scalar([0, 0, 10, 10], [10.0, 0, 20, 10]) == float(g["touch"]) == 0.0
    for bb, raises in zip(g["bad_boxes"], g["bad_raises"]):
        assert raises == 1
        with pytest.raises(AssertionError):
            oracle.iou_scalar(bb, g["grt"])


def test_nms(golden, oracle):
    g = golden("nms")
    names = sorted(k[:-5] for k in g.files if k.endswith("_keep"))
    assert len(names) >= 20
    for name in names:
        keep = oracle.nms(g[name + "_boxes"], g[name + "_scores"], float(g[name + "_thr"]))
        assert np.array_equal(keep, g[name + "_keep"]), name
    # the quirk probes of SURVEY §8(a) N1
    assert g["one_keep"].size == 0 and list(g["two_disjoint_keep"]) == [0]
    assert list(g["dup_then_disjoint_keep"]) == [0, 2]
    t = golden("nms_frcnn_top")
    r, s = oracle.nms_frcnn_top(list(t["rects"]), list(t["scores"]))
    assert np.array_equal(r, t["best_rect"]) and s == float(t["best_score"])


def test_decode_v1(golden, oracle):
    g = golden("decode_v1")
    for tag in "abc":
        ct, pt = g[tag + "_thr"]
        for i in range(g["output"].shape[0]):
            box, lab, conf, score, tb, tl = oracle.yolo1_decode(g["output"][i], g["target"][i], ct, pt)
            assert np.array_equal(box, g[f"{tag}_box_{i}"]), (tag, i)
            assert np.array_equal(lab, g[f"{tag}_label_{i}"])
            assert np.array_equal(conf, g[f"{tag}_conf_{i}"])
            assert np.array_equal(score, g[f"{tag}_score_{i}"])
            assert np.array_equal(tb, g[f"{tag}_tbox_{i}"])
            assert tl == int(g[f"{tag}_tlabel_{i}"])
    box, *_ = oracle.yolo1_decode(g["low_output"], g["target"][0], 0.2, 0.1)
    assert box.shape == (0, 4)


def test_evaluate_and_ap(golden, oracle):
    g = golden("evaluate_v1")
    det, tgt = oracle.evaluate_batch(g["output"], g["target"], 5, prob_threshold=0.01, nms_threshold=0.3)
    for c in range(5):
        assert np.array_equal(np.array(det[c], np.float64).reshape(-1, 6), g[f"detects_{c}"]), c
        assert np.array_equal(np.array(tgt[c], np.float64).reshape(-1, 6), g[f"targets_{c}"]), c
    assert np.array_equal(np.array(oracle.average_precision(det, tgt, 0.05)), g["aps_iou005"])      # float64, bit for bit
    assert np.array_equal(np.array(oracle.average_precision(det, tgt, 0.5)), g["aps_iou05"])


def test_ap_known_answers(oracle):
    """hand-checkable cases of yolo.py:306-367: perfect detections, a duplicate counted as FP, a class without targets."""
    box = [0.1, 0.1, 0.5, 0.5]
    tgt = {0: [[0, 1] + box, [1, 1] + box], 1: []}
    det = {0: [[0, 0.9] + box, [1, 0.8] + box], 1: []}
    assert oracle.average_precision(det, tgt, 0.5) == [1.0, 0.0, 0.0]                 # the empty class appends 0.0 twice
    det = {0: [[0, 0.9] + box, [0, 0.8] + box, [1, 0.7] + box], 1: []}                # second hit on image 0 is a FP
    ap0 = oracle.average_precision(det, tgt, 0.5)[0]
    assert ap0 == 0.5 * 1.0 + 0.5 * (2.0 / 3.0)
    det = {0: [], 1: [[0, 0.9] + box]}                                                # detections but no targets: 0/0 recall
    out = oracle.average_precision(det, tgt, 0.5)
    assert out[0] == 0.0 and out[1] == 0.0 and np.isnan(out[2])


def test_roipool(golden, oracle):
    g = golden("roipool")
    out, arg = oracle.roipool_fwd(g["img"], g["rois"], g["idx"], 7, "max")
    assert np.array_equal(out, g["max_out"])
    assert np.array_equal(arg, g["max_argmax"])
    avg, _ = oracle.roipool_fwd(g["img"], g["rois"], g["idx"], 7, "avg")
    np.testing.assert_allclose(avg, g["avg_out"], rtol=1e-6, atol=1e-7)
    gin = oracle.roipool_bwd(g["grad_out"], arg, g["img"].shape, g["rois"], g["idx"], 7, "max")
    np.testing.assert_allclose(gin, g["max_grad_in"], rtol=1e-5, atol=1e-5)
    gin = oracle.roipool_bwd(g["grad_out"], None, g["img"].shape, g["rois"], g["idx"], 7, "avg")
    np.testing.assert_allclose(gin, g["avg_grad_in"], rtol=1e-5, atol=1e-5)
    assert int(g["empty_raises"]) == 1
    with pytest.raises(RuntimeError):
        oracle.roipool_fwd(g["img"], g["empty_rois"], g["idx"][:2])


def test_roi_window_is_python_slicing(oracle):
    """rcnn/fast_rcnn.py:366-374: `np.floor(x1 * w).astype(int)` ... then `img[roi_idx[i], :, y1:y2, x1:x2]`.  The oracle's window
    (orc_roi_window) against numpy's own conversion and python's own slice normalisation (slice.indices) on RoIs far outside
    [0, 1], reversed, on cell boundaries, a hair below zero (start -1 = the last column), with a NEGATIVE STOP (wraps too),
    NaN and infinities (INT64_MIN from astype(int) on this platform: bound 0)."""
    import ctypes as C
    import warnings
    lib = oracle.lib()
    rng = np.random.default_rng(12)
    special = np.array([-1e-18, 1e-18, 1 - 1e-16, 1 + 1e-15, 0.0, 1.0, -1.0, 2.0, 1e12, -1e12, 1e300, -1e300, np.nan, np.inf, -np.inf, -0.4, -0.02])
    checked = nonempty = 0
    for _ in range(60):
        H, W = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        n = 200
        kind = rng.integers(0, 4, (n, 4))
        grid = np.stack([rng.integers(-2, W + 3, n) / W, rng.integers(-2, H + 3, n) / H, rng.integers(-2, W + 3, n) / W, rng.integers(-2, H + 3, n) / H], 1)
        rois = np.where(kind == 0, rng.uniform(-1.5, 2.5, (n, 4)), np.where(kind == 1, grid, np.where(kind == 2, rng.uniform(0, 1, (n, 4)),
                        special[rng.integers(0, len(special), (n, 4))])))
        rois = np.ascontiguousarray(rois, np.float64)
        with warnings.catch_warnings(), np.errstate(invalid="ignore"):
            warnings.simplefilter("ignore")
            x1 = np.floor(rois[:, 0] * W).astype(int); x2 = np.ceil(rois[:, 2] * W).astype(int)      # the reference's lines
            y1 = np.floor(rois[:, 1] * H).astype(int); y2 = np.ceil(rois[:, 3] * H).astype(int)
        for r in range(n):
            xs, xe, _ = slice(int(x1[r]), int(x2[r])).indices(W)
            ys, ye, _ = slice(int(y1[r]), int(y2[r])).indices(H)
            a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            rc = lib.orc_roi_window(rois[r].ctypes.data_as(C.POINTER(C.c_double)), C.c_int(H), C.c_int(W), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
            empty = xe <= xs or ye <= ys
            assert (rc != 0) == empty, (H, W, rois[r])
            if not empty:
                assert (a.value, b.value, c.value, d.value) == (xs, ys, xe, ye), (H, W, rois[r])
                nonempty += 1
            checked += 1
    assert checked == 12000 and nonempty > 1500


def test_roipool_other_output_sizes(golden, oracle):
    """slowroipool(output_size) with sizes other than (7,7) -- the reference's class as written (avg) and with
    AdaptiveMaxPool2d substituted, values, argmax and autograd gradients generated from the reference."""
    g = golden("roipool_sizes")
    for tag in ("s3x5", "s14", "s1", "s2x9"):
        P = tuple(int(v) for v in g[f"{tag}_size"])
        out, arg = oracle.roipool_fwd(g["img"], g["rois"], g["idx"], P, "max")
        assert np.array_equal(out, g[f"{tag}_max_out"]) and np.array_equal(arg, g[f"{tag}_max_argmax"]), tag
        avg, _ = oracle.roipool_fwd(g["img"], g["rois"], g["idx"], P, "avg")
        np.testing.assert_allclose(avg, g[f"{tag}_avg_out"], rtol=1e-6, atol=1e-6)
        gin = oracle.roipool_bwd(g[f"{tag}_grad_out"], arg, g["img"].shape, g["rois"], g["idx"], P, "max")
        np.testing.assert_allclose(gin, g[f"{tag}_max_grad_in"], rtol=1e-5, atol=1e-5)
        gin = oracle.roipool_bwd(g[f"{tag}_grad_out"], None, g["img"].shape, g["rois"], g["idx"], P, "avg")
        np.testing.assert_allclose(gin, g[f"{tag}_avg_grad_in"], rtol=1e-5, atol=1e-5)


def test_yolo2_get_point(golden, oracle):
    g = golden("yolo2_get_point")
    pts, enc = oracle.yolo2_get_point(g["bbox"])
    assert np.array_equal(pts, g["points"])
    assert np.array_equal(enc, g["enc"])
    # decode_v2 is the inverse (parity unpinned: round trip only)
    xyxy = oracle.yolo2_decode_cells(pts, enc)
    want = np.concatenate([g["bbox"][:, :2], g["bbox"][:, :2] + g["bbox"][:, 2:]], 1)
    np.testing.assert_allclose(xyxy, want, rtol=0, atol=2e-6)


def test_rcnn_boxreg(golden, oracle):
    g = golden("rcnn_boxreg")
    np.testing.assert_allclose(oracle.rcnn_calcul(g["positive"], g["grt"]), g["t_target"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(oracle.rcnn_decode(g["p_rect"], g["tset"], 0).reshape(g["g_rect_rcnn"].shape),
                               g["g_rect_rcnn"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(oracle.rcnn_decode(g["p_rect"], g["tset"], 1).reshape(g["g_rect_frcnn"].shape),
                               g["g_rect_frcnn"], rtol=1e-6, atol=1e-4)


def test_do_ss_quota(golden, oracle):
    g = golden("do_ss_quota")
    rois, grt, labels = oracle.label_proposals(g["rects"], g["bbox"], int(g["label"]), int(g["min_num"]), 0.1, 0.1, 2)
    assert np.array_equal(rois, g["rcnn_rois"]) and np.array_equal(labels, g["rcnn_labels"])
    assert np.array_equal(grt, g["rcnn_grt"])
    rois, grt, labels = oracle.label_proposals(g["rects"], g["bbox"], int(g["label"]), int(g["min_num"]), 0.5, 0.1, 1,
                                               require_full=True)
    assert np.array_equal(rois, g["frcnn_rois"]) and np.array_equal(labels, g["frcnn_labels"])
    rois, _, labels = oracle.label_proposals(g["rects"][:30], g["bbox"], 3, 50, 0.5, 0.1, 1, require_full=True)
    assert rois.shape == (0, 4) and labels.size == 0


def test_kmeans_spec_selfconsistency(oracle):
    """1-IoU k-means is NOT in the reference (parity unpinned): only the spec's own invariants."""
    from tools import workloads as wl
    wh = wl.anchor_wh(0, 20000)
    cent, labels, n_iter, inertia = oracle.kmeans_fit(wh, 5, 42)
    assert cent.shape == (5, 2) and labels.shape == (20000,) and 1 <= n_iter <= 300
    # converged: one more step changes nothing
    lab8 = labels.astype(np.uint8)
    acc = oracle.kmeans_step(wh, cent, lab8)
    assert acc[15] == 0
    # sharded accumulation is bit-identical to the single pass (integer sums)
    acc2 = np.zeros(17, np.int64)
    l2 = labels.astype(np.uint8)
    for s in range(4):
        oracle.kmeans_step(wh[s * 5000:(s + 1) * 5000], cent, l2[s * 5000:(s + 1) * 5000], acc2)
    assert np.array_equal(acc, acc2)
    assert np.array_equal(oracle.kmeans_update(acc, cent), cent)


def test_cluster_demo_api_shape(golden):
    g = golden("cluster_demo")
    assert g["labels"].shape == (100,) and g["centers"].shape == (5, 2)
    assert set(np.unique(g["labels"])) == set(range(5))


def test_yolo1_loss_oracle_vs_reference(golden, oracle):
    """Loss.forward + autograd of the reference (modules.py:234-306) run on CPU by oracle/gen_golden.py.  The loss is a
    sum of O(1e3) fp32 squares: the oracle sums in double, so it is compared with the reference's float64 run at 1e-6
    relative (north_star's fp32 tolerance) and with its fp32 run at the fp32 summation noise; gradients at 1e-5/1e-6."""
    g = golden("yolo1_loss")
    for name in ("b16c5", "b4c20", "b1c5"):
        loss, comp, grad = oracle.yolo1_loss(g[name + "_modely"], g[name + "_targety"], quirk=True)
        assert abs(loss - g[name + "_loss_f64"]) <= 1e-6 * abs(g[name + "_loss_f64"]), name
        assert abs(loss - g[name + "_loss"]) <= 4e-6 * abs(g[name + "_loss"]), name
        np.testing.assert_allclose(grad, g[name + "_grad"], rtol=1e-5, atol=1e-6, err_msg=name)
        n = g[name + "_modely"].shape[0]
        assert abs((5 * (comp[0] + comp[1]) + comp[2] + 0.5 * comp[3] + comp[4]) / n - loss) <= 1e-6 * loss


def test_yolo1_encode_restatement_vs_reference(golden, oracle):
    """oracle.yolo1_encode (the NumPy restatement) against the reference's encode, and the synthetic one-object targets of
    tools/workloads.yolo1_batch against it."""
    from tools import workloads as wl
    g = golden("yolo1_encode")
    for k in range(len(g["bb"])):
        C = 5 if k < 30 else 20
        want = g["target_c5"][k] if k < 30 else g["target_c20"][k - 30]
        assert np.array_equal(oracle.yolo1_encode(g["bb"][k], int(g["label"][k]), C), want), k
    # the one-object targets workloads.yolo1_batch builds for bench.py and the decode tests are what encode would build
    rng = np.random.default_rng(77)
    _, tgt = wl.yolo1_batch(77, batch=16, num_classes=5)
    rng.uniform(0, 1, (16, 7, 7, 15))                             # (the generator draws the predictions first)
    for b in range(16):
        xy = rng.uniform(0.05, 0.6, 2); wh = rng.uniform(0.05, 0.35, 2); lab = int(rng.integers(1, 6))
        assert np.array_equal(tgt[b], oracle.yolo1_encode(np.concatenate([xy, wh]), lab, 5).astype(np.float32)), b
