"""Evaluation row: pose chaining and the KITTI odometry metrics against golden vectors produced by the
reference's own rslo/utils/geometric.py + kitti_evaluation.py (tests/golden/make_golden_eval.py)."""
import os

import numpy as np
import pytest

import rslo_amd  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "eval_side.npz"))


def test_odom_to_abs_pose_and_matrix_round_trip(g):
    from rslo.utils import geometric as G
    np.testing.assert_allclose(G.odom_to_abs_pose(g["gt_odo"]), g["gt_abs"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(G.odom_to_abs_pose(g["pred_odo"]), g["pred_abs"], rtol=1e-12, atol=1e-12)
    RT = np.stack([G.tq_to_RT(p) for p in g["gt_abs"][::100]])
    np.testing.assert_allclose(RT, g["RT"], rtol=1e-12, atol=1e-12)
    back = np.stack([G.RT_to_tq(m) for m in RT])
    np.testing.assert_allclose(back, g["back"], rtol=1e-9, atol=1e-9)
    assert G.tq_to_RT(g["gt_abs"][3], expand=True).shape == (4, 4)
    with pytest.raises(ValueError):
        G.expand_rigid_transformation(np.zeros((3, 3)))


def test_kitti_segment_metrics_match_reference(g):
    from rslo.utils.geometric import tq_to_RT
    from rslo.utils.kitti_evaluation import kittiOdomEval
    ev = kittiOdomEval()
    seq = ev.calcSequenceErrors(g["pred_abs"], g["gt_abs"])
    np.testing.assert_allclose(np.array(seq), g["seq"], rtol=1e-9, atol=1e-12)
    assert abs(ev.distance - float(g["distance"])) < 1e-9
    avg = ev.computeSegmentErr(seq)
    np.testing.assert_allclose(np.array([[k, *v] for k, v in sorted(avg.items())]), g["seg"], rtol=1e-9)
    np.testing.assert_allclose(np.array(ev.computeOverallErr(seq)), g["overall"], rtol=1e-9)
    np.testing.assert_allclose(np.array(ev.computeSegmentAvgErr(avg)), g["seg_avg"], rtol=1e-9)
    np.testing.assert_allclose(np.array(ev.computeSegmentRMSEErr(avg)), g["seg_rmse"], rtol=1e-9)
    sp = ev.computeSpeedErr(seq)
    got = np.array([[k, *(v if v else [np.nan, np.nan])] for k, v in sorted(sp.items())])
    np.testing.assert_allclose(got, g["speed"], rtol=1e-9, equal_nan=True)
    np.testing.assert_allclose(np.array(ev.calcOdomErrors(g["pred_odo"][:200], g["gt_odo"][:200])), g["odo_err"],
                               rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(ev.toCameraCoord(tq_to_RT(g["gt_abs"][500], expand=True)), g["cam"], rtol=1e-12, atol=1e-12)
    # known answer: a perfect prediction has zero error everywhere
    perfect = ev.calcSequenceErrors(g["gt_abs"], g["gt_abs"])
    assert len(perfect) == len(seq) and max(e[2] for e in perfect) < 1e-12 and max(e[1] for e in perfect) < 1e-7
