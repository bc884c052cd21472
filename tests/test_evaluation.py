"""TUM export (evaluation/conversions.py:5-8) and the APE / RPE metrics of evaluation/run.py:31-57 (host tooling, CPU only)."""
import os

import numpy as np

from hyperslam_amd import evaluation as ev

HERE = os.path.dirname(os.path.abspath(__file__))
GT = os.path.join(HERE, "golden", "euroc_MH_02_easy_head.tum")  # first 400 poses of resources/datasets/euroc/sequences/MH_02_easy.txt


def rot(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def mat_to_quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def test_hyper_to_tum_column_order(tmp_path):
    """estimation.hyper rows are `stamp, qx, qy, qz, qw, px, py, pz` (main.cpp:72-79); TUM wants `stamp tx ty tz qx qy qz qw`."""
    src, dst = tmp_path / "estimation.hyper", tmp_path / "estimation.tum"
    rows = np.array([[1.5, 0.1, 0.2, 0.3, 0.9, 10.0, 20.0, 30.0], [2.5, 0.0, 0.0, 0.0, 1.0, -1.0, -2.0, -3.0]])
    np.savetxt(src, rows, delimiter=", ", fmt="%.20e")
    assert ev.convert_hyper_to_tum(src, dst) == 2
    t, p, q = ev.read_tum(dst)
    assert np.array_equal(t, rows[:, 0]) and np.array_equal(p, rows[:, 5:8]) and np.array_equal(q, rows[:, 1:5])
    text = open(dst).read().split()
    assert len(text) == 16 and all("e" in x and len(x.split("e")[0].replace("-", "").replace(".", "")) == 21 for x in text)  # '%.20e'
    ev.write_tum(tmp_path / "w.tum", rows[:, 0], rows[:, 1:8])
    assert open(tmp_path / "w.tum").read() == open(dst).read()


def test_reads_the_reference_ground_truth_format():
    t, p, q = ev.read_tum(GT)
    assert len(t) == 400 and t[0] == 1403636859.536666393280 and np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-4)
    assert np.allclose(p[0], [4.62115, -1.837605, 0.739627])


def test_ape_rpe_on_a_rigidly_moved_copy():
    """An estimate that is the ground truth in another world frame has zero error after alignment (`-a`); without alignment it has not."""
    t, p, q = ev.read_tum(GT)
    Rw, tw = rot([0.3, -1, 0.5], 0.8), np.array([5.0, -2.0, 1.0])
    R = ev.quat_to_matrix(q)
    est = (t + 0.002, p @ Rw.T + tw, np.array([mat_to_quat(Rw @ Ri) for Ri in R]))  # stamps 2 ms off: still associated
    ref = (t, p, q)
    for rel in ("trans_part", "angle_deg"):
        assert ev.ape(ref, est, rel)["rmse"] < 1e-5  # (angle_deg: arccos near 1 keeps half the digits)
        assert ev.rpe(ref, est, rel)["rmse"] < 1e-5
    assert ev.ape(ref, est, "trans_part", align=False)["rmse"] > 1.0
    assert ev.ape(ref, est, "trans_part")["n"] == 400


def test_ape_measures_known_errors():
    t, p, q = ev.read_tum(GT)
    R = ev.quat_to_matrix(q)
    # constant body-frame rotation error of 2 degrees: APE rotation = 2 deg everywhere, translation untouched
    Rerr = rot([0, 0, 1], np.radians(2.0))
    est = (t, p, np.array([mat_to_quat(Ri @ Rerr) for Ri in R]))
    a = ev.ape((t, p, q), est, "angle_deg")
    assert abs(a["mean"] - 2.0) < 1e-3 and a["std"] < 1e-3
    assert ev.ape((t, p, q), est, "trans_part")["rmse"] < 1e-9
    assert ev.rpe((t, p, q), est, "angle_deg")["rmse"] < 1e-2  # relative rotations Rerr^T dR Rerr vs dR: second order in (frame motion x error)
    # zero-mean position noise: APE translation rmse ~ sigma sqrt(3) (alignment absorbs almost nothing of it)
    rng = np.random.default_rng(3)
    noisy = (t, p + rng.normal(scale=0.05, size=p.shape), q)
    e = ev.ape((t, p, q), noisy, "trans_part")["rmse"]
    assert 0.9 * 0.05 * np.sqrt(3) < e < 1.1 * 0.05 * np.sqrt(3)
    # samples farther than 10 ms from every reference stamp are not associated
    assert len(ev.associate(t[::10], t[::10] + 0.02)) == 0 and len(ev.associate(t[::10], t[::10] + 0.004)) == 40


def test_evaluate_round_trip(tmp_path):
    t, p, q = ev.read_tum(GT)
    hyper = np.column_stack([t, q, p])
    np.savetxt(tmp_path / "estimation.hyper", hyper, delimiter=", ", fmt="%.20e")
    ev.convert_hyper_to_tum(tmp_path / "estimation.hyper", tmp_path / "estimation.tum")
    out = ev.evaluate(GT, tmp_path / "estimation.tum")
    assert set(out) == {"ape_rotation_deg", "ape_translation_m", "rpe_rotation_deg", "rpe_translation_m"}
    assert all(v["rmse"] < 1e-5 and v["n"] >= 399 for v in out.values())  # angle_deg: arccos near 1 costs half the digits
