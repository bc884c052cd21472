"""Trajectory export and accuracy evaluation.

Replaces, for the replay and the tests of this repository,
  * evaluation/conversions.py:5-8   estimation.hyper (`stamp, qx, qy, qz, qw, px, py, pz`, the SIGUSR1 dump of apps/hyperslam/main.cpp:72-79)
                                    -> TUM (`stamp tx ty tz qx qy qz qw`, '%.20e'), column order [0, 5, 6, 7, 1, 2, 3, 4];
  * evaluation/run.py:31-57         the four `evo_ape` / `evo_rpe` calls (`-a -r angle_deg`, `-a -r trans_part`) against the TUM ground truth
                                    of resources/datasets/euroc/sequences/*.txt — evo is not installed in this image, so the metrics are
                                    restated here: timestamp association (max 0.01 s), SE3 Umeyama alignment (`-a`, no scale), absolute pose
                                    error E_i = Q_i^-1 S P_i and relative pose error over consecutive associated pairs (delta = 1 frame), with
                                    the two pose relations the reference asks for (rotation angle in degrees, norm of the translation part).

Host-side numpy only: this is evaluation tooling, not part of the solve path.
"""
from __future__ import annotations

import argparse
import json

import numpy as np

HYPER_TO_TUM = [0, 5, 6, 7, 1, 2, 3, 4]  # conversions.py:7


def convert_hyper_to_tum(src, dst):
    """estimation.hyper (comma separated) -> TUM (space separated, 20 significant digits)."""
    data = np.atleast_2d(np.loadtxt(src, delimiter=","))
    np.savetxt(dst, data[:, HYPER_TO_TUM], fmt="%.20e")
    return len(data)


def write_tum(path, stamps, poses_q_p):
    """poses_q_p: (n, 7) [qx qy qz qw px py pz] (the layout of hs_sample_trajectory)."""
    poses_q_p = np.asarray(poses_q_p, float)
    data = np.column_stack([np.asarray(stamps, float), poses_q_p[:, 4:7], poses_q_p[:, 0:4]])
    np.savetxt(path, data, fmt="%.20e")


def read_tum(path):
    """(stamps (n,), xyz (n, 3), quat_xyzw (n, 4)); '#' comments and both separators accepted."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            rows.append([float(x) for x in line.replace(",", " ").split()])
    a = np.array(rows, float).reshape(-1, 8)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def quat_to_matrix(q):
    q = np.asarray(q, float)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)
    return R


def associate(ref_stamps, est_stamps, max_diff=0.01, offset=0.0):
    """Index pairs (i_ref, i_est) of nearest stamps within max_diff seconds, each sample used once (evo's associate_trajectories)."""
    ref_stamps, est = np.asarray(ref_stamps, float), np.asarray(est_stamps, float) + offset
    pairs, used = [], -1
    idx = np.searchsorted(ref_stamps, est)
    for j, (t, i) in enumerate(zip(est, idx)):
        best = None
        for c in (i - 1, i):
            if 0 <= c < len(ref_stamps) and c > used and abs(ref_stamps[c] - t) <= max_diff and (best is None or abs(ref_stamps[c] - t) < abs(ref_stamps[best] - t)):
                best = c
        if best is not None:
            pairs.append((best, j))
            used = best
    return np.array(pairs, int).reshape(-1, 2)


def umeyama(est_xyz, ref_xyz):
    """Least-squares rigid alignment (Umeyama 1991, no scale): R, t with ref ~ R est + t."""
    est_xyz, ref_xyz = np.asarray(est_xyz, float), np.asarray(ref_xyz, float)
    mu_e, mu_r = est_xyz.mean(0), ref_xyz.mean(0)
    C = (ref_xyz - mu_r).T @ (est_xyz - mu_e) / len(est_xyz)
    U, _, Vt = np.linalg.svd(C)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    return R, mu_r - R @ mu_e


def _stats(e):
    e = np.asarray(e, float)
    return {"rmse": float(np.sqrt(np.mean(e * e))), "mean": float(e.mean()), "median": float(np.median(e)), "std": float(e.std()),
            "min": float(e.min()), "max": float(e.max()), "sse": float(np.sum(e * e)), "n": int(len(e))}


def _relation(E_R, E_t, relation):
    if relation == "trans_part":
        return np.linalg.norm(E_t, axis=-1)
    if relation == "angle_deg":
        c = np.clip((np.trace(E_R, axis1=-2, axis2=-1) - 1.0) / 2.0, -1.0, 1.0)
        return np.degrees(np.arccos(c))
    raise ValueError("relation must be 'trans_part' or 'angle_deg'")


def ape(ref, est, relation="trans_part", align=True, max_diff=0.01, offset=0.0):
    """Absolute pose error (evo_ape tum REF EST -a -r <relation>). ref / est: (stamps, xyz, quat_xyzw)."""
    pairs = associate(ref[0], est[0], max_diff, offset)
    if len(pairs) < 3:
        raise ValueError("fewer than three associated poses")
    Pr, Rr = ref[1][pairs[:, 0]], quat_to_matrix(ref[2][pairs[:, 0]])
    Pe, Re = est[1][pairs[:, 1]], quat_to_matrix(est[2][pairs[:, 1]])
    if align:
        R, t = umeyama(Pe, Pr)
        Pe, Re = Pe @ R.T + t, R @ Re
    # E_i = Q_i^-1 P_i (reference^-1 * aligned estimate)
    E_R = np.swapaxes(Rr, -1, -2) @ Re
    E_t = np.einsum("nji,nj->ni", Rr, Pe - Pr)
    return _stats(_relation(E_R, E_t, relation))


def rpe(ref, est, relation="trans_part", delta=1, align=True, max_diff=0.01, offset=0.0):
    """Relative pose error over associated pairs `delta` frames apart (evo_rpe's default: consecutive frames)."""
    pairs = associate(ref[0], est[0], max_diff, offset)
    if len(pairs) < delta + 2:
        raise ValueError("too few associated poses")
    Pr, Rr = ref[1][pairs[:, 0]], quat_to_matrix(ref[2][pairs[:, 0]])
    Pe, Re = est[1][pairs[:, 1]], quat_to_matrix(est[2][pairs[:, 1]])
    if align:
        R, t = umeyama(Pe, Pr)
        Pe, Re = Pe @ R.T + t, R @ Re

    def rel(P, Rm):  # T_i^-1 T_{i+delta}
        Rt = np.swapaxes(Rm[:-delta], -1, -2)
        return Rt @ Rm[delta:], np.einsum("nij,nj->ni", Rt, P[delta:] - P[:-delta])
    dRr, dtr = rel(Pr, Rr)
    dRe, dte = rel(Pe, Re)
    # E_i = (Q_i^-1 Q_{i+d})^-1 (P_i^-1 P_{i+d})
    E_R = np.swapaxes(dRr, -1, -2) @ dRe
    E_t = np.einsum("nji,nj->ni", dRr, dte - dtr)
    return _stats(_relation(E_R, E_t, relation))


def evaluate(reference_tum, estimation_tum, offset=0.0):
    """The four numbers of evaluation/run.py:31-57."""
    ref, est = read_tum(reference_tum), read_tum(estimation_tum)
    return {"ape_rotation_deg": ape(ref, est, "angle_deg", offset=offset), "ape_translation_m": ape(ref, est, "trans_part", offset=offset),
            "rpe_rotation_deg": rpe(ref, est, "angle_deg", offset=offset), "rpe_translation_m": rpe(ref, est, "trans_part", offset=offset)}


def main():
    ap = argparse.ArgumentParser(description="estimation.hyper -> TUM and APE / RPE against a TUM ground truth")
    ap.add_argument("estimation", help="estimation.hyper (comma separated) or a TUM file")
    ap.add_argument("reference", nargs="?", help="TUM ground truth (resources/datasets/euroc/sequences/*.txt)")
    ap.add_argument("--tum-out", help="write the converted estimation here")
    ap.add_argument("--offset", type=float, default=0.0)
    args = ap.parse_args()
    est_path = args.estimation
    with open(est_path) as f:
        first = next((l for l in f if l.strip() and not l.startswith("#")), "")
    if "," in first:
        est_path = args.tum_out or args.estimation + ".tum"
        convert_hyper_to_tum(args.estimation, est_path)
    if args.reference:
        print(json.dumps(evaluate(args.reference, est_path, args.offset)))


if __name__ == "__main__":
    main()
