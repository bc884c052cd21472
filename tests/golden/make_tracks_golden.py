#!/usr/bin/env python3
"""Generates tests/golden/tracks.json — 100-digit golden vectors for the front half of AbstractOptimizer::process(VisualTracks)
(internal/hyper/optimizers/abstract.cpp:197-223,250-255; hs_process_tracks): pixel -> unit bearing in both cameras of a stereo
pair and the triangulation of each pair into the world frame through the spline pose at the track stamp.

Independent of the oracle and of the kernels: the radial-tangential undistortion is the EXACT root of distort(n) = n_d (mpmath
findroot; the libraries iterate a fixed point 20 times), the triangulation solves the 2 x 2 normal equations of the
closest-points problem of the two rays and takes the midpoint, the spline pose comes from make_golden.spline_pose (Cox-de Boor
basis, quaternion exp / log). Inputs are rounded to doubles first.
Run:  python tests/golden/make_tracks_golden.py   (seconds)
"""
import json
import os
import sys

import mpmath as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import SplitMix64, qconj, qexp, qmul, qnorm, qrot, spline_pose, tofloat  # noqa: E402  (sets mp.dps = 100)

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hyperslam_amd.synthetic import EUROC_CAM_DISTORTION, EUROC_CAM_INTRINSICS, EUROC_CAM_T_BS  # noqa: E402  (constants only)


def distort(x, y, d):
    k1, k2, p1, p2 = d
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2
    return x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y


def pixel_to_bearing(px, intr, dist):
    cx, cy, fx, fy = intr
    xd, yd = (px[0] - cx) / fx, (px[1] - cy) / fy
    sol = mp.findroot(lambda x, y: [distort(x, y, dist)[0] - xd, distort(x, y, dist)[1] - yd], (xd, yd), tol=mp.mpf(10) ** -80)
    x, y = sol[0], sol[1]
    n = mp.sqrt(x * x + y * y + 1)
    return [x / n, y / n, 1 / n]


def main():
    rng = SplitMix64(0x545241434B)
    k, dt = 4, mp.mpf("0.1")

    def u(lo, hi):
        return mp.mpf(float(rng.uniform(lo=lo, hi=hi)))

    q = qnorm([u(-1, 1) for _ in range(4)])
    cps, t_first = [], u(0, 3)
    for j in range(k):
        q = qmul(q, qexp([u(-0.2, 0.2) for _ in range(3)]))
        cps.append(q + [u(-1, 1) for _ in range(3)] + [t_first + dt * j])
    cps = [[mp.mpf(float(v)) for v in cp] for cp in cps]  # exactly representable inputs
    stamp = mp.mpf(float(cps[1][7] + dt * u(0.1, 0.9)))
    cams = [[mp.mpf(float(v)) for v in EUROC_CAM_T_BS[c]] for c in range(2)]
    intr = [[mp.mpf(float(v)) for v in EUROC_CAM_INTRINSICS[c]] for c in range(2)]
    dist = [[mp.mpf(float(v)) for v in EUROC_CAM_DISTORTION[c]] for c in range(2)]
    q_wb, p_wb = spline_pose(cps, k, stamp)

    px0, px1, b0s, b1s, pws = [], [], [], [], []
    for _ in range(16):
        # a point in front of camera 0 -> its (distorted) pixels in both cameras, rounded to doubles -> golden outputs of those pixels
        ps0 = [u(-1.5, 1.5), u(-1.0, 1.0), u(2.0, 8.0)]
        pb = [a + b for a, b in zip(qrot(cams[0][:4], ps0), cams[0][4:7])]
        ps1 = qrot(qconj(cams[1][:4]), [a - b for a, b in zip(pb, cams[1][4:7])])
        pix = []
        for c, ps in ((0, ps0), (1, ps1)):
            xd, yd = distort(ps[0] / ps[2], ps[1] / ps[2], dist[c])
            pix.append([mp.mpf(float(intr[c][0] + intr[c][2] * xd + u(-0.3, 0.3))), mp.mpf(float(intr[c][1] + intr[c][3] * yd + u(-0.3, 0.3)))])
        b0, b1 = pixel_to_bearing(pix[0], intr[0], dist[0]), pixel_to_bearing(pix[1], intr[1], dist[1])
        # rays in the body frame: o_c + s_c d_c; closest points -> midpoint -> world
        o0, o1 = cams[0][4:7], cams[1][4:7]
        d0, d1 = qrot(cams[0][:4], b0), qrot(cams[1][:4], b1)
        dot = lambda a, b: sum(x * y for x, y in zip(a, b))  # noqa: E731
        r = [a - b for a, b in zip(o1, o0)]
        A = mp.matrix([[dot(d0, d0), -dot(d0, d1)], [-dot(d0, d1), dot(d1, d1)]])
        s = mp.lu_solve(A, mp.matrix([dot(d0, r), -dot(d1, r)]))
        mid = [(o0[i] + s[0] * d0[i] + o1[i] + s[1] * d1[i]) / 2 for i in range(3)]
        pw = [a + b for a, b in zip(qrot(q_wb, mid), p_wb)]
        px0.append(tofloat(pix[0])), px1.append(tofloat(pix[1])), b0s.append(tofloat(b0)), b1s.append(tofloat(b1)), pws.append(tofloat(pw))
    out = {"generator": "tests/golden/make_tracks_golden.py (mpmath, 100 digits)", "k": k, "cps": tofloat(cps), "stamp": float(stamp),
           "cam_T_bs": tofloat(cams), "intrinsics": tofloat(intr), "distortion": tofloat(dist),
           "pixels0": px0, "pixels1": px1, "bearings0": b0s, "bearings1": b1s, "positions_w": pws}
    with open(os.path.join(HERE, "tracks.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", len(px0), "tracks")


if __name__ == "__main__":
    main()
