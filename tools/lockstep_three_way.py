#!/usr/bin/env python3
"""Three-way lock-step comparison: oracle in double (master), oracle in long double and the HIP library, from identical tables.

TEST TOOLING (runs on the GPU box; uses oracle/ as the checker only). The lock-step harness (tests/harness/replay_lockstep)
is run twice with the same master — the double oracle, whose results drive the replay, so both runs see the very same windows — once
with libhyperslam_hip.so and once with oracle/liboracle_ld.so (capi_ld.cpp: the same restatement compiled in 80-bit long double) as
the shadow. HS_LOCKSTEP_DUMP gives the raw end points of every call; this script prints, per gauge-fixed call,

    |hip - double| , |long double - double| , |hip - long double|      (max-norm relative, landmarks and control points)

If the first two agree and the third is orders of magnitude smaller, the distance between the HIP library and the double oracle on
that window is the DOUBLE ORACLE's rounding error (an ill-conditioned window amplifies the rounding of its normal equations), not
the product's.

    usage: python tools/lockstep_three_way.py [seconds=3.6] [imu=0] [order=4] [--out profiles/rNN_three_way.txt]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_dump(path):
    out = {}
    raw = open(path, "rb").read()
    o = 0
    while o < len(raw):
        call, ncp, nlm = np.frombuffer(raw, np.int32, 3, o)
        o += 12
        arr = []
        for n in (ncp, nlm, ncp, nlm):
            arr.append(np.frombuffer(raw, np.float64, n, o).copy())
            o += 8 * n
        out[int(call)] = arr  # shadow cp, shadow lm, master cp, master lm
    return out


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)) if a.size else 0.0


def run(shadow, args, tmp, tag):
    env = dict(os.environ, HS_LOCKSTEP_DUMP=os.path.join(tmp, tag + ".bin"))
    prefix = "hsl_" if os.path.basename(shadow) == "liboracle_ld.so" else "hs_"
    res = subprocess.run([os.path.join(ROOT, "tests/harness/replay_lockstep"), shadow] + args + [prefix], env=env, capture_output=True, text=True, check=True)
    rows = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    return rows, read_dump(env["HS_LOCKSTEP_DUMP"])


def main():
    argv = [a for a in sys.argv[1:]]
    out = None
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        del argv[i : i + 2]
    hip_lib = os.path.join(ROOT, "hyperslam_amd/libhyperslam_hip.so")
    if "--hip" in argv:  # (another library exporting hs_*: the script's own dry run on a machine without a GPU)
        i = argv.index("--hip")
        hip_lib = argv[i + 1]
        del argv[i : i + 2]
    args = (argv + ["3.6", "0", "4"][len(argv) :])[:3]
    with tempfile.TemporaryDirectory() as tmp:
        rows_h, hip = run(hip_lib, args, tmp, "hip")
        rows_l, ld = run(os.path.join(ROOT, "oracle/liboracle_ld.so"), args, tmp, "ld")
    lines = ["# lock-step three-way, replay %s s, imu %s, order %s: max-norm relative distances of the end points of every gauge-fixed optimize()" % tuple(args),
             "# d = oracle in double (master), ld = oracle in 80-bit long double, hip = libhyperslam_hip.so; S = reduced system of the first iteration",
             "%5s %6s %9s %9s | %-32s | %-32s" % ("call", "lm", "S hip-d", "S ld-d", "landmarks  hip-d    ld-d     hip-ld", "ctrl pts   hip-d    ld-d     hip-ld")]
    worst = [0.0, 0.0, 0.0]
    for rh, rl in zip(rows_h, rows_l):
        if rh.get("summary") or not rh["gauge_fixed"]:
            continue
        c = rh["call"]
        hc, hl, dc, dl = hip[c]
        lc, ll, dc2, dl2 = ld[c]
        assert np.array_equal(dc, dc2) and np.array_equal(dl, dl2), "the two runs did not see the same master"
        lm = (rel(hl, dl), rel(ll, dl), rel(hl, ll))
        cp = (rel(hc, dc), rel(lc, dc), rel(hc, lc))
        worst = [max(w, a, b) for w, a, b in zip(worst, lm, cp)]
        lines.append("%5d %6d %9.2e %9.2e |          %9.2e %9.2e %9.2e |          %9.2e %9.2e %9.2e" % (c, rh["landmarks"], rh["S_rel"], rl["S_rel"], *lm, *cp))
    lines.append("# worst over the gauge-fixed calls: hip-d %.3e   ld-d %.3e   hip-ld %.3e" % tuple(worst))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
