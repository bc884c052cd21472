"""Host-side structure build (hyperslam_amd/csrc/host_structure.hpp: the sort orders every kernel relies on, rebuilt in front of every
optimize() of a sliding window) against an independent numpy restatement of its rules, on CPU: landmarks in device order = observed ones by the
first control point they touch (stable), unobserved last; residuals landmark-major (stable: the caller's order within a landmark, pixel table
before bearing table); record slots segment-major (stable over the landmark-major order). `tests/test_gpu_parity.py::test_structure_bit_exact`
checks the same tables against the oracle through the C ABI on the GPU."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HS_PIXEL, HS_BEARING = 0, 1


@pytest.fixture(scope="module")
def dumper(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("structure") / "structure_dump")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "hyperslam_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "helpers", "structure_dump.cpp")], stderr=subprocess.DEVNULL)
    return exe


def run(exe, tmp_path, k, n_cp, n_lm, t0, dt, px_stamp, px_lm, br_stamp, br_lm, geometry=()):
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.txt")
    with open(src, "wb") as f:
        f.write(struct.pack("=5i2d", k, n_cp, n_lm, len(px_stamp), len(br_stamp), t0, dt))
        for a, t in ((px_stamp, "<f8"), (br_stamp, "<f8"), (px_lm, "<i4"), (br_lm, "<i4")):
            f.write(np.asarray(a, dtype=t).tobytes())
    subprocess.check_call([exe, src, dst] + [str(g) for g in geometry])
    out = {}
    for line in open(dst):
        parts = line.split()
        if parts[0] == "error":
            return {"error": " ".join(parts[1:])}
        out[parts[0]] = np.array(parts[2:], dtype=np.int64)
        assert len(out[parts[0]]) == int(parts[1])
    return out


def expected(k, n_cp, n_lm, t0, dt, px_stamp, px_lm, br_stamp, br_lm):
    stamp = np.concatenate([px_stamp, br_stamp])
    lm = np.concatenate([px_lm, br_lm]).astype(np.int64)
    typ = np.concatenate([np.full(len(px_stamp), HS_PIXEL), np.full(len(br_stamp), HS_BEARING)])
    idx = np.concatenate([np.arange(len(px_stamp)), np.arange(len(br_stamp))])
    first = (np.floor((stamp - t0) / dt) - (k - 1) // 2).astype(np.int64)  # abstract.cpp:89 (uniform basis)
    cf = np.full(n_lm, n_cp, dtype=np.int64)
    cl = np.full(n_lm, -1, dtype=np.int64)
    np.minimum.at(cf, lm, first)
    np.maximum.at(cl, lm, first + k - 1)
    table_of_dev = np.argsort(cf, kind="stable")
    dev_of_table = np.empty(n_lm, dtype=np.int64)
    dev_of_table[table_of_dev] = np.arange(n_lm)
    ncp = np.where(cl[table_of_dev] >= 0, cl[table_of_dev] - cf[table_of_dev] + 1, 0)
    order = np.argsort(dev_of_table[lm], kind="stable")
    first_q = first[order]
    by_seg = np.argsort(first_q, kind="stable")
    pos = np.empty(len(stamp), dtype=np.int64)
    pos[by_seg] = np.arange(len(stamp))
    n_seg = n_cp - k + 1
    lm_cfirst = cf[table_of_dev]
    return {
        "table_type": typ[order], "table_idx": idx[order], "lm_dev": dev_of_table[lm][order], "first": first_q, "pos": pos,
        "seg_ptr": np.concatenate([[0], np.cumsum(np.bincount(first, minlength=n_seg))]),
        "dev_of_table": dev_of_table, "table_of_dev": table_of_dev,
        "lm_ptr": np.concatenate([[0], np.cumsum(np.bincount(dev_of_table[lm], minlength=n_lm))]),
        "lm_cfirst": lm_cfirst, "lm_ncp": ncp, "lm_yoff": np.concatenate([[0], np.cumsum(18 * ncp)]),
        "cf_ptr": np.searchsorted(lm_cfirst, np.arange(n_cp + 2), side="left"),
        "bw": np.array([max(k, ncp.max() if n_lm else 0)]), "y_total": np.array([18 * ncp.sum()]),
    }


@pytest.mark.parametrize("seed,k,n_cp,n_lm,n_px,n_br", [(1, 4, 16, 40, 300, 0), (2, 6, 24, 60, 0, 500), (3, 4, 63, 479, 9000, 4000), (4, 6, 12, 25, 130, 70),
                                                     (5, 4, 8, 5, 0, 0), (6, 4, 128, 1000, 10000, 0)])
def test_structure_against_numpy(seed, k, n_cp, n_lm, n_px, n_br, dumper, tmp_path):
    rng = np.random.default_rng(seed)
    t0, dt = 0.25 * seed, 0.1 / seed
    lo, hi = t0 + (k - 1) // 2 * dt, t0 + ((k - 1) // 2 + n_cp - k + 1) * dt  # stamps whose segment lies inside the window

    def table(n):
        lm = rng.integers(0, max(1, n_lm - n_lm // 5), n)  # the last fifth of the landmarks is never observed
        anchor = rng.uniform(lo, hi, n_lm)
        stamp = np.clip(anchor[lm] + rng.uniform(-6 * dt, 6 * dt, n), lo, np.nextafter(hi, lo))  # tracks of <= 12 segments, unsorted in time
        on_knot = rng.random(n) < 0.1
        stamp[on_knot] = np.minimum(t0 + np.round((stamp[on_knot] - t0) / dt) * dt, np.nextafter(hi, lo))  # stamps exactly on a knot
        stamp = np.maximum(stamp, lo)
        first = np.floor((stamp - t0) / dt) - (k - 1) // 2  # (a stamp on the window's first knot may round into the segment before it)
        stamp[(first < 0) | (first >= n_cp - k + 1)] = 0.5 * (lo + hi)
        return stamp, lm

    px_stamp, px_lm = table(n_px)
    br_stamp, br_lm = table(n_br)
    got = run(dumper, tmp_path, k, n_cp, n_lm, t0, dt, px_stamp, px_lm, br_stamp, br_lm)
    want = expected(k, n_cp, n_lm, t0, dt, px_stamp, px_lm, br_stamp, br_lm)
    assert "error" not in got, got
    for name, w in want.items():
        assert np.array_equal(got[name], w), name


def test_structure_rejects_bad_tables(dumper, tmp_path):
    ok = dict(k=4, n_cp=10, n_lm=3, t0=0.0, dt=0.1)
    inside = 0.35
    got = run(dumper, tmp_path, px_stamp=[inside], px_lm=[3], br_stamp=[], br_lm=[], **ok)
    assert "landmark outside" in got["error"]
    got = run(dumper, tmp_path, px_stamp=[], px_lm=[], br_stamp=[0.05], br_lm=[0], **ok)
    assert "outside the valid range" in got["error"]
    got = run(dumper, tmp_path, px_stamp=[0.85], px_lm=[0], br_stamp=[], br_lm=[], **ok)  # first control point 7: segment 7 of 0 .. 6
    assert "outside the valid range" in got["error"]


def expected_chunks(want, k, n_cp, R, L):
    """The chunks of the fused build (host_structure.hpp: build_chunks): observed landmarks of one first control point, filled greedily up to
    L landmarks and R residuals; descriptor = [first landmark, landmarks, first control point, first residual, residuals, chunk id, 0, 0]."""
    lm_ptr, cf_ptr, cfirst = want["lm_ptr"], want["cf_ptr"], want["lm_cfirst"]
    per_lm = np.diff(lm_ptr)
    n_obs = int(np.count_nonzero(per_lm))  # unobserved landmarks are last in device order
    ch_ptr, gw_ptr, gw_cf = [], [], []
    for c in range(n_cp):
        gw_ptr.append(len(gw_cf))
        d, d1 = min(cf_ptr[c], n_obs), min(cf_ptr[c + 1], n_obs)
        while d < d1:
            ch_ptr.append(d), gw_cf.append(c)
            e, cnt = d, 0
            while e < d1 and e - d < L and cnt + per_lm[e] <= R:
                cnt += per_lm[e]
                e += 1
            assert e > d, "a landmark with more than R residuals: build_chunks declines"
            d = e
    gw_ptr.append(len(gw_cf))
    ch_ptr.append(n_obs), gw_cf.append(0)
    desc = [[lo, hi - lo, cfirst[lo], lm_ptr[lo], lm_ptr[hi] - lm_ptr[lo], w, 0, 0] for w, (lo, hi) in enumerate(zip(ch_ptr[:-1], ch_ptr[1:]))]
    return np.array(ch_ptr), np.array(gw_ptr), np.array(gw_cf), np.array(desc, dtype=np.int64).reshape(-1, 8)


def chunk_weight(desc, first, k, bw):
    """Cost model of order_chunks_for_dispatch: the most records any control-point column of the chunk collects (the longest stream of its J'J phase)."""
    q = first[desc[3]:desc[3] + desc[4]] - desc[2]
    cnt = np.bincount(q[(q >= 0) & (q < bw)], minlength=bw + k)
    return max(int(cnt[max(0, o - k + 1):o + 1].sum()) for o in range(bw + k))


def expected_dispatch(desc, first, k, bw, n_cu):
    n = len(desc)
    if n <= 1:
        return desc
    weight = np.array([chunk_weight(d, first, k, bw) for d in desc])
    order = np.argsort(-weight, kind="stable")
    slot = np.empty(n, dtype=np.int64)
    if n_cu < n <= 2 * n_cu:  # workgroup w runs on compute unit w mod n_cu: the heaviest chunks get a unit to themselves, the others pair heavy + light
        n_lone, n_pair = 2 * n_cu - n, n - n_cu
        slot[n_pair:n_pair + n_lone] = order[:n_lone]
        slot[:n_pair] = order[n_lone:n_lone + n_pair]
        slot[n_cu:] = order[::-1][:n_pair]
    else:
        slot[:] = order
    return desc[slot]


@pytest.mark.parametrize("seed,k,n_cp,n_lm,n_px,R,L", [(11, 4, 16, 40, 700, 138, 14), (12, 4, 63, 479, 9000, 138, 14), (13, 6, 24, 60, 800, 84, 8), (14, 5, 40, 300, 3000, 112, 11),
                                                     (15, 4, 10, 12, 90, 40, 3), (16, 4, 128, 2000, 20000, 138, 14), (17, 4, 128, 5000, 50000, 138, 14)])
def test_chunks_and_dispatch_order(seed, k, n_cp, n_lm, n_px, R, L, dumper, tmp_path):
    rng = np.random.default_rng(seed)
    t0, dt = 0.0, 0.05
    lo, hi = t0 + (k - 1) // 2 * dt, t0 + ((k - 1) // 2 + n_cp - k + 1) * dt
    lm = rng.integers(0, n_lm, n_px)
    anchor = rng.uniform(lo, hi, n_lm)
    stamp = np.clip(anchor[lm] + rng.uniform(-3 * dt, 3 * dt, n_px), lo + 1e-9, hi - 1e-9)
    none = np.array([], dtype=np.float64)
    got = run(dumper, tmp_path, k, n_cp, n_lm, t0, dt, stamp, lm, none, none.astype(np.int32), geometry=(R, L))
    want = expected(k, n_cp, n_lm, t0, dt, stamp, lm, none, none.astype(np.int64))
    assert "error" not in got, got
    ch_ptr, gw_ptr, gw_cf, desc = expected_chunks(want, k, n_cp, R, L)
    assert np.array_equal(got["ch_ptr"], ch_ptr) and np.array_equal(got["gw_ptr"], gw_ptr) and np.array_equal(got["gw_cf"], gw_cf)
    assert np.array_equal(got["ch_desc"].reshape(-1, 8), desc)
    assert desc[:, 1].max() <= L and desc[:, 4].max() <= R and desc[:, 4].sum() == n_px  # every residual in exactly one chunk
    bw = int(want["bw"][0])
    for n_cu in (8, 256):
        d = got["ch_desc_cu%d" % n_cu].reshape(-1, 8)
        assert sorted(d[:, 5]) == list(range(len(desc)))  # a permutation: every chunk dispatched once, its partial slot unchanged
        assert np.array_equal(d, expected_dispatch(desc, want["first"], k, bw, n_cu)), n_cu
