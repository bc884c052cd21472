"""The look-ahead band Cholesky of the reduced system (hyperslam_amd/csrc/kernels_factor.hpp: k_band_factor_la — the longest kernel of an
iteration) compiled from the product's source for the HOST (tests/emul/: one thread per lane) and checked against numpy's dense Cholesky on
CPU: factor rows, inverted diagonal blocks and the forward-solved right-hand side, one-ended and from both ends (the far end on the reversed
system, the middle rows as the factor of the Schur complement both ends leave on them: launch_factor's job layout, host_launch.hpp), followed
by the backward sweeps in super-blocks with their inverse builders (kernels_backward_sb.hpp: k_band_backward_sb, the far sweep's phase A
included) against numpy's solve: solution, step, scaled step and both ends' shares of the model cost change.
The `-m gpu` tests (test_every_band_width, test_band_widths_from_both_ends) remain the parity tests of the compiled kernel; this one makes
its index arithmetic — ring slots, look-ahead, hand-over of the far end's window in the near end's coordinates — checkable without a GPU.
Further down: the kernels launch_factor picks for other band shapes, the sliding window's frozen prefix, and the bordered solve of windows
with an IMU (kernels_border.hpp: forward sweep of the border columns from one or both ends, border Schur complement, dense Cholesky, y').
Since round 5 the two-ended factorisation of bands up to 14 control points is k_band_factor_mx (kernels_factor_mx.hpp: the trailing window
in the accumulators of the f64 matrix cores, ring coordinates, 16 compile-time phases, panel / loader / storer / inverse waves): the harness
runs it too (variant 5; tests/emul/hip/hip_runtime.h emulates v_mfma_f64_16x16x4_f64 with the lane layout tools/microbench/mfma_probe.hip
confirmed on the GPU), one-ended and from both ends, at every band width it holds.
Replaces what CHOLMOD does for /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:46-48 (SPARSE_NORMAL_CHOLESKY)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("emul_factor") / "factor_harness")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-pthread", "-Wno-psabi", "-I", EMUL, "-o", exe, os.path.join(EMUL, "factor_harness.cpp")])
    return exe


def banded_spd(rng, n_blk, bw):
    """Symmetric positive definite, block (i, j) non-zero iff |i - j| < bw: a sum of window Gram matrices, as the reduced system is."""
    n = 6 * n_blk
    M = np.zeros((n, n))
    for s in range(n_blk):
        e = min(s + bw, n_blk)
        J = rng.standard_normal((3 * (e - s), 6 * (e - s)))
        M[6 * s:6 * e, 6 * s:6 * e] += J.T @ J
    M += np.diag(rng.uniform(0.5, 1.5, n))
    d = 1.0 / np.sqrt(np.diag(M))  # Jacobi scaled, like the system the kernel sees
    return M * d[:, None] * d[None, :] + 1e-3 * np.eye(n)


def band_rows(M, bw):
    """Row rho stores M[rho][6 (rho / 6) + c], c < 6 bw (zero beyond the matrix): the kernels' band layout (kernels_factor.hpp header)."""
    n, ncb = len(M), 6 * bw
    B = np.zeros((n, ncb))
    for r in range(n):
        c0 = 6 * (r // 6)
        w = min(ncb, n - c0)
        B[r, :w] = M[r, c0:c0 + w]
    return B


def check_job(Ub, Ubk, yb, U, y, rows, bw, col_limit=None):
    """Rows [0, rows) of a job against the upper factor U (job coordinates) and y = U^-T g. The kernel leaves the diagonal and the lower part
    of a diagonal block unspecified (never read: the sweeps use the inverted blocks)."""
    ncb = 6 * bw
    n = len(U)
    for r in range(rows):
        c0 = 6 * (r // 6)
        lim = n if col_limit is None else col_limit
        for c in range(r - c0 + 1, ncb):
            want = U[r, c0 + c] if c0 + c < lim else 0.0
            assert abs(Ub[r, c] - want) <= 1e-11 * max(1.0, abs(want)), (r, c, Ub[r, c], want)
    for i in range(rows // 6):
        W = np.linalg.inv(U[6 * i:6 * i + 6, 6 * i:6 * i + 6])
        for a in range(6):
            for c in range(a, 6):
                got = Ubk[24 * i + a * 6 - a * (a - 1) // 2 + (c - a)]
                assert abs(got - W[a, c]) <= 1e-10 * max(1.0, abs(W[a, c])), (i, a, c)
    assert np.allclose(yb[:rows], y[:rows], rtol=0, atol=1e-11)


def run(exe, tmp_path, M, g, bw, two_ended, variant=0, f0=0, border=None):
    """border = (S_pb, S_bb, g_b, first non-zero block row per group of two border columns): the bordered system [M S_pb; S_pb' S_bb]."""
    n = len(M)
    P = np.arange(n)[::-1]
    src, dst = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    aux = np.random.default_rng(n + bw)
    scale, g_full, d2 = aux.uniform(0.5, 2.0, n), aux.standard_normal(n), aux.uniform(0.0, 1.0, n)  # operands of the step outputs
    nb = 0 if border is None else len(border[2])
    scale_b, d2b = aux.uniform(0.5, 2.0, nb), aux.uniform(0.0, 1.0, nb)
    with open(src, "wb") as f:
        f.write(struct.pack("=6i", n, bw, int(two_ended) | (f0 << 8), variant, nb, 0))
        for a in (band_rows(M, bw), g, band_rows(M[np.ix_(P, P)], bw), g[P], scale, g_full, d2):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        if border is not None:
            for a in (border[0], border[1], border[2], scale_b, d2b):
                f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(border[3], dtype="<i4").tobytes())
    subprocess.check_call([exe, src, dst], timeout=180)
    raw = open(dst, "rb").read()
    m, mB, failed, _ = struct.unpack("=4i", raw[:16])
    v = np.frombuffer(raw[16:], dtype="<f8")
    ncb, nU, nK = 6 * bw, n * 6 * bw, 24 * (n // 6)
    parts, o = [], 0
    for size in (nU, nK, n, nU, nK, n, n, n, n, 4, nb, nb):
        parts.append(v[o:o + size])
        o += size
    assert failed == 0 and o == len(v)
    # ---- the sweeps: solution, step = -x, scaled step, and the two sums of the model cost change (decide_step adds the two ends' shares) ----
    xsol, step, delta, sums = parts[6:10]
    if border is not None:  # bordered solve: Z = U^-T S_pb, C = S_bb - Z'Z, dense Cholesky, y' = y - Z x_b, then the sweeps on y'
        A = np.block([[M, border[0]], [border[0].T, border[1]]])
        xa = np.linalg.solve(A, np.concatenate([g, border[2]]))
        xb, delta_b = parts[10], parts[11]
        tol = 1e-9 * max(1.0, np.abs(xa).max())
        assert np.allclose(xb, xa[n:], rtol=0, atol=tol) and np.allclose(step, -xa[:n], rtol=0, atol=tol)
        assert np.array_equal(delta_b, scale_b * -xb) and np.array_equal(delta, scale * step)
        want_g, want_d = g_full @ step + border[2] @ -xb, (d2 * step) @ step + (d2b * xb) @ xb
        assert abs(sums[0] + sums[2] - want_g) <= 1e-12 * (np.abs(g_full * step).sum() + np.abs(border[2] * xb).sum())
        assert abs(sums[1] + sums[3] - want_d) <= 1e-12 * want_d
        return m, mB, parts[0].reshape(n, ncb), parts[1], parts[2], parts[3].reshape(n, ncb), parts[4], parts[5]
    x = np.linalg.solve(M, g)
    assert np.allclose(step, -x, rtol=0, atol=1e-9 * max(1.0, np.abs(x).max()))
    if two_ended:
        assert np.array_equal(xsol, -step)
    assert np.array_equal(delta, scale * step)
    assert abs(sums[0] + sums[2] - g_full @ step) <= 1e-12 * np.abs(g_full * step).sum()
    assert abs(sums[1] + sums[3] - (d2 * step) @ step) <= 1e-12 * (d2 * step * step).sum()
    return m, mB, parts[0].reshape(n, ncb), parts[1], parts[2], parts[3].reshape(n, ncb), parts[4], parts[5]


@pytest.mark.parametrize("n_blk,bw", [(20, 4), (17, 6), (30, 14), (24, 16), (9, 5)])
def test_one_ended_factor_against_numpy(n_blk, bw, harness, tmp_path):
    rng = np.random.default_rng(100 * n_blk + bw)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    _, _, Ub, Ubk, yb, _, _, _ = run(harness, tmp_path, M, g, bw, False)
    U = np.linalg.cholesky(M).T
    check_job(Ub, Ubk, yb, U, np.linalg.solve(U.T, g), 6 * n_blk, bw)


def check_two_ended(n_blk, bw, harness, tmp_path, variant):
    rng = np.random.default_rng(7 * n_blk + bw)
    M = banded_spd(rng, n_blk, bw)
    n = 6 * n_blk
    g = rng.standard_normal(n)
    m, mB, Ub, Ubk, yb, Ub2, Ubk2, yb2 = run(harness, tmp_path, M, g, bw, True, variant)
    w = bw - 1
    assert m + w + mB == n_blk and m >= mB
    # near end: the leading rows of the factor of M; far end: the leading rows of the factor of the reversed matrix
    U = np.linalg.cholesky(M).T
    check_job(Ub, Ubk, yb, U, np.linalg.solve(U.T, g), 6 * m, bw)
    P = np.arange(n)[::-1]
    U2 = np.linalg.cholesky(M[np.ix_(P, P)]).T
    check_job(Ub2, Ubk2, yb2, U2, np.linalg.solve(U2.T, g[P]), 6 * mB, bw)
    # middle rows: factor of the Schur complement that eliminating both ends leaves on them (entries that coupled to the far end are zero)
    t, mid, b = np.arange(0, 6 * m), np.arange(6 * m, 6 * (m + w)), np.arange(6 * (m + w), n)
    A, h = M[np.ix_(mid, mid)].copy(), g[mid].copy()
    for e in (t, b):
        X = np.linalg.solve(M[np.ix_(e, e)], np.column_stack([M[np.ix_(e, mid)], g[e]]))
        A -= M[np.ix_(mid, e)] @ X[:, :-1]
        h -= M[np.ix_(mid, e)] @ X[:, -1]
    Um = np.linalg.cholesky(A).T
    # in the job's own coordinates: rows 6 m .. of a matrix whose leading part is irrelevant for the comparison
    Ufull = np.zeros((6 * (m + w), 6 * (m + w)))
    Ufull[6 * m:, 6 * m:] = Um
    yfull = np.zeros(6 * (m + w))
    yfull[6 * m:] = np.linalg.solve(Um.T, h)
    ncb = 6 * bw
    for r in range(6 * m, 6 * (m + w)):
        c0 = 6 * (r // 6)
        for c in range(r - c0 + 1, ncb):
            want = Ufull[r, c0 + c] if c0 + c < 6 * (m + w) else 0.0
            assert abs(Ub[r, c] - want) <= 1e-10 * max(1.0, abs(want)), (r, c, Ub[r, c], want)
    assert np.allclose(yb[6 * m:6 * (m + w)], yfull[6 * m:], rtol=0, atol=1e-10)
    for i in range(m, m + w):
        W = np.linalg.inv(Ufull[6 * i:6 * i + 6, 6 * i:6 * i + 6])
        for a in range(6):
            for c in range(a, 6):
                assert abs(Ubk[24 * i + a * 6 - a * (a - 1) // 2 + (c - a)] - W[a, c]) <= 1e-9 * max(1.0, abs(W[a, c]))



@pytest.mark.parametrize("n_blk,bw", [(20, 4), (58, 14), (31, 6), (64, 16), (40, 10)])
def test_two_ended_factor_against_numpy(n_blk, bw, harness, tmp_path):
    check_two_ended(n_blk, bw, harness, tmp_path, 0)


@pytest.mark.parametrize("n_blk,bw", [(12, 3), (20, 4), (31, 6), (40, 10), (55, 13), (58, 14), (128, 14), (60, 15), (64, 16), (70, 16)])
def test_mx_two_ended_factor_against_numpy(n_blk, bw, harness, tmp_path):
    """k_band_factor_mx from both ends: junction merge into accumulators, rowbuf rows and the storer's right-hand sides; the far end's window
    handed over from all three places; (128, 14) is configs[1]'s shape; 15 and 16 control points per landmark (order-6 windows, configs[2]) are
    the WIDE instance: every slot of the ring in use, the last band blocks of a row through the loader."""
    check_two_ended(n_blk, bw, harness, tmp_path, 5)


@pytest.mark.parametrize("n_blk,bw", [(5, 3), (8, 3), (9, 5), (17, 6), (40, 10), (33, 13), (20, 14), (30, 14), (24, 15), (18, 16), (40, 16)])
def test_mx_one_ended_factor_against_numpy(n_blk, bw, harness, tmp_path):
    """k_band_factor_mx on one job: every phase of the ring (more than 16 block rows), systems shorter than the ring, shorter than the band."""
    rng = np.random.default_rng(100 * n_blk + bw)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    _, _, Ub, Ubk, yb, _, _, _ = run(harness, tmp_path, M, g, bw, False, 5)
    U = np.linalg.cholesky(M).T
    check_job(Ub, Ubk, yb, U, np.linalg.solve(U.T, g), 6 * n_blk, bw)


@pytest.mark.parametrize("variant,n_blk,bw", [(1, 20, 12), (1, 14, 16), (2, 24, 18), (2, 23, 21), (3, 20, 24), (3, 17, 30), (4, 24, 20), (4, 30, 17)])
def test_other_band_kernels_against_numpy(variant, n_blk, bw, harness, tmp_path):
    """The one-ended kernels launch_factor picks for other band shapes — k_band_factor<1> (every tile in one lane: bw^2 <= 256), <2> (bw <= 21),
    k_band_factor_wide (long feature tracks: trailing window in L2) and k_dense_factor (window-wide bands of the sliding-window replay) — and,
    from 18 control points per landmark on, the sweep with one block row per step (k_band_backward): factor, inverted diagonal blocks,
    forward-solved right-hand side and the solution against numpy."""
    rng = np.random.default_rng(1000 * variant + 10 * n_blk + bw)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    _, _, Ub, Ubk, yb, _, _, _ = run(harness, tmp_path, M, g, bw, False, variant)
    U = np.linalg.cholesky(M).T
    check_job(Ub, Ubk, yb, U, np.linalg.solve(U.T, g), 6 * n_blk, bw)


@pytest.mark.parametrize("variant,n_blk,bw,f0", [(0, 30, 14, 9), (1, 20, 12, 5), (2, 24, 18, 11), (3, 20, 24, 7), (4, 36, 18, 14), (0, 24, 6, 23)])
def test_frozen_prefix_against_numpy(variant, n_blk, bw, f0, harness, tmp_path):
    """Sliding window: the leading block rows belong to constant control points — unit diagonal, no coupling, zero right-hand side. They are
    factored one wave each (k_factor_decoupled_rows; extra workgroups of the k_dense_factor launch), the dependency chain of the factorisation
    and of the sweeps starts behind them (launch_factor: pointer offsets into the row-relative band storage, j_lo of the sweeps)."""
    rng = np.random.default_rng(77 * variant + n_blk + bw + f0)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    k = 6 * f0
    M[:k, :], M[:, :k] = 0.0, 0.0
    M[:k, :k] = np.eye(k)
    g[:k] = 0.0
    _, _, Ub, Ubk, yb, _, _, _ = run(harness, tmp_path, M, g, bw, False, variant, f0)
    U = np.linalg.cholesky(M).T
    check_job(Ub, Ubk, yb, U, np.linalg.solve(U.T, g), 6 * n_blk, bw)


def bordered(rng, M, n_b, n_blk):
    """Border columns as bias points and gravity make them: a column meets a contiguous stretch of the pose rows (zero above its first block
    row), columns come in groups of two that start together; S_bb makes the whole system positive definite."""
    n = len(M)
    B = np.zeros((n, n_b))
    n_groups = (n_b + 1) // 2
    start = np.sort(rng.integers(0, max(1, n_blk - 4), n_groups))
    start[-1] = 0  # (gravity: every row)
    for c in range(n_b):
        r0 = 6 * start[c // 2]
        r1 = n if c // 2 == n_groups - 1 else min(n, r0 + 6 * int(rng.integers(3, 9)))
        B[r0:r1, c] = 0.1 * rng.standard_normal(r1 - r0)
    C = B.T @ np.linalg.solve(M, B) + np.diag(rng.uniform(0.5, 1.5, n_b))
    return B, C, rng.standard_normal(n_b), start


@pytest.mark.parametrize("n_blk,bw,n_b,two_ended", [(24, 6, 9, False), (30, 14, 21, False), (64, 16, 57, True), (40, 10, 99, True), (31, 6, 3, True),
                                                    (26, 5, 131, False)])  # (more than 127 border unknowns: k_border_solve, the trailing matrix in LDS)
def test_bordered_solve_against_numpy(n_blk, bw, n_b, two_ended, harness, tmp_path):
    """Windows with an IMU: the border chain behind the band factorisation — k_border_forward / k_border_forward2 (both ends, hand-over of the
    far end's updates of the middle rows per column group), k_border_schur, k_border_solve_reg (trailing matrix in registers, two columns per
    barrier, one-wave backward sweep), k_border_apply — and the sweeps with the border's step outputs, against numpy's solve of the whole
    bordered system. Replaces the dense tail of CHOLMOD's factorisation for optimizer.cpp:46-48."""
    rng = np.random.default_rng(5 * n_blk + bw + n_b)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    run(harness, tmp_path, M, g, bw, two_ended, border=bordered(rng, M, n_b, n_blk))


@pytest.mark.parametrize("n_blk,bw,n_b", [(64, 16, 57), (40, 10, 99), (60, 15, 21)])
def test_mx_bordered_solve_against_numpy(n_blk, bw, n_b, harness, tmp_path):
    """The same chain behind k_band_factor_mx, which publishes the count of complete block rows per end (MfmaJob::progress) for a forward
    sweep of the border columns that runs next to it on the device (k_border_forward2 with its polling wave, BfJob::progress). Here the
    factorisation has finished when the sweep starts: the words must hold the final counts (harness exit code 9) and the sweep must read
    the factor through them."""
    rng = np.random.default_rng(5 * n_blk + bw + n_b)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    run(harness, tmp_path, M, g, bw, True, variant=5, border=bordered(rng, M, n_b, n_blk))


@pytest.mark.parametrize("n_blk,bw,f0,n_b", [(33, 33, 0, 0), (33, 33, 0, 44), (36, 18, 14, 0), (63, 33, 30, 47), (63, 33, 30, 0), (42, 20, 0, 3), (5, 4, 0, 0),
                                             (4, 4, 2, 2), (20, 6, 0, 9), (2, 2, 0, 45), (34, 34, 0, 51), (42, 42, 0, 0)])
def test_dense_solve_mx_against_numpy(n_blk, bw, f0, n_b, harness, tmp_path):
    """k_dense_solve_mx (kernels_dense_mx.hpp): the whole solve of a small reduced system — the sliding window's steady state: ~33 free block
    rows with window-wide bands behind a frozen prefix, bordered by the bias / gravity unknowns with an IMU — in one launch, trailing matrix in
    the accumulators of the f64 matrix cores, the border as the last columns of ONE dense Cholesky: 16 x 16 tiles 2-D cyclic over eight waves,
    readlane panel, W_k rows for the sweep, padding to whole tiles (identity), every size from one tile to sixteen (255 unknowns + the right-hand
    side column: (34, 34, 0, 51), (42, 20, 0, 3)), band narrower than the window (zero corner blocks), frozen prefix (zero step on its rows)."""
    rng = np.random.default_rng(31 * n_blk + bw + 7 * f0 + n_b)
    M = banded_spd(rng, n_blk, bw)
    g = rng.standard_normal(6 * n_blk)
    k = 6 * f0
    if f0:
        M[:k, :], M[:, :k] = 0.0, 0.0
        M[:k, :k] = np.eye(k)
        g[:k] = 0.0
    border = None
    if n_b:
        border = bordered(rng, M, n_b, n_blk)
        border[0][:k, :] = 0.0  # (a constant control point has no Jacobian columns: its rows of S_pb are zero)
        border = (border[0], border[0].T @ np.linalg.solve(M, border[0]) + np.diag(rng.uniform(0.5, 1.5, n_b)), border[2], border[3])
    run(harness, tmp_path, M, g, bw, False, 6, f0, border=border)


@pytest.mark.parametrize("n_b", [110, 64, 255, 17, 1])
def test_dense_solve_mx_border_mode_against_numpy(n_b, harness, tmp_path):
    """k_dense_solve_mx in border mode (Tables::dense_border, launch_factor): the border Schur complement C | h of a two-ended bordered system
    (configs[2]: 110 unknowns) in the dense layout — padding by k_dense_border_init, entries as k_border_schur stores them — comes out as x_b
    and nothing else is written."""
    rng = np.random.default_rng(977 + n_b)
    A = rng.standard_normal((n_b, n_b + 8))
    C = A @ A.T + np.diag(rng.uniform(0.5, 1.5, n_b))
    h = rng.standard_normal(n_b)
    n_blk, bw = 4, 2
    M = banded_spd(rng, n_blk, bw)
    n = 6 * n_blk
    src, dst = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    P = np.arange(n)[::-1]
    z = np.zeros(n)
    with open(src, "wb") as f:
        f.write(struct.pack("=6i", n, bw, 0, 7, n_b, 0))
        for a in (band_rows(M, bw), z, band_rows(M[np.ix_(P, P)], bw), z, z, z, z):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        for a in (np.zeros((n, n_b)), C, h, np.ones(n_b), np.zeros(n_b)):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        f.write(np.zeros((n_b + 1) // 2 + 8, dtype="<i4").tobytes())
    subprocess.check_call([harness, src, dst], timeout=180)
    raw = open(dst, "rb").read()
    _, _, failed, _ = struct.unpack("=4i", raw[:16])
    xb = np.frombuffer(raw[16:], dtype="<f8")
    assert failed == 0 and len(xb) == n_b
    want = np.linalg.solve(C, h)
    assert np.allclose(xb, want, rtol=0, atol=1e-10 * max(1.0, np.abs(want).max()))
