"""GPU parity of the inertial factor (inertial.cpp:13-205) and of the bordered (bias splines + gravity) solve."""
import numpy as np
import pytest

import hyperslam_amd as ha
from hyperslam_amd import synthetic
from util import check_against_golden, golden_cases, golden_window, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx", [i for i, c in enumerate(golden_cases()) if c["type"] == "inertial"])
def test_hip_inertial_matches_golden(idx, hip):
    case = golden_cases()[idx]
    with ha.Problem(golden_window(case), lib=hip) as p:
        check_against_golden(p, case, 1e-9)


@pytest.mark.parametrize("order,identity", [(4, False), (6, False), (4, True)])
@pytest.mark.parametrize("robustify", [False, True])
def test_inertial_linearization_vs_oracle(order, identity, robustify, hip, oracle):
    w = synthetic.small_inertial(order=order, n_cp=18, identity=identity)
    with ha.Problem(w, lib=hip) as g, ha.Problem(w, lib=oracle) as c:
        a, b = g.linearize(ha.HS_INERTIAL, robustify), c.linearize(ha.HS_INERTIAL, robustify)
        assert np.array_equal(a["first_cp"], b["first_cp"]) and np.array_equal(a["first_bias"], b["first_bias"])
        for k in ("r", "J_state", "J_bias_g", "J_bias_a", "J_gravity", "cost"):
            assert rel(a[k], b[k]) < 1e-9, (k, rel(a[k], b[k]))
        L = g.residual_layout(ha.HS_INERTIAL, 3)
        Lc = c.residual_layout(ha.HS_INERTIAL, 3)
        for k in L:
            assert np.array_equal(L[k], Lc[k]), k
