"""CPU: the chamfer oracle (oracle/edge_oracle.py) against fixtures produced by the reference's chamfer_distance
(tests/golden/make_golden_chamfer.py; reference packnet_sfm/utils/edge.py:19-64)."""
import glob
import os

import numpy as np
import pytest

from oracle import edge_oracle as eo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[8:-4] for p in glob.glob(os.path.join(GOLDEN, "chamfer_*.npz")))


def close(a, b):
    if np.isnan(b):
        assert np.isnan(a)
    else:
        assert abs(a - b) <= 1e-12 * max(1.0, abs(b))


def test_fixture_set_is_complete():
    assert set(CASES) >= {"a", "b", "far", "same", "nopred"}


@pytest.mark.parametrize("name", CASES)
def test_chamfer_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, "chamfer_%s.npz" % name))
    c, p, m = eo.chamfer_distance(z["pred"], z["gt"])
    close(c, float(z["c_pg"])); close(p, float(z["p_pg"]))
    np.testing.assert_array_equal(m, z["m_pg"])
    if "c_gp" in z:
        c, p, m = eo.chamfer_distance(z["gt"], z["pred"])
        close(c, float(z["c_gp"])); close(p, float(z["p_gp"]))
        np.testing.assert_array_equal(m, z["m_gp"])
        c, p, _ = eo.chamfer_distance(z["pred"], z["gt"], 2.5)
        close(c, float(z["c_t25"])); close(p, float(z["p_t25"]))
        c, p, _ = eo.chamfer_distance(z["pred"], z["soft_gt"])
        close(c, float(z["c_soft"])); close(p, float(z["p_soft"]))
        pr, rc, f1 = eo.precision_recall_f1(z["pred"], z["gt"])
        close(pr, float(z["p_pg"])); close(rc, float(z["p_gp"]))
        with np.errstate(invalid="ignore"):
            close(f1, 2 * np.float64(z["p_pg"]) * np.float64(z["p_gp"]) / (np.float64(z["p_pg"]) + np.float64(z["p_gp"])))


def test_squared_edt_is_exact():
    m = np.zeros((5, 7), bool)
    m[1, 2] = m[4, 6] = True
    d2 = eo.squared_edt(m)
    ys, xs = np.mgrid[0:5, 0:7]
    want = np.minimum((ys - 1) ** 2 + (xs - 2) ** 2, (ys - 4) ** 2 + (xs - 6) ** 2)
    np.testing.assert_array_equal(d2, want)
