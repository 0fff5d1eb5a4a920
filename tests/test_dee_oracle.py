"""CPU: the annotation post-processing oracle (oracle/dee_oracle.py) against fixtures produced by the reference's own
loops (tests/golden/make_golden_dee.py; reference packnet_sfm/utils/tools.py:9-92)."""
import glob
import os

import numpy as np
import pytest

from oracle import dee_oracle as do

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "dee_*.npz")) if "snake" not in p)


def same(a, b):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_array_equal(np.nan_to_num(a), np.nan_to_num(b))


def test_fixture_set_is_complete():
    assert set(CASES) >= {"a", "b", "c", "tiny", "thin", "nostrong"}


@pytest.mark.parametrize("name", CASES)
def test_nms_and_hysteresis_bit_exact(name):
    z = np.load(os.path.join(GOLDEN, "dee_%s.npz" % name))
    p = z["prob"]
    nms = do.non_max_suppression(p)
    assert nms.dtype == np.float64
    same(nms, z["nms"])
    same(do.hysteresis(nms), z["nms_hyst"])
    same(do.hysteresis(p.astype(np.float64)), z["hyst_only"])
    custom = (0.1, 0.45) if name == "nostrong" else (0.1, 0.5)
    same(do.hysteresis(nms, *custom), z["hyst_custom"])
    e, n = do.annotate(p)
    same(e, z["nms_hyst"])
    assert n.dtype == np.uint8 and n.shape == p.shape


def test_snake_needs_many_sweeps_and_drops_the_disconnected_piece():
    z = np.load(os.path.join(GOLDEN, "dee_snake.npz"))
    got = do.hysteresis(z["img"])
    same(got, z["hyst"])
    assert (got > 0).sum() == (z["img"] > 0).sum() - 4


def test_sobel5_known_answers():
    """OpenCV's 5x5 Sobel on a unit ramp: d/dx of img = x is sum(smooth) * sum(j * deriv[j]) = 16 * 6... checked from the
    published kernels: deriv taps (-1,-2,0,2,1) at offsets (-2..2) -> 2+2+2+2 = 8; smoothing sums to 16 -> 128 per unit slope;
    reflect-101 borders make the outermost two columns smaller."""
    H, W = 7, 9
    ramp = np.tile(np.arange(W, dtype=np.float32), (H, 1))
    sx, sy = do.sobel5(ramp, 1, 0), do.sobel5(ramp, 0, 1)
    assert np.all(sx[:, 2:-2] == 128.0) and np.all(sy == 0.0)
    assert np.all(sx[:, 0] == 0.0) and np.all(sx[:, 1] == 16.0 * 6.0)     # reflect: (-1)*2 + (-2)*0... = -2+0+0+4+4 = 6
    imp = np.zeros((9, 9), np.float32)
    imp[4, 4] = 1.0
    k = np.outer(do.SMOOTH5, do.DERIV5)
    np.testing.assert_array_equal(do.sobel5(imp, 1, 0)[2:7, 2:7], k[::-1, ::-1])   # correlation: impulse response = flipped kernel


def test_normals_quantisation_points():
    flat = np.zeros((6, 6), np.float32)
    assert np.all(do.normals_u8(flat) == 127)                             # atan2(-0, 0) = -0 -> (180/360)*255 = 127.5 -> 127
    ramp = np.tile(np.arange(8, dtype=np.float32), (8, 1))
    n = do.normals_u8(ramp)
    assert np.all(n[:, 2:-2] == 127)                                      # gradient along +x: angle 0
    assert np.all(do.normals_u8(ramp.T.copy())[2:-2, :] == 63)            # along +y: atan2(-sy, 0) = -pi/2 -> 63.75 -> 63
