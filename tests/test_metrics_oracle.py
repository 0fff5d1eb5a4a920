"""CPU: the metrics oracle (oracle/metrics_oracle.py) against the fixtures generated from the reference
(tests/golden/make_golden_metrics.py; reference utils/depth.py:230-325)."""
import glob
import os

import numpy as np
import pytest

from oracle import metrics_oracle as mo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[len("metrics_"):-4] for p in glob.glob(os.path.join(GOLDEN, "metrics_*.npz"))
               if not p.endswith("post_process.npz"))


def test_fixture_set_is_complete():
    assert set(CASES) >= {"resize_garg", "resize_nocrop", "same_garg", "topcenter", "empty_image", "even_count"}


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("use_gt_scale", [False, True])
def test_compute_depth_metrics_matches_reference(name, use_gt_scale):
    z = np.load(os.path.join(GOLDEN, "metrics_%s.npz" % name))
    got = mo.compute_depth_metrics(z["gt"], z["pred"], crop=str(z["crop"]), scale_output=str(z["scale_output"]),
                                   min_depth=float(z["min_depth"]), max_depth=float(z["max_depth"]), use_gt_scale=use_gt_scale)
    want = z["metrics_gt%d" % int(use_gt_scale)]
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7)     # float32 means in the reference vs float64 here


@pytest.mark.parametrize("method", ["mean", "max", "min"])
def test_post_process_inv_depth_matches_reference(method):
    z = np.load(os.path.join(GOLDEN, "metrics_post_process.npz"))
    got = mo.post_process_inv_depth(z["inv_depth"], z["inv_depth_flipped"], method)
    np.testing.assert_allclose(got, z["pp_" + method], rtol=1e-6, atol=1e-7)


def test_lower_median_of_even_count():
    assert mo.lower_median(np.array([4.0, 1.0, 3.0, 2.0], np.float32)) == 2.0


def test_unknown_modes_raise_like_the_reference():
    with pytest.raises(ValueError):
        mo.fuse_inv_depth(np.zeros(1, np.float32), np.zeros(1, np.float32), "median")
    with pytest.raises(NotImplementedError):
        mo.scale_depth(np.zeros((1, 1, 2, 2), np.float32), (1, 1, 4, 4), "bottom")
