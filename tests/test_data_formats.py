"""CPU: split-file / multi-scale naming logic and the data oracle against fixtures from the reference
(data half; reference gta_dataset.py:184-211,354-422, augmentations.py:58-100)."""
import os

import numpy as np

from mindtheedge_amd.datasets.kitti_edges import SPLIT_COLUMNS, multiscale_paths, parse_split_line, read_split
from oracle import data_oracle as do

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_prep.npz")


def test_split_line_columns():
    line = "img/0001.png gt/0001.png ann/00000000_lidar_000.png velo/0001.bin None None None ann/normals/00000000_lidar_000.png\n"
    rec = parse_split_line(line)
    assert tuple(rec) == SPLIT_COLUMNS
    assert rec["rgb"] == "img/0001.png" and rec["lidar"] == "velo/0001.bin" and rec["seg"] is None and rec["rgb_edge"] is None
    assert rec["normal"] == "ann/normals/00000000_lidar_000.png"
    short = parse_split_line("img/0001.png gt/0001.png \n")                 # trailing-newline column is dropped (:187-188)
    assert short == {"rgb": "img/0001.png", "depth": "gt/0001.png"}


def test_split_line_matches_what_the_annotator_writes(tmp_path):
    """infer_edge_estimation.save_split_list (reference :103-117) writes: rgb lidar edges lidar None None None normals."""
    p = os.path.join(tmp_path, "split.txt")
    with open(p, "w") as f:
        f.write("a.png l.bin out/00000000_lidar_000.png l.bin None None None out/normals/00000000_lidar_000.png\n\n")
    (rec,) = read_split(p)
    assert rec["depth"] == "l.bin" and rec["edge"] == "out/00000000_lidar_000.png" and rec["rgb_edge_for_loss"] is None


def test_multiscale_naming(tmp_path):
    base = os.path.join(tmp_path, "00000012_lidar_000.png")
    assert multiscale_paths(base) == [base]                                 # no _001 on disk: single scale, like the reference
    for i in range(4):
        open(base.replace("_000", "_00%d" % i), "w").close()
    got = multiscale_paths(base)
    assert [os.path.basename(g) for g in got] == ["00000012_lidar_00%d.png" % i for i in range(4)]
    assert multiscale_paths("x/7_regular_000.png", require_existing=False)[3] == "x/7_regular_003.png"


def test_data_oracle_matches_reference_fixtures():
    z = np.load(GOLDEN)
    for name in ("down", "kitti", "up", "same", "empty"):
        got = do.resize_depth_preserve(z["rdp_%s_in" % name], tuple(z["rdp_%s_shape" % name]))
        np.testing.assert_array_equal(got, z["rdp_%s_out" % name])
    np.testing.assert_array_equal(do.normal_from_u8(z["u8"]), z["normal"])
    np.testing.assert_array_equal(do.edge_from_u8(z["u8"]), z["edge"])
    assert (z["rdp_down_out"] > 0).sum() < (z["rdp_down_in"] > 0).sum()     # collisions: several points share a target
