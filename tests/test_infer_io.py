"""CPU: the host-side file handling of infer_edges.py (reference infer_edges.py:266-282, 349-353): image
files are opened with PIL, resized to the configured (H, W) with LANCZOS and scaled to [0,1] CHW exactly as the reference's
load_image / resize_image / to_tensor chain does; depth maps are written as float32 .npy + 8-bit depth/max*255 .png."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import infer_edges  # noqa: E402


def test_png_is_resized_with_lanczos_like_the_reference(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(75, 250, 3), dtype=np.uint8)          # KITTI-like aspect, not a multiple of 32
    path = os.path.join(tmp_path, "frame.png")
    Image.fromarray(a).save(path)
    t = infer_edges.load_frame(path, (64, 192))
    assert t.dtype == torch.float32 and tuple(t.shape) == (3, 64, 192) and 0.0 <= float(t.min()) and float(t.max()) <= 1.0
    want = np.asarray(Image.open(path).resize((192, 64), Image.LANCZOS), dtype=np.float32).transpose(2, 0, 1) / 255.0
    assert np.array_equal(t.numpy(), want)                               # = transforms.Resize((H, W), ANTIALIAS) + ToTensor
    same = infer_edges.load_frame(path, (75, 250))                       # already at the configured size: untouched
    assert np.array_equal(same.numpy(), a.transpose(2, 0, 1).astype(np.float32) / 255.0)


def test_npy_inputs_and_depth_outputs(tmp_path):
    chw = np.random.default_rng(1).random((3, 8, 16), dtype=np.float32)
    p1, p2 = os.path.join(tmp_path, "a.npy"), os.path.join(tmp_path, "b.npy")
    np.save(p1, chw)
    np.save(p2, (chw.transpose(1, 2, 0) * 255).astype(np.uint8))
    assert np.allclose(infer_edges.load_frame(p1).numpy(), chw)
    assert np.allclose(infer_edges.load_frame(p2).numpy(), np.floor(chw * 255) / 255.0)
    depth = torch.tensor([[0.5, 1.0], [20.0, 80.0]])
    base = os.path.join(tmp_path, "00000000")
    infer_edges.save_depth(base, depth)
    assert np.array_equal(np.load(base + "_regular.npy"), depth.numpy())
    png = np.asarray(Image.open(base + "_regular.png"))
    assert png.dtype == np.uint8 and png.tolist() == [[2, 3], [64, 255]]          # depth / max * 255, rounded
