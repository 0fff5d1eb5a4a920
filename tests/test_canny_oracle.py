"""CPU: known-answer tests of oracle/canny_oracle.py.  PARITY UNPINNED (OpenCV absent): these pin the restatement to the
behaviour OpenCV's Canny is documented to have, not to cv2 itself."""
import numpy as np

from oracle import canny_oracle as co


def test_constants_and_sobel():
    assert co.TG22 == 13573                                   # round(0.41421356 * 2^15), canny.cpp
    ramp = np.tile(np.arange(9, dtype=np.uint8) * 10, (7, 1))
    dx, dy = co.sobel3_s16(ramp)
    assert np.all(dx[:, 1:-1] == 80) and np.all(dy == 0)      # (1+2+1) * 2 * 10
    assert np.all(dx[:, 0] == 40) and np.all(dx[:, -1] == 40)  # replicated border halves the central difference


def test_step_edge_is_one_pixel_wide_on_the_low_side():
    im = np.zeros((12, 12), np.uint8)
    im[:, 6:] = 100
    e = co.canny(im, 10, 20)
    assert np.all(e[:, 5] == 255) and e.sum() == 255 * 12
    assert co.canny(im, 20, 10).tolist() == e.tolist()         # thresholds are ordered internally
    assert co.canny(im, 500, 600).sum() == 0                   # magnitude 400 is below both


def test_hysteresis_keeps_only_weak_pixels_connected_to_strong_ones():
    im = np.zeros((20, 40), np.uint8)
    im[:, 10:] = 12                                            # weak step: magnitude 48
    im[:10, 10:] = 100                                         # strong step in the upper half: magnitude 400
    im[12:, 30:] += 5                                          # detached faint step: magnitude 20
    e = co.canny(im, 30, 200)
    assert np.all(e[:9, 9] == 255)                             # strong part (row 9 turns the corner along the horizontal edge)
    assert np.all(e[11:, 9] == 255)                            # weak continuation survives through connectivity
    assert np.all(e[9, 10:] == 255)                            # the horizontal strong edge between the two halves
    assert e[11:, 25:].sum() == 0                              # the detached faint step is below the low threshold
    assert co.canny(im, 10, 2000).sum() == 0                   # nothing is strong -> nothing survives


def test_depth_to_u8_truncates():
    d = np.array([[1.0, 2.0], [3.0, 4.0]], np.float32)
    assert co.depth_to_u8(d).tolist() == [[63, 127], [191, 255]]
