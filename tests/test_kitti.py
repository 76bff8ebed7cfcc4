"""KITTI evaluation helpers: D1 definition and the 16-bit disparity PNG encoding."""
import numpy as np

from oracle import stereo_oracle as O
from redtail_amd import kitti


def test_d1_definition():
    gt = np.array([[0.0, 10.0, 10.0, 100.0, 100.0, 2.0]])
    est = np.array([[55.0, 12.9, 13.1, 104.0, 106.0, 5.5]])
    # no gt | 2.9 px ok | 3.1 px and 31 % -> outlier | 4 px but 4 % ok | 6 px and 6 % -> outlier | 3.5 px, 175 % -> outlier
    assert abs(kitti.d1_all(est, gt) - 100.0 * 3 / 5) < 1e-9
    assert np.isnan(kitti.d1_all(est, np.zeros_like(gt)))


def test_disparity_png_round_trip(tmp_path):
    disp = np.random.default_rng(0).uniform(0, 250, (37, 53)).astype(np.float32)
    disp[3, 4] = 0.0
    p = str(tmp_path / "d.png")
    kitti.write_disparity_png(p, disp)
    back = kitti.read_disparity_png(p)
    assert np.abs(back - disp).max() <= 0.5 / 256 + 1e-6
    assert back[3, 4] == 0.0
    # same integers as the device encoder / the oracle restatement of main.cpp:324-330
    assert np.array_equal((back * 256).astype(np.uint16), O.disparity_to_u16(disp, 256.0))
