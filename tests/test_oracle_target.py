"""Pin oracle/target.py against the REFERENCE's own numpy outputs (tests/golden/target_*.npz)."""
import numpy as np
import pytest

from oracle import target as ot
from tests.golden_io import TARGET_CASES, load_target_case


@pytest.mark.parametrize("name", TARGET_CASES)
def test_oracle_target_matches_reference_bit_exact(name):
    c = load_target_case(name)
    target, weight = ot.generate_target(c["keypoints"], c["image_size"], c["heatmap_size"], sigma=c["sigma"],
                                        use_udp=c["use_udp"], joint_weights=c["joint_weights"])
    assert target.dtype == np.float32 and target.shape == c["target"].shape
    assert np.array_equal(target.view(np.uint32), c["target"].view(np.uint32))
    assert np.array_equal(weight, c["target_weight"])


def test_known_answers():
    ka = ot.known_answers()
    assert ka["centre"] == 1.0
    assert abs(ka["corner"] - ka["corner_expected"]) < 1e-9
    assert ka["round_2_5"] == 2
    # half-to-even tie: x = 10 -> 10/4 = 2.5 -> mu_x = 2 (not 3)
    kp = np.array([[[10.0, 40.0, 1.0]]], dtype=np.float32)
    t, w = ot.generate_target(kp, [192, 256], [48, 64])
    assert np.unravel_index(np.argmax(t[0, 0]), t[0, 0].shape) == (10, 2)
    assert t[0, 0, 10, 2] == 1.0 and w[0, 0] == 1.0


def test_out_of_bounds_joint_gets_zero_weight():
    kp = np.array([[[-40.0, 10.0, 1.0], [100.0, 100.0, 0.0], [188.0, 252.0, 1.0]]], dtype=np.float32)
    t, w = ot.generate_target(kp, [192, 256], [48, 64])
    assert w.tolist() == [[0.0, 0.0, 1.0]]
    assert t[0, 0].max() == 0 and t[0, 1].max() == 0 and t[0, 2].max() == 1.0
