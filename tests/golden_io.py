"""Helpers to read the committed golden fixtures."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TARGET_CASES = ["target_plain_64x48", "target_udp_64x48", "target_plain_96x72_s3",
                "target_udp_96x72_s3", "target_plain_jw"]


def load_target_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    shape = tuple(int(v) for v in z["target_shape"])
    target = np.zeros(int(np.prod(shape)), dtype=np.float32)
    target[z["target_nz_idx"]] = z["target_nz_val"]
    jw = z["joint_weights"]
    return dict(
        keypoints=z["keypoints"], target=target.reshape(shape), target_weight=z["target_weight"],
        sigma=float(z["sigma"]), use_udp=bool(z["use_udp"]),
        image_size=[int(v) for v in z["image_size"]], heatmap_size=[int(v) for v in z["heatmap_size"]],
        joint_weights=None if jw.size == 0 else jw)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
