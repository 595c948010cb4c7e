"""CPU: the depth-alignment oracle against golden vectors PRODUCED BY THE REFERENCE ITSELF
(tests/golden/enhancement_golden.npz <- tests/golden/make_golden_enhancement.py, which imports
/root/reference/nerf_vo/enhancement/enhancement_module.py and runs EnhancementModule.step on CPU).
This pins oracle/enhancement.py."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ("a", "b", "fallback")


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(os.path.join(HERE, "golden", "enhancement_golden.npz")))


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_outputs_bit_exactly(golden, case):
    from oracle import enhancement as E

    out = E.enhance_depth(torch.from_numpy(golden[f"{case}_mono_depth"]), torch.from_numpy(golden[f"{case}_patches"]),
                          torch.from_numpy(golden[f"{case}_noise"]))
    assert np.array_equal(out.numpy(), golden[f"{case}_ref_frames_depth"])


def test_outlier_removal_keeps_five_sixths_and_fallback_keeps_all(golden):
    from oracle import enhancement as E

    p = E.dpvo_remove_outliers(torch.from_numpy(golden["a_patches"]), torch.from_numpy(golden["a_noise"]))
    assert p.shape[1] == 80  # int(96 * 5 / 6)
    q = E.dpvo_remove_outliers(torch.from_numpy(golden["fallback_patches"]), torch.from_numpy(golden["fallback_noise"]))
    assert q.shape[1] == 90  # 74 survivors per frame != int(90 * 5 / 6): the reference's `except` path keeps all
    assert (q >= 1e-3).all()  # ... after replacing every element < 1e-3 with the global mean


def test_reference_side_effects_recorded_in_golden(golden):
    """colours / 255, the nerfstudio axis flip of the poses and the normal re-normalisation are host logic of
    the step (ref: enhancement_module.py:47, 102-104, 113-114); the mirror's CPU-checkable part."""
    for case in CASES:
        np.testing.assert_array_equal(golden[f"{case}_ref_frames_color"],
                                      golden[f"{case}_frames_color_u8"].astype(np.float32) / np.float32(255.0))
        flipped = golden[f"{case}_extrinsics_in"].copy()
        flipped[:, :3, 1:3] *= -1
        np.testing.assert_array_equal(golden[f"{case}_ref_extrinsics"], flipped)
    n = torch.nn.functional.normalize(torch.from_numpy(golden["a_mono_normal"]) * 2.0 - 1.0, p=2, dim=1)
    np.testing.assert_array_equal(golden["a_ref_frames_normal"], n.numpy())
