"""oracle/loss.py known answers + frozen goldens (parity unpinned by the reference)."""
import numpy as np

from oracle import loss as ol
from tests.golden import recipes
from tests.golden_io import load_npz


def test_loss_known_answer():
    pred = np.ones((2, 3, 4, 5), dtype=np.float32)
    target = np.zeros_like(pred)
    w = np.full((2, 3), 0.5, dtype=np.float32)
    assert ol.joints_mse(pred, target, w, use_target_weight=True) == np.float32(0.5)
    assert ol.joints_mse(pred, target) == np.float32(1.0)
    g = ol.joints_mse_grad(pred, target, w, use_target_weight=True)
    assert np.allclose(g, 2 * 0.5 / pred.size)


def test_loss_frozen_golden_and_torch_crosscheck():
    import torch
    g = load_npz("loss.npz")
    for name, (shape, seed) in recipes.LOSS_CASES.items():
        pred, target, w = recipes.loss_inputs(shape, seed)
        lp = ol.joints_mse(pred, target)
        lw = ol.joints_mse(pred, target, w, use_target_weight=True)
        assert lp == g[name + "/loss_plain"] and lw == g[name + "/loss_weighted"]
        tp = torch.tensor(pred, requires_grad=True)
        tl = (((tp - torch.tensor(target)) ** 2) * torch.tensor(w)[..., None, None]).mean()
        tl.backward()
        assert abs(float(tl.detach()) - float(lw)) < 1e-6
        gr = ol.joints_mse_grad(pred, target, w, use_target_weight=True)
        np.testing.assert_allclose(gr, tp.grad.numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_array_equal(gr[:, :, ::8, ::8], g[name + "/grad_sample"])
