"""oracle/lpips.py against the reference's own LPIPS class (tests/golden/lpips.npz, made by tests/golden/make_golden_lpips.py)."""
import os
import sys

import numpy as np
import torch

import oracle.lpips as ol

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
G = np.load(os.path.join(HERE, "golden", "lpips.npz"))


def _vgg():
    from make_golden_lpips import vgg16_features_state
    return vgg16_features_state()


def test_constants_are_the_reference_s():
    assert np.allclose(G["shift"], ol.SHIFT) and np.allclose(G["scale"], ol.SCALE) and G["lin"].shape == (sum(ol.CHNS),)


def test_oracle_lpips_value_layers_and_gradient():
    vgg, lin = _vgg(), torch.from_numpy(G["lin"])
    for tag in ("a", "b", "c"):
        pred = torch.from_numpy(G[f"{tag}_pred"]).requires_grad_(True)
        targ = torch.from_numpy(G[f"{tag}_target"])
        val, layers = ol.lpips(2.0 * pred.permute(0, 3, 1, 2) - 1.0, 2.0 * targ.permute(0, 3, 1, 2) - 1.0, vgg, lin)
        assert np.abs(val.detach().numpy() - G[f"{tag}_val"]).max() < 1e-6
        # (the reference's `val = res[0]; val += res[l]` (L:106-109) adds IN PLACE: the first entry of its per-layer list is the total)
        assert np.abs(layers.detach().numpy()[1:] - G[f"{tag}_layers"][1:]).max() < 1e-6
        assert np.abs(G[f"{tag}_layers"][0] - G[f"{tag}_val"]).max() == 0 and np.abs(layers.detach().numpy().sum(0) - G[f"{tag}_val"]).max() < 1e-6
        loss = ol.lpips_loss(pred, targ, vgg, lin)
        assert abs(float(loss) - float(G[f"{tag}_loss"])) < 1e-6
        loss.backward()
        g = G[f"{tag}_grad"]
        assert np.abs(pred.grad.numpy() - g).max() < 1e-6 * max(1.0, np.abs(g).max()) + 1e-8
