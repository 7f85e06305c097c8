"""hosnerf_amd.lpips.LPIPS (csrc/hos_lpips.hip + the GEMM entry points) against the reference's own LPIPS class (tests/golden/lpips.npz:
value and gradient w.r.t. the prediction) and, for patches cut by the box, against the oracle on the unpacked image."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle.lpips as ol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
G = np.load(os.path.join(HERE, "golden", "lpips.npz"))
DEV = "cuda"


@pytest.fixture(scope="module")
def net():
    from make_golden_lpips import vgg16_features_state
    from hosnerf_amd.lpips import LPIPS
    return LPIPS().load_vgg16_features(vgg16_features_state(), DEV).load_lin(G["lin"], DEV)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_value_and_gradient_vs_the_reference_class(net, tag):
    pred, targ = torch.from_numpy(G[f"{tag}_pred"]), torch.from_numpy(G[f"{tag}_target"])
    n, P = pred.shape[0], pred.shape[1]
    rgb = pred.reshape(-1, 3).to(DEV).requires_grad_(True)
    idx = torch.arange(n * P * P, dtype=torch.int32, device=DEV)
    loss = net.loss(rgb, targ.to(DEV), idx, torch.zeros(3, device=DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(G[f"{tag}_loss"])) < 2e-6 * max(1.0, abs(float(G[f"{tag}_loss"]))), (float(loss), float(G[f"{tag}_loss"]))
    g, ref = rgb.grad.cpu().numpy().reshape(G[f"{tag}_grad"].shape), G[f"{tag}_grad"]
    assert np.abs(g - ref).max() < 2e-5 * np.abs(ref).max(), (np.abs(g - ref).max(), np.abs(ref).max())


def test_patches_cut_by_the_box_vs_oracle(net):
    """Stage-2 form: some patch pixels have no ray and are filled with the background colour (model.py:41-50); the rays come patch
    after patch.  Upstream gradient 1.0 of the loss weight included (a scaled loss)."""
    from make_golden_lpips import vgg16_features_state
    from hosnerf_amd.lpips import patch_ray_index
    rs = np.random.RandomState(3)
    n, P = 3, 32
    masks = rs.uniform(size=(n, P, P)) > 0.2
    masks[1, :, :7] = False
    B = int(masks.sum())
    rgb = torch.from_numpy(rs.uniform(0, 1, size=(B, 3)).astype(np.float32))
    targ = torch.from_numpy(rs.uniform(0, 1, size=(n, P, P, 3)).astype(np.float32))
    bg = torch.tensor([30.0, 200.0, 120.0])
    # oracle on the unpacked image
    r64 = rgb.clone().requires_grad_(True)
    img = (bg / 255.0).expand(n, P, P, 3).clone()
    img[torch.from_numpy(masks)] = r64
    lo = 0.7 * ol.lpips_loss(img, targ, vgg16_features_state(), torch.from_numpy(G["lin"]))
    lo.backward()
    rg = rgb.to(DEV).requires_grad_(True)
    idx = patch_ray_index(torch.from_numpy(masks).to(DEV))
    assert int((idx >= 0).sum()) == B and int(idx.max()) == B - 1
    loss = 0.7 * net.loss(rg, targ.to(DEV), idx, bg.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(lo)) < 2e-6 * max(1.0, abs(float(lo)))
    ref = r64.grad.numpy()
    assert np.abs(rg.grad.cpu().numpy() - ref).max() < 2e-5 * np.abs(ref).max()


def test_unloaded_module_raises():
    from hosnerf_amd.lpips import LPIPS
    with pytest.raises(RuntimeError):
        LPIPS().loss(torch.zeros(4, 3, device=DEV), torch.zeros(1, 2, 2, 3, device=DEV), torch.arange(4, dtype=torch.int32, device=DEV), torch.zeros(3, device=DEV))
