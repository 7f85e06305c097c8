"""Training losses of stages 2 and 3: the oracle's restatement (oracle/losses.py) against the reference's
own `get_loss` (tests/golden/make_golden_losses.py -> tests/golden/losses.npz), values and gradients.  The HIP kernels
are checked against the same vectors in tests/test_gpu_losses.py."""
import os

import numpy as np
import pytest
import torch

import oracle.losses as ol

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))


def _t(p, k):
    return torch.from_numpy(np.asarray(G[p + k]))


def _check(total, parts, grads, p):
    assert abs(float(total.detach()) - float(G[p + "total"])) < 1e-7
    for k, w in (("mse", 0.2), ("flow", 0.01), ("cycle", 0.01)):            # the reference reports the WEIGHTED terms
        assert abs(w * float(parts[k]) - float(G[p + k])) < 1e-7 + 1e-5 * abs(float(G[p + k])), k
    for x, key in grads:
        got = torch.zeros_like(x) if x.grad is None else x.grad
        want = torch.from_numpy(G[p + key])
        if want.numel():
            assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())) + 1e-9, key


@pytest.mark.parametrize("tag", ["mix", "nofg", "t0"])
def test_stage3_losses_vs_reference(tag):
    p = tag + "_"
    rgb, hw, pts_prev, deform = (_t(p, k).requires_grad_(True) for k in ("rgb", "hw", "pts_prev", "deform"))
    N, P = G[p + "targets"].shape[:2]
    out = {"rgb": rgb, "idx_fg": _t(p, "idx_fg"), "human_weights_onlyfg": hw, "deform_pts_prev_final": pts_prev,
           "observe_pts": _t(p, "observe"), "deform_pts_final": deform}
    batch = {"target_patches": _t(p, "targets"), "patch_masks": torch.ones(N, P, P, dtype=torch.bool), "bgcolor": torch.zeros(3),
             "patch_div_indices": [0, P * P, 2 * P * P], "ray_grid": _t(p, "ray_grid"), "newsmpl_to_camera_prev": _t(p, "cam"),
             "intrinsics_prev": _t(p, "K")}
    total, parts = ol.stage3_losses(out, batch, float(G[p + "time"]))
    total.backward()
    _check(total, parts, ((rgb, "g_rgb"), (deform, "g_deform"), (hw, "g_hw"), (pts_prev, "g_pts_prev")), p)


@pytest.mark.parametrize("tag", ["s2_mix", "s2_t0", "s2_one"])
def test_stage2_losses_vs_reference(tag):
    p = tag + "_"
    rgb, w, pts_prev, deform = (_t(p, k).requires_grad_(True) for k in ("rgb", "weights", "pts_prev", "deform"))
    out = {"rgb": rgb, "weights": w, "deform_pts_prev_final": pts_prev, "observe_pts": _t(p, "observe"), "deform_pts_final": deform}
    batch = {"target_patches": _t(p, "targets"), "patch_masks": _t(p, "patch_masks"), "bgcolor": _t(p, "bgcolor"),
             "patch_div_indices": G[p + "div"].tolist(), "ray_grid": _t(p, "ray_grid"), "newsmpl_to_camera_prev": _t(p, "cam"),
             "intrinsics_prev": _t(p, "K")}
    total, parts = ol.stage2_losses(out, batch, float(G[p + "time"]))
    total.backward()
    _check(total, parts, ((rgb, "g_rgb"), (deform, "g_deform"), (w, "g_weights"), (pts_prev, "g_pts_prev")), p)


def test_patch_target_preparation_matches_unpack():
    """hosnerf_amd.train.prepare_patch_targets (host-side constants of the patch MSE) against `_unpack_imgs` + `img2mse`."""
    from hosnerf_amd.train import prepare_patch_targets
    p = "s2_mix_"
    batch = {"target_patches": _t(p, "targets"), "patch_masks": _t(p, "patch_masks"), "bgcolor": _t(p, "bgcolor"),
             "patch_div_indices": G[p + "div"].tolist()}
    pb = prepare_patch_targets(batch)
    rgb = _t(p, "rgb")
    mse = (((rgb - pb["target_rgbs"]) ** 2).sum() + pb["mse_const"]) / pb["mse_count"]
    ref = torch.mean((ol.unpack_imgs(rgb, batch["patch_masks"], batch["bgcolor"] / 255.0, batch["target_patches"], batch["patch_div_indices"])
                      - batch["target_patches"]) ** 2)
    assert abs(float(mse) - float(ref)) < 1e-7
    assert abs(0.2 * float(mse) - float(G[p + "mse"])) < 1e-7
