"""Stage-3 training losses against the reference's own `get_loss` (tests/golden/make_golden_losses.py):
values and gradients of the sync-free formulation in hosnerf_amd.train.stage3_losses."""
import os

import numpy as np
import pytest
import torch

from hosnerf_amd.train import stage3_losses

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))


@pytest.mark.parametrize("tag", ["mix", "nofg", "t0"])
def test_stage3_losses_vs_reference(tag):
    p = tag + "_"
    t = lambda k: torch.from_numpy(np.asarray(G[p + k]))
    idx_fg = t("idx_fg")
    B, S = G[p + "pts_prev"].shape[:2]
    rgb = t("rgb").requires_grad_(True)
    hw_fg = t("hw").requires_grad_(True)
    pts_prev = t("pts_prev").requires_grad_(True)
    deform = t("deform").requires_grad_(True)
    hw = torch.zeros(B, S).masked_scatter(idx_fg[:, None].expand(B, S), hw_fg)        # rows of background rays are zero
    out = {"rgb": rgb, "idx_fg": idx_fg.to(torch.int32), "human_weights_sorted": hw, "observe_pts": t("observe"), "deform_pts_final": deform}
    batch = {"target_patches": t("targets"), "ray_grid": t("ray_grid"), "newsmpl_to_camera_prev": t("cam"), "intrinsics_prev": t("K")}
    if float(G[p + "time"]) > 0.005:                   # the flow set only exists for time > 0.005 (N:474, M:1703)
        out["deform_pts_prev_final"] = pts_prev
    total, parts = stage3_losses(out, batch)
    total.backward()
    assert abs(float(total.detach()) - float(G[p + "total"])) < 1e-7
    # the reference reports the WEIGHTED terms (M:1711-1716)
    for k, w in (("mse", 0.2), ("flow", 0.01), ("cycle", 0.01)):
        assert abs(w * float(parts[k]) - float(G[p + k])) < 1e-7 + 1e-5 * abs(float(G[p + k])), k
    g = lambda x: torch.zeros_like(x) if x.grad is None else x.grad
    for got, key in ((g(rgb), "g_rgb"), (g(deform), "g_deform"), (g(hw_fg), "g_hw"), (g(pts_prev), "g_pts_prev")):
        want = torch.from_numpy(G[p + key])
        if want.numel() == 0:
            continue
        assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())) + 1e-9, key
    if tag == "mix":
        assert float(parts["flow"]) > 0 and float(g(pts_prev).abs().max()) > 0
        # poison the rows that must not count: a background ray's flow points may project to infinity
        with torch.no_grad():
            bad = pts_prev.detach().clone()
            bad[~idx_fg] = 0.0
            cam = t("cam")
            bad[~idx_fg] = (-cam[:3, :3].T @ cam[:3, 3])                # the previous camera's centre: depth exactly 0
        out2 = {k: v.detach() for k, v in out.items()}
        out2["deform_pts_prev_final"] = bad.requires_grad_(True)
        total2, _ = stage3_losses(out2, batch)
        total2.backward()
        assert abs(float(total2.detach()) - float(total.detach())) < 1e-7 and bool(torch.isfinite(out2["deform_pts_prev_final"].grad).all())
