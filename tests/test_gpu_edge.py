"""Ragged and degenerate batches through the whole renderer (the reference has no unit tests for these; its code simply
works for any batch): a single ray, batch sizes that are no multiple of any tile, rays that all miss the subject, a time
exactly on a state boundary.  Outputs against the oracle on the same inputs, gradients finite and against the oracle."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.background as ob
import oracle.human as oh
from hosnerf_amd import synth


def _basedir():
    d = tempfile.mkdtemp(prefix="hos_edge_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


@pytest.mark.parametrize("B", [1, 3, 65, 257])
def test_stage1_ragged_batches(B):
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import stage1_loss
    dev = torch.device("cuda")
    model = MipNeRF360(_basedir(), opaque_background=True)
    sd0 = synth.background_state_dict(777, 2)
    model.load_state_dict(sd0, strict=False)
    model = model.to(dev)
    b = synth.stage1_batch(B, seed=100 + B, time=0.4)              # time on the state boundary (<= tau + 1e-5 -> state 1)
    g = torch.Generator().manual_seed(B)
    jit = [torch.rand(B, generator=g) for _ in range(3)]
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    rend_o, hist_o = ob.mipnerf360_forward(sd, b, 0.3, True, 0.1, 1e6, transitions_times=[0.4], jitters=[j.view(B, 1) for j in jit])
    loss_o, _ = ob.stage1_loss(rend_o[-1]["rgb"], b["target"], hist_o)
    loss_o.backward()
    gb = {k: v.to(dev) for k, v in b.items()}
    rend, hist = model(gb, 0.3, True, True, 0.1, 1e6, jitters=[j.to(dev) for j in jit])
    loss, _ = stage1_loss(rend[-1]["rgb"], gb["target"], hist)
    loss.backward()
    assert rend[-1]["rgb"].shape == (B, 3)
    assert float((rend[-1]["rgb"].cpu() - rend_o[-1]["rgb"]).abs().max()) < 1e-4
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 1e-4 * max(1.0, abs(float(loss_o.detach())))
    for name in ("mlps.2.pts_linear.5.weight", "mlps.0.pts_linear.2.weight"):
        go = sd[name].grad
        gh = dict(model.named_parameters())[name].grad.cpu()
        assert bool(torch.isfinite(gh).all())
        assert float((gh.reshape(go.shape) - go).abs().max()) < 1e-2 * float(go.abs().max()) + 1e-9, name


@pytest.mark.parametrize("B,miss", [(1, False), (5, False), (7, True)])
def test_stage3_ragged_and_all_background(B, miss):
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    from hosnerf_amd.train import batch_to_device, stage3_losses
    dev = torch.device("cuda")
    cfg = default_cfg(_basedir())
    cfg.perturb = 0.0
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    hos = hos.to(dev)
    b = synth.human_batch(B, seed=200 + B, time=0.5, is_train=True, iter_val=3e5)
    if miss:                                                     # samples far outside the subject: no foreground ray at all
        b["near"] += 50.0
        b["far"] += 50.0
    b["ray_grid"] = torch.cat([torch.rand(B, 2) * 100, torch.randn(B, 2), torch.ones(B, 1)], -1)
    b["newsmpl_to_camera_prev"] = torch.eye(4)
    b["newsmpl_to_camera_prev"][2, 3] = 3.0
    b["intrinsics_prev"] = torch.tensor([[500.0, 0, 50], [0, 500.0, 50], [0, 0, 1]])
    g = torch.Generator().manual_seed(B)
    jit = [torch.rand(B, generator=g) for _ in range(3)]
    gb = batch_to_device(b, dev)
    out = hos.render(gb, randomized=True, is_train=True, jitters=[j.to(dev) for j in jit])
    loss, parts = stage3_losses(out, gb)
    loss.backward()
    assert out["rgb"].shape == (B, 3) and bool(torch.isfinite(out["rgb"]).all()) and bool(torch.isfinite(loss))
    assert bool(torch.isfinite(hos.human.flat_grad).all()) and bool(torch.isfinite(hos.model.flat_grad).all())
    if miss:
        assert int(out["idx_fg"].sum()) == 0 and float(parts["flow"]) == 0.0
    # oracle on the same inputs
    bsd, hsd = synth.background_state_dict(777, 2), synth.human_state_dict(777, 2)
    bb = {"rays_o": b["rays_o_bkg"], "rays_d": b["rays_d_bkg"], "viewdirs": b["viewdirs_bkg"], "radii": b["radii"], "times": b["time"]}
    with torch.no_grad():
        _, hist = ob.mipnerf360_forward(bsd, bb, 1.0, True, 0.1, 1e6, transitions_times=[0.4], jitters=[j.view(B, 1) for j in jit], render=False)
        human = oh.human_forward(hsd, b, transitions_times=[0.4])
        rgb_o, fg_o = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human, bb["rays_o"], bb["rays_d"],
                                          b["newsmpl_to_scale_world"])[:2]
    assert np.array_equal(out["idx_fg"].bool().cpu().numpy(), fg_o.numpy())
    assert float((out["rgb"].cpu() - rgb_o).abs().max()) < 1e-4
