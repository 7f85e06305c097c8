"""SURVEY row C4 / 8(f).2 on the device: hos_train_losses_{fwd,bwd} (through hosnerf_amd.train.stage{2,3}_losses) against the
reference's own `get_loss` vectors (tests/golden/losses.npz): values and gradients within 1e-6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))
DEV = "cuda"


def _t(p, k):
    return torch.from_numpy(np.asarray(G[p + k])).to(DEV)


def _check(total, parts, grads, p):
    assert abs(float(total.detach()) - float(G[p + "total"])) < 1e-6
    for k, w in (("mse", 0.2), ("flow", 0.01), ("cycle", 0.01)):
        assert abs(w * float(parts[k]) - float(G[p + k])) < 1e-7 + 1e-5 * abs(float(G[p + k])), k
    for x, key, want in grads:
        got = torch.zeros_like(x) if x.grad is None else x.grad
        if want.numel():
            assert float((got.cpu() - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())) + 1e-9, key


@pytest.mark.parametrize("tag", ["mix", "nofg", "t0"])
@pytest.mark.parametrize("dyn_count", [False, True])
def test_stage3_losses_vs_reference(tag, dyn_count):
    from hosnerf_amd.train import stage3_losses
    p = tag + "_"
    idx_fg = torch.from_numpy(G[p + "idx_fg"])
    B, S = G[p + "pts_prev"].shape[:2]
    rgb, pts_prev, deform = (_t(p, k).requires_grad_(True) for k in ("rgb", "pts_prev", "deform"))
    hw_fg = torch.from_numpy(G[p + "hw"])
    hw = torch.zeros(B, S).masked_scatter(idx_fg[:, None].expand(B, S), hw_fg).to(DEV).requires_grad_(True)   # bg rows are zero
    observe = _t(p, "observe")
    out = {"rgb": rgb, "idx_fg": idx_fg.to(torch.int32).to(DEV), "human_weights_sorted": hw, "observe_pts": observe, "deform_pts_final": deform}
    if dyn_count:       # fixed-capacity cycle buffers with the row count in device memory (rows past it hold garbage)
        n = observe.shape[0]
        pad = torch.full((9, 3), 7.0, device=DEV)
        deform = torch.cat([deform.detach(), pad], 0).requires_grad_(True)
        out.update(observe_pts=torch.cat([observe, -pad], 0), deform_pts_final=deform, cycle_count=torch.tensor([n], dtype=torch.int32, device=DEV))
    batch = {"target_patches": _t(p, "targets"), "ray_grid": _t(p, "ray_grid"), "newsmpl_to_camera_prev": _t(p, "cam"), "intrinsics_prev": _t(p, "K")}
    if float(G[p + "time"]) > 0.005:
        out["deform_pts_prev_final"] = pts_prev
    total, parts = stage3_losses(out, batch)
    total.backward()
    g_hw = torch.zeros(B, S) if hw.grad is None else hw.grad.cpu()
    want_hw = torch.from_numpy(G[p + "g_hw"])
    if want_hw.numel():
        assert float((g_hw[idx_fg] - want_hw).abs().max()) <= 1e-6 * max(1.0, float(want_hw.abs().max())) + 1e-9
    want_def = torch.from_numpy(G[p + "g_deform"])
    if dyn_count:
        assert float(deform.grad[want_def.shape[0]:].abs().max()) == 0.0
        got = deform.grad[:want_def.shape[0]].cpu()
        assert float((got - want_def).abs().max()) <= 1e-6 * max(1.0, float(want_def.abs().max())) + 1e-9
    _check(total, parts, [(rgb, "g_rgb", torch.from_numpy(G[p + "g_rgb"])), (pts_prev, "g_pts_prev", torch.from_numpy(G[p + "g_pts_prev"]))]
           + ([] if dyn_count else [(deform, "g_deform", want_def)]), p)


@pytest.mark.parametrize("tag", ["s2_mix", "s2_t0", "s2_one"])
def test_stage2_losses_vs_reference(tag):
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets, stage2_losses
    p = tag + "_"
    c = lambda k: torch.from_numpy(np.asarray(G[p + k]))
    host = prepare_patch_targets({"target_patches": c("targets"), "patch_masks": c("patch_masks"), "bgcolor": c("bgcolor"),
                                  "patch_div_indices": c("div"), "ray_grid": c("ray_grid"), "newsmpl_to_camera_prev": c("cam"),
                                  "intrinsics_prev": c("K")})
    batch = batch_to_device(host, DEV)
    rgb, w, pts_prev, deform = (_t(p, k).requires_grad_(True) for k in ("rgb", "weights", "pts_prev", "deform"))
    out = {"rgb": rgb, "weights": w, "observe_pts": _t(p, "observe"), "deform_pts_final": deform}
    if float(G[p + "time"]) > 0.005:
        out["deform_pts_prev_final"] = pts_prev
    total, parts = stage2_losses(out, batch)
    total.backward()
    _check(total, parts, [(x, k, torch.from_numpy(G[p + k])) for x, k in
                          ((rgb, "g_rgb"), (deform, "g_deform"), (w, "g_weights"), (pts_prev, "g_pts_prev"))], p)


def test_losses_deterministic_and_scaled_upstream():
    """Same inputs -> bit-identical totals (fixed summation order); the backward scales with the upstream gradient."""
    from hosnerf_amd import ops
    g = torch.Generator().manual_seed(3)
    B, S = 4096, 128
    rgb = torch.rand(B, 3, generator=g).to(DEV).requires_grad_(True)
    tgt = torch.rand(B, 3, generator=g).to(DEV)
    pts = (torch.randn(B, S, 3, generator=g) * 0.3).to(DEV).requires_grad_(True)
    w = (torch.rand(B, S, generator=g) * 0.02).to(DEV).requires_grad_(True)
    grid = torch.cat([torch.rand(B, 2, generator=g) * 60, torch.randn(B, 2, generator=g), (torch.rand(B, 1, generator=g) > 0.3).float()], -1).to(DEV)
    cam = torch.eye(4); cam[2, 3] = 3.0
    K = torch.tensor([[55.0, 0.0, 30.0], [0.0, 57.0, 28.0], [0.0, 0.0, 1.0]])
    obs = torch.randn(B * S // 3, 3, generator=g).to(DEV)
    dfm = (obs + 0.01).requires_grad_(True)
    vals = []
    for _ in range(3):
        t, parts = ops.train_losses(rgb, tgt, pts_prev=pts, weights=w, ray_grid=grid, cam_prev=cam.to(DEV), intrinsics_prev=K.to(DEV),
                                    observe=obs, deform=dfm)
        vals.append(parts.clone())
    assert torch.equal(vals[0], vals[1]) and torch.equal(vals[1], vals[2])
    (3.0 * t).backward()
    g3 = rgb.grad.clone()
    rgb.grad = pts.grad = w.grad = dfm.grad = None
    t2, _ = ops.train_losses(rgb, tgt, pts_prev=pts, weights=w, ray_grid=grid, cam_prev=cam.to(DEV), intrinsics_prev=K.to(DEV), observe=obs, deform=dfm)
    t2.backward()
    assert torch.allclose(g3, 3.0 * rgb.grad, rtol=1e-6, atol=0)
    # against torch autograd on the same device (the oracle's formulation)
    import oracle.losses as ol
    r2, p2, w2, d2 = (x.detach().clone().requires_grad_(True) for x in (rgb, pts, w, dfm))
    mse = torch.mean((r2 - tgt) ** 2)
    flow = ol.flow_func(grid, cam.to(DEV), K.to(DEV), w2, p2)
    cyc = torch.mean(torch.sum((obs - d2) ** 2, 1) / 2.0)
    ref = 0.2 * mse + 0.01 * flow + 0.01 * cyc
    ref.backward()
    assert abs(float(ref) - float(t2)) < 1e-6
    for a, b in ((rgb.grad, r2.grad), (pts.grad, p2.grad), (w.grad, w2.grad)):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-10


def test_stage3_mse_is_against_target_patches_not_target_rgbs():
    """M:1598-1602 + `_unpack_imgs` (M:41-50): stage 3 compares the plain reshape of the rendered rays with `target_patches`.  A patch that
    leaves the subject's box keeps its size (core/data/human_nerf/train.py:322-330): its outside pixels are rendered by a substituted box
    ray, so the item's `target_rgbs` (colour of that ray's own pixel) differs from `target_patches` there -- the loss must not use it."""
    import oracle.losses as ol
    from hosnerf_amd.train import stage3_losses
    g = torch.Generator().manual_seed(5)
    B = 2 * 16 * 16
    rgb = torch.rand(B, 3, generator=g).to(DEV).requires_grad_(True)
    patches = torch.rand(2, 16, 16, 3, generator=g).to(DEV)
    rgbs = patches.reshape(-1, 3).clone()
    rgbs[40:90] = torch.rand(50, 3, generator=g).to(DEV)                       # the substituted rays' own colours
    obs = torch.randn(64, 3, generator=g).to(DEV)
    dfm = (obs + 0.02).requires_grad_(True)
    out = {"rgb": rgb, "idx_fg": torch.ones(B, dtype=torch.int32, device=DEV), "observe_pts": obs, "deform_pts_final": dfm}
    batch = {"target_patches": patches, "target_rgbs": rgbs, "patch_masks": torch.ones(2, 16, 16, dtype=torch.bool, device=DEV),
             "patch_div_indices": torch.tensor([0, 256, 512]), "bgcolor": torch.zeros(3, device=DEV), "mse_const": 0.0, "mse_count": float(patches.numel())}
    total, parts = stage3_losses(out, batch)
    total.backward()
    r2, d2 = rgb.detach().clone().requires_grad_(True), dfm.detach().clone().requires_grad_(True)
    ref, pref = ol.stage3_losses({"rgb": r2, "idx_fg": torch.ones(B, dtype=torch.bool, device=DEV), "observe_pts": obs, "deform_pts_final": d2}, batch, 0.0)
    ref.backward()
    assert abs(float(total) - float(ref)) < 1e-7 and abs(float(parts["mse"]) - float(pref["mse"])) < 1e-7
    assert float((rgb.grad - r2.grad).abs().max()) <= 1e-6 * float(r2.grad.abs().max())
    wrong = float(torch.mean((rgb.detach() - rgbs) ** 2))
    assert abs(wrong - float(pref["mse"])) > 1e-4                               # the two targets do differ in this item
