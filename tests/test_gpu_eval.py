"""Full-frame evaluation loop: device-side frame set-up + chunked render of `free_view` (M:1293-1494)
against the oracle's chunk-free restatement of the same frame, and the invariances the loop must have."""
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


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


@pytest.fixture(scope="module")
def hos(dev):
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    m = HOSNeRF(default_cfg(d))
    m.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    m.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    return m.to(dev)


def make_frame(dev, H, W, seed=41):
    from hosnerf_amd import eval as ev
    hb = synth.human_batch(8, seed=seed, time=0.5, is_train=False, iter_val=3e5)
    K, E, Ec = synth.eval_camera(H, W, hb)
    bbox = {"min_xyz": hb["dst_bbox_min_xyz"].numpy(), "max_xyz": hb["dst_bbox_max_xyz"].numpy()}
    fr = ev.frame_rays(H, W, K, E, bbox, Ec, device=dev)
    for k in ev.FRAME_KEYS:
        fr[k] = hb[k].to(dev) if isinstance(hb[k], torch.Tensor) else hb[k]
    g = torch.Generator().manual_seed(seed)
    fr["target_rgbs"] = torch.rand(int(fr["ray_mask"].sum()), 3, generator=g)
    fr["target_rgbs_bkg"] = torch.rand(int(fr["ray_mask_bkg"].sum()), 3, generator=g)
    return fr, hb


def test_render_frame_vs_oracle(dev, hos):
    from hosnerf_amd import eval as ev
    H, W = 20, 16
    fr, hb = make_frame(dev, H, W)
    n_fg, n_bg = int(fr["ray_mask"].sum()), int(fr["ray_mask_bkg"].sum())
    assert n_fg > 40 and n_bg > 40 and n_fg + n_bg == H * W
    rendered = ev.render_frame(hos, fr, chunk_bkg=8192, randomized=False)
    assert rendered.shape == (H * W, 3) and hos.training and hos.cfg.perturb == 1.0      # test_end restored the modes

    # the oracle on the same frame, no chunking
    bsd, hsd = synth.background_state_dict(777, 2), synth.human_state_dict(777, 2)
    c = lambda k: fr[k].detach().cpu()
    with torch.no_grad():
        bb = {"rays_o": c("rays_o_bkg"), "rays_d": c("rays_d_bkg"), "viewdirs": c("viewdirs_bkg"), "radii": c("radii"), "times": hb["time"]}
        _, hist = ob.mipnerf360_forward(bsd, bb, 1.0, False, 0.1, 1e6, transitions_times=[0.4], render=False)
        b = dict(hb)
        b.update(rays=c("rays"), near=c("near"), far=c("far"), rays_o_bkg=c("rays_o_bkg"), rays_d_bkg=c("rays_d_bkg"),
                 viewdirs_bkg=c("viewdirs_bkg"), radii=c("radii"), is_train=False)
        human = oh.human_forward(hsd, b, transitions_times=[0.4])
        rgb_fg = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human, bb["rays_o"], bb["rays_d"],
                                     hb["newsmpl_to_scale_world"])[0]
        bo = {"rays_o": c("rays_o_bkg_only"), "rays_d": c("rays_d_bkg_only"), "viewdirs": c("viewdirs_bkg_only"),
              "radii": c("radii_bkg_only"), "times": hb["time"]}
        _, hist_o = ob.mipnerf360_forward(bsd, bo, 1.0, False, 0.1, 1e6, transitions_times=[0.4], render=False)
        rgb_bg = oh.raw2outputs(hist_o[-1]["rgb"], hist_o[-1]["density"], hist_o[-1]["tdist"][..., :-1], bo["rays_d"],
                                torch.ones_like(hist_o[-1]["density"]))[0]
    want = (hb["bgcolor"] / 255.0).expand(H * W, 3).clone()
    want[c("ray_mask")] = rgb_fg
    want[c("ray_mask_bkg")] = rgb_bg
    err = float((rendered.cpu() - want).abs().max())
    assert err < 1e-4, f"north-star: 1e-4 RGB L-inf, got {err}"
    # some rays of the box really carry the subject (else the merge path would be untested)
    fg_used = hos_fg_fraction(hos, fr)
    assert fg_used > 0.05

    # PSNR / truth scatter against the numpy formulas of M:101-112, :1456-1460
    truth = ev.truth_frame(fr)
    t_np = np.full((H * W, 3), hb["bgcolor"].numpy() / 255.0, dtype="float32")
    t_np[c("ray_mask").numpy()] = fr["target_rgbs"].numpy()
    t_np[c("ray_mask_bkg").numpy()] = fr["target_rgbs_bkg"].numpy()
    assert np.array_equal(truth.cpu().numpy(), t_np)
    r_np = rendered.cpu().numpy()
    mse = np.mean((r_np - t_np) ** 2)
    assert abs(ev.psnr_metric(rendered, truth) - (-10 * np.log(mse) / np.log(10))) < 1e-4
    img = ev.to_8b_image(rendered.view(H, W, 3))
    assert img.dtype == torch.uint8 and np.array_equal(img.cpu().numpy(), (255.0 * np.clip(r_np, 0.0, 1.0)).astype(np.uint8).reshape(H, W, 3))


def hos_fg_fraction(hos, fr):
    from hosnerf_amd import eval as ev
    with ev.evaluating(hos):
        b = {k: fr[k] for k in ev.FRAME_KEYS}
        b.update({k: fr[k] for k in ("rays", "near", "far", "rays_o_bkg", "rays_d_bkg", "viewdirs_bkg", "radii")})
        b["is_train"] = False
        out = hos.render(b, randomized=False, is_train=False, with_cycle=False)
    return float(out["idx_fg"].float().mean())


def test_render_frame_invariances(dev, hos):
    """Chunk size, prologue caching and the cycle set must not change a pixel: rows of the MLP GEMMs are independent and
    the prologue depends on the frame only."""
    from hosnerf_amd import eval as ev
    H, W = 36, 28
    fr, _ = make_frame(dev, H, W, seed=43)
    a = ev.render_frame(hos, fr, chunk_bkg=8192)
    b = ev.render_frame(hos, fr, chunk_bkg=97)                    # ragged chunks
    c = ev.render_frame(hos, fr, chunk_bkg=8192, cache_prologue=False)
    assert float((a - b).abs().max()) < 1e-6
    assert torch.equal(a, c)
    # a frame that misses the subject entirely: background-only path for every pixel, no human call
    fr2 = dict(fr)
    hb = synth.human_batch(8, seed=43, time=0.5, is_train=False, iter_val=3e5)
    K, E, Ec = synth.eval_camera(H, W, hb)
    far_box = {"min_xyz": np.array([50.0, 50.0, 50.0]), "max_xyz": np.array([51.0, 51.0, 51.0])}
    fr2.update(ev.frame_rays(H, W, K, E, far_box, Ec, device=dev))
    assert int(fr2["ray_mask"].sum()) == 0
    d = ev.render_frame(hos, fr2)
    assert d.shape == (H * W, 3) and bool(torch.isfinite(d).all())
    # ... and its pixels equal the background-only render of the same camera rays
    bb = {"rays_o": fr2["rays_o_bkg_only"], "rays_d": fr2["rays_d_bkg_only"], "viewdirs": fr2["viewdirs_bkg_only"],
          "radii": fr2["radii_bkg_only"], "times": fr2["time"]}
    with ev.evaluating(hos):
        e = hos.render_bkg_only(bb)
    assert float((d - e).abs().max()) < 1e-6


def test_training_item_on_device(dev, hos):
    """the stage-3 training item (frame rays -> box test -> random patches) built on the device gives the
    renderer the reference's batch keys; the patch choice does not depend on where the masks live."""
    from hosnerf_amd import eval as ev, rays as R
    H, W, P, N = 72, 64, 16, 2
    fr, hb = make_frame(dev, H, W, seed=47)
    n_box = int(fr["ray_mask"].sum())
    g = torch.Generator().manual_seed(1)
    img = torch.rand(H, W, 3, generator=g).to(dev)
    fr["ray_img"] = img.view(-1, 3)[fr["ray_mask"]]
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    subject = (((yy - H / 2) / (H / 4)) ** 2 + ((xx - W / 2) / (W / 6)) ** 2 < 1.0)
    np.random.seed(5)
    item = R.sample_patch_rays(fr, img, subject.to(dev), N, P, 0.8)
    np.random.seed(5)
    cpu_fr = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in fr.items()}
    item_cpu = R.sample_patch_rays(cpu_fr, img.cpu(), subject, N, P, 0.8)
    assert item["rays"].shape == (2, N * P * P, 3) and item["near"].shape == (N * P * P, 1)
    for k in ("rays", "near", "far", "rays_o_bkg", "radii", "target_patches", "target_rgbs"):
        assert torch.equal(item[k].cpu(), item_cpu[k]), k
    assert item["patch_div_indices"].tolist() == [0, P * P, 2 * P * P] and bool(item["patch_masks"].all())
    assert item["rays_o_bkg_only"].shape[0] == H * W - n_box                  # frame-level keys pass through unchanged
    b = dict(item)
    b["is_train"] = False
    with ev.evaluating(hos):
        out = hos.render(b, randomized=False, is_train=False, with_cycle=False)
    assert out["rgb"].shape == (N * P * P, 3) and bool(torch.isfinite(out["rgb"]).all())


def test_render_frame_with_poisoned_allocations(dev, hos):
    """The inference loop under NaN-poisoned `torch.empty` (scripts/soak_poison.py): every pixel finite and equal to the normal
    run -- no kernel of the evaluation path reads memory nobody wrote (ragged chunks included)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import soak_poison as sp
    from hosnerf_amd import eval as ev
    fr, _ = make_frame(dev, 36, 28, seed=43)
    ref = ev.render_frame(hos, fr, chunk_bkg=300)
    torch.empty, torch.empty_like = sp.pempty, sp.pempty_like
    try:
        got = ev.render_frame(hos, fr, chunk_bkg=300)
    finally:
        torch.empty, torch.empty_like = sp._empty, sp._empty_like
    assert bool(torch.isfinite(got).all())
    assert float((got - ref).abs().max()) == 0.0
