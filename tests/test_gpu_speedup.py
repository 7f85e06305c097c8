"""north_star target: >= 10x the reference PyTorch path in train rays/s at 1xMI355X.  The reference itself cannot travel
to the GPU box, so its op graph is represented by the oracle (fixture-verified restatement) run as plain PyTorch-ROCm
ops on device tensors -- same model, losses, clip and Adam.  Writes the measured numbers to
gpurun_out/speedup_vs_torch.json (copied to profiles/ by hand)."""
import json
import os
import tempfile
import time

import pytest
import torch

import oracle.background as ob
import oracle.human as oh
from hosnerf_amd import synth

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "speedup_vs_torch.json")


def _record(key, value):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    d = json.load(open(OUT)) if os.path.exists(OUT) else {}
    d[key] = value
    json.dump(d, open(OUT, "w"), indent=1)


def _time(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _basedir():
    d = tempfile.mkdtemp(prefix="hos_speed_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


def test_stage1_vs_torch_rocm():
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import FusedAdam, stage1_loss
    dev = torch.device("cuda")
    B = 1024
    batch = {k: v.to(dev) for k, v in synth.stage1_batch(B, seed=777).items()}
    # the reference op graph as PyTorch-ROCm ops
    sd = {k: v.to(dev).requires_grad_(True) for k, v in synth.background_state_dict(777, 2).items()}
    params = list(sd.values())
    topt = torch.optim.Adam(params, lr=2e-3)

    def torch_step():
        topt.zero_grad()
        rend, hist = ob.mipnerf360_forward(sd, batch, 0.5, True, 0.1, 1e6, transitions_times=[0.4])
        loss, _ = ob.stage1_loss(rend[-1]["rgb"], batch["target"], hist)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.001)
        topt.step()

    model = MipNeRF360(_basedir(), opaque_background=True)
    model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    model = model.to(dev)
    opt = FusedAdam(model, lr=2e-3, max_grad_norm=0.001)
    hb = dict(batch)
    hb["times"] = 0.5

    # ---- FULL-SIZE parity first (BASELINE configs[1]: 1024 rays, 64/64/32 samples): same weights, rays and per-ray
    # jitters through the oracle's op graph on this device and through the HIP path; the north-star tolerance on RGB.
    g = torch.Generator().manual_seed(11)
    jit = [torch.rand(B, generator=g).to(dev) for _ in range(3)]
    topt.zero_grad()
    rend_o, hist_o = ob.mipnerf360_forward(sd, batch, 0.5, True, 0.1, 1e6, transitions_times=[0.4], jitters=[j.view(B, 1).cpu() for j in jit])
    loss_o, _ = ob.stage1_loss(rend_o[-1]["rgb"], batch["target"], hist_o)
    loss_o.backward()
    opt.zero_grad()
    rend_h, hist_h = model(hb, 0.5, True, True, 0.1, 1e6, jitters=jit)
    loss_h, _ = stage1_loss(rend_h[-1]["rgb"], hb["target"], hist_h)
    loss_h.backward()
    par = {"rays": B, "rgb_linf": float((rend_h[-1]["rgb"] - rend_o[-1]["rgb"]).abs().max()),
           "loss_rel": abs(float(loss_h.detach()) - float(loss_o.detach())) / abs(float(loss_o.detach()))}
    assert par["rgb_linf"] < 1e-4, par
    for lvl in range(3):
        par[f"tdist_linf_level{lvl}"] = float((hist_h[lvl]["tdist"] - hist_o[lvl]["tdist"]).abs().max() / hist_o[lvl]["tdist"].abs().max())
    assert par["loss_rel"] < 1e-4, par
    hg = {k: v.grad for k, v in model.named_parameters()}
    for name in ("mlps.2.pts_linear.3.weight", "mlps.2.pts_linear.7.weight", "mlps.2.density_layer.weight" if "mlps.2.density_layer.weight" in hg else "mlps.2.pts_linear.0.weight",
                 "mlps.0.pts_linear.1.weight", "mlps.1.pts_linear.2.weight"):
        go, gh = sd[name].grad, hg[name]
        par["grad " + name] = float((gh.reshape(go.shape) - go).abs().max()) / max(1e-20, float(go.abs().max()))
        assert par["grad " + name] < 5e-3, par
    _record("stage1_fullsize_parity", par)
    del rend_o, hist_o, rend_h, hist_h, hg
    opt.zero_grad()

    t_torch = _time(torch_step, 2, 5)

    def hip_step():
        opt.zero_grad()
        rend, hist = model(hb, 0.5, True, True, 0.1, 1e6)
        loss, _ = stage1_loss(rend[-1]["rgb"], hb["target"], hist)
        loss.backward()
        opt.step(2e-3)

    t_hip = _time(hip_step, 3, 10)
    _record("stage1", {"rays": B, "torch_rocm_rays_per_s": B / t_torch, "hip_eager_rays_per_s": B / t_hip, "speedup": t_torch / t_hip})
    assert t_torch / t_hip > 3.0, (t_torch, t_hip)


def test_stage2_vs_torch_rocm():
    """Human-object branch (stage-2 style step: forward with flow + cycle sets, backward, Adam), 2048 rays x 128."""
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import FusedAdam, human_lr_ranges
    dev = torch.device("cuda")
    B = 2048
    b = synth.human_batch(B, seed=777, time=0.5, is_train=True, iter_val=3e5)
    from hosnerf_amd.train import batch_to_device
    gb = batch_to_device(b, dev)           # control scalars (time, iter_val) stay on the host: no round trip per step
    # the baseline leg gets what the reference's training_step gives its network: `cpu_data_to_gpu` moves every tensor of
    # the item, control scalars included (M:1507), and the network reads them back
    gb_ref = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    t_rand = torch.rand(B, 128, generator=torch.Generator().manual_seed(4)).to(dev)
    sd = {k: v.to(dev).requires_grad_(True) for k, v in synth.human_state_dict(777, 2).items()}
    params = list(sd.values())
    topt = torch.optim.Adam(params, lr=5e-4)

    def loss_of(out):
        loss = (out["human_rgb"] ** 2).mean() + (out["human_density"] ** 2).mean() * 1e-3
        if "deform_pts_prev_final" in out:
            loss = loss + (out["deform_pts_prev_final"] ** 2).mean() * 1e-3 + (out["deform_pts_final"] ** 2).mean() * 1e-3
        return loss

    def torch_step():
        topt.zero_grad()
        out = oh.human_forward(sd, gb_ref, transitions_times=[0.4], t_rand=t_rand, stage=3)
        loss_of(out).backward()
        topt.step()

    cfg = default_cfg(_basedir())
    cfg.perturb = 1.0
    net = Network(cfg, stage=3)
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    net = net.to(dev)
    opt = FusedAdam(net, lr=5e-4, lr_ranges=human_lr_ranges(net))

    # ---- FULL-SIZE parity before anything is timed (2048 rays x 128 samples = 262 144 points: the only size at which the
    # many-row routes -- thin GEMMs with register-resident weights, fused layer backward, wide WGRAD, LDS volume scatter --
    # are all taken inside the model): same weights, same rays, same jitter; outputs and parameter gradients of one step
    # against the oracle's op graph on the same device.
    topt.zero_grad()
    ref = oh.human_forward(sd, gb_ref, transitions_times=[0.4], t_rand=t_rand, stage=3)
    loss_of(ref).backward()
    opt.zero_grad()
    got = net(**gb, t_rand=t_rand)
    loss_of(got).backward()
    # Where the skinning mask (sum of LBS weights) vanishes, x_skel = sum(w q) / max(sum w, 1e-4) amplifies fp32 noise and
    # the canonical MLP's Fourier features (frequencies up to 512) amplify it again: the reference's own CPU and GPU runs
    # differ by 2e-3 there.  Like tests/test_gpu_human.py the radiance is therefore compared weighted by the mask -- the
    # quantity the composite consumes (alpha = mask * (1 - exp(-sigma delta))).
    m = ref["pts_mask"].detach()
    errs = {"pts_mask": float((got["pts_mask"] - m).abs().max()),
            "human_rgb*mask": float(((got["human_rgb"] - ref["human_rgb"]) * m[..., None]).abs().max()),
            "human_density*mask": float(((got["human_density"] - ref["human_density"]) * m).abs().max()) / max(1.0, float((ref["human_density"] * m).abs().max())),
            "deform_pts_prev_final": float((got["deform_pts_prev_final"] - ref["deform_pts_prev_final"]).abs().max())}
    assert errs["pts_mask"] < 1e-5 and errs["human_rgb*mask"] < 1e-4 and errs["human_density*mask"] < 2e-4, errs
    # (the forward warp divides by max(sum w, 1e-4) as well: same amplification, no weight to compare under)
    assert errs["deform_pts_prev_final"] < 5e-3, errs
    same_set = got["deform_pts_final"].shape == ref["deform_pts_final"].shape          # data-dependent cycle set (mask > 0.005)
    if same_set:
        errs["deform_pts_final"] = float((got["deform_pts_final"] - ref["deform_pts_final"]).abs().max())
        assert errs["deform_pts_final"] < 5e-3, errs
    hip_grads = {k: v.grad for k, v in net.named_parameters()}
    worst = 0.0
    for name in ("cnl_mlp.pts_linears.2.weight", "cnl_mlp.pts_linears.10.weight", "cnl_mlp.output_linear.0.weight",
                 "non_rigid_mlp.block_mlps.4.weight", "non_rigid_mlp.block_mlps.8.weight", "non_rigid_forward_mlp.block_mlps.0.weight",
                 "non_rigid_forward_mlp.block_mlps.12.bias", "mweight_vol_decoder.decoder.block_conv.0.weight",
                 "mweight_vol_decoder.const_embedding", "human_stateembeds.1", "pose_decoder.block_mlps_dstR.2.weight"):
        g_ref, g_hip = sd[name].grad, hip_grads[name]
        assert g_ref is not None and g_hip is not None, name
        rel = float((g_hip.reshape(g_ref.shape) - g_ref).abs().max()) / max(1e-12, float(g_ref.abs().max()))
        worst = max(worst, rel)
        errs["grad " + name] = rel
        # fixed bounds only where the graph is well conditioned; the decoder / pose-decoder gradients carry the fp32 noise
        # of the skinning normalisation (tests/test_gpu_conditioning.py measures them against fp64) and are recorded
        bound = 1e-3 if name.startswith("cnl_mlp") or name.startswith("human_stateembeds") else (2e-2 if "non_rigid" in name else 0.5)
        assert rel < bound, (name, rel, errs)
    _record("stage2_fullsize_parity", {"rays": B, "worst_relative_gradient_error": worst, **errs})
    del ref, got, hip_grads
    opt.zero_grad()

    t_torch = _time(torch_step, 1, 3)

    def hip_step():
        opt.zero_grad()
        out = net(**gb, t_rand=t_rand)
        loss_of(out).backward()
        opt.step(5e-4)

    del sd, params, topt
    torch.cuda.empty_cache()
    t_hip = _time(hip_step, 4, 16)
    _record("stage2_human", {"rays": B, "torch_rocm_rays_per_s": B / t_torch, "hip_eager_rays_per_s": B / t_hip, "speedup": t_torch / t_hip})
    assert t_torch / t_hip > 2.0, (t_torch, t_hip)


def test_stage3_fullsize_parity_and_speedup():
    """Stage 3 (both branches + z-merged composite) at the bench size of 2048 rays: RGB against the oracle's op graph on the
    same device (same weights, rays and jitters), then one training step of each timed."""
    import oracle.background as ob_
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    from hosnerf_amd.train import FusedAdam, batch_to_device, human_lr_ranges, stage3_losses
    dev = torch.device("cuda")
    B = 2048
    b = synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5)
    b["ray_grid"] = torch.cat([torch.rand(B, 2) * 100, torch.randn(B, 2), torch.ones(B, 1)], -1)
    b["newsmpl_to_camera_prev"] = torch.eye(4)
    b["newsmpl_to_camera_prev"][2, 3] = 3.0
    b["intrinsics_prev"] = torch.tensor([[500.0, 0, 50], [0, 500.0, 50], [0, 0, 1]])
    gb = batch_to_device(b, dev)
    gb_ref = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    g = torch.Generator().manual_seed(5)
    t_rand = torch.rand(B, 128, generator=g).to(dev)
    jit = [torch.rand(B, generator=g) for _ in range(3)]
    bsd = {k: v.to(dev).requires_grad_(True) for k, v in synth.background_state_dict(777, 2).items()}
    hsd = {k: v.to(dev).requires_grad_(True) for k, v in synth.human_state_dict(777, 2).items()}
    bb = {"rays_o": gb_ref["rays_o_bkg"], "rays_d": gb_ref["rays_d_bkg"], "viewdirs": gb_ref["viewdirs_bkg"], "radii": gb_ref["radii"], "times": b["time"]}

    def oracle_render(jitters):
        _, hist = ob_.mipnerf360_forward(bsd, bb, 1.0, True, 0.1, 1e6, transitions_times=[0.4], jitters=jitters, render=False)
        human = oh.human_forward(hsd, gb_ref, transitions_times=[0.4], t_rand=t_rand, stage=3)
        rgb, fg, order, hw, _ = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human, bb["rays_o"], bb["rays_d"],
                                                    gb_ref["newsmpl_to_scale_world"])
        return rgb, fg, hw, human

    cfg = default_cfg(_basedir())
    cfg.perturb = 1.0
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    hos = hos.to(dev)
    with torch.no_grad():
        rgb_o, fg_o, _, _ = oracle_render([j.view(B, 1) for j in jit])
        out = hos.render(gb, randomized=True, is_train=True, jitters=[j.to(dev) for j in jit], t_rand=t_rand)
    fg_h = out["idx_fg"].bool()
    # a ray whose mask sum sits within fp32 noise of the 5e-3 threshold may flip sides; everything else must agree
    flips = int((fg_h != fg_o).sum())
    same = fg_h == fg_o
    err = float((out["rgb"] - rgb_o)[same].abs().max())
    _record("stage3_fullsize_parity", {"rays": B, "rgb_linf": err, "fg_rays": int(fg_o.sum()), "fg_flips": flips})
    assert flips <= 2 and err < 1e-4, (flips, err)

    params = list(bsd.values()) + list(hsd.values())
    topt = torch.optim.Adam(params, lr=5e-4)

    def torch_step():
        topt.zero_grad()
        rgb, fg, hw, human = oracle_render(None)
        hw_full = torch.zeros(B, hw.shape[1], device=dev).masked_scatter(fg[:, None].expand(B, hw.shape[1]), hw)
        o = dict(human, rgb=rgb, idx_fg=fg.to(torch.int32), human_weights_sorted=hw_full)
        loss, _ = stage3_losses(o, gb_ref)
        loss.backward()
        topt.step()

    t_torch = _time(torch_step, 1, 3)
    del bsd, hsd, params, topt
    torch.cuda.empty_cache()
    ob1 = FusedAdam(hos.model, lr=5e-4)
    oh1 = FusedAdam(hos.human, lr=5e-4, lr_ranges=human_lr_ranges(hos.human))

    def hip_step():
        ob1.zero_grad(); oh1.zero_grad()
        o = hos.render(gb, randomized=True, is_train=True)
        loss, _ = stage3_losses(o, gb)
        loss.backward()
        ob1.step(5e-4); oh1.step(5e-4)

    t_hip = _time(hip_step, 3, 10)
    _record("stage3", {"rays": B, "torch_rocm_rays_per_s": B / t_torch, "hip_eager_rays_per_s": B / t_hip, "speedup": t_torch / t_hip})
    assert t_torch / t_hip > 2.0, (t_torch, t_hip)
