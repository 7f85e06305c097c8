"""Training equivalence and parity on TRAINED weights (north-star "PSNR within 0.1 dB").

The reference's psnr (3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:101-112), its evaluation PSNR (:1456-1462) and its
training_step (:1501-1658; stage 1: 1st_State-Conditional_Scene/src/model/mipnerf360/model.py:491-569; stage 2:
2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:571-634) are what is compared: a synthetic, multi-view
consistent scene DIRECTORY (the on-disk formats of SURVEY 8(f).4) is trained twice from identical initial weights, items,
sampling draws, learning-rate schedule and gradient clip --
  (a) stage 1, 400 steps x 1024 background rays (single-image batches that alternate between the two states, train_frac annealing,
      warm-up + log-linear decay at 0.3 x the reference's rate),
  (b) stage 2, 500 steps x two 32x32 patches cut by the subject's box (<= 2048 rays x 128 samples, flow + cycle terms; 0.3 x the
      reference's rates, its 0.1 ** (step / 500 k) decay compressed into the run),
  (c) stage 3, 150 joint steps x 2048 rays warm-started from (a) and (b) as the reference's launcher does (S3/run.py:206-212), at 0.3 x
      the reference's rates (ONE Adam and ONE gradient norm over both modules),
once through the HIP path (`MipNeRF360` / `Network` + `FusedAdam`) and once through the reference's op graph as PyTorch-ROCm ops
(`oracle.steps.stage1_trainer` / `stage2_trainer` / `stage3_trainer`: torch autograd + `clip_grad_norm_` + torch Adam).  Asserted: the
PSNR on HELD-OUT frames agrees within 0.1 dB (stage 2: 0.15, its same-path spread alone is 0.08) and the smoothed final losses agree.  Two fp32 trainings are two chaotic trajectories (the
inverse-CDF resampling and Adam's normalisation amplify a last-bit difference), so weight-for-weight equality is not expected
and not asserted; the learning rates are those at which two trainings of the SAME path agree to a few hundredths of a dB (see the
constants below).  This test found a real difference in round 6: a flat Adam that updates parameters WITHOUT a gradient (the other
state's embeddings) where torch's skips them -- +0.3 dB against the reference graph at stage 1's full rate, gone with the lazily
updated spans of hos_adam_lazy_prepare (profiles/r06_convergence_pairs_before_lazy_adam.jsonl vs r06_convergence_pairs.jsonl).
And a second one: the stage-3 MSE was taken against the item's `target_rgbs` where the reference takes `target_patches` (they differ
where a patch leaves the subject's box) -- a systematic 0.06 dB, visible as ONE step at which the two NeRF-MLP trajectories parted.
Then, on the HIP-TRAINED weights (sharper densities -> more empty proposal bins; larger pre-activations against the fp16-hi
planes' +-65504): the full-size forward parity tables of tests/test_gpu_selfnoise.py again -- 1e-4 RGB L-inf on rays whose
discrete decisions agree, flip counts within the multiple of the reference's own fp32-vs-fp32 noise -- for stage 1 and for
stage 3 warm-started from the two trained modules (what 3rd_Complete_HOSNeRF/run.py:206-212 does), the stage-2 maps at 1e-4,
`train.range_skips == 0`, and the per-layer max |pre-activation| of every linear layer recorded next to 65504.
Figures -> gpurun_out/parity_counts.json (copied to profiles/r06_parity_counts.json)."""
import math
import os

import numpy as np
import pytest
import torch

import oracle.background as ob
import oracle.human as oh
import oracle.steps as osteps
from hosnerf_amd import synth
from tests import _parity as par
from tests._record import record

pytestmark = pytest.mark.gpu

S1_STEPS = int(os.environ.get("HOS_CONV_S1_STEPS", "400"))
S2_STEPS = int(os.environ.get("HOS_CONV_S2_STEPS", "500"))
S1_RAYS = 1024
S2_DECAY_STEPS = int(os.environ.get("HOS_CONV_S2_DECAY", "1"))     # > 0: the decay 0.1 ** (2 step / steps) instead of the reference's 500 k-step one
# The regime: a comparison of two trainings resolves what two fp32 trainings of the SAME path differ by.  Measured with the HIP path
# from initial weights perturbed by one ulp (scripts/convergence_spread.py, profiles/r06_convergence_spread_hip.jsonl): at the full
# stage-1 rate (2e-3) the held-out PSNR of four runs spreads over 0.16 dB after 400 steps and 0.75 dB after 1200; at 0.3 x the rate
# over 0.06 dB.  Stage 2: 0.27 dB at the full rate with the reference's (here: flat) decay, 0.03-0.08 dB at 0.3 x with the decay
# compressed into the run.  HIP / oracle pairs in the chosen regimes differ by 0.00-0.03 dB (stage 1) and 0.02-0.07 dB (stage 2)
# (profiles/r06_convergence_pairs.jsonl); at the full stage-1 rate by +-0.12 dB in either direction, i.e. by the spread.  Longer,
# slower or smaller-batch stage-2 regimes (900 steps x 1 patch, 700 steps at 0.2 x) do not narrow the stage-2 pairs further: 0.03-0.06 dB
# (profiles/r06_convergence_stage2_regimes.jsonl); over all fourteen stage-2 pairs measured the largest difference is 0.07 dB.
S3_STEPS = int(os.environ.get("HOS_CONV_S3_STEPS", "150"))
S2_PATCHES = int(os.environ.get("HOS_CONV_S2_PATCHES", "2"))
S1_LR_SCALE = float(os.environ.get("HOS_CONV_S1_LR", "0.3"))
S2_LR_SCALE = float(os.environ.get("HOS_CONV_S2_LR", "0.3"))
# Stage 3 (profiles/r06_convergence_stage3_pairs.jsonl): four pairs at 0.3 x the reference's rates differ by 0.001-0.009 dB, at the full
# rates by 0.001-0.05 dB.  BEFORE the loss target was fixed (r06_convergence_stage3_pairs_before_target_fix.jsonl) the same pairs sat
# 0.06-0.08 dB apart at 0.3 x -- with 0.01 dB of spread on either side: a systematic difference, traced to ONE step whose patch left the box.
S3_LR_SCALE = float(os.environ.get("HOS_CONV_S3_LR", "0.3"))
HW = 96
N_FRAMES = 16
HELD_OUT = (5, 11)
TRANSITIONS = (0.4,)
NET_DROP = ("frame_name", "ray_mask", "img_width", "img_height", "patch_div_indices")


def _psnr(pred, truth):
    mse = float(torch.mean((pred.double() - truth.double()) ** 2))
    return -10.0 * math.log10(mse)


# ------------------------------------------------------------------------------------------ the scene
def _make_scene(root, dev):
    """A scene directory + pixels that ARE learnable: the background is an environment map (colour = smooth function of the
    world-space viewing direction, i.e. consistent between the orbiting cameras), the subject a smooth colour disc at the image
    centre (every camera looks at the subject).  Returns (px, per-frame full-image background rays)."""
    from hosnerf_amd import formats
    from hosnerf_amd.dataset import SceneItems
    scene = os.path.join(root, "scene")
    px = synth.write_scene_dir(scene, N_FRAMES, HW, HW, seed=3)
    formats.load_scene(scene, (HW, HW), masks=px["alphas"], near=0.1, far=1e6)
    ds = SceneItems(scene, px["images"], px["alphas"], px["flows"], n_patches=2, patch_size=32, device=dev, seed=5)
    rays, images = [], []
    a = torch.tensor([[1.0, 0.4, -0.3], [-0.5, 1.0, 0.2], [0.3, -0.6, 1.0]], device=dev) * 4.0
    yy, xx = torch.meshgrid(torch.arange(HW, device=dev, dtype=torch.float32), torch.arange(HW, device=dev, dtype=torch.float32), indexing="ij")
    subject = torch.stack([0.55 + 0.3 * torch.sin(0.11 * xx), 0.45 + 0.3 * torch.cos(0.09 * yy), 0.5 + 0.25 * torch.sin(0.07 * (xx - yy))], -1)
    for i in range(N_FRAMES):
        fr = ds.eval_frame(i)
        full = {}
        for k in ("rays_o", "rays_d", "viewdirs", "radii"):
            src_in, src_out = fr[f"{k}_bkg" if k != "radii" else "radii"], fr[f"{k}_bkg_only" if k != "radii" else "radii_bkg_only"]
            t = torch.empty(HW * HW, src_in.shape[-1], device=dev)
            t[fr["ray_mask"]] = src_in
            t[fr["ray_mask_bkg"]] = src_out
            full[k] = t
        env = 0.5 + 0.42 * torch.sin(full["viewdirs"] @ a.T + torch.tensor([0.3, 1.1, 2.0], device=dev))
        alpha = torch.as_tensor(px["alphas"][i], device=dev).reshape(-1, 1)
        img = alpha * subject.reshape(-1, 3) + (1.0 - alpha) * env
        images.append(img.reshape(HW, HW, 3).cpu().numpy())
        full["target"] = img
        full["bg_pixels"] = torch.nonzero(alpha.reshape(-1) == 0).reshape(-1)
        full["time"] = float(ds.times[i])
        rays.append(full)
    px = dict(px, images=np.stack(images, 0).astype(np.float32))
    return scene, px, rays


def _stage1_lr(step, steps, scale=1.0):
    from hosnerf_amd.train import stage1_lr
    return scale * stage1_lr(step, steps, 2e-3, 2e-5, lr_delay_steps=64, lr_delay_mult=0.01)


def _stage1_batches(rays, steps, seed):
    """Single-image batches (S1 Backpack.gin `LitData.batch_sampler = "single_image"`): a training frame, 1024 of its
    background pixels, three per-ray jitters; all from one numpy / torch CPU stream so that both trainers see the same."""
    rs = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    train_frames = [i for i in range(N_FRAMES) if i not in HELD_OUT]
    for step in range(steps):
        f = rays[train_frames[rs.randint(len(train_frames))]]
        sel = f["bg_pixels"][torch.from_numpy(rs.choice(int(f["bg_pixels"].shape[0]), S1_RAYS, replace=False)).to(f["bg_pixels"].device)]
        b = {k: f[k][sel].contiguous() for k in ("rays_o", "rays_d", "viewdirs", "radii", "target")}
        b["times"] = f["time"]
        yield step, b, [torch.rand(S1_RAYS, generator=g) for _ in range(3)]


def _stage1_heldout(render, rays):
    """PSNR over the background pixels of the held-out frames, evaluation sampling (`randomized=False`)."""
    pred, truth = [], []
    for i in HELD_OUT:
        f = rays[i]
        sel = f["bg_pixels"]
        for c in range(0, int(sel.shape[0]), 4096):
            s = sel[c:c + 4096]
            b = {k: f[k][s].contiguous() for k in ("rays_o", "rays_d", "viewdirs", "radii")}
            b["times"] = f["time"]
            pred.append(render(b))
            truth.append(f["target"][s])
    return _psnr(torch.cat(pred), torch.cat(truth))


def _train_stage1(rays, dev, sd0=None, steps=None, lr_scale=None, oracle=True):
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import FusedAdam, stage1_loss
    sd0 = synth.background_state_dict(777, 2) if sd0 is None else sd0
    steps = S1_STEPS if steps is None else steps
    lr_scale = S1_LR_SCALE if lr_scale is None else lr_scale
    model = MipNeRF360(par.basedir(TRANSITIONS), opaque_background=True)
    model.load_state_dict(sd0, strict=False)
    model = model.to(dev)
    opt = FusedAdam(model, lr=2e-3, max_grad_norm=osteps.GRAD_MAX_NORM)
    p_ora, ora_step = osteps.stage1_trainer(sd0, dev, TRANSITIONS) if oracle else (None, None)
    loss_h, loss_o = [], []
    for step, b, jit in _stage1_batches(rays, steps, 17):
        lr, frac = _stage1_lr(step, steps, lr_scale), step / steps
        opt.zero_grad()
        rend, hist = model(b, frac, True, True, 0.1, 1e6, jitters=[j.to(dev) for j in jit])
        loss, _ = stage1_loss(rend[-1]["rgb"], b["target"], hist)
        loss.backward()
        opt.step(lr)
        loss_h.append(loss.detach())
        loss_o.append(ora_step(b, lr, frac, [j.view(-1, 1) for j in jit]) if oracle else loss.detach())          # (the oracle draws / takes its jitters on the host, H:364)
    loss_h, loss_o = torch.stack(loss_h).cpu(), torch.stack(loss_o).cpu()

    def render_hip(b):
        with torch.no_grad():
            return model(b, 1.0, False, False, 0.1, 1e6)[0][-1]["rgb"]

    def render_ora(b):
        with torch.no_grad():
            return ob.mipnerf360_forward(p_ora, b, 1.0, False, 0.1, 1e6, transitions_times=list(TRANSITIONS))[0][-1]["rgb"]

    res = {"steps": steps, "rays_per_step": S1_RAYS, "psnr_hip": _stage1_heldout(render_hip, rays),
           "psnr_oracle": _stage1_heldout(render_ora, rays) if oracle else None, "loss_first": [float(loss_h[0]), float(loss_o[0])],
           "loss_last20_mean": [float(loss_h[-20:].mean()), float(loss_o[-20:].mean())]}
    sd_hip = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if k in sd0}
    return res, sd_hip, model


def _stage2_items(scene, px, dev, steps):
    from hosnerf_amd.dataset import SceneItems
    ds = SceneItems(scene, px["images"], px["alphas"], px["flows"], n_patches=S2_PATCHES, patch_size=32, sample_subject_ratio=0.8, device=dev,
                    seed=9, stage=2)
    rs = np.random.RandomState(23)
    g = torch.Generator().manual_seed(23)
    train_frames = [i for i in range(N_FRAMES) if i not in HELD_OUT]
    items = []
    for _ in range(steps):
        it = ds[train_frames[rs.randint(len(train_frames))]]
        it["iter_val"] = torch.full((1,), 3e5)                       # every branch active (pose refinement, full non-rigid band)
        n = int(it["near"].shape[0])
        items.append((it, torch.rand(n, 128, generator=g).to(dev)))
    return ds, items


def _train_stage2(scene, px, dev, sd0=None, steps=None, lr_scale=None, oracle=True):
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import FusedAdam, human_lr_ranges, train_step_stage2
    LR = 6.667e-4 * (S2_LR_SCALE if lr_scale is None else lr_scale)
    sd0 = synth.human_state_dict(777, 2) if sd0 is None else sd0
    steps = S2_STEPS if steps is None else steps
    cfg = default_cfg(par.basedir(TRANSITIONS))
    cfg.perturb = 1.0
    net = Network(cfg, stage=2)
    net.load_state_dict(sd0, strict=True)
    net = net.to(dev)
    opt = FusedAdam(net, lr=LR, lr_ranges=human_lr_ranges(net, LR, LR / 10.0), max_grad_norm=osteps.GRAD_MAX_NORM)
    p_ora, ora_step = osteps.stage2_trainer(sd0, dev, LR, TRANSITIONS) if oracle else (None, None)
    ds, items = _stage2_items(scene, px, dev, steps)
    from hosnerf_amd.train import human_lr_decay
    loss_h, loss_o, rays = [], [], 0
    for step, (it, t_rand) in enumerate(items):
        decay = human_lr_decay(step) if S2_DECAY_STEPS <= 0 else 0.1 ** (2.0 * step / steps)
        batch = {k: v for k, v in it.items() if k not in NET_DROP}
        loss, _ = train_step_stage2(net, opt, batch, LR * decay, t_rand=t_rand)
        loss_h.append(loss)
        loss_o.append(ora_step(it, t_rand, decay) if oracle else loss)
        rays += int(it["near"].shape[0])
    loss_h, loss_o = torch.stack(loss_h).cpu(), torch.stack(loss_o).cpu()
    frames = [ds.eval_frame_stage2(i) for i in HELD_OUT]

    def heldout(render):
        return _psnr(torch.cat([render(f) for f in frames]), torch.cat([f["target_rgbs"] for f in frames]))

    def render_hip(f):
        net.eval()
        cfg.perturb = 0.0
        try:
            with torch.no_grad():
                return net(with_cycle=False, **{k: v for k, v in f.items() if k not in NET_DROP})["rgb"]
        finally:
            cfg.perturb = 1.0
            net.train()

    def render_ora(f):
        with torch.no_grad():
            return oh.human_forward(p_ora, f, transitions_times=list(TRANSITIONS), stage=2)["rgb"]

    res = {"steps": steps, "mean_rays_per_step": rays / max(1, steps), "psnr_hip": heldout(render_hip), "psnr_oracle": heldout(render_ora) if oracle else None,
           "loss_first": [float(loss_h[0]), float(loss_o[0])], "loss_last20_mean": [float(loss_h[-20:].mean()), float(loss_o[-20:].mean())]}
    sd_hip = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if k in sd0}
    return res, sd_hip, frames


def _stage3_items(scene, px, dev, frames, n, seed, n_patches=2):
    from hosnerf_amd.dataset import SceneItems
    ds = SceneItems(scene, px["images"], px["alphas"], px["flows"], n_patches=n_patches, patch_size=32, device=dev, seed=seed)
    rs = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    items = []
    for k in range(n):
        it = ds[frames[rs.randint(len(frames))] if n > len(frames) else frames[k]]
        it["iter_val"] = torch.full((1,), 3e5)
        B = int(it["near"].shape[0])
        items.append((it, torch.rand(B, 128, generator=g).to(dev), [torch.rand(B, generator=g) for _ in range(3)]))
    return items


def _train_stage3(scene, px, dev, bsd, hsd, steps=None, lr_scale=None, oracle=True):
    """Stage 3 warm-started from the two trained modules (S3/run.py:206-212), `steps` joint steps (M:1501-1658: both renderers, merged
    composite, ONE Adam / ONE clip over both modules) through the HIP path and through the reference's op graph; held-out PSNR =
    the rays of eight 32x32 patches per held-out frame, training sampling with the SAME injected draws for both."""
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    from hosnerf_amd.train import FusedAdam, GradClip, human_lr_decay, human_lr_ranges, train_step_stage3
    LR = 6.667e-5 * (S3_LR_SCALE if lr_scale is None else lr_scale)
    steps = S3_STEPS if steps is None else steps
    cfg = default_cfg(par.basedir(TRANSITIONS))
    cfg.perturb = 1.0
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(bsd, strict=False)
    hos.human.load_state_dict(hsd, strict=True)
    hos = hos.to(dev)
    clip = GradClip(osteps.GRAD_MAX_NORM)
    o_b = FusedAdam(hos.model, lr=LR, clip=clip)
    o_h = FusedAdam(hos.human, lr=LR, lr_ranges=human_lr_ranges(hos.human, LR, LR / 10.0), clip=clip)
    pb, ph, ora_step = osteps.stage3_trainer(bsd, hsd, dev, LR, TRANSITIONS) if oracle else (None, None, None)
    train_frames = [i for i in range(N_FRAMES) if i not in HELD_OUT]
    loss_h, loss_o = [], []
    for step, (it, t_rand, jit) in enumerate(_stage3_items(scene, px, dev, train_frames, steps, 41)):
        decay = human_lr_decay(step)
        batch = {k: v for k, v in it.items() if k not in NET_DROP}
        loss, _ = train_step_stage3(hos, o_b, o_h, batch, LR * decay, jitters=[j.to(dev) for j in jit], t_rand=t_rand)
        loss_h.append(loss)
        loss_o.append(ora_step(it, t_rand, [j.view(-1, 1) for j in jit], decay) if oracle else loss)
    loss_h, loss_o = torch.stack(loss_h).cpu(), torch.stack(loss_o).cpu()
    held = _stage3_items(scene, px, dev, list(HELD_OUT), len(HELD_OUT), 43, n_patches=8)

    def heldout(render):
        with torch.no_grad():
            return _psnr(torch.cat([render(it, t, j) for it, t, j in held]), torch.cat([it["target_rgbs"] for it, _, _ in held]))

    def render_hip(it, t_rand, jit):
        return hos.render({k: v for k, v in it.items() if k not in NET_DROP}, randomized=True, is_train=True, jitters=[j.to(dev) for j in jit],
                          t_rand=t_rand, with_cycle=False)["rgb"]

    def render_ora(it, t_rand, jit):
        return osteps.stage3_render(pb, ph, it, TRANSITIONS, t_rand=t_rand, jitters=[j.view(-1, 1) for j in jit])["rgb"]

    res = {"steps": steps, "rays_per_step": 2048, "psnr_hip": heldout(render_hip), "psnr_oracle": heldout(render_ora) if oracle else None,
           "loss_first": [float(loss_h[0]), float(loss_o[0])], "loss_last20_mean": [float(loss_h[-20:].mean()), float(loss_o[-20:].mean())]}
    # the jointly trained weights of the HIP path (for the whole-frame evaluation check); not JSON: kept out of the record
    _train_stage3.trained = ({k: v.detach().cpu().clone() for k, v in hos.model.state_dict().items() if k in bsd},
                             {k: v.detach().cpu().clone() for k, v in hos.human.state_dict().items() if k in hsd})
    return res


class _LinearPeaks:
    """Records max |output| of every `F.linear` / matmul-form linear the oracle runs (call order = layer order)."""

    def __init__(self):
        self.peaks = []

    def __enter__(self):
        self._lin = torch.nn.functional.linear
        peaks = self.peaks

        def linear(x, w, b=None):
            y = self._lin(x, w, b)
            peaks.append((tuple(w.shape), float(y.abs().max())))
            return y
        torch.nn.functional.linear = linear
        return self

    def __exit__(self, *exc):
        torch.nn.functional.linear = self._lin


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    assert torch.cuda.is_available()
    dev = torch.device("cuda")
    from hosnerf_amd.train import range_skips
    root = str(tmp_path_factory.mktemp("conv"))
    scene, px, rays = _make_scene(root, dev)
    skips0 = range_skips(dev)
    s1, bsd, model = _train_stage1(rays, dev)
    del model
    torch.cuda.empty_cache()
    s2, hsd, frames2 = _train_stage2(scene, px, dev)
    torch.cuda.empty_cache()
    s3 = _train_stage3(scene, px, dev, bsd, hsd)
    bsd3, hsd3 = _train_stage3.trained
    torch.cuda.empty_cache()
    skipped = int(range_skips(dev)) - int(skips0)
    record("convergence.stage1", s1)
    record("convergence.stage2", s2)
    record("convergence.stage3", s3)
    record("convergence.range_skips_during_training", skipped)
    return {"dev": dev, "scene": scene, "px": px, "rays": rays, "s1": s1, "s2": s2, "s3": s3, "bsd": bsd, "hsd": hsd, "bsd3": bsd3, "hsd3": hsd3, "skipped": skipped, "frames2": frames2}


def test_stage1_heldout_psnr_matches_the_reference_graph(trained):
    s = trained["s1"]
    assert s["loss_last20_mean"][0] < 0.6 * s["loss_first"][0], ("the HIP path did not train", s)
    assert abs(s["psnr_hip"] - s["psnr_oracle"]) <= 0.1, s                                     # north-star: PSNR within 0.1 dB
    assert abs(s["loss_last20_mean"][0] - s["loss_last20_mean"][1]) <= 0.03 * abs(s["loss_last20_mean"][1]), s
    assert abs(s["loss_first"][0] - s["loss_first"][1]) <= 1e-4 * abs(s["loss_first"][1]), s   # identical first step


def test_stage2_heldout_psnr_matches_the_reference_graph(trained):
    s = trained["s2"]
    assert s["loss_last20_mean"][0] < 0.8 * s["loss_first"][0], ("the HIP path did not train", s)
    # Stage 2 is the one stage whose SAME-path spread is of the size of the north-star's 0.1 dB: four HIP runs from initial weights one
    # ulp apart land 0.08 dB apart in this regime (r06_convergence_spread_hip.jsonl), and no slower / longer regime narrows it
    # (r06_convergence_stage2_regimes.jsonl).  Twenty (HIP, oracle) pairs measured: |difference| 0.02-0.07 dB, both signs, mean +0.02.
    # The bound is the 0.1 dB plus half of that spread -- a single comparison cannot resolve less -- and the measured value is recorded.
    assert abs(s["psnr_hip"] - s["psnr_oracle"]) <= 0.15, s
    assert abs(s["loss_last20_mean"][0] - s["loss_last20_mean"][1]) <= 0.03 * abs(s["loss_last20_mean"][1]), s
    assert abs(s["loss_first"][0] - s["loss_first"][1]) <= 1e-4 * abs(s["loss_first"][1]), s


def test_stage3_heldout_psnr_matches_the_reference_graph(trained):
    """BASELINE's metric config: the joint stage-3 step, warm-started from the trained stage-1 / stage-2 modules."""
    s = trained["s3"]
    assert abs(s["psnr_hip"] - s["psnr_oracle"]) <= 0.1, s
    assert abs(s["loss_last20_mean"][0] - s["loss_last20_mean"][1]) <= 0.03 * abs(s["loss_last20_mean"][1]), s
    assert abs(s["loss_first"][0] - s["loss_first"][1]) <= 2e-4 * abs(s["loss_first"][1]), s


def test_no_step_was_skipped_by_the_range_guard(trained):
    """`train.range_skips`: optimiser steps the device-side fp16 range guard skipped during both trainings."""
    assert trained["skipped"] == 0, trained["skipped"]


def test_trained_weights_stage1_fullsize_parity(trained):
    dev, rays = trained["dev"], trained["rays"]
    f = rays[HELD_OUT[0]]
    rs = np.random.RandomState(4)
    sel = f["bg_pixels"][torch.from_numpy(rs.choice(int(f["bg_pixels"].shape[0]), 1024, replace=False)).to(dev)]
    batch = {k: f[k][sel].cpu() for k in ("rays_o", "rays_d", "viewdirs", "radii", "target")}
    batch["times"] = torch.full((1024,), f["time"])
    g = torch.Generator().manual_seed(11)
    jit = [torch.rand(1024, generator=g) for _ in range(3)]
    with _LinearPeaks() as lp:
        pairs = par.stage1_tables(trained["bsd"], batch, jit, dev, train_frac=1.0, transitions=TRANSITIONS)
    n_layers = len(lp.peaks) // 3                               # three oracle evaluations
    peaks = [{"weight": list(s), "max_abs_preactivation": v, "of_fp16_max": v / 65504.0} for s, v in lp.peaks[n_layers:2 * n_layers]]
    record("trained.stage1[1024 held-out rays, trained weights]", pairs)
    record("trained.stage1.preactivation_peaks[oracle fp32 rocm]", peaks)
    par.assert_stage1(pairs)
    assert max(p["max_abs_preactivation"] for p in peaks) < 65504.0, peaks


def test_trained_weights_stage2_maps(trained):
    """The stage-2 network's maps on a held-out full frame (evaluation sampling), trained weights: 1e-4 against the oracle on the
    host cores and on the device."""
    from hosnerf_amd.human_nerf import Network, default_cfg
    dev = trained["dev"]
    cfg = default_cfg(par.basedir(TRANSITIONS))
    cfg.perturb = 0.0
    net = Network(cfg, stage=2)
    net.load_state_dict(trained["hsd"], strict=True)
    net = net.to(dev).eval()
    f = trained["frames2"][0]
    fb = {k: v for k, v in f.items() if k not in NET_DROP}
    with torch.no_grad():
        got = net(with_cycle=False, **fb)
        assert net.gemm_mode is None, "the HIP forward left the exact fp16 hi/lo range on the trained weights"
        with _LinearPeaks() as lp:
            ref_d = oh.human_forward(par.cast(trained["hsd"], dev, torch.float32), f, transitions_times=list(TRANSITIONS), stage=2)
        fc = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in f.items()}
        ref_c = oh.human_forward(trained["hsd"], fc, transitions_times=list(TRANSITIONS), stage=2)
    errs = {}
    for name, ref in (("rocm", ref_d), ("cpu", ref_c)):
        for k in ("rgb", "alpha", "weights"):
            errs[f"{k} vs oracle_fp32_{name}"] = float((got[k].cpu() - ref[k].cpu()).abs().max())
    errs["oracle cpu vs rocm rgb"] = float((ref_d["rgb"].cpu() - ref_c["rgb"]).abs().max())
    errs["rays"] = int(f["near"].shape[0])
    errs["mean_alpha"] = float(ref_c["alpha"].mean())
    record("trained.stage2[held-out full frame, trained weights]", errs)
    record("trained.stage2.preactivation_peaks[oracle fp32 rocm]", [{"weight": list(s), "max_abs_preactivation": v} for s, v in lp.peaks])
    assert all(v < 1e-4 for k, v in errs.items() if " vs oracle_fp32_" in k), errs


def test_trained_weights_stage3_fullsize_parity(trained):
    """Stage 3 warm-started from the trained background model and the trained human network (3rd_Complete_HOSNeRF/run.py:206-212),
    2048 rays of a held-out frame's patches, training sampling with injected draws."""
    from hosnerf_amd.dataset import SceneItems
    dev, px = trained["dev"], trained["px"]
    ds = SceneItems(trained["scene"], px["images"], px["alphas"], px["flows"], n_patches=2, patch_size=32, device=dev, seed=31)
    it = ds[HELD_OUT[1]]
    it["iter_val"] = torch.full((1,), 3e5)
    b = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in it.items()}
    B = int(b["near"].shape[0])
    g = torch.Generator().manual_seed(5)
    t_rand = torch.rand(B, 128, generator=g)
    jit = [torch.rand(B, generator=g) for _ in range(3)]
    pairs = par.stage3_tables(trained["bsd"], trained["hsd"], b, t_rand, jit, dev, transitions=TRANSITIONS)
    record("trained.stage3[2048 rays of a held-out frame, trained weights]", pairs)
    par.assert_stage3(pairs, outliers=1)


def test_trained_weights_whole_frame_evaluation_psnr(trained):
    """The reference's evaluation PSNR (M:101-112 psnr, M:1293-1494 / :1456-1462: rays through the subject's box take both branches and
    the merged composite, the others the background model alone; evaluation sampling, no jitter) of a HELD-OUT frame on the jointly
    trained stage-3 weights: `eval.render_frame` (8192-ray chunks, cached prologue) against the oracle's chunk-free restatement of the
    same frame on the same weights -- the two PSNRs agree to 0.005 dB, the frames to 1e-4 but for counted rays."""
    from hosnerf_amd import eval as ev
    from hosnerf_amd.dataset import SceneItems
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    dev, px = trained["dev"], trained["px"]
    bsd, hsd = trained["bsd3"], trained["hsd3"]
    cfg = default_cfg(par.basedir(TRANSITIONS))
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(bsd, strict=False)
    hos.human.load_state_dict(hsd, strict=True)
    hos = hos.to(dev)
    ds = SceneItems(trained["scene"], px["images"], px["alphas"], px["flows"], n_patches=2, patch_size=32, device=dev, seed=3)
    fr = ds.eval_frame(HELD_OUT[0])
    H = W = HW
    rendered = ev.render_frame(hos, fr, chunk_bkg=8192, randomized=False)
    assert hos.model.gemm_mode is None and hos.human.gemm_mode is None
    truth = ev.truth_frame(fr)
    pb, ph = par.cast(bsd, dev, torch.float32), par.cast(hsd, dev, torch.float32)
    with torch.no_grad():
        bb = {"rays_o": fr["rays_o_bkg"], "rays_d": fr["rays_d_bkg"], "viewdirs": fr["viewdirs_bkg"], "radii": fr["radii"], "times": fr["time"]}
        _, hist = ob.mipnerf360_forward(pb, bb, 1.0, False, 0.1, 1e6, transitions_times=list(TRANSITIONS), render=False)
        b = {k: v for k, v in fr.items()}
        b["is_train"] = False
        human = oh.human_forward(ph, b, transitions_times=list(TRANSITIONS))
        rgb_fg, fg_o = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human, bb["rays_o"], bb["rays_d"],
                                           fr["newsmpl_to_scale_world"])[:2]
        bo = {"rays_o": fr["rays_o_bkg_only"], "rays_d": fr["rays_d_bkg_only"], "viewdirs": fr["viewdirs_bkg_only"], "radii": fr["radii_bkg_only"],
              "times": fr["time"]}
        parts = []
        for c in range(0, int(bo["radii"].shape[0]), 4096):
            bc = {k: (v[c:c + 4096] if isinstance(v, torch.Tensor) else v) for k, v in bo.items()}
            _, ho = ob.mipnerf360_forward(pb, bc, 1.0, False, 0.1, 1e6, transitions_times=list(TRANSITIONS), render=False)
            parts.append(oh.raw2outputs(ho[-1]["rgb"], ho[-1]["density"], ho[-1]["tdist"][..., :-1], bc["rays_d"], torch.ones_like(ho[-1]["density"]))[0])
    want = (torch.as_tensor(fr["bgcolor"], device=dev).float().reshape(3) / 255.0).expand(H * W, 3).clone()
    want[fr["ray_mask"]] = rgb_fg
    want[fr["ray_mask_bkg"]] = torch.cat(parts)
    d = (rendered - want).abs().max(-1).values
    p_hip, p_ora = ev.psnr_metric(rendered, truth), ev.psnr_metric(want, truth)
    rep = {"psnr_hip": p_hip, "psnr_oracle": p_ora, "rgb_linf": float(d.max()), "rays": H * W, "fg_rays": int(fr["ray_mask"].sum()),
           "rays_over_1e-4": int((d > 1e-4).sum())}
    record("trained.stage3.whole_frame_eval[held-out frame, jointly trained weights]", rep)
    assert abs(p_hip - p_ora) < 0.005, rep
    assert rep["rays_over_1e-4"] <= 0.005 * H * W and float(d.max()) < 5e-3, rep          # (swapped coinciding pairs / moved samples: counted)
