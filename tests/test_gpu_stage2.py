"""The reference's STAGE-2 step on the device (BASELINE configs[2]; VERDICT r1 'top_next'):
  * `Network(cfg, stage=2)` -- in-network `_raw2outputs` with background colour, output dict {rgb, alpha, depth, weights,
    observe_pts, deform_pts_final[, deform_pts_prev_final]} -- against fixtures made by the reference's own stage-2
    network (2nd_State_Conditional_Human-Object/core/nets/human_nerf/network.py:273-299, 538-556; tests/golden/human_forward.npz
    keys s2_*);
  * the stage-2 training step (model.py:571-605, 918-944: 0.2 MSE on unpacked patches + 0.01 flow with the network's
    `weights` + 0.01 cycle) -- loss values and parameter gradients against the oracle's autograd, with PARTIAL patch masks;
  * the Lightning-style module driven the way a Trainer drives it (zero_grad(set_to_none=True) included)."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

import oracle.human as oh
import oracle.losses as ol
from hosnerf_amd import synth

pytestmark = pytest.mark.gpu
# Gradients of the pose / volume decoders against fp64: the bound is this factor times the fp32 oracle's own error (+ the discrete
# floor).  Their gradients are near-total cancellations of terms the backward GEMMs carry as bf16 PAIRS (16-17 significant bits
# per operand, where the fp32 graph has 24): the representation of the operands, not a missing product, sets a 2^7 ceiling;
# measured factors are recorded (profiles/*parity_counts.json: stage2.decoder_gradients_vs_fp64).
DECODER_GRAD_FACTOR = float(os.environ.get("HOS_DECODER_GRAD_FACTOR", "32"))      # measured worst factor: 10.4 (round 4)
HERE = os.path.dirname(os.path.abspath(__file__))


def _basedir():
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


@pytest.fixture(scope="module")
def net2(dev):
    from hosnerf_amd.human_nerf import Network, default_cfg
    cfg = default_cfg(_basedir())
    cfg.perturb = 0.0
    n = Network(cfg, stage=2)
    n.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    return n.to(dev)


def maxerr(a, b):
    b = torch.from_numpy(np.asarray(b)) if not isinstance(b, torch.Tensor) else b
    return float((a.detach().cpu() - b.cpu()).abs().max())


@pytest.mark.parametrize("tag,time,is_train,it,perturb", [("evalA", 0.5, False, 3e5, 0.0), ("trainA", 0.5, True, 3e5, 1.0),
                                                          ("earlyB", 0.3, True, 1000.0, 0.0), ("t0C", 0.0, True, 3e5, 0.0)])
def test_network_stage2_vs_reference(dev, net2, tag, time, is_train, it, perturb):
    hf = np.load(os.path.join(HERE, "golden", "human_forward.npz"))
    p = f"s2_{tag}_"
    b = synth.human_batch(8, seed=21, time=time, is_train=is_train, iter_val=it)
    gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    t_rand = torch.from_numpy(hf[f"s3_{tag}_t_rand"]).to(dev) if perturb > 0 else None
    net2.cfg.perturb = float(perturb)
    try:
        with torch.no_grad():
            out = net2(t_rand=t_rand, **gb)
    finally:
        net2.cfg.perturb = 0.0
    assert set(out.keys()) == set(hf[p + "keys"].tolist()), (sorted(out.keys()), hf[p + "keys"].tolist())
    assert maxerr(out["rgb"], hf[p + "rgb"]) < 1e-4                       # north-star tolerance: RGB L-inf
    assert maxerr(out["alpha"], hf[p + "alpha"]) < 1e-4
    assert maxerr(out["weights"], hf[p + "weights"]) < 1e-4
    assert maxerr(out["depth"], hf[p + "depth"]) < 5e-4                  # sum w z, z ~ 3
    assert out["observe_pts"].shape == hf[p + "observe_pts"].shape
    assert maxerr(out["observe_pts"], hf[p + "observe_pts"]) < 2e-6
    assert maxerr(out["deform_pts_final"], hf[p + "deform_pts_final"]) < 1e-4
    if (p + "deform_pts_prev_final") in hf:
        assert maxerr(out["deform_pts_prev_final"], hf[p + "deform_pts_prev_final"]) < 5e-4
    else:
        assert "deform_pts_prev_final" not in out


def _stage2_item(B, seed, n_patches=2, size=32):
    b = synth.add_patch_supervision(synth.human_batch(B, seed=seed, time=0.5, is_train=True, iter_val=3e5), n_patches, size, seed)
    return b


@pytest.fixture(params=[False, True], ids=["layer_bwd", "group_bwd"])
def group_bwd(request):
    """True: chain forward + group backward of the non-rigid MLPs below their row thresholds (the full-size training path)."""
    from hosnerf_amd import ops
    prev = ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_BWD_MIN_ROWS
    if request.param:
        ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_BWD_MIN_ROWS = 1, 1
    yield request.param
    ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_BWD_MIN_ROWS = prev


def test_stage2_step_vs_oracle(dev, net2, group_bwd):
    """One stage-2 training step at a size the oracle finishes in seconds (96 rays in two 8x8 patches, partial masks):
    loss terms against the fp32 oracle, and EVERY parameter gradient against the oracle evaluated in float64 -- the bound is
    the fp32 oracle's own distance from float64 (parts of this graph are ill-conditioned in fp32 by construction, see
    tests/test_gpu_conditioning.py), not a fixed constant."""
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets, stage2_losses
    B = 96
    b = _stage2_item(B, seed=41, n_patches=2, size=8)
    assert int(b["patch_masks"].sum()) == B and not bool(b["patch_masks"].all())
    grads, outs = {}, {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in synth.human_state_dict(777, 2).items()}
        bb = {k: (v.to(dt) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in b.items()}
        out_o = oh.human_forward(sd, bb, transitions_times=[0.4], stage=2)
        tot_o, parts_o = ol.stage2_losses(out_o, bb, 0.5)
        tot_o.backward()
        grads[tag] = {n: (p_.grad.double() if p_.grad is not None else None) for n, p_ in sd.items()}
        outs[tag] = (out_o["rgb"].detach().double(), float(tot_o), {k: float(v) for k, v in parts_o.items()})

    gb = batch_to_device(prepare_patch_targets(b), dev)
    net2.zero_grad()
    out = net2(static_cycle=True, **gb)
    total, parts = stage2_losses(out, gb)
    total.backward()
    rgb64, tot64, parts64 = outs["f64"]
    assert maxerr(out["rgb"].double(), rgb64) < 1e-4
    assert abs(float(total) - tot64) < 1e-5 * max(1.0, abs(tot64))
    for k in ("mse", "flow", "cycle"):
        e_ref = abs(outs["f32"][2][k] - parts64[k])
        assert abs(float(parts[k]) - parts64[k]) < 3.0 * e_ref + 1e-5 * max(1e-3, abs(parts64[k])), (k, float(parts[k]), parts64[k])
    net2.scatter_compact_grads()           # the live taps of the first deconvolution layer -> its reference-shaped p.grad
    params = dict(net2.named_parameters())
    seen, report, ratios = 0, {}, []
    for n, t in grads["f64"].items():
        if t is None or float(t.abs().max()) == 0:
            continue
        e_ref = float((grads["f32"][n] - t).norm() / (t.norm() + 1e-30))
        e_hip = float((params[n].grad.detach().double().cpu().reshape(t.shape) - t).norm() / (t.norm() + 1e-30))
        report[n] = (e_ref, e_hip)
        seen += 1
        # the additive term covers discrete events that any two fp32 evaluations of this graph disagree on (a ReLU or |.| sign
        # flipping on a pre-activation within rounding of 0, a sample crossing the 0.005 cycle threshold): each moves a
        # gradient by ~1/(number of sample points)
        if n.startswith(("pose_decoder", "mweight_vol_decoder")):
            # these gradients are sums over all sample points of terms amplified by 1 / max(sum of skinning weights, 1e-4)
            # (up to 1e4) that cancel almost completely; the backward GEMMs multiply bf16 hi/lo pairs (products exact to
            # 2^-17, fp32 accumulation) where the fp32 graph has 2^-24, and the cancellation exposes exactly that factor:
            # measured ~1e-2 here against ~1e-3 for the fp32 graph (direction: cosine > 0.9999)
            a, t_ = params[n].grad.detach().double().cpu().reshape(-1), t.reshape(-1)
            cos = float((a @ t_) / (a.norm() * t_.norm() + 1e-30))
            ratios.append(((e_hip - 3e-3) / max(e_ref, 1e-12), n, e_ref, e_hip))
            assert cos > 0.999 and e_hip < DECODER_GRAD_FACTOR * e_ref + 3e-3, (n, e_ref, e_hip, cos)
        else:
            assert e_hip < 3.0 * e_ref + 3e-3, (n, e_ref, e_hip)
    assert seen >= 70
    from tests._record import record
    w = max(ratios)
    record("stage2.decoder_gradients_vs_fp64" + ("[group_bwd]" if group_bwd else ""), {"bound_factor": DECODER_GRAD_FACTOR, "worst_factor": w[0], "worst_param": w[1],
                                                "fp32_oracle_rel_err": w[2], "hip_rel_err": w[3]})
    net2.zero_grad()


def test_lit_stage2_module_under_a_trainer_style_loop(dev):
    """`select_model('state_humanobject')`: configure_optimizers / training_step / optimizer_step the way Lightning calls
    them -- including `optimizer.zero_grad()` with torch's default set_to_none=True, which must NOT detach the HIP weight
    gradients from the flat buffer -- trains: the loss falls and every module's parameters move."""
    from hosnerf_amd.select_option import select_model
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets
    lit = select_model("state_humanobject", _basedir())
    lit.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    lit = lit.to(dev)
    lit._step = 300000                       # every branch active (pose refinement, full hann band)
    opt = lit.configure_optimizers()
    gb = batch_to_device(prepare_patch_targets(_stage2_item(256, seed=43, n_patches=2, size=16)), dev)
    before = {k: v.detach().clone() for k, v in lit.human.state_dict().items()}
    losses = []
    for i in range(6):
        opt.zero_grad()                      # set_to_none=True default signature
        torch.nn.Module.zero_grad(lit)       # and the nn.Module flavour a training loop may call
        lit.zero_grad()
        loss = lit.training_step(gb, i)
        loss.backward()
        lit.optimizer_step(0, i, opt, optimizer_closure=None)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    after = lit.human.state_dict()
    for mod in ("cnl_mlp", "non_rigid_mlp", "non_rigid_forward_mlp", "mweight_vol_decoder", "pose_decoder", "human_stateembeds"):
        moved = max(float((after[k] - before[k]).abs().max()) for k in before if k.startswith(mod))
        assert moved > 0, f"{mod} did not train"
    # M2:606-634: the rate written after step i is base * 0.1 ** (i / 500k), i = trainer.global_step read BEFORE optimizer.step
    assert lit._global_step() == 300006
    assert abs(opt.param_groups[0]["lr"] - 6.667e-4 * 0.1 ** ((lit._global_step() - 1) / 5e5)) < 1e-9
