"""Arithmetic outside random-init (VERDICT r1 item 5 / SURVEY 7.1): the default forward arithmetic carries activations as fp16
(hi, lo) pairs -- exact to 2^-22 for |x| <= 65 504, 2^-11 up to 131 008, saturating beyond.  These tests scale the weights
x0.03 / x3 / x30, drive one layer's activations to ~9e4, and compare the three-level forward against the oracle evaluated in
float64 at the north-star tolerance (1e-4 RGB L-inf); where the format cannot hold it (x30: densities of 1e13) the range guard
must notice and switch the module to exact fp32 MFMA by itself."""
import json
import os
import tempfile
import warnings

import pytest
import torch

import oracle.background as ob
from hosnerf_amd import ops, synth

pytestmark = pytest.mark.gpu


def _basedir():
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


def _variant(kind):
    sd = synth.background_state_dict(777, 2)
    if kind.startswith("scale"):
        s = float(kind[5:])
        for k in sd:
            if k.endswith(".weight") and k.startswith("mlps."):
                sd[k] = sd[k] * s
    elif kind == "bigact":
        # NeRF trunk: layer 1's outputs pushed to ~9e4 (between the exact limit 65 504 and the hard limit 131 008), layer 2's
        # weights scaled down so the rest of the network sees ordinary magnitudes again
        sd["mlps.2.pts_linear.1.bias"] = sd["mlps.2.pts_linear.1.bias"] + 9.0e4
        sd["mlps.2.pts_linear.2.weight"] = sd["mlps.2.pts_linear.2.weight"] * 1e-4
    elif kind == "bigact_prop":
        # the same in the first proposal MLP, whose forward planes are fp16 in every mode (its densities steer the resampling)
        sd["mlps.0.pts_linear.1.bias"] = sd["mlps.0.pts_linear.1.bias"] + 9.0e4
        sd["mlps.0.pts_linear.2.weight"] = sd["mlps.0.pts_linear.2.weight"] * 1e-4
    elif kind == "tinyact":
        sd["mlps.2.pts_linear.1.weight"] = sd["mlps.2.pts_linear.1.weight"] * 1e-5        # activations ~1e-6 (fp16 subnormals)
        sd["mlps.2.pts_linear.2.weight"] = sd["mlps.2.pts_linear.2.weight"] * 1e5
    return sd


def _run(kind, mode):
    from hosnerf_amd.mipnerf360 import MipNeRF360
    dev = torch.device("cuda")
    sd = _variant(kind)
    batch = synth.stage1_batch(64, seed=5)
    with torch.no_grad():
        r64, h64 = ob.mipnerf360_forward({k: v.double() for k, v in sd.items()},
                                         {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, 0.5, False, 0.1, 1e6,
                                         transitions_times=[0.4])
    m = MipNeRF360(_basedir(), opaque_background=True)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev)
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        with torch.no_grad(), warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            rend, hist = m({k: v.to(dev) for k, v in batch.items()}, 0.5, False, False, 0.1, 1e6)
    finally:
        ops.set_gemm_mode(prev)
    err = float((rend[-1]["rgb"].double().cpu() - r64[-1]["rgb"]).abs().max())
    return err, m, [str(x.message) for x in w], float(h64[-1]["density"].max())


@pytest.mark.parametrize("mode", [ops.GEMM_PLANES, ops.GEMM_BF16X3])
@pytest.mark.parametrize("kind", ["scale1", "scale0.03", "tinyact"])
def test_scaled_weights_hold_the_tolerance_without_fallback(kind, mode):
    err, m, warned, _ = _run(kind, mode)
    assert err < 1e-4, (kind, err)
    assert m.gemm_mode is None and not warned, "the range guard must stay quiet inside the exact range"


@pytest.mark.parametrize("mode", [ops.GEMM_PLANES, ops.GEMM_BF16X3])
def test_weights_x3_hold_the_tolerance(mode):
    """x3 per layer: densities of 2e4 and hidden activations around the 6e4 flag level -- whichever side of the guard a layer lands
    on (fp16 hi/lo measured 4e-7 here, exact fp32 after a fallback), the rendering holds the tolerance."""
    err, m, _, dmax = _run("scale3", mode)
    assert err < 1e-4 and dmax > 1e4, (err, dmax)


@pytest.mark.parametrize("mode", [ops.GEMM_PLANES, ops.GEMM_BF16X3])
@pytest.mark.parametrize("kind", ["bigact", "bigact_prop", "scale30"])
def test_out_of_range_activations_fall_back_to_exact_fp32(kind, mode):
    err, m, warned, dmax = _run(kind, mode)
    if kind == "bigact" and mode == ops.GEMM_PLANES:
        # round 5: the NeRF MLP's planes are bf16 pairs (8-bit exponent): 9e4 is inside their range, nothing to guard, and the
        # rendering still holds the tolerance
        assert m.gemm_mode is None and not warned, (kind, warned)
    else:
        assert m.gemm_mode == ops.GEMM_FP32 and any("exact fp32" in s for s in warned), (kind, warned)
    assert err < 1e-4, (kind, err, dmax)
    # the module stays in exact-fp32 mode: the next call does not need the guard
    dev = torch.device("cuda")
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m({k: v.to(dev) for k, v in synth.stage1_batch(8, seed=6).items()}, 0.5, False, False, 0.1, 1e6)
    assert not w


def test_training_loop_poll_switches_the_module():
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import FusedAdam, check_range, stage1_loss
    dev = torch.device("cuda")
    m = MipNeRF360(_basedir(), opaque_background=True)
    m.load_state_dict(_variant("scale30"), strict=False)
    m = m.to(dev)
    opt = FusedAdam(m, lr=1e-4)
    b = {k: v.to(dev) for k, v in synth.stage1_batch(32, seed=7).items()}
    ops.arm_range_flag(dev)
    ops.range_events(dev)                       # clear
    assert not check_range([m], dev)
    opt.zero_grad()
    rend, hist = m(b, 0.5, True, True, 0.1, 1e6)
    stage1_loss(rend[-1]["rgb"], b["target"], hist)[0].backward()
    assert check_range([m], dev) and m.gemm_mode == ops.GEMM_FP32
    opt.zero_grad()
    rend, hist = m(b, 0.5, True, True, 0.1, 1e6)             # runs in exact fp32 now
    loss = stage1_loss(rend[-1]["rgb"], b["target"], hist)[0]
    loss.backward()
    opt.step(1e-4)
    assert bool(torch.isfinite(loss)) and not check_range([m], dev)
