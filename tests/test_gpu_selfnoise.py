"""Are the relaxed full-size bounds reference self-noise?  (VERDICT r2 "next round" item 2.)

The full-size parity tests (tests/test_gpu_speedup.py) assert the north-star tolerance -- 1e-4 RGB L-inf -- on the rays whose
DISCRETE decisions agree with the oracle's, and count / bound the others: a sample that jumps across an empty bin of a proposal
histogram when the CDF moves by an ulp (stage 1, inverse-CDF resampling, H:343-399), a background sample and a human sample
that coincide and swap places in the z-merge (stage 3, M:1547-1585).  Here the REFERENCE's arithmetic is evaluated three ways
on the same rays, weights and draws -- the oracle in fp32 on the host cores (MKL sums), in fp32 on the MI355X (rocBLAS sums)
and in float64 -- and the same statistics are taken between those: if two fp32 evaluations of the reference's own op graph
disagree on as many rays, by as much, as the HIP path does, the relaxed bounds measure the reference, not this implementation.
The HIP path's counts are asserted against a small multiple of the oracle-vs-oracle counts (the `e_ref` pattern of the
gradient tests); the figures go to gpurun_out/parity_counts.json -> profiles/r03_parity_counts.json."""
import json
import os
import tempfile

import pytest
import torch

import oracle.background as ob
import oracle.human as oh
from hosnerf_amd import synth
from tests._record import record

pytestmark = pytest.mark.gpu


def _basedir():
    d = tempfile.mkdtemp(prefix="hos_noise_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


def _cast(d, device, dtype):
    return {k: (v.to(device=device, dtype=dtype if v.is_floating_point() else None) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


def _stage1_pair(a, b):
    """a, b: (rgb [B,3], [tdist per level]) as float64 CPU tensors.  Rays whose 160 interval edges agree to 1e-4 relative /
    the others; worst RGB difference on each class; rays over the north-star tolerance."""
    moved = torch.zeros(a[0].shape[0], dtype=torch.bool)
    for ta, tb in zip(a[1], b[1]):
        moved |= ((ta - tb).abs() / tb.abs()).max(-1).values > 1e-4
    diff = (a[0] - b[0]).abs().max(-1).values
    return {"rays_with_moved_samples": int(moved.sum()), "rays_over_1e-4": int((diff > 1e-4).sum()),
            "rgb_linf_same_samples": float(diff[~moved].max()), "rgb_linf_moved_samples": float(diff[moved].max()) if bool(moved.any()) else 0.0}


def test_stage1_fullsize_bounds_are_reference_self_noise():
    from hosnerf_amd.mipnerf360 import MipNeRF360
    dev = torch.device("cuda")
    B = 1024
    batch = synth.stage1_batch(B, seed=777)
    sd = synth.background_state_dict(777, 2)
    g = torch.Generator().manual_seed(11)
    jit = [torch.rand(B, generator=g) for _ in range(3)]               # the same fp32 draws for every evaluation

    def oracle(device, dtype):
        with torch.no_grad():
            rend, hist = ob.mipnerf360_forward(_cast(sd, device, dtype), _cast(batch, device, dtype), 0.5, True, 0.1, 1e6,
                                               transitions_times=[0.4], jitters=[j.view(B, 1) for j in jit])
        return rend[-1]["rgb"].double().cpu(), [h["tdist"].double().cpu() for h in hist]

    ev = {"oracle_fp32_cpu": oracle("cpu", torch.float32), "oracle_fp32_rocm": oracle(dev, torch.float32), "oracle_fp64": oracle(dev, torch.float64)}
    model = MipNeRF360(_basedir(), opaque_background=True)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev)
    hb = {k: v.to(dev) for k, v in batch.items()}
    hb["times"] = 0.5
    with torch.no_grad():
        rend, hist = model(hb, 0.5, True, True, 0.1, 1e6, jitters=[j.to(dev) for j in jit])
    ev["hip"] = (rend[-1]["rgb"].double().cpu(), [h["tdist"].double().cpu() for h in hist])
    pairs = {f"{x} vs {y}": _stage1_pair(ev[x], ev[y]) for x, y in (
        ("oracle_fp32_cpu", "oracle_fp64"), ("oracle_fp32_rocm", "oracle_fp64"), ("oracle_fp32_cpu", "oracle_fp32_rocm"),
        ("hip", "oracle_fp64"), ("hip", "oracle_fp32_rocm"), ("hip", "oracle_fp32_cpu"))}
    record("selfnoise.stage1[1024 rays x 64/64/32 samples, jitter seed 11]", pairs)
    ref = [pairs[k] for k in pairs if not k.startswith("hip")]
    noise_moved = max(r["rays_with_moved_samples"] for r in ref)
    noise_over = max(r["rays_over_1e-4"] for r in ref)
    noise_rgb = max(r["rgb_linf_moved_samples"] for r in ref)
    # measured (profiles/r03_parity_counts.json): oracle fp32-CPU vs fp32-ROCm move a sample on 58 of the 1024 rays, each of
    # them vs float64 on 179 (worst RGB on such rays 9.4e-5); the HIP path: 69 / 90 rays vs the two fp32 evaluations, 206 vs
    # float64, 1 ray over 1e-4 (1.2-1.6e-4)
    for k in ("hip vs oracle_fp64", "hip vs oracle_fp32_rocm", "hip vs oracle_fp32_cpu"):
        h = pairs[k]
        assert h["rgb_linf_same_samples"] < 1e-4, (k, h)                                         # the north-star tolerance, identical samples
        assert h["rays_with_moved_samples"] <= 1.5 * noise_moved + 8, (k, h, noise_moved)
        assert h["rays_over_1e-4"] <= noise_over + 3, (k, h, noise_over)
        assert h["rgb_linf_moved_samples"] <= 2 * noise_rgb + 1e-4, (k, h, noise_rgb)
    # and the reference's own two fp32 evaluations are NOT within the tolerance of each other on every ray -- the premise
    assert noise_moved > 0


def _stage3_pair(a, b):
    """a, b: (rgb [B,3], idx_fg [B] bool, dense total_order [B,160])."""
    same_fg = a[1] == b[1]
    both = a[1] & b[1]
    same_order = torch.ones_like(same_fg)
    same_order[both] = (a[2][both] == b[2][both]).all(-1)
    diff = (a[0] - b[0]).abs().max(-1).values
    ok = same_fg & same_order
    sw = same_fg & ~same_order
    return {"fg_flips": int((~same_fg).sum()), "rays_with_a_swapped_pair": int(sw.sum()), "rgb_linf_same_order": float(diff[ok].max()),
            "rgb_linf_swapped": float(diff[sw].max()) if bool(sw.any()) else 0.0, "rays_over_1e-4": int((diff > 1e-4).sum())}


def test_stage3_fullsize_bounds_are_reference_self_noise():
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets
    dev = torch.device("cuda")
    B = 2048
    b = synth.add_patch_supervision(synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5), 2, 32, 778)
    g = torch.Generator().manual_seed(5)
    t_rand = torch.rand(B, 128, generator=g)
    jit = [torch.rand(B, generator=g) for _ in range(3)]
    bsd, hsd = synth.background_state_dict(777, 2), synth.human_state_dict(777, 2)

    def oracle(device, dtype):
        bb = _cast(b, device, dtype)
        bk = {"rays_o": bb["rays_o_bkg"], "rays_d": bb["rays_d_bkg"], "viewdirs": bb["viewdirs_bkg"], "radii": bb["radii"], "times": b["time"]}
        with torch.no_grad():
            _, hist = ob.mipnerf360_forward(_cast(bsd, device, dtype), bk, 1.0, True, 0.1, 1e6, transitions_times=[0.4],
                                            jitters=[j.view(B, 1) for j in jit], render=False)
            human = oh.human_forward(_cast(hsd, device, dtype), bb, transitions_times=[0.4], t_rand=t_rand.to(device=device, dtype=dtype), stage=3)
            rgb, fg, order, _, _ = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human, bk["rays_o"], bk["rays_d"],
                                                       bb["newsmpl_to_scale_world"])
        dense = torch.zeros(B, 160, dtype=torch.int64)
        dense[fg.cpu()] = order.cpu().long()
        return rgb.double().cpu(), fg.cpu().bool(), dense

    ev = {"oracle_fp32_cpu": oracle("cpu", torch.float32), "oracle_fp32_rocm": oracle(dev, torch.float32), "oracle_fp64": oracle(dev, torch.float64)}
    cfg = default_cfg(_basedir())
    cfg.perturb = 1.0
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(bsd, strict=False)
    hos.human.load_state_dict(hsd, strict=True)
    hos = hos.to(dev)
    gb = batch_to_device(prepare_patch_targets(b), dev)
    with torch.no_grad():
        out = hos.render(gb, randomized=True, is_train=True, jitters=[j.to(dev) for j in jit], t_rand=t_rand.to(dev))
    fg_h = out["idx_fg"].bool().cpu()
    dense = torch.zeros(B, 160, dtype=torch.int64)
    dense[fg_h] = out["total_order"].cpu().long()[fg_h]
    ev["hip"] = (out["rgb"].double().cpu(), fg_h, dense)
    pairs = {f"{x} vs {y}": _stage3_pair(ev[x], ev[y]) for x, y in (
        ("oracle_fp32_cpu", "oracle_fp64"), ("oracle_fp32_rocm", "oracle_fp64"), ("oracle_fp32_cpu", "oracle_fp32_rocm"),
        ("hip", "oracle_fp64"), ("hip", "oracle_fp32_rocm"), ("hip", "oracle_fp32_cpu"))}
    record("selfnoise.stage3[2048 rays, 160 merged samples, seed 778 / draws 5]", pairs)
    ref = [pairs[k] for k in pairs if not k.startswith("hip")]
    noise_sw = max(r["rays_with_a_swapped_pair"] for r in ref)
    noise_fg = max(r["fg_flips"] for r in ref)
    noise_rgb = max(r["rgb_linf_swapped"] for r in ref)
    # measured (profiles/r03_parity_counts.json): the reference's two fp32 evaluations swap a coinciding pair on 10 of the 2048
    # rays (worst 1.8e-3, 3 rays over 1e-4) and each is 19-21 swapped rays / 1.0e-4 on same-order rays away from float64; the
    # HIP path: 2 and 8 swapped rays against the two fp32 evaluations, 19 against float64
    for k in ("hip vs oracle_fp32_rocm", "hip vs oracle_fp32_cpu"):
        assert pairs[k]["rgb_linf_same_order"] < 1e-4, (k, pairs[k])             # the north-star tolerance: against the fp32 reference
    e64 = max(pairs[k]["rgb_linf_same_order"] for k in ("oracle_fp32_cpu vs oracle_fp64", "oracle_fp32_rocm vs oracle_fp64"))
    assert pairs["hip vs oracle_fp64"]["rgb_linf_same_order"] <= 1.5 * e64 + 1e-5, (pairs["hip vs oracle_fp64"], e64)
    for k in ("hip vs oracle_fp64", "hip vs oracle_fp32_rocm", "hip vs oracle_fp32_cpu"):
        h = pairs[k]
        assert h["rays_with_a_swapped_pair"] <= noise_sw + 2, (k, h, noise_sw)
        assert h["fg_flips"] <= noise_fg + 2, (k, h, noise_fg)
        assert h["rgb_linf_swapped"] <= 2 * noise_rgb + 1e-4, (k, h, noise_rgb)
    assert noise_sw > 0, "premise: two fp32 evaluations of the reference's own graph order some coinciding pair differently"
