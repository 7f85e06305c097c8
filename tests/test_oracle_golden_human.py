"""Pin oracle/human.py (human-object branch P1-P10, stage-3 composite C1-C3) against the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

import oracle.background as ob
import oracle.human as oh
from hosnerf_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


@pytest.fixture(scope="module")
def hp():
    return load("human_parts.npz")


@pytest.fixture(scope="module")
def hsd():
    return synth.human_state_dict(777, 2)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    a, b = a.astype(np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    assert np.all(err <= atol + rtol * np.abs(b)), f"max err {err.max():.3e} (atol {atol}, rtol {rtol})"


def test_prologue(hp, hsd):
    b = synth.human_batch(8, seed=3)
    close(oh.rodrigues(T(hp["rod_in"])), hp["rod_out"], 1e-6)
    Rs, Ts = oh.pose_refiner(hsd, b["dst_posevec"][None])
    close(Rs, hp["pose_Rs"], 1e-6)
    close(Ts, hp["pose_Ts"], 1e-6)
    R, Tt, Rf, Tf = oh.motion_basis(b["dst_Rs"], b["dst_Ts"], b["cnl_gtfms"])
    close(R, hp["mb_R"], 2e-6); close(Tt, hp["mb_T"], 2e-6); close(Rf, hp["mb_Rf"], 2e-6); close(Tf, hp["mb_Tf"], 2e-6)
    vol = oh.motion_weight_volume(hsd, b["motion_weights_priors"])
    close(vol[:, ::4, ::4, ::4], hp["vol_sub"], 1e-6)
    assert abs(float(vol.double().sum()) - float(hp["vol_sum"])) < 1e-2


def test_lbs(hp, hsd):
    b = synth.human_batch(8, seed=3)
    vol = oh.motion_weight_volume(hsd, b["motion_weights_priors"])
    R, Tt, Rf, Tf = (T(hp[k]) for k in ("mb_R", "mb_T", "mb_Rf", "mb_Tf"))
    x, m = oh.backward_lbs(T(hp["lbs_pts"]), R, Tt, vol, b["cnl_bbox_min_xyz"], b["cnl_bbox_scale_xyz"])
    close(m[:, 0], hp["lbs_mask"], 2e-6)
    close(x, hp["lbs_x_skel"], 2e-5, 1e-5)
    assert float(np.max(hp["lbs_mask"])) > 0.5 and float(np.min(hp["lbs_mask"])) < 1e-3   # inside and outside the body
    xf = oh.forward_lbs(T(hp["flbs_pts"]), Rf, Tf, vol, b["cnl_bbox_min_xyz"], b["cnl_bbox_scale_xyz"])
    close(xf, hp["flbs_x"], 2e-5, 1e-5)


def test_embedders_and_mlps(hp, hsd):
    b = synth.human_batch(8, seed=3)
    cn = T(hp["flbs_pts"])
    for it in (0, 150000, 300000):
        w = oh.hannw_weights(float(it), 100000, 200000, 6)
        close(oh.hannw_embed(cn, w), hp[f"hann_{it}"], 2e-6)
    close(oh.fourier_embed(cn), hp["fourier"], 2e-6)
    w = oh.hannw_weights(3e5, 100000, 200000, 6)
    cond = b["dst_posevec"][None]
    close(oh.nonrigid_mlp(hsd, "non_rigid_mlp.", oh.hannw_embed(cn, w), cn, cond), hp["nonrigid_xyz"], 2e-6)
    close(oh.nonrigid_mlp(hsd, "non_rigid_forward_mlp.", oh.hannw_embed(cn, w), cn, cond), hp["nonrigid_fwd_xyz"], 2e-6)
    emb = torch.cat([oh.fourier_embed(cn), hsd["human_stateembeds.1"].repeat(96, 1)], -1)
    close(oh.canonical_mlp(hsd, emb), hp["cnl_raw"], 2e-5)


def test_raw2outputs(hp):
    raw = T(hp["r2o_raw"])
    out = oh.raw2outputs(raw[..., :3], raw[..., 3], T(hp["r2o_z"]), T(hp["r2o_d"]), T(hp["r2o_mask"]))
    close(out[0], hp["r2o_rgb"], 1e-6); close(out[1], hp["r2o_acc"], 1e-6); close(out[2], hp["r2o_w"], 1e-6); close(out[3], hp["r2o_depth"], 1e-5)
    out = oh.raw2outputs(raw[..., :3], raw[..., 3], T(hp["r2o_z"]), T(hp["r2o_d"]), T(hp["r2o_mask"]), torch.tensor([10.0, 120.0, 250.0]))
    close(out[0], hp["r2o_rgb_bg"], 1e-6)


@pytest.mark.parametrize("tag", ["evalA", "trainA", "earlyB", "t0C"])
def test_forward_s3(hsd, tag):
    hf = load("human_forward.npz")
    p = f"s3_{tag}_"
    time, is_train, it, perturb = hf[p + "meta"]
    b = synth.human_batch(8, seed=21, time=float(time), is_train=bool(is_train), iter_val=float(it))
    t_rand = T(hf[p + "t_rand"]) if perturb > 0 else None
    with torch.no_grad():
        out = oh.human_forward(hsd, b, transitions_times=[0.4], t_rand=t_rand)
    close(out["newsmpl_pts"], hf[p + "newsmpl_pts"], 2e-6)
    close(out["pts_mask"], hf[p + "pts_mask"], 5e-6)
    # x_skel = sum(w_i x_i)/clamp(sum w_i, 1e-4) is ill-conditioned where the skinning mask is ~0 (and is then
    # amplified by the 2^9 Fourier band); those samples are multiplied by the mask in the composite (M:85-86),
    # so per-sample outputs are compared mask-weighted, plus a loose unweighted bound
    m = hf[p + "pts_mask"]
    close(out["human_rgb"] * T(m)[..., None], hf[p + "human_rgb"] * m[..., None], 2e-5)
    close(out["human_density"] * T(m), hf[p + "human_density"] * m, 1e-4, 1e-4)
    close(out["human_rgb"], hf[p + "human_rgb"], 2e-3)
    close(out["observe_pts"], hf[p + "observe_pts"], 2e-6)
    close(out["deform_pts_final"], hf[p + "deform_pts_final"], 5e-5)
    if (p + "deform_pts_prev_final") in hf:
        close(out["deform_pts_prev_final"], hf[p + "deform_pts_prev_final"], 3e-4)   # all samples, incl. mask ~ 0 (ill-conditioned blend)
    else:
        assert "deform_pts_prev_final" not in out
        close(out["z_vals"], hf[p + "z_vals"], 1e-6)


@pytest.mark.parametrize("tag,time,is_train,it,perturb", [("evalA", 0.5, False, 3e5, 0.0), ("trainA", 0.5, True, 3e5, 1.0),
                                                          ("earlyB", 0.3, True, 1000.0, 0.0), ("t0C", 0.0, True, 3e5, 0.0)])
def test_forward_s2(hsd, tag, time, is_train, it, perturb):
    """oracle.human_forward(stage=2) against the reference's STAGE-2 Network (2nd_State_Conditional_Human-Object/core/nets/
    human_nerf/network.py:273-299, 538-556: composites inside, returns rgb / alpha / depth / weights), same items and jitter
    as the stage-3 cases."""
    hf = load("human_forward.npz")
    p = f"s2_{tag}_"
    b = synth.human_batch(8, seed=21, time=time, is_train=is_train, iter_val=it)
    t_rand = T(hf[f"s3_{tag}_t_rand"]) if perturb > 0 else None
    with torch.no_grad():
        out = oh.human_forward(hsd, b, transitions_times=[0.4], t_rand=t_rand, stage=2)
    close(out["rgb"], hf[p + "rgb"], 2e-5)
    close(out["alpha"], hf[p + "alpha"], 2e-5)
    close(out["weights"], hf[p + "weights"], 2e-5)
    close(out["depth"], hf[p + "depth"], 1e-4)
    close(out["observe_pts"], hf[p + "observe_pts"], 2e-6)
    close(out["deform_pts_final"], hf[p + "deform_pts_final"], 5e-5)
    ref_keys = set(hf[p + "keys"].tolist())
    assert ref_keys <= set(out.keys()), ref_keys - set(out.keys())
    if (p + "deform_pts_prev_final") in hf:
        close(out["deform_pts_prev_final"], hf[p + "deform_pts_prev_final"], 3e-4)
    else:
        assert "deform_pts_prev_final" not in out and "deform_pts_prev_final" not in ref_keys


@pytest.mark.parametrize("tag,B,seed", [("A", 16, 31), ("tinyd", 8, 32), ("nofg", 8, 33)])
def test_stage3_composite(hsd, tag, B, seed):
    st = load("stage3_step.npz")
    p = f"c_{tag}_"
    b = synth.human_batch(B, seed=seed, time=0.5, is_train=True, iter_val=3e5)
    if tag == "tinyd":
        b["rays_d_bkg"][0, 0] = 1e-7
        b["rays_d_bkg"][1, 1] = 5e-6
    if tag == "nofg":
        b["near"] += 50.0
        b["far"] += 50.0
    bsd = synth.background_state_dict(777, 2)
    bb = {"rays_o": b["rays_o_bkg"], "rays_d": b["rays_d_bkg"], "viewdirs": b["viewdirs_bkg"], "radii": b["radii"], "times": b["time"]}
    with torch.no_grad():
        _, hist = ob.mipnerf360_forward(bsd, bb, 1.0, True, 0.1, 1e6, transitions_times=[0.4],
                                        jitters=[T(st[p + f"jitter{l}"]) for l in range(3)], render=False)
        human = oh.human_forward(hsd, b, transitions_times=[0.4])
        rgb, idx_fg, order, hw, _ = oh.stage3_composite(hist[-1]["tdist"], hist[-1]["rgb"], hist[-1]["density"], human,
                                                       b["rays_o_bkg"], b["rays_d_bkg"], b["newsmpl_to_scale_world"])
    assert np.array_equal(idx_fg.numpy(), st[p + "idx_fg"])
    assert np.array_equal(order.numpy(), st[p + "total_order"]), "merge order must be bit-exact"
    close(rgb, st[p + "rgb"], 5e-5)
    close(hw, st[p + "human_weights_onlyfg"], 2e-5)
    if tag == "nofg":
        assert not idx_fg.any()
    else:
        assert idx_fg.any()
