"""How exact is "exact"?  Several gradients of the human branch are ill-conditioned in fp32 by construction of the reference's
graph -- x = sum(w q) / max(sum w, 1e-4) where the skinning weights vanish, Fourier features up to frequency 512 behind it --
so the reference's own fp32 arithmetic is far from the fp64 value of the same graph (tens of percent on the pose decoder
with random-init weights).  A fixed tolerance against the fp32 oracle would either be meaningless or fail on noise; this test
pins the statement that matters instead: against the fp64 oracle the HIP path is as accurate as the fp32 oracle is."""
import json
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.human as oh
from hosnerf_amd import synth

NAMES = ("pose_decoder.block_mlps_dstR.2.weight", "mweight_vol_decoder.const_embedding", "non_rigid_mlp.block_mlps.4.weight",
         "non_rigid_forward_mlp.block_mlps.0.weight", "cnl_mlp.pts_linears.2.weight", "cnl_mlp.output_linear.0.weight")


def _loss(out):
    return ((out["human_rgb"] ** 2).mean() + (out["human_density"] ** 2).mean() * 1e-3
            + (out["deform_pts_prev_final"] ** 2).mean() * 1e-3 + (out["deform_pts_final"] ** 2).mean() * 1e-3)


def test_hip_gradients_as_accurate_as_the_fp32_reference_graph():
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import batch_to_device
    dev = torch.device("cuda")
    B = 96
    b = synth.human_batch(B, seed=777, time=0.5, is_train=True, iter_val=3e5)
    t_rand = torch.rand(B, 128, generator=torch.Generator().manual_seed(3))
    grads = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in synth.human_state_dict(777, 2).items()}
        bb = {k: (v.to(dt) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in b.items()}
        out = oh.human_forward(sd, bb, transitions_times=[0.4], t_rand=t_rand.to(dt), stage=3)
        _loss(out).backward()
        grads[tag] = {n: sd[n].grad.double() for n in NAMES}
        if tag == "f64":
            truth_rgbm = (out["human_rgb"] * out["pts_mask"][..., None]).detach()
    d = tempfile.mkdtemp(prefix="hos_cond_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    cfg = default_cfg(d)
    cfg.perturb = 1.0
    net = Network(cfg, stage=3)
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    net = net.to(dev)
    out = net(**batch_to_device(b, dev), t_rand=t_rand.to(dev))
    _loss(out).backward()
    net.scatter_compact_grads()
    hip = {k: v.grad.double().cpu() for k, v in net.named_parameters() if k in NAMES}
    # forward, mask weighted (what the composite consumes): fp32-grade against fp64
    got = (out["human_rgb"] * out["pts_mask"][..., None]).detach().double().cpu()
    assert float((got - truth_rgbm).abs().max()) < 5e-5
    report = {}
    for n in NAMES:
        t = grads["f64"][n]
        s = float(t.abs().max())
        e_ref = float((grads["f32"][n] - t).abs().max()) / s
        e_hip = float((hip[n].reshape(t.shape) - t).abs().max()) / s
        report[n] = (e_ref, e_hip)
        assert e_hip < 2.0 * e_ref + 2e-4, (n, e_ref, e_hip)
    # the well-conditioned part of the graph is simply accurate
    assert report["cnl_mlp.output_linear.0.weight"][1] < 1e-4 and report["cnl_mlp.pts_linears.2.weight"][1] < 5e-4, report
