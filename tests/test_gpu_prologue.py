"""SURVEY rows P2 / P3 as HIP kernels (hos_pose_refine_*, hos_motion_basis_*): forward against the reference's fixtures
(tests/golden/human_parts.npz: `pose_Rs`, `pose_Ts` from the reference's BodyPoseRefiner, `mb_*` from its
MotionBasisComputer) and against the oracle, backward against the oracle's autograd in float64."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

import oracle.human as oh
from hosnerf_amd import ops, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


@pytest.fixture(scope="module")
def net(dev):
    from hosnerf_amd.human_nerf import Network, default_cfg
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    n = Network(default_cfg(d))
    n.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    return n.to(dev)


def maxerr(a, b):
    b = torch.from_numpy(np.asarray(b)) if not isinstance(b, torch.Tensor) else b
    return float((a.detach().double().cpu() - b.double().cpu()).abs().max())


def test_pose_refiner_vs_reference_fixture(dev, net):
    """Identity rotations / zero translations in -> the kernel's composed output IS the decoder's (dR, dT): compare with what
    the reference's `pose_decoder` returned for the same weights and pose vector."""
    hp = np.load(os.path.join(HERE, "golden", "human_parts.npz"))
    b = synth.human_batch(8, seed=3)
    K = 26
    Rs = torch.eye(3, device=dev).expand(1, K, 3, 3).contiguous()
    Ts = torch.zeros(1, K, 3, device=dev)
    with torch.no_grad():
        Ro, To = net._pose_refine(Rs, Ts, b["dst_posevec"][None].to(dev))
    assert maxerr(Ro[0, 1:], hp["pose_Rs"][0]) < 2e-6 and maxerr(To[0, 1:], hp["pose_Ts"][0]) < 2e-6
    assert maxerr(Ro[0, 0], torch.eye(3)) == 0.0 and maxerr(To[0, 0], torch.zeros(3)) == 0.0


def test_prologue_two_frames_vs_oracle(dev, net):
    """F = 2 (current + previous frame, as a training step runs it): refined joints and all four motion bases against the
    oracle (torch.inverse on the non-orthonormal refined chain, U:134-174)."""
    b = synth.human_batch(8, seed=21)
    sd = synth.human_state_dict(777, 2)
    Rs = torch.stack([b["dst_Rs"], b["dst_Rs_prev"]], 0).to(dev)
    Ts = torch.stack([b["dst_Ts"], b["dst_Ts_prev"]], 0).to(dev)
    pv = torch.stack([b["dst_posevec"], b["dst_posevec_prev"]], 0).to(dev)
    with torch.no_grad():
        Ro, To = net._pose_refine(Rs, Ts, pv)
        Rb, Tb, Rf, Tf = net._motion_basis(Ro, To, b["cnl_gtfms"].to(dev))
    for f, (r, t, p) in enumerate(((b["dst_Rs"], b["dst_Ts"], b["dst_posevec"]), (b["dst_Rs_prev"], b["dst_Ts_prev"], b["dst_posevec_prev"]))):
        dR, dT = oh.pose_refiner(sd, p[None])
        r2 = torch.cat([r[0:1], torch.matmul(r[1:], dR[0])], 0)
        t2 = torch.cat([t[0:1], t[1:] + dT[0]], 0)
        assert maxerr(Ro[f], r2) < 2e-6 and maxerr(To[f], t2) < 2e-6
        want = oh.motion_basis(r2, t2, b["cnl_gtfms"])
        for got, w in zip((Rb[f], Tb[f], Rf[f], Tf[f]), want):
            assert maxerr(got, w) < 1e-5, (f, maxerr(got, w))


def test_prologue_backward_vs_fp64_autograd(dev, net):
    """Random cotangents on the four bases of both frames: gradients of every pose-decoder parameter against the oracle's
    autograd evaluated in float64 (the kernels are plain fp32 FMA chains)."""
    b = synth.human_batch(8, seed=22)
    K = 26
    sd64 = {k: v.double().requires_grad_(True) for k, v in synth.human_state_dict(777, 2, ).items() if k.startswith("pose_decoder.")}
    # the reference initialises the last head layers at +-1e-5; larger values exercise Rodrigues' backward properly
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for head in ("dstR", "dstT"):
            w = torch.randn(75, 256, generator=g) * 0.02
            sd64[f"pose_decoder.block_mlps_{head}.2.weight"].copy_(w.double())
            dict(net.named_parameters())[f"pose_decoder.block_mlps_{head}.2.weight"].copy_(w.to(dev))
    cot = [torch.randn(2, K, 3, 3, generator=g), torch.randn(2, K, 3, generator=g), torch.randn(2, K, 3, 3, generator=g), torch.randn(2, K, 3, generator=g)]
    frames = ((b["dst_Rs"], b["dst_Ts"], b["dst_posevec"]), (b["dst_Rs_prev"], b["dst_Ts_prev"], b["dst_posevec_prev"]))
    loss64 = 0.0
    for f, (r, t, p) in enumerate(frames):
        dR, dT = oh.pose_refiner(sd64, p[None].double())
        r2 = torch.cat([r[0:1].double(), torch.matmul(r[1:].double(), dR[0])], 0)
        t2 = torch.cat([t[0:1].double(), t[1:].double() + dT[0]], 0)
        outs = oh.motion_basis(r2, t2, b["cnl_gtfms"].double())
        loss64 = loss64 + sum((o * c[f].double()).sum() for o, c in zip(outs, cot))
    loss64.backward()

    net.zero_grad()
    Rs = torch.stack([fr[0] for fr in frames], 0).to(dev)
    Ts = torch.stack([fr[1] for fr in frames], 0).to(dev)
    pv = torch.stack([fr[2] for fr in frames], 0).to(dev)
    Ro, To = net._pose_refine(Rs, Ts, pv)
    outs = net._motion_basis(Ro, To, b["cnl_gtfms"].to(dev))
    loss = sum((o * c.to(dev)).sum() for o, c in zip(outs, cot))
    assert abs(float(loss.detach()) - float(loss64.detach())) < 1e-4 * max(1.0, abs(float(loss64.detach())))
    loss.backward()
    net.scatter_compact_grads()
    params = dict(net.named_parameters())
    for name, p64 in sd64.items():
        got, want = params[name].grad.detach().double().cpu(), p64.grad
        rel = float((got - want).norm() / (want.norm() + 1e-30))
        assert rel < 1e-4, (name, rel)
    net.zero_grad()
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)


def test_motion_basis_partial_cotangents(dev, net):
    """Only some of the four outputs carry a gradient (eval of the flow frame uses the forward bases alone): NULL
    cotangents must behave as zeros."""
    b = synth.human_batch(8, seed=23)
    Rs = b["dst_Rs"][None].to(dev).requires_grad_(True)
    Ts = b["dst_Ts"][None].to(dev).requires_grad_(True)
    _, _, Rf, Tf = ops.motion_basis(Rs, Ts, b["cnl_gtfms"].to(dev))
    (Rf.sum() + 2.0 * Tf.sum()).backward()
    r64 = b["dst_Rs"].double().requires_grad_(True)
    t64 = b["dst_Ts"].double().requires_grad_(True)
    o = oh.motion_basis(r64, t64, b["cnl_gtfms"].double())
    (o[2].sum() + 2.0 * o[3].sum()).backward()
    assert maxerr(Rs.grad[0], r64.grad) < 1e-4 * float(r64.grad.abs().max())
    assert maxerr(Ts.grad[0], t64.grad) < 1e-4 * float(t64.grad.abs().max())
