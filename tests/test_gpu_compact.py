"""Device-side cycle-set selection (hos_compact_rows / hos_scatter_rows) against torch.nonzero / index_select, and the
fixed-capacity (`static_cycle=True`) form of the network against the reference-shaped form: same selected rows, same
losses, same parameter gradients."""
import json
import os
import tempfile

import pytest
import torch

from hosnerf_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("P,frac", [(1, 1.0), (5, 0.0), (1000, 0.3), (1024, 1.0), (262144, 0.27), (300001, 0.5), (4096 * 128, 0.9)])
def test_compact_rows_matches_nonzero(P, frac):
    g = torch.Generator().manual_seed(P)
    mask = torch.rand(P, generator=g).to(DEV)
    thr = 1.0 - frac
    a = torch.randn(P, 3, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(P, 3, generator=g).to(DEV)
    a_sel, b_sel, sel, count = ops.compact_rows(mask, thr, a, b)
    want = torch.nonzero(mask > thr).reshape(-1)
    n = int(count)
    assert n == want.numel()
    assert torch.equal(sel[:n].long(), want) and bool((sel[n:] == -1).all())
    assert torch.equal(a_sel[:n], a.detach()[want]) and torch.equal(b_sel[:n], b[want])
    assert float(a_sel[n:].abs().sum()) == 0.0 and float(b_sel[n:].abs().sum()) == 0.0
    cot = torch.randn(P, 3, generator=g).to(DEV)
    (a_sel * cot).sum().backward()
    ref = torch.zeros(P, 3, device=DEV)
    ref[want] = cot[:n]
    assert torch.equal(a.grad, ref)


def test_static_cycle_equals_reference_shaped_cycle():
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets, stage2_losses
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    cfg = default_cfg(d)
    cfg.perturb = 0.0
    net = Network(cfg, stage=2)
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    net = net.to(DEV)
    b = synth.add_patch_supervision(synth.human_batch(64, seed=5, time=0.5, is_train=True, iter_val=3e5), 1, 8, 5)
    gb = batch_to_device(prepare_patch_targets(b), DEV)
    res = {}
    for static in (False, True):
        net.zero_grad()
        out = net(static_cycle=static, **gb)
        total, parts = stage2_losses(out, gb)
        total.backward()
        res[static] = (out, float(total), {k: float(v) for k, v in parts.items()}, net.flat_grad.clone())
    o0, o1 = res[False][0], res[True][0]
    n = int(o1["cycle_count"])
    assert n == o0["observe_pts"].shape[0] and 0 < n < 64 * 128
    assert torch.equal(o1["observe_pts"][:n], o0["observe_pts"])
    assert float((o1["deform_pts_final"][:n] - o0["deform_pts_final"]).abs().max()) < 1e-6
    assert abs(res[True][1] - res[False][1]) < 1e-7 * max(1.0, abs(res[False][1]))
    for k in ("mse", "flow", "cycle"):
        assert abs(res[True][2][k] - res[False][2][k]) <= 1e-6 * abs(res[False][2][k]) + 1e-12, k
    g0, g1 = res[False][3], res[True][3]
    assert float((g1 - g0).norm() / g0.norm()) < 1e-5
