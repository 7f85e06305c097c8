"""hos_mlp_chain128_fwd (the non-rigid MLP forward in one launch, activations on chip) against the layer-by-layer path and
against float64: every hidden activation and the output, ragged row counts, the device-side row limit, and the reference's
fixture for the whole MLP (`nonrigid_xyz`)."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from hosnerf_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def net():
    from hosnerf_amd.human_nerf import Network, default_cfg
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    n = Network(default_cfg(d))
    n.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    return n.to(DEV)


def _inputs(P, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(P, 3, generator=g) * 2 - 1).to(DEV)
    cond = (torch.randn(75, generator=g) * 0.3).to(DEV)
    band = torch.tensor([1.0, 1.0, 0.8, 0.3, 0.0, 0.0]).to(DEV)
    return x, cond, band


@pytest.fixture(params=[True, False], ids=["fold", "rows"])
def fold(request):
    """Both forms of the chain: the condition code folded into the first layer's bias (default), and as columns of every row."""
    prev = ops.MLP_CHAIN_FOLD
    ops.MLP_CHAIN_FOLD = request.param
    yield request.param
    ops.MLP_CHAIN_FOLD = prev


def _both(net, specs, x, cond, band, rows_dev=None):
    prev_c, prev_m = ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS
    try:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS = True, 1
        xyz_c, (Ec, _, acts_c, fold_c) = net._nonrigid_fwd(specs, x, cond, band, save=True, rows_dev=rows_dev)
        assert (Ec is None) == ops.MLP_CHAIN_FOLD and (fold_c is not None) == ops.MLP_CHAIN_FOLD
        ops.MLP_CHAIN = False
        xyz_l, (E, PE, acts_l, _) = net._nonrigid_fwd(specs, x, cond, band, save=True, rows_dev=rows_dev)
    finally:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS = prev_c, prev_m
    return xyz_c, acts_c, xyz_l, acts_l, E, PE


def _ref64(net, specs, E, PE, x):
    h = E.double()
    acts = []
    for i in range(6):
        W, b = net._w(specs[i])
        W, b = W.double(), b.double()
        inp = torch.cat([h, PE.double()], 1) if i == 4 else h
        h = torch.relu(inp @ W[:128, :inp.shape[1]].T + b[:128])
        acts.append(h)
    W, b = net._w(specs[6])
    return x.double() + h @ W.double()[:3, :128].T + b.double()[:3], acts


@pytest.mark.parametrize("P", [1, 31, 128, 129, 1000, 4096 + 77, 65536])
def test_chain_matches_layers_and_fp64(net, P, fold):
    for which, specs in (("nr", net._nr), ("nrf", net._nrf)):
        x, cond, band = _inputs(P, seed=P)
        with torch.no_grad():
            xyz_c, acts_c, xyz_l, acts_l, E, PE = _both(net, specs, x, cond, band)
            ref, racts = _ref64(net, specs, E, PE, x)
        for l in range(6):
            scale = max(1.0, float(racts[l].abs().max()))
            assert float((acts_c[l].double() - racts[l]).abs().max()) < 2e-6 * scale, (which, l)
            assert float((acts_c[l] - acts_l[l]).abs().max()) < 2e-6 * scale, (which, l)
        assert float((xyz_c.double() - ref).abs().max()) < 1e-6
        assert float((xyz_c - xyz_l).abs().max()) < 1e-6


def test_chain_respects_the_device_row_count(net, fold):
    P, n = 5000, 1234
    x, cond, band = _inputs(P, seed=3)
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        full = _both(net, net._nr, x, cond, band)
        prev = ops.MLP_CHAIN_MIN_ROWS
        ops.MLP_CHAIN_MIN_ROWS = 1
        try:
            xyz = torch.full((P, 3), 7.0, device=DEV)
            E = torch.empty(P, 128, device=DEV); PE = torch.empty(P, 64, device=DEV)
            ops.embed_hannw(x, band, cond, E, PE)
            bufs = ops.mlp_chain_buffers(DEV)
            ws = [net._w(L) for L in net._nr]
            acts = [torch.full((P, 128), 7.0, device=DEV) for _ in range(6)]
            if fold:
                ops.mlp_chain_pack_fold([w for w, _ in ws], [b for _, b in ws], cond, 36, bufs[0], bufs[1], torch.empty(128, 64, device=DEV))
                ops.mlp_chain128_fwd(None, PE, x, bufs[0], bufs[1], acts, xyz, rows_dev=cnt)
            else:
                ops.mlp_chain_pack([w for w, _ in ws], [b for _, b in ws], bufs[0], bufs[1])
                ops.mlp_chain128_fwd(E, PE, x, bufs[0], bufs[1], acts, xyz, rows_dev=cnt)
        finally:
            ops.MLP_CHAIN_MIN_ROWS = prev
    assert torch.equal(xyz[:n], full[0][:n]) and bool((xyz[n:] == 7.0).all())
    assert torch.equal(acts[5][:n], full[1][5][:n]) and bool((acts[5][n:] == 7.0).all())


def test_chain_vs_reference_fixture(net, fold):
    """The reference's NonRigidMotionMLP on its own hann embedding (tests/golden/human_parts.npz: nonrigid_xyz)."""
    hp = np.load(os.path.join(HERE, "golden", "human_parts.npz"))
    b = synth.human_batch(8, seed=3)
    cn = torch.from_numpy(hp["flbs_pts"]).to(DEV)
    band = net._band_weights(3e5, DEV)
    prev_c, prev_m = ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS
    ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS = True, 1
    try:
        with torch.no_grad():
            xyz, _ = net._nonrigid_fwd(net._nr, cn, b["dst_posevec"].to(DEV), band, save=False)
            xyzf, _ = net._nonrigid_fwd(net._nrf, cn, b["dst_posevec"].to(DEV), band, save=False)
    finally:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS = prev_c, prev_m
    assert float((xyz.cpu() - torch.from_numpy(hp["nonrigid_xyz"])).abs().max()) < 2e-6
    assert float((xyzf.cpu() - torch.from_numpy(hp["nonrigid_fwd_xyz"])).abs().max()) < 2e-6


@pytest.mark.parametrize("P,n_live", [(4096 + 77, None), (20000, 7777)])
def test_folded_chain_backward_equals_the_row_form(net, P, n_live):
    """`_NonRigidFn` through both forms of the chain: d loss / d x and EVERY parameter gradient of the MLP.  The folded first
    layer gets its weight gradient in two parts (hann columns from the GEMM over PE, condition columns as bias-gradient (x)
    cond) -- against the row form, where all 111 columns come out of one GEMM over E, and against float64 autograd."""
    from hosnerf_amd.human_nerf import _NonRigidFn
    x, cond, band = _inputs(P, seed=11)
    g = torch.randn(P, 3, generator=torch.Generator().manual_seed(5)).to(DEV)
    rows_dev = None if n_live is None else torch.tensor([n_live], dtype=torch.int32, device=DEV)
    n = P if n_live is None else n_live
    if n_live is not None:
        g[n:] = 0
    res = {}
    prev = ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_FOLD
    try:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS = True, 1
        for f in (True, False):
            ops.MLP_CHAIN_FOLD = f
            net.zero_grad()
            xx = x.clone().requires_grad_(True)
            xyz = _NonRigidFn.apply(torch.zeros((), device=DEV, requires_grad=True), net, "nr", xx, cond, band, rows_dev)
            xyz.backward(g)
            res[f] = (xyz.detach()[:n].clone(), xx.grad[:n].clone(),
                      {k: v.grad.detach().clone() for k, v in net.named_parameters() if k.startswith("non_rigid_mlp.")})
    finally:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_FOLD = prev
        net.zero_grad()
    # float64 autograd of the same MLP on the live rows
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in net.named_parameters() if k.startswith("non_rigid_mlp.")}
    x64 = x[:n].double().clone().requires_grad_(True)
    freqs = 2.0 ** torch.arange(6, dtype=torch.float64, device=DEV)
    ang = x64[:, None, :] * freqs[None, :, None]
    pe = (torch.cat([torch.sin(ang), torch.cos(ang)], -1) * band.double()[None, :, None]).reshape(n, 36)
    names = sorted({k.rsplit(".", 1)[0] for k in sd}, key=lambda s_: int(s_.split(".")[-1]))
    h = torch.cat([cond.double()[None].expand(n, -1), pe], 1)
    for i, nm in enumerate(names):
        if i == 4:
            h = torch.cat([h, pe], 1)
        h = h @ sd[nm + ".weight"].T + sd[nm + ".bias"]
        if i < len(names) - 1:
            h = torch.relu(h)
    (x64 + h).backward(g[:n].double())
    assert float((res[True][0] - res[False][0]).abs().max()) < 1e-6
    gx_scale = float(x64.grad.abs().max())
    assert torch.equal(res[True][1], res[False][1])          # same masks, same weights, same kernels behind the first layer
    # against float64: all rows but the few where an fp32 pre-activation within rounding of 0 takes the other ReLU branch
    row_err = (res[True][1].double() - x64.grad).abs().max(1).values
    assert int((row_err > 2e-5 * gx_scale).sum()) <= 3 and float(row_err.max()) < 2e-2 * gx_scale, (int((row_err > 2e-5 * gx_scale).sum()), float(row_err.max()))
    assert len(res[True][2]) == 14
    for k, t in sd.items():
        ref = t.grad
        scale = float(ref.abs().max()) + 1e-30
        e_fold = float((res[True][2][k].double() - ref).abs().max()) / scale
        e_rows = float((res[False][2][k].double() - ref).abs().max()) / scale
        assert e_fold < max(3.0 * e_rows, 1e-4), (k, e_fold, e_rows)


@pytest.mark.parametrize("P,n_live", [(16384 + 77, None), (65536, None), (40000, 17777), (20000, 63), (20000, 0)])
def test_group_backward_equals_the_layer_launches(net, P, n_live):
    """hos_mlp_chain_bwd (three group launches, dZ in LDS between the layers of a group) against the eight hos_linear_bwd_fused
    launches it replaces: d loss / d x and every parameter gradient.  Same arithmetic (bf16 pairs, same product order), so the
    input gradient must agree to rounding of the fp32 sums and the parameter gradients to the slab reduction's order."""
    from hosnerf_amd.human_nerf import _NonRigidFn
    x, cond, band = _inputs(P, seed=13)
    g = torch.randn(P, 3, generator=torch.Generator().manual_seed(6)).to(DEV)
    rows_dev = None if n_live is None else torch.tensor([n_live], dtype=torch.int32, device=DEV)
    n = P if n_live is None else n_live
    if n_live is not None:
        g[n:] = 0
    res = {}
    prev = ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_FOLD, ops.MLP_CHAIN_BWD, ops.MLP_CHAIN_BWD_MIN_ROWS
    try:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_FOLD, ops.MLP_CHAIN_BWD_MIN_ROWS = True, 1, True, 1
        for cb in (True, False):
            ops.MLP_CHAIN_BWD = cb
            net.zero_grad()
            xx = x.clone().requires_grad_(True)
            xyz = _NonRigidFn.apply(torch.zeros((), device=DEV, requires_grad=True), net, "nr", xx, cond, band, rows_dev)
            xyz.backward(g)
            torch.cuda.synchronize()
            res[cb] = (xx.grad[:n].clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters() if k.startswith("non_rigid_mlp.")})
    finally:
        ops.MLP_CHAIN, ops.MLP_CHAIN_MIN_ROWS, ops.MLP_CHAIN_FOLD, ops.MLP_CHAIN_BWD, ops.MLP_CHAIN_BWD_MIN_ROWS = prev
        net.zero_grad()
    gx, gx_ref = res[True][0], res[False][0]
    assert torch.isfinite(gx).all()
    if n > 0:
        assert float((gx - gx_ref).abs().max()) <= 1e-6 * float(gx_ref.abs().max()), float((gx - gx_ref).abs().max())
    assert len(res[True][1]) == 14
    for k, ref in res[False][1].items():
        got = res[True][1][k]
        scale = float(ref.abs().max()) + 1e-30
        assert float((got - ref).abs().max()) <= 2e-6 * scale, (k, float((got - ref).abs().max()), scale)


# ------------------------------------------------------------------------------------------------ canonical MLP, folded state embedding
def _cnl_ref64(net, cnl, state, g=None):
    """float64 autograd of CanonicalMLP on [fourier(cnl) | state embedding] (mlp_rgb_sigma.py:49-58 + N:539-540)."""
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in net.named_parameters()
          if k.startswith("cnl_mlp.") or k == f"human_stateembeds.{state}"}
    x = cnl.double().clone().requires_grad_(True)
    freqs = 2.0 ** torch.arange(10, dtype=torch.float64, device=DEV)
    ang = x[:, None, :] * freqs[None, :, None]
    emb = torch.cat([x, torch.cat([torch.sin(ang), torch.cos(ang)], -1).reshape(x.shape[0], 60),
                     sd[f"human_stateembeds.{state}"][None].expand(x.shape[0], -1)], 1)
    names = sorted({k.rsplit(".", 1)[0] for k in sd if k.startswith("cnl_mlp.pts_linears")}, key=lambda s_: int(s_.split(".")[-1]))
    h = emb
    for i, nm in enumerate(names):
        if i == 5:
            h = torch.cat([emb, h], 1)
        h = torch.relu(h @ sd[nm + ".weight"].T + sd[nm + ".bias"])
    out_name = [k for k in sd if k.startswith("cnl_mlp.") and "pts_linears" not in k and k.endswith(".weight")]
    assert len(out_name) == 1, out_name
    o = h @ sd[out_name[0]].T + sd[out_name[0][:-6] + "bias"]
    raw = torch.cat([torch.sigmoid(o[:, :3]), torch.relu(o[:, 3:])], 1)
    if g is not None:
        raw.backward(g.double())
    return raw.detach(), x.grad, sd


@pytest.mark.parametrize("P", [16384, 20000 + 13, 65536])
@pytest.mark.parametrize("state", [0, 1])
def test_folded_canonical_forward_equals_the_row_form(net, P, state):
    """The state embedding as a per-call bias of the input layer and the skip layer (64 / 320-column rows, the skip layer on the
    20-step thin kernel) against the row form ([fourier | state] rows, 384-wide skip layer on the tiled GEMM) and float64."""
    gen = torch.Generator().manual_seed(P + state)
    cnl = (torch.rand(P, 3, generator=gen) * 2 - 1).to(DEV)
    prev = ops.CNL_FOLD
    try:
        with torch.no_grad():
            ops.CNL_FOLD = True
            raw_f, (E_f, acts_f, bits_f, fold) = net._canonical_fwd(cnl, state, save=True)
            ops.CNL_FOLD = False
            raw_r, (E_r, acts_r, bits_r, none) = net._canonical_fwd(cnl, state, save=True)
    finally:
        ops.CNL_FOLD = prev
    assert fold is not None and none is None and E_f.shape[1] == 64 and E_r.shape[1] == 128 and acts_f[4].shape[1] == 320
    assert torch.equal(E_f[:, :63], E_r[:, :63]) and bool((E_f[:, 63] == 0).all())
    assert torch.equal(acts_f[4][:, :63], acts_r[4][:, :63]) and bool((acts_f[4][:, 63] == 0).all())
    assert bits_f[5] is not None and bits_r[5] is None          # the folded skip layer runs on the thin kernel and writes its mask bits
    ref, _, _ = _cnl_ref64(net, cnl, state)
    for l in range(8):
        a_f = acts_f[l][:, 64:] if l == 4 else acts_f[l]
        a_r = acts_r[l][:, 127:383] if l == 4 else acts_r[l]
        assert float((a_f - a_r).abs().max()) < 3e-6 * max(1.0, float(a_r.abs().max())), l
    rs = max(1.0, float(ref.abs().max()))
    e_f, e_r = float((raw_f.double() - ref).abs().max()), float((raw_r.double() - ref).abs().max())
    assert e_f < 3e-6 * rs and e_f < 2.0 * e_r + 1e-7 * rs, (e_f, e_r)


@pytest.mark.parametrize("state", [0, 1])
def test_folded_canonical_backward_equals_the_row_form(net, state):
    """The backward pass of both forms on the SAME saved activations and ReLU masks (the folded form's rows are cut out of the row
    form's buffers, so no pre-activation within rounding of 0 can take different branches in the two): d loss / d points, every
    cnl_mlp gradient and the state embedding's.  The folded layers get their weight gradient in parts -- Fourier columns from the
    GEMM over the 64-column rows, state columns as bias-gradient (x) embedding, h columns from the aligned window."""
    P = 20000 + 13
    gen = torch.Generator().manual_seed(77 + state)
    cnl = (torch.rand(P, 3, generator=gen) * 2 - 1).to(DEV)
    g = torch.randn(P, 4, generator=gen).to(DEV)
    prev = ops.CNL_FOLD
    res = {}
    try:
        with torch.no_grad():
            ops.CNL_FOLD = False
            raw, (E_r, acts_r, bits_r, _) = net._canonical_fwd(cnl, state, save=True)
            ops.CNL_FOLD = True
            _, (_, _, _, fold) = net._canonical_fwd(cnl, state, save=True)           # the packed weights of the folded layers
            z = torch.zeros(P, 1, device=DEV)
            E_f = torch.cat([E_r[:, :63], z], 1).contiguous()
            acts_f = list(acts_r)
            acts_f[4] = torch.cat([E_r[:, :63], z, acts_r[4][:, 127:383]], 1).contiguous()
            for f, saved in ((False, (E_r, acts_r, bits_r, None)), (True, (E_f, acts_f, bits_r, fold))):
                net.zero_grad()
                g_cnl = net._canonical_bwd(saved, cnl, raw, g, state)
                res[f] = (g_cnl.clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters()
                                          if k.startswith("cnl_mlp.") or k.startswith("human_stateembeds.")})
    finally:
        ops.CNL_FOLD = prev
        net.zero_grad()
    _, gx64, sd = _cnl_ref64(net, cnl, state, g)
    gs = float(gx64.abs().max())
    assert float((res[True][0] - res[False][0]).abs().max()) < 2e-5 * gs
    row_err = (res[True][0].double() - gx64).abs().max(1).values
    n_bad = int((row_err > 5e-5 * gs).sum())                    # rows where fp32 and float64 disagree about a ReLU branch
    assert n_bad <= 16 and float(row_err.max()) < 5e-2 * gs, (n_bad, float(row_err.max()))
    for k in res[True][1]:                                      # the embeddings of the other states receive nothing
        if k.startswith("human_stateembeds.") and k != f"human_stateembeds.{state}":
            assert float(res[True][1][k].abs().max()) == 0.0
    assert len(sd) == 19
    for k, t in sd.items():
        ref = t.grad
        scale = float(ref.abs().max()) + 1e-30
        e_fold = float((res[True][1][k].double() - ref).abs().max()) / scale
        e_rows = float((res[False][1][k].double() - ref).abs().max()) / scale
        d_fr = float((res[True][1][k] - res[False][1][k]).abs().max()) / scale
        assert d_fr < 1e-4 and e_fold < 1.5 * e_rows + 1e-4, (k, e_fold, e_rows, d_fr)


@pytest.mark.parametrize("state", [0, 1])
def test_folded_canonical_autograd_end_to_end(net, state):
    """`_CanonicalFn` through both forms, each on its own forward: the two evaluations round differently, so a sample whose
    pre-activation lies within rounding of 0 may take the other ReLU branch in one of them (one such row moves a gradient by
    ~1 / sqrt(P) of its largest element when the upstream gradients are random) -- the bound is that, on top of the row form's
    own distance from float64; direction cosines stay at 1."""
    from hosnerf_amd.human_nerf import _CanonicalFn
    P = 20000 + 13
    gen = torch.Generator().manual_seed(77 + state)
    cnl = (torch.rand(P, 3, generator=gen) * 2 - 1).to(DEV)
    g = torch.randn(P, 4, generator=gen).to(DEV)
    res = {}
    prev = ops.CNL_FOLD
    try:
        for f in (True, False):
            ops.CNL_FOLD = f
            net.zero_grad()
            xx = cnl.clone().requires_grad_(True)
            raw = _CanonicalFn.apply(torch.zeros((), device=DEV, requires_grad=True), net, xx, state)
            raw.backward(g)
            res[f] = (xx.grad.clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters()
                                        if k.startswith("cnl_mlp.") or k == f"human_stateembeds.{state}"})
    finally:
        ops.CNL_FOLD = prev
        net.zero_grad()
    _, gx64, sd = _cnl_ref64(net, cnl, state, g)
    gs = float(gx64.abs().max())
    for f in (True, False):
        row_err = (res[f][0].double() - gx64).abs().max(1).values
        n_bad = int((row_err > 5e-5 * gs).sum())
        assert n_bad <= 16 and float(row_err.max()) < 5e-2 * gs, (f, n_bad, float(row_err.max()))
    for k, t in sd.items():
        ref = t.grad.reshape(-1)
        scale = float(ref.abs().max()) + 1e-30
        a = res[True][1][k].double().reshape(-1)
        e_fold = float((a - ref).abs().max()) / scale
        e_rows = float((res[False][1][k].double().reshape(-1) - ref).abs().max()) / scale
        cos = float((a @ ref) / (a.norm() * ref.norm() + 1e-30))
        assert cos > 0.99999 and e_fold < 3.0 * e_rows + 3.0 / P ** 0.5, (k, e_fold, e_rows, cos)
