"""GPU parity tests (run with -m gpu on MI355X): HIP kernels vs the CPU oracle and the golden vectors.

Everything goes through the C ABI (hosnerf_amd._lib -> libhosrender.so).  Tolerances are the
north-star ones: <= 1e-4 RGB L-inf against the reference's golden outputs, bit-exact indices.
"""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle.background as ob
from hosnerf_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from hosnerf_amd import _lib
    _lib.load()
    return torch.device("cuda")


@pytest.fixture(scope="module")
def hv():
    return {k: v for k, v in np.load(os.path.join(G, "bkgd_helpers.npz")).items()}


@pytest.fixture(scope="module")
def fw():
    return {k: v for k, v in np.load(os.path.join(G, "bkgd_forward.npz")).items()}


def T(x, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    return t.to(dev) if dev is not None else t


def maxerr(a, b):
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    return float((a.detach().double().cpu() - torch.as_tensor(np.asarray(b)).double()).abs().max())


# ------------------------------------------------------------------------------------------ GEMM
@pytest.fixture(params=["fp32", "bf16x3"])
def gemm_mode(request, dev):
    """Run the GEMM unit tests in both arithmetic modes; bf16x3 carries ~2^-17 relative error per product."""
    from hosnerf_amd import ops
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_FP32 if request.param == "fp32" else ops.GEMM_BF16X3)
    yield 1.0 if request.param == "fp32" else 10.0
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize("M,N,K0,K1,epi", [
    (256, 256, 576, 0, 1), (384, 1024, 1024, 576, 1), (200, 128, 288, 0, 1), (128, 257, 1024, 0, 4),
    (96, 1, 256, 0, 2), (160, 3, 128, 0, 3), (131, 256, 256, 0, 0), (64, 4, 256, 0, 5)])
def test_linear_fwd(dev, gemm_mode, M, N, K0, K1, epi):
    tol = 2e-5 * gemm_mode
    from hosnerf_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A0 = torch.randn(M, K0, generator=g)
    A1 = torch.randn(M, K1, generator=g) if K1 else None
    Wt = torch.randn(N, K0 + K1, generator=g) / np.sqrt(K0 + K1)
    b = torch.randn(N, generator=g)
    A = A0 if A1 is None else torch.cat([A0, A1], -1)
    ref = A.double() @ Wt.double().T + b.double()
    out = torch.full((M, max(N, 4) + 3), -7.0, device=dev)      # odd ldc, sentinel to catch stray writes
    out = torch.full((M, ((N + 3) // 4) * 4 + 4), -7.0, device=dev)
    aux = torch.full((M,), -7.0, device=dev)
    ops.linear_fwd(A0.to(dev), K0, Wt.to(dev), b.to(dev), N, out, epi, A1=None if A1 is None else A1.to(dev), K1=K1,
                   aux=aux, aux_col=256, p0={2: -1.0, 3: 0.001, 4: -1.0}.get(epi, 0.0))
    torch.cuda.synchronize()
    o = out.cpu().double()
    if epi == 1:
        ref = ref.clamp(min=0)
    if epi == 3:
        ref = torch.sigmoid(ref) * 1.002 - 0.001
    if epi == 5:
        ref = torch.cat([torch.sigmoid(ref[:, :3]), ref[:, 3:].clamp(min=0)], -1)
    if epi == 2:
        assert maxerr(aux, torch.nn.functional.softplus(ref[:, 0] - 1)) < tol
        assert torch.all(o == -7.0)
        return
    if epi == 4:
        assert maxerr(aux, torch.nn.functional.softplus(ref[:, 256] - 1)) < tol
        assert maxerr(out[:, :256], ref[:, :256]) < tol
        assert torch.all(o[:, 256:] == -7.0)
        return
    assert maxerr(out[:, :N], ref) < tol
    assert torch.all(o[:, N:] == -7.0)


@pytest.mark.parametrize("M,N,K,mask", [(256, 256, 256, True), (200, 1024, 1024, True), (128, 288, 1024, True),
                                         (96, 32, 128, True), (192, 128, 256, False)])
def test_linear_dgrad(dev, gemm_mode, M, N, K, mask):
    tol = 2e-5 * gemm_mode
    from hosnerf_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g)
    Wt = torch.randn(N, K + 32, generator=g) / np.sqrt(N)      # ldw > K: column sub-range
    X = torch.randn(M, K, generator=g)
    ref = dY.double() @ Wt[:, 8:8 + K].double()
    if mask:
        ref = ref * (X > 0)
    out = torch.full((M, K + 4), -7.0, device=dev)
    ops.linear_dgrad(dY.to(dev), Wt.to(dev), N, K, out, mask_src=X.to(dev) if mask else None, w_col0=8)
    assert maxerr(out[:, :K], ref) < tol
    assert torch.all(out[:, K:] == -7.0)
    ops.linear_dgrad(dY.to(dev), Wt.to(dev), N, K, out, mask_src=X.to(dev) if mask else None, w_col0=8, accumulate=True)
    assert maxerr(out[:, :K], 2 * ref) < 2 * tol


@pytest.mark.parametrize("M,N,K", [(4096, 256, 576), (2048, 1024, 1024), (1024, 257, 1024), (8192, 1, 256),
                                    (2048, 3, 128), (512, 128, 288), (96, 256, 256), (1000, 128, 128), (77, 256, 384)])
def test_linear_wgrad(dev, gemm_mode, M, N, K):
    from hosnerf_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    Npad = (N + 31) // 32 * 32
    dY = torch.zeros(M, Npad)
    dY[:, :N] = torch.randn(M, N, generator=g)
    X = torch.randn(M, K, generator=g)
    ref = dY[:, :N].double().T @ X.double()
    refb = dY[:, :N].double().sum(0)
    dW = torch.zeros(Npad, K + 64, device=dev)
    db = torch.zeros(Npad, device=dev)
    ops.linear_wgrad(dY.to(dev), X.to(dev), dW, db, N, K, w_col0=32)
    scale = np.sqrt(M) * gemm_mode
    assert maxerr(dW[:N, 32:32 + K], ref) < 3e-6 * scale * 4
    assert maxerr(db[:N], refb) < 3e-6 * scale * 4
    assert float(dW[:, :32].abs().max()) == 0 and float(dW[:, 32 + K:].abs().max()) == 0
    if Npad > N:
        assert float(dW[N:].abs().max()) == 0 and float(db[N:].abs().max()) == 0
    ops.linear_wgrad(dY.to(dev), X.to(dev), dW, db, N, K, w_col0=32)    # accumulates
    assert maxerr(dW[:N, 32:32 + K], 2 * ref) < 6e-6 * scale * 4


# ------------------------------------------------------------------------------------------ per-ray kernels
@pytest.mark.parametrize("S", [64, 32])
def test_resample_golden(dev, hv, S):
    from hosnerf_amd import ops
    # feed the *pre-dilation* histogram: the kernel fuses max_dilate + trim + logits + sampling
    t, w = T(hv["dil_l1_t"], dev), T(hv["dil_l1_w"], dev)
    dil = float(hv["dil_l1_dilation"])
    sd, td, idx = ops.resample(t, w, S, dil, 0.7, False, 0.1, 1e6, want_index=True)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), hv[f"rs_binidx_eval_S{S}"]), "sample index must be bit-exact"
    # The inverse CDF is ill-conditioned where the pdf is ~0 (a 1e-7 CDF perturbation moves the sample
    # across the whole empty bin), so positions are compared through the CDF: F(got) == F(want).
    cdf_t, cdf_w = T(hv["rs_t"]).double(), torch.softmax(T(hv["rs_logits"]).double(), -1)
    cdf = torch.cat([torch.zeros(cdf_w.shape[0], 1, dtype=torch.float64), torch.cumsum(cdf_w, -1)], -1)

    def F(x):
        x = x.double().cpu()
        i = (torch.searchsorted(cdf_t.contiguous(), x.contiguous(), right=True) - 1).clamp(0, cdf_w.shape[1] - 1)
        t0, t1 = torch.gather(cdf_t, -1, i), torch.gather(cdf_t, -1, i + 1)
        frac = ((x - t0) / (t1 - t0).clamp(min=1e-30)).clamp(0, 1)
        return torch.gather(cdf, -1, i) + frac * torch.gather(cdf_w, -1, i)

    # CDF knots are 190 sequential fp32 adds (reference: torch.cumsum) -> ~1e-5 absolute noise in CDF space
    assert float((F(sd) - F(T(hv[f"rs_eval_S{S}"]))).abs().max()) < 2e-5
    assert float(np.median(np.abs(sd.cpu().numpy() - hv[f"rs_eval_S{S}"]))) < 1e-7
    sd2, _ = ops.resample(t, w, S, dil, 0.7, True, 0.1, 1e6, jitter=T(hv[f"rs_jitter_S{S}"], dev).reshape(-1))
    assert float((F(sd2) - F(T(hv[f"rs_train_S{S}"]))).abs().max()) < 2e-5
    assert maxerr(1.0 / td, 1.0 / ob.s_to_t(sd.cpu(), 0.1, 1e6)) < 1e-5


def test_resample_level0_and_dilate(dev, hv):
    from hosnerf_amd import ops
    B = 8
    sd, td = ops.resample(torch.tensor([[0.0, 1.0]], device=dev).repeat(B, 1), torch.ones(B, 1, device=dev), 64, 0.5025, 1.0, False, 0.1, 1e6)
    assert maxerr(sd, hv["rs0_eval_S64"]) < 1e-7
    # max_dilate alone, through the oracle-equivalent path: compare sdist from resampling both dilations
    for tag in ("l1", "l2"):
        t, w = T(hv[f"dil_{tag}_t"]), T(hv[f"dil_{tag}_w"])
        dil = float(hv[f"dil_{tag}_dilation"])
        td_o, wd_o = ob.max_dilate_weights(t, w, dil, (0.0, 1.0))
        lg = ob.resample_logits(td_o[..., 1:-1], wd_o[..., 1:-1], 1.0)
        want, widx = ob.sample_intervals(False, td_o[..., 1:-1], lg, 32, (0.0, 1.0), return_index=True)
        got, _, gidx = ops.resample(t.to(dev), w.to(dev), 32, dil, 1.0, False, 0.1, 1e6, want_index=True)
        assert np.array_equal(gidx.cpu().numpy(), widx.numpy().astype(np.int32))
        assert float(np.median(np.abs(got.cpu().numpy() - want.numpy()))) < 1e-7
        assert maxerr(got, want) < 5e-3    # empty-bin samples are ill-conditioned (see test_resample_golden)
    # well-conditioned histogram (no near-empty bins): positions agree to fp32 rounding
    g = torch.Generator().manual_seed(9)
    t = torch.sort(torch.rand(16, 65, generator=g), -1).values
    t[:, 0], t[:, -1] = 0.0, 1.0
    w = torch.rand(16, 64, generator=g) + 0.5
    w = w / w.sum(-1, keepdim=True)
    td_o, wd_o = ob.max_dilate_weights(t, w, 0.0103125, (0.0, 1.0))
    lg = ob.resample_logits(td_o[..., 1:-1], wd_o[..., 1:-1], 1.0)
    for S, rnd in ((64, False), (32, True)):
        jit = torch.rand(16, 1, generator=g)
        want, widx = ob.sample_intervals(rnd, td_o[..., 1:-1], lg, S, (0.0, 1.0), jitter=jit, return_index=True)
        got, _, gidx = ops.resample(t.to(dev), w.to(dev), S, 0.0103125, 1.0, rnd, 0.1, 1e6, jitter=jit.to(dev).reshape(-1), want_index=True)
        assert np.array_equal(gidx.cpu().numpy(), widx.numpy().astype(np.int32))
        assert maxerr(got, want) < 1e-5


def test_encode_ipe(dev, hv):
    from hosnerf_amd import ops
    tdist, o, d, radii = (T(hv[k], dev) for k in ("cast_tdist", "cast_o", "cast_d", "cast_radii"))
    basis = T(hv["basis"], dev)
    embed = torch.arange(64, dtype=torch.float32, device=dev)
    X = ops.encode_ipe(tdist, o, d, radii, basis, embed, 576)
    B, S = tdist.shape[0], tdist.shape[1] - 1
    X = X.view(B, S, 576)
    # feature at level l is sin(2^l * mean): an fp32 rounding of the contracted mean (|z| < 2, ulp 2.4e-7,
    # a handful of roundings in cast + contract) is amplified by 2^l, so the tolerance scales with the level
    want = ob.encode_samples(T(hv["cast_means"]), T(hv["cast_covs"]), T(hv["basis"]))
    for half in (0, 252):
        for lvl in range(12):
            sl = slice(half + lvl * 21, half + (lvl + 1) * 21)
            tol = 2e-6 + (2.0 ** lvl) * 1.5e-6
            assert maxerr(X[..., sl], hv["ipe"][..., sl]) < tol, (lvl, "golden")
            assert maxerr(X[..., sl], want[..., sl]) < tol, (lvl, "oracle")
    assert torch.equal(X[..., 504:568].cpu(), embed.cpu().expand(B, S, 64))
    assert float(X[..., 568:].abs().max()) == 0
    # planes variant: the same fp32 values, split into fp16 / bf16 hi-lo pairs by the encoder itself
    p16, pb = ops.encode_ipe_planes(tdist, o, d, radii, basis, embed, 576)
    q16, qb = ops.split_planes2(X.view(B * S, 576), C=576, ld=576)
    assert torch.equal(p16.t, q16.t) and torch.equal(pb.t, qb.t)
    Xv = torch.full((B * S, 288), -7.0, device=dev)
    ops.encode_viewdirs(d, S, Xv, 256)
    assert maxerr(Xv.view(B, S, 288)[:, 0, 256:283], hv["dir_enc"]) < 2e-6
    assert torch.all(Xv[:, :256] == -7.0) and float(Xv[:, 283:].abs().max()) == 0


def test_encode_ipe_beyond_the_own_sine_range(dev, hv):
    """Round 5: the encoder's own sine covers |2^l x lifted mean| <= 2^15 (the model: <= 4100); a workgroup whose lifted means leave that
    range takes the library's sinf for all its samples.  A caller's free basis can get there: the same rays with the basis scaled by 40
    (|2^11 x mean . b| up to ~1.6e5) against the oracle on the same scaled basis, tolerance scaled with the argument (an fp32 rounding
    of the lifted mean is amplified by 2^l x 40), finite everywhere, embedding columns untouched -- and the in-range basis still takes
    the fast path bit for bit next to it (two launches, same inputs, same outputs)."""
    from hosnerf_amd import ops
    tdist, o, d, radii = (T(hv[k], dev) for k in ("cast_tdist", "cast_o", "cast_d", "cast_radii"))
    embed = torch.arange(64, dtype=torch.float32, device=dev)
    B, S = tdist.shape[0], tdist.shape[1] - 1
    scale = 40.0
    basis = T(hv["basis"]) * scale
    X = ops.encode_ipe(tdist, o, d, radii, basis.to(dev), embed, 576).view(B, S, 576)
    assert torch.isfinite(X).all()
    want = ob.encode_samples(T(hv["cast_means"]), T(hv["cast_covs"]), basis)
    for half in (0, 252):
        for lvl in range(12):
            sl = slice(half + lvl * 21, half + (lvl + 1) * 21)
            tol = 2e-6 + (2.0 ** lvl) * 1.5e-6 * scale
            assert maxerr(X[..., sl], want[..., sl]) < tol, lvl
    assert float(maxerr(X[..., :63], want[..., :63])) < 4e-4                     # levels 0-2: tight enough to tell a wrong quadrant (errors of order 1)
    assert torch.equal(X[..., 504:568].cpu(), embed.cpu().expand(B, S, 64))
    a = ops.encode_ipe(tdist, o, d, radii, T(hv["basis"], dev), embed, 576)
    b = ops.encode_ipe(tdist, o, d, radii, T(hv["basis"], dev), embed, 576)
    assert torch.equal(a, b)


@pytest.mark.parametrize("tag,opq", [("opq", True), ("nopq", False)])
def test_alpha_weights_volrender(dev, hv, tag, opq):
    from hosnerf_amd import ops
    dens = T(hv["aw_density"], dev).requires_grad_(True)
    td, dirs, rgbs = T(hv["aw_tdist"], dev), T(hv["aw_dirs"], dev), T(hv["vr_rgbs"], dev).requires_grad_(True)
    w = ops.alpha_weights(dens, td, dirs, opq)
    assert maxerr(w, hv[f"aw_{tag}_w"]) < 2e-6
    rgb = ops.volumetric_rendering(rgbs, w, 1.0)
    assert maxerr(rgb, hv[f"vr_{tag}_rgb"]) < 2e-6
    gout = torch.randn(rgb.shape, generator=torch.Generator().manual_seed(3))
    gw_extra = torch.randn(w.shape, generator=torch.Generator().manual_seed(4))
    ((rgb * gout.to(dev)).sum() + (w * gw_extra.to(dev)).sum()).backward()
    # oracle autograd
    d2 = T(hv["aw_density"]).requires_grad_(True)
    r2 = T(hv["vr_rgbs"]).requires_grad_(True)
    w2 = ob.compute_alpha_weights(d2, T(hv["aw_tdist"]), T(hv["aw_dirs"]), opq)[0]
    rgb2 = ob.volumetric_rendering(r2, w2, 1.0)
    ((rgb2 * gout).sum() + (w2 * gw_extra).sum()).backward()
    assert maxerr(rgbs.grad, r2.grad) < 1e-5
    scale = float(d2.grad.abs().max())
    assert maxerr(dens.grad, d2.grad) < 2e-5 * max(scale, 1.0)


def test_alpha_weights_128_samples(dev):
    """S=128 (human branch sample count): two samples per lane."""
    from hosnerf_amd import ops
    g = torch.Generator().manual_seed(0)
    B, S = 6, 128
    dens = torch.rand(B, S, generator=g) * 3
    td = torch.sort(torch.rand(B, S + 1, generator=g) * 4 + 0.5, -1).values
    dirs = torch.randn(B, 3, generator=g)
    d1 = dens.to(dev).requires_grad_(True)
    w = ops.alpha_weights(d1, td.to(dev), dirs.to(dev), True)
    d2 = dens.clone().requires_grad_(True)
    w2 = ob.compute_alpha_weights(d2, td, dirs, True)[0]
    assert maxerr(w, w2) < 2e-6
    gw = torch.randn(B, S, generator=g)
    (w * gw.to(dev)).sum().backward()
    (w2 * gw).sum().backward()
    assert maxerr(d1.grad, d2.grad) < 1e-5


def test_losses(dev, hv):
    from hosnerf_amd import ops
    c, w, cp, wp = (T(hv[k], dev) for k in ("lo_c", "lo_w", "lo_cp", "lo_wp"))
    lo, hi = ops.interlevel_indices(c, w, cp, wp)
    assert np.array_equal(lo.cpu().numpy().astype(np.int64), hv["lo_idx_lo"])
    assert np.array_equal(hi.cpu().numpy().astype(np.int64), hv["lo_idx_hi"])
    wp_g = wp.clone().requires_grad_(True)
    per_ray = ops.interlevel_loss_per_ray(c, w, cp, wp_g)
    assert maxerr(per_ray, hv["lo_loss"].sum(-1)) < 1e-6
    coef = torch.arange(1, per_ray.numel() + 1, dtype=torch.float32)
    (per_ray * coef.to(dev)).sum().backward()
    wp2 = T(hv["lo_wp"]).requires_grad_(True)
    (ob.lossfun_outer(T(hv["lo_c"]), T(hv["lo_w"]), T(hv["lo_cp"]), wp2).sum(-1) * coef).sum().backward()
    # d/dwp ~ (w - w_outer)/(w + eps): fp32 cancellation divided by tiny w -> compare relative to the ray's scale
    assert maxerr(wp_g.grad, wp2.grad) < 1e-3 * float(wp2.grad.abs().max())
    w_g = w.clone().requires_grad_(True)
    d = ops.distortion_loss_per_ray(c, w_g)
    assert maxerr(d, hv["dist_loss"]) < 1e-6
    (d * coef.to(dev)).sum().backward()
    w2 = T(hv["lo_w"]).requires_grad_(True)
    (ob.lossfun_distortion(T(hv["lo_c"]), w2) * coef).sum().backward()
    assert maxerr(w_g.grad, w2.grad) < 1e-5


# ------------------------------------------------------------------------------------------ end to end
def _basedir():
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


def _batch(B, seed, time):
    b = synth.stage1_batch(B, seed=seed, time=time)
    b["rays_d"][B // 2:] *= 1.7
    return b


@pytest.fixture(scope="module")
def model(dev):
    from hosnerf_amd.mipnerf360 import MipNeRF360
    m = MipNeRF360(_basedir(), opaque_background=True)
    m.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    return m.to(dev)


@pytest.fixture(params=["split", "planes"])
def mlp_mode(request, dev):
    """Model-level tests run with fp32 activations (on-the-fly split GEMMs) and with the NeRF trunk on pre-split
    16-bit planes (hos_linearp_*); both must meet the same tolerances."""
    from hosnerf_amd import ops
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(ops.GEMM_BF16X3 if request.param == "split" else ops.GEMM_PLANES)
    yield request.param
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize("case", ["s1_evalA", "s1_evalB", "s1_trainA"])
def test_forward_vs_golden(dev, model, fw, case, mlp_mode):
    time, frac = float(fw[case + "_time"]), float(fw[case + "_train_frac"])
    randomized = "train" in case
    jit = [T(fw[f"{case}_jitter{l}"], dev).reshape(-1) for l in range(3)] if randomized else None
    batch = {k: v.to(dev) for k, v in _batch(8, 11, time).items()}
    with torch.no_grad():
        rend, hist = model(batch, frac, randomized, randomized, 0.1, 1e6, jitters=jit)
    for l in range(3):
        assert maxerr(hist[l]["sdist"], fw[f"{case}_sdist{l}"]) < 5e-5
        assert maxerr(hist[l]["weights"], fw[f"{case}_weights{l}"]) < 1e-4
        assert maxerr(rend[l]["rgb"], fw[f"{case}_render{l}"]) < 1e-4, "north-star: 1e-4 RGB L-inf vs the reference"
    # per-sample colours are not composited yet (samples with ~zero weight carry fp32 noise of the 2^11-frequency features):
    # bounded by the distance of the reference's OWN fp32 values from the fp64 value of its graph
    sd64 = {k: v.double() for k, v in synth.background_state_dict(777, 2).items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in _batch(8, 11, time).items()}
    with torch.no_grad():
        _, hist64 = ob.mipnerf360_forward(sd64, b64, frac, randomized, 0.1, 1e6, transitions_times=[0.4],
                                          jitters=[T(fw[f"{case}_jitter{l}"]).double() for l in range(3)] if randomized else None)
    e_ref = maxerr(T(fw[case + "_rgb2"]).double(), hist64[2]["rgb"])
    e_hip = maxerr(hist[2]["rgb"].double().cpu(), hist64[2]["rgb"])
    assert e_hip <= 2.0 * e_ref + 2e-5, (e_hip, e_ref)


def test_forward_vs_oracle_indices(dev, model, mlp_mode):
    """Larger batch vs the oracle: RGB within 1e-4, inverse-CDF bin indices bit-exact."""
    B = 64
    cpu_batch = _batch(B, 21, 0.5)
    sd = synth.background_state_dict(777, 2)
    rend_o, hist_o = ob.mipnerf360_forward(sd, cpu_batch, 0.6, False, 0.1, 1e6, transitions_times=[0.4])
    with torch.no_grad():
        rend, hist = model({k: v.to(dev) for k, v in cpu_batch.items()}, 0.6, False, False, 0.1, 1e6, want_index=True)
    assert maxerr(rend[-1]["rgb"], rend_o[-1]["rgb"]) < 1e-4
    mism = 0
    for l in range(3):
        mism += int((hist[l]["bin_idx"].cpu().long() != hist_o[l]["bin_idx"]).sum())
    # level 0 is input-independent and must match exactly; deeper levels see fp32-rounded MLP outputs,
    # so an index may flip only where u sits within rounding of a CDF knot
    from tests._record import record
    record(f"bkgd.bin_idx_flips_vs_oracle[64 rays x 160 samples, mode {mlp_mode}]", {"flips": mism, "of": 64 * 160,
           "rgb_linf": maxerr(rend[-1]["rgb"], rend_o[-1]["rgb"])})
    assert int((hist[0]["bin_idx"].cpu().long() != hist_o[0]["bin_idx"]).sum()) == 0
    assert mism <= 2, f"{mism} bin indices differ"


def test_gradients_vs_oracle(dev, model, fw, mlp_mode):
    from hosnerf_amd.train import stage1_loss
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.background_state_dict(777, 2).items()}
    b = _batch(4, 12, 0.5)
    jit = [T(fw[f"grad_jitter{l}"]) for l in range(3)]
    frac = float(fw["grad_train_frac"])
    rend_o, hist_o = ob.mipnerf360_forward(sd, b, frac, True, 0.1, 1e6, transitions_times=[0.4], jitters=jit)
    loss_o, _ = ob.stage1_loss(rend_o[-1]["rgb"], b["target"], hist_o)
    loss_o.backward()

    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in synth.background_state_dict(777, 2).items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in b.items()}
    rend64, hist64 = ob.mipnerf360_forward(sd64, b64, frac, True, 0.1, 1e6, transitions_times=[0.4], jitters=[j.double() for j in jit])
    ob.stage1_loss(rend64[-1]["rgb"], b64["target"], hist64)[0].backward()

    model.zero_grad()
    rend, hist = model({k: v.to(dev) for k, v in b.items()}, frac, True, True, 0.1, 1e6, jitters=[j.to(dev).reshape(-1) for j in jit])
    loss, parts = stage1_loss(rend[-1]["rgb"], b["target"].to(dev), hist)
    loss.backward()
    assert abs(float(loss.detach()) - float(fw["grad_loss"])) < 2e-5
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 2e-5
    names = [str(n) for n in fw["grad_names"]]
    params = dict(model.named_parameters())
    for n, ref_norm in zip(names, fw["grad_norms"]):
        g = params[n].grad
        got = float(g.double().norm())
        assert abs(got - ref_norm) <= 5e-3 * ref_norm + 1e-8, (n, got, ref_norm)       # vs the reference's own autograd
        go = sd[n].grad
        if go is not None and float(go.abs().max()) > 0:
            # elementwise comparison is fragile with 128 samples (one ReLU flipping under the fp32 noise of the
            # 2^11-frequency features changes a whole row), so: cosine similarity + a loose elementwise bound
            # ... anchored on the fp64 value of the same graph: as close to it as the fp32 oracle's own gradient is
            a, bo, t = g.detach().double().cpu().reshape(-1), go.double().reshape(-1), sd64[n].grad.reshape(-1)
            cos = float((a @ t) / (a.norm() * t.norm() + 1e-30))
            assert cos > 0.9995, (n, cos)
            e_hip, e_ref = float((a - t).abs().max()), float((bo - t).abs().max())
            assert e_hip <= 2.0 * e_ref + 2e-4 * float(t.abs().max()), (n, e_hip, e_ref)
    # padded regions of the flat gradient stay exactly zero
    L = model.mlps[2]._views
    assert float(L.W.view(model.flat_grad)[:, 283:].abs().max()) == 0


def test_fused_adam_matches_torch(dev):
    from hosnerf_amd import ops
    n = 100003
    g0 = torch.Generator().manual_seed(5)
    p = torch.randn(n, generator=g0)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2)
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ss = torch.zeros(1, device=dev)
    for step in range(1, 4):
        g = torch.randn(n, generator=g0) * 0.1
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 0.5)
        opt.step()
        ss.zero_()
        gd = g.to(dev)
        ops.sumsq(gd, ss)
        ops.adam_step(pd, gd, m, v, 1e-2, 0.9, 0.999, 1e-8, step, 1.0, ss, 0.5)
    assert maxerr(pd, ref.detach()) < 2e-6


def test_train_step_reduces_loss(dev, mlp_mode):
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import FusedAdam, train_step_stage1
    torch.manual_seed(0)
    m = MipNeRF360(_basedir(), opaque_background=True)
    m.load_state_dict(synth.background_state_dict(3, 2), strict=False)
    m = m.to(dev)
    from hosnerf_amd.train import stage1_loss
    opt = FusedAdam(m, lr=3e-5, max_grad_norm=0.0)
    batch = {k: v.to(dev) for k, v in synth.stage1_batch(256, seed=1).items()}
    batch["target"] = torch.full_like(batch["target"], 0.25)
    losses = []
    for _ in range(12):      # deterministic (eval) sampling so the loss sequence is comparable step to step
        opt.zero_grad()
        rend, hist = m(batch, 0.5, False, True, 0.1, 1e6)
        loss, _ = stage1_loss(rend[-1]["rgb"], batch["target"], hist)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all()
    assert min(losses[-3:]) < losses[0] - 1e-4, losses
    # and the randomized training entry point runs
    l2, _ = train_step_stage1(m, opt, batch, 0.5, 0.1, 1e6)
    assert np.isfinite(float(l2))


def test_no_cpu_fallback(dev):
    from hosnerf_amd import _lib, ops
    with pytest.raises(_lib.HosLibraryError):
        ops.alpha_weights(torch.ones(2, 4), torch.ones(2, 5), torch.ones(2, 3), True)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1024, 512, 256), (2048, 257, 1024), (512, 128, 576), (4096, 1024, 64), (448, 257, 576),
                                   (96, 33, 32)])
def test_planes_gemm_matches_fp64(M, N, K):
    """hos_split_planes* + hos_linearp_{fwd,dgrad,wgrad}: pre-split 16-bit hi/lo planes, LDS-DMA staging, transpose
    reads for the weight gradient.  fp32-grade accuracy against fp64, incl. ragged N and both tile widths."""
    from hosnerf_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    dY = torch.randn(M, N, device=dev, generator=g)
    X16, Xb = ops.split_planes2(X)
    W16, _ = ops.split_planes(W, dtype=torch.float16)
    _, WTb = ops.split_planes(W, dtype=torch.bfloat16, transposed=True, row_major=False)
    _, dZ = ops.split_planes2(dY, want16=False)
    Npad = dZ.ld
    assert WTb.ld == Npad
    Y = ops.Planes.empty(M, Npad, torch.float16, dev, N)
    Yb = ops.Planes.empty(M, Npad, torch.bfloat16, dev, N)
    ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb)
    ref = torch.relu(X.double() @ W.double().T + b.double())
    assert (Y.float().double() - ref).abs().max().item() < 2e-5
    assert (Yb.float().double() - ref).abs().max().item() < 2e-4           # bf16 hi/lo planes: 2^-17 relative
    if Npad > N:                                                            # padding columns zeroed
        assert Y.hi[:, N:].abs().max().item() == 0 and Y.lo[:, N:].abs().max().item() == 0
    C = torch.empty(M, N, device=dev)
    ops.linearp_fwd(X16, K, W16, b, M, N, True, None, None, C=C, epilogue=ops.EPI_RELU)
    assert (C.double() - ref).abs().max().item() < 2e-5
    dX = ops.Planes.empty(M, X16.ld, torch.bfloat16, dev, K)
    ops.linearp_dgrad(dZ, WTb, Npad, M, K, mask=X16, dX=dX)
    refd = (dY.double() @ W.double()) * (X > 0)
    assert (dX.float().double() - refd).abs().max().item() < 2e-4 * refd.abs().max().item() + 1e-5
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    refw = dY.double().T @ X.double()
    for use_ws in (True, False):             # slab reduction through the workspace / fp32 atomics
        dW.zero_(); db.zero_()
        ops.linearp_wgrad(dZ, Xb, dW, db, M, N, K, use_ws=use_ws)
        assert (dW.double() - refw).abs().max().item() < 2e-4 * refw.abs().max().item()
        assert (db.double() - dY.double().sum(0)).abs().max().item() < 2e-3
    ops.linearp_wgrad(dZ, Xb, dW, db, M, N, K)                                  # accumulates
    assert (dW.double() - 2 * refw).abs().max().item() < 4e-4 * refw.abs().max().item()


@pytest.mark.gpu
def test_lit_module_with_torch_adam(dev):
    """select_model('state_mipnerf360'): the Lightning-style surface -- training_step returns the loss, a plain
    torch.optim.Adam from configure_optimizers() steps the parameters (their grads are views of the flat buffer)."""
    from hosnerf_amd.select_option import select_model
    lit = select_model("state_mipnerf360", _basedir())
    lit.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    lit = lit.to(dev)
    opt = lit.configure_optimizers()
    batch = {k: v.to(dev) for k, v in synth.stage1_batch(256, seed=1).items()}
    batch["target"] = torch.full_like(batch["target"], 0.25)
    losses = []
    for i in range(12):
        opt.zero_grad(set_to_none=False)
        loss = lit.training_step(batch, i)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(lit.parameters(), 0.001)
        for g in opt.param_groups:
            g["lr"] = 5e-4
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and min(losses[-3:]) < losses[0] - 1e-4, losses
    assert abs(lit.learning_rate(0) - 2e-5) < 1e-9 and abs(lit.learning_rate(512) - 2e-3) / 2e-3 < 0.02
