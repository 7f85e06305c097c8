"""Generate golden vectors by RUNNING THE REFERENCE (stub-imported from /root/reference).

Run in the build container only:   python -m tests.golden.make_golden [bkgd|human|all]
Outputs small .npz files next to this script; they are committed and are what pins oracle/
(tests/test_oracle_golden.py) and, through the oracle, the HIP kernels on the GPU box.

Inputs are seeded numpy; weights come from hosnerf_amd.synth (seeded numpy), loaded into the
reference modules with load_state_dict, so fixtures hold inputs + expected outputs only.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import refload  # noqa: E402
from hosnerf_amd import synth  # noqa: E402

warnings.filterwarnings("ignore")


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: _np(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KB, {len(arrays)} arrays")


# ------------------------------------------------------------------------------ background


def _hist(rs, B, n, zero_bins=True):
    """random sorted edges in [0,1] and non-negative weights; a few zero-width bins."""
    t = np.sort(rs.uniform(0, 1, size=(B, n + 1)).astype(np.float32), axis=-1)
    t[:, 0], t[:, -1] = 0.0, 1.0
    if zero_bins and n > 8:
        t[:, 5] = t[:, 4]
        t[1, 9] = t[1, 8] = t[1, 7]
    w = rs.uniform(0, 1, size=(B, n)).astype(np.float32) ** 3
    w /= w.sum(-1, keepdims=True)
    return torch.from_numpy(t), torch.from_numpy(w)


def golden_bkgd_helpers():
    H = refload.helper(3)
    rs = np.random.RandomState(20230929)
    out = {}
    B = 8
    # B2
    s = torch.from_numpy(np.sort(rs.uniform(0, 1, (B, 33)).astype(np.float32), -1))
    out["s2t_s"] = s
    out["s2t_t"] = H.construct_ray_warps(0.1, 1e6)[1](s)
    # B3 (both dilations used by the 3-level model)
    t, w = _hist(rs, B, 64)
    for tag, dil in (("l1", 0.0025 + 0.5 / 64), ("l2", 0.0025 + 0.5 / 4096)):
        td, wd = H.max_dilate_weights(t, w.clone(), dil, (0.0, 1.0), True)
        out[f"dil_{tag}_t"], out[f"dil_{tag}_w"], out[f"dil_{tag}_dilation"] = t, w, np.float64(dil)
        out[f"dil_{tag}_td"], out[f"dil_{tag}_wd"] = td, wd
    # B4: resample from the dilated histogram of level 1 (191 edges) -- eval and train (jitter)
    td, wd = H.max_dilate_weights(t, w.clone(), 0.0025 + 0.5 / 64, (0.0, 1.0), True)
    td, wd = td[..., 1:-1], wd[..., 1:-1]
    logits = torch.where(td[..., 1:] > td[..., :-1], 0.7 * torch.log(wd), torch.full_like(wd, -torch.inf))
    out["rs_t"], out["rs_logits"] = td, logits
    for S in (64, 32):
        out[f"rs_eval_S{S}"] = H.sample_intervals(False, td, logits, S, single_jitter=True, domain=(0.0, 1.0))
        torch.manual_seed(1234 + S)
        jit = torch.rand(B, 1)
        torch.manual_seed(1234 + S)
        out[f"rs_train_S{S}"] = H.sample_intervals(True, td, logits, S, single_jitter=True, domain=(0.0, 1.0))
        out[f"rs_jitter_S{S}"] = jit
        # bin index implied by the reference's own mask semantics (H:109-114 uses the same x >= xp test)
        u_eval = torch.linspace(1 / (2 * S), 1 - 1 / (2 * S) - H.eps, S).expand(B, S).contiguous()
        cw = H.integrate_weights(torch.softmax(logits, -1))
        out[f"rs_binidx_eval_S{S}"] = H.searchsorted(cw, u_eval)[0]
    # level 0: t=[0,1], logits=[0]
    t01 = torch.tensor([[0.0, 1.0]]).repeat(B, 1)
    out["rs0_eval_S64"] = H.sample_intervals(False, t01, torch.zeros(B, 1), 64, single_jitter=True, domain=(0.0, 1.0))
    # B5
    Bc, Sc = 4, 8
    tdist = torch.from_numpy(np.sort(np.exp(rs.uniform(np.log(0.1), np.log(50.0), (Bc, Sc + 1))).astype(np.float32), -1))
    o = torch.from_numpy((rs.standard_normal((Bc, 3)) * 0.1).astype(np.float32))
    d = torch.from_numpy(rs.standard_normal((Bc, 3)).astype(np.float32))
    d[:2] = d[:2] / d[:2].norm(dim=-1, keepdim=True)  # two unit, two non-unit directions
    radii = torch.from_numpy((1e-3 * rs.uniform(0.5, 2, (Bc, 1))).astype(np.float32))
    means, covs = H.cast_rays(tdist, o, d, radii, "cone", diag=False)
    out.update(cast_tdist=tdist, cast_o=o, cast_d=d, cast_radii=radii, cast_means=means, cast_covs=covs)
    # B6
    cm, cc = H.contract(means, covs, True)
    out.update(contract_means=cm, contract_covs=cc)
    # B7
    basis = H.generate_basis("icosahedron", 2)
    lm, lv = H.lift_and_diagonalize(cm, cc, basis)
    out.update(basis=basis, lift_mean=lm, lift_var=lv, ipe=H.integrated_pos_enc(lm, lv, 0, 12))
    # B8
    out["dir_enc"] = H.pos_enc(d, 0, 4, True)
    # B10 / B11
    dens = torch.from_numpy(rs.gamma(0.5, 2.0, (B, 32)).astype(np.float32))
    td32 = torch.from_numpy(np.sort(np.exp(rs.uniform(np.log(0.1), np.log(1e4), (B, 33))).astype(np.float32), -1))
    dirs = torch.from_numpy(rs.standard_normal((B, 3)).astype(np.float32))
    rgbs = torch.from_numpy(rs.uniform(0, 1, (B, 32, 3)).astype(np.float32))
    for tag, opq in (("opq", True), ("nopq", False)):
        wts, al, tr = H.compute_alpha_weights(dens, td32, dirs, opaque_background=opq)
        out[f"aw_{tag}_w"], out[f"aw_{tag}_alpha"], out[f"aw_{tag}_trans"] = wts, al, tr
        out[f"vr_{tag}_rgb"] = H.volumetric_rendering(rgbs, wts, td32, 1.0, 1e6, False)["rgb"]
    out.update(aw_density=dens, aw_tdist=td32, aw_dirs=dirs, vr_rgbs=rgbs)
    # B12
    c, wc = _hist(rs, B, 32, zero_bins=False)
    cp, wp = _hist(rs, B, 64, zero_bins=False)
    lo, hi = H.searchsorted(cp, c)
    out.update(lo_c=c, lo_w=wc, lo_cp=cp, lo_wp=wp, lo_idx_lo=lo, lo_idx_hi=hi,
               lo_loss=H.lossfun_outer(c, wc, cp, wp), dist_loss=H.lossfun_distortion(c, wc))
    save("bkgd_helpers.npz", **out)


def _bkgd_batch(B, seed, time):
    b = synth.stage1_batch(B, seed=seed, time=time)
    b["rays_d"][B // 2:] *= 1.7  # stage-3 style: rays_d not normalised, viewdirs unit
    return b


def golden_bkgd_forward():
    sd = synth.background_state_dict(seed=777, n_states=2)
    out = {}
    for stage_n in (1, 3):
        mod, model = refload.background_model(stage_n, transitions=(0.4,), opaque_background=True)
        missing = model.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all("pos_basis_t" in k for k in missing.missing_keys), missing
        B = 8
        for tag, time, randomized, frac in (("evalA", 0.5, False, 0.3), ("evalB", 0.39998, False, 1.0),
                                            ("trainA", 0.5, True, 1.0)):
            if stage_n == 3 and tag == "evalB":
                continue
            batch = _bkgd_batch(B, 11, time)
            seed = 4321
            torch.manual_seed(seed)
            jit = [torch.rand(B, 1) for _ in range(3)]
            torch.manual_seed(seed)
            b = dict(batch)
            if stage_n == 3:
                b["times"] = torch.tensor(time)
            with refload.stage(stage_n):
                rend, hist = model(b, frac, randomized, randomized, 0.1, 1e6)
            p = f"s{stage_n}_{tag}_"
            out[p + "time"], out[p + "train_frac"] = np.float32(time), np.float64(frac)
            if randomized:
                for l in range(3):
                    out[p + f"jitter{l}"] = jit[l]
            for l in range(3):
                for k in ("density", "sdist", "weights"):
                    out[p + f"{k}{l}"] = hist[l][k]
                if stage_n == 3:
                    out[p + f"tdist{l}"] = hist[l]["tdist"]
            out[p + "rgb2"] = hist[2]["rgb"]
            if stage_n == 1:
                for l in range(3):
                    out[p + f"render{l}"] = rend[l]["rgb"]
            else:
                assert rend == []
        if stage_n == 1:
            # gradient fixture: stage-1 training loss on 4 rays (M1:491-514)
            B = 4
            batch = _bkgd_batch(B, 12, 0.5)
            torch.manual_seed(99)
            jit = [torch.rand(B, 1) for _ in range(3)]
            torch.manual_seed(99)
            model.zero_grad()
            with refload.stage(1):
                rend, hist = model(batch, 0.25, True, True, 0.1, 1e6)
            rgb = rend[-1]["rgb"]
            H = refload.helper(1)
            mse = H.img2mse(rgb, batch["target"])
            lit = mod.LitMipNeRF360
            inter = lit.interlevel_loss(None, hist)
            dist = lit.distortion_loss(None, hist)
            loss = torch.sqrt(mse + 0.001**2) + inter + 0.01 * dist
            loss.backward()
            out.update(grad_loss=loss, grad_mse=mse, grad_inter=inter, grad_dist=dist, grad_train_frac=np.float64(0.25))
            for l in range(3):
                out[f"grad_jitter{l}"] = jit[l]
            names, norms = [], []
            for n_, p_ in model.named_parameters():
                names.append(n_)
                norms.append(0.0 if p_.grad is None else float(p_.grad.double().norm()))
                if p_.grad is not None and p_.grad.numel() <= 1024:
                    out["grad__" + n_] = p_.grad
            out["grad_names"] = np.array(names)
            out["grad_norms"] = np.array(norms, dtype=np.float64)
    save("bkgd_forward.npz", **out)


def main(which="all"):
    assert refload.available(), "reference mount not present"
    if which in ("bkgd", "all"):
        golden_bkgd_helpers()
        golden_bkgd_forward()
    if which in ("human", "all"):
        from tests.golden import make_golden_human
        make_golden_human.main()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "all")
