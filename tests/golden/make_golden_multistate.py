"""Golden vectors that pin the multi-transition state selection by RUNNING THE REFERENCE:
`MipNeRF360MLP.forward`'s ladder (S3/src/model/mipnerf360/model.py:224-293) and `Network._query_mlp`'s
(S3/core/nets/human_nerf/network.py:179-246) with 3 and 6 transitions (4 and 7 state embeddings), `time` on both sides of
every tau_k - 1e-5 and tau_k + 1e-5, plus the ends of [0, 1].

Build container only:   python -m tests.golden.make_golden_multistate
Output: multistate.npz = inputs (times, transitions) + expected outputs (the index of the embedding the reference fed to its
first layer, per-sample rgb / density of the finest level, human rgb / density).  Weights come from hosnerf_amd.synth.
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

TAUS = {3: (0.2, 0.45, 0.7), 6: (0.1, 0.25, 0.4, 0.55, 0.7, 0.85)}
DELTA = 2e-6     # distance from a threshold: 60 float32 ulps at 0.4, so float32 / float64 evaluation of `tau -+ eps` cannot matter
B_BKGD, B_HUMAN = 4, 2


def probe_times(taus):
    """both sides of tau_k - 1e-5 and of tau_k + 1e-5 for every k, the interval mid-points and the ends."""
    ts = [0.0, 1.0]
    for k, t in enumerate(taus):
        t = float(np.float32(t))
        for thr in (t - 1e-5, t + 1e-5):
            ts += [thr - DELTA, thr + DELTA]
        nxt = float(taus[k + 1]) if k + 1 < len(taus) else 1.0
        ts.append(0.5 * (t + nxt))
    return np.array(sorted(ts), dtype=np.float64)


def _which(embeds, used):
    """index of the state embedding whose values `used` (64 columns of the first layer's input row) equals."""
    hits = [i for i, e in enumerate(embeds) if torch.equal(e.detach().reshape(-1), used.reshape(-1))]
    assert len(hits) == 1, hits
    return hits[0]


def main():
    assert refload.available()
    out = {}
    for K, taus in TAUS.items():
        times = probe_times(taus)
        out[f"k{K}_transitions"] = np.array(taus, dtype=np.float32)
        out[f"k{K}_times"] = times
        # ---- background: the whole 3-level model, eval sampling; the NeRF MLP's first layer sees [IPE 504 | state 64]
        sd = synth.background_state_dict(seed=777, n_states=K + 1)
        mod, model = refload.background_model(3, transitions=taus, opaque_background=True)
        miss = model.load_state_dict(sd, strict=False)
        assert not miss.unexpected_keys and all("pos_basis_t" in k for k in miss.missing_keys), miss
        assert len(model.mlps[2].bkgd_stateembeds) == K + 1
        seen = {}
        assert len(model.mlps) == 3                       # M:411: two proposal MLPs + the NeRF MLP
        for name, mlp in enumerate(model.mlps):
            mlp.pts_linear[0].register_forward_pre_hook(
                lambda m, inp, name=name, mlp=mlp: seen.__setitem__(name, _which(list(mlp.bkgd_stateembeds), inp[0][0, 0, 504:568])))
        idx, rgb, dens = [], [], []
        for t in times:
            b = synth.stage1_batch(B_BKGD, seed=31, time=float(t))
            b["times"] = torch.tensor(float(t))
            seen.clear()
            with refload.stage(3), torch.no_grad():
                rend, hist = model(b, 1.0, False, False, 0.1, 1e6)
            assert len(set(seen.values())) == 1, seen     # every MLP of the model picks the same state
            idx.append(seen[2])
            rgb.append(hist[2]["rgb"].numpy())
            dens.append(hist[2]["density"].numpy())
        out[f"k{K}_bkgd_state"] = np.array(idx, dtype=np.int64)
        out[f"k{K}_bkgd_rgb2"] = np.stack(rgb)
        out[f"k{K}_bkgd_density2"] = np.stack(dens)
        print(f"K={K} background states:", idx)

        # ---- human-object network (stage-3 form), eval
        cfg, net = refload.human_network(3, transitions=taus)
        hsd = synth.human_state_dict(777, K + 1)
        print(net.load_state_dict(hsd, strict=True))
        net.eval()
        cfg.perturb = 0.0
        hseen = {}
        lin = net.cnl_mlp.pts_linears[0]      # mlp_rgb_sigma.py:29: the input Linear sees [fourier 63 | state 64]
        lin.register_forward_pre_hook(
            lambda m, inp: hseen.__setitem__("s", _which(list(net.human_stateembeds), inp[0][0, 63:127])))
        hidx, hrgb, hden, hmask = [], [], [], []
        for t in times:
            hb = synth.human_batch(B_HUMAN, seed=41, time=float(t), is_train=False, iter_val=3e5)
            skip = ("rays_o_bkg", "rays_d_bkg", "viewdirs_bkg", "radii", "newsmpl_to_scale_world")
            hseen.clear()
            with refload.stage(3), torch.no_grad():
                res = net(**{k: v for k, v in hb.items() if k not in skip})
            hidx.append(hseen["s"])
            hrgb.append(res["human_rgb"].numpy())
            hden.append(res["human_density"].numpy())
            hmask.append(res["pts_mask"].numpy())
        out[f"k{K}_human_state"] = np.array(hidx, dtype=np.int64)
        out[f"k{K}_human_rgb"] = np.stack(hrgb)
        out[f"k{K}_human_density"] = np.stack(hden)
        out[f"k{K}_human_mask"] = np.stack(hmask)
        print(f"K={K} human states:     ", hidx)
        assert hidx == idx, "both ladders are the same function of (time, transitions)"

    # ---- more than 7 states: the reference's ladders define no branch (M:224-293) -> `embed_state_` is unbound
    taus8 = tuple(np.linspace(0.1, 0.8, 7))
    mod, model = refload.background_model(3, transitions=taus8, opaque_background=True)
    b = synth.stage1_batch(2, seed=31, time=0.5)
    b["times"] = torch.tensor(0.5)
    try:
        with refload.stage(3), torch.no_grad():
            model(b, 1.0, False, False, 0.1, 1e6)
        raised = ""
    except Exception as e:  # UnboundLocalError (a NameError subclass)
        raised = type(e).__name__
    out["k7_raises"] = np.array(raised)
    print("8 states ->", raised)
    path = os.path.join(HERE, "multistate.npz")
    np.savez_compressed(path, **out)
    print(f"multistate.npz: {os.path.getsize(path)/1024:.1f} KB")


if __name__ == "__main__":
    main()
