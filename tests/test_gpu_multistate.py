"""-m gpu: the product (HIP path) against tests/golden/multistate.npz -- the reference run with 3 and 6 transitions and `time`
on both sides of every tau_k -+ 1e-5 (M:224-293, N:179-246): the state the modules select and what they render with it."""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hosnerf_amd import synth

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ms():
    return {k: v for k, v in np.load(os.path.join(G, "multistate.npz")).items()}


def _basedir(taus):
    d = tempfile.mkdtemp(prefix="hos_ms_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({f"f{i}": {"time": float(t)} for i, t in enumerate(taus)}, f)
    return d


@pytest.mark.parametrize("K", [3, 6])
def test_background_states(ms, K):
    from hosnerf_amd.mipnerf360 import MipNeRF360, select_state
    dev = torch.device("cuda")
    taus, times = ms[f"k{K}_transitions"], ms[f"k{K}_times"]
    model = MipNeRF360(_basedir(taus), opaque_background=True)
    assert all(len(m.bkgd_stateembeds) == K + 1 for m in model.mlps)
    model.load_state_dict(synth.background_state_dict(777, K + 1), strict=False)
    model = model.to(dev)
    worst = 0.0
    for i, t in enumerate(times):
        assert select_state(float(t), model.mlps[2].transitions_times) == int(ms[f"k{K}_bkgd_state"][i]), (K, i, t)
        b = {k: v.to(dev) for k, v in synth.stage1_batch(4, seed=31, time=float(t)).items()}
        with torch.no_grad():
            _, hist = model(b, 1.0, False, False, 0.1, 1e6)
        e = float((hist[2]["rgb"].cpu().numpy() - ms[f"k{K}_bkgd_rgb2"][i]).__abs__().max())
        worst = max(worst, e)
        # per-sample colours (not composited): the bound of tests/test_oracle_golden_bkgd.py for the same quantity; picking the
        # wrong embedding moves them by > 1e-3 (asserted on the fixture in the CPU test)
        assert e < 5e-4, (K, i, e)
    print(f"K={K}: worst per-sample rgb deviation {worst:.2e} over {len(times)} probe times")


@pytest.mark.parametrize("K", [3, 6])
def test_human_states(ms, K):
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.mipnerf360 import select_state
    dev = torch.device("cuda")
    taus, times = ms[f"k{K}_transitions"], ms[f"k{K}_times"]
    cfg = default_cfg(_basedir(taus))
    cfg.perturb = 0.0
    net = Network(cfg)
    net.load_state_dict(synth.human_state_dict(777, K + 1), strict=True)
    net = net.to(dev)
    for i, t in enumerate(times):
        assert select_state(float(t), net.transitions_times) == int(ms[f"k{K}_human_state"][i]), (K, i, t)
        hb = synth.human_batch(2, seed=41, time=float(t), is_train=False, iter_val=3e5)
        gb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in hb.items()}
        with torch.no_grad():
            out = net(**gb)
        m = ms[f"k{K}_human_mask"][i]
        e = float(np.abs(out["human_rgb"].cpu().numpy() * m[..., None] - ms[f"k{K}_human_rgb"][i] * m[..., None]).max())
        assert e < 5e-5, (K, i, e)
        e = float(np.abs(out["human_density"].cpu().numpy() * m - ms[f"k{K}_human_density"][i] * m).max())
        assert e < 2e-4, (K, i, e)


def test_more_than_seven_states_raises():
    from hosnerf_amd.mipnerf360 import MipNeRF360
    dev = torch.device("cuda")
    taus = np.linspace(0.1, 0.8, 7)
    model = MipNeRF360(_basedir(taus), opaque_background=True).to(dev)
    b = {k: v.to(dev) for k, v in synth.stage1_batch(2, seed=31, time=0.5).items()}
    with pytest.raises((NotImplementedError, NameError)):
        with torch.no_grad():
            model(b, 1.0, False, False, 0.1, 1e6)
