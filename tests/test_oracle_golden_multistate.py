"""State selection with MORE THAN ONE transition, pinned by the reference itself (tests/golden/multistate.npz, made by
tests/golden/make_golden_multistate.py from S3/src/model/mipnerf360/model.py:224-293 and
S3/core/nets/human_nerf/network.py:179-246): 3 and 6 transitions, `time` on both sides of every tau_k -+ 1e-5.
CPU: the oracle's and the product's `select_state` reproduce the index of the embedding the reference fed to its first layer,
and the oracle's finest-level colours / human radiance equal the reference's at every probe time."""
import os

import numpy as np
import pytest
import torch

import oracle.background as ob
import oracle.human as oh
from hosnerf_amd import synth
from hosnerf_amd.mipnerf360 import select_state

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ms():
    return {k: v for k, v in np.load(os.path.join(G, "multistate.npz")).items()}


@pytest.mark.parametrize("K", [3, 6])
def test_state_index_is_the_reference_ladder(ms, K):
    taus, times = ms[f"k{K}_transitions"], ms[f"k{K}_times"]
    want = ms[f"k{K}_bkgd_state"].tolist()
    assert ms[f"k{K}_human_state"].tolist() == want          # N:179-246 is the same function as M:224-293
    assert sorted(set(want)) == list(range(K + 1))            # every state is reached by some probe time
    assert [ob.select_state(float(t), list(taus)) for t in times] == want
    assert [select_state(float(t), taus) for t in times] == want
    # a torch scalar `time` (what training_step hands over) selects the same state
    assert [select_state(torch.tensor(float(t), dtype=torch.float32).item(), taus) for t in times] == want


def test_more_than_seven_states_is_an_error_like_the_reference(ms):
    """M:224-293 has ladders for 1..7 embeddings; with 8 the reference dies with an unbound `embed_state_`."""
    assert str(ms["k7_raises"]) in ("UnboundLocalError", "NameError")
    with pytest.raises((NotImplementedError, NameError)):
        select_state(0.5, np.linspace(0.1, 0.8, 7).astype(np.float32))


@pytest.mark.parametrize("K", [3, 6])
def test_oracle_outputs_at_every_probe_time(ms, K):
    taus, times = [float(t) for t in ms[f"k{K}_transitions"]], ms[f"k{K}_times"]
    sd = synth.background_state_dict(777, K + 1)
    hsd = synth.human_state_dict(777, K + 1)
    step = 1 if K == 3 else 3          # K = 6: every third probe time keeps the CPU suite short; the GPU test takes all
    for i in range(0, len(times), step):
        t = float(times[i])
        b = synth.stage1_batch(4, seed=31, time=t)
        with torch.no_grad():
            _, hist = ob.mipnerf360_forward(sd, b, 1.0, False, 0.1, 1e6, transitions_times=taus, render=False)
        assert float(np.abs(hist[2]["rgb"].numpy() - ms[f"k{K}_bkgd_rgb2"][i]).max()) < 5e-4, (K, i)
        hb = synth.human_batch(2, seed=41, time=t, is_train=False, iter_val=3e5)
        with torch.no_grad():
            out = oh.human_forward(hsd, hb, transitions_times=taus, stage=3)
        m = ms[f"k{K}_human_mask"][i]
        assert float(np.abs(out["human_rgb"].numpy() * m[..., None] - ms[f"k{K}_human_rgb"][i] * m[..., None]).max()) < 5e-5, (K, i)
    # the states differ in what they render: neighbouring states give different colours on the same rays
    a, bb = ms[f"k{K}_bkgd_rgb2"][0], ms[f"k{K}_bkgd_rgb2"][-1]
    assert float(np.abs(a - bb).max()) > 1e-3
