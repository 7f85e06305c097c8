"""Data-parallel backward of the human network: cutting the autograd graph at the motion-weight volume and
backpropagating the volume gradient through the decoder afterwards (Network.decoder_backward -- the point where the ranks
exchange 3.5 MB instead of the decoder's 253 MB of parameter gradients) gives the same flat gradient as one backward."""
import json
import os
import tempfile

import pytest
import torch

from hosnerf_amd import synth

pytestmark = pytest.mark.gpu


def test_split_decoder_backward_equals_plain_backward():
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets, stage2_losses
    dev = torch.device("cuda")
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    cfg = default_cfg(d)
    cfg.perturb = 0.0
    net = Network(cfg, stage=2)
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    net = net.to(dev)
    b = synth.add_patch_supervision(synth.human_batch(256, seed=9, time=0.5, is_train=True, iter_val=3e5), 1, 16, 9)
    gb = batch_to_device(prepare_patch_targets(b), dev)
    grads = {}
    for split in (False, True):
        net.zero_grad()
        net.split_decoder_backward = split
        out = net(static_cycle=True, **gb)
        loss, _ = stage2_losses(out, gb)
        loss.backward()
        if split:
            off, n = net.decoder_span()
            assert float(net.flat_grad[off:off + n].abs().max()) == 0.0, "the decoder must not have been reached yet"
            g = net.pending_volume_grad()
            assert g is not None and g.shape == (27, 32, 32, 32)
            net.decoder_backward()
            assert net.pending_volume_grad() is None
        grads[split] = net.flat_grad.clone()
    net.split_decoder_backward = False
    off, n = net.decoder_span()
    a, bb = grads[False], grads[True]
    assert float(a[off:off + n].abs().max()) > 0
    assert float((a - bb).abs().max()) <= 1e-6 * float(a.abs().max())
    lo, cnt = net.reduce_ranges()[0]
    assert lo == off + n and lo + cnt == net.flat_param.numel() and cnt < 2_000_000
