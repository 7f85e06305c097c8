"""Backward of one 6 x 128 non-rigid MLP: the three group launches (hos_mlp_chain_bwd) against the eight fused-layer launches.
usage: python scripts/bench_chainbwd.py [rows] [iters]"""
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hosnerf_amd import ops, synth  # noqa: E402
from hosnerf_amd.human_nerf import Network, default_cfg  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda"
d = tempfile.mkdtemp(prefix="hos_basedir_")
with open(os.path.join(d, "transitions_times.json"), "w") as f:
    json.dump({"f0": {"time": 0.4}}, f)
net = Network(default_cfg(d))
net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
net = net.to(dev)
g = torch.Generator().manual_seed(0)
x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
cond = (torch.randn(75, generator=g) * 0.3).to(dev)
band = torch.tensor([1.0, 1.0, 0.8, 0.3, 0.0, 0.0]).to(dev)
gx = torch.randn(P, 3, generator=g).to(dev)
specs = net._nr
xyz, saved = net._nonrigid_fwd(specs, x, cond, band, save=True)
for cb in (False, True, False, True):
    ops.MLP_CHAIN_BWD = cb
    for _ in range(3):
        net._nonrigid_bwd(specs, saved, x, band, gx)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(IT):
        net._nonrigid_bwd(specs, saved, x, band, gx)
    b.record()
    torch.cuda.synchronize()
    print(f"rows {P}  chain_bwd={int(cb)}  {1e3 * a.elapsed_time(b) / IT:8.1f} us per MLP backward (embed backward and unfold included)")
