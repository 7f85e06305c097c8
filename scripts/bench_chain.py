"""Micro-benchmark: the non-rigid MLP forward as one chain launch (hos_chain.hip) vs the layer-by-layer path.
  python scripts/bench_chain.py [rows]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.human_nerf import Network, default_cfg
P = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda")
d = tempfile.mkdtemp(); json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
net = Network(default_cfg(d)); net.load_state_dict(synth.human_state_dict(777, 2), strict=True); net = net.to(dev)
x = torch.rand(P, 3, device=dev) * 2 - 1; cond = torch.randn(75, device=dev) * 0.3; band = torch.ones(6, device=dev)
out = {}
for name, flag in (("chain", True), ("layers", False)):
    ops.MLP_CHAIN = flag
    with torch.no_grad():
        for _ in range(3): net._nonrigid_fwd(net._nr, x, cond, band, save=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net._nonrigid_fwd(net._nr, x, cond, band, save=True)
        torch.cuda.synchronize(); out[name + "_us"] = (time.perf_counter() - t0) / 20 * 1e6
E = torch.empty(P, 128, device=dev); PE = torch.empty(P, 64, device=dev)
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.embed_hannw(x, band, cond, E, PE)
    torch.cuda.synchronize(); out["embed_us"] = (time.perf_counter() - t0) / 20 * 1e6
out["rows"] = P
out["chain_algorithmic_tflops"] = 2.0 * P * 101120 / ((out["chain_us"] - out["embed_us"]) * 1e-6) / 1e12
out["chain_write_GBps"] = P * (6 * 512 + 12) / ((out["chain_us"] - out["embed_us"]) * 1e-6) / 1e9
print(json.dumps(out))
