"""Micro-benchmark: the canonical MLP forward as one chain launch vs the layer-by-layer path.  python scripts/bench_chain256.py [rows]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.human_nerf import Network, default_cfg
P = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda")
d = tempfile.mkdtemp(); json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
net = Network(default_cfg(d)); net.load_state_dict(synth.human_state_dict(777, 2), strict=True); net = net.to(dev)
x = torch.rand(P, 3, device=dev) * 2 - 1
out = {"rows": P}
for name, flag in (("chain", True), ("layers", False)):
    ops.MLP_CHAIN256 = flag
    with torch.no_grad():
        for _ in range(3): net._canonical_fwd(x, 1, save=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net._canonical_fwd(x, 1, save=True)
        torch.cuda.synchronize(); out[name + "_us"] = (time.perf_counter() - t0) / 10 * 1e6
out["chain_algorithmic_tflops"] = 2.0 * P * 524800 / (out["chain_us"] * 1e-6) / 1e12
print(json.dumps(out))
