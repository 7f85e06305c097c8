"""LPIPS term alone: forward + backward on 4 patches of 32 x 32 (the stage-3 item), seeded random VGG-16 weights.
usage: python scripts/bench_lpips.py [iters]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hosnerf_amd.lpips import LPIPS, VGG16_CFG, CHNS  # noqa: E402

IT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
g = torch.Generator().manual_seed(4321)
sd, cin, idx = {}, 3, 0
for v in VGG16_CFG:
    if v == "M":
        idx += 1
        continue
    sd[f"{idx}.weight"] = torch.randn(v, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    sd[f"{idx}.bias"] = torch.zeros(v)
    cin, idx = v, idx + 2
net = LPIPS().load_vgg16_features(sd, dev).load_lin(torch.rand(sum(CHNS), generator=g) * 0.1, dev)
Np, P = 4, 32
rgb = torch.rand(Np * P * P, 3, generator=g).to(dev).requires_grad_(True)
targ = torch.rand(Np, P, P, 3, generator=g).to(dev)
ridx = torch.arange(Np * P * P, dtype=torch.int32, device=dev)
bg = torch.zeros(3, device=dev)
for _ in range(3):
    net.loss(rgb, targ, ridx, bg).backward()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(IT):
    net.loss(rgb, targ, ridx, bg).backward()
b.record()
torch.cuda.synchronize()
print(f"LPIPS fwd + bwd, {Np} patches of {P}x{P}: {1e3 * a.elapsed_time(b) / IT:.1f} us (eager launches)")
