import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd._lib import call, ptr
from hosnerf_amd import synth
dev = torch.device("cuda")
b = synth.human_batch(2048, seed=5, time=0.5, is_train=True, iter_val=3e5)
P, K, V = 2048 * 128, 26, 32
o, d = b["rays"][0], b["rays"][1]
t = torch.linspace(0, 1, 128).view(1, 128, 1)
z = b["near"][:, None] * (1 - t) + b["far"][:, None] * t
pts = (o[:, None] + d[:, None] * z).reshape(-1, 3).contiguous().to(dev)
g = torch.Generator().manual_seed(0)
R = torch.eye(3).repeat(K, 1, 1); R += 0.05 * torch.randn(K, 3, 3, generator=g); T = 0.1 * torch.randn(K, 3, generator=g)
R, T = R.to(dev), T.to(dev)
vol = torch.softmax(torch.randn(K + 1, V, V, V, generator=g) * 2, 0).to(dev)
bmin = b["cnl_bbox_min_xyz"].to(dev); bscale = b["cnl_bbox_scale_xyz"].to(dev)
gx = torch.randn(P, 3, device=dev); gm = torch.randn(P, device=dev)
g_vol = torch.zeros_like(vol); g_R = torch.zeros(K, 9, device=dev); g_T = torch.zeros(K, 3, device=dev)
for name, gv in (("run", g_vol),):
    fn = lambda: call("hos_human_sample_warp_bwd", ptr(pts), ptr(R), ptr(T), ptr(vol), V, ptr(bmin), ptr(bscale), P, K, ptr(gx), ptr(gm), ptr(gv), ptr(g_R), ptr(g_T))
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(e) * 100, "us")
