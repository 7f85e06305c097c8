import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd._lib import call, ptr
dev = torch.device("cuda")
P, K, V, CL = 262144, 26, 32, 32
g = torch.Generator().manual_seed(0)
# points along 2048 rays x 128 samples inside the box (spatially coherent like the real thing)
o = torch.rand(2048, 1, 3, generator=g) * 0.4 - 0.2
d = torch.nn.functional.normalize(torch.randn(2048, 1, 3, generator=g), dim=-1)
t = torch.linspace(-0.8, 0.8, 128).view(1, 128, 1)
cnl = (o + d * t).reshape(-1, 3).contiguous().to(dev)
R = torch.eye(3).repeat(K, 1, 1).to(dev); T = torch.zeros(K, 3, device=dev)
vol = torch.softmax(torch.randn(V, V, V, CL, generator=g), -1).to(dev)
bmin = torch.tensor([-1.0, -1.0, -1.0], device=dev); bscale = torch.tensor([1.0, 1.0, 1.0], device=dev)
gx = torch.randn(P, 3, device=dev)
g_cnl = torch.empty_like(cnl); g_vol = torch.zeros_like(vol); g_R = torch.zeros_like(R); g_T = torch.zeros_like(T)
for name, gv in (("with volume scatter", g_vol), ("no scatter", None)):
    fn = lambda: call("hos_lbs_forward_bwd", ptr(cnl), ptr(R), ptr(T), ptr(vol), V, CL, ptr(bmin), ptr(bscale), P, K, ptr(gx), ptr(g_cnl), ptr(gv), ptr(g_R), ptr(g_T))
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    print(name, a.elapsed_time(b) * 100, "us")
