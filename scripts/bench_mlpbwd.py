"""Fused thin-layer backward (hos_mlpbwd.hip) against the wgrad + dgrad pair it replaces, [M,128,128] at M = 2048 x 128."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
res = {}
for N, K in ((128, 128), (3, 128), (128, 64)):
    Np = (N + 31) // 32 * 32
    dY = torch.zeros(M, Np, device=dev); dY[:, :N] = torch.randn(M, N, device=dev) * 1e-3
    X = torch.relu(torch.randn(M, max(K, 64), device=dev)); W = torch.randn(Np, 128, device=dev) / 11
    dW = torch.zeros(Np, 128, device=dev); db = torch.zeros(Np, device=dev); out = torch.empty(M, K, device=dev)
    def fused(): ops.linear_bwd_fused(dY, X, W, dW, db, N, K, out, True)
    def pair():
        ops.linear_wgrad(dY, X, dW, db, N, K)
        ops.linear_dgrad(dY, W, Np, K, out, mask_src=X)
    for name, fn in (("fused", fused), ("pair", pair)):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 50
        res[f"{name}[{M},{N},{K}]"] = {"us": us, "GB/s": (M * (Np + K) * 4 + M * K * 4) / us / 1e3}
print(json.dumps(res, indent=1))
