"""Micro-benchmark of the linear entry points on the dominant NeRF-layer shape (M=32768, N=K=1024).
Usage: python scripts/bench_gemm.py [fp32|bf16x3] [fwd|dgrad|wgrad|all] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
which = sys.argv[2] if len(sys.argv) > 2 else "all"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = int(os.environ.get("GM", 32768)); N = int(os.environ.get("GN", 1024)); K = int(os.environ.get("GK", 1024))
ops.set_gemm_mode(ops.GEMM_FP32 if mode == "fp32" else ops.GEMM_BF16X3)
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
PADK = int(os.environ.get("PADK", 0)); PADN = int(os.environ.get("PADN", 0))   # leading-dimension padding experiments
X = torch.randn(M, K + PADK, device=dev, generator=g)
W = torch.randn(N, K + PADK, device=dev, generator=g) / K**0.5
b = torch.randn(N, device=dev, generator=g)
Y = torch.empty(M, N + PADN, device=dev)
dY = torch.randn(M, N + PADN, device=dev, generator=g)
dX = torch.empty(M, K + PADK, device=dev)
dW = torch.zeros(N, K + PADK, device=dev)
db = torch.zeros(N, device=dev)
fl = 2.0 * M * N * K

def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / iters
    print(f"{mode:7s} {name:6s} M={M} N={N} K={K}: {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s")

if which in ("fwd", "all"): run("fwd", lambda: ops.linear_fwd(X, K, W, b, N, Y, ops.EPI_RELU))
if which in ("dgrad", "all"): run("dgrad", lambda: ops.linear_dgrad(dY, W, N, K, dX, mask_src=X))
if which in ("wgrad", "all"): run("wgrad", lambda: ops.linear_wgrad(dY, X, dW, db, N, K))

# accuracy on a 256-row slice vs an fp64 reference
ops.linear_fwd(X, K, W, b, N, Y, ops.EPI_NONE)
ref = X[:256, :K].double() @ W[:, :K].double().T + b.double()
err = (Y[:256, :N].double() - ref).abs().max().item()
print(f"{mode:7s} fwd max abs err vs fp64 (outputs ~N(0,1)): {err:.3e}")
