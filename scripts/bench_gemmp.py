"""Micro-benchmark + accuracy check of the planes GEMM on the dominant shape (M=32768, N=K=1024)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
M = int(os.environ.get("GM", 32768)); N = int(os.environ.get("GN", 1024)); K = int(os.environ.get("GK", 1024))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn(M, K, device=dev, generator=g); W = torch.randn(N, K, device=dev, generator=g) / K**0.5
b = torch.randn(N, device=dev, generator=g); dYf = torch.randn(M, N, device=dev, generator=g)
PAD = int(os.environ.get("PAD", 0))
Xp, XTp = ops.split_planes(X, dtype=torch.float16, transposed=False, ldo=K + PAD)
_, XTb = ops.split_planes(X, dtype=torch.bfloat16, transposed=True, row_major=False, ldt=M + PAD)
Wp, _ = ops.split_planes(W, dtype=torch.float16, ldo=K + PAD)
_, WTb = ops.split_planes(W, dtype=torch.bfloat16, transposed=True, row_major=False, ldt=N + PAD)      # [K][N]
dZ, dZT = ops.split_planes(dYf, dtype=torch.bfloat16, transposed=True, ldo=N + PAD, ldt=M + PAD)
Y = ops.Planes.empty(M, N, torch.float16, dev); YT = ops.Planes.empty(N, M, torch.bfloat16, dev)
dX = ops.Planes.empty(M, K, torch.bfloat16, dev); dXT = ops.Planes.empty(K, M, torch.bfloat16, dev)
dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
fl = 2.0 * M * N * K

def run(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / iters
    print(f"planes {name:12s} M={M} N={N} K={K}: {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s")

run("fwd", lambda: ops.linearp_fwd(Xp, K, Wp, b, M, N, True, Y, YT))
run("fwd(noT)", lambda: ops.linearp_fwd(Xp, K, Wp, b, M, N, True, Y, None))
run("dgrad", lambda: ops.linearp_dgrad(dZ, WTb, N, M, K, mask=Xp, dX=dX, dXT=dXT))
run("dgrad(noT)", lambda: ops.linearp_dgrad(dZ, WTb, N, M, K, mask=Xp, dX=dX, dXT=None))
run("wgrad", lambda: ops.linearp_wgrad(dZT, XTb, dW, db, M, N, K))
# accuracy
ops.linearp_fwd(Xp, K, Wp, b, M, N, True, Y, YT)
ref = torch.relu(X[:512].double() @ W.double().T + b.double())
print("fwd err rowmajor", (Y.float()[:512].double() - ref).abs().max().item(), "transposed", (YT.float()[:, :512].double().T - ref).abs().max().item())
ops.linearp_dgrad(dZ, WTb, N, M, K, mask=Xp, dX=dX, dXT=dXT)
refd = (dYf[:512].double() @ W.double()) * (X[:512] > 0)
print("dgrad err", (dX.float()[:512].double() - refd).abs().max().item(), (dXT.float()[:, :512].double().T - refd).abs().max().item(), "scale", refd.abs().max().item())
dW.zero_(); db.zero_()
ops.linearp_wgrad(dZT, XTb, dW, db, M, N, K)
refw = dYf.double().T @ X.double()
print("wgrad err", (dW.double() - refw).abs().max().item(), "scale", refw.abs().max().item(), "db err", (db.double() - dYf.double().sum(0)).abs().max().item())
