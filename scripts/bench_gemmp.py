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
if os.environ.get("GZERO"):     # DVFS evidence: the same binary on zero-filled operands (no bit toggling in the MFMA / LDS data paths)
    X.zero_(); W.zero_(); b.zero_(); dYf.zero_()
X16, Xb = ops.split_planes2(X)
W16, _ = ops.split_planes(W, dtype=torch.float16)
_, WTb = ops.split_planes(W, dtype=torch.bfloat16, transposed=True, row_major=False)      # [K][N]
_, dZ = ops.split_planes2(dYf, want16=False)
Y = ops.Planes.empty(M, N, torch.float16, dev, relu_bits=True); Yn = ops.Planes.empty(M, N, torch.float16, dev); Yb = ops.Planes.empty(M, N, torch.bfloat16, dev)
dX = ops.Planes.empty(M, K, torch.bfloat16, dev)
C = torch.empty(M, N, device=dev)
dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
fl = 2.0 * M * N * K

ONLY = os.environ.get("GONLY", "")          # comma list of line names: run only these (PMC passes: one configuration per kernel)


def run(name, fn):
    if ONLY and name not in ONLY.split(","):
        return
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / iters
    print(f"planes {name:18s} M={M} N={N} K={K}: {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s")

run("(warm-up line)", lambda: ops.linearp_fwd(X16, K, W16, b, M, N, True, Yn, Yb))
run("fwd(2fmt)", lambda: ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb))
run("fwd(2fmt,nobits)", lambda: ops.linearp_fwd(X16, K, W16, b, M, N, True, Yn, Yb))
run("fwd(f16)", lambda: ops.linearp_fwd(X16, K, W16, b, M, N, True, Yn, None))
# round 5: the NeRF MLP's one-format layer (bf16 planes in, bf16 planes + ReLU bits out)
Wb, _ = ops.split_planes(W, dtype=torch.bfloat16)
Ybb = ops.Planes.empty(M, N, torch.bfloat16, dev, relu_bits=True)
run("fwd(bf16)", lambda: ops.linearp_fwd(Xb, K, Wb, b, M, N, True, None, Ybb))
run("fwd(f32out)", lambda: ops.linearp_fwd(X16, K, W16, b, M, N, True, None, None, C=C, epilogue=ops.EPI_RELU))
def relu_bits(Xf):
    """The ReLU bit mask of include/hosrender.h (hos_linearp_fwd, relu_bits) built with torch ops: int32 [M/32, K/64, 64]."""
    Mr, Kc = Xf.shape
    m = (Xf > 0).view(Mr // 32, 32, Kc // 64, 2, 32).permute(0, 2, 1, 3, 4)             # [rb, cb, row, y, l]
    r = torch.arange(16, device=Xf.device)
    rowidx = ((r & 3) + 8 * (r >> 2))[None, :] + 4 * torch.arange(2, device=Xf.device)[:, None]      # [h, r]
    mm = m[:, :, rowidx].permute(0, 1, 2, 5, 4, 3)                                      # [rb, cb, h, l, y, r]
    w = (2 ** (31 - (16 * torch.arange(2, device=Xf.device)[:, None] + r[None, :]))).to(torch.int64)
    v = (mm.to(torch.int64) * w).sum((-1, -2))
    return torch.where(v >= 2**31, v - 2**32, v).to(torch.int32).reshape(Mr // 32, Kc // 64, 64).contiguous()


Xm = ops.Planes.empty(M, K, torch.float16, dev, relu_bits=True)       # X as a layer output WITH its bits
Xm.t.copy_(X16.t)
if M % 32 == 0 and K % 64 == 0:
    Xm.bits.copy_(relu_bits(X))
else:
    eye16, _ = ops.split_planes(torch.eye(K, device=dev), dtype=torch.float16)
    ops.linearp_fwd(X16, K, eye16, None, M, K, True, Xm, None)
run("dgrad(planes mask)", lambda: ops.linearp_dgrad(dZ, WTb, N, M, K, mask=X16, dX=dX))
run("dgrad(bits)", lambda: ops.linearp_dgrad(dZ, WTb, N, M, K, mask=Xm, dX=dX))
run("wgrad", lambda: ops.linearp_wgrad(dZ, Xb, dW, db, M, N, K))
run("split2", lambda: ops.split_planes2(X))
if ONLY:
    sys.exit(0)
# accuracy
if M % 32 == 0 and N % 64 == 0:
    ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb)
    print("relu bits: kernel vs torch restatement, mismatching dwords", int((Y.bits != relu_bits(Y.float())).sum()))
ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb)
ref = torch.relu(X[:512].double() @ W.double().T + b.double())
print("fwd err fp16 planes", (Y.float()[:512].double() - ref).abs().max().item(), "bf16 planes", (Yb.float()[:512].double() - ref).abs().max().item())
ops.linearp_fwd(X16, K, W16, b, M, N, True, None, None, C=C, epilogue=ops.EPI_RELU)
ops.linearp_fwd(Xb, K, Wb, b, M, N, True, None, Ybb)
print("fwd err bf16 one-format layer", (Ybb.float()[:512].double() - ref).abs().max().item())
print("fwd err fp32 out", (C[:512].double() - ref).abs().max().item())
ops.linearp_dgrad(dZ, WTb, N, M, K, mask=X16, dX=dX)
dX1 = dX.float().clone()
ops.linearp_dgrad(dZ, WTb, N, M, K, mask=Xm, dX=dX)
print("dgrad bits vs planes mask: max diff", (dX.float() - dX1).abs().max().item())
refd = (dYf[:512].double() @ W.double()) * (X[:512] > 0)
print("dgrad err", (dX.float()[:512].double() - refd).abs().max().item(), "scale", refd.abs().max().item())
dW.zero_(); db.zero_()
ops.linearp_wgrad(dZ, Xb, dW, db, M, N, K)
refw = dYf.double().T @ X.double()
print("wgrad err", (dW.double() - refw).abs().max().item(), "scale", refw.abs().max().item(), "db err", (db.double() - dYf.double().sum(0)).abs().max().item())
