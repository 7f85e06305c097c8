"""Micro-benchmark of the thin-layer kernels at the human branch's size (M = 2048 rays x 128 samples):
hos_thin_linear_fwd / _dgrad [M,256,256], hos_linear_bwd_fused [M,128,128], hos_linear_wgrad_tr [256,256,M].
  python scripts/bench_thin.py [reps]        (reps > 0: also prints event timings)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M = 262144
X = torch.relu(torch.randn(M, 256, device=dev)); W = torch.randn(256, 256, device=dev) / 16; b = torch.zeros(256, device=dev)
dY = torch.randn(M, 256, device=dev) * 1e-3
Y = torch.empty(M, 256, device=dev); dX = torch.empty(M, 256, device=dev)
dW = torch.zeros(256, 256, device=dev); db = torch.zeros(256, device=dev)
X1 = X[:, :128].contiguous(); dY1 = dY[:, :128].contiguous(); W1 = W[:128, :128].contiguous()
dX1 = torch.empty(M, 128, device=dev); dW1 = torch.zeros(128, 128, device=dev); db1 = torch.zeros(128, device=dev)
bits = ops.thin_relu_bits(M, dev)
fns = {
    "thin_fwd[262144,256,256]": (lambda: ops.linear_fwd(X, 256, W, b, 256, Y, ops.EPI_RELU), M * 256 * 8),
    "thin_fwd+relu_bits[262144,256,256]": (lambda: ops.linear_fwd(X, 256, W, b, 256, Y, ops.EPI_RELU, relu_bits=bits), M * 256 * 8 + M * 32),
    "thin_dgrad[262144,256,256]": (lambda: ops.linear_dgrad(dY, W, 256, 256, dX, mask_src=X), M * 256 * 12),
    "thin_dgrad(mask_bits)[262144,256,256]": (lambda: ops.linear_dgrad(dY, W, 256, 256, dX, mask_bits=bits), M * 256 * 8 + M * 32),
    "thin_dgrad[262144,64,256]": (lambda: ops.linear_dgrad(dY, W, 256, 64, dX, thin=True), M * (256 + 64) * 4),
    "mlp_bwd_fused[262144,128,128]": (lambda: ops.linear_bwd_fused(dY1, X1, W1, dW1, db1, 128, 128, dX1, True), M * 128 * 12),
    "wgrad_tr[256,256,262144]": (lambda: ops.linear_wgrad(dY, X, dW, db, 256, 256), M * 256 * 8),
}
res = {}
for rnd in range(3):            # three rounds, best of: the first launches of a process run at a lower clock (order effects of ~20 %)
    for name, (fn, nbytes) in fns.items():
        for _ in range(3): fn()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) * 1e3 / reps
        if name not in res or us < res[name]["us"]:
            res[name] = {"us": us, "algorithmic_bytes": nbytes, "TB/s": nbytes / us / 1e6}
print(json.dumps(res, indent=1))
