"""Phase stamps of hos_thin_linear_fwd (needs a -DHOS_TH_TRACE=1 build; the bias array receives the stamps):
   scripts/build_variant.sh thtrace -DHOS_TH_TRACE=1 && HOS_LIB_PATH=build/variants/thtrace/libhosrender.so python scripts/trace_thin.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
M, N, K = 262144, 256, 256
X = torch.relu(torch.randn(M, K, device=dev)); W = torch.randn(N, K, device=dev) / 16
bias = torch.zeros(256, device=dev); out = torch.empty(M, N, device=dev)
for _ in range(3):
    bias.zero_()
    ops.linear_fwd(X, K, W, bias, N, out, ops.EPI_RELU)
torch.cuda.synchronize()
v = [int(x) for x in bias.view(torch.int64).cpu() if int(x) != 0]
d = [v[i + 1] - v[i] for i in range(len(v) - 1)]
print("stamps", len(v), "total", v[-1] - v[0])
print("weights->regs", d[0], " first tile load+stage", d[1])
body = d[2:]
print("per tile [MFMAs, stage next + issue prefetch, epilogue stores, barrier]:")
for i in range(0, len(body) - 1, 4):
    print("   ", body[i:i + 4])
