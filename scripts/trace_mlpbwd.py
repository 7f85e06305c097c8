"""Phase stamps of the fused thin-layer backward (needs a -DHOS_MB_TRACE=1 build):
   scripts/build_variant.sh mbtrace -DHOS_MB_TRACE=1 && HOS_LIB_PATH=build/variants/mbtrace/libhosrender.so python scripts/trace_mlpbwd.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
N = K = 128
dY = torch.randn(M, N, device=dev) * 1e-3; X = torch.relu(torch.randn(M, K, device=dev)); W = torch.randn(N, K, device=dev) / 11
dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev); out = torch.empty(M, K, device=dev)
ws = ops._bwd_workspace(dY.device)
for _ in range(3):
    ws.zero_()
    ops.linear_bwd_fused(dY, X, W, dW, db, N, K, out, True)
torch.cuda.synchronize()
st = ws[:256].view(torch.int64).cpu()
for b, name in ((0, "block 0"), (1, "block 200")):
    v = [int(x) for x in st[b * 64:(b + 1) * 64] if int(x) != 0]
    d = [v[i + 1] - v[i] for i in range(len(v) - 1)]
    print(name, "stamps", len(v), "total", v[-1] - v[0])
    print("   entry->W staged", d[0], "| per iteration [sstore, barrier, compute, barrier]:")
    body = d[1:]
    for i in range(0, min(len(body) - 2, 24), 4):
        print("     ", body[i:i + 4])
    print("   tail (dW/db out):", body[-2:])
