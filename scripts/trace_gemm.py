"""Phase timeline of the split-precision forward GEMM (one workgroup, K tiles 8..11): HOS_GEMM_ABLATE=16."""
import os, sys
os.environ["HOS_GEMM_ABLATE"] = str(16 | int(os.environ.get("EXTRA_ABLATE", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
M, N, K = 32768, 1024, 1024
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 32; b = torch.zeros(N, device=dev); Y = torch.empty(M, N, device=dev)
tr = torch.zeros(8 * 4 * 8, dtype=torch.int64, device=dev)
aux = tr.view(torch.float32)
for _ in range(3):
    ops.linear_fwd(X, K, W, b, N, Y, ops.EPI_RELU, aux=aux)
torch.cuda.synchronize()
t = tr.cpu().view(8, 4, 8)
t0 = int(t[t > 0].min())
names = ["top", "B:stored", "postbarB", "loads issued", "computed", "postbar", "A:stored", "end"]
for w in range(8):
    print(f"wave {w} ({'B' if w >= 4 else 'A'})")
    for it in range(4):
        row = t[w, it]
        print("   iter", it, " ".join(f"{names[k]}={int(row[k]) - t0 if row[k] > 0 else -1:6d}" for k in range(8)))
