"""Phase timeline of the planes forward GEMM (workgroup 0, K tiles 8..11).  Needs a -DHOS_TRACE=1 build:
   scripts/build_variant.sh trace -DHOS_TRACE=1 && HOS_LIB_PATH=build/variants/trace/libhosrender.so python scripts/trace_gemmp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
M, N, K = int(os.environ.get('GM', 32768)), 1024, 1024
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 32; b = torch.zeros(N, device=dev)
X16, _ = ops.split_planes2(X, wantb=False); W16, _ = ops.split_planes(W, dtype=torch.float16)
Y = ops.Planes.empty(M, N, torch.float16, dev)
tr = torch.zeros(8 * 4 * 8 + 64 + 32, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, None, aux=tr.view(torch.float32))
torch.cuda.synchronize()
t = tr.cpu()[:256].view(8, 4, 8)
bt = tr.cpu()[256:320].view(2, 8, 4)
rt = tr.cpu()[320:352].view(2, 8, 2)
t0 = int(t[t > 0].min())
names = ["top", "G1", "G2", "G3", "dma+lds waited", "barrier", "dma issued", "G4"]
for w in range(8):
    print(f"wave {w}")
    for it in range(4):
        row = t[w, it]
        print("   tile", it, " ".join(f"{names[k]}={int(row[k]) - t0 if row[k] > 0 else -1:6d}" for k in range(8)))

for bb in range(2):
    cyc = int(bt[bb, 0, 3] - bt[bb, 0, 0]); ticks = int(rt[bb, 0, 1] - rt[bb, 0, 0])
    print(f"effective shader clock, block {'0' if bb == 0 else '300'}: {cyc} cycles in {ticks} ticks of the 100 MHz counter = {cyc / max(ticks, 1) * 0.1:.3f} GHz")
print("block timeline (cycles from entry): loop start, loop end, exit")
for bb in range(2):
    for w in range(8):
        r = bt[bb, w]
        print(f"   block {'0' if bb == 0 else '300'} wave {w}: prologue {int(r[1]-r[0]):7d}  loop {int(r[2]-r[1]):7d}  epilogue {int(r[3]-r[2]):7d}  total {int(r[3]-r[0]):7d}")
import time
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, None)
e.record(); torch.cuda.synchronize()
print("fwd us per launch", s.elapsed_time(e) * 100)
