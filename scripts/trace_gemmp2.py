"""Per-tile timeline of the planes GEMM (workgroups 0 and 100, waves 0 and 4): K loop / epilogue / hand-over per output tile, the
workgroup's lifetime in shader cycles and in 100 MHz ticks (= effective clock), and the launch time.  Needs a -DHOS_TRACE2=1 build:
   scripts/build_variant.sh trace2 -DHOS_TRACE2=1 && HOS_LIB_PATH=build/variants/trace2/libhosrender.so python scripts/trace_gemmp2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
dev = torch.device("cuda")
M, N, K = int(os.environ.get('GM', 131072)), int(os.environ.get('GN', 1024)), int(os.environ.get('GK', 1024))
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 32; b = torch.zeros(N, device=dev)
X16, Xb = ops.split_planes2(X); W16, _ = ops.split_planes(W, dtype=torch.float16)
Y = ops.Planes.empty(M, N, torch.float16, dev, relu_bits=True)
Yb = ops.Planes.empty(M, N, torch.bfloat16, dev) if os.environ.get("TWOFMT") else None
tr = torch.zeros(512 + 4 * 4096, dtype=torch.int64, device=dev)
for _ in range(3):
    tr.zero_()
    ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb, aux=tr.view(torch.float32))
torch.cuda.synchronize()
c = tr.cpu()
t = c[:256].view(2, 2, 16, 4)
w = c[256:272].view(2, 2, 2, 2)
for bi, bname in enumerate(("0", "100")):
    for wi, wname in enumerate(("0", "4")):
        ticks = int(w[bi, wi, 1, 0] - w[bi, wi, 0, 0]); cyc = int(w[bi, wi, 1, 1] - w[bi, wi, 0, 1])
        if ticks <= 0:
            continue
        print(f"block {bname} wave {wname}: lifetime {cyc} cycles = {ticks * 0.01:.1f} us -> {cyc / ticks * 0.1:.3f} GHz")
        prev_end = int(w[bi, wi, 0, 1])
        for k in range(16):
            r = [int(v) for v in t[bi, wi, k]]
            if r[0] == 0:
                break
            print(f"    tile {k:2d}: before loop {r[0] - prev_end:7d}  K loop {r[1] - r[0]:7d}  epilogue {r[2] - r[1]:7d}  hand-over {(r[3] - r[2]) if r[3] else 0:6d}")
            prev_end = r[3] if r[3] else r[2]
import numpy as np
ab = c[512:].view(4096, 4).numpy()
live = ab[:, 0] > 0
if live.any():
    st, en, xcc = ab[live, 0], ab[live, 1], ab[live, 2] & 0xf
    t0 = st.min()
    print(f"all {int(live.sum())} workgroups: first start 0, last start {(st.max() - t0) * 0.01:.1f} us, first end {(en.min() - t0) * 0.01:.1f} us, last end {(en.max() - t0) * 0.01:.1f} us")
    life = (en - st) * 0.01
    print(f"   lifetime us: min {life.min():.1f} median {np.median(life):.1f} max {life.max():.1f}")
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"   XCC {x}: {int(m.sum()):4d} workgroups, start {(st[m].min() - t0) * 0.01:6.1f}..{(st[m].max() - t0) * 0.01:6.1f}, end {(en[m].min() - t0) * 0.01:6.1f}..{(en[m].max() - t0) * 0.01:6.1f} us, lifetime median {np.median(life[m]):.1f}")
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb)
e.record(); torch.cuda.synchronize()
print("fwd us per launch (traced build)", s.elapsed_time(e) * 100)
# inter-kernel gap: three back-to-back launches with their own stamp buffers; wall_clock64 is one global 100 MHz counter
trs = [torch.zeros(512 + 4 * 4096, dtype=torch.int64, device=dev) for _ in range(4)]
for t_ in trs:
    ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, Yb, aux=t_.view(torch.float32))
torch.cuda.synchronize()
prev_end = None
for i, t_ in enumerate(trs):
    ab = t_.cpu()[512:].view(4096, 4).numpy(); live = ab[:, 0] > 0
    st, en = ab[live, 0], ab[live, 1]
    msg = f"launch {i}: first start -> last end {(en.max() - st.min()) * 0.01:.1f} us (last start {(st.max() - st.min()) * 0.01:.1f}, first end {(en.min() - st.min()) * 0.01:.1f})"
    if prev_end is not None:
        msg += f"; gap from the previous launch's last end to this first start {(st.min() - prev_end) * 0.01:.1f} us"
    print(msg)
    prev_end = en.max()
