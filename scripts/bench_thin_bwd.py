"""Microbenchmark of hos_linear_bwd_fused (mlp_bwd_kernel<4,4,true>) and its slab reduction over the row count:
fixed cost per launch vs per-row slope.  Usage: python scripts/bench_thin_bwd.py [grid override via HOS_MB_GRID]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops

dev = "cuda"
W = torch.randn(128, 128, device=dev) * 0.1
gW = torch.zeros(128, 128, device=dev)
gb = torch.zeros(128, device=dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M in (4096, 16384, 65536, 131072, 262144, 524288):
    dz = torch.randn(M, 128, device=dev)
    X = torch.relu(torch.randn(M, 128, device=dev))
    out = torch.empty(M, 128, device=dev)
    t_imm = timeit(lambda: ops.linear_bwd_fused(dz, X, W, gW, gb, 128, 128, out, True))

    def chain7():
        with ops.deferred_bwd_reduce():
            for _ in range(7):
                ops.linear_bwd_fused(dz, X, W, gW, gb, 128, 128, out, True)
    t_def = timeit(chain7, 20)
    rows = torch.tensor([min(M, 3000)], dtype=torch.int32, device=dev)
    t_rows = timeit(lambda: ops.linear_bwd_fused(dz, X, W, gW, gb, 128, 128, out, True, rows_dev=rows))
    print(f"M={M:7d}  immediate (kernel + reduce) {t_imm:7.1f} us   7 deferred + 1 batched reduce {t_def:7.1f} us ({t_def / 7:6.1f} per layer)   "
          f"rows_dev=3000: {t_rows:6.1f} us", flush=True)
