"""encode_ipe_planes: us per launch and TB/s of plane stores at the stage-3 step's level sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
dev = torch.device("cuda")
for B, S in ((4096, 64), (4096, 32), (65536, 64)):
    b = {k: v.to(dev) for k, v in synth.stage1_batch(B, seed=3).items()}
    tdist = torch.sort(torch.rand(B, S + 1, device=dev), -1).values * 4 + 0.1
    basis = torch.randn(3, 21, device=dev)
    embed = torch.randn(64, device=dev)
    for name, kw in (("fp16+bf16", dict(want_bf16=True, want_fp16=True)), ("fp16", dict(want_bf16=False, want_fp16=True)), ("bf16", dict(want_bf16=True, want_fp16=False))):
        fn = lambda: ops.encode_ipe_planes(tdist, b["rays_o"], b["rays_d"], b["radii"], basis, embed, 576, **kw)
        try:
            for _ in range(5): fn()
        except Exception as ex:            # (a round-4 library has no bf16-only form)
            print(f"encode_ipe_planes[{B}x{S}] {name:10s} not available: {type(ex).__name__}")
            continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        nbytes = B * S * 576 * 4 * (2 if name == "fp16+bf16" else 1)
        print(f"encode_ipe_planes[{B}x{S}] {name:10s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s of stores")
