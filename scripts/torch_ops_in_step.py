"""Which torch (aten) launches does a stage-3 step still contain, and where do they come from?  One eager step under
torch.profiler with python stacks; aggregated by (aten op, innermost hosnerf_amd source line).   python scripts/torch_ops_in_step.py [rays]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hosnerf_amd import ops
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
w = bench.Stage3(dev, 0, 1, rays)
w.hos.two_streams = False
for i in range(3):
    w.eager_step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    w.eager_step(3)
    torch.cuda.synchronize()
agg = collections.Counter()
dur = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12):
    if not ev.key.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "hosnerf_amd/" in fr or "bench.py" in fr:
            site = fr.split("/")[-1][:80]
            break
    agg[(ev.key, site)] += ev.count
    dur[(ev.key, site)] += ev.self_device_time_total
tot = sum(dur.values())
print(f"aten ops with device time of their own in one {rays}-ray stage-3 step: {sum(agg.values())} launches, {tot:.0f} us")
for k, n in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:50]:
    print(f"{n:4d} x {k[0]:30s} {dur[k]:8.1f} us   {k[1]}")
