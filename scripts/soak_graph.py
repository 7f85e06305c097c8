"""Captured-graph soak: the stage's step is captured like bench.py does and replayed `steps` times with NO host synchronisation;
after every replay the loss and the per-module gradient norms are copied into device-side history buffers (stream-ordered eager
ops).  At the end: the first step whose loss / gradient norm / parameters are non-finite and the trajectory before it.
  [SOAK_GEMM=planes|split|fp32] python scripts/soak_graph.py [1|2|3] [steps] [rays]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hosnerf_amd import ops

a = sys.argv[1:]
stage, steps, rays = int(a[0]) if a else 2, int(a[1]) if len(a) > 1 else 600, int(a[2]) if len(a) > 2 else 2048
dev = torch.device("cuda")
ops.set_gemm_mode({"planes": ops.GEMM_PLANES, "split": ops.GEMM_BF16X3, "fp32": ops.GEMM_FP32}[os.environ.get("SOAK_GEMM", "planes")])
w = {1: bench.Stage1, 2: bench.Stage2, 3: bench.Stage3}[stage](dev, 0, 1, rays)
warm = 5
if os.environ.get("SOAK_NOSPLIT") and stage == 2:
    from hosnerf_amd.train import stage2_losses
    def _fb(i):
        w.opt.zero_grad()
        w.net.split_decoder_backward = False
        out = w.net(static_cycle=True, **w.batch)
        loss, _ = stage2_losses(out, w.batch)
        loss.backward()
        return loss.detach()
    w.fwd_bwd = _fb
for i in range(warm):
    w.host_prepare(i); w.eager_step(i)
torch.cuda.synchronize()
for o in w.opts():
    o.set_step_hyper(w.lr(warm))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        w.fwd_bwd(warm); w.reduce(); w.finish(warm, True)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    static_loss = w.fwd_bwd(warm)
    w.finish(warm, True)
mods = [o.module for o in w.opts()]
spans = []
for m in mods:
    d = collections.OrderedDict()
    for name, p in m.named_parameters():
        key = ".".join(name.split(".")[:2])
        off = (p.data_ptr() - m.store.param.data_ptr()) // 4
        lo, hi = d.get(key, (off, off + p.numel()))
        d[key] = (min(lo, off), max(hi, off + p.numel()))
    spans.append(d)
keys = [f"{type(m).__name__}.{k}" for m, d in zip(mods, spans) for k in d]
H = torch.zeros(steps, 2 + len(keys), device=dev)
ptrs = set()
if os.environ.get("SOAK_FILL_FREE"):
    # occupy the free blocks of the default pool with NaN: a tensor the graph still reads after its owner freed it shows at once
    junk = [torch.full((n,), float("nan"), device=dev) for n in [128] * 4096 + [1 << 18] * 64 + [1 << 22] * 32]
    torch.cuda.synchronize()
    print("free-pool poison: %d tensors" % len(junk))
for i in range(steps):
    step = warm + i
    w.host_prepare(step)
    for o in w.opts():
        o.set_step_hyper(w.lr(step))
    graph.replay()
    if os.environ.get("SOAK_SYNC"):
        torch.cuda.synchronize()
    H[i, 0] = static_loss.detach()
    H[i, 1] = torch.stack([m.store.param.abs().max() for m in mods] + [o.exp_avg.abs().max() for o in w.opts()]).max()
    ptrs.add(tuple(m.store.grad.data_ptr() for m in mods))
    c = 2
    for m, d in zip(mods, spans):
        for k, (lo, hi) in d.items():
            H[i, c] = torch.linalg.vector_norm(m.store.grad[lo:hi]); c += 1
torch.cuda.synchronize()
Hc = H.cpu()
bad = (~torch.isfinite(Hc)).any(1).nonzero()
print("distinct flat-gradient addresses seen on the host:", len(ptrs))
if os.environ.get("SOAK_TRAJ"):
    print("loss every 20 replays:", " ".join("%.4f" % float(Hc[i, 0]) for i in range(0, steps, 20)))
    print("max grad-norm per column over the run:", " ".join("%.2e" % float(x) for x in Hc[:, 2:].max(0).values))
if len(bad) == 0:
    print(f"ok: {steps} replays, final loss {float(Hc[-1, 0]):.6f}")
else:
    b = int(bad[0])
    print("EVENT: first non-finite record at replay", b, "; columns: loss, param_absmax,", keys)
    for i in range(max(0, b - 4), min(steps, b + 2)):
        print(i, " ".join("%.3e" % float(x) for x in Hc[i]))
    sys.exit(1)
