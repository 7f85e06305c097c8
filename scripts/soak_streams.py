"""Race hunt for the two-stream stage-3 step (HOSNeRF.two_streams): the forward + backward of one FIXED batch with FIXED
parameters (no optimiser step) is captured like bench.py captures it and replayed `n` times without host synchronisation; every
replay must reproduce the same loss and the same per-module gradient norms (fp32 atomics allow ~1e-6 relative).  A kernel that
reads a buffer the other stream is still writing shows up as a replay that deviates.
  [HOS_TWO_STREAMS=0|1] python scripts/soak_streams.py [replays] [rays]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from hosnerf_amd import ops

a = sys.argv[1:]
steps, rays = int(a[0]) if a else 300, int(a[1]) if len(a) > 1 else 1024
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
w = bench.Stage3(dev, 0, 1, rays)
from hosnerf_amd.train import stage3_losses
g = torch.Generator().manual_seed(7)
TR = torch.rand(rays, 128, generator=g).to(dev)                 # the same stratified draws / jitters in every replay
JIT = [torch.rand(rays, generator=g).to(dev) for _ in range(3)]


def fb():
    w.ob.zero_grad(); w.oh.zero_grad()
    w.hos.human.split_decoder_backward = True
    out = w.hos.render(w.batch, randomized=True, is_train=True, static_cycle=True, jitters=JIT, t_rand=TR)
    loss, _ = stage3_losses(out, w.batch)
    loss.backward()
    w.hos.human.finish_decoder_backward()
    return loss.detach()


for _ in range(3):
    fb()
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fb()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
EAGER, SYNC = os.environ.get("SOAK_EAGER") == "1", os.environ.get("SOAK_SYNC") == "1"
if not EAGER:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = fb()
mods = [w.hos.model, w.hos.human]
cuts = [0, w.hos.human.decoder_span()[1], w.hos.human.flat_grad.numel()]
H = torch.zeros(steps, 4, device=dev, dtype=torch.float64)
for i in range(steps):
    if EAGER:
        static_loss = fb()
    else:
        graph.replay()
    if SYNC:
        torch.cuda.synchronize()
    H[i, 0] = static_loss.detach()
    H[i, 1] = torch.linalg.vector_norm(mods[0].flat_grad.double())
    H[i, 2] = torch.linalg.vector_norm(mods[1].flat_grad[cuts[0]:cuts[1]].double())
    H[i, 3] = torch.linalg.vector_norm(mods[1].flat_grad[cuts[1]:cuts[2]].double())
torch.cuda.synchronize()
Hc = H.cpu()
med = Hc.median(0).values
dev_rel = ((Hc - med).abs() / med.abs().clamp_min(1e-30)).max(0).values
print("two_streams =", w.hos.two_streams, "eager" if EAGER else "graph", "sync" if SYNC else "nosync", "| medians: loss %.8f |g_bkgd| %.6e |g_decoder| %.6e |g_human_rest| %.6e" % tuple(float(x) for x in med))
print("max relative deviation over %d replays:" % steps, " ".join("%.2e" % float(x) for x in dev_rel))
if not bool(torch.isfinite(Hc).all()) or float(dev_rel.max()) > 1e-4:
    bad = int(((Hc - med).abs() / med.abs().clamp_min(1e-30)).max(1).values.argmax())
    print("EVENT at replay", bad, Hc[bad].tolist())
    sys.exit(1)
print(f"ok: {steps} replays")
