"""Un-profiled timeline of the two-stream stage-3 step: stamp kernels (hos_debug_stamp: the 100 MHz device counter) captured INSIDE the
step's graph at the start / end of each branch's forward and backward, read back after the replays.
  python scripts/diag_overlap.py [rays] [steps]        (HOS_TWO_STREAMS=0: the one-stream order for comparison)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hosnerf_amd import _lib, ops  # noqa: E402
from hosnerf_amd.train import stage3_losses, step_all  # noqa: E402

RAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ops.set_gemm_mode(ops.GEMM_PLANES)
wl = bench.Stage3(dev, 0, 1, RAYS)
hos, batch = wl.hos, wl.batch
NAMES = ["step start", "human fwd start", "human fwd end", "bkgd fwd start", "bkgd fwd end", "merge+loss end",
         "bkgd bwd start", "human bwd start", "bkgd bwd end (main stream at join)", "human bwd end (side stream at join)",
         "decoder bwd end", "adam end"]
S = {n: i for i, n in enumerate(NAMES)}
stamps = torch.zeros(len(NAMES), dtype=torch.int64, device=dev)


def stamp(name, stream=None):
    s = stream if stream is not None else torch.cuda.current_stream(dev)
    with torch.cuda.stream(s):
        _lib.call("hos_debug_stamp", stamps.data_ptr(), S[name])


class Stamp(torch.autograd.Function):
    """Identity; stamps `fwd` when the forward reaches it and `bwd` when the backward does (on the stream autograd runs it on)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        if fwd:
            stamp(fwd)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.bwd:
            stamp(ctx.bwd)
        return g, None, None


# wrap the two branches
human_fwd, model_fwd = hos.human.forward, hos.model.forward


def human_wrapped(*a, **k):
    stamp("human fwd start")
    out = human_fwd(*a, **k)
    if out["human_rgbsigma"].requires_grad:
        out["human_rgbsigma"] = Stamp.apply(out["human_rgbsigma"], "human fwd end", "human bwd start")
    return out


def model_wrapped(*a, **k):
    stamp("bkgd fwd start")
    rend, hist = model_fwd(*a, **k)
    last = hist[-1]
    if last["density"].requires_grad:
        last["density"] = Stamp.apply(last["density"], "bkgd fwd end", "bkgd bwd start")
    return rend, hist


hos.human.forward, hos.model.forward = human_wrapped, model_wrapped
join = hos.join_side_stream


def join_wrapped(main=None):
    for s in hos._side.values():
        stamp("human bwd end (side stream at join)", s)
    stamp("bkgd bwd end (main stream at join)", main)
    join(main)


hos.join_side_stream = join_wrapped


def step(i):
    stamp("step start")
    wl.ob.zero_grad(); wl.oh.zero_grad()
    hos.human.split_decoder_backward = True
    out = hos.render(batch, randomized=True, is_train=True, static_cycle=True)
    loss, _ = stage3_losses(out, batch)
    stamp("merge+loss end")
    loss.backward()
    hos.human.finish_decoder_backward()
    stamp("decoder bwd end")
    step_all(wl.opts(), None, dynamic=True, reduced=True)
    stamp("adam end")
    return loss.detach()


for o in wl.opts():
    o.set_step_hyper(wl.lr(0))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        step(i)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step(3)
torch.cuda.synchronize()
acc = torch.zeros(len(NAMES), dtype=torch.float64)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(5):
    g.replay()
torch.cuda.synchronize()
e0.record()
for i in range(STEPS):
    g.replay()
    torch.cuda.synchronize()
    t = stamps.cpu().double()
    acc += (t - t[0]) * 0.01
e1.record()
torch.cuda.synchronize()
print(f"rays {RAYS}, two_streams {hos.two_streams}: timeline of one replayed step, us from its first stamp (mean of {STEPS} replays, synchronised between replays)")
for n in NAMES:
    print(f"   {acc[S[n]] / STEPS:9.1f}  {n}")
e0.record()
for i in range(STEPS):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"back-to-back replays: {e0.elapsed_time(e1) / STEPS:.3f} ms per step (with the 12 stamp kernels), loss {float(loss):.5f}")
