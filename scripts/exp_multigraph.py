"""Experiment: the stage-3 step as SIX captured graphs on two real streams (human forward || background forward -> merge + losses +
head backward -> human backward || background backward -> norm + Adam) instead of one graph with a fork / join inside.  The replay of
a single graph does not run its two captured branches concurrently where it matters at 512 rays; separate graphs
launched on two streams are ordered by events only.   usage: python scripts/exp_multigraph.py [rays] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hosnerf_amd import ops  # noqa: E402
from hosnerf_amd.train import stage3_losses, step_all  # noqa: E402

RAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ops.set_gemm_mode(ops.GEMM_PLANES)
wl = bench.Stage3(dev, 0, 1, RAYS)
hos, batch = wl.hos, wl.batch
hos.two_streams = False                       # the fork / join is made here, not inside render()
torch.cuda.synchronize()
S0 = torch.cuda.Stream()                       # graphs cannot be captured on the default stream
S1 = torch.cuda.Stream()
torch.cuda.set_stream(S0)


def leafify(d, keys=None):
    """Detached leaves for every differentiable tensor of `d` (the cut between a branch and the head)."""
    cut, orig = {}, {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor) and v.requires_grad and (keys is None or k in keys):
            orig[k] = v
            cut[k] = v.detach().requires_grad_(True)
        else:
            cut[k] = v
    return cut, orig


st = {}


def human_fwd():
    wl.oh.zero_grad()
    hos.human.split_decoder_backward = False
    st["out_h"] = hos.human(static_cycle=True, **batch)


def bkgd_fwd():
    wl.ob.zero_grad()
    batch_bkg = {"rays_o": batch["rays_o_bkg"], "rays_d": batch["rays_d_bkg"], "viewdirs": batch["viewdirs_bkg"],
                 "radii": batch["radii"], "times": batch["time"]}
    _, st["hist"] = hos.model(batch_bkg, 1.0, True, True, hos.near_bkg, hos.far_bkg)


def head():
    out_c, st["orig_h"] = leafify(st["out_h"])
    last = st["hist"][-1]
    last_c, st["orig_b"] = leafify(last, ("rgb", "density"))
    rgb, hw, idx_fg, order, zh = ops.merge_composite(
        last_c["tdist"], last_c["rgb"], last_c["density"], out_c["human_rgbsigma"], out_c["newsmpl_pts"], out_c["pts_mask"],
        batch["rays_o_bkg"], batch["rays_d_bkg"], batch["newsmpl_to_scale_world"])
    out_c.update(rgb=rgb, idx_fg=idx_fg, total_order=order, human_weights_sorted=hw, z_vals_human=zh)
    loss, _ = stage3_losses(out_c, batch)
    loss.backward()
    st["loss"] = loss.detach()
    st["g_h"] = {k: out_c[k].grad for k in st["orig_h"] if out_c[k].grad is not None}
    st["g_b"] = {k: last_c[k].grad for k in st["orig_b"] if last_c[k].grad is not None}


def human_bwd():
    ks = list(st["g_h"])
    torch.autograd.backward([st["orig_h"][k] for k in ks], [st["g_h"][k] for k in ks])


def bkgd_bwd():
    ks = list(st["g_b"])
    torch.autograd.backward([st["orig_b"][k] for k in ks], [st["g_b"][k] for k in ks])


def tail(dynamic):
    step_all(wl.opts(), None if dynamic else wl.lr(0), dynamic=dynamic, reduced=True)


def eager_step():
    S1.wait_stream(S0)
    with torch.cuda.stream(S1):
        human_fwd()
    bkgd_fwd()
    S0.wait_stream(S1)
    head()
    S1.wait_stream(S0)
    with torch.cuda.stream(S1):
        human_bwd()
    bkgd_bwd()
    S0.wait_stream(S1)
    tail(False)


for _ in range(3):
    eager_step()
torch.cuda.synchronize()
for o in wl.opts():
    o.set_step_hyper(wl.lr(0))


def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g


# warm-up on side streams as torch asks, then capture piece by piece (every piece sees the static outputs of the previous ones)
G = {}
G["hf"] = capture(human_fwd, S1)
G["bf"] = capture(bkgd_fwd, S0)
torch.cuda.synchronize()
G["head"] = capture(head, S0)
torch.cuda.synchronize()
G["hb"] = capture(human_bwd, S1)
G["bb"] = capture(bkgd_bwd, S0)
torch.cuda.synchronize()
G["tail"] = capture(lambda: tail(True), S0)
torch.cuda.synchronize()


def replay_step():
    S1.wait_stream(S0)
    with torch.cuda.stream(S1):
        G["hf"].replay()
    G["bf"].replay()
    S0.wait_stream(S1)
    G["head"].replay()
    S1.wait_stream(S0)
    with torch.cuda.stream(S1):
        G["hb"].replay()
    G["bb"].replay()
    S0.wait_stream(S1)
    G["tail"].replay()


for _ in range(3):
    replay_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    for o in wl.opts():
        o.set_step_hyper(wl.lr(i))
    replay_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
print(f"rays {RAYS}: six graphs on two streams {1e3 * dt:.3f} ms per step, loss {float(st['loss']):.6f}, finite "
      f"{bool(torch.isfinite(hos.human.flat_param).all() and torch.isfinite(hos.model.flat_param).all())}")
