"""Diagnostic for the two-stream stage-3 step under graph replay: which tensor goes wrong first?
  python scripts/diag_streams.py [rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hosnerf_amd import ops
from hosnerf_amd.train import stage3_losses

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
w = bench.Stage3(dev, 0, 1, rays)
g = torch.Generator().manual_seed(7)
TR = torch.rand(rays, 128, generator=g).to(dev)
JIT = [torch.rand(rays, generator=g).to(dev) for _ in range(3)]
KEYS = ("rgb", "human_rgbsigma", "pts_mask", "newsmpl_pts", "deform_pts_final", "observe_pts", "deform_pts_prev_final", "human_weights_sorted")


def run(mode, n=12):
    def fb():
        w.ob.zero_grad(); w.oh.zero_grad()
        w.hos.human.split_decoder_backward = mode == "split"
        if mode == "fwd":
            with torch.no_grad():
                out = w.hos.render(w.batch, randomized=True, is_train=True, static_cycle=True, jitters=JIT, t_rand=TR)
            return out, None
        out = w.hos.render(w.batch, randomized=True, is_train=True, static_cycle=True, jitters=JIT, t_rand=TR)
        loss, _ = stage3_losses(out, w.batch)
        loss.backward()
        if mode == "split":
            w.hos.human.finish_decoder_backward()
        return out, loss.detach()
    for _ in range(2):
        fb()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fb()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, loss = fb()
    hist = out["ray_history"]
    tens = {k: out[k] for k in KEYS if k in out and isinstance(out[k], torch.Tensor)}
    for l, h in enumerate(hist):
        for k in ("density", "rgb", "weights", "tdist"):
            if k in h:
                tens[f"bkgd{l}.{k}"] = h[k]
    if loss is not None:
        tens["loss"] = loss
        tens["g_bkgd"] = w.hos.model.flat_grad
        tens["g_human"] = w.hos.human.flat_grad
    ref = None
    for i in range(n):
        graph.replay()
        snap = {k: v.detach().float().clone() for k, v in tens.items()}
        torch.cuda.synchronize()
        if ref is None:
            ref = snap
            continue
        bad = []
        for k, v in snap.items():
            nn = int((~torch.isfinite(v)).sum())
            d = float((v - ref[k]).abs().max()) if nn == 0 else float("nan")
            sc = float(ref[k].abs().max())
            if nn or d > 1e-5 * max(sc, 1e-30):
                bad.append(f"{k}: nonfinite {nn} maxdiff {d:.3e} (scale {sc:.3e})")
        print(f"[{mode} two_streams={w.hos.two_streams}] replay {i}: " + ("clean" if not bad else "; ".join(bad[:8])))
    del graph


for two in (True, False):
    w.hos.two_streams = two
    for mode in ("fwd", "bwd", "split"):
        run(mode, 6)
