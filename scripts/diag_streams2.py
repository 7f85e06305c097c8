"""Eager two-stream stage-3 steps with a device sync and a full finiteness / determinism audit after every step.
  [HOS_POISON2=1] python scripts/diag_streams2.py [steps] [rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("HOS_POISON2") == "1":
    import soak_poison as sp
    torch.empty, torch.empty_like = sp.pempty, sp.pempty_like
import bench
from hosnerf_amd import ops
from hosnerf_amd.train import stage3_losses

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
w = bench.Stage3(dev, 0, 1, rays)
g = torch.Generator().manual_seed(7)
TR = torch.rand(rays, 128, generator=g).to(dev)
JIT = [torch.rand(rays, generator=g).to(dev) for _ in range(3)]
KEYS = ("rgb", "human_rgbsigma", "pts_mask", "newsmpl_pts", "deform_pts_final", "observe_pts", "deform_pts_prev_final", "human_weights_sorted", "idx_fg", "cycle_count")
mods = {"bkgd": w.hos.model, "human": w.hos.human}
ref = None
def csum(t):
    return float(t.detach().double().sum())
inputs = {"p_bkgd": w.hos.model.flat_param, "p_human": w.hos.human.flat_param, "rays_o_bkg": w.batch["rays_o_bkg"], "radii": w.batch["radii"], "jit0": JIT[0], "TR": TR}
in_ref = {k: csum(v) for k, v in inputs.items()}
for i in range(steps):
    chg = [k for k, v in inputs.items() if csum(v) != in_ref[k]]
    if chg:
        print("INPUT CHANGED before step", i, chg, flush=True)
    w.ob.zero_grad(); w.oh.zero_grad()
    w.hos.human.split_decoder_backward = True
    out = w.hos.render(w.batch, randomized=True, is_train=True, static_cycle=True, jitters=JIT, t_rand=TR)
    torch.cuda.synchronize()
    snap = {k: out[k].detach().float().clone() for k in KEYS if k in out and isinstance(out[k], torch.Tensor)}
    n_cyc = int(out["cycle_count"]) if "cycle_count" in out else None
    for k in ("deform_pts_final", "observe_pts"):
        if n_cyc is not None and k in snap:
            snap[k] = snap[k][:n_cyc]
    for l, h in enumerate(out["ray_history"]):
        for k in ("density", "rgb", "weights", "tdist"):
            if k in h:
                snap[f"bkgd{l}.{k}"] = h[k].detach().float().clone()
    loss, _ = stage3_losses(out, w.batch)
    loss.backward()
    w.hos.human.finish_decoder_backward()
    torch.cuda.synchronize()
    snap["loss"] = loss.detach().float().clone()
    snap["g_bkgd"] = w.hos.model.flat_grad.clone()
    dn = w.hos.human.decoder_span()[1]
    snap["g_decoder"] = w.hos.human.flat_grad[:dn].clone()
    snap["g_human_rest"] = w.hos.human.flat_grad[dn:].clone()
    if ref is None:
        ref = snap
    bad = []
    for k, v in snap.items():
        nn = int((~torch.isfinite(v)).sum())
        d = float((v - ref[k]).abs().max()) if nn == 0 else float("nan")
        sc = float(ref[k].abs().max())
        if nn or d > 1e-4 * max(sc, 1e-30):
            bad.append(f"{k}: nonfinite {nn}/{v.numel()} maxdiff {d:.3e} (scale {sc:.3e})")
    print(f"[two_streams={w.hos.two_streams}] step {i}: " + ("clean" if not bad else " | ".join(bad[:10])), flush=True)
    if bad and i > 0 and os.environ.get("STOP_AT_FIRST", "1") == "1":
        break
