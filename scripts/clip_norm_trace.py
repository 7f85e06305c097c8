"""Per step of a stage-3 training run: the global gradient norm the clip divides by, HIP path vs the reference's op graph (same items and
draws), overall and for the background model alone.  The background norms agree to 1e-4..1e-3; the overall norm -- dominated by the
ill-conditioned pose-decoder / non-rigid head gradients of the human branch -- differs by up to a few per cent on some steps, and through the
ONE shared clip coefficient that is what makes the two NeRF-MLP trajectories part (profiles/r06_stage3_clip_norm_hip_vs_oracle.txt).
  python scripts/clip_norm_trace.py            (HUMAN_FP32=1: the same with the human network pinned to exact fp32 MFMA)"""
import os, sys, json, tempfile, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle.steps as osteps
from tests import test_gpu_convergence as tc, _parity as par
from hosnerf_amd.hosnerf import HOSNeRF
from hosnerf_amd.human_nerf import default_cfg
from hosnerf_amd.train import FusedAdam, GradClip, human_lr_decay, human_lr_ranges, train_step_stage3
dev = torch.device("cuda")
scene, px, rays = tc._make_scene(tempfile.mkdtemp(), dev)
_, bsd, m = tc._train_stage1(rays, dev, oracle=False); del m
_, hsd, _ = tc._train_stage2(scene, px, dev, oracle=False)
LR = 6.667e-5 * 0.3
cfg = default_cfg(par.basedir(tc.TRANSITIONS)); cfg.perturb = 1.0
hos = HOSNeRF(cfg); hos.model.load_state_dict(bsd, strict=False); hos.human.load_state_dict(hsd, strict=True); hos = hos.to(dev)
if os.environ.get("HUMAN_FP32") == "1":       # diagnosis: the human network's GEMMs on the exact fp32 MFMA (module-level arithmetic pin)
    from hosnerf_amd import ops
    hos.human.gemm_mode = ops.GEMM_FP32
clip = GradClip(osteps.GRAD_MAX_NORM)
o_b = FusedAdam(hos.model, lr=LR, clip=clip)
o_h = FusedAdam(hos.human, lr=LR, lr_ranges=human_lr_ranges(hos.human, LR, LR / 10.0), clip=clip)
pb, ph, ora_step = osteps.stage3_trainer(bsd, hsd, dev, LR, tc.TRANSITIONS)
norms = []
_orig = torch.nn.utils.clip_grad_norm_
def rec(params, max_norm, *a, **k):
    params = list(params)
    per = {}
    n = _orig(params, max_norm, *a, **k)
    norms.append(float(n))
    return n
torch.nn.utils.clip_grad_norm_ = rec
train_frames = [i for i in range(tc.N_FRAMES) if i not in tc.HELD_OUT]
pm = dict(hos.model.named_parameters())
for step, (it, t_rand, jit) in enumerate(tc._stage3_items(scene, px, dev, train_frames, 30, 41)):
    batch = {k: v for k, v in it.items() if k not in tc.NET_DROP}
    train_step_stage3(hos, o_b, o_h, batch, LR * human_lr_decay(step), jitters=[j.to(dev) for j in jit], t_rand=t_rand)
    nh = float(clip._partials.double().sum().sqrt())
    # per-module HIP norms of this step's gradient (still in the flat buffers)
    nb = float(hos.model.flat_grad.double().norm()); 
    ora_step(it, t_rand, [j.view(-1, 1) for j in jit], human_lr_decay(step))
    nob = sum(float(v.grad.double().pow(2).sum()) for v in pb.values() if v.grad is not None) ** 0.5     # (clipped in place: rescale by coef)
    coef_o = min(osteps.GRAD_MAX_NORM / (norms[-1] + 1e-6), 1.0)
    d = float((pm["mlps.2.pts_linear.3.weight"].detach() - pb["mlps.2.pts_linear.3.weight"].detach()).double().norm())
    print("step %2d time %.4f  norm hip %.6e ora %.6e ratio %.5f   bkgd norm hip %.5e ora %.5e ratio %.5f  |hip-ora| %.3e" %
          (step, it["time"], nh, norms[-1], nh / norms[-1], nb, nob / coef_o, nb / (nob / coef_o), d), flush=True)
