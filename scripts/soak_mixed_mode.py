"""Determinism + NaN-poison soak of the MIXED arithmetic mode (human network pinned to exact fp32 MFMA by `module.gemm_mode`, background on\nthe planes path, two streams): the same stage-3 forward + backward repeated REPS times per item; any repetition whose gradient norms leave the\nfirst one's by more than atomics-order noise is an EVENT.  Round 6: 0 events in 6 x 60 repetitions.   python scripts/soak_mixed_mode.py"""
import os, sys, json, tempfile, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import scripts.soak_poison as sp
torch.empty, torch.empty_like = sp.pempty, sp.pempty_like          # NaN-poisoned product allocations
from tests import test_gpu_convergence as tc, _parity as par
from hosnerf_amd import ops
from hosnerf_amd.hosnerf import HOSNeRF
from hosnerf_amd.human_nerf import default_cfg
from hosnerf_amd.train import stage3_losses
dev = torch.device("cuda")
scene, px, rays = tc._make_scene(tempfile.mkdtemp(), dev)
_, bsd, m = tc._train_stage1(rays, dev, steps=100, oracle=False); del m
_, hsd, _ = tc._train_stage2(scene, px, dev, steps=100, oracle=False)
cfg = default_cfg(par.basedir(tc.TRANSITIONS)); cfg.perturb = 1.0
hos = HOSNeRF(cfg); hos.model.load_state_dict(bsd, strict=False); hos.human.load_state_dict(hsd, strict=True); hos = hos.to(dev)
hos.human.gemm_mode = ops.GEMM_FP32
train_frames = [i for i in range(tc.N_FRAMES) if i not in tc.HELD_OUT]
items = tc._stage3_items(scene, px, dev, train_frames, 6, 41)
bad = 0
for it, t_rand, jit in items:
    batch = {k: v for k, v in it.items() if k not in tc.NET_DROP}
    ref = None
    for rep in range(int(os.environ.get("REPS", "60"))):
        hos.zero_grad()
        out = hos.render(batch, randomized=True, is_train=True, static_cycle=True, jitters=[j.to(dev) for j in jit], t_rand=t_rand)
        loss, _ = stage3_losses(out, batch)
        loss.backward()
        hos.model.store.ensure_bound(); hos.human.store.ensure_bound()
        torch.cuda.synchronize()
        nb = float(hos.model.flat_grad.double().norm())
        nh = sum(float(hos.human.flat_grad[o:o + n].double().pow(2).sum()) for o, n in hos.human.store.active_spans()) ** 0.5
        if ref is None:
            ref = (nb, nh)
        if not (abs(nb - ref[0]) <= 2e-3 * ref[0] and abs(nh - ref[1]) <= 2e-2 * ref[1]):
            bad += 1
            print("EVENT time %.4f rep %d: bkgd %.6e (ref %.6e) human %.6e (ref %.6e) loss %.7f" % (it["time"], rep, nb, ref[0], nh, ref[1], float(loss)), flush=True)
    print("item time %.4f: reference norms bkgd %.6e human %.6e" % (it["time"], ref[0], ref[1]), flush=True)
print("events:", bad)
