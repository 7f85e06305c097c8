"""Which loss term carries the ill-conditioned part of the human network's gradient, and who is closer to float64?  One stage-3 item on
trained weights; per loss term (MSE / flow / cycle alone) the gradient of selected parameters from the HIP path (human network pinned to
exact fp32 MFMA, so that no 16-bit split is involved) against the oracle evaluated in fp32 AND in float64 on the same device.
Round 6 result (profiles/r06_stage3_gradient_terms_vs_fp64.txt, two runs): on the cycle term the HIP path is 0.4-1.5 % away
from float64 in both runs, the reference's op graph in fp32 0.7-1.5 % in one run (0.004 % from the HIP path) and 2-5 % in the other.
  python scripts/gradient_terms_vs_fp64.py"""
import os, sys, json, tempfile, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle.steps as osteps, oracle.losses as ol
from tests import test_gpu_convergence as tc, _parity as par
from hosnerf_amd import ops
from hosnerf_amd.hosnerf import HOSNeRF
from hosnerf_amd.human_nerf import default_cfg
from hosnerf_amd.train import stage3_losses
dev = torch.device("cuda")
scene, px, rays = tc._make_scene(tempfile.mkdtemp(), dev)
_, bsd, m = tc._train_stage1(rays, dev, oracle=False); del m
_, hsd, _ = tc._train_stage2(scene, px, dev, oracle=False)
cfg = default_cfg(par.basedir(tc.TRANSITIONS)); cfg.perturb = 1.0
train_frames = [i for i in range(tc.N_FRAMES) if i not in tc.HELD_OUT]
items = tc._stage3_items(scene, px, dev, train_frames, 30, 41)
hos = HOSNeRF(cfg); hos.model.load_state_dict(bsd, strict=False); hos.human.load_state_dict(hsd, strict=True); hos = hos.to(dev)
hos.human.gemm_mode = ops.GEMM_FP32
watch = ["non_rigid_mlp.block_mlps.12.weight", "non_rigid_mlp.block_mlps.0.weight", "non_rigid_forward_mlp.block_mlps.12.weight", "pose_decoder.block_mlps_dstT.2.weight",
         "cnl_mlp.pts_linears.0.weight", "mweight_vol_decoder.decoder.block_conv.8.weight"]
it, t_rand, jit = items[26]
batch = {k: v for k, v in it.items() if k not in tc.NET_DROP}
for dt in (torch.float32, torch.float64):
  pb = par.cast(bsd, dev, dt); ph = {k: v.requires_grad_(True) for k, v in par.cast(hsd, dev, dt).items()}
  itc = {k: (v.to(dt) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in it.items()}
  for name, w in (("mse", (0.2, 0.0, 0.0)), ("flow", (0.0, 0.01, 0.0)), ("cycle", (0.0, 0.0, 0.01))):
    hos.zero_grad()
    out = hos.render(batch, randomized=True, is_train=True, static_cycle=True, jitters=[j.to(dev) for j in jit], t_rand=t_rand)
    loss, parts = stage3_losses(out, batch, w_mse=w[0], w_flow=w[1], w_cycle=w[2])
    loss.backward()
    hos.human.store.ensure_bound(); hos.human.scatter_compact_grads()
    for p in ph.values(): p.grad = None
    o = osteps.stage3_render(pb, ph, itc, tc.TRANSITIONS, t_rand=t_rand.to(dt), jitters=[j.view(-1, 1).to(dt) for j in jit])
    lo, po = ol.stage3_losses(o, itc, float(it["time"]), w_mse=w[0], w_flow=w[1], w_cycle=w[2])
    lo.backward()
    hp = dict(hos.human.named_parameters())
    print(str(dt)[6:], name, "loss %.3e %.3e" % (float(loss), float(lo)))
    for n in watch:
        go = ph[n].grad
        if go is None: print("      %-48s oracle None, hip %.3e" % (n, float(hp[n].grad.abs().max()))); continue
        gh = hp[n].grad.reshape(go.shape).double(); go = go.double()
        print("      %-48s |ora| %.3e rel err %.5f" % (n, float(go.norm()), float((gh - go).norm() / max(float(go.norm()), 1e-30))))
