"""Bug-hunting companion of tests/test_gpu_convergence.py: train the HIP path and the reference's op graph side by side on the SAME items
and draws and print, per step, how far selected parameters of the two have drifted apart.  Chaos grows that distance smoothly; a STEP
at which it jumps marks an item on which the two compute a different gradient (this is how the stage-3 `target_patches` difference of
round 6 was found: one step, one patch that left the subject's box).   python scripts/trajectory_split.py {1|2|3} [steps]"""
import os, sys, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.steps as osteps
from tests import test_gpu_convergence as tc, _parity as par
from hosnerf_amd import synth

def main():
    stage, steps = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda")
    scene, px, rays = tc._make_scene(tempfile.mkdtemp(), dev)
    prev = None
    if stage == 1:
        from hosnerf_amd.mipnerf360 import MipNeRF360
        from hosnerf_amd.train import FusedAdam, stage1_loss
        sd0 = synth.background_state_dict(777, 2)
        model = MipNeRF360(par.basedir(tc.TRANSITIONS), opaque_background=True); model.load_state_dict(sd0, strict=False); model = model.to(dev)
        opt = FusedAdam(model, lr=2e-3, max_grad_norm=osteps.GRAD_MAX_NORM)
        p_ora, ora_step = osteps.stage1_trainer(sd0, dev, tc.TRANSITIONS)
        watch = ["mlps.0.pts_linear.1.weight", "mlps.1.pts_linear.2.weight", "mlps.2.pts_linear.3.weight", "mlps.2.bkgd_stateembeds.1"]
        params = dict(model.named_parameters())
        for step, b, jit in tc._stage1_batches(rays, steps, 17):
            lr, frac = tc._stage1_lr(step, steps, tc.S1_LR_SCALE), step / steps
            opt.zero_grad()
            rend, hist = model(b, frac, True, True, 0.1, 1e6, jitters=[j.to(dev) for j in jit])
            loss, _ = stage1_loss(rend[-1]["rgb"], b["target"], hist); loss.backward(); opt.step(lr)
            lo = ora_step(b, lr, frac, [j.view(-1, 1) for j in jit])
            d = [float((params[n].detach() - p_ora[n].detach()).double().norm()) for n in watch]
            prev = report(step, b["times"], float(loss), float(lo), d, prev)
    elif stage == 3:
        from hosnerf_amd.hosnerf import HOSNeRF
        from hosnerf_amd.human_nerf import default_cfg
        from hosnerf_amd.train import FusedAdam, GradClip, human_lr_decay, human_lr_ranges, train_step_stage3
        _, bsd, m = tc._train_stage1(rays, dev, oracle=False)
        del m
        _, hsd, _ = tc._train_stage2(scene, px, dev, oracle=False)
        LR = 6.667e-5 * tc.S3_LR_SCALE
        cfg = default_cfg(par.basedir(tc.TRANSITIONS)); cfg.perturb = 1.0
        hos = HOSNeRF(cfg); hos.model.load_state_dict(bsd, strict=False); hos.human.load_state_dict(hsd, strict=True); hos = hos.to(dev)
        clip = GradClip(osteps.GRAD_MAX_NORM)
        o_b = FusedAdam(hos.model, lr=LR, clip=clip)
        o_h = FusedAdam(hos.human, lr=LR, lr_ranges=human_lr_ranges(hos.human, LR, LR / 10.0), clip=clip)
        pb, ph, ora_step = osteps.stage3_trainer(bsd, hsd, dev, LR, tc.TRANSITIONS)
        train_frames = [i for i in range(tc.N_FRAMES) if i not in tc.HELD_OUT]
        wb, wh = ["mlps.2.pts_linear.3.weight", "mlps.2.rgb_layer.weight"], ["cnl_mlp.pts_linears.4.weight", "non_rigid_mlp.block_mlps.4.weight", "pose_decoder.block_mlps.2.weight"]
        pm, phm = dict(hos.model.named_parameters()), dict(hos.human.named_parameters())
        for step, (it, t_rand, jit) in enumerate(tc._stage3_items(scene, px, dev, train_frames, steps, 41)):
            decay = human_lr_decay(step)
            batch = {k: v for k, v in it.items() if k not in tc.NET_DROP}
            loss, _ = train_step_stage3(hos, o_b, o_h, batch, LR * decay, jitters=[j.to(dev) for j in jit], t_rand=t_rand)
            lo = ora_step(it, t_rand, [j.view(-1, 1) for j in jit], decay)
            d = [float((pm[n].detach() - pb[n].detach()).double().norm()) for n in wb] + \
                [float((phm[n].detach().reshape(ph[n].shape) - ph[n].detach()).double().norm()) for n in wh]
            prev = report(step, it["time"], float(loss), float(lo), d, prev)
    else:
        from hosnerf_amd.human_nerf import Network, default_cfg
        from hosnerf_amd.train import FusedAdam, human_lr_ranges, train_step_stage2
        LR = 6.667e-4 * tc.S2_LR_SCALE
        sd0 = synth.human_state_dict(777, 2)
        cfg = default_cfg(par.basedir(tc.TRANSITIONS)); cfg.perturb = 1.0
        net = Network(cfg, stage=2); net.load_state_dict(sd0, strict=True); net = net.to(dev)
        opt = FusedAdam(net, lr=LR, lr_ranges=human_lr_ranges(net, LR, LR / 10.0), max_grad_norm=osteps.GRAD_MAX_NORM)
        p_ora, ora_step = osteps.stage2_trainer(sd0, dev, LR, tc.TRANSITIONS)
        _, items = tc._stage2_items(scene, px, dev, steps)
        watch = ["cnl_mlp.pts_linears.4.weight", "non_rigid_mlp.block_mlps.4.weight", "non_rigid_forward_mlp.block_mlps.4.weight",
                 "pose_decoder.block_mlps.2.weight", "mweight_vol_decoder.decoder.block_conv.4.weight", "human_stateembeds.1"]
        params = dict(net.named_parameters())
        for step, (it, t_rand) in enumerate(items):
            decay = 0.1 ** (2.0 * step / steps)
            batch = {k: v for k, v in it.items() if k not in tc.NET_DROP}
            loss, _ = train_step_stage2(net, opt, batch, LR * decay, t_rand=t_rand)
            lo = ora_step(it, t_rand, decay)
            d = [float((params[n].detach().reshape(p_ora[n].shape) - p_ora[n].detach()).double().norm()) for n in watch]
            prev = report(step, it["time"], float(loss), float(lo), d, prev)

def report(step, time, lh, lo, d, prev):
    flag = ""
    if prev is not None:
        inc = [a - b for a, b in zip(d, prev["d"])]
        for i, (x, s) in enumerate(zip(inc, prev["inc"])):
            if x > 4.0 * max(s, 1e-7) and x > 0.02 * max(d[i], 1e-9):
                flag += f" JUMP[{i}]"
        sm = [0.8 * s + 0.2 * max(x, 0.0) for s, x in zip(prev["inc"], inc)]
    else:
        sm = [0.0] * len(d)
    rel = abs(lh - lo) / max(abs(lo), 1e-12)
    if flag or rel > 1e-3 or step % 25 == 0:
        print(f"step {step:4d} time {float(time):.4f} loss {lh:.6f} / {lo:.6f} (rel {rel:.1e})  |hip-ora| " + " ".join(f"{x:.2e}" for x in d) + flag, flush=True)
    return {"d": d, "inc": sm}

if __name__ == "__main__":
    main()
