#!/usr/bin/env python
"""Launcher with the reference's command line (`run.py` of each stage directory: 1st_State-Conditional_Scene/run.py:141-156,
3rd_Complete_HOSNeRF/run.py:75-292): `--ginc` / `--ginb` (gin files and bindings), `--scene_name`, `--seed`, `--logbase`,
`--resume_training`, `--ckpt_path`, `--cfg`; `run.model_name` selects the stage (`state_mipnerf360 | state_humanobject |
hosnerf`), `run.max_steps`, `run.grad_max_norm`, `run.bkgd_path` / `run.human_path` (stage-3 warm start, run.py:206-212),
`run.run_train`, `run.run_eval` (S3/run.py:224-231 -> `trainer.test`: the held-out frames rendered by their own cameras, PSNR,
`test_metrics`, M:884-1085) and `run.run_render` (S3/run.py:233-239 -> `trainer.predict`: the free-viewpoint turn about the subject
of `freeview.frame_idx`, `free_view`, M:1293-1494, cameras of core/utils/camera_util.py:106-131), both from `last.ckpt` (stage 3).
The reference's .gin files parse unchanged (hosnerf_amd/gin_lite.py).

What is NOT here: Lightning's Trainer (a plain loop drives `training_step` / `optimizer_step` / checkpoints the way the Trainer
does; DDP = one process per GPU under torch.distributed.run, ray shards + one flat-gradient all-reduce per module), the
datasets and image IO (out of scope -- training items are synthetic unless `--items` points at a torch-saved
list of dataset items with the reference's keys), LPIPS / SSIM.

    python run.py --ginc configs/hosnerf_backpack.gin --scene_name Backpack --logbase logs --ginb "run.max_steps=200"
    python run.py --ginc ... --cpu        # BASELINE configs[0] "CPU plumbing": everything except the kernels, no GPU needed
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--ginc", action="append", help="gin config file")
    p.add_argument("--ginb", action="append", help="gin bindings")
    p.add_argument("--resume_training", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--ckpt_path", type=str, default=None, help="path to checkpoints")
    p.add_argument("--scene_name", type=str, default=None, help="scene name to render")
    p.add_argument("--seed", type=int, default=220901, help="seed to use")
    p.add_argument("--logbase", type=str, default=None, help="folder to save results")
    p.add_argument("--cfg", default=None, type=str, help="yaml of the human-object network (configs/human_nerf/...), optional")
    # additions of this build
    p.add_argument("--cpu", action="store_true", help="plumbing only (BASELINE configs[0]): no kernel is launched, no GPU needed")
    p.add_argument("--items", type=str, default=None, help="torch-saved list of dataset items (reference keys); default: synthetic")
    p.add_argument("--lpips_vgg16", type=str, default=None, help="torchvision's vgg16 checkpoint (vgg16-397923af.pth): with --lpips_lin "
                   "enables the reference's LPIPS term (weight 1.0, configs/default.yaml:97-101) in stages 2 / 3")
    p.add_argument("--lpips_lin", type=str, default=None, help="the reference's third_parties/lpips/weights/v0.1/vgg.pth")
    p.add_argument("--rays", type=int, default=0, help="rays per step and GPU for synthetic items (default: the stage's reference batch)")
    p.add_argument("--scene_dir", type=str, default=None, help="scene directory in the reference's on-disk formats (cameras.pkl / "
                   "poses_bounds.npy, mesh_infos.pkl, canonical_joints.pkl, images/, masks/, images_flow/): training items, evaluation "
                   "frames and free-viewpoint frames are built from it on the device (stages 2 / 3)")
    p.add_argument("--eval_skip", type=int, default=0, help="run_eval: render every eval_skip-th frame of the scene (default: 8 frames spread over the sequence)")
    p.add_argument("--render_frames", type=int, default=100, help="run_render: cameras per free-viewpoint turn (cfg.render_frames)")
    p.add_argument("--render_limit", type=int, default=0, help="run_render: stop after this many cameras of the turn (0 = all)")
    return p.parse_args(argv)


def make_cfg(args, basedir):
    """S3/run.py:33-62: defaults + configs/default.yaml + --cfg merged; here the hot-path fields of those yaml files are the
    defaults of `default_cfg`, and an optional --cfg yaml overrides them."""
    from hosnerf_amd.human_nerf import Cfg, default_cfg
    cfg = default_cfg(basedir)
    if args.cfg:
        import yaml

        def merge(dst, src):
            for k, v in src.items():
                if isinstance(v, dict):
                    node = dst.get(k)
                    if not isinstance(node, dict):
                        node = Cfg()
                        dst[k] = node
                    merge(node, v)
                else:
                    dst[k] = v
        with open(args.cfg) as f:
            merge(cfg, yaml.safe_load(f) or {})
    return cfg


def synthetic_item(model_name: str, rays: int, seed: int, step: int):
    from hosnerf_amd import synth
    from hosnerf_amd.train import prepare_patch_targets
    if model_name == "state_mipnerf360":
        return synth.stage1_batch(rays, seed=seed + step)
    b = synth.human_batch(rays, seed=seed + step, time=((step * 37) % 100) / 99.0, is_train=True, iter_val=float(step))
    return prepare_patch_targets(synth.add_patch_supervision(b, max(1, rays // 1024), 32, seed + step))


def crop_rays_64(seed: int = 777):
    """BASELINE configs[0] / SURVEY 8(d).1: the 64x64 crop = 4096 rays of a pin-hole camera (f = 64 * 1.2) at z = +1 looking
    at the origin, radii from the neighbouring-ray distance * 2 / sqrt(12) (S1/src/data/ray_utils.py:94-108)."""
    H = W = 64
    f = 64 * 1.2
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d = torch.stack([(i + 0.5 - W / 2) / f, -(j + 0.5 - H / 2) / f, -torch.ones_like(i)], -1).reshape(-1, 3)
    o = torch.tensor([0.0, 0.0, 1.0]).expand_as(d).contiguous()
    dn = d.view(H, W, 3)
    dx = torch.sqrt(((dn[:-1] - dn[1:]) ** 2).sum(-1))
    dx = torch.cat([dx, dx[-2:-1]], 0).reshape(-1, 1)
    vd = d / d.norm(dim=-1, keepdim=True)
    g = torch.Generator().manual_seed(seed)
    return {"rays_o": o, "rays_d": d, "viewdirs": vd, "radii": dx * 2 / (12 ** 0.5), "times": torch.full((H * W,), 0.5),
            "target": torch.rand(H * W, 3, generator=g)}


def run(args, gin):
    from hosnerf_amd import select_option
    from hosnerf_amd.train import batch_to_device, prepare_patch_targets
    kw = gin.kwargs("run")
    model_name = kw.get("model_name")
    if model_name is None:
        raise SystemExit("run.model_name is not bound (pass a --ginc file or --ginb 'run.model_name=\"hosnerf\"')")
    dataset_name = kw.get("dataset_name", "synthetic")
    scene = args.scene_name or "scene"
    exp_name = f"{model_name}_{dataset_name}_{scene}_{str(args.seed).zfill(3)}"          # S3/run.py:108-110
    if kw.get("postfix"):
        exp_name += "_" + kw["postfix"]
    logdir = os.path.join(args.logbase or "logs", exp_name)
    os.makedirs(logdir, exist_ok=True)
    basedir = kw.get("datadir") if (kw.get("datadir") and os.path.isdir(str(kw.get("datadir")))) else logdir
    if not os.path.exists(os.path.join(basedir, "transitions_times.json")):
        with open(os.path.join(logdir, "transitions_times.json"), "w") as f:       # one transition -> two states, like Backpack
            json.dump({"f0": {"time": 0.4}}, f)
        basedir = logdir
    torch.manual_seed(args.seed)                                                         # seed_everything (run.py:155)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # `run.grad_max_norm` -> Trainer(gradient_clip_val=..., algorithm "norm") in all three launchers (S1/run.py:155,
    # 2nd_.../run.py:185-186, 3rd_.../run.py:188-189); every Backpack.gin binds 0.001
    model_kw = {"grad_max_norm": float(kw.get("grad_max_norm", 0.0))}
    if str(kw.get("grad_clip_algorithm", "norm")) != "norm":
        raise SystemExit("run.grad_clip_algorithm: only 'norm' (the reference's default and the one its configs use) is implemented")
    if model_name == "state_mipnerf360":
        model_kw.update(max_steps=int(kw.get("max_steps", 500000)))
        for k in ("lr_init", "lr_final", "lr_delay_steps", "lr_delay_mult"):
            if gin.get_param(f"LitMipNeRF360.{k}") is not None:
                model_kw[k] = gin.get_param(f"LitMipNeRF360.{k}")
        for side, name in (("near", "LitDataNeRF360V2.near"), ("far", "LitDataNeRF360V2.far")):
            if gin.get_param(name) is not None:
                model_kw[side] = float(gin.get_param(name))
    else:
        model_kw["cfg"] = make_cfg(args, basedir)
    lit = select_option.select_model(model_name, basedir, **model_kw)
    n_params = sum(p.numel() for p in lit.parameters())
    if model_name == "hosnerf":                                                          # warm start, S3/run.py:206-212
        for key in ("human_path", "bkgd_path"):
            path = kw.get(key)
            if path and os.path.exists(str(path)):
                missing, unexpected = select_option.load_checkpoint(lit, path, strict=False)
                print(f"[run] {key}: loaded {path} ({len(missing)} missing, {len(unexpected)} unexpected keys)")
            elif path and args.items:
                # the reference fails in pl_load when a bound warm-start file is missing; training a real scene's stage 3
                # from random initialisation is never what was asked for
                raise SystemExit(f"run.{key} = {path!r} does not exist")
            elif path:
                print(f"[run] WARNING: run.{key} = {path!r} does not exist -- stage 3 starts from RANDOM initialisation "
                      f"(tolerated for synthetic items only)", file=sys.stderr)
    opt = lit.configure_optimizers()
    ckpt = args.ckpt_path or os.path.join(logdir, "last.ckpt")
    step0 = 0
    if args.resume_training and os.path.exists(ckpt):
        select_option.load_checkpoint(lit, ckpt, strict=True)
        step0 = select_option.load_optimizer_states(ckpt, opt)
        print(f"[run] resumed from {ckpt} at step {step0}")
    max_steps = int(kw.get("max_steps", 0))
    default_rays = {"state_mipnerf360": int(gin.get_param("LitData.batch_size", 4096)) // max(world, 1), "state_humanobject": 2048, "hosnerf": 2048}
    rays = args.rays or default_rays[model_name]
    print(f"[run] {exp_name}: {model_name}, {n_params / 1e6:.2f} M parameters, {rays} rays/step/GPU, world {world}, max_steps {max_steps}, logdir {logdir}")

    if args.cpu:
        # BASELINE configs[0] "CPU PyTorch single process (plumbing, no GPU)": flags, gin, module construction under the
        # reference's names, optimiser + schedule, a 64x64-crop ray batch, checkpoint round trip.  The renderer itself has no
        # CPU implementation (by design: hosnerf_amd never falls back); its CPU restatement is the test oracle (tests/).
        crop = crop_rays_64(args.seed)
        lit._step = step0
        lit.apply_lr(opt, step0)
        select_option.save_checkpoint(lit, ckpt, global_step=step0, optimizer=opt)
        missing, unexpected = select_option.load_checkpoint(lit, ckpt, strict=True)
        plan = {"mode": "cpu-plumbing", "exp_name": exp_name, "model_name": model_name, "parameters": n_params, "state_dict_keys": len(lit.state_dict()),
                "crop_rays": int(crop["rays_o"].shape[0]), "samples_per_ray": "64/64/32 background, 128 human", "lr": [g["lr"] for g in opt.param_groups],
                "checkpoint": ckpt, "checkpoint_roundtrip": {"missing": len(missing), "unexpected": len(unexpected)}, "max_steps": max_steps,
                "gin": {k: v for k, v in sorted(gin.items())}}
        print(json.dumps(plan))
        return plan

    if not torch.cuda.is_available():
        raise SystemExit("run.py needs an MI355X for training / rendering (there is no CPU renderer); use --cpu for the plumbing check")
    # HOS_BENCH_ONE_GPU=1 (testing only, as in bench.py): all ranks share cuda:0 and exchange through gloo, so that the multi-rank
    # launcher path -- shards of rays / frames, gradient exchange, sharded decoder, checkpoint of rank 0 -- runs on a one-GPU box
    one_gpu = os.environ.get("HOS_BENCH_ONE_GPU") == "1"
    local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lit = lit.to(dev)
    if world > 1 and bool(kw.get("shard_decoder", False)) and hasattr(lit, "human"):
        # `run.shard_decoder = True` (this build's addition; the reference replicates the module under DDP): the volume decoder's
        # first three layers sharded over the ranks (Network.shard_decoder); the optimiser is rebuilt on the remaining active spans
        from hosnerf_amd.train import ShardComm
        lit.human.shard_decoder(ShardComm(rank, world))
        opt = lit.configure_optimizers()
        if args.resume_training and os.path.exists(ckpt):
            select_option.load_optimizer_states(ckpt, opt)
        print(f"[run] volume decoder sharded over {world} ranks: this rank owns {sum(n for _, n in lit.human.decoder_shard_spans()) / 1e6:.2f} M of its parameters")
    if args.lpips_vgg16 or args.lpips_lin:
        if not (args.lpips_vgg16 and args.lpips_lin) or not hasattr(lit, "human"):
            raise SystemExit("--lpips_vgg16 and --lpips_lin go together and apply to the stages that own the human-object network")
        from hosnerf_amd.lpips import LPIPS
        lit.lpips = LPIPS.from_files(args.lpips_vgg16, args.lpips_lin, dev)          # M:582-584
        print(f"[run] LPIPS term enabled (weight 1.0): {args.lpips_vgg16}, {args.lpips_lin}")
    items = torch.load(args.items, weights_only=False) if args.items else None
    if items and model_name == "hosnerf":
        # The stage-3 reference unpacks the rendered rays with a PLAIN reshape (`_unpack_imgs`, S3 model.py:41-50: patches are never
        # cut by the box in stage 3, core/data/human_nerf/train.py:322-330), for the MSE and for the LPIPS term alike: an item whose
        # `patch_masks` has holes is a stage-2 item and would be rendered against the wrong pixels -- refuse it here, once, on the host
        for n, it in enumerate(items):
            if "patch_masks" in it and not bool(torch.as_tensor(it["patch_masks"]).all()):
                raise SystemExit(f"--items[{n}]: patch_masks has holes; stage 3 takes whole patches (the reference's stage-3 _unpack_imgs is a reshape)")
    scene = None
    if args.scene_dir:
        if model_name == "state_mipnerf360":
            raise SystemExit("--scene_dir builds human-object items (stages 2 / 3); stage 1 takes --items or synthetic rays")
        from hosnerf_amd import formats
        from hosnerf_amd.dataset import SceneItems
        from hosnerf_amd.freeview import load_scene_pixels
        px = load_scene_pixels(args.scene_dir)
        if not os.path.exists(os.path.join(args.scene_dir, "cameras_scaleworld.pkl")):     # what the stage-1 loader leaves behind
            formats.load_scene(args.scene_dir, px["images"].shape[1:3], masks=px["alphas"], near=0.1, far=1e6)
        scene = SceneItems(args.scene_dir, px["images"], px["alphas"], px["flows"], frames=px["frames"], device=dev, seed=args.seed + rank,
                           stage=2 if model_name == "state_humanobject" else 3,
                           n_patches=int(model_kw["cfg"].patch.N_patches), patch_size=int(model_kw["cfg"].patch.size),
                           sample_subject_ratio=float(model_kw["cfg"].patch.sample_subject_ratio), bbox_offset=float(model_kw["cfg"].bbox_offset),
                           resize_img_scale=float(model_kw["cfg"].resize_img_scale))
        print(f"[run] scene {args.scene_dir}: {len(scene)} frames of {px['images'].shape[2]}x{px['images'].shape[1]}")
    run_train = bool(kw.get("run_train", True))
    log_every = int(kw.get("log_every_n_steps", 100))
    t0 = time.perf_counter()
    lit._step = step0
    # `last.ckpt` every `run.save_every_n_steps` steps (default 5000; the reference's ModelCheckpoint writes per validation
    # epoch), written to a temporary name and renamed so that a crash during the write never leaves a truncated file
    save_every = int(kw.get("save_every_n_steps", 5000)) if bool(kw.get("save_last", True)) else 0

    def save_last(global_step):
        tmp = ckpt + ".tmp"
        select_option.save_checkpoint(lit, tmp, global_step=global_step, optimizer=opt)
        os.replace(tmp, ckpt)

    if run_train:
        for step in range(step0, max_steps):
            if scene is not None:
                item = scene[int(torch.randint(len(scene), (1,)))] if len(scene) > 1 else scene[0]      # shuffled frames (DataLoader(shuffle=True))
            else:
                item = items[step % len(items)] if items else synthetic_item(model_name, rays, args.seed + 1000 * rank, step)
            if items and model_name != "state_mipnerf360" and "mse_count" not in item:
                item = prepare_patch_targets(item)          # items straight from the reference's dataset: derive the patch-MSE constants
            batch = batch_to_device(item, dev) if model_name != "state_mipnerf360" else {k: v.to(dev) for k, v in item.items()}
            opt.zero_grad()
            loss = lit.training_step(batch, step)
            lit.backward(loss)           # human stages: volume decoder reduced at its 3.5 MB output gradient (train.backward_human)
            lit.optimizer_step(0, step, opt)
            if save_every > 0 and (step + 1) % save_every == 0 and step + 1 < max_steps:
                if getattr(getattr(lit, "human", None), "decoder_shard", None) is not None:
                    lit.human.gather_decoder_shards(opt)        # collective: every rank's flat buffer is complete before rank 0 writes it
                if rank == 0:
                    save_last(step + 1)
            if (step + 1) % log_every == 0 and rank == 0:
                dt = time.perf_counter() - t0
                print(f"[run] step {step + 1}/{max_steps} loss {float(loss):.5f} lr {opt.param_groups[0]['lr']:.3e} "
                      f"{(step + 1 - step0) * rays * world / dt:.0f} rays/s")
        if getattr(getattr(lit, "human", None), "decoder_shard", None) is not None:
            lit.human.gather_decoder_shards(opt)
        if rank == 0 and bool(kw.get("save_last", True)):
            save_last(max_steps)
            print(f"[run] wrote {ckpt}")
        if world > 1:
            import torch.distributed as dist
            dist.barrier()              # the other ranks' evaluation reads the checkpoint rank 0 has just written
    result = {"exp_name": exp_name, "checkpoint": ckpt}
    run_eval, run_render = bool(kw.get("run_eval", False)), bool(kw.get("run_render", False))
    if run_eval or run_render:
        result.update(evaluate_and_render(args, kw, lit, model_name, scene, ckpt, logdir, dev, rank, world, run_eval, run_render))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return result


def evaluate_and_render(args, kw, lit, model_name, scene, ckpt, logdir, dev, rank, world, run_eval: bool, run_render: bool):
    """`trainer.test` / `trainer.predict` of the stage-3 launcher (S3/run.py:224-239) from `last.ckpt`: `test_metrics` over held-out
    frames (PSNR against the frame's pixels, images under <logdir>/test_vis) and `free_view` over the orbit cameras of
    `freeview.frame_idx` (images under <logdir>/freeview_vis_newtrans/view_<idx>), every frame through `eval.render_frame`; with
    several ranks each frame's rays are split over the group.  Writes <logdir>/results.json."""
    from hosnerf_amd import eval as ev, select_option
    from hosnerf_amd.freeview import save_image
    if model_name != "hosnerf":
        raise SystemExit("run.run_eval / run.run_render: full-frame rendering is the stage-3 (`hosnerf`) launcher's; stages 1 / 2 report their training loss only")
    if scene is None:
        raise SystemExit("run.run_eval / run.run_render need --scene_dir (frames, cameras and SMPL fits to render)")
    if not os.path.exists(ckpt):
        raise SystemExit(f"run.run_eval / run.run_render: {ckpt} does not exist (train first, or pass --ckpt_path)")
    select_option.load_checkpoint(lit, ckpt, strict=True)
    hos = lit.net                                   # the composite renderer that owns `model` and `human`
    group = None
    if world > 1:
        import torch.distributed as dist
        group = dist.group.WORLD
    chunk = int(lit.cfg.chunk_bkg)
    bgc = tuple(float(c) for c in lit.cfg.bgcolor)
    out = {}
    if run_eval:
        n = len(scene)
        idxs = list(range(0, n, args.eval_skip)) if args.eval_skip > 0 else sorted({int(round(i * (n - 1) / 7.0)) for i in range(8)} if n > 1 else {0})
        psnrs = {}
        for i in idxs:
            fr = scene.eval_frame(i, bgcolor=bgc)
            rendered = ev.render_frame(hos, fr, chunk_bkg=chunk, randomized=False, group=group)
            psnrs[fr["frame_name"]] = ev.psnr_metric(rendered, ev.truth_frame(fr))
            if rank == 0:
                save_image(os.path.join(logdir, "test_vis", fr["frame_name"] + ".png"), rendered, int(fr["img_height"]), int(fr["img_width"]))
        out["test"] = {"psnr": float(sum(psnrs.values()) / len(psnrs)), "frames": psnrs}
        if rank == 0:
            print(f"[run] Test PSNR is {out['test']['psnr']:.4f} over {len(psnrs)} frames")
    if run_render:
        fidx = min(int(lit.cfg.freeview.frame_idx), len(scene) - 1)
        total = int(args.render_frames)
        count = min(total, args.render_limit) if args.render_limit > 0 else total
        psnrs = []
        for k in range(count):
            fr = scene.freeview_frame(fidx, k, total, bgcolor=bgc)
            rendered = ev.render_frame(hos, fr, chunk_bkg=chunk, randomized=False, group=group)
            psnrs.append(ev.psnr_metric(rendered, ev.truth_frame(fr)))          # the reference reports it against the training frame too (M:1462)
            if rank == 0:
                save_image(os.path.join(logdir, "freeview_vis_newtrans", f"view_{fidx:05d}", f"image-{k:05d}.jpg"), rendered,
                           int(fr["img_height"]), int(fr["img_width"]))
        out["freeview"] = {"frame_idx": fidx, "frames": count, "of": total, "psnr_vs_training_frame": float(sum(psnrs) / len(psnrs))}
        if rank == 0:
            print(f"[run] Freeview: {count} of {total} cameras about frame {fidx} written to {os.path.join(logdir, 'freeview_vis_newtrans')}")
    if rank == 0:
        with open(os.path.join(logdir, "results.json"), "w") as f:
            json.dump(out, f, indent=1)
    return {"results": out}


def main(argv=None):
    from hosnerf_amd import gin_lite
    args = parse_args(argv)
    gin = gin_lite.parse_config_files_and_bindings(args.ginc, args.ginb)
    return run(args, gin)


if __name__ == "__main__":
    main()
