"""SURVEY 8(f).4 end to end on the device: a synthetic scene DIRECTORY in the reference's on-disk formats
-> `formats.load_scene` (writes cameras_scaleworld.pkl, like the stage-1 loader) -> `dataset.SceneItems` (mesh_infos.pkl,
canonical_joints.pkl, device-side rays / box test / patch gather) -> `run.py --items` (stage 3, real optimiser steps)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

APPENDIX_B = {"rays", "near", "far", "rays_o_bkg", "rays_d_bkg", "viewdirs_bkg", "radii", "newsmpl_to_scale_world", "dst_Rs", "dst_Ts",
              "cnl_gtfms", "canonical_joints", "motion_weights_priors", "cnl_bbox_min_xyz", "cnl_bbox_max_xyz", "cnl_bbox_scale_xyz",
              "dst_posevec", "bgcolor", "time", "is_train", "patch_masks", "target_patches", "patch_div_indices", "target_rgbs",
              "ray_mask", "ray_mask_bkg", "img_width", "img_height", "frame_name"}
FLOW_KEYS = {"dst_Rs_prev", "dst_Ts_prev", "dst_posevec_prev", "newsmpl_to_camera_prev", "intrinsics_prev", "ray_grid"}


def test_scene_directory_to_training_steps(tmp_path):
    from hosnerf_amd import formats, synth
    from hosnerf_amd.dataset import SceneItems
    dev = torch.device("cuda")
    scene = str(tmp_path / "scene")
    H = W = 96
    px = synth.write_scene_dir(scene, 16, H, W, seed=3)
    s = formats.load_scene(scene, (H, W), masks=px["alphas"], near=0.1, far=1e6)
    assert os.path.exists(os.path.join(scene, "cameras_scaleworld.pkl")) and s["bkgrays_sizes"].shape == (16,)
    ds = SceneItems(scene, px["images"], px["alphas"], px["flows"], n_patches=2, patch_size=16, device=dev, seed=5)
    assert len(ds) == 16
    first, item = ds[0], ds[7]
    assert APPENDIX_B <= set(item) and FLOW_KEYS <= set(item), sorted(APPENDIX_B | FLOW_KEYS - set(item))
    assert not (FLOW_KEYS & set(first)) and first["time"] == 0.0                  # frame 0 has no previous frame (time <= 0.005)
    n = 2 * 16 * 16
    assert item["rays"].shape == (2, n, 3) and item["near"].shape == (n, 1) and item["ray_grid"].shape == (n, 5)
    assert item["target_patches"].shape == (2, 16, 16, 3) and bool(item["patch_masks"].all())
    assert item["motion_weights_priors"].shape == (27, 32, 32, 32) and item["dst_posevec"].shape == (75,)
    assert 0 < int(item["ray_mask"].sum()) < H * W, "the subject's box must cover part of the frame"
    assert torch.equal(item["target_rgbs"], item["target_patches"].reshape(-1, 3))
    # the two ray sets describe the SAME pixels: a point on the body-frame ray, carried into the scaled world by the item's
    # similarity, lies on the background ray of that pixel (this is what the stage-3 z-merge, C1, relies on)
    o, d = item["rays"][0].double(), item["rays"][1].double()
    A = item["newsmpl_to_scale_world"].double()
    for z in (float(item["near"].mean()), float(item["far"].mean())):
        pw = (o + z * d) @ A[:3, :3].T + A[:3, 3]
        rel = pw - item["rays_o_bkg"].double()
        dirb = item["rays_d_bkg"].double()
        t = (rel * dirb).sum(-1, keepdim=True) / (dirb * dirb).sum(-1, keepdim=True)
        off = (rel - t * dirb).norm(dim=-1) / rel.norm(dim=-1)
        assert float(off.max()) < 1e-4, float(off.max())
    assert float((item["far"] - item["near"]).min()) > 0 and float(item["near"].min()) > 0
    # items -> the launcher: three real stage-3 optimiser steps on these items (clip, both Adams), checkpoint written
    items = [ds[i] for i in (3, 7, 11)]
    path = str(tmp_path / "items.pt")
    torch.save(items, path)
    cmd = [sys.executable, os.path.join(ROOT, "run.py"), "--ginc", os.path.join(ROOT, "configs", "hosnerf_backpack.gin"), "--ginb", "run.max_steps=3",
           "--ginb", "run.log_every_n_steps=1", "--ginb", f'run.datadir="{scene}"', "--ginb", 'run.human_path=""', "--ginb", 'run.bkgd_path=""',
           "--logbase", str(tmp_path / "logs"), "--scene_name", "synthetic", "--items", path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("[run] step")]
    assert len(lines) == 3 and all(np.isfinite(float(l.split("loss")[1].split()[0])) for l in lines), r.stdout[-2000:]
    assert "wrote" in r.stdout


S2_KEYS = {"rays", "near", "far", "dst_Rs", "dst_Ts", "cnl_gtfms", "canonical_joints", "motion_weights_priors", "cnl_bbox_min_xyz",
           "cnl_bbox_max_xyz", "cnl_bbox_scale_xyz", "dst_posevec", "bgcolor", "time", "is_train", "patch_masks", "target_patches",
           "patch_div_indices", "target_rgbs", "ray_mask", "img_width", "img_height", "frame_name"}


def test_scene_directory_to_stage2_items_and_steps(tmp_path):
    """`SceneItems(stage=2)`: the items of the stage-2 dataset (S2 core/data/human_nerf/train.py:460-658) -- frame composited over
    the item's background colour, the subject's rays only, patches CUT by the box (ragged selection) -- and two real stage-2
    optimiser steps on them (in-network composite, patch MSE with the cut pixels as background colour, flow + cycle terms)."""
    import json
    from hosnerf_amd import formats, synth
    from hosnerf_amd.dataset import SceneItems
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import FusedAdam, human_lr_ranges, train_step_stage2
    dev = torch.device("cuda")
    scene = str(tmp_path / "scene")
    H = W = 96
    px = synth.write_scene_dir(scene, 16, H, W, seed=3)
    formats.load_scene(scene, (H, W), masks=px["alphas"], near=0.1, far=1e6)
    ds = SceneItems(scene, px["images"], px["alphas"], px["flows"], n_patches=6, patch_size=20, sample_subject_ratio=0.5, device=dev,
                    seed=9, stage=2)
    cut_items = 0
    items = []
    for i in (2, 5, 9, 12, 14):
        it = ds[i]
        assert S2_KEYS <= set(it) and FLOW_KEYS <= set(it), sorted((S2_KEYS | FLOW_KEYS) - set(it))
        assert not ({"rays_o_bkg", "newsmpl_to_scale_world", "radii"} & set(it))
        n = int(it["patch_masks"].sum())
        assert it["rays"].shape == (2, n, 3) and it["near"].shape == (n, 1) and it["ray_grid"].shape == (n, 5)
        assert int(it["patch_div_indices"][-1]) == n and it["target_patches"].shape == (6, 20, 20, 3)
        # the selected rays ARE the patch pixels that hit the box, in patch order: their colours are the unmasked patch pixels
        assert torch.equal(it["target_rgbs"], it["target_patches"][it["patch_masks"]])
        # the composite: pixels outside the silhouette carry the item's background colour
        bgc = it["bgcolor"] / 255.0
        assert it["mse_count"] == 6 * 20 * 20 * 3
        want = float((((bgc.expand(it["target_patches"].shape) - it["target_patches"]) ** 2)[~it["patch_masks"]]).sum())
        assert abs(it["mse_const"] - want) <= 1e-5 * max(1.0, want)
        cut_items += int(not bool(it["patch_masks"].all()))
        items.append(it)
    assert cut_items > 0, "no patch of the sample was cut by the subject's box"
    d = str(tmp_path / "base")
    os.makedirs(d)
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    net = Network(default_cfg(d), stage=2)
    net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    net = net.to(dev)
    opt = FusedAdam(net, lr=1e-4, lr_ranges=human_lr_ranges(net, 1e-4, 1e-5))
    for it in items[:2]:
        batch = {k: v for k, v in it.items() if k not in ("frame_name", "ray_mask", "img_width", "img_height", "patch_div_indices")}
        loss, parts = train_step_stage2(net, opt, batch, 1e-4)
        torch.cuda.synchronize()
        assert torch.isfinite(loss) and all(torch.isfinite(v) for v in parts.values())
    assert torch.isfinite(net.flat_param).all()


def _line_offset(item, z):
    """Relative distance of the body-frame ray point at depth z, carried into the scaled world, from the background ray of its pixel."""
    o, d = item["rays"][0].double(), item["rays"][1].double()
    A = item["newsmpl_to_scale_world"].double()
    pw = (o + z * d) @ A[:3, :3].T + A[:3, 3]
    rel = pw - item["rays_o_bkg"].double()
    dirb = item["rays_d_bkg"].double()
    t = (rel * dirb).sum(-1, keepdim=True) / (dirb * dirb).sum(-1, keepdim=True)
    return float(((rel - t * dirb).norm(dim=-1) / rel.norm(dim=-1)).max())


def test_freeview_and_eval_frames_are_consistent(tmp_path):
    """`SceneItems.eval_frame` / `freeview_frame` (freeview.py:199-337): camera 0 of the turn IS the frame's own camera; for every
    camera of the turn the human-branch rays and the background rays describe the same pixels (what the z-merge relies on), the box
    stays in view, and the camera keeps its distance to the subject."""
    from hosnerf_amd import formats, synth
    from hosnerf_amd.dataset import SceneItems
    dev = torch.device("cuda")
    scene = str(tmp_path / "scene")
    H = W = 64
    px = synth.write_scene_dir(scene, 8, H, W, seed=4)
    formats.load_scene(scene, (H, W), masks=px["alphas"], near=0.1, far=1e6)
    ds = SceneItems(scene, px["images"], px["alphas"], px["flows"], n_patches=2, patch_size=16, device=dev, seed=5)
    ev = ds.eval_frame(5)
    f0 = ds.freeview_frame(5, 0, 40)
    for k in ("rays", "near", "far", "rays_o_bkg", "rays_d_bkg", "radii", "ray_mask", "newsmpl_to_scale_world", "target_rgbs"):
        assert torch.allclose(ev[k].float(), f0[k].float(), atol=1e-5), k
    assert ev["is_train"] is False and ev["time"] == float(ds.times[5]) and int(ev["ray_mask"].sum()) + int(ev["ray_mask_bkg"].sum()) == H * W
    n0 = int(f0["ray_mask"].sum())
    c0 = None
    for k in (0, 7, 20, 33):
        fr = ds.freeview_frame(5, k, 40)
        n = int(fr["ray_mask"].sum())
        assert 0.3 * n0 < n < 3.0 * n0, (k, n, n0)                           # the subject's box stays in the picture
        for z in (float(fr["near"].mean()), float(fr["far"].mean())):
            assert _line_offset(fr, z) < 1e-4, (k, z)
        A = fr["newsmpl_to_scale_world"].double()
        cam_world = fr["rays_o_bkg"][0].double()                              # all background rays start at the camera centre
        subj_world = A[:3, 3]                                                 # the body frame's origin (= Th in SMPL space) in the scaled world
        dist = float((cam_world - subj_world).norm())
        c0 = dist if c0 is None else c0
        assert abs(dist - c0) < 1e-4 * c0, (k, dist, c0)


def test_launcher_trains_evaluates_and_renders_from_a_scene_directory(tmp_path):
    """run.py end to end on a scene DIRECTORY (images/, masks/, images_flow/ decoded by the launcher): two stage-3 optimiser steps,
    `run.run_eval` -> one held-out frame + PSNR, `run.run_render` -> two cameras of the free-viewpoint turn, all from last.ckpt."""
    import json
    from hosnerf_amd import synth
    from hosnerf_amd.freeview import write_scene_pixels
    scene = str(tmp_path / "scene")
    H = W = 64
    px = synth.write_scene_dir(scene, 6, H, W, seed=9)
    write_scene_pixels(scene, px)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("patch:\n  N_patches: 2\n  size: 16\nfreeview:\n  frame_idx: 3\n")
    logs = str(tmp_path / "logs")
    cmd = [sys.executable, os.path.join(ROOT, "run.py"), "--ginc", os.path.join(ROOT, "configs", "hosnerf_backpack.gin"),
           "--ginb", "run.max_steps=2", "--ginb", "run.log_every_n_steps=1", "--ginb", f'run.datadir="{scene}"', "--ginb", 'run.human_path=""',
           "--ginb", 'run.bkgd_path=""', "--ginb", "run.run_eval=True", "--ginb", "run.run_render=True", "--logbase", logs,
           "--scene_name", "synthetic", "--scene_dir", scene, "--cfg", str(cfg), "--eval_skip", "100", "--render_frames", "40", "--render_limit", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-4000:])
    assert len([l for l in r.stdout.splitlines() if l.startswith("[run] step")]) == 2 and "Test PSNR" in r.stdout and "Freeview" in r.stdout
    logdir = [os.path.join(logs, d) for d in os.listdir(logs)][0]
    res = json.load(open(os.path.join(logdir, "results.json")))
    assert list(res["test"]["frames"]) == ["frame_000000"] and np.isfinite(res["test"]["psnr"]) and 0.0 < res["test"]["psnr"] < 60.0
    assert res["freeview"] == {"frame_idx": 3, "frames": 2, "of": 40, "psnr_vs_training_frame": res["freeview"]["psnr_vs_training_frame"]}
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(logdir, "test_vis", "frame_000000.png")))
    assert img.shape == (H, W, 3) and img.std() > 0
    for k in (0, 1):
        assert os.path.exists(os.path.join(logdir, "freeview_vis_newtrans", "view_00003", f"image-{k:05d}.jpg"))
    # the modes also run WITHOUT training, from the checkpoint the first call wrote
    cmd2 = cmd + ["--ginb", "run.run_train=False", "--ginb", "run.run_render=False"]          # later bindings win
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r2.returncode == 0 and "[run] step" not in r2.stdout and "Test PSNR" in r2.stdout, (r2.stdout[-2000:], r2.stderr[-3000:])
    res2 = json.load(open(os.path.join(logdir, "results.json")))
    assert abs(res2["test"]["psnr"] - res["test"]["psnr"]) < 1e-3 and "freeview" not in res2


@pytest.mark.parametrize("shard", [False, True], ids=["replicated_decoder", "sharded_decoder"])
def test_launcher_two_ranks_on_one_gpu(tmp_path, shard):
    """The launcher's multi-rank path (S3/run.py:173-190: DDP) executed as TWO processes on the one GPU (HOS_BENCH_ONE_GPU=1: gloo
    transport, testing only): each rank builds its own items from the scene directory, three stage-3 optimiser steps with the
    gradient exchange (and, `run.shard_decoder = True`, the volume decoder sharded over the ranks: per-layer collectives in the forward,
    norm completion, shards gathered before rank 0 writes the checkpoint), then `run.run_eval` with the frame's rays split over the
    ranks.  The checkpoint rank 0 wrote loads with strict keys and holds finite, fully populated decoder weights."""
    import json
    import torch
    from hosnerf_amd import synth
    from hosnerf_amd.freeview import write_scene_pixels
    scene = str(tmp_path / "scene")
    H = W = 64
    px = synth.write_scene_dir(scene, 6, H, W, seed=9)
    write_scene_pixels(scene, px)
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text("patch:\n  N_patches: 2\n  size: 16\nfreeview:\n  frame_idx: 3\n")
    logs = str(tmp_path / "logs")
    e = dict(os.environ)
    e.update(HOS_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561" if shard else "29560", os.path.join(ROOT, "run.py"),
           "--ginc", os.path.join(ROOT, "configs", "hosnerf_backpack.gin"),
           "--ginb", "run.max_steps=3", "--ginb", "run.log_every_n_steps=1", "--ginb", f'run.datadir="{scene}"', "--ginb", 'run.human_path=""',
           "--ginb", 'run.bkgd_path=""', "--ginb", "run.run_eval=True", "--ginb", f"run.shard_decoder={shard}", "--logbase", logs,
           "--scene_name", "synthetic", "--scene_dir", scene, "--cfg", str(cfg), "--eval_skip", "100"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-4000:])
    steps = [l for l in r.stdout.splitlines() if l.startswith("[run] step")]
    assert len(steps) == 3 and "world 2" in r.stdout and "Test PSNR" in r.stdout
    assert ("volume decoder sharded over 2 ranks" in r.stdout) == shard
    losses = [float(l.split(" loss ")[1].split()[0]) for l in steps]
    assert all(np.isfinite(losses)) and all(0 < v < 10 for v in losses)
    logdir = [os.path.join(logs, d) for d in os.listdir(logs)][0]
    res = json.load(open(os.path.join(logdir, "results.json")))
    assert np.isfinite(res["test"]["psnr"]) and 0.0 < res["test"]["psnr"] < 60.0
    ck = torch.load(os.path.join(logdir, "last.ckpt"), map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    w = [v for k, v in sd.items() if "mweight_vol_decoder.decoder.block_conv" in k and k.endswith("weight")]
    assert len(w) >= 3 and all(torch.isfinite(t).all() for t in w)
    assert all(float(t.abs().sum()) > 0 for t in w)            # (a rank's stale / missing rows would show as untrained or garbage)


def _launch_two_ranks(gin, port, extra, logs):
    e = dict(os.environ)
    e.update(HOS_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "run.py"), "--ginc", os.path.join(ROOT, "configs", gin),
           "--ginb", "run.max_steps=3", "--ginb", "run.log_every_n_steps=1", "--logbase", logs, "--scene_name", "synthetic"] + extra
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-4000:])
    steps = [l for l in r.stdout.splitlines() if l.startswith("[run] step")]
    assert len(steps) == 3 and "world 2" in r.stdout and "[run] wrote" in r.stdout
    return [float(l.split(" loss ")[1].split()[0]) for l in steps], r.stdout


def test_launcher_two_ranks_stage1_and_stage2_sharded_equals_replicated(tmp_path):
    """Stages 1 and 2 of the launcher as two ranks on the one GPU (synthetic items, seeds per rank): stage 1 trains and writes its
    checkpoint; stage 2 prints the SAME losses with the volume decoder sharded over the ranks as with it replicated (the sharded
    forward is the replicated forward up to summation order, and three clipped steps do not amplify that past the printed digits)."""
    l1, _ = _launch_two_ranks("state_mipnerf360_backpack.gin", 29572, [], str(tmp_path / "s1"))
    assert all(np.isfinite(l1)) and l1[-1] < l1[0]
    rep, out_r = _launch_two_ranks("state_humanobject_backpack.gin", 29573, ["--ginb", "run.shard_decoder=False"], str(tmp_path / "s2r"))
    shd, out_s = _launch_two_ranks("state_humanobject_backpack.gin", 29574, ["--ginb", "run.shard_decoder=True"], str(tmp_path / "s2s"))
    assert "volume decoder sharded over 2 ranks" in out_s and "volume decoder sharded" not in out_r
    assert all(np.isfinite(rep)) and max(abs(a - b) for a, b in zip(rep, shd)) <= 2e-5, (rep, shd)


def test_launcher_refuses_stage3_items_with_cut_patches(tmp_path):
    """The reference's stage-3 `_unpack_imgs` is a plain reshape (S3 model.py:41-50): stage 3 takes whole patches.  An item whose
    `patch_masks` has holes (a stage-2 item) is refused by the launcher when `--items` is loaded (ADVICE r4 / VERDICT r5 item 8)."""
    import torch
    from hosnerf_amd import synth
    from hosnerf_amd.train import prepare_patch_targets
    b = prepare_patch_targets(synth.add_patch_supervision(synth.human_batch(512, seed=3, time=0.5, is_train=True, iter_val=3e5), 2, 16, 3))
    b["patch_masks"] = b["patch_masks"].clone()
    b["patch_masks"][0, 0, 0] = False
    path = str(tmp_path / "items.pt")
    torch.save([b], path)
    cmd = [sys.executable, os.path.join(ROOT, "run.py"), "--ginc", os.path.join(ROOT, "configs", "hosnerf_backpack.gin"), "--ginb", "run.max_steps=1",
           "--ginb", 'run.human_path=""', "--ginb", 'run.bkgd_path=""', "--logbase", str(tmp_path / "logs"), "--scene_name", "synthetic", "--items", path]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "patch_masks has holes" in (r.stderr + r.stdout), (r.stdout[-1000:], r.stderr[-2000:])
