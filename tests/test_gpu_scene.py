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
