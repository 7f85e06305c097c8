"""Launcher surface (BASELINE configs[0] "CPU plumbing"): gin files in the reference's format, the reference's
command-line flags, `run.py --cpu` for the three stages, checkpoint + optimiser-state round trip -- no GPU, no kernels."""
import json
import os
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import run as launcher  # noqa: E402
from hosnerf_amd import gin_lite  # noqa: E402


def test_gin_lite_parses_reference_style_files_and_bindings():
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "base.gin"), "w") as f:
        f.write('### 360-v2 Specific Arguments\n\nrun.dataset_name = "nerf_360_v2"\nrun.datadir = "Path to # the dataset"\n'
                'LitData.batch_size = 4096  # rays\nLitDataNeRF360V2.far = 1e6\nMipNeRF360.opaque_background = True\n'
                'MipNeRF360.bg_intensity_range = (1.0,\n    1.0)\n')
    with open(os.path.join(d, "top.gin"), "w") as f:
        f.write("include 'base.gin'\nrun.model_name = \"state_mipnerf360\"\nrun.max_steps = 500000\nrun.grad_max_norm = 0.001\n")
    g = gin_lite.parse_config_files_and_bindings([os.path.join(d, "top.gin")], ["run.max_steps=12", "LitMipNeRF360.lr_init = 1e-3"])
    assert g["run.datadir"] == "Path to # the dataset" and g["LitData.batch_size"] == 4096 and g["LitDataNeRF360V2.far"] == 1e6
    assert g["MipNeRF360.opaque_background"] is True and g["MipNeRF360.bg_intensity_range"] == (1.0, 1.0)
    assert g.kwargs("run") == {"dataset_name": "nerf_360_v2", "datadir": "Path to # the dataset", "model_name": "state_mipnerf360",
                               "max_steps": 12, "grad_max_norm": 0.001}
    assert g["LitMipNeRF360.lr_init"] == 1e-3
    with pytest.raises(ValueError):
        gin_lite.parse_lines(["run.max_steps"])


def test_reference_gin_files_parse_unchanged():
    """The reference's own .gin files, when the mount is present (build container)."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference mount absent")
    for rel in ("1st_State-Conditional_Scene/configs/state_mipnerf360/Backpack.gin", "3rd_Complete_HOSNeRF/configs/HOSNeRF/Backpack.gin"):
        g = gin_lite.parse_file(os.path.join(ref, rel))
        assert g["run.model_name"] in ("state_mipnerf360", "hosnerf") and g["run.grad_max_norm"] == 0.001


@pytest.mark.parametrize("gin,name,keys", [("state_mipnerf360_backpack.gin", "state_mipnerf360", 53), ("state_humanobject_backpack.gin", "state_humanobject", 75),
                                           ("hosnerf_backpack.gin", "hosnerf", 128)])
def test_run_cpu_plumbing(gin, name, keys, capsys):
    logs = tempfile.mkdtemp()
    plan = launcher.main(["--ginc", os.path.join(ROOT, "configs", gin), "--scene_name", "Backpack", "--logbase", logs, "--cpu",
                          "--ginb", "run.max_steps=7", "--seed", "5"])
    assert plan["model_name"] == name and plan["state_dict_keys"] == keys and plan["crop_rays"] == 4096 and plan["max_steps"] == 7
    assert plan["checkpoint_roundtrip"] == {"missing": 0, "unexpected": 0} and os.path.exists(plan["checkpoint"])
    assert plan["exp_name"].startswith(name) and plan["exp_name"].endswith("_Backpack_005")            # S3/run.py:108-110
    ck = torch.load(plan["checkpoint"], map_location="cpu", weights_only=False)
    assert "state_dict" in ck and "optimizer_states" in ck and len(ck["state_dict"]) == keys
    # resume: the optimiser state and the global step come back
    plan2 = launcher.main(["--ginc", os.path.join(ROOT, "configs", gin), "--scene_name", "Backpack", "--logbase", logs, "--cpu", "--seed", "5",
                           "--resume_training", "true"])
    assert plan2["checkpoint"] == plan["checkpoint"]


def test_crop_rays_match_the_config0_description():
    c = launcher.crop_rays_64()
    assert c["rays_o"].shape == (4096, 3) and c["radii"].shape == (4096, 1) and float(c["radii"].min()) > 0
    assert torch.allclose(c["viewdirs"].norm(dim=-1), torch.ones(4096), atol=1e-6)


def test_flags_of_the_reference_launcher_exist():
    a = launcher.parse_args(["--ginc", "a.gin", "--ginc", "b.gin", "--ginb", "run.x=1", "--resume_training", "--ckpt_path", "p", "--scene_name", "s",
                             "--seed", "3", "--logbase", "l", "--cfg", "c.yaml"])
    assert a.ginc == ["a.gin", "b.gin"] and a.ginb == ["run.x=1"] and a.resume_training is True and a.ckpt_path == "p"
    assert a.scene_name == "s" and a.seed == 3 and a.logbase == "l" and a.cfg == "c.yaml"
    assert launcher.parse_args([]).seed == 220901                                                      # S3/run.py:270


def _basedir(tmp_path):
    with open(os.path.join(str(tmp_path), "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)


class _RecordingOpt:
    """Stands in for the optimiser: remembers the learning rate each group had at the moment step() ran."""

    def __init__(self, n_groups=1):
        self.param_groups = [{"lr": -1.0} for _ in range(n_groups)]
        self.seen = []

    def step(self, closure=None):
        self.seen.append([g["lr"] for g in self.param_groups])


def test_schedule_order_matches_the_reference(tmp_path):
    """stage 1 writes lr(step) BEFORE optimizer.step (M1:541-569: the first update runs at the warm-up rate
    lr_init * lr_delay_mult), stages 2/3 step first and then write base * 0.1 ** (step / 500k) with step = the index of the
    step just taken (M2:606-634, M:1631-1656)."""
    from hosnerf_amd.select_option import select_model
    from hosnerf_amd.train import stage1_lr
    _basedir(tmp_path)
    lit = select_model("state_mipnerf360", str(tmp_path), max_steps=1000, grad_max_norm=0.001)
    opt = _RecordingOpt()
    for i in range(3):
        lit.optimizer_step(0, i, opt)
    assert [s[0] for s in opt.seen] == [stage1_lr(i, 1000) for i in range(3)]
    assert abs(opt.seen[0][0] - 2e-3 * 0.01) < 1e-12 and lit._global_step() == 3
    lit3 = select_model("hosnerf", str(tmp_path), grad_max_norm=0.001)
    lit3._step = 1000
    opt3 = _RecordingOpt(2)
    lit3.optimizer_step(0, 0, opt3)
    assert opt3.seen[0] == [-1.0, -1.0]                                      # stepped with the rates it had
    want = 6.667e-5 * 0.1 ** (1000 / 5e5)
    assert all(abs(g["lr"] - want) < 1e-15 for g in opt3.param_groups) and lit3._global_step() == 1001


def test_per_module_learning_rates_follow_cfg_train(tmp_path):
    """`cfg.train.lr_<module>` of a --cfg yaml decide the rates (optimizer.py:19-60); the shipped defaults otherwise.
    The ranges tile the flat buffer exactly and reproduce `human_lr_ranges` for the defaults."""
    from hosnerf_amd.human_nerf import Cfg, default_cfg
    from hosnerf_amd.select_option import select_model
    from hosnerf_amd.train import human_lr_from_cfg, human_lr_ranges, lr_ranges_by_name
    _basedir(tmp_path)
    lit = select_model("state_humanobject", str(tmp_path), grad_max_norm=0.001)
    f = lit.fused_optimizer()
    assert f.max_grad_norm == 0.001 and abs(f.lr - 6.667e-4) < 1e-12
    n = lit.human.flat_param.numel()
    # the ranges tile exactly the ACTIVE part of the flat buffer: everything but the reference-shaped first deconvolution weight,
    # whose 8 live taps per (Cin, Cout) pair are updated through the compact copy (human_nerf.Network._w0c)
    import numpy as np
    cover = np.zeros(n, np.int32)
    for off, cnt, _ in f.lr_ranges:
        cover[off:off + cnt] += 1
    active = np.zeros(n, np.int32)
    for off, cnt in lit.human.store.active_spans():
        active[off:off + cnt] = 1
    assert np.array_equal(cover, active) and all(r[0] % 4 == 0 for r in f.lr_ranges)
    (ioff, inum), = lit.human.store.inactive
    w0 = lit.human._plain["mweight_vol_decoder.decoder.block_conv.0.weight"]
    assert ioff == (w0.data_ptr() - lit.human.flat_param.data_ptr()) // 4 and inum >= w0.numel() == 1024 * 512 * 64
    ref = human_lr_ranges(lit.human, 6.667e-4, 6.667e-5)
    # same rate at every flat element (the merge boundaries may differ by padding that belongs to neither module)
    import numpy as np

    def dense(ranges):
        v = np.zeros(n)
        for off, cnt, mult in ranges:
            v[off:off + cnt] = mult
        return v
    d1, d2 = dense(f.lr_ranges), dense(ref)
    live = np.zeros(n, bool)
    for p, r, _, _ in lit.human.store._bindings:
        live[r.offset:r.offset + r.numel] = True
    live &= active.astype(bool)
    assert np.allclose(d1[live], d2[live])
    cfg = default_cfg(str(tmp_path))
    cfg.train = Cfg(lr=1e-3, lr_cnl_mlp=2e-4, lr_pose_decoder=5e-5, lr_bkgd=3e-4, lrate_decay=250)
    base, lr_of = human_lr_from_cfg(cfg, 6.667e-4)
    assert base == 2e-4 and lr_of("pose_decoder.block_mlps.0.weight") == 5e-5 and lr_of("cnl_mlp.pts_linears.0.weight") == 2e-4
    assert lr_of("non_rigid_mlp.block_mlps.0.weight") == 2e-5
    lit3 = select_model("hosnerf", str(tmp_path), cfg=cfg, grad_max_norm=0.001)
    ob_, oh_ = lit3.fused_optimizers()
    assert ob_.clip is oh_.clip and ob_.lr == 3e-4 and oh_.lr == 2e-4
    rng = lr_ranges_by_name(lit3.human, lr_of, base)
    pd = lit3.human._plain["pose_decoder.block_mlps.0.weight"]
    off = (pd.data_ptr() - lit3.human.flat_param.data_ptr()) // 4
    assert any(r[0] <= off < r[0] + r[1] and abs(r[2] - 0.25) < 1e-12 for r in rng)
    opt3 = _RecordingOpt(2)
    lit3._step = 500
    lit3.optimizer_step(0, 0, opt3)
    dec = 0.1 ** (500 / 250e3)
    assert abs(opt3.param_groups[0]["lr"] - 3e-4 * dec) < 1e-15 and abs(opt3.param_groups[1]["lr"] - 2e-4 * dec) < 1e-15
