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
