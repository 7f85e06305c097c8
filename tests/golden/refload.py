"""Import the *reference* (read-only mount at /root/reference) with stub modules.

Only usable in the build container -- the GPU box has no /root/reference.  Used by
make_golden.py (fixture generation) and by tests that are skipped when the mount is absent.
Nothing from the reference is copied: the modules are imported from where they lie.
Recipe: SURVEY.md Appendix C.
"""
from __future__ import annotations

import contextlib
import importlib
import json
import os
import sys
import tempfile
import types

REF = os.environ.get("HOSNERF_REFERENCE", "/root/reference")
STAGE = {
    1: os.path.join(REF, "1st_State-Conditional_Scene"),
    2: os.path.join(REF, "2nd_State_Conditional_Human-Object"),
    3: os.path.join(REF, "3rd_Complete_HOSNeRF"),
}


def available() -> bool:
    return os.path.isdir(STAGE[3])


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    import torch

    sys.dont_write_bytecode = True

    def configurable(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _stub("gin", configurable=configurable, query_parameter=lambda *_: 1, REQUIRED=None)
    pl = _stub("pytorch_lightning", LightningModule=torch.nn.Module, LightningDataModule=object)
    pl.seed_everything = lambda *a, **k: None
    _stub("piqa")
    _stub("piqa.lpips", LPIPS=object)
    _stub("piqa.ssim", SSIM=object)
    for n in ("imageio", "skimage", "cv2", "termcolor"):
        _stub(n)
    sys.modules["termcolor"].colored = lambda s, *a, **k: s
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")


_PURGE = ("src", "core", "utils", "third_parties", "configs")


@contextlib.contextmanager
def stage(n: int):
    """chdir + sys.path into one reference stage directory (needed by its imp.load_source)."""
    _install_stubs()
    for k in [k for k in sys.modules if k.split(".")[0] in _PURGE]:
        del sys.modules[k]
    cwd = os.getcwd()
    sys.path.insert(0, STAGE[n])
    os.chdir(STAGE[n])
    try:
        yield
    finally:
        os.chdir(cwd)
        sys.path.remove(STAGE[n])


def make_basedir(transitions=(0.4,)):
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    if transitions is not None:
        with open(os.path.join(d, "transitions_times.json"), "w") as f:
            json.dump({f"f{i}": {"time": float(t)} for i, t in enumerate(transitions)}, f)
    return d


def helper(n: int = 3):
    with stage(n):
        return importlib.import_module("src.model.mipnerf360.helper")


def background_model(n: int = 3, transitions=(0.4,), **kw):
    """Instantiate the reference MipNeRF360 of stage n (1 or 3)."""
    with stage(n):
        mod = importlib.import_module("src.model.mipnerf360.model")
        return mod, mod.MipNeRF360(make_basedir(transitions), **kw)


def human_cfg(n: int = 3, transitions=(0.4,)):
    with stage(n):
        from third_parties.yacs import CfgNode as CN
        cfg = CN()
        # run.py:33-49 defaults that precede the yaml merge
        cfg.resume = False
        cfg.eval_iter = 10000000
        cfg.render_folder_name = ""
        cfg.ignore_non_rigid_motions = False
        cfg.render_skip = 1
        cfg.render_frames = 100
        cfg.num_workers = 4
        cfg.merge_from_file("configs/default.yaml")
        cfg.merge_from_file("configs/human_nerf/wild/monocular/adventure.yaml")
        cfg.basedir = make_basedir(transitions)
        return cfg


def human_network(n: int = 3, transitions=(0.4,)):
    cfg = human_cfg(n, transitions)
    with stage(n):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from core.nets import create_network
            net = create_network(cfg)
    return cfg, net
