"""Golden vectors for the STAGE-2 training item's patch selection: outputs of the reference's own
`Dataset.get_patch_ray_indices` / `sample_patch_rays` / `sample_patch_rays_original`
(2nd_State_Conditional_Human-Object/core/data/human_nerf/train.py:215-455), imported here (build container only; cv2 /
termcolor / PIL are absent and unused by these methods, so empty modules stand in for them at import time) and driven with seeded
`np.random` on synthetic masks.  Unlike stage 3, a stage-2 patch IS intersected with the subject's box (T2:321-332): the
selection is ragged and `patch_masks` marks the pixels that kept their ray.
  python tests/golden/make_golden_patches_s2.py   ->  tests/golden/patches_s2.npz"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/2nd_State_Conditional_Human-Object"


def main():
    for name in ("cv2", "termcolor", "PIL", "PIL.Image"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["termcolor"].colored = lambda *a, **k: a[0]
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    if not hasattr(np, "bool"):
        np.bool = bool                    # the reference asserts `dtype == np.bool` (numpy < 1.24 spelling)
    sys.path.insert(0, REF)
    import core.data.human_nerf.train as T

    H, W, P, N = 48, 64, 8, 6
    yy, xx = np.mgrid[0:H, 0:W]
    bbox_mask = (yy >= 6) & (yy < 40) & (xx >= 10) & (xx < 50)                       # rays that hit the subject's box
    subject_mask = ((yy - 22.0) / 15.0) ** 2 + ((xx - 30.0) / 9.0) ** 2 < 1.0       # silhouette, pokes out of the box on top
    subject_mask[2:5, 28:33] = True
    ray_mask = bbox_mask.reshape(-1)
    nv = int(ray_mask.sum())
    rs = np.random.RandomState(5)
    arrs = {k: rs.standard_normal((nv, c)).astype(np.float32) for k, c in
            (("rays_o", 3), ("rays_d", 3), ("ray_img", 3), ("ray_grid", 5), ("near", 1), ("far", 1))}
    img = rs.uniform(0, 1, size=(H, W, 3)).astype(np.float32)

    fake = types.SimpleNamespace()
    fake.cfg = types.SimpleNamespace(patch=types.SimpleNamespace(sample_subject_ratio=0.6, N_patches=N, size=P))
    fake._get_patch_ray_indices = lambda *a, **k: T.Dataset._get_patch_ray_indices(fake, *a, **k)
    fake.get_patch_ray_indices = lambda *a, **k: T.Dataset.get_patch_ray_indices(fake, *a, **k)
    fake.select_rays = T.Dataset.select_rays
    fake.select_rays_original = T.Dataset.select_rays_original

    out = {"H": H, "W": W, "P": P, "N": N, "ratio": 0.6, "bbox_mask": bbox_mask, "subject_mask": subject_mask, "img": img}
    out.update({"in_" + k: arrs[k] for k in ("rays_o", "rays_d", "ray_img", "ray_grid", "near", "far")})
    seeds = [21, 22, 23, 24, 25]
    out["seeds"] = np.array(seeds)
    cut = 0
    for s in seeds:
        np.random.seed(s)
        sel, info, div = T.Dataset.get_patch_ray_indices(fake, N, ray_mask, subject_mask, bbox_mask, P, H, W)
        out[f"s{s}_select_inds"] = sel
        out[f"s{s}_xy_min"] = info["xy_min"]
        out[f"s{s}_xy_max"] = info["xy_max"]
        out[f"s{s}_mask"] = info["mask"]
        out[f"s{s}_div"] = div
        cut += int((~info["mask"]).sum())
        np.random.seed(s)
        r = T.Dataset.sample_patch_rays(fake, img, H, W, subject_mask, bbox_mask, ray_mask, arrs["rays_o"], arrs["rays_d"],
                                        arrs["ray_img"], arrs["ray_grid"], arrs["near"], arrs["far"])
        names = ("rays_o", "rays_d", "ray_img", "ray_grid", "near", "far", "target_patches", "patch_masks", "patch_div_indices")
        for nme, v in zip(names, r):
            out[f"s{s}_out_{nme}"] = np.asarray(v)
        out[f"s{s}_rng_after"] = np.random.rand(2)          # the stream position after the call is part of the behaviour
        np.random.seed(s)
        r0 = T.Dataset.sample_patch_rays_original(fake, img, H, W, subject_mask, bbox_mask, ray_mask, arrs["rays_o"], arrs["rays_d"],
                                                  arrs["ray_img"], arrs["near"], arrs["far"])
        assert all(np.array_equal(a, b) for a, b in zip(r0[:3] + r0[3:], r[:3] + r[4:])), "the no-flow form selects the same rays"
    assert cut > 0, "the fixture must contain patches cut by the box"
    np.savez_compressed(os.path.join(HERE, "patches_s2.npz"), **out)
    print("patches_s2.npz: cut pixels", cut, {s: out[f"s{s}_div"].tolist() for s in seeds})


if __name__ == "__main__":
    main()
