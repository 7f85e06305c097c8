"""Golden vectors for the on-disk formats / scene normalisation: the REFERENCE's own stage-1 loader
(`load_nerf_360_v2_data`, 1st_State-Conditional_Scene/src/data/data_util/nerf_360_v2.py:367-488) run on a synthetic scene
directory built here (20 frames of 12x10 PNG images and masks, `poses_bounds.npy`, `cameras.pkl`), with imageio stubbed by PIL.
Stores the scene's inputs and everything the loader returned or wrote (`cameras_scaleworld.pkl`).
  python tests/golden/make_golden_formats.py   ->  tests/golden/formats.npz"""
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import refload


def synthetic_scene(d, n=20, H=12, W=10, seed=5):
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(d, "images")); os.makedirs(os.path.join(d, "masks"))
    rows, cams, masks = [], {}, []
    for i in range(n):
        ang = 2 * np.pi * i / n
        c = np.array([2.5 * np.cos(ang), 2.5 * np.sin(ang), 0.4 + 0.2 * np.sin(3 * ang)]) + rs.normal(0, 0.05, 3)
        fwd = -c / np.linalg.norm(c)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        upc = np.cross(right, fwd)
        # LLFF pose columns: [down, right, backwards] | translation | hwf
        R = np.stack([-upc, right, -fwd], 1)
        pose = np.concatenate([R, c[:, None], np.array([[H], [W], [14.0]])], 1)
        rows.append(np.concatenate([pose.reshape(-1), [0.5, 6.0]]))
        name = "frame_%06d" % i
        Image.fromarray(rs.randint(0, 255, size=(H, W, 3)).astype(np.uint8)).save(os.path.join(d, "images", name + ".png"))
        m = (rs.uniform(size=(H, W)) > 0.7).astype(np.uint8) * 255
        Image.fromarray(m).save(os.path.join(d, "masks", name + ".png"))
        masks.append(m / 255.0)
        s2w = np.eye(4); s2w[:3, :3] = 0.9 * np.eye(3); s2w[:3, 3] = rs.normal(0, 0.2, 3)
        s2c = np.eye(4); s2c[:3, 3] = rs.normal(0, 0.3, 3) + np.array([0, 0, 3.0])
        cams[name] = {"intrinsics": np.array([[14.0, 0, W / 2], [0, 14.0, H / 2], [0, 0, 1]], np.float32), "smpl_to_camera": s2c.astype(np.float32),
                      "smpl_to_world": s2w.astype(np.float32)}
    np.save(os.path.join(d, "poses_bounds.npy"), np.stack(rows))
    with open(os.path.join(d, "cameras.pkl"), "wb") as f:
        pickle.dump(cams, f)
    return np.stack(rows), cams, np.stack(masks)


def main():
    root = tempfile.mkdtemp(prefix="hos_scene_")
    scene = "Synth"
    d = os.path.join(root, scene)
    os.makedirs(d)
    rows, cams, masks = synthetic_scene(d)
    with refload.stage(1):
        from PIL import Image
        im = sys.modules["imageio"]
        im.imread = lambda f, **k: np.asarray(Image.open(f))
        import importlib
        L = importlib.import_module("src.data.data_util.nerf_360_v2")
        res = L.load_nerf_360_v2_data(root, scene, 0, 0.95, 1, 1, 1, 0.1, 1e6, False)
    (images, rmasks, intr, extr, sizes, near, far, ndc, i_split, render_poses, bkg, times, render_times) = res
    with open(os.path.join(d, "cameras_scaleworld.pkl"), "rb") as f:
        csw = pickle.load(f)
    names = sorted(cams.keys())
    out = {"poses_bounds": rows, "masks": masks, "H": 12, "W": 10, "names": np.array(names),
           "cam_intrinsics": np.stack([cams[k]["intrinsics"] for k in names]), "cam_smpl_to_camera": np.stack([cams[k]["smpl_to_camera"] for k in names]),
           "cam_smpl_to_world": np.stack([cams[k]["smpl_to_world"] for k in names]),
           "intrinsics": intr, "extrinsics": extr, "image_sizes": sizes, "near": near, "far": far, "i_train": i_split[0], "i_val": i_split[1],
           "i_test": i_split[2], "i_all": i_split[3], "render_poses": render_poses, "bkgrays_sizes": bkg, "times": times, "render_times": render_times,
           "csw_smpl_to_scale_world": np.stack([csw[k]["smpl_to_scale_world"] for k in names]),
           "csw_scaleworld_to_camera": np.stack([csw[k]["scaleworld_to_camera"] for k in names])}
    out.update(skeleton_golden())
    np.savez_compressed(os.path.join(HERE, "formats.npz"), **out)
    print("formats.npz", os.path.getsize(os.path.join(HERE, "formats.npz")) / 1024, "KB", extr.shape, render_poses.shape)


def skeleton_golden():
    """The reference's own body_util functions (3rd_Complete_HOSNeRF/core/utils/body_util.py) on a synthetic SMPL fit."""
    rs = np.random.RandomState(11)
    from hosnerf_amd import synth
    tj24 = synth.tpose_joints()[:24].astype(np.float32)
    poses72 = (rs.standard_normal(72) * 0.25).astype(np.float32)
    joints24 = tj24 + rs.normal(0, 0.05, tj24.shape).astype(np.float32)
    with refload.stage(3):
        import importlib
        B = importlib.import_module("core.utils.body_util")
        tj26 = np.concatenate([tj24, (tj24[23] + (tj24[23] - tj24[19]))[None], (tj24[22] + (tj24[22] - tj24[18]))[None]], 0)
        p78 = np.concatenate([poses72, np.zeros(6, np.float32)])
        Rs, Ts = B.body_pose_to_body_RTs(p78, tj26)
        g = B.get_canonical_global_tfms(tj26)
        bmin, bmax = tj26.min(0) - 0.6, tj26.max(0) + 0.6
        vol = B.approx_gaussian_bone_volumes(tj26, bmin, bmax, grid_size=32).astype(np.float32)
    return {"sk_tpose24": tj24, "sk_poses72": poses72, "sk_joints24": joints24, "sk_Rs": Rs, "sk_Ts": Ts, "sk_gtfms": g,
            "sk_vol_sub": vol[:, ::4, ::4, ::4], "sk_vol_sum": vol.astype(np.float64).sum(), "sk_vol_max": vol.max(axis=(1, 2, 3))}


if __name__ == "__main__":
    main()
