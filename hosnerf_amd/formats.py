"""On-disk formats of a HOSNeRF scene directory and the scene normalisation the stage-1 loader derives from them
(1st_State-Conditional_Scene/src/data/data_util/nerf_360_v2.py:80-160, 295-488, src/data/pose_utils.py:129-205):

    poses_bounds.npy        [N, 17] LLFF rows: 3x5 pose (rotation | translation | h, w, focal) + near / far bounds
    cameras.pkl             {frame: {intrinsics [3,3], smpl_to_camera [4,4], smpl_to_world [4,4], ...}} (HumanNeRF-style preprocessing)
    cameras_scaleworld.pkl  WRITTEN by the stage-1 loader, read by stages 2/3: {frame: {intrinsics, smpl_to_camera,
                            smpl_to_scale_world, scaleworld_to_camera}} -- the similarity that maps the SMPL frame of every
                            frame into the normalised ("scale world") frame the background model is trained in
    transitions_times.json  {name: {"time": t}} state transitions (one learned embedding per interval)
    masks/*.png             human masks; pixels with mask < 1 are the stage-1 training rays (`bkgrays_sizes`)

Host-side numpy only (this is what runs once per scene, before any ray exists); image decoding stays with the caller (the
reference uses imageio, which this build does not depend on): `load_scene` takes the image size and, optionally, the masks.
Pinned by tests/golden/formats.npz = the reference's own `load_nerf_360_v2_data` run on a synthetic scene directory
(tests/golden/make_golden_formats.py).
"""
from __future__ import annotations

import json
import os
import pickle
from typing import Dict, Optional, Sequence, Tuple

import numpy as np


def load_poses_bounds(path: str, image_hw: Tuple[int, int], factor: float = 1.0):
    """`_load_data` (nerf_360_v2.py:80-145) without the images: poses [3,5,N] with the image size / scaled focal written into
    column 4, bounds [2,N]."""
    arr = np.load(path)
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0]).copy()
    bds = arr[:, -2:].transpose([1, 0])
    poses[:2, 4, :] = np.array(image_hw[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1.0 / factor
    return poses, bds


def similarity_from_cameras(c2w: np.ndarray, strict_scaling: bool = False):
    """nerf_360_v2.py:295-350: rotate the world so that z+ is up (mean camera up axis), recentre on the median of the points
    of the camera centre rays closest to the origin, rescale by the median (or max) camera distance.  Returns (T [4,4], scale)."""
    t = c2w[:, :3, 3]
    R = c2w[:, :3, :3]
    ups = np.sum(R * np.array([0, -1.0, 0]), axis=-1)
    world_up = np.mean(ups, axis=0)
    world_up /= np.linalg.norm(world_up)
    up_camspace = np.array([0.0, -1.0, 0.0])
    c = (up_camspace * world_up).sum()
    cross = np.cross(world_up, up_camspace)
    skew = np.array([[0.0, -cross[2], cross[1]], [cross[2], 0.0, -cross[0]], [-cross[1], cross[0], 0.0]])
    if c > -1:
        R_align = np.eye(3) + skew + (skew @ skew) * 1 / (1 + c)
    else:
        R_align = np.array([[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    R = R_align @ R
    fwds = np.sum(R * np.array([0, 0.0, 1.0]), axis=-1)
    t = (R_align @ t[..., None])[..., 0]
    nearest = t + (fwds * -t).sum(-1)[:, None] * fwds
    translate = -np.median(nearest, axis=0)
    transform = np.eye(4)
    transform[:3, 3] = translate
    transform[:3, :3] = R_align
    scale_fn = np.max if strict_scaling else np.median
    scale = 1.0 / scale_fn(np.linalg.norm(t + translate, axis=-1))
    return transform, scale


def _r_to_axis_angle(m):
    axis = np.stack([m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]], -1)
    r = np.hypot(axis[:, 0], np.hypot(axis[:, 1], axis[:, 2]))
    t = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    return axis / r[:, None], np.arctan2(r, t - 1)


def _r_axis_angle(angle, axis):
    ca, sa = np.cos(angle), np.sin(angle)
    C = 1 - ca
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    m = np.zeros((len(angle), 3, 3))
    m[:, 0, 0] = x * x * C + ca; m[:, 0, 1] = x * y * C - z * sa; m[:, 0, 2] = z * x * C + y * sa
    m[:, 1, 0] = x * y * C + z * sa; m[:, 1, 1] = y * y * C + ca; m[:, 1, 2] = y * z * C - x * sa
    m[:, 2, 0] = z * x * C - y * sa; m[:, 2, 1] = y * z * C + x * sa; m[:, 2, 2] = z * z * C + ca
    return m


def pose_interp(poses: np.ndarray, factor: int) -> np.ndarray:
    """pose_utils.py:129-152: `factor - 1` interpolated poses between neighbours (axis-angle on the rotation, linear on the
    translation), 4x as many between the last and the first."""
    out = []
    for i in range(len(poses)):
        out.append(poses[i])
        if i == len(poses) - 1:
            factor = 4 * factor
        nxt = (i + 1) % len(poses)
        axis, angle = _r_to_axis_angle((poses[nxt, :3, :3] @ poses[i, :3, :3].T)[None])
        for j in range(factor - 1):
            ret = np.eye(4)
            f = (j + 1) / factor
            ret[:3, :3] = _r_axis_angle(angle * f, axis) @ poses[i, :3, :3]
            ret[:3, 3] = (1 - f) * poses[i, :3, 3] + f * poses[nxt, :3, 3]
            out.append(ret)
    return np.stack(out)


def load_cameras(path: str) -> Dict:
    with open(path, "rb") as f:
        return pickle.load(f)


def load_transitions_times(basedir: str) -> Optional[np.ndarray]:
    """M:163-172 / N:38-48: sorted by file order, one state embedding more than there are transitions; None if absent."""
    p = os.path.join(basedir, "transitions_times.json")
    if not os.path.exists(p):
        return None
    with open(p, "r") as f:
        infos = json.load(f)
    return np.stack([np.array(infos[k]["time"], dtype=np.float32) for k in infos], axis=0)


def load_scene(basedir: str, image_hw: Tuple[int, int], masks: Optional[np.ndarray] = None, cam_scale_factor: float = 0.95,
               strict_scaling: bool = False, factor: float = 1.0, near: Optional[float] = None, far: Optional[float] = None,
               write_cameras_scaleworld: bool = True) -> Dict:
    """`load_nerf_360_v2_data` (nerf_360_v2.py:367-488) minus image decoding: normalised extrinsics, intrinsics, splits, render
    path, per-frame times, `bkgrays_sizes` (if `masks` [N,H,W] in 0..1 is given), and -- like the reference -- writes
    `cameras_scaleworld.pkl` next to `cameras.pkl` for stages 2 and 3."""
    poses, bds = load_poses_bounds(os.path.join(basedir, "poses_bounds.npy"), image_hw, factor)
    cams = load_cameras(os.path.join(basedir, "cameras.pkl"))
    # rotation-matrix ordering of LLFF -> OpenCV, variable dimension to axis 0 (nerf_360_v2.py:387-389)
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.concatenate([poses[:, 0:1, :], -poses[:, 1:2, :], -poses[:, 2:3, :], poses[:, 3:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    n = poses.shape[0]
    times = np.linspace(0.0, 1.0, n).astype(np.float32)
    extr = np.stack([np.eye(4) for _ in range(n)])
    extr[:, :3, :4] = poses[:, :3, :4]
    T, sscale = similarity_from_cameras(extr, strict_scaling)
    extr = np.einsum("nij, ki -> nkj", extr, T)
    scene_scale = cam_scale_factor * sscale
    extr[:, :3, 3] *= scene_scale
    cam_to_scaleworld = extr.copy()
    s = np.eye(4)
    s[:3, :3] *= scene_scale
    world_to_scaleworld = s @ T
    scaleworld = {}
    for idx, name in enumerate(cams):
        scaleworld[name] = {
            "intrinsics": cams[name]["intrinsics"],
            "smpl_to_camera": cams[name]["smpl_to_camera"],
            "smpl_to_scale_world": np.array((world_to_scaleworld @ cams[name]["smpl_to_world"]).tolist(), dtype=np.float32),
            "scaleworld_to_camera": np.array(np.linalg.inv(cam_to_scaleworld[idx]).tolist(), dtype=np.float32),
        }
    if write_cameras_scaleworld:
        with open(os.path.join(basedir, "cameras_scaleworld.pkl"), "wb") as f:
            pickle.dump(scaleworld, f)
    render_poses = pose_interp(extr, 2)
    test_skip = n // 16
    i_test = np.arange(n)[::test_skip][:16]
    i_train = np.array([i for i in range(n) if i not in i_test])
    h, w, focal = poses[0, :3, -1]
    h, w = int(h), int(w)
    intr = np.array([[[focal, 0.0, 0.5 * w], [0.0, focal, 0.5 * h], [0.0, 0.0, 1.0]] for _ in range(n)])
    out = {"extrinsics": extr, "intrinsics": intr, "image_sizes": np.array([[h, w] for _ in range(n)]),
           "near": 0.0 if near is None else near, "far": 1.0 if far is None else far, "ndc_coeffs": (-1.0, -1.0),
           "i_split": (i_train, i_train[:2], i_test, np.arange(n)), "render_poses": render_poses, "times": times,
           "render_times": np.linspace(0.0, 1.0, render_poses.shape[0]).astype(np.float32),
           "world_to_scaleworld": world_to_scaleworld, "scene_scale": scene_scale, "cameras_scaleworld": scaleworld,
           "transitions_times": load_transitions_times(basedir), "bounds": bds}
    if masks is not None:
        out["bkgrays_sizes"] = np.sum(np.asarray(masks, dtype=np.float32) < 1, axis=(1, 2))
    return out
