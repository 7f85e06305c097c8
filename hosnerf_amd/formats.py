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


def _minimal_rotation(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """The rotation about a x b that carries the unit vector a onto the unit vector b:  R = c I + [v]x + v v^T / (1 + c)  with
    v = a x b, c = a . b (Rodrigues' formula with sin / cos expressed through v and c, so no angle is ever formed)."""
    v = np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    c = float(a @ b)
    vx = np.zeros((3, 3))
    vx[0, 1], vx[0, 2], vx[1, 2] = -v[2], v[1], -v[0]
    vx -= vx.T
    return c * np.eye(3) + vx + np.outer(v, v) / (1.0 + c)


def similarity_from_cameras(c2w: np.ndarray, strict_scaling: bool = False):
    """Scene normalisation of the stage-1 loader (what nerf_360_v2.py:295-350 computes), from camera-to-world matrices in the
    OpenCV convention (x right, y DOWN, z forward).  Returns (T [4,4], scale):
      1. orientation -- a camera's up direction in the world is minus the second column of its rotation; the mean of those,
         normalised, is carried onto (0, -1, 0) by the minimal rotation `align`;
      2. origin -- for every camera the foot of the perpendicular from the origin onto its (aligned) optical axis,
         p = t - (t . f) f; the component-wise median of the feet moves to the origin;
      3. size -- the median (strict_scaling: the largest) distance of the recentred camera centres becomes 1.
    T = [align | shift] acts on world points; `scale` multiplies the translated result."""
    rot, centres = c2w[:, :3, :3], c2w[:, :3, 3]
    down = np.array([0.0, -1.0, 0.0])
    mean_up = -rot[:, :, 1].mean(axis=0)
    mean_up = mean_up / np.sqrt(mean_up @ mean_up)
    if float(mean_up @ down) > -1.0:
        align = _minimal_rotation(mean_up, down)
    else:
        # exactly opposite directions have no minimal rotation; the reference returns this matrix for the case (:322-325)
        align = np.diag([-1.0, 1.0, 1.0])
    centres = centres @ align.T
    axes = np.einsum("ij,njk->nik", align, rot)[:, :, 2]               # optical axes after the alignment
    feet = centres - np.einsum("ni,ni->n", centres, axes)[:, None] * axes
    shift = -np.median(feet, axis=0)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = align, shift
    dist = np.sqrt(((centres + shift) ** 2).sum(axis=-1))
    return T, 1.0 / float(dist.max() if strict_scaling else np.median(dist))


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
    # LLFF keeps the camera axes as the columns (down, right, back) of each 3x5 block; everything downstream expects OpenCV's
    # (right, down, forward): exchange the first two columns, negate the third, frames to axis 0 (the net effect of
    # nerf_360_v2.py:387-389)
    poses = poses[:, [1, 0, 2, 3, 4], :] * np.array([1.0, 1.0, -1.0, 1.0, 1.0])[None, :, None]
    poses = np.ascontiguousarray(np.transpose(poses, (2, 0, 1))).astype(np.float32)
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
    test_skip = max(1, n // 16)          # (short synthetic scenes: every frame; the reference needs >= 16 frames)
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


# ------------------------------------------------------------------------------------------------ skeleton files (stages 2 / 3)
# mesh_infos.pkl        {frame: {poses [72], tpose_joints [24,3], joints [24,3], Rh, Th, ...}} per-frame SMPL fits
# canonical_joints.pkl  {joints [24,3]} canonical T-pose
# The two object joints are extrapolated from the hands (3rd_Complete_HOSNeRF/core/data/human_nerf/train.py:132-191); the
# per-frame network inputs are derived with core/utils/body_util.py:211-369.
SMPL_PARENT = {1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 13: 9, 14: 9, 15: 12,
               16: 13, 17: 14, 18: 16, 19: 17, 20: 18, 21: 19, 22: 20, 23: 21, 24: 23, 25: 22}
_TORSO = (0, 3, 6, 9, 13, 14)
_HEAD, _OBJ_R, _OBJ_L = 15, 24, 25


def add_object_joints(joints24: np.ndarray) -> np.ndarray:
    """[24,3] -> [26,3]: right object joint = J23 + (J23 - J19), left = J22 + (J22 - J18) (train.py:136-140, 167-174)."""
    j = np.asarray(joints24, dtype=np.float32)
    return np.concatenate([j, (j[23] + (j[23] - j[19]))[None], (j[22] + (j[22] - j[18]))[None]], 0)


def skeleton_bbox(joints: np.ndarray, offset: float = 0.6) -> Dict[str, np.ndarray]:
    return {"min_xyz": np.min(joints, axis=0) - offset, "max_xyz": np.max(joints, axis=0) + offset}


def load_canonical_joints(path: str, bbox_offset: float = 0.6):
    with open(path, "rb") as f:
        joints = add_object_joints(pickle.load(f)["joints"].astype("float32"))
    return joints, skeleton_bbox(joints, bbox_offset)


def load_mesh_infos(path: str, bbox_offset: float = 0.6) -> Dict:
    """train.py:160-181: 26-joint T-pose, 78-vector pose (object joints carry zero rotation), per-frame bbox of the posed joints."""
    with open(path, "rb") as f:
        infos = pickle.load(f)
    for name in infos:
        m = infos[name]
        m["tpose_joints"] = add_object_joints(m["tpose_joints"])
        m["poses"] = np.concatenate([m["poses"].astype("float32"), np.zeros(6, dtype="float32")], 0)
        m["bbox"] = skeleton_bbox(m["joints"], bbox_offset)
    return infos


def _rodrigues(rvec: np.ndarray) -> np.ndarray:
    """body_util.py:211-230 (the axis is normalised by norm + 1e-5)."""
    v = np.asarray(rvec, dtype=np.float64).reshape(3, 1)
    theta = np.linalg.norm(v)
    r = v / (theta + 1e-5)
    x, y, z = r.ravel()
    skew = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    return np.cos(theta) * np.eye(3) + np.sin(theta) * skew + (1 - np.cos(theta)) * r.dot(r.T)


def body_pose_to_body_RTs(jangles: np.ndarray, tpose_joints: np.ndarray):
    """body_util.py:233-259: per-joint local rotation (Rodrigues) and translation (offset to the parent in the T-pose)."""
    ja = np.asarray(jangles).reshape(-1, 3)
    K = ja.shape[0]
    assert tpose_joints.shape[0] == K
    Rs = np.zeros((K, 3, 3), dtype="float32")
    Ts = np.zeros((K, 3), dtype="float32")
    for i in range(K):
        Rs[i] = _rodrigues(ja[i])
        Ts[i] = tpose_joints[i] if i == 0 else tpose_joints[i] - tpose_joints[SMPL_PARENT[i]]
    return Rs, Ts


def get_canonical_global_tfms(canonical_joints: np.ndarray) -> np.ndarray:
    """body_util.py:262-282: chain of pure translations along the tree."""
    K = canonical_joints.shape[0]
    g = np.zeros((K, 4, 4), dtype="float32")
    for i in range(K):
        L = np.eye(4, dtype="float32")
        L[:3, 3] = canonical_joints[i] if i == 0 else canonical_joints[i] - canonical_joints[SMPL_PARENT[i]]
        g[i] = L if i == 0 else g[SMPL_PARENT[i]].dot(L)
    return g


def _rotation_between(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """body_util.py:96-125 for one pair (float32 result)."""
    a = a / np.clip(np.linalg.norm(a), 1e-5, None)
    b = b / np.clip(np.linalg.norm(b), 1e-5, None)
    n = np.cross(a, b)
    c = a.dot(b)
    skew = np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]], dtype=np.float32)
    return (np.eye(3) + skew + skew.dot(skew) * (1.0 / (1.0 + c))).astype(np.float32)


def _gaussian_volume(grid_size, bmin, bmax, center, S, R):
    """body_util.py:149-190: exp(-d^T (R S S R^T) d) on the [z][y][x] grid of the box."""
    sigma = R.dot(S).dot(S).dot(R.T)
    zg, yg, xg = np.meshgrid(np.linspace(bmin[2], bmax[2], grid_size), np.linspace(bmin[1], bmax[1], grid_size),
                             np.linspace(bmin[0], bmax[0], grid_size), indexing="ij")
    d = np.stack([xg - center[0], yg - center[1], zg - center[2]], axis=-1)
    return np.exp(-1 * np.einsum("abci, abci->abc", np.einsum("abci, ij->abcj", d, sigma), d))


def approx_gaussian_bone_volumes(tpose_joints: np.ndarray, bbox_min_xyz, bbox_max_xyz, grid_size: int = 32) -> np.ndarray:
    """body_util.py:285-369: one Gaussian per bone (sum over the bones that start at a joint; isotropic blobs at the leaves, wider
    for the head and the two object joints), plus the background channel; [K+1, V, V, V] normalised over the channels."""
    tj = tpose_joints.astype(np.float32)
    K = tj.shape[0]
    up = np.array([0.0, 1.0, 0.0], dtype=np.float32)

    def scale_mtx(stds):
        return np.diag(1.0 / np.asarray(stds, dtype=np.float32)).astype(np.float32)

    vols = []
    for j in range(K):
        vol = np.zeros((grid_size,) * 3, dtype="float32")
        parent_of_any = False
        for bone, par in SMPL_PARENT.items():
            if par != j:
                continue
            S = scale_mtx(np.array([0.03, 0.06, 0.03]) * 2.0)
            if j in _TORSO:
                S[0][0] *= 1 / 1.5
                S[2][2] *= 1 / 1.5
            a, b = tj[SMPL_PARENT[bone]], tj[bone]
            vol = vol + _gaussian_volume(grid_size, bbox_min_xyz, bbox_max_xyz, (a + b) / 2.0, S, _rotation_between(up, b - a))
            parent_of_any = True
        if not parent_of_any:
            stds = np.array([0.06] * 3) if j in (_HEAD, _OBJ_R, _OBJ_L) else np.array([0.02] * 3)
            vol = _gaussian_volume(grid_size, bbox_min_xyz, bbox_max_xyz, tj[j], scale_mtx(stds * 2.0), np.eye(3, dtype="float32"))
        vols.append(vol)
    g = np.stack(vols, 0)
    g = np.concatenate([g, 1.0 - np.sum(g, axis=0, keepdims=True).clip(min=0.0, max=1.0)], 0)
    return g / np.sum(g, axis=0, keepdims=True).clip(min=0.001)


def skeleton_item(mesh_info: Dict, canonical_joints: np.ndarray, canonical_bbox: Dict, volume_size: int = 32) -> Dict[str, np.ndarray]:
    """The pose part of a training item (train.py:640-727): dst_Rs / dst_Ts / cnl_gtfms / motion_weights_priors / dst_posevec and
    the canonical box, from one frame of mesh_infos.pkl and canonical_joints.pkl."""
    Rs, Ts = body_pose_to_body_RTs(mesh_info["poses"], mesh_info["tpose_joints"])
    bmin, bmax = canonical_bbox["min_xyz"].astype("float32"), canonical_bbox["max_xyz"].astype("float32")
    return {"dst_Rs": Rs, "dst_Ts": Ts, "cnl_gtfms": get_canonical_global_tfms(canonical_joints),
            "motion_weights_priors": approx_gaussian_bone_volumes(canonical_joints, bmin, bmax, volume_size).astype("float32"),
            "dst_posevec": mesh_info["poses"][3:] + 1e-2, "cnl_bbox_min_xyz": bmin, "cnl_bbox_max_xyz": bmax,
            "cnl_bbox_scale_xyz": 2.0 / (bmax - bmin)}
