"""Free-viewpoint cameras and evaluation frames of a scene directory (BASELINE configs[4]).

The reference renders its free-viewpoint sequence from ONE training frame: `FreeviewDataset` (3rd_Complete_HOSNeRF/core/data/
human_nerf/freeview.py:39-337) takes the cameras / SMPL fit of frame `cfg.freeview.frame_idx` and, for k = 0 .. render_frames-1,
turns the camera about the subject with `rotate_camera_by_frame_idx` (core/utils/camera_util.py:106-131 -> `_update_extrinsics`,
:14-70): a rotation by 2 pi k / period about an axis tilted 15 degrees out of the vertical, centred on the subject's translation
Th.  `LitMipNeRF360.free_view` (src/model/mipnerf360/model.py:1293-1494) renders those frames; `test_metrics` (:884-1085) renders
the held-out frames with their own cameras and reports PSNR.

Here: `orbit_camera` is that camera (pinned against the reference's function by tests/golden/freeview.npz), `freeview_frame` /
`eval_frame` build the frame dicts `eval.render_frame` consumes with every per-pixel quantity produced on the device
(`eval.frame_rays`), and `load_scene_pixels` decodes `images/*.png` / `masks/*.png` / `images_flow/*_bwd.npz` of a scene
directory (PIL; the reference uses cv2 / PIL, core/utils/image_util.py)."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

TILT_DEG = 15.0          # the orbit's axis: (0, cos 15, sin 15) in SMPL space (camera_util.py:45)


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """Rotation matrix of an axis-angle vector (what cv2.Rodrigues(rvec)[0] returns), float64."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r @ r))
    if theta < 1e-12:
        return np.eye(3)
    k = r / theta
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + math.sin(theta) * Kx + (1.0 - math.cos(theta)) * (Kx @ Kx)


def orbit_camera(extrinsics: np.ndarray, frame_idx: int, period: int, trans: Optional[np.ndarray] = None,
                 inv_angle: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """`rotate_camera_by_frame_idx` (camera_util.py:106-131): camera k of a `period`-frame turn about the subject.

    Returns (E_k [4,4], T_smpl [4,4]): E_k = E @ T_smpl, where T_smpl is the inverse of the rigid motion "rotate by the angle about
    the tilted axis through `trans`" -- moving the scene by T_smpl^-1 in front of the fixed camera is the same picture as moving
    the camera.  The angle changes sign when the camera's up vector points down in SMPL space (camera_util.py:40-42) and with
    `inv_angle` (the 'zju_mocap' source type; 'wild' = False, freeview.py:34-37).  The rotation is rounded to float32 like the
    reference's (`.astype('float32')`, camera_util.py:46)."""
    E = np.asarray(extrinsics, dtype=np.float64)
    angle = 2.0 * math.pi * (frame_idx / period)
    if inv_angle:
        angle = -angle
    cam_to_smpl = np.linalg.inv(E)
    if cam_to_smpl[:3, :3].T[1, 1] < 0.0:
        angle = -angle
    axis = np.array([0.0, math.cos(math.radians(TILT_DEG)), math.sin(math.radians(TILT_DEG))])
    G = rodrigues(axis * angle).astype(np.float32).astype(np.float64)
    T = np.eye(4)
    T[:3, :3] = G
    if trans is not None:
        t = np.asarray(trans, dtype=np.float64).reshape(3)
        # x -> G (x - t) + t
        T[:3, 3] = t - G @ t
    T_smpl = np.linalg.inv(T)
    return E @ T_smpl, T_smpl


def load_scene_pixels(scene_dir: str, frames: Optional[Sequence[str]] = None) -> Dict:
    """Decode what the reference's datasets read per frame (core/data/human_nerf/train.py:300-345, freeview.py:173-197):
    `images/<frame>.png` -> [N,H,W,3] in 0..1, `masks/<frame>.png` -> [N,H,W] in 0..1 (all ones if the folder is missing),
    `images_flow/<frame>_bwd.npz` {flow [H,W,2], mask [H,W]} -> [N,H,W,3] (zeros where a file is missing), and the frame names
    (sorted file stems, `list_files` order)."""
    from PIL import Image
    img_dir = os.path.join(scene_dir, "images")
    if frames is None:
        frames = sorted(os.path.splitext(f)[0] for f in os.listdir(img_dir) if f.lower().endswith(".png"))
    images, alphas, flows = [], [], []
    any_flow = False
    for name in frames:
        img = np.asarray(Image.open(os.path.join(img_dir, name + ".png")).convert("RGB"), dtype=np.float32) / 255.0
        images.append(img)
        mp = os.path.join(scene_dir, "masks", name + ".png")
        if os.path.exists(mp):
            m = np.asarray(Image.open(mp), dtype=np.float32)
            alphas.append((m[..., 0] if m.ndim == 3 else m) / 255.0)
        else:
            alphas.append(np.ones(img.shape[:2], np.float32))
        fp = os.path.join(scene_dir, "images_flow", name + "_bwd.npz")
        if os.path.exists(fp):
            d = np.load(fp)
            flows.append(np.concatenate([d["flow"].astype(np.float32), d["mask"].astype(np.float32)[..., None]], -1))
            any_flow = True
        else:
            flows.append(np.zeros(img.shape[:2] + (3,), np.float32))
    return {"frames": list(frames), "images": np.stack(images, 0), "alphas": np.stack(alphas, 0),
            "flows": np.stack(flows, 0) if any_flow else None}


def write_scene_pixels(scene_dir: str, px: Dict):
    """The inverse of `load_scene_pixels` for synthetic scenes (tests, the launcher's synthetic mode): 8-bit PNGs + flow archives."""
    from PIL import Image
    for sub in ("images", "masks", "images_flow"):
        os.makedirs(os.path.join(scene_dir, sub), exist_ok=True)
    for i, name in enumerate(px["frames"]):
        Image.fromarray((np.clip(px["images"][i], 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)).save(os.path.join(scene_dir, "images", name + ".png"))
        Image.fromarray((np.clip(px["alphas"][i], 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)).save(os.path.join(scene_dir, "masks", name + ".png"))
        if px.get("flows") is not None:
            np.savez(os.path.join(scene_dir, "images_flow", name + "_bwd.npz"), flow=px["flows"][i][..., :2], mask=px["flows"][i][..., 2])


def save_image(path: str, rendered: torch.Tensor, H: int, W: int):
    """`to_8b_image` + `Image.fromarray(...).save` (model.py:1471-1486)."""
    from PIL import Image
    from .eval import to_8b_image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(to_8b_image(rendered.view(H, W, 3)).cpu().numpy()).save(path)
