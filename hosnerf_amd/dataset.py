"""From a scene directory to stage-3 training items, with everything per-pixel on the device (SURVEY 8(f).1 + 8(f).4).

The reference's `Dataset.__getitem__` (3rd_Complete_HOSNeRF/core/data/human_nerf/train.py:469-727) builds, for EVERY step, two
full-image ray sets, their radii and a six-plane box test in numpy inside a DataLoader worker, and ships the selected rays
through pinned memory.  `SceneItems` yields the same dict (SURVEY Appendix B keys) from the same files --

    cameras_scaleworld.pkl   written by the stage-1 loader (`formats.load_scene`)
    mesh_infos.pkl           per-frame SMPL fits (`formats.load_mesh_infos`)
    canonical_joints.pkl     canonical T-pose (`formats.load_canonical_joints`)
    transitions_times.json   read by the networks themselves

-- but the rays, the box test and the patch gather run on the MI355X (`eval.frame_rays` -> hos_rays.hip,
`rays.sample_patch_rays`); only the 26-joint pose algebra (host numpy, once per frame) and the two random patch decisions
stay on the host.  Image decoding is the caller's (imageio / cv2 are not dependencies of this build): frames are handed over
as arrays -- `images` [N,H,W,3] in 0..1, `alphas` [N,H,W] in 0..1, optionally `flows` [N,H,W,3] = (flow_x, flow_y, mask), the
content of `images_flow/<frame>_bwd.npz`."""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import formats
from . import rays as rays_mod
from .eval import frame_rays


def smpl_frame_camera(E: np.ndarray, Rh: np.ndarray, Th: np.ndarray):
    """The camera seen from the per-frame body frame ("new SMPL" space) in which the pose has no global orientation
    (`apply_global_tfm_to_camera`, core/utils/camera_util.py:134-151): with G = Rodrigues(Rh), a body-frame point p sits at
    G p + Th in SMPL space, so  newsmpl_to_smpl = [G | Th]  and the extrinsics become E @ newsmpl_to_smpl.
    Returns (E_new [4,4], newsmpl_to_smpl [4,4])."""
    rh = np.asarray(Rh, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(rh @ rh))
    if theta < 1e-12:
        G = np.eye(3)
    else:
        k = rh / theta
        Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
        G = np.eye(3) + np.sin(theta) * Kx + (1.0 - np.cos(theta)) * (Kx @ Kx)
    M = np.eye(4)
    M[:3, :3] = G
    M[:3, 3] = np.asarray(Th, dtype=np.float64).reshape(3)
    return np.asarray(E, dtype=np.float64) @ M, M


def pixel_flow_grid(flow_xy_mask: torch.Tensor) -> torch.Tensor:
    """[H,W,3] (flow_x, flow_y, mask) -> [H*W,5] rows (x, y, flow_x, flow_y, mask): `get_grid` (train.py:39-55), on the device."""
    H, W = flow_xy_mask.shape[:2]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=flow_xy_mask.device),
                            torch.arange(W, dtype=torch.float32, device=flow_xy_mask.device), indexing="ij")
    return torch.cat([xs[..., None], ys[..., None], flow_xy_mask.float()], -1).reshape(H * W, 5)


class SceneItems:
    """Indexable source of stage-2 / stage-3 training items of one scene.  `items[i]` is the dict the reference's dataset
    returns for frame i in 'patch' ray-shoot mode, tensors on `device`, control scalars (`time`, `is_train`, sizes) on the host."""

    def __init__(self, scene_dir: str, images, alphas, flows=None, frames: Optional[Sequence[str]] = None,
                 n_patches: int = 2, patch_size: int = 32, sample_subject_ratio: float = 0.8, bbox_offset: float = 0.6,
                 volume_size: int = 32, resize_img_scale: float = 1.0, bgcolor=None, device="cuda", seed: Optional[int] = None,
                 stage: int = 3):
        """`stage=2` builds the items of the stage-2 dataset (2nd_State_Conditional_Human-Object/core/data/human_nerf/train.py:
        460-658): the frame is composited over the item's background colour with its alpha mask (T2:345), only the subject's
        rays are kept (no background-branch rays, no `newsmpl_to_scale_world`), and a patch is CUT by the subject's box
        (T2:321-332: ragged selection, `patch_masks` with holes, their pixels enter the MSE as background colour)."""
        if stage not in (2, 3):
            raise ValueError("stage must be 2 or 3")
        self.stage = stage
        with open(os.path.join(scene_dir, "cameras_scaleworld.pkl"), "rb") as f:
            self.cameras = pickle.load(f)
        self.mesh_infos = formats.load_mesh_infos(os.path.join(scene_dir, "mesh_infos.pkl"), bbox_offset)
        self.canonical_joints, self.canonical_bbox = formats.load_canonical_joints(os.path.join(scene_dir, "canonical_joints.pkl"), bbox_offset)
        self.frames: List[str] = list(frames) if frames is not None else list(self.mesh_infos.keys())
        n = len(self.frames)
        self.times = np.linspace(0.0, 1.0, n).astype(np.float32)            # train.py:120 (one time per frame of the list)
        self.device = torch.device(device)
        self.images = torch.as_tensor(np.asarray(images), dtype=torch.float32)
        self.alphas = torch.as_tensor(np.asarray(alphas), dtype=torch.float32)
        self.flows = None if flows is None else torch.as_tensor(np.asarray(flows), dtype=torch.float32)
        assert self.images.shape[0] == n and self.alphas.shape[0] == n, "one image / mask per frame"
        self.n_patches, self.patch_size, self.subject_ratio = n_patches, patch_size, sample_subject_ratio
        self.resize = resize_img_scale
        self.bgcolor = bgcolor
        self.rng = np.random.RandomState(seed) if seed is not None else np.random
        bmin, bmax = self.canonical_bbox["min_xyz"].astype("float32"), self.canonical_bbox["max_xyz"].astype("float32")
        # per-SUBJECT constants: built once and kept on the device (the reference copies the 3.5 MB prior into every item)
        self._prior = torch.from_numpy(formats.approx_gaussian_bone_volumes(self.canonical_joints, bmin, bmax, volume_size).astype("float32")).to(self.device)
        self._cnl = {"cnl_gtfms": formats.get_canonical_global_tfms(self.canonical_joints), "canonical_joints": self.canonical_joints,
                     "cnl_bbox_min_xyz": bmin, "cnl_bbox_max_xyz": bmax, "cnl_bbox_scale_xyz": 2.0 / (bmax - bmin)}
        assert np.all(self._cnl["cnl_bbox_scale_xyz"] >= 0)

    def __len__(self) -> int:
        return len(self.frames)

    def _pose(self, name: str):
        m = self.mesh_infos[name]
        Rs, Ts = formats.body_pose_to_body_RTs(m["poses"].astype("float32"), m["tpose_joints"].astype("float32"))
        return Rs, Ts, m["poses"].astype("float32")[3:] + 1e-2

    def _camera(self, name: str):
        cam, m = self.cameras[name], self.mesh_infos[name]
        K = np.array(cam["intrinsics"][:3, :3], dtype=np.float64)
        K[:2] *= self.resize
        E, newsmpl_to_smpl = smpl_frame_camera(cam["smpl_to_camera"], m["Rh"], m["Th"])
        return K, E, newsmpl_to_smpl

    def __getitem__(self, idx: int) -> Dict:
        name = self.frames[idx]
        time = float(self.times[idx])
        dev = self.device
        flow_on = time > 0.005 and self.flows is not None            # train.py:486, :560: flow supervision needs a previous frame
        bg = (self.rng.rand(3) * 255.0).astype("float32") if self.bgcolor is None else np.asarray(self.bgcolor, dtype="float32")
        img = self.images[idx].to(dev)
        H, W = int(img.shape[0]), int(img.shape[1])
        K, E, newsmpl_to_smpl = self._camera(name)
        cam = self.cameras[name]
        if self.stage == 2:
            return self._item_stage2(idx, name, time, flow_on, bg, img, H, W, K, E)
        item = frame_rays(H, W, K, E, self.mesh_infos[name]["bbox"], np.asarray(cam["scaleworld_to_camera"], dtype=np.float64), device=dev)
        rm = item["ray_mask"]
        item["ray_img"] = img.reshape(-1, 3)[rm]
        if flow_on:
            item["ray_grid"] = pixel_flow_grid(self.flows[idx].to(dev))[rm]
        item = rays_mod.sample_patch_rays(item, img, self.alphas[idx].to(dev) > 0.0, self.n_patches, self.patch_size,
                                          self.subject_ratio, self.rng)
        item.pop("ray_img", None)
        # constants of the patch MSE (`train.prepare_patch_targets`): stage-3 patches are never cut by the box (train.py:322-330),
        # so no patch pixel is filled with the background colour -- known here without reading a mask back from the device
        item["mse_const"], item["mse_count"] = 0.0, float(item["target_patches"].numel())
        Rs, Ts, posevec = self._pose(name)
        host = {"dst_Rs": Rs, "dst_Ts": Ts, "dst_posevec": posevec, "bgcolor": bg,
                "newsmpl_to_scale_world": (np.asarray(cam["smpl_to_scale_world"], dtype=np.float64) @ newsmpl_to_smpl).astype("float32"), **self._cnl}
        if time > 0.005:
            prev = self.frames[idx - 1]
            Rp, Tp, pp = self._pose(prev)
            Kp, Ep, _ = self._camera(prev)
            host.update(dst_Rs_prev=Rp, dst_Ts_prev=Tp, dst_posevec_prev=pp, newsmpl_to_camera_prev=Ep.astype("float32"),
                        intrinsics_prev=Kp.astype("float32"))
        item.update({k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in host.items()})
        item["motion_weights_priors"] = self._prior
        item.update(frame_name=name, time=time, is_train=True)
        return item


    # ------------------------------------------------------------------ full-frame evaluation items (stage 3)
    def _frame_item(self, idx: int, K, E_smpl, E_colmap, newsmpl_to_scale_world, bg, with_targets: bool) -> Dict:
        """The dict `eval.render_frame` consumes (keys of FreeviewDataset.__getitem__, freeview.py:284-335): full-image rays split by
        the box test, the frame's pose, the per-subject constants; tensors on the device."""
        name = self.frames[idx]
        dev = self.device
        img = self.images[idx].to(dev)
        H, W = int(img.shape[0]), int(img.shape[1])
        item = frame_rays(H, W, K, E_smpl, self.mesh_infos[name]["bbox"], E_colmap, device=dev)
        if with_targets:
            flat = img.reshape(-1, 3)
            item["target_rgbs"], item["target_rgbs_bkg"] = flat[item["ray_mask"]], flat[item["ray_mask_bkg"]]
        Rs, Ts, posevec = self._pose(name)
        host = {"dst_Rs": Rs, "dst_Ts": Ts, "dst_posevec": posevec, "bgcolor": np.asarray(bg, dtype="float32"),
                "newsmpl_to_scale_world": np.asarray(newsmpl_to_scale_world, dtype="float32"), **self._cnl}
        item.update({k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in host.items()})
        item["motion_weights_priors"] = self._prior
        item.update(frame_name=name, time=float(self.times[idx]), is_train=False, iter_val=torch.full((1,), 1e7))
        return item

    def eval_frame(self, idx: int, bgcolor=(255.0, 255.0, 255.0)) -> Dict:
        """Frame `idx` seen by its own camera, with the ground-truth pixels (`test_metrics`, model.py:884-1085)."""
        if self.stage != 3:
            raise ValueError("full-frame evaluation items are stage-3 items")
        name = self.frames[idx]
        cam = self.cameras[name]
        K, E, newsmpl_to_smpl = self._camera(name)
        A = np.asarray(cam["smpl_to_scale_world"], dtype=np.float64) @ newsmpl_to_smpl
        return self._frame_item(idx, K, E, np.asarray(cam["scaleworld_to_camera"], dtype=np.float64), A, bgcolor, True)

    def freeview_frame(self, idx: int, k: int, total_frames: int, inv_angle: bool = False, bgcolor=(255.0, 255.0, 255.0)) -> Dict:
        """Camera k of the `total_frames`-camera turn about the subject of frame `idx` (FreeviewDataset.__getitem__,
        freeview.py:199-337).  With T_smpl from `freeview.orbit_camera`:  the SMPL-space camera is E T_smpl, the same motion in
        the scaled world is T_world = S T_smpl S^-1 (S = smpl_to_scale_world), so the background camera is
        scaleworld_to_camera T_world and the body-frame -> scaled-world map becomes T_world^-1 S T_smpl [G | Th]."""
        from .freeview import orbit_camera
        if self.stage != 3:
            raise ValueError("free-viewpoint items are stage-3 items")
        name = self.frames[idx]
        cam, m = self.cameras[name], self.mesh_infos[name]
        K = np.array(cam["intrinsics"][:3, :3], dtype=np.float64)
        K[:2] *= self.resize
        E_k, T_smpl = orbit_camera(cam["smpl_to_camera"], k, total_frames, trans=m["Th"], inv_angle=inv_angle)
        S = np.asarray(cam["smpl_to_scale_world"], dtype=np.float64)
        T_world = S @ T_smpl @ np.linalg.inv(S)
        E_colmap = np.asarray(cam["scaleworld_to_camera"], dtype=np.float64) @ T_world
        S_k = np.linalg.inv(T_world) @ S @ T_smpl
        E_new, newsmpl_to_smpl = smpl_frame_camera(E_k, m["Rh"], m["Th"])
        return self._frame_item(idx, K, E_new, E_colmap, S_k @ newsmpl_to_smpl, bgcolor, True)

    def eval_frame_stage2(self, idx: int, bgcolor=(255.0, 255.0, 255.0)) -> Dict:
        """Stage 2's full-frame evaluation item (the 'image' ray-shoot mode of the stage-2 dataset, T2:437-455 / :585-586, which its
        `progress` / `test_metrics` loops render): EVERY ray that hits the subject's box, the frame composited over `bgcolor`,
        `target_rgbs` = the composited pixels of those rays, no flow supervision (`is_train=False`)."""
        if self.stage != 2:
            raise ValueError("eval_frame_stage2 is a stage-2 item (stage 3: eval_frame)")
        name = self.frames[idx]
        img = self.images[idx].to(self.device)
        K, E, _ = self._camera(name)
        item = self._item_stage2(idx, name, float(self.times[idx]), False, np.asarray(bgcolor, dtype="float32"), img,
                                 int(img.shape[0]), int(img.shape[1]), K, E, full_frame=True)
        item.update(is_train=False, iter_val=torch.full((1,), 1e7))
        return item

    def _item_stage2(self, idx: int, name: str, time: float, flow_on: bool, bg, img, H: int, W: int, K, E, full_frame: bool = False) -> Dict:
        """T2:460-658 with the per-pixel work on the device: composite, rays, box test, patch gather."""
        dev = self.device
        alpha = self.alphas[idx].to(dev)
        bgc = torch.from_numpy(bg).to(dev) / 255.0
        img = alpha[..., None] * img + (1.0 - alpha[..., None]) * bgc                       # T2:345 (+ the /255 of T2:479)
        o, d = rays_mod.get_rays_from_KRT(H, W, K, E[:3, :3], E[:3, 3], device=dev)           # T2:501
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        near, far, rm = rays_mod.rays_intersect_3d_bbox(self.mesh_infos[name]["bbox"], o, d)  # T2:509
        item = {"img_width": W, "img_height": H, "ray_mask": rm, "rays": torch.stack([o[rm], d[rm]], 0),
                "near": near[:, None], "far": far[:, None], "ray_img": img.reshape(-1, 3)[rm]}
        if flow_on:
            item["ray_grid"] = pixel_flow_grid(self.flows[idx].to(dev))[rm]
        if full_frame:
            item["target_rgbs"] = item.pop("ray_img")
        else:
            item = rays_mod.sample_patch_rays(item, img, alpha > 0.0, self.n_patches, self.patch_size, self.subject_ratio, self.rng,
                                              cut_by_box=True)
            item.pop("ray_img", None)
            # constants of the patch MSE (`train.prepare_patch_targets`, M2:41-50): the cut pixels are filled with the background colour
            pm, tp = item["patch_masks"], item["target_patches"]
            item["mse_const"] = float((((bgc.expand(tp.shape) - tp) ** 2)[~pm]).sum())
            item["mse_count"] = float(tp.numel())
        Rs, Ts, posevec = self._pose(name)
        host = {"dst_Rs": Rs, "dst_Ts": Ts, "dst_posevec": posevec, "bgcolor": bg, **self._cnl}
        if time > 0.005:
            prev = self.frames[idx - 1]
            Rp, Tp, pp = self._pose(prev)
            Kp, Ep, _ = self._camera(prev)
            host.update(dst_Rs_prev=Rp, dst_Ts_prev=Tp, dst_posevec_prev=pp, newsmpl_to_camera_prev=Ep.astype("float32"),
                        intrinsics_prev=Kp.astype("float32"))
        item.update({k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in host.items()})
        item["motion_weights_priors"] = self._prior
        item.update(frame_name=name, time=time, is_train=True)
        return item
