"""Deterministic synthetic weights and ray batches for the BASELINE.json configurations.

There is no dataset and no checkpoint in the build/bench environment, so every measurement runs
on random-init weights of the reference architecture and seeded synthetic rays of the reference
batch-dict shapes.  numpy RandomState is used (not torch RNG) so
the same seed gives bit-identical tensors on every torch build -- the golden fixtures under
tests/golden depend on that.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

# ------------------------------------------------------------------ background weights


def _linear(rs: np.random.RandomState, n_out: int, n_in: int, kaiming: bool = True):
    """nn.Linear default bias init + kaiming_uniform_(a=0) weight (M:174-209)."""
    wb = math.sqrt(6.0 / n_in) if kaiming else 1.0 / math.sqrt(n_in)
    bb = 1.0 / math.sqrt(n_in)
    w = rs.uniform(-wb, wb, size=(n_out, n_in)).astype(np.float32)
    b = rs.uniform(-bb, bb, size=(n_out,)).astype(np.float32)
    return torch.from_numpy(w), torch.from_numpy(b)


def mlp_state_dict(rs, prefix: str, depth: int, width: int, n_states: int, rgb: bool,
                   pos_size: int = 568, skip: int = 4) -> Dict[str, torch.Tensor]:
    sd = {}
    fan = pos_size
    for i in range(depth):
        w, b = _linear(rs, width, fan)
        sd[f"{prefix}pts_linear.{i}.weight"], sd[f"{prefix}pts_linear.{i}.bias"] = w, b
        fan = width + pos_size if (i % skip == 0 and i > 0) else width
    sd[f"{prefix}density_layer.weight"], sd[f"{prefix}density_layer.bias"] = _linear(rs, 1, fan)
    if rgb:
        sd[f"{prefix}bottleneck_layer.weight"], sd[f"{prefix}bottleneck_layer.bias"] = _linear(rs, 256, fan)
        sd[f"{prefix}views_linear.0.weight"], sd[f"{prefix}views_linear.0.bias"] = _linear(rs, 128, 256 + 27)
        sd[f"{prefix}rgb_layer.weight"], sd[f"{prefix}rgb_layer.bias"] = _linear(rs, 3, 128)
    for k in range(n_states):
        sd[f"{prefix}bkgd_stateembeds.{k}"] = torch.from_numpy(rs.standard_normal(64).astype(np.float32))
    return sd


def background_state_dict(seed: int = 777, n_states: int = 2, prop_width: int = 256, nerf_width: int = 1024,
                          prop_depth: int = 4, nerf_depth: int = 8) -> Dict[str, torch.Tensor]:
    """state_dict of the reference `MipNeRF360` (keys `mlps.{0,1,2}.*`)."""
    rs = np.random.RandomState(seed)
    sd = {}
    sd.update(mlp_state_dict(rs, "mlps.0.", prop_depth, prop_width, n_states, rgb=False))
    sd.update(mlp_state_dict(rs, "mlps.1.", prop_depth, prop_width, n_states, rgb=False))
    sd.update(mlp_state_dict(rs, "mlps.2.", nerf_depth, nerf_width, n_states, rgb=True))
    return sd


# ------------------------------------------------------------------ stage-1 rays (config 2)


def stage1_batch(num_rays: int = 1024, seed: int = 777, time: float = 0.5) -> Dict[str, torch.Tensor]:
    """o ~ N(0,0.1^2), d unit, radii = 1e-3*U(0.5,2), viewdirs == rays_d."""
    rs = np.random.RandomState(seed)
    o = (rs.standard_normal((num_rays, 3)) * 0.1).astype(np.float32)
    d = rs.standard_normal((num_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    radii = (1e-3 * rs.uniform(0.5, 2.0, size=(num_rays, 1))).astype(np.float32)
    target = rs.uniform(0, 1, size=(num_rays, 3)).astype(np.float32)
    t = torch.from_numpy
    return {
        "rays_o": t(o), "rays_d": t(d), "viewdirs": t(d.copy()), "radii": t(radii),
        "times": torch.full((num_rays,), float(time)), "target": t(target),
    }


def pinhole_batch(hw: int = 64, seed: int = 777, time: float = 0.5) -> Dict[str, torch.Tensor]:
    """hw x hw pin-hole camera, f = 1.2*hw, at z=+1 looking at the origin."""
    f = 1.2 * hw
    j, i = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    dirs = np.stack([(i - hw / 2 + 0.5) / f, -(j - hw / 2 + 0.5) / f, -np.ones_like(i)], -1)
    d = dirs.reshape(-1, 3)
    dx = np.linalg.norm(dirs[:-1] - dirs[1:], axis=-1)
    dx = np.concatenate([dx, dx[-1:]], 0).reshape(-1, 1)
    radii = (dx * 2 / math.sqrt(12)).astype(np.float32)
    dn = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(np.array([0, 0, 1], np.float32), d.shape).copy()
    rs = np.random.RandomState(seed)
    target = rs.uniform(0, 1, size=d.shape).astype(np.float32)
    t = torch.from_numpy
    return {
        "rays_o": t(o), "rays_d": t(dn.astype(np.float32)), "viewdirs": t(dn.astype(np.float32).copy()),
        "radii": t(radii), "times": torch.full((d.shape[0],), float(time)), "target": t(target),
    }


# ------------------------------------------------------------------ human-object branch (configs 3/4)
SMPL_PARENT = {1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 13: 9, 14: 9, 15: 12,
               16: 13, 17: 14, 18: 16, 19: 17, 20: 18, 21: 19, 22: 20, 23: 21, 24: 23, 25: 22}
TORSO = (0, 3, 6, 9, 13, 14)


def tpose_joints() -> np.ndarray:
    """A fixed SMPL-like T-pose (metres) with the two object joints extrapolated from the hands
    like the reference does (train.py:132-145): J24 = J23 + (J23 - J19), J25 = J22 + (J22 - J18)."""
    J = np.array([
        (0.00, 0.00, 0.00), (0.07, -0.09, 0.00), (-0.07, -0.09, 0.00), (0.00, 0.11, -0.02),
        (0.10, -0.47, 0.00), (-0.10, -0.47, 0.00), (0.00, 0.25, 0.00), (0.09, -0.87, -0.03),
        (-0.09, -0.87, -0.03), (0.00, 0.30, 0.02), (0.11, -0.93, 0.09), (-0.11, -0.93, 0.09),
        (0.00, 0.51, -0.01), (0.08, 0.42, 0.00), (-0.08, 0.42, 0.00), (0.00, 0.60, 0.04),
        (0.19, 0.45, -0.01), (-0.19, 0.45, -0.01), (0.45, 0.44, -0.03), (-0.45, 0.44, -0.03),
        (0.70, 0.45, -0.03), (-0.70, 0.45, -0.03), (0.79, 0.44, -0.04), (-0.79, 0.44, -0.04)], dtype=np.float32)
    obj_r = J[23] + (J[23] - J[19])
    obj_l = J[22] + (J[22] - J[18])
    return np.concatenate([J, obj_r[None], obj_l[None]], 0).astype(np.float32)


def _axis_angle_to_matrix(r: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(r))
    k = r / (th + 1e-5)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.float64)
    return (math.cos(th) * np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * np.outer(k, k)).astype(np.float32)


def _rotation_between(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = a / (np.linalg.norm(a) + 1e-12)
    b = b / (np.linalg.norm(b) + 1e-12)
    v = np.cross(a, b)
    s, c = np.linalg.norm(v), float(a @ b)
    if s < 1e-8:
        return np.eye(3, dtype=np.float32) if c > 0 else np.diag([1.0, -1.0, -1.0]).astype(np.float32)
    return _axis_angle_to_matrix(v / s * math.atan2(s, c))


def bone_prior_volume(joints: np.ndarray, bmin: np.ndarray, bmax: np.ndarray, V: int = 32) -> np.ndarray:
    """Gaussian bone-occupancy prior [K+1,V,V,V] (last channel = background), strictly positive,
    channel-normalised -- same construction idea as body_util.approx_gaussian_bone_volumes
    (anisotropic Gaussians along bones, isotropic at leaf joints); volume axes are (z,y,x)."""
    K = joints.shape[0]
    zz, yy, xx = np.meshgrid(np.linspace(bmin[2], bmax[2], V), np.linspace(bmin[1], bmax[1], V),
                             np.linspace(bmin[0], bmax[0], V), indexing="ij")
    grid = np.stack([xx, yy, zz], -1).astype(np.float32)
    vols = []
    for j in range(K):
        vol = np.zeros((V, V, V), np.float32)
        kids = [c for c, p in SMPL_PARENT.items() if p == j]
        for c in kids:
            std = np.array([0.03, 0.06, 0.03]) * 2.0
            S = np.diag(1.0 / std)
            if j in TORSO:
                S[0, 0] /= 1.5
                S[2, 2] /= 1.5
            a, b = joints[j], joints[c]
            R = _rotation_between(np.array([0.0, 1.0, 0.0]), b - a)
            Sig = R @ S @ S @ R.T
            d = grid - (a + b) / 2.0
            vol += np.exp(-np.einsum("...i,ij,...j->...", d, Sig, d)).astype(np.float32)
        if not kids:
            std = (0.06 if j in (15, 24, 25) else 0.02) * 2.0
            d = grid - joints[j]
            vol = np.exp(-np.sum(d * d, -1) / (std * std)).astype(np.float32)
        vols.append(vol)
    vols = np.stack(vols, 0)
    bg = 1.0 - np.clip(vols.sum(0, keepdims=True), 0.0, 1.0)
    vols = np.concatenate([vols, bg], 0)
    vols = vols / np.clip(vols.sum(0, keepdims=True), 0.001, None)
    return np.clip(vols, 1e-6, None).astype(np.float32)       # log(prior) must stay finite


def _ray_bbox(o, d, bmin, bmax):
    inv = 1.0 / np.where(np.abs(d) < 1e-9, 1e-9, d)
    t0, t1 = (bmin - o) * inv, (bmax - o) * inv
    near = np.max(np.minimum(t0, t1), -1)
    far = np.min(np.maximum(t0, t1), -1)
    return near, far


def human_batch(num_rays: int = 2048, seed: int = 777, time: float = 0.5, is_train: bool = True,
                iter_val: float = 3e5, pose_sigma: float = 0.2) -> Dict[str, torch.Tensor]:
    """SURVEY 8(d) configs 3/4 + Appendix B: one synthetic training item of the human-object branch
    (reference kwargs of Network.forward) plus the stage-3 extras (`*_bkg`, `radii`, similarity)."""
    rs = np.random.RandomState(seed)
    J = tpose_joints()
    K = J.shape[0]
    pose = (rs.standard_normal((K, 3)) * pose_sigma).astype(np.float32)
    pose[24:] = 0.0
    pose_prev = pose + (rs.standard_normal((K, 3)) * 0.02).astype(np.float32)
    pose_prev[24:] = 0.0

    def rts(p):
        Rs = np.stack([_axis_angle_to_matrix(p[i]) for i in range(K)], 0)
        Ts = np.stack([J[0]] + [J[i] - J[SMPL_PARENT[i]] for i in range(1, K)], 0).astype(np.float32)
        return Rs.astype(np.float32), Ts

    dst_Rs, dst_Ts = rts(pose)
    dst_Rs_prev, dst_Ts_prev = rts(pose_prev)
    gt = np.zeros((K, 4, 4), np.float32)
    gt[:, :3, :3] = np.eye(3)
    gt[:, 3, 3] = 1.0
    gt[:, :3, 3] = J                                   # chain of pure translations
    bmin, bmax = J.min(0) - 0.6, J.max(0) + 0.6
    prior = bone_prior_volume(J, bmin, bmax)

    # posed joints (for aiming rays at the body) via the kinematic chain
    G = [np.block([[dst_Rs[0], dst_Ts[0][:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])]
    for i in range(1, K):
        L = np.block([[dst_Rs[i], dst_Ts[i][:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])
        G.append(G[SMPL_PARENT[i]] @ L)
    posed = np.stack([g[:3, 3] for g in G], 0).astype(np.float32)
    pmin, pmax = posed.min(0) - 0.6, posed.max(0) + 0.6

    cam = np.array([0.3, 0.2, 3.0], np.float32)
    tgt = posed[rs.randint(0, K, size=num_rays)] + (rs.standard_normal((num_rays, 3)) * 0.25).astype(np.float32)
    d = tgt - cam
    d = d / np.abs(d[:, 2:3])                           # pixel-style directions: |d_z| = 1, not unit length
    o = np.broadcast_to(cam, d.shape).copy()
    near, far = _ray_bbox(o, d, pmin, pmax)
    near = np.maximum(near, 0.0)
    bad = far <= near
    far = np.where(bad, near + 1.0, far)

    # stage-3 extras: the same rays in the (scaled) background world
    s = 0.2
    Rw = _axis_angle_to_matrix(np.array([0.1, -0.4, 0.05], np.float32))
    A = np.eye(4, dtype=np.float32)
    A[:3, :3] = s * Rw
    A[:3, 3] = np.array([0.05, -0.02, 0.1], np.float32)
    o_b = o @ A[:3, :3].T + A[:3, 3]
    d_b = d @ A[:3, :3].T
    vd = d_b / np.linalg.norm(d_b, axis=-1, keepdims=True)
    radii = (1e-3 * rs.uniform(0.5, 2.0, size=(num_rays, 1))).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return {
        "rays": torch.stack([t(o), t(d)], 0), "near": t(near[:, None]), "far": t(far[:, None]),
        "dst_Rs": t(dst_Rs), "dst_Ts": t(dst_Ts), "cnl_gtfms": t(gt), "motion_weights_priors": t(prior),
        "dst_posevec": t(pose[1:].reshape(-1) + 0.01), "dst_Rs_prev": t(dst_Rs_prev), "dst_Ts_prev": t(dst_Ts_prev),
        "dst_posevec_prev": t(pose_prev[1:].reshape(-1) + 0.01),
        "cnl_bbox_min_xyz": t(bmin), "cnl_bbox_max_xyz": t(bmax), "cnl_bbox_scale_xyz": t(2.0 / (bmax - bmin)),
        "bgcolor": t(rs.uniform(0, 255, size=3)), "time": torch.tensor(float(time)), "is_train": is_train,
        "iter_val": torch.full((1,), float(iter_val)),
        "rays_o_bkg": t(o_b), "rays_d_bkg": t(d_b), "viewdirs_bkg": t(vd), "radii": t(radii),
        "newsmpl_to_scale_world": t(A), "target_rgbs": t(rs.uniform(0, 1, size=(num_rays, 3))),
        "canonical_joints": t(J), "dst_bbox_min_xyz": t(pmin), "dst_bbox_max_xyz": t(pmax),
    }


def eval_camera(H: int, W: int, hb: Dict[str, torch.Tensor]):
    """A synthetic evaluation camera for the subject of `human_batch`: pinhole K, the SMPL-space extrinsics E and the
    background-world extrinsics E_colmap that see the same pixels (what freeview.py:226-239 derives from the dataset's
    cameras), as float64 host arrays."""
    f = 0.45 * H                                          # wide enough that the box covers about a third of the frame
    K = np.array([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]])
    c = np.array([0.3, 0.2, 3.0])                         # the camera centre `human_batch` aims its rays from
    R = np.diag([1.0, -1.0, -1.0])                        # x right, y down, z forward = -z of the SMPL frame
    E = np.eye(4); E[:3, :3] = R; E[:3, 3] = -R @ c
    A = hb["newsmpl_to_scale_world"].double().numpy()
    s = np.cbrt(np.linalg.det(A[:3, :3]))
    Rw = A[:3, :3] / s
    Rc = R @ Rw.T
    cb = A[:3, :3] @ c + A[:3, 3]
    Ec = np.eye(4); Ec[:3, :3] = Rc; Ec[:3, 3] = -Rc @ cb
    return K, E, Ec


def human_state_dict(seed: int = 777, n_states: int = 2, small_decoder_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random-init state_dict of the reference human `Network` (64.67 M params; key names of SURVEY
    section 5).  Xavier-uniform with ReLU gain like network_util.initseq; last layers of the offset /
    pose heads use +-`off` (the reference uses 1e-5; a larger value exercises those paths in tests)."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, n_out, n_in, gain=math.sqrt(2.0), bound=None):
        b = gain * math.sqrt(2.0 / (n_in + n_out)) * math.sqrt(3.0) if bound is None else bound
        sd[name + ".weight"] = torch.from_numpy(rs.uniform(-b, b, size=(n_out, n_in)).astype(np.float32))
        sd[name + ".bias"] = torch.zeros(n_out)

    # motion-weight volume decoder (U:21-59): Linear 256->1024, 5 ConvTranspose3d
    sd["mweight_vol_decoder.const_embedding"] = torch.from_numpy(rs.standard_normal(256).astype(np.float32))
    lg = math.sqrt(2.0 / (1 + 0.2**2))
    lin("mweight_vol_decoder.decoder.block_mlp.0", 1024, 256, gain=lg)
    chans = [(1024, 512), (512, 512), (512, 256), (256, 256), (256, 27)]
    for n, (ci, co) in enumerate(chans):
        gain = lg if n < len(chans) - 1 else 1.0
        std = gain * math.sqrt(2.0 / ((ci + co) * 8.0))          # ksize = 4^3 / 2^3
        b = std * math.sqrt(3.0)
        w = rs.uniform(-b, b, size=(ci, co, 2, 2, 2)).astype(np.float32)
        # blockwise init (U:283-297): every 2x2x2 phase copies the (0::2,0::2,0::2) entries
        full = np.zeros((ci, co, 4, 4, 4), np.float32)
        for a in range(2):
            for bb in range(2):
                for c in range(2):
                    full[:, :, a::2, bb::2, c::2] = w[:, :, 0:2, 0:2, 0:2]
        sd[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.weight"] = torch.from_numpy(full * small_decoder_scale)
        sd[f"mweight_vol_decoder.decoder.block_conv.{2 * n}.bias"] = torch.zeros(co)
    for pre in ("non_rigid_mlp", "non_rigid_forward_mlp"):
        for i in range(6):
            lin(f"{pre}.block_mlps.{2 * i}", 128, 164 if i == 4 else (111 if i == 0 else 128))
        lin(f"{pre}.block_mlps.12", 3, 128, bound=1e-2)
    for k in range(n_states):
        sd[f"human_stateembeds.{k}"] = torch.from_numpy(rs.standard_normal(64).astype(np.float32))
    for i in range(8):
        lin(f"cnl_mlp.pts_linears.{2 * i}", 256, 127 if i == 0 else (383 if i == 5 else 256))
    lin("cnl_mlp.output_linear.0", 4, 256, gain=1.0)
    lin("pose_decoder.block_mlps.0", 256, 75)
    lin("pose_decoder.block_mlps.2", 256, 256)
    lin("pose_decoder.block_mlps.4", 256, 256)
    for head in ("dstR", "dstT"):
        lin(f"pose_decoder.block_mlps_{head}.0", 256, 256)
        lin(f"pose_decoder.block_mlps_{head}.2", 75, 256, bound=1e-3)
    return sd


def add_patch_supervision(b: Dict[str, torch.Tensor], n_patches: int = 2, size: int = 32, seed: int = 777) -> Dict[str, torch.Tensor]:
    """The supervision part of a stage-2 / stage-3 training item (S2/core/data/human_nerf/train.py:
    405-455, 523-586) for the rays of `human_batch`: `patch_masks` [Np,size,size] bool with exactly as many set pixels as
    there are rays (full patches when B = Np*size*size, otherwise a random subset per patch -- the reference's patches
    keep only the pixels whose rays hit the subject's box), `target_patches`, `patch_div_indices`, and the flow inputs
    `ray_grid` [B,5] = (x, y, flow_x, flow_y, valid), `newsmpl_to_camera_prev`, `intrinsics_prev`."""
    rs = np.random.RandomState(seed + 1000)
    B = b["near"].shape[0]
    per = [B // n_patches + (1 if i < B % n_patches else 0) for i in range(n_patches)]
    assert max(per) <= size * size, "more rays than patch pixels"
    masks = np.zeros((n_patches, size * size), bool)
    for i, n in enumerate(per):
        masks[i, np.sort(rs.permutation(size * size)[:n])] = True
    out = dict(b)
    out["patch_masks"] = torch.from_numpy(masks.reshape(n_patches, size, size))
    out["target_patches"] = torch.from_numpy(rs.uniform(0, 1, size=(n_patches, size, size, 3)).astype(np.float32))
    out["patch_div_indices"] = torch.from_numpy(np.concatenate([[0], np.cumsum(per)]).astype(np.int64))
    out.pop("target_rgbs", None)
    grid = np.concatenate([rs.uniform(0, 100, size=(B, 2)), rs.standard_normal((B, 2)), (rs.uniform(size=(B, 1)) > 0.2).astype(np.float64)], -1)
    out["ray_grid"] = torch.from_numpy(grid.astype(np.float32))
    cam = np.eye(4, dtype=np.float32)
    cam[:3, :3] = np.diag([1.0, -1.0, -1.0])                # x right, y down, z forward, at the camera `human_batch` aims from
    cam[:3, 3] = -cam[:3, :3] @ np.array([0.3, 0.2, 3.0], np.float32)
    out["newsmpl_to_camera_prev"] = torch.from_numpy(cam)
    out["intrinsics_prev"] = torch.tensor([[500.0, 0.0, 50.0], [0.0, 500.0, 50.0], [0.0, 0.0, 1.0]])
    return out


# ------------------------------------------------------------------ a whole scene DIRECTORY
def write_scene_dir(path: str, n_frames: int = 16, H: int = 96, W: int = 96, seed: int = 777):
    """Write a geometrically consistent synthetic scene in the on-disk formats the stages read -- `poses_bounds.npy` (LLFF rows),
    `cameras.pkl`, `mesh_infos.pkl`, `canonical_joints.pkl`, `transitions_times.json` -- and return the per-frame arrays a
    loader would decode from the image files: {"images" [N,H,W,3] 0..1, "alphas" [N,H,W], "flows" [N,H,W,3], "frames"}.
    A camera orbits a posed SMPL-like subject that stands somewhere in a 'world' frame; every frame has its own pose, global
    orientation Rh and translation Th.  Consistency that the pipeline relies on: smpl_to_camera = world_to_camera @ smpl_to_world."""
    import json
    import os
    import pickle
    rs = np.random.RandomState(seed)
    J24 = tpose_joints()[:24]
    Rw = _axis_angle_to_matrix(np.array([0.2, 0.5, -0.1], np.float32)).astype(np.float64)       # SMPL frame -> world
    smpl_to_world = np.eye(4)
    smpl_to_world[:3, :3] = Rw
    smpl_to_world[:3, 3] = np.array([0.4, -0.3, 0.8])
    f = 1.1 * H
    K = np.array([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]], dtype=np.float32)
    names = [f"frame_{i:06d}" for i in range(n_frames)]
    rows, cams, infos = [], {}, {}
    base_pose = (rs.standard_normal((24, 3)) * 0.15).astype(np.float32)
    for i, name in enumerate(names):
        pose = base_pose + (rs.standard_normal((24, 3)) * 0.03).astype(np.float32)
        Rh = np.array([0.05 * i, 0.3 + 0.04 * i, 0.0], np.float32)
        Th = np.array([0.02 * i, 0.9, 0.01 * i], np.float32)
        pose[0] = Rh
        # posed joints in SMPL space: kinematic chain in the body frame, then the global transform
        Rs = np.stack([_axis_angle_to_matrix(pose[k]) if k else np.eye(3, dtype=np.float32) for k in range(24)], 0)
        G = [np.block([[Rs[0], J24[0][:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])]
        for k in range(1, 24):
            L = np.block([[Rs[k], (J24[k] - J24[SMPL_PARENT[k]])[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])
            G.append(G[SMPL_PARENT[k]] @ L)
        body = np.stack([g[:3, 3] for g in G], 0)
        poses72 = pose.reshape(-1).copy()
        # the stages pose the skeleton in the BODY frame (global orientation removed: `dst_poses[:3]` still holds Rh, but the
        # joints / box they use are these) -- mesh_infos stores the body-frame joints
        infos[name] = {"poses": poses72.astype(np.float32), "tpose_joints": J24.astype(np.float32), "joints": body.astype(np.float32),
                       "Rh": Rh, "Th": Th}
        # camera: orbit around the subject's position in the world, looking at it (OpenCV axes: x right, y down, z forward)
        centre_world = smpl_to_world[:3, :3] @ (_axis_angle_to_matrix(Rh).astype(np.float64) @ np.array([0.0, 0.1, 0.0]) + Th) + smpl_to_world[:3, 3]
        ang = 2.0 * math.pi * i / n_frames * 0.35
        up_w = smpl_to_world[:3, :3] @ np.array([0.0, 1.0, 0.0])
        side = smpl_to_world[:3, :3] @ np.array([math.sin(ang), 0.1, math.cos(ang)])
        cam_c = centre_world + 3.2 * side / np.linalg.norm(side)
        fwd = centre_world - cam_c
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, up_w)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, cam_c
        w2c = np.linalg.inv(c2w)
        cams[name] = {"intrinsics": K.copy(), "smpl_to_camera": (w2c @ smpl_to_world).astype(np.float32),
                      "smpl_to_world": smpl_to_world.astype(np.float32)}
        # LLFF row: columns (down, right, back) of the camera-to-world rotation | centre | (H, W, focal); then near / far bounds
        llff = np.concatenate([np.stack([down, right, -fwd, cam_c], 1), np.array([[H], [W], [f]])], 1)
        rows.append(np.concatenate([llff.reshape(-1), [0.5, 20.0]]))
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, "poses_bounds.npy"), np.stack(rows, 0))
    for fname, obj in (("cameras.pkl", cams), ("mesh_infos.pkl", infos), ("canonical_joints.pkl", {"joints": J24.astype(np.float32)})):
        with open(os.path.join(path, fname), "wb") as fh:
            pickle.dump(obj, fh)
    with open(os.path.join(path, "transitions_times.json"), "w") as fh:
        json.dump({"f0": {"time": 0.4}}, fh)
    # per-frame pixels: a smooth colour field, a disc-shaped subject mask around the image centre, a small smooth flow
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    images = np.stack([np.stack([0.5 + 0.5 * np.sin(0.07 * xx + i), 0.5 + 0.5 * np.cos(0.05 * yy - i), 0.5 + 0.4 * np.sin(0.03 * (xx + yy))], -1)
                       for i in range(n_frames)], 0).astype(np.float32)
    alphas = np.stack([((xx - 0.5 * W) ** 2 / (0.16 * W) ** 2 + (yy - 0.5 * H) ** 2 / (0.3 * H) ** 2 < 1.0).astype(np.float32)
                       for _ in range(n_frames)], 0)
    flows = np.stack([np.stack([0.5 * np.sin(0.1 * yy + i), 0.5 * np.cos(0.1 * xx), (rs.rand(H, W) > 0.2).astype(np.float32)], -1)
                      for i in range(n_frames)], 0).astype(np.float32)
    return {"images": images, "alphas": alphas, "flows": flows, "frames": names}
