"""Oracle (CPU, torch fp32) for the human-object branch and the stage-3 composite.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Abbreviations:
  N: = 3rd_Complete_HOSNeRF/core/nets/human_nerf/network.py       (S3)
  N2: = 2nd_State_Conditional_Human-Object/core/nets/human_nerf/network.py
  U: = 3rd_Complete_HOSNeRF/core/utils/network_util.py
  M: = 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py
Row ids (P1..P10, C1..C4) are the rows of SURVEY.md section 8(a).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .background import select_state

SMPL_PARENT = {1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 13: 9, 14: 9, 15: 12,
               16: 13, 17: 14, 18: 16, 19: 17, 20: 18, 21: 19, 22: 20, 23: 21, 24: 23, 25: 22}   # U:100-103

DEFAULT_CFG = dict(total_bones=26, N_samples=128, perturb=0.0, pose_kick_in_iter=20000,
                   nonrigid_kick_in_iter=100000, nonrigid_full_band_iter=200000,
                   nonrigid_multires=6, canonical_multires=10, volume_size=32)


# ----------------------------------------------------------------------------- P2
def rodrigues(rvec: torch.Tensor) -> torch.Tensor:
    """U:66-92: theta = sqrt(1e-5 + |r|^2), r/theta, standard Rodrigues rotation matrix."""
    theta = torch.sqrt(1e-5 + torch.sum(rvec**2, dim=1))
    r = rvec / theta[:, None]
    c, s = torch.cos(theta), torch.sin(theta)
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    rows = [x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s,
            x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s,
            x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def _seq(sd, prefix, x, idxs, last_act=False):
    for n, i in enumerate(idxs):
        x = F.linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"])
        if n < len(idxs) - 1 or last_act:
            x = torch.relu(x)
    return x


def pose_refiner(sd, posevec: torch.Tensor, prefix="pose_decoder.", total_bones=26):
    """pose_decoders/mlp_delta_body_pose.py:14-73 (mlp_depth=4): trunk 75->256->256->256 (ReLU each),
    heads 256->256(ReLU)->75; Rs via Rodrigues, Ts raw."""
    h = _seq(sd, prefix + "block_mlps", posevec, [0, 2, 4], last_act=True)
    rvec = _seq(sd, prefix + "block_mlps_dstR", h, [0, 2]).view(-1, 3)
    Rs = rodrigues(rvec).view(-1, total_bones - 1, 3, 3)
    Ts = _seq(sd, prefix + "block_mlps_dstT", h, [0, 2]).view(-1, total_bones - 1, 3)
    return Rs, Ts


# ----------------------------------------------------------------------------- P3
def motion_basis(dst_Rs, dst_Ts, cnl_gtfms):
    """U:134-174: kinematic chain -> backward (cnl <- dst) and forward (dst <- cnl) rigid maps.
    Inputs [K,3,3], [K,3], [K,4,4]; returns (R_bwd [K,3,3], T_bwd [K,3], R_fwd, T_fwd)."""
    K = dst_Rs.shape[0]
    G = torch.zeros(K, 4, 4, dtype=dst_Rs.dtype, device=dst_Rs.device)
    G[:, :3, :3] = dst_Rs
    G[:, :3, 3] = dst_Ts
    G[:, 3, 3] = 1.0
    chain = [G[0]]
    for i in range(1, K):
        chain.append(chain[SMPL_PARENT[i]] @ G[i])
    dst = torch.stack(chain, 0)
    bwd = cnl_gtfms @ torch.inverse(dst)
    fwd = dst @ torch.inverse(cnl_gtfms)
    return bwd[:, :3, :3], bwd[:, :3, 3], fwd[:, :3, :3], fwd[:, :3, 3]


# ----------------------------------------------------------------------------- P4
def motion_weight_volume(sd, priors: torch.Tensor, prefix="mweight_vol_decoder."):
    """mweight_vol_decoders/deconv_vol_decoder.py:34-42 + U:21-59: softmax_ch(deconv(embedding) + log prior).
    priors [27,V,V,V] -> [27,V,V,V]."""
    h = F.leaky_relu(F.linear(sd[prefix + "const_embedding"][None], sd[prefix + "decoder.block_mlp.0.weight"],
                              sd[prefix + "decoder.block_mlp.0.bias"]), 0.2).view(-1, 1024, 1, 1, 1)
    idx = sorted({int(k.split(".")[3]) for k in sd if k.startswith(prefix + "decoder.block_conv.")})
    for n, i in enumerate(idx):
        h = F.conv_transpose3d(h, sd[f"{prefix}decoder.block_conv.{i}.weight"], sd[f"{prefix}decoder.block_conv.{i}.bias"],
                               stride=2, padding=1)
        if n < len(idx) - 1:
            h = F.leaky_relu(h, 0.2)
    return F.softmax(h + torch.log(priors[None]), dim=1)[0]


# ----------------------------------------------------------------------------- P5
def samples_along_ray(near, far, N_samples: int, t_rand: Optional[torch.Tensor] = None):
    """N:409-424: z = near(1-t)+far t ; optional stratified jitter with the given uniform draws [B,N]."""
    t = torch.linspace(0.0, 1.0, steps=N_samples).to(near)
    z = (near * (1.0 - t) + far * t).expand(near.shape[0], N_samples)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


# ----------------------------------------------------------------------------- P6 / P9
def trilinear_sample(vol: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """F.grid_sample(vol[None], grid[None,None,None], padding_mode='zeros', align_corners=True) restated.

    vol [C,D,H,W]; grid [P,3] = (x,y,z) in [-1,1] addressing (W,H,D).  Returns [P,C].
    index = (g+1)/2*(size-1); the 8 corner taps are weighted trilinearly, taps outside the volume
    contribute 0 (published PyTorch semantics of grid_sample, 5-D bilinear mode).
    """
    C, D, H, W = vol.shape
    ix = (grid[:, 0] + 1) / 2 * (W - 1)
    iy = (grid[:, 1] + 1) / 2 * (H - 1)
    iz = (grid[:, 2] + 1) / 2 * (D - 1)
    x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
    out = torch.zeros(grid.shape[0], C, dtype=vol.dtype, device=vol.device)
    flat = vol.reshape(C, -1)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                wx = (ix - x0) if dx else (x0 + 1 - ix)
                wy = (iy - y0) if dy else (y0 + 1 - iy)
                wz = (iz - z0) if dz else (z0 + 1 - iz)
                ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1) & (zi >= 0) & (zi <= D - 1)
                lin = (zi.clamp(0, D - 1) * H + yi.clamp(0, H - 1)) * W + xi.clamp(0, W - 1)
                tap = flat[:, lin.long()].T
                out = out + torch.where(ok[:, None], tap * (wx * wy * wz)[:, None], torch.zeros((), device=vol.device))
    return out


def backward_lbs(pts, R, T, vol, bbox_min, bbox_scale):
    """N:304-355 _sample_motion_fields: pts [P,3] -> (x_skel [P,3], mask [P,1]).  vol [K+1,V,V,V]."""
    K = vol.shape[0] - 1
    ws, xs = [], []
    for i in range(K):
        pos = (R[i] @ pts.T).T + T[i]
        g = (pos - bbox_min[None]) * bbox_scale[None] - 1.0
        ws.append(trilinear_sample(vol[i:i + 1], g))
        xs.append(pos)
    w = torch.cat(ws, -1)                                    # [P,K]
    wsum = w.sum(-1, keepdim=True)
    x_skel = sum(w[:, i:i + 1] * xs[i] for i in range(K)) / wsum.clamp(min=1e-4)
    return x_skel, wsum


def forward_lbs(cnl_pts, R_f, T_f, vol, bbox_min, bbox_scale):
    """N:357-399 _sample_motion_fields_forward: one K-channel tap at the canonical point."""
    K = vol.shape[0] - 1
    g = (cnl_pts - bbox_min[None]) * bbox_scale[None] - 1.0
    w = trilinear_sample(vol[:K], g)
    wsum = w.sum(-1, keepdim=True)
    x = sum(w[:, i:i + 1] * ((R_f[i] @ cnl_pts.T).T + T_f[i]) for i in range(K)) / wsum.clamp(min=1e-4)
    return x


# ----------------------------------------------------------------------------- P7 / P8 embedders + MLPs
def hannw_weights(iter_val: float, kick_in: float, full_band: float, multires: int) -> torch.Tensor:
    """embedders/hannw_fourier.py:29-44: w_j = (1 - cos(pi clamp(alpha - j, 0, 1)))/2, alpha = m t / N."""
    t = torch.clamp(torch.as_tensor(float(iter_val)) - torch.tensor(float(kick_in)), min=0.0)
    N = full_band - torch.tensor(float(kick_in))
    alpha = multires * t / N
    j = torch.arange(multires, dtype=torch.float32)
    return (1.0 - torch.cos(np.pi * torch.clamp(alpha - j, min=0.0, max=1.0))) / 2.0


def hannw_embed(x: torch.Tensor, band_w: torch.Tensor) -> torch.Tensor:
    """[w0 sin(2^0 x), w0 cos(2^0 x), w1 sin(2 x), ...] (no identity) -> [P, 6*multires]."""
    out = []
    for j in range(band_w.numel()):
        f = 2.0 ** j
        out += [band_w[j] * torch.sin(x * f), band_w[j] * torch.cos(x * f)]
    return torch.cat(out, -1)


def fourier_embed(x: torch.Tensor, multires: int = 10) -> torch.Tensor:
    """embedders/fourier.py:18-40: [x, sin(2^0 x), cos(2^0 x), ..., sin(2^9 x), cos(2^9 x)] -> 63."""
    out = [x]
    for j in range(multires):
        f = 2.0 ** j
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


def nonrigid_mlp(sd, prefix, pos_embed, pos_xyz, cond):
    """non_rigid_motion_mlps/mlp_offset.py:54-70 (depth 6, width 128, skip before Linear #4)."""
    h = torch.cat([cond.expand(pos_embed.shape[0], -1), pos_embed], -1)
    idxs = [0, 2, 4, 6, 8, 10, 12]
    for i in idxs:
        if i == 8:
            h = torch.cat([h, pos_embed], -1)
        h = F.linear(h, sd[f"{prefix}block_mlps.{i}.weight"], sd[f"{prefix}block_mlps.{i}.bias"])
        if i != 12:
            h = torch.relu(h)
    return pos_xyz + h


def canonical_mlp(sd, pos_embed, prefix="cnl_mlp."):
    """canonical_mlps/mlp_rgb_sigma.py:49-58 (8x256; input re-concatenated FIRST before Linear #5)."""
    h = pos_embed
    for i in [0, 2, 4, 6, 8, 10, 12, 14]:
        if i == 10:
            h = torch.cat([pos_embed, h], -1)
        h = torch.relu(F.linear(h, sd[f"{prefix}pts_linears.{i}.weight"], sd[f"{prefix}pts_linears.{i}.bias"]))
    return F.linear(h, sd[prefix + "output_linear.0.weight"], sd[prefix + "output_linear.0.bias"])


# ----------------------------------------------------------------------------- P10
def raw2outputs(rgb, sigma, z_vals, rays_d, pts_mask=None, bgcolor=None, last_dist: float = 1e10):
    """M:73-99 (S3 form: activations already applied).  rgb [B,S,3], sigma [B,S], mask [B,S]."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], last_dist)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-sigma * dists)
    if pts_mask is not None:
        alpha = alpha * pts_mask
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    rgb_map = torch.sum(w[..., None] * rgb, -2)
    depth = torch.sum(w * z_vals, -1)
    acc = torch.sum(w, -1)
    if bgcolor is not None:
        rgb_map = rgb_map + (1.0 - acc[..., None]) * bgcolor[None, :] / 255.0
    return rgb_map, acc, w, depth


# ----------------------------------------------------------------------------- P1
def human_forward(sd: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], cfg: Optional[dict] = None,
                  transitions_times: Optional[Sequence[float]] = None, t_rand: Optional[torch.Tensor] = None,
                  stage: int = 3):
    """N:574-698 Network.forward (stage=3) / N2 (stage=2: composites with _raw2outputs + bgcolor).

    batch keys as the reference kwargs: rays [2,B,3], dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors,
    dst_posevec, near, far [B,1], iter_val, cnl_bbox_min_xyz, cnl_bbox_scale_xyz, bgcolor, time, is_train,
    (+ dst_Rs_prev, dst_Ts_prev, dst_posevec_prev when time > 0.005 and is_train).
    `t_rand` [B,N] uniform draws enable the stratified jitter (cfg.perturb > 0).
    """
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    K = c["total_bones"]
    iter_val = float(torch.as_tensor(batch["iter_val"]).reshape(-1)[0])
    time = float(batch["time"])
    is_train = bool(batch["is_train"])
    flow = time > 0.005 and is_train
    state = select_state(time, transitions_times)

    def refine(Rs, Ts, posevec):
        if iter_val >= c["pose_kick_in_iter"]:
            dR, dT = pose_refiner(sd, posevec[None], total_bones=K)
            Rs = torch.cat([Rs[0:1], torch.matmul(Rs[1:], dR[0])], 0)          # N:595-600
            Ts = torch.cat([Ts[0:1], Ts[1:] + dT[0]], 0)
        return Rs, Ts

    def cond_of(posevec):
        if iter_val < c["nonrigid_kick_in_iter"]:
            return torch.zeros_like(posevec) * posevec                            # N:653-656
        return posevec

    dst_Rs, dst_Ts = refine(batch["dst_Rs"], batch["dst_Ts"], batch["dst_posevec"])
    R_b, T_b, R_f, T_f = motion_basis(dst_Rs, dst_Ts, batch["cnl_gtfms"])
    if flow:
        Rp, Tp = refine(batch["dst_Rs_prev"], batch["dst_Ts_prev"], batch["dst_posevec_prev"])
        _, _, R_fp, T_fp = motion_basis(Rp, Tp, batch["cnl_gtfms"])
        cond_prev = cond_of(batch["dst_posevec_prev"])[None]
    band_w = hannw_weights(iter_val, c["nonrigid_kick_in_iter"], c["nonrigid_full_band_iter"], c["nonrigid_multires"])
    cond = cond_of(batch["dst_posevec"])[None]
    vol = motion_weight_volume(sd, batch["motion_weights_priors"])
    bmin, bscale = batch["cnl_bbox_min_xyz"], batch["cnl_bbox_scale_xyz"]

    rays_o, rays_d = batch["rays"][0].float(), batch["rays"][1].float()
    B = rays_o.shape[0]
    N = c["N_samples"]
    z = samples_along_ray(batch["near"], batch["far"], N, t_rand)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]
    flat = pts.reshape(-1, 3)
    x_skel, mask = backward_lbs(flat, R_b, T_b, vol, bmin, bscale)
    cnl = nonrigid_mlp(sd, "non_rigid_mlp.", hannw_embed(x_skel, band_w), x_skel, cond)
    emb = torch.cat([fourier_embed(cnl, c["canonical_multires"]), sd[f"human_stateembeds.{state}"].repeat(cnl.shape[0], 1)], -1)
    raw = canonical_mlp(sd, emb).view(B, N, 4)
    mask = mask.view(B, N)

    out = {}
    if flow:   # N:474-502
        d_prev = forward_lbs(cnl, R_fp, T_fp, vol, bmin, bscale)
        out["deform_pts_prev_final"] = nonrigid_mlp(sd, "non_rigid_forward_mlp.", hannw_embed(d_prev, band_w), d_prev, cond_prev).view(B, N, 3)
    sel = mask.reshape(-1) > 0.005                                              # N:505-536
    if int(sel.sum()) > 0:
        d_cur = forward_lbs(cnl[sel], R_f, T_f, vol, bmin, bscale)
        out["deform_pts_final"] = nonrigid_mlp(sd, "non_rigid_forward_mlp.", hannw_embed(d_cur, band_w), d_cur, cond)
        out["observe_pts"] = flat[sel]
    else:
        out["deform_pts_final"] = pts[0, 0][None]
        out["observe_pts"] = pts[0, 0][None]
    rgb, sigma = torch.sigmoid(raw[..., :3]), torch.relu(raw[..., 3])
    if stage == 3:
        out.update(human_rgb=rgb, human_density=sigma, newsmpl_pts=pts, pts_mask=mask, bgcolor=batch["bgcolor"])
        if not flow:
            out.update(z_vals=z, rays_d=rays_d)
    else:  # stage 2: N2:273-299 composites here (last interval 1e10, bg colour added)
        rgb_map, acc, w, depth = raw2outputs(rgb, sigma, z, rays_d, mask, batch["bgcolor"])
        out.update(rgb=rgb_map, alpha=acc, weights=w, depth=depth)
    out["cnl_pts"] = cnl.view(B, N, 3)
    out["motion_weights_vol"] = vol
    return out


# ----------------------------------------------------------------------------- C1..C3
def stage3_composite(bkg_tdist, bkg_rgb, bkg_density, human, rays_o_bkg, rays_d_bkg, newsmpl_to_scale_world):
    """M:1524-1596.  bkg_* come from ray_history[-1] ([B,33], [B,32,3], [B,32]); `human` is the dict of
    human_forward(stage=3).  Returns rgb [B,3], idx_fg [B] bool, total_order [B_fg,160] int64,
    human_weights_onlyfg [B_fg,128]."""
    pts = human["newsmpl_pts"]
    hom = torch.cat([pts, torch.ones_like(pts[..., :1])], -1)
    world = torch.einsum("ji,bni->bnj", newsmpl_to_scale_world, hom)[..., :3]          # M:1524
    d = rays_d_bkg[..., None, :]
    if torch.any(torch.abs(d) < 1e-5):                                                 # M:1526-1542
        ok = torch.abs(rays_d_bkg) > 1e-5
        first = torch.argmax(ok.int(), dim=-1)           # first non-tiny component
        assert bool(ok.any(-1).all()), "ray with all-tiny direction components"
        num = torch.gather(world - rays_o_bkg[:, None, :], -1, first[:, None, None].expand(-1, world.shape[1], 1))[..., 0]
        den = torch.gather(rays_d_bkg + 1e-10, -1, first[:, None])
        z_h = num / den
    else:
        z_h = torch.mean((world - rays_o_bkg[:, None, :]) / (d + 1e-10), dim=-1)       # M:1544-1545
    mask = human["pts_mask"]
    idx_fg = mask.sum(-1) > 5e-3                                                        # M:1547-1551
    B = mask.shape[0]
    z_b = bkg_tdist[..., :-1]
    rgb_out = torch.zeros(B, 3, device=z_b.device, dtype=bkg_rgb.dtype)
    bkg = torch.cat([bkg_rgb, bkg_density[..., None]], -1)
    hum = torch.cat([human["human_rgb"], human["human_density"][..., None]], -1)
    fg, bg = idx_fg, ~idx_fg
    total_order = torch.zeros(0, z_b.shape[1] + z_h.shape[1], dtype=torch.int64, device=z_b.device)
    hw = torch.zeros(0, z_h.shape[1], device=z_b.device, dtype=bkg_rgb.dtype)
    if int(fg.sum()) > 0:
        zz, total_order = torch.sort(torch.cat([z_b[fg], z_h[fg]], -1), dim=-1, stable=True)   # M:1565
        allv = torch.cat([bkg[fg], hum[fg]], 1)
        allv = torch.gather(allv, 1, total_order[..., None].expand(-1, -1, 4))
        m = torch.cat([torch.ones_like(z_b[fg]), mask[fg]], -1)
        m = torch.gather(m, 1, total_order)
        rgb_fg, _, w_fg, _ = raw2outputs(allv[..., :3], allv[..., 3], zz, rays_d_bkg[fg], m)
        is_h = total_order >= z_b.shape[1]
        hw = w_fg[is_h].reshape(-1, z_h.shape[1])                                       # M:1588
        rgb_out[fg] = rgb_fg
    if int(bg.sum()) > 0:
        rgb_bg, _, _, _ = raw2outputs(bkg[bg][..., :3], bkg[bg][..., 3], z_b[bg], rays_d_bkg[bg], torch.ones_like(z_b[bg]))
        rgb_out[bg] = rgb_bg
    return rgb_out, idx_fg, total_order, hw, z_h
