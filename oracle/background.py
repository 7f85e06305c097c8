"""Oracle (CPU, torch fp32) for the state-conditional mip-NeRF-360 background branch.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Each function cites the reference
lines it restates.  Abbreviations:
  H: = 3rd_Complete_HOSNeRF/src/model/mipnerf360/helper.py
  M: = 3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py
  M1: = 1st_State-Conditional_Scene/src/model/mipnerf360/model.py
Row ids (B1..B12) are the rows of SURVEY.md section 8(a).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1.1920929e-07  # H:18 -- fp32 machine epsilon, used as a clip floor everywhere
HALF_PI32 = float(np.float32(0.5 * np.pi))


# ----------------------------------------------------------------------------- B2
def s_to_t(s: torch.Tensor, near: float, far: float) -> torch.Tensor:
    """H:169-174 construct_ray_warps()[1]:  t = 1 / (s/far + (1-s)/near)."""
    s_near, s_far = 1.0 / near, 1.0 / far
    return 1.0 / (s * s_far + (1 - s) * s_near)


# ----------------------------------------------------------------------------- B3
def max_dilate_weights(t: torch.Tensor, w: torch.Tensor, dilation: float,
                       domain: Tuple[float, float]) -> Tuple[torch.Tensor, torch.Tensor]:
    """H:187-194 (max_dilate_weights, renormalize=True) -> H:177, H:153-166, H:182.

    t [B,n+1] histogram edges, w [B,n] bin weights -> t_d [B,3n+1], w_d [B,3n].
    """
    n = w.shape[-1]
    pdf = w / torch.clip(t[..., 1:] - t[..., :-1], min=EPS)          # H:177-178
    lo = t[..., :-1] - dilation                                       # H:154
    hi = t[..., 1:] + dilation                                        # H:155
    edges = torch.sort(torch.cat([t, lo, hi], dim=-1), dim=-1).values  # H:156
    edges = torch.clip(edges, domain[0], domain[1])                   # H:157
    # For every new edge e_i: the largest pdf among dilated bins [lo_j, hi_j) containing it.
    e = edges[..., :, None]
    covered = (lo[..., None, :] <= e) & (hi[..., None, :] > e)        # H:158-160
    pdf_d = torch.where(covered, pdf[..., None, :], torch.zeros((), device=pdf.device)).amax(dim=-1)[..., :-1]
    w_d = pdf_d * (edges[..., 1:] - edges[..., :-1])                  # H:182-183
    w_d = w_d / torch.clip(w_d.sum(dim=-1, keepdim=True), min=EPS)    # H:191-192
    assert w_d.shape[-1] == 3 * n
    return edges, w_d


def resample_logits(sdist: torch.Tensor, weights: torch.Tensor, anneal: float,
                    resample_padding: float = 0.0) -> torch.Tensor:
    """M:478-482: anneal*log(w + padding), -inf for zero-width bins."""
    return torch.where(sdist[..., 1:] > sdist[..., :-1],
                       anneal * torch.log(weights + resample_padding),
                       torch.full_like(weights, -math.inf))


# ----------------------------------------------------------------------------- B4
def sample_positions_u(num_samples: int, randomized: bool, batch_shape: Sequence[int],
                       jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """H:343-367 `sample()` up to `u` (single_jitter=True, deterministic_center=True).

    `jitter` is the [B,1] uniform(0,1) draw; if None and randomized, it is drawn from the
    global CPU generator exactly like the reference does (H:364).
    """
    if not randomized:
        pad = 1 / (2 * num_samples)
        u = torch.linspace(pad, 1 - pad - EPS, num_samples)
        return torch.broadcast_to(u, tuple(batch_shape) + (num_samples,))
    u_max = EPS + (1 - EPS) / num_samples
    max_jitter = (1 - u_max) / (num_samples - 1) - EPS
    if jitter is None:
        jitter = torch.rand(tuple(batch_shape) + (1,))
    return torch.linspace(0, 1 - u_max, num_samples) + jitter * max_jitter


def cdf_from_logits(w_logits: torch.Tensor) -> torch.Tensor:
    """H:227-229 + H:197-204: softmax -> [0, clip(cumsum(w[:-1]),max=1), 1]."""
    w = F.softmax(w_logits, dim=-1)
    cw = torch.cumsum(w[..., :-1], dim=-1).clip(max=1.0)
    z = torch.zeros(cw.shape[:-1] + (1,), dtype=cw.dtype, device=cw.device)
    return torch.cat([z, cw, torch.ones_like(z)], dim=-1)


def sorted_interp_indexed(x: torch.Tensor, xp: torch.Tensor, fp: torch.Tensor):
    """H:208-224 restated with an explicit bin index.

    The reference takes max/min over the mask  x >= xp_j.  With xp non-decreasing the mask is a
    prefix of length c = #{j : xp_j <= x}  (= searchsorted(xp, x, right=True)), hence
        (xp0,fp0) = entry  max(c-1, 0)       (falls back to entry 0 when c == 0)
        (xp1,fp1) = entry  min(c, n-1)       (falls back to the last entry when c == n)
    Returns (value, idx_lo) where idx_lo is the bit-exact "sample index" of SURVEY 8(a) B4.
    """
    n = xp.shape[-1]
    c = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)
    i_lo = (c - 1).clamp(min=0)
    i_hi = c.clamp(max=n - 1)
    xp0 = torch.gather(xp, -1, i_lo)
    xp1 = torch.gather(xp, -1, i_hi)
    fp0 = torch.gather(fp, -1, i_lo)
    fp1 = torch.gather(fp, -1, i_hi)
    offset = torch.clip(torch.nan_to_num((x - xp0) / (xp1 - xp0), 0), 0, 1)
    return fp0 + offset * (fp1 - fp0), i_lo


def sample_intervals(randomized: bool, t: torch.Tensor, w_logits: torch.Tensor, num_samples: int,
                     domain: Tuple[float, float], jitter: Optional[torch.Tensor] = None,
                     return_index: bool = False):
    """H:373-399 sample_intervals (single_jitter=True) -> S+1 interval edges."""
    # (the reference draws u on the CPU and moves it with .type_as, H:363-365; .to(t) does the same)
    u = sample_positions_u(num_samples, randomized, t.shape[:-1], jitter).to(t)
    u = torch.broadcast_to(u, t.shape[:-1] + (num_samples,))
    centers, idx = sorted_interp_indexed(u, cdf_from_logits(w_logits), t)   # H:227-231
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    first = torch.clip(2 * centers[..., :1] - mid[..., :1], min=domain[0])
    last = torch.clip(2 * centers[..., -1:] - mid[..., -1:], max=domain[1])
    out = torch.cat([first, mid, last], dim=-1)
    return (out, idx) if return_index else out


# ----------------------------------------------------------------------------- B5
def cast_rays_cone(tdist: torch.Tensor, origins: torch.Tensor, directions: torch.Tensor,
                   radii: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """H:279-290, H:294-304, H:318-339 with ray_shape="cone", diag=False.

    tdist [B,S+1], origins/directions [B,3], radii [B,1] -> means [B,S,3], covs [B,S,3,3].
    """
    t0, t1 = tdist[..., :-1], tdist[..., 1:]
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    denom = (3 * mu**2 + hw**2).clip(min=EPS)
    t_mean = mu + (2 * mu * hw**2) / denom
    t_var = (hw**2) / 3 - (4 / 15) * hw**4 * (12 * mu**2 - hw**2) / denom**2
    r_var = ((mu**2) / 4 + (5 / 12) * hw**2 - (4 / 15) * (hw**4) / denom) * radii**2
    d = directions
    mean = d[..., None, :] * t_mean[..., None]
    d_mag_sq = torch.sum(d**2, dim=-1, keepdim=True).clip(min=1e-10)
    d_outer = d[..., :, None] * d[..., None, :]
    null_outer = torch.eye(3, dtype=d.dtype, device=d.device) - d[..., :, None] * (d / d_mag_sq)[..., None, :]
    cov = t_var[..., None, None] * d_outer[..., None, :, :] + r_var[..., None, None] * null_outer[..., None, :, :]
    return mean + origins[..., None, :], cov


# ----------------------------------------------------------------------------- B6
def contract(mean: torch.Tensor, cov: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """H:33-68.  mip-NeRF-360 scene contraction of a Gaussian.

    z = x if |x|^2<=1 else ((2|x|-1)/|x|^2) x ; cov' = J cov J^T.  The reference obtains J by
    functorch.vmap(jacrev); the closed form is  J = s I + ((2/r^3 - 2/r^2)/r) x x^T,  s=(2r-1)/r^2
    for r>1 and J = I otherwise (agrees to 1.5e-8).  Outputs carry no gradient
    (H:64-67 `.detach()`).
    """
    x = mean.detach()
    r2 = torch.sum(x * x, dim=-1, keepdim=True).clip(min=1e-32)
    r = torch.sqrt(r2)
    inside = r2 <= 1
    s = (2 * r - 1) / r2
    z = torch.where(inside, x, s * x)
    c = (2 / (r2 * r) - 2 / r2) / r
    eye = torch.eye(3, dtype=x.dtype, device=x.device)
    J = s[..., None] * eye + c[..., None] * (x[..., :, None] * x[..., None, :])
    J = torch.where(inside[..., None], eye.expand_as(J), J)
    cov2 = J @ cov.detach() @ J.transpose(-1, -2)
    return z, cov2


# ----------------------------------------------------------------------------- B7 / B8
def _icosahedron():
    a = (math.sqrt(5) + 1) / 2
    v = np.array([(-1, 0, a), (1, 0, a), (-1, 0, -a), (1, 0, -a), (0, a, 1), (0, a, -1),
                  (0, -a, 1), (0, -a, -1), (a, 1, 0), (-a, 1, 0), (a, -1, 0), (-a, -1, 0)]) / math.sqrt(a + 2)
    f = np.array([(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10),
                  (5, 3, 8), (5, 2, 3), (2, 7, 3), (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6),
                  (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2), (9, 2, 5), (7, 2, 11)])
    return v, f


def generate_basis(subdivision: int = 2, tol: float = 1e-4) -> torch.Tensor:
    """H:457-531 generate_basis("icosahedron", 2) -> pos_basis_t [3,21] (fp32).

    Tessellate each icosahedron face with barycentric weights (i,j,v-i-j)/v, project to the unit
    sphere, merge duplicates (first occurrence wins, ascending index), drop antipodal copies
    (keep the one that appears first), then reverse the xyz column order.
    """
    verts, faces = _icosahedron()
    v = subdivision
    bary = np.array([(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)]) / v
    pts = []
    for face in faces:
        p = bary @ verts[face, :]
        pts.append(p / np.sqrt(np.sum(p**2, 1, keepdims=True)))
    pts = np.concatenate(pts, 0)

    def sqdist(a, b):
        return np.maximum(0, np.sum(a**2, 0)[:, None] + np.sum(b**2, 0)[None, :] - 2 * a.T @ b)

    d = sqdist(pts.T, pts.T)
    first = np.array([np.min(np.argwhere(row <= tol)) for row in d])
    pts = pts[np.unique(first), :]
    match = sqdist(pts.T, -pts.T) < tol
    pts = pts[np.any(np.triu(match), 1), :]
    return torch.from_numpy(pts[:, ::-1].copy().T).to(torch.float32)


def lift_and_diagonalize(means, covs, basis):
    """H:71-74."""
    return means @ basis, torch.sum(basis[None, None, ...] * (covs @ basis), dim=-2)


def integrated_pos_enc(mean: torch.Tensor, var: torch.Tensor, min_deg: int, max_deg: int) -> torch.Tensor:
    """H:78-89 + H:104-105.  level-major / direction-minor; [sin-part | sin(.+pi/2)-part]."""
    scales = 2.0 ** torch.arange(min_deg, max_deg, dtype=mean.dtype, device=mean.device)
    shape = tuple(mean.shape[:-1]) + (-1,)
    sm = (mean[..., None, :] * scales[:, None]).reshape(shape)
    sv = (var[..., None, :] * scales[:, None] ** 2).reshape(shape)
    damp = torch.exp(-0.5 * sv)
    return torch.cat([damp * torch.sin(sm), damp * torch.sin(sm + HALF_PI32)], dim=-1)


def pos_enc(x: torch.Tensor, min_deg: int, max_deg: int, append_identity: bool = True) -> torch.Tensor:
    """H:93-100 (view directions: min_deg=0, max_deg=4 -> 27 features)."""
    scales = 2.0 ** torch.arange(min_deg, max_deg, dtype=x.dtype, device=x.device)
    xb = (x[..., None, :] * scales[:, None]).reshape(tuple(x.shape[:-1]) + (-1,))
    feat = torch.sin(torch.cat([xb, xb + HALF_PI32], dim=-1))
    return torch.cat([x, feat], dim=-1) if append_identity else feat


# ----------------------------------------------------------------------------- B9
def select_state(time: float, transitions_times: Optional[Sequence[float]]) -> int:
    """M:224-293 (same ladder in N:179-246): index of the state embedding for `time`.

    K transition times -> K+1 embeddings.  state 0 iff time < tau_0 - 1e-5; otherwise the first
    k in 1..K-1 with time <= tau_k + 1e-5; otherwise K.
    """
    if transitions_times is None or len(transitions_times) == 0:
        return 0
    tau = [float(np.float32(t)) for t in transitions_times]
    time = float(time)
    eps = 1e-5
    if time < np.float32(tau[0]) - eps:
        return 0
    for k in range(1, len(tau)):
        if time <= np.float32(tau[k]) + eps:
            return k
    return len(tau)


class MLPWeights:
    """Plain container of one MipNeRF360MLP's tensors under the reference's state_dict names."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        g = lambda k: sd[prefix + k]
        self.pts = []
        i = 0
        while (prefix + f"pts_linear.{i}.weight") in sd:
            self.pts.append((g(f"pts_linear.{i}.weight"), g(f"pts_linear.{i}.bias")))
            i += 1
        self.density = (g("density_layer.weight"), g("density_layer.bias"))
        self.has_rgb = (prefix + "rgb_layer.weight") in sd
        if self.has_rgb:
            self.bottleneck = (g("bottleneck_layer.weight"), g("bottleneck_layer.bias"))
            self.views = (g("views_linear.0.weight"), g("views_linear.0.bias"))
            self.rgb = (g("rgb_layer.weight"), g("rgb_layer.bias"))
        self.embeds = []
        i = 0
        while (prefix + f"bkgd_stateembeds.{i}") in sd:
            self.embeds.append(g(f"bkgd_stateembeds.{i}"))
            i += 1
        self.basis = g("pos_basis_t") if (prefix + "pos_basis_t") in sd else generate_basis()


def encode_samples(means, covs, basis, max_deg_point: int = 12):
    """M:213-222: contract -> lift -> IPE  ([B,S,504])."""
    m, c = contract(means, covs)
    lm, lv = lift_and_diagonalize(m, c, basis.to(device=m.device, dtype=m.dtype))
    return integrated_pos_enc(lm, lv, 0, max_deg_point)


def mlp_forward(wts: MLPWeights, means, covs, viewdirs, state: int, skip_layer: int = 4,
                density_bias: float = -1.0, rgb_padding: float = 0.001, deg_view: int = 4):
    """M:213-351 MipNeRF360MLP.forward (noise terms are 0 in every config)."""
    x = encode_samples(means, covs, wts.basis)
    B, S, _ = x.shape
    x = torch.cat([x, wts.embeds[state].repeat(B, S, 1)], dim=-1)            # M:295-296
    inputs = x
    for idx, (W, b) in enumerate(wts.pts):                                    # M:299-303
        x = torch.relu(F.linear(x, W, b))
        if idx % skip_layer == 0 and idx > 0:
            x = torch.cat([x, inputs], dim=-1)
    raw_density = F.linear(x, *wts.density)[..., 0]
    density = F.softplus(raw_density + density_bias)                          # M:316
    if not wts.has_rgb:
        return {"density": density, "rgb": torch.zeros_like(means)}
    bott = F.linear(x, *wts.bottleneck)
    dir_enc = pos_enc(viewdirs, 0, deg_view, True)
    dir_enc = torch.broadcast_to(dir_enc[..., None, :], bott.shape[:-1] + (dir_enc.shape[-1],))
    h = torch.relu(F.linear(torch.cat([bott, dir_enc], dim=-1), *wts.views))  # M:337-342
    rgb = torch.sigmoid(F.linear(h, *wts.rgb))
    rgb = rgb * (1 + 2 * rgb_padding) - rgb_padding                           # M:345-346
    return {"density": density, "rgb": rgb}


# ----------------------------------------------------------------------------- B10 / B11
def compute_alpha_weights(density, tdist, dirs, opaque_background: bool):
    """H:235-261."""
    delta = (tdist[..., 1:] - tdist[..., :-1]) * torch.norm(dirs[..., None, :], dim=-1)
    dd = density * delta
    if opaque_background:
        dd = torch.cat([dd[..., :-1], torch.full_like(dd[..., -1:], 1e10)], dim=-1)
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    return alpha * trans, alpha, trans


def volumetric_rendering(rgbs, weights, bg_rgb: float):
    """H:265-275."""
    acc = weights.sum(dim=-1)
    bg_w = torch.clip(1 - acc[..., None], min=0)
    return (weights[..., None] * rgbs).sum(dim=-2) + bg_w * bg_rgb


# ----------------------------------------------------------------------------- B1
def mipnerf360_forward(state_dict: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor],
                       train_frac: float, randomized: bool, near: float, far: float,
                       transitions_times: Optional[Sequence[float]] = None,
                       num_prop_samples: int = 64, num_nerf_samples: int = 32, num_levels: int = 3,
                       anneal_slope: float = 10, dilation_multiplier: float = 0.5,
                       dilation_bias: float = 0.0025, opaque_background: bool = True,
                       bg_intensity: float = 1.0, jitters: Optional[List[torch.Tensor]] = None,
                       render: bool = True, prefix: str = ""):
    """M:418-540 (S3) / M1:331-461 (S1) MipNeRF360.forward.

    `render=True` reproduces stage 1 (per-level `renderings`), `render=False` stage 3 (empty list).
    `batch["times"]` may be 0-d or [B] (only the first entry is used, M1:335).
    Returns (renderings, ray_history); ray_history[l] has density, rgb, sdist, tdist, weights
    and additionally `bin_idx` (the bit-exact inverse-CDF bin of every sample centre).
    """
    o, d, vd, radii = batch["rays_o"], batch["rays_d"], batch["viewdirs"], batch["radii"]
    B = o.shape[0]
    time = float(torch.as_tensor(batch["times"]).reshape(-1)[0])
    state = select_state(time, transitions_times)
    sdist = torch.cat([torch.zeros(B, 1, device=o.device, dtype=o.dtype), torch.ones(B, 1, device=o.device, dtype=o.dtype)], dim=-1)
    weights = torch.ones(B, 1, device=o.device, dtype=o.dtype)
    prod = 1
    renderings, history = [], []
    for lvl in range(num_levels):
        S = num_prop_samples if lvl < num_levels - 1 else num_nerf_samples
        dilation = dilation_bias + dilation_multiplier * 1.0 / prod            # M:450-455
        prod *= S
        if lvl > 0:
            sdist, weights = max_dilate_weights(sdist, weights, dilation, (0.0, 1.0))
            sdist, weights = sdist[..., 1:-1], weights[..., 1:-1]              # M:469-470
        anneal = (anneal_slope * train_frac) / ((anneal_slope - 1) * train_frac + 1) if anneal_slope > 0 else 1.0
        logits = resample_logits(sdist, weights, anneal)
        jit = None if jitters is None else jitters[lvl]
        sdist, bin_idx = sample_intervals(randomized, sdist, logits, S, (0.0, 1.0), jit, return_index=True)
        sdist = sdist.detach()                                                 # M:493-494
        tdist = s_to_t(sdist, near, far)
        means, covs = cast_rays_cone(tdist, o, d, radii)
        wts = MLPWeights(state_dict, f"{prefix}mlps.{lvl}.")
        res = mlp_forward(wts, means, covs, vd, state)
        weights = compute_alpha_weights(res["density"], tdist, d, opaque_background)[0]
        res.update(sdist=sdist, tdist=tdist, weights=weights, bin_idx=bin_idx)
        history.append(res)
        if render:
            renderings.append({"rgb": volumetric_rendering(res["rgb"], weights, bg_intensity)})
    return renderings, history


# ----------------------------------------------------------------------------- B12
def searchsorted_lo_hi(a: torch.Tensor, v: torch.Tensor):
    """H:109-114 (int64 indices; the reference builds them with masked max/min over arange).

    idx_lo = last i with a_i <= v (0 if none); idx_hi = first i with a_i > v (n-1 if none).
    """
    n = a.shape[-1]
    c = torch.searchsorted(a.contiguous(), v.contiguous(), right=True)
    return (c - 1).clamp(min=0), c.clamp(max=n - 1)


def lossfun_outer(t, w, t_env, w_env):
    """H:136-138 via H:117-132: proposal (envelope) histogram must upper-bound the NeRF one."""
    cy = torch.cat([torch.zeros_like(w_env[..., :1]), torch.cumsum(w_env, dim=-1)], dim=-1)
    lo, hi = searchsorted_lo_hi(t_env, t)
    w_outer = torch.gather(cy, -1, hi)[..., 1:] - torch.gather(cy, -1, lo)[..., :-1]
    return torch.clip(w - w_outer, min=0) ** 2 / (w + EPS)


def lossfun_distortion(t, w):
    """H:142-149."""
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    intra = torch.sum(w**2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return inter + intra


def stage1_loss(rgb, target, history, data_mult=1.0, inter_mult=1.0, dist_mult=0.01, charb=0.001):
    """M1:491-514 + M1:611-627: Charbonnier(MSE) + interlevel + distortion."""
    mse = torch.mean((rgb - target) ** 2)
    loss = torch.sqrt(mse + charb**2) * data_mult
    c = history[-1]["sdist"].detach()
    w = history[-1]["weights"].detach()
    inter = 0.0
    for h in history[:-1]:
        inter = inter + torch.mean(lossfun_outer(c, w, h["sdist"], h["weights"]))
    dist = torch.mean(lossfun_distortion(history[-1]["sdist"], history[-1]["weights"]))
    return loss + inter * inter_mult + dist * dist_mult, {"mse": mse, "interlevel": inter, "distortion": dist}
