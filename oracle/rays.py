"""TEST INFRASTRUCTURE -- CPU restatement of the reference's per-frame ray set-up, numpy like the
reference: `core/utils/camera_util.py` of stage 3 (`C:` below).  Imported only by tests.

  rays_from_krt      C:154-183  get_rays_from_KRT       camera origin + per-pixel directions (|d_z| = 1 in camera space)
  rays_from_krt_bkg  C:185-216  get_rays_from_KRT_bkg   same + unit view directions + mip-NeRF radii (row differences)
  rays_aabb          C:219-265  rays_intersect_3d_bbox  six-plane AABB test, rays with exactly two hits, near/far
  patch_ray_indices  T:225-332  Dataset.get_patch_ray_indices  training item: random P x P patches -> ray indices
  patch_ray_indices_s2  T2:215-332  the stage-2 dataset's form (2nd_State_Conditional_Human-Object/core/data/human_nerf/train.py):
                                    the patch IS intersected with the box -> ragged selection, patch masks with holes
"""
import numpy as np


def rays_from_krt(H, W, K, R, T):
    rays_o = -np.dot(R.T, T).ravel()                                         # C:172
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")   # C:174-176
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)
    pixel_camera = np.dot(xy1, np.linalg.inv(K).T)                           # C:178
    pixel_world = np.dot(pixel_camera - T.ravel(), R)                        # C:179
    rays_d = pixel_world - rays_o[None, None]                                # C:181
    return np.broadcast_to(rays_o, rays_d.shape), rays_d


def rays_from_krt_bkg(H, W, K, R, T):
    rays_o, rays_d = rays_from_krt(H, W, K, R, T)
    viewdirs = rays_d / np.linalg.norm(rays_d, axis=-1, keepdims=True)       # C:210
    dx = np.sqrt(np.sum((rays_d[:-1, :, :] - rays_d[1:, :, :]) ** 2, -1))    # C:212
    dx = np.concatenate([dx, dx[-2:-1, :]], 0)                               # C:213 (last row repeats row H-2)
    radii = dx[..., None] * 2 / np.sqrt(12)                                  # C:214
    return rays_o, rays_d, viewdirs, radii


def rays_aabb(bounds, ray_o, ray_d):
    """bounds [2,3] (min, max); returns near, far for the valid rays and the [N] validity mask.
    NOTE: like the reference this clamps tiny direction components of ray_d IN PLACE (C:238)."""
    bounds = np.asarray(bounds) + np.array([-0.01, 0.01])[:, None]           # C:234
    nominator = bounds[None] - ray_o[:, None]
    ray_d[np.abs(ray_d) < 1e-5] = 1e-5                                       # C:238
    d_intersect = (nominator / ray_d[:, None]).reshape(-1, 6)
    p_intersect = d_intersect[..., None] * ray_d[:, None] + ray_o[:, None]
    min_x, min_y, min_z, max_x, max_y, max_z = bounds.ravel()
    eps = 1e-6
    p_mask = ((p_intersect[..., 0] >= (min_x - eps)) * (p_intersect[..., 0] <= (max_x + eps)) *
              (p_intersect[..., 1] >= (min_y - eps)) * (p_intersect[..., 1] <= (max_y + eps)) *
              (p_intersect[..., 2] >= (min_z - eps)) * (p_intersect[..., 2] <= (max_z + eps)))     # C:245-250
    mask_at_box = p_mask.sum(-1) == 2                                        # C:252
    p_intervals = p_intersect[mask_at_box][p_mask[mask_at_box]].reshape(-1, 2, 3)
    o = ray_o[mask_at_box]
    d = ray_d[mask_at_box]
    norm_ray = np.linalg.norm(d, axis=1)
    d0 = np.linalg.norm(p_intervals[:, 0] - o, axis=1) / norm_ray
    d1 = np.linalg.norm(p_intervals[:, 1] - o, axis=1) / norm_ray
    return np.minimum(d0, d1), np.maximum(d0, d1), mask_at_box


def patch_ray_indices(N_patch, ray_mask, subject_mask, bbox_mask, patch_size, H, W, sample_subject_ratio, rng=np.random):
    """T:225-332 (`Dataset.get_patch_ray_indices` + `_get_patch_ray_indices`, T = core/data/human_nerf/train.py of stage 3).
    Returns select_inds [N*P*P] (indices into the box-compacted ray arrays, possibly -1), xy_min [N,2], patch masks
    [N,P,P] (all True: the patch is not intersected with the box, T:323) and patch_div_indices [N+1]."""
    excl = np.bitwise_and(bbox_mask, np.bitwise_not(subject_mask))                   # T:238-241
    sels, xy, masks, div = [], [], [], [0]
    for _ in range(N_patch):
        cand = subject_mask if rng.rand(1)[0] < sample_subject_ratio else excl       # T:256-259
        ys, xs = np.where(cand)                                                      # T:294
        k = rng.choice(ys.shape[0], size=[1], replace=False)[0]                      # T:297-298
        half = patch_size // 2
        x0 = np.clip(xs[k] - half, 0, W - patch_size)                                # T:304-311
        y0 = np.clip(ys[k] - half, 0, H - patch_size)
        m = np.zeros((H, W), dtype=bool)
        m[y0:y0 + patch_size, x0:x0 + patch_size] = True                             # T:313-314
        sels.append((np.cumsum(ray_mask) - 1)[np.where(m.reshape(-1))])              # T:322-328
        xy.append(np.array([x0, y0]))
        masks.append(m[y0:y0 + patch_size, x0:x0 + patch_size])
        div.append(div[-1] + sels[-1].shape[0])
    return np.concatenate(sels, 0), np.stack(xy, 0), np.stack(masks, 0), np.array(div)


def patch_ray_indices_s2(N_patch, ray_mask, subject_mask, bbox_mask, patch_size, H, W, sample_subject_ratio, rng=np.random):
    """T2:215-332 (`Dataset.get_patch_ray_indices` + `_get_patch_ray_indices` of STAGE 2).  Same random decisions as stage 3, but
    the patch rectangle is intersected with the rays that hit the subject's box (T2:321-322): select_inds holds only those rays
    (ragged, `patch_div_indices` delimits the patches) and the patch mask [P,P] marks the pixels that kept their ray (T2:329-332)."""
    excl = np.bitwise_and(bbox_mask, np.bitwise_not(subject_mask))                   # T2:227-230
    sels, xy, masks, div = [], [], [], [0]
    masked_indices = np.cumsum(ray_mask) - 1                                         # T2:324
    for _ in range(N_patch):
        cand = subject_mask if rng.rand(1)[0] < sample_subject_ratio else excl       # T2:243-246
        ys, xs = np.where(cand)                                                      # T2:286
        k = rng.choice(ys.shape[0], size=[1], replace=False)[0]                      # T2:289-290
        half = patch_size // 2
        x0 = np.clip(xs[k] - half, 0, W - patch_size)                                # T2:296-303
        y0 = np.clip(ys[k] - half, 0, H - patch_size)
        m = np.zeros((H, W), dtype=bool)
        m[y0:y0 + patch_size, x0:x0 + patch_size] = True                             # T2:305-306
        inter = np.bitwise_and(m.reshape(-1), ray_mask)                              # T2:321
        sels.append(masked_indices[np.where(inter)])                                 # T2:322-325
        xy.append(np.array([x0, y0]))
        masks.append(inter.reshape(H, W)[y0:y0 + patch_size, x0:x0 + patch_size])    # T2:327-330
        div.append(div[-1] + sels[-1].shape[0])
    return np.concatenate(sels, 0), np.stack(xy, 0), np.stack(masks, 0), np.array(div)
