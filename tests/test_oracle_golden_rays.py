"""oracle/rays.py against vectors produced by the reference's own camera_util functions (tests/golden/make_golden_rays.py)."""
import os

import numpy as np

import oracle.rays as orays

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rays.npz"))


def test_rays_from_krt():
    H, W = int(G["H"]), int(G["W"])
    o, d = orays.rays_from_krt(H, W, G["K"], G["R"], G["T"])
    assert np.abs(o - G["rays_o"]).max() < 1e-6 and np.abs(d - G["rays_d"]).max() < 1e-6
    o2, d2, vd, rad = orays.rays_from_krt_bkg(H, W, G["K"], G["R"], G["T"])
    assert np.abs(vd - G["viewdirs"]).max() < 1e-6 and np.abs(rad - G["radii"]).max() < 1e-8
    assert np.abs(np.linalg.norm(vd, axis=-1) - 1).max() < 1e-6


def test_rays_aabb():
    o = np.ascontiguousarray(G["rays_o"].reshape(-1, 3))
    d = np.ascontiguousarray(G["rays_d"].reshape(-1, 3))
    near, far, mask = orays.rays_aabb(G["bounds"], o.copy(), d.copy())
    assert np.array_equal(mask, G["mask"])
    assert np.abs(near - G["near"]).max() < 1e-6 and np.abs(far - G["far"]).max() < 1e-6
    assert (far >= near).all() and 0 < mask.sum() < mask.size
