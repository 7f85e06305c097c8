"""Golden vectors for the per-frame ray set-up: outputs of the REFERENCE's own
core/utils/camera_util.py functions (imported here, in the build container only) on a seeded synthetic camera.
  python tests/golden/make_golden_rays.py   ->  tests/golden/rays.npz"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/3rd_Complete_HOSNeRF/core/utils/camera_util.py"


def main():
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))        # imported at module level, unused by these functions
    spec = importlib.util.spec_from_file_location("ref_camera_util", REF)
    cu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cu)
    rs = np.random.RandomState(777)
    H, W = 40, 56
    K = np.array([[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]], np.float32)
    ax = rs.randn(3); ax /= np.linalg.norm(ax); ang = 0.4
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = (np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx).astype(np.float32)
    T = np.array([0.1, -0.2, 3.0], np.float32)
    ro, rd = cu.get_rays_from_KRT(H, W, K, R, T)
    rob, rdb, vd, radii = cu.get_rays_from_KRT_bkg(H, W, K, R, T)
    bounds = np.array([[-0.7, -0.9, -0.5], [0.6, 0.8, 0.7]], np.float32)
    o2 = np.ascontiguousarray(ro.reshape(-1, 3)).astype(np.float32)
    d2 = np.ascontiguousarray(rd.reshape(-1, 3)).astype(np.float32)
    near, far, mask = cu.rays_intersect_3d_bbox(bounds, o2.copy(), d2.copy())
    np.savez_compressed(os.path.join(HERE, "rays.npz"), H=H, W=W, K=K, R=R, T=T, rays_o=np.asarray(ro, np.float32),
                        rays_d=np.asarray(rd, np.float32), viewdirs=np.asarray(vd, np.float32),
                        radii=np.asarray(radii, np.float32), bounds=bounds, near=near.astype(np.float32),
                        far=far.astype(np.float32), mask=mask)
    print("rays.npz:", ro.shape, int(mask.sum()), "of", mask.size, "rays hit the box")


if __name__ == "__main__":
    main()
