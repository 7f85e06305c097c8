"""Golden vectors for the free-viewpoint orbit (VERDICT r4 missing 3): outputs of the REFERENCE's own
core/utils/camera_util.py::rotate_camera_by_frame_idx / apply_global_tfm_to_camera (imported here, in the build container only)
for three cameras of a 100-frame turn, both source types, with and without a flipped camera.
cv2 is not installed in this image; the reference calls exactly one cv2 function on this path, `cv2.Rodrigues(vec)[0]` -- the
generator provides a numpy stand-in for it (the closed-form rotation matrix cv2 documents), nothing of the reference is altered.
  python tests/golden/make_golden_freeview.py   ->  tests/golden/freeview.npz"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/3rd_Complete_HOSNeRF/core/utils/camera_util.py"


def _rodrigues(v):
    v = np.asarray(v, dtype=np.float64).reshape(3)
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3), None
    k = v / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx, None


def main():
    cv2 = types.ModuleType("cv2")
    cv2.Rodrigues = _rodrigues
    sys.modules["cv2"] = cv2
    spec = importlib.util.spec_from_file_location("ref_camera_util", REF)
    cu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cu)
    rs = np.random.RandomState(2024)
    out = {}
    cases = []
    for flip in (False, True):
        ax = rs.randn(3); ax /= np.linalg.norm(ax)
        R = _rodrigues(ax * 0.3)[0]
        if flip:
            R = R @ np.diag([1.0, -1.0, -1.0])        # a camera whose up vector points down in SMPL space
        E = np.eye(4); E[:3, :3] = R; E[:3, 3] = np.array([0.1, -0.2, 3.1]) + 0.1 * rs.randn(3)
        trans = np.array([0.05, 0.9, 0.02]) + 0.05 * rs.randn(3)
        for inv_angle in (False, True):
            for k in (0, 7, 33, 61):
                tag = f"flip{int(flip)}_inv{int(inv_angle)}_k{k}"
                E_k, T_smpl = cu.rotate_camera_by_frame_idx(extrinsics=E.copy(), frame_idx=k, trans=trans.copy(), period=100,
                                                            inv_angle=inv_angle, rotate_axis="y")
                out[tag + "_E"], out[tag + "_T"] = np.asarray(E_k, np.float64), np.asarray(T_smpl, np.float64)
                cases.append(tag)
        out[f"flip{int(flip)}_E0"], out[f"flip{int(flip)}_trans"] = E, trans
    # without a translation (trans=None): rotation about the SMPL origin
    E = np.eye(4); E[:3, 3] = [0.0, 0.0, 2.5]
    E_k, T_smpl = cu.rotate_camera_by_frame_idx(extrinsics=E.copy(), frame_idx=25, trans=None, period=100)
    out["notrans_E0"], out["notrans_k25_E"], out["notrans_k25_T"] = E, np.asarray(E_k, np.float64), np.asarray(T_smpl, np.float64)
    # apply_global_tfm_to_camera on an orbit camera (freeview.py:232-235)
    Rh, Th = np.array([0.2, 0.5, -0.1]), np.array([0.05, 0.9, 0.02])
    E2, M = cu.apply_global_tfm_to_camera(E=out["flip0_inv0_k33_E"].copy(), Rh=Rh, Th=Th)
    out["gtfm_Rh"], out["gtfm_Th"], out["gtfm_E"], out["gtfm_M"] = Rh, Th, np.asarray(E2, np.float64), np.asarray(M, np.float64)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "freeview.npz"), **out)
    print("freeview.npz:", len(cases), "orbit cameras")


if __name__ == "__main__":
    main()
