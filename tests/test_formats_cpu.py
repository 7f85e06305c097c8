"""SURVEY 8(f).4 on-disk formats: hosnerf_amd.formats against the reference's own stage-1 loader run on a synthetic scene
directory (tests/golden/formats.npz, made by tests/golden/make_golden_formats.py): normalised extrinsics, intrinsics, splits,
render path, background-ray counts and the `cameras_scaleworld.pkl` the loader writes for stages 2 and 3."""
import json
import os
import pickle
import tempfile

import numpy as np

from hosnerf_amd import formats

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "formats.npz"))


def _scene_dir():
    d = tempfile.mkdtemp(prefix="hos_scene_")
    np.save(os.path.join(d, "poses_bounds.npy"), G["poses_bounds"])
    cams = {str(n): {"intrinsics": G["cam_intrinsics"][i], "smpl_to_camera": G["cam_smpl_to_camera"][i], "smpl_to_world": G["cam_smpl_to_world"][i]}
            for i, n in enumerate(G["names"])}
    with open(os.path.join(d, "cameras.pkl"), "wb") as f:
        pickle.dump(cams, f)
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"put_down": {"time": 0.31}, "pick_up": {"time": 0.77}}, f)
    return d


def test_scene_normalisation_matches_the_reference_loader():
    d = _scene_dir()
    s = formats.load_scene(d, (int(G["H"]), int(G["W"])), masks=G["masks"], cam_scale_factor=0.95, strict_scaling=False, near=0.1, far=1e6)
    assert np.abs(s["extrinsics"] - G["extrinsics"]).max() < 1e-6
    assert np.abs(s["intrinsics"] - G["intrinsics"]).max() < 1e-6
    assert np.array_equal(s["image_sizes"], G["image_sizes"])
    for got, key in zip(s["i_split"], ("i_train", "i_val", "i_test", "i_all")):
        assert np.array_equal(got, G[key]), key
    assert np.abs(s["render_poses"] - G["render_poses"]).max() < 1e-6
    assert np.array_equal(s["bkgrays_sizes"], G["bkgrays_sizes"])
    assert np.array_equal(s["times"], G["times"]) and np.array_equal(s["render_times"], G["render_times"])
    assert s["near"] == float(G["near"]) and s["far"] == float(G["far"])
    # transitions: one state more than transitions, file order kept
    assert np.allclose(s["transitions_times"], [0.31, 0.77])


def test_cameras_scaleworld_file_matches_the_reference_loader():
    d = _scene_dir()
    formats.load_scene(d, (int(G["H"]), int(G["W"])))
    with open(os.path.join(d, "cameras_scaleworld.pkl"), "rb") as f:
        csw = pickle.load(f)
    names = [str(n) for n in G["names"]]
    assert list(csw.keys()) == names
    for i, n in enumerate(names):
        assert set(csw[n]) == {"intrinsics", "smpl_to_camera", "smpl_to_scale_world", "scaleworld_to_camera"}
        assert csw[n]["smpl_to_scale_world"].dtype == np.float32
        assert np.abs(csw[n]["smpl_to_scale_world"] - G["csw_smpl_to_scale_world"][i]).max() < 1e-6
        assert np.abs(csw[n]["scaleworld_to_camera"] - G["csw_scaleworld_to_camera"][i]).max() < 1e-5
    # the similarity this file carries is what the stage-3 composite consumes (`newsmpl_to_scale_world`, C1): scale > 0, rigid part orthonormal
    A = csw[names[0]]["smpl_to_scale_world"][:3, :3].astype(np.float64)
    sc = np.cbrt(np.linalg.det(A))
    assert sc > 0 and np.abs((A / sc) @ (A / sc).T - np.eye(3)).max() < 1e-5


def test_skeleton_files_match_the_reference_body_util():
    """mesh_infos.pkl / canonical_joints.pkl -> network inputs, against the reference's own body_util functions."""
    d = tempfile.mkdtemp(prefix="hos_scene_")
    with open(os.path.join(d, "mesh_infos.pkl"), "wb") as f:
        pickle.dump({"frame_000000": {"poses": G["sk_poses72"], "tpose_joints": G["sk_tpose24"], "joints": G["sk_joints24"],
                                      "Rh": np.zeros(3, np.float32), "Th": np.zeros(3, np.float32)}}, f)
    with open(os.path.join(d, "canonical_joints.pkl"), "wb") as f:
        pickle.dump({"joints": G["sk_tpose24"]}, f)
    infos = formats.load_mesh_infos(os.path.join(d, "mesh_infos.pkl"))
    cj, cbox = formats.load_canonical_joints(os.path.join(d, "canonical_joints.pkl"))
    m = infos["frame_000000"]
    assert m["tpose_joints"].shape == (26, 3) and m["poses"].shape == (78,) and cj.shape == (26, 3)
    assert np.allclose(m["bbox"]["min_xyz"], G["sk_joints24"].min(0) - 0.6) and np.allclose(cbox["max_xyz"], cj.max(0) + 0.6)
    item = formats.skeleton_item(m, cj, cbox)
    assert np.abs(item["dst_Rs"] - G["sk_Rs"]).max() < 1e-6 and np.abs(item["dst_Ts"] - G["sk_Ts"]).max() < 1e-7
    assert np.abs(item["cnl_gtfms"] - G["sk_gtfms"]).max() < 1e-6
    vol = item["motion_weights_priors"]
    assert vol.shape == (27, 32, 32, 32) and vol.dtype == np.float32
    assert np.abs(vol[:, ::4, ::4, ::4] - G["sk_vol_sub"]).max() < 1e-6
    assert abs(vol.astype(np.float64).sum() - float(G["sk_vol_sum"])) < 1e-2 and np.abs(vol.max(axis=(1, 2, 3)) - G["sk_vol_max"]).max() < 1e-6
    assert item["dst_posevec"].shape == (75,) and np.allclose(item["dst_posevec"], m["poses"][3:] + 1e-2)
    assert np.allclose(item["cnl_bbox_scale_xyz"], 2.0 / (cbox["max_xyz"] - cbox["min_xyz"]))
