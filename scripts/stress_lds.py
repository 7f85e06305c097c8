"""LDS canary run (scripts/probe/lds_canary.hip): canary workgroups hold a pattern in their LDS on one stream while the
background forward / the human training forward run on another.  python scripts/stress_lds.py"""
import ctypes, gc, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.hosnerf import HOSNeRF
from hosnerf_amd.human_nerf import default_cfg
from hosnerf_amd.train import batch_to_device, prepare_patch_targets, stage3_losses

_so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "liblds_canary.so")
if not os.path.exists(_so):
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", _so, _so.replace("liblds_canary.so", "lds_canary.hip")], check=True)
lib = ctypes.CDLL(_so)
lib.lds_canary_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
cfg = default_cfg(d); cfg.perturb = 1.0
hos = HOSNeRF(cfg); hos.two_streams = False
hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
hos = hos.to(dev)
B = 1024
item = synth.add_patch_supervision(synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5), 1, 32, 778)
gb = batch_to_device(prepare_patch_targets(item), dev)
bb = {"rays_o": gb["rays_o_bkg"], "rays_d": gb["rays_d_bkg"], "viewdirs": gb["viewdirs_bkg"], "radii": gb["radii"], "times": gb["time"]}
side = torch.cuda.Stream()


def work(kind):
    if kind == "bkgd_fwd":
        with torch.no_grad():
            hos.model(bb, 1.0, True, True, 0.1, 1e6)
    elif kind == "bkgd_train":
        _, hist = hos.model(bb, 1.0, True, True, 0.1, 1e6)
        (hist[-1]["weights"].sum() + hist[-1]["rgb"].sum()).backward()
    elif kind == "human_train":
        hos.human.split_decoder_backward = False
        out = hos.human(static_cycle=True, **gb)
        (out["human_rgbsigma"].sum() + out["deform_pts_final"].sum() + out["deform_pts_prev_final"].sum()).backward()
    elif kind == "chain_only":
        with torch.no_grad():
            x = torch.randn(131072, 3, device=dev) * 0.3
            pro = hos.human.frame_prologue(**gb)
            for _ in range(4):
                hos.human._nonrigid_fwd(hos.human._nrf, x, pro["cond"], pro["band_w"], save=True)
    elif kind == "step":
        hos.zero_grad()
        out = hos.render(gb, randomized=True, is_train=True, static_cycle=True)
        loss, _ = stage3_losses(out, gb)
        loss.backward()


THREADS = int(os.environ.get("CANARY_THREADS", "64"))
for kind in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("none", "bkgd_fwd", "bkgd_train", "human_train", "step")):
    for lds_kb in (4, 14, 24, 40):
        out = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        rc = lib.lds_canary_launch(4096, THREADS, lds_kb * 1024, 200, out.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
        if kind != "none":
            for _ in range(3):
                work(kind)
        torch.cuda.synchronize()
        o = out.cpu().tolist()
        print(f"{kind:12s} canary {lds_kb:2d} KB x 2048 WGs: corrupted words {o[0]}" + (f" (first: word {o[1]} = {o[2] & 0xffffffff:#x}, expected {o[4] & 0xffffffff:#x}, WG {o[5]})" if o[0] else "") + f", WGs run {o[3]}", flush=True)
        gc.collect()
