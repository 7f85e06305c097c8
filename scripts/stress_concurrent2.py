"""Which part of the human forward perturbs a concurrently running background forward (training mode, 3 levels)?
  python scripts/stress_concurrent2.py [iters]"""
import gc, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.hosnerf import HOSNeRF
from hosnerf_amd.human_nerf import default_cfg
from hosnerf_amd.train import batch_to_device, prepare_patch_targets

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
cfg = default_cfg(d)
cfg.perturb = 1.0
hos = HOSNeRF(cfg)
hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
hos = hos.to(dev)
B = 1024
item = synth.add_patch_supervision(synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5), 1, 32, 778)
gb = batch_to_device(prepare_patch_targets(item), dev)
g = torch.Generator().manual_seed(5)
TR = torch.rand(B, 128, generator=g).to(dev)
JIT = [torch.rand(B, generator=g).to(dev) for _ in range(3)]
bb = {"rays_o": gb["rays_o_bkg"], "rays_d": gb["rays_d_bkg"], "viewdirs": gb["viewdirs_bkg"], "radii": gb["radii"], "times": gb["time"]}
def masked_stream(words):
    """A HIP stream restricted to a set of CUs (hipExtStreamCreateWithCUMask), wrapped for torch."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


CUMASK = os.environ.get("CUMASK")
if CUMASK == "split":        # disjoint halves of the chip
    side = masked_stream([0xFFFFFFFF] * 4 + [0] * 4)
    main = masked_stream([0] * 4 + [0xFFFFFFFF] * 4)
elif CUMASK == "same":       # both restricted, to the SAME half (control)
    side = masked_stream([0xFFFFFFFF] * 4 + [0] * 4)
    main = masked_stream([0xFFFFFFFF] * 4 + [0] * 4)
else:
    side = torch.cuda.Stream()
    main = torch.cuda.Stream()
torch.cuda.set_stream(main)
big = torch.randn(64 * 1024 * 1024, device=dev); big2 = torch.empty_like(big)
net = hos.human


class _Stop(Exception):
    pass


def stop_at(name):
    """Truncate the human forward: the named stage raises instead of running (everything before it has been queued)."""
    import hosnerf_amd.human_nerf as hn
    def boom(*a, **k):
        raise _Stop()
    if name == "warp":
        ops.human_sample_warp_ad = boom
    elif name == "nonrigid":
        hn._NonRigidFn.apply = staticmethod(boom)
    elif name == "canonical":
        hn._CanonicalFn.apply = staticmethod(boom)
    elif name == "flow":
        ops.lbs_forward_ad = boom
    elif name == "compact":
        ops.compact_rows = boom


STOP = os.environ.get("STOP_AT")
if STOP:
    stop_at(STOP)


def bkgd():
    _, hist = hos.model(bb, 1.0, True, True, 0.1, 1e6, jitters=JIT)
    return {f"L{l}.{k}": h[k].detach() for l, h in enumerate(hist) for k in ("tdist", "density", "weights")}


def disturb(kind):
    with torch.cuda.stream(side):
        if kind == "copy":
            for _ in range(6):
                big2.copy_(big)
        elif kind == "human_train":
            net.split_decoder_backward = False
            try:
                return net(static_cycle=True, t_rand=TR, **gb)
            except _Stop:
                return None
        elif kind == "human_train_split":
            net.split_decoder_backward = True
            o = net(static_cycle=True, t_rand=TR, **gb)
            net._pending_vol = None
            return o
        elif kind == "human_nograd":
            with torch.no_grad():
                return net(static_cycle=True, t_rand=TR, **gb)
        elif kind == "prologue":
            return net.frame_prologue(**gb)
        elif kind == "prologue_nograd":
            with torch.no_grad():
                return net.frame_prologue(**gb)
        elif kind == "after_prologue":
            return net(static_cycle=True, t_rand=TR, prologue=PRO, **gb)


ref = {k: v.clone() for k, v in bkgd().items()}
torch.cuda.synchronize()
with torch.no_grad():
    PRO = net.frame_prologue(**gb)
torch.cuda.synchronize()
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "copy", "human_nograd", "human_train"]
for kind in KINDS:
    bad = {}
    for it in range(iters):
        side.wait_stream(torch.cuda.current_stream())
        keep = disturb(kind) if kind != "none" else None
        outs = bkgd()
        torch.cuda.synchronize()
        for k, v in outs.items():
            if not torch.equal(v, ref[k]):
                bad.setdefault(k, []).append(it)
        del keep, outs
        gc.collect()
    print(f"disturber {kind:18s}: " + ("all bit-identical" if not bad else "; ".join(f"{k}: {len(v)}/{iters}" for k, v in bad.items())), flush=True)
