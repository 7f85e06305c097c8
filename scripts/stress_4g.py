"""Address-dependence hunt: every buffer the product code allocates with torch.empty is placed so that it STRADDLES a 4 GiB
address boundary (one buffer per boundary, inside an 80 GiB arena), and the results are compared bit for bit with the normal
placement.  A kernel that forms addresses with 32-bit arithmetic on the low half (no carry into the high half) gives different
results exactly then.   python scripts/stress_4g.py [bkgd|human|step]"""
import json, os, sys, tempfile, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

what = sys.argv[1] if len(sys.argv) > 1 else "bkgd"
dev = torch.device("cuda")
G4 = 1 << 32
arena = torch.empty(80 * (1 << 30), dtype=torch.uint8, device=dev)
base = arena.data_ptr()
bounds = [a for a in range((base // G4 + 1) * G4, base + arena.numel() - G4 // 2, G4)]
state = {"on": False, "next": 0, "placed": 0, "skipped": 0}
_empty = torch.empty


def _ours():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "hosnerf_amd/" in fr.filename and "_lib.py" not in fr.filename:
            return True
    return False


def sempty(*a, **k):
    t = _empty(*a, **k)
    if not (state["on"] and t.is_cuda and _ours()):
        return t
    nbytes = t.numel() * t.element_size()
    if nbytes < 4096 or nbytes > G4 // 2 or state["next"] >= len(bounds):
        state["skipped"] += 1
        return t
    bnd = bounds[state["next"]]
    state["next"] += 1
    start = (bnd - nbytes // 2) // 256 * 256
    off = start - base
    v = arena[off:off + nbytes].view(t.dtype).view(t.shape)
    assert v.data_ptr() < bnd < v.data_ptr() + nbytes
    state["placed"] += 1
    return v


torch.empty = sempty
from hosnerf_amd import ops, synth
from hosnerf_amd.mipnerf360 import MipNeRF360
from hosnerf_amd.hosnerf import HOSNeRF
from hosnerf_amd.human_nerf import default_cfg
from hosnerf_amd.train import batch_to_device, prepare_patch_targets, stage3_losses

ops.set_gemm_mode(ops.GEMM_PLANES)
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
cfg = default_cfg(d)
cfg.perturb = 1.0
hos = HOSNeRF(cfg)
hos.two_streams = False
hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
hos = hos.to(dev)
B = 1024
item = synth.add_patch_supervision(synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5), 1, 32, 778)
gb = batch_to_device(prepare_patch_targets(item), dev)
g = torch.Generator().manual_seed(5)
TR = torch.rand(B, 128, generator=g).to(dev)
JIT = [torch.rand(B, generator=g).to(dev) for _ in range(3)]


def step():
    hos.zero_grad()
    out = hos.render(gb, randomized=True, is_train=True, static_cycle=True, jitters=JIT, t_rand=TR)
    loss, _ = stage3_losses(out, gb)
    loss.backward()
    torch.cuda.synchronize()
    res = {"rgb": out["rgb"].detach().clone(), "loss": loss.detach().clone(), "g_bkgd": hos.model.flat_grad.clone(), "g_human": hos.human.flat_grad.clone()}
    for l, h in enumerate(out["ray_history"]):
        res[f"bkgd{l}.density"] = h["density"].detach().clone()
    for k in ("human_rgbsigma", "pts_mask", "newsmpl_pts"):
        res[k] = out[k].detach().clone()
    return res


ref = step()
ref2 = step()
print("normal placement, run-to-run:", {k: float((ref2[k] - ref[k]).abs().max()) for k in ref})
# the allocations of one step are many more than there are boundaries: slide a window over the allocation sequence
n_alloc = None
first = 0
while True:
    state.update(on=True, next=0, placed=0, skipped=0)
    cnt = {"seen": 0}
    orig_next = 0

    def gated(*a, **k):
        return sempty(*a, **k)
    # skip the first `first` eligible allocations by pre-consuming the counter
    state["skip_until"] = first
    seen = {"n": 0}
    _s = sempty

    def windowed(*a, **k):
        t = _empty(*a, **k)
        if not (t.is_cuda and _ours()):
            return t
        nbytes = t.numel() * t.element_size()
        if nbytes < 4096 or nbytes > G4 // 2:
            return t
        seen["n"] += 1
        if seen["n"] <= first or state["next"] >= len(bounds):
            return t
        bnd = bounds[state["next"]]
        state["next"] += 1
        start = (bnd - nbytes // 2) // 256 * 256
        v = _empty(0, dtype=t.dtype, device=t.device).set_(arena.untyped_storage(), (start - base) // t.element_size(), t.shape)   # not an autograd view
        assert v.data_ptr() == start
        state["placed"] += 1
        return v
    torch.empty = windowed
    got = step()
    torch.empty = _empty
    diffs = {k: float((got[k] - ref[k]).abs().max()) for k in ref}
    bad = {k: v for k, v in diffs.items() if not (v <= 1e-4 * max(float(ref[k].abs().max()), 1e-30))}
    print(f"allocations {first + 1}..{first + state['placed']} of {seen['n']} straddle a 4 GiB boundary: " + ("same results" if not bad else f"DIFFERENT {bad}"), flush=True)
    if state["placed"] == 0 or first + state["placed"] >= seen["n"]:
        break
    first += state["placed"]
