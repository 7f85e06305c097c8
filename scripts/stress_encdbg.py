import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.mipnerf360 import MipNeRF360
from hosnerf_amd.human_nerf import Network, default_cfg
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
model = MipNeRF360(d, opaque_background=True); model.load_state_dict(synth.background_state_dict(777, 2), strict=False); model = model.to(dev)
cfg = default_cfg(d); cfg.perturb = 0.0
net = Network(cfg, stage=3); net.load_state_dict(synth.human_state_dict(777, 2), strict=True); net = net.to(dev)
B, S = 1024, 64
b = {k: v.to(dev) for k, v in synth.stage1_batch(B, seed=777).items()}
mlp = model.mlps[1]
tdist = torch.linspace(0.2, 5.0, S + 1, device=dev).expand(B, S + 1).contiguous()
embed = mlp._embeds.view(mlp.store.param)[1]
P_H = 131072
xh = torch.randn(P_H, 3, device=dev) * 0.3
bufs = ops.mlp_chain_buffers(dev)
ws = [net._w(L) for L in net._nrf]
ops.mlp_chain_pack([w for w, _ in ws], [b_ for _, b_ in ws], bufs[0], bufs[1])
E, PE = torch.randn(P_H, 128, device=dev), torch.randn(P_H, 64, device=dev)
acts, xyz = [torch.empty(P_H, 128, device=dev) for _ in range(6)], torch.empty(P_H, 3, device=dev)
side = torch.cuda.Stream()
enc = lambda: ops.encode_ipe(tdist, b["rays_o"], b["rays_d"], b["radii"], mlp.pos_basis_t, embed, 576)
with torch.no_grad():
    ref = enc().clone(); torch.cuda.synchronize()
    for it in range(20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):
                ops.mlp_chain128_fwd(E, PE, xh, bufs[0], bufs[1], acts, xyz)
        out = enc(); torch.cuda.synchronize()
        if not torch.equal(out, ref):
            rows = (out != ref).any(1).nonzero().flatten()
            print("iteration", it, "bad rows", rows.numel(), rows[:40].tolist())
            names = ["z0", "z1", "z2"] + [f"cov{i}" for i in range(9)] + [f"lm{j}" for j in range(21)] + [f"lv{j}" for j in range(21)]
            for r in rows[:3].tolist():
                dbg_g, dbg_r = out[r, 504:558], ref[r, 504:558]
                diff = (dbg_g != dbg_r).nonzero().flatten().tolist()
                print(" row", r, "differing debug fields:", [names[i] for i in diff][:30])
                for i in diff[:8]:
                    print("    ", names[i], "ref %.9g got %.9g  (ref bits %08x got %08x)" % (float(dbg_r[i]), float(dbg_g[i]), dbg_r[i].view(torch.int32).item() & 0xffffffff, dbg_g[i].view(torch.int32).item() & 0xffffffff))
                ipe_diff = (out[r, :504] != ref[r, :504]).nonzero().flatten()
                print("    IPE columns differing:", ipe_diff.numel(), "first", ipe_diff[:12].tolist())
            break
    else:
        print("no differing run in 20")
