"""Which kernels compute different bits while chain128_kernel runs on another stream?  python scripts/stress_victims.py [iters]"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.mipnerf360 import MipNeRF360
from hosnerf_amd.human_nerf import Network, default_cfg

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
model = MipNeRF360(d, opaque_background=True)
model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
model = model.to(dev)
cfg = default_cfg(d); cfg.perturb = 0.0
net = Network(cfg, stage=3)
net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
net = net.to(dev)
B, S = 1024, 64
b = {k: v.to(dev) for k, v in synth.stage1_batch(B, seed=777).items()}
mlp = model.mlps[1]
tdist = torch.linspace(0.2, 5.0, S + 1, device=dev).expand(B, S + 1).contiguous()
embed = mlp._embeds.view(mlp.store.param)[1]
xs = torch.randn(65536, 576, device=dev)
w_prev = torch.rand(B, 64, device=dev) + 0.01
w_prev = w_prev / w_prev.sum(-1, keepdim=True)
sd_prev = torch.sort(torch.rand(B, 65, device=dev), -1).values
jit = torch.rand(B, device=dev)
P_H = 131072
xh = torch.randn(P_H, 3, device=dev) * 0.3
bufs = ops.mlp_chain_buffers(dev)
ws = [net._w(L) for L in net._nrf]
ops.mlp_chain_pack([w for w, _ in ws], [b_ for _, b_ in ws], bufs[0], bufs[1])
E, PE = torch.randn(P_H, 128, device=dev), torch.randn(P_H, 64, device=dev)
acts, xyz = [torch.empty(P_H, 128, device=dev) for _ in range(6)], torch.empty(P_H, 3, device=dev)
side = torch.cuda.Stream()
REP = int(os.environ.get("REP", "4"))

victims = {
    "encode_fp32": lambda: ops.encode_ipe(tdist, b["rays_o"], b["rays_d"], b["radii"], mlp.pos_basis_t, embed, 576),
    "encode_planes": lambda: ops.encode_ipe_planes(tdist, b["rays_o"], b["rays_d"], b["radii"], mlp.pos_basis_t, embed, 576, want_bf16=True)[0].t,
}
with torch.no_grad():
    for name, fn in victims.items():
        ref = fn().clone()
        torch.cuda.synchronize()
        res = {}
        for mode in ("quiet", "chain128"):
            nbad = 0
            for it in range(iters):
                side.wait_stream(torch.cuda.current_stream())
                if mode == "chain128":
                    with torch.cuda.stream(side):
                        for _ in range(REP):
                            ops.mlp_chain128_fwd(E, PE, xh, bufs[0], bufs[1], acts, xyz)
                out = fn()
                torch.cuda.synchronize()
                nbad += int(not torch.equal(out.view(torch.int16) if out.element_size() == 2 else out, ref.view(torch.int16) if ref.element_size() == 2 else ref))
            res[mode] = nbad
        print(f"victim {name:14s}: runs that differ  quiet {res['quiet']}/{iters}   with chain128 on the side stream {res['chain128']}/{iters}", flush=True)
