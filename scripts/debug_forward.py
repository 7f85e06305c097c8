import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import oracle.background as ob
from hosnerf_amd import synth, ops
from hosnerf_amd.mipnerf360 import MipNeRF360
d = tempfile.mkdtemp(); json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
dev = torch.device("cuda")
sd = synth.background_state_dict(777, 2)
m = MipNeRF360(d, opaque_background=True); print(m.load_state_dict(sd, strict=False)); m = m.to(dev)
print("param ids stable:", all(p.data_ptr() >= m.flat_param.data_ptr() and p.data_ptr() < m.flat_param.data_ptr()+4*m.flat_param.numel() for p in m.parameters()))
B = 8
batch = synth.stage1_batch(B, seed=5)
gb = {k: v.to(dev) for k, v in batch.items()}
def E(a, b): return float((a.detach().cpu().double() - b.detach().double()).abs().max())
rend_o, hist_o = ob.mipnerf360_forward(sd, batch, 0.5, False, 0.1, 1e6, transitions_times=[0.4])
with torch.no_grad():
    rend, hist = m(gb, 0.5, False, False, 0.1, 1e6)
for l in range(3):
    print(l, {k: E(hist[l][k], hist_o[l][k]) for k in ("sdist", "density", "weights", "rgb")}, "render", E(rend[l]["rgb"], rend_o[l]["rgb"]))
# stage by stage at level 0 using oracle inputs
h0 = hist_o[0]
tdist = h0["tdist"].to(dev)
mlp = m.mlps[0]
X = ops.encode_ipe(tdist, gb["rays_o"], gb["rays_d"], gb["radii"], mlp.pos_basis_t, mlp._embeds.view(m.flat_param)[1], 576)
means, covs = ob.cast_rays_cone(h0["tdist"], batch["rays_o"], batch["rays_d"], batch["radii"])
Xo = ob.encode_samples(means, covs, ob.generate_basis())
print("X", E(X.view(B, 64, 576)[..., :504], Xo), "embed", E(X.view(B,64,576)[..., 504:568], sd["mlps.0.bkgd_stateembeds.1"].expand(B,64,64)))
wts = ob.MLPWeights(sd, "mlps.0.")
x = torch.cat([Xo, wts.embeds[1].repeat(B, 64, 1)], -1)
with torch.no_grad():
    dens, rgb, _ = mlp._forward_impl(X, gb["viewdirs"], B, 64, save=True)
    acts = _[1]
hh = x
for i, (W_, b_) in enumerate(wts.pts):
    hh = torch.relu(torch.nn.functional.linear(hh, W_, b_))
    print("layer", i, E(acts[i].view(B, 64, -1), hh), float(hh.abs().max()))
raw = torch.nn.functional.linear(hh, *wts.density)[..., 0]
print("density", E(dens.view(B, 64), torch.nn.functional.softplus(raw - 1)))
