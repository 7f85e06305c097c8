"""Do the background kernels give the same bits when OTHER kernels run concurrently on a second stream?
One PropMLP query (encoder -> 4 x 256 planes layers -> density row dot) of 1024 rays x 64 samples is repeated on the main stream
while a disturber runs on a side stream: 'copy' (HBM traffic), 'matmul' (rocBLAS fp32), 'human' (the human network's forward).
Every intermediate is compared bit for bit with an undisturbed run.   python scripts/stress_concurrent.py [iters]"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops, synth
from hosnerf_amd.mipnerf360 import MipNeRF360
from hosnerf_amd.human_nerf import Network, default_cfg

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
model = MipNeRF360(d, opaque_background=True)
model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
model = model.to(dev)
B = 1024
b = {k: v.to(dev) for k, v in synth.stage1_batch(B, seed=777).items()}
mlp = model.mlps[int(os.environ.get("LEVEL", "0"))]
S = 64 if mlp.disable_rgb else 32
tdist = torch.linspace(0.2, 5.0, S + 1, device=dev).expand(B, S + 1).contiguous()
embed = mlp._embeds.view(mlp.store.param)[1]
SAVE = os.environ.get("SAVE", "1") == "1"


def query():
    X = ops.encode_ipe_planes(tdist, b["rays_o"], b["rays_d"], b["radii"], mlp.pos_basis_t, embed, 576, want_bf16=SAVE)
    density, rgb, saved = mlp._forward_planes(X, b["viewdirs"], B, S, save=SAVE)
    outs = {"X16": X[0].t, "density": density}
    if rgb is not None:
        outs["rgb"] = rgb
    if SAVE:
        sv = saved[0]
        outs["Xb"] = sv.Xb.t
        for i, (a, c) in enumerate(zip(sv.y16, sv.yb)):
            outs[f"y16[{i}]"] = a.t
            outs[f"yb[{i}]"] = c.t
            if a.bits is not None:
                outs[f"bits[{i}]"] = a.bits
    return outs


cfg = default_cfg(d)
cfg.perturb = 0.0
net = Network(cfg, stage=3)
net.load_state_dict(synth.human_state_dict(777, 2), strict=True)
net = net.to(dev)
hb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in synth.human_batch(2048, seed=3, time=0.5, is_train=True, iter_val=3e5).items()}
big = torch.randn(64 * 1024 * 1024, device=dev)
big2 = torch.empty_like(big)
ma = torch.randn(4096, 4096, device=dev)
side = torch.cuda.Stream()


P_H = 131072
xh = torch.randn(P_H, 3, device=dev) * 0.3
with torch.no_grad():
    PRO = net.frame_prologue(**hb)
bmin, bscale = hb["cnl_bbox_min_xyz"].contiguous(), hb["cnl_bbox_scale_xyz"].contiguous()
REP = int(os.environ.get("REP", "4"))


def disturb(kind):
    with torch.cuda.stream(side), torch.no_grad():
        if kind == "lbs":
            for _ in range(REP):
                ops.lbs_forward(xh, PRO["R_f"], PRO["T_f"], PRO["vol_cl"], bmin, bscale, 26)
        elif kind == "chain_save":
            for _ in range(REP):
                net._nonrigid_fwd(net._nrf, xh, PRO["cond"], PRO["band_w"], save=True)
        elif kind == "chain_nosave":
            for _ in range(REP):
                net._nonrigid_fwd(net._nrf, xh, PRO["cond"], PRO["band_w"], save=False)
        elif kind in ("pack_only", "chain_only"):
            specs = net._nrf
            bufs = net._chain_bufs.get("dbg")
            if bufs is None:
                bufs = net._chain_bufs["dbg"] = ops.mlp_chain_buffers(dev)
                ws = [net._w(L) for L in specs]
                ops.mlp_chain_pack([w for w, _ in ws], [b_ for _, b_ in ws], bufs[0], bufs[1])
                net._dbg = (torch.randn(P_H, 128, device=dev), torch.randn(P_H, 64, device=dev), [torch.empty(P_H, 128, device=dev) for _ in range(6)], torch.empty(P_H, 3, device=dev))
                torch.cuda.synchronize()
            E, PE, acts, xyz = net._dbg
            ws = [net._w(L) for L in specs]
            for _ in range(REP):
                if kind == "pack_only":
                    for _ in range(8):
                        ops.mlp_chain_pack([w for w, _ in ws], [b_ for _, b_ in ws], bufs[0], bufs[1])
                else:
                    ops.mlp_chain128_fwd(E, PE, xh, bufs[0], bufs[1], acts, xyz)
        elif kind == "embed":
            E = torch.empty(P_H, 128, device=dev); PE = torch.empty(P_H, 64, device=dev)
            for _ in range(REP * 3):
                ops.embed_hannw(xh, PRO["band_w"], PRO["cond"].reshape(-1), E, PE)
        elif kind == "canonical_save":
            for _ in range(REP):
                net._canonical_fwd(xh, 1, save=True)
        elif kind == "warp":
            for _ in range(REP):
                ops.human_sample_warp(hb["rays"][0].contiguous(), hb["rays"][1].contiguous(), hb["near"].contiguous(), hb["far"].contiguous(), 128,
                                      PRO["R_b"], PRO["T_b"], PRO["vol"], bmin, bscale, None, 26)
    with torch.cuda.stream(side):
        if kind == "copy":
            for _ in range(6):
                big2.copy_(big)
        elif kind == "matmul":
            for _ in range(4):
                torch.mm(ma, ma)
        elif kind == "human":
            with torch.no_grad():
                net(with_cycle=True, static_cycle=True, **hb)


with torch.no_grad():
    ref = {k: v.clone() for k, v in query().items()}
    torch.cuda.synchronize()
    KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["none", "copy", "matmul", "human"]
    for kind in KINDS:
        bad = {}
        for it in range(iters):
            side.wait_stream(torch.cuda.current_stream())
            if kind != "none":
                disturb(kind)
            outs = query()
            torch.cuda.synchronize()
            if os.environ.get("DUMP") == "1" and not torch.equal(outs["X16"], ref["X16"]) and not bad:
                Xg = (outs["X16"][:, :, 0, :].float() + outs["X16"][:, :, 1, :].float()).reshape(outs["X16"].shape[0], -1)
                Xr = (ref["X16"][:, :, 0, :].float() + ref["X16"][:, :, 1, :].float()).reshape(ref["X16"].shape[0], -1)
                rows = (Xg != Xr).any(1).nonzero().flatten()
                print("bad rows:", rows.tolist()[:70], "count", rows.numel())
                r0 = int(rows[0])
                cols = (Xg[r0] != Xr[r0]).nonzero().flatten()
                print("row", r0, "bad cols", cols.numel(), "first", cols[:24].tolist())
                print(" got", [round(float(x), 5) for x in Xg[r0, cols[:10]]])
                print(" ref", [round(float(x), 5) for x in Xr[r0, cols[:10]]])
                # is the wrong row some other row's correct content?
                d = (Xr[:, :504] - Xg[r0, :504]).abs().max(1).values
                print(" closest ref row to the wrong row:", int(d.argmin()), float(d.min()))
                for r1 in rows[:6].tolist():
                    c1 = (Xg[r1] != Xr[r1]).nonzero().flatten()
                    print("  row", r1, "ncols", c1.numel(), "col range", int(c1.min()), int(c1.max()), "max |got|", float(Xg[r1].abs().max()), "nan", int(torch.isnan(Xg[r1]).sum()))
            for k, v in outs.items():
                if not torch.equal(v, ref[k]):
                    n = int((v != ref[k]).sum()) if v.dtype != torch.bfloat16 and v.dtype != torch.float16 else int((v.view(torch.int16) != ref[k].view(torch.int16)).sum())
                    bad.setdefault(k, []).append((it, n))
        print(f"disturber {kind:7s}: " + ("all bit-identical" if not bad else "; ".join(f"{k}: {len(v)} of {iters} runs differ (first at {v[0][0]}, {v[0][1]} elements)" for k, v in bad.items())), flush=True)
