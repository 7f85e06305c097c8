"""Golden vectors for the stage-3 AND stage-2 training losses: the REFERENCE's own `LitMipNeRF360.get_loss` /
`flow_func` / `img2mae` / `_unpack_imgs` (3rd_Complete_HOSNeRF/src/model/mipnerf360/model.py:41-71, 1680-1716;
2nd_State_Conditional_Human-Object/src/model/mipnerf360/model.py:41-71, 908-944) with the configured weights
of the non-LPIPS terms (configs/default.yaml: mse 0.2, flow 0.01, cycle 0.01), values and gradients, on seeded inputs.
Stage-2 cases (`s2_*`) use PARTIAL patch masks (rays only where the patch overlaps the subject's box), so `_unpack_imgs`'s
background fill enters the MSE.
  python tests/golden/make_golden_losses.py   ->  tests/golden/losses.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import refload


def main():
    out = {}
    with refload.stage(3):
        import importlib
        M = importlib.import_module("src.model.mipnerf360.model")
        lit = object.__new__(M.LitMipNeRF360)
        torch.nn.Module.__init__(lit)
        lit.cfg = types.SimpleNamespace(train=types.SimpleNamespace(lossweights={"mse": 0.2, "flow": 0.01, "cycle": 0.01}))
        N, P, S = 2, 4, 32
        B = N * P * P
        for tag, seed, time, fg_mode in (("mix", 1, 0.5, "mix"), ("nofg", 2, 0.5, "none"), ("t0", 3, 0.001, "mix")):
            g = torch.Generator().manual_seed(seed)
            r = lambda *s: torch.rand(*s, generator=g)
            rgb = r(B, 3).requires_grad_(True)
            targets = r(N, P, P, 3)
            idx_fg = (r(B) > 0.4) if fg_mode == "mix" else torch.zeros(B, dtype=torch.bool)
            nfg = int(idx_fg.sum())
            hw = (r(nfg, S) * 0.05).requires_grad_(True)
            pts_prev = (torch.randn(B, S, 3, generator=g) * 0.3).requires_grad_(True)
            ncyc = 57
            observe = torch.randn(ncyc, 3, generator=g)
            deform = (observe + 0.05 * torch.randn(ncyc, 3, generator=g)).requires_grad_(True)
            ray_grid = torch.cat([r(B, 2) * 60, torch.randn(B, 2, generator=g), (r(B, 1) > 0.3).float()], -1)
            cam = torch.eye(4); cam[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]; cam[:3, 3] = torch.tensor([0.1, -0.2, 3.0])
            Kp = torch.tensor([[55.0, 0.0, 30.0], [0.0, 57.0, 28.0], [0.0, 0.0, 1.0]])
            net_output = {"rgb": rgb, "deform_pts_prev_final": pts_prev, "observe_pts": observe, "deform_pts_final": deform}
            total, parts = M.LitMipNeRF360.get_loss(lit, net_output, torch.ones(N, P, P, dtype=torch.bool), torch.zeros(3), targets,
                                                    ray_grid, time, idx_fg, hw, torch.tensor([0, P * P, 2 * P * P]), cam, Kp)
            total.backward()
            p = tag + "_"
            out.update({p + "rgb": rgb.detach().numpy(), p + "targets": targets.numpy(), p + "idx_fg": idx_fg.numpy(),
                        p + "hw": hw.detach().numpy(), p + "pts_prev": pts_prev.detach().numpy(), p + "observe": observe.numpy(),
                        p + "deform": deform.detach().numpy(), p + "ray_grid": ray_grid.numpy(), p + "cam": cam.numpy(), p + "K": Kp.numpy(),
                        p + "time": time, p + "total": float(total),
                        p + "mse": float(parts["mse"]), p + "flow": float(parts["flow"]), p + "cycle": float(parts["cycle"]),
                        p + "g_rgb": rgb.grad.numpy(), p + "g_deform": deform.grad.numpy(),
                        p + "g_hw": (hw.grad if hw.grad is not None else torch.zeros_like(hw)).numpy(),
                        p + "g_pts_prev": (pts_prev.grad if pts_prev.grad is not None else torch.zeros_like(pts_prev)).numpy()})
            print(tag, float(total), {k: float(v) for k, v in parts.items()})
    with refload.stage(2):
        import importlib
        M2 = importlib.import_module("src.model.mipnerf360.model")
        lit = object.__new__(M2.LitMipNeRF360)
        torch.nn.Module.__init__(lit)
        lit.cfg = types.SimpleNamespace(train=types.SimpleNamespace(lossweights={"mse": 0.2, "flow": 0.01, "cycle": 0.01}))
        N, P, S = 2, 6, 24
        for tag, seed, time, ncyc in (("s2_mix", 11, 0.5, 41), ("s2_t0", 12, 0.001, 17), ("s2_one", 13, 0.7, 1)):
            g = torch.Generator().manual_seed(seed)
            r = lambda *s: torch.rand(*s, generator=g)
            patch_masks = r(N, P, P) > 0.35
            div = torch.tensor([0, int(patch_masks[0].sum()), int(patch_masks.sum())])
            B = int(patch_masks.sum())
            rgb = r(B, 3).requires_grad_(True)
            targets = r(N, P, P, 3)
            bgcolor = r(3) * 255.0
            weights = (r(B, S) * 0.05).requires_grad_(True)
            pts_prev = (torch.randn(B, S, 3, generator=g) * 0.3).requires_grad_(True)
            observe = torch.randn(ncyc, 3, generator=g)
            deform = (observe + (0.05 if ncyc > 1 else 0.0) * torch.randn(ncyc, 3, generator=g)).requires_grad_(True)
            ray_grid = torch.cat([r(B, 2) * 60, torch.randn(B, 2, generator=g), (r(B, 1) > 0.3).float()], -1)
            cam = torch.eye(4); cam[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]; cam[:3, 3] = torch.tensor([0.1, -0.2, 3.0])
            Kp = torch.tensor([[55.0, 0.0, 30.0], [0.0, 57.0, 28.0], [0.0, 0.0, 1.0]])
            net_output = {"rgb": rgb, "weights": weights, "deform_pts_prev_final": pts_prev, "observe_pts": observe, "deform_pts_final": deform}
            total, parts = M2.LitMipNeRF360.get_loss(lit, net_output, patch_masks, bgcolor / 255.0, targets,
                                                     ray_grid if time > 0.005 else None, time, div,
                                                     cam if time > 0.005 else None, Kp if time > 0.005 else None)
            total.backward()
            p = tag + "_"
            z = lambda x: (x.grad if x.grad is not None else torch.zeros_like(x)).numpy()
            out.update({p + "rgb": rgb.detach().numpy(), p + "targets": targets.numpy(), p + "patch_masks": patch_masks.numpy(),
                        p + "div": div.numpy(), p + "bgcolor": bgcolor.numpy(), p + "weights": weights.detach().numpy(),
                        p + "pts_prev": pts_prev.detach().numpy(), p + "observe": observe.numpy(), p + "deform": deform.detach().numpy(),
                        p + "ray_grid": ray_grid.numpy(), p + "cam": cam.numpy(), p + "K": Kp.numpy(), p + "time": time,
                        p + "total": float(total), p + "mse": float(parts["mse"]), p + "flow": float(parts["flow"]),
                        p + "cycle": float(parts["cycle"]), p + "g_rgb": z(rgb), p + "g_deform": z(deform),
                        p + "g_weights": z(weights), p + "g_pts_prev": z(pts_prev)})
            print(tag, float(total), {k: float(v) for k, v in parts.items()})
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


if __name__ == "__main__":
    main()
