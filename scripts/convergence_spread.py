"""One-off experiment behind tests/test_gpu_convergence.py: how far apart do two fp32 trainings of the SAME path land?  The HIP
path is trained K times from initial weights that differ by a relative 1e-7 perturbation (one fp32 ulp: a stand-in for "another
fp32 evaluation order"), per regime (steps, learning-rate scale); the held-out PSNR spread is what a single HIP-vs-oracle
comparison can resolve.  Usage (GPU box): python scripts/convergence_spread.py > gpurun_out/convergence_spread.jsonl"""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hosnerf_amd import synth  # noqa: E402
from tests import test_gpu_convergence as tc  # noqa: E402


def perturbed(sd, k):
    if k == 0:
        return sd
    g = torch.Generator().manual_seed(100 + k)
    return {n: (v * (1.0 + 1e-7 * torch.randn(v.shape, generator=g))).to(v.dtype) if v.is_floating_point() else v for n, v in sd.items()}


def main():
    dev = torch.device("cuda")
    K = int(os.environ.get("K", "4"))
    scene, px, rays = tc._make_scene(tempfile.mkdtemp(prefix="hos_spread_"), dev)
    regimes1 = [(400, 1.0), (400, 0.3), (1200, 1.0), (1200, 0.3)]
    regimes2 = [(300, 1.0), (300, 0.3), (900, 1.0), (900, 0.3)]
    stages = os.environ.get("STAGES", "1,2").split(",")
    if os.environ.get("MODE") == "pairs":
        # (HIP, oracle) pairs from the same perturbed initial weights: is there an OFFSET between the two paths beyond their spreads?
        b0, h0 = synth.background_state_dict(777, 2), synth.human_state_dict(777, 2)
        for steps, scale, ks in ((400, 1.0, 2), (400, 0.3, 3)):
            out = []
            for k in range(ks):
                r, _, m = tc._train_stage1(rays, dev, perturbed(b0, k), steps, scale, oracle=True)
                del m
                out.append((r["psnr_hip"], r["psnr_oracle"]))
                torch.cuda.empty_cache()
            print(json.dumps({"stage": 1, "steps": steps, "lr_scale": scale, "psnr_hip_oracle": out}), flush=True)
        tc.S2_DECAY_STEPS = 1
        for steps, scale, ks in ((500, 0.3, 3),):
            out = []
            for k in range(ks):
                r, _, _ = tc._train_stage2(scene, px, dev, perturbed(h0, k), steps, scale, oracle=True)
                out.append((r["psnr_hip"], r["psnr_oracle"]))
                torch.cuda.empty_cache()
            print(json.dumps({"stage": 2, "steps": steps, "lr_scale": scale, "fast_decay": 1, "psnr_hip_oracle": out}), flush=True)
        return
    if "1" in stages:
        b0 = synth.background_state_dict(777, 2)
        for steps, scale in regimes1:
            ps, ls = [], []
            for k in range(K):
                r, _, m = tc._train_stage1(rays, dev, perturbed(b0, k), steps, scale, oracle=False)
                del m
                ps.append(r["psnr_hip"]); ls.append(r["loss_last20_mean"][0])
            print(json.dumps({"stage": 1, "steps": steps, "lr_scale": scale, "psnr": ps, "spread_db": max(ps) - min(ps), "loss_last20": ls}), flush=True)
    if "2" in stages:
        h0 = synth.human_state_dict(777, 2)
        for decay in (0, 1):
            tc.S2_DECAY_STEPS = decay
            for steps, scale in regimes2:
                ps, ls = [], []
                for k in range(K):
                    r, _, _ = tc._train_stage2(scene, px, dev, perturbed(h0, k), steps, scale, oracle=False)
                    ps.append(r["psnr_hip"]); ls.append(r["loss_last20_mean"][0])
                print(json.dumps({"stage": 2, "steps": steps, "lr_scale": scale, "fast_decay": decay, "psnr": ps, "spread_db": max(ps) - min(ps), "loss_last20": ls}), flush=True)


if __name__ == "__main__":
    main()
