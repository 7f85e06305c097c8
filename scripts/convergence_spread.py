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
    if os.environ.get("MODE") == "reference_rates":
        # The REFERENCE's own learning rates (stage 1: 2e-3 -> 2e-5 with warm-up; stage 2: 6.667e-4 with its flat 500 k-step decay),
        # where one comparison cannot resolve 0.1 dB (the spread of one path alone is larger): K (HIP, oracle) pairs from perturbed
        # initial weights -- do the two DISTRIBUTIONS of held-out PSNR coincide?
        import statistics as st
        b0, h0 = synth.background_state_dict(777, 2), synth.human_state_dict(777, 2)
        K1, K2 = int(os.environ.get("K1", "4")), int(os.environ.get("K2", "4"))
        out = []
        for k in range(K1):
            r, _, m = tc._train_stage1(rays, dev, perturbed(b0, k), 800, 1.0, oracle=True)
            del m
            out.append((r["psnr_hip"], r["psnr_oracle"]))
            torch.cuda.empty_cache()
        h, o = [a for a, _ in out], [b for _, b in out]
        print(json.dumps({"stage": 1, "steps": 800, "lr_scale": 1.0, "psnr_hip_oracle": out, "hip_mean_sd": [st.mean(h), st.pstdev(h)],
                          "oracle_mean_sd": [st.mean(o), st.pstdev(o)], "mean_difference_db": st.mean(h) - st.mean(o)}), flush=True)
        tc.S2_DECAY_STEPS = 0
        out = []
        for k in range(K2):
            r, _, _ = tc._train_stage2(scene, px, dev, perturbed(h0, k), 400, 1.0, oracle=True)
            out.append((r["psnr_hip"], r["psnr_oracle"]))
            torch.cuda.empty_cache()
        h, o = [a for a, _ in out], [b for _, b in out]
        print(json.dumps({"stage": 2, "steps": 400, "lr_scale": 1.0, "fast_decay": 0, "psnr_hip_oracle": out, "hip_mean_sd": [st.mean(h), st.pstdev(h)],
                          "oracle_mean_sd": [st.mean(o), st.pstdev(o)], "mean_difference_db": st.mean(h) - st.mean(o)}), flush=True)
        return
    if os.environ.get("MODE") == "stage3_regimes":
        # stage 3 warm-started from ONE pair of HIP-trained modules; (steps, lr scale), K (HIP, oracle) pairs from perturbed copies
        _, bsd, m = tc._train_stage1(rays, dev, oracle=False)
        del m
        _, hsd, _ = tc._train_stage2(scene, px, dev, oracle=False)
        torch.cuda.empty_cache()
        for steps, scale, ks in ((150, 1.0, 4), (150, 0.3, 4), (300, 0.3, 3)):
            out = []
            for k in range(ks):
                r = tc._train_stage3(scene, px, dev, perturbed(bsd, k), perturbed(hsd, k), steps, scale, oracle=True)
                out.append((r["psnr_hip"], r["psnr_oracle"]))
                torch.cuda.empty_cache()
            print(json.dumps({"stage": 3, "steps": steps, "lr_scale": scale, "psnr_hip_oracle": out, "abs_diff": [abs(a - b) for a, b in out]}), flush=True)
        return
    if os.environ.get("MODE") == "stage2_regimes":
        # candidate stage-2 regimes for the test: (steps, lr scale, patches per item), K pairs each
        h0 = synth.human_state_dict(777, 2)
        tc.S2_DECAY_STEPS = 1
        for steps, scale, patches, ks in ((900, 0.3, 1, 3), (700, 0.2, 2, 2)):
            tc.S2_PATCHES = patches
            out = []
            for k in range(ks):
                r, _, _ = tc._train_stage2(scene, px, dev, perturbed(h0, k), steps, scale, oracle=True)
                out.append((r["psnr_hip"], r["psnr_oracle"]))
                torch.cuda.empty_cache()
            print(json.dumps({"stage": 2, "steps": steps, "lr_scale": scale, "patches": patches, "fast_decay": 1, "psnr_hip_oracle": out,
                              "abs_diff": [abs(a - b) for a, b in out]}), flush=True)
        return
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
