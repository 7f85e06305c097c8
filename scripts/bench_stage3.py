"""Stage-2 / stage-3 training-step timing on one MI355X -- coverage of the P* / C* scope
rows; the headline metric stays bench.py (stage 1).  Prints one JSON line per stage.
  python scripts/bench_stage3.py [--rays 2048] [--steps 10] [--warmup 3] [--gemm planes|split|fp32]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

S2_TRAIN_FLOP_PER_RAY = 558e6 + 77.7e6          # 558 M + 77.7 M x f_cyc (f_cyc ~ 1 for rays aimed at the body)
S3_TRAIN_FLOP_PER_RAY = 2261e6 + 77.7e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gemm", choices=["planes", "split", "fp32"], default="planes")
    ap.add_argument("--only", choices=["stage3", "stage2-human-only"], default=None)
    args = ap.parse_args()
    from hosnerf_amd import ops, synth
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    from hosnerf_amd.train import FusedAdam, human_lr_ranges, train_step_stage3
    dev = torch.device("cuda")
    ops.set_gemm_mode({"planes": ops.GEMM_PLANES, "split": ops.GEMM_BF16X3, "fp32": ops.GEMM_FP32}[args.gemm])
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    cfg = default_cfg(d)
    cfg.perturb = 1.0
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    hos = hos.to(dev)
    b = synth.human_batch(args.rays, seed=777, time=0.5, is_train=True, iter_val=3e5)
    b["ray_grid"] = torch.cat([torch.rand(args.rays, 2) * 100, torch.randn(args.rays, 2), torch.ones(args.rays, 1)], -1)
    b["newsmpl_to_camera_prev"] = torch.eye(4)
    b["newsmpl_to_camera_prev"][2, 3] = 3.0
    b["intrinsics_prev"] = torch.tensor([[500.0, 0, 50], [0, 500.0, 50], [0, 0, 1]])
    from hosnerf_amd.train import batch_to_device
    gb = batch_to_device(b, dev)           # control scalars (time, iter_val) stay on the host: no round trip per step
    ob_ = FusedAdam(hos.model, lr=6.667e-5)
    oh_ = FusedAdam(hos.human, lr=6.667e-5, lr_ranges=human_lr_ranges(hos.human))

    def s3():
        return train_step_stage3(hos, ob_, oh_, gb)[0]

    def s2():                                       # human branch alone: forward (flow + cycle sets) + mse loss + backward + Adam
        oh_.zero_grad()
        out = hos.human(**gb)
        loss = (out["human_rgb"] ** 2).mean() + (out["human_density"] ** 2).mean() * 1e-3
        if "deform_pts_prev_final" in out:
            loss = loss + (out["deform_pts_prev_final"] ** 2).mean() * 1e-3 + (out["deform_pts_final"] ** 2).mean() * 1e-3
        loss.backward()
        oh_.step(6.667e-5)
        return loss.detach()

    for name, fn, flop in (("stage3", s3, S3_TRAIN_FLOP_PER_RAY), ("stage2-human-only", s2, S2_TRAIN_FLOP_PER_RAY)):
        if args.only is not None and name != args.only:
            continue
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = ops.KernelEvents()
        ops.set_kernel_events(prof)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ops.set_kernel_events(None)
        table = prof.summary()
        print(json.dumps({"stage": name, "metric": "train rays/s (eager launches)", "value": args.rays * args.steps / dt,
                          "ms_per_step": 1e3 * dt / args.steps, "rays": args.rays, "gemm": args.gemm, "loss": float(loss),
                          "algorithmic_tflops": args.rays * args.steps * flop / dt / 1e12,
                          "gemm_ms_per_step": sum(r["total_ms"] for r in table) / 3, "kernels": table[:12]}))


if __name__ == "__main__":
    main()
