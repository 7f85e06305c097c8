"""Inference throughput (1080p free-viewpoint frames, forward only) on one MI355X.
A 1920x1080 frame = 2 073 600 rays; ~25 % of them cross the human bounding box and go through both branches and the
160-sample merged composite, the rest are background-only.  Rays are independent, so N GPUs render N disjoint ray
ranges of a frame (one all-gather of [rays,3] RGB per frame, 24.9 MB) -- this script times one GPU's share per chunk.
  python scripts/bench_infer.py [--chunk 16384] [--frames 1]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

FWD_FLOP_BG = 651.0e6        # background forward per ray
FWD_FLOP_FG = 811.3e6        # foreground ray: + 160.2 M for the human branch core


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=16384, help="rays per call")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--fg-frac", type=float, default=0.25)
    args = ap.parse_args()
    from hosnerf_amd import ops, synth
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    dev = torch.device("cuda")
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    cfg = default_cfg(d)
    cfg.perturb = 0.0
    cfg.chunk = args.chunk
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    hos = hos.to(dev).eval()
    from hosnerf_amd.mipnerf360 import MipNeRF360
    bg_model = MipNeRF360(d, opaque_background=True)          # background-only rays: rendered levels (stage-1 style)
    bg_model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    bg_model = bg_model.to(dev).eval()
    n_rays = 1920 * 1080
    n_fg = int(n_rays * args.fg_frac)
    n_bg = n_rays - n_fg
    bg = {k: v.to(dev) for k, v in synth.stage1_batch(args.chunk, seed=1).items()}
    bg["times"] = 0.5
    fb = synth.human_batch(args.chunk, seed=2, time=0.5, is_train=False, iter_val=3e5)
    fg = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in fb.items()}

    def bg_chunk():
        with torch.no_grad():
            rend, _ = bg_model(bg, 1.0, False, False, 0.1, 1e6)
        return rend[-1]["rgb"]

    def fg_chunk():
        with torch.no_grad():
            return hos.render(fg, randomized=False, is_train=False)["rgb"]

    res = {}
    for name, fn in (("background-only", bg_chunk), ("foreground (both branches + merge)", fg_chunk)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 6
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = args.chunk * n / (time.perf_counter() - t0)
    t_frame = n_bg / res["background-only"] + n_fg / res["foreground (both branches + merge)"]
    flops = n_bg * FWD_FLOP_BG + n_fg * FWD_FLOP_FG
    print(json.dumps({"metric": "inference rays/s, 1080p frame (25 % foreground), 1 GPU", "chunk": args.chunk,
                      "rays_per_s_background": res["background-only"], "rays_per_s_foreground": res["foreground (both branches + merge)"],
                      "rays_per_s_frame": n_rays / t_frame, "frames_per_s": 1.0 / t_frame,
                      "algorithmic_tflops": flops / t_frame / 1e12, "gemm": "planes"}))


if __name__ == "__main__":
    main()
