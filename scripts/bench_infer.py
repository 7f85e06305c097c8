"""Inference throughput (1080p free-viewpoint frames, forward only) on one MI355X.
Renders whole synthetic 1920x1080 frames with `hosnerf_amd.eval.render_frame` -- the reference's `free_view` loop
(M:1293-1494): rays through the subject's box go through both branches and the 160-sample merged composite, the rest
are background-only -- including the device-side ray set-up of the frame (`eval.frame_rays`).  Rays are independent,
so N GPUs render N disjoint ray ranges of a frame (one all-gather of [rays,3] RGB per list, 24.9 MB per frame).
  python scripts/bench_infer.py [--chunk 65536] [--frames 2] [--no-cache]"""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

FWD_FLOP_BG = 651.0e6        # background forward per ray
FWD_FLOP_FG = 811.3e6        # foreground ray: + 160.2 M for the human branch core


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=65536, help="rays per call (the reference's chunk_bkg is 8192)")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--no-cache", action="store_true", help="rebuild the human prologue for every chunk, like the reference")
    args = ap.parse_args()
    from hosnerf_amd import eval as ev, synth
    from hosnerf_amd.hosnerf import HOSNeRF
    from hosnerf_amd.human_nerf import default_cfg
    dev = torch.device("cuda")
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    cfg = default_cfg(d)
    cfg.chunk = min(max(args.chunk, int(cfg.chunk)), 32768)     # human inner chunk: 32768 rays x 128 samples = 4.3 GB per activation
    hos = HOSNeRF(cfg)
    hos.model.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    hos.human.load_state_dict(synth.human_state_dict(777, 2), strict=True)
    hos = hos.to(dev)
    H, W = args.height, args.width
    hb = synth.human_batch(8, seed=2, time=0.5, is_train=False, iter_val=3e5)
    K, E, Ec = synth.eval_camera(H, W, hb)
    bbox = {"min_xyz": hb["dst_bbox_min_xyz"].numpy(), "max_xyz": hb["dst_bbox_max_xyz"].numpy()}
    per_frame = {k: (hb[k].to(dev) if isinstance(hb[k], torch.Tensor) else hb[k]) for k in ev.FRAME_KEYS}

    def one_frame():
        fr = ev.frame_rays(H, W, K, E, bbox, Ec, device=dev)
        fr.update(per_frame)
        return fr, ev.render_frame(hos, fr, chunk_bkg=args.chunk, cache_prologue=not args.no_cache)

    fr, img = one_frame()                                   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        fr, img = one_frame()
    torch.cuda.synchronize()
    t_frame = (time.perf_counter() - t0) / args.frames
    t0 = time.perf_counter()
    for _ in range(5):
        ev.frame_rays(H, W, K, E, bbox, Ec, device=dev)
    torch.cuda.synchronize()
    t_rays = (time.perf_counter() - t0) / 5
    n_rays = H * W
    n_fg = int(fr["ray_mask"].sum())
    flops = (n_rays - n_fg) * FWD_FLOP_BG + n_fg * FWD_FLOP_FG
    print(json.dumps({"metric": "inference rays/s, whole synthetic frame, 1 GPU", "height": H, "width": W, "chunk": args.chunk,
                      "prologue": "per chunk" if args.no_cache else "per frame", "foreground_fraction": n_fg / n_rays,
                      "rays_per_s_frame": n_rays / t_frame, "frames_per_s": 1.0 / t_frame, "ms_per_frame": 1e3 * t_frame,
                      "ms_ray_setup": 1e3 * t_rays, "algorithmic_tflops": flops / t_frame / 1e12, "gemm": "planes",
                      "finite": bool(torch.isfinite(img).all())}))


if __name__ == "__main__":
    main()
