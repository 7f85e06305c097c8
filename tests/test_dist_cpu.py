"""N>1 path on CPU: world-size-2 gloo.  Ray sharding partitions the batch like the reference sampler
(S1/src/data/sampler.py:96) and the single flat-gradient all-reduce sums what per-parameter DDP buckets would."""
import json
import os
import socket
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, basedir, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hosnerf_amd import synth
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import allreduce_flat_grad, shard_rays
    torch.manual_seed(0)
    m = MipNeRF360(basedir, opaque_background=True)
    m.load_state_dict(synth.background_state_dict(777, 2), strict=False)
    batch = synth.stage1_batch(64, seed=1)
    mine = shard_rays(batch, rank, world)
    assert mine["rays_o"].shape[0] == 64 // world
    assert torch.equal(mine["rays_o"], batch["rays_o"][rank::world])
    # rank-dependent fake gradient written through the per-parameter views
    for i, p in enumerate(m.parameters()):
        p.grad.fill_(float(rank + 1) * (1 + (i % 3)))
    got_world = allreduce_flat_grad(m)
    assert got_world == world
    expect = sum(r + 1 for r in range(world))
    ok = all(torch.all(p.grad == expect * (1 + (i % 3))) for i, p in enumerate(m.parameters()))
    pad_ok = float(m.mlps[2]._views.W.view(m.flat_grad)[:, 283:].abs().max()) == 0     # padding never becomes non-zero
    # the enqueue-only form bench.py overlaps with the volume decoder's backward (N > 1): two spans, handles waited afterwards
    from hosnerf_amd.train import allreduce_flat_grad_async
    n_flat = m.flat_grad.numel()
    keep = m.flat_grad.clone()
    m.flat_grad.fill_(float(rank + 1))
    hs = allreduce_flat_grad_async(m, None, [(0, 1000), (n_flat - 1000, 1000)])
    for h in hs:
        h.wait()
    ok = ok and len(hs) == 2 and bool(torch.all(m.flat_grad[:1000] == expect)) and bool(torch.all(m.flat_grad[-1000:] == expect)) \
        and bool(torch.all(m.flat_grad[1000:n_flat - 1000] == rank + 1))
    m.flat_grad.copy_(keep)
    # inference side: contiguous ray ranges of a frame + one RGB all-gather (a frame of 101 rays: ragged over 2 ranks)
    from hosnerf_amd.train import gather_frame, shard_frame
    n = 101
    frame = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    idx, per = shard_frame(n, rank, world)
    full = gather_frame(frame[idx] * 2.0, n)           # "render" = times two
    gather_ok = bool(torch.equal(full, frame * 2.0)) and per == 51
    # the frame loop of eval.render_frame over the group, with a stand-in renderer (rgb = 2 * origin / 3 * origin): every
    # rank must end up with the whole frame, ragged chunks and ragged per-rank shares included
    from hosnerf_amd import eval as ev

    class _Stub:
        training = True

        class cfg:
            perturb = 1.0

        class human:
            @staticmethod
            def frame_prologue(**kw):
                return {"calls": 1}

        def eval(self): self.training = False
        def train(self, mode=True): self.training = mode
        def render(self, b, **kw):
            assert kw["prologue"] == {"calls": 1} and kw["with_cycle"] is False and b["rays"].shape[1] == b["near"].shape[0]
            return {"rgb": b["rays_o_bkg"] * 2.0}
        def render_bkg_only(self, bb, **kw): return bb["rays_o"] * 3.0

    g = torch.Generator().manual_seed(5)
    Hh, Ww = 3, 29
    rm = torch.rand(Hh * Ww, generator=g) > 0.55
    nf, nb = int(rm.sum()), int((~rm).sum())
    fr = {"img_height": Hh, "img_width": Ww, "ray_mask": rm, "ray_mask_bkg": ~rm, "time": torch.tensor(0.5), "bgcolor": torch.zeros(3),
          "rays": torch.rand(2, nf, 3, generator=g), "near": torch.rand(nf, 1, generator=g), "far": torch.rand(nf, 1, generator=g),
          "rays_o_bkg": torch.rand(nf, 3, generator=g), "rays_d_bkg": torch.rand(nf, 3, generator=g),
          "viewdirs_bkg": torch.rand(nf, 3, generator=g), "radii": torch.rand(nf, 1, generator=g),
          "rays_o_bkg_only": torch.rand(nb, 3, generator=g), "rays_d_bkg_only": torch.rand(nb, 3, generator=g),
          "viewdirs_bkg_only": torch.rand(nb, 3, generator=g), "radii_bkg_only": torch.rand(nb, 1, generator=g)}
    stub = _Stub()
    img = ev.render_frame(stub, fr, chunk_bkg=7, group=dist.group.WORLD)
    frame_ok = bool(torch.equal(img[rm], fr["rays_o_bkg"] * 2.0) and torch.equal(img[~rm], fr["rays_o_bkg_only"] * 3.0)) \
        and stub.training and stub.cfg.perturb == 1.0
    gather_ok = gather_ok and frame_ok
    torch.save({"ok": bool(ok), "pad_ok": pad_ok, "gather_ok": gather_ok, "sum": float(m.flat_grad.double().sum())},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    d = tempfile.mkdtemp(prefix="hos_dist_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    out = tempfile.mkdtemp(prefix="hos_dist_out_")
    mp.spawn(_worker, args=(2, _free_port(), d, out), nprocs=2, join=True)
    res = [torch.load(os.path.join(out, f"r{r}.pt")) for r in range(2)]
    assert all(r["ok"] and r["pad_ok"] and r["gather_ok"] for r in res)
    assert res[0]["sum"] == res[1]["sum"]


def _shard_worker(rank, world, port, basedir, out_dir):
    """`run.shard_decoder = True` checkpoints: `gather_decoder_shards(optimizer)` completes the parameters AND both Adam
    moment buffers on every rank, so the optimiser state rank 0 writes holds the rows of every owner."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.train import FusedAdam, FusedAdamOptimizer, ShardComm
    torch.manual_seed(0)
    net = Network(default_cfg(basedir), stage=2)
    net.shard_decoder(ShardComm(rank, world))
    fused = FusedAdam(net, lr=1e-3)
    opt = FusedAdamOptimizer(fused)
    mine = net.decoder_shard_spans()
    assert len(mine) == 3
    # what a few sharded steps leave behind: every rank has written ITS rows of the parameters and of both moments, the other
    # ranks' rows are stale (here: poisoned, so that a row nobody fetched is visible)
    for flat, base in ((net.flat_param, 1.0), (fused.exp_avg, 10.0), (fused.exp_avg_sq, 100.0)):
        with torch.no_grad():
            for n in net._shard_layers:
                off, row, cin = net._shard_rows(n)
                flat[off:off + cin * row].fill_(float("nan"))
            for off, n in mine:
                flat[off:off + n].fill_(base * (rank + 1))
    net.gather_decoder_shards(opt)
    ok = True
    for flat, base in ((net.flat_param, 1.0), (fused.exp_avg, 10.0), (fused.exp_avg_sq, 100.0)):
        for n in net._shard_layers:
            off, row, cin = net._shard_rows(n)
            cs = cin // world
            for r in range(world):
                ok = ok and bool(torch.all(flat[off + r * cs * row: off + (r + 1) * cs * row] == base * (r + 1)))
    sd = opt.state_dict()["fused"][0]
    ok = ok and bool(torch.isfinite(sd["exp_avg"]).all()) and bool(torch.isfinite(sd["exp_avg_sq"]).all())     # nothing stale is left to save
    torch.save({"ok": ok, "sum": float(sd["exp_avg"].double().sum())}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_decoder_checkpoint_carries_every_ranks_adam_state():
    d = tempfile.mkdtemp(prefix="hos_dist_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    out = tempfile.mkdtemp(prefix="hos_dist_out_")
    mp.spawn(_shard_worker, args=(2, _free_port(), d, out), nprocs=2, join=True)
    res = [torch.load(os.path.join(out, f"s{r}.pt")) for r in range(2)]
    assert all(r["ok"] for r in res)
    assert res[0]["sum"] == res[1]["sum"]          # the state rank 0 saves is the state every rank would save


def test_crash_line_is_the_last_word_of_a_killed_process():
    """bench.py arms libhoscomm's signal handlers around its optional multi-rank legs (include/hoscomm.h `hos_crash_line_set`): a process
    that dies of SIGABRT / SIGSEGV / SIGTERM inside a leg still prints the one JSON line it has and exits 0; cleared handlers die normally."""
    import subprocess
    import sys
    from hosnerf_amd import comm
    if not os.path.exists(comm.LIB_PATH):
        import pytest
        pytest.skip("libhoscomm.so not built (no RCCL under the ROCm path)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, signal, sys; sys.path.insert(0, %r); from hosnerf_amd import comm; "
            "comm.crash_line_set('{\"value\": 1}\\n'); SIG = getattr(signal, sys.argv[1]); "
            "comm.crash_line_clear() if sys.argv[2] == 'clear' else None; os.kill(os.getpid(), SIG); import time; time.sleep(5); print('survived')") % root
    for sig in ("SIGABRT", "SIGSEGV", "SIGTERM"):
        r = subprocess.run([sys.executable, "-c", code, sig, "armed"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout == '{"value": 1}\n', (sig, r.returncode, r.stdout, r.stderr[-300:])
    r = subprocess.run([sys.executable, "-c", code, "SIGTERM", "clear"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout == ""
