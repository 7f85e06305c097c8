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
    # inference side: contiguous ray ranges of a frame + one RGB all-gather (a frame of 101 rays: ragged over 2 ranks)
    from hosnerf_amd.train import gather_frame, shard_frame
    n = 101
    frame = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    idx, per = shard_frame(n, rank, world)
    full = gather_frame(frame[idx] * 2.0, n)           # "render" = times two
    gather_ok = bool(torch.equal(full, frame * 2.0)) and per == 51
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
