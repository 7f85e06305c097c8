"""Multi-rank paths on the 1-GPU box:
  * RCCL executes: a 1-rank `nccl` process group all-reduces the flat gradient and FusedAdam steps through it;
  * bench.py's N > 1 code path -- stage-3, 4096 GLOBAL rays split over the ranks (strong scaling), fwd+bwd hipGraph replay +
    eager all-reduce of both flat gradients + dynamic Adam, max-over-ranks timing -- runs as two processes sharing cuda:0
    (HOS_BENCH_ONE_GPU=1: gloo carries the collectives, RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RCCL_SELF_TEST = r"""
import os, sys, json, tempfile
sys.path.insert(0, os.environ["HOS_ROOT"])
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from hosnerf_amd import synth
from hosnerf_amd.mipnerf360 import MipNeRF360
from hosnerf_amd.train import FusedAdam, allreduce_flat_grad, stage1_loss
d = tempfile.mkdtemp()
json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
m = MipNeRF360(d, opaque_background=True)
m.load_state_dict(synth.background_state_dict(777, 2), strict=False)
m = m.to(dev)
opt = FusedAdam(m, lr=1e-3, max_grad_norm=0.001)
b = {k: v.to(dev) for k, v in synth.stage1_batch(64, seed=1).items()}
opt.zero_grad()
rend, hist = m(b, 0.5, True, True, 0.1, 1e6)
loss, _ = stage1_loss(rend[-1]["rgb"], b["target"], hist)
loss.backward()
g0 = m.flat_grad.clone()
assert allreduce_flat_grad(m) == 1                      # world 1: no collective issued by the helper
dist.all_reduce(m.flat_grad)                            # RCCL all-reduce of the 38 MB flat gradient, sum over 1 rank = identity
torch.cuda.synchronize()
assert torch.equal(m.flat_grad, g0)
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t); dist.barrier()
out = torch.empty(8, device=dev); dist.all_gather_into_tensor(out, t)
assert torch.equal(out, t)
p0 = m.flat_param.clone()
opt.step(1e-3)
torch.cuda.synchronize()
assert not torch.equal(p0, m.flat_param) and bool(torch.isfinite(m.flat_param).all())
print("RCCL_OK", dist.get_backend())
dist.destroy_process_group()
"""


def _env():
    e = dict(os.environ)
    e["HOS_ROOT"] = ROOT
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    e["MASTER_ADDR"] = "127.0.0.1"
    return e


def test_rccl_single_rank_allreduce_and_step():
    r = subprocess.run([sys.executable, "-c", _RCCL_SELF_TEST], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK nccl" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("shard", [False, True], ids=["replicated_decoder", "sharded_decoder"])
def test_bench_two_ranks_share_one_gpu_strong_scaling(shard):
    e = _env()
    e["HOS_BENCH_ONE_GPU"] = "1"
    if shard:
        e["HOS_SHARD_DECODER"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29542" if shard else "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--only-primary", "--no-kernel-events"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_rays"] == 4096 and d["config"]["rays_per_gpu"] == 2048
    assert d["value"] > 0 and d["steps"] == 3 and "stage-3" in d["config"]["workload"]
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 1.0
    if shard:
        # round 5: the sharded decoder's collectives go through torch.distributed here (gloo), i.e. they sit inside the forward and
        # nothing is captured -- the bench must say so and must not try (a capture invalidated by a collective used to leave a
        # sticky runtime error that the next launch reported: hos_clear_last_error)
        assert d["launch"].startswith("eager (volume decoder sharded"), d["launch"]
    else:
        # the step was captured; the collectives stay outside the graphs (default: enqueued asynchronously, the decoder's backward
        # replays under them; HOS_BENCH_OVERLAP=0: sequential)
        assert d["launch"].startswith("hipGraph replay (fwd+bwd)") and "all-reduce" in d["launch"], d["launch"]


def test_bench_default_command_two_ranks_rehearsal():
    """The command the driver runs at N > 1 -- `bench.py --gpus N --steps K --warmup W`, nothing skipped -- rehearsed as two processes
    on the one GPU (gloo stands in for RCCL): the ONE JSON line must carry the proof of who took part (`comm`), the strong-scaling
    primary, the weak-scaling stage-3 leg (4096 rays PER rank, the reference's DDP semantic, 3rd_Complete_HOSNeRF/run.py:173-190),
    the sharded-decoder leg, stage 1 and the 1080p frame -- every secondary leg inside its guard."""
    e = _env()
    e["HOS_BENCH_ONE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rays_per_gpu"] == 2048
    c = d["comm"]
    assert c["rccl_ranks_seen"] == 2 and len(c["devices"]) == 2 and c["devices"][0].startswith("rank 0:") and c["devices"][1].startswith("rank 1:")
    assert "gloo" in c["backend"]                       # said, not hidden: this rehearsal is not an RCCL run
    assert "parity_note" in d and "roofline" in d
    st = d["stages"]
    for k in ("stage1", "stage3_weak", "stage3_sharded_onegraph", "infer_1080p"):
        assert k in st and "error" not in st[k], (k, st.get(k))
    w = st["stage3_weak"]
    assert w["scaling"] == "weak" and w["rays_per_gpu"] == 4096 and w["global_rays"] == 8192 and w["value"] > 0
    sh = st["stage3_sharded_onegraph"]
    assert sh["rays_per_gpu"] == 2048 and sh["launch"].startswith("eager (volume decoder sharded") and 0 < sh["final_loss"] < 1.0
    assert st["infer_1080p"]["n_gpus"] == 2 and st["infer_1080p"]["finite"]
    out = os.path.join(ROOT, "gpurun_out", "bench_two_ranks_one_gpu_rehearsal.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write(lines[0] + "\n")


_FAILED_CAPTURE = r"""
import os, sys
sys.path.insert(0, os.environ["HOS_ROOT"])
import torch
from hosnerf_amd import _lib, ops
dev = torch.device("cuda")
x = torch.ones(1024, device=dev)
torch.cuda.synchronize()
s = torch.cuda.Stream()
failed = False
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            ops.zero_(x)
            float(x.sum())                     # a device-to-host read: not capturable, invalidates the capture
    except Exception as e:
        failed = True
assert failed
torch.cuda.synchronize()
first = ops.clear_last_error()
assert ops.clear_last_error() == 0             # cleared: a second call finds nothing
ops.zero_(x)                                   # on the default stream; reported the capture's error (901) before the clear existed
torch.cuda.synchronize()
assert float(x.abs().sum()) == 0.0
print("CLEAR_OK", first)
"""


def test_failed_capture_does_not_poison_the_next_launch():
    """An operation that cannot be captured invalidates a hipGraph capture and leaves a sticky runtime error; every entry point of the
    library reports launches through hipGetLastError(), so the NEXT launch used to fail with the capture's error.  A caller that
    recovers calls ops.clear_last_error() (hos_clear_last_error) once; after that launches succeed again.  (Own process: a failed
    capture leaves torch's capture bookkeeping of the process in a state later tests must not inherit.)"""
    e = _env()
    e["HOS_ROOT"] = ROOT
    r = subprocess.run([sys.executable, "-c", _FAILED_CAPTURE], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CLEAR_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("shard", [False, True], ids=["replicated_decoder", "sharded_decoder"])
def test_two_rank_step_equals_gradient_averaging(tmp_path, shard):
    """Data-parallel correctness of the human network's exchange (3.5 MB volume gradient + 4.3 MB of other parameters, decoder
    backward on the SUM): two ranks, each on its own item, take two optimiser steps; the parameters equal those of one process
    that accumulates the plain (un-split) backward of both items, halves the gradient and steps -- the definition of DDP.
    `sharded_decoder` (round 5): the same with the volume decoder's first three transposed convolutions SHARDED over the two ranks
    by input channel (`Network.shard_decoder`: partial forward products summed, input-gradient slices gathered, each rank's Adam
    on its rows only, the clip norm completed with the other rank's shard) -- the parameters after two clipped steps, gathered
    back from their owners, must be the replicated ones."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from tests import _dp_worker as W
    e = _env()
    e["HOS_BENCH_ONE_GPU"] = "1"
    e["HOS_DP_OUT"] = str(tmp_path / "dp.pt")
    if shard:
        e["HOS_SHARD_DECODER"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29548" if shard else "29547", os.path.join(ROOT, "tests", "_dp_worker.py")]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    got = torch.load(e["HOS_DP_OUT"])
    dev = torch.device("cuda")
    A = bench.Stage2(dev, 0, 2, W.RAYS)          # the network + rank 0's item
    B = bench.Stage2(dev, 1, 2, W.RAYS)          # only its item is used
    assert got["rays_local"] == A.rays_local == W.RAYS // 2
    from hosnerf_amd.train import stage2_losses
    p0 = A.net.store.param.detach().clone()
    sumsq0 = None
    for i in range(W.STEPS):
        A.opt.zero_grad()
        A.net.split_decoder_backward = False
        for rank, item in ((0, A.batch), (1, B.batch)):
            torch.manual_seed(W.seed_for(rank, i))
            loss, _ = stage2_losses(A.net(static_cycle=True, **item), item)
            loss.backward()                       # accumulates into the flat gradient
        A.net.store.ensure_bound()
        if i == 0:
            sumsq0 = float((A.net.store.grad.double() ** 2).sum())        # of the SUMMED gradient, like the ranks' clip kernels see it
        A.net.store.grad.mul_(0.5)
        A.opt.step(A.lr(i))
    want = A.net.store.param.detach().cpu()
    moved = float((want - p0.cpu()).abs().max())
    err = float((got["param"] - want).abs().max())
    assert moved > 1e-4, "the steps must have moved the parameters"
    # Adam normalises the update to ~lr per element, so the comparison is in units of the step, not of the parameter
    # (an element whose gradient is rounding noise gets a noise-sized update from Adam's normalisation: the worst element is
    # bounded loosely, the bulk tightly)
    frac_close = float(((got["param"] - want).abs() < 1e-3 * moved).float().mean())
    from tests._record import record
    record("dist.two_rank_step_vs_gradient_averaging[stage 2, 2 x 256 rays, 2 steps]" + ("[sharded decoder]" if shard else ""),
           {"max_param_step": moved, "max_abs_diff": err, "fraction_within_1e-3_of_a_step": frac_close})
    # the gradient norm both ranks clip by: every shard's share is in it (`train._shard_norm_correction`)
    assert got.get("sumsq_step0") is not None and abs(got["sumsq_step0"] - sumsq0) < 1e-3 * sumsq0, (got.get("sumsq_step0"), sumsq0)
    if not shard:
        assert err < 0.1 * moved, (err, moved)
        assert frac_close > 0.999, frac_close
        return
    # Sharded: the decoder's forward sums its input channels in another order (two partial products + an all-reduce), so the
    # volume differs from the replicated one at fp32 rounding (asserted below).  The DECODER's parameters must still come out
    # as if it were replicated; downstream of the volume the warp's x / max(sum w, 1e-4) and the 2^9-frequency Fourier features
    # amplify that rounding (tests/test_gpu_conditioning.py), and Adam turns a few-per-cent change of a 1e-7 gradient element
    # into a few per cent of a step: those parameters are bounded in units of the step, bulk and worst element.
    assert got["volume_rel_err"] < 2e-6, got["volume_rel_err"]
    off, n = A.net.decoder_span()
    d = (got["param"] - want).abs()
    assert float(d[off:off + n].max()) < 1e-2 * moved and float((d[off:off + n] < 1e-3 * moved).float().mean()) > 0.9995, \
        (float(d[off:off + n].max()), moved)
    assert err < 0.25 * moved and frac_close > 0.995, (err, moved, frac_close)
