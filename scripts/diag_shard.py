"""Where do the parameters of the sharded-decoder data-parallel step differ from gradient averaging?  (diagnostic of
tests/test_gpu_dist.py::test_two_rank_step_equals_gradient_averaging[sharded_decoder])"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tests import _dp_worker as W
from hosnerf_amd.train import stage2_losses
e = dict(os.environ); e.update(HOS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", HOS_BENCH_ONE_GPU="1", HOS_SHARD_DECODER=os.environ.get("SHARD", "1"))
out = os.path.join(tempfile.mkdtemp(), "dp.pt"); e["HOS_DP_OUT"] = out
r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29549",
                    os.path.join(ROOT, "tests", "_dp_worker.py")], env=e, capture_output=True, text=True, cwd=ROOT)
assert r.returncode == 0, r.stderr[-3000:]
got = torch.load(out)["param"]
dev = torch.device("cuda")
A = bench.Stage2(dev, 0, 2, W.RAYS); B = bench.Stage2(dev, 1, 2, W.RAYS)
p0 = A.net.store.param.detach().clone()
for i in range(W.STEPS):
    A.opt.zero_grad(); A.net.split_decoder_backward = False
    for rank, item in ((0, A.batch), (1, B.batch)):
        torch.manual_seed(W.seed_for(rank, i))
        loss, _ = stage2_losses(A.net(static_cycle=True, **item), item); loss.backward()
    A.net.store.ensure_bound(); A.net.store.grad.mul_(0.5); A.opt.step(A.lr(i))
want = A.net.store.param.detach().cpu(); g = A.net.store.grad.detach().cpu(); v = A.opt.exp_avg_sq.detach().cpu()
moved = float((want - p0.cpu()).abs().max())
diff = (got - want).abs()
print("moved", moved, "max diff", float(diff.max()), "frac > 1e-3 step", float((diff > 1e-3 * moved).float().mean()))
base = A.net.flat_param.data_ptr()
for name, p in A.net.named_parameters():
    off = (p.data_ptr() - base) // 4; n = p.numel()
    d = diff[off:off + n]
    bad = d > 1e-3 * moved
    if bad.any():
        gg = g[off:off + n].abs()
        print(f"{name:60s} n={n:9d} bad={int(bad.sum()):8d} max={float(d.max()):.2e}  |g| at bad: median {float(gg[bad].median()):.2e} max {float(gg[bad].max()):.2e}; |g| overall median {float(gg.median()):.2e}")
if A.net._w0c is not None:
    off, n = A.net._w0c.offset, A.net._w0c.numel
    d = diff[off:off + n]; bad = d > 1e-3 * moved; gg = g[off:off + n].abs()
    print(f"{'(compact first deconv)':60s} n={n:9d} bad={int(bad.sum()):8d} max={float(d.max()):.2e}  |g| at bad: median {float(gg[bad].median()) if bad.any() else 0:.2e}; |g| overall median {float(gg.median()):.2e}")
    rows = bad.view(-1, n // 1024 if False else A.net._deconv_chans[0][1] * 8).any(1)
    print("   rows of the compact weight with a bad element:", int(rows.sum()), "of", rows.numel(), "first", rows.nonzero().flatten()[:10].tolist())
