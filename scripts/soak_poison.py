"""Soak run with NaN-poisoned allocations: every float `torch.empty` / `empty_like` made by the product code is filled with NaN,
so a kernel that reads memory nobody wrote -- or garbage a racing kernel produced -- turns a gradient non-finite at once instead
of perturbing training silently.  Per step (with a host synchronisation) the flat gradient is checked per module; the outputs of
the chain kernel are checked right after each launch.

  [SOAK_GEMM=planes|split|fp32] python scripts/soak_poison.py [1|2|3] [steps] [seed] [rays]      # stage, default 2 / 400 / 0 / 2048
This is how the intermittent garbage of `chain128_kernel` (hipcc copying a fragment register an asm ds_read was still filling) was
found and its fix verified: 9-10 events per 14 runs of 400 steps before, 0 of 12 after."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def _ours():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "hosnerf_amd/" in fr.filename and "_lib.py" not in fr.filename:
            return True
    return False


def pempty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point() and _ours():
        t.fill_(float("nan"))
    return t


def pempty_like(x, **k):
    t = _empty_like(x, **k)
    if t.is_cuda and t.is_floating_point() and _ours():
        t.fill_(float("nan"))
    return t


def run(stage=2, steps=400, seed=0, rays=2048, verbose=True):
    torch.empty, torch.empty_like = pempty, pempty_like
    try:
        import bench
        from hosnerf_amd import ops
        dev = torch.device("cuda")
        ops.set_gemm_mode({"planes": ops.GEMM_PLANES, "split": ops.GEMM_BF16X3, "fp32": ops.GEMM_FP32}[os.environ.get("SOAK_GEMM", "planes")])
        w = {1: bench.Stage1, 2: bench.Stage2, 3: bench.Stage3}[stage](dev, 0, 1, rays)
        torch.manual_seed(seed)
        state = {}
        chain = ops.mlp_chain128_fwd

        def chain_checked(E, PE, x, planes, aux, acts, xyz, rows_dev=None):
            chain(E, PE, x, planes, aux, acts, xyz, rows_dev=rows_dev)
            n = x.shape[0] if rows_dev is None else int(rows_dev)
            for l, a in enumerate(list(acts) + [xyz]):
                nf = ~torch.isfinite(a[:n])
                if nf.any() and "chain" not in state:
                    rows = nf.any(1).nonzero().flatten()
                    state["chain"] = f"chain output {l} (6 = xyz): {rows.numel()} bad rows of {n}, first {int(rows[0])}, last {int(rows[-1])}"
        ops.mlp_chain128_fwd = chain_checked
        mods = [o.module for o in w.opts()]
        spans = []
        for m in mods:
            d = collections.OrderedDict()
            for name, p in m.named_parameters():
                key = ".".join(name.split(".")[:2])
                off = (p.data_ptr() - m.store.param.data_ptr()) // 4
                lo, hi = d.get(key, (off, off + p.numel()))
                d[key] = (min(lo, off), max(hi, off + p.numel()))
            spans.append(d)
        orig_finish = w.finish

        def finish(i, dynamic):
            for m, d in zip(mods, spans):
                g = m.store.grad
                bad = [k for k, (a, b) in d.items() if not bool(torch.isfinite(g[a:b]).all())]
                if bad and "first" not in state:
                    state["first"] = (i, bad)
            orig_finish(i, dynamic)
        w.finish = finish
        try:
            for i in range(steps):
                w.host_prepare(i)
                loss = w.eager_step(i)
                if "first" in state or "chain" in state or not bool(torch.isfinite(loss)):
                    msg = f"EVENT at step {i}: loss {float(loss)}, non-finite gradient spans {state.get('first')}, chain check: {state.get('chain')}"
                    if verbose:
                        print(msg)
                    return msg
        finally:
            ops.mlp_chain128_fwd = chain
        if verbose:
            print(f"no event in {steps} steps (stage {stage}, seed {seed}); final loss {float(loss):.6f}")
        return None
    finally:
        torch.empty, torch.empty_like = _empty, _empty_like


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(1 if run(int(a[0]) if a else 2, int(a[1]) if len(a) > 1 else 400, int(a[2]) if len(a) > 2 else 0,
                      int(a[3]) if len(a) > 3 else 2048) else 0)
