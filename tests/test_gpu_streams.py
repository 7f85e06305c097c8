"""Kernels of the two branches of a stage-3 step run concurrently on two HIP streams (HOSNeRF.two_streams).  That is only
admissible if every kernel computes the same bits no matter what runs next to it:
  * the IPE encoder next to chain128_kernel -- the pair that exposed the packed-FP32 problem (28 of 30 runs differed before the
    library was built without packed-FP32 instructions, see Makefile / DESIGN section 6);
  * a whole PropMLP query (encoder -> planes GEMMs -> row dot) next to the human network's training-mode forward;
  * the captured two-stream step replayed without host synchronisation: every replay reproduces the same loss and gradient
    norms (scripts/soak_streams.py), in the one-stream order too."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + list(args), capture_output=True, text=True, timeout=1200, cwd=ROOT, env=e)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_encoder_is_bit_stable_next_to_the_chain_kernel():
    out = _run("stress_victims.py", "24")
    lines = [l for l in out.splitlines() if l.startswith("victim")]
    assert len(lines) >= 2, out
    for l in lines:
        assert "quiet 0/24" in l and "side stream 0/24" in l, l


def test_prop_mlp_query_is_bit_stable_next_to_the_human_forward():
    out = _run("stress_concurrent.py", "16", "none,chain_save,canonical_save,warp,lbs", env={"LEVEL": "1"})
    lines = [l for l in out.splitlines() if l.startswith("disturber")]
    assert len(lines) == 5, out
    assert all("all bit-identical" in l for l in lines), lines


@pytest.mark.parametrize("two", ["1", "0"])
def test_captured_step_replays_are_deterministic(two):
    out = _run("soak_streams.py", "120", "1024", env={"HOS_TWO_STREAMS": two})
    assert "ok: 120 replays" in out and f"two_streams = {two == '1'}" in out, out[-1500:]
