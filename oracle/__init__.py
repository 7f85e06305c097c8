"""CPU oracle for the HOSNeRF per-ray hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU (torch fp32, single process) restatement of the reference
algorithm for the path named in BASELINE.json:north_star.  It exists so that the
HIP kernels in ``hosnerf_amd/csrc`` can be checked on a GPU box where
``/root/reference`` does not exist.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it -- and there only as the checker / the timed CPU
    baseline, never as the thing measured or shipped;
  * nothing in ``hosnerf_amd/`` imports it; the product path raises when the
    HIP library is missing instead of falling back to this code.

Pinning: every function is checked against golden vectors exported from the
reference itself (imported in the build container with stub modules, see
``tests/golden/make_golden.py``) in ``tests/test_oracle_golden.py``.  The
reference ships no tests / fixtures of its own, so parity is
"pinned by vectors generated from the reference run here", not by reference
tests.

Floating point: the reference is a chain of torch fp32 ops, so the oracle uses
torch fp32 CPU ops as its arithmetic substrate (this also gives the gradient
oracle through autograd).  Where the reference uses a third-party fused op the
oracle restates the published algorithm explicitly (trilinear grid_sample,
closed-form contraction Jacobian, searchsorted-style inverse CDF).
"""
