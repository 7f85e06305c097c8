"""Are the relaxed full-size bounds reference self-noise?  (VERDICT r2 "next round" item 2.)

The full-size parity tests (tests/test_gpu_speedup.py) assert the north-star tolerance -- 1e-4 RGB L-inf -- on the rays whose
DISCRETE decisions agree with the oracle's, and count / bound the others: a sample that jumps across an empty bin of a proposal
histogram when the CDF moves by an ulp (stage 1, inverse-CDF resampling, H:343-399), a background sample and a human sample
that coincide and swap places in the z-merge (stage 3, M:1547-1585).  Here the REFERENCE's arithmetic is evaluated three ways
on the same rays, weights and draws -- the oracle in fp32 on the host cores (MKL sums), in fp32 on the MI355X (rocBLAS sums)
and in float64 -- and the same statistics are taken between those: if two fp32 evaluations of the reference's own op graph
disagree on as many rays, by as much, as the HIP path does, the relaxed bounds measure the reference, not this implementation.
The HIP path's counts are asserted against a small multiple of the oracle-vs-oracle counts (the `e_ref` pattern of the
gradient tests); the figures go to gpurun_out/parity_counts.json -> profiles/r03_parity_counts.json."""
import pytest
import torch

from hosnerf_amd import synth
from tests import _parity as par
from tests._record import record

pytestmark = pytest.mark.gpu


def test_stage1_fullsize_bounds_are_reference_self_noise():
    dev = torch.device("cuda")
    B = 1024
    batch = synth.stage1_batch(B, seed=777)
    sd = synth.background_state_dict(777, 2)
    g = torch.Generator().manual_seed(11)
    jit = [torch.rand(B, generator=g) for _ in range(3)]               # the same fp32 draws for every evaluation
    pairs = par.stage1_tables(sd, batch, jit, dev, train_frac=0.5)
    record("selfnoise.stage1[1024 rays x 64/64/32 samples, jitter seed 11]", pairs)
    # measured (profiles/r03_parity_counts.json): oracle fp32-CPU vs fp32-ROCm move a sample on 58 of the 1024 rays, each of
    # them vs float64 on 179 (worst RGB on such rays 9.4e-5); the HIP path: 69 / 90 rays vs the two fp32 evaluations, 206 vs
    # float64, 1 ray over 1e-4 (1.2-1.6e-4)
    noise_moved = par.assert_stage1(pairs)
    # and the reference's own two fp32 evaluations are NOT within the tolerance of each other on every ray -- the premise
    assert noise_moved > 0


def test_stage3_fullsize_bounds_are_reference_self_noise():
    dev = torch.device("cuda")
    B = 2048
    b = synth.add_patch_supervision(synth.human_batch(B, seed=778, time=0.5, is_train=True, iter_val=3e5), 2, 32, 778)
    g = torch.Generator().manual_seed(5)
    t_rand = torch.rand(B, 128, generator=g)
    jit = [torch.rand(B, generator=g) for _ in range(3)]
    bsd, hsd = synth.background_state_dict(777, 2), synth.human_state_dict(777, 2)
    pairs = par.stage3_tables(bsd, hsd, b, t_rand, jit, dev)
    record("selfnoise.stage3[2048 rays, 160 merged samples, seed 778 / draws 5]", pairs)
    # measured (profiles/r03_parity_counts.json): the reference's two fp32 evaluations swap a coinciding pair on 10 of the 2048
    # rays (worst 1.8e-3, 3 rays over 1e-4) and each is 19-21 swapped rays / 1.0e-4 on same-order rays away from float64; the
    # HIP path: 2 and 8 swapped rays against the two fp32 evaluations, 19 against float64
    noise_sw = par.assert_stage3(pairs)
    assert noise_sw > 0, "premise: two fp32 evaluations of the reference's own graph order some coinciding pair differently"
