"""Build-time checks of the generated ISA (no GPU).
(1) no packed-FP32 VALU instruction in any kernel (Makefile: -packed-fp32-ops; scripts/check_packed_fp32.py) -- measured on gfx950:
such instructions go wrong in lanes 48-63 when MFMA waves of another kernel share the SIMD (two streams);
(2) no instruction of the kernels that read LDS through inline-asm `ds_read` + counted `s_waitcnt`
touches a fragment register that such a read is still filling (scripts/scan_inflight_reads.py; the bug class behind the
intermittent garbage of `chain128_kernel` found in round 2)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")


def test_chain_kernels_never_touch_an_inflight_fragment():
    import scan_inflight_reads as S
    rep = S.scan(os.path.join(ROOT, "hosnerf_amd", "csrc", "hos_chain.hip"), ["chain128_kernel"])
    assert len(rep) == 2, list(rep)              # chain128, folded and not
    for k, found in rep.items():
        assert not found, (k, found[:4])


def test_planes_gemm_main_loops_never_touch_an_inflight_fragment():
    """All twelve instantiations of the default build (round 5: + bf16-operand FWD; the persistent FWD / DGRAD forms are compiled out
    by default, -DHOS_GEMMP_PERSIST=1 adds three); the scan covers the main loop (first to last MFMA)."""
    import scan_inflight_reads as S
    rep = S.scan(os.path.join(ROOT, "hosnerf_amd", "csrc", "hos_gemmp.hip"), ["gemmp_kernel"], region="mfma")
    assert len(rep) == 12, list(rep)
    for k, found in rep.items():
        assert not found, (k, found[:4])


def test_planes_gemm_untracked_epilogue_loads_are_never_touched_before_their_wait():
    """the FWD bias / DGRAD bit-mask operands of the planes GEMM are requested by inline-asm `global_load_dword` before the
    K loop; nothing may name their destination VGPRs until the first vmcnt(0) (re-run on every ROCm bump: it checks the
    register allocation of THIS compiler)."""
    import scan_inflight_reads as S
    rep = S.scan_untracked_global_loads(os.path.join(ROOT, "hosnerf_amd", "csrc", "hos_gemmp.hip"), ["gemmp_kernel"])
    assert len(rep) == 12, list(rep)
    assert sum(n for n, _ in rep.values()) >= 12, "the scan did not see the asm loads (ASMSTART markers / mnemonic changed?)"
    for k, (n, found) in rep.items():
        assert not found, (k, found[:4])


def test_no_packed_fp32_instruction_in_any_kernel():
    """The two-stream step (HOSNeRF.two_streams) lets kernels of the two branches share CUs.  v_pk_{mul,add,fma}_f32 / v_pk_mov_b32
    in a wave that shares its SIMD with MFMA-issuing waves of another kernel gave wrong results in lanes 48-63 (28 of 30 runs
    of the IPE encoder next to chain128_kernel; 0 of 30 without the packed forms; scripts/stress_victims.py)."""
    import check_packed_fp32 as C
    rep = C.scan()
    assert len(rep) >= 18
    assert all(p == 0 for p, _ in rep.values()), {k: v for k, v in rep.items() if v[0]}
    assert sum(s for _, s in rep.values()) > 10000          # the scalar forms are there: the scan looked at real ISA
