"""Build-time check (no GPU): no instruction of the kernels that read LDS through inline-asm `ds_read` + counted `s_waitcnt`
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
    rep = S.scan(os.path.join(ROOT, "hosnerf_amd", "csrc", "hos_chain.hip"), ["chain128_kernel", "chain256_kernel"])
    assert len(rep) == 2, list(rep)
    for k, found in rep.items():
        assert not found, (k, found[:4])


def test_planes_gemm_main_loops_never_touch_an_inflight_fragment():
    """All eight instantiations; the scan covers the main loop (first to last MFMA): at the loop exit no read is outstanding."""
    import scan_inflight_reads as S
    rep = S.scan(os.path.join(ROOT, "hosnerf_amd", "csrc", "hos_gemmp.hip"), ["gemmp_kernel"], region="mfma")
    assert len(rep) == 8, list(rep)
    for k, found in rep.items():
        assert not found, (k, found[:4])
