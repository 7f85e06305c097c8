"""Build-time check (no GPU): the device ISA of every translation unit, compiled with the Makefile's flags, contains no packed-FP32
VALU instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32).  Why: DESIGN.md section 6 "packed FP32 under MFMA
co-residency" -- on gfx950 / ROCm 7.2 such instructions were measured to produce wrong values in lanes 48-63 of a wave when waves of
another kernel that issue MFMAs share its SIMD (kernels of two HIP streams running concurrently).
  python scripts/check_packed_fp32.py            # prints {file: count}, exit status 1 if any count is non-zero"""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACKED = re.compile(r"\bv_pk_(mul|add|fma)_f32\b|\bv_pk_mov_b32\b")


def makefile_flags():
    with open(os.path.join(ROOT, "Makefile")) as f:
        for line in f:
            if line.startswith("FLAGS :="):
                return line.split(":=", 1)[1].replace("$(ARCH)", "gfx950").split()
    raise RuntimeError("FLAGS not found in the Makefile")


def count(src, flags):
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    r = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", src, "-o", "-"], capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    return len(PACKED.findall(r.stdout)), len(re.findall(r"\bv_(mul|add|fma)_f32", r.stdout))


def scan():
    flags = makefile_flags()
    srcs = sorted(glob.glob(os.path.join(ROOT, "hosnerf_amd", "csrc", "*.hip")))
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda s: count(s, flags), srcs))
    return {os.path.basename(s): r for s, r in zip(srcs, res)}


if __name__ == "__main__":
    rep = scan()
    for k, (p, s) in rep.items():
        print(f"{k:22s} packed fp32 instructions {p:5d}   scalar-form fp32 mul/add/fma {s}")
    sys.exit(1 if any(p for p, _ in rep.values()) else 0)
