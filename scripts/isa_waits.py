"""List the s_waitcnt vmcnt / s_barrier / MFMA / LDS-DMA structure of every gemmp_kernel instantiation (hipcc -S of hos_gemmp.hip).
Run on the build host; no GPU needed.  Used to check that hipcc put NO vmcnt wait of its own inside the K loop."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "hosnerf_amd/csrc/hos_gemmp.hip")
out = "/tmp/isa_waits.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
                "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", f"-I{root}/include", f"-I{root}/hosnerf_amd/csrc",
                "-Wno-unused-result", "-S", "--cuda-device-only", src, "-o", out] + sys.argv[2:], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN12_GLOBAL__N_112gemmp_kernel.*:", l)]
for a in starts:
    name = lines[a].split(":")[0]
    end = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    body = lines[a:end]
    ev = []
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            asm = body[i - 1].strip().startswith(";;#ASMSTART")
            ev.append(f"{i}:{'asm' if asm else 'HIPCC'}:{t.replace('s_waitcnt ', '')}")
        elif t.startswith("s_barrier"):
            ev.append(f"{i}:barrier")
    n_mfma = sum("v_mfma" in l for l in body)
    n_dma = sum(("global_load_lds" in l) or ("buffer_load" in l and " lds" in l) for l in body)
    print(name[len("_ZN12_GLOBAL__N_112gemmp_kernel"):-len("EvNS_5PArgsE")], "lines", len(body), "mfma", n_mfma, "dma", n_dma)
    print("   ", " ".join(ev[:40]), "..." if len(ev) > 40 else "")
