"""ISA check for kernels that read LDS with inline-asm `ds_read` + explicit `s_waitcnt lgkmcnt(N)` (hos_chain.hip, hos_gemmp.hip).

hipcc believes such an asm statement has WRITTEN its output register when it is issued.  If anything -- typically a `v_mov`
the register allocator places at a control-flow merge -- touches that register before the matching `s_waitcnt`, it sees a
half-filled fragment.  That cost intermittent garbage in `chain128_kernel` (a few rows of one wave, a few times per 10^5 launches;
found by a soak run with NaN-poisoned allocations, scripts/soak_poison.py).  The scan walks the kernel's instructions in text
order (so it also sees hazards that cross basic blocks on the fall-through path) and reports every instruction that names a
register of a still-outstanding ds_read.

  python scripts/scan_inflight_reads.py hosnerf_amd/csrc/hos_chain.hip chain128_kernel
Exit status 1 if a hazard is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_kernel(lines, start, end, stop_at_loop_exit=False):
    pending, found = [], []
    for i in range(start, end):
        t = lines[i].strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
        if op.startswith("ds_read"):
            pending.append((regs(toks[0]) if toks else set(), i))
            continue
        if op == "s_waitcnt" and "lgkmcnt" in t:
            k = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
            pending = pending[len(pending) - k:] if k else []
            continue
        if op in ("s_endpgm",):
            pending = []
            continue
        used = set()
        for tk in toks:
            used |= regs(tk)
        for dst, li in pending:
            if used & dst:
                found.append((i - start, t[:80], lines[li].strip()[:60]))
    return found


def assemble(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",          # the Makefile's flags: the scan must see the shipped allocation
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "hosnerf_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def scan(src, kernels, region="kernel"):
    """region 'kernel': the whole kernel in text order; 'mfma': only between the first and the last MFMA (the main loop -- a
    kernel whose reads are never outstanding at the loop exit)."""
    lines = [l.rstrip() for l in open(assemble(src))]
    report = {}
    for name in kernels:
        starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l)]
        for st in starts:
            end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
            a, b = st, end
            if region == "mfma":
                mf = [i for i in range(st, end) if "v_mfma" in lines[i]]
                a, b = mf[0], mf[-1]
            report[lines[st][:-1]] = scan_kernel(lines, a, b)
    return report


if __name__ == "__main__":
    src, kernels = sys.argv[1], sys.argv[2:]
    bad = 0
    for k, found in scan(os.path.join(ROOT, src) if not os.path.isabs(src) else src, kernels).items():
        print(k[:100], "->", len(found), "instructions touch a register of an outstanding asm ds_read")
        for f in found[:8]:
            print("   +%d  %s   | outstanding: %s" % f)
        bad += len(found)
    sys.exit(1 if bad else 0)


def scan_untracked_global_loads(src, kernels):
    """destinations of INLINE-ASM `global_load_dword` (hos_gemmp.hip: the epilogue operands bias_r[] / bw[][] requested
    before the K loop, `"=v"` outputs the compiler believes are defined when the asm statement ends).  Between such a load and the
    first `s_waitcnt vmcnt(0)` behind it NO instruction may name its destination VGPR -- neither a read (stale value) nor a write
    or copy (the late-arriving data would land in a re-used register).  Loads the compiler emitted itself are tracked by its own
    waits and are not looked at (inline asm is bracketed by `;;#ASMSTART` / `;;#ASMEND` in the -S output).
    Returns {kernel: (number of asm loads seen, [hazards])}."""
    lines = [l.rstrip() for l in open(assemble(src))]
    report = {}
    for name in kernels:
        for st in [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l)]:
            end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
            pending, found, seen, in_asm = [], [], 0, False
            for i in range(st, end):
                t = lines[i].strip()
                if t.startswith(";;#ASMSTART"):
                    in_asm = True
                    continue
                if t.startswith(";;#ASMEND"):
                    in_asm = False
                    continue
                if not t or t[0] in ";." or t.endswith(":"):
                    continue
                op = t.split()[0]
                toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
                if in_asm and op == "global_load_dword" and toks:
                    pending.append((regs(toks[0]), i))
                    seen += 1
                    used = set()
                    for tk in toks[1:]:
                        used |= regs(tk)
                elif op == "s_waitcnt" and re.search(r"vmcnt\(0\)", t):
                    pending = []
                    continue
                else:
                    used = set()
                    for tk in toks:
                        used |= regs(tk)
                for dst, li in pending:
                    if li != i and used & dst:
                        found.append((i - st, t[:80], lines[li].strip()[:60]))
            report[lines[st][:-1]] = (seen, found)
    return report
