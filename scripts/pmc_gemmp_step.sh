#!/bin/bash
# PMC passes (rounds 2 and 3) of the planes GEMM at the shape of the primary bench line (M = 131072 = 4096 rays x 32 samples, N = K = 1024)
# and at the stage-1 shape (M = 32768): HBM traffic (FETCH_SIZE, WRITE_SIZE) and MFMA busy cycles, separate passes, no tracing
# domains combined with --pmc.  Writes gpurun_out/pmc_gemmp/pmc_gemmp_traffic.json in the format bench.py reads
# (keyed by bench.py's kernel names, with the hash of the kernel sources the library was built from).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemmp
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for GM in 131072 32768; do
  export GM
  # only the three launches of a training step: forward with both output formats + ReLU bits, dgrad with the bit mask, wgrad
  export GONLY="fwd(bf16),dgrad(bits),wgrad"          # round 5: the forward is the one-format bf16 layer
  run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1_$GM -o t -- python $R/scripts/bench_gemmp.py 3 > /dev/null 2>&1; }
  run fetch "FETCH_SIZE"
  run write "WRITE_SIZE"
  run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
  GONLY= python $R/scripts/bench_gemmp.py 10 > $OUT/bench_$GM.txt 2>&1
done
python - <<PY
import csv, glob, json, hashlib, os
R = "$R"; OUT = "$OUT"
def mean(path, tag, counter=None):
    vals = []
    for f in glob.glob("%s/%s/*counter_collection.csv" % (OUT, path)):
        for r in csv.DictReader(open(f)):
            if "gemmp_kernel" in r["Kernel_Name"] and tag in r["Kernel_Name"] and (counter is None or r["Counter_Name"] == counter):
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None
h = hashlib.sha256()
for rel in ("hosnerf_amd/csrc/hos_gemmp.hip", "hosnerf_amd/csrc/hos_gemm_common.h", "Makefile"):
    h.update(open(os.path.join(R, rel), "rb").read())
out = {"source_hash": h.hexdigest()[:16], "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/bench_gemmp.py; "
       "hbm_bytes_per_launch = 2 x FETCH_SIZE (gfx950 wide-load correction, MI355X_MICROARCH.md) + WRITE_SIZE, counters in KB; "
       "mfma_busy_cycles = SQ_VALU_MFMA_BUSY_CYCLES, which on gfx950 counts 32 cycles per issued 32x32x16 MFMA (= 32 x insts_mfma by construction: "
       "an issue count, not an independent utilisation measurement); MFMA-busy fraction = mfma_busy_cycles / (4 SIMDs x 256 CUs) / (gui_active_cycles / 8 XCDs)", "kernels": {}}
for GM in (131072, 32768):
    # ELi1E = planes forward epilogue, ELi2E = dgrad, '3, ' = wgrad (transpose reads); the warm-up dispatches of each line are the same configuration
    for name, tag, key in (("gemmp_fwd", "ELi1EDF16b", "gemmp_fwd[M=%d,N=1024,K=1024]" % GM), ("gemmp_dgrad", "ELi2E", "gemmp_dgrad[M=%d,N=1024,K=1024]" % GM),
                           ("gemmp_wgrad", "3, ", "gemmp_wgrad[M=1024,N=1024,K=%d]" % GM)):
        f, w = mean("fetch_%d" % GM, tag), mean("write_%d" % GM, tag)
        if f is None or w is None:
            continue
        # fwd: A planes in, two plane formats out (+ 1 bit per element); dgrad: dZ in, dX out (+ bits in); wgrad: dZ and X in
        alg = {"gemmp_fwd": 4.0 * GM * 1024 * 2 + GM * 1024 / 8, "gemmp_dgrad": 4.0 * GM * 1024 * 2 + GM * 1024 / 8, "gemmp_wgrad": 4.0 * GM * 1024 * 2}[name]
        out["kernels"][key] = {"fetch_size_kb": f, "write_size_kb": w, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
                               "algorithmic_bytes_approx": alg,
                               "mfma_busy_cycles": mean("sq1_%d" % GM, tag, "SQ_VALU_MFMA_BUSY_CYCLES"), "insts_mfma": mean("sq1_%d" % GM, tag, "SQ_INSTS_MFMA"),
                               "gui_active_cycles": mean("sq1_%d" % GM, tag, "GRBM_GUI_ACTIVE")}
json.dump(out, open(OUT + "/pmc_gemmp_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat $OUT/bench_131072.txt | head -8
