export GM=131072
for v in base noph base noph; do if [ $v = base ]; then unset HOS_LIB_PATH; else export HOS_LIB_PATH=build/variants/$v/libhosrender.so; fi; echo "== $v"; GONLY="(warm-up line),fwd(2fmt),fwd(f16),dgrad(bits),wgrad" python scripts/bench_gemmp.py 20 2>&1 | grep "^planes"; done
GM=32768 python scripts/bench_gemmp.py 20 2>&1 | grep "err\|diff"
GM=1000 GN=96 GK=64 python scripts/bench_gemmp.py 5 2>&1 | grep "err\|diff"
