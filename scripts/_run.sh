export GM=131072
for v in base nopf base nopf; do if [ $v = base ]; then unset HOS_LIB_PATH; else export HOS_LIB_PATH=build/variants/$v/libhosrender.so; fi; echo "== $v"; GONLY="(warm-up line),fwd(2fmt),fwd(f16),dgrad(bits)" python scripts/bench_gemmp.py 20 2>&1 | grep "^planes"; done
GM=32768 python scripts/bench_gemmp.py 20 2>&1 | grep "^planes\|err\|diff"
