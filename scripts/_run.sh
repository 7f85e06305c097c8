export GM=131072
python scripts/bench_gemmp.py 10 2>&1 | grep "^planes\|err\|diff"
GM=1000 GN=96 GK=64 python scripts/bench_gemmp.py 10 2>&1 | grep "err\|diff"
python -m pytest tests -x -q -m gpu -k "gemm or plane or mip or stage1 or golden or bkgd or stress" 2>&1 | tail -3
