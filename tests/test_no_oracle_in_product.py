"""The oracle is test infrastructure: nothing under hosnerf_amd/ may import or reference it, and the only
other importers allowed are tests/, __graft_entry__.smoke() and bench.py's baseline leg (`cpu_baseline`: the oracle timed as
the reference would run -- on the host cores, and as the same op graph on the GPU -- never as the product)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                yield os.path.join(base, f)


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for path in _py_files(os.path.join(ROOT, "hosnerf_amd")):
        src = open(path).read()
        assert not pat.search(src), f"{path} imports the oracle"
        assert "/root/reference" not in src, f"{path} reads the reference mount"


def test_bench_and_entry_use_oracle_only_as_checker():
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("import oracle") == 1 and "def cpu_baseline" in bench
    assert bench.index("import oracle") > bench.index("def cpu_baseline") and bench.index("import oracle") < bench.index("def main")
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert entry.index("import oracle") > entry.index("def smoke")
    for path in (os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")):
        assert "/root/reference" not in open(path).read()
