"""Who launches the small torch-native kernels of one eager stage-2 / stage-3 step (TorchDispatchMode + Python stacks)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
dev = torch.device("cuda")
w = (bench.Stage3 if len(sys.argv) > 1 and sys.argv[1] == "3" else bench.Stage2)(dev, 0, 1, 2048)
for i in range(2):
    w.host_prepare(i); w.fwd_bwd(i); w.finish(i, False)
torch.cuda.synchronize()
cnt = collections.Counter()
SKIP = ("view", "reshape", "as_strided", "detach", "slice", "select", "expand", "permute", "transpose", "t.default", "unsqueeze", "squeeze", "alias", "empty", "_unsafe_view", "unbind", "split", "is_", "sym_", "stride", "size", "numel", "lift", "_local_scalar")
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            site = "(no python frame of ours: autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if ("hosnerf_amd/" in fr.filename or fr.filename.endswith("bench.py")) and not fr.filename.endswith("_lib.py"):
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:70]}"
                    break
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    w.host_prepare(2); w.fwd_bwd(2); w.finish(2, False)
torch.cuda.synchronize()
for (n, s), c in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{c:4d}  {n:28s} {s}")
print("total", sum(cnt.values()))
