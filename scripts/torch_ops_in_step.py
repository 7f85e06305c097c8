"""Which torch (aten) kernels does a stage-3 step still launch, and from which source line?  A TorchDispatchMode records every
aten call of one eager step (forward on this thread, backward on the autograd thread) with the innermost hosnerf_amd / bench frame.
  python scripts/torch_ops_in_step.py [rays] [stage3|stage2|stage1]"""
import collections, os, sys, threading, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from hosnerf_amd import ops

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STAGE = sys.argv[2] if len(sys.argv) > 2 else "stage3"
dev = torch.device("cuda")
ops.set_gemm_mode(ops.GEMM_PLANES)
w = {"stage3": bench.Stage3, "stage2": bench.Stage2, "stage1": bench.Stage1}[STAGE](dev, 0, 1, rays)
if STAGE == "stage3":
    w.hos.two_streams = False
for i in range(3):
    w.eager_step(i)
torch.cuda.synchronize()
SKIP = {"aten::view", "aten::_unsafe_view", "aten::reshape", "aten::detach", "aten::alias", "aten::as_strided", "aten::slice", "aten::select",
        "aten::t", "aten::transpose", "aten::permute", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::empty", "aten::empty_like",
        "aten::empty_strided", "aten::is_pinned", "aten::_local_scalar_dense", "aten::unbind", "aten::split", "aten::narrow", "aten::view_as",
        "aten::lift_fresh", "aten::_to_copy", "aten::resize_", "aten::set_", "aten::record_stream", "aten::is_same_size", "aten::new_empty"}
agg = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        out = func(*args, **(kwargs or {}))
        if name not in SKIP:
            on_gpu = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or (isinstance(out, torch.Tensor) and out.is_cuda)
            if on_gpu:
                site = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if ("hosnerf_amd/" in fr.filename or fr.filename.endswith("bench.py")) and "_lib.py" not in fr.filename:
                        site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                        break
                agg[(name, site)] += 1
        return out


# the autograd engine runs backward nodes on its own thread: modes are thread-local, so run the backward in THIS thread
torch.autograd.set_multithreading_enabled(False)
with Rec():
    w.eager_step(3)
torch.cuda.synchronize()
print(f"device aten calls in one {rays}-ray {STAGE} step (views / allocations not counted): {sum(agg.values())}")
for (name, site), n in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d} x {name:26s} {site}")
