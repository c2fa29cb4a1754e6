"""tools/e2e_infer.py (BASELINE configs[1] end to end, 10 DDIM steps) under the dispatch spy of r03ap: vendor GEMM / convolution call sites of the inference path."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
seen = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("aten.mm", "aten.addmm", "aten.bmm", "aten.baddbmm", "aten.convolution", "aten.mv", "aten.addmv", "aten._scaled_mm")):
            if any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
                st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1] if "adaface" in f.filename]
                seen[(name, tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), " < ".join(st[-4:]))] += 1
        return func(*args, **(kwargs or {}))


sys.argv = ["e2e_infer.py", "10"]
import e2e_infer
with Spy():
    e2e_infer.main()
for (name, shapes, st), n in seen.most_common(40):
    print(n, name, shapes, "|", st)
print("vendor call sites:", len(seen))
