"""Which call sites of a Stage-2 micro-batch reach a vendor GEMM (aten mm / addmm / bmm / convolution on device tensors)."""
import argparse, os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

seen = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("aten.mm", "aten.addmm", "aten.bmm", "aten.baddbmm", "aten.convolution", "aten.linear", "aten.matmul", "aten.mv", "aten.addmv", "aten._scaled")):
            dev = [a.device.type for a in args if isinstance(a, torch.Tensor)]
            if "cuda" in dev:
                st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1] if "adaface" in f.filename or "bench.py" in f.filename]
                seen[(name, tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), " < ".join(st[-4:]))] += 1
        return func(*args, **(kwargs or {}))


import bench
ns = argparse.Namespace(batch=4, no_ffn_lora=False, no_train_graphs=True, train_steps=2, train_warmup=0, no_roofline=True, distill_only=False)
with Spy():
    bench.run_train(ns, (1, 0, 0, False), torch.device("cuda:0"), stage=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
for (name, shapes, st), n in seen.most_common(40):
    print(n, name, shapes, "|", st)
