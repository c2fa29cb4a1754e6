#!/usr/bin/env python
"""Which torch (aten) kernels the training legs still launch on device tensors, and from where: a TorchDispatchMode counts every aten op with a
CUDA tensor argument by (op, innermost call site inside this package) over whole micro-batches of bench.py's leg (after its warm-up).
    python tools/torch_op_census.py --leg train2 [--micro-batches 2]  ->  the 60 most frequent (op, call site) pairs per micro-batch
The package's own kernels go through ctypes and do not show up here: this lists what is LEFT on torch (fills, copies, casts, adds, cats ...)."""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", default="train2", choices=["train", "train2", "distill"])
    ap.add_argument("--micro-batches", type=int, default=2)
    a = ap.parse_args()
    args = argparse.Namespace(batch=4, no_ffn_lora=False, distill_only=a.leg == "distill", no_train_graphs=False, reference_pass_structure=False)
    dev = torch.device("cuda:0")
    tr, batches, step_kw, B, n_train, ldm, teacher, id2ada, text_enc = bench.build_train(args, (1, 0, 0, False), dev, 2 if a.leg == "train2" else 1)
    for i in range(8):
        tr.training_step(batches[i % 4], i, **step_kw)
    torch.cuda.synchronize()
    counts = collections.Counter()

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            flat = list(args) + list((kwargs or {}).values())
            if any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat) or "empty" in str(func) or "zeros" in str(func) or "full" in str(func):
                where = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "adaface-dev_amd" in f.filename or "adaface_dev_amd" in f.filename]
                counts[(str(func), where[-1] if where else "?")] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        for i in range(a.micro_batches):
            tr.training_step(batches[i % 4], 8 + i, **step_kw)
    torch.cuda.synchronize()
    n = a.micro_batches
    by_op = collections.Counter()
    for (op, _), c in counts.items():
        by_op[op] += c
    print(f"# {a.leg}: aten ops on device tensors per micro-batch (mean of {n}); views / metadata ops launch nothing")
    for op, c in by_op.most_common(25):
        print(f"{c / n:9.1f}  {op}")
    print("# by call site")
    for (op, where), c in counts.most_common(70):
        print(f"{c / n:9.1f}  {op:48s} {where}")


if __name__ == "__main__":
    main()
