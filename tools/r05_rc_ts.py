"""GPU box: per-phase wave cycles of coupling_rqs_bwd_recompute_kernel (library built with -DBGK_RC_TS=1 for bgk_fused2_train.hip: lane 0
of every wave writes its phase sums over g_y[b0][0..7]): z1 wait | act + split | first GEMM | chunks -> LDS | VJP | next GEMMs |
gradient stores | whole tile."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import dense                                     # noqa: E402
from tests.test_gpu_round5 import _fields, _spline_layer         # noqa: E402

dev = torch.device("cuda:0")
B = 1 << 18
layer = _spline_layer(dev, what="TORSIONS", on="FIXED")
stash = []
orig = dense._rqs_backward_recompute


def wrap(*a):
    r = orig(*a)
    stash.append(r[0])
    return r


dense._rqs_backward_recompute = wrap
for it in range(3):
    stash.clear()
    xs = _fields(dev, B)
    *out, dl = layer(*xs)
    (sum((o * o).sum() for o in out) - dl.sum()).backward()
torch.cuda.synchronize()
st = stash[0].view(torch.int32)[0::32, :8].cpu().numpy().astype(np.int64) & 0xffffffff
names = ["z1 tile wait", "activation + f16 split", "first GEMM (chunk 0)", "chunks -> LDS (+ ring start)", "VJP (10 slots)", "next chunks' GEMMs", "gradient stores", "whole tile"]
ok = (st < 1 << 24).all(axis=1)
st = st[ok]
print(f"{ok.sum()} tiles")
for k, nm in enumerate(names):
    print(f"  {nm:30s} median {np.median(st[:, k]):8.0f}   p90 {np.percentile(st[:, k], 90):8.0f}")
