"""GPU box: per-phase wave cycles of bgk_dense_layer (library built with -DBGK_LAYER_TS=1 for bgk_dense_layer.hip: lane 0 stamps s_memtime at
the phase boundaries of its tile and writes the stamps over the tile's first output row).  BGK_LIB=gpurun_variants/lib_layer_ts.so"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import dense
from bgflow_amd.utils import hash_init_
dev = "cuda:0"
B = 1 << 20
for n_in, n_out, act in ((256, 256, 1), (128, 256, 1)):
    lin = hash_init_(torch.nn.Linear(n_in, n_out)).to(dev)
    x = torch.randn(B, n_in, device=dev)
    for _ in range(3):
        y = dense.dense_layer(x, lin, act=act)
    torch.cuda.synchronize()
    st = y.view(torch.int32)[0::32, :8].cpu().numpy().astype(np.int64) & 0xffffffff
    d = np.diff(st[:, :7], axis=1) & 0xffffffff
    ok = (d < 1 << 24).all(axis=1)
    d = d[ok]
    names = ["loads -> LDS", "LDS -> fragments, split", "group 0: GEMM", "group 0: epilogue math -> LDS", "group 0: stores issued", "group 1 (all of it)" if n_out > 128 else "-"]
    print(f"{n_in} -> {n_out}: {ok.sum()} of {len(ok)} tiles; s_memtime ticks per tile (median) {np.median(d.sum(1)):.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:36s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
