"""round 6: bgk_dense_layer alone -- ms per launch at 2^20 samples for a few layer shapes (HIP events, 20 launches), and the error
against the f64 product on 4096 rows.  Run ON THE GPU BOX:  python tools/r06_layer_time.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bgflow_amd import dense
from bgflow_amd.utils import hash_init_
dev = "cuda:0"
B = 1 << 20
for n_in, n_out, act in ((256, 256, 1), (256, 425, 0), (128, 256, 1), (64, 64, 3), (300, 300, 1)):
    lin = hash_init_(torch.nn.Linear(n_in, n_out)).to(dev)
    x = torch.randn(B, n_in, device=dev)
    for _ in range(3):
        y = dense.dense_layer(x, lin, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = dense.dense_layer(x, lin, act=act)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    xs = x[:4096].double().cpu()
    want = xs @ lin.weight.detach().double().cpu().T + lin.bias.detach().double().cpu()
    want = {0: want, 1: torch.nn.functional.silu(want), 3: torch.tanh(want)}[act]
    err = float((y[:4096].double().cpu() - want).abs().max())
    gb = 4.0 * B * (n_in + n_out) / 1e9
    print(f"dense_layer {n_in:4d} -> {n_out:4d} act {act}: {ms:7.3f} ms  ({gb / ms:6.1f} GB/ms = {gb / ms / 8 * 100:4.1f} % of HBM peak)  max |err| vs f64 {err:.2e}")
