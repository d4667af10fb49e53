"""GPU: per-phase wave cycles of the cfg 2 coupling kernel (library built with -DBGK_AFF_TS=1).  usage: BGK_LIB=... python tools/r04_cfg2_ts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs
dev = torch.device("cuda:0")
gen = configs.make_affine8_generator(device=dev)
B = 1 << 20
x = torch.randn(B, 64, device=dev)
layer = [b for b in gen.flow if type(b).__name__ == "CouplingFlow"][0]
a, b = x[:, :32].contiguous(), x[:, 32:].contiguous()
with torch.no_grad():
    for _ in range(3):
        out = layer(a, b)
    torch.cuda.synchronize()
y = out[1]
ts = y.view(torch.int32).view(-1, 32 * 32)[:, :7].cpu().numpy().astype(np.int64)
dt = (ts[:, 1:] - ts[:, :-1]) % (1 << 32)
names = ["wait for the conditioner half", "layer 0 (both nets) + next DMA", "shift net (2 act + 2 GEMM)", "scale net", "tanh / log-det", "epilogue + stores + next DMA"]
tot = (ts[:, 6] - ts[:, 0]) % (1 << 32)
print(f"{ts.shape[0]} wave tiles, cycles per tile: mean {tot.mean():.0f} median {np.median(tot):.0f}")
for k, n in enumerate(names):
    print(f"   {n:36s} mean {dt[:, k].mean():8.0f}  median {np.median(dt[:, k]):8.0f}  p10 {np.percentile(dt[:, k], 10):8.0f}  p90 {np.percentile(dt[:, k], 90):8.0f}")
