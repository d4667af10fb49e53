"""GPU box: per-phase wave cycles of the sampling tail icdf_ic2xyz_uni_kernel (library built with -DBGK_TAIL_TS=1 for bgk_tail.hip: lane 0
stamps s_memtime at the phase boundaries and writes the stamps over the tile's first output row).
BGK_LIB=gpurun_variants/lib_tail_ts.so python tools/r06_tail_ts.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import configs
dev = torch.device("cuda:0")
B = 1 << 20
gen = configs.make_ala2_spline_generator(dev)
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
blocks = list(gen.flow)
tail = [b for b in blocks if "Coupling" not in type(b).__name__]
with torch.no_grad():
    for _ in range(3):
        out = gen.flow(*xs)
    torch.cuda.synchronize()
x = out[0]
st = x.view(torch.int32)[0::64, :8].cpu().numpy().astype(np.int64) & 0xffffffff
d = np.diff(st[:, :7], axis=1) & 0xffffffff
ok = (d < 1 << 24).all(axis=1)
d = d[ok]
names = ["DMA issue + wait for the fixed tile", "fixed pass (icdf + blacken)", "wait for the rest + bonds / angles / torsions elementwise + row sums",
         "17 placements", "dlogp + rows -> LDS -> stores issued", "stores acknowledged"]
print(f"{ok.sum()} of {len(ok)} tiles; wave cycles per 64-sample tile (median) {np.median(d.sum(1)):.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:72s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
