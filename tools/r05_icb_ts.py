"""GPU box: per-phase wave cycles of the DMA-staged IC -> xyz backward sweep (library built with -DBGK_ICB_TS=1 for bgk_ic.hip: lane 0
stamps s_memtime at the phase boundaries of every tile and writes the 8 stamps over the tile's first g_xfix row)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import configs                                    # noqa: E402

dev = torch.device("cuda:0")
B = 1 << 18
gen = configs.make_ala2_spline_generator(dev)
blk = list(gen.flow)[-1]
ic = [m for m in blk.modules() if hasattr(m, "_ic2xyz_backward")][0]
mixed = [m for m in blk.modules() if type(m).__name__ == "MixedCoordinateTransformation"][0]
g = torch.Generator(device=dev).manual_seed(3)
ins = [0.1 + 0.05 * torch.rand(B, 17, device=dev, generator=g), 0.2 + 0.6 * torch.rand(B, 17, device=dev, generator=g),
       torch.rand(B, 17, device=dev, generator=g), torch.randn(B, 9, device=dev, generator=g)]
with torch.no_grad():
    x, dl = blk(*ins)
wh = mixed._whiten
black = (wh.X0mean, wh.Tblacken, float(wh.jacobian_xz))
gx = torch.randn(B, 66, device=dev, generator=g)
gl = torch.full((B,), -1.0 / B, device=dev)
for it in range(3):
    gb, ga, gt, gf = ic._ic2xyz_backward(ins[0], ins[1], ins[2], x, black, gx, gl)
torch.cuda.synchronize()
st = gf.view(torch.int32)[0::64, :8].cpu().numpy().astype(np.int64) & 0xffffffff
d = np.diff(st, axis=1) & 0xffffffff
ok = (d < 1 << 24).all(axis=1)
d = d[ok]
names = ["DMA of x, g_x: issue + wait", "lift x rows, IC DMA issue, live scan", "wait for the IC tiles", "reverse sweep (17 placements)",
         "whitened coordinates + flags", "stores issued", "stores acknowledged"]
print(f"{ok.sum()} of {len(ok)} tiles; cycles per tile (median): {np.median(d.sum(1)):.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:40s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
