"""GPU box: per-phase wave cycles of bgk_dense_backward_dx from s_memtime stamps (library built with -DBGK_SBD_TS=1 for
bgk_dense_bwd.hip: lane 0 of every wave stamps its phases and writes them over the first 32 floats of row b0 of g_z0).
Stamps: 0 start | 14 first GEMM done | 22 z1 arrived | 23 act' done | 24 g_z1 stores issued | 25 stores acknowledged | 19 | 20 second
GEMM done | 26 z0 arrived | 27 | 28 | 29 | 21 | 15 third GEMM + g_cond out, everything acknowledged"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import dense                                     # noqa: E402

dev = torch.device("cuda:0")
B, P, n_in = 1 << 18, 425, int(sys.argv[1]) if len(sys.argv) > 1 else 17
g = torch.Generator(device=dev).manual_seed(1)
W0, W1, W2 = (torch.randn(128, n_in, device=dev, generator=g) * 0.2, torch.randn(128, 128, device=dev, generator=g) * 0.09,
              torch.randn(P, 128, device=dev, generator=g) * 0.09)
cs = torch.tensor([2.0 ** 15, 2.0 ** -15] * 3, device=dev)
gp = torch.empty(B, 448, device=dev).normal_(generator=g)[:, :P] * 1e-6
z1, z0 = torch.randn(B, 128, device=dev, generator=g), torch.randn(B, 128, device=dev, generator=g)
x = torch.rand(B, n_in, device=dev, generator=g)
am = dense.absmax_of(gp, None, None)
for it in range(3):
    bufs = {}
    g_z1, g_z0, _, _, g_x = dense._dense_backward_dx(gp, z1, z0, x, W0, W1, W2, cs, 1, False, True, bufs, want_h=False, absmax=am.clone())
torch.cuda.synchronize()
st = g_z0.view(torch.int32)[0::32, :32].cpu().numpy().astype(np.int64) & 0xffffffff
order = [0, 14, 22, 23, 24, 25, 19, 20, 26, 27, 28, 29, 21, 15]
names = ["start", "GEMM1 (g stream)", "z1 wait", "act' z1", "g_z1 -> LDS -> stores issued", "stores acked", "split g_z1", "GEMM2",
         "z0 wait", "act' z0", "g_z0 stores issued", "stores acked", "(tail)", "GEMM3 + g_cond + drain"]
d = np.diff(st[:, order], axis=1) & 0xffffffff
ok = (d < 1 << 24).all(axis=1)
d = d[ok]
print(f"{ok.sum()} of {len(ok)} tiles; total cycles per tile (median): {np.median(d.sum(1)):.0f}")
for k, nm in enumerate(names[1:]):
    print(f"  {nm:34s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
acc = st[ok][:, 1:6]          # slots 1..5: cycles of the first GEMM spent waiting for the wave's own requests | at the workgroup barrier | computing (of which: mask + split | issuing the next group's requests)
for nm, col in (("first GEMM: waiting for own DMA + gradient requests", 0), ("first GEMM: at the barrier (other waves)", 1), ("first GEMM: split + issue + LDS reads + MFMAs", 2),
                ("   of which mask + f16 split of the gradient", 3), ("   of which issuing the next group's requests", 4)):
    print(f"  {nm:52s} median {np.median(acc[:, col]):8.0f}   p90 {np.percentile(acc[:, col], 90):8.0f}")
