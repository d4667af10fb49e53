"""GPU box: bgk_dense_backward_dx on a gradient whose rows start on 16-byte boundaries (pitch 448 / 428 floats) and on one whose rows do
not (pitch 425), a few batch sizes; prints checksums and NaN counts, and saves the outputs for a comparison between two libraries
(python tools/r05_dx_align.py save <file> ; python tools/r05_dx_align.py cmp <fileA> <fileB>)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "cmp":
    A, Bt = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in A:
        a, b = A[k].double(), Bt[k].double()
        print(f"{k:28s} nan {int(torch.isnan(a).sum())}/{int(torch.isnan(b).sum())}  max|a-b| {float((a - b).abs().max()):.3e}  max|b| {float(b.abs().max()):.3e}")
    sys.exit(0)

from bgflow_amd import dense                                     # noqa: E402

dev = torch.device("cuda:0")
out = {}
for B in (64, 100, 4096 + 7):
    for P, pitch, n_in in ((425, 448, 17), (425, 428, 17), (425, 425, 17), (225, 225, 9), (200, 200, 8)):
        g = torch.Generator(device=dev).manual_seed(B + pitch)
        W0, W1, W2 = (torch.randn(128, n_in, device=dev, generator=g) * 0.2, torch.randn(128, 128, device=dev, generator=g) * 0.09,
                      torch.randn(P, 128, device=dev, generator=g) * 0.09)
        cs = torch.tensor([2.0 ** 15, 2.0 ** -15] * 3, device=dev)
        gp = torch.empty(B, pitch, device=dev).normal_(generator=g)[:, :P] * 1e-3
        z1, z0 = torch.randn(B, 128, device=dev, generator=g), torch.randn(B, 128, device=dev, generator=g)
        x = torch.rand(B, n_in, device=dev, generator=g)
        am = dense.absmax_of(gp, None, None)
        g_z1, g_z0, _, _, g_x = dense._dense_backward_dx(gp, z1, z0, x, W0, W1, W2, cs, 1, False, True, {}, want_h=False, absmax=am.clone())
        torch.cuda.synchronize()
        for nm, t in (("g_z1", g_z1), ("g_z0", g_z0), ("g_x", g_x)):
            out[f"B{B} P{P} pitch{pitch} {nm}"] = t.detach().cpu().clone()
torch.save(out, sys.argv[2])
