"""GPU box: bgk_affine_coupling_backward64 (the one-call backward of cfg 2's coupling class) against the separate launches
(bgk_affine_backward + 2 x bgk_affine_net_backward64) and against itself run to run -- both directions, saved and recomputed forms, batches
that end in partial tiles.  Written after a run-time `inverse` flag in the tile loop turned the forward-direction instance's bias / log_alpha
sums into garbage (104 spilled scalar registers; the flag is a template parameter now): every gradient must be BIT-stable over 20 runs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bgflow_amd as bg
from bgflow_amd import dense
from bgflow_amd.utils import hash_init_
dev = "cuda:0"
def layer():
    net = lambda act: bg.DenseNet([32, 64, 64, 32], activation=act())     # noqa: E731
    tr = bg.AffineTransformer(shift_transformation=net(torch.nn.ReLU), scale_transformation=net(torch.nn.Tanh))
    return hash_init_(bg.SequentialFlow([bg.CouplingFlow(tr, transformed_indices=[1], cond_indices=[0])]), scale=1.5).to(dev)
bad = 0
for B in (1 << 16, 4133, 1000, 64, 33, 32, 1):
    for inverse in (False, True):
        flow = layer()
        g = torch.Generator().manual_seed(B)
        x0, y0 = torch.randn(B, 32, generator=g).to(dev), torch.rand(B, 32, generator=g).to(dev)
        w = torch.randn(B, 32, generator=g).to(dev)
        def run():
            for p in flow.parameters():
                p.grad = None
            x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
            _, out, dl = flow(x, y, inverse=inverse)
            ((out * w).sum() + 0.5 * dl.sum()).backward()
            return [x.grad, y.grad] + [p.grad.clone() for p in flow.parameters()]
        res = {}
        for mode in ("separate", "one call", "recompute"):
            dense.TAIL_FUSED64, dense.RECOMPUTE64 = mode != "separate", mode == "recompute"
            first = run()
            unstable = sum(any(not torch.equal(a, b) for a, b in zip(run(), first)) for _ in range(20))
            res[mode] = first
            rel = max(float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(first, res["separate"]))
            # (the separate launches are NOT bit-stable: bgk_affine_backward adds its log_alpha partials with float atomics)
            ok = (unstable == 0 or mode == "separate") and rel <= 5e-6 and all(bool(torch.isfinite(t).all()) for t in first)
            bad += not ok
            print(f"B {B:6d} inverse {int(inverse)} {mode:10s}: runs that differ {unstable:2d} of 20, max rel L2 distance to the separate launches {rel:.1e}" + ("" if ok else "   <-- FAIL"))
        dense.TAIL_FUSED64, dense.RECOMPUTE64 = True, False
print("FAILURES:", bad)
