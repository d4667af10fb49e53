"""debug: run-to-run stability of the fused affine training layer, per parameter (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from bgflow_amd import dense
import test_gpu_round6 as t6
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
flow = t6._affine_layer(32, (64, 64), 32, (torch.nn.ReLU, torch.nn.Tanh)).to(dev)
g = torch.Generator(device=dev).manual_seed(11)
x0, y0 = torch.randn(B, 32, device=dev, generator=g), torch.randn(B, 32, device=dev, generator=g)
def run():
    for p in flow.parameters(): p.grad = None
    x = x0.clone().requires_grad_(True); y = y0.clone().requires_grad_(True)
    _, out, dl = flow(x, y)
    (out.square().mean() - dl.mean() + (out * x).mean()).backward()
    r = {n: p.grad.clone() for n, p in flow.named_parameters() if "log_alpha" not in n}
    r["g_x"] = x.grad.clone(); r["g_y"] = y.grad.clone(); r["out"] = out.detach().clone()
    return r
ref = run()
bad = {}
for it in range(200):
    junk = torch.full((1 << 20,), float("nan"), device=dev); del junk
    cur = run()
    for n in ref:
        e = float((cur[n] - ref[n]).abs().max() / ref[n].abs().max())
        if e > 0:
            bad.setdefault(n, []).append((it, e))
print("B", B, "FUSED_BWD64", dense.FUSED_BWD64, "FUSED_FWD64", dense.FUSED_FWD64)
for n, v in bad.items():
    print("  ", n[-40:], len(v), "of 200 runs differ; max rel", max(e for _, e in v))
if not bad: print("   bit-stable over 200 runs")
