"""GPU box: the KL gradient's distance from f64, stage by stage and sample by sample.  The cfg-3 flow block by block on a chunk of
uniform prior samples, the cotangent of every intermediate field tensor retained -- GPU kernels vs f64 autograd of the reference's
op chain (oracle/torch_flow.py) -- relative L2 per stage and field, with and without the 8 worst samples, and what those samples are."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import configs                                    # noqa: E402
from oracle import torch_flow as tfl                              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
use_gpu = torch.cuda.is_available()
dev = torch.device("cuda:0" if use_gpu else "cpu")
gen64 = configs.make_ala2_spline_generator().double()
mean64 = gen64._target._mean.detach().double()
g = torch.Generator().manual_seed(2024)
z = [torch.rand(B, d, generator=g) for d in (17, 17, 17, 9)]
if len(sys.argv) > 2 and use_gpu:      # `B chunk`: chunk `chunk` of B samples of tests/test_gpu_round4.py::test_kl_gradient_at_the_bench_batch's batch
    gg = torch.Generator(device=dev).manual_seed(2024)
    zz = [torch.rand(1 << 18, d, device=dev, generator=gg) for d in (17, 17, 17, 9)]
    c = int(sys.argv[2])
    z = [v[c * B:(c + 1) * B].cpu() for v in zz]
names = ("bonds", "angles", "torsions", "fixed")


def run(blocks, xs, step, mean):
    stages = []
    total = 0.0
    for bi, blk in enumerate(blocks):
        xs, dl = step(blk, xs)
        xs = list(xs)
        for v in xs:
            if v.requires_grad:
                v.retain_grad()
        total = total + dl
        stages.append(list(xs))
    x = xs[0]
    loss = ((0.5 * ((x - mean) ** 2).sum(-1, keepdim=True) - total).sum() / B)
    loss.backward()
    return stages, x, total


z64 = [v.double().requires_grad_(True) for v in z]
st64, x64, dl64 = run(list(gen64.flow._blocks), z64, lambda b, xs: tfl.run_block(b, xs, False, grad=True), mean64)

if use_gpu:
    gen = configs.make_ala2_spline_generator(dev)
    gen.flow.FUSE_TRAINING_TAIL = False
    gen.flow.FUSE_TRAINING_CHAINS = False
    mean = gen._target._mean

    def gstep(b, xs):
        *o, dl = b(*xs)
        return o, dl
    zg = [v.to(dev).requires_grad_(True) for v in z]
    stg, xg, dlg = run(list(gen.flow._blocks), zg, gstep, mean)
    tag = "GPU kernels, block by block"
else:                                  # no device: the reference's op chain in f32 on the host (what the comparison prints for an f32 autograd)
    gen32 = configs.make_ala2_spline_generator()
    zg = [v.clone().requires_grad_(True) for v in z]
    stg, xg, dlg = run(list(gen32.flow._blocks), zg, lambda b, xs: tfl.run_block(b, xs, False, grad=True), mean64.float())
    tag = "reference op chain in f32 on the host"

print(tag, "vs f64;  B =", B)
print("forward: x rel L2 %.2e max %.2e; dlogp abs max %.2e" % (float((xg.detach().cpu().double() - x64.detach()).norm() / x64.detach().norm()),
      float((xg.detach().cpu().double() - x64.detach()).abs().max()), float((dlg.detach().cpu().double() - dl64.detach()).abs().max())))


def cmp(a, b):
    a, b = a.cpu().double(), b.double()
    e = ((a - b) ** 2).sum(-1)
    n = (b ** 2).sum(-1)
    order = torch.argsort(e, descending=True)
    keep = torch.ones_like(e, dtype=torch.bool)
    keep[order[:8]] = False
    return float((e.sum() / n.sum()) ** 0.5), float((e[keep].sum() / n[keep].sum()) ** 0.5), order[:4].tolist(), float(e[order[0]] / e.sum()), float(n[order[0]] / n.sum())


rows = [("prior z", zg, z64)] + [(f"after block {i:2d}", a, b) for i, (a, b) in enumerate(zip(stg, st64))]
for label, a_l, b_l in rows[::-1]:
    parts = []
    for k, (a, b) in enumerate(zip(a_l, b_l)):
        if a.grad is None or b.grad is None:
            continue
        nm = names[k] if len(a_l) == 4 else f"t{k}"
        r, r8, worst, share_e, share_n = cmp(a.grad, b.grad)
        parts.append(f"{nm} {r:.1e} (w/o 8 worst {r8:.1e}; worst {worst[0]} carries {share_e:.0%} of err^2, {share_n:.0%} of |g|^2)")
    print(f"{label}: " + " | ".join(parts))

# the worst samples of the prior's cotangent: where do they sit?
for k in range(4):
    if zg[k].grad is None:
        continue
    a, b = zg[k].grad.cpu().double(), z64[k].grad
    e = (a - b).abs()
    flat = torch.argsort(e.reshape(-1), descending=True)[:6]
    for f in flat.tolist():
        i, j = divmod(f, e.shape[1])
        print(f"   g_z {names[k]}[{i},{j}]: got {float(a[i, j]):+.6e} want {float(b[i, j]):+.6e} (rel {float(e[i, j] / b[i, j].abs()):.1e}); z = {float(z[k][i, j]):.7f}; "
              f"sample's min distance to 0/1 over its fields: " + " ".join(f"{float(torch.minimum(z[q][i], 1 - z[q][i]).min()):.1e}" for q in range(4)))
