import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bgflow_amd import configs
dev = torch.device("cuda:0")
gen = configs.make_ala2_spline_generator(dev)
g = torch.Generator(device=dev).manual_seed(1234)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
zs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
xs = tuple(zs); inter = []
total = 0
for i, block in enumerate(gen.flow):
    *xs, dd = block(*xs)
    for v in xs:
        if v.requires_grad: v.retain_grad()
    inter.append((i, xs, dd))
    total = total + dd
x = xs[0]
e = gen._target.energy(x)
per = e - total
print("per-sample loss finite:", torch.isfinite(per).all().item(), "nonfinite count", (~torch.isfinite(per)).sum().item(), "max", per[torch.isfinite(per)].max().item())
loss = per.mean()
loss.backward()
for i, xs_i, dd in inter[::-1]:
    bad = [(~torch.isfinite(v.grad)).sum().item() if v.grad is not None else -1 for v in xs_i]
    mx = [float(v.grad[torch.isfinite(v.grad)].abs().max()) if v.grad is not None else -1 for v in xs_i]
    print("block", i, type(gen.flow[i]).__name__, "nonfinite grads per tensor", bad, "max|g|", ["%.3g" % m for m in mx])
nbad = sum((~torch.isfinite(p.grad)).sum().item() for p in gen.flow.parameters())
print("nonfinite param grads", nbad)
