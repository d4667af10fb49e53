import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bgflow_amd import configs
dev = torch.device("cuda:0")
gen = configs.make_ala2_spline_generator(dev)
g = torch.Generator(device=dev).manual_seed(1234)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
_ = [torch.rand(1 << 20, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
with torch.no_grad():
    gen.flow(*_)
zs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
params = list(gen.flow.parameters())
opt = torch.optim.Adam(params, lr=1e-5)
for step in range(3):
    opt.zero_grad(set_to_none=True)
    xs = tuple(zs); total = 0
    for i, block in enumerate(gen.flow):
        *xs, dd = block(*xs)
        bad = [int((~torch.isfinite(v)).sum()) for v in xs] + [int((~torch.isfinite(dd)).sum())]
        if any(bad):
            print("step", step, "block", i, "nonfinite outputs", bad); break
        total = total + dd
    else:
        loss = (gen._target.energy(xs[0]) - total).mean()
        loss.backward()
        nb = sum(int((~torch.isfinite(p.grad)).sum()) for p in params)
        print("step", step, "loss", float(loss.detach()), "nonfinite grads", nb, "max|grad|", max(float(p.grad.abs().max()) for p in params))
        opt.step()
        print("   nonfinite params after step", sum(int((~torch.isfinite(p)).sum()) for p in params))
        continue
    break
