"""One B|A spline coupling layer (17 dims) at 2^20 samples with 1 / 2 / 3 / 4 hidden layers of 128 units: one launch against the
layer-by-layer path.  python tools/r05_deep.py [B] [reps]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bgflow_amd import configs
from bgflow_amd.utils import hash_init_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]


def ms(fn):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for L in (1, 2, 3, 4):
    layer = hash_init_(configs._spline_coupling("BONDS", "ANGLES", dims, circ, slot, hidden=(128,) * L)).to(dev)
    layer.transformer.allow_fused = True
    t_f = ms(lambda: layer(*xs))
    layer.transformer.allow_fused = False
    t_g = ms(lambda: layer(*xs))
    print(f"{L} hidden layers of 128, B={B}: one launch {t_f:.3f} ms   layer by layer {t_g:.3f} ms")
