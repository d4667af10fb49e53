"""One affine coupling layer (32 conditioning features -> 32 transformed dims, shift + scale networks of 1 / 2 / 3 / 4 hidden layers of
H units) at 2^20 samples: one launch against the layer-by-layer path.  python tools/r05_deep_affine.py [B] [reps]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bgflow_amd as bg
from bgflow_amd.utils import hash_init_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(B, 32, device=dev, generator=g) for _ in range(2)]


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


for H in (64, 128):
    for L in (1, 2, 3, 4):
        layer = hash_init_(bg.CouplingFlow(bg.AffineTransformer(
            bg.DenseNet([32] + [H] * L + [32], torch.nn.ReLU()), bg.DenseNet([32] + [H] * L + [32], torch.nn.Tanh())),
            transformed_indices=(1,), cond_indices=(0,))).to(dev)
        layer.transformer.allow_fused = True
        t_f = ms(lambda: layer(*xs))
        layer.transformer.allow_fused = False
        t_g = ms(lambda: layer(*xs))
        print(f"affine, {L} hidden layers of {H}, B={B}: one launch {t_f:.3f} ms   layer by layer {t_g:.3f} ms")
