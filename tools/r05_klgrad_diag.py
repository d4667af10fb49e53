"""GPU box: where does the flat KL gradient's distance from its f64 value come from?  Chunk of 8192 samples of the cfg-3 flow; the
gradient through (a) the bench's path, (b) library GEMMs in the backward, (c) unfused layers, (d) the block-by-block tail, and (e) the
reference's op chain in f32 on the host (what an f32 autograd evaluation gives: the floor of any f32 implementation) -- each against
f64 autograd of the same chain (oracle/torch_flow.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_round4 import _grad_errors, _kl_gradient_f64, _kl_gradient_gpu      # noqa: E402
from bgflow_amd import configs, dense                                              # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gen = configs.make_ala2_spline_generator(dev)
gen64 = configs.make_ala2_spline_generator().double()
gen32 = configs.make_ala2_spline_generator()
mean = gen._target._mean.detach().cpu().double()
g = torch.Generator(device=dev).manual_seed(2024)
z = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
ref, _ = _kl_gradient_f64(gen64, mean, [v.cpu().double() for v in z], 1)


def report(name, got):
    rel, worst = _grad_errors(got, ref)
    per = sorted(((float((got[n] - ref[n]).norm()) / max(float(ref[n].norm()), 1e-30), n) for n in ref), reverse=True)
    print(f"{name:42s} rel L2 {rel:.2e}   worst entry {worst[0]:.2e} ({worst[1]})")
    print("      per-tensor rel L2, top 4:", "  ".join(f"{n.replace('_blocks.', 'b').replace('.transformer._params_net', '').replace('_layers.', 'L')} {e:.1e}" for e, n in per[:4]))
    by_block = {}
    for e, n in per:
        by_block.setdefault(int(n.split(".")[1]), []).append(e)
    print("      per block (max over its tensors):", " ".join(f"{b}:{max(v):.0e}" for b, v in sorted(by_block.items())))


report("(a) bench path", _kl_gradient_gpu(gen, z)[0])
dense.FUSED_MLP_BACKWARD, dense.FUSED_WEIGHT_GRAD = False, False
report("(b) library GEMMs in the backward", _kl_gradient_gpu(gen, z)[0])
dense.FUSED_MLP_BACKWARD, dense.FUSED_WEIGHT_GRAD = True, True
for blk in gen.flow:
    if hasattr(blk, "transformer"):
        blk.transformer.allow_fused = False
report("(c) unfused layers (GEMMs + spline kernels)", _kl_gradient_gpu(gen, z)[0])
for blk in gen.flow:
    if hasattr(blk, "transformer"):
        blk.transformer.allow_fused = True
gen.flow.FUSE_TRAINING_TAIL = False
report("(d) block-by-block tail", _kl_gradient_gpu(gen, z)[0])
gen.flow.FUSE_TRAINING_TAIL = True
g32, _ = _kl_gradient_f64(gen32, mean.float(), [v.cpu() for v in z], 1)
report("(e) reference op chain, f32, host", {n: v.double() for n, v in g32.items()})
