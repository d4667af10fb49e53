"""Micro-benchmark of single kernels (used under rocprofv3): python tools/prof_layer.py [kind] [B] [reps]
kind: fused-BA | fused-TF | fused-FT | rqs-BA (generic spline kernel only) | ic | affine
env BGK_GEMM=f32|f16x2 selects the conditioner GEMM mode of the fused kernel"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs, dense
if os.environ.get("BGK_GEMM"):
    dense.GEMM_MODE = os.environ["BGK_GEMM"]
from bgflow_amd.utils import hash_init_

kind = sys.argv[1] if len(sys.argv) > 1 else "fused-BA"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
pairs = {"BA": ("BONDS", "ANGLES"), "TF": ("TORSIONS", "FIXED"), "FT": ("FIXED", "TORSIONS")}
if kind.startswith("fused-") or kind.startswith("rqs-"):
    what, on = pairs[kind.split("-")[1]]
    layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot)).to(dev)
    if kind.startswith("rqs-"):
        layer.transformer.allow_fused = False
    fn = lambda: layer(*xs)
elif kind == "ic":
    gen = configs.make_ala2_spline_generator(dev)
    ic = gen.flow[20]
    with torch.no_grad():
        *ics, _ = gen.flow[:20](*xs)
    fn = lambda: ic(*ics)
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
print(f"{kind} B={B}: {e0.elapsed_time(e1) / reps:.3f} ms per call")
