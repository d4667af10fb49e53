"""GPU box: per-phase wave cycles of coupling_affine_dense_v2_kernel on cfg 5's affine couplings (library built with -DBGK_V2_AFF_TS=1 for
bgk_fused2.hip: lane 0 stamps s_memtime at the phase boundaries and writes the stamps over the tile's first output row).
BGK_LIB=gpurun_variants/lib_aff_ts.so python tools/r06_aff_ts.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgflow_amd import configs
from bgflow_amd.utils import hash_init_
dev = torch.device("cuda:0")
B = 1 << 20
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9, "AUGMENTED": 66}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False, "AUGMENTED": False}
fields = configs.IC_FIELDS + ("AUGMENTED",)
slot = {f: i for i, f in enumerate(fields)}
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, dims[f], device=dev, generator=g) for f in fields]
names = ["stage (issue + finish)", "shift: layer 0", "shift: hidden + output layers", "scale: layer 0 + park mu", "scale: hidden + output layers",
         "affine tail math", "dlogp + store"]
for on in ("TORSIONS", ("FIXED", "BONDS", "ANGLES")):
    l = hash_init_(configs._affine_coupling("AUGMENTED", on, dims, circ, slot)).to(dev)
    with torch.no_grad():
        for _ in range(3):
            out = l(*xs)
        torch.cuda.synchronize()
    y = out[slot["AUGMENTED"]]
    st = y.view(torch.int32)[0::32, :8].cpu().numpy().astype(np.int64) & 0xffffffff
    d = np.diff(st, axis=1) & 0xffffffff
    ok = (d < 1 << 24).all(axis=1)
    d = d[ok]
    print(f"AUGMENTED | {on}: {ok.sum()} of {len(ok)} tiles; cycles per tile (median) {np.median(d.sum(1)):.0f}")
    for k, nm in enumerate(names):
        print(f"  {nm:36s} median {np.median(d[:, k]):8.0f}   p90 {np.percentile(d[:, k], 90):8.0f}")
