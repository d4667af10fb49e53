"""GPU: per-phase wave time of the spline coupling kernel (library built with -DBGK_V2_TS=1: lane 0 of every wave stores s_memtime at
phase boundaries where the bin indices would go).  usage: BGK_LIB=gpurun_variants/lib_ts.so python tools/r04_phase_ts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs
from bgflow_amd.utils import hash_init_

dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
pairs = {"T|F": ("TORSIONS", "FIXED"), "F|T": ("FIXED", "TORSIONS"), "B|A": ("BONDS", "ANGLES")}
B = 1 << 20
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
names = ["stage", "layer0", "layer1", "chunk0 gemm", "chunk0 (spline0 | gemm1)", "chunk1", "chunk2", "chunk3", "chunk4"]
for kind in ("B|A", "T|F", "F|T"):
    what, on = pairs[kind]
    l = hash_init_(configs._spline_coupling(what, on, dims, circ, slot)).to(dev)
    l.transformer.gemm_mode = "f16x2"
    l.transformer.return_bin_indices = True
    d = dims[what]
    nck = (d + 4) // 5
    for inverse in (False, True):
        with torch.no_grad():
            for _ in range(2):
                l(*xs, inverse=inverse)
            torch.cuda.synchronize()
        raw = l.transformer.last_bin_indices.view(-1, 32 * d)[:, :16].cpu().numpy().astype(np.int64)
        ts = raw[:, : 6 + nck]
        ex = lambda a, b: ((raw[:, a] - raw[:, b]) % (1 << 32)).mean()
        print(f"   [stage: loads issued {ex(12, 0):.0f} | wait + cos-sin pass {ex(1, 12):.0f} ]  [layer1: ring start + tile-0 activation {ex(14, 2):.0f} | threaded tiles 1-3 {ex(15, 14):.0f} | tail events {ex(3, 15):.0f}]")
        dt = (ts[:, 1:] - ts[:, :-1]) % (1 << 32)
        tot = (ts[:, -1] - ts[:, 0]) % (1 << 32)
        span = None
        print(f"{kind} inv={int(inverse)}: {ts.shape[0]} wave tiles; wave time per tile: mean {tot.mean():.0f} median {np.median(tot):.0f} cycles (memtime ticks)")
        lab = names[:4] + [f"chunk{c} (to LDS, spline{c} | gemm{c + 1})" for c in range(nck)] + ["output"]
        for k in range(dt.shape[1]):
            print(f"   {lab[k]:38s} mean {dt[:, k].mean():8.0f}  median {np.median(dt[:, k]):8.0f}  p10 {np.percentile(dt[:, k], 10):8.0f}  p90 {np.percentile(dt[:, k], 90):8.0f}")
        if "--hist" in sys.argv and not inverse:
            t0 = raw[:, 0].astype(np.int64); t0 = (t0 - t0.min()) % (1 << 32)
            T = float(np.median(tot))
            h, _ = np.histogram(t0 / T, bins=np.arange(0, 20.01, 0.125))
            print("   start-time histogram (bins of 1/8 tile time):", " ".join(str(x) for x in h))
