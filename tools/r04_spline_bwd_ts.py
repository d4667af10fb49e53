"""GPU: per-phase wave time of bgk_spline_backward_dx (library built with -DBGK_SBD_TS=1 for bgk_dense_bwd.hip: lane 0 of every wave
keeps s_memtime stamps of its phases and writes them over row b0 of g_z0).
usage: BGK_LIB=gpurun_variants/lib_sbdts.so python tools/r04_spline_bwd_ts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs, dense
dense.FUSED_SPLINE_BACKWARD = True
from bgflow_amd.utils import hash_init_

dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
B = 1 << 18
keep = {}
orig = dense._spline_backward_dx


def spy(*a, **k):
    out = orig(*a, **k)
    keep["g_z0"] = out[3]
    return out


dense._spline_backward_dx = spy
for what, on in (("BONDS", "ANGLES"), ("TORSIONS", "FIXED"), ("FIXED", "TORSIONS")):
    layer = hash_init_(configs._spline_coupling(what, on, dims, circ, slot)).to(dev)
    d = dims[what]
    T = (d + 1) // 2
    for it in range(3):
        layer.zero_grad()
        xs = [torch.rand(B, dims[f], device=dev, generator=torch.Generator(device=dev).manual_seed(5 + i)).requires_grad_(True)
              for i, f in enumerate(configs.IC_FIELDS)]
        *out, dl = layer(*xs, inverse=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        loss = sum((o * o).sum() for o in out) + dl.sum()
        e0.record()
        loss.backward()
        e1.record()
        torch.cuda.synchronize()
    raw = keep["g_z0"].view(torch.int32).view(-1, 32, 128)[:, 0, :32].cpu().numpy().astype(np.int64)
    dt = lambda a, b: ((raw[:, a] - raw[:, b]) % (1 << 32))
    print(f"{what}|{on}: d = {d}, {raw.shape[0]} wave tiles, backward {e0.elapsed_time(e1):.3f} ms (all kernels); memtime ticks (100 MHz) per wave:")
    print(f"   whole wave {dt(15, 0).mean():8.0f}   prologue {dt(1, 0).mean():7.0f}   slots {dt(1 + T, 1).mean():8.0f}   knot-K k-steps {dt(14, 1 + T).mean():7.0f}   rest of the chain {dt(15, 14).mean():8.0f}")
    print("   per slot:", " ".join(f"{dt(2 + t, 1 + t).mean():.0f}" for t in range(min(T, 12))))
    print(f"   chain: activation backward of z1 (loads, stores) {dt(19, 14).mean():.0f} | second GEMM {dt(20, 19).mean():.0f} | activation backward of z0 {dt(21, 20).mean():.0f} | third GEMM + g_cond + drain {dt(15, 21).mean():.0f}")
    for nm, b, st in (("z1", 22, 14), ("z0", 26, 20)):
        print(f"   activation backward of {nm}: loads issued -> all arrived {dt(b, st).mean():.0f} | arithmetic {dt(b + 1, b).mean():.0f} | slab + store issue {dt(b + 2, b + 1).mean():.0f} | stores acknowledged {dt(b + 3, b + 2).mean():.0f}")
    print(f"   slot 1: wait for its element {dt(16, 2).mean():.0f} | request + VJP {dt(17, 16).mean():.0f} | 36 MFMAs + operand loads {dt(18, 17).mean():.0f} | stores {dt(3, 18).mean():.0f}")
