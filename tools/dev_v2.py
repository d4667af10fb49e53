"""Dev harness (GPU): second-generation split-f16 coupling kernel vs the first-generation one and the CPU oracle; timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs, _lib
from bgflow_amd.utils import hash_init_, synth
from oracle import flow_oracle as fo

dev = torch.device("cuda:0")
L = _lib.lib()
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
pairs = {"T|F": ("TORSIONS", "FIXED"), "F|T": ("FIXED", "TORSIONS"), "B|A": ("BONDS", "ANGLES")}


def layer(kind, d=None):
    what, on = pairs[kind]
    l = hash_init_(configs._spline_coupling(what, on, dims, circ, slot))
    return (l.to(d) if d is not None else l), slot[what]


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


def run(l, xs, inverse, variant):
    L.bgk_set_option(1, variant)
    l.transformer.gemm_mode = "f16x2"
    l.transformer.return_bin_indices = True
    with torch.no_grad():
        *outs, dl = l(*[t(v) for v in xs], inverse=inverse)
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in outs], dl.cpu().numpy(), l.transformer.last_bin_indices.cpu().numpy()


if "--check" in sys.argv or len(sys.argv) == 1:
    for kind in ("T|F", "F|T", "B|A"):
        for inverse in (False, True):
            for B in (1, 31, 4133):
                lc, ti = layer(kind)
                lg, _ = layer(kind, dev)
                xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
                o1, dl1, b1 = run(lg, xs, inverse, 1)
                o2, dl2, b2 = run(lg, xs, inverse, 2)
                trace = []
                o64, dl64 = fo.run_block(lc, [v.astype(np.float64) for v in xs], inverse, np.float64, trace)
                trace32 = []
                o32, dl32 = fo.run_block(lc, xs, inverse, np.float32, trace32)
                e1 = np.abs(o1[ti] - o64[ti]).max(), (np.abs(dl1 - dl64) / np.abs(dl64).clip(1e-3)).max()
                e2 = np.abs(o2[ti] - o64[ti]).max(), (np.abs(dl2 - dl64) / np.abs(dl64).clip(1e-3)).max()
                e3 = np.abs(o32[ti] - o64[ti]).max(), (np.abs(dl32 - dl64) / np.abs(dl64).clip(1e-3)).max()
                nb1, nb2 = int((b1 != trace32[0]["bin_idx"]).sum()), int((b2 != trace32[0]["bin_idx"]).sum())
                others_same = all(np.array_equal(a, b) for i, (a, b) in enumerate(zip(o1, o2)) if i != ti)
                print(f"{kind} inv={int(inverse)} B={B:5d}: out err v1 {e1[0]:.2e} v2 {e2[0]:.2e} orc32 {e3[0]:.2e} | "
                      f"dlogp rel v1 {e1[1]:.2e} v2 {e2[1]:.2e} orc32 {e3[1]:.2e} | bin mismatches vs orc32 v1 {nb1} v2 {nb2} of {b1.size} | pass-through same {others_same} "
                      f"| max|dl| {np.abs(dl64).max():.1f}")

if "--time" in sys.argv or "--time-v2" in sys.argv or len(sys.argv) == 1:
    B = 1 << 20
    g = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
    for kind in ("B|A", "T|F", "F|T"):
        lg, _ = layer(kind, dev)
        lg.transformer.gemm_mode = "f16x2"
        for variant in ((2,) if "--time-v2" in sys.argv else (1, 2)):
            L.bgk_set_option(1, variant)
            for inverse in (False, True):
                with torch.no_grad():
                    for _ in range(3):
                        lg(*xs, inverse=inverse)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        lg(*xs, inverse=inverse)
                    e1.record()
                    torch.cuda.synchronize()
                print(f"time {kind} v{variant} inv={int(inverse)}: {e0.elapsed_time(e1) / 20:.3f} ms per call")
    L.bgk_set_option(1, 2)
