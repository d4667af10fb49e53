"""Error margins of the wide / deep envelope tests (tests/test_gpu_round5.py): the quantities the tests bound, printed.  GPU box."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bgflow_amd as bg
from bgflow_amd import configs
from bgflow_amd.utils import hash_init_, synth
from oracle import flow_oracle as fo

dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
t = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev)
warnings.simplefilter("ignore")
B = 1037
xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
for hidden in ((256, 256), (200, 130), (128,), (128, 128, 128), (64, 128, 32, 100), (96,) * 8):
    worst_y = worst_dl = 0.0
    for inverse in (False, True):
        for what, on in (("TORSIONS", "FIXED"), ("BONDS", "TORSIONS"), ("FIXED", "TORSIONS")):
            lc = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden))
            lg = hash_init_(configs._spline_coupling(what, on, dims, circ, slot, hidden=hidden)).to(dev)
            with torch.no_grad():
                *outs, dl = lg(*[t(v) for v in xs], inverse=inverse)
            o64, dl64 = fo.run_block(lc, [v.astype(np.float64) for v in xs], inverse, np.float64, [])
            ti = slot[what]
            worst_y = max(worst_y, float(np.abs(outs[ti].cpu().numpy() - o64[ti]).max()))
            worst_dl = max(worst_dl, float((np.abs(dl.cpu().numpy() - dl64) / (1.0 + np.abs(dl64))).max()))
    print(f"spline hidden {hidden}: max |y - y64| {worst_y:.2e} (bound 2e-5), max |dl - dl64| / (1 + |dl64|) {worst_dl:.2e} (bound 2e-5)")
B = 2111
xa = [synth(B + 3 * i, B, d) for i, d in enumerate((12, 20))]
for hidden, acts in (((4,), ("ReLU", "Tanh")), ((64,), ("SiLU", "SiLU")), ((128, 64, 32, 100), ("ReLU", "Tanh")), ((48,) * 5, ("Tanh", "Tanh")),
                     ((128,) * 8, ("SiLU", "ReLU"))):
    wy = wd = 0.0
    for inverse in (False, True):
        mk = lambda: hash_init_(bg.CouplingFlow(bg.AffineTransformer(bg.DenseNet([12, *hidden, 20], getattr(torch.nn, acts[0])()),
                                                                    bg.DenseNet([12, *hidden, 20], getattr(torch.nn, acts[1])())),
                                                transformed_indices=(1,), cond_indices=(0,)))
        lc, lg = mk(), mk().to(dev)
        with torch.no_grad():
            _, y, dl = lg(*[t(v) for v in xa], inverse=inverse)
        o64, dl64 = fo.run_block(lc, [v.astype(np.float64) for v in xa], inverse, np.float64, [])
        wy = max(wy, float((np.abs(y.cpu().numpy() - o64[1]) / (1.0 + np.abs(o64[1]))).max()))
        wd = max(wd, float((np.abs(dl.cpu().numpy() - dl64) / (1.0 + np.abs(dl64))).max()))
    print(f"affine hidden {hidden} {acts}: max rel |y - y64| {wy:.2e} (bound 2e-5), max rel |dl - dl64| {wd:.2e} (bound 2e-5)")
