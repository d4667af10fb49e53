"""Error of the cfg-3 flow (16 spline couplings + icdf maps + IC) against the reference's f64 evaluation (tests/golden/g_flow16.npz),
per conditioner-GEMM mode, next to the reference's own f32-vs-f64 deviation on the same inputs.  Run on an MI355X."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs, dense

G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g_flow16.npz"))
dev = torch.device("cuda:0")
u = [torch.tensor(G[k], device=dev) for k in ("u_bonds", "u_angles", "u_torsions", "u_fixed")]
ref = G["dlogp64"].reshape(-1)
print(f"reference f32 vs its own f64: max |d dlogp| / |dlogp| = {np.abs(G['dlogp32'].reshape(-1) - ref).max() / np.abs(ref).max():.2e}"
      f"   max |d x| = {np.abs(G['x32'] - G['x64']).max():.2e}")
for mode in ("f32", "f16x2", "bf16"):
    dense.GEMM_MODE = mode
    gen = configs.make_ala2_spline_generator(dev)
    with torch.no_grad():
        x, dl = gen.flow(*u)
    dl = dl.cpu().numpy().reshape(-1)
    print(f"{mode:6s}: max |d dlogp| / |dlogp| = {np.abs(dl - ref).max() / np.abs(ref).max():.2e}   rms = "
          f"{np.sqrt(((dl - ref) ** 2).mean()) / np.abs(ref).max():.2e}   max |d x| = {np.abs(x.cpu().numpy() - G['x64']).max():.2e}")
