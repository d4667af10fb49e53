"""Accuracy of the GPU flows against the reference's f64 evaluation (tests/golden) and the f32 CPU oracle, per conditioner-GEMM
mode / kernel generation, PER SAMPLE; bin-index mismatches of the split-f16 kernels with their distance to the nearest knot.
Run on an MI355X:  python tools/accuracy_report.py [n_random]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bgflow_amd as bg
from bgflow_amd import configs, dense, _lib
from oracle import flow_oracle as fo

dev = torch.device("cuda:0")
L = _lib.lib()


def t(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


def rel(a, ref, floor=0.0):
    a, ref = a.reshape(-1), ref.reshape(-1)
    return np.abs(a - ref) / np.maximum(np.abs(ref), floor)


def tie_report(gen, gen_cpu, u, label):
    """every coupling layer fed with the ORACLE's inputs of that layer: mismatching bin indices and their knot distance"""
    pb, trace = [], []
    fo.run_flow(gen_cpu.flow, u, dtype=np.float32, per_block=pb, trace=trace)
    n_mis = n_tot = 0
    worst = 0.0
    ti_trace = 0
    with torch.no_grad():
        for i, block in enumerate(gen.flow):
            if not isinstance(block, bg.CouplingFlow) or not hasattr(block.transformer, "return_bin_indices"):
                continue
            block.transformer.return_bin_indices = True
            ins = u if i == 0 else pb[i - 1][0]
            block(*[t(v) for v in ins])
            idx = block.transformer.last_bin_indices.cpu().numpy()
            block.transformer.return_bin_indices = False
            det = trace[ti_trace]; ti_trace += 1
            y = np.asarray(ins[block.transformed_indices[0]])
            mis = idx != det["bin_idx"]
            n_mis += int(mis.sum()); n_tot += mis.size
            if mis.any():
                dist = np.abs(det["knots"] - y[..., None]).min(-1)
                worst = max(worst, float(dist[mis].max()))
                assert np.abs(idx - det["bin_idx"]).max() <= 1
    print(f"  {label}: bin-index mismatches vs the f32 oracle {n_mis} of {n_tot} ({n_mis / max(n_tot, 1):.2e}); "
          f"largest |x - knot| among them {worst:.2e}")


for name, make, keys in (("g_flow16", configs.make_ala2_spline_generator, ("u_bonds", "u_angles", "u_torsions", "u_fixed")),
                         ("g_aug", configs.make_ala2_augmented_generator, ("u_bonds", "u_angles", "u_torsions", "u_fixed", "u_aug"))):
    G = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    u = [G[k] for k in keys]
    nc = int(G["n_couplings"])
    gen_cpu = make()
    outs32, dl32 = fo.run_flow(gen_cpu.flow, u, dtype=np.float32)
    _, dlc32 = fo.run_flow(gen_cpu.flow[:nc], u, dtype=np.float32)
    print(f"== {name}: |dlogp64| in [{np.abs(G['dlogp64']).min():.1f}, {np.abs(G['dlogp64']).max():.1f}], couplings-only |dlogp| in "
          f"[{np.abs(G['dlogp_c64']).min():.2f}, {np.abs(G['dlogp_c64']).max():.2f}]")
    print(f"  reference f32 vs its f64: whole flow per-sample rel {rel(G['dlogp32'], G['dlogp64']).max():.2e}; couplings abs {np.abs(G['dlogp_c32'] - G['dlogp_c64']).max():.2e}")
    print(f"  f32 CPU oracle vs f64:    whole flow per-sample rel {rel(dl32, G['dlogp64']).max():.2e}; couplings abs {np.abs(dlc32 - G['dlogp_c64']).max():.2e}")
    for mode, variant in (("f32", 2), ("f16x2", 2), ("f16x2", 1), ("bf16", 2)):
        dense.GEMM_MODE = mode
        L.bgk_set_option(1, variant)
        gen = make(dev)
        with torch.no_grad():
            *xs, dl = gen.flow(*[t(v) for v in u])
            *st, dlc = gen.flow[:nc](*[t(v) for v in u])
        dl, dlc = dl.cpu().numpy(), dlc.cpu().numpy()
        state = torch.cat(st, -1).cpu().numpy()
        print(f"  {mode:6s} gen{variant}: whole flow per-sample rel vs f64 {rel(dl, G['dlogp64']).max():.2e}  vs f32 oracle {rel(dl, dl32).max():.2e} | "
              f"couplings: abs vs f64 {np.abs(dlc - G['dlogp_c64']).max():.2e}  rel(floor 1) {rel(dlc, G['dlogp_c64'], 1.0).max():.2e}  vs f32 oracle abs {np.abs(dlc - dlc32).max():.2e} | "
              f"state after couplings max abs vs f64 {np.abs(state - G['state_c64']).max():.2e} | x max abs vs f64 {np.abs(xs[0].cpu().numpy() - G['x64']).max():.2e}")
        if mode == "f16x2":
            tie_report(gen, gen_cpu, u, f"{mode} gen{variant}")
    L.bgk_set_option(1, 2)
    dense.GEMM_MODE = "f16x2"

# larger random batch through the cfg-3 couplings: tie statistics of the shipped default
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
rng = np.random.default_rng(7)
u = [rng.random((n, d), dtype=np.float32) for d in (17, 17, 17, 9)]
gen, gen_cpu = configs.make_ala2_spline_generator(dev), configs.make_ala2_spline_generator()
tie_report(gen, gen_cpu, u, f"cfg 3, {n} random samples, f16x2 gen2")
_, dl32 = fo.run_flow(gen_cpu.flow, u, dtype=np.float32)
_, dl64 = fo.run_flow(gen_cpu.flow, [v.astype(np.float64) for v in u], dtype=np.float64)
with torch.no_grad():
    x, dl = gen.flow(*[t(v) for v in u])
dl = dl.cpu().numpy()
print(f"  whole flow per-sample rel: vs f64 oracle max {rel(dl, dl64).max():.2e} (f32 oracle vs f64: {rel(dl32, dl64).max():.2e}), vs f32 oracle max {rel(dl, dl32).max():.2e}")
