import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bgflow_amd as bg
from bgflow_amd import configs
from oracle import flow_oracle as fo
dev = torch.device("cuda:0")
gen, gen_cpu = configs.make_ala2_spline_generator(dev), configs.make_ala2_spline_generator()
g = torch.Generator(device=dev).manual_seed(1234)
B = 1 << 14
z = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
zr = [v.cpu().numpy() for v in z]
pb32 = []
x32, d32 = fo.run_flow(gen_cpu.flow, zr, dtype=np.float32, per_block=pb32)
x64, d64 = fo.run_flow(gen_cpu.flow, [v.astype(np.float64) for v in zr], dtype=np.float64)
tot = np.abs(d64).reshape(-1)
bg.SequentialFlow.FUSE_GENERATION_TAIL = True
with torch.no_grad():
    for i, (block, block_cpu) in enumerate(zip(gen.flow, gen_cpu.flow)):
        ins = zr if i == 0 else list(pb32[i - 1][0])
        *o, dl = block(*[torch.as_tensor(np.ascontiguousarray(v)).to(dev) for v in ins])
        o32, dl32 = fo.run_block(block_cpu, ins, False, np.float32)
        o64, dl64 = fo.run_block(block_cpu, [np.asarray(v, np.float64) for v in ins], False, np.float64)
        eg = np.abs(dl.cpu().numpy().reshape(-1) - dl64.reshape(-1)); eo = np.abs(dl32.reshape(-1) - dl64.reshape(-1))
        print(f"{i:2d} {type(block).__name__:12s} GPU: med {np.median(eg):.1e} p99 {np.quantile(eg,.99):.1e} max {eg.max():.1e} frac>1e-5|dl| {(eg > 1e-5*tot).mean():.4f} | f32 oracle: med {np.median(eo):.1e} p99 {np.quantile(eo,.99):.1e} max {eo.max():.1e} frac {(eo > 1e-5*tot).mean():.4f}")
    # the fused tail fed the oracle's state after the couplings
    ins = list(pb32[15][0])
    tail = gen.flow.segments()[-1][1]
    x, dl = tail(*[torch.as_tensor(np.ascontiguousarray(v)).to(dev) for v in ins])
    d64t = sum(fo.run_block(b, None, False, np.float64)[1] if False else 0 for b in [])  # placeholder
    st64 = [np.asarray(v, np.float64) for v in ins]; st32 = ins
    t64 = 0; t32 = 0
    for bcpu in list(gen_cpu.flow)[16:]:
        st64, dd = fo.run_block(bcpu, st64, False, np.float64); t64 = t64 + dd
        st32, dd = fo.run_block(bcpu, st32, False, np.float32); t32 = t32 + dd
    eg = np.abs(dl.cpu().numpy().reshape(-1) - t64.reshape(-1)); eo = np.abs(t32.reshape(-1) - t64.reshape(-1))
    print(f"fused tail GPU: med {np.median(eg):.1e} p99 {np.quantile(eg,.99):.1e} max {eg.max():.1e} frac {(eg > 1e-5*tot).mean():.4f} | f32 oracle chain: med {np.median(eo):.1e} p99 {np.quantile(eo,.99):.1e} max {eo.max():.1e} frac {(eo > 1e-5*tot).mean():.4f}")
    worst = np.argsort(-eg)[:5]
    print("worst rows", worst, eg[worst], eo[worst], "angles in", st32[0][worst, :0] if False else "")
    xa = np.asarray(ins[1])[worst]; print("min normalised angle input of worst rows", xa.min(-1))
