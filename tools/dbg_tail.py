"""GPU: accuracy of the two sampling-tail kernels against the f64 oracle chain, and their time (dev tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bgflow_amd as bg
from bgflow_amd import configs
from bgflow_amd.ic import RelativeInternalCoordinateTransformation as Rel
from oracle import flow_oracle as fo
dev = torch.device("cuda:0")
gen, gen_cpu = configs.make_ala2_spline_generator(dev), configs.make_ala2_spline_generator()
TIME_ONLY = "--time-only" in sys.argv
n = 1 << 15
rng = np.random.default_rng(1234)
u = [rng.random((n, d), dtype=np.float32) for d in (17, 17, 17, 9)]
pb32 = []
fo.run_flow(gen_cpu.flow, u, dtype=np.float32, per_block=pb32)
x64, d64 = fo.run_flow(gen_cpu.flow, [v.astype(np.float64) for v in u], dtype=np.float64)
tot = np.abs(d64).reshape(-1)
ins = list(pb32[15][0])
st64 = [np.asarray(v, np.float64) for v in ins]; st32 = ins; t64 = 0; t32 = 0
for bcpu in list(gen_cpu.flow)[16:]:
    st64, dd = fo.run_block(bcpu, st64, False, np.float64); t64 = t64 + dd
    st32, dd = fo.run_block(bcpu, st32, False, np.float32); t32 = t32 + dd
tail = gen.flow.segments()[-1][1]
tin = [torch.as_tensor(np.ascontiguousarray(v)).to(dev) for v in ins]
def rep(name, dl, x):
    e = np.abs(dl.reshape(-1) - t64.reshape(-1)); ex = np.abs(x - st64[0]).max(-1)
    print(f"{name}: dlogp med {np.median(e):.1e} p99 {np.quantile(e,.99):.1e} max {e.max():.1e} frac>1e-5|dl| {(e > 1e-5*tot).mean():.4f} | x med {np.median(ex):.1e} p99 {np.quantile(ex,.99):.1e} max {ex.max():.1e}")
rep("C oracle f32 chain", t32, st32[0])
with torch.no_grad():
    for flag, uni in ((True, True), (True, False), (False, False)):
        if TIME_ONLY:
            break
        Rel.REGISTER_TAIL, Rel.UNIFORM_TAIL = flag, uni
        x, dl = tail(*tin)
        rep(f"GPU tail register={flag} uniform={uni}", dl.cpu().numpy(), x.cpu().numpy())
    B = 1 << 20
    g = torch.Generator(device=dev).manual_seed(1)
    z = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
    for flag, uni in (((True, True), (True, False), (False, False)) if TIME_ONLY else ((True, True), (True, False), (False, False)) * 2):
        Rel.REGISTER_TAIL, Rel.UNIFORM_TAIL = flag, uni
        for _ in range(1 if TIME_ONLY else 3): tail(*z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3 if TIME_ONLY else 20): tail(*z)
        e1.record(); torch.cuda.synchronize()
        print(f"register={flag} uniform={uni}: {e0.elapsed_time(e1) / (3 if TIME_ONLY else 20):.4f} ms per 2^20 samples")
    Rel.REGISTER_TAIL = Rel.UNIFORM_TAIL = True
