"""GPU: the fused inverse head (xyz -> IC + cdf maps) against the block path and the f64 oracle; time of both (dev tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bgflow_amd as bg
from bgflow_amd import configs
from oracle import flow_oracle as fo
dev = torch.device("cuda:0")
gen, gen_cpu = configs.make_ala2_spline_generator(dev), configs.make_ala2_spline_generator()
g = torch.Generator(device=dev).manual_seed(1)
B = 1 << 20
z = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
with torch.no_grad():
    x, dl = gen.flow(*z)
    segs = gen.flow.segments(inverse=True)
    head = segs[0][1]
    n = 1 << 14
    xr = x[:n].contiguous()
    b, a, t, zf, dlh = head(xr, inverse=True)
    bb, ab, tb, zfb, dlb = head._blocks_path(xr)
    st64, d64 = [xr.cpu().numpy().astype(np.float64)], 0
    st32, d32 = [xr.cpu().numpy()], 0
    for bcpu in reversed(list(gen_cpu.flow)[16:]):
        st64, dd = fo.run_block(bcpu, st64, True, np.float64); d64 = d64 + dd
        st32, dd = fo.run_block(bcpu, st32, True, np.float32); d32 = d32 + dd
    def rep(name, outs, dlv):
        e = np.abs(np.asarray(dlv).reshape(-1) - d64.reshape(-1))
        ez = max(np.median(np.abs(np.asarray(o) - s).max(-1)) for o, s in zip(outs, st64))
        print(f"{name}: dlogp med {np.median(e):.1e} p99 {np.quantile(e,.99):.1e} max {e.max():.1e} | latent med-of-rowmax {ez:.1e}")
    rep("C oracle f32", st32, d32)
    rep("GPU fused head", [v.cpu().numpy() for v in (b, a, t, zf)], dlh.cpu().numpy())
    rep("GPU blocks", [v.cpu().numpy() for v in (bb, ab, tb, zfb)], dlb.cpu().numpy())
    for name, fn in (("fused", lambda: head(x, inverse=True)), ("blocks", lambda: head._blocks_path(x)), ("fused", lambda: head(x, inverse=True))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 10:.4f} ms per 2^20 samples")
    ms = []
    for lbl, seg in segs:
        pass
