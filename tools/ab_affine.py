"""GPU box: A/B of the two kernels behind bgk_coupling_affine_dense_h2 on the cfg 2 flow (same process, HIP events)."""
import sys, torch
sys.path.insert(0, ".")
from bgflow_amd import configs, _lib

dev = torch.device("cuda:0")
gen = configs.make_affine8_generator(device=dev)
B = 1 << 20
z = torch.randn(B, 64, device=dev)
res = {}
with torch.no_grad():
    for variant in (1, 2, 1, 2):
        _lib.lib().bgk_set_option(2, variant)
        for _ in range(3):
            x, dl = gen.flow(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            x, dl = gen.flow(z)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.setdefault(variant, []).append((ms, x.clone(), dl.clone()))
        print(f"variant {variant}: {ms:.3f} ms/flow  {B / ms * 1e3:.3e} samples/s  hbm frac {4672 * B / ms * 1e3 / 8e12:.3f}")
x1, d1 = res[1][0][1:]; x2, d2 = res[2][0][1:]
print("max |dx|", float((x1 - x2).abs().max()), "max |ddlogp|", float((d1 - d2).abs().max()))
