"""debug: bgk_affine_net_backward64 called directly on fixed inputs: run-to-run stability per output, per activation (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bgflow_amd import _lib, dense
dev = torch.device('cuda:0')
lib = _lib.lib()
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
d = n_in = 32; H0 = H1 = 64
W0 = torch.randn(H0, n_in, device=dev) / 6; W1 = torch.randn(H1, H0, device=dev) / 8; W2 = torch.randn(d, H1, device=dev) / 8
b0 = torch.zeros(H0, device=dev); b1 = torch.zeros(H1, device=dev); b2 = torch.zeros(d, device=dev)
f16 = lambda n: torch.empty((n, 64, 8), dtype=torch.float16, device=dev)
A0, A1, A2, cs = f16(12), f16(18), f16(9), torch.empty(6, device=dev)
tb = {}
dense._t_operand_bufs(tb, d, n_in, dev)
st = lib.bgk_pack_mlp_h2(_lib.ptr(W0), _lib.ptr(b0), n_in, H0, _lib.ptr(W1), _lib.ptr(b1), H1, _lib.ptr(W2), _lib.ptr(b2), d, None, 1, 1, 2,
                         _lib.ptr(A0), _lib.ptr(A1), _lib.ptr(A2), _lib.ptr(cs), _lib.stream_ptr(dev)); assert st == 0
st = lib.bgk_pack_mlp_h2_t(_lib.ptr(W0), n_in, H0, _lib.ptr(W1), H1, _lib.ptr(W2), d, _lib.ptr(cs), _lib.ptr(tb["T0"]), _lib.ptr(tb["T1"]), _lib.ptr(tb["T2"]),
                           _lib.stream_ptr(dev)); assert st == 0
x = torch.randn(B, n_in, device=dev); z0 = torch.randn(B, 64, device=dev); z1 = torch.randn(B, 64, device=dev)
g = torch.randn(B, d, device=dev) * 1e-5
am = torch.zeros(3, device=dev); am[0] = g.abs().max()
ws = torch.empty(int(lib.bgk_affine_net_backward64_workspace(B, d, H1, H0, n_in)), device=dev)
def run(act):
    outs = [torch.empty(s, device=dev) for s in ((d, H1), (d,), (H1, H0), (H1,), (H0, n_in), (H0,))]
    gx = torch.empty(B, n_in, device=dev)
    st = lib.bgk_affine_net_backward64(_lib.ptr(g), d, d, _lib.ptr(z1), _lib.ptr(z0), _lib.ptr(x), n_in, n_in, H1, H0,
                                       _lib.ptr(tb["T0"]), _lib.ptr(tb["T1"]), _lib.ptr(tb["T2"]), _lib.ptr(cs), act, B,
                                       _lib.ptr(gx), n_in, None, 0, _lib.ptr(am), _lib.ptr(ws), ws.numel(), *[_lib.ptr(o) for o in outs], 0, _lib.stream_ptr(dev))
    assert st == 0, lib.bgk_last_error()
    torch.cuda.synchronize()
    return [gx] + outs
names = ["g_x", "gW2", "gb2", "gW1", "gb1", "gW0", "gb0"]
for act in (1, 2, 3):
    ref = run(act)
    cnt = {n: 0 for n in names}; worst = {n: 0.0 for n in names}
    for it in range(30):
        cur = run(act)
        for n, a, b in zip(names, cur, ref):
            e = float((a - b).abs().max() / b.abs().max())
            if e > 0: cnt[n] += 1; worst[n] = max(worst[n], e)
    print("B", B, "act", act, {n: (cnt[n], f"{worst[n]:.1e}") for n in names if cnt[n]} or "bit-stable",
          "non-finite:", {n: int((~torch.isfinite(t)).sum()) for n, t in zip(names, ref) if not torch.isfinite(t).all()})
    if not torch.isfinite(ref[0]).all():
        badrows = (~torch.isfinite(ref[0])).any(1).nonzero().flatten()
        print("   g_x rows non-finite:", badrows[:10].tolist(), "...", len(badrows), "tiles:", sorted(set((badrows // 32).tolist()))[:12])
