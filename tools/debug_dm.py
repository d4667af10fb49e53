import numpy as np, torch, sys
sys.path.insert(0,'.')
from bgflow_amd import _lib
from bgflow_amd.utils import synth
from oracle import oracle
hip_lib=_lib.lib(); dev=torch.device('cuda:0')
x = np.concatenate([synth(1, 1 << 16, scale=8.0), synth(2, 4096, scale=40.0), np.array([0.0, -0.0, 1.0, -87.5, 88.5, 20.0, 28.9,-79.5,-80.5,1e-30,-1e-30,1e-20], np.float32)]).astype(np.float32)
for which, code in (("exp", 0), ("log", 1), ("softplus", 2), ("silu", 3), ("tanh", 4)):
    xin = np.abs(x) + np.float32(1e-30) if which == "log" else x
    ref = oracle.detmath_probe(xin, which)
    xd = torch.from_numpy(xin).to(dev); out=torch.empty_like(xd)
    st = hip_lib.bgk_detmath_probe(_lib.ptr(xd), xd.numel(), code, _lib.ptr(out), _lib.stream_ptr(dev))
    got=out.cpu().numpy()
    bad=np.nonzero(got.view(np.uint32)!=ref.view(np.uint32))[0]
    print(which, len(bad), [(float(xin[i]), float(got[i]), float(ref[i])) for i in bad[:8]])
