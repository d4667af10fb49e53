import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bgflow_amd import configs
from bgflow_amd.utils import hash_init_, synth
from oracle import flow_oracle as fo
dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
layer_cpu = hash_init_(configs._spline_coupling("BONDS", "ANGLES", dims, circ, slot))
layer = hash_init_(configs._spline_coupling("BONDS", "ANGLES", dims, circ, slot)).to(dev)
xs = [synth(B + 7 * i, B, d, uniform=True) for i, d in enumerate((17, 17, 17, 9))]
with torch.no_grad():
    *outs, dl = layer(*[torch.as_tensor(v).to(dev) for v in xs])
fo.MFMA_ORDER = True
outs_o, dl_o = fo.run_block(layer_cpu, xs, False, np.float32, None)
got = outs[0].cpu().numpy(); ref = outs_o[0]
bad = np.abs(got - ref).max(1) > 1e-6
print("bad samples:", np.nonzero(bad)[0][:40], "count", bad.sum(), "of", B)
badd = np.abs(got - ref) > 1e-6
print("bad dims histogram:", badd.sum(0))
print("dlogp err max", np.abs(dl.cpu().numpy() - dl_o).max())
