"""python tools/r05_w256_one.py H fused|layer [B] [reps]: one B|A spline coupling layer with hidden width H, one path only (for counter passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bgflow_amd import configs
from bgflow_amd.utils import hash_init_

H, path = int(sys.argv[1]), sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
dims = {"BONDS": 17, "ANGLES": 17, "TORSIONS": 17, "FIXED": 9}
circ = {"BONDS": False, "ANGLES": False, "TORSIONS": True, "FIXED": False}
slot = {f: i for i, f in enumerate(configs.IC_FIELDS)}
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, d, device=dev, generator=g) for d in (17, 17, 17, 9)]
layer = hash_init_(configs._spline_coupling("BONDS", "ANGLES", dims, circ, slot, hidden=(H, H))).to(dev)
layer.transformer.allow_fused = path == "fused"
with torch.no_grad():
    for _ in range(reps):
        layer(*xs)
torch.cuda.synchronize()
