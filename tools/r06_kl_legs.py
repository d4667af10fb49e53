"""round 6: the KL-step legs of cfg 2 / cfg 5 alone (bench.py's kl_side_leg), optionally with the A/B against the layer-by-layer path:
    python tools/r06_kl_legs.py [cfg2|cfg5|both] [steps]          (BGK_BENCH_AB=1: + the library-GEMM path of rounds 1 - 5)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
for nm, bt in (("cfg2", 1 << 20), ("cfg5", 1 << 18)):
    if which in (nm, "both"):
        print(json.dumps({nm: bench.kl_side_leg(nm, dev, bt, steps)}))
        torch.cuda.empty_cache()
