"""The E-step extras of bench.py alone: segment ratios at 35 states, the 100-state model (tehmm_wide_estep.hip.h).
usage: python tools/wide_estep_bench.py [ratios|wide ...] [--no-verify]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
import bench                                   # noqa: E402

which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["ratios", "wide"]
dev = torch.device("cuda:0")
for w in which:
    print(w, json.dumps(bench.estep_other_routes(w, torch, dev, verify="--no-verify" not in sys.argv)), flush=True)
