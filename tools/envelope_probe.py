"""bench.py's fallback envelope alone (one 60 Mb chromosome under -NOGC, -m 201, 300x): ms, per-base kernel times.  Twice: the first
pass warms the pool's workspaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsicnv_amd import api
lib = api.load_library()
pool = api.RsiPool(0, 16)
bench.fallback_envelope(lib, pool, torch.device("cuda", 0))
for k, v in bench.fallback_envelope(lib, pool, torch.device("cuda", 0)).items():
    print(k, v, flush=True)
