#!/usr/bin/env python3
"""Host-side profile of a pass (cProfile over PN passes of tools/prof_rank.py's workload): where the Python between the launches goes."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = sys.argv[:1]
import bench  # noqa: E402
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = bench.build_model(dev)
B, W = int(os.environ.get('PB', 64)), int(os.environ.get('PW', 8))
wl = bench.Workload(model, dev, B, 1 if W > 1 else 0, W, emulate=W > 1)
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
n = int(os.environ.get('PN', 10))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(n):
    wl.step(10 + i)
torch.cuda.synchronize()
pr.disable()
print(f'{(time.perf_counter() - t0) / n * 1e3:.2f} ms per pass under cProfile')
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
