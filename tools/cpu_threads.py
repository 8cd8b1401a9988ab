import sys, time, os, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from collections import OrderedDict
import json
from bisinger_amd import synth
from oracle import diffnet as odn
torch.set_grad_enabled(False)
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
spec = OrderedDict((k, tuple(s)) for k, s in json.load(open(root + '/tests/golden/state_dict_spec.json'))['GaussianDiffusion'] if k.startswith('denoise_fn.'))
sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 0).items()}
B, T = 16, 1000
x = torch.randn(B, 1, 80, T); cond = torch.randn(B, 256, T); t = torch.full((B,), 50)
print('cpu_count', os.cpu_count())
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.')
    t0 = time.perf_counter()
    for _ in range(2): odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.')
    dt = (time.perf_counter() - t0) / 2
    print(f'threads {n}: {dt:.3f} s per DiffNet call -> {B*T/(dt*100):.0f} frames/s @100 steps', flush=True)
