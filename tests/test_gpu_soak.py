"""GPU soak: the launches that hand data between workgroups stay CORRECT when the GPU is shared with another process.

The stack launches (one workgroup per CU, neighbouring tiles exchange their edges through L2 every layer) and the channel-split
launches assume that every workgroup of a launch is resident; with a second process occupying CUs that can fail, a polling workgroup
gives up after a bounded spin, the launch reports it, and the Python drop-in repeats the evaluation without hand-offs before anything
is returned (bisinger_amd/diffnet.py DiffNet.guarded).  This test runs that situation for real: a second, fresh process keeps the chip
busy with large matrix products while this one repeats 100-step sampler passes at a stack-launch shape and a small-batch shape.
Required: every result equals the undisturbed reference (bit for bit while no hand-off gave up, to 1e-5 — the rounding of the
fallback kernels — after one did), nothing hangs, and the slowdown stays bounded.  Both children are started fresh; no process that
has touched the GPU is ever re-executed.  (Round 2's tools/soak_handoffs.py, promoted to a test: VERDICT r02 item 8b.)"""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOG = r'''
import sys, time, torch
torch.set_grad_enabled(False)
a = torch.randn(8192, 8192, device='cuda'); b = torch.randn(8192, 8192, device='cuda')
print('hog ready', flush=True)
t_end = time.time() + float(sys.argv[1])
n = 0
while time.time() < t_end:
    for _ in range(4):
        c = a @ b
    torch.cuda.synchronize()
    n += 4
print('hog done', n, flush=True)
'''

SOAK = r'''
import sys, json, time, warnings, torch
sys.path.insert(0, %r)
import bench
torch.set_grad_enabled(False)
m = bench.build_model(torch.device('cuda', 0))
shapes = [(16, 1000), (4, 500)]
refs, base = {}, {}
for B, T in shapes:                                   # undisturbed references and their time
    g = torch.Generator().manual_seed(B)
    cond = torch.randn(B, 256, T, generator=g).cuda(); x0 = torch.randn(B, 1, 80, T, generator=g).cuda()
    m.sample(cond, x0.clone(), seed=3)
    torch.cuda.synchronize(); t0 = time.time()
    refs[(B, T)] = (cond, x0, m.sample(cond, x0.clone(), seed=3).clone())
    torch.cuda.synchronize(); base[(B, T)] = time.time() - t0
# what a HEALED handle costs (VERDICT r03 item 9e), measured while the chip is still quiet: a give-up is injected, the call that sees it
# repeats itself without hand-offs (per-layer launches on the fp32 matrix pipe; after a give-up inside a PART launch: on the one-workgroup-per-tile stack launch) and the handle stays there for CLEAN_CALLS_TO_REENABLE calls
from bisinger_amd import _lib
net = m.denoise_fn
healed = {}
for (B, T), (cond, x0, ref) in refs.items():
    net.prepare(cond)
    net.debug_inject_giveup(1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.sample(cond, x0.clone(), seed=3)
    net.debug_inject_giveup(0)
    assert getattr(net, 'split_disabled', False) or getattr(net, 'parts_disabled', False)   # (a give-up inside a part launch takes only the part forms off: round 5)
    torch.cuda.synchronize(); t0 = time.time()
    got = m.sample(cond, x0.clone(), seed=3)
    torch.cuda.synchronize(); dt = time.time() - t0
    healed[f'B{B}'] = {'healed_ms_per_pass': round(dt * 1e3, 1), 'normal_ms_per_pass': round(base[(B, T)] * 1e3, 1),
                       'healed_over_normal': round(dt / base[(B, T)], 2), 'healed_path': net.last_path(),
                       'healed_max_abs_vs_normal': float((got - ref).abs().max())}
    _lib.check(_lib.load().bsg_diffnet_set_split(net._h, 1), 'set_split')      # back to the hand-off launches for the soak below
    _lib.check(_lib.load().bsg_diffnet_set_parts(net._h, 1), 'set_parts')
    net.split_disabled, net._clean_calls, net.parts_disabled = False, 0, False
print('refs ready', flush=True)
sys.stdin.readline()                                  # the parent starts the second process now
out = {'runs': 0, 'exact': 0, 'close': 0, 'wrong': 0, 'warnings': 0, 'worst_dev': 0.0, 'slowdown': 0.0, 'healed_state': healed}
t_end = time.time() + float(sys.argv[1])
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    while time.time() < t_end:
        for (B, T), (cond, x0, ref) in refs.items():
            torch.cuda.synchronize(); t0 = time.time()
            got = m.sample(cond, x0.clone(), seed=3)
            torch.cuda.synchronize(); dt = time.time() - t0
            dev = float((got - ref).abs().max())
            out['runs'] += 1
            out['exact'] += int(dev == 0.0)
            out['close'] += int(0.0 < dev <= 1e-5)
            out['wrong'] += int(not (dev <= 1e-5))
            out['worst_dev'] = max(out['worst_dev'], dev)
            out['slowdown'] = max(out['slowdown'], dt / base[(B, T)])
    out['warnings'] = len(w)
out['pending_timeouts'] = m.denoise_fn.handoff_timeouts()
print(json.dumps(out), flush=True)
''' % ROOT


def test_handoff_launches_stay_correct_beside_a_second_process():
    soak = subprocess.Popen([sys.executable, '-c', SOAK, '25'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        line = soak.stdout.readline()
        while line and 'refs ready' not in line:
            line = soak.stdout.readline()
        assert 'refs ready' in line, soak.stderr.read()[-2000:]
        hog = subprocess.Popen([sys.executable, '-c', HOG, '30'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            ready = hog.stdout.readline()
            assert 'hog ready' in ready, hog.stderr.read()[-2000:]
            soak.stdin.write('go\n')
            soak.stdin.flush()
            t0 = time.time()
            so, se = soak.communicate(timeout=420)
            ho, _ = hog.communicate(timeout=120)
        finally:
            if hog.poll() is None:
                hog.kill()
    finally:
        if soak.poll() is None:
            soak.kill()
    assert soak.returncode == 0, se[-3000:]
    res = json.loads(so.strip().splitlines()[-1])
    print(f'soak beside a second process ({time.time() - t0:.0f} s): {res}; {ho.strip().splitlines()[-1] if ho.strip() else ""}')
    assert res['runs'] >= 4 and res['wrong'] == 0, res
    assert all(v['healed_max_abs_vs_normal'] <= 1e-5 for v in res['healed_state'].values()), res['healed_state']
    assert res['pending_timeouts'] == 0            # every give-up was taken and healed inside the call that saw it
    # bounded: a healed pass costs a bounded spin + one repeat; besides, the chip is time-shared with the second process — measured 17x for the
    # one-launch-per-step default and ~200x for the 2000 per-layer launches per pass of the fp32-pipe fallback (BSG_H2=0), each of which
    # queues behind the other process's kernels.  The bar only has to tell "slow" from "stuck"
    assert res['slowdown'] <= 1000.0, res
