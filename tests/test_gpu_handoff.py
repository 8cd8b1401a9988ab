"""GPU: a hand-off that gives up never yields a bad tensor (VERDICT r01 item 4 / ADVICE r01), and the bound-condition cache
cannot be fooled by the allocator.

Small batches run channel-split launches: a tile is computed by 2 or 4 workgroups that exchange their z halves through
flags (csrc/diffnet.hip residual_split_kernel).  If a partner is not resident the consumer's bounded spin gives up and
the tile is garbage.  ``bsg_diffnet_debug_inject_giveup`` forces exactly that (consumers skip the wait and count a
give-up); the drop-ins must notice in the SAME call, switch the handle to one-workgroup-per-tile launches and repeat.
"""
import warnings

import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from tests.util import load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


def _model():
    use_config()
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    m = GaussianDiffusion(_Enc(), 80, DIFF_DECODERS[hparams['diff_decoder_type']](hparams), timesteps=100, K_step=100,
                          spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, 0, synth.DIFFNET_GAIN)
    return m.cuda().eval()


@pytest.mark.parametrize('B', [1, 16])     # T=320: 10 tiles (4-way split, one chain) / 160 tiles (two chains of pair-split launches)
def test_injected_giveup_self_heals_in_the_same_call(B):
    T = 320
    rs = np.random.RandomState(7)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    noise = T_(synth.synth_noise(8, B, 80, T, seed=3)).cuda()
    x0 = noise[0][:, None].contiguous()
    t = torch.full((B,), 42, device='cuda', dtype=torch.long)

    good = _model()
    net = good.denoise_fn
    assert net.uses_handoffs(B, T)
    want_eps = net(x0, t, cond).clone()
    want_x = good.sample(cond, x0.clone(), noise=noise[1:], n_steps=8).clone()
    assert net.handoff_timeouts() == 0
    # B=16 (80 tiles of 64 frames: a third of the CUs) runs the split-fp16 stack launch by default; its repeat after a give-up runs the
    # per-layer F(2,3) kernels — another rounding of the same sums (1e-5).  The channel-split forms (B=1) heal bit for bit.
    stack = net.last_path().startswith('stack')
    same = (lambda a, b: maxabs(a, b) <= 1e-5) if stack else torch.equal

    # (1) DiffNet.forward: inject into the 20 layer launches of one evaluation
    m = _model()
    net = m.denoise_fn
    net.prepare(cond)
    net.debug_inject_giveup(20)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = net(x0, t, cond).clone()
    assert same(got, want_eps), 'a tensor computed from a given-up hand-off left the call'
    if B == 1:          # (a single evaluation at 160 tiles runs the 16-wave one-workgroup-per-tile form: nothing to inject into)
        assert any('hand-offs gave up' in str(x.message) for x in w), 'the give-up went unnoticed'
        assert getattr(net, 'split_disabled', False) and not net.uses_handoffs(B, T)
    assert net.handoff_timeouts() == 0                      # the take reset the counter

    # (2) the sampler loop (in place on x): inject mid-way through a fresh handle's loop
    m = _model()
    net = m.denoise_fn
    net.prepare(cond)
    net.debug_inject_giveup(3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=8).clone()
    assert any('hand-offs gave up' in str(x.message) for x in w)
    assert same(got, want_x)
    # and the healed handle keeps producing the same bits without hand-offs
    again = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=8).clone()
    assert torch.equal(again, got) and net.handoff_timeouts() == 0


def test_injected_giveup_self_heals_f43_stack_launch():
    """B=16, T=1000 runs a stack launch (split-fp16 form by default, F(4,3) with BSG_H2=0; 256 tiles, neighbours exchange edges every layer).  An injected give-up
    (odd tiles do not publish, nobody waits) must be noticed in the same call; the repeat runs per-layer F(2,3) launches —
    another rounding, so the healed result agrees to 1e-5 instead of bit for bit — and the handle stays healed."""
    B, T = 16, 1000
    rs = np.random.RandomState(9)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    noise = T_(synth.synth_noise(3, B, 80, T, seed=4)).cuda()
    x0 = noise[0][:, None].contiguous()
    good = _model()
    want = good.sample(cond, x0.clone(), noise=noise[1:], n_steps=3).clone()
    assert good.denoise_fn.last_path().startswith(('stack_h2', 'stack_f43')) and good.denoise_fn.uses_handoffs(B, T)
    assert good.denoise_fn.handoff_timeouts() == 0
    m = _model()
    net = m.denoise_fn
    net.prepare(cond)
    net.debug_inject_giveup(1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=3).clone()
    assert any('hand-offs gave up' in str(x.message) for x in w), 'the give-up went unnoticed'
    assert maxabs(got, want) <= 1e-5, 'a tensor computed from a given-up hand-off left the call'
    assert not net.uses_handoffs(B, T) and not net.last_path().startswith('stack') and net.handoff_timeouts() == 0
    # the injection is a real fault: the same launch without the guard gives another result
    from bisinger_amd import _lib
    from ctypes import c_int32, byref
    bad = _model().denoise_fn
    t = torch.full((B,), 42, device='cuda', dtype=torch.long)
    ref = bad(x0, t, cond).clone()
    bad.debug_inject_giveup(1)
    eps = torch.empty(B, 80, T, device='cuda')
    _lib.check(_lib.load().bsg_diffnet_forward(bad._h, _lib.ptr(x0[:, 0].contiguous()), _lib.ptr(t), _lib.ptr(eps), B, T, _lib.stream_ptr()), 'fwd')
    torch.cuda.synchronize()
    n = c_int32()
    _lib.check(_lib.load().bsg_diffnet_handoff_take(bad._h, byref(n), _lib.stream_ptr()), 'take')
    assert n.value > 0 and maxabs(eps[:, None], ref) > 1e-4


def test_injection_really_corrupts_without_the_guard():
    """the fault injection is a real fault: with the check bypassed the result differs (otherwise the test above proves nothing)"""
    B, T = 1, 320
    rs = np.random.RandomState(7)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    x = T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda()
    t = torch.full((B,), 42, device='cuda', dtype=torch.long)
    m = _model()
    net = m.denoise_fn
    want = net(x, t, cond).clone()
    from bisinger_amd import _lib
    from ctypes import c_int32, byref
    net.debug_inject_giveup(20)
    eps = torch.empty(B, 80, T, device='cuda')
    xx = x[:, 0].contiguous()
    _lib.check(_lib.load().bsg_diffnet_forward(net._h, _lib.ptr(xx), _lib.ptr(t), _lib.ptr(eps), B, T, _lib.stream_ptr()), 'fwd')
    torch.cuda.synchronize()
    n = c_int32()
    _lib.check(_lib.load().bsg_diffnet_handoff_take(net._h, byref(n), _lib.stream_ptr()), 'take')
    assert n.value > 0
    # consumers that do not wait read a partner's z tile from the previous launch (or half-written): not the right result
    assert not torch.equal(eps[:, None], want)


def test_bound_condition_is_not_keyed_on_the_address():
    """two different same-shape conds allocated back to back (the caching allocator reuses the freed block): the second call
    must re-bind, not reuse the first utterance's hoisted conditioner term"""
    m = _model()
    net = m.denoise_fn
    B, T = 2, 96
    rs = np.random.RandomState(11)
    x = T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda()
    t = torch.tensor([5, 60], device='cuda')
    c1 = rs.standard_normal((B, 256, T)).astype(np.float32)
    c2 = rs.standard_normal((B, 256, T)).astype(np.float32)
    a = T_(c1).cuda()
    p1 = a.data_ptr()
    e1 = net(x, t, a).clone()
    del a
    b = T_(c2).cuda()                      # same size: typically lands at the same address, _version 0 again
    same_address = b.data_ptr() == p1
    e2 = net(x, t, b).clone()
    ref2 = _model().denoise_fn(x, t, T_(c2).cuda()).clone()
    assert torch.equal(e2, ref2), f'stale conditioner term reused (same address: {same_address})'
    assert not torch.equal(e1, e2)
    # the same tensor object, unmodified, IS reused (no second prepare) and modified in place is re-bound
    bound = net._bound
    e2b = net(x, t, b).clone()
    assert net._bound is bound and torch.equal(e2b, e2)
    b.mul_(0.5)
    e3 = net(x, t, b).clone()
    assert not torch.equal(e3, e2)
    # p_sample goes through the same check
    c = T_(c1).cuda()
    y1 = m.p_sample(x, torch.full((B,), 9, device='cuda'), c, noise=torch.zeros_like(x)).clone()
    del c
    c = T_(c2).cuda()
    y2 = m.p_sample(x, torch.full((B,), 9, device='cuda'), c, noise=torch.zeros_like(x)).clone()
    assert not torch.equal(y1, y2)


def test_demoted_handle_tries_handoffs_again_after_clean_calls():
    """a give-up demotes the handle to launches without hand-offs; that must EXPIRE.  After CLEAN_CALLS_TO_REENABLE guarded
    calls (a demoted handle launches nothing that can give up, so every one of them counts) the hand-off forms are back."""
    B, T = 1, 320
    rs = np.random.RandomState(5)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    x0 = T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda()
    t = torch.full((B,), 17, device='cuda', dtype=torch.long)
    m = _model()
    net = m.denoise_fn
    want = net(x0, t, cond).clone()
    path0 = net.last_path()
    assert net.uses_handoffs(B, T)
    net.debug_inject_giveup(20)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = net(x0, t, cond).clone()
    assert any('hand-offs gave up' in str(x.message) for x in w)
    assert getattr(net, 'split_disabled', False) and not net.uses_handoffs(B, T)
    assert maxabs(got, want) <= 1e-5
    net.debug_inject_giveup(0)          # (one part launch took one of the 20 injections: drop the rest)
    demoted_path = net.last_path()
    assert demoted_path != path0
    for i in range(net.CLEAN_CALLS_TO_REENABLE):
        assert getattr(net, 'split_disabled', False), f'came back after {i} calls'
        out = net(x0, t, cond)
    assert not getattr(net, 'split_disabled', False) and net.uses_handoffs(B, T)
    out = net(x0, t, cond)
    assert net.last_path() == path0 and torch.equal(out, want)
    assert net.handoff_timeouts() == 0


def test_deferred_guard_mode_raises_one_call_late():
    """hparams['guard_mode'] = 'deferred': no stream wait inside a guarded call; the health words are copied to pinned host memory behind the
    work, and a LATER call (the first one that finds the copy completed; check_deferred() waits for it) raises for the invalid result —
    after switching the fallback on, so that the repeated work is valid."""
    from bisinger_amd import _lib
    B, T = 1, 320
    rs = np.random.RandomState(6)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    noise = T_(synth.synth_noise(6, B, 80, T, seed=9)).cuda()
    x0 = noise[0][:, None].contiguous()
    m = _model()
    net = m.denoise_fn
    want = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6).clone()
    hparams['guard_mode'] = 'deferred'
    try:
        got = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6)
        assert torch.equal(got, want)
        net.check_deferred()                       # clean: nothing raised
        _lib.check_deferred(cond)
        net.debug_inject_giveup(3)
        bad = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6)      # invalid, and nobody waited for it
        with pytest.raises(_lib.BsgError, match='PREVIOUS call'):
            net.check_deferred()
        assert getattr(net, 'parts_disabled', False) and not getattr(net, 'split_disabled', False)   # first tier: the give-up came out of a part launch
        redo = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6)     # the repeated work: valid, on the one-workgroup-per-tile launch
        net.check_deferred()
        assert maxabs(redo, want) <= 1e-5
        del bad
        # a split-fp16 GEMM operand beyond the fp16 range (the conditioner projection of a huge cond): raised by the next call too
        big = cond * 1e4
        m.sample(big, x0.clone(), noise=noise[1:3], n_steps=2)
        torch.cuda.synchronize()
        with pytest.raises(_lib.BsgError, match='PREVIOUS call'):
            m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6)
        ok = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6)
        net.check_deferred()
        _lib.check_deferred(cond)
        assert maxabs(ok, want) <= 1e-5
    finally:
        hparams.pop('guard_mode', None)
        torch.cuda.synchronize()
        net.gemm_range_take()
        net.take_health()
        net.set_gemm_split(True)


WRAP_CHILD = r'''
import sys, json, torch, numpy as np
sys.path.insert(0, %r)
from tests.util import load_formula_weights, use_config
from bisinger_amd import synth
torch.set_grad_enabled(False)
use_config()
from bisinger_amd.diffnet import DiffNet
net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
rs = np.random.RandomState(3)
def inputs(B, T):
    return (torch.from_numpy(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda(),
            torch.full((B,), 37, device='cuda', dtype=torch.long),
            torch.from_numpy(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda())
P, S, P2 = inputs(1, 320), inputs(10, 900), [inputs(1, 320) for _ in range(4)]   # a part form (quads of 32-frame tiles) and a whole-tile stack form on ONE handle
ref_p = net(*P).clone(); path_p = net.last_path()
ref_s = net(*S).clone(); path_s = net.last_path()
ref_p2 = [net(*q).clone() for q in P2]           # OTHER inputs of the part shape: stale exchange slots hold other values than these need
net(*P)                                          # the part flags exist and hold small epochs
net.debug_set_epoch(2 ** 25 - 3)
a = net(*P).clone()                              # epoch 2^25 - 3: the part flags now hold values near 2^31
b = net(*S).clone()                              # epoch 2^25 - 2
c = net(*S).clone()                              # epoch 2^25 - 1: its last workgroup wraps the epoch to 1 and zeroes the flags — in a WHOLE-TILE launch
outs = [net(*q).clone() for q in P2]             # epochs 1 .. 4: part launches again, on other inputs
torch.cuda.synchronize()
print(json.dumps({'paths': [path_p, path_s], 'timeouts': net.handoff_timeouts(),
                  'pre': [bool(torch.equal(a, ref_p)), bool(torch.equal(b, ref_s)), bool(torch.equal(c, ref_s))],
                  'post': [bool(torch.equal(o, r)) for o, r in zip(outs, ref_p2)],
                  'post_err': [float((o - r).abs().max()) for o, r in zip(outs, ref_p2)]}))
'''


def test_epoch_wrap_inside_a_whole_tile_launch_zeroes_the_part_flags(tmp_path):
    """the device-side launch epoch wraps at 2^25 (about two hours of single-utterance serving) and the launch that
    wraps zeroes the hand-off flags.  Round 4 zeroed only the arrays the WRAPPING launch itself waits on: a wrap inside a whole-tile stack
    launch left the part forms' flags near 2^31, and the next part launch (epoch 1) saw every flag as published — consumers read their
    partners' parts before they were written.  Driven here through bsg_diffnet_debug_set_epoch: part and whole-tile shapes alternate on one
    handle across the wrap and every result must stay bit-identical to the one computed at small epochs.  Negative control: the same
    sequence with round 4's behaviour (BSG_DEBUG_WRAP_R04=1) is NOT required to fail — the race may go either way — but it is run and
    reported, so that the log shows what the fix is worth on this box."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode, env in (('fixed', {}), ('r04', {'BSG_DEBUG_WRAP_R04': '1'})):
        out = subprocess.run([sys.executable, '-c', WRAP_CHILD % root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads(out.stdout.strip().splitlines()[-1])
    print('epoch wrap:', res)
    r = res['fixed']
    assert r['paths'][0].startswith('stack_h2_quad') and r['paths'][1] in ('stack_h2', 'stack_h2q'), r['paths']
    assert all(r['pre']) and all(r['post']) and r['timeouts'] == 0, r


def test_xcc_mismatch_in_a_part_launch_falls_back_to_the_one_workgroup_stack_launch():
    """the part forms exchange z and the image among the workgroups of a tile with PLAIN stores, visible only inside one
    XCD's L2, and assume workgroup i -> XCD i mod 8.  Under another dispatch order the partners' flags never become visible: round 4 noticed
    that behind a full bounded spin (seconds) and then took EVERY hand-off launch off the handle (per-layer launches on the fp32 matrix pipe).
    Now every part publishes its XCC id with an agent-scope store first, the partners compare it before their first wait, and the host takes
    only the part forms off: the repeat runs the one-workgroup-per-tile stack launch.  The mismatch is injected (odd parts report another
    id; CU masks in the environment do not change the placement on this pool: tools/xcc_probe.hip prints 0 1 2 3 4 5 6 7 under every
    HSA_CU_MASK tried).  Asserted: the right result, within a second of extra wall time, from a launch whose name is not 'layer'."""
    import time
    B, T = 1, 320
    rs = np.random.RandomState(7)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    x = T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda()
    t = torch.full((B,), 42, device='cuda', dtype=torch.long)
    m = _model()
    net = m.denoise_fn
    want = net(x, t, cond).clone()
    assert net.last_path() == 'stack_h2_quad'
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        net(x, t, cond)
    torch.cuda.synchronize()
    clean = (time.perf_counter() - t0) / 3
    net.debug_inject_xcc(1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        t0 = time.perf_counter()
        got = net(x, t, cond).clone()
        torch.cuda.synchronize()
        took = time.perf_counter() - t0
    assert any('part launch' in str(v.message) for v in w), [str(v.message) for v in w]
    print(f'XCC mismatch: recovered in {took * 1e3:.1f} ms (a clean call: {clean * 1e3:.1f} ms); repeat ran {net.last_path()}')
    assert took - clean < 1.0, (took, clean)
    assert net.last_path() in ('stack_h2', 'stack_h2q') and getattr(net, 'parts_disabled', False) and not getattr(net, 'split_disabled', False)
    assert maxabs(got, want) <= 1e-5 and net.handoff_timeouts() == 0        # another launch form: agreement to rounding
    again = net(x, t, cond).clone()                                          # and the handle stays on the one-workgroup launch, quietly
    assert torch.equal(again, got) and net.last_path() in ('stack_h2', 'stack_h2q')
