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
        assert getattr(net, 'split_disabled', False)
        redo = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=6)     # the repeated work: valid, without hand-offs
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
        _lib.gemm_range_take()
        net.take_health()
        _lib.check(_lib.load().bsg_gemm_set_split(1), 'bsg_gemm_set_split')
