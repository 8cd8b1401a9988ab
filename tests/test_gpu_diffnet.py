"""GPU parity: the HIP DiffNet / samplers (through the C ABI) against the oracle and the goldens."""
import os

import numpy as np
import pytest
import torch

from bisinger_amd import _lib, synth
from oracle import diffnet as odn, diffusion as odf
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


@pytest.fixture(scope='module')
def net():
    use_config()
    from bisinger_amd.diffnet import DiffNet
    m = DiffNet(80)
    load_formula_weights(m, 0, synth.DIFFNET_GAIN, prefix='denoise_fn.')
    return m.cuda()


def test_library_is_native():
    lib = _lib.load()
    assert lib.bsg_device_arch().decode().startswith('gfx950')


@pytest.mark.parametrize('M,N,K,trans_b,batch', [(128, 128, 32, 0, 1), (256, 1000, 80, 0, 3), (80, 77, 256, 0, 2),
                                                 (300, 768, 256, 1, 1), (1000, 1000, 128, 1, 4), (7, 5, 3, 1, 2),
                                                 (64, 200, 1024, 0, 1)])
def test_gemm_f32(M, N, K, trans_b, batch):
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((batch, M, K)).astype(np.float32)
    Bm = rs.standard_normal((batch, N, K) if trans_b else (batch, K, N)).astype(np.float32)
    bias = rs.standard_normal(M).astype(np.float32)
    ref = np.einsum('bmk,bnk->bmn' if trans_b else 'bmk,bkn->bmn', A.astype(np.float64), Bm.astype(np.float64)) + bias[None, :, None]
    ref = np.maximum(ref, 0)
    a, b, bi = T_(A).cuda(), T_(Bm).cuda(), T_(bias).cuda()
    c = torch.full((batch, M, N), float('nan'), device='cuda')
    lib = _lib.load()
    _lib.check(lib.bsg_gemm_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(bi), None, M, N, K, K, K if trans_b else N, N,
                                trans_b, batch, M * K, N * K, M * N, 1, _lib.stream_ptr()), 'gemm')
    torch.cuda.synchronize()
    assert maxabs(c, ref) <= 2e-6 * K ** 0.5 * 4 + 1e-5


@pytest.mark.parametrize('rows,Wn,K,taps,act_is_a,batch', [(128, 128, 64, 1, 1, 1), (1000, 768, 256, 1, 1, 1), (333, 1024, 256, 9, 1, 3),
                                                           (77, 256, 1024, 1, 1, 2), (1000, 512, 256, 1, 0, 5), (50, 256, 256, 3, 1, 4),
                                                           (130, 128, 128, 17, 0, 2), (1, 128, 64, 5, 1, 2)])
def test_gemm_presplit(rows, Wn, K, taps, act_is_a, batch):
    """gemm_h2w_kernel (pre-split weights, activation planes; FS2 linears, the k-tap Conv1d FFN with SAME padding per batch item, and the
    [feature][frame] form of the conditioner projections) against float64: ragged row counts, both tile heights, both output layouts."""
    rs = np.random.RandomState(rows + Wn + K + taps)
    A = rs.standard_normal((batch, rows, K)).astype(np.float32)
    W = rs.standard_normal((taps, Wn, K)).astype(np.float32)
    bias = rs.standard_normal(Wn).astype(np.float32)
    pad = taps // 2
    Ap = np.pad(A.astype(np.float64), ((0, 0), (pad, pad), (0, 0)))
    ref = sum(np.einsum('brk,nk->brn', Ap[:, t:t + rows], W[t].astype(np.float64)) for t in range(taps)) + bias[None, None, :]
    ref = np.maximum(ref, 0)
    if not act_is_a:
        ref = ref.transpose(0, 2, 1)
    a, w, bi = T_(A).cuda(), T_(W).cuda(), T_(bias).cuda()
    c = torch.full(ref.shape, float('nan'), device='cuda')
    lib = _lib.load()
    _lib.check(lib.bsg_gemm_presplit_f32(_lib.ptr(a), _lib.ptr(w), _lib.ptr(c), _lib.ptr(bi), rows, Wn, K, taps, act_is_a, batch, 1, 1,
                                         _lib.stream_ptr()), 'gemm_presplit')
    torch.cuda.synchronize()
    assert maxabs(c, ref) <= 2e-6 * (K * taps) ** 0.5 * 4 + 1e-5
    assert _lib.gemm_range_take() == 0


@pytest.mark.parametrize('B,T', [(2, 64), (3, 77), (1, 31), (5, 333), (2, 1000)])
@pytest.mark.parametrize('layer', [0, 3, 19])
def test_residual_layer(net, B, T, layer):
    """One fused ResidualBlock vs oracle.residual_block (net.py:66-78): vector (T%4==0) and scalar staging,
    partial last tiles, every dilation class, first-layer store and last-layer scaling of the skip sum."""
    sd = cpu_sd(net, 'denoise_fn.')
    rs = np.random.RandomState(100 * B + T + layer)
    x = rs.standard_normal((B, 256, T)).astype(np.float32)
    cond = rs.standard_normal((B, 256, T)).astype(np.float32)
    skip0 = rs.standard_normal((B, 256, T)).astype(np.float32)
    t = rs.randint(0, 100, size=(B,)).astype(np.int64)
    d = odn.step_embedding(sd, T_(t), 256, 'denoise_fn.')
    rx, rskip = odn.residual_block(sd, f'denoise_fn.residual_layers.{layer}.', T_(x), T_(cond), d, 2 ** (layer % 4))
    want_skip = rskip if layer == 0 else T_(skip0) + rskip
    if layer == 19:
        want_skip = want_skip / 20 ** 0.5
    net.prepare(T_(cond).cuda())
    skip = T_(skip0).cuda()
    if layer == 0:
        skip.fill_(float('nan'))          # the first layer must store, never read, the skip buffer
    out = net.residual_layer(layer, T_(x).cuda(), T_(t).cuda(), skip)
    torch.cuda.synchronize()
    assert maxabs(out, rx) <= 2e-5
    assert maxabs(skip, want_skip) <= 2e-5


def test_diffnet_forward_golden(net, gold):
    g = gold('diffnet')
    rs = np.random.RandomState(11)
    x = rs.standard_normal((2, 1, 80, 64)).astype(np.float32)
    cond = rs.standard_normal((2, 256, 64)).astype(np.float32)
    t = np.array([7, 93], np.int64)
    eps = net(T_(x).cuda(), T_(t).cuda(), T_(cond).cuda())
    assert eps.shape == (2, 1, 80, 64)
    assert maxabs(eps, g['eps_B2T64']) <= 1e-4
    rs = np.random.RandomState(12)
    x2 = rs.standard_normal((3, 1, 80, 77)).astype(np.float32)
    c2 = rs.standard_normal((3, 256, 77)).astype(np.float32)
    t2 = np.array([0, 50, 99], np.int64)
    eps2 = net(T_(x2).cuda(), T_(t2).cuda(), T_(c2).cuda())
    assert maxabs(eps2, g['eps_B3T77']) <= 1e-4


def test_diffnet_forward_vs_fp64_oracle(net):
    """At a size that uses the wide tiles: error vs the float64 oracle is rounding-level."""
    sd = cpu_sd(net, 'denoise_fn.')
    rs = np.random.RandomState(5)
    B, T = 8, 300
    x = rs.standard_normal((B, 1, 80, T)).astype(np.float32)
    cond = rs.standard_normal((B, 256, T)).astype(np.float32)
    t = rs.randint(0, 100, size=(B,)).astype(np.int64)
    eps = net(T_(x).cuda(), T_(t).cuda(), T_(cond).cuda())
    ref64 = odn.diffnet_forward(sd, T_(x), T_(t), T_(cond), 'denoise_fn.', dtype=torch.float64)
    ref32 = odn.diffnet_forward(sd, T_(x), T_(t), T_(cond), 'denoise_fn.')
    e_hip, e_cpu = maxabs(eps, ref64), maxabs(ref32, ref64)
    assert e_hip <= 2e-5, (e_hip, e_cpu)
    assert e_hip <= 10 * e_cpu + 1e-6, (e_hip, e_cpu)


def test_philox_stream_matches_host():
    lib = _lib.load()
    x = torch.empty(4096, device='cuda')
    _lib.check(lib.bsg_philox_normal(_lib.ptr(x), 4096, 1234, 7, 1024, _lib.stream_ptr()), 'philox')
    host = synth.philox_normal(1234, 7, 1024 + 4096)[1024:]
    assert maxabs(x, host) <= 5e-6
