"""GPU: BASELINE.json's full sizes (B=16, T=1000 fp32; B=64, T=1000 bf16) through size-independent properties.

The oracle takes minutes at these sizes, so beyond one DiffNet evaluation checked against it directly the tests use
properties the domain offers:
  * locality   - 20 dilated-conv layers (dilations 1,2,4,8 x 5) see +-75 frames: perturbing x beyond that leaves eps
                 bit-identical, and rows of a batch never interact (this is what catches tiling / halo / tile-order bugs
                 at sizes where every tile class occurs: first, interior, last partial, all 8 XCD runs).  The F(4,3)
                 stack launch (the fp32 default at this size) computes quads of frames t, t+d, t+2d, t+3d from the six
                 inputs t-d .. t+4d; the terms outside an output's own +-d cancel in exact arithmetic, in fp32 they leave
                 rounding noise: bit-identity holds outside 4 x 75 frames, and between 75 and 300 the leak is < 1e-5;
  * permutation equivariance of the sampler over the rows of a batch (supplied noise permuted alike), bit-exact;
  * shard invariance - rows [r0, r1) generated alone (Philox noise indexed by global row) equal the same rows of the
                 unsharded run; same seed -> same bits, other seed -> other result.  Bit-exact when both runs take the same
                 GEMM1 form (always under BSG_WINO=1 BSG_H2=0); by default B=16 runs the split-fp16 stack launch (diffnet_h2.hip) and a
                 shard of 2 rows the F(2,3) kernels — two roundings of the same sums — and the rows agree to 1e-5 instead.
"""
import os

import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from oracle import diffnet as odn
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy
B, T = 16, 1000
RF = 75    # receptive field of the residual stack on each side: 5 cycles x (1 + 2 + 4 + 8)


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


@pytest.fixture(scope='module')
def model():
    use_config()
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    m = GaussianDiffusion(_Enc(), 80, DIFF_DECODERS[hparams['diff_decoder_type']](hparams), timesteps=100, K_step=100,
                          spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, 0, synth.DIFFNET_GAIN)
    m = m.cuda()
    yield m
    m.denoise_fn.set_compute('fp32')


@pytest.fixture(scope='module')
def data():
    rs = np.random.RandomState(21)
    return (T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32)).cuda(),
            T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda(),
            T_(rs.randint(0, 100, size=(B,)).astype(np.int64)).cuda())


def test_diffnet_full_size_vs_oracle(model, data):
    x, cond, t = data
    eps = model.denoise_fn(x, t, cond)
    want = odn.diffnet_forward(cpu_sd(model), x.cpu(), t.cpu(), cond.cpu(), 'denoise_fn.')
    assert maxabs(eps, want) <= 1e-4


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_locality_and_row_independence(model, data, mode):
    x, cond, t = data
    net = model.denoise_fn
    net.set_compute(mode)
    try:
        base = net(x, t, cond).clone()
        rf = 4 * RF if net.last_path() == 'stack_f43' else RF       # bit-exact locality radius (module docstring)
        # (a) frames >= 600 of row 3 perturbed: only row 3, frames >= 600 - RF may change
        x2 = x.clone()
        x2[3, :, :, 600:] += 1.0
        e2 = net(x2, t, cond).clone()
        keep = torch.ones(B, T, dtype=torch.bool, device='cuda')
        keep[3, 600 - rf:] = False
        same = (e2 == base).all(dim=(1, 2))              # [B, T]
        assert bool(same[keep].all()), 'a perturbation leaked outside the receptive field / into another row'
        assert not bool(same[3, 600:].all())
        assert maxabs(e2[3, :, :, :600 - RF], base[3, :, :, :600 - RF]) < 1e-5
        # (b) the same for the condition, at a tile boundary of both tilings (frame 64*k) and at the very first frames
        c2 = cond.clone()
        c2[7, :, :64] *= 0.5
        e3 = net(x, t, c2).clone()
        keep = torch.ones(B, T, dtype=torch.bool, device='cuda')
        keep[7, :64 + rf] = False
        same = (e3 == base).all(dim=(1, 2))
        assert bool(same[keep].all())
        assert not bool(same[7, :64].all())
        assert maxabs(e3[7, :, :, 64 + RF:], base[7, :, :, 64 + RF:]) < 1e-5
    finally:
        net.set_compute('fp32')


def test_sampler_permutation_equivariance(model, data):
    _, cond, _ = data
    n = 4
    noise = T_(synth.synth_noise(n, B, 80, T, seed=9)).cuda()
    perm = torch.from_numpy(np.random.RandomState(2).permutation(B)).cuda()
    a = model.sample(cond, noise[0][:, None].contiguous().clone(), noise=noise[1:], n_steps=n).clone()
    b = model.sample(cond[perm].contiguous(), noise[0][perm][:, None].contiguous().clone(), noise=noise[1:][:, perm].contiguous(),
                     n_steps=n).clone()
    assert torch.equal(a[perm], b)


def test_full_size_shard_invariance_and_seed(model):
    inp = synth.synth_inputs(B, T // 10, T, seed=1)
    d = {k: T_(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    model.K_step = 6                                   # 6 sampler steps keep the test short; every kernel of the loop runs
    try:
        run = lambda **k: model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], infer=True, **kw, **k)['mel_out'].clone()
        full = run(seed=5)
        full_path = model.denoise_fn.last_path()
        assert full.shape == (B, T, 80) and bool(torch.isfinite(full).all())
        assert torch.equal(full, run(seed=5))
        assert maxabs(full, run(seed=6)) > 1e-2
        for r0, r1 in ((0, 8), (8, 16), (5, 7)):
            part = run(seed=5, rows=slice(r0, r1))
            if full_path.startswith(('stack_f43', 'stack_h2')) and model.denoise_fn.last_path() != full_path:
                assert maxabs(part, full[r0:r1]) <= 1e-5, f'rows [{r0},{r1}) differ from the unsharded run'
            else:
                assert torch.equal(part, full[r0:r1]), f'rows [{r0},{r1}) differ from the unsharded run'
    finally:
        model.K_step = 100


def test_bf16_config_full_size():
    """BASELINE configs[2] shape (B=64, T=1000): finite, deterministic, and close to the fp32 configuration — a SELF-COMPARISON (HIP bf16 path
    against the HIP fp32 path); the independent check of the bf16 arithmetic at this size is one DiffNet evaluation against the emulating
    oracle, tests/test_gpu_configs.py::test_config2_bf16_full_size_vs_emulating_oracle"""
    use_config()
    from bisinger_amd.diffnet import DiffNet
    net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
    rs = np.random.RandomState(4)
    x = T_(rs.standard_normal((64, 1, 80, T)).astype(np.float32)).cuda()
    cond = T_(rs.standard_normal((64, 256, T)).astype(np.float32)).cuda()
    t = T_(rs.randint(0, 100, size=(64,)).astype(np.int64)).cuda()
    ref = net(x, t, cond).clone()
    net.set_compute('bf16')
    a = net(x, t, cond).clone()
    b = net(x, t, cond).clone()
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    dev = maxabs(a, ref)
    print(f'B=64 T=1000 eps, bf16 vs fp32: max abs {dev:.3e}')
    assert dev <= 2e-2


def test_split_launch_equals_regular_launch(model, data):
    """The kernels of the fp32 matrix pipe (DiffNet.set_split_fp16(False), what BSG_H2=0 selects for a whole process).  Small launches
    (<= 128 tiles) run a tile as a PAIR (or four) of workgroups with a z hand-off (residual_split_kernel); the arithmetic per output
    element is the same k-ordered chain as the one-workgroup-per-tile launch, so row 3 alone (B=1: split) must equal row 3 inside a B=8
    batch (256 tiles: one workgroup per tile) bit for bit, and no hand-off may have timed out.  B=16 fills the chip and runs the F(4,3)
    stack launch: another rounding of the same sums (1e-5)."""
    x, cond, t = data
    net = model.denoise_fn
    net.set_split_fp16(False)
    try:
        full = net(x, t, cond).clone()
        assert net.last_path() in ('stack_f43', 'layer')
        f43 = net.last_path() == 'stack_f43'
        one = net(x[3:4].contiguous(), t[3:4].contiguous(), cond[3:4].contiguous()).clone()
        assert net.last_path().startswith('split')
        if f43:
            assert maxabs(one[0], full[3]) <= 1e-5
        else:
            assert torch.equal(one[0], full[3])
        half = net(x[:8].contiguous(), t[:8].contiguous(), cond[:8].contiguous()).clone()      # 256 tiles of 32 frames, one workgroup per tile
        assert not net.last_path().startswith('stack')
        assert torch.equal(one[0], half[3])
        assert net.handoff_timeouts() == 0
    finally:
        net.set_split_fp16(True)
    # and the default form (split-fp16 stack launch, 32- or 64-frame tiles by batch size): a row does not depend on the batch around it
    a = net(x, t, cond).clone()
    b1 = net(x[3:4].contiguous(), t[3:4].contiguous(), cond[3:4].contiguous()).clone()
    # (under BSG_H2=0 — the fallback-matrix run of the whole suite, profiles/r03_fallback_suite.txt — the process has no split-fp16 launch)
    assert net.last_path() == ('stack_h2_quad' if os.environ.get('BSG_H2', '1') != '0' else net.last_path())   # one utterance: the quad form
    assert maxabs(b1[0], a[3]) <= 1e-5      # 32-frame against 64-frame tiles: the same sums, fp32 accumulation order alike
    assert maxabs(b1[0], one[0]) <= 1e-5
