"""GPU parity: FastSpeech2-MIDI (HIP, via the C ABI) against the goldens produced by the reference."""
import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from oracle import fs2 as ofs2
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


@pytest.fixture(scope='module')
def fs2():
    use_config()
    from bisinger_amd.fs2 import FastSpeech2MIDI
    m = FastSpeech2MIDI(_Enc(), 80)
    load_formula_weights(m, 0, prefix='fs2.')
    return m.cuda()


def _run(fs2, inp, with_mel2ph=True, skip_decoder=False):
    d = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    return fs2(d['txt_tokens'], d['mel2ph'] if with_mel2ph else None, d['spk_embed'], None, None, None, None,
               skip_decoder=skip_decoder, infer=True, **kw)


@pytest.mark.parametrize('tag,B,Tt,Tm,ragged', [('B2', 2, 12, 64, False), ('B3r', 3, 12, 64, True), ('B1', 1, 12, 64, False)])
def test_fs2_golden(fs2, gold, tag, B, Tt, Tm, ragged):
    g = gold('fs2')
    inp = synth.synth_inputs(B, Tt, Tm, seed=1, ragged=ragged)
    r = _run(fs2, inp)
    assert maxabs(r['decoder_inp'], g[f'{tag}.decoder_inp']) <= 5e-5
    assert maxabs(r['mel_out'], g[f'{tag}.mel_out']) <= 1e-4
    r = _run(fs2, inp, skip_decoder=True)
    assert 'mel_out' not in r and maxabs(r['decoder_inp'], g[f'{tag}.decoder_inp']) <= 5e-5
    # predicted durations: integer outputs must be identical
    r = _run(fs2, inp, with_mel2ph=False)
    assert np.array_equal(r['mel2ph'].cpu().numpy(), g[f'{tag}.pred.mel2ph'])
    assert maxabs(r['dur'], g[f'{tag}.pred.dur']) <= 5e-5
    assert maxabs(r['mel_out'], g[f'{tag}.pred.mel_out']) <= 1e-4


def test_fs2_larger_vs_oracle(fs2):
    """Bench-like shape (long rows exercise the multi-tile attention and the odd-T scalar paths)."""
    sd = cpu_sd(fs2, 'fs2.')
    for B, Tt, Tm, ragged in [(4, 30, 301, True), (2, 100, 1000, False)]:
        inp = synth.synth_inputs(B, Tt, Tm, seed=5, ragged=ragged)
        want = ofs2.fs2_forward(sd, {k: torch.from_numpy(v) for k, v in inp.items()})
        got = _run(fs2, inp)
        assert maxabs(got['decoder_inp'], want['decoder_inp']) <= 1e-4
        assert maxabs(got['mel_out'], want['mel_out']) <= 2e-4


def test_length_regulator_hand_case(gold):
    from bisinger_amd import _lib
    g = gold('fs2')
    dur = torch.tensor([[2, 2, 3, 0], [1, 0, 4, 2]]).cuda()
    txt = torch.tensor([[5, 6, 7, 0], [5, 6, 7, 8]]).cuda()
    out = torch.empty(2, 7, dtype=torch.long, device='cuda')
    _lib.check(_lib.load().bsg_length_regulator(_lib.ptr(dur), _lib.ptr(txt), _lib.ptr(out), 2, 4, 7, _lib.stream_ptr()), 'lr')
    assert np.array_equal(out.cpu().numpy(), g['lr.mel2ph'])
