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


@pytest.mark.parametrize('B,Tt,Tm,rows,ragged', [(3, 12, 64, slice(1, 2), True), (5, 30, 301, slice(2, 5), True), (64, 100, 200, slice(8, 16), False),
                                                 (8, 40, 120, slice(0, 8), False)])
def test_fs2_rank_rows_front(fs2, B, Tt, Tm, rows, ragged):
    """a rank's token-level front (bsg_fs2midi_encode_rows) — K / V of the ESM from EVERY row's lang ids (the ESM attends over the
    batch axis, common_layers.py:848-860), everything else on the rank's rows only — equals the same rows of the whole-batch front
    (bsg_fs2midi_encode), and the oracle with the whole-batch front; the encoder ran on the rank's token rows, not on the batch's."""
    inp = synth.synth_inputs(B, Tt, Tm, seed=11, ragged=ragged)
    inp['lang'] = np.random.RandomState(5).randint(0, 2, (B, Tt)).astype(np.int64)      # rows differ in lang: K / V differ by row
    d = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    full = fs2.encode(d['txt_tokens'], d['spk_embed'], predict_dur=True, **kw)
    assert fs2.last_rows()[0] == B * Tt
    part = fs2.encode(d['txt_tokens'], d['spk_embed'], predict_dur=True, rows=rows, **kw)
    nb = rows.stop - rows.start
    assert fs2.last_rows()[0] == nb * Tt, 'the rank front encoded rows it does not own'
    assert part['enc_out'].shape == (nb, Tt, 256)
    assert maxabs(part['enc_out'], full['enc_out'][rows]) <= 5e-6
    assert maxabs(part['dur_xs'], full['dur_xs'][rows]) <= 1e-5      # (the predictor's GEMM picks its tiles by the batch: another order of the same sums)
    assert torch.equal(part['dur'], full['dur'][rows])
    # the whole forward with rows: against the oracle's whole-batch front sliced to the rows
    got = fs2(d['txt_tokens'], d['mel2ph'], d['spk_embed'], None, None, None, None, infer=True, rows=rows, **kw)
    assert fs2.last_rows()[0] == nb * Tt
    want = ofs2.fs2_forward(cpu_sd(fs2, 'fs2.'), {k: torch.from_numpy(v) for k, v in inp.items()}, 'fs2.', rows=rows)
    assert maxabs(got['decoder_inp'], want['decoder_inp']) <= 1e-4
    assert maxabs(got['mel_out'], want['mel_out']) <= 2e-4
    # a front that ignored the other rows' lang would be wrong by far more than that (tests/test_dist_cpu.py shows the same on the oracle)
    alone = fs2.encode(d['txt_tokens'][rows], d['spk_embed'][rows], predict_dur=False, **{k: v[rows] for k, v in kw.items() if k != 'speechsing'})
    if B > nb:
        assert maxabs(alone['enc_out'], full['enc_out'][rows]) > 1e-3
    # predicted durations with rows: T is the batch's maximum, so the front runs on every row (documented) and is sliced
    pred = fs2(d['txt_tokens'], None, d['spk_embed'], None, None, None, None, infer=True, rows=rows, **kw)
    allp = fs2(d['txt_tokens'], None, d['spk_embed'], None, None, None, None, infer=True, **kw)
    assert torch.equal(pred['mel2ph'], allp['mel2ph'][rows]) and maxabs(pred['mel_out'], allp['mel_out'][rows]) <= 1e-5


def test_length_regulator_hand_case(gold):
    from bisinger_amd import _lib
    g = gold('fs2')
    dur = torch.tensor([[2, 2, 3, 0], [1, 0, 4, 2]]).cuda()
    txt = torch.tensor([[5, 6, 7, 0], [5, 6, 7, 8]]).cuda()
    out = torch.empty(2, 7, dtype=torch.long, device='cuda')
    _lib.check(_lib.load().bsg_length_regulator(_lib.ptr(dur), _lib.ptr(txt), _lib.ptr(out), 2, 4, 7, _lib.stream_ptr()), 'lr')
    assert np.array_equal(out.cpu().numpy(), g['lr.mel2ph'])
