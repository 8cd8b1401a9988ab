"""CPU: the oracle (oracle/) against every golden vector produced by the reference itself
(tools/make_golden.py).  This is what pins the oracle."""
import hashlib

import numpy as np
import torch

from bisinger_amd import synth
from oracle import diffnet as odn, diffusion as odf, fs2 as ofs2, hifigan as ohg, melgen as omg

torch.set_grad_enabled(False)
T = torch.from_numpy


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def tin(d):
    return {k: T(v) for k, v in d.items()}


def test_schedules_bit_exact(gold):
    g = gold('schedules')
    for tag, kw in {'lin100_006': dict(timesteps=100, schedule_type='linear', max_beta=0.06),
                    'lin1000_002': dict(timesteps=1000, schedule_type='linear', max_beta=0.02),
                    'cos100': dict(timesteps=100, schedule_type='cosine')}.items():
        sch = odf.make_schedule(**kw)
        assert len(sch) == 12
        for k, v in sch.items():
            assert np.array_equal(v.numpy(), g[f'{tag}.{k}']), (tag, k)


def test_formula_weights_are_stable(gold, gd_sd):
    g = gold('diffnet')
    assert sha(*[gd_sd[k].numpy() for k in gd_sd if k.startswith('denoise_fn.')]) == str(g['sha_w'])


def test_diffnet(gold, gd_sd):
    g = gold('diffnet')
    rs = np.random.RandomState(11)
    x = rs.standard_normal((2, 1, 80, 64)).astype(np.float32)
    cond = rs.standard_normal((2, 256, 64)).astype(np.float32)
    t = np.array([7, 93], np.int64)
    eps = odn.diffnet_forward(gd_sd, T(x), T(t), T(cond), 'denoise_fn.')
    assert (eps.numpy() - g['eps_B2T64']).__abs__().max() <= 2e-6
    d = odn.step_embedding(gd_sd, T(t), 256, 'denoise_fn.')
    assert np.abs(d.numpy() - g['step_emb']).max() <= 1e-6
    rx, rs_ = odn.residual_block(gd_sd, 'denoise_fn.residual_layers.3.', T(g['rb3_in']), T(cond), d, 8)
    assert np.abs(rx.numpy() - g['rb3_x']).max() <= 2e-6
    assert np.abs(rs_.numpy() - g['rb3_skip']).max() <= 2e-6
    rs = np.random.RandomState(12)
    x2 = rs.standard_normal((3, 1, 80, 77)).astype(np.float32)
    c2 = rs.standard_normal((3, 256, 77)).astype(np.float32)
    t2 = np.array([0, 50, 99], np.int64)
    assert sha(x, cond, t, x2, c2, t2) == str(g['sha_in'])
    eps2 = odn.diffnet_forward(gd_sd, T(x2), T(t2), T(c2), 'denoise_fn.')
    assert np.abs(eps2.numpy() - g['eps_B3T77']).max() <= 2e-6


def test_sampler_trajectory(gold, gd_sd):
    g = gold('sampler')
    rs = np.random.RandomState(11)
    rs.standard_normal((2, 1, 80, 64))
    cond = T(rs.standard_normal((2, 256, 64)).astype(np.float32))
    noise = synth.synth_noise(100, 2, 80, 64, seed=1)
    assert sha(noise) == str(g['sha_noise'])
    sch = odf.make_schedule(100, 'linear', 0.06)
    den = lambda x_, t_: odn.diffnet_forward(gd_sd, x_, t_, cond, 'denoise_fn.')
    x1 = odf.p_sample(sch, den, T(noise[0][:, None]), torch.full((2,), 99, dtype=torch.long), T(noise[1][:, None]))
    assert np.abs(x1.numpy() - g['p_sample_t99']).max() <= 5e-6
    x0 = odf.ddpm_sample(sch, den, T(noise[0][:, None]), T(noise[1:][:, :, None]), 100)
    assert np.abs(x0.numpy() - g['x_t0']).max() <= 2e-5
    # the fp64 trajectory stays next to the reference's fp32 one: the synthetic denoiser is
    # well-conditioned, so 1e-3 on the mel is a meaningful bar for the HIP path
    den64 = lambda x_, t_: odn.diffnet_forward(gd_sd, x_, t_, cond.double(), 'denoise_fn.', dtype=torch.float64)
    x64 = odf.ddpm_sample(sch, den64, T(noise[0][:, None]).double(), T(noise[1:][:, :, None]).double(), 100)
    assert np.abs(x64.numpy() - g['x_t0']).max() <= 5e-5


def test_fs2(gold, gd_sd):
    g = gold('fs2')
    for tag, (B, Tt, Tm, ragged) in {'B2': (2, 12, 64, False), 'B3r': (3, 12, 64, True), 'B1': (1, 12, 64, False)}.items():
        inp = tin(synth.synth_inputs(B, Tt, Tm, seed=1, ragged=ragged))
        r = ofs2.fs2_forward(gd_sd, inp)
        assert np.abs(r['decoder_inp'].numpy() - g[f'{tag}.decoder_inp']).max() <= 5e-6, tag
        assert np.abs(r['mel_out'].numpy() - g[f'{tag}.mel_out']).max() <= 1e-5, tag
        inp.pop('mel2ph')
        r = ofs2.fs2_forward(gd_sd, inp)
        assert np.array_equal(r['mel2ph'].numpy(), g[f'{tag}.pred.mel2ph']), tag
        assert np.abs(r['dur'].numpy() - g[f'{tag}.pred.dur']).max() <= 1e-5
        assert np.abs(r['mel_out'].numpy() - g[f'{tag}.pred.mel_out']).max() <= 1e-5, tag


def test_esm_couples_batch_rows(gd_sd):
    """Reference quirk (common_layers.py:853): ESM attends over the batch axis."""
    inp2 = tin(synth.synth_inputs(2, 12, 64, seed=1))
    inp1 = {k: v[:1] for k, v in inp2.items()}
    a = ofs2.fs2_forward(gd_sd, inp2, skip_decoder=True)['decoder_inp'][0]
    b = ofs2.fs2_forward(gd_sd, inp1, skip_decoder=True)['decoder_inp'][0]
    assert (a - b).abs().max() > 1e-2


def test_enc_sa_layer_and_length_regulator(gold, gd_sd):
    g = gold('fs2')
    rs = np.random.RandomState(13)
    xe = rs.standard_normal((10, 2, 256)).astype(np.float32)
    pm = np.zeros((2, 10), bool)
    pm[1, 6:] = True
    y = ofs2.enc_sa_layer(gd_sd, 'fs2.decoder.layers.1.op.', T(xe), T(pm), 2, 9, torch.float32)
    assert np.abs(y.numpy() - g['encsa.y']).max() <= 2e-6
    dur = torch.tensor([[2, 2, 3, 0], [1, 0, 4, 2]])
    dpad = torch.tensor([[False, False, False, True], [False, False, False, False]])
    assert np.array_equal(ofs2.length_regulator(dur, dpad).numpy(), g['lr.mel2ph'])
    # docstring example of the reference (tts_modules.py:163-174)
    assert ofs2.length_regulator(torch.tensor([[2, 2, 3]]), None).tolist() == [[1, 1, 2, 2, 3, 3, 3]]


def test_melgen_end_to_end(gold, gd_sd):
    g = gold('melgen')
    for tag, (B, Tt, Tm, ragged) in {'B2': (2, 12, 64, False), 'B3r': (3, 10, 50, True)}.items():
        inp = tin(synth.synth_inputs(B, Tt, Tm, seed=2, ragged=ragged))
        noise = T(synth.synth_noise(100, B, 80, Tm, seed=3))
        r = omg.mel_gen(gd_sd, inp, noise)
        assert np.abs(r['mel_out'].numpy() - g[f'{tag}.mel_out']).max() <= 1e-4, tag
    inp = tin(synth.synth_inputs(2, 12, 64, seed=2))
    r = omg.mel_gen(gd_sd, inp, T(synth.synth_noise(51, 2, 80, 64, seed=4)), K_step=51, gaussian_start=False)
    assert np.abs(r['mel_out'].numpy() - g['shallow51.mel_out']).max() <= 1e-4
    inp = tin(synth.synth_inputs(1, 12, 64, seed=2))
    r = omg.mel_gen(gd_sd, inp, T(synth.synth_noise(100, 1, 80, 64, seed=5)), pndm_speedup=5)
    assert np.abs(r['mel_out'].numpy() - g['plms5.mel_out']).max() <= 2e-4


def test_hifigan(gold, hifigan_sd, sd_spec):
    g = gold('hifigan')
    rs = np.random.RandomState(21)
    for tag, (B, Th) in {'B1T16': (1, 16), 'B2T37': (2, 37)}.items():
        mel = (rs.standard_normal((B, 80, Th)) * 1.5 - 3.0).astype(np.float32)
        assert sha(mel) == str(g[f'{tag}.sha_in'])
        y = ohg.hifigan_forward(hifigan_sd, T(mel), sd_spec['hifigan_cfg'])
        assert y.shape == (B, 1, Th * 256)
        assert np.abs(y.numpy() - g[f'{tag}.wav']).max() <= 5e-6
        y2 = ohg.hifigan_forward(ohg.fold_weight_norm(hifigan_sd), T(mel), sd_spec['hifigan_cfg'])
        assert np.abs(y2.numpy() - g[f'{tag}.wav']).max() <= 5e-6


def test_philox_stream_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): ctr=0,key=0 and the pi digits case."""
    z = synth.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros((1, 2), np.uint32))[0]
    assert [hex(int(v)) for v in z] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    c = np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32)
    k = np.array([[0xa4093822, 0x299f31d0]], np.uint32)
    z = synth.philox4x32_10(c, k)[0]
    assert [hex(int(v)) for v in z] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']
    n = synth.philox_normal(1, 5, 100000)
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1) < 0.02


def _bn_fix(w, spec):
    for k in spec:
        if k.endswith('running_var'):
            w[k] = (0.5 + np.abs(w[k]) * 5).astype(np.float32)
        if k.endswith('running_mean'):
            w[k] = (w[k] * 3).astype(np.float32)
    return w


def test_f2_pitch_extractor_and_nsf(gold, sd_spec):
    """SURVEY.md §8 row f2 goldens (tools/make_golden_f2.py)."""
    from collections import OrderedDict
    from oracle import nsf as onsf, pe as ope
    g = gold('f2')
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['PitchExtractor'])
    sd = {k: T(v) for k, v in _bn_fix(synth.synth_state_dict(spec, seed=11), spec).items()}
    rs = np.random.RandomState(31)
    for tag, (B, Tn) in {'B2T50': (2, 50), 'B1T133': (1, 133)}.items():
        mel = (rs.standard_normal((B, Tn, 80)) * 1.5 - 3.0).astype(np.float32)
        if B > 1:
            mel[1, Tn - 7:] = 0
        r = ope.pitch_extractor_forward(sd, T(mel))
        assert np.abs(r['pitch_pred'].numpy() - g[f'pe.{tag}.pitch_pred']).max() <= 1e-5
        assert np.abs(r['f0_denorm_pred'].numpy() - g[f'pe.{tag}.f0']).max() <= 1e-4
    nspec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_nsf_weight_norm'])
    nsd = {k: T(v) for k, v in synth.synth_state_dict(nspec, seed=13).items()}
    cfg = dict(sd_spec['hifigan_cfg'], use_pitch_embed=True)
    for tag, (B, Tn) in {'B1T12': (1, 12), 'B2T21': (2, 21)}.items():
        mel = (rs.standard_normal((B, 80, Tn)) * 1.5 - 3.0).astype(np.float32)
        f0 = (rs.uniform(90, 500, size=(B, Tn))).astype(np.float32)
        f0[:, Tn // 3: Tn // 3 + 3] = 0
        ri = rs.uniform(0, 1, size=(B, 9)).astype(np.float32)
        nz = rs.standard_normal((B, Tn * 256, 9)).astype(np.float32)
        har = onsf.sine_source(nsd, T(f0), T(ri), T(nz), cfg['audio_sample_rate'], 256)
        assert np.abs(har.numpy() - g[f'nsf.{tag}.har']).max() <= 1e-5
        y = onsf.nsf_hifigan_forward(nsd, T(mel), T(f0), T(ri), T(nz), cfg)
        assert np.abs(y.numpy() - g[f'nsf.{tag}.wav']).max() <= 1e-5


def test_f4_fft_candidate_denoiser(gold, sd_spec):
    """SURVEY.md §8 row f4 golden (tools/make_golden_f4.py)."""
    from collections import OrderedDict
    from oracle import candidate_decoder as ocd
    g = gold('f4')
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['FFT'])
    sd = {k: T(v) for k, v in synth.synth_state_dict(spec, seed=17).items()}
    rs = np.random.RandomState(41)
    for tag, (B, Tn) in {'B2T40': (2, 40), 'B1T77': (1, 77)}.items():
        x = rs.standard_normal((B, 1, 80, Tn)).astype(np.float32)
        cond = rs.standard_normal((B, 256, Tn)).astype(np.float32)
        t = rs.randint(0, 100, size=(B,)).astype(np.int64)
        y = ocd.fft_denoiser_forward(sd, T(x), T(t), T(cond))
        assert np.abs(y.numpy() - g[f'{tag}.eps']).max() <= 1e-5


def test_oracle_bf16_operand_emulation_is_a_small_perturbation():
    """oracle.diffnet.residual_block(operand_bf16=True) is the checker of the build's bf16-operand configuration
    (tests/test_gpu_bf16.py): deterministic, and within the bf16 rounding budget of the fp32 restatement."""
    import torch
    from oracle import diffnet as odn
    g = torch.Generator().manual_seed(3)
    C = 256
    sd = {'dilated_conv.weight': torch.randn(2 * C, C, 3, generator=g) * 0.03, 'dilated_conv.bias': torch.randn(2 * C, generator=g) * 0.1,
          'diffusion_projection.weight': torch.randn(C, C, generator=g) * 0.05, 'diffusion_projection.bias': torch.zeros(C),
          'conditioner_projection.weight': torch.randn(2 * C, C, 1, generator=g) * 0.05, 'conditioner_projection.bias': torch.zeros(2 * C),
          'output_projection.weight': torch.randn(2 * C, C, 1, generator=g) * 0.05, 'output_projection.bias': torch.zeros(2 * C)}
    x, cond, d = torch.randn(2, C, 50, generator=g), torch.randn(2, C, 50, generator=g), torch.randn(2, C, generator=g)
    with torch.no_grad():
        a = odn.residual_block(sd, '', x, cond, d, 4, operand_bf16=True)
        b = odn.residual_block(sd, '', x, cond, d, 4, operand_bf16=True)
        f = odn.residual_block(sd, '', x, cond, d, 4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for u, v in zip(a, f):
        dev = float((u - v).abs().max())
        assert 1e-5 < dev < 3e-2, dev


# ---- BASELINE's stated configurations at their stated sizes (tools/make_golden_cfg.py) -------------------------------
def _cfg_oracle(c, tag, gold, sd_spec, gd_sd):
    from tests.cfg_fixtures import inputs_and_noise, oracle_state_dict
    g = gold('cfg')
    inp, noise = inputs_and_noise(c)
    assert sha(*[inp[k] for k in sorted(inp)], noise) == str(g[f'{tag}.sha_in'])
    sd = oracle_state_dict(c, sd_spec, gd_sd['spec_min'], gd_sd['spec_max'])
    out = omg.mel_gen(sd, tin(inp), T(noise), timesteps=c['timesteps'], K_step=c['K_step'], max_beta=c['max_beta'],
                      pndm_speedup=c['pndm_speedup'])
    return float(np.abs(out['mel_out'].numpy() - g[f'{tag}.mel_out']).max())


def test_config0_single_utterance_T500(gold, sd_spec, gd_sd):
    """BASELINE configs[0] / SURVEY §8(d) Config 1: B=1, T_txt=50, T=500, FS2 + 100-step DDPM; oracle vs reference <= 1e-4"""
    from tests.cfg_fixtures import CFG0
    assert _cfg_oracle(CFG0, 'cfg0', gold, sd_spec, gd_sd) <= 1e-4


def test_shipped_config_plms1000(gold, sd_spec, gd_sd):
    """the shipped sampler (1000-step schedule to beta 0.02, PLMS interval 5: 201 evaluations); oracle vs reference <= 1e-4"""
    from tests.cfg_fixtures import SHIPPED
    assert _cfg_oracle(SHIPPED, 'shipped', gold, sd_spec, gd_sd) <= 1e-4


def test_bf16_emulation_of_diffnet_is_close_to_fp32(gd_sd):
    """oracle.diffnet.diffnet_forward(operand_bf16=True) (what the GPU's bf16 configuration is checked against at full size)
    stays within bf16 rounding of the fp32 oracle and is not identical to it"""
    rs = np.random.RandomState(3)
    x, cond = T(rs.standard_normal((1, 1, 80, 40)).astype(np.float32)), T(rs.standard_normal((1, 256, 40)).astype(np.float32))
    t = torch.tensor([17])
    a = odn.diffnet_forward(gd_sd, x, t, cond, 'denoise_fn.')
    b = odn.diffnet_forward(gd_sd, x, t, cond, 'denoise_fn.', operand_bf16=True)
    d = float((a - b).abs().max())
    assert 1e-5 < d < 5e-2


def test_bf16_emulation_noise_floor(gd_sd):
    """How closely can ANY two correct implementations of the bf16 configuration agree?  The same emulation with fp64 instead
    of fp32 accumulation (identical roundings, only the summation precision differs) lands as far from the fp32-accumulating
    emulation as that one is from the fp32 oracle: a 1e-7 difference flips a bf16 operand by one ulp (0.4 %) now and then, and
    20 layers carry every flip forward.  This is the floor tests/test_gpu_configs.py::test_config2_* holds the HIP path to
    (measured there: HIP vs emulation rms 7.1e-4, max 4.1e-3; here fp64 vs fp32 emulation rms ~7e-4, max ~3.6e-3)."""
    rs = np.random.RandomState(4)
    B, Tn = 2, 200
    x, cond = T(rs.standard_normal((B, 1, 80, Tn)).astype(np.float32)), T(rs.standard_normal((B, 256, Tn)).astype(np.float32))
    t = T(rs.randint(0, 100, size=(B,)).astype(np.int64))
    f32 = odn.diffnet_forward(gd_sd, x, t, cond, 'denoise_fn.')
    e32 = odn.diffnet_forward(gd_sd, x, t, cond, 'denoise_fn.', operand_bf16=True)
    e64 = odn.diffnet_forward(gd_sd, x, t, cond, 'denoise_fn.', operand_bf16=True, dtype=torch.float64).float()
    rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
    cost, floor = rms(e32, f32), rms(e64, e32)
    assert 2e-4 < cost < 3e-3
    assert 0.3 * cost < floor < 1.5 * cost


def test_r4_resblock2_and_plms_over_the_fft_denoiser(gold, sd_spec):
    """Round 4 fixtures (tools/make_golden_r4.py, reference outputs): HifiGanGenerator with ResBlock2 (hifigan.py:70-91) and the PLMS loop
    over DIFF_DECODERS['fft'] (shallow_diffusion_tts.py:168-201 with candidate_decoder.FFT as denoise_fn) pin the oracle's restatements."""
    from collections import OrderedDict
    from bisinger_amd import synth
    from oracle import candidate_decoder as ocd, diffusion as odf, hifigan as ohg
    g = gold('r4')
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_rb2_weight_norm'])
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 27).items()}
    rs = np.random.RandomState(31)
    mel = (rs.standard_normal((2, 80, 37)) * 1.5 - 3.0).astype(np.float32)
    y = ohg.hifigan_forward(sd, torch.from_numpy(mel), sd_spec['hifigan_rb2_cfg'])
    assert float((y - torch.from_numpy(g['rb2.wav'])).abs().max()) <= 2e-6
    fspec = OrderedDict((k, tuple(s)) for k, s in sd_spec['FFT'])
    fsd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(fspec, seed=17).items()}
    rs = np.random.RandomState(43)
    cond = torch.from_numpy(rs.standard_normal((1, 256, 40)).astype(np.float32))
    xT = torch.from_numpy(rs.standard_normal((1, 1, 80, 40)).astype(np.float32))
    x0 = odf.plms_sample(odf.make_schedule(100, 'linear', 0.06), lambda x_, t_: ocd.fft_denoiser_forward(fsd, x_, t_, cond), xT, 100, 5)
    assert float((x0 - torch.from_numpy(g['plmsfft.x0'])).abs().max()) <= 5e-5
