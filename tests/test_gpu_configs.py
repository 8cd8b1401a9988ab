"""GPU parity at BASELINE.json's STATED configurations and sizes (VERDICT r01 "pin the stated configurations").

  configs[0]  B=1, T_txt=50, T=500, FS2 + 100-step DDPM      HIP vs the reference's own output (tests/golden/cfg.npz) <= 1e-3
  shipped     timesteps=K_step=1000, beta to 0.02, PLMS/5    HIP (bsg_plms_sample over a 1000-row step table, 201
              (usr/configs/lang-esm-style-ori-shift/diff.yaml:16-23)   evaluations) vs the reference's own output <= 1e-3
  configs[1]  B=16, T=1000 fp32, FS2 + the FULL 100 steps     HIP vs the oracle, same supplied noise: max-abs <= 1e-3 on the
              with supplied noise                            de-normalised mel (north_star's bar, asserted at full size)
  configs[2]  B=64, T=1000 bf16: one DiffNet evaluation       HIP-bf16 vs oracle.diffnet_forward(operand_bf16=True), i.e. an
                                                              independent CPU emulation of the same roundings
"""
import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from oracle import diffnet as odn, melgen as omg
from tests.cfg_fixtures import CFG0, SHIPPED, build_model, inputs_and_noise
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy
KEYS = ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')


def _run(model, inp, noise):
    d = {k: T_(v).cuda() for k, v in inp.items()}
    return model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True,
                 noise=T_(noise), **{k: d[k] for k in KEYS})


def test_config0_single_utterance_T500(gold):
    c = CFG0
    model = build_model(c)
    inp, noise = inputs_and_noise(c)
    out = _run(model, inp, noise)
    assert out['mel_out'].shape == (1, 500, 80)
    dev = maxabs(out['mel_out'], gold('cfg')['cfg0.mel_out'])
    print(f'configs[0] B=1 T=500, 100 DDPM steps: HIP vs reference max-abs {dev:.3e}')
    assert dev <= 1e-3
    assert model.denoise_fn.handoff_timeouts() == 0      # B=1 runs the 4-way channel-split launches


def test_shipped_config_plms1000(gold):
    c = SHIPPED
    model = build_model(c)
    try:
        assert model.num_timesteps == 1000 and hparams['pndm_speedup'] == 5
        assert abs(float(model.betas[-1]) - 0.02) < 1e-8
        inp, noise = inputs_and_noise(c)
        out = _run(model, inp, noise)
        dev = maxabs(out['mel_out'], gold('cfg')['shipped.mel_out'])
        print(f'shipped config (1000-step schedule, PLMS interval 5, 201 evaluations): HIP vs reference max-abs {dev:.3e}')
        assert dev <= 1e-3
        assert model.denoise_fn.handoff_timeouts() == 0
    finally:
        use_config()


def test_config1_full_100_steps_vs_oracle():
    """B=16, T=1000: FS2 + all 100 sampler steps with supplied noise against the CPU oracle (about a minute of CPU)."""
    B, T, Tt = 16, 1000, 100
    c = dict(CFG0, B=B, T=T, T_txt=Tt)
    model = build_model(c)
    inp = synth.synth_inputs(B, Tt, T, seed=1)
    noise = synth.synth_noise(100, B, 80, T, seed=1)
    out = _run(model, inp, noise)
    torch.cuda.synchronize()
    sd = cpu_sd(model)
    want = omg.mel_gen(sd, {k: T_(v) for k, v in inp.items()}, T_(noise))
    dev = maxabs(out['mel_out'], want['mel_out'])
    dfs = maxabs(out['fs2_mel'], want['fs2_mel'])
    print(f'configs[1] B=16 T=1000, full 100 steps: HIP vs oracle max-abs {dev:.3e} on the de-normalised mel (fs2_mel {dfs:.3e})')
    assert out['mel_out'].shape == (B, T, 80)
    assert dev <= 1e-3 and dfs <= 1e-3
    assert model.denoise_fn.handoff_timeouts() == 0


def test_config2_bf16_full_size_vs_emulating_oracle():
    """B=64, T=1000, bf16-operand configuration: one DiffNet evaluation against an independent CPU emulation of the same
    roundings (not against the HIP fp32 path)."""
    use_config()
    from bisinger_amd.diffnet import DiffNet
    net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
    B, T = 64, 1000
    rs = np.random.RandomState(4)
    x = T_(rs.standard_normal((B, 1, 80, T)).astype(np.float32))
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32))
    t = T_(rs.randint(0, 100, size=(B,)).astype(np.int64))
    net.set_compute('bf16')
    try:
        got = net(x.cuda(), t.cuda(), cond.cuda()).cpu()
    finally:
        net.set_compute('fp32')
    sd = cpu_sd(net, 'denoise_fn.')
    emu = odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.', operand_bf16=True)
    f32 = odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.')
    e, q = maxabs(got, emu), maxabs(emu, f32)
    rms = float((got - emu).pow(2).mean().sqrt())
    rms_q = float((emu - f32).pow(2).mean().sqrt())
    rms_eps = float(f32.pow(2).mean().sqrt())
    print(f'configs[2] B=64 T=1000 eps (rms {rms_eps:.3f}): HIP-bf16 vs bf16-emulating oracle max-abs {e:.3e}, rms {rms:.2e}; '
          f'the roundings themselves cost max-abs {q:.3e}, rms {rms_q:.2e} vs the fp32 oracle')
    # bf16 roundings sit at decision boundaries: a 1e-7 summation-order difference flips an operand by one bf16 ulp (0.4 %)
    # now and then, and 20 layers carry every flip forward, so over 5 M outputs the agreement with the emulation is
    # statistical: closer (rms) to the emulation than the emulation is to fp32, and a max-abs of the same order as the
    # roundings' own — a wrong tile, tap or channel order shows as O(eps rms) = 0.1
    assert rms <= rms_q and e <= 2.0 * q and e <= 0.1 * rms_eps


def test_config3_rank_rows_of_64():
    """BASELINE configs[3] as ONE rank sees it (B_total = 64 utterances sharded 8 per GPU over 8 GPUs): the token-level front
    (embeddings, ESM over the batch axis — common_layers.py:848-860 — with K / V from all 64 rows' lang ids, the encoder on the rank's rows
    8..15: bsg_fs2midi_encode_rows), the frame-level part and the 100-step sampler on this rank's rows 8..15 only (shallow_diffusion_tts.py:230-273 with rows=slice(8, 16)).
      * against the same rows of the UNSHARDED B=64 run on this GPU (the 8-row shard takes 32-frame tiles, the unsharded run 64-frame
        tiles in four launch groups: another summation order of the same sums): <= 1e-5 of the de-normalised mel;
      * against the CPU oracle with the full-batch front and the same supplied noise: <= 1e-3 (north_star's bar)."""
    from oracle import fs2 as ofs2
    B, T, Tt = 64, 1000, 100
    rows = slice(8, 16)
    c = dict(CFG0, B=B, T=T, T_txt=Tt)
    model = build_model(c)
    inp = synth.synth_inputs(B, Tt, T, seed=3)
    noise = synth.synth_noise(100, B, 80, T, seed=3)
    d = {k: T_(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in KEYS}
    nz = T_(noise).cuda()
    full = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True, noise=nz, **kw)['mel_out'].clone()
    path_full = model.denoise_fn.last_path()
    part = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True, noise=nz, rows=rows, **kw)
    path_part = model.denoise_fn.last_path()
    torch.cuda.synchronize()
    # round 6: the rank's token-level front encodes ITS 8 utterances (K / V of the ESM from all 64 rows' lang ids), not the batch's 64
    assert model.fs2.last_rows()[0] == 8 * Tt, model.fs2.last_rows()
    assert part['mel_out'].shape == (8, T, 80) and full.shape == (B, T, 80)
    dev_shard = maxabs(part['mel_out'], full[rows])
    sd = cpu_sd(model)
    inp_t = {k: T_(v) for k, v in inp.items()}
    fs2_out = ofs2.fs2_forward(sd, inp_t, 'fs2.', rows=rows)          # the front on 64 rows, the decoder on rows 8..15
    want = omg.mel_gen(sd, dict(inp_t, mel2ph=inp_t['mel2ph'][rows]), T_(noise[:, rows]), fs2_out=fs2_out)
    dev = maxabs(part['mel_out'], want['mel_out'])
    dci = maxabs(part['decoder_inp'], want['decoder_inp'])
    print(f'configs[3] rank (rows 8..15 of B=64, T=1000, 100 steps; launch forms {path_part} / unsharded {path_full}): vs the same rows of the '
          f'unsharded run {dev_shard:.3e}; vs the oracle with the 64-row front {dev:.3e} (decoder_inp {dci:.3e})')
    assert dev_shard <= 1e-5
    assert dev <= 1e-3 and dci <= 1e-3
    assert model.denoise_fn.handoff_timeouts() == 0
