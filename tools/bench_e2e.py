#!/usr/bin/env python3
"""BASELINE.json configs[4]: end-to-end mel generation + HiFi-GAN on one MI355X, 22.05 kHz, real-time factor.
Also times the vocoder alone.  One JSON line per configuration."""
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bisinger_amd import synth  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = bench.build_model(dev)
from collections import OrderedDict  # noqa: E402
from bisinger_amd.hifigan import HifiGanGenerator  # noqa: E402
cfg = yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml'))
voc = HifiGanGenerator(cfg)
spec = OrderedDict((k, tuple(v.shape)) for k, v in voc.state_dict().items())
voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 7).items()})
voc = voc.to(dev)
voc.remove_weight_norm()
SR, HOP = cfg['audio_sample_rate'], 256

for B, T in [(1, 1000), (8, 1000), (16, 1000)]:
    inp = synth.synth_inputs(B, T // 10, T, seed=1)
    d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}

    def run():
        out = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], infer=True, seed=3, **kw)
        return out['mel_out'], voc(out['mel_out'].transpose(1, 2))
    mel, wav = run()
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        mel, wav = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    for _ in range(3):                        # the vocoder alone: warm-up, then the median of 3 batches of 10 forwards (behind the light B = 1
        w2 = voc(mel.transpose(1, 2))         # sampler a single batch of 10 x 0.85 ms reads up to 45 % high: the clocks are still ramping)
    torch.cuda.synchronize()
    dvs = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            w2 = voc(mel.transpose(1, 2))
        torch.cuda.synchronize()
        dvs.append((time.perf_counter() - t0) / 10)
    dv = sorted(dvs)[1]
    audio_s = B * T * HOP / SR
    print(json.dumps({'config': f'e2e mel-gen(100 steps)+HiFi-GAN B={B} T={T}', 'seconds': dt, 'audio_seconds': audio_s,
                      'rtf': dt / audio_s, 'vocoder_ms': dv * 1e3, 'vocoder_rtf': dv / audio_s,
                      'vocoder_gflops': 38.51e6 * B * T / dv / 1e9, 'wav_finite': bool(torch.isfinite(wav).all())}), flush=True)

# ---- a STREAM of requests (round 4): the same work issued back to back, (a) guard_mode 'same_call' — every guarded
# entry waits for its stream once, so consecutive requests cannot overlap — against (b) guard_mode 'deferred' on two HIP streams: the health
# words of a call are copied to pinned host memory behind its work and looked at by the next call, nothing waits on the host, and the vocoder
# of request i (stream B, behind an event) runs under the mel generation of request i + 1 (stream A).  All launches that hand data between
# workgroups come from ONE stream, in order — two streams of such launches could each hold a part of the chip and wait for the rest.
from bisinger_amd.hparams import hparams  # noqa: E402
for B, T, N in [(1, 1000, 16), (16, 1000, 6)]:
    inp = synth.synth_inputs(B, T // 10, T, seed=1)
    d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    res = {}
    for mode in ('same_call', 'deferred'):
        hparams['guard_mode'] = mode
        two = mode == 'deferred'
        sa = torch.cuda.Stream() if two else torch.cuda.current_stream()
        sb = torch.cuda.Stream() if two else sa

        def request(seed):
            with torch.cuda.stream(sa):
                mel = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], infer=True, seed=seed, **kw)['mel_out']
                ev = torch.cuda.Event()
                ev.record()
            with torch.cuda.stream(sb):
                sb.wait_event(ev)
                mel.record_stream(sb)
                return voc(mel.transpose(1, 2))
        request(100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            wav = request(i)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / N
        model.denoise_fn.check_deferred()                 # deferred: the last call's words are still pending — look at them
    hparams.pop('guard_mode', None)
    audio_s = B * T * HOP / SR
    print(json.dumps({'config': f'stream of {N} requests, mel-gen + HiFi-GAN, B={B} T={T}', 'same_call_serial_ms_per_request': res['same_call'] * 1e3,
                      'deferred_two_streams_ms_per_request': res['deferred'] * 1e3, 'speedup': res['same_call'] / res['deferred'],
                      'rtf_deferred': res['deferred'] / audio_s, 'wav_finite': bool(torch.isfinite(wav).all())}), flush=True)
