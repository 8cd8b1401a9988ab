"""``GaussianDiffusion`` — drop-in for usr/diff/shallow_diffusion_tts.py:71-285 (inference direction).

Same constructor, sub-module names (``fs2``, ``denoise_fn``), 14 registered buffers (computed in
float64 numpy and cast, :90-126, so they are bit-identical and a checkpoint's buffers override them)
and ``forward(..., infer=True)`` result keys.  The step loop itself (p_sample :159-166 /
p_sample_plms :168-201 over DiffNet) runs inside libbisinger_hip (``bsg_ddpm_sample`` /
``bsg_plms_sample``): one C call enqueues all ``K_step`` iterations on the current stream.

Noise: the reference draws ``torch.randn`` per step.  Here either the caller supplies the draws
(``noise=[K_step+1,B,M,T]``: parity mode, see bisinger_amd/synth.py) or they come from the on-device
Philox4x32-10 stream keyed by ``seed`` and the *global* batch row (so utterance shards reproduce the
unsharded run).
"""
from ctypes import POINTER, byref, c_float, cast

import numpy as np
import torch
from torch import nn

from . import _lib
from .hparams import hparams


def linear_beta_schedule(timesteps, max_beta=None):
    """:44-49.  The reference captures hparams['max_beta'] at import time; here it is read at call time."""
    if max_beta is None:
        max_beta = hparams.get('max_beta', 0.01)
    return np.linspace(1e-4, max_beta, timesteps)


def cosine_beta_schedule(timesteps, s=0.008):
    """:52-62."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)


beta_schedule = {'cosine': cosine_beta_schedule, 'linear': linear_beta_schedule}

_SCHED_KEYS = ('sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_mean_coef1',
               'posterior_mean_coef2', 'alphas_cumprod')


class GaussianDiffusion(nn.Module):
    def __init__(self, phone_encoder, out_dims, denoise_fn, timesteps=1000, K_step=1000, loss_type=None,
                 betas=None, spec_min=None, spec_max=None):
        super().__init__()
        self.denoise_fn = denoise_fn
        if hparams.get('use_midi'):
            from .fs2 import FastSpeech2MIDI
            self.fs2 = FastSpeech2MIDI(phone_encoder, out_dims)
        else:
            raise NotImplementedError('only the MIDI front (use_midi: true) is on the BiSinger path (SURVEY.md §8)')
        self.mel_bins = out_dims
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else betas
        elif 'schedule_type' in hparams:
            betas = beta_schedule[hparams['schedule_type']](timesteps)
        else:
            betas = cosine_beta_schedule(timesteps)
        alphas = 1. - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.K_step = K_step
        self.loss_type = loss_type if loss_type is not None else hparams.get('diff_loss_type', 'l1')
        f = lambda a: torch.tensor(a, dtype=torch.float32)
        self.register_buffer('betas', f(betas))
        self.register_buffer('alphas_cumprod', f(ac))
        self.register_buffer('alphas_cumprod_prev', f(ac_prev))
        self.register_buffer('sqrt_alphas_cumprod', f(np.sqrt(ac)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', f(np.sqrt(1. - ac)))
        self.register_buffer('log_one_minus_alphas_cumprod', f(np.log(1. - ac)))
        self.register_buffer('sqrt_recip_alphas_cumprod', f(np.sqrt(1. / ac)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', f(np.sqrt(1. / ac - 1)))
        pv = betas * (1. - ac_prev) / (1. - ac)
        self.register_buffer('posterior_variance', f(pv))
        self.register_buffer('posterior_log_variance_clipped', f(np.log(np.maximum(pv, 1e-20))))
        self.register_buffer('posterior_mean_coef1', f(betas * np.sqrt(ac_prev) / (1. - ac)))
        self.register_buffer('posterior_mean_coef2', f((1. - ac_prev) * np.sqrt(alphas) / (1. - ac)))
        self.register_buffer('spec_min', torch.FloatTensor(spec_min)[None, None, :hparams['keep_bins']])
        self.register_buffer('spec_max', torch.FloatTensor(spec_max)[None, None, :hparams['keep_bins']])
        self._sched_cache = None

    # ------------------------------------------------------------------ schedule -> C struct
    def _schedule(self):
        """Host copies of the (possibly checkpoint-loaded) buffers for bsg_schedule."""
        key = tuple((getattr(self, k).data_ptr(), getattr(self, k)._version) for k in _SCHED_KEYS) + \
            (self.posterior_log_variance_clipped._version,)
        if self._sched_cache is not None and self._sched_cache[0] == key:
            return self._sched_cache[1]
        host = {k: getattr(self, k).detach().float().cpu().contiguous() for k in _SCHED_KEYS}
        sigma = (0.5 * self.posterior_log_variance_clipped.detach().float().cpu()).exp()   # :166
        sigma[0] = 0.0                                                                      # nonzero_mask (:165)
        host['sigma'] = sigma.contiguous()
        s = _lib.Schedule()
        s.num_timesteps = self.num_timesteps
        for k, v in host.items():
            setattr(s, k, cast(v.data_ptr(), POINTER(c_float)))
        self._sched_cache = (key, (s, host))   # keep the host tensors alive with the struct
        return s, host

    # ------------------------------------------------------------------ samplers
    @torch.no_grad()
    def sample(self, cond, x, noise=None, seed=0, row0=0, B_total=None, n_steps=None):
        """Run the inference loop (:258-267) from ``x`` ([B,1,M,T], modified in place) under ``cond`` [B,H,T]."""
        lib = _lib.load()
        B, _, M, T = x.shape
        assert x.is_contiguous() and x.dtype == torch.float32
        if hparams.get('pndm_speedup') and (n_steps is not None or noise is not None):
            raise ValueError('sample(): the PLMS loop (pndm_speedup) is deterministic after x_T and always runs the whole '
                             'schedule; n_steps / noise only apply to the DDPM loop')
        self.denoise_fn.prepare(cond)
        s, _keep = self._schedule()
        h = self.denoise_fn._h
        t = self.K_step
        from .diffnet import DiffNet
        if not isinstance(self.denoise_fn, DiffNet):
            # any other denoise_fn (DIFF_DECODERS['fft']): the reference loops (:258-267) step by step — the denoiser through its own
            # kernels, the update of x through the generic step entries (bsg_ddpm_step / bsg_plms_step)
            n = t if n_steps is None else n_steps
            capturing = torch.cuda.is_current_stream_capturing()
            keep = None if capturing else x.clone()
            interval = int(hparams.get('pndm_speedup') or 0)

            def tt(i):
                return torch.full((B,), i, device=x.device, dtype=torch.long)

            def loop_plms():
                # p_sample_plms (:168-201): a 4-deep history of the noise predictions; the first iteration is a predictor / corrector
                # pair (two evaluations).  Batched semantics as bsg_plms_sample's: t - interval clamps at 0 per element
                hist = []                                  # newest first
                with torch.cuda.device(x.device):
                    for i in reversed(range(0, t, interval)):
                        ip = max(i - interval, 0)
                        e0 = self.denoise_fn(x, tt(i), cond).contiguous()
                        if not hist:
                            xp = torch.empty_like(x)
                            _lib.check(lib.bsg_plms_step(_lib.ptr(x), _lib.ptr(xp), _lib.ptr(e0), None, None, None, 0, 0, byref(s), i, ip,
                                                         x.numel(), _lib.stream_ptr()), 'bsg_plms_step')
                            e1 = self.denoise_fn(xp, tt(ip), cond).contiguous()
                            _lib.check(lib.bsg_plms_step(_lib.ptr(x), _lib.ptr(x), _lib.ptr(e0), _lib.ptr(e1), None, None, 1, 1, byref(s), i, ip,
                                                         x.numel(), _lib.stream_ptr()), 'bsg_plms_step')
                        else:
                            h = hist + [None] * (3 - len(hist))
                            _lib.check(lib.bsg_plms_step(_lib.ptr(x), _lib.ptr(x), _lib.ptr(e0), _lib.ptr(h[0]), _lib.ptr(h[1]), _lib.ptr(h[2]),
                                                         len(hist), 0, byref(s), i, ip, x.numel(), _lib.stream_ptr()), 'bsg_plms_step')
                        hist = [e0] + hist[:2]

            def loop():
                if interval:
                    return loop_plms()
                with torch.cuda.device(x.device):
                    for k in range(n):
                        i = t - 1 - k
                        eps = self.denoise_fn(x, tt(i), cond)
                        nz = None if noise is None else noise[k].contiguous()
                        _lib.check(lib.bsg_ddpm_step(_lib.ptr(x), _lib.ptr(eps.contiguous()), _lib.ptr(nz), byref(s), i, x.numel(), seed,
                                                     row0 * M * T, _lib.stream_ptr()), 'bsg_ddpm_step')
            # this denoiser is GEMMs only: the range guard of the split-fp16 GEMMs (an operand beyond the fp16 range is counted, not
            # clipped) is read once per call; on an event every GEMM moves to the fp32 matrix pipe and the loop is repeated from x_T
            def again():
                x.copy_(keep)
                self.denoise_fn.prepare(cond)
            _lib.range_guarded(loop, 'FFT denoiser sampler loop', on_retry=None if capturing else again, device=x,
                               owners=(self.denoise_fn,) if isinstance(self.denoise_fn, _lib.GemmGuarded) else ())
            return x
        n = t if n_steps is None else n_steps
        if noise is not None:
            noise = noise.contiguous()
            assert tuple(noise.shape) == (n, B, M, T), noise.shape

        def run():
            with torch.cuda.device(x.device):
                if hparams.get('pndm_speedup'):
                    _lib.check(lib.bsg_plms_sample(h, byref(s), _lib.ptr(x), t, int(hparams['pndm_speedup']), B, T,
                                                   _lib.stream_ptr()), 'bsg_plms_sample')
                else:
                    _lib.check(lib.bsg_ddpm_sample(h, byref(s), _lib.ptr(x), _lib.ptr(noise), seed, t - 1, n, B, T, row0,
                                                   B if B_total is None else B_total, _lib.stream_ptr()), 'bsg_ddpm_sample')
        # the loop updates x in place: keep x_T (B*M*T floats) so that the call can repeat itself — without hand-off launches if a
        # workgroup gave up, on the fp32 matrix pipe if a value left the fp16 range — and an invalid x never leaves this function
        keep = None if torch.cuda.is_current_stream_capturing() else x.clone()
        self.denoise_fn.guarded(run, B, T, restore=None if keep is None else (lambda: x.copy_(keep)))
        return x

    @torch.no_grad()
    def p_sample(self, x, t, cond, clip_denoised=True, repeat_noise=False, noise=None, seed=0):
        """One ancestral step with the reference signature (:159-166); all rows must share ``t``."""
        assert clip_denoised and not repeat_noise
        ti = int(t[0])
        assert bool((t == ti).all()), 'p_sample: the HIP sampler takes one timestep per call'
        lib = _lib.load()
        B, _, M, T = x.shape
        x = x.contiguous().clone()
        self.denoise_fn._ensure_bound(cond)
        s, _keep = self._schedule()
        nz = None if noise is None else noise.reshape(1, B, M, T).contiguous()
        x_in = x.clone()

        def run():
            with torch.cuda.device(x.device):
                _lib.check(lib.bsg_ddpm_sample(self.denoise_fn._h, byref(s), _lib.ptr(x), _lib.ptr(nz), seed, ti, 1, B, T, 0, B,
                                               _lib.stream_ptr()), 'bsg_ddpm_sample')
        self.denoise_fn.guarded(run, B, T, restore=lambda: x.copy_(x_in))
        return x

    def philox_normal(self, shape, device, seed, stream_id, offset=0):
        x = torch.empty(shape, device=device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().bsg_philox_normal(_lib.ptr(x), x.numel(), seed, stream_id, offset, _lib.stream_ptr()),
                       'bsg_philox_normal')
        return x

    # ------------------------------------------------------------------ reference forward (:230-273)
    @torch.no_grad()
    def forward(self, txt_tokens, mel2ph=None, spk_embed=None, ref_mels=None, f0=None, uv=None, energy=None,
                infer=False, noise=None, seed=None, rows=None, **kwargs):
        """Extensions over the reference signature: ``noise`` (supplied draws, parity mode), ``seed`` (Philox
        key, default hparams['seed']) and ``rows`` (slice of the batch this process generates; outputs then
        have len(rows) rows and reproduce the same rows of the unsharded call — SURVEY.md §8e).
        How closely: the launch form of the sampler is chosen by the LOCAL batch size (part forms — four workgroups per tile — while
        4 * B * ceil(T/64) <= CUs, 32-frame tiles up to one launch group, 64-frame tiles beyond).  The 32- and 64-frame forms sum in the
        same order (sharded rows were bit-identical to the unsharded run at every shape tested, e.g. 8 of 64 rows at T=1000); the part
        forms order GEMM1's k-steps differently and add the conditioner term last (1e-6 on the mel).  ``BSG_H2_PART=0`` makes the
        arithmetic of a row independent of the batch around it."""
        if not infer:
            raise NotImplementedError('training (p_losses) is outside the accelerated hot path (SURVEY.md §8)')
        # one range guard around the whole call (FS2, the conditioner projections, the sampler): an operand beyond the fp16 range of the
        # split-fp16 GEMMs repeats all of it with THAT handle (FS2's or the denoiser's) on the fp32 matrix pipe (_lib.range_guarded; the nested
        # guards register their handles with this one and leave the check to it)
        return _lib.range_guarded(lambda: self._forward_infer(txt_tokens, mel2ph, spk_embed, ref_mels, f0, uv, energy, noise, seed, rows,
                                                              **kwargs), 'GaussianDiffusion.forward', device=self,
                                  owners=tuple(m for m in (self.fs2, self.denoise_fn) if isinstance(m, _lib.GemmGuarded)))

    def _forward_infer(self, txt_tokens, mel2ph, spk_embed, ref_mels, f0, uv, energy, noise, seed, rows, **kwargs):
        B_total = txt_tokens.shape[0]
        row0 = 0
        if rows is not None:
            row0, stop, stride = rows.indices(B_total)
            assert stride == 1 and stop > row0, 'rows must be a contiguous non-empty slice'
        ret = self.fs2(txt_tokens, mel2ph, spk_embed, ref_mels, f0, uv, energy, skip_decoder=False, infer=True,
                       rows=rows, **kwargs)
        if mel2ph is not None and rows is not None:
            mel2ph = mel2ph[rows]
        cond = ret['decoder_inp'].transpose(1, 2).contiguous()
        ret['fs2_mel'] = ret['mel_out']
        B, H, T = cond.shape
        t = self.K_step
        seed = int(hparams.get('seed', 1234)) if seed is None else int(seed)
        M = self.mel_bins
        if noise is not None:
            noise = noise.to(cond.device, torch.float32)
            if rows is not None and noise.shape[1] == B_total:
                noise = noise[:, rows]
            # x_T is updated in place by the sampler: it must be a COPY — `noise` may be the caller's own device tensor (.to() and the
            # row slice are views), and a second call with the same tensor would start from the first call's result
            draw0, steps = noise[0][:, None].clone(memory_format=torch.contiguous_format), noise[1:]
        else:
            off = row0 * M * T
            draw0 = self.philox_normal((B, 1, M, T), cond.device, seed, 0, off)
            steps = None
        lib = _lib.load()
        smin, smax = self.spec_min.reshape(-1).contiguous(), self.spec_max.reshape(-1).contiguous()
        if hparams.get('gaussian_start'):
            x = draw0
        else:
            _s, host = self._schedule()
            x = torch.empty_like(draw0)
            with torch.cuda.device(cond.device):
                _lib.check(lib.bsg_mel_start(_lib.ptr(ret['mel_out'].contiguous()), _lib.ptr(smin), _lib.ptr(smax), _lib.ptr(draw0),
                                             float(self.sqrt_alphas_cumprod[t - 1]), float(self.sqrt_one_minus_alphas_cumprod[t - 1]),
                                             _lib.ptr(x), B, M, T, _lib.stream_ptr()), 'bsg_mel_start')
        x = self.sample(cond, x, noise=None if hparams.get('pndm_speedup') else steps, seed=seed, row0=row0, B_total=B_total)
        out = torch.empty(B, T, M, device=cond.device)
        m2p = None if mel2ph is None else mel2ph.to(device=cond.device, dtype=torch.long).contiguous()
        with torch.cuda.device(cond.device):
            _lib.check(lib.bsg_mel_finish(_lib.ptr(x), _lib.ptr(smin), _lib.ptr(smax), _lib.ptr(m2p), _lib.ptr(out), B, M, T,
                                          _lib.stream_ptr()), 'bsg_mel_finish')
        ret['mel_out'] = out
        return ret

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min

    def out2mel(self, x):
        return x
