"""GaussianDiffusion.forward(infer=True) — CPU oracle of the whole mel-generation path.

Follows /root/reference/train_bisinger/usr/diff/shallow_diffusion_tts.py:230-273:
fs2 (decoder runs, skip_decoder = not infer) -> cond = decoder_inp^T -> x_T (gaussian_start) or
q_sample(norm(fs2_mel), K_step-1) -> DDPM loop (or PLMS) -> denorm * (mel2ph > 0).
Noise is *supplied* (see bisinger_amd/synth.py:synth_noise): noise[0] = x_T draw (also the
q_sample draw in the shallow branch), noise[1+k] = draw of the k-th executed p_sample.
"""
import torch

from .diffnet import diffnet_forward
from .diffusion import ddpm_sample, denorm_spec, make_schedule, norm_spec, plms_sample, q_sample
from .fs2 import fs2_forward


def mel_gen(sd, inp, noise, *, timesteps=100, K_step=100, max_beta=0.06, schedule_type='linear',
            gaussian_start=True, pndm_speedup=0, residual_layers=20, dilation_cycle_length=4,
            hp=None, dtype=torch.float32, fs2_out=None, n_steps=None):
    """sd: full GaussianDiffusion state_dict (keys fs2.*, denoise_fn.*, schedule buffers, spec_min/max).
    noise: [K_step+1, B, M, T].  Returns dict(mel_out [B,T,M], fs2_mel, mel2ph, decoder_inp).
    ``n_steps`` (bench cpu_baseline only): run just the first n sampler steps and return x."""
    ret = fs2_out if fs2_out is not None else fs2_forward(sd, inp, 'fs2.', hp, False, dtype)
    cond = ret['decoder_inp'].transpose(1, 2)
    sch = {k: sd[k] for k in make_schedule(2, 'linear', 0.01)} if 'betas' in sd else \
        make_schedule(timesteps, schedule_type, max_beta)
    smin, smax = sd['spec_min'].to(dtype), sd['spec_max'].to(dtype)
    ret['fs2_mel'] = ret['mel_out']
    noise = noise.to(dtype)
    if gaussian_start:
        x = noise[0][:, None]
    else:
        fm = norm_spec(ret['mel_out'], smin, smax).transpose(1, 2)[:, None]
        x = q_sample({k: v.to(dtype) for k, v in sch.items()}, fm,
                     torch.tensor([K_step - 1]).long(), noise[0][:, None])
    den = lambda x_, t_: diffnet_forward(sd, x_, t_, cond, 'denoise_fn.', residual_layers,
                                         dilation_cycle_length, dtype)
    if pndm_speedup:
        x = plms_sample(sch, den, x, K_step, pndm_speedup)
    elif n_steps is not None:
        from .diffusion import p_sample
        B = x.shape[0]
        for k, i in enumerate(list(reversed(range(K_step)))[:n_steps]):
            x = p_sample(sch, den, x, torch.full((B,), i, dtype=torch.long), noise[1 + k][:, None])
        ret['x'] = x
        return ret
    else:
        x = ddpm_sample(sch, den, x, noise[1:][:, :, None], K_step)
    x = x[:, 0].transpose(1, 2)
    ret['mel_out'] = denorm_spec(x, smin, smax)
    if inp.get('mel2ph') is not None:      # "for singing" branch, :269-270
        ret['mel_out'] = ret['mel_out'] * (inp['mel2ph'] > 0).to(dtype)[:, :, None]
    return ret
