"""Gaussian diffusion schedule + samplers — CPU oracle.

Follows /root/reference/train_bisinger/usr/diff/shallow_diffusion_tts.py:
  linear/cosine_beta_schedule :44-62, buffers :90-126, extract :32-35,
  predict_start_from_noise :134-138, q_posterior :140-147, p_mean_variance :149-157,
  p_sample :159-166, p_sample_plms :168-201, q_sample :203-208, norm/denorm_spec :275-279,
  inference loop :244-272.
"""
from collections import deque

import numpy as np
import torch


def linear_beta_schedule(timesteps, max_beta=0.01):
    return np.linspace(1e-4, max_beta, timesteps)


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return np.clip(betas, a_min=0, a_max=0.999)


def make_schedule(timesteps, schedule_type='linear', max_beta=0.01, betas=None):
    """float64 numpy -> 12 float32 buffers, exactly as :90-122."""
    if betas is None:
        betas = (linear_beta_schedule(timesteps, max_beta) if schedule_type == 'linear'
                 else cosine_beta_schedule(timesteps))
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1. - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1., ac[:-1])
    pv = betas * (1. - ac_prev) / (1. - ac)
    f = lambda a: torch.tensor(a, dtype=torch.float32)
    return {
        'betas': f(betas),
        'alphas_cumprod': f(ac),
        'alphas_cumprod_prev': f(ac_prev),
        'sqrt_alphas_cumprod': f(np.sqrt(ac)),
        'sqrt_one_minus_alphas_cumprod': f(np.sqrt(1. - ac)),
        'log_one_minus_alphas_cumprod': f(np.log(1. - ac)),
        'sqrt_recip_alphas_cumprod': f(np.sqrt(1. / ac)),
        'sqrt_recipm1_alphas_cumprod': f(np.sqrt(1. / ac - 1)),
        'posterior_variance': f(pv),
        'posterior_log_variance_clipped': f(np.log(np.maximum(pv, 1e-20))),
        'posterior_mean_coef1': f(betas * np.sqrt(ac_prev) / (1. - ac)),
        'posterior_mean_coef2': f((1. - ac_prev) * np.sqrt(alphas) / (1. - ac)),
    }


def extract(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def norm_spec(x, spec_min, spec_max):
    return (x - spec_min) / (spec_max - spec_min) * 2 - 1


def denorm_spec(x, spec_min, spec_max):
    return (x + 1) / 2 * (spec_max - spec_min) + spec_min


def q_sample(sch, x_start, t, noise):
    return (extract(sch['sqrt_alphas_cumprod'], t, x_start.shape) * x_start +
            extract(sch['sqrt_one_minus_alphas_cumprod'], t, x_start.shape) * noise)


def p_sample(sch, denoise, x, t, noise, clip_denoised=True):
    """One ancestral step (:149-166).  ``denoise(x, t)`` -> eps; ``noise`` = the N(0,1) draw."""
    dt = x.dtype
    c = lambda k: extract(sch[k].to(dt), t, x.shape)
    eps = denoise(x, t)
    x0 = c('sqrt_recip_alphas_cumprod') * x - c('sqrt_recipm1_alphas_cumprod') * eps
    if clip_denoised:
        x0 = x0.clamp(-1., 1.)
    mean = c('posterior_mean_coef1') * x0 + c('posterior_mean_coef2') * x
    logvar = c('posterior_log_variance_clipped')
    nonzero = (1 - (t == 0).to(dt)).reshape(x.shape[0], *((1,) * (x.dim() - 1)))
    return mean + nonzero * (0.5 * logvar).exp() * noise


def ddpm_sample(sch, denoise, x_T, noise_steps, K_step):
    """Loop B (:265-267): for i in reversed(range(K_step)): x = p_sample(x, full(i)).
    noise_steps[k] is the draw of the k-th executed step (k=0 <-> i=K_step-1)."""
    x = x_T
    B = x.shape[0]
    for k, i in enumerate(reversed(range(K_step))):
        t = torch.full((B,), i, dtype=torch.long)
        x = p_sample(sch, denoise, x, t, noise_steps[k])
    return x


def plms_sample(sch, denoise, x_T, K_step, interval):
    """Loop A (:258-264) + p_sample_plms (:168-201).  The reference raises for B>1 at
    ``max(t-interval, 0)`` (:189); all rows share the same t in the inference loop, so the
    well-defined batched meaning used here (and by the HIP path) is the element-wise clamp,
    which reduces to the reference for B=1."""
    ac_all = sch['alphas_cumprod']

    def get_x_pred(x, noise_t, t):
        dt = x.dtype
        a_t = extract(ac_all.to(dt), t, x.shape)
        a_prev = extract(ac_all.to(dt), torch.max(t - interval, torch.zeros_like(t)), x.shape)
        a_t_sq, a_prev_sq = a_t.sqrt(), a_prev.sqrt()
        x_delta = (a_prev - a_t) * ((1 / (a_t_sq * (a_t_sq + a_prev_sq))) * x -
                                    1 / (a_t_sq * (((1 - a_prev) * a_t).sqrt() +
                                                   ((1 - a_t) * a_prev).sqrt())) * noise_t)
        return x + x_delta

    x = x_T
    B = x.shape[0]
    hist = deque(maxlen=4)
    for i in reversed(range(0, K_step, interval)):
        t = torch.full((B,), i, dtype=torch.long)
        eps = denoise(x, t)
        if len(hist) == 0:
            x_pred = get_x_pred(x, eps, t)
            eps_prev = denoise(x_pred, torch.clamp(t - interval, min=0))
            eps_prime = (eps + eps_prev) / 2
        elif len(hist) == 1:
            eps_prime = (3 * eps - hist[-1]) / 2
        elif len(hist) == 2:
            eps_prime = (23 * eps - 16 * hist[-1] + 5 * hist[-2]) / 12
        else:
            eps_prime = (55 * eps - 59 * hist[-1] + 37 * hist[-2] - 9 * hist[-3]) / 24
        x = get_x_pred(x, eps_prime, t)
        hist.append(eps)
    return x
