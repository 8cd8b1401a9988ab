"""NSF harmonic source + NSF-HiFiGAN generator — CPU oracle.  SURVEY.md §8 row f2.

Follows /root/reference/train_bisinger/modules/parallel_wavegan/models/source.py: SineGen :8-138 (flag_for_pulse False),
SourceModuleHnNSF :352-399; modules/hifigan/hifigan.py:111-132 (noise_convs), :145-160 (source add).
Randomness is supplied: ``rand_ini`` [B,9] (uniform, column 0 is zeroed as in :56) and ``noise`` [B,L,9] (N(0,1)).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .hifigan import DEFAULT_CFG, LRELU_SLOPE, fold_weight_norm


def sine_source(sd, f0, rand_ini, noise, sr, hop, prefix='m_source.', harmonic_num=8, sine_amp=0.1, noise_std=0.003,
                dtype=torch.float32):
    """f0 [B,T] -> harmonic source [B,1,T*hop]."""
    g = lambda k: sd[prefix + k].to(dtype)
    f0 = f0.to(dtype)
    f0u = f0[:, :, None].repeat_interleave(hop, dim=1)                     # Upsample(nearest), hifigan.py:147
    mult = torch.arange(1, harmonic_num + 2, dtype=dtype)
    f0_buf = f0u * mult                                                     # :112-116
    rad = (f0_buf / sr) % 1                                                 # :50
    ri = rand_ini.to(dtype).clone()
    ri[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ri                                        # :53-57
    tmp = torch.cumsum(rad, 1) % 1                                          # :67
    over = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * sine_amp   # :73-74, :119
    uv = (f0u > 0).to(dtype)                                                # :42-43
    noise_amp = uv * noise_std + (1 - uv) * sine_amp / 3                    # :129
    sine_waves = sines * uv + noise_amp * noise.to(dtype)                   # :130-134
    merged = torch.tanh(F.linear(sine_waves, g('l_linear.weight'), g('l_linear.bias')))   # :391
    return merged.transpose(1, 2)


def nsf_hifigan_forward(sd, mel, f0, rand_ini, noise, cfg=None, prefix='', dtype=torch.float32):
    """mel [B,80,T], f0 [B,T] -> wav [B,1,T*hop]   (hifigan.py:144-173 with use_pitch_embed)."""
    cfg = {**DEFAULT_CFG, **(cfg or {})}
    sd = fold_weight_norm({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    g = lambda k: sd[k].to(dtype)
    hop = int(np.prod(cfg['upsample_rates']))
    har = sine_source(sd, f0, rand_ini, noise, cfg['audio_sample_rate'], hop, dtype=dtype)
    nk = len(cfg['resblock_kernel_sizes'])
    x = F.conv1d(mel.to(dtype), g('conv_pre.weight'), g('conv_pre.bias'), padding=3)
    for i, (u, k) in enumerate(zip(cfg['upsample_rates'], cfg['upsample_kernel_sizes'])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, g(f'ups.{i}.weight'), g(f'ups.{i}.bias'), stride=u, padding=(k - u) // 2)
        if i + 1 < len(cfg['upsample_rates']):
            s = int(np.prod(cfg['upsample_rates'][i + 1:]))
            xs_ = F.conv1d(har, g(f'noise_convs.{i}.weight'), g(f'noise_convs.{i}.bias'), stride=s, padding=s // 2)
        else:
            xs_ = F.conv1d(har, g(f'noise_convs.{i}.weight'), g(f'noise_convs.{i}.bias'))
        xs_ = F.relu(xs_)
        xs_ = F.layer_norm(xs_.transpose(1, -1), (xs_.shape[1],)).transpose(1, -1)
        x = x + xs_
        acc = None
        for j, (ks, dil) in enumerate(zip(cfg['resblock_kernel_sizes'], cfg['resblock_dilation_sizes'])):
            p = f'resblocks.{i * nk + j}.'
            y = x
            for m, d in enumerate(dil):
                xt = F.leaky_relu(y, LRELU_SLOPE)
                xt = F.conv1d(xt, g(f'{p}convs1.{m}.weight'), g(f'{p}convs1.{m}.bias'), dilation=d, padding=(ks * d - d) // 2)
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, g(f'{p}convs2.{m}.weight'), g(f'{p}convs2.{m}.bias'), padding=(ks - 1) // 2)
                y = xt + y
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, g('conv_post.weight'), g('conv_post.bias'), padding=3)
    return torch.tanh(x)
