"""HiFi-GAN generator forward — CPU oracle.

Follows /root/reference/train_bisinger/modules/hifigan/hifigan.py:
  ResBlock1.forward :54-61, ResBlock2.forward :83-87, HifiGanGenerator.__init__ :105-142, forward :144-173
  (note the default-slope 0.01 leaky_relu before conv_post, :169), remove_weight_norm :175-182.
Config: TB/configs/tts/hifigan.yaml:3-10.  NSF (use_pitch_embed) is SURVEY.md §8 row f2 (next).
"""
import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(resblock='1', upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                   upsample_initial_channel=128, resblock_kernel_sizes=[3, 7, 11],
                   resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], use_pitch_embed=False,
                   audio_sample_rate=22050)
LRELU_SLOPE = 0.1


def fold_weight_norm(sd):
    """weight = g * v / ||v|| with the norm over every dim but 0 (torch.nn.utils.weight_norm, dim=0).
    Accepts both checkpoint layouts."""
    out = {}
    for k, v in sd.items():
        if k.endswith('.weight_v'):
            base = k[:-len('.weight_v')]
            g = sd[base + '.weight_g']
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            out[base + '.weight'] = v * (g / norm)
        elif k.endswith('.weight_g'):
            continue
        else:
            out[k] = v
    return out


def hifigan_forward(sd, mel, cfg=None, prefix='', dtype=torch.float32):
    """mel [B,80,T] -> wav [B,1,T*prod(upsample_rates)].  ``sd`` may be weight-normed or folded."""
    cfg = {**DEFAULT_CFG, **(cfg or {})}
    sd = fold_weight_norm({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    g = lambda k: sd[k].to(dtype)
    nk = len(cfg['resblock_kernel_sizes'])
    x = F.conv1d(mel.to(dtype), g('conv_pre.weight'), g('conv_pre.bias'), padding=3)
    for i, (u, k) in enumerate(zip(cfg['upsample_rates'], cfg['upsample_kernel_sizes'])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, g(f'ups.{i}.weight'), g(f'ups.{i}.bias'), stride=u, padding=(k - u) // 2)
        xs = None
        for j, (ks, dil) in enumerate(zip(cfg['resblock_kernel_sizes'], cfg['resblock_dilation_sizes'])):
            p = f'resblocks.{i * nk + j}.'
            y = x
            for m, d in enumerate(dil):
                xt = F.leaky_relu(y, LRELU_SLOPE)
                if str(cfg['resblock']) != '1':      # ResBlock2.forward :83-87: one conv per dilation
                    y = F.conv1d(xt, g(f'{p}convs.{m}.weight'), g(f'{p}convs.{m}.bias'), dilation=d, padding=(ks * d - d) // 2) + y
                    continue
                xt = F.conv1d(xt, g(f'{p}convs1.{m}.weight'), g(f'{p}convs1.{m}.bias'),
                              dilation=d, padding=(ks * d - d) // 2)
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, g(f'{p}convs2.{m}.weight'), g(f'{p}convs2.{m}.bias'),
                              padding=(ks - 1) // 2)
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)            # default slope 0.01 (hifigan.py:169)
    x = F.conv1d(x, g('conv_post.weight'), g('conv_post.bias'), padding=3)
    return torch.tanh(x)
