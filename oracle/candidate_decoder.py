"""FFT candidate denoiser — CPU oracle.  SURVEY.md §8 row f4.
Follows /root/reference/train_bisinger/usr/diff/candidate_decoder.py:39-100."""
import torch
import torch.nn.functional as F

from .diffnet import mish, sinusoidal_pos_emb
from .fs2 import fft_blocks


def fft_denoiser_forward(sd, spec, t, cond, prefix='', n_layers=4, num_heads=2, kernel_size=9, dtype=torch.float32):
    """spec [B,1,M,T], t [B], cond [B,H,T] -> [B,1,M,T]."""
    g = lambda k: sd[prefix + k].to(dtype)
    x = F.conv1d(spec[:, 0].to(dtype), g('input_projection.weight'), g('input_projection.bias')).permute(0, 2, 1)
    e = sinusoidal_pos_emb(t, x.shape[-1]).to(dtype)
    d = F.linear(mish(F.linear(e, g('mlp.0.weight'), g('mlp.0.bias'))), g('mlp.2.weight'), g('mlp.2.bias'))
    c = cond.to(dtype).permute(0, 2, 1)
    te = d[:, None, :].repeat(1, c.shape[1], 1)
    x = F.linear(torch.cat([x, c, te], dim=-1), g('get_decode_inp.weight'), g('get_decode_inp.bias'))
    x = fft_blocks(sd, prefix, x, None, n_layers, num_heads, kernel_size, True, dtype)
    x = F.linear(x, g('get_mel_out.weight'), g('get_mel_out.bias')).permute(0, 2, 1)
    return x[:, None]
