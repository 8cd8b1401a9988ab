"""PitchExtractor (mel -> f0) — CPU oracle.  SURVEY.md §8 row f2.

Follows /root/reference/train_bisinger/modules/fastspeech/pe.py: Prenet :9-42, ConvBlock :45-78, ConvStacks :81-117,
PitchExtractor :120-149; PitchPredictor modules/fastspeech/tts_modules.py:194-247; denorm_f0 utils/pitch_utils.py:63-76.
"""
import torch
import torch.nn.functional as F

from .fs2 import make_positions, sinusoidal_table


def pitch_extractor_forward(sd, mel, prefix='', conv_layers=2, predictor_layers=5, predictor_kernel=5, use_uv=True,
                            dtype=torch.float32):
    """mel [B,T,80] -> dict(pitch_pred [B,T,2], f0_denorm_pred [B,T])  (pitch_norm: log, pitch_type: frame)."""
    g = lambda k: sd[prefix + k].to(dtype)
    mel = mel.to(dtype)
    pad = mel.abs().sum(-1).eq(0)
    keep = 1 - pad.to(dtype)[:, None, :]
    x = mel.transpose(1, 2)
    for i in range(3):                                                  # Prenet: conv k5 -> ReLU -> BatchNorm (eval)
        p = f'mel_prenet.layers.{i}.'
        x = F.conv1d(x, g(p + '0.weight'), g(p + '0.bias'), padding=2)
        x = F.relu(x)
        x = F.batch_norm(x, g(p + '2.running_mean'), g(p + '2.running_var'), g(p + '2.weight'), g(p + '2.bias'), False, 0.1, 1e-5)
        x = x * keep
    x = F.linear(x.transpose(1, 2), g('mel_prenet.out_proj.weight'), g('mel_prenet.out_proj.bias')) * keep.transpose(1, 2)
    if conv_layers > 0:                                                 # ConvStacks (GroupNorm, residual)
        x = F.linear(x, g('mel_encoder.in_proj.weight'), g('mel_encoder.in_proj.bias')).transpose(1, -1)
        for i in range(conv_layers):
            p = f'mel_encoder.conv.{i}.'
            h = F.conv1d(x, g(p + 'conv.conv.weight'), g(p + 'conv.conv.bias'), padding=2)
            h = F.group_norm(h, h.shape[1] // 16, g(p + 'norm.weight'), g(p + 'norm.bias'), 1e-5)
            x = x + F.relu(h)
        x = F.linear(x.transpose(1, -1), g('mel_encoder.out_proj.weight'), g('mel_encoder.out_proj.bias'))
    # PitchPredictor :233-247
    p = 'pitch_predictor.'
    pos = make_positions(x[..., 0], 0)
    table = sinusoidal_table(max(4096, int(pos.max()) + 1), x.shape[-1], 0).to(dtype)
    xs = x + g(p + 'pos_embed_alpha') * table.index_select(0, pos.view(-1)).view(*pos.shape, -1)
    xs = xs.transpose(1, -1)
    kp = (predictor_kernel - 1) // 2
    for i in range(predictor_layers):
        xs = F.pad(xs, (kp, kp))
        xs = F.relu(F.conv1d(xs, g(f'{p}conv.{i}.1.weight'), g(f'{p}conv.{i}.1.bias')))
        C = xs.shape[1]
        xs = F.layer_norm(xs.transpose(1, -1), (C,), g(f'{p}conv.{i}.3.weight'), g(f'{p}conv.{i}.3.bias'), 1e-12).transpose(1, -1)
    pred = F.linear(xs.transpose(1, -1), g(p + 'linear.weight'), g(p + 'linear.bias'))
    f0 = 2 ** pred[:, :, 0]
    if use_uv:
        f0 = f0.masked_fill(pred[:, :, 1] > 0, 0.)
    f0 = f0.masked_fill(pad, 0.)
    return dict(pitch_pred=pred, f0_denorm_pred=f0)
