"""DiffNet (WaveNet-style noise predictor) — CPU oracle.

Follows /root/reference/train_bisinger/usr/diff/net.py:
  SinusoidalPosEmb :32-44, ResidualBlock :58-78, DiffNet.forward :107-130,
  Mish = x*tanh(softplus(x)) (usr/diff/diffusion.py:68-70).
"""
import math

import torch
import torch.nn.functional as F


def mish(x):
    return x * torch.tanh(F.softplus(x))


def sinusoidal_pos_emb(t, dim):
    """net.py:37-44 — evaluated in float32 from the int64 step, cat(sin, cos), divisor half-1."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)            # float32, as in the reference
    e = t[:, None] * e[None, :]                        # int64 * float32 -> float32
    return torch.cat((e.sin(), e.cos()), dim=-1)


def step_embedding(sd, t, C, prefix='', dtype=torch.float32):
    """net.py:119-120 — SinusoidalPosEmb -> Linear -> Mish -> Linear.  Returns [B, C]."""
    g = lambda k: sd[prefix + k].to(dtype)
    e = sinusoidal_pos_emb(t, C).to(dtype)
    h = mish(F.linear(e, g('mlp.0.weight'), g('mlp.0.bias')))
    return F.linear(h, g('mlp.2.weight'), g('mlp.2.bias'))


def _bf16(x):
    """round-to-nearest-even to bfloat16, kept in the working dtype (emulates a bf16 MFMA operand)"""
    return x.to(torch.bfloat16).to(x.dtype)


def residual_block(sd, p, x, cond, d, dilation, dtype=torch.float32, operand_bf16=False):
    """net.py:66-78.  gate = first C channels -> sigmoid, filter = last C -> tanh;
    residual = first C of the output projection, skip = last C.
    operand_bf16: emulate the build's bf16 configuration (BASELINE config 3) — the two conv weights, the conv input x + d,
    the hoisted conditioner term and the gated activation are rounded to bf16, everything else (accumulation, gate,
    residual stream) as above.  (The build also STORES the running skip sum as bf16; the caller of this function sums.)"""
    g = lambda k: sd[p + k].to(dtype)
    if operand_bf16:
        dp = F.linear(d, g('diffusion_projection.weight'), g('diffusion_projection.bias')).unsqueeze(-1)
        c = _bf16(F.conv1d(cond, g('conditioner_projection.weight'), g('conditioner_projection.bias')) + g('dilated_conv.bias')[None, :, None])
        y = F.conv1d(_bf16(x + dp), _bf16(g('dilated_conv.weight')), None, padding=dilation, dilation=dilation) + c
        gate, filt = torch.chunk(y, 2, dim=1)
        y = _bf16(torch.sigmoid(gate) * torch.tanh(filt))
        y = F.conv1d(y, _bf16(g('output_projection.weight')), g('output_projection.bias'))
        res, skip = torch.chunk(y, 2, dim=1)
        return (x + res) / math.sqrt(2.0), skip
    dp = F.linear(d, g('diffusion_projection.weight'), g('diffusion_projection.bias')).unsqueeze(-1)
    c = F.conv1d(cond, g('conditioner_projection.weight'), g('conditioner_projection.bias'))
    y = F.conv1d(x + dp, g('dilated_conv.weight'), g('dilated_conv.bias'),
                 padding=dilation, dilation=dilation) + c
    gate, filt = torch.chunk(y, 2, dim=1)
    y = torch.sigmoid(gate) * torch.tanh(filt)
    y = F.conv1d(y, g('output_projection.weight'), g('output_projection.bias'))
    res, skip = torch.chunk(y, 2, dim=1)
    return (x + res) / math.sqrt(2.0), skip


def diffnet_forward(sd, spec, t, cond, prefix='', n_layers=20, cycle=4, dtype=torch.float32, operand_bf16=False,
                    skip_rounding='running', tail_bf16=False, in_bf16=False):
    """spec [B,1,M,T], t [B] int64, cond [B,H,T] -> eps [B,1,M,T]   (net.py:107-130).
    operand_bf16: emulate the build's bf16 configuration end to end — residual_block's operand roundings plus the skip sum
    stored as bf16: skip_rounding='running' after every layer (s_0 = bf16(o_0), s_i = bf16(s_{i-1} + o_i), last layer
    bf16((s + o)/sqrt(L)): the build's per-layer launches), 'final' once (bf16(sum / sqrt(L)): the build's stack launch, which
    keeps the sum in fp32 registers).
    tail_bf16: the skip and output projections with bf16 operands too (weights, the skip sum, relu(h)), fp32 accumulation —
    the build's fused bf16 step tail; in_bf16: the input projection likewise (x and its weight rounded) — the build's tail
    computes the NEXT evaluation's input projection, so every evaluation of a sampler run but the first.  Without them the
    in/skip/out projections are fp32 (the build's bsg_diffnet_forward)."""
    g = lambda k: sd[prefix + k].to(dtype)
    spec = spec.to(dtype)
    cond = cond.to(dtype)
    x = spec[:, 0]
    if in_bf16:
        x = F.relu(F.conv1d(_bf16(x), _bf16(g('input_projection.weight')), g('input_projection.bias')))
    else:
        x = F.relu(F.conv1d(x, g('input_projection.weight'), g('input_projection.bias')))
    d = step_embedding(sd, t, x.shape[1], prefix, dtype)
    if operand_bf16:
        run = None
        for i in range(n_layers):
            x, s = residual_block(sd, f'{prefix}residual_layers.{i}.', x, cond, d, 2 ** (i % cycle), dtype, True)
            run = s if run is None else run + s
            if skip_rounding == 'running':
                run = _bf16(run / math.sqrt(n_layers) if i == n_layers - 1 else run)
        if skip_rounding != 'running':
            run = _bf16(run / math.sqrt(n_layers))
        x = run
    else:
        skips = []
        for i in range(n_layers):
            x, s = residual_block(sd, f'{prefix}residual_layers.{i}.', x, cond, d, 2 ** (i % cycle), dtype)
            skips.append(s)
        x = torch.sum(torch.stack(skips), dim=0) / math.sqrt(n_layers)
    if tail_bf16:
        x = F.relu(F.conv1d(_bf16(x), _bf16(g('skip_projection.weight')), g('skip_projection.bias')))
        x = F.conv1d(_bf16(x), _bf16(g('output_projection.weight')), g('output_projection.bias'))
    else:
        x = F.relu(F.conv1d(x, g('skip_projection.weight'), g('skip_projection.bias')))
        x = F.conv1d(x, g('output_projection.weight'), g('output_projection.bias'))
    return x[:, None, :, :]
