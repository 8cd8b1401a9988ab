"""FastSpeech2-MIDI encoder/decoder — CPU oracle (state_dict in, tensors out).

Follows /root/reference/train_bisinger/ :
  modules/diffsinger_midi/fs2.py      FastspeechMIDIEncoder :14-65, FastSpeech2MIDI.forward :94-197
  modules/fastspeech/fs2.py           add_dur :154-177, run_decoder :236-240
  modules/fastspeech/tts_modules.py   LayerNorm(eps=1e-12) :39-58, DurationPredictor :61-153,
                                      LengthRegulator :156-191, FFTBlocks :253-309
  modules/commons/common_layers.py    SinusoidalPositionalEmbedding :106-179, MultiheadAttention :199-370
                                      (-> F.multi_head_attention_forward), TransformerFFNLayer :598-644,
                                      EncSALayer :664-730, ESM :832-860
  modules/commons/espnet_positional_embedding.py  RelPositionalEncoding :90-114 (reverse table :25-46)
  utils/__init__.py                   make_positions :146-158
"""
import math

import torch
import torch.nn.functional as F

DEFAULT_HP = dict(hidden_size=256, enc_layers=4, dec_layers=4, num_heads=2, enc_ffn_kernel_size=9,
                  dec_ffn_kernel_size=9, dur_predictor_layers=5, dur_predictor_kernel=3,
                  esm_heads=8, rel_pos_max_len=5000)


# ----------------------------------------------------------------------------- tables
def sinusoidal_table(num, dim, padding_idx=0):
    """common_layers.py:124-146 (tensor2tensor layout: [sin | cos], row padding_idx zeroed)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(num, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num, -1)
    if padding_idx is not None:
        e[padding_idx, :] = 0
    return e


def make_positions(x0, padding_idx=0):
    """utils/__init__.py:146-158 on the first channel of the decoder input."""
    mask = x0.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


def rel_pos_table(length, d_model):
    """espnet_positional_embedding.py:25-46 with reverse=True: row j holds position length-1-j,
    interleaved sin/cos."""
    pe = torch.zeros(length, d_model)
    position = torch.arange(length - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


# ----------------------------------------------------------------------------- layers
def self_attention(x, in_w, out_w, key_padding_mask, num_heads):
    """x [T,B,C]; bias-free in/out projections (common_layers.py:686-692); q * head_dim^-0.5;
    padded keys -> -inf; softmax over keys (F.multi_head_attention_forward semantics)."""
    T, B, C = x.shape
    hd = C // num_heads
    q, k, v = F.linear(x, in_w).chunk(3, dim=-1)
    q = q * math.sqrt(1.0 / float(hd))
    sh = lambda a: a.contiguous().view(T, B * num_heads, hd).transpose(0, 1)   # [B*H, T, hd]
    q, k, v = sh(q), sh(k), sh(v)
    s = torch.bmm(q, k.transpose(1, 2))                                        # [B*H, T, T]
    if key_padding_mask is not None:
        m = torch.zeros(B, 1, 1, T, dtype=x.dtype).masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
        s = (s.view(B, num_heads, T, T) + m).view(B * num_heads, T, T)
    p = F.softmax(s, dim=-1)
    o = torch.bmm(p, v).transpose(0, 1).contiguous().view(T, B, C)
    return F.linear(o, out_w)


def ffn(x, w1, b1, w2, b2, kernel_size):
    """TransformerFFNLayer :625-644 — Conv1d(k, pad k//2) * k^-0.5 -> GELU(erf) -> Linear.  x [T,B,C]."""
    h = F.conv1d(x.permute(1, 2, 0), w1, b1, padding=kernel_size // 2).permute(2, 0, 1)
    h = h * kernel_size ** -0.5
    h = F.gelu(h)
    return F.linear(h, w2, b2)


def enc_sa_layer(sd, p, x, pad_mask, num_heads, kernel_size, dtype):
    """EncSALayer.forward :706-730 (eval: dropout = identity).  x [T,B,C], pad_mask [B,T] bool."""
    g = lambda k: sd[p + k].to(dtype)
    keep = (1 - pad_mask.to(dtype)).transpose(0, 1)[..., None]
    C = x.shape[-1]
    r = x
    h = F.layer_norm(x, (C,), g('layer_norm1.weight'), g('layer_norm1.bias'), 1e-5)
    h = self_attention(h, g('self_attn.in_proj_weight'), g('self_attn.out_proj.weight'), pad_mask, num_heads)
    x = (r + h) * keep
    r = x
    h = F.layer_norm(x, (C,), g('layer_norm2.weight'), g('layer_norm2.bias'), 1e-5)
    h = ffn(h, g('ffn.ffn_1.weight'), g('ffn.ffn_1.bias'), g('ffn.ffn_2.weight'), g('ffn.ffn_2.bias'), kernel_size)
    return (r + h) * keep


def fft_blocks(sd, p, x, pad_mask, n_layers, num_heads, kernel_size, use_pos_embed, dtype):
    """FFTBlocks.forward :284-309.  x [B,T,C] -> [B,T,C]."""
    g = lambda k: sd[p + k].to(dtype)
    if pad_mask is None:
        pad_mask = x.abs().sum(-1).eq(0)
    keep_TB = 1 - pad_mask.transpose(0, 1).to(dtype)[:, :, None]
    if use_pos_embed:
        pos = make_positions(x[..., 0], 0)
        table = sinusoidal_table(max(2000, int(pos.max()) + 1), x.shape[-1], 0).to(dtype)
        x = x + g('pos_embed_alpha') * table.index_select(0, pos.view(-1)).view(*pos.shape, -1)
    x = x.transpose(0, 1) * keep_TB
    for i in range(n_layers):
        x = enc_sa_layer(sd, f'{p}layers.{i}.op.', x, pad_mask, num_heads, kernel_size, dtype) * keep_TB
    x = F.layer_norm(x, (x.shape[-1],), g('layer_norm.weight'), g('layer_norm.bias'), 1e-5) * keep_TB
    return x.transpose(0, 1)


def esm(sd, p, Eo, LP, nhead, dtype, rows=None):
    """ESM.forward :848-860.  nn.MultiheadAttention is sequence-first but is fed [B,T,C]:
    L = B, N = T_txt — the softmax runs over the *batch* axis.
    ``rows`` (slice): the result for these batch rows only, computed the way a rank of a sharded run computes it
    (bsg_fs2midi_encode_rows): K and V — projections of LN(LP), the only thing other rows contribute — for every row, the
    queries, the residual and the FFN for the rows asked for.  Row for row the same sums as the whole-batch call."""
    g = lambda k: sd[p + k].to(dtype)
    L, N, C = Eo.shape
    hd = C // nhead
    LPn = F.layer_norm(LP, (C,), g('ln1.weight'), g('ln1.bias'), 1e-5)
    w, b = g('mh.in_proj_weight'), g('mh.in_proj_bias')
    if rows is not None:
        Eo, LP = Eo[rows], LP[rows]
    Lq = Eo.shape[0]
    q = F.linear(Eo, w[:C], b[:C])
    k = F.linear(LPn, w[C:2 * C], b[C:2 * C])
    v = F.linear(LPn, w[2 * C:], b[2 * C:])
    sh = lambda a: a.contiguous().view(a.shape[0], N * nhead, hd).transpose(0, 1)       # [N*H, L, hd]
    q, k, v = sh(q) * math.sqrt(1.0 / float(hd)), sh(k), sh(v)
    a = F.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    o = torch.bmm(a, v).transpose(0, 1).contiguous().view(Lq, N, C)
    Mo = F.linear(o, g('mh.out_proj.weight'), g('mh.out_proj.bias')) + LP
    h = F.layer_norm(Mo, (C,), g('ln2.weight'), g('ln2.bias'), 1e-5)
    h = F.linear(F.relu(F.linear(h, g('ffn.0.weight'), g('ffn.0.bias'))), g('ffn.2.weight'), g('ffn.2.bias'))
    return h + Mo


def duration_predictor(sd, p, xs, pad_mask, n_layers, kernel, dtype):
    """DurationPredictor._forward(is_inference=True) :108-133 (dur_loss = mse, offset 1, SAME pad).
    Returns (dur int64 [B,T_txt], log-domain xs [B,T_txt,1])."""
    g = lambda k: sd[p + k].to(dtype)
    keep = (1 - pad_mask.to(dtype))
    xs = xs.transpose(1, -1)
    for i in range(n_layers):
        xs = F.pad(xs, ((kernel - 1) // 2, (kernel - 1) // 2))
        xs = F.relu(F.conv1d(xs, g(f'conv.{i}.1.weight'), g(f'conv.{i}.1.bias')))
        C = xs.shape[1]
        xs = F.layer_norm(xs.transpose(1, -1), (C,), g(f'conv.{i}.3.weight'), g(f'conv.{i}.3.bias'),
                          1e-12).transpose(1, -1)
        xs = xs * keep[:, None, :]
    xs = F.linear(xs.transpose(1, -1), g('linear.weight'), g('linear.bias'))
    xs = xs * keep[:, :, None]
    dur = torch.clamp(torch.round(xs.squeeze(-1).exp() - 1.0), min=0).long()
    return dur, xs


def length_regulator(dur, dur_padding):
    """LengthRegulator.forward :161-191 (alpha = 1)."""
    dur = torch.round(dur.float()).long()
    if dur_padding is not None:
        dur = dur * (1 - dur_padding.long())
    token_idx = torch.arange(1, dur.shape[1] + 1)[None, :, None]
    cs = torch.cumsum(dur, 1)
    cs_prev = F.pad(cs, [1, -1], mode='constant', value=0)
    pos_idx = torch.arange(int(dur.sum(-1).max()))[None, None]
    token_mask = (pos_idx >= cs_prev[:, :, None]) & (pos_idx < cs[:, :, None])
    return (token_idx * token_mask.long()).sum(1)


# ----------------------------------------------------------------------------- model
def fs2_forward(sd, inp, prefix='fs2.', hp=None, skip_decoder=False, dtype=torch.float32, rows=None, local_front=False):
    """FastSpeech2MIDI.forward(infer=True) with use_spk_id, no pitch/energy embed
    (diffsinger_midi/fs2.py:94-197).  ``inp``: dict of tensors (see bisinger_amd/synth.py).
    ``rows``: the outputs for these batch rows (a rank of a sharded run).  By default the token-level front
    is the reference's — evaluated on the whole batch — and sliced; ``local_front=True`` (needs ``mel2ph``) restates what
    bsg_fs2midi_encode_rows does instead: only the ESM's K / V see every row, everything else runs on ``rows``
    (tests/test_dist_cpu.py holds the two against each other)."""
    hp = {**DEFAULT_HP, **(hp or {})}
    H = hp['hidden_size']
    g = lambda k: sd[prefix + k].to(dtype)
    txt = inp['txt_tokens']
    ret = {}
    midi = F.embedding(inp['pitch_midi'], g('midi_embed.weight'))
    mdur = F.linear(inp['midi_dur'].to(dtype)[:, :, None], g('midi_dur_layer.weight'), g('midi_dur_layer.bias'))
    slur = F.embedding(inp['is_slur'], g('is_slur_embed.weight'))
    lang = F.embedding(inp['lang'], g('lang_embed.weight'))
    # FastspeechMIDIEncoder.forward_embedding :19-39
    x = math.sqrt(H) * F.embedding(txt, g('encoder.embed_tokens.weight'))
    if local_front:
        assert rows is not None and inp.get('mel2ph') is not None, 'local_front: a row slice and a given mel2ph'
        dyn = esm(sd, prefix + 'esm.', x, lang, hp['esm_heads'], dtype, rows=rows)
        x, midi, mdur, slur, txt = x[rows], midi[rows], mdur[rows], slur[rows], txt[rows]
        inp = dict(inp, spk_embed=inp['spk_embed'][rows], speechsing=inp['speechsing'][rows], mel2ph=inp['mel2ph'][rows])
        rows = None
    else:
        dyn = esm(sd, prefix + 'esm.', x, lang, hp['esm_heads'], dtype)
    x = x + midi + mdur + slur + dyn
    T_txt = x.shape[1]
    pe = rel_pos_table(max(hp['rel_pos_max_len'], T_txt), H).to(dtype)
    x = x * math.sqrt(H) + pe[None, :T_txt]
    pad = txt.eq(0)
    enc = fft_blocks(sd, prefix + 'encoder.', x, pad, hp['enc_layers'], hp['num_heads'],
                     hp['enc_ffn_kernel_size'], False, dtype)
    src_keep = (txt > 0).to(dtype)[:, :, None]
    spk = F.embedding(inp['spk_embed'], g('spk_embed_proj.weight'))[:, None, :]
    style = F.embedding(inp['speechsing'], g('style_embed.weight'))[:, None, :]
    dur_inp = (enc + spk) * src_keep
    mel2ph = inp.get('mel2ph')
    if mel2ph is None:
        dur, xs = duration_predictor(sd, prefix + 'dur_predictor.', dur_inp, txt == 0,
                                     hp['dur_predictor_layers'], hp['dur_predictor_kernel'], dtype)
        ret['dur'] = xs
        ret['dur_choice'] = dur
        mel2ph = length_regulator(dur, txt == 0)
    if rows is not None:
        # sharded evaluation: the token-level front above saw the whole batch (ESM couples
        # rows); everything frame-level is per-row, so only this rank's rows continue
        enc, spk, style, mel2ph = enc[rows], spk[rows], style[rows], mel2ph[rows]
    ret['mel2ph'] = mel2ph
    dec_in = F.pad(enc, [0, 0, 1, 0])
    dec_in = torch.gather(dec_in, 1, mel2ph[..., None].repeat([1, 1, H]))
    tgt_keep = (mel2ph > 0).to(dtype)[:, :, None]
    ret['decoder_inp'] = dec_in = (dec_in + spk + style) * tgt_keep
    if skip_decoder:
        return ret
    y = fft_blocks(sd, prefix + 'decoder.', dec_in, None, hp['dec_layers'], hp['num_heads'],
                   hp['dec_ffn_kernel_size'], True, dtype)
    ret['mel_out'] = F.linear(y, g('mel_out.weight'), g('mel_out.bias')) * tgt_keep
    return ret
