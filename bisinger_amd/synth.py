"""Deterministic synthetic weights, inputs and noise for parity tests and the benchmark.

No trained checkpoints ship with the reference, so every parity
fixture is built from *formula* weights: ``numpy.random.RandomState`` streams keyed by the
state_dict entry name.  The GPU box regenerates the multi-MB weights itself from this file;
only the small golden outputs are committed under ``tests/golden/``.

The input recipe follows SURVEY.md §8(d) "Synthetic inputs".
"""
import zlib

import numpy as np

# Buffers that are *computed* (schedule) or configured (spec range) and therefore never
# replaced by synthetic values (reference: usr/diff/shallow_diffusion_tts.py:103-126,
# modules/commons/common_layers.py:121 `_float_tensor`).
COMPUTED_BUFFERS = (
    'betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
    'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod',
    'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_variance',
    'posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2',
    'spec_min', 'spec_max', '_float_tensor',
)


def is_computed_buffer(key):
    return key.split('.')[-1] in COMPUTED_BUFFERS


def canonical_key(key):
    """The reference registers two modules twice (FastSpeech2MIDI.esm is also encoder.esm,
    encoder_embed_tokens is also encoder.embed_tokens: diffsinger_midi/fs2.py:84-87,
    fastspeech/fs2.py:31-32), so their tensors appear under two state_dict names."""
    return key.replace('encoder.esm.', 'esm.').replace('encoder.embed_tokens.', 'encoder_embed_tokens.')


def _rs(key, seed):
    key = canonical_key(key)
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def synth_tensor(key, shape, seed=0):
    """One float32 array for state_dict entry ``key`` of ``shape``.

    Scales are chosen so activations stay O(1) through the 20 residual layers / 8 FFT layers:
      * matrices / conv kernels: N(0, 1/fan_in)  (fan_in = prod(shape[1:]))
      * embeddings:              N(0, 1/dim)      (reference init, common_layers.py:80-81)
      * 1-D ``weight`` (LayerNorm gain), ``weight_g``, ``pos_embed_alpha``: 1 + 0.1 N(0,1)
      * biases:                  0.1 N(0,1)
    """
    rs = _rs(key, seed)
    shape = tuple(int(s) for s in shape)
    leaf = key.split('.')[-1]
    n = rs.standard_normal(shape).astype(np.float32)
    if leaf == 'pos_embed_alpha':
        return (1.0 + 0.1 * n).astype(np.float32)
    if leaf == 'weight_g':
        # weight-norm gain g: folded weight = g*v/||v||, rows of unit norm -> per-element
        # variance 1/fan_in, the same scale as the plain matrices below
        return (1.0 + 0.1 * n).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if 'embed' in key and len(shape) == 2 and 'proj' not in leaf:
            std = shape[1] ** -0.5
        elif leaf == 'weight_v':
            std = 1.0  # direction only; weight_g carries the scale (see fold below)
        else:
            std = fan_in ** -0.5
        return (n * np.float32(std)).astype(np.float32)
    if leaf == 'weight':          # LayerNorm gain
        return (1.0 + 0.1 * n).astype(np.float32)
    return (0.1 * n).astype(np.float32)  # biases


def synth_state_dict(spec, seed=0, gain=None):
    """``spec``: mapping key -> shape.  Returns key -> float32 ndarray for every entry that is
    not a computed buffer.  ``gain``: optional mapping substring -> multiplier applied to the
    matching *weights* (used to keep the synthetic denoiser well-conditioned)."""
    out = {}
    for k, shp in spec.items():
        if is_computed_buffer(k):
            continue
        if k.endswith('num_batches_tracked'):
            continue
        a = synth_tensor(k, shp, seed)
        if gain:
            for sub, g in gain.items():
                if sub in k:
                    a = (a * np.float32(g)).astype(np.float32)
        out[k] = a
    return out


# Gains applied on top of synth_tensor for the denoiser.  A random 20-layer gated network has
# a large Lipschitz constant; trained denoisers do not.  These multipliers keep
# d(eps)/d(x) small enough that a 100-step ancestral trajectory is not chaotic, so that
# fp32 summation-order differences (MFMA k-order vs oneDNN) stay ~1e-5 on the mel.
DIFFNET_GAIN = {
    'denoise_fn.output_projection.weight': 0.5,
}


def synth_inputs(B, T_txt, T, seed=1, vocab=65, num_spk=21, ragged=False):
    """SURVEY.md §8(d) synthetic batch.  Returns dict of numpy arrays:
    txt_tokens[B,T_txt] i64 in [3,vocab), pitch_midi [40,80), midi_dur U(0,1) f32, is_slur, lang
    Bernoulli(1/2), speechsing=1, spk_embed [0,num_spk), mel2ph[b,j] = j*T_txt//T + 1.
    ``ragged=True`` pads row b>0 (token id 0 / mel2ph 0) to exercise the padding masks."""
    rs = np.random.RandomState(seed)
    d = dict(
        txt_tokens=rs.randint(3, vocab, size=(B, T_txt)).astype(np.int64),
        pitch_midi=rs.randint(40, 80, size=(B, T_txt)).astype(np.int64),
        midi_dur=rs.uniform(0, 1, size=(B, T_txt)).astype(np.float32),
        is_slur=rs.randint(0, 2, size=(B, T_txt)).astype(np.int64),
        lang=rs.randint(0, 2, size=(B, T_txt)).astype(np.int64),
        speechsing=np.ones((B,), np.int64),
        spk_embed=rs.randint(0, num_spk, size=(B,)).astype(np.int64),
    )
    mel2ph = (np.arange(T)[None, :] * T_txt // T + 1).astype(np.int64).repeat(B, 0)
    if ragged:
        for b in range(1, B):
            n_tok = max(2, T_txt - (b * 3) % max(1, T_txt // 2))
            d['txt_tokens'][b, n_tok:] = 0
            d['pitch_midi'][b, n_tok:] = 0
            d['midi_dur'][b, n_tok:] = 0
            d['is_slur'][b, n_tok:] = 0
            d['lang'][b, n_tok:] = 0
            n_frm = (T * n_tok) // T_txt
            mel2ph[b] = np.minimum(np.arange(T) * T_txt // T + 1, n_tok)
            mel2ph[b, n_frm:] = 0
    d['mel2ph'] = mel2ph
    return d


def synth_noise(steps, B, M, T, seed=1):
    """Host-supplied sampler noise, parity mode (SURVEY.md §7 hard part 2):
    noise[0] = x_T, noise[1 + k] = the N(0,1) draw of the k-th executed p_sample
    (k = 0 is timestep t = steps-1).  Shape [steps+1, B, M, T] float32."""
    rs = np.random.RandomState(seed)
    return rs.standard_normal((steps + 1, B, M, T)).astype(np.float32)


# --------------------------------------------------------------------------------------
# Philox4x32-10 + Box-Muller: the bench-mode noise stream.  The HIP sampler kernel
# (csrc/sampler.hip) implements the same counter layout, so the stream can be reproduced
# on the host for parity checks in Philox mode.
# --------------------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(ctr, key):
    """ctr: uint32[..., 4], key: uint32[..., 2] -> uint32[..., 4]."""
    c = [ctr[..., i].astype(np.uint32) for i in range(4)]
    k0 = key[..., 0].astype(np.uint32)
    k1 = key[..., 1].astype(np.uint32)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = c[0].astype(np.uint64) * _PHILOX_M0
            p1 = c[2].astype(np.uint64) * _PHILOX_M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + _PHILOX_W0).astype(np.uint32)
            k1 = (k1 + _PHILOX_W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox_normal(seed, stream, n):
    """n standard normals for (seed, stream): element i uses counter (i//4, stream, 0, 0),
    lane i%4; Box-Muller pairs (0,1)->(z0,z1), (2,3)->(z2,z3)."""
    nq = (n + 3) // 4
    ctr = np.zeros((nq, 4), np.uint32)
    ctr[:, 0] = np.arange(nq, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = np.uint32(stream)
    key = np.zeros((nq, 2), np.uint32)
    key[:, 0] = np.uint32(seed & 0xFFFFFFFF)
    key[:, 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10(ctr, key)
    # uniform in (0,1]: (x + 1) * 2^-32 ; computed in float32 like the kernel
    u = (r.astype(np.float32) + np.float32(1.0)) * np.float32(2.3283064365386963e-10)
    u = np.minimum(u, np.float32(1.0))
    rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 0::2]))
    ang = np.float32(6.283185307179586) * u[:, 1::2]
    z = np.empty((nq, 4), np.float32)
    z[:, 0] = rad[:, 0] * np.cos(ang[:, 0])
    z[:, 1] = rad[:, 0] * np.sin(ang[:, 0])
    z[:, 2] = rad[:, 1] * np.cos(ang[:, 1])
    z[:, 3] = rad[:, 1] * np.sin(ang[:, 1])
    return z.reshape(-1)[:n]
