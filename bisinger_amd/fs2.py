"""``FastSpeech2MIDI`` — drop-in for modules/diffsinger_midi/fs2.py:80-197 (+ the base
modules/fastspeech/fs2.py:24-89 constructor), inference direction, on HIP kernels.

The sub-modules below reproduce the reference's module tree so that ``state_dict()`` has exactly the
reference's 143 ``fs2.*`` entries (names, shapes, order — including the two aliased sub-trees
``encoder.esm``/``esm`` and ``encoder.embed_tokens``/``encoder_embed_tokens`` and the
``decoder.embed_positions._float_tensor`` buffer).  They only hold parameters: the arithmetic is
``bsg_fs2midi_*`` in libbisinger_hip (csrc/fs2.hip).
"""
import math
from ctypes import POINTER, byref, c_void_p, cast

import torch
import torch.nn as nn

from . import _lib
from .hparams import hparams

DEFAULT_MAX_TARGET_POSITIONS = 2000


def Embedding(num_embeddings, embedding_dim, padding_idx=None):
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0, std=embedding_dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    return m


def Linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.0)
    return m


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise _lib.BsgError(f'{type(self).__name__} only holds parameters; it runs inside libbisinger_hip')


class MultiheadAttention(_Holder):
    """common_layers.py:199-280 with bias=False: in_proj_weight [3C,C], out_proj.weight [C,C]."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)


class TransformerFFNLayer(_Holder):
    """common_layers.py:598-644 (padding SAME, gelu)."""

    def __init__(self, hidden_size, filter_size, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.ffn_1 = nn.Conv1d(hidden_size, filter_size, kernel_size, padding=kernel_size // 2)
        self.ffn_2 = Linear(filter_size, hidden_size)


class EncSALayer(_Holder):
    """common_layers.py:664-704."""

    def __init__(self, c, num_heads, kernel_size):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c)
        self.self_attn = MultiheadAttention(c, num_heads)
        self.layer_norm2 = nn.LayerNorm(c)
        self.ffn = TransformerFFNLayer(c, 4 * c, kernel_size)


class TransformerEncoderLayer(_Holder):
    def __init__(self, hidden_size, kernel_size, num_heads):
        super().__init__()
        self.op = EncSALayer(hidden_size, num_heads, kernel_size)


class SinusoidalPositionalEmbedding(_Holder):
    """common_layers.py:106-179.  ``table`` builds the lookup the decoder kernels index by position."""

    def __init__(self, embedding_dim, padding_idx, init_size=1024):
        super().__init__()
        self.embedding_dim, self.padding_idx, self.init_size = embedding_dim, padding_idx, init_size
        self.register_buffer('_float_tensor', torch.zeros(1))

    def table(self, num):
        half = self.embedding_dim // 2
        e = math.log(10000) / (half - 1)
        e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
        e = torch.arange(num, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
        e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num, -1)
        if self.padding_idx is not None:
            e[self.padding_idx, :] = 0
        return e


class RelPositionalEncoding(_Holder):
    """espnet_positional_embedding.py:90-114 (reverse=True table of :25-46); no parameters."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.d_model, self.max_len = d_model, max_len

    def table(self, length):
        pe = torch.zeros(length, self.d_model)
        pos = torch.arange(length - 1, -1, -1.0, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d_model))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return pe


class FFTBlocks(_Holder):
    """tts_modules.py:253-282."""

    def __init__(self, hidden_size, num_layers, ffn_kernel_size=9, num_heads=2, use_pos_embed=True):
        super().__init__()
        self.num_layers, self.hidden_size, self.use_pos_embed = num_layers, hidden_size, use_pos_embed
        if use_pos_embed:
            self.padding_idx = 0
            self.pos_embed_alpha = nn.Parameter(torch.Tensor([1]))
            self.embed_positions = SinusoidalPositionalEmbedding(hidden_size, 0, init_size=DEFAULT_MAX_TARGET_POSITIONS)
        self.layers = nn.ModuleList([TransformerEncoderLayer(hidden_size, ffn_kernel_size, num_heads)
                                     for _ in range(num_layers)])
        self.layer_norm = nn.LayerNorm(hidden_size)


class FastspeechDecoder(FFTBlocks):
    def __init__(self):
        super().__init__(hparams['hidden_size'], hparams['dec_layers'], hparams['dec_ffn_kernel_size'], hparams['num_heads'])


class ESM(_Holder):
    """common_layers.py:832-846."""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.mh = nn.MultiheadAttention(d_model, nhead)
        self.ffn = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.ln1 = nn.LayerNorm(d_model)
        self.ln2 = nn.LayerNorm(d_model)


class FastspeechMIDIEncoder(FFTBlocks):
    """diffsinger_midi/fs2.py:14-17 on tts_modules.py:312-328."""

    def __init__(self, esm, embed_tokens):
        super().__init__(hparams['hidden_size'], hparams['enc_layers'], hparams['enc_ffn_kernel_size'],
                         hparams['num_heads'], use_pos_embed=False)
        self.embed_tokens = embed_tokens
        self.embed_scale = math.sqrt(hparams['hidden_size'])
        self.padding_idx = 0
        self.embed_positions = RelPositionalEncoding(hparams['hidden_size'])
        self.esm = esm


class PredictorLayerNorm(nn.LayerNorm):
    """tts_modules.py:39-58: LayerNorm over the channel axis of [B,C,T], eps = 1e-12."""

    def __init__(self, nout):
        super().__init__(nout, eps=1e-12)


class DurationPredictor(_Holder):
    """tts_modules.py:61-106 (dur_loss = mse)."""

    def __init__(self, idim, n_layers, n_chans, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv = nn.ModuleList()
        for i in range(n_layers):
            self.conv.append(nn.Sequential(
                nn.ConstantPad1d(((kernel_size - 1) // 2, (kernel_size - 1) // 2), 0),
                nn.Conv1d(idim if i == 0 else n_chans, n_chans, kernel_size, stride=1, padding=0),
                nn.ReLU(), PredictorLayerNorm(n_chans), nn.Dropout(hparams['predictor_dropout'])))
        assert hparams['dur_loss'] == 'mse', 'only dur_loss: mse is on the BiSinger path'
        self.linear = nn.Linear(n_chans, 1)


class FastSpeech2MIDI(nn.Module, _lib.HandleOwner, _lib.GemmGuarded):
    GUARD_KIND = 'fs2midi'

    def __init__(self, dictionary, out_dims=None):
        super().__init__()
        hp = hparams
        assert hp['encoder_type'] == 'fft' and hp['decoder_type'] == 'fft'
        assert hp['use_spk_id'] and not hp['use_spk_embed'] and not hp.get('use_split_spk_id'), \
            'BiSinger path: use_spk_id speaker table (SURVEY.md §8)'
        assert not hp['use_pitch_embed'] and not hp['use_energy_embed'], \
            'pitch/energy predictors are not executed by any BiSinger config (SURVEY.md §2 row 5)'
        assert hp['ffn_act'] == 'gelu' and hp['ffn_padding'] == 'SAME' and hp['use_pos_embed'] and hp.get('rel_pos')
        self.dictionary = dictionary
        self.padding_idx = dictionary.pad()
        self.enc_layers, self.dec_layers = hp['enc_layers'], hp['dec_layers']
        self.hidden_size = H = hp['hidden_size']
        self.encoder_embed_tokens = Embedding(len(dictionary), H, self.padding_idx)
        self.decoder = FastspeechDecoder()
        self.out_dims = out_dims if out_dims is not None else hp['audio_num_mel_bins']
        self.mel_out = Linear(H, self.out_dims, bias=True)
        self.spk_embed_proj = Embedding(hp['num_spk'] + 1, H)
        ph = hp['predictor_hidden'] if hp['predictor_hidden'] > 0 else H
        self.dur_predictor = DurationPredictor(H, hp['dur_predictor_layers'], ph, hp['dur_predictor_kernel'])
        self.esm = ESM(d_model=H, nhead=8)
        self.encoder = FastspeechMIDIEncoder(self.esm, self.encoder_embed_tokens)
        self.midi_embed = Embedding(300, H, self.padding_idx)
        self.midi_dur_layer = Linear(1, H)
        self.is_slur_embed = Embedding(2, H)
        self.lang_embed = Embedding(2, H)
        self.style_embed = Embedding(3, H)
        self._h = None
        self._h_key = None

    # ------------------------------------------------------------------ handle management
    def handle(self):
        key = self._key()
        if self._h is not None and key == self._h_key:
            return self._h
        self.release()
        ws = [p.detach() for p in self._weights()]
        for p in ws:
            if not p.is_cuda:
                raise _lib.BsgError('FastSpeech2MIDI parameters must live on the GPU (model.cuda()); there is no CPU path')
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.BsgError('FastSpeech2MIDI parameters must be contiguous float32')
        hp = hparams
        lib = _lib.load()
        self._n_pos = max(DEFAULT_MAX_TARGET_POSITIONS, int(hp.get('max_frames', 5000))) + 2
        self._n_rel = self.encoder.embed_positions.max_len
        cfg = _lib.Fs2Cfg(self.hidden_size, self.encoder_embed_tokens.num_embeddings, self.enc_layers, self.dec_layers,
                          hp['num_heads'], hp['enc_ffn_kernel_size'], hp['dec_ffn_kernel_size'], self.out_dims,
                          hp['dur_predictor_layers'], hp['dur_predictor_kernel'], self.spk_embed_proj.num_embeddings, 8,
                          self._n_pos, self._n_rel)
        assert lib.bsg_fs2midi_n_weights(byref(cfg)) == len(ws)
        dev = ws[0].device
        dec_table = self.decoder.embed_positions.table(self._n_pos).to(dev).contiguous()
        rel_table = self.encoder.embed_positions.table(self._n_rel).to(dev).contiguous()
        arr = (c_void_p * len(ws))(*[p.data_ptr() for p in ws])
        h = c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.bsg_fs2midi_create(byref(h), byref(cfg), cast(arr, POINTER(c_void_p)), len(ws),
                                              _lib.ptr(dec_table), _lib.ptr(rel_table), _lib.stream_ptr()),
                       'bsg_fs2midi_create')
        self._h, self._h_key = h, key
        self._apply_guard_state()
        return h

    def release(self):
        if self._h is not None:
            _lib.load().bsg_fs2midi_destroy(self._h)
        self._h = self._h_key = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference forward (fs2.py:94-197)
    @torch.no_grad()
    def encode(self, txt_tokens, spk_embed, predict_dur=False, rows=None, **kwargs):
        """Token-level front: embeddings + ESM + FFT encoder (+ duration predictor)   (fs2.py:111-165).
        ESM attends over the batch axis (common_layers.py:853): the inputs are always the WHOLE batch's.  ``rows`` (a contiguous slice,
        extension): the outputs for these batch rows only — K / V of the ESM (projections of LN(lang_embed[lang]), :850-853, the only thing
        other rows contribute) are still formed for every row; Q, the ESM's FFN, the encoder and the duration predictor run on the rows
        asked for (bsg_fs2midi_encode_rows; SURVEY.md §8e: a rank of a sharded batch encodes its own utterances, not everybody's)."""
        lib = _lib.load()
        h = self.handle()
        dev = txt_tokens.device
        i64 = lambda t: t.to(device=dev, dtype=torch.long).contiguous()
        txt = i64(txt_tokens)
        B, Tt = txt.shape
        if Tt > self._n_rel:
            raise _lib.BsgError(f'T_txt={Tt} exceeds the positional table ({self._n_rel})')
        if kwargs.get('midi_dur') is None or kwargs.get('is_slur') is None:
            raise _lib.BsgError('midi_dur and is_slur are required (every BiSinger entry point passes them)')
        pitch_midi, lang, is_slur = i64(kwargs['pitch_midi']), i64(kwargs['lang']), i64(kwargs['is_slur'])
        midi_dur = kwargs['midi_dur'].to(dev, torch.float32).contiguous()
        spk = i64(spk_embed)
        row0, nb = 0, B
        if rows is not None:
            row0, stop, stride = rows.indices(B)
            if stride != 1 or stop <= row0:
                raise _lib.BsgError('rows must be a contiguous non-empty slice')
            nb = stop - row0
        enc_out = torch.empty(nb, Tt, self.hidden_size, device=dev)
        dur_xs = torch.empty(nb, Tt, device=dev) if predict_dur else None
        dur = torch.empty(nb, Tt, dtype=torch.long, device=dev) if predict_dur else None
        with torch.cuda.device(dev):
            if rows is None:
                _lib.check(lib.bsg_fs2midi_encode(h, _lib.ptr(txt), _lib.ptr(pitch_midi), _lib.ptr(midi_dur), _lib.ptr(is_slur),
                                                  _lib.ptr(lang), _lib.ptr(spk), B, Tt, _lib.ptr(enc_out), _lib.ptr(dur_xs),
                                                  _lib.ptr(dur), _lib.stream_ptr()), 'bsg_fs2midi_encode')
            else:
                _lib.check(lib.bsg_fs2midi_encode_rows(h, _lib.ptr(txt), _lib.ptr(pitch_midi), _lib.ptr(midi_dur), _lib.ptr(is_slur),
                                                       _lib.ptr(lang), _lib.ptr(spk), B, Tt, row0, nb, _lib.ptr(enc_out),
                                                       _lib.ptr(dur_xs), _lib.ptr(dur), _lib.stream_ptr()), 'bsg_fs2midi_encode_rows')
        return dict(enc_out=enc_out, txt=txt, spk=spk, dur_xs=dur_xs, dur=dur)

    def last_rows(self):
        """-> (token rows the last encode ran its encoder on, rows of the last FFT stack): test introspection."""
        from ctypes import c_int32
        a, b = c_int32(), c_int32()
        _lib.check(_lib.load().bsg_fs2midi_last_rows(self._h if self._h is not None else self.handle(), byref(a), byref(b)), 'bsg_fs2midi_last_rows')
        return a.value, b.value

    @torch.no_grad()
    def regulate(self, enc):
        """LengthRegulator on the predicted durations (tts_modules.py:161-191); one host sync, as in the
        reference (the output length is data dependent, :182)."""
        dur, txt = enc['dur'], enc['txt']
        B, Tt = txt.shape
        T = int((dur * (txt != 0)).sum(-1).max().item())
        if T <= 0:
            raise _lib.BsgError('duration predictor produced an empty utterance')
        mel2ph = torch.empty(B, T, dtype=torch.long, device=txt.device)
        with torch.cuda.device(txt.device):
            _lib.check(_lib.load().bsg_length_regulator(_lib.ptr(dur), _lib.ptr(txt), _lib.ptr(mel2ph), B, Tt, T,
                                                        _lib.stream_ptr()), 'bsg_length_regulator')
        return mel2ph

    @torch.no_grad()
    def decode(self, enc_out, mel2ph, spk, speechsing, skip_decoder=False):
        """Frame-level part: gather by mel2ph, +spk +style, mask; FFT decoder + mel_out   (fs2.py:166-195)."""
        dev = enc_out.device
        h = self.handle()
        B, Tt, _ = enc_out.shape
        mel2ph = mel2ph.to(device=dev, dtype=torch.long).contiguous()
        T = mel2ph.shape[1]
        if T >= self._n_pos:
            raise _lib.BsgError(f'T={T} exceeds the decoder position table ({self._n_pos})')
        speechsing = speechsing.to(device=dev, dtype=torch.long).contiguous()
        decoder_inp = torch.empty(B, T, self.hidden_size, device=dev)
        mel_out = None if skip_decoder else torch.empty(B, T, self.out_dims, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().bsg_fs2midi_decode(h, _lib.ptr(enc_out.contiguous()), _lib.ptr(mel2ph), _lib.ptr(spk.contiguous()),
                                                      _lib.ptr(speechsing), B, Tt, T, _lib.ptr(decoder_inp), _lib.ptr(mel_out),
                                                      _lib.stream_ptr()), 'bsg_fs2midi_decode')
        return decoder_inp, mel_out

    def forward(self, txt_tokens, mel2ph=None, spk_embed=None, ref_mels=None, f0=None, uv=None, energy=None,
                skip_decoder=False, spk_embed_dur_id=None, spk_embed_f0_id=None, infer=False, rows=None, **kwargs):
        """``rows`` (slice, extension): generate only these batch rows — what a rank of a sharded run asks for.  The inputs
        stay the WHOLE batch's: the ESM attends over the batch axis, so every row's ``lang`` enters this rank's K / V; with ``mel2ph``
        given nothing else of the other rows is computed (``encode(rows=...)``).  With predicted durations the frame count T is the
        maximum over the whole batch (tts_modules.py:182), so the token-level front then runs on every row and is sliced afterwards."""
        return _lib.range_guarded(lambda: self._forward(txt_tokens, mel2ph, spk_embed, skip_decoder, rows, **kwargs),
                                  'FastSpeech2MIDI.forward', device=self, owners=(self,))

    def _forward(self, txt_tokens, mel2ph, spk_embed, skip_decoder, rows, **kwargs):
        ret = {}
        local = rows is not None and mel2ph is not None      # the token front on this rank's rows only
        enc = self.encode(txt_tokens, spk_embed, predict_dur=mel2ph is None, rows=rows if local else None, **kwargs)
        if mel2ph is None:
            mel2ph = self.regulate(enc)
            ret['dur'] = enc['dur_xs'][:, :, None]
            ret['dur_choice'] = enc['dur']
        # (with mel2ph given the reference also runs the predictor for its training loss; inference does not use it)
        enc_out, spk, speechsing = enc['enc_out'], enc['spk'], kwargs['speechsing']
        if rows is not None:
            if not local:
                enc_out = enc_out[rows]
            spk, speechsing, mel2ph = spk[rows], speechsing[rows], mel2ph[rows]
        ret['mel2ph'] = mel2ph
        ret['decoder_inp'], mel_out = self.decode(enc_out, mel2ph, spk, speechsing, skip_decoder)
        if not skip_decoder:
            ret['mel_out'] = mel_out
        return ret
