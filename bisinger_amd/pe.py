"""``PitchExtractor`` — drop-in for modules/fastspeech/pe.py:120-149 (mel -> f0), SURVEY.md §8 row f2, on HIP kernels.

Used by the inference harness when ``pe_enable`` is set to feed the NSF vocoder (a-*.py:600-603, :629-630).
Same module tree / ``state_dict`` (59 entries) as the reference; parameters only, the arithmetic is ``bsg_pitchext_*``.
"""
from ctypes import POINTER, byref, c_void_p, cast

import torch
import torch.nn as nn

from . import _lib
from .fs2 import Linear, SinusoidalPositionalEmbedding, _Holder
from .hparams import hparams


class Prenet(_Holder):
    def __init__(self, in_dim=80, out_dim=256, kernel=5, n_layers=3):
        super().__init__()
        layers = []
        for _ in range(n_layers):
            layers.append(nn.Sequential(nn.Conv1d(in_dim, out_dim, kernel_size=kernel, padding=kernel // 2), nn.ReLU(),
                                        nn.BatchNorm1d(out_dim)))
            in_dim = out_dim
        self.layers = nn.ModuleList(layers)
        self.out_proj = nn.Linear(out_dim, out_dim)


class ConvNorm(_Holder):
    def __init__(self, cin, cout, kernel_size):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, kernel_size=kernel_size, padding=(kernel_size - 1) // 2)


class ConvBlock(_Holder):
    def __init__(self, idim, n_chans, kernel_size):
        super().__init__()
        self.conv = ConvNorm(idim, n_chans, kernel_size)
        self.norm = nn.GroupNorm(n_chans // 16, n_chans)


class ConvStacks(_Holder):
    def __init__(self, idim=80, n_layers=5, n_chans=256, odim=32, kernel_size=5):
        super().__init__()
        self.conv = nn.ModuleList()
        self.in_proj = Linear(idim, n_chans)
        for _ in range(n_layers):
            self.conv.append(ConvBlock(n_chans, n_chans, kernel_size))
        self.out_proj = Linear(n_chans, odim)


class PredictorLayerNorm(nn.LayerNorm):
    def __init__(self, nout):
        super().__init__(nout, eps=1e-12)


class PitchPredictor(_Holder):
    """tts_modules.py:194-231."""

    def __init__(self, idim, n_layers=5, n_chans=384, odim=2, kernel_size=5, dropout_rate=0.1):
        super().__init__()
        self.conv = nn.ModuleList()
        for i in range(n_layers):
            self.conv.append(nn.Sequential(
                nn.ConstantPad1d(((kernel_size - 1) // 2, (kernel_size - 1) // 2), 0),
                nn.Conv1d(idim if i == 0 else n_chans, n_chans, kernel_size, stride=1, padding=0),
                nn.ReLU(), PredictorLayerNorm(n_chans), nn.Dropout(dropout_rate)))
        self.linear = nn.Linear(n_chans, odim)
        self.embed_positions = SinusoidalPositionalEmbedding(idim, 0, init_size=4096)
        self.pos_embed_alpha = nn.Parameter(torch.Tensor([1]))


class PitchExtractor(nn.Module, _lib.HandleOwner, _lib.GemmGuarded):
    GUARD_KIND = 'pitchext'

    def __init__(self, n_mel_bins=80, conv_layers=2):
        super().__init__()
        self.hidden_size = 256
        self.n_mel_bins = n_mel_bins
        ph = hparams['predictor_hidden'] if hparams['predictor_hidden'] > 0 else self.hidden_size
        assert ph == 256 and hparams['ffn_padding'] == 'SAME'
        self.conv_layers = conv_layers
        self.predictor_kernel = hparams['predictor_kernel']
        self.mel_prenet = Prenet(n_mel_bins, self.hidden_size)
        if conv_layers > 0:
            self.mel_encoder = ConvStacks(idim=self.hidden_size, n_chans=self.hidden_size, odim=self.hidden_size, n_layers=conv_layers)
        self.pitch_predictor = PitchPredictor(self.hidden_size, n_chans=ph, n_layers=5, dropout_rate=0.5, odim=2,
                                              kernel_size=self.predictor_kernel)
        self._h = self._h_key = None

    def handle(self):
        key = self._key()
        if self._h is not None and key == self._h_key:
            return self._h
        self.release()
        ws = [p.detach() for p in self._weights()]
        for p in ws:
            if not p.is_cuda or not p.is_contiguous():
                raise _lib.BsgError('PitchExtractor parameters must be contiguous tensors on the GPU; there is no CPU path')
        assert hparams.get('pitch_norm', 'log') == 'log', "only pitch_norm: log is on the BiSinger path"
        use_uv = int(hparams.get('pitch_type', 'frame') == 'frame' and bool(hparams.get('use_uv', True)))
        self._n_pos = max(4096, int(hparams.get('max_frames', 5000)) + 2)
        cfg = _lib.PitchextCfg(256, self.n_mel_bins, self.conv_layers, 5, self.predictor_kernel, use_uv, self._n_pos)
        lib = _lib.load()
        assert lib.bsg_pitchext_n_weights(byref(cfg)) == len(ws), (lib.bsg_pitchext_n_weights(byref(cfg)), len(ws))
        table = self.pitch_predictor.embed_positions.table(self._n_pos).to(ws[0].device).contiguous()
        arr = (c_void_p * len(ws))(*[p.data_ptr() for p in ws])
        h = c_void_p()
        with torch.cuda.device(ws[0].device):
            _lib.check(lib.bsg_pitchext_create(byref(h), byref(cfg), cast(arr, POINTER(c_void_p)), len(ws), _lib.ptr(table),
                                               _lib.stream_ptr()), 'bsg_pitchext_create')
        self._h, self._h_key = h, key
        self._apply_guard_state()
        return h

    def release(self):
        if self._h is not None:
            _lib.load().bsg_pitchext_destroy(self._h)
        self._h = self._h_key = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, mel_input=None):
        """mel [B,T,80] -> {'pitch_pred': [B,T,2], 'f0_denorm_pred': [B,T]}   (pe.py:136-149)."""
        return _lib.range_guarded(lambda: self._forward(mel_input), 'PitchExtractor.forward', device=self, owners=(self,))

    def _forward(self, mel_input):
        h = self.handle()
        mel = mel_input.contiguous().float()
        B, T, M = mel.shape
        if T >= self._n_pos:
            raise _lib.BsgError(f'T={T} exceeds the position table ({self._n_pos})')
        pred = torch.empty(B, T, 2, device=mel.device)
        f0 = torch.empty(B, T, device=mel.device)
        with torch.cuda.device(mel.device):
            _lib.check(_lib.load().bsg_pitchext_forward(h, _lib.ptr(mel), _lib.ptr(pred), _lib.ptr(f0), B, T, _lib.stream_ptr()),
                       'bsg_pitchext_forward')
        return {'pitch_pred': pred, 'f0_denorm_pred': f0}
