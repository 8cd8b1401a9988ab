"""``FFT`` candidate denoiser — drop-in for usr/diff/candidate_decoder.py:39-100 (DIFF_DECODERS['fft']), SURVEY.md §8 row f4.

Same constructor ``FFT(hidden_size, num_layers, kernel_size, num_heads)``, state_dict (49 entries at 4 layers) and
``denoise_fn(spec [B,1,M,T], t [B], cond [B,H,T])`` contract as the reference; arithmetic in ``bsg_fftden_*``.
Not selected by any shipped BiSinger config (all use 'wavenet'), so its sampler loop is driven from Python:
one ``bsg_fftden_forward`` + ``bsg_ddpm_step`` per step (see GaussianDiffusion.sample).
"""
from ctypes import POINTER, byref, c_void_p, cast

import torch
import torch.nn as nn

from . import _lib
from .diffnet import Mish, SinusoidalPosEmb, _conv1d
from .fs2 import FFTBlocks
from .hparams import hparams


class FFT(FFTBlocks, _lib.HandleOwner, _lib.GemmGuarded):
    GUARD_KIND = 'fftden'

    def __init__(self, hidden_size=None, num_layers=None, kernel_size=None, num_heads=None):
        num_heads = hparams['num_heads'] if num_heads is None else num_heads
        hidden_size = hparams['hidden_size'] if hidden_size is None else hidden_size
        kernel_size = hparams['dec_ffn_kernel_size'] if kernel_size is None else kernel_size
        num_layers = hparams['dec_layers'] if num_layers is None else num_layers
        super().__init__(hidden_size, num_layers, kernel_size, num_heads=num_heads)
        dim = hparams['residual_channels']
        assert dim == 256 and hidden_size == 256
        self.in_dims = hparams['audio_num_mel_bins']
        self.num_heads, self.kernel_size = num_heads, kernel_size
        self.max_steps = max(int(hparams.get('timesteps', 1000)), 1000)
        self.input_projection = _conv1d(self.in_dims, dim, 1)
        self.diffusion_embedding = SinusoidalPosEmb(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, dim * 4), Mish(), nn.Linear(dim * 4, dim))
        self.get_mel_out = nn.Linear(hidden_size, 80, bias=True)
        self.get_decode_inp = nn.Linear(hidden_size + dim + dim, hidden_size)
        self._h = self._h_key = self._bound = None

    def handle(self):
        key = self._key()
        if self._h is not None and key == self._h_key:
            return self._h
        self.release()
        ws = [p.detach() for p in self._weights()]
        for p in ws:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.BsgError('FFT parameters must be contiguous float32 on the GPU; there is no CPU path')
        lib = _lib.load()
        assert lib.bsg_fftden_n_weights(self.num_layers) == len(ws), (lib.bsg_fftden_n_weights(self.num_layers), len(ws))
        dev = ws[0].device
        self._n_pos = max(2000, int(hparams.get('max_frames', 5000))) + 2
        step_table = self.diffusion_embedding.table(self.max_steps).to(dev).contiguous()
        pos_table = self.embed_positions.table(self._n_pos).to(dev).contiguous()
        arr = (c_void_p * len(ws))(*[p.data_ptr() for p in ws])
        h = c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.bsg_fftden_create(byref(h), self.in_dims, self.num_layers, self.num_heads, self.kernel_size,
                                             self.max_steps, self._n_pos, cast(arr, POINTER(c_void_p)), len(ws),
                                             _lib.ptr(step_table), _lib.ptr(pos_table), _lib.stream_ptr()), 'bsg_fftden_create')
        self._h, self._h_key, self._bound = h, key, None
        self._apply_guard_state()
        return h

    def release(self):
        if self._h is not None:
            _lib.load().bsg_fftden_destroy(self._h)
        self._h = self._h_key = self._bound = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def prepare(self, cond):
        h = self.handle()
        cond = cond.contiguous().float()
        B, H, T = cond.shape
        with torch.cuda.device(cond.device):
            _lib.check(_lib.load().bsg_fftden_prepare(h, _lib.ptr(cond), B, T, _lib.stream_ptr()), 'bsg_fftden_prepare')
        self._bound = (cond, cond._version, B, T)      # strong reference (see DiffNet._ensure_bound)
        return B, T

    def _ensure_bound(self, cond):
        b = self._bound
        if (self._h is None or self._key() != self._h_key or b is None or b[0] is not cond or b[1] != cond._version
                or not cond.is_contiguous() or cond.dtype != torch.float32):
            self.prepare(cond)
            if cond.is_contiguous() and cond.dtype == torch.float32:
                self._bound = (cond, cond._version, cond.shape[0], cond.shape[2])

    @torch.no_grad()
    def forward(self, spec, diffusion_step, cond, padding_mask=None, attn_mask=None, return_hiddens=False):
        assert padding_mask is None and attn_mask is None and not return_hiddens, 'inference contract only'
        B, _, M, T = spec.shape
        x = spec[:, 0].contiguous().float()
        t = diffusion_step.to(device=x.device, dtype=torch.long).contiguous()
        eps = torch.empty_like(x)

        def run():
            self._ensure_bound(cond)
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().bsg_fftden_forward(self._h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(eps), B, T, _lib.stream_ptr()),
                           'bsg_fftden_forward')

        def again():
            self._bound = None          # the hoisted condition part was projected by the same GEMMs: bind again

        _lib.range_guarded(run, 'FFT denoiser forward', on_retry=again, device=self, owners=(self,))
        return eps[:, None, :, :]
