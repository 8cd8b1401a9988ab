"""``HifiGanGenerator`` — drop-in for modules/hifigan/hifigan.py:104-182 (generator forward only) on HIP kernels.

Constructed like the reference (``HifiGanGenerator(h)`` with ``h`` = the vocoder config dict); parameters are
registered in the checkpoint layout (``bias, weight_g, weight_v`` per conv) so that
``load_state_dict(ckpt['state_dict']['model_gen'], strict=True)`` followed by ``remove_weight_norm()`` works as in
vocoders/hifigan.py:27-29; an already folded ``weight`` layout loads too.  Discriminators / GAN losses
(hifigan.py:185-369) are training-only and out of scope.
"""
from ctypes import POINTER, byref, c_void_p, cast

import numpy as np
import torch
import torch.nn as nn

from . import _lib

LRELU_SLOPE = 0.1


class _WNConv(nn.Module):
    """Parameter holder for one weight-normed Conv1d / ConvTranspose1d (torch's weight_norm, dim=0)."""

    def __init__(self, weight_shape, n_bias):
        super().__init__()
        self.weight_shape = tuple(weight_shape)
        self.bias = nn.Parameter(torch.zeros(n_bias))
        self.weight_g = nn.Parameter(torch.ones(weight_shape[0], *([1] * (len(weight_shape) - 1))))
        self.weight_v = nn.Parameter(torch.empty(*weight_shape).normal_(0.0, 0.01))

    @property
    def folded(self):
        return 'weight' in self._parameters

    def fold(self):
        """remove_weight_norm: weight = g * v / ||v|| (computed by the library on the GPU, by torch on the CPU)."""
        if self.folded:
            return
        g, v = self.weight_g.detach(), self.weight_v.detach()
        if v.is_cuda:
            w = torch.empty_like(v)
            with torch.cuda.device(v.device):
                _lib.check(_lib.load().bsg_weight_norm_fold(_lib.ptr(g.contiguous()), _lib.ptr(v.contiguous()), _lib.ptr(w),
                                                            v.shape[0], v[0].numel(), _lib.stream_ptr()), 'bsg_weight_norm_fold')
        else:  # host-side model surgery before .cuda(); not on the inference path
            w = v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))
        del self._parameters['weight_g'], self._parameters['weight_v']
        self.register_parameter('weight', nn.Parameter(w))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kw):
        # accept the other layout than the one currently registered
        if prefix + 'weight' in state_dict and not self.folded:
            del self._parameters['weight_g'], self._parameters['weight_v']
            self.register_parameter('weight', nn.Parameter(torch.empty(self.weight_shape, device=self.bias.device)))
        elif prefix + 'weight_v' in state_dict and self.folded:
            dev = self.bias.device
            del self._parameters['weight']
            self.register_parameter('weight_g', nn.Parameter(torch.ones(self.weight_shape[0], *([1] * (len(self.weight_shape) - 1)), device=dev)))
            self.register_parameter('weight_v', nn.Parameter(torch.empty(self.weight_shape, device=dev)))
        super()._load_from_state_dict(state_dict, prefix, *args, **kw)


class SourceModuleHnNSF(nn.Module):
    """parallel_wavegan/models/source.py:352-399 (parameters: the harmonic merge Linear)."""

    def __init__(self, harmonic_num):
        super().__init__()
        self.l_linear = nn.Linear(harmonic_num + 1, 1)


class ResBlock1(nn.Module):
    """hifigan.py:30-52 (parameters only)."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels) for _ in dilation])
        self.convs2 = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels) for _ in dilation])

    def remove_weight_norm(self):
        for c in list(self.convs1) + list(self.convs2):
            c.fold()


class ResBlock2(nn.Module):
    """hifigan.py:70-91 (parameters only): per dilation ONE conv, x = conv_d(lrelu(x)) + x."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.convs = nn.ModuleList([_WNConv((channels, channels, kernel_size), channels) for _ in dilation])

    def remove_weight_norm(self):
        for c in self.convs:
            c.fold()


class HifiGanGenerator(nn.Module, _lib.HandleOwner, _lib.GemmGuarded):
    GUARD_KIND = 'hifigan'

    def __init__(self, h, c_out=1):
        super().__init__()
        self.h = h
        self.use_nsf = bool(h.get('use_pitch_embed'))
        self.resblock = 1 if str(h['resblock']) == '1' else 2      # hifigan.py:117
        assert c_out == 1
        self.num_kernels = len(h['resblock_kernel_sizes'])
        self.num_upsamples = len(h['upsample_rates'])
        C0 = h['upsample_initial_channel']
        if self.use_nsf:      # hifigan.py:111-132: SourceModuleHnNSF(harmonic_num=8) + one noise conv per stage
            self.harmonic_num = 8
            self.m_source = SourceModuleHnNSF(self.harmonic_num)
            self.noise_convs = nn.ModuleList()
            rates = list(h['upsample_rates'])
            for i in range(len(rates)):
                c_cur = C0 // (2 ** (i + 1))
                if i + 1 < len(rates):
                    s = int(np.prod(rates[i + 1:]))
                    self.noise_convs.append(nn.Conv1d(1, c_cur, kernel_size=s * 2, stride=s, padding=s // 2))
                else:
                    self.noise_convs.append(nn.Conv1d(1, c_cur, kernel_size=1))
        self.conv_pre = _WNConv((C0, 80, 7), C0)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
            c_cur = C0 // (2 ** (i + 1))
            self.ups.append(_WNConv((c_cur * 2, c_cur, k), c_cur))      # ConvTranspose1d weight [Cin, Cout, K]
        self.resblocks = nn.ModuleList()
        ch = C0
        for i in range(len(self.ups)):
            ch = C0 // (2 ** (i + 1))
            for k, d in zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes']):
                self.resblocks.append((ResBlock1 if self.resblock == 1 else ResBlock2)(h, ch, k, d))
        self.conv_post = _WNConv((c_out, ch, 7), c_out)
        self._h = None
        self._h_key = None

    def remove_weight_norm(self):
        print('Removing weight norm...')
        for l in self.ups:
            l.fold()
        for l in self.resblocks:
            l.remove_weight_norm()
        self.conv_pre.fold()
        self.conv_post.fold()
        self.__dict__.pop('_handle_slots', None)      # the parameters changed NAMES (weight_g / weight_v -> weight): look the slots up again

    # ------------------------------------------------------------------ handle
    def handle(self):
        key = self._key()
        if self._h is not None and key == self._h_key:
            return self._h
        self.release()
        ws = [p.detach() for p in self._weights()]
        for p in ws:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.BsgError('HifiGanGenerator parameters must be contiguous float32 on the GPU; there is no CPU path')
        states = {c.folded for c in self.modules() if isinstance(c, _WNConv)}
        if len(states) != 1:
            raise _lib.BsgError('mixed weight-norm / folded layers')
        h = self.h
        cfg = _lib.HifiganCfg()
        cfg.n_mel, cfg.upsample_initial_channel, cfg.n_ups = 80, h['upsample_initial_channel'], self.num_upsamples
        for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.n_kernels = self.num_kernels
        cfg.n_dil = len(h['resblock_dilation_sizes'][0])
        for j, (k, d) in enumerate(zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes'])):
            cfg.resblock_kernel_sizes[j] = k
            assert len(d) == cfg.n_dil
            for m, dd in enumerate(d):
                cfg.resblock_dilations[j][m] = dd
        cfg.weight_norm = 0 if states.pop() else 1
        cfg.use_nsf = int(self.use_nsf)
        cfg.sample_rate = int(h.get('audio_sample_rate', 22050))
        cfg.harmonic_num = self.harmonic_num if self.use_nsf else 0
        cfg.resblock = self.resblock
        lib = _lib.load()
        assert lib.bsg_hifigan_n_weights(byref(cfg)) == len(ws), (lib.bsg_hifigan_n_weights(byref(cfg)), len(ws))
        arr = (c_void_p * len(ws))(*[p.data_ptr() for p in ws])
        hd = c_void_p()
        with torch.cuda.device(ws[0].device):
            _lib.check(lib.bsg_hifigan_create(byref(hd), byref(cfg), cast(arr, POINTER(c_void_p)), len(ws), _lib.stream_ptr()),
                       'bsg_hifigan_create')
        self._h, self._h_key = hd, key
        self._apply_guard_state()
        return hd

    def release(self):
        if self._h is not None:
            _lib.load().bsg_hifigan_destroy(self._h)
        self._h = self._h_key = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, x, f0=None, rand_ini=None, noise=None, seed=0):
        """x [B,80,T] (, f0 [B,T]) -> [B,1,T*hop]   (hifigan.py:144-173).
        NSF draws (source.py:53, :130): supplied (``rand_ini`` [B,9], ``noise`` [B,T*hop,9]) or generated
        (numpy RandomState(seed) for the 9 initial phases, the library's Philox stream 0x4E5346 for the noise)."""
        # range guard of the split-fp16 ResBlock pairs (an activation beyond the fp16 range is counted by the kernels, never clipped):
        # read once per outermost call; on an event the call is repeated on the fp32 matrix pipe (_lib.range_guarded)
        return _lib.range_guarded(lambda: self._forward(x, f0, rand_ini, noise, seed), 'HifiGanGenerator.forward', device=self, owners=(self,))

    def _forward(self, x, f0, rand_ini, noise, seed):
        hd = self.handle()
        x = x.contiguous().float()
        B, M, T = x.shape
        assert M == 80
        hop = int(np.prod(self.h['upsample_rates']))
        y = torch.empty(B, 1, T * hop, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            if self.use_nsf:
                if f0 is None:
                    raise _lib.BsgError('this generator has the NSF source (use_pitch_embed): f0 is required')
                NH = self.harmonic_num + 1
                f0 = f0.to(x.device, torch.float32).contiguous()
                if rand_ini is None:       # 9 initial phases per utterance: host draw, reproducible from the seed
                    rand_ini = torch.from_numpy(np.random.RandomState(seed & 0x7FFFFFFF).uniform(size=(B, NH)).astype(np.float32))
                if noise is None:
                    noise = torch.empty(B, T * hop, NH, device=x.device)
                    _lib.check(lib.bsg_philox_normal(_lib.ptr(noise), noise.numel(), seed, 0x4E5346, 0, _lib.stream_ptr()), 'bsg_philox_normal')
                rand_ini = rand_ini.to(x.device, torch.float32).contiguous()
                noise = noise.to(x.device, torch.float32).contiguous()
                assert tuple(rand_ini.shape) == (B, NH) and tuple(noise.shape) == (B, T * hop, NH) and tuple(f0.shape) == (B, T)
                _lib.check(lib.bsg_hifigan_forward_nsf(hd, _lib.ptr(x), _lib.ptr(f0), _lib.ptr(rand_ini), _lib.ptr(noise), _lib.ptr(y),
                                                       B, T, _lib.stream_ptr()), 'bsg_hifigan_forward_nsf')
            else:
                if f0 is not None:
                    raise _lib.BsgError('f0 given but this generator was built without use_pitch_embed')
                _lib.check(lib.bsg_hifigan_forward(hd, _lib.ptr(x), _lib.ptr(y), B, T, _lib.stream_ptr()), 'bsg_hifigan_forward')
        return y
