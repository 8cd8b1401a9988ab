"""Vocoder registry + the HiFi-GAN wrapper — vocoders/base_vocoder.py:6-40 and vocoders/hifigan.py:17-69.

``register_vocoder`` / ``get_vocoder_cls(hparams)`` (short name, or the reference's dotted path
``vocoders.hifigan.HifiGAN``), ``BaseVocoder.spec2wav(mel[T,80]) -> wav[T*hop]``.  ``HifiGAN`` loads
``<vocoder_ckpt>/config.yaml`` + the newest ``model_ckpt_steps_*.ckpt`` (``['state_dict']['model_gen']``, weight-norm
layout, strict) — or, when there is no config.yaml, the original release's ``config.json`` + ``generator_v1``
(``['generator']``) — folds the weight norm and runs the generator on the HIP kernels.  PWG and the spectral
denoiser (vocoder_denoise_c, needs librosa) are out of scope.
"""
import importlib
import os

import numpy as np
import torch

from .ckpt import latest_ckpt
from .hparams import hparams, set_hparams

VOCODERS = {}


def register_vocoder(cls):
    VOCODERS[cls.__name__.lower()] = cls
    VOCODERS[cls.__name__] = cls
    return cls


def get_vocoder_cls(hp):
    name = hp['vocoder']
    if name in VOCODERS:
        return VOCODERS[name]
    if name.split('.')[-1] in VOCODERS and name.startswith('vocoders.'):
        return VOCODERS[name.split('.')[-1]]          # the reference's module path maps onto this registry
    pkg, cls_name = '.'.join(name.split('.')[:-1]), name.split('.')[-1]
    return getattr(importlib.import_module(pkg), cls_name)


class BaseVocoder:
    def spec2wav(self, mel):
        """mel [T,80] -> wav [T*hop]"""
        raise NotImplementedError

    @staticmethod
    def wav2spec(wav_fn):
        raise NotImplementedError('analysis (wav -> mel) is data preparation, outside the hot path')


def load_model(config_path, checkpoint_path, device=None):
    """vocoders/hifigan.py:17-33: `config.yaml` + a trainer checkpoint (['state_dict']['model_gen']), or the layout of the original HiFi-GAN
    release: `config.json` + `generator_v1` (['generator']).  Strict load in the weight-norm layout, then the fold."""
    from .hifigan import HifiGanGenerator
    device = device or torch.device('cuda')
    ckpt = torch.load(checkpoint_path, map_location='cpu')
    if '.yaml' in config_path:
        config = set_hparams(config_path, global_hparams=False, print_hparams=False)
        state = ckpt['state_dict']['model_gen']
    elif '.json' in config_path:
        import json
        config = json.load(open(config_path, 'r'))
        state = ckpt['generator']
        if 'audio_sample_rate' not in config and 'sampling_rate' in config:      # the release's key; only the NSF source reads it
            config['audio_sample_rate'] = config['sampling_rate']
    else:
        raise ValueError(f'{config_path}: expected config.yaml or config.json')
    config.setdefault('use_pitch_embed', False)       # absent from both config chains; the reference reads it unconditionally (hifigan.py:111)
    model = HifiGanGenerator(config)
    model.load_state_dict(state, strict=True)
    model = model.eval().to(device)
    model.remove_weight_norm()
    print(f'| Loaded model parameters from {checkpoint_path}.')
    return model, config, device


@register_vocoder
class HifiGAN(BaseVocoder):
    def __init__(self, device=None):
        base_dir = hparams['vocoder_ckpt']
        config_path = f'{base_dir}/config.yaml'
        if os.path.exists(config_path):                                   # vocoders/hifigan.py:41-47
            ckpt = latest_ckpt(base_dir)
            assert ckpt, f'no model_ckpt_steps_*.ckpt under {base_dir}'
            print('| load HifiGAN: ', ckpt)
        else:                                                             # :48-52: the original release's files
            config_path, ckpt = f'{base_dir}/config.json', f'{base_dir}/generator_v1'
            assert os.path.exists(config_path) and os.path.exists(ckpt), f'no HiFi-GAN checkpoint under {base_dir}'
            print('| load HifiGAN: ', ckpt)
        self.model, self.config, self.device = load_model(config_path, ckpt, device)

    def spec2wav(self, mel, **kwargs):
        with torch.no_grad():
            c = torch.as_tensor(np.asarray(mel), dtype=torch.float32).unsqueeze(0).transpose(2, 1).to(self.device)
            f0 = kwargs.get('f0')
            if f0 is not None and hparams.get('use_nsf'):          # vocoders/hifigan.py:60-63
                f0 = torch.as_tensor(np.asarray(f0), dtype=torch.float32)[None, :].to(self.device)
                y = self.model(c, f0, seed=int(kwargs.get('seed', hparams.get('seed', 1234)))).view(-1)
            else:
                y = self.model(c).view(-1)
        if hparams.get('vocoder_denoise_c', 0.0) > 0:
            raise NotImplementedError('vocoder_denoise_c needs the librosa spectral denoiser (out of scope)')
        return y.cpu().numpy()
