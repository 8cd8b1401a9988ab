"""Import recipe for the reference.  BUILD-CONTAINER ONLY: needs
/root/reference, which does not exist on the GPU box.  Used by tools/make_golden.py."""
import importlib
import os
import sys
import types

REF = '/root/reference/train_bisinger'


def import_reference(hparams_str='timesteps=100,K_step=100,max_beta=0.06,pndm_speedup=0'):
    sys.dont_write_bytecode = True                      # the mount is read-only
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)                                       # relative YAML paths
    for n in ('librosa', 'pycwt'):                      # imported at module top, unused on the path
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules['pycwt'].wavelet = types.SimpleNamespace()
    from utils.hparams import hparams, set_hparams
    set_hparams(config='usr/configs/lang-esm-style-ori-shift/diff.yaml', print_hparams=False,
                hparams_str=hparams_str)                # BEFORE importing the diffusion module
    from utils.text_encoder import TokenTextEncoder
    from usr.diff.net import DiffNet
    import usr.diff.shallow_diffusion_tts as sdt
    sdt.tqdm = lambda it, **kw: it
    enc = TokenTextEncoder(None, vocab_list=['<AP>', '<SP>'] + [f'p{i}' for i in range(60)], replace_oov=',')
    return dict(hparams=hparams, set_hparams=set_hparams, DiffNet=DiffNet, sdt=sdt,
                GaussianDiffusion=sdt.GaussianDiffusion, phone_encoder=enc)


def import_hifigan():
    import scipy.signal
    import scipy.signal.windows
    if not hasattr(scipy.signal, 'kaiser'):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    import modules
    import modules.parallel_wavegan
    pkg = types.ModuleType('modules.parallel_wavegan.layers')
    pkg.__path__ = [REF + '/modules/parallel_wavegan/layers']
    sys.modules['modules.parallel_wavegan.layers'] = pkg
    for sub in ('causal_conv', 'pqmf', 'residual_block', 'upsample', 'residual_stack'):
        m = importlib.import_module('modules.parallel_wavegan.layers.' + sub)
        for k, v in vars(m).items():
            if not k.startswith('_'):
                setattr(pkg, k, v)
    from modules.hifigan.hifigan import HifiGanGenerator
    from utils.hparams import set_hparams
    cfg = set_hparams(config='configs/tts/hifigan.yaml', print_hparams=False, global_hparams=False)
    cfg['use_pitch_embed'] = False
    return HifiGanGenerator, cfg
