"""Global ``hparams`` dict + ``set_hparams`` — the reference's configuration surface
(/root/reference/train_bisinger/utils/hparams.py:24-123), re-implemented.

Kept semantics: argparse flags ``--config --exp_name --hparams --infer --validate --reset --debug``
when ``config == ''``; recursive ``base_config`` YAML inheritance (depth first, later files override
earlier ones, nested dicts merged, ``./``-relative paths resolved against the including file);
``checkpoints/<exp_name>/config.yaml`` overrides the chain unless ``--reset``; ``k=v,k2=v2`` overrides
are cast with the type of the existing key (``True``/``False`` strings and bool keys are evaluated);
``global_hparams=False`` returns the dict without touching the global one.
One deliberate difference: modules in this package read ``hparams`` at construction/call time only
(no import-time default capture, cf. shallow_diffusion_tts.py:44,73), so ``set_hparams`` may run after
the imports.
"""
import argparse
import os

import yaml

hparams = {}
_printed = False


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_chain(path, seen, chain):
    with open(path) as f:
        cur = yaml.safe_load(f) or {}
    seen.add(path)
    out = {}
    bases = cur.get('base_config', [])
    if not isinstance(bases, (list, tuple)):
        bases = [bases]
    cur['base_config'] = list(bases) if bases else cur.get('base_config', bases)
    for b in bases:
        if b in seen:
            continue
        if b.startswith('.'):
            b = os.path.normpath(os.path.join(os.path.dirname(path), b))
        _merge(out, _load_chain(b, seen, chain))
    if not bases:
        cur.pop('base_config', None)
    _merge(out, cur)
    chain.append(path)
    return out


def _cast(old, text):
    if text in ('True', 'False') or isinstance(old, bool):
        return eval(text)  # same permissiveness as the reference (:92-93)
    return type(old)(text)


def set_hparams(config='', exp_name='', hparams_str='', print_hparams=True, global_hparams=True):
    global _printed
    if config == '':
        ap = argparse.ArgumentParser(description='bisinger_amd')
        ap.add_argument('--config', type=str, default='')
        ap.add_argument('--exp_name', type=str, default='')
        ap.add_argument('--hparams', type=str, default='')
        for flag in ('infer', 'validate', 'reset', 'debug'):
            ap.add_argument('--' + flag, action='store_true')
        a, _ = ap.parse_known_args()
        config, exp_name, hparams_str = a.config, a.exp_name, a.hparams
        flags = {f: getattr(a, f) for f in ('infer', 'validate', 'reset', 'debug')}
    else:
        flags = dict(infer=False, validate=False, reset=False, debug=False)
    work_dir = f'checkpoints/{exp_name}' if exp_name != '' else ''
    assert config != '' or work_dir != '', 'need --config or --exp_name'

    saved = {}
    saved_path = f'{work_dir}/config.yaml'
    if work_dir != 'checkpoints/':
        if os.path.exists(saved_path):
            try:
                with open(saved_path) as f:
                    saved.update(yaml.safe_load(f) or {})
            except Exception:
                pass
        if config == '':
            config = saved_path

    chain = []
    hp = {}
    hp.update(_load_chain(config, set(), chain))
    if not flags['reset']:
        hp.update(saved)
    hp['work_dir'] = work_dir

    if hparams_str != '':
        for item in hparams_str.split(','):
            k, v = item.split('=')
            hp[k] = _cast(hp[k], v)

    if work_dir != '' and (not os.path.exists(saved_path) or flags['reset']) and not flags['infer']:
        os.makedirs(hp['work_dir'], exist_ok=True)
        with open(saved_path, 'w') as f:
            yaml.safe_dump(hp, f)

    hp.update(flags)
    if global_hparams:
        hparams.clear()
        hparams.update(hp)
        if print_hparams and not _printed:
            print('| Hparams chains: ', chain)
            print('| Hparams: ' + ', '.join(f'{k}: {v}' for k, v in sorted(hp.items())))
            _printed = True
    if hparams.get('exp_name') is None:
        hparams['exp_name'] = exp_name
    if hp.get('exp_name') is None:
        hp['exp_name'] = exp_name
    return hp
