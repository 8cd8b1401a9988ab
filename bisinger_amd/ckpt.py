"""Checkpoint loading — the reference's ``utils.load_ckpt`` surface (utils/__init__.py:179-210).

Files: ``<dir>/model_ckpt_steps_<N>.ckpt`` = ``torch.save`` dict whose ``['state_dict']`` keys are
``<prefix>.<module key>``; the highest-step file wins; ``strict=False`` drops shape-mismatched entries.
"""
import glob
import os
import re

import torch


def latest_ckpt(base_dir):
    paths = glob.glob(f'{base_dir}/model_ckpt_steps_*.ckpt')
    step = lambda p: int(re.findall(r'model_ckpt_steps_(\d+)\.ckpt$', p)[0])
    return sorted(paths, key=step)[-1] if paths else None


def load_ckpt(cur_model, ckpt_base_dir, prefix_in_ckpt='model', force=True, strict=True):
    if os.path.isfile(ckpt_base_dir):
        base_dir, path = os.path.dirname(ckpt_base_dir), ckpt_base_dir
    else:
        base_dir, path = ckpt_base_dir, latest_ckpt(ckpt_base_dir)
    if path is None:
        msg = f'| ckpt not found in {base_dir}.'
        assert not force, msg
        print(msg)
        return None
    sd = torch.load(path, map_location='cpu')['state_dict']
    pre = prefix_in_ckpt + '.'
    sd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    if not strict:
        have = cur_model.state_dict()
        for k in [k for k, v in sd.items() if k in have and have[k].shape != v.shape]:
            print('| Unmatched keys: ', k, tuple(have[k].shape), tuple(sd[k].shape))
            del sd[k]
    cur_model.load_state_dict(sd, strict=strict)
    print(f"| load '{prefix_in_ckpt}' from '{path}'.")
    return path
