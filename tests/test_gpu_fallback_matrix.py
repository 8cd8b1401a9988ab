"""GPU: the FALLBACK forms stay green under the driver's own test run.

The defaults multiply on the 16-bit matrix pipe (split-fp16 stack / part launches, pre-split GEMMs).  What a handle or a process runs after
a range event or a hand-off give-up — every product on the fp32 matrix pipe, one workgroup per tile — is selected by switches that are read
once per process, so a representative slice of the parity suite is repeated in CHILD processes with those switches set:
  fp32_pipe   BSG_H2=0 BSG_GEMM_SPLIT=0 BSG_H2_PART=0   (healed handle + healed GEMMs: Winograd kernels, gemm_fast_kernel, fp32 flash attention)
  split_gemm  BSG_GEMM_H2W=0                             (round 3's gemm_split_kernel: what FS2 runs when a weight cannot be pre-split)
  split_attn  BSG_FLASH_PLANES=0 BSG_ESM_H2W=0           (K / V split while staged, thread-per-query ESM attention: the forms short sequences and
                                                          batches of more than 64 utterances still run; FS2 files only)
  qkv_split   BSG_QKV_FUSED=0 BSG_FLASH_KS=1 BSG_H2W_TINY=0 BSG_H2W_RING=4 BSG_H2W_DEEP=0 BSG_H2W_DIRECT=0
                                                         (planes attention fed by qkv_split_kernel from an fp32 QKV tensor, no key split; pre-split
                                                          GEMM without 32-row tiles, deep rings / slices and the direct-store epilogue: the first
                                                          forms of round 4; FS2 + mel-generation files only)
One pytest child per setting, one after the other (the GPU box allows few processes on the card at once)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ['tests/test_gpu_diffnet.py', 'tests/test_gpu_fs2.py', 'tests/test_gpu_melgen.py', 'tests/test_gpu_f4.py', 'tests/test_gpu_sampler.py',
         'tests/test_gpu_configs.py']
# (-k also matches module names: everything in the first four files, the trajectory goldens of the sampler, configs[0] at full size)
KEYS = 'test_gpu_diffnet or test_gpu_fs2 or test_gpu_melgen or test_gpu_f4 or trajectory or config0'


@pytest.mark.parametrize('name,env', [('fp32_pipe', {'BSG_H2': '0', 'BSG_GEMM_SPLIT': '0', 'BSG_H2_PART': '0'}), ('split_gemm', {'BSG_GEMM_H2W': '0'}),
                                      ('split_attn', {'BSG_FLASH_PLANES': '0', 'BSG_ESM_H2W': '0'}),
                                      ('qkv_split', {'BSG_QKV_FUSED': '0', 'BSG_FLASH_KS': '1', 'BSG_H2W_TINY': '0', 'BSG_H2W_RING': '4', 'BSG_H2W_DEEP': '0',
                                                     'BSG_H2W_DIRECT': '0'})])
def test_parity_slice_under_fallback_switches(name, env):
    files = FILES if name not in ('split_attn', 'qkv_split') else ['tests/test_gpu_fs2.py', 'tests/test_gpu_melgen.py', 'tests/test_gpu_f4.py']
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'] + files + (['-k', KEYS] if name not in ('split_attn', 'qkv_split') else [])
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    tail = (p.stdout or '')[-1500:]
    print(f'[{name}] {tail.strip().splitlines()[-1] if tail.strip() else ""}')
    assert p.returncode == 0, tail + (p.stderr or '')[-1500:]
    assert ' passed' in tail and 'failed' not in tail.splitlines()[-1]
