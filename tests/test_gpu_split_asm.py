"""GPU: the 3-instruction hi + lo fp16 split (v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16; csrc/diffnet_h2_shared.h split2, csrc/hifigan.hip
split4_scaled) is the plain split `hi = (f16)x; lo = (f16)(x - (float)hi)` BIT FOR BIT.

Every split-fp16 image of the residual stack, the step tail and the 8- / 16-channel HiFi-GAN ResBlocks is written through it, so the parity
tests cover it end to end; this test pins the instruction sequence itself on 8.4 M values of every binade the images see (fp16 subnormal lo
terms, the range guard's edge, values that overflow fp16, signed zeros): tools/split_asm_check.hip, built here with hipcc and run on the card."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asm_split_equals_plain_split(tmp_path):
    exe = str(tmp_path / 'split_asm_check')
    b = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fno-gpu-flush-denormals-to-zero', '-Wno-unused-value',
                        os.path.join(ROOT, 'tools', 'split_asm_check.hip'), '-o', exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout.strip())
    assert r.returncode == 0 and ' 0 mismatching dwords' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
