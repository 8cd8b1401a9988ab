"""CPU: register allocation of the one-workgroup-per-CU stack launches stays off the spill cliff.

residual_stack_h2_kernel (csrc/diffnet_h2.hip) lives at 256 VGPRs by design (64 accumulators + x + skip sum + weight ring); the
allocator has no slack there, and small source changes have moved it from ~15 spilled registers (outside the matrix loops: free) to
100-180 (inside the per-layer code: 158 k -> 125 k mel-frames/s in a same-box A/B).  hipcc cross-compiles without a GPU, so the cliff is
checked at build time."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spills(src, kernel):
    out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function',
                          '-fno-gpu-flush-denormals-to-zero', '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                         capture_output=True, text=True, timeout=900).stderr
    res, name = {}, None
    for line in out.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
        m = re.search(r'VGPRs Spill: (\d+)', line)
        if m and name and kernel in name:
            res[name] = int(m.group(1))
    return res


def test_h2_stack_kernel_register_spills():
    res = _spills(os.path.join(ROOT, 'bisinger_amd', 'csrc', 'diffnet_h2.hip'), 'residual_stack_h2_kernel')
    print(res)
    assert len(res) >= 4, res      # <FAIR = true> x <TAIL> x <NCT>
    assert max(res.values()) <= 32, f'residual_stack_h2_kernel spills {res}: the register allocation fell off the cliff (see the module docstring)'


def test_q_stack_kernel_register_spills():
    """residual_stack_q_kernel (csrc/diffnet_h2q.hip): x lives in the conv image, not in registers, so that NOTHING is spilled inside
    the layer loop (5-6 registers at phase boundaries on 64-frame tiles, none on 32-frame tiles).  Holding 16 more registers across the
    GEMM loops (the edge tiles' tap terms, loaded early) put 29-144 spilled registers back and cost 3-6 % (profiles/r05_q_layouts.txt)."""
    res = _spills(os.path.join(ROOT, 'bisinger_amd', 'csrc', 'diffnet_h2q.hip'), 'residual_stack_q_kernel')
    # the product instantiations: <FAIRB, TAIL, NCT, DIAG = 0, NS = 2>; DIAG != 0 are timing-only
    prod = {k: v for k, v in res.items() if re.search(r'Li[12]ELi0ELi2EEE', k)}
    print(prod)
    assert len(prod) >= 8, res
    wide = [v for k, v in prod.items() if 'Li2ELi0ELi2EEE' in k]
    narrow = [v for k, v in prod.items() if 'Li1ELi0ELi2EEE' in k]
    assert max(wide) <= 12, f'64-frame tiles spill {prod}'
    assert max(narrow) == 0, f'32-frame tiles spill {prod}'
