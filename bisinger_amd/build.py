"""Build libbisinger_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m bisinger_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
The .so lands in bisinger_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libbisinger_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-fgpu-flush-denormals-to-zero' if os.environ.get('BSG_FTZ') else '-fno-gpu-flush-denormals-to-zero']


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join('..', '..', 'include', 'bisinger_hip.h')]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, 'build.sha256')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs = []
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f'hipcc failed on {src}')
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
